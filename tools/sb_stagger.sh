for m in 0 50 100 150 200 300 0; do echo "stagger $m: $(FID_SB_STAGGER=$m python tools/profile_ops.py arcface_r50 500 2>/dev/null | grep -E 'layer1.0.conv1' | awk '{print $7}')"; done
