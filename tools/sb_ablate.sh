for m in 0 1 2 3 4 8 16 31; do echo "ablate $m: $(FID_SB_ABLATE=$m python tools/profile_ops.py arcface_r50 500 2>/dev/null | grep -E 'layer1.0.conv1' | awk '{print $7}')"; done
