for m in ${MASKS:-0 8 32 64 96}; do
  echo "ablate $m: $(FID_WR_ABLATE=$m FID_FORCE_GEN=9 python tools/profile_ops.py arcface_r50 ${BATCH:-500} 2>/dev/null | grep -E 'layer3.5.conv1|layer3.5.conv2|layer2.1.conv1|layer4.0.conv1' | awk '{printf "%s %s  ", $1, $7}')"
done
