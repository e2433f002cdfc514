# conv3x3_wr timing ablations (wrong results): MASKS bits 1 no step barrier, 2 no patch pieces, 4 no weight reloads, 8 no epilogue, 32 stores dropped, 64 no residual loads
# NS = the FID_FORCE_NS code of the variant (2: pair of tiles x 128 couts, 29: the same on STRIP tiles, ...)
for m in ${MASKS:-0 8 32 64 96}; do
  echo "ablate $m: $(FID_WR_ABLATE=$m FID_FORCE_GEN=9 FID_FORCE_NS=${NS:-2} python tools/profile_ops.py arcface_r50 ${BATCH:-500} 2>/dev/null | grep -E 'layer3.5.conv1|layer3.5.conv2|layer2.1.conv1|layer4.0.conv1' | awk '{printf "%s %s  ", $1, $7}')"
done
