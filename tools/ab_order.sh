#!/bin/bash
# does the position of conv_wr.o in the link (= where its kernels land in the code object) change its speed?
cd $GRAFT_REPO_ROOT/scrfd_arcface_facerecognition_amd/csrc
make -j16 > /dev/null 2>&1
ALL="ctx.hip postproc.hip align.hip conv.hip conv_direct.hip conv_chunked.hip conv_pp.hip conv_pc.hip conv_pc2.hip conv_pcr.hip conv_s2.hip stem_fused.hip net.hip match.hip comm.hip repack.hip"
for ord in first last mid; do
  case $ord in
    first) S="conv_wr.hip $ALL";;
    last) S="$ALL conv_wr.hip";;
    mid) S="ctx.hip postproc.hip align.hip conv.hip conv_direct.hip conv_wr.hip conv_chunked.hip conv_pp.hip conv_pc.hip conv_pc2.hip conv_pcr.hip conv_s2.hip stem_fused.hip net.hip match.hip comm.hip repack.hip";;
  esac
  rm -f ../libfaceid.so
  make -j16 SRCS="$S" > /dev/null 2>&1
  echo "== conv_wr.o $ord: $(cd $GRAFT_REPO_ROOT && FID_FORCE_GEN=9 FID_FORCE_NS=2 python tools/profile_ops.py arcface_r50 500 2>/dev/null | grep -E "layer3.5.conv1|layer3.5.conv2|layer2.1.conv1|layer4.0.conv1" | awk '{printf "%s %s  ", $1, $7}')"
done
rm -f ../libfaceid.so; make -j16 > /dev/null 2>&1
