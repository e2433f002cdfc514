#!/bin/bash
# same-box comparison of whole-library builds from source snapshots under tmp_ab/<tag>/ (made with git archive):
#   tools/ab_commits.sh "<cmd>" tag1 tag2 ...   -- the current tree is measured last as "work"
cmd="$1"; shift
R=$GRAFT_REPO_ROOT
cp $R/scrfd_arcface_facerecognition_amd/libfaceid.so /tmp/libfaceid.work.so
for t in "$@"; do
  (cd $R/tmp_ab/$t/scrfd_arcface_facerecognition_amd/csrc && make -j16 > /tmp/build_$t.log 2>&1) || { echo "build $t failed"; tail -3 /tmp/build_$t.log; continue; }
  cp $R/tmp_ab/$t/scrfd_arcface_facerecognition_amd/libfaceid.so $R/scrfd_arcface_facerecognition_amd/libfaceid.so
  echo "== $t: $(cd $R && eval "$cmd")"
done
cp /tmp/libfaceid.work.so $R/scrfd_arcface_facerecognition_amd/libfaceid.so
echo "== work: $(cd $R && eval "$cmd")"
