#!/bin/bash
# same-box comparison of source variants of one csrc file: tools/ab_files.sh <file.hip> "<cmd>" variant1.hip variant2.hip ... (paths under the repo)
f=$1; cmd="$2"; shift 2
R=$GRAFT_REPO_ROOT
cd $R/scrfd_arcface_facerecognition_amd/csrc
cp $f /tmp/orig_$f
for rep in 1 2; do
for v in "$@"; do
  cp $R/$v $f
  if make -j16 2>&1 | grep -E " error"; then echo "BUILD FAILED $v"; continue; fi
  echo "== $v rep $rep: $(cd $R && eval "$cmd")"
done
done
cp /tmp/orig_$f $f; make -j16 > /dev/null 2>&1
