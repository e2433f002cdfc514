#!/bin/bash
# same-box comparison of several macro builds of one source file: tools/ab_build4.sh <file.hip> "<cmd>" "<EXTRA 1>" "<EXTRA 2>" ...
f=$1; cmd="$2"; shift 2
cd $GRAFT_REPO_ROOT/scrfd_arcface_facerecognition_amd/csrc
for rep in 1 2; do
  for X in "$@"; do
    rm -f build/${f%.hip}.o
    if make -j16 EXTRA="$X" 2>&1 | grep -E " error"; then echo "BUILD FAILED [$X]"; continue; fi
    echo "== [$X] rep $rep: $(cd $GRAFT_REPO_ROOT && eval "$cmd")"
  done
done
rm -f build/${f%.hip}.o; make -j16 > /dev/null 2>&1
