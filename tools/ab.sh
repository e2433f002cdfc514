#!/bin/bash
# same-box A/B of two source variants of one file: ab.sh <file-in-csrc> <variantA> <variantB> <cmd...>
set -e
f=$1; A=$2; B=$3; shift 3
cd $GRAFT_REPO_ROOT/scrfd_arcface_facerecognition_amd/csrc
for rep in 1 2; do
  for v in A B; do
    src=$A; [ $v = B ] && src=$B
    cp $GRAFT_REPO_ROOT/$src $f
    if make -j16 2>&1 | grep -E "error"; then echo "BUILD FAILED for variant $v"; exit 1; fi
    echo "== variant $v rep $rep"
    (cd $GRAFT_REPO_ROOT && eval "$@")
  done
done
