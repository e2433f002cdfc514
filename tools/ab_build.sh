#!/bin/bash
# same-box A/B of two BUILDS of one source file (macro variants): tools/ab_build.sh <file.hip> "<EXTRA A>" "<EXTRA B>" <cmd...>
set -e
f=$1; A="$2"; B="$3"; shift 3
cd $GRAFT_REPO_ROOT/scrfd_arcface_facerecognition_amd/csrc
for rep in 1 2; do
  for v in A B; do
    X="$A"; [ $v = B ] && X="$B"
    rm -f build/${f%.hip}.o
    if make -j16 EXTRA="$X" 2>&1 | grep -E " error"; then echo "BUILD FAILED for variant $v"; exit 1; fi
    echo "== variant $v [$X] rep $rep"
    (cd $GRAFT_REPO_ROOT && eval "$@")
  done
done
rm -f build/${f%.hip}.o; make -j16 > /dev/null 2>&1
