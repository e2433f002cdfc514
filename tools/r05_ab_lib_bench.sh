#!/bin/bash
# same-box A/B of two LIBRARY builds on the bench step only: tools/r05_ab_lib_bench.sh <tag under tmp_ab/> [rounds]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
t=$1; n=${2:-4}
(cd $R/tmp_ab/$t/scrfd_arcface_facerecognition_amd/csrc && make -j16 > /tmp/build_$t.log 2>&1) || { echo "build $t failed"; tail -3 /tmp/build_$t.log; exit 1; }
export FID_PLAN_RO=$R/plans/mi355x.plan
cd $R
cp scrfd_arcface_facerecognition_amd/libfaceid.so /tmp/libfaceid.work.so
for i in $(seq $n); do
  for T in $t work; do
    if [ $T = work ]; then cp /tmp/libfaceid.work.so scrfd_arcface_facerecognition_amd/libfaceid.so; else cp tmp_ab/$t/scrfd_arcface_facerecognition_amd/libfaceid.so scrfd_arcface_facerecognition_amd/libfaceid.so; fi
    python3 bench.py --steps 40 --warmup 6 --cpu-frames 0 --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$T bench ms_per_step', d['ms_per_step'], 'one lane', d.get('ms_per_step_1lane'), 'rg1', d.get('ms_per_step_rec_group1'))"
  done
done | tee -a $O/ab_lib_bench_$t.txt
cp /tmp/libfaceid.work.so scrfd_arcface_facerecognition_amd/libfaceid.so
