#!/bin/bash
# same-box A/B of two LIBRARY builds under the current python tree and plan: tools/r05_ab_lib.sh <tag under tmp_ab/> [rounds]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
t=$1; n=${2:-2}
(cd $R/tmp_ab/$t/scrfd_arcface_facerecognition_amd/csrc && make -j16 > /tmp/build_$t.log 2>&1) || { echo "build $t failed"; tail -3 /tmp/build_$t.log; exit 1; }
export FID_PLAN_RO=$R/plans/mi355x.plan
cd $R
meas() {
  for spec in "arcface_r50 500" "arcface_r50 128" "scrfd_10g 64"; do
    set -- $spec
    FID_LIB_DIR=$L python3 tools/run_r50_steady.py $1 $2 30 2>&1 | tail -1 | sed "s/^/$T /"
  done
  python3 bench.py --steps 40 --warmup 6 --cpu-frames 0 --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$T bench ms_per_step', d['ms_per_step'], 'one lane', d.get('ms_per_step_1lane'))"
}
cp scrfd_arcface_facerecognition_amd/libfaceid.so /tmp/libfaceid.work.so
for i in $(seq $n); do
  cp tmp_ab/$t/scrfd_arcface_facerecognition_amd/libfaceid.so scrfd_arcface_facerecognition_amd/libfaceid.so; T=$t; meas
  cp /tmp/libfaceid.work.so scrfd_arcface_facerecognition_amd/libfaceid.so; T=work; meas
done | tee -a $O/ab_lib_$t.txt
