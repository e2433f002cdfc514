#!/bin/bash
# Round 5 vs the round-4 tree (tmp_ab/r04 = git archive f7f7ab1), same box, interleaved: the default bench line (each tree's own bench.py, library and plan)
# and IResNet-50 at batch 500 with each tree's plan loaded
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
n=${1:-3}; t=r04
(cd $R/tmp_ab/$t/scrfd_arcface_facerecognition_amd/csrc && make -j16 > /tmp/build_$t.log 2>&1) || { echo "build $t failed"; tail -3 /tmp/build_$t.log; exit 1; }
one() { (cd $1 && python3 bench.py --steps 40 --warmup 5 --cpu-frames 0 --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$2 bench ms_per_step', d['ms_per_step'], 'one lane', d.get('ms_per_step_1lane'), 'rec_group1', d.get('ms_per_step_rec_group1'))"); }
r50() { (cd $1 && FID_PLAN_RO=$1/plans/mi355x.plan python3 tools/run_r50_steady.py arcface_r50 500 30 2>&1 | tail -1 | sed "s/^/$2 /"); }
sc() { (cd $1 && FID_PLAN_RO=$1/plans/mi355x.plan python3 tools/run_r50_steady.py scrfd_10g 64 30 2>&1 | tail -1 | sed "s/^/$2 /"); }
for i in $(seq $n); do
  one $R/tmp_ab/$t $t; one $R r05
  r50 $R/tmp_ab/$t $t; r50 $R r05
  sc $R/tmp_ab/$t $t; sc $R r05
done | tee -a $O/ab_r04.txt
