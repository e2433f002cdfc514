#!/bin/bash
# Round-5 plan generation on the GPU box: tools/make_plan.sh (N tunings judged by the two-lane bench step; bench.py now runs the recogniser once per two
# steps of a lane: IResNet-50 picks at 128 crops) + the recogniser's picks at 64 (the rec_group-1 side leg and tests), 500 and 585 crops (BASELINE configs[3])
N=${1:-4}
R=$GRAFT_REPO_ROOT
bash $R/tools/make_plan.sh $N
f=$R/gpurun_out/plan_best.plan
for b in 64 500 585; do
  FID_PLAN=$f FID_TUNE_REPS=9 python $R/tools/run_r50_steady.py arcface_r50 $b 5
done
sort -u $f -o $f
cut -d'|' -f1,2,4 $f | sort | uniq -c
