#!/bin/bash
# Round-4 plan generation on the GPU box: tools/make_plan.sh (N tunings judged by the two-lane bench step) + the recogniser's batch-500 picks
# (BASELINE configs[3]; tools/run_r50_steady.py tunes them under the same cold protocol) appended to the winner.  Output: gpurun_out/plan_best.plan
N=${1:-4}
R=$GRAFT_REPO_ROOT
bash $R/tools/make_plan.sh $N
f=$R/gpurun_out/plan_best.plan
FID_PLAN=$f FID_TUNE_REPS=9 python $R/tools/run_r50_steady.py arcface_r50 500 5
sort -u $f -o $f
cut -d'|' -f1,2,4 $f | sort | uniq -c
