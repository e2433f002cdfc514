#!/bin/bash
# Generate kernel plan files on the GPU box and keep the one with the best bench step: the autotuner's picks vary run to run
# (candidates within timing noise of each other alone, but not under the two-lane overlap), so several tunings are tried and
# each resulting plan is judged by the step time it gives.  Output: gpurun_out/plan_best.plan (+ gpurun_out/plan_log.txt).
# FID_TUNE_SHARE (default 0.25 here): the tuner's score weighs in the fraction of the CUs a launch occupies -- the bench keeps two batches in flight,
# and a launch that leaves CUs free lets the other lane's kernels run there (net.hip; 0.5 / 0.75 measured equal, 1.0 measured 15 % slower).
export FID_TUNE_SHARE=${FID_TUNE_SHARE:-0.25}
N=${1:-5}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -f $O/plan_*.plan $O/plan_log.txt
best=999; bestf=""
for i in $(seq $N); do
  f=$O/plan_$i.plan
  FID_PLAN=$f FID_TUNE_REPS=9 python $R/bench.py --steps 10 --warmup 3 --cpu-frames 8 > /dev/null 2>&1
  sort -u $f -o $f
  ms=$(for k in 1 2 3; do FID_PLAN=$f python $R/bench.py --steps 40 --warmup 5 --cpu-frames 0 --no-roofline 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])"; done | sort -n | sed -n 2p)
  echo "plan $i: $(wc -l < $f) picks, median ms_per_step $ms" | tee -a $O/plan_log.txt
  if python -c "import sys; sys.exit(0 if float('$ms') < float('$best') else 1)"; then best=$ms; bestf=$f; fi
done
cp $bestf $O/plan_best.plan
echo "best: $bestf ($best ms)" | tee -a $O/plan_log.txt
