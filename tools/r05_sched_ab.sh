#!/bin/bash
# Round 5: schedule knobs of the bench step on one box, alternating: --rec-group G x --streams S (plan loaded; picks missing for a batch are tuned at start-up)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
for rep in 1 2; do
  for spec in ${1:-2:2 4:2 2:3 3:2}; do
    g=${spec%%:*}; s=${spec##*:}
    python3 bench.py --steps 24 --warmup 6 --repeats 3 --rec-group $g --streams $s --cpu-frames 0 --no-roofline --no-one-lane 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.read()); print('rec_group $g streams $s:', o['ms_per_step'], 'ms/step', o['value'], 'faces/s')" | tee -a $O/sched_ab.txt
  done
done
