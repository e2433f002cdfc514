#!/bin/bash
# Round 5: the recogniser batched over G consecutive steps of a lane (bench.py --rec-group G), alternating on one box; no plan file (every variant tunes its own picks)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export FID_PLAN=
for rep in 1 2; do
  for g in ${1:-1 2 3}; do
    python3 bench.py --steps 24 --warmup 6 --repeats 3 --rec-group $g --cpu-frames 0 --no-roofline 2> $O/group_$g.err | python3 -c "
import json,sys
o=json.loads(sys.stdin.read()); print('rec_group $g:', o['ms_per_step'], 'ms/step two lanes,', o.get('ms_per_step_1lane'), 'one lane,', o['value'], 'faces/s')" | tee -a $O/group_ab.txt
  done
done
