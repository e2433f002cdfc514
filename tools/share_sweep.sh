#!/bin/bash
# plans tuned with FID_TUNE_SHARE = s (the CU-share weight of the autotuner's score) judged by the bench step, against a base plan on the same box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; BASE=${BASE:-$R/plans/mi355x.plan}
ms() { FID_PLAN_RO=$1 python $R/bench.py --cpu-frames 0 --no-roofline 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])"; }
for s in "$@"; do
  f=$O/share_$s.plan; rm -f $f
  FID_TUNE_SHARE=$s FID_PLAN=$f FID_TUNE_REPS=9 python $R/bench.py --steps 10 --warmup 3 --cpu-frames 0 --no-roofline > /dev/null 2>&1
  sort -u $f -o $f
  for k in 1 2 3; do echo "share $s: $(ms $f)   base: $(ms $BASE)"; done
done
