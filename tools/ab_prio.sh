#!/bin/bash
# Same-box A/B of HIP stream priorities for the two lanes of the bench step (FID_BENCH_PRIO, docs/HOOKS.md):
#   tools/ab_prio.sh [rounds]      -> ms_per_step with equal priorities (A) and lane 0 high / lane 1 default (B), interleaved
export FID_PLAN_RO=$PWD/plans/mi355x.plan
for i in $(seq ${1:-3}); do
  for p in "" "-1,0"; do
    ms=$(FID_BENCH_PRIO="$p" python bench.py --steps 40 --warmup 5 --cpu-frames 0 --no-roofline --no-one-lane 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])") || exit 1
    echo "prio [$p] $ms"
  done
done
