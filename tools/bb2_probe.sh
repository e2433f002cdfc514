#!/bin/bash
# conv_bb2 phase ablations (wrong results) on one box, SCRFD-10G steady state (three blocks per run)
export FID_PLAN_RO=$PWD/plans/mi355x.plan
for e in "FID_BB_V=2" "FID_BB_ABLATE=3" "FID_BB_ABLATE=15" "FID_BB_ABLATE=31" "FID_BB_ABLATE=47" "FID_BB_ABLATE=63" "FID_BB_ABLATE=127" "FID_BB_ABLATE=16" "FID_BB_ABLATE=32" "FID_BB_ABLATE=48" "FID_BB_ABLATE=64" "FID_BB_V=2"; do
  echo "[$e] $(env $e python3 tools/run_r50_steady.py scrfd_10g 64 60 2>/dev/null | tail -1)"
done
