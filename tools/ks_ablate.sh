#!/bin/bash
# phase ablations of conv3x3_ks on IResNet-50 at 64 faces (FID_KS_ABLATE: 1 no step barrier, 2 no patch pieces, 4 no weight reloads, 8 no epilogue, 16 no matrix work; wrong results), ms per run (27 + 4 launches of the kernel per run)
export FID_PLAN_RO=$PWD/plans/mi355x.plan
for e in 0 2 4 8 16 6 14 30 31 0; do
  echo "[$e] $(FID_KS_ABLATE=$e python3 tools/run_r50_steady.py arcface_r50 64 100 2>/dev/null | tail -1)"
done
