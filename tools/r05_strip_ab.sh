#!/bin/bash
# Round 5: STRIP tiles A/B on one box -- parity tests of the strip variants, then IResNet-50 at batch 500 (and other nets) tuned fresh with and without them.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python3 -m pytest tests/test_gpu_conv_families.py -x -q -k "${1:-98 or strip}" > $O/strip_tests.log 2>&1 || { tail -30 $O/strip_tests.log; exit 1; }
tail -3 $O/strip_tests.log
for spec in ${2:-"arcface_r50:500"}; do
  a=${spec%%:*}; b=${spec##*:}
  for v in strip nostrip strip nostrip; do
    if [ $v = nostrip ]; then export FID_NO_STRIP=1; else unset FID_NO_STRIP; fi
    python3 tools/run_r50_steady.py $a $b 30 2>&1 | tail -1 | sed "s/^/$v /" | tee -a $O/strip_ab.txt
  done
  unset FID_NO_STRIP
  FID_TUNE_LOG=2 python3 tools/profile_ops.py $a $b > $O/ops_${a}_b${b}_strip.txt 2>&1
  grep -c "ns 7" $O/ops_${a}_b${b}_strip.txt
done
