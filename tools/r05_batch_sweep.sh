#!/bin/bash
# Round 5: IResNet-50 steady state over chunk sizes (STRIP tiles: 8 images x 14 columns = 7 exact fragments; 585 crops = 512 tiles of 14 x 16)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
python3 -m pytest tests/test_gpu_conv_families.py -x -q -k "strip" > $O/strip_tests2.log 2>&1 || { tail -30 $O/strip_tests2.log; exit 1; }
tail -1 $O/strip_tests2.log
for b in ${1:-500 512 576 584 585 592}; do
  python3 tools/run_r50_steady.py arcface_r50 $b 30 2>&1 | tail -1 | tee -a $O/batch_sweep.txt
done
