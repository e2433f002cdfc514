#!/bin/bash
# MFMA utilisation and stall counters per kernel (rocprofv3 PMC, counters in their own passes with --kernel-trace only, the
# program directly after `--`).  Run on the GPU box:   tools/pmc_mfma.sh <tag> <script> [args...]
#   e.g. tools/pmc_mfma.sh r50_b500 tools/profile_ops.py arcface_r50 500
# Output: gpurun_out/pmc_<tag>/summary.txt (copy what should be judged into profiles/rNN/).
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
    --output-format csv -d $O/a -o p -- python3 $R/$1 "${@:2}" > $O/a.log 2>&1
echo "pass a done"
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA \
    --output-format csv -d $O/b -o p -- python3 $R/$1 "${@:2}" > $O/b.log 2>&1
echo "pass b done"
# KLOG="arch batch runs": also the executed / algorithmic MFMA work per kernel (tools/klog_map.py: op -> kernel map of that net at that batch; runs = net runs of the profiled program)
if [ -n "$KLOG" ]; then
  set -- $KLOG
  (cd $R && python3 tools/klog_map.py $1 $2 > $O/klog_map.json 2> $O/klog_map.err) || true
  python3 $R/tools/pmc_mfma.py $O $O/klog_map.json $3 > $O/summary.txt
else
  python3 $R/tools/pmc_mfma.py $O > $O/summary.txt
fi
rm -f $O/*/*counter_collection.csv $O/*/*kernel_trace.csv $O/*/*/*counter_collection.csv $O/*/*/*kernel_trace.csv
cat $O/summary.txt
