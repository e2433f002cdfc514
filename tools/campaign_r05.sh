#!/bin/bash
# Round-5 measurement campaign on the GPU box (run through gpurun in parts; everything lands under gpurun_out/r05/).
#   tools/campaign_r05.sh a : default bench line, rocprofv3 kernel stats (two lanes / one lane), HBM traffic PMC passes of the bench step
#   tools/campaign_r05.sh b : per-op profiles, steady-state numbers, side configurations, gallery match
#   tools/campaign_r05.sh c : IResNet-50 at batch 500 / 585 and SCRFD-10G at 64 with the plan loaded: MFMA-busy PMC passes with executed / algorithmic work per kernel + HBM traffic at 500
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
part=${1:-a}
export FID_PLAN_RO=$R/plans/mi355x.plan
if [ "$part" = a ]; then
  python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
  tail -c 600 $O/bench_default.json
  bash $R/tools/collect_profiles.sh r05 > $O/collect.log 2>&1
  P=$R/gpurun_out/prof_r05
  cp $(find $P/lanes2 -name "*kernel_stats.csv" | head -1) $O/bench_lanes2_kernel_stats.csv
  cp $(find $P/lanes1 -name "*kernel_stats.csv" | head -1) $O/bench_lanes1_kernel_stats.csv
  cp $P/pmc_traffic.json $O/pmc_traffic.json
  cat $O/pmc_traffic.json
elif [ "$part" = b ]; then
  for spec in "scrfd_10g 64" "arcface_r50 64" "arcface_r50 128" "arcface_r50 500" "arcface_r50 585"; do
    set -- $spec
    FID_TUNE_LOG=1 python3 $R/tools/profile_ops.py $1 $2 > $O/ops_$1_b$2.txt 2>&1
    python3 $R/tools/run_r50_steady.py $1 $2 30 2>&1 | tail -1 | tee -a $O/steady.txt
  done
  python3 $R/tools/run_stem.py 64 40 2>&1 | tail -1 | tee -a $O/steady.txt
  python3 $R/tools/bench_configs.py > $O/bench_configs.json 2> $O/bench_configs.err; cat $O/bench_configs.json
  python3 $R/tools/bench_match.py > $O/bench_match.txt 2>&1
else
  KLOG="arcface_r50 500 11" bash $R/tools/pmc_mfma.sh r50_b500_r05 tools/run_r50_steady.py arcface_r50 500 10 > $O/pmc_mfma_r50_b500.log 2>&1
  cp $R/gpurun_out/pmc_r50_b500_r05/summary.txt $O/pmc_mfma_r50_b500.txt
  KLOG="arcface_r50 585 11" bash $R/tools/pmc_mfma.sh r50_b585_r05 tools/run_r50_steady.py arcface_r50 585 10 > $O/pmc_mfma_r50_b585.log 2>&1
  cp $R/gpurun_out/pmc_r50_b585_r05/summary.txt $O/pmc_mfma_r50_b585.txt
  KLOG="arcface_r50 128 11" bash $R/tools/pmc_mfma.sh r50_b128_r05 tools/run_r50_steady.py arcface_r50 128 10 > $O/pmc_mfma_r50_b128.log 2>&1
  cp $R/gpurun_out/pmc_r50_b128_r05/summary.txt $O/pmc_mfma_r50_b128.txt
  KLOG="scrfd_10g 64 11" bash $R/tools/pmc_mfma.sh scrfd_b64_r05 tools/run_r50_steady.py scrfd_10g 64 10 > $O/pmc_mfma_scrfd_b64.log 2>&1
  cp $R/gpurun_out/pmc_scrfd_b64_r05/summary.txt $O/pmc_mfma_scrfd_b64.txt
  T=$R/gpurun_out/prof_r05_b500
  mkdir -p $T
  cd /tmp && export TMPDIR=/tmp
  for c in f:FETCH_SIZE w:WRITE_SIZE; do
    for n in 2 6; do
      rocprofv3 --kernel-trace --pmc ${c#*:} --output-format csv -d $T/${c%%:*}$n -o p -- python3 $R/tools/run_r50_steady.py arcface_r50 500 $n > $T/${c%%:*}$n.log 2>&1
      echo "pmc ${c#*:} $n done"
    done
  done
  python3 $R/tools/pmc_traffic.py $T "python3 tools/run_r50_steady.py arcface_r50 500 {2,6} (a step = one run of the net on 500 crops, plan loaded)" > $O/pmc_traffic_r50_b500.json
  rm -f $T/*/*counter_collection.csv $T/*/*kernel_trace.csv $T/*/*/*counter_collection.csv $T/*/*/*kernel_trace.csv
  cat $O/pmc_traffic_r50_b500.json; tail -30 $O/pmc_mfma_r50_b500.txt
fi
