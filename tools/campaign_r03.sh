#!/bin/bash
# Round-3 measurement campaign on the GPU box (run through gpurun in two parts; everything lands under gpurun_out/r03/).
#   tools/campaign_r03.sh a : default bench line, rocprofv3 kernel stats (two lanes / one lane), HBM traffic PMC passes
#   tools/campaign_r03.sh b : MFMA-utilisation PMC passes of the single-lane bench, per-op profiles, steady-state numbers, side configurations
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
part=${1:-a}
if [ "$part" = a ]; then
  python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
  tail -c 600 $O/bench_default.json
  bash $R/tools/collect_profiles.sh r03 > $O/collect.log 2>&1
  P=$R/gpurun_out/prof_r03
  cp $P/lanes2/*kernel_stats.csv $O/bench_lanes2_kernel_stats.csv 2>/dev/null || cp $(find $P/lanes2 -name "*kernel_stats.csv" | head -1) $O/bench_lanes2_kernel_stats.csv
  cp $(find $P/lanes1 -name "*kernel_stats.csv" | head -1) $O/bench_lanes1_kernel_stats.csv
  cp $P/pmc_traffic.json $O/pmc_traffic.json
  cat $O/pmc_traffic.json
else
  bash $R/tools/pmc_mfma.sh bench_r03 bench.py --steps 4 --warmup 1 --streams 1 --cpu-frames 0 --no-roofline --repeats 1 > $O/pmc_mfma.log 2>&1
  cp $R/gpurun_out/pmc_bench_r03/summary.txt $O/pmc_mfma_bench_lanes1.txt
  export FID_PLAN_RO=$R/plans/mi355x.plan
  for spec in "scrfd_10g 64" "arcface_r50 64" "arcface_r50 500"; do
    set -- $spec
    FID_TUNE_LOG=1 python3 $R/tools/profile_ops.py $1 $2 > $O/ops_$1_b$2.txt 2>&1
    python3 $R/tools/run_r50_steady.py $1 $2 30 > $O/steady_$1_b$2.txt 2>&1
    cat $O/steady_$1_b$2.txt
  done
  python3 $R/tools/run_stem.py 64 40 > $O/stem_steady.txt 2>&1; cat $O/stem_steady.txt
  python3 $R/tools/bench_configs.py > $O/bench_configs.json 2> $O/bench_configs.err; cat $O/bench_configs.json
  python3 $R/tools/bench_match.py > $O/bench_match.txt 2>&1
fi
