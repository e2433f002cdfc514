#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace stats of the default bench and of a single-lane bench, then the four PMC
# passes tools/pmc_traffic.py needs.  Output under gpurun_out/prof_$1/.
set -e
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lanes2 -o k -- python3 $R/bench.py --steps 60 --warmup 3 --cpu-frames 0 --no-roofline --repeats 1 --no-one-lane > $O/lanes2.log 2>&1
echo "lanes2 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lanes1 -o k -- python3 $R/bench.py --steps 60 --warmup 3 --cpu-frames 0 --no-roofline --repeats 1 --streams 1 --no-one-lane > $O/lanes1.log 2>&1
echo "lanes1 done"
for c in f:FETCH_SIZE w:WRITE_SIZE; do
  for n in 2 6; do
    rocprofv3 --kernel-trace --pmc ${c#*:} --output-format csv -d $O/${c%%:*}$n -o p -- python3 $R/bench.py --steps $n --warmup 1 --cpu-frames 0 --no-roofline --repeats 1 --streams 1 --no-one-lane > $O/${c%%:*}$n.log 2>&1
    echo "pmc ${c#*:} $n done"
  done
done
python3 $R/tools/pmc_traffic.py $O > $O/pmc_traffic.json
rm -f $O/*/*counter_collection.csv $O/*/*kernel_trace.csv   # large; the summaries stay
cat $O/pmc_traffic.json
