export FID_PLAN_RO=$PWD/plans/mi355x.plan
for i in 1 2; do
  for s in 2 3; do
    ms=$(python bench.py --steps 42 --warmup 6 --cpu-frames 0 --no-roofline --no-one-lane --streams $s 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])") || exit 1
    echo "streams $s: $ms"
  done
done
