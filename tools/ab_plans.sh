# same-box A/B of two plan files on the bench step: tools/ab_plans.sh planA planB [rounds]
A=$1; B=$2; R=${3:-4}
for i in $(seq $R); do for v in A B; do
  f=$A; [ $v = B ] && f=$B
  echo "$v $(FID_PLAN_RO=$f python bench.py --steps 40 --warmup 5 --cpu-frames 0 --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['ms_per_step_1lane'], d['config']['ops_autotuned_at_startup'])")"
done; done
