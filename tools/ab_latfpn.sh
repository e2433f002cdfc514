# two-lane bench with / without the fused lateral + fpn op, each variant with its own persisted plan (tuned once), alternating
for v in A B; do
  if [ $v = A ]; then export FID_NO_LATFPN_FUSE=1; else unset FID_NO_LATFPN_FUSE; fi
  FID_PLAN=/tmp/plan_$v.plan python bench.py --steps 10 --warmup 3 --cpu-frames 0 --no-roofline --no-one-lane > /dev/null 2>&1
done
for i in 1 2 3 4; do
  for v in A B; do
    if [ $v = A ]; then export FID_NO_LATFPN_FUSE=1; else unset FID_NO_LATFPN_FUSE; fi
    ms=$(FID_PLAN=/tmp/plan_$v.plan python bench.py --steps 40 --warmup 5 --cpu-frames 0 --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['ms_per_step_1lane'])")
    echo "$v (A = unfused, B = fused) $ms"
  done
done
