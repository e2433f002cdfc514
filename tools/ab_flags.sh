#!/bin/bash
# Same-box comparison of bench.py flag sets on the bench step:  tools/ab_flags.sh rounds "<flags A>" "<flags B>" ...
R="$1"; shift
for i in $(seq $R); do
  for F in "$@"; do
    ms=$(python bench.py --steps 40 --warmup 5 --cpu-frames 0 --no-roofline $F 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])") || exit 1
    echo "[$F] $ms"
  done
done
