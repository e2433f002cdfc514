#!/bin/bash
# Same-box A/B of two environments (e.g. a kernel switched off through its env hook) on the bench step:
#   tools/ab_env.sh "FID_STEM_VALU=1" "" [rounds]      -> ms_per_step of A and B, interleaved
A="$1"; B="$2"; R="${3:-3}"
for i in $(seq $R); do
  for v in A B; do
    if [ $v = A ]; then E="$A"; else E="$B"; fi
    ms=$(env $E python bench.py --steps 40 --warmup 5 --cpu-frames 0 --no-roofline | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])") || exit 1
    echo "$v [$E] $ms"
  done
done
