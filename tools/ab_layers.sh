#!/bin/bash
# same-box A/B of kernel families on chosen layers: tools/ab_layers.sh "<env A>" "<env B>" <arch> <batch> "<layer regex>" [rounds]
A="$1"; B="$2"; arch=$3; batch=$4; rx="$5"; R="${6:-3}"
for i in $(seq $R); do
  for v in A B; do
    if [ $v = A ]; then E="$A"; else E="$B"; fi
    echo "$v [$E] $(env $E python tools/profile_ops.py $arch $batch 2>/dev/null | grep -E "$rx" | awk '{printf "%s %s  ", $1, $7}')"
  done
done
