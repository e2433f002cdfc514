#!/bin/bash
# Same-box A/B of two environments on any command that prints one line: tools/ab_env_cmd.sh "ENV_A" "ENV_B" rounds cmd...
A="$1"; B="$2"; R="$3"; shift 3
for i in $(seq $R); do
  for v in A B; do
    if [ $v = A ]; then E="$A"; else E="$B"; fi
    echo "$v [$E] $(env $E "$@" 2>/dev/null | tail -1)"
  done
done
