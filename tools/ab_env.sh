#!/bin/bash
# usage: ab_env.sh "ENV1=.. ENV2=.." "ENVB=.." rounds
A="$1"; B="$2"; R=${3:-3}
for i in $(seq 1 $R); do
  for E in "$A" "$B"; do
    env $E python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-boundary --no-readme-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s %8.1f ops/s  step %.4f ms  read %.4f rpw %.4f write %.4f' % ('$E', d['value'], d['ms_per_step'], d['read_ms'], d['read_prepare_write_ms'], d['write_ms']))"
  done
done
