#!/bin/bash
# A/B of two builds of the library on the bench workload, alternating (same box, same clocks): tools/bench_ab.sh libA.so libB.so [rounds] [extra bench args]
A=$1; B=$2; R=${3:-3}; shift 3 2>/dev/null
for i in $(seq 1 $R); do
  for L in $A $B; do
    FHERAM_LIB=$L python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-boundary --no-readme-leg "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s %8.1f ops/s  step %.4f ms  read %.4f rpw %.4f write %.4f  trace step %.2f us (frac %.3f)' % ('$L'.split('/')[-1], d['value'], d['ms_per_step'], d['read_ms'], d['read_prepare_write_ms'], d['write_ms'], d['roofline']['avg_launch_ms']*1e3, d['roofline']['frac']))"
  done
done
