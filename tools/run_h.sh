#!/bin/bash
source tools/gpu_steps.sh
export O=gpurun_out/r6h; mkdir -p $O
step 500 golden.log python -m pytest tests/test_gpu_golden.py -q -x -m gpu -k "262144 or tail or products or pair or 16384 or 65536"
step 200 ab_tailpair.txt bash tools/ab_env.sh "FHERAM_TAIL_PAIR=0" "FHERAM_TAIL_PAIR=1" 3
step 120 ab_readme.txt bash tools/ab_env.sh "FHERAM_TAIL_PAIR=0" "FHERAM_TAIL_PAIR=1" 1
step 120 bench.json python bench.py --steps 30 --warmup 10 --no-cpu-baseline
tail -n 6 $O/golden.log; cat $O/ab_tailpair.txt
python3 -c "
import json
for f in ('bench',):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['read_ms'], d['read_prepare_write_ms'], d['write_ms'], d.get('value_incl_boundary'), d['trace_tail'])
    for k in d['roofline_by_kernel']['kernels']: print('   %-44s %8.1f us share %.3f frac %.3f alg %.3f'%(k['kernel'][:44], k['avg_launch_ms']*1e3, k['share_of_gpu_time'], k['frac'], k.get('frac_algorithmic') or 0))
"
