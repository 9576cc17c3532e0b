#!/bin/bash
source tools/gpu_steps.sh
export O=gpurun_out/r6e; mkdir -p $O
step 120 bench.json python bench.py --steps 30 --warmup 10 --no-cpu-baseline
step 120 bench_readme.json python bench.py --params readme --steps 30 --warmup 10 --no-cpu-baseline
step 200 ab_gate.txt bash tools/ab_env.sh "FHERAM_PRE_INV=1" "FHERAM_PRE_INV=2" 2
step 700 pytest_all.log python -m pytest tests -q -x -m gpu
tail -n 3 $O/pytest_all.log; cat $O/ab_gate.txt
python3 -c "
import json
for f in ('bench','bench_readme'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['read_ms'], d['read_prepare_write_ms'], d['write_ms'], d.get('value_incl_boundary'), d['roundoff_max']['value'], d['reference_published'].get('speedup_read'), d['reference_published'].get('speedup_write'))
    for k in d['roofline_by_kernel']['kernels']: print('   %-44s %8.1f us share %.3f frac %.3f alg %.3f'%(k['kernel'][:44], k['avg_launch_ms']*1e3, k['share_of_gpu_time'], k['frac'], k.get('frac_algorithmic') or 0))
"
