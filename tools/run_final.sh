#!/bin/bash
# Final validation of the round on one box: the whole GPU suite, the default bench line (with the CPU legs and parity_in_run), the soaks, a long search.
source tools/gpu_steps.sh
export O=gpurun_out/r6z; mkdir -p $O
step 900 pytest_all.log python -m pytest tests -q -x -m gpu
step 300 bench_default.json python bench.py
step 200 soak_ops.txt python tests/soak_gpu.py 120
step 300 soak_flow.txt python tests/soak_flow_gpu.py 200
step 100 soak_tail.txt python tests/soak_tail_gpu.py 60
step 100 soak_tail2.txt python tests/soak_tail_gpu.py 40 20261007 9 64
tail -n 3 $O/pytest_all.log; tail -n 2 $O/soak_ops.txt $O/soak_flow.txt $O/soak_tail.txt $O/soak_tail2.txt
python3 -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('default bench', d['value'], d['ms_per_step'], d['read_ms'], d['read_prepare_write_ms'], d['write_ms'], d.get('value_incl_boundary'), d['roundoff_max']['value'], d['parity_in_run']['read'], d['parity_in_run']['rpw'], d['parity_in_run']['rows_after_write'], d['reference_published']['speedup_read'], d['reference_published']['speedup_write'], d['roofline']['frac'], d['roofline']['frac_algorithmic'], d['roofline']['traffic'], d['trace_tail'])
"
