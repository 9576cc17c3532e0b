#!/bin/bash
source tools/gpu_steps.sh
export O=gpurun_out/r6m; mkdir -p $O
step 900 pytest_all.log python -m pytest tests -q -x -m gpu
step 300 bench_default.json python bench.py
tail -n 3 $O/pytest_all.log
python3 -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('default bench', d['value'], d['ms_per_step'], d['read_ms'], d['read_prepare_write_ms'], d['write_ms'], d.get('value_incl_boundary'), d['ops_enqueued_back_to_back']['ram_ops_s'], d['parity_in_run']['read'], d['parity_in_run']['rpw'], d['parity_in_run']['rows_after_write'], d['trace_tail'])
"
