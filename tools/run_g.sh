#!/bin/bash
source tools/gpu_steps.sh
export O=gpurun_out/r6g; mkdir -p $O
step 900 pytest_all.log python -m pytest tests -q -x -m gpu --durations=15
tail -n 25 $O/pytest_all.log
