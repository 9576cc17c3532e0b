#!/bin/bash
# One gpurun call: the round's profile set (tools/collect_profiles.sh -> gpurun_out/r6q) and the one-GPU scaling prediction.
source tools/gpu_steps.sh
export O=gpurun_out/r6q; mkdir -p $O
step 1000 collect.log bash tools/collect_profiles.sh
step 160 scaling_model.txt python tools/scaling_model.py --json gpurun_out/r6q/scaling_model.json
ls $O
