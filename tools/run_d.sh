#!/bin/bash
source tools/gpu_steps.sh
export O=gpurun_out/r6d; mkdir -p $O
export FHERAM_TAIL2=0
step 200 ab_v256.txt env FHERAM_PRE_INV=2 bash tools/bench_ab.sh fhe-ram_amd/libfheram.so fhe-ram_amd/libfheram_v256.so 4
step 120 ab_v256_readme.txt env FHERAM_PRE_INV=2 bash tools/bench_ab.sh fhe-ram_amd/libfheram.so fhe-ram_amd/libfheram_v256.so 2 --params readme
step 100 ab_gate.txt bash tools/ab_env.sh "FHERAM_PRE_INV=1" "FHERAM_PRE_INV=2" 3
cat $O/ab_v256.txt $O/ab_v256_readme.txt $O/ab_gate.txt
