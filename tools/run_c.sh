#!/bin/bash
source tools/gpu_steps.sh
export O=gpurun_out/r6c; mkdir -p $O
step 120 exchange.txt ./tools/tail_exchange_bench
step 300 golden.log python -m pytest tests/test_gpu_golden.py tests/test_gpu_fft.py -q -x -m gpu -k "not 2097152"
step 200 ab_tail2.txt bash tools/ab_env.sh "FHERAM_TAIL2=0" "FHERAM_TAIL2=1" 3
step 200 ab_launder.txt bash tools/bench_ab.sh fhe-ram_amd/libfheram_nomon.so fhe-ram_amd/libfheram_nomon_l.so 3
step 200 ab_mon.txt bash tools/bench_ab.sh fhe-ram_amd/libfheram_nomon_l.so fhe-ram_amd/libfheram_mon_l.so 3
step 200 ab_readme.txt bash tools/bench_ab.sh fhe-ram_amd/libfheram_nomon.so fhe-ram_amd/libfheram_nomon_l.so 2 --params readme
step 60 sweep12.txt python bench.py --log-max-addr 12 --steps 20 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-boundary --no-readme-leg
cat $O/exchange.txt; tail -n 3 $O/golden.log; cat $O/ab_tail2.txt $O/ab_launder.txt $O/ab_mon.txt $O/ab_readme.txt
