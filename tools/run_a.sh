#!/bin/bash
source tools/gpu_steps.sh
export O=gpurun_out/r6a; mkdir -p $O
step 240 fft.log python -m pytest tests/test_gpu_fft.py -q -x -m gpu
step 120 bench1.json python bench.py --steps 30 --warmup 10 --no-cpu-baseline
step 200 ab_monitor.txt bash tools/ab_env.sh "FHERAM_MONITOR=0" "FHERAM_MONITOR=1" 3
step 100 ab_monitor2.txt bash tools/ab_env.sh "FHERAM_MONITOR=2" "FHERAM_MONITOR=1" 2
step 150 fft_search.txt ./tools/fft_search 100 gpurun_out/r6a/fft_search_best.bin 1
step 420 extremes.log python -m pytest tests/test_gpu_extremes.py -q -x -m gpu -s
tail -3 $O/fft.log $O/extremes.log; cat $O/ab_monitor.txt $O/ab_monitor2.txt; tail -4 $O/fft_search.txt
