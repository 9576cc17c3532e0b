#!/bin/bash
source tools/gpu_steps.sh
export O=gpurun_out/r6b; mkdir -p $O
step 240 fft.log python -m pytest tests/test_gpu_fft.py tests/test_gpu_extremes.py -q -x -m gpu -s
step 300 ab_monitor.txt bash tools/bench_ab.sh fhe-ram_amd/libfheram_nomon.so fhe-ram_amd/libfheram.so 4
step 120 bench_readme.json python bench.py --params readme --no-cpu-baseline
step 200 golden_readme.log python -m pytest tests/test_gpu_golden.py -q -x -m gpu -k "readme or 16384"
step 160 fft_search2.txt ./tools/fft_search 130 gpurun_out/r6b/fft_search_best2.bin 2
step 160 fft_search3.txt ./tools/fft_search 130 gpurun_out/r6b/fft_search_best3.bin 3 tests/golden/fft_worst_pattern.bin
step 600 pytest_all.log python -m pytest tests -q -x -m gpu
tail -n 3 $O/fft.log $O/golden_readme.log $O/pytest_all.log; cat $O/ab_monitor.txt; tail -n 2 $O/fft_search2.txt $O/fft_search3.txt
python3 -c "
import json; d=json.load(open('$O/bench_readme.json')); print('readme', d['value'], d['ms_per_step'], d['read_ms'], d['read_prepare_write_ms'], d['write_ms'])"
