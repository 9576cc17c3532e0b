#!/bin/bash
# A longer search for the operands with the largest round-off (tools/fft_search.hip): fresh seeds and climbs resumed from the pinned worst pattern.
source tools/gpu_steps.sh
export O=gpurun_out/r6t; mkdir -p $O
step 290 search_s21.txt ./tools/fft_search 260 gpurun_out/r6t/best_s21.bin 21
step 290 search_s22.txt ./tools/fft_search 260 gpurun_out/r6t/best_s22.bin 22
step 290 search_s23_resume.txt ./tools/fft_search 260 gpurun_out/r6t/best_s23.bin 23 tests/golden/fft_worst_pattern.bin
step 290 search_s24_resume.txt ./tools/fft_search 260 gpurun_out/r6t/best_s24.bin 24 tests/golden/fft_worst_pattern.bin
tail -n 2 $O/search_*.txt
