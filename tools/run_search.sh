#!/bin/bash
# A longer search for the operands with the largest round-off (tools/fft_search.hip): fresh seeds and climbs resumed from the pinned worst pattern.
source tools/gpu_steps.sh
export O=gpurun_out/r6s; mkdir -p $O
step 290 search_s11.txt ./tools/fft_search 260 gpurun_out/r6s/best_s11.bin 11
step 290 search_s12.txt ./tools/fft_search 260 gpurun_out/r6s/best_s12.bin 12
step 290 search_s13_resume.txt ./tools/fft_search 260 gpurun_out/r6s/best_s13.bin 13 tests/golden/fft_worst_pattern.bin
step 290 search_s14_resume.txt ./tools/fft_search 260 gpurun_out/r6s/best_s14.bin 14 tests/golden/fft_worst_pattern.bin
tail -n 2 $O/search_*.txt
