set -x
# Collects the round's profile set on an MI355X box (run through gpurun); summaries land in gpurun_out/r6q, from where
# the ones to be judged are copied into profiles/.  Counters are collected in separate passes (FETCH_SIZE, WRITE_SIZE, SQ).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r6q; mkdir -p $O
for t in fetch_calib tail_exchange_bench; do [ -x $R/tools/$t ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $R/tools/$t $R/tools/$t.hip; done
[ -x $R/tools/fft_bench ] || /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wno-unused-value -Wno-unused-result -I$R/fhe-ram_amd/csrc -o $R/tools/fft_bench $R/tools/fft_bench.hip
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-boundary --no-readme-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-boundary --no-readme-leg > $O/stats.json 2> $O/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.json 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.json 2> $O/write.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cfetch -- $R/tools/fetch_calib > $O/cfetch.txt 2> $O/cfetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cwrite -- $R/tools/fetch_calib > $O/cwrite.txt 2> $O/cwrite.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq -- $B > $O/sq.json 2> $O/sq.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/lds -- $B > $O/lds.json 2> $O/lds.err
cd $R
python tools/trace_summary.py $O/stats 40 > $O/kernel_trace_by_grid.txt
(cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $O/tl -- python3 $R/bench.py --steps 4 --warmup 8 --no-cpu-baseline --no-boundary --no-kernel-timing --no-readme-leg > $O/tl.json 2> $O/tl.err)
python tools/trace_timeline.py $O/tl 0 100000 | tail -420 | head -260 > $O/timeline.txt   # steps of the timed region, no per-launch events
rm -rf $O/tl
(cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $O/tl14 -- python3 $R/bench.py --log-max-addr 14 --steps 2 --warmup 8 --no-cpu-baseline --no-boundary --no-kernel-timing --no-readme-leg > $O/tl14.json 2> $O/tl14.err)
python tools/trace_timeline.py $O/tl14 0 100000 | tail -110 > $O/timeline_2_14.txt   # the source default size: the mid-batch chains
rm -rf $O/tl14
cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv
python tools/pmc_hbm.py $O/cfetch $O/cwrite $O/fetch $O/write > $O/pmc_hbm_traffic.json
python tools/pmc_sq_summary.py pmc_sq=$O/sq pmc_lds=$O/lds > $O/pmc_sq_summary.txt
python bench.py > $O/bench.json 2> $O/bench.err
rm -rf $O/stats $O/fetch $O/write $O/cfetch $O/cwrite $O/sq $O/lds
ls -la $O
python bench.py --log-max-addr 12 --no-cpu-baseline > $O/bench_2_12.json 2> $O/bench_2_12.err
python bench.py --log-max-addr 21 --steps 5 --no-cpu-baseline > $O/bench_2_21.json 2> $O/bench_2_21.err
python bench.py --workload ep > $O/bench_ep.json 2> $O/bench_ep.err
python bench.py --params readme > $O/bench_readme.json 2> $O/bench_readme.err
python tools/chain_bench.py 256 600 > $O/chain_bench.txt 2>&1
( echo "# python bench.py --log-max-addr K --steps 20 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-boundary   (one MI355X, WORDSIZE 4)"; echo "log2(MAX_ADDR)  read_ms  rpw_ms  write_ms  ms_per_step  RAM ops/s  single-launch trace chains / fallbacks   mid-batch chains / fallbacks";
  for K in 12 13 14 15 16 18 20 21 22 24; do ST=20; [ $K -ge 22 ] && ST=4; python bench.py --log-max-addr $K --steps $ST --warmup 10 --no-cpu-baseline --no-kernel-timing --no-boundary --no-readme-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%6d %10.3f %9.3f %9.3f %10.3f %10.1f   %d/%d   %d/%d' % ($K, d['read_ms'], d['read_prepare_write_ms'], d['write_ms'], d['ms_per_step'], d['value'], d['trace_tail']['launches'], d['trace_tail']['fallbacks'], d['mid_chain']['launches'], d['mid_chain']['fallbacks']))"; done ) > $O/size_sweep.txt
./tools/fft_bench > $O/fft_bench.txt 2>&1
python tools/chain_n.py > $O/chain_n.txt 2>&1
FHERAM_SAFE=1 python bench.py --no-cpu-baseline > $O/bench_safe.json 2> $O/bench_safe.err
FHERAM_PRE_INV=2 python bench.py --no-cpu-baseline > $O/bench_event_fork.json 2> $O/bench_event_fork.err
[ -f fhe-ram_amd/libfheram_stamp.so ] && FHERAM_LIB=$R/fhe-ram_amd/libfheram_stamp.so python tools/stamp_z.py > $O/stamps_chain_step.txt 2>&1
