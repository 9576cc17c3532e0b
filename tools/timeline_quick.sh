# kernel timeline of a few timed steps (no per-launch events): gpurun_out/r4tl/timeline.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5tl; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tl -- python3 $R/bench.py --steps 4 --warmup 8 --no-cpu-baseline --no-boundary --no-kernel-timing ${BENCH_EXTRA:-} > $O/tl.json 2> $O/tl.err
cd $R
python tools/trace_timeline.py $O/tl 0 100000 | tail -${TL_TAIL:-420} | head -${TL_HEAD:-260} > $O/timeline.txt
rm -rf $O/tl
