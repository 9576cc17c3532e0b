R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4sq; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-boundary"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq -- $B > $O/sq.json 2> $O/sq.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/lds -- $B > $O/lds.json 2> $O/lds.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $O/vm -- $B > $O/vm.json 2> $O/vm.err
cd $R
python tools/pmc_sq_summary.py pmc_sq=$O/sq pmc_lds=$O/lds pmc_vmem=$O/vm 2>&1 | grep -E "^==|k_keyswitch_chain<3, 4, 3, 2>|k_ext_product_chain" > $O/summary.txt
rm -rf $O/sq $O/lds $O/vm
cat $O/summary.txt
