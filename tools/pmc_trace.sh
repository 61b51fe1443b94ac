export TMPDIR=/tmp
B="python3 bench.py --inflight 1 --steps 3 --warmup 1 --no-extras --no-cpu-baseline"   # one launch at a time: clean per-launch counters
O=$GRAFT_REPO_ROOT/gpurun_out/pmc1
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-include-regex trace_kernel --output-format csv -d $O/a -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-include-regex trace_kernel --output-format csv -d $O/b -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc_b.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-include-regex trace_kernel --output-format csv -d $O/c -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc_c.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex trace_kernel --output-format csv -d $O/d -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc_d.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT --kernel-include-regex trace_kernel --output-format csv -d $O/e -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc_e.log 2>&1
echo done
