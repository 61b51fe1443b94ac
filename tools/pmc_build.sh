# PMC passes over the build kernels (1M-triangle grid): python3 tools/build_loop.py runs N builds.
export TMPDIR=/tmp
B="python3 tools/build_loop.py 3 ${G:-708}"
O=$GRAFT_REPO_ROOT/gpurun_out/pmcb
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/a -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmcb_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $O/b -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmcb_b.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/c -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmcb_c.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/d -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmcb_d.log 2>&1
echo done
