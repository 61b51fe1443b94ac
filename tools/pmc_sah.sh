# PMC passes over the SAH build kernels (1M-triangle grid): python3 tools/sah_loop.py runs 6 builds.
export TMPDIR=/tmp
B="python3 tools/sah_loop.py"
O=$GRAFT_REPO_ROOT/gpurun_out/pmcs
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/a -- $B > gpurun_out/pmcs_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $O/b -- $B > gpurun_out/pmcs_b.log 2>&1
echo done
