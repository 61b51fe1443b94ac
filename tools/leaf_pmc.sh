cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m; mkdir -p $O
for G in 708 2237; do
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --kernel-include-regex lbvh_leaf --output-format csv -d $O/a$G -- python3 tools/build_loop.py 3 $G > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE --kernel-include-regex lbvh_leaf --output-format csv -d $O/b$G -- python3 tools/build_loop.py 3 $G > /dev/null 2>&1
done
python3 tools/pmc_summary.py $O
