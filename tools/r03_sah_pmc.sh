set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT/gpu-raytracing_amd/csrc
cp librt_amd.so /tmp/librt_amd.orig.so
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I."
for W in 1 4; do
  /opt/rocm/bin/hipcc $F -DRT_SAH_SMALL_WINDOW=$W -c sah_build.hip -o /tmp/sah_w$W.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o librt_amd.so build_front.o radix_sort.o lbvh_levels.o hybrid_top.o /tmp/sah_w$W.o rt_abi.o trace_kernel.o || exit 1
  cd $GRAFT_REPO_ROOT
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-include-regex sah_small --output-format csv -d gpurun_out/r3l/w$W/a -- python3 tools/sah_loop.py > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE --kernel-include-regex sah_small --output-format csv -d gpurun_out/r3l/w$W/b -- python3 tools/sah_loop.py > /dev/null 2>&1
  echo "== window $W"; python3 tools/pmc_summary.py gpurun_out/r3l/w$W
  cd gpu-raytracing_amd/csrc
done
cp /tmp/librt_amd.orig.so librt_amd.so
