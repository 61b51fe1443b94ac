# Prices the register spills of trace_kernel's 64-VGPR instantiations: the shipped build (kDepth at
# 8 waves/SIMD = 64 VGPRs, 21 spilled), the same kernel at 7 waves (72 VGPRs, 6 spilled) and at 64 VGPRs without the
# wave-step counters.  Per arm: Mrays/s serial and with 8 frames in flight (camera A, LBVH, 1M triangles, 1080p) and the
# kernel's WRITE_SIZE (one --pmc pass, program directly after `--`).  Run on the GPU box from the repo root.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3h; mkdir -p $O
cd gpu-raytracing_amd/csrc
cp librt_amd.so /tmp/librt_amd.orig.so
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I."
/opt/rocm/bin/hipcc $F -DRT_TRACE_LEAN_EXTRA=0 -c trace_kernel.hip -o /tmp/trace_lean0.o || exit 1
/opt/rocm/bin/hipcc $F -DRT_TRACE_NO_STEPS -c trace_kernel.hip -o /tmp/trace_nosteps.o || exit 1
for v in lean0 nosteps; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/librt_amd_$v.so build_front.o radix_sort.o lbvh_levels.o hybrid_top.o sah_build.o rt_abi.o /tmp/trace_$v.o || exit 1
done
for v in orig lean0 nosteps; do
  cp /tmp/librt_amd$([ $v = orig ] && echo .orig || echo _$v).so librt_amd.so
  (cd $GRAFT_REPO_ROOT && python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'camera a: inflight', d['value'], 'serial', d['serial_mrays'])") >> $O/spill_experiment.txt
  (cd $GRAFT_REPO_ROOT && python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --type sah 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'camera a, SAH tree: inflight', d['value'], 'serial', d['serial_mrays'])") >> $O/spill_experiment.txt
  (cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex trace_kernel --output-format csv -d $O/w_$v -- python3 bench.py --inflight 1 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/w_$v.log 2>&1)
  (cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py $O/w_$v | grep -E "trace_kernel|WRITE_SIZE" | sed "s/^/$v  /") >> $O/spill_experiment.txt
done
cp /tmp/librt_amd.orig.so librt_amd.so
cat $O/spill_experiment.txt
