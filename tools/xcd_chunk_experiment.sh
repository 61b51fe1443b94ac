# Experiment behind trace_kernel's tile -> XCD mapping: chunks of C workgroups dealt round-robin to the XCDs,
# C = 1 (the hardware's own order) ... 60 (one tile row), against the shipped library.  Run on the GPU box from the repo root.
set -e
cd $GRAFT_REPO_ROOT/gpu-raytracing_amd/csrc
cp librt_amd.so /tmp/librt_amd.orig.so
for C in 1 2 4 8 15 60; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -DRT_TRACE_XCD_CHUNK=$C -c trace_kernel.hip -o /tmp/trace_c$C.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/librt_amd_chunk$C.so build_front.o radix_sort.o lbvh_levels.o hybrid_top.o sah_build.o rt_abi.o /tmp/trace_c$C.o
done
for v in orig chunk1 chunk2 chunk4 chunk8 chunk15 chunk60; do
  if [ $v = orig ]; then cp /tmp/librt_amd.orig.so librt_amd.so; else cp /tmp/librt_amd_$v.so librt_amd.so; fi
  for cfg in "--camera a" "--camera b" "--camera a --type sah" "--camera b --type sah"; do
    (cd $GRAFT_REPO_ROOT && python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras $cfg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$cfg', 'inflight', d['value'], 'serial', d['serial_mrays'])")
  done
done
cp /tmp/librt_amd.orig.so librt_amd.so
