# experiment: waves (tiles) per workgroup of trace_kernel x XCD chunk.  Run on the GPU box from the repo root.
set -e
cd $GRAFT_REPO_ROOT/gpu-raytracing_amd/csrc
cp librt_amd.so /tmp/librt_amd.orig.so
for W in 2 8; do for C in 4 8 16; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -DRT_TRACE_WAVES=$W -DRT_TRACE_XCD_CHUNK=$C -c trace_kernel.hip -o /tmp/trace_w.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o librt_amd.so build_front.o radix_sort.o lbvh_levels.o hybrid_top.o sah_build.o rt_abi.o /tmp/trace_w.o
  for cfg in "--camera a" "--camera b" "--camera a --type sah"; do
    (cd $GRAFT_REPO_ROOT && python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras $cfg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves $W chunk $C', '$cfg', 'inflight', d['value'], 'serial', d['serial_mrays'])")
  done
done; done
cp /tmp/librt_amd.orig.so librt_amd.so
