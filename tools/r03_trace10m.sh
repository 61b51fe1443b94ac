# Traversal where HBM can matter: the 10M-triangle scene (BVH 1.28 GB > 256 MiB Infinity Cache).  Counter passes on
# bench.py --preset config4 (program directly after `--`; FETCH_SIZE and WRITE_SIZE in separate passes), then the XCD chunk
# sweep of the tile order on that scene.  Outputs under gpurun_out/r3g.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3g; mkdir -p $O
B="python3 bench.py --preset config4 --inflight 1 --steps 3 --warmup 1 --no-extras --no-cpu-baseline"
$B > $O/bench_config4_serial.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cat $O/bench_config4_serial.json
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-include-regex trace_kernel --output-format csv -d $O/a -- $B > $O/pmc_a.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-include-regex trace_kernel --output-format csv -d $O/b -- $B > $O/pmc_b.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-include-regex trace_kernel --output-format csv -d $O/c -- $B > $O/pmc_c.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex trace_kernel --output-format csv -d $O/d -- $B > $O/pmc_d.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT --kernel-include-regex trace_kernel --output-format csv -d $O/e -- $B > $O/pmc_e.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE --kernel-include-regex trace_kernel --output-format csv -d $O/f -- $B > $O/pmc_f.log 2>&1 || exit 1
python3 tools/pmc_summary.py $O > $O/trace_pmc_10m.txt 2>&1
python3 tools/pmc_traffic.py $O/d $O/e $O/trace_traffic_10m.json "$(date -u +%Y-%m-%d) config4" > $O/pmc_traffic.log 2>&1
cat $O/trace_traffic_10m.json
# ---- XCD chunk sweep on the 10M scene (camera A and B, serial + in flight)
cd $GRAFT_REPO_ROOT/gpu-raytracing_amd/csrc
cp librt_amd.so /tmp/librt_amd.orig.so
for C in 1 4 8 16 60; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -DRT_TRACE_XCD_CHUNK=$C -c trace_kernel.hip -o /tmp/trace_c$C.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/librt_amd_chunk$C.so build_front.o radix_sort.o lbvh_levels.o hybrid_top.o sah_build.o rt_abi.o /tmp/trace_c$C.o || exit 1
done
for v in orig chunk1 chunk4 chunk8 chunk16 chunk60; do
  if [ $v = orig ]; then cp /tmp/librt_amd.orig.so librt_amd.so; else cp /tmp/librt_amd_$v.so librt_amd.so; fi
  for cfg in "--camera a" "--camera b"; do
    (cd $GRAFT_REPO_ROOT && python3 bench.py --preset config4 --steps 10 --warmup 2 --no-cpu-baseline --no-extras $cfg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$cfg', 'inflight', d['value'], 'serial', d['serial_mrays'])") >> $O/xcd_chunk_sweep_10m.txt
  done
done
cp /tmp/librt_amd.orig.so librt_amd.so
cat $O/xcd_chunk_sweep_10m.txt
