# park-threshold sweep of trace_kernel (tuning build: make -C gpu-raytracing_amd/csrc TUNING=1).  Run on the GPU box.
cd $GRAFT_REPO_ROOT/gpu-raytracing_amd/csrc && cp librt_amd.so /tmp/librt_amd.ship.so && make -s clean && make -s -j8 TUNING=1 librt_amd.so && cd $GRAFT_REPO_ROOT
for pk in 2,1 4,1 6,1 8,1 12,1 16,1 32,1; do
  for cfg in "--camera a" "--camera b" "--camera a --type sah"; do
    echo -n "park=$pk $cfg: " ; RT_TRACE_PARK=$pk python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $cfg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['serial_mrays'], d['wave_steps'])"
  done
done
cd $GRAFT_REPO_ROOT/gpu-raytracing_amd/csrc && make -s clean && make -s -j8 && cmp librt_amd.so /tmp/librt_amd.ship.so && echo "shipped library restored"
