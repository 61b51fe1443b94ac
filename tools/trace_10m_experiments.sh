# Round-4 tracer experiments on the workload where misses, not the address path, limit traversal (bench.py --preset config4:
# 10M triangles, 1.28 GB of nodes + leaves): register cap / occupancy arms, waves per workgroup, and the pair-prefetch arm,
# each also on the 1M headline frame (a per-size choice is only worth keeping if it costs nothing there).
#   gpurun -- 'bash tools/trace_10m_experiments.sh <tag> v1 v2 ...'     variants = csrc/librt_amd_exp_<v>.so
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O; shift
R=$O/summary.txt; : > $R
for V in base "$@"; do
  if [ $V = base ]; then unset RT_LIB; else export RT_LIB=gpu-raytracing_amd/csrc/librt_amd_exp_$V.so; fi
  for cfg in "10m_a:--preset config4" "10m_b:--preset config4 --camera b" "1m_a:" "1m_sah:--type sah" "1m_b:--camera b"; do
    tag=${cfg%%:*}; args=${cfg#*:}
    python3 tools/trace_exp.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras $args > $O/bench_${V}_$tag.json 2> $O/bench_${V}_$tag.err || { tail -5 $O/bench_${V}_$tag.err; exit 1; }
    python3 -c "import json; d=json.loads(open('$O/bench_${V}_$tag.json').read()); print('$V', '$tag', 'inflight', d['value'], 'serial', d['serial_mrays'], 'box/ray', d['box_tests_per_ray'])" | tee -a $R
  done
  for C in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
    rocprofv3 --kernel-trace --pmc $C --kernel-include-regex trace_kernel --output-format csv -d $O/pmc_${V}_${C%% *} -- python3 tools/trace_exp.py --preset config4 --inflight 1 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/pmc_${V}_${C%% *}.log 2>&1 || echo "pmc $V $C failed" | tee -a $R
  done
  python3 tools/pmc_summary.py $O/pmc_${V}_TCC_HIT_sum 2>/dev/null | grep -E "trace_kernel<1>|TCC_" | sed "s/^/$V  /" | tee -a $R
  python3 tools/pmc_summary.py $O/pmc_${V}_FETCH_SIZE 2>/dev/null | grep -A1 "trace_kernel<1>" | grep FETCH | sed "s/^/$V  /" | tee -a $R
done
# ---- scene-size sweep of the same variants (where does the pair prefetch start to pay?): G=1000 -> 2.0M, 1500 -> 4.5M, 2237 -> 10M
if [ -n "$RT_SIZE_SWEEP" ]; then
  : > $O/size_sweep.txt
  for V in base "$@"; do
    if [ $V = base ]; then unset RT_LIB; else export RT_LIB=gpu-raytracing_amd/csrc/librt_amd_exp_$V.so; fi
    for G in 1000 1500 2237; do for cam in a b; do
      python3 tools/trace_exp.py --grid $G --camera $cam --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/b.json 2> $O/b.err || { tail -3 $O/b.err; exit 1; }
      python3 -c "import json; d=json.loads(open('$O/b.json').read()); print('$V', 'G=$G cam $cam', 'inflight', d['value'], 'serial', d['serial_mrays'])" | tee -a $O/size_sweep.txt
    done; done
  done
fi
