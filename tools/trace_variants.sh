# Tracer Mrays/s (serial and 8 frames in flight, camera A, LBVH and SAH tree) for experiment variants of the library:
# gpurun -- 'bash tools/trace_variants.sh tag v1 v2 ...'   (variants = csrc/librt_amd_exp_<v>.so)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O; shift
for V in base "$@"; do
  if [ $V = base ]; then unset RT_LIB; else export RT_LIB=gpu-raytracing_amd/csrc/librt_amd_exp_$V.so; fi
  for T in bottom-up sah; do
    python3 tools/trace_exp.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --type $T > $O/bench_${V}_$T.json 2> $O/bench_${V}_$T.err || { tail -5 $O/bench_${V}_$T.err; exit 1; }
    python3 -c "import sys,json; d=json.loads(open('$O/bench_${V}_$T.json').read()); print('$V', '$T', 'inflight', d['value'], 'serial', d['serial_mrays'])"
  done
  python3 tools/trace_exp.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --camera b > $O/bench_${V}_camb.json 2> $O/bench_${V}_camb.err || { tail -5 $O/bench_${V}_camb.err; exit 1; }
  python3 -c "import sys,json; d=json.loads(open('$O/bench_${V}_camb.json').read()); print('$V', 'camera b', 'inflight', d['value'], 'serial', d['serial_mrays'])"
done
