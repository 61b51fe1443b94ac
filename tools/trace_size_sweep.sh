cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i; mkdir -p $O; : > $O/summary.txt
for V in base pf96 pf80; do
  if [ $V = base ]; then unset RT_LIB; else export RT_LIB=gpu-raytracing_amd/csrc/librt_amd_exp_$V.so; fi
  for G in 1000 1500 2237; do for cam in a b; do
    python3 tools/trace_exp.py --grid $G --camera $cam --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/b.json 2> $O/b.err || { tail -3 $O/b.err; exit 1; }
    python3 -c "import json; d=json.loads(open('$O/b.json').read()); print('$V', 'G=$G cam $cam', 'inflight', d['value'], 'serial', d['serial_mrays'])" | tee -a $O/summary.txt
  done; done
done
