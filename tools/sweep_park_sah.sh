# sweep the park threshold of the trace kernel for rays traced through the SAH tree
for pk in 1,2 1,1 2,1 4,1 8,1 16,1 64,1; do
  echo -n "park=$pk " ; RT_TRACE_PARK=$pk python3 bench.py --type sah --steps 20 --warmup 3 --no-cpu-baseline --no-extras --other-camera | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['camera_b_mrays'], d['wave_steps'])"
done
