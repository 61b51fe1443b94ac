# sweep the park threshold of the trace kernel (box-phase vs leaf-phase switch)
for pk in 1,4 1,2 1,1 2,1 3,1 4,1 8,1 64,1; do
  echo -n "park=$pk " ; RT_TRACE_PARK=$pk python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --build-reps 1 --other-camera | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['camera_b_mrays'], d['wave_steps'])"
done
