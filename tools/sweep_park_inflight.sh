# park threshold sweep with 8 frames in flight (LBVH and SAH)
for ty in bottom-up sah; do
for pk in 2,1 4,1 6,1 8,1 12,1 16,1; do
  echo -n "type=$ty park=$pk " ; RT_TRACE_PARK=$pk python3 bench.py --type $ty --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['wave_steps'])"
done; done
