# Copies the summaries of the last tools/r03_collect.sh run from gpurun_out/r03/ into profiles/ (run here, after gpurun).
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r03
for f in default serial:inflight1 camera_b:lbvh_camera_b config4 config5 sah pairs sah_pairs hybrid 2ranks_rehearsal:2ranks_one_gpu_rehearsal; do
  src=${f%%:*}; dst=${f##*:}
  cp $O/bench_$src.json profiles/r03_bench_$dst.json
done
cp $O/bench_kernel_stats.txt profiles/r03_bench_kernel_stats.txt
cp "$(ls -t $O/prof_bench/*/*kernel_stats.csv | head -1)" profiles/r03_bench_kernel_stats.csv
cp $O/build1m_kernel_stats.txt profiles/r03_build_1m_kernel_stats.txt
cp $O/build10m_kernel_stats.txt profiles/r03_build_10m_kernel_stats.txt
cp $O/build_pmc_1m.txt profiles/r03_build_pmc_1m.txt
cp $O/build_pmc_10m.txt profiles/r03_build_pmc_10m.txt
cp $O/trace_pmc.txt profiles/r03_trace_pmc.txt
cp $O/sah_build_1m_kernel_stats.txt profiles/r03_sah_build_1m_kernel_stats.txt
cp $O/sort_yardstick.txt profiles/r03_sort_yardstick.txt
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r03/trace_traffic.json'))
o = json.load(open('profiles/trace_traffic.json'))
d["note"] = o.get("note", "")
json.dump(d, open('profiles/trace_traffic.json', 'w'), indent=1)
for f in ["default", "inflight1", "lbvh_camera_b", "config4", "config5", "sah", "pairs", "sah_pairs", "hybrid"]:
    x = json.loads(open(f"profiles/r03_bench_{f}.json").read().strip().splitlines()[-1])
    print(f, x["value"], x.get("serial_mrays"), x.get("build_ms"), x.get("build_frac_of_hbm_peak"), x.get("sah_build_ms"), (x.get("build") or {}).get("sort", {}).get("us"))
PY
