# Copies the summaries of the last tools/r02_collect.sh run from gpurun_out/r02/ into profiles/ (run here, after gpurun).
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r02
cp $O/bench_default.json profiles/r02_bench_default.json
cp $O/bench_serial.json profiles/r02_bench_inflight1.json
cp $O/bench_camera_b.json profiles/r02_bench_lbvh_camera_b.json
cp $O/bench_config4.json profiles/r02_bench_config4.json
cp $O/bench_config5.json profiles/r02_bench_config5.json
cp $O/bench_sah.json profiles/r02_bench_sah.json
cp $O/bench_pairs.json profiles/r02_bench_pairs.json
cp $O/bench_sah_pairs.json profiles/r02_bench_sah_pairs.json
cp $O/bench_hybrid.json profiles/r02_bench_hybrid.json
cp $O/bench_2ranks_rehearsal.json profiles/r02_bench_2ranks_one_gpu_rehearsal.json
cp $O/bench_4ranks_rehearsal_camera_b.json profiles/r02_bench_4ranks_one_gpu_rehearsal_camera_b.json
cp $O/bench_kernel_stats.txt profiles/r02_bench_kernel_stats.txt
cp "$(ls -t $O/prof_bench/*/*kernel_stats.csv | head -1)" profiles/r02_bench_kernel_stats.csv
cp $O/build1m_kernel_stats.txt profiles/r02_build_1m_kernel_stats.txt
cp $O/build10m_kernel_stats.txt profiles/r02_build_10m_kernel_stats.txt
cp $O/trace_pmc.txt profiles/r02_trace_pmc.txt
cp $O/trace_l1_pmc.txt profiles/r02_trace_l1_pmc.txt
cp $O/lbvh_phases_1m.txt profiles/r02_lbvh_phases_1m.txt
cp $O/lbvh_phases_10m.txt profiles/r02_lbvh_phases_10m.txt
cp $O/lds_latency.txt profiles/r02_lds_latency.txt
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r02/trace_traffic.json'))
o = json.load(open('profiles/trace_traffic.json'))
d["note"] = o.get("note", "")
json.dump(d, open('profiles/trace_traffic.json', 'w'), indent=1)
for f in ["default", "inflight1", "lbvh_camera_b", "config4", "config5", "sah", "pairs", "sah_pairs", "hybrid"]:
    x = json.loads(open(f"profiles/r02_bench_{f}.json").read().strip().splitlines()[-1])
    print(f, x["value"], x.get("serial_mrays"), x.get("build_ms"), x.get("build_frac_of_hbm_peak"), x.get("sah_build_ms"))
PY
