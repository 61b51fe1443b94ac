# Round-3 evidence run (one gpurun call): tests, bench lines, rocprof summaries, counter passes.  Outputs under gpurun_out/r03/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
python3 bench.py --steps 50 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_profiled.json 2> $O/prof_bench.log || exit 1
python3 tools/kstats.py $O/prof_bench > $O/bench_kernel_stats.txt
python3 bench.py --steps 30 --warmup 5 --inflight 1 --no-cpu-baseline > $O/bench_serial.json 2>> $O/bench_default.err || exit 1
python3 bench.py --steps 30 --warmup 5 --camera b --no-cpu-baseline --no-extras > $O/bench_camera_b.json 2>> $O/bench_default.err || exit 1
python3 bench.py --preset config4 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_config4.json 2>> $O/bench_default.err || exit 1
python3 bench.py --preset config5 --steps 5 --warmup 1 > $O/bench_config5.json 2>> $O/bench_default.err || exit 1
python3 bench.py --type sah --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_sah.json 2>> $O/bench_default.err || exit 1
python3 bench.py --gpus 2 --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_2ranks_rehearsal.json 2>> $O/bench_default.err || exit 1
python3 bench.py --type bottom-up-pairs --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_pairs.json 2>> $O/bench_default.err
python3 bench.py --type sah-pairs --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_sah_pairs.json 2>> $O/bench_default.err
python3 bench.py --type hybrid --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_hybrid.json 2>> $O/bench_default.err
bash tools/pmc_trace.sh > $O/pmc_trace.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/pmc1/d gpurun_out/pmc1/e $O/trace_traffic.json "$(date -u +%Y-%m-%d)" > $O/pmc_traffic.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc1 > $O/trace_pmc.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_build10m -- python3 tools/build_loop.py 10 2237 > $O/build10m.log 2>&1
python3 tools/kstats.py $O/prof_build10m > $O/build10m_kernel_stats.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_build1m -- python3 tools/build_loop.py 20 708 > $O/build1m.log 2>&1
python3 tools/kstats.py $O/prof_build1m > $O/build1m_kernel_stats.txt
rm -rf gpurun_out/pmcb; G=708 bash tools/pmc_build.sh > $O/pmc_build1m.log 2>&1; python3 tools/pmc_summary.py gpurun_out/pmcb > $O/build_pmc_1m.txt 2>&1
rm -rf gpurun_out/pmcb; G=2237 bash tools/pmc_build.sh > $O/pmc_build10m.log 2>&1; python3 tools/pmc_summary.py gpurun_out/pmcb > $O/build_pmc_10m.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sah1m -- python3 tools/sah_loop.py > $O/sah1m.log 2>&1
python3 tools/kstats.py $O/prof_sah1m > $O/sah_build_1m_kernel_stats.txt
timeout -k 10 300 tools/bin/sort_yardstick gpu-raytracing_amd/csrc/librt_amd.so 708 2237 > $O/sort_yardstick.txt 2>&1
echo collected
