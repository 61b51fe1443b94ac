# LBVH parity tests, then kernel stats of the 10M and 1M builds (gpurun -- 'bash tools/lbvh_check.sh [tag]')
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-lbvh}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pairs.py tests/test_gpu_hybrid.py tests/test_gpu_graph.py -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_build10m -- python3 tools/build_loop.py 10 2237 > $O/build10m.log 2>&1 || { tail -5 $O/build10m.log; exit 1; }
python3 tools/kstats.py $O/prof_build10m > $O/build10m_kernel_stats.txt; cat $O/build10m_kernel_stats.txt; grep "build ms" $O/build10m.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_build1m -- python3 tools/build_loop.py 20 708 > $O/build1m.log 2>&1 || { tail -5 $O/build1m.log; exit 1; }
python3 tools/kstats.py $O/prof_build1m > $O/build1m_kernel_stats.txt; cat $O/build1m_kernel_stats.txt; grep "build ms" $O/build1m.log
