set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
timeout -k 10 300 tools/bin/sort_yardstick gpu-raytracing_amd/csrc/librt_amd.so 708 2237 > $O/sort_yardstick.txt 2>&1 || { tail -5 $O/sort_yardstick.txt; exit 1; }
cat $O/sort_yardstick.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_build10m -- python3 tools/build_loop.py 10 2237 > $O/build10m.log 2>&1 || { tail -5 $O/build10m.log; exit 1; }
python3 tools/kstats.py $O/prof_build10m > $O/build10m_kernel_stats.txt; cat $O/build10m_kernel_stats.txt; grep "build ms" $O/build10m.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_build1m -- python3 tools/build_loop.py 20 708 > $O/build1m.log 2>&1 || { tail -5 $O/build1m.log; exit 1; }
python3 tools/kstats.py $O/prof_build1m > $O/build1m_kernel_stats.txt; cat $O/build1m_kernel_stats.txt; grep "build ms" $O/build1m.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
