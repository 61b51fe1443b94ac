# Round-3 run 1 (one gpurun call): new sort -- correctness, rocPRIM yardstick, parameter sweep, build kernel stats, then the suite.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "radix or boundaries or scene_aabb or nodes_and or stale_scratch or standalone or tiny" > $O/pytest_sort.log 2>&1 || { tail -30 $O/pytest_sort.log; exit 1; }
tail -2 $O/pytest_sort.log
timeout -k 10 300 tools/bin/sort_yardstick gpu-raytracing_amd/csrc/librt_amd.so 708 2237 > $O/sort_yardstick.txt 2>&1 || { tail -5 $O/sort_yardstick.txt; exit 1; }
cat $O/sort_yardstick.txt
export YARD_QUICK=1
for three in 0 1; do for tpw in 1 2 3 4 6 8; do for pf in 0 1; do
  if [ $tpw = 1 ] && [ $pf = 1 ]; then continue; fi
  echo "## 3pass=$three tpw=$tpw pf=$pf" >> $O/sort_sweep.txt
  RT_SORT_3PASS=$three RT_SORT_TPW=$tpw RT_SORT_PF=$pf timeout -k 10 120 tools/bin/sort_yardstick gpu-raytracing_amd/csrc/librt_amd_tuning.so 2237 2>&1 | grep bits >> $O/sort_sweep.txt || exit 1
done; done; done
for G in 708 1000 1500; do for three in 0 1; do for tpw in 1 2; do
  echo "## G=$G 3pass=$three tpw=$tpw" >> $O/sort_sweep.txt
  RT_SORT_3PASS=$three RT_SORT_TPW=$tpw RT_SORT_PF=1 timeout -k 10 120 tools/bin/sort_yardstick gpu-raytracing_amd/csrc/librt_amd_tuning.so $G 2>&1 | grep bits >> $O/sort_sweep.txt || exit 1
done; done; done
cat $O/sort_sweep.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_build10m -- python3 tools/build_loop.py 10 2237 > $O/build10m.log 2>&1 || { tail -5 $O/build10m.log; exit 1; }
python3 tools/kstats.py $O/prof_build10m > $O/build10m_kernel_stats.txt; cat $O/build10m_kernel_stats.txt; grep "build ms" $O/build10m.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_build1m -- python3 tools/build_loop.py 20 708 > $O/build1m.log 2>&1 || { tail -5 $O/build1m.log; exit 1; }
python3 tools/kstats.py $O/prof_build1m > $O/build1m_kernel_stats.txt; cat $O/build1m_kernel_stats.txt; grep "build ms" $O/build1m.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
