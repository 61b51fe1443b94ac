set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "radix or boundaries or scene_aabb or nodes_and or stale_scratch or standalone or tiny" > $O/pytest_sort.log 2>&1 || { tail -30 $O/pytest_sort.log; exit 1; }
tail -2 $O/pytest_sort.log
timeout -k 10 300 tools/bin/sort_yardstick gpu-raytracing_amd/csrc/librt_amd.so 708 2237 > $O/sort_yardstick.txt 2>&1 || { tail -5 $O/sort_yardstick.txt; exit 1; }
cat $O/sort_yardstick.txt
export YARD_QUICK=1
for three in 0 1; do for tpw in 1 2 3 4; do
  echo "## 3pass=$three tpw=$tpw" >> $O/sort_sweep.txt
  RT_SORT_3PASS=$three RT_SORT_TPW=$tpw timeout -k 10 120 tools/bin/sort_yardstick gpu-raytracing_amd/csrc/librt_amd_tuning.so 2237 2>&1 | grep "bits(" >> $O/sort_sweep.txt || exit 1
done; done
for G in 708 1000 1500; do for three in 0 1; do for tpw in 1 2; do
  echo "## G=$G 3pass=$three tpw=$tpw" >> $O/sort_sweep.txt
  RT_SORT_3PASS=$three RT_SORT_TPW=$tpw timeout -k 10 120 tools/bin/sort_yardstick gpu-raytracing_amd/csrc/librt_amd_tuning.so $G 2>&1 | grep "bits(" >> $O/sort_sweep.txt || exit 1
done; done; done
export YARD_NOCHECK=1
for e in 1 2 3; do
  echo "## 3pass=0 tpw=1 exp=$e" >> $O/sort_sweep.txt
  RT_SORT_3PASS=0 RT_SORT_TPW=1 RT_SORT_EXP=$e timeout -k 10 120 tools/bin/sort_yardstick gpu-raytracing_amd/csrc/librt_amd_tuning.so 2237 2>&1 | grep "bits(" >> $O/sort_sweep.txt
done
paste - - < $O/sort_sweep.txt | sed -E 's/morton +n=[0-9]+ +rt_radix_sort_u32_pairs_bits\(30\)//; s/ of the 80.*//'
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_build10m -- python3 tools/build_loop.py 10 2237 > $O/build10m.log 2>&1 || { tail -5 $O/build10m.log; exit 1; }
python3 tools/kstats.py $O/prof_build10m > $O/build10m_kernel_stats.txt; cat $O/build10m_kernel_stats.txt; grep "build ms" $O/build10m.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_build1m -- python3 tools/build_loop.py 20 708 > $O/build1m.log 2>&1 || { tail -5 $O/build1m.log; exit 1; }
python3 tools/kstats.py $O/prof_build1m > $O/build1m_kernel_stats.txt; cat $O/build1m_kernel_stats.txt; grep "build ms" $O/build1m.log
