set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
export YARD_QUICK=1
for big in 0 1; do for tpw in 2 4 6; do
  echo "## big=$big tpw=$tpw" >> $O/sort_sweep.txt
  RT_SORT_3PASS=0 RT_SORT_BIG=$big RT_SORT_TPW=$tpw timeout -k 10 120 tools/bin/sort_yardstick gpu-raytracing_amd/csrc/librt_amd_tuning.so 2237 2>&1 | grep -E "bits\(|DIFFER" >> $O/sort_sweep.txt || exit 1
done; done
for G in 1000 1500; do for big in 0 1; do
  echo "## G=$G big=$big tpw=2" >> $O/sort_sweep.txt
  RT_SORT_3PASS=0 RT_SORT_BIG=$big RT_SORT_TPW=2 timeout -k 10 120 tools/bin/sort_yardstick gpu-raytracing_amd/csrc/librt_amd_tuning.so $G 2>&1 | grep -E "bits\(|DIFFER" >> $O/sort_sweep.txt || exit 1
done; done
paste - - < $O/sort_sweep.txt | sed -E 's/morton +n=[0-9]+ +rt_radix_sort_u32_pairs_bits\(30\)//; s/ of the 80.*//'
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "radix or boundaries or 10m" > $O/pytest_sort.log 2>&1 || { tail -30 $O/pytest_sort.log; exit 1; }
tail -2 $O/pytest_sort.log
