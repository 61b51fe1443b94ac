set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
export YARD_QUICK=1 YARD_NOCHECK=1 RT_SORT_3PASS=0 RT_SORT_PF=0
for tpw in 1 2; do for e in 0 1 2 3; do
  echo "## tpw=$tpw exp=$e" >> $O/exp.txt
  RT_SORT_TPW=$tpw RT_SORT_EXP=$e timeout -k 10 120 tools/bin/sort_yardstick gpu-raytracing_amd/csrc/librt_amd_tuning.so 2237 2>&1 | grep bits >> $O/exp.txt
done; done
cat $O/exp.txt
