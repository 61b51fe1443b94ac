# The leaf kernel's HBM traffic on the bench mesh (input in file order) and on the same triangles in Morton order:
# kernel stats + FETCH_SIZE / WRITE_SIZE / request-counter passes (gpurun -- 'bash tools/leaf_gather.sh [tag]')
# One TCC pass holds FETCH_SIZE alone (it takes 3 of the 4 TCC slots; a fuller pass hangs the profiler).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-leafg}; mkdir -p $O
for K in grid sorted; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$K -- python3 tools/build_loop.py 10 2237 $K > $O/build_$K.log 2>&1 || { tail -5 $O/build_$K.log; exit 1; }
python3 tools/kstats.py $O/st_$K > $O/kstats_$K.txt; echo "== $K"; cat $O/kstats_$K.txt; grep "build ms" $O/build_$K.log
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex lbvh_leaf --output-format csv -d $O/pmc_$K/f -- python3 tools/build_loop.py 3 2237 $K > $O/pmc_f_$K.log 2>&1 || { tail -5 $O/pmc_f_$K.log; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-include-regex lbvh_leaf --output-format csv -d $O/pmc_$K/w -- python3 tools/build_loop.py 3 2237 $K > $O/pmc_w_$K.log 2>&1 || { tail -5 $O/pmc_w_$K.log; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-include-regex lbvh_leaf --output-format csv -d $O/pmc_$K/r -- python3 tools/build_loop.py 3 2237 $K > $O/pmc_r_$K.log 2>&1 || { echo "request counters: not collected"; tail -3 $O/pmc_r_$K.log; }
python3 tools/pmc_summary.py $O/pmc_$K > $O/pmc_$K.txt 2>&1; cat $O/pmc_$K.txt
done
