# Sort tile size sweep: LBVH build of 0.5M / 1M / 2M triangles with 4096- (shipped), 2048- and 1024-key sort tiles.
#   gpurun -- 'bash tools/sort_tile_sweep.sh <tag> v1 v2 ...'      variants = csrc/librt_amd_exp_<v>.so
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O; shift
: > $O/summary.txt
for V in base "$@"; do
  if [ $V = base ]; then unset RT_LIB; else export RT_LIB=gpu-raytracing_amd/csrc/librt_amd_exp_$V.so; fi
  for G in 500 708 1000; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_${V}_$G -- python3 tools/build_loop.py 16 $G > $O/build_${V}_$G.log 2>&1 || { tail -5 $O/build_${V}_$G.log; exit 1; }
    python3 tools/kstats.py $O/st_${V}_$G > $O/kstats_${V}_$G.txt
    echo "== $V G=$G $(grep -o 'median [0-9.]*' $O/build_${V}_$G.log) ms $(grep -o 'checksum [0-9a-fx]*' $O/build_${V}_$G.log)" | tee -a $O/summary.txt
    grep -E "sort_|morton_hist" $O/kstats_${V}_$G.txt | awk '{printf "     %-46s calls %s avg %s us\n", $1, $3, $5}' | tee -a $O/summary.txt
  done
done
