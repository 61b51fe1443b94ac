# 10M (and 1M) build kernel stats for experiment variants of the library: gpurun -- 'bash tools/exp_variants.sh tag v1 v2 ...'
# (variants = gpu-raytracing_amd/csrc/librt_amd_exp_<v>.so, built with make librt_amd_exp.so EXPFLAGS=... EXPNAME=...)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O; shift
for V in base "$@"; do
  if [ $V = base ]; then unset RT_LIB; else export RT_LIB=gpu-raytracing_amd/csrc/librt_amd_exp_$V.so; fi
  for G in 2237 708; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_${V}_$G -- python3 tools/build_loop.py 12 $G > $O/build_${V}_$G.log 2>&1 || { tail -5 $O/build_${V}_$G.log; exit 1; }
    python3 tools/kstats.py $O/st_${V}_$G > $O/kstats_${V}_$G.txt; echo "== $V G=$G"; grep -E "lbvh_leaf|morton_hist|scene_aabb" $O/kstats_${V}_$G.txt; grep "build ms" $O/build_${V}_$G.log
  done
done
