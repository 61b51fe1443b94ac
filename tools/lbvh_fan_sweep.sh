cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp gpu-raytracing_amd/csrc/librt_amd.so /tmp/orig.so; cp gpu-raytracing_amd/csrc/librt_amd_tuning.so gpu-raytracing_amd/csrc/librt_amd.so
for G in 708 2237; do for F in 16 24 32 40 48 56 64; do
  rm -rf gpurun_out/fan; RT_LBVH_FAN=$F rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fan -- python3 tools/build_loop.py 8 $G > /tmp/b.log 2>&1
  echo "G=$G fan=$F $(python3 tools/kstats.py gpurun_out/fan | grep upper | sed 's/ *calls.*avg/ avg/; s/total.*//') $(grep 'build ms' /tmp/b.log | sed 's/.*\[[^,]*, \([^,]*\), \([^,]*\),.*/\1 \2/')"
done; done
cp /tmp/orig.so gpu-raytracing_amd/csrc/librt_amd.so
