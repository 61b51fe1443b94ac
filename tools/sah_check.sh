# SAH builder: bit-exactness tests, kernel split, build ms (one gpurun call while working on sah_build.hip)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_sah.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
rm -rf gpurun_out/sahchk; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sahchk -- python3 tools/sah_loop.py > /dev/null 2>&1
python3 tools/kstats.py gpurun_out/sahchk | head -9
python3 tools/sah_bench.py 2>&1 | grep -o "sah_build_ms_median[^,]*"
