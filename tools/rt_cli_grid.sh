cd $GRAFT_REPO_ROOT
C=gpu-raytracing_amd/host/rt_cli
for t in bottom-up sah; do
echo "== rt_cli - --grid 708 --camera a --type $t (single-device path)"; $C - --grid 708 --camera a --type $t --width 1920 --height 1080 --frames 12 2>&1 | grep -E "triangles|time elapsed|number of tests|frame (0|1|11):|frames: mean|num nodes"
echo "== the same with --gpus 1 (one-device RCCL communicator, bands)"; $C - --grid 708 --camera a --type $t --width 1920 --height 1080 --frames 12 --gpus 1 2>&1 | grep -E "time elapsed|number of tests|frame (0|1|11):|frames: mean"
done
echo "== camera b, --gpus 1 --partition strips"; $C - --grid 708 --camera b --type bottom-up --width 1920 --height 1080 --frames 12 --gpus 1 --partition strips 2>&1 | grep -E "number of tests|frame (1|11):|frames: mean"
echo "== camera b, single"; $C - --grid 708 --camera b --type bottom-up --width 1920 --height 1080 --frames 12 2>&1 | grep -E "number of tests|frame (1|11):|frames: mean"
