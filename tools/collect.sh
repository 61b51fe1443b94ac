#!/bin/bash
# One evidence collector for every round (replaces the r0N_run*/collect/publish scripts).
#
#   on the GPU box (through gpurun):   bash tools/collect.sh run <round> [section ...]
#   here, after gpurun merged the run: bash tools/collect.sh publish <round>
#
# `run` writes under gpurun_out/<round>/; `publish` copies the summaries to profiles/<round>_*.  Sections (default: all):
#   tests     pytest -m gpu
#   bench     the default bench line, the same command under rocprofv3 --kernel-trace --stats, and the other configurations
#   build     rocprofv3 --kernel-trace --stats of the 1M and 10M LBVH builds and of the SAH build
#   pmc       counter passes of the trace kernel (HBM traffic, TA / L1 utilisation), 1M scene
#   pmc10m    the same on the 10M scene (bench.py --preset config4)
#   sort      the sort against rocPRIM (tools/bin/sort_yardstick)
set -o pipefail
MODE=$1; R=$2; shift 2
SECTIONS=${*:-tests bench build pmc pmc10m sort}
has() { [[ " $SECTIONS " == *" $1 "* ]]; }

if [ "$MODE" = run ]; then
  cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
  O=gpurun_out/$R; mkdir -p $O
  prof() { rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1 -- "${@:2}" > $O/$1.log 2>&1 && python3 tools/kstats.py $O/$1 > $O/$1_kernel_stats.txt; }
  B1="--inflight 1 --steps 3 --warmup 1 --no-extras --no-cpu-baseline"   # one launch at a time: clean per-launch counters
  pmc() { mkdir -p $(dirname $1); rocprofv3 --kernel-trace --pmc ${@:3} --kernel-include-regex trace_kernel --output-format csv -d $1 -- python3 bench.py $B1 $2 > $1.log 2>&1 || echo "pmc pass $1 failed"; }
  pmcset() {   # $1 = output dir prefix, $2 = extra bench arguments
    pmc $1/a "$2" SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
    pmc $1/c "$2" TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
    pmc $1/d "$2" FETCH_SIZE
    pmc $1/e "$2" WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT
    pmc $1/l1 "$2" TA_TA_BUSY_sum GRBM_GUI_ACTIVE
    pmc $1/l2 "$2" TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
    pmc $1/l3 "$2" TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum
  }
  if has tests; then
    timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
    tail -2 $O/pytest.log
  fi
  if has bench; then
    python3 bench.py --steps 50 --warmup 5 > $O/bench_default.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
    prof prof_bench python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_profiled.json || exit 1
    python3 bench.py --steps 30 --warmup 5 --inflight 1 --no-cpu-baseline > $O/bench_inflight1.json 2>> $O/bench.err || exit 1
    python3 bench.py --steps 30 --warmup 5 --camera b --no-cpu-baseline --no-extras > $O/bench_lbvh_camera_b.json 2>> $O/bench.err || exit 1
    python3 bench.py --preset config4 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_config4.json 2>> $O/bench.err || exit 1
    python3 bench.py --preset config5 --steps 5 --warmup 1 > $O/bench_config5.json 2>> $O/bench.err || exit 1
    for t in sah bottom-up-pairs sah-pairs hybrid; do
      python3 bench.py --type $t --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_$t.json 2>> $O/bench.err
    done
    python3 bench.py --gpus 2 --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_2ranks_one_gpu_rehearsal.json 2>> $O/bench.err
    for cam in a b; do
      for k in 1 4; do
        echo -n "rt_cli --gpus 1 --inflight $k, grid 708 (1M triangles), 1920x1080, camera $cam: "
        gpu-raytracing_amd/host/rt_cli - --grid 708 --camera $cam --type bottom-up --width 1920 --height 1080 --gpus 1 --inflight $k --frames 40 2>&1 | tail -1
      done
    done > $O/rt_cli_inflight.txt
  fi
  if has build; then
    prof prof_build1m python3 tools/build_loop.py 20 708
    prof prof_build10m python3 tools/build_loop.py 10 2237
    prof prof_sah1m python3 tools/sah_loop.py
  fi
  if has pmc; then
    pmcset $O/pmc1 ""
    python3 tools/pmc_summary.py $O/pmc1 > $O/trace_pmc.txt 2>&1
    python3 tools/pmc_traffic.py $O/pmc1/d $O/pmc1/e $O/trace_traffic.json "$(date -u +%Y-%m-%d)" > $O/pmc_traffic.log 2>&1
    python3 tools/pmc_l1_json.py $O/pmc1 $O/pmc1/c $O/trace_l1_pmc.json "$(date -u +%Y-%m-%d)" > $O/pmc_l1.log 2>&1
  fi
  if has pmc10m; then
    pmcset $O/pmc10m "--preset config4"
    python3 tools/pmc_summary.py $O/pmc10m > $O/trace_pmc_10m.txt 2>&1
    python3 tools/pmc_traffic.py $O/pmc10m/d $O/pmc10m/e $O/trace_traffic_10m.json "$(date -u +%Y-%m-%d) config4" > $O/pmc_traffic_10m.log 2>&1
    python3 tools/pmc_l1_json.py $O/pmc10m $O/pmc10m/c $O/trace_l1_pmc_10m.json "$(date -u +%Y-%m-%d) config4" > $O/pmc_l1_10m.log 2>&1
  fi
  if has sort; then
    timeout -k 10 300 tools/bin/sort_yardstick gpu-raytracing_amd/csrc/librt_amd.so 708 2237 > $O/sort_yardstick.txt 2>&1
  fi
  echo collected $R: $SECTIONS
elif [ "$MODE" = publish ]; then
  cd "$(dirname "$0")/.." || exit 1
  O=gpurun_out/$R
  for f in $O/bench_*.json; do
    b=$(basename $f .json); b=${b#bench_}
    [ "$b" = profiled ] && continue
    [ -s $f ] && cp $f profiles/${R}_bench_$b.json
  done
  for k in bench build1m build10m sah1m; do
    [ -s $O/prof_${k}_kernel_stats.txt ] && cp $O/prof_${k}_kernel_stats.txt profiles/${R}_${k/bench/bench}_kernel_stats.txt
  done
  for f in trace_pmc.txt trace_pmc_10m.txt sort_yardstick.txt rt_cli_inflight.txt; do [ -s $O/$f ] && cp $O/$f profiles/${R}_$f; done
  # the objects bench.py attaches to its line (exact-workload counters): current copies, unprefixed
  for f in trace_traffic.json trace_traffic_10m.json trace_l1_pmc.json trace_l1_pmc_10m.json; do [ -s $O/$f ] && cp $O/$f profiles/$f; done
  ls profiles | grep "^${R}_"
else
  echo "usage: collect.sh run|publish <round> [section ...]"; exit 2
fi
