# L1 (TCP) / TA pressure of the trace kernel (2 counters per pass: TA / TCP blocks have few slots)
export TMPDIR=/tmp
B="python3 bench.py --inflight 1 --steps 3 --warmup 1 --no-extras --no-cpu-baseline"   # one launch at a time: clean per-launch counters
O=$GRAFT_REPO_ROOT/gpurun_out/pmcl1
i=0
for C in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --kernel-include-regex trace_kernel --output-format csv -d $O/p$i -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmcl1_$i.log 2>&1 || echo "pass $i failed"
done
echo done
