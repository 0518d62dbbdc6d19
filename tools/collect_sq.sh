#!/bin/bash
# Runs on the GPU box (gpurun): SQ / TA / TCP / TD / GRBM counters of the bench step, per kernel -> gpurun_out/r06/sq_<TAG>/ and the
# summary gpurun_out/r06/sq_summary_<TAG>.json (tools/summarize_sq.py).  Counters are collected in their own runs (--pmc with
# --kernel-trace only), <= 8 SQ + 2 GRBM counters per pass; TA / TCP / TD counters two per pass (larger groups hang rocprofv3 on
# this box until the timeout).  rocprofv3 serialises the dispatches while it collects, so one stream count is enough for the
# per-kernel figures; the uninstrumented step of the same command at 1 and 12 streams is timed next to them.
#   bash tools/collect_sq.sh [TAG] [quick]      TAG names the output (default "product"); the library in place is the one profiled
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${1:-product}
QUICK=${2:-}
OUT=gpurun_out/r06/sq_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
[ -f "$OUT/../counters_available.txt" ] || rocprofv3 -L > "$OUT/../counters_available.txt" 2>&1 || true
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE GRBM_COUNT"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
 "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_FLAT"
 "TA_TA_BUSY_sum TA_BUSY_avr"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
 "TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
 "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"
)
[ -n "$QUICK" ] && PASSES=("${PASSES[@]:0:1}")
s=1
i=0
for C in "${PASSES[@]}"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$OUT/s${s}_p$i" -o p -- \
      python bench.py --steps 3 --warmup 1 --streams $s --no-cpu-baseline > "$OUT/s${s}_p$i.log" 2>&1
  echo "pass $i ($C) rc $?" >> "$OUT/passes.txt"
done
for s in 1 12; do
  timeout 300 python bench.py --steps 20 --warmup 5 --streams $s --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_streams$s.json"
done
python tools/summarize_sq.py "$OUT" > "$OUT/../sq_summary_$TAG.json"
find "$OUT" -name '*.csv' -size +2M -delete
du -sh "$OUT"
