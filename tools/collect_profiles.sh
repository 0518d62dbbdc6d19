#!/bin/bash
# Runs on the GPU box (gpurun): regenerates everything under profiles/ for one round tag.
#   bash tools/collect_profiles.sh r02 [tc|all]
# Timing (kernel-trace/stats) and counters (--pmc) are separate rocprofv3 runs, as MI355X_MICROARCH.md prescribes.
set -u
TAG=${1:-r03}
ROWS=${2:-tc}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
# 1. the bench line itself (default flags, as the driver runs it)
timeout 900 python bench.py --rows $ROWS > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench.json"
# 2. per-kernel timing, one stream and the default eight
for s in 1 8; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats$s" -o s -- \
      python bench.py --steps 10 --warmup 2 --streams $s --rows $ROWS --no-cpu-baseline > /dev/null 2>&1
  cp "$(find "$OUT/stats$s" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_streams$s.csv"
done
# 3. HBM-side counters, one pass per counter group (single stream so dispatches do not overlap)
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$OUT/pmc$i" -o p -- \
      python bench.py --steps 3 --warmup 1 --streams 1 --rows $ROWS --no-cpu-baseline > /dev/null 2>&1
done
python tools/summarize_pmc.py "$OUT" $ROWS > "$OUT/pmc_hbm.json"
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" > "$OUT/host_cpu.txt"
timeout 300 python tools/pass_stats.py > "$OUT/pass_stats.txt" 2>&1
ls -la "$OUT"
