#!/bin/bash
# Rounds 2-4 (round 5: tools/collect_r05.sh + tools/r05_static_probe.sh).
# Runs on the GPU box (gpurun): regenerates everything under profiles/ for one round tag.
#   bash tools/collect_profiles.sh r02 [tc|all]
# Timing (kernel-trace/stats) and counters (--pmc) are separate rocprofv3 runs, as MI355X_MICROARCH.md prescribes.
set -u
TAG=${1:-r04}
ROWS=${2:-tc}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
# 1. the bench line itself: with the driver's flags (--steps 20 --warmup 5) and with the script's own defaults (40 steps)
timeout 900 python bench.py --rows $ROWS --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2> "$OUT/bench.err"
timeout 900 python bench.py --rows $ROWS --no-cpu-baseline > "$OUT/bench.json" 2>> "$OUT/bench.err"
tail -c 600 "$OUT/bench.json"
# 1b. small batches (a rank's share of a sharded 100 000-storm ensemble at N = 2 / 4 / 8) and where a batch's time goes under load
for B in 50000 25000 12500; do
  timeout 600 python bench.py --no-cpu-baseline --scaling weak --storms $B --streams 16 --steps 240 --warmup 32 2>/dev/null | tail -1 > "$OUT/bench_small_$B.json"
done
for cfg in "12500 1 60" "12500 4 120" "12500 16 240" "100000 1 10" "100000 12 48"; do
  set -- $cfg
  timeout 600 python bench.py --no-cpu-baseline --scaling weak --storms $1 --streams $2 --steps $3 --warmup 8 --stage-trace 2>/dev/null | tail -1 > "$OUT/stage_trace_${1}_streams$2.json"
done
# 2. per-kernel timing, one stream and the default twelve
for s in 1 12; do
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
