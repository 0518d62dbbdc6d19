#!/bin/bash
# Runs on the GPU box (gpurun): everything under profiles/r06_* that comes from the product library in place.
#   bash tools/collect_r06.sh
# Timing (kernel-trace / stats) and counters (--pmc) are separate rocprofv3 runs, as MI355X_MICROARCH.md prescribes.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06c
rm -rf "$OUT"; mkdir -p "$OUT"
B="timeout 600 python bench.py"
# 1. the bench line: the driver's flags, the script's own defaults
$B --steps 20 --warmup 5 2>"$OUT/bench.err" | tail -1 > "$OUT/bench_driver_flags.json"
$B --no-cpu-baseline 2>>"$OUT/bench.err" | tail -1 > "$OUT/bench.json"
$B --steps 20 --warmup 5 --streams 1 --no-cpu-baseline 2>>"$OUT/bench.err" | tail -1 > "$OUT/bench_streams1.json"
# 2. small batches (a rank's share of a sharded 100 000-storm ensemble at N = 2 / 4 / 8)
for S in 50000 25000 12500; do
  $B --no-cpu-baseline --scaling weak --storms $S --streams 16 --steps 240 --warmup 32 2>/dev/null | tail -1 > "$OUT/bench_small_$S.json"
done
# 3. BASELINE config 5's speed claims from current code
$B --steps 20 --warmup 5 --no-cpu-baseline --dtype f32 2>>"$OUT/bench.err" | tail -1 > "$OUT/bench_f32.json"
$B --steps 20 --warmup 5 --no-cpu-baseline --dtype f32 --shape gfdl 2>>"$OUT/bench.err" | tail -1 > "$OUT/bench_f32_gfdl.json"
$B --steps 20 --warmup 5 --no-cpu-baseline --shape gfdl 2>>"$OUT/bench.err" | tail -1 > "$OUT/bench_f64_gfdl.json"
# 4. per-kernel timing, one stream and the default twelve; the 12-stream trace also feeds the timeline
for s in 1 12; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats$s" -o s -- \
      python bench.py --steps 10 --warmup 2 --streams $s --no-cpu-baseline > /dev/null 2>&1
  cp "$(find "$OUT/stats$s" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_streams$s.csv"
  [ $s = 12 ] && python tools/timeline.py "$(find "$OUT/stats$s" -name '*kernel_trace.csv' | head -1)" 6 5 > "$OUT/timeline_streams12.txt" 2>&1
  rm -rf "$OUT/stats$s"
done
# 5. HBM-side counters, one pass per counter group (single stream so dispatches do not overlap)
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$OUT/pmc$i" -o p -- \
      python bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline > /dev/null 2>&1
done
python tools/summarize_pmc.py "$OUT" tc 0.125/auto > "$OUT/pmc_hbm.json"
rm -rf "$OUT"/pmc[123]
# 6. integrator pass statistics, host CPU
timeout 300 python tools/pass_stats.py > "$OUT/integrate_pass_stats.txt" 2>&1
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" > "$OUT/host_cpu.txt"
python - <<PY
import json, glob
for f in sorted(glob.glob('$OUT/bench*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print('%-40s %.4g storm-steps/s  %.4f ms/step  chain %.3f ms  frac %.3f  step_alg %.3f' % (f.split('/')[-1], d['value'], d['ms_per_step'], r['launch_ms'], r['frac'], r['step_algorithmic']['frac']))
    except Exception as e:
        print(f, 'failed', e)
PY
cat "$OUT/timeline_streams12.txt"
# 7. (optional, `bash tools/collect_r06.sh studies`) the product surface and the studies
if [ "${1:-}" = studies ]; then
  timeout 600 python tools/run_config3.py > "$OUT/config3.json" 2>/dev/null
  timeout 600 python tools/fp32_study.py > "$OUT/fp32_study.json" 2>/dev/null
  TCR_PARITY_STUDY=20000 timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k parity_study > "$OUT/parity_study_20000.log" 2>&1
  cp gpurun_out/parity_study.json "$OUT/parity_study.json"
  tail -3 "$OUT/parity_study_20000.log"
fi
