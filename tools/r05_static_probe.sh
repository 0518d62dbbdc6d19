#!/bin/bash
# Runs on the GPU box: the 100 000-storm step on the reference's real static-field shape (0.125 degree int8 land, whole-metre
# bathymetry) next to the 0.25 degree fp64 planes of rounds 1-4.  TAG names the output directory; STORE = auto | f64.
#   bash tools/r05_static_probe.sh r05_static_f64 f64
set -u
TAG=${1:-r05_static}
STORE=${2:-auto}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
for RES in 0.25 0.125; do
  for rep in 1 2; do
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --static-res $RES --static-store $STORE 2>>"$OUT/bench.err" | tail -1 > "$OUT/bench_res${RES}_$rep.json"
  done
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats$RES" -o s -- \
      python bench.py --steps 10 --warmup 2 --streams 1 --no-cpu-baseline --static-res $RES --static-store $STORE > /dev/null 2>&1
  cp "$(find "$OUT/stats$RES" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_streams1_res$RES.csv"
  rm -rf "$OUT/stats$RES"
  i=0
  mkdir -p "$OUT/pmc_res$RES"
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$OUT/pmc_res$RES/pmc$i" -o p -- \
        python bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --static-res $RES --static-store $STORE > /dev/null 2>&1
  done
  python tools/summarize_pmc.py "$OUT/pmc_res$RES" tc > "$OUT/pmc_hbm_res$RES.json"
  rm -rf "$OUT/pmc_res$RES"
done
python - <<PY
import json, glob
for f in sorted(glob.glob('$OUT/bench_res*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f, 'ms/step %.4f' % d['ms_per_step'], 'chain %.3f' % r['launch_ms'], 'frac %.3f' % r['frac'], 'simd %.3f' % r['integrate_passes']['simd_time_ms'])
    except Exception as e:
        print(f, 'failed', e)
for f in sorted(glob.glob('$OUT/pmc_hbm_res*.json')):
    d = json.load(open(f))
    for k, v in d['kernels'].items():
        if 'k_integrate' in k or 'k_seed' == k.split('::')[-1]:
            print(f, k[:40], 'hbm GB %.3f' % (v.get('hbm_bytes_per_batch', 0) / 1e9), 'l2 hit %.3f' % v.get('l2_hit_rate', 0), 'miss M %.2f' % (v.get('TCC_MISS_sum_per_batch', 0) / 1e6))
    print(f, 'step total GB %.3f' % (d['step_total']['hbm_bytes_per_batch'] / 1e9))
PY
