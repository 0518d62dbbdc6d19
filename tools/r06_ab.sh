#!/bin/bash
# On the GPU box: same-box A / B of the product library against every build_variants/*.so (tools/build_variant.py):
# the integrator chain alone (pass_stats, --streams 1 bench), the 100k step at the default 12 streams, the 12 500-storm step.
#   bash tools/r06_ab.sh [REPS]
cd "${GRAFT_REPO_ROOT:-.}"
REPS=${1:-2}
LIB=tropical_cyclone_risk_amd/libtcrisk_hip.so
OUT=gpurun_out/r06
mkdir -p $OUT
cp $LIB /tmp/product.so
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
    pp = r.get('integrate_passes_pipelined') or {}
    print('$1 ms/step %.4f  chain(launch_ms) %.4f  simd_time %.4f  simd_time_pipelined %.4f  frac %.4f  value %.4g' % (d['ms_per_step'], r['launch_ms'], r['integrate_passes']['simd_time_ms'], pp.get('simd_time_ms', float('nan')), r['frac'], d['value']))
except Exception as e:
    print('$1 failed', e)
"; }
for rep in $(seq 1 $REPS); do
for v in /tmp/product.so build_variants/*.so; do
  [ -f "$v" ] || continue
  n=$(basename $v .so)
  cp "$v" $LIB
  echo "== $n (rep $rep)"
  timeout 300 python tools/pass_stats.py 2>&1 | grep -E "^integrate|^total"
  timeout 300 python bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline 2>/dev/null | line "streams1 "
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "streams12"
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | line "streams12/40"
  timeout 300 python bench.py --no-cpu-baseline --scaling weak --storms 12500 --streams 16 --steps 240 --warmup 32 2>/dev/null | line "12500x16 "
done
done 2>&1 | tee $OUT/ab.txt
cp /tmp/product.so $LIB
