#!/bin/bash
# On the GPU box: integrator pass statistics (and a short bench) of the product library and of every build_variants/*.so
cd "${GRAFT_REPO_ROOT:-.}"
LIB=tropical_cyclone_risk_amd/libtcrisk_hip.so
cp $LIB /tmp/product.so
for v in /tmp/product.so build_variants/*.so; do
  [ -f "$v" ] || continue
  echo "== $v"
  cp "$v" $LIB
  python tools/pass_stats.py 2>&1 | grep -E "integrate|pass  0|total|phase" | sort | uniq -c | sort -rn | head -8
  [ -n "${NOBENCH:-}" ] || python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('bench ms/step %.3f'%d['ms_per_step'], r['kernel_ms'], 'simd %.3f'%r['integrate_passes']['simd_time_ms'])"
done
cp /tmp/product.so $LIB
