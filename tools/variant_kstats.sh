#!/bin/bash
# On the GPU box: rocprofv3 kernel stats (one stream) of the product library and of every build_variants/*.so; $1 = kernel name filter
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
LIB=tropical_cyclone_risk_amd/libtcrisk_hip.so
cp $LIB /tmp/product.so
for v in /tmp/product.so build_variants/*.so; do
  [ -f "$v" ] || continue
  echo "== $v"
  cp "$v" $LIB
  rm -rf /tmp/prof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gs -- python bench.py --steps 10 --warmup 2 --streams 1 --no-cpu-baseline > /dev/null 2>&1
  python tools/print_kernel_stats.py /tmp/prof | grep -E "${1:-.}" | head -${2:-6}
done
cp /tmp/product.so $LIB
