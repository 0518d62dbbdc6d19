cd $GRAFT_REPO_ROOT
for st in 4 6 8; do
timeout 900 python bench.py --steps 24 --warmup 3 --no-cpu-baseline --streams $st 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $st 100k', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
for st in 4 8; do
timeout 900 python bench.py --steps 24 --warmup 3 --no-cpu-baseline --storms 10000 --streams $st 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $st 10k', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
