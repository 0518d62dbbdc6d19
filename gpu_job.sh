cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('100k', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['accepted_fraction'])"
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --storms 10000 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('10k', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
