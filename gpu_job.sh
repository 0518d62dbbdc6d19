cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r01
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" > gpurun_out/r01/host_cpu.txt
python bench.py --steps 20 --warmup 3 > gpurun_out/r01/bench.json 2> gpurun_out/r01/bench.err
tail -c 3000 gpurun_out/r01/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01/stats -o run -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r01/stats_bench.json 2>/dev/null
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/r01/pmc_fetch -o run -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d gpurun_out/r01/pmc_write -o run -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d gpurun_out/r01/pmc_l2 -o run -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
find gpurun_out/r01 -type f | head -30
