cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc4 gpurun_out/pmc5
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d gpurun_out/pmc4 -o run -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --storms 10000 > gpurun_out/pmc4/log.txt 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM -d gpurun_out/pmc5 -o run -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --storms 10000 > gpurun_out/pmc5/log.txt 2>&1
