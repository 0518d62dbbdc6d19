#!/bin/bash
# Build the timing-experiment variants of DESIGN.md §9 into build_variants/ (here, no GPU needed); then on the GPU box:
#   NOBENCH=1 bash tools/variant_stats.sh > gpurun_out/ablations.txt
set -e
cd "$(dirname "$0")/.."
rm -f build_variants/*.so
python tools/build_variant.py a_no_reads        -DTCR_ABLATE_ALL_READS
python tools/build_variant.py b_uniform_address -DTCR_ABLATE_UNIFORM_ADDR
python tools/build_variant.py c_no_fs_read      -DTCR_ABLATE_FS_READ
python tools/build_variant.py d_no_step_stores  -DTCR_ABLATE_STEP_STORES
python tools/build_variant.py e_pipelined       -DTCR_INT_PIPELINE=1
python tools/build_variant.py f_kernarg_consts  -DTCR_K_KERNARG
python tools/build_variant.py g_phase_clocks_wait   -DTCR_INT_PHASE_CLOCKS=1
python tools/build_variant.py h_phase_clocks_nowait -DTCR_INT_PHASE_CLOCKS=2
python tools/build_variant.py i_phase_clocks_pipelined -DTCR_INT_PHASE_CLOCKS=2 -DTCR_INT_PIPELINE=1
python tools/build_variant.py j_machine_licm    -mllvm -disable-machine-licm=false
ls -la build_variants
