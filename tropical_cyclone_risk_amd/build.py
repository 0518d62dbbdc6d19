"""Build libtcrisk_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m tropical_cyclone_risk_amd.build [--force]

-ffp-contract=off: bilinear weights/sums and the `land == 1` test must keep
FITPACK's operation order without fused multiply-adds; the one place that opts in to
contraction (the RK45 step of k_integrate) does so with a pragma (tcr_device.h, "Arithmetic policy").
-mllvm -disable-machine-licm: see FLAGS.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
OUT = os.path.join(PKG, 'libtcrisk_hip.so')
SOURCES = ['tcr_abi.hip', 'tcr_kernels.hip', 'tcr_seed.hip', 'tcr_compact.hip', 'tcr_prep.hip', 'tcr_thermo.hip', 'tcr_comm.hip', 'tcr_device.h', 'tcr_experiments.h',
           os.path.join('..', '..', 'include', 'tcrisk_hip.h')]
# -disable-machine-licm: the kernels here are register-bound loops around libm-heavy bodies; hoisting the bodies' constant
# materialisations out of the loops costs k_emit 40 VGPRs + spills (0.36 instead of 0.15 ms) and k_integrate 70 AGPRs.
# -DTCR_K_KERNARG: k_integrate reads its ~100 evaluation constants with scalar loads from the kernel arguments instead of from the
# workgroup's LDS copy behind an opaque offset (-DTCR_OPAQUE_K, rounds 2-5: needed while LLVM hoisted them into registers and the
# kernel spilled; without machine LICM neither form spills).  Round 6, same box: chain -3 %, 100 000-storm step -1.5 %, bit-identical.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-DTCR_K_KERNARG', '-mllvm', '-disable-machine-licm',
         '-fPIC', '-shared']


def hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found; libtcrisk_hip.so cannot be built (no CPU fallback exists)')


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return os.path.getmtime(os.path.abspath(__file__)) > t or any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES)


def build(force=False, verbose=False):
    if not (force or stale()):
        return OUT
    extra = os.environ.get('TCR_HIPCC_FLAGS', '').split()      # tuning experiments, e.g. -DTCR_EMIT_WPS=4
    cmd = [hipcc()] + FLAGS + extra + ['-o', OUT, os.path.join(CSRC, 'tcr_abi.hip')]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
