"""The C-ABI library builds for gfx950, loads, and exports every symbol that
include/tcrisk_hip.h declares.  No compute calls (CPU-only container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'tcrisk_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(tcr_[a-z_0-9]+)\s*\(', text)))


def test_header_and_binding_agree():
    from tropical_cyclone_risk_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared()


def test_library_exports_every_declared_symbol(built_lib):
    L = ctypes.CDLL(built_lib)
    for name in _declared():
        assert hasattr(L, name), name
    from tropical_cyclone_risk_amd import _lib
    assert L.tcr_abi_version() == _lib.TCR_ABI_VERSION == 7


def test_struct_layout_matches_header(built_lib):
    """sizeof(tcr_params) as compiled by a C compiler == ctypes mirror."""
    import subprocess
    import tempfile
    from tropical_cyclone_risk_amd import _lib
    src = ('#include <stdio.h>\n#include "tcrisk_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",'
           'sizeof(tcr_params),sizeof(tcr_storms),sizeof(tcr_tracks),sizeof(tcr_seeds),sizeof(tcr_grid),sizeof(tcr_round),'
           'offsetof(tcr_round, tracks),offsetof(tcr_round, seed_hist));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, 'sz.c')
        open(c, 'w').write(src)
        exe = os.path.join(d, 'sz')
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    mine = [ctypes.sizeof(t) for t in (_lib.Params, _lib.Storms, _lib.Tracks, _lib.Seeds, _lib.Grid, _lib.Round)]
    mine += [_lib.Round.tracks.offset, _lib.Round.seed_hist.offset]
    assert sizes == mine


def test_integration_stub_is_complete():
    """INTEGRATION.md's reference-side binding: the code block compiles, every name it uses is defined in it
    (imported, assigned, a parameter, a builtin), every library function it calls is declared in the header, and its
    ctypes mirrors have the sizes of the package's checked mirrors."""
    import ast
    import builtins
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    code = text.split('```python')[1].split('```')[0]
    tree = ast.parse(code)
    defined = set(dir(builtins))
    for node in ast.walk(tree):
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            defined |= {(a.asname or a.name).split('.')[0] for a in node.names}
        elif isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            defined.add(node.name)
            if isinstance(node, ast.FunctionDef):
                defined |= {a.arg for a in node.args.args + node.args.kwonlyargs}
        elif isinstance(node, ast.Lambda):
            defined |= {a.arg for a in node.args.args}
        elif isinstance(node, ast.Name) and isinstance(node.ctx, ast.Store):
            defined.add(node.id)
        elif isinstance(node, ast.comprehension):
            defined |= {n.id for n in ast.walk(node.target) if isinstance(n, ast.Name)}
    used = {n.id for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)}
    assert not (used - defined), sorted(used - defined)
    called = set(re.findall(r'_L\.(tcr_[a-z_0-9]+)', code))
    assert called and called <= set(_declared()), sorted(called - set(_declared()))
    # struct mirrors: same sizes as the package's (which are checked against the C compiler's sizeof)
    ns = {}
    mirrors = code.split('def _f64')[0].replace("_L = C.CDLL('libtcrisk_hip.so')", '_L = None').replace('_L.tcr_last_error', '# ')
    mirrors = '\n'.join(l for l in mirrors.splitlines() if not l.startswith(('import xarray', 'import namelist', 'from ')))
    exec(compile(mirrors, 'INTEGRATION.md', 'exec'), ns)
    from tropical_cyclone_risk_amd import _lib
    for a, b in (('Grid', _lib.Grid), ('Params', _lib.Params), ('Storms', _lib.Storms), ('Tracks', _lib.Tracks), ('Seeds', _lib.Seeds)):
        assert ctypes.sizeof(ns[a]) == ctypes.sizeof(b), a
        assert [f[0] for f in ns[a]._fields_] == [f[0] for f in b._fields_], a


def test_no_gpu_fails_loudly(built_lib):
    """Without a HIP device the product path must raise, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from tropical_cyclone_risk_amd import _lib
    from tropical_cyclone_risk_amd.engine import TCEngine
    with pytest.raises(_lib.TcrError) as e:
        TCEngine('NA')
    assert 'no CPU fallback' in str(e.value) or 'HIP' in str(e.value)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'tropical_cyclone_risk_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, fn)).read()
                assert 'oracle' not in re.sub(r'#.*|//.*', '', text).replace('"oracle"', ''), (dirpath, fn)


def test_entry_scripts_compile():
    """bench.py / run.py / __graft_entry__.py are only executed on the GPU box: a syntax error in them must not wait for it."""
    import py_compile
    for f in ('bench.py', 'run.py', '__graft_entry__.py'):
        py_compile.compile(os.path.join(ROOT, f), doraise=True)
    for f in os.listdir(os.path.join(ROOT, 'tools')):
        if f.endswith('.py'):
            py_compile.compile(os.path.join(ROOT, 'tools', f), doraise=True)


def test_kernel_register_budgets():
    """The resident-next-to-the-integrator contract of round 4: an integrator wave leaves 88 registers on its
    SIMD, so every kernel of a batch other than the integrator and the forcing-table GEMM must need at most that many
    (VGPRs + AGPRs, in the allocation granule of 8), and the fp64 integrator must not spill.  Read from the compiler's own
    resource report (hipcc -Rpass-analysis=kernel-resource-usage: no GPU needed), so that a compiler upgrade that changes
    the trade shows up here and not as a slower bench."""
    import subprocess
    from tropical_cyclone_risk_amd import build as B
    cmd = [B.hipcc()] + B.FLAGS + ['-Rpass-analysis=kernel-resource-usage', '-o', os.devnull, os.path.join(B.CSRC, 'tcr_abi.hip')]
    out = subprocess.run(cmd, cwd=B.CSRC, capture_output=True, text=True).stderr
    rows, cur = {}, None
    for line in out.splitlines():
        m = re.search(r'remark: \S+ +(Function Name|Name): (\S+)', line)
        if m:
            cur = rows.setdefault(m.group(2), {})
            continue
        m = re.search(r'(VGPRs|AGPRs|ScratchSize \[bytes/lane\]): (\d+)', line)
        if m and cur is not None:
            cur[m.group(1).split()[0]] = int(m.group(2))
    assert len(rows) > 40, 'no resource report from hipcc'
    small = ('k_seed', 'k_compact', 'k_gather_seeds', 'k_cell_key', 'k_cell_scan', 'k_cell_scatter', 'k_cell_rank', 'k_phase_factors_frag',
             'k_batch_reset', 'k_screenId', 'k_denseId', 'k_emitId', 'k_flagsId', 'k_pack_tracksId', 'k_seed_hist', 'k_stats')
    seen = set()
    for name, r in rows.items():
        if 'k_integrateIdLb' in name:
            assert r.get('ScratchSize', 0) == 0, (name, r)            # the fp64 integrator owns its SIMD's registers without spilling
        for s in small:
            if ('3tcr' + str(len(s.replace('Id', ''))) + s.replace('Id', '') in name) and (not s.endswith('Id') or 'Id' in name.split(s.replace('Id', ''))[1][:3]):
                regs = r.get('VGPRs', 0) + r.get('AGPRs', 0)
                assert -(-regs // 8) * 8 <= 88, (name, r)
                seen.add(s)
    assert seen >= set(small) - {'k_batch_reset'}, sorted(set(small) - seen)
    # The 80-register budgets are bought with a little scratch (measured as a net win).  The amounts
    # are pinned so that a compiler upgrade that starts spilling in earnest shows up here: k_seed <= 56 B per lane,
    # k_screen<double> <= 20 B, and the production k_emit (fp64, affine grids, one list entry per workgroup) none at all.
    ceilings = {'_ZN3tcr6k_seedE': 56, '_ZN3tcr8k_screenIdE': 20, '_ZN3tcr6k_emitIdLb1ELb0EE': 0}
    for prefix, cap in ceilings.items():
        hit = [(n, r) for n, r in rows.items() if n.startswith(prefix)]
        assert len(hit) == 1, (prefix, [n for n, _ in hit])
        assert hit[0][1].get('ScratchSize', 0) <= cap, hit[0]


def test_product_library_reads_the_environment_once():
    """no per-launch getenv in the product build.  The launch-shape knobs are read once, in tcr_ctx_create (one
    helper), and changed through tcr_tune_set; the scheduling probes and per-call knobs of the experiments live in
    csrc/tcr_experiments.h, which only -DTCR_EXPERIMENTS builds include."""
    from tropical_cyclone_risk_amd import build as B
    n = sum(l.count('getenv') for l in open(os.path.join(B.CSRC, 'tcr_abi.hip')) if not l.lstrip().startswith('//'))
    assert n <= 2, n
    for f in ('tcr_kernels.hip', 'tcr_seed.hip', 'tcr_compact.hip', 'tcr_prep.hip', 'tcr_thermo.hip', 'tcr_device.h'):
        assert 'getenv' not in open(os.path.join(B.CSRC, f)).read(), f
    assert 'TCR_EXPERIMENTS' not in ' '.join(B.FLAGS)
