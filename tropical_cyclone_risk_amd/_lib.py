"""ctypes binding of ``libtcrisk_hip.so`` (C ABI declared in ``include/tcrisk_hip.h``).

There is deliberately no CPU fallback: if the shared library is missing the import
of :func:`lib` raises with build instructions, and if no HIP device is visible
``tcr_ctx_create`` fails with the driver's message.
"""
import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, 'libtcrisk_hip.so')

TCR_ABI_VERSION = 7
TCR_NW, TCR_NCOV, TCR_MAX_SERIES, TCR_N_BASINS = 4, 10, 32, 7
STATUS_GATED, STATUS_FINISHED, STATUS_EVENT, STATUS_STEP_FAIL, STATUS_STEP_OVERFLOW = -1, 0, 1, -2, -3
FLAG_IS_TC, FLAG_ACCEPTED = 1, 2
STAGES = ('start', 'seed', 'select', 'order', 'gather', 'fourier', 'integrate', 'screen', 'select_tc', 'dense', 'emit', 'flags', 'stats', 'pack')
STATIC_MODES = ('f64', 'f64_split', 'pack16', 'u8_f32', 'pack64')     # StaticMode (tcr_static_info)
N_STATS = 10        # TCR_N_STATS: words of a tcr_stats_dev / tcr_round.stats counter block

DP = C.POINTER(C.c_double)
IP = C.POINTER(C.c_int32)
U8P = C.POINTER(C.c_uint8)

# every symbol include/tcrisk_hip.h declares (checked by tests/test_abi.py)
EXPORTS = ('tcr_abi_version', 'tcr_ctx_create', 'tcr_ctx_destroy', 'tcr_last_error',
           'tcr_params_set', 'tcr_static_upload', 'tcr_fields_upload', 'tcr_rh_upload', 'tcr_masks_upload',
           'tcr_integrate_host', 'tcr_integrate_dev', 'tcr_seed_dev', 'tcr_seed_host',
           'tcr_probe_rhs_host', 'tcr_fourier_table_host', 'tcr_timing_enable',
           'tcr_timing_last', 'tcr_timing_sum', 'tcr_sync', 'tcr_compact_dev',
           'tcr_gather_seeds_dev', 'tcr_pack_tracks_dev', 'tcr_stats_dev', 'tcr_integrate_pass_stats', 'tcr_wind_stats_dev', 'tcr_wind_stats_host', 'tcr_entropy_table_upload',
           'tcr_potential_intensity_host', 'tcr_potential_intensity_dev', 'tcr_chi_rh_host',
           'tcr_integrate_probe_host', 'tcr_integrate_f32_dev', 'tcr_integrate_f32_host', 'tcr_pack_tracks_f32_dev',
           'tcr_wind_stats_f32_dev', 'tcr_wind_stats_f32_host', 'tcr_static_upload2', 'tcr_init_m_dev', 'tcr_init_m_host', 'tcr_cell_order_dev',
           'tcr_round_dev', 'tcr_round_graph_stats', 'tcr_schedule_set', 'tcr_stage_trace_enable', 'tcr_stage_trace_sum', 'tcr_seed_hist_dev', 'tcr_pack_tracks_meta_dev',
           'tcr_static_store', 'tcr_static_info', 'tcr_tune_set', 'tcr_tune_get', 'tcr_slot_upload', 'tcr_stage_timing',
           'tcr_probe_math_host', 'tcr_comm_unique_id', 'tcr_comm_create', 'tcr_comm_destroy', 'tcr_comm_rank', 'tcr_comm_world', 'tcr_allgather_dev',
           'tcr_allgather_rows_dev', 'tcr_allgather_counts_dev', 'tcr_allreduce_sum_i64_dev', 'tcr_concat_rows_dev')
TCR_COMM_ID_BYTES = 128


class Grid(C.Structure):
    _fields_ = [('nlon', C.c_int32), ('nlat', C.c_int32), ('lon', DP), ('lat', DP)]


class Params(C.Structure):
    _fields_ = [('Ck', C.c_double), ('epsilon', C.c_double), ('kappa', C.c_double),
                ('u_beta', C.c_double), ('v_beta', C.c_double), ('T_Fs', C.c_double),
                ('y_alpha', C.c_double * 2), ('m_alpha', C.c_double * 2),
                ('alpha_max', C.c_double * 2), ('alpha_min', C.c_double * 2),
                ('steering_coefs', C.c_double * 2),
                ('dt_out', C.c_double), ('total_time', C.c_double),
                ('rtol', C.c_double), ('atol', C.c_double), ('max_step', C.c_double),
                ('v_thresh', C.c_double), ('v_2d_thresh', C.c_double), ('vmax_thresh', C.c_double),
                ('v_dissipate', C.c_double), ('earth_R', C.c_double), ('box', C.c_double * 4),
                ('fs_amp', C.c_double), ('fs_wgt', C.c_double * TCR_MAX_SERIES),
                ('n_series', C.c_int32), ('n_steps', C.c_int32),
                ('coupled_track', C.c_int32), ('max_rk_steps', C.c_int32),
                ('seed_v_init', C.c_double), ('pi_gate', C.c_double), ('lat_vort_fac', C.c_double),
                ('lat_vort_power', C.c_double * TCR_N_BASINS),
                ('atm_bl_depth', C.c_double * TCR_N_BASINS),
                ('minit_a', C.c_double), ('minit_b', C.c_double),
                ('minit_c', C.c_double), ('minit_d', C.c_double)]


class Storms(C.Structure):
    _fields_ = [('n', C.c_int64), ('lon0', C.c_void_p), ('lat0', C.c_void_p), ('v0', C.c_void_p),
                ('m0', C.c_void_p), ('h_bl', C.c_void_p), ('slot', C.c_void_p), ('phases', C.c_void_p),
                ('n_dev', C.c_void_p)]


class Tracks(C.Structure):
    _fields_ = [('lon', C.c_void_p), ('lat', C.c_void_p), ('v', C.c_void_p), ('m', C.c_void_p),
                ('vmax', C.c_void_p), ('envw', C.c_void_p), ('n_valid', C.c_void_p),
                ('status', C.c_void_p), ('flags', C.c_void_p), ('nfev', C.c_void_p),
                ('n_accept', C.c_void_p), ('n_reject', C.c_void_p), ('pad_state', C.c_void_p),
                ('tc_rows_only', C.c_int32)]


class Seeds(C.Structure):
    _fields_ = [('n', C.c_int64), ('lon0', C.c_void_p), ('lat0', C.c_void_p), ('v0', C.c_void_p),
                ('m0', C.c_void_p), ('h_bl', C.c_void_p), ('slot', C.c_void_p),
                ('phases', C.c_void_p), ('basin_idx', C.c_void_p), ('seed_flags', C.c_void_p)]


class Round(C.Structure):
    """tcr_round: one round of the accept loop (tcr_round_dev)."""
    _fields_ = [('n_cand', C.c_int64), ('n_storms', C.c_int64), ('cand', Seeds), ('storms', Seeds),
                ('cand_idx', C.c_void_p), ('n_passed', C.c_void_p), ('cell_deg', C.c_double),
                ('exact_count', C.c_int32), ('f32', C.c_int32), ('tracks', Tracks),
                ('stats', C.c_void_p), ('acc_idx', C.c_void_p), ('n_accepted', C.c_void_p),
                ('packed', C.c_void_p), ('pack_cap', C.c_int64), ('pack_stride', C.c_int64), ('seed_hist', C.c_void_p),
                ('n_expected', C.c_int64)]


class Tune(C.Structure):
    """tcr_tune: launch-shape knobs (negative = the library's choice)."""
    _fields_ = [(k, C.c_int32) for k in ('waves', 'park', 'park_final', 'table_segments', 'prune', 'emit_grid_cap', 'copy_threads', 'reserved')]


class TcrError(RuntimeError):
    pass


_lib = None


def _pin_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64;
    libtcrisk_hip.so is linked against the system ROCm.  If this library initialised the system
    runtime first and torch then brought up its bundled one, the second bring-up fails ("No HIP GPUs
    are available").  Both have the same SONAME, so loading torch's copies first (when torch is
    installed but not imported yet) makes the dynamic loader resolve our dependency to them too."""
    import importlib.util
    import sys
    if 'torch' in sys.modules:
        return
    try:
        spec = importlib.util.find_spec('torch')
    except Exception:
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), 'lib')
    for name in ('libhsa-runtime64.so', 'libamdhip64.so'):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def lib():
    """Load the shared library (once).  Fails loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TcrError(
            'libtcrisk_hip.so not found at %s — build it with '
            '`python -c "import __graft_entry__ as g; g.build()"` or '
            '`python -m tropical_cyclone_risk_amd.build`; there is no CPU fallback.' % LIB_PATH)
    _pin_hip_runtime()
    L = C.CDLL(LIB_PATH)
    L.tcr_abi_version.restype = C.c_int
    L.tcr_last_error.restype = C.c_char_p
    L.tcr_last_error.argtypes = [C.c_void_p]
    L.tcr_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.tcr_ctx_destroy.argtypes = [C.c_void_p]
    L.tcr_params_set.argtypes = [C.c_void_p, C.POINTER(Params)]
    L.tcr_static_upload.argtypes = [C.c_void_p, C.POINTER(Grid), DP, DP]
    L.tcr_static_upload2.argtypes = [C.c_void_p, C.POINTER(Grid), DP, C.POINTER(Grid), DP]
    L.tcr_stage_timing.argtypes = [C.c_void_p, DP, C.c_int32]
    L.tcr_tune_set.argtypes = [C.c_void_p, C.POINTER(Tune)]
    L.tcr_tune_get.argtypes = [C.c_void_p, C.POINTER(Tune)]
    L.tcr_static_store.argtypes = [C.c_void_p, C.c_int32]
    L.tcr_static_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    L.tcr_init_m_dev.argtypes = [C.c_void_p, C.POINTER(Storms), C.c_double, C.c_void_p, C.c_void_p]
    L.tcr_init_m_host.argtypes = [C.c_void_p, C.POINTER(Storms), C.c_double, DP]
    L.tcr_cell_order_dev.argtypes = [C.c_void_p, C.POINTER(Seeds), C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_void_p]
    L.tcr_fields_upload.argtypes = [C.c_void_p, C.c_int, C.POINTER(Grid), C.POINTER(DP), C.POINTER(DP),
                                    C.POINTER(Grid), DP, DP, DP, DP]
    L.tcr_rh_upload.argtypes = [C.c_void_p, C.c_int, C.POINTER(Grid), DP]
    L.tcr_slot_upload.argtypes = [C.c_void_p, C.c_int, C.POINTER(Grid), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(Grid)] + [C.c_void_p] * 4 + [C.POINTER(Grid), C.c_void_p]
    L.tcr_masks_upload.argtypes = [C.c_void_p, C.POINTER(Grid), U8P, C.POINTER(U8P)]
    L.tcr_integrate_host.argtypes = [C.c_void_p, C.POINTER(Storms), C.POINTER(Tracks)]
    L.tcr_integrate_probe_host.argtypes = [C.c_void_p, C.POINTER(Storms), C.POINTER(Tracks), C.c_void_p, C.c_int32]
    L.tcr_integrate_f32_dev.argtypes = [C.c_void_p, C.POINTER(Storms), C.POINTER(Tracks), C.c_void_p]    # tcr_tracks_f32 has tcr_tracks' layout
    L.tcr_integrate_f32_host.argtypes = [C.c_void_p, C.POINTER(Storms), C.POINTER(Tracks)]
    L.tcr_integrate_dev.argtypes = [C.c_void_p, C.POINTER(Storms), C.POINTER(Tracks), C.c_void_p]
    L.tcr_seed_dev.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_int64, C.POINTER(Seeds), C.c_void_p]
    L.tcr_seed_host.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_int64, C.POINTER(Seeds)]
    L.tcr_probe_rhs_host.argtypes = [C.c_void_p, C.c_int, C.c_double, DP, C.c_int64,
                                     DP, DP, DP, DP, DP, DP, DP, DP]
    L.tcr_fourier_table_host.argtypes = [C.c_void_p, C.c_int64, DP, DP]
    L.tcr_timing_enable.argtypes = [C.c_void_p, C.c_int]
    L.tcr_timing_last.argtypes = [C.c_void_p, DP]
    L.tcr_sync.argtypes = [C.c_void_p, C.c_void_p]
    L.tcr_timing_sum.argtypes = [C.c_void_p, DP, C.POINTER(C.c_int64)]
    L.tcr_integrate_pass_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int]
    L.tcr_wind_stats_dev.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.c_void_p, C.c_int32,
                                     C.c_void_p, C.c_void_p]
    L.tcr_entropy_table_upload.argtypes = [C.c_void_p, C.c_int32, C.c_int32, DP, DP, DP]
    L.tcr_potential_intensity_host.argtypes = [C.c_void_p, C.c_int64, C.c_int32, DP, DP, DP, DP, DP, C.c_double, DP]
    L.tcr_potential_intensity_dev.argtypes = [C.c_void_p, C.c_int64, C.c_int32] + [C.c_void_p] * 5 + [C.c_double, C.c_void_p, C.c_void_p]
    L.tcr_chi_rh_host.argtypes = [C.c_void_p, C.c_int64, DP, DP, DP, DP, C.c_double, DP, DP]
    L.tcr_wind_stats_host.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.c_void_p, C.c_int32,
                                      C.c_void_p]
    L.tcr_compact_dev.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int64,
                                  C.c_void_p, C.c_void_p, C.c_void_p]
    L.tcr_gather_seeds_dev.argtypes = [C.c_void_p, C.POINTER(Seeds), C.c_void_p, C.c_int64, C.c_void_p,
                                       C.POINTER(Seeds), C.c_uint64, C.c_int32, C.c_int64, C.c_void_p]
    L.tcr_stats_dev.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(Tracks), C.c_void_p, C.c_int32, C.c_void_p]
    L.tcr_round_dev.argtypes = [C.c_void_p, C.POINTER(Round), C.c_uint64, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]
    L.tcr_schedule_set.argtypes = [C.c_void_p, C.c_int32]
    L.tcr_stage_trace_enable.argtypes = [C.c_void_p, C.c_int]
    L.tcr_stage_trace_sum.argtypes = [C.c_void_p, DP, C.POINTER(C.c_int64)]
    L.tcr_round_graph_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.tcr_seed_hist_dev.argtypes = [C.c_void_p, C.POINTER(Seeds), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tcr_pack_tracks_meta_dev.argtypes = [C.c_void_p, C.POINTER(Tracks), C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                           C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.tcr_pack_tracks_dev.argtypes = [C.c_void_p, C.POINTER(Tracks), C.c_void_p, C.c_void_p,
                                      C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    L.tcr_pack_tracks_f32_dev.argtypes = L.tcr_pack_tracks_dev.argtypes
    L.tcr_wind_stats_f32_dev.argtypes = L.tcr_wind_stats_dev.argtypes
    L.tcr_wind_stats_f32_host.argtypes = L.tcr_wind_stats_host.argtypes
    L.tcr_probe_math_host.argtypes = [C.c_void_p, C.c_int32, C.c_int64, DP, DP, DP]
    L.tcr_comm_unique_id.argtypes = [U8P]
    L.tcr_comm_create.argtypes = [C.c_void_p, U8P, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.tcr_comm_destroy.argtypes = [C.c_void_p]
    L.tcr_comm_rank.argtypes = [C.c_void_p]
    L.tcr_comm_world.argtypes = [C.c_void_p]
    L.tcr_allgather_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.tcr_allgather_rows_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.tcr_allgather_counts_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tcr_allreduce_sum_i64_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.tcr_concat_rows_dev.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    if L.tcr_abi_version() != TCR_ABI_VERSION:
        raise TcrError('libtcrisk_hip.so ABI version %d != binding version %d'
                       % (L.tcr_abi_version(), TCR_ABI_VERSION))
    _lib = L
    return L
