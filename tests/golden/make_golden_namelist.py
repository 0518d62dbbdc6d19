#!/usr/bin/env python3
"""Golden tracks for NON-DEFAULT namelist physics, produced by the reference's own code (build container only):

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_namelist.py

  tracks_NA_uncoupled.npz   namelist.coupled_track = False: `_calc_steering_coefs` returns the constant
                            namelist.steering_coefs = [0.2, 0.8] (intensity/coupled_fast.py:183-192, namelist.py:71-72);
  tracks_NA_physics.npz     u_beta = -1.5, v_beta = 2.0, Ck = 1.0e-3, PI_reduc = 0.9, seed_v_2d_threshold_ms = 7,
                            atm_bl_depth['NA'] + 200 m (namelist.py:56-94).  PI_reduc and Ck enter the potential intensity
                            where the fields are staged (util/compute.py:113: vmax * PI_reduc * sqrt(Ck / Cd)): the synthetic
                            vpot — which stands for that product under the default namelist — is rescaled by the ratio of the
                            two factors, and the fixture records the ratio (`vpot_scale`).

The reference's namelist module is modified in memory only (attributes set on the imported module); its files are not
touched.  The files hold data only: the storm inputs, the changed scalars (`nl_*`) and what the reference returned.
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

from tests.golden import make_golden as G            # noqa: E402
from tests.golden import ref_harness as H            # noqa: E402
from tropical_cyclone_risk_amd import synthetic      # noqa: E402


def main():
    ref = H.import_reference()
    nl = ref.namelist
    warnings.simplefilter('ignore')
    defaults = {k: getattr(nl, k) for k in ('coupled_track', 'steering_coefs', 'u_beta', 'v_beta', 'Ck', 'Cd', 'PI_reduc',
                                              'seed_v_2d_threshold_ms')}
    bl0 = dict(nl.atm_bl_depth)

    # ---- case 1: uncoupled steering
    env = synthetic.make_env(**G.ENV_KW)
    nl.coupled_track = False
    d = G.run_set(ref, env, 'NA', 260, 606, per_class=4)
    nl.coupled_track = defaults['coupled_track']
    d.update(nl_coupled_track=np.int32(0), nl_steering_coefs=np.array(defaults['steering_coefs'], dtype=np.float64))
    np.savez_compressed(os.path.join(HERE, 'tracks_NA_uncoupled.npz'), **d, **{'meta_' + k: v for k, v in G.META.items()})

    # ---- case 2: non-default physics scalars
    over = dict(u_beta=-1.5, v_beta=2.0, Ck=1.0e-3, PI_reduc=0.9, seed_v_2d_threshold_ms=7.0)
    for k, v in over.items():
        setattr(nl, k, v)
    nl.atm_bl_depth = {k: v + 200.0 for k, v in bl0.items()}
    scale = (nl.PI_reduc * np.sqrt(nl.Ck / nl.Cd)) / (defaults['PI_reduc'] * np.sqrt(defaults['Ck'] / defaults['Cd']))
    env2 = synthetic.make_env(**G.ENV_KW)
    env2.vpot = env2.vpot * scale

    def deeper(S):
        S['h_bl'] = S['h_bl'] + 200.0               # fast.h_bl = namelist.atm_bl_depth[basin] (util/compute.py:175)
    d = G.run_set(ref, env2, 'NA', 260, 707, per_class=4, mutate=deeper)
    d.update({'nl_' + k: np.float64(v) for k, v in over.items()})
    d.update(nl_atm_bl_depth_plus=np.float64(200.0), vpot_scale=np.float64(scale))
    for k, v in defaults.items():
        setattr(nl, k, v)
    nl.atm_bl_depth = bl0
    np.savez_compressed(os.path.join(HERE, 'tracks_NA_physics.npz'), **d, **{'meta_' + k: v for k, v in G.META.items()})
    for fn in ('tracks_NA_uncoupled.npz', 'tracks_NA_physics.npz'):
        g = np.load(os.path.join(HERE, fn))
        print('%-28s %7d bytes  %d storms  accepted %d  tc %d' % (fn, os.path.getsize(os.path.join(HERE, fn)), len(g['status']),
                                                                int(g['accepted'].sum()), int(g['is_tc'].sum())))


if __name__ == '__main__':
    main()
