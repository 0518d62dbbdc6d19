#!/usr/bin/env python3
"""Fixtures that pin the oracle's *decision-forced replay* (oracle/tc_oracle.c, orc_run_ensemble_forced)
to the reference's own code.  Build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_forced.py

The reference's over-land test `f_land.ev(lon, lat) == 1` (intensity/coupled_fast.py:35-38) is decided by
rounding in the interior of land, so two implementations stop agreeing at the first evaluation where it
lands differently.  The parity checker closes that hole by re-running the C oracle with the other side's
decision sequence forced at exactly those rounding-sensitive evaluations.  That mechanism itself needs a
pin, and the only ground truth is the reference:

  natural   storms (found by scanning) on which the reference's own decisions differ from the C oracle's
            natural ones.  Forcing the *reference's* recorded decisions must make the C oracle reproduce
            the reference's track — status, n_valid, nfev, accept flags exactly, samples pointwise.
  scripted  the reference run with `_get_over_land` replaced (on the instance, ref_harness.gen_track) by
            its own test except at rounding-sensitive points, where a fixed pseudo-random script decides.
            Everything else — gen_track, dydt, solve_ivp, the interpolators — is the reference's code.
            These tracks differ from the natural ones by O(1) wherever PI != 0 over land, so a replay that
            ignored or misapplied the forced sequence cannot reproduce them.

Data only: storm inputs, the decisions the reference used, the outputs it produced.
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

from tests.golden import ref_harness as H            # noqa: E402
from tests.golden.make_golden import ENV_KW, META, N_STEPS      # noqa: E402
from tropical_cyclone_risk_amd import synthetic      # noqa: E402
from oracle import c_oracle, parity                  # noqa: E402


def script_for(salt):
    """Deterministic decision script: at a rounding-sensitive evaluation k, say `over land` unless a hash
    of (k, salt) lands in 1 of 5 — far from both the natural pattern (~98.5 % True) and its complement."""
    def script(k, own):
        return ((k * 2654435761 + salt * 40503) >> 7) % 5 != 0
    return script


def pack(basin, rows):
    n = len(rows)
    out = dict(
        lon0=np.array([r['inp']['lon'] for r in rows]), lat0=np.array([r['inp']['lat'] for r in rows]),
        v0=np.array([r['inp']['v0'] for r in rows]), m0=np.array([r['inp']['m0'] for r in rows]),
        h_bl=np.array([r['inp']['h_bl'] for r in rows]), month=np.array([r['inp']['month'] for r in rows], np.int32),
        phases=np.array([r['inp']['phases'] for r in rows]),
        status=np.array([r['res']['status'] for r in rows], np.int32),
        n_valid=np.array([r['res']['n'] for r in rows], np.int32),
        nfev=np.array([r['res']['nfev'] for r in rows], np.int32),
        is_tc=np.array([r['post']['is_tc'] for r in rows]), accepted=np.array([r['post']['accepted'] for r in rows]),
        kind=np.array([r['kind'] for r in rows]), salt=np.array([r['salt'] for r in rows], np.int64),
        basin=np.array(basin))
    traj = np.full((n, 4, N_STEPS), np.nan); envw = np.full((n, N_STEPS, 4), np.nan); vmax = np.full((n, N_STEPS), np.nan)
    for i, r in enumerate(rows):
        m = r['res']['n']
        traj[i, :, :m] = r['res']['y']; envw[i, :m] = r['post']['envw']
        vmax[i, :m] = r['post']['vmax'][:m] if m else []
    out.update(traj=traj, envw=envw, vmax=vmax)
    out['dec_off'] = np.concatenate([[0], np.cumsum([len(r['res']['dec']) for r in rows])]).astype(np.int64)
    out['dec'] = np.concatenate([r['res']['dec'] for r in rows]).astype(np.uint8)
    out['dec_t0'] = np.concatenate([r['res']['dec_t0'] for r in rows]).astype(np.float64)
    return out


def make(ref, env, basin, n_scan, seed, n_scripted, max_natural):
    S = synthetic.draw_storm_inputs(n_scan, basin, seed)
    orc = c_oracle.run_ensemble(env, basin, S, probe=True)
    exposed = np.nonzero(((orc['dec'] != 0xff) & ((orc['dec'] & 6) == 6)).any(axis=1))[0]
    print('%s: %d of %d scanned storms are flicker-exposed' % (basin, len(exposed), n_scan))
    fast = {}
    rows, n_nat, n_scr = [], 0, 0
    for i in exposed:
        inp = dict(lon=S['lon'][i], lat=S['lat'][i], month=int(S['month'][i]), v0=S['v0'][i], m0=S['m0'][i],
                   h_bl=S['h_bl'][i], phases=S['phases'][i])
        mo = inp['month'] - 1
        if mo not in fast:
            fast[mo] = H.build_coupled_fast(ref, env, basin, mo)
        f = fast[mo]
        args = (inp['lon'], inp['lat'], inp['v0'], inp['m0'], inp['h_bl'], inp['phases'])
        res = H.gen_track(ref, f, *args)
        dref = parity.ragged_to_padded(res['dec'], np.array([0, len(res['dec'])]), c_oracle.PROBE_CAP)
        if parity.first_divergence(orc['dec'][i:i + 1], dref)[0] >= 0 and n_nat < max_natural:
            rows.append(dict(inp=inp, res=res, post=H.post_track(ref, f, res), kind='natural', salt=0))
            n_nat += 1
        if n_scr < n_scripted:
            salt = int(i) + 1
            res = H.gen_track(ref, f, *args, script=script_for(salt))
            rows.append(dict(inp=inp, res=res, post=H.post_track(ref, f, res), kind='scripted', salt=salt))
            n_scr += 1
        if n_scr >= n_scripted and n_nat >= max_natural:
            break
    print('%s: kept %d natural divergences, %d scripted' % (basin, n_nat, n_scr))
    return pack(basin, rows)


def main():
    ref = H.import_reference()
    env = synthetic.make_env(**ENV_KW)
    warnings.simplefilter('ignore')
    for basin, n_scan, seed in (('NA', 1500, 4242), ('AU', 600, 4243)):
        out = make(ref, env, basin, n_scan, seed, n_scripted=14 if basin == 'NA' else 8, max_natural=12)
        out.update({'meta_' + k: np.array(v) for k, v in META.items()})
        np.savez_compressed(os.path.join(HERE, 'forced_%s.npz' % basin), **out)


if __name__ == '__main__':
    main()
