"""Probe: does work on OTHER threads (allocations, field staging, direct rounds) disturb a stream capture (tcr_round_dev
use_graph) on this thread, or the other way round?  Prints the errors seen per activity."""
import os, sys, threading, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tropical_cyclone_risk_amd import _lib, synthetic
from tropical_cyclone_risk_amd.engine import TCEngine
from tropical_cyclone_risk_amd.pipeline import DevicePipeline

env = synthetic.make_env('era5')
stop = False
errs = {'capture': [], 'stage': [], 'alloc': [], 'direct': []}
counts = {'capture': 0, 'stage': 0, 'alloc': 0, 'direct': 0, 'graphs': 0}


def capturer():
    try:
        dev = torch.device('cuda', 0)
        with torch.cuda.stream(torch.cuda.Stream(dev)):
            eng = TCEngine('GL', device=0).stage_env(env)
            p = DevicePipeline(eng, 8192, 2048, sort_storms=2.0, tc_rows_only=True)
            stats = [torch.zeros(_lib.N_STATS, dtype=torch.int64, device=dev) for _ in range(40)]
            k = 0
            while not stop:
                try:
                    if k % 2 == 0:
                        eng.schedule(1 + (k // 2) % 2)          # bumps the context's epoch: the next round is captured again
                    p.round(2000, 8192 * k, stats=stats[k % 40], graph=True)
                    torch.cuda.current_stream().synchronize()
                    counts['capture'] += 1
                except Exception as e:
                    errs['capture'].append(repr(e)[:200])
                k += 1
            counts['graphs'] = p.graph_stats()
    except Exception:
        errs['capture'].append(traceback.format_exc()[-400:])


def stager():
    try:
        eng = TCEngine('GL', device=0)
        while not stop:
            try:
                eng.stage_env(env); counts['stage'] += 1
            except Exception as e:
                errs['stage'].append(repr(e)[:200])
    except Exception:
        errs['stage'].append(traceback.format_exc()[-400:])


def allocator():
    dev = torch.device('cuda', 0)
    while not stop:
        try:
            x = torch.empty(64 << 20, dtype=torch.uint8, device=dev); del x
            torch.cuda.empty_cache(); counts['alloc'] += 1
        except Exception as e:
            errs['alloc'].append(repr(e)[:200])


def direct():
    try:
        dev = torch.device('cuda', 0)
        with torch.cuda.stream(torch.cuda.Stream(dev)):
            eng = TCEngine('GL', device=0).stage_env(env)
            k = 0
            while not stop:
                try:
                    p = DevicePipeline(eng, 4096 + 256 * (k % 8), 1024 + 64 * (k % 8), sort_storms=2.0, tc_rows_only=True)   # workspaces grow
                    p.round(2001, 4096 * k, graph=False)
                    torch.cuda.current_stream().synchronize(); counts['direct'] += 1
                except Exception as e:
                    errs['direct'].append(repr(e)[:200])
                k += 1
    except Exception:
        errs['direct'].append(traceback.format_exc()[-400:])


ths = [threading.Thread(target=f) for f in (capturer, stager, allocator, direct)]
for t in ths:
    t.start()
time.sleep(float(sys.argv[1]) if len(sys.argv) > 1 else 12.0)
stop = True
for t in ths:
    t.join()
print('iterations', counts)
for k, v in errs.items():
    print(k, len(v), 'errors', sorted(set(v))[:4])
