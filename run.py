#!/usr/bin/env python3
"""CLI mirror of the reference's run.py (`run.py:8-18`):  python3 run.py <BASIN> [--namelist FILE]

Creates the output directory, copies the namelist there, and runs the downscaling
for one basin on the MI355X(s) of this node.  Launch under torchrun for several GPUs:
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 run.py GL
"""
import argparse
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('basin')
    ap.add_argument('--namelist', default=None, help='user namelist file in the reference format')
    ap.add_argument('--synthetic', action='store_true', help='run on regenerated ERA5-shaped fields')
    a = ap.parse_args()
    from tropical_cyclone_risk_amd import compute, distributed, namelist
    if a.namelist:
        namelist.load(a.namelist)
    if a.synthetic:
        namelist.dataset_type = 'SYNTHETIC'
    rank, world, local = distributed.init_from_env()
    f_base = '%s/%s/' % (namelist.output_directory, namelist.exp_name)
    if rank == 0:
        os.makedirs(f_base, exist_ok=True)
        print('Saving model output to %s' % f_base)
        shutil.copyfile(a.namelist or namelist.__file__, '%s/namelist.py' % f_base)
        print('Running tracks for basin %s...' % a.basin)
    compute.run_downscaling(a.basin, nl=namelist)


if __name__ == '__main__':
    main()
