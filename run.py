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
# run_downscaling keeps several years in flight, each on its own stream: a hardware queue for each (ROCm multiplexes a process's
# streams onto 4 by default); must be set before the HIP runtime starts
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('basin')
    ap.add_argument('--namelist', default=None, help='user namelist file in the reference format')
    ap.add_argument('--synthetic', action='store_true', help='run on regenerated ERA5-shaped fields')
    ap.add_argument('--fields', default=None, metavar='DIR',
                    help='read thermo_*.nc, env_wnd_*.nc, {mld,strat}_climatology.nc, land.nc, bathymetry.nc and '
                         'land/<B>.nc from DIR (the layout --export-synthetic writes) instead of the namelist paths')
    ap.add_argument('--preprocess', action='store_true',
                    help='first write env_wnd_*.nc / thermo_*.nc from the raw daily winds and monthly sst / mslp / t / q '
                         'files under namelist.base_directory, as the reference\'s compute_downscaling_inputs does')
    ap.add_argument('--export-synthetic', default=None, metavar='DIR',
                    help='write the synthetic fields for the namelist\'s years in the reference\'s file schema and exit')
    a = ap.parse_args()
    from tropical_cyclone_risk_amd import compute, distributed, namelist
    if a.namelist:
        namelist.load(a.namelist)
    if a.synthetic:
        namelist.dataset_type = 'SYNTHETIC'
    if a.export_synthetic:
        from tropical_cyclone_risk_amd import fields, synthetic
        files = fields.write_reference_files(synthetic.make_env('era5'), a.export_synthetic, namelist.start_year, namelist,
                                             last_year=namelist.end_year)
        print('\n'.join('%s: %s' % kv for kv in sorted(files.items())))
        return
    env = None
    if a.fields:
        import glob
        from tropical_cyclone_risk_amd import fields
        one = lambda pat: sorted(glob.glob(os.path.join(a.fields, pat)))[0]
        env = fields.FileEnvironment(namelist, dict(
            thermo=one('thermo_*.nc'), env_wnd=one('env_wnd_*.nc'), mld=one('mld_climatology.nc'),
            strat=one('strat_climatology.nc'), land=one('land.nc'), bathy=one('bathymetry.nc'),
            basin_dir=os.path.join(a.fields, 'land')))
    rank, world, local = distributed.init_from_env()
    if not a.synthetic and not a.fields and rank == 0:
        # the reference's run.py:14: land.nc and the basin masks next to the sources, generated when missing
        from tropical_cyclone_risk_amd import fields, masks
        fl = fields.default_files(namelist)
        try:
            masks.generate_land_masks(fl['basin_dir'], land_file=fl['land'])
        except FileNotFoundError as e:
            print('land masks not generated: %s' % e)
    if a.preprocess and rank == 0:
        from tropical_cyclone_risk_amd import preprocess
        from tropical_cyclone_risk_amd.engine import TCEngine
        eng = TCEngine(a.basin, device=distributed.local_device(local), nl=namelist)
        preprocess.run_preprocessing(eng, namelist)
        eng.close()
    distributed.barrier()
    f_base = '%s/%s/' % (namelist.output_directory, namelist.exp_name)
    if rank == 0:
        os.makedirs(f_base, exist_ok=True)
        print('Saving model output to %s' % f_base)
        shutil.copyfile(a.namelist or namelist.__file__, '%s/namelist.py' % f_base)
        print('Running tracks for basin %s...' % a.basin)
    compute.run_downscaling(a.basin, env=env, nl=namelist)


if __name__ == '__main__':
    main()
