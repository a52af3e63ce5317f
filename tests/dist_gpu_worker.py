"""
Worker of tests/test_gpu_dropin.py::test_two_ranks_share_the_jobs: one process per rank under torch.distributed.run,
backend 'gloo' so that two ranks can rehearse the multi-GPU path on the ONE GPU of a test box (the ranks share cuda:0;
on a node RCCL would be the backend and every rank would have its own GPU -- the code path is the same).

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tests/dist_gpu_worker.py <outdir>
"""
import contextlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(outdir):
    import torch.distributed as dist
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()

    import er3t_amd.rtm.mca as mca
    from er3t_amd.synth import atm_synth, abs_synth, cld_synth
    from er3t_amd.rtm.mca.mca_exe import JobRunner, get_runner, run_job
    from tests.golden import inputs as gin

    atm = atm_synth(np.concatenate([np.arange(0, 11)*0.2, np.arange(3, 21)*1.0]))
    ab = abs_synth(650.0, atm, Ng=3)
    cld = cld_synth(atm, nx=12, ny=10, nz=10, z_base=0.4, z_top=1.6, cot_mean=8.0, seed=5)
    with contextlib.redirect_stdout(io.StringIO()):
        a1 = mca.mca_atm_1d(atm_obj=atm, abs_obj=ab)
        a3 = mca.mca_atm_3d(atm_obj=atm, cld_obj=cld, fname=os.path.join(outdir, 'atm3d.bin'), quiet=True)
    res = {}
    for target in ('radiance', 'flux'):
        kw = dict(atm_1ds=[a1], atm_3ds=[a3], Ng=3, target=target, surface_albedo=0.05, solar_zenith_angle=40.0, Nrun=2,
                  photons=3e5, weights=ab.coef['weight']['data'], solver='3D', mp_mode='py', overwrite=True, date=gin.DATE, quiet=True)
        # (1) the reference's route: one output file per job, every job's photons split over the ranks, one all-reduce per job
        m = mca.mcarats_ng(fdir=os.path.join(outdir, target), **kw)
        out = mca.mca_out_ng(mca_obj=m, abs_obj=ab, mode='all', squeeze=True, quiet=True).data
        # (2) fused: run statistics on the device, one all-reduce per run, no files
        mf = mca.mcarats_ng(fdir=os.path.join(outdir, target+'_fused'), abs_obj=ab, keep_files=False, **kw)
        outf = mca.mca_out_ng(mca_obj=mf, abs_obj=ab, mode='mean', squeeze=True, quiet=True).data
        key = 'rad' if target == 'radiance' else 'f_up'
        import torch
        t = torch.tensor([m.run0.photons_done, mf.photons_done], dtype=torch.int64)
        dist.all_reduce(t)
        if rank == 0:
            res[target+'_photons'] = np.array([t[0].item(), int(m.photons.sum()), t[1].item(), int(mf.photons.sum())])
            res[target+'_dist_mean'] = out[key]['data'].mean(axis=-1)
            res[target+'_fused_mean'] = outf[key]['data']
            if target == 'flux':
                # every output variable, level by level (domain means): file route (runs averaged) and fused route
                for v in ('f_down_direct', 'f_down', 'f_up'):
                    res['flux_dist_'+v] = out[v]['data'].mean(axis=-1).mean(axis=(0, 1))
                    res['flux_fused_'+v] = outf[v]['data'].mean(axis=(0, 1))
                res['flux_kdir'] = np.array([int(a3.nml['Atm_iz3l']['data']) - 1 + int(a3.nml['Atm_nz3']['data'])])
            res[target+'_fused_files'] = np.array([int(os.path.exists(f)) for row in mf.fnames_out for f in row])
            # one job of route (1) again, from the same input file, through ONE rank holding all its photons
            solo = JobRunner(device=0); solo.rank, solo.world = 0, 1
            ir, ig = 1, 2
            r = run_job(m.fnames_inp[ir][ig], os.path.join(outdir, 'solo_%s.bin' % target), int(m.photons[ir*m.Ng+ig]), 0, runner=solo)
            raw = mca.mca_out_raw(m.fnames_out[ir][ig])
            if target == 'radiance':
                res[target+'_job_dist'] = raw.data[0]['data'][:, :, 0, 0]; res[target+'_job_solo'] = r['rad'][0].T
            else:
                res[target+'_job_dist'] = raw.data[2]['data'][:, :, :, 0]; res[target+'_job_solo'] = np.transpose(r['flux'][2], (2, 1, 0))
    # heating rates (a fourth tally, exchanged with the others in the batch's one all-reduce): one job against one rank holding all of it
    kw = dict(atm_1ds=[a1], atm_3ds=[a3], Ng=3, target='heating rate', surface_albedo=0.05, solar_zenith_angle=40.0, Nrun=1,
              photons=3e5, weights=ab.coef['weight']['data'], solver='3D', mp_mode='py', overwrite=True, date=gin.DATE, quiet=True)
    m = mca.mcarats_ng(fdir=os.path.join(outdir, 'heating'), **kw)
    if rank == 0:
        solo = JobRunner(device=0); solo.rank, solo.world = 0, 1
        r = run_job(m.fnames_inp[0][1], os.path.join(outdir, 'solo_heating.bin'), int(m.photons[1]), 0, runner=solo)
        raw = mca.mca_out_raw(m.fnames_out[0][1])
        res['heating_job_dist'] = raw.data[3]['data'][:, :, :, 0]; res['heating_job_solo'] = np.transpose(r['heat'], (2, 1, 0))
        res['heating_flux_dist'] = raw.data[1]['data'][:, :, :, 0]; res['heating_flux_solo'] = np.transpose(r['flux'][1], (2, 1, 0))
    if rank == 0:
        np.savez(os.path.join(outdir, 'result.npz'), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1])
