"""
Job execution on the GPU: one job = one (run, g) pair = what the reference hands to one solver process,
`<exe> <Nphoton> <solver 0|1|2> <inp.txt> <out.bin>` (er3t/rtm/mca/mca_run.py:101-115).

`JobRunner` keeps one `Mi3dSolver` per process and re-uses what consecutive jobs share: the 3-D arrays stay on the
device while only the 1-D profiles change from one g to the next.  With torch.distributed initialised (one process
per GPU) every rank transports its contiguous share of the job's photon ids and the raw tallies are summed by a
single all-reduce before rank 0 writes the output.

The module is also the drop-in solver executable:

    python -m er3t_amd.rtm.mca.mca_exe <Nphoton> <solver> <inp.txt> <out.bin>
"""

import os
import sys

import numpy as np

from er3t_amd.scene import Scene, TARGET_FLUX, TARGET_RADIANCE, TARGET_HEAT
from er3t_amd.dist import photon_shard, allreduce_tallies, world_info
from er3t_amd.rtm.mca.mca_inp import mca_inp_read
from er3t_amd.rtm.mca.mca_out import mca_out_write

__all__ = ['JobRunner', 'run_job', 'main']


_WARNED = set()


def _check_supported(nml):
    if int(nml.get('Wld_moptim', 0) or 0) != 0 and 'moptim' not in _WARNED and world_info()[0] == 0:
        # (MCARaTS' biasing optimisations, er3t/rtm/mca/mca_inp.py:27-33; er3t sets 2 for tune=True, mcarats.py:257-260)
        _WARNED.add('moptim')
        print('Warning [mca_exe]: <Wld_moptim=%s> asks for MCARaTS\' variance-reduction approximations; this solver runs its unbiased estimator (Wld_moptim=0).' % nml.get('Wld_moptim'), file=sys.stderr)
    if int(nml.get('Wld_mtarget', 1)) not in (1, 2):
        raise OSError('Error [mca_exe]: <Wld_mtarget=%s> is not supported (1: flux, 2: radiance).' % nml.get('Wld_mtarget'))
    if int(nml.get('Wld_mtarget', 1)) == 2 and int(nml.get('Rad_mrkind', 2)) not in (1, 2):
        raise OSError('Error [mca_exe]: <Rad_mrkind=%s> is not supported (1: all-sky camera, 2: satellite sensor).' % nml.get('Rad_mrkind'))
    if int(nml.get('Src_mtype', 1)) != 1:
        raise OSError('Error [mca_exe]: only the solar source (<Src_mtype=1>) is supported.')


class JobRunner:

    def __init__(self, device=None, column_le=True):
        from er3t_amd.solver import Mi3dSolver
        rank, world = world_info()
        if device is None:
            # one rank per GPU; more ranks than GPUs (a rehearsal on a one-GPU box) share devices round robin
            from er3t_amd.solver import load_library
            ndev = max(load_library().mi3d_device_count(), 1)
            device = int(os.environ.get('LOCAL_RANK', '0')) % ndev if world > 1 else 0
        # Slot 0 is THE solver handle.  The fused g-loop (mcarats_ng.run_fused) takes a second one on the same GPU and gives
        # every other job to it: a launch ends with a tail as long as its longest history (~2.5 ms, profiles/r02/small_launches.log)
        # during which the next job's launch finds the GPU all but empty.  MI3D_FUSED_SLOTS=1 switches that off.
        self.sols = [Mi3dSolver(device=device)]
        self.scenes = [None]
        self._key3d = [None]
        self._tensors = [None]
        self._streams = [None]
        self.column_le = column_le
        self.rank, self.world = rank, world
        self.photons_done = 0
        self.kernel_ms = 0.0
        self.nslot = 1
        self._stats_slots = set()

    @property
    def sol(self):
        return self.sols[0]

    @property
    def scene(self):
        return self.scenes[0]

    def use_slots(self, n):
        """how many solver handles share the jobs of the fused g-loop (1 or 2)"""
        from er3t_amd.solver import Mi3dSolver
        n = max(1, min(4, int(os.environ.get("MI3D_FUSED_SLOTS", n))))
        while len(self.sols) < n:
            self.sols.append(Mi3dSolver(device=self.sols[0].device))
            self.scenes.append(None); self._key3d.append(None); self._tensors.append(None); self._streams.append(None)
        if n > 1 and self.world == 1:
            # side by side means: not both on the null stream (under torchrun the second slot gets a torch stream, _bind_tensors)
            for sol in self.sols:
                sol.set_tuning(own_stream=1)
        self.nslot = n
        return n

    # What consecutive jobs may share: the side files (identified by path, size and time stamp) and EVERY namelist entry
    # except the per-g 1-D profiles and the seed.  On a hit only the 1-D profiles are replaced on the device.
    _PER_JOB = ('Atm_ext1d', 'Atm_omg1d', 'Atm_apf1d', 'Atm_abs1d', 'Atm_tmp1d', 'Wld_jseed')

    @classmethod
    def _file_key(cls, nml, fdir):
        keys = []
        for k in ('Atm_inpfile', 'Sca_inpfile', 'Sfc_inpfile'):
            v = nml.get(k)
            if v:
                p = os.path.join(fdir, v)
                st = os.stat(p)
                keys.append((os.path.abspath(p), st.st_size, st.st_mtime_ns))
        for k in sorted(nml):
            if not k.startswith(cls._PER_JOB):
                keys.append((k, np.asarray(nml[k]).tobytes() if isinstance(nml[k], (list, tuple, np.ndarray)) else str(nml[k])))
        return tuple(keys)

    def load(self, nml, fdir, solver, slot=0):
        _check_supported(nml)
        sol = self.sols[slot]
        key = (self._file_key(nml, fdir), int(solver))
        if key == self._key3d[slot]:
            # same 3-D inputs: only the 1-D profiles (the per-g part) are replaced
            scene = self.scenes[slot]
            nml1 = {k: v for k, v in nml.items() if not k.endswith('inpfile')}
            nml1.update(Atm_nz3=0, Sca_npf=0)
            s1 = Scene.from_nml(nml1, fdir, solver=solver)
            if s1.nz != scene.nz or s1.np1d != scene.np1d:      # (the key holds Atm_nz and Atm_np1d: cannot happen)
                raise OSError('Error [mca_exe]: the 1-D grid changed shape between two jobs that share their 3-D inputs.')
            sol.update_atm1d(s1)
            scene.zgrd, scene.ext1d, scene.omg1d, scene.apf1d, scene.abs1d = s1.zgrd, s1.ext1d, s1.omg1d, s1.apf1d, s1.abs1d
        else:
            self.scenes[slot] = Scene.from_nml(nml, fdir, solver=solver)
            self._bind_tensors(self.scenes[slot], slot)
            sol.load_scene(self.scenes[slot], column_le=self.column_le)
            self._key3d[slot] = key
            self._stats_slots.discard(slot)
        return self.scenes[slot]

    def _bind_tensors(self, scene, slot=0):
        # multi-process runs accumulate into torch tensors so that the all-reduce works in place
        sol = self.sols[slot]
        if self.world > 1:
            import torch
            dev = torch.device('cuda', sol.device)
            rad = torch.zeros(max(scene.nview, 1)*scene.nyr*scene.nxr, dtype=torch.float64, device=dev)       # raw tallies: float64
            flux = torch.zeros(3*(scene.nz+1)*scene.ny*scene.nx, dtype=torch.float64, device=dev)
            heat = torch.zeros(scene.nz*scene.ny*scene.nx, dtype=torch.float64, device=dev) if scene.target & TARGET_HEAT else None
            self._tensors[slot] = (rad, flux, heat)
            if slot == 0:
                stream = torch.cuda.current_stream(dev)
            else:
                stream = self._streams[slot] = self._streams[slot] or torch.cuda.Stream(dev)
            sol.bind(rad_ptr=rad.data_ptr(), flux_ptr=flux.data_ptr(), stream=stream.cuda_stream, heat_ptr=None if heat is None else heat.data_ptr())
        else:
            self._tensors[slot] = None
            sol.bind(None, None, None)

    def run(self, nphoton, seed):
        """transport <nphoton> histories of the loaded job (this rank's share of them); returns the result arrays"""
        nphoton = int(nphoton)
        off, cnt = photon_shard(nphoton, self.world, self.rank)
        self.sol.reset()
        self.sol.run(cnt, seed=seed, offset=off)
        if self._tensors[0] is not None:
            import torch
            self.sol.sync()
            torch.cuda.current_stream().synchronize()
            allreduce_tallies(*self._tensors[0])
            torch.cuda.synchronize(self.sol.device)
        ms, _ = self.sol.timing()
        self.kernel_ms += ms
        self.photons_done += cnt
        out = {}
        if self.scene.target & TARGET_RADIANCE:
            out['rad'] = self.sol.radiance(nphoton)
        if self.scene.target & TARGET_FLUX:
            out['flux'] = self.sol.flux(nphoton)
        if self.scene.target & TARGET_HEAT:
            out['heat'] = self.sol.heating(nphoton)
        return out

    def collect(self, nphoton, slot=0):
        """results of the job launched last on <slot> (single process: no exchange)"""
        sol, scene = self.sols[slot], self.scenes[slot]
        ms, _ = sol.timing()
        self.kernel_ms += ms
        out = {}
        if scene.target & TARGET_RADIANCE:
            out['rad'] = sol.radiance(int(nphoton))
        if scene.target & TARGET_FLUX:
            out['flux'] = sol.flux(int(nphoton))
        if scene.target & TARGET_HEAT:
            out['heat'] = sol.heating(int(nphoton))
        return out

    # ---- several ranks, file route: one exchange per BATCH of jobs instead of one per job -------------------------------------
    def run_batched(self, jobs, solver, max_bytes=2.0e9):
        """
        jobs: [(input file, output file, photons)] of one simulation, every rank transporting its share of every job.
        The raw tallies of up to `max_bytes` worth of consecutive jobs go to slices of ONE device tensor, two solver handles take
        turns through the jobs (the tail of a launch runs beside the next job's), and the ranks exchange that tensor with ONE
        all-reduce per batch -- per run of 16 g for a radiance target -- where `run` pays one exchange, two host waits and a
        launch tail per job.  Rank 0 then normalises job by job (what mi3d_get_radiance / mi3d_get_flux / mi3d_get_heating do) and
        writes the output files.
        """
        import torch
        nslot = self.use_slots(2)
        i, njob = 0, len(jobs)
        while i < njob:
            buf, metas, shape = None, [], None
            untimed = set()                             # slots whose last job's kernel time has not been added to self.kernel_ms yet
            while i < njob and (buf is None or len(metas) < buf.shape[0]):
                fname_inp, fname_out, nphoton = jobs[i]
                slot = len(metas) % nslot
                nml = mca_inp_read(fname_inp)
                fdir = os.path.dirname(os.path.abspath(fname_inp))
                sc = self.load(nml, fdir, int(solver), slot=slot)
                sizes = (max(sc.nview, 1)*sc.nyr*sc.nxr if sc.target & TARGET_RADIANCE else 0,
                         3*(sc.nz+1)*sc.ny*sc.nx if sc.target & TARGET_FLUX else 0,
                         sc.nz*sc.ny*sc.nx if sc.target & TARGET_HEAT else 0)
                if buf is None:
                    shape = sizes
                    nb = max(1, min(njob-i, int(max_bytes//(8*sum(sizes)))))
                    dev = torch.device('cuda', self.sol.device)
                    buf = torch.zeros((nb, sum(sizes)), dtype=torch.float64, device=dev)
                elif sizes != shape:
                    break                       # (a job of another shape opens the next batch; its scene is loaded again there: cached)
                row = buf[len(metas)]
                ptr = lambda a, n: row[a:a+n].data_ptr() if n else None
                sol = self.sols[slot]
                stream = torch.cuda.current_stream(dev) if slot == 0 else self._streams[slot]
                # (a tally this job does not have still needs somewhere to point: the handle's own buffer)
                sol.bind(rad_ptr=ptr(0, sizes[0]), flux_ptr=ptr(sizes[0], sizes[1]), stream=stream.cuda_stream,
                         heat_ptr=ptr(sizes[0]+sizes[1], sizes[2]))
                seed = int(nml.get('Wld_jseed', 0) or 0)
                if seed == 0:
                    seed = _fresh_seed(self)
                off, cnt = photon_shard(int(nphoton), self.world, self.rank)
                if slot in untimed:                     # (this handle's previous job of THIS batch: reset clears its timing; what ran
                    self.kernel_ms += sol.timing()[0]   #  before the batch -- run, collect, stats_add, an earlier batch -- has been counted)
                sol.reset()
                sol.run(cnt, seed=seed, offset=off)
                untimed.add(slot)
                self.photons_done += cnt
                metas.append(dict(fname_out=fname_out, nphoton=int(nphoton), slot=slot, scene=sc,
                                  direct=sol.direct_levels() if sc.target & TARGET_FLUX else None,
                                  norm=dict(src_flx=sc.src_flx, mu0=sc.mu0, rad_kind=getattr(sc, 'rad_kind', 2), area=sc.nx*sc.dx*sc.ny*sc.dy,
                                            dz=np.diff(sc.zgrd))))
                i += 1
            for slot, sol in enumerate(self.sols[:nslot]):
                sol.sync()
                if slot in untimed:
                    self.kernel_ms += sol.timing()[0]
            untimed.clear()
            torch.cuda.synchronize(self.sol.device)
            allreduce_tallies(buf)                      # ONE exchange for the whole batch
            torch.cuda.synchronize(self.sol.device)
            if self.rank == 0:
                # (normalised on the device -- float32 crosses to the host, half of the raw float64 tallies' bytes, and no host arithmetic over
                #  millions of cells --, then read back and written by a few threads side by side: the copies and the file writes release the
                #  interpreter lock.  A flux job's file is 14 MB on 128 x 128 x 69, 0.3 GB on 480 x 480 x 117; one after the other the 48 files of a
                #  16 g x 3 run simulation took longer than its photons)
                outs = [self._normalise(row, shape, m) for row, m in zip(buf[:len(metas)], metas)]
                torch.cuda.synchronize(self.sol.device)

                def finish(om):
                    out, m = om
                    self.write(m['fname_out'], {k: v.cpu().numpy() for k, v in out.items()})
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max_workers=max(1, min(8, len(metas), os.cpu_count() or 1))) as pool:
                    list(pool.map(finish, zip(outs, metas)))
                del outs
            # the handles go back to tensors of their own (the next caller may be `run`)
            for slot in range(nslot):
                if self.scenes[slot] is not None:
                    self._bind_tensors(self.scenes[slot], slot)

    @staticmethod
    def _normalise(row, sizes, m):
        """raw all-reduced tallies of one job (a row of the batch's float64 device tensor) -> the float32 arrays of its output file, still on
        the device (include/mi3d.h: mi3d_get_radiance, mi3d_get_flux, mi3d_get_heating state the same factors; float64 products and sums in
        their order, rounded to float32 once)"""
        import torch
        sc, n, p = m['scene'], float(m['nphoton']), m['norm']
        out = {}
        a, b, c = sizes
        if a:
            fac = p['src_flx']*p['mu0']*(p['area'] if p['rad_kind'] == 1 else sc.nxr*sc.nyr)/n
            out['rad'] = (row[:a]*fac).to(torch.float32).reshape(max(sc.nview, 1), sc.nyr, sc.nxr)[:sc.nview]
        if b:
            raw = row[a:a+b].reshape(3, sc.nz+1, sc.ny, sc.nx).clone()
            raw[1] += raw[0]                                           # raw planes: direct-down, DIFFUSE-down, up
            f = raw*(p['src_flx']*p['mu0']*sc.nx*sc.ny/n)
            f[:2] += torch.as_tensor(np.asarray(m['direct'], dtype=np.float64), device=row.device)[None, :, None, None]   # the known part of the direct beam
            out['flux'] = f.to(torch.float32)
        if c:
            dz = torch.as_tensor(np.asarray(p['dz'], dtype=np.float64), device=row.device)
            h = row[a+b:a+b+c].reshape(sc.nz, sc.ny, sc.nx)*(p['src_flx']*p['mu0']*sc.nx*sc.ny/n)/dz[:, None, None]
            out['heat'] = h.to(torch.float32)
        return out

    # ---- fused g-loop: results stay on the device, only run statistics come back ---------------
    def launch(self, nphoton, seed, slot=0):
        """transport this rank's share of the job loaded on <slot>; no read-back, no exchange"""
        nphoton = int(nphoton)
        off, cnt = photon_shard(nphoton, self.world, self.rank)
        self.sols[slot].reset()
        self.sols[slot].run(cnt, seed=seed, offset=off)
        self.photons_done += cnt

    def stats_begin(self):
        """start gathering run statistics for the loaded scene (its shape and target)"""
        self._run_tensors = None
        if self.world > 1:
            import torch
            dev = torch.device('cuda', self.sol.device)
            rad = torch.zeros(max(self.scene.nview, 1)*self.scene.nyr*self.scene.nxr, dtype=torch.float32, device=dev)
            flux = torch.zeros(3*(self.scene.nz+1)*self.scene.ny*self.scene.nx if self.scene.target & TARGET_FLUX else 1,
                               dtype=torch.float32, device=dev)
            self._run_tensors = (rad, flux)
            self.sol.stats_begin(rad.data_ptr(), flux.data_ptr())
            # the analytic direct beam joins the run field on ONE rank: the fields are summed over the ranks
            self.sol.stats_set_analytic_share(1.0 if self.rank == 0 else 0.0)
        else:
            self.sol.stats_begin()
        self._stats_slots = {0}

    def _stats_join(self, slot):
        """a further handle joins the statistics of slot 0: it adds its jobs into slot 0's run fields"""
        if slot in self._stats_slots:
            return
        if 0 not in self._stats_slots:
            raise OSError('Error [mca_exe]: stats_begin has not been called.')
        sol = self.sols[slot]
        sol.stats_join(self.sol)
        sol.stats_set_analytic_share(1.0 if self.rank == 0 or self.world == 1 else 0.0)
        self._stats_slots.add(slot)

    def stats_add(self, nphoton, factors, slot=0):
        """fold the job that ran on <slot> into the current run: factors[level] (flux) / factors[view] (radiance).
        The jobs of a run must be added in job order, whichever slot ran them: the run field is a float32 sum."""
        sol, scene = self.sols[slot], self.scenes[slot]
        self._stats_join(slot)
        ms, _ = sol.timing()                 # before the next reset clears it
        self.kernel_ms += ms
        f = np.asarray(factors, dtype=np.float32)
        for other in range(self.nslot):
            if other != slot:
                sol.stats_chain(self.sols[other])
        sol.stats_add(int(nphoton), factor_rad=f if scene.target & TARGET_RADIANCE else None,
                      factor_flux=f if scene.target & TARGET_FLUX else None)

    def stats_end_run(self, keep=False):
        for other in range(1, self.nslot):
            if other in self._stats_slots:
                self.sols[other].sync()
        if self._run_tensors is not None:
            import torch
            self.sol.sync()
            torch.cuda.current_stream().synchronize()
            allreduce_tallies(*self._run_tensors)          # one exchange per run: the run field is linear in the tallies
            torch.cuda.synchronize(self.sol.device)
        return self.sol.stats_end_run(keep=keep)

    def stats_result(self):
        out = {}
        for key, which in (('rad', TARGET_RADIANCE), ('flux', TARGET_FLUX)):
            if self.scene.target & which:
                mean, sdev, nrun = self.sol.stats_get(which)
                out[key] = {'mean': mean, 'std': sdev, 'nrun': nrun}
        return out

    def write(self, fname_out, result):
        """MCARaTS-format out.bin + .ctl (what er3t/rtm/mca/mca_out.py:48-103 parses); rank 0 only"""
        if self.rank != 0:
            return
        os.makedirs(os.path.dirname(os.path.abspath(fname_out)), exist_ok=True)
        if 'flux' in result:
            f = result['flux']                                   # (3, nz+1, ny, nx) -> (nx, ny, nz+1)
            names = [('fdnd', 'direct downward flux density'), ('fdn', 'total downward flux density'), ('fup', 'upward flux density')]
            variables = [(n, d, np.transpose(f[i], (2, 1, 0))) for i, (n, d) in enumerate(names)]
            if 'heat' in result:                                 # (nz, ny, nx) -> (nx, ny, nz): a fourth variable on the layer grid (Flx_mhrt = 1)
                variables.append(('hrt', 'absorbed power per unit volume (heating rate x air density x c_p)', np.transpose(result['heat'], (2, 1, 0))))
            mca_out_write(fname_out, variables)
        else:
            r = result['rad']                                    # (nview, nyr, nxr) -> (nxr, nyr, nview)
            mca_out_write(fname_out, [('rad', 'pixel-averaged radiance', np.transpose(r, (2, 1, 0)))])


_RUNNER = None


def get_runner(**kwargs):
    global _RUNNER
    if _RUNNER is None:
        _RUNNER = JobRunner(**kwargs)
    return _RUNNER


_SEED_COUNTER = [0]


def _fresh_seed(runner):
    """<Wld_jseed=0>: a seed of the solver's own choosing -- drawn ONCE (rank 0: operating-system entropy plus a job counter,
    so that jobs started within the same second differ) and handed to the other ranks, which must follow the same histories"""
    _SEED_COUNTER[0] += 1
    seed = (int.from_bytes(os.urandom(6), 'little') + _SEED_COUNTER[0]) & 0x7FFFFFFFFFFF
    if runner.world > 1:
        import torch
        import torch.distributed as dist
        dev = torch.device('cuda', runner.sol.device) if dist.get_backend() == 'nccl' else torch.device('cpu')
        t = torch.tensor([seed], dtype=torch.int64, device=dev)
        dist.broadcast(t, src=0)
        seed = int(t.item())
    return max(seed, 1)


def run_job(fname_inp, fname_out, nphoton, solver=0, runner=None):

    """one job from its input file, exactly like one invocation of the reference's solver executable"""

    runner = runner or get_runner()
    nml = mca_inp_read(fname_inp)
    fdir = os.path.dirname(os.path.abspath(fname_inp))
    runner.load(nml, fdir, int(solver))
    seed = int(nml.get('Wld_jseed', 0) or 0)
    if seed == 0:
        seed = _fresh_seed(runner)
    result = runner.run(nphoton, seed)
    runner.write(fname_out, result)
    return result


def submit_job(fname_inp, nphoton, solver, runner, slot=0):

    """load one job on solver handle <slot> and launch it; the results are fetched later by collect_job"""

    nml = mca_inp_read(fname_inp)
    fdir = os.path.dirname(os.path.abspath(fname_inp))
    runner.load(nml, fdir, int(solver), slot=slot)
    seed = int(nml.get('Wld_jseed', 0) or 0)
    if seed == 0:
        seed = _fresh_seed(runner)
    runner.launch(nphoton, seed, slot=slot)
    return (slot, int(nphoton))


def collect_job(job, fname_out, runner):

    """wait for a job submitted by submit_job, read its results back and write its output file"""

    slot, nphoton = job
    result = runner.collect(nphoton, slot=slot)
    runner.write(fname_out, result)
    return result


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 4:
        print('usage: python -m er3t_amd.rtm.mca.mca_exe <Nphoton> <solver 0|1|2> <input file> <output file>', file=sys.stderr)
        return 2
    nphoton, solver, fname_inp, fname_out = int(float(argv[0])), int(argv[1]), argv[2], argv[3]
    try:
        run_job(fname_inp, fname_out, nphoton, solver)
    except (OSError, ValueError) as err:
        print(str(err), file=sys.stderr)
        return 1
    return 0


if __name__ == '__main__':
    sys.exit(main())
