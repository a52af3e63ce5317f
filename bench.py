"""
bench.py -- photons/s of the photon-transport hot path on the BASELINE.json domain.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        (no launcher: bench.py starts the N ranks itself, see `spawn`)

One "step" = one photon batch: zero the tally, transport `--photons` histories (BASELINE config 4: 1e9) through the
synthetic 480x480x100 LES cloud domain (nadir radiance, HG g=0.85 cloud + Rayleigh + gas
absorption, Lambert surface), and -- for N > 1 -- one RCCL all-reduce of the radiance tally.  Photon ids are
disjoint across ranks and steps.  `--scaling weak` (default): `--photons` per GPU, per-GPU work fixed; `--scaling
strong`: `--photons` in all, split over the ranks (config 4 as written: 1e9 photons on 8 GPUs).  Inputs are resident in
HBM before the timed region.  Rank 0 prints ONE JSON line.

Extra objects on the line
  roofline     : algorithmic bytes per launch (measured with the instrumented kernel build on a sub-sample of the
                 timed photon ids, formula in DESIGN.md §6) / average launch duration (HIP events on the launch
                 stream, inside libmi3drt) against the 8 TB/s HBM3E peak; `traffic` = HBM bytes per launch from PMC passes
                 made after the timed region (child processes under rocprofv3 --pmc, N=1; --no-pmc or no rocprofv3: the
                 figures recorded in profiles/traffic.json, labelled as replayed; null if absent); `valu` = vector-ALU issue slots in
                 use (SQ_INSTS_VALU pass of the same command x 4 cycles / (1024 SIMDs x 2.4 GHz x launch time)) -- the
                 limit this kernel actually runs into
                 `peak_measured` / `frac_of_measured`: the rate a float4 stream copy reaches on this device, measured after
                 the timed region (tools/microbench/stream_copy), as the second denominator SURVEY.md §8(d) asks for
  cpu_baseline : the CPU oracle (oracle/mi3d_oracle.c, OpenMP) timed on this box's host cores on a bounded
                 sample of the same workload, rank 0 at N=1 only
  parity       : the second half of BASELINE.json's metric ("radiance sigma-error"): the HIP path on the photon ids the CPU leg has
                 just transported, eight batches on either side: difference of the domain means in sigma of two independent estimates
                 of that size, the same difference paired (relative), z-scores of 16 x 16 block means (`parity_stats`; the
                 full-size tests of tests/test_gpu_fullsize.py assert on the same numbers)
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak BW, 8.0 TB/s spec
# The chip's vector-issue ceiling for the photon loop's OWN instruction mix, measured (tools/microbench/mix_rates, profiles/r05/mix_rates_mix.log):
# a loop of the mix's instruction classes in the proportions the SQ_INSTS_VALU_* counters give for one launch of the headline kernel
# (profiles/r05/mix_rates_mix_args.txt), all lanes active, no memory access, occupancy forced to six waves per SIMD: 7.52e11
# wave-instructions a second for the whole chip = 2.94 cycles per wave-instruction and SIMD at the 2.16 GHz the chip holds under that
# load.  (Not one price per instruction: v_fma / v_mul / v_add / v_xor with register operands cost 2.1-2.3 cycles, anything with an SGPR
# operand, compares, selects through an SGPR mask, shifts, conversions, min / max, the three-operand integer forms and packed math 4.1,
# transcendentals 8.1: profiles/r05/mix_rates_ops.log, mix_rates_ops2.log.  Round 4 priced every instruction at 4 cycles and 2.4 GHz
# and read 1.02-1.05 of the issue rate -- a fraction above 1.)  issue_frac = wave-instructions per second of the run / this ceiling.
VALU_MIX_CEILING = 7.52e11


def parity_stats(g, o, nblk=16, min_block_rel=0.0):
    """Two estimates of the same images from the SAME photon ids, batch by batch: g, o of shape (batches, views, ny, nx) -- the HIP
    path and the oracle.  Per view (SURVEY.md §8(d) "Metric"): the difference of the domain means in units of the standard error of
    the difference of two INDEPENDENT estimates of this size (north_star's sigma: sqrt(2) x the oracle's batch-to-batch standard
    error), the same difference relative to the mean and in standard errors of the PAIRED difference (same ids on both sides: most
    of the noise cancels, a bias far below the Monte-Carlo noise shows), and the z-scores of nblk x nblk block means (of the blocks
    whose oracle mean exceeds min_block_rel x the image mean: the corners of a camera's round image are empty)."""
    g = np.asarray(g, dtype=np.float64); o = np.asarray(o, dtype=np.float64)
    nb = g.shape[0]
    out = []
    for iv in range(g.shape[1]):
        gm, om = g[:, iv].mean(axis=(1, 2)), o[:, iv].mean(axis=(1, 2))
        d = gm-om
        se_ind = np.sqrt(2.0)*om.std(ddof=1)/np.sqrt(nb)
        se_pair = d.std(ddof=1)/np.sqrt(nb)
        ny, nx = g.shape[2:]
        nbk = max(1, min(nblk, ny, nx))
        by, bx = ny//nbk, nx//nbk
        gb = g[:, iv, :by*nbk, :bx*nbk].reshape(nb, nbk, by, nbk, bx).mean(axis=(2, 4))
        ob = o[:, iv, :by*nbk, :bx*nbk].reshape(nb, nbk, by, nbk, bx).mean(axis=(2, 4))
        se = np.maximum(ob.std(axis=0, ddof=1)/np.sqrt(nb), 1e-12*max(om.mean(), 1e-30))
        z = (gb.mean(axis=0)-ob.mean(axis=0))/(np.sqrt(2.0)*se)
        z = z[ob.mean(axis=0) > min_block_rel*om.mean()] if min_block_rel > 0.0 else z.ravel()
        if z.size == 0:
            z = np.zeros(1)
        out.append({'view': iv, 'mean_gpu': float(gm.mean()), 'mean_oracle': float(om.mean()), 'diff': float(d.mean()),
                    'se_independent': float(se_ind), 'se_paired': float(se_pair),
                    # (a value that is the same in every batch -- the top level of a flux job reads mu0 exactly -- has no Monte-Carlo error:
                    #  the float32 rounding of the outputs, 1e-6 of the value, stands in as its standard error)
                    'domain_mean_diff_sigma': float(d.mean()/max(se_ind, 1.0e-6*abs(om.mean()), 1e-300)),
                    # (north_star's criterion with the floor of tests/test_gpu_fullsize.py: two sigma of the Monte-Carlo error, plus 0.03 % of the value --
                    #  the float32 arithmetic of the GPU path against the oracle's float64 shows at a few 1e-5 of a flux whose own
                    #  Monte-Carlo error is smaller still: a direct beam, a level every photon crosses)
                    'within_2_sigma': bool(abs(d.mean()) < 2.0*se_ind + 3.0e-4*abs(om.mean())),
                    'paired_rel_diff': float(d.mean()/max(abs(om.mean()), 1e-300)),
                    'paired_diff_in_paired_se': float(d.mean()/max(se_pair, 1.0e-6*abs(om.mean()), 1e-300)),
                    'block_z_mean': float(z.mean()), 'block_z_std': float(z.std()), 'block_abs_z_max': float(np.abs(z).max()),
                    'frac_abs_z_gt_2': float(np.mean(np.abs(z) > 2.0)), 'n_abs_z_ge_4': int(np.sum(np.abs(z) >= 4.0)), 'blocks': int(z.size)})
    return out


def stream_peak():
    """(GB/s of a float4 stream copy on this device -- tools/microbench/stream_copy, a child process after the timed region --, None) or
    (None, why not)."""
    import re
    import subprocess
    exe = os.path.join(ROOT, 'tools', 'microbench', 'stream_copy')
    if not os.path.exists(exe):
        return None, 'tools/microbench/stream_copy is not built (__graft_entry__.build() builds it)'
    try:
        r = subprocess.run([exe, '4', '20'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
        m = re.search(r'stream_copy ([0-9.]+) GB/s', r.stdout)
        if r.returncode == 0 and m:
            return float(m.group(1)), None
        return None, 'stream_copy exited with code %d: %s' % (r.returncode, (r.stderr or r.stdout).strip()[-200:])
    except Exception as e:
        return None, 'stream_copy did not run: %r' % (e,)


def make_scene(workload, base=None):
    from er3t_amd.synth import les_scene, z_levels_config4
    if workload == 'les480_mv9' and base is not None and (base.nx, base.ny, base.nz3) == (480, 480, 100):
        # config 5 on config 4's grid that the caller holds already (the cloud field takes seconds to make): nine views, LSRT surface
        import dataclasses
        from er3t_amd.synth import sfc_lsrt_synth
        vza = np.array([0.0, 26.1, 26.1, 45.6, 45.6, 60.0, 60.0, 70.5, 70.5]); vaa = np.array([0.0, 0.0, 180.0, 0.0, 180.0, 0.0, 180.0, 0.0, 180.0])
        sfc = sfc_lsrt_synth(base.nx, base.ny)
        psfc = np.zeros((5, base.ny, base.nx), dtype=np.float32)
        psfc[:3] = np.transpose(sfc.data['sfc']['data'], (2, 1, 0))
        return dataclasses.replace(base, view_the=list(180.0-vza), view_phi=list((270.0-vaa) % 360.0), view_zloc=[base.view_zloc[0]]*9,
                                   jsfc=np.full((base.ny, base.nx), 4.0, dtype=np.float32), psfc=psfc)
    if workload == 'les480':
        return les_scene(nx=480, ny=480, nz3=100, levels=z_levels_config4(), z_top=1.6, seed=20251004)
    elif workload == 'les128':
        return les_scene()
    elif workload in ('les480_mv9', 'les480_mv9_lambert'):
        # BASELINE config 5: nine MISR-like view zenith angles along track, LSRT land surface
        # (_lambert: the same over a Lambertian surface, for experiments with the ray kernel's light build)
        vza = [0.0, 26.1, 26.1, 45.6, 45.6, 60.0, 60.0, 70.5, 70.5]
        vaa = [0.0, 0.0, 180.0, 0.0, 180.0, 0.0, 180.0, 0.0, 180.0]
        return les_scene(nx=480, ny=480, nz3=100, levels=z_levels_config4(), z_top=1.6, seed=20251004, vza=vza, vaa=vaa,
                         lsrt=(workload == 'les480_mv9'))
    elif workload == 'les128_flux':
        return les_scene(target='flux', aerosol=True)
    elif workload == 'les480_flux':
        # BASELINE config 4's grid as a flux job (117 levels: 5000 bins of tally cells)
        return les_scene(nx=480, ny=480, nz3=100, levels=z_levels_config4(), z_top=1.6, seed=20251004, target='flux')
    elif workload == 'les128_cam':
        # er3t's all-sky camera (mcarats.py:291-296: on the ground, 178 degree cone, 500 x 500 pixels) on the config-2 grid
        sc = les_scene(surface_albedo=0.1)
        sc.rad_kind = 1
        sc.view_the = [0.0]; sc.view_phi = [0.0]; sc.view_zloc = [0.0]
        sc.cam_psi = [0.0]; sc.cam_xpos = [0.5]; sc.cam_ypos = [0.5]
        sc.cam_qmax = [178.0]; sc.cam_umax = [178.0]; sc.cam_vmax = [178.0]; sc.cam_apsize = [0.05]
        sc.nxr = 500; sc.nyr = 500
        return sc
    elif workload == 'les128_mie':
        # config 2's grid with TABULATED phase functions in the cloud (north_star: "LDS-staged phase-function CDF tables"): four Mie-like
        # tables of 498 angles, a real-valued table index per voxel (er3t_amd/synth.py: les_scene(mie=True))
        return les_scene(mie=True)
    elif workload == 'les128_aer':
        # BASELINE config 3, radiance leg: cloud + 3-D aerosol (two 3-D constituents), nadir view
        return les_scene(aerosol=True)
    raise SystemExit('unknown workload %s' % workload)


def algorithmic_bytes(cnt, np3d):
    """4 B per extinction read (transport + local-estimate steps in the 3-D region, and one
    column-table read per LE answered from it), 8 B per scattering component per collision, 8 B per tally RMW."""
    nph = max(cnt['photons'], 1)
    b = 4.0*(cnt['steps3d'] + cnt['le_steps3d'] + cnt['le_column']) + 8.0*np3d*cnt['scatter'] + 8.0*cnt['le_rays'] \
        + 12.0*cnt['flux_tally']
    return b/nph


STREAM_KERNELS = ('k_entry', 'k_bin_', 'k_tl_', 'k_fold_rad', 'k_stats_')      # kernels that read wide coalesced streams


def live_pmc(workload, photons, full=True):
    """HBM traffic and vector-ALU figures of ONE launch of every kernel of a step (photon order, entry records, photon loop, ray
    kernels / record sort, fold), measured now: short child processes, each `rocprofv3 --pmc <one counter group> -- python3
    tools/pmc_run.py <photons> <workload>` (separate passes, nothing combined with tracing, as MI355X_MICROARCH.md prescribes; the
    children are fresh processes -- this one keeps its GPU context and is idle meanwhile).  Returns per-photon figures, None (no
    rocprofv3, BENCH_NO_PMC set) or a string saying which pass failed or timed out: the caller then falls back to the figures
    recorded in profiles/traffic.json and says so, with the reason.

    FETCH_SIZE on gfx950 reports half the bytes of a wide coalesced streaming read (the guide's HBM section); 16-byte gathers are
    counted in full (profiles/r02/fetch_size_calibration_16B_gathers.txt).  So: x 2 on the FETCH_SIZE of the kernels that stream
    (STREAM_KERNELS), and inside the transport kernels the two coalesced streams they read are added once more at their known
    size -- the entry records (48 B per photon when k_entry ran) and the event records the ray kernel reads back (as many bytes as the
    event-writing loop wrote)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3')
    if exe is None or os.environ.get('BENCH_NO_PMC'):
        return None
    tmp = tempfile.mkdtemp(prefix='bench_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp', BENCH_NO_PMC='1')
    vals, byk = {}, {}
    groups = [['FETCH_SIZE'], ['WRITE_SIZE'], ['SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_THREAD_CYCLES_VALU']]
    if full:
        groups += [['TCC_HIT_sum', 'TCC_MISS_sum'], ['SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY']]
    try:
        for group in groups:
            d = os.path.join(tmp, group[0])
            cmd = [exe, '--pmc'] + group + ['-d', d, '-o', 'p', '--output-format', 'csv', '--', sys.executable,
                                            os.path.join(ROOT, 'tools', 'pmc_run.py'), '%d' % photons, workload]
            # (a process group of its own: a pass that hangs is killed WITH the profiled child, which would otherwise keep the GPU)
            pr = subprocess.Popen(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=150)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except Exception:
                    pass
                pr.wait()
                return 'a rocprofv3 --pmc pass (%s) timed out and was killed' % group[0]
            if rc != 0:
                if group[0].startswith('TCC') or group[0].startswith('SQ_WAVE'):
                    continue          # (extras: the traffic figures stand without them)
                return 'a rocprofv3 --pmc pass (%s) failed with exit code %d' % (group[0], rc)
            for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
                for row in csv.DictReader(open(f)):
                    kn = row['Kernel_Name']
                    if 'mi3d::' not in kn:
                        continue          # (runtime helpers: fills and copies)
                    short = kn.split('mi3d::')[1].split('(')[0]
                    if short.startswith(('k_build_', 'k_layer_', 'k_apf_', 'k_xcc_')):
                        continue          # (scene builders: once per scene, not per step)
                    v = float(row['Counter_Value'])
                    vals[row['Counter_Name']] = vals.get(row['Counter_Name'], 0.0) + v
                    byk.setdefault(short, {})
                    byk[short][row['Counter_Name']] = byk[short].get(row['Counter_Name'], 0.0) + v
        if not all(k in vals for k in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_INSTS_VALU', 'SQ_THREAD_CYCLES_VALU')):
            return 'the rocprofv3 --pmc passes returned no rows for the transport kernels'
        n = float(photons)
        is_stream = lambda k: any(k.startswith(q) for q in STREAM_KERNELS)
        fetch_stream = sum(c.get('FETCH_SIZE', 0.0) for k, c in byk.items() if is_stream(k))*1024.0
        fetch_other = sum(c.get('FETCH_SIZE', 0.0) for k, c in byk.items() if not is_stream(k))*1024.0
        entry_b = 48.0*n if any(k.startswith('k_entry') for k in byk) else 0.0
        emit_b = sum(c.get('WRITE_SIZE', 0.0) for k, c in byk.items() if k.startswith('k_transport_lean') and any(q.startswith('k_rays') for q in byk))*1024.0
        fetch_corr = fetch_other + 2.0*fetch_stream + entry_b + emit_b
        write_b = vals['WRITE_SIZE']*1024.0
        hit, miss = vals.get('TCC_HIT_sum'), vals.get('TCC_MISS_sum')
        out = {'fetch_bytes_per_photon': fetch_corr/n, 'write_bytes_per_photon': write_b/n, 'hbm_bytes_per_photon': (fetch_corr + write_b)/n,
               'fetch_bytes_per_photon_as_counted': (fetch_other + fetch_stream)/n,
               'correction_bytes_per_photon': {'streaming_kernels_x2': fetch_stream/n, 'entry_records_read_by_the_photon_loop': entry_b/n,
                                               'event_records_read_by_the_ray_kernel': emit_b/n},
               'valu_insts_per_photon': vals['SQ_INSTS_VALU']/n, 'salu_insts_per_photon': vals.get('SQ_INSTS_SALU', float('nan'))/n,
               'lane_utilisation': vals['SQ_THREAD_CYCLES_VALU']/(64.0*vals['SQ_INSTS_VALU']),
               'tcc_hit_rate': (hit/(hit+miss)) if (hit is not None and miss is not None and hit+miss > 0) else None,
               'photons_of_the_measured_run': n,
               'by_kernel_bytes_per_photon': {k: {'fetch_as_counted': c.get('FETCH_SIZE', 0.0)*1024.0/n, 'write': c.get('WRITE_SIZE', 0.0)*1024.0/n,
                                                  'valu_insts': c.get('SQ_INSTS_VALU', 0.0)/n} for k, c in sorted(byk.items())}}
        if vals.get('SQ_WAVE_CYCLES'):
            # where the waves of the dominant kernel spend their cycles: parked on s_waitcnt (memory, LDS), waiting for an issue slot, issuing
            dom = max(byk.items(), key=lambda kc: kc[1].get('SQ_WAVE_CYCLES', 0.0))
            wc = dom[1]['SQ_WAVE_CYCLES']
            out['sq_active_inst_valu'] = {'kernel': dom[0], 'wait_any_frac': dom[1].get('SQ_WAIT_ANY', 0.0)/wc, 'wait_inst_any_frac': dom[1].get('SQ_WAIT_INST_ANY', 0.0)/wc,
                                          'active_inst_any_frac': dom[1].get('SQ_ACTIVE_INST_ANY', 0.0)/wc}
        return out
    except Exception as e:
        return 'live PMC passes failed: %r' % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


SECONDARY = (('les128', 2.0e8, 'BASELINE config 2'), ('les128_aer', 2.0e8, 'BASELINE config 3, radiance leg'),
             ('les128_flux', 1.0e8, 'BASELINE config 3, flux leg'), ('les480_mv9', 2.0e8, 'BASELINE config 5'),
             ('les480_flux', 5.0e7, 'BASELINE config 4\'s grid as a flux job (480 x 480 columns x 117 levels: the reference\'s default target on the largest grid)'),
             ('les128_mie', 2.0e8, 'config 2 with tabulated phase functions in the cloud (north_star: LDS-staged phase-function tables)'))


def secondary_leg(workload, photons, device, seed, ncore, base_scene=None, min_seconds=3.2, oracle_seconds=3.0, general=False):
    """One of the other BASELINE configurations, measured by the same command as the headline: a handle of its own, a warm-up step (pilot
    launches, list sizing), then as many steps of `photons` histories as fill `min_seconds` -- timed from the first launch to the
    synchronisation after the last, inputs resident; the algorithmic bytes from the instrumented build on a sub-sample; the HIP path
    against the oracle on the same photon ids (a CPU sample of `oracle_seconds`)."""
    from er3t_amd.solver import Mi3dSolver
    from er3t_amd.scene import TARGET_FLUX
    scene = make_scene(workload, base=base_scene)
    is_flux = bool(scene.target & TARGET_FLUX)
    sol = Mi3dSolver(device=device)
    try:
        if general:
            sol.set_kernel(general=True)      # (the general photon loop, k_transport: what serves the jobs the lean loops do not -- VERDICT r5 item 7: a number for it)
        sol.load_scene(scene); sol.set_counting(False); sol.reset()
        P = int(photons)
        sol.run(P, seed=seed, offset=0); sol.sync()                       # warm-up
        sol.reset()
        tp = time.perf_counter(); sol.run(P, seed=seed, offset=P); sol.sync(); t1 = time.perf_counter()-tp
        steps = int(max(2, min(64, np.ceil(min_seconds/max(t1, 1.0e-3)))))
        sol.reset(); sol.sync()
        tp = time.perf_counter()
        for i in range(steps):
            sol.run(P, seed=seed, offset=(2+i)*P)
        sol.sync()
        dt = time.perf_counter()-tp
        kernel_ms, launches = sol.timing()
        kname = sol.kernel_name()
        nsub = 1000000
        sol.set_counting(True); sol.reset(); sol.run(nsub, seed=seed, offset=2*P); sol.sync()
        cnt = sol.counters()
        bpp = algorithmic_bytes(cnt, scene.np3d)
        # (flux legs: the record sort of one step runs beside the photon loop of the next on a stream of its own, so the launches' own
        #  intervals overlap and their sum exceeds the wall time: a launch then counts for its share of the wall time)
        overlapped = kernel_ms > 1.0e3*dt
        avg_ms = min(kernel_ms, 1.0e3*dt)/max(launches, 1)
        achieved = bpp*(P*steps/max(launches, 1))/(avg_ms*1.0e-3)/1.0e9
        # frac: the dominant kernel's own intervals (HIP events around every launch of the photon loop and what it queues behind itself);
        # frac_step: the same bytes over the WALL time of the timed steps -- photon order, entry records, sorts, folds, gaps and tails included
        achieved_step = bpp*P*steps/dt/1.0e9
        leg = {'value': P*steps/dt, 'unit': 'photons/s', 'ms_per_step': 1.0e3*dt/steps, 'steps': steps, 'photons_per_step': P, 'timed_s': dt,
               'kernel': kname, 'launches': launches, 'avg_launch_ms': avg_ms, 'bytes_per_photon': bpp,
               'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved/HBM_PEAK_GBS,
                            'achieved_step': achieved_step, 'frac_step': achieved_step/HBM_PEAK_GBS},
               'views': scene.nview, 'target': 'flux' if is_flux else 'radiance'}
        leg['avg_launch_ms_how'] = ('wall time of the timed steps / launches (consecutive launches overlap: the previous launch\'s record sort runs beside the photon loop)' if overlapped else
                                    'HIP events on the launch stream around each launch of the photon loop and the kernels queued behind it (ray kernels, record sort); the pre-pass '
                                    '(photon order, entry records) and the fold are outside the bracket: frac_step has them')
        if ncore > 0:
            from oracle import oracle
            sol.set_counting(False)
            tp = time.perf_counter()
            oracle.run_raw(scene, 20000, seed=seed, offset=0, nthreads=ncore)
            rate = 20000/(time.perf_counter()-tp)
            nbatch = 8
            nper = int(min(max(rate*oracle_seconds, 1.6e4), 2.0e7))//nbatch
            gimg, oimg = [], []
            for b in range(nbatch):
                osum = oracle.run_raw(scene, nper, seed=seed, offset=2*P + b*nper, nthreads=ncore)
                orad, oflux = oracle.normalise(scene, osum[0], osum[1], nper)
                sol.reset(); sol.run(nper, seed=seed, offset=2*P + b*nper); sol.sync()
                if is_flux:
                    gimg.append(sol.flux(nper).astype(np.float64).reshape(-1, scene.ny, scene.nx)); oimg.append(oflux.reshape(-1, scene.ny, scene.nx))
                else:
                    gimg.append(sol.radiance(nper).astype(np.float64)); oimg.append(orad)
            ps = parity_stats(np.stack(gimg), np.stack(oimg))
            worst = max(ps, key=lambda q: abs(q['domain_mean_diff_sigma']))
            worst_p = max(ps, key=lambda q: abs(q['paired_diff_in_paired_se']))
            leg['parity'] = {'photons': nper*nbatch, 'batches': nbatch, 'same_photon_ids': True, 'domain_mean_diff_sigma': worst['domain_mean_diff_sigma'],
                             'paired_rel_diff': worst['paired_rel_diff'], 'paired_diff_in_paired_se': worst['paired_diff_in_paired_se'],
                             'worst_paired_diff_in_paired_se': worst_p['paired_diff_in_paired_se'], 'worst_paired_rel_diff': worst_p['paired_rel_diff'],
                             'worst_of': '%d %s' % (len(ps), 'flux variables x levels' if is_flux else 'views'),
                             'within_tolerance': bool(all(q['within_2_sigma'] for q in ps)),
                             'within_tolerance_how': '|difference of the domain means| < 2 sigma (two independent estimates of this size) + 0.03 % of the value, for every one of them',
                             'note': 'paired: same photon ids on both sides, batch by batch -- most of the Monte-Carlo noise cancels; the criterion of tests/test_gpu_fullsize.py is |paired difference| < 4 paired standard errors + 0.03 %'}
        return leg
    finally:
        sol.close()


def published_case_leg(device, ncore=0):
    """The one case the reference publishes a wall-clock for (docs/source/other/contest.rst:15-28: `00_er3t_mca.py - example_05`, 45 s on 24 CPUs
    of CU Research Computing, 133 s on an 8-core M2; case definition examples/00_er3t_mca.py:38-39,973,1038-1059): 3-D radiance of the LES field
    coarsened 25 : 1 in z -- 480 x 480 x 4 voxels --, Mie cloud through `mca_sca`, 16 g x 3 runs x 1e8 photons per run, nadir view, END TO END through the
    drop-in classes as a user runs it: `mcarats_ng` (48 job files written, every job transported, 48 output files written) + `mca_out_ng` (files
    read back, g-sum, mean and std over the runs).  Synthetic stand-ins for the LES field and the data bases (er3t_amd/synth.py).  Context, not
    target: other hardware, another implementation of the solver."""
    import contextlib
    import datetime
    import io
    import shutil
    import tempfile
    import er3t_amd.rtm.mca as mca
    from er3t_amd.rtm.mca.mca_exe import get_runner
    from er3t_amd.synth import atm_synth, abs_synth, cld_synth, pha_mie_synth

    def quiet(fn, *a, **k):
        with contextlib.redirect_stdout(io.StringIO()):
            return fn(*a, **k)

    tmp = tempfile.mkdtemp(prefix='bench_case_', dir='/tmp')
    try:
        t0 = time.perf_counter()
        atm = atm_synth(np.linspace(0.0, 20.0, 21))
        ab = abs_synth(650.0, atm, Ng=16)
        cld = cld_synth(atm, nx=480, ny=480, nz=4, z_base=0.2, z_top=1.8, cot_mean=10.0, seed=20251004)
        pha = pha_mie_synth(650.0)
        a1 = quiet(mca.mca_atm_1d, atm_obj=atm, abs_obj=ab)
        sca = quiet(mca.mca_sca, pha_obj=pha, fname=os.path.join(tmp, 'mca_sca.bin'), overwrite=True)
        a3 = quiet(mca.mca_atm_3d, atm_obj=atm, cld_obj=cld, pha_obj=pha, fname=os.path.join(tmp, 'mca_atm_3d.bin'), overwrite=True)
        t_setup = time.perf_counter()-t0
        kw = dict(date=datetime.datetime(2017, 8, 13), atm_1ds=[a1], atm_3ds=[a3], Ng=16, target='radiance', surface_albedo=0.03, sca=sca,
                  solar_zenith_angle=30.0, solar_azimuth_angle=45.0, sensor_zenith_angle=0.0, sensor_azimuth_angle=0.0, sensor_altitude=705000.0,
                  Nrun=3, weights=ab.coef['weight']['data'], solver='3D', Ncpu='auto', mp_mode='py', overwrite=True, quiet=True)
        # warm-up on the same grid (the runner's handle, its scene buffers and list sizes: what a second call of the same script finds)
        quiet(mca.mcarats_ng, fdir=os.path.join(tmp, 'warm'), photons=2.0e6, **kw)
        secs = []
        for rep in range(2):
            t0 = time.perf_counter()
            m = quiet(mca.mcarats_ng, fdir=os.path.join(tmp, 'rad_3d_%d' % rep), photons=1.0e8, **kw)
            t1 = time.perf_counter()
            out = quiet(mca.mca_out_ng, mca_obj=m, abs_obj=ab, mode='mean', squeeze=True, quiet=True).data      # (no result cache: h5py is not part of this image)
            t2 = time.perf_counter()
            secs.append((t2-t0, t1-t0, t2-t1))
        best = min(secs)
        rad = np.asarray(out['rad']['data'], dtype=np.float64)
        return {'seconds': best[0], 'seconds_mcarats_ng': best[1], 'seconds_mca_out_ng': best[2], 'seconds_all_repeats': [q[0] for q in secs],
                'photons': 3.0e8, 'jobs': int(m.Nrun*m.Ng), 'voxels': '480x480x4', 'kernel': get_runner().sol.kernel_name(),
                'photons_per_s_end_to_end': 3.0e8/best[0], 'setup_seconds_not_timed': t_setup,
                'mean_radiance': float(rad.mean()), 'rad_std_over_runs_rel': float(np.asarray(out['rad_std']['data']).mean()/max(rad.mean(), 1e-30)),
                'reference_published': {'cu_research_computing_24_cpus_s': 45.0, 'macbook_air_m2_8_cpus_s': 133.0,
                                        'source': 'docs/source/other/contest.rst:15-28 (00_er3t_mca.py - example_05)'},
                'what': 'mcarats_ng (48 job files written, 48 jobs transported, 48 output files written) + mca_out_ng (files read back, g-sum, mean and std over 3 runs); '
                        'synthetic LES field and Mie table; context, not a target: other hardware, another implementation of the solver'}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def sclk_mhz():
    """the shader clock of device 0 right now, MHz (`rocm-smi --showclocks`: the level marked current), or None"""
    import re
    import shutil
    import subprocess
    exe = shutil.which('rocm-smi') or '/opt/rocm/bin/rocm-smi'
    try:
        r = subprocess.run([exe, '-d', '0', '--showclocks'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=20)
        m = re.search(r'sclk clock level[^\n]*?\((\d+)\s*Mhz\)', r.stdout, re.IGNORECASE)
        return float(m.group(1)) if m else None
    except Exception:
        return None


def gpu_count_sysfs():
    """GPUs of this machine counted WITHOUT opening the HIP runtime (the parent of an N-rank run must not touch the GPU before it
    starts its ranks): the KFD topology nodes that have SIMDs.  None where the topology is not readable."""
    import glob
    n, seen = 0, False
    for f in glob.glob('/sys/class/kfd/kfd/topology/nodes/*/properties'):
        try:
            with open(f) as fh:
                seen = True
                for ln in fh:
                    if ln.startswith('simd_count'):
                        n += int(ln.split()[1]) > 0
        except Exception:
            pass
    if not seen:
        return None
    # a process sees no more devices than the runtime's masks leave it (cgroup masking shows in the topology itself)
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES', 'GPU_DEVICE_ORDINAL'):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([q for q in v.split(',') if q.strip() != '']))
    return n


def spawn(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh processes under torch.distributed.run and
    relay rank 0's JSON line.  This process never touches the GPU: it counts the devices in the KFD topology (sysfs), and every
    rank's output goes to a log file of its own, so that a failed run can show what EACH rank said last."""
    import glob
    import shutil
    import socket
    import subprocess
    import tempfile
    if not args.dry_run:
        have = gpu_count_sysfs()
        if have is None:
            import torch
            have = torch.cuda.device_count()      # (no KFD topology to read: counting devices does not initialise the runtime on this image)
        if have < args.gpus:
            print('bench.py: --gpus %d asked for but this machine shows %d GPU(s); refusing to report a %d-GPU number '
                  'from fewer devices' % (args.gpus, have, args.gpus), file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    logdir = tempfile.mkdtemp(prefix='bench_ranks_', dir='/tmp')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), '--log-dir', logdir, '--redirects', '3',
           os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)

    def rank_logs():
        logs = {}
        for f in glob.glob(os.path.join(logdir, '**', '*.log'), recursive=True):
            rk = os.path.basename(os.path.dirname(f))
            logs.setdefault(rk, {})[os.path.basename(f)] = f
        return logs

    line = None
    for rk, files in rank_logs().items():
        f = files.get('stdout.log')
        if f:
            for ln in open(f, errors='replace').read().splitlines():
                if ln.startswith('{') and '"metric"' in ln:
                    line = ln
    for ln in r.stdout.splitlines():              # (a launcher that did not redirect: the line is on its own stdout)
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
    rc = 0
    if r.returncode != 0 or line is None:
        print('bench.py: the %d-rank run failed (exit code %d); last lines of the launcher and of every rank:' % (args.gpus, r.returncode), file=sys.stderr)
        sys.stderr.write('\n'.join(r.stdout.splitlines()[-20:]) + '\n')
        for rk, files in sorted(rank_logs().items()):
            for name, f in sorted(files.items()):
                tail = open(f, errors='replace').read().splitlines()[-20:]
                if tail:
                    sys.stderr.write('---- rank %s %s\n%s\n' % (rk, name, '\n'.join(tail)))
        rc = r.returncode or 1
    else:
        print(line)
    shutil.rmtree(logdir, ignore_errors=True)
    return rc


def dry_run(args, world, rank, use_dist):
    """The launcher, the rendezvous, one all-reduce per step and the JSON line WITHOUT any transport: a CPU test of the N-rank
    plumbing (tests/test_dist_gloo.py).  The line carries `"data": "dry-run"` and a null value: it is not a measurement."""
    import torch
    import torch.distributed as dist
    from er3t_amd.dist import photon_shard
    if use_dist:
        dist.init_process_group('gloo')
    P = int(args.photons)
    Ptot = world*P if args.scaling == 'weak' else P
    done = torch.zeros(1, dtype=torch.float64)
    for i in range(args.steps):
        off, n = photon_shard(Ptot, world, rank)
        t = torch.tensor([float(n)], dtype=torch.float64)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        done += t
    if rank == 0:
        print(json.dumps({'metric': 'photons/sec', 'value': None, 'unit': 'photons/s', 'n_gpus': world, 'steps': args.steps,
                          'warmup': args.warmup, 'higher_is_better': True, 'scaling': args.scaling, 'data': 'dry-run',
                          'rccl_ranks': dist.get_world_size() if use_dist else 1, 'backend': 'gloo' if use_dist else None,
                          'config': {'workload': args.workload, 'photons_per_step': Ptot, 'photon_ids_covered': float(done.item())}}))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--photons', type=float, default=1.0e9, help='photon histories per step: per GPU (weak) or in all (strong)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'])
    ap.add_argument('--workload', default='les480', choices=["les480", "les128", "les480_mv9", "les128_flux", "les128_aer", "les480_mv9_lambert", "les480_flux", "les128_cam", "les128_mie"])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='headline workload only: no short legs of the other BASELINE configurations')
    ap.add_argument('--no-pmc', action='store_true', help='no live rocprofv3 --pmc passes after the timed region: traffic figures replayed from profiles/traffic.json')
    ap.add_argument('--march-le', action='store_true', help='march every local-estimate ray (no column table)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help='gloo: rehearsal of the N-rank plumbing')
    ap.add_argument('--dry-run', action='store_true',
                    help='no transport at all (CPU test of the launcher and the exchange): the line says so and is not a measurement')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn(args, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d: start one rank per GPU (torch.distributed.run --nproc-per-node %d, '
                         'or plain `python bench.py --gpus %d`)' % (args.gpus, world, args.gpus, args.gpus))
    use_dist = world > 1 or ('RANK' in os.environ and 'MASTER_ADDR' in os.environ)   # launched by torch.distributed.run
    if args.dry_run:
        return dry_run(args, world, rank, use_dist)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (no CPU fallback)')
    if torch.cuda.device_count() < world and args.backend == 'nccl':
        raise SystemExit('bench.py: %d ranks but %d GPU(s): one rank per GPU' % (world, torch.cuda.device_count()))
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if use_dist:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group('gloo')

    from er3t_amd.solver import Mi3dSolver
    from er3t_amd.dist import photon_shard

    # photons per step: per rank (weak scaling) or in all (strong scaling)
    P = int(args.photons)
    Ptot = world*P if args.scaling == 'weak' else P
    scene = make_scene(args.workload)
    sol = Mi3dSolver(device=local_rank)
    from er3t_amd.scene import TARGET_FLUX
    is_flux = bool(scene.target & TARGET_FLUX)
    # the raw tallies are float64: the radiance image, or the three flux planes per level for a flux workload
    rad = torch.zeros(3*(scene.nz+1)*scene.ny*scene.nx if is_flux else scene.nview*scene.nyr*scene.nxr, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev)
    if is_flux:
        sol.bind(flux_ptr=rad.data_ptr(), stream=stream.cuda_stream)
    else:
        sol.bind(rad_ptr=rad.data_ptr(), stream=stream.cuda_stream)
    sol.load_scene(scene, column_le=not args.march_le)
    sol.set_counting(False)
    sol.reset()
    seed = 1234
    red = rad if args.backend == 'nccl' else None

    def step(istep):
        rad.zero_()
        off, n = photon_shard(Ptot, world, rank)               # contiguous id ranges, one per rank
        sol.run(n, seed=seed, offset=istep*Ptot + off)
        if use_dist:
            if red is not None:
                dist.all_reduce(red, op=dist.ReduceOp.SUM)     # RCCL, in place on the tally the kernel wrote
            else:                                              # gloo rehearsal: through host memory
                torch.cuda.synchronize(dev)
                h = rad.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                rad.copy_(h)

    sclk_mid = None
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    sol.timing()
    sol.reset()

    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
        if i == args.steps//2 and rank == 0 and world == 1 and args.steps >= 4:
            sclk_mid = sclk_mhz()      # (the shader clock UNDER the timed load: a child process beside the queued launches, the host does not wait for the device here)
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_ms, launches = sol.timing()
    kernel_name = sol.kernel_name()

    # ---- N > 1: BASELINE config 4 as it is written -- `--photons` IN ALL, sharded over the ranks -- timed beside the weak-scaling headline
    #      (the driver runs `--scaling weak`; the strong line is the config's own wording).  Same barrier / max-over-ranks bracket.
    strong = None
    if world > 1 and args.scaling == 'weak' and not args.no_secondary:
        Pst = P
        ks = max(2, args.steps//2)

        def step_strong(istep):
            rad.zero_()
            off, n = photon_shard(Pst, world, rank)
            sol.run(n, seed=seed+1, offset=istep*Pst + off)
            if red is not None:
                dist.all_reduce(red, op=dist.ReduceOp.SUM)
            else:
                torch.cuda.synchronize(dev); hh = rad.cpu(); dist.all_reduce(hh, op=dist.ReduceOp.SUM); rad.copy_(hh)
        step_strong(0)
        torch.cuda.synchronize(dev); dist.barrier()
        ts0 = time.perf_counter()
        for i in range(ks):
            step_strong(1+i)
        torch.cuda.synchronize(dev); dist.barrier()
        es = time.perf_counter()-ts0
        tt = torch.tensor([es], dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        es = float(tt.item())
        strong = {'scaling': 'strong', 'photons_per_step': Pst, 'photons_per_gpu_per_step': photon_shard(Pst, world, rank)[1], 'steps': ks,
                  'value': Pst*ks/es, 'unit': 'photons/s', 'ms_per_step': 1.0e3*es/ks,
                  'what': 'BASELINE config 4 as written: %g photons in all, sharded over %d GPUs, one all-reduce per step' % (Pst, world)}
        sol.timing()

    # sanity: the tally of the last step is finite and positive
    mean_rad = float(rad.sum().item())*scene.src_flx*scene.mu0/Ptot
    if not (mean_rad > 0.0 and np.isfinite(mean_rad)):
        raise SystemExit('bench.py: radiance tally is not finite/positive (%r)' % mean_rad)

    if rank == 0:
        # ---- algorithmic bytes per photon: instrumented build on a sub-sample of the timed ids
        nsub = min(P, 2000000)
        sol.bind(rad_ptr=None, flux_ptr=None, stream=stream.cuda_stream)
        sol.set_counting(True)
        sol.reset()
        sol.run(nsub, seed=seed, offset=args.warmup*Ptot)
        sol.sync()
        cnt = sol.counters()
        bpp = algorithmic_bytes(cnt, scene.np3d)
        # a step is transported in launches of at most 2^30 photons (the photon order of a launch is sorted by start tile):
        # the roofline figure is per launch of the transport kernel, averaged over the timed launches of this rank
        avg_ms = kernel_ms/max(launches, 1)
        n_rank = photon_shard(Ptot, world, rank)[1]
        per_launch = n_rank*args.steps/max(launches, 1)
        achieved = bpp*per_launch/(avg_ms*1.0e-3)/1.0e9
        traffic = None
        traffic_src = None
        valu = None
        note = ('(FETCH_SIZE + WRITE_SIZE) x 1024 summed over EVERY kernel of a step (photon order, entry records, photon loop, ray kernels / record sort, fold); '
                'FETCH_SIZE x 2 for the kernels that read wide coalesced streams, the entry and event records the transport kernels read as streams added '
                'once more at their known size (gfx950 counts such reads at half, MI355X_MICROARCH.md); no x 2 on the 16-byte voxel gathers '
                '(calibrated: profiles/r02/fetch_size_calibration_16B_gathers.txt)')
        t, source, why_not = None, None, None
        if world == 1 and not args.no_pmc:
            # measured now: one launch of this workload's size (at most 5e8 photons) under rocprofv3 --pmc, in child processes
            t = live_pmc(args.workload, int(min(per_launch, 5.0e8)))
            source = 'measured in this run: child processes under rocprofv3 --pmc, one counter group per pass (bench.py: live_pmc)'
            if isinstance(t, str):
                why_not, t = t, None
        if t is None:
            ftraffic = os.path.join(ROOT, 'profiles', 'traffic.json')
            try:
                with open(ftraffic) as f:
                    tj = json.load(f)
                if args.workload in tj:
                    # the per-photon figures of the rocprofv3 --pmc passes recorded in profiles/traffic.json (same workload; its
                    # `session` field says when), scaled to this run's photons per launch
                    t = tj[args.workload]
                    source = 'replayed from profiles/traffic.json (session: %s)' % t.get('session') + (' -- %s' % why_not if why_not else '')
            except Exception:
                t = None
        if t is not None:
            traffic = t['hbm_bytes_per_photon']*per_launch
            traffic_src = {'source': source, 'fetch_bytes_per_photon': t.get('fetch_bytes_per_photon'),
                           'write_bytes_per_photon': t.get('write_bytes_per_photon'), 'tcc_hit_rate': t.get('tcc_hit_rate'),
                           'photons_of_the_measured_run': t.get('photons_of_the_measured_run'), 'correction': note,
                           'fetch_bytes_per_photon_as_counted': t.get('fetch_bytes_per_photon_as_counted'),
                           'correction_bytes_per_photon': t.get('correction_bytes_per_photon'), 'by_kernel_bytes_per_photon': t.get('by_kernel_bytes_per_photon')}
            if 'valu_insts_per_photon' in t:
                # vector-ALU issue against the MEASURED ceiling of the loop's instruction mix (VALU_MIX_CEILING above): <= 1 by construction
                # for a kernel with that mix; the other workloads' kernels are priced against the same ceiling (their mixes are close: the
                # same walk and collision code)
                v = t['valu_insts_per_photon']
                valu = {'wave_insts_per_photon': v, 'scalar_insts_per_photon': t.get('salu_insts_per_photon'), 'lane_utilisation': t.get('lane_utilisation'),
                        'wave_insts_per_second': v*per_launch/(avg_ms*1.0e-3), 'ceiling_wave_insts_per_second': VALU_MIX_CEILING,
                        'ceiling_how': 'tools/microbench/mix_rates mix <class counts of the headline kernel>, six waves per SIMD (profiles/r05/mix_rates_mix.log)',
                        'issue_frac': v*per_launch/(avg_ms*1.0e-3)/VALU_MIX_CEILING, 'wave_cycles': t.get('sq_active_inst_valu'), 'source': source}
        peak_meas, peak_why_not = stream_peak() if world == 1 else (None, 'measured at N=1 only')

        out = {
            'metric': 'photons/sec', 'value': Ptot*args.steps/elapsed, 'unit': 'photons/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1.0e3*elapsed/args.steps,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'rccl_ranks': dist.get_world_size() if use_dist else 1, 'backend': args.backend if use_dist else None,
            'config': {'workload': '%s: %dx%dx%d-voxel LES cloud domain (Atm_nz=%d), %s, HG g=0.85 + Rayleigh%s'
                                   % (args.workload, scene.nx, scene.ny, scene.nz3, scene.nz,
                                      {'les480': 'nadir radiance', 'les128': 'nadir radiance', 'les480_mv9': 'nine view zenith angles',
                                       'les128_flux': 'flux + 3-D aerosol', 'les128_aer': 'nadir radiance + 3-D aerosol',
                                       'les480_mv9_lambert': 'nine view zenith angles', 'les480_flux': 'flux', 'les128_cam': 'all-sky camera on the ground, 500 x 500 pixels',
                                       'les128_mie': 'nadir radiance, tabulated (Mie-like) phase functions in the cloud'}[args.workload],
                                      ', LSRT surface' if args.workload == 'les480_mv9' else ', Lambert 0.03'),
                       'photons_per_step': Ptot, 'photons_per_gpu_per_step': n_rank, 'views': scene.nview, 'target': 'flux' if is_flux else 'radiance',
                       'local_estimate': ('none (flux job)' if is_flux else 'marched' if args.march_le else
                                          'marched to the camera (event lists + ray kernel)' if args.workload == 'les128_cam' else
                                          'column table for the nadir view, the eight slant views marched (event lists + ray kernel)' if args.workload.startswith('les480_mv9') else
                                          'column-table (exact for nadir)'),
                       'parallelism': 'photon-sharded x%d, 1 all-reduce/step' % world if world > 1 else 'single GPU',
                       'tallies': ('float64 sums of 8-byte level-crossing records, sorted and summed in LDS after every launch' if 'k_tl_scatter' in kernel_name
                                   else 'float32 sums per workgroup in LDS around the photons\' tile (tally window), added to the float64 image by atomics; tallies outside the window: float64 atomics'
                                   if args.workload in ('les480', 'les128', 'les128_aer', 'les128_mie') and not args.march_le else 'float64 atomics') + ' (arithmetic of the path: float32)', 'mean_radiance': mean_rad,
                       'le_roulette': {'tau1': getattr(scene, 'le_tau1', 0.0), 'cmin': getattr(scene, 'le_cmin', 0.0),
                                       'note': 'unbiased Russian roulettes on marched local-estimate rays (none on column-table views)'}},
            # `bound` / `frac`: the HBM roofline SURVEY.md §8(d) prescribes for this path.  `bound_actual`: what the dominant kernel of
            # the workload runs into on this chip: the voxel reads are L2 hits, so none of them is HBM-bound
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved/HBM_PEAK_GBS, 'peak_measured': peak_meas, 'frac_of_measured': (achieved/peak_meas) if peak_meas else None,
                         'peak_measured_how': 'float4 stream copy, read + written bytes per second (tools/microbench/stream_copy), after the timed region' if peak_meas else None,
                         'peak_measured_why_not': peak_why_not,
                         'traffic': traffic, 'traffic_source': traffic_src,
                         'bound_actual': {'les480': 'valu_issue: %s of the measured ceiling of its own instruction mix at six waves per SIMD (profiles/r05/mix_rates_mix.log); the rest: waves parked on the '
                                                    'walk\'s 16-byte reads (SQ_WAIT_ANY 46-47 %% of the wave cycles, profiles/r05/pmc_busy_wait_les480.txt); by ablation a gathered line costs 0.68 CU-clocks, the walk\'s 61.5 lines a fifth '
                                                    'of a photon\'s 208, with the texture path busy 12 %% of the cycles (profiles/r06/ab_gather_sensitivity_les480.log): vector instructions at half-empty lanes are the rest' % (('%.0f %%' % (100.0*valu['issue_frac'])) if valu and valu.get('issue_frac') else '87-90 %'), 'les128': 'valu_issue', 'les128_aer': 'valu_issue',
                                          'les128_flux': 'valu_issue (photon loop) with the sort of the previous launch\'s tally records beside it on a stream of its own (memory latency; one workgroup per CU fits beside the loop\'s four)',
                                          'les480_mv9': 'memory latency: the ray kernel\'s and the event-writing loop\'s waves are parked on reads 54-59 % of their cycles (profiles/r05/pmc_wait_mv9.txt); until round 5 the L2, which 1.1 KB of event records per photon swept clean (non-temporal since)',
                                          'les480_mv9_lambert': 'valu_issue + l2_gather_rate',
                                          'les480_flux': 'valu_issue (photon loop) + memory latency (sort of the tally records: a third of the time; its 5000-bin tables leave it no room beside the loop)',
                                          'les128_cam': 'valu_issue (photon loop, start batches) + l2_gather_rate (the rays\' walk)',
                                          'les128_mie': 'valu_issue + LDS look-ups of the phase tables'}[args.workload],
                         'achieved_step': bpp*n_rank/(1.0e-3*1.0e3*elapsed/args.steps)/1.0e9,
                         'frac_step': bpp*n_rank/(elapsed/args.steps)/1.0e9/HBM_PEAK_GBS,
                         'frac_step_how': 'the same algorithmic bytes over ms_per_step (the wall time of a whole step: photon order, entry records, photon loop, fold, zeroing, gaps and tails, '
                                          'the all-reduce for N > 1) -- what `value` is measured over; `frac` brackets the dominant kernel alone',
                         'kernel': kernel_name, 'avg_launch_ms': avg_ms, 'launches': launches,
                         'avg_launch_ms_how': 'HIP events recorded inside libmi3drt on the launch stream around each launch of the photon loop (and the ray kernels / record sort queued behind it); '
                                              'since round 5 the bracket does NOT contain the pre-pass of the launch (k_bin_*: photon order; k_entry: entry records; ~15 ms per 1e9 photons), '
                                              'the fold of the accumulation image or the zeroing of the tally: `frac_step` has them',
                         'sclk_mhz': {'during_the_timed_steps': sclk_mid, 'after': sclk_mhz() if world == 1 else None,
                                      'how': 'rocm-smi --showclocks, the level marked current: a child process started half way through the queued steps / after the timed region',
                                      'mix_ceiling_measured_at_mhz': 2160.0},
                         'photons_per_launch': per_launch,
                         'bytes_per_photon': bpp, 'valu': (dict(valu, issue_frac_at_the_clock_of_this_run=(valu['issue_frac']*2160.0/sclk_mid if (valu.get('issue_frac') and sclk_mid) else None),
                                                                 issue_frac_at_the_clock_of_this_run_how='the ceiling was measured at 2160 MHz (tools/microbench/mix_rates reads s_memtime against s_memrealtime); the chip holds '
                                                                 'a higher clock under this loop, which waits on memory half of the time -- the ceiling scales with the clock, the fraction with its inverse: the 0.90 by rate and the '
                                                                 '0.77 per busy cycle of VERDICT r5 (SQ_ACTIVE_INST_VALU x 4 / SQ_BUSY_CYCLES: 35.0 against 45.6) are one number once both sides are counted in cycles') if valu else None),
                         'per_photon': {k: cnt[k]/nsub for k in ('steps3d', 'le_steps3d', 'le_column', 'scatter', 'surface', 'le_rays', 'flux_tally')}},
        }

        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle
            ncore = os.cpu_count() or 1
            try:
                ncore = len(os.sched_getaffinity(0))
            except Exception:
                pass
            # a 1-GPU box grants a 16-core CPU share however many logical CPUs it shows
            ncore = int(os.environ.get('BENCH_CPU_THREADS', min(ncore, 16)))
            tp = time.perf_counter()
            oracle.run_raw(scene, 50000, seed=seed, offset=0, nthreads=ncore)
            pilot = 50000/(time.perf_counter()-tp)
            nbatch = 8
            nper = int(min(max(pilot*15.0, 1.0e5), 2.0e7))//nbatch
            nsample = nper*nbatch
            off0 = args.warmup*Ptot
            tp = time.perf_counter()
            osum = [oracle.run_raw(scene, nper, seed=seed, offset=off0 + b*nper, nthreads=ncore) for b in range(nbatch)]
            dt = time.perf_counter()-tp
            out['cpu_baseline'] = {'value': nsample/dt, 'unit': 'photons/s', 'cores': ncore, 'kind': 'port',
                                   'sample': '%d photon ids of the first timed step of the same scene in %d batches, oracle/mi3d_oracle.c '
                                             '(double precision, every local-estimate ray marched), OpenMP %d threads, %.1f s'
                                             % (nsample, nbatch, ncore, dt)}
            # ---- "radiance sigma-error": the HIP path on the SAME photon ids, batch by batch, against what the oracle has just returned
            sol.set_counting(False)
            gimg, oimg = [], []
            for b in range(nbatch):
                sol.reset(); sol.run(nper, seed=seed, offset=off0 + b*nper); sol.sync()
                orad, oflux = oracle.normalise(scene, osum[b][0], osum[b][1], nper)
                if is_flux:
                    # the three flux variables at every level as "views" of (ny, nx) images: direct-down, total-down, up
                    gimg.append(sol.flux(nper).astype(np.float64).reshape(-1, scene.ny, scene.nx))
                    oimg.append(oflux.reshape(-1, scene.ny, scene.nx))
                else:
                    gimg.append(sol.radiance(nper).astype(np.float64)); oimg.append(orad)
            ps = parity_stats(np.stack(gimg), np.stack(oimg))
            worst = max(ps, key=lambda q: abs(q['domain_mean_diff_sigma']))
            out['parity'] = {'against': 'oracle/mi3d_oracle.c (CPU restatement, float64; unpinned against MCARaTS itself: DESIGN.md §2)',
                             'photons': nsample, 'batches': nbatch, 'same_photon_ids': True, 'tolerance_sigma': 2.0,
                             'within_tolerance': bool(all(q['within_2_sigma'] for q in ps)),
                             'within_tolerance_how': '|difference of the domain means| < 2 sigma (two independent estimates of this size) + 0.03 % of the value (the float32 floor of tests/test_gpu_fullsize.py), for every one of them',
                             'domain_mean_diff_sigma': worst['domain_mean_diff_sigma'], 'paired_rel_diff': worst['paired_rel_diff'],
                             'block_z_mean': worst['block_z_mean'], 'block_z_std': worst['block_z_std'], 'frac_abs_z_gt_2': worst['frac_abs_z_gt_2'],
                             'worst_of': '%d %s' % (len(ps), 'flux variables x levels' if is_flux else 'views'),
                             'per_view': ps if len(ps) <= 16 else None,
                             'note': 'sigma = standard error of the difference of two independent estimates of this size; the two sides follow the '
                                     'same histories, so block z-scores far below 1 are expected (1 would be two independent runs)'}
        else:
            out['cpu_baseline'] = None
            out['parity'] = None
        # ---- the other BASELINE configurations, in the same command and the same JSON line: short legs after the headline's
        # timed region; the headline fields above are untouched by them
        if world == 1 and args.workload == 'les480' and not args.no_secondary:
            ncore_s = 0
            if not args.no_cpu_baseline:
                try:
                    ncore_s = len(os.sched_getaffinity(0))
                except Exception:
                    ncore_s = os.cpu_count() or 1
                ncore_s = int(os.environ.get('BENCH_CPU_THREADS', min(ncore_s, 16)))
            sec = {}
            for name, nph, what in SECONDARY:
                try:
                    # (nine views with every local-estimate ray marched by the oracle: twice the CPU sample, or its 2-sigma bound is 1 % of the mean)
                    leg = secondary_leg(name, nph, local_rank, seed, ncore_s, base_scene=scene, oracle_seconds=(6.0 if name == 'les480_mv9' else 3.0))
                    leg['config'] = what
                    sec[name] = leg
                except Exception as e:          # (a leg that fails says so in the line; the headline stands)
                    sec[name] = {'error': repr(e)[:300], 'config': what}
            # the general photon loop (k_transport: flux together with radiance, more than two 3-D constituents, tables beyond the LDS) on config 3's
            # radiance leg, beside the lean loop's figure for the same scene above: what a job that lands there pays (no oracle leg: same scene)
            try:
                leg = secondary_leg('les128_aer', 1.0e8, local_rank, seed, 0, base_scene=scene, min_seconds=1.5, general=True)
                leg['config'] = 'BASELINE config 3, radiance leg, through the GENERAL photon loop (mi3d_set_kernel 1)'
                sec['les128_aer_general_kernel'] = leg
            except Exception as e:
                sec['les128_aer_general_kernel'] = {'error': repr(e)[:300]}
            out['secondary'] = sec
            # ---- the one case the reference publishes a wall-clock for, end to end through the drop-in classes (context, not target)
            try:
                sol.close()          # (the headline's handle and its buffers go first: the job runner holds handles of its own)
                out['published_case'] = published_case_leg(local_rank)
            except Exception as e:
                out['published_case'] = {'error': repr(e)[:300]}
        if strong is not None:
            out['strong'] = strong
        print(json.dumps(out))

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
