"""
Job runner: executes the (run, g) jobs of a simulation.

Counterpart of the reference's `mca_run` (er3t/rtm/mca/mca_run.py:17-181).  There the jobs are shell commands
handed to a `multiprocessing.Pool` of CPU workers (mp_mode='py'), run one after the other under `mpirun`
('mpi'), or written to a batch script ('sh').  Here 'py' and 'mpi' run the jobs one after the other on the GPU(s)
of the calling process group -- each job's photons are what is parallelised -- and 'sh' writes a script of
`python -m er3t_amd.rtm.mca.mca_exe ...` commands with the reference's four-argument command line.

`rearrange_jobs` (the reference's load-balancing order for CPU workers, mca_run.py:185-315) is kept for callers
that distribute whole jobs over workers.
"""

import datetime
import os
import sys

import numpy as np

import er3t_amd.common

__all__ = ['mca_run', 'rearrange_jobs']


class mca_run:

    """
    Input:
        fnames_inp, fnames_out: lists of input / output file paths, one pair per job
        executable=: command used in 'sh' scripts (default: this package's solver module)
        photons=   : number of photons, scalar or one value per job
        solver=    : 0 3-D, 1 partial 3-D, 2 IPA
        Ncpu=      : accepted for interface parity (job order under `optimize`)
        mp_mode=   : 'py' | 'mpi' (run now on the GPU) | 'sh' (write a batch script)
    """

    def __init__(self, fnames_inp, fnames_out, executable=None, photons=1.0e6, solver=0, Ncpu=1, mp_mode='py',
                 optimize=True, fname_sh=None, verbose=er3t_amd.common.params['verbose'], quiet=False):

        if executable is None:
            executable = '%s -m er3t_amd.rtm.mca.mca_exe' % sys.executable

        Nfile = len(fnames_inp)
        if len(fnames_out) != Nfile:
            raise OSError('\nError [mca_run]: Inconsistent input and output files.')

        self.Ncpu    = Ncpu
        self.quiet   = quiet
        self.verbose = verbose
        self.solver  = solver

        if not isinstance(photons, np.ndarray):
            photons_dist = np.repeat(photons, Nfile)
        elif photons.size == Nfile:
            photons_dist = photons.copy()
        else:
            raise ValueError('Error [mca_run]: Cannot distribute photon set of %d over %d runs.' % (photons.size, Nfile))

        mp_mode = mp_mode.lower()
        if mp_mode in ['mpi', 'openmpi']:
            mp_mode = 'mpi'
        elif mp_mode in ['python', 'multiprocessing', 'py', 'mp', 'pymp']:
            mp_mode = 'py'
        elif mp_mode in ['batch', 'shell', 'bash', 'hpc', 'sh']:
            mp_mode = 'sh'
        else:
            raise OSError('\nError [mca_run]: Cannot understand input <mp_mode=\'%s\'>.' % mp_mode)
        self.mp_mode = mp_mode

        # on one GPU the jobs run back to back, so their order only matters for scripts meant for CPU workers
        indices = rearrange_jobs(Ncpu, photons_dist) if (mp_mode == 'sh' and Ncpu > 1 and optimize) else np.arange(Nfile)

        self.jobs = []
        self.commands = []
        for i in indices:
            input_file  = os.path.abspath(fnames_inp[i])
            output_file = os.path.abspath(fnames_out[i])
            os.makedirs(os.path.dirname(output_file), exist_ok=True)
            self.jobs.append((input_file, output_file, int(photons_dist[i])))
            self.commands.append('%s %d %d %s %s' % (executable, photons_dist[i], solver, input_file, output_file))

        if self.mp_mode in ('mpi', 'py'):
            self.run()
        else:
            self.save(fname=fname_sh)

    def run(self):
        from er3t_amd.rtm.mca.mca_exe import get_runner, run_job, submit_job, collect_job
        runner = get_runner()
        ms0, n0 = runner.kernel_ms, runner.photons_done
        # One process: two solver handles take turns, job i+1 is launched before job i is read back and written, so that the
        # tail of a launch (as long as its longest history) and the writing of the output file run beside the next launch.
            # Under torchrun the jobs go through JobRunner.run_batched: one exchange per batch of jobs.
        if runner.world > 1 and len(self.jobs) > 1:
            # several ranks: the raw tallies of a batch of jobs are exchanged with ONE all-reduce (JobRunner.run_batched)
            if self.verbose:
                for command in self.commands:
                    print('Message [mca_run]: Executing <%s> ...' % command)
            runner.run_batched(self.jobs, self.solver)
            self.kernel_ms = runner.kernel_ms - ms0
            self.photons_done = runner.photons_done - n0
            return
        nslot = runner.use_slots(2) if (runner.world == 1 and len(self.jobs) > 1) else 1
        waiting = None
        for i, (command, (fname_inp, fname_out, nphoton)) in enumerate(zip(self.commands, self.jobs)):
            if self.verbose:
                print('Message [mca_run]: Executing <%s> ...' % command)
            if nslot == 1:
                run_job(fname_inp, fname_out, nphoton, self.solver, runner=runner)
                continue
            job = submit_job(fname_inp, nphoton, self.solver, runner, slot=i % nslot)
            if waiting is not None:
                collect_job(waiting[0], waiting[1], runner)
            waiting = (job, fname_out)
        if waiting is not None:
            collect_job(waiting[0], waiting[1], runner)
        self.kernel_ms = runner.kernel_ms - ms0
        self.photons_done = runner.photons_done - n0
        if not self.quiet and self.kernel_ms > 0.0:
            print('Message [mca_run]: %d jobs, %.3g photon histories per rank in %.1f ms of transport kernels (%.3g photons/s/GPU).'
                  % (len(self.jobs), self.photons_done, self.kernel_ms, self.photons_done/(self.kernel_ms*1.0e-3)))

    def save(self, fname=None):
        if fname is None:
            fname = 'mca-run_%s.sh' % datetime.datetime.now().strftime('%Y-%m-%d_%H:%M:%S')
        with open(fname, 'w') as f:
            f.write('#!/bin/bash\n\n')
            if not self.quiet:
                print('Message [mca_run]: Creating batch script <%s> ...' % fname)
            for command in self.commands:
                f.write(command + '\n')
        os.chmod(fname, 0o755)
        self.fname_sh = fname


def rearrange_jobs(Ncpu, weights_in):

    """
    Order in which to start jobs of unequal cost on Ncpu workers that each take the next job when free, so that the
    workers finish at about the same time (reference: er3t/rtm/mca/mca_run.py:185-315).

    Two stages: (1) heaviest job first, each job to the worker where it leaves the smallest spread between that
    worker and the least loaded one; (2) the workers' job lists are interleaved round by round, lighter
    accumulated load first.  Returns the job indices in start order.
    """

    f_dtype = er3t_amd.common.f_dtype
    weights = np.array(np.asarray(weights_in).ravel())
    weights = weights + weights.min()
    order = np.argsort(weights)[::-1]

    # stage 1: assignment
    lists = [[] for _ in range(Ncpu)]
    loads = np.zeros(Ncpu, dtype=f_dtype)
    for j in order:
        wj = weights[j]
        spread = ((loads+wj)-loads.min())**2
        iw = int(np.argmin(spread))
        loads[iw] += wj
        lists[iw].append((int(j), wj))

    # workers with more jobs first
    nlist = np.array([len(l) for l in lists])
    lists = [lists[i] for i in np.argsort(nlist)[::-1]]

    # stage 2: interleave
    out = []
    next_round = np.arange(Ncpu)
    nround_max = max(len(l) for l in lists)
    base = None
    while max(len(l) for l in lists) > 0:
        first = (max(len(l) for l in lists) == nround_max)
        w_round = np.array([], dtype=f_dtype)
        i_round = np.array([], dtype=np.int32)
        for i in next_round:
            if len(lists[i]) > 0:
                j, wj = lists[i].pop(0)
                out.append(j)
                w_round = np.append(w_round, wj)
                i_round = np.append(i_round, i)
        if first:
            base = w_round.copy()
        else:
            base = base[:w_round.size].copy()
            base += w_round
        next_round = i_round[np.argsort(base)]
    return np.array(out)
