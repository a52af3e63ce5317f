"""
Multi-GPU layer: photons shard, tallies all-reduce.

The reference parallelises the same way at process level: independent (run, g) jobs in a
`multiprocessing.Pool` (er3t/rtm/mca/mca_run.py:144-152) and, optionally, MCARaTS' own MPI photon split
under `mpirun -n Ncpu` (mca_run.py:110-111).  Here one process drives one GPU (torch.distributed, backend
'nccl' = RCCL over xGMI; 'gloo' in the CPU tests); every rank holds the full scene, transports a contiguous
range of global photon ids -- the Philox stream is keyed by photon id, so the sampling does not depend on
the number of ranks -- and the raw tally buffers are summed with ONE all-reduce per job.
"""

import numpy as np

__all__ = ['photon_shard', 'allreduce_tallies', 'world_info', 'barrier']


def photon_shard(nphoton, world_size, rank):

    """
    Contiguous split of photon ids [0, nphoton) over ranks: returns (offset, count).
    The first `nphoton % world_size` ranks take one extra photon, so counts differ by at most one.
    """

    nphoton = int(nphoton); world_size = int(world_size); rank = int(rank)
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError('Error [photon_shard]: bad rank %d of %d.' % (rank, world_size))
    base, rest = divmod(nphoton, world_size)
    count  = base + (1 if rank < rest else 0)
    offset = rank*base + min(rank, rest)
    return offset, count


def world_info():
    """(rank, world_size) of the default process group, (0, 1) when torch.distributed is not initialised"""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


def allreduce_tallies(*tensors):

    """
    Sum raw tally tensors over all ranks in place (no-op for a single process).  The tensors are the
    buffers the solver accumulates into (`Mi3dSolver.bind`), so no copy is made; message sizes are
    nview*nx*ny*4 B for radiance and 3*(nz+1)*nx*ny*4 B for flux.
    """

    rank, world = world_info()
    if world == 1:
        return
    import torch.distributed as dist
    for t in tensors:
        if t is not None:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)


def barrier():
    """wait for every rank (no-op for a single process): rank 0 writes the job files, everybody reads them"""
    rank, world = world_info()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
