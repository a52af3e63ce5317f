"""
The N > 1 path on CPU: two processes, torch.distributed 'gloo' backend, photon ids sharded with
er3t_amd.dist.photon_shard, raw tallies summed with er3t_amd.dist.allreduce_tallies.  The oracle stands in for
the per-rank transport (tests may do that; the product never does), so the check is exactly the property the
multi-GPU design rests on: sharded-and-reduced == single-process, for the same global photon ids.
"""

import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_photon_shard_partitions_exactly():
    from er3t_amd.dist import photon_shard
    for n in (0, 1, 7, 100, 10**9+7):
        for w in (1, 2, 3, 8):
            parts = [photon_shard(n, w, r) for r in range(w)]
            assert parts[0][0] == 0
            assert sum(c for _, c in parts) == n
            for (o0, c0), (o1, c1) in zip(parts[:-1], parts[1:]):
                assert o0+c0 == o1 and 0 <= c0-c1 <= 1
    with pytest.raises(ValueError):
        photon_shard(10, 2, 2)


def _worker(rank, world, port, nphoton, seed, outdir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch
    import torch.distributed as dist
    from er3t_amd.dist import photon_shard, allreduce_tallies, world_info
    from oracle import oracle
    from tests.util import slab_scene
    dist.init_process_group('gloo', rank=rank, world_size=world)
    assert world_info() == (rank, world)
    sc = slab_scene(tau=3.0, apf=0.8, albedo=0.2, nx=4, ny=3, nz3=2, vza=(0.0, 40.0), vaa=(0.0, 120.0))
    off, cnt = photon_shard(nphoton, world, rank)
    rad, flux, counters = oracle.run_raw(sc, cnt, seed=seed, offset=off, nthreads=1)
    t_rad = torch.from_numpy(rad.copy()); t_flux = torch.from_numpy(flux.copy()); t_cnt = torch.from_numpy(counters.astype(np.int64))
    allreduce_tallies(t_rad, t_flux, t_cnt, None)
    if rank == 0:
        np.savez(os.path.join(outdir, 'reduced.npz'), rad=t_rad.numpy(), flux=t_flux.numpy(), cnt=t_cnt.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_run_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    from oracle import oracle
    from tests.util import slab_scene
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    nphoton, seed, world = 4001, 17, 2
    mp.spawn(_worker, args=(world, port, nphoton, seed, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), 'reduced.npz'))
    sc = slab_scene(tau=3.0, apf=0.8, albedo=0.2, nx=4, ny=3, nz3=2, vza=(0.0, 40.0), vaa=(0.0, 120.0))
    rad, flux, cnt = oracle.run_raw(sc, nphoton, seed=seed, offset=0, nthreads=1)
    assert np.array_equal(got['cnt'], cnt.astype(np.int64))
    assert np.allclose(got['rad'], rad, rtol=1e-12, atol=0.0)
    assert np.allclose(got['flux'], flux, rtol=1e-12, atol=0.0)


def test_allreduce_is_a_noop_without_a_process_group():
    import torch
    from er3t_amd.dist import allreduce_tallies, world_info
    t = torch.ones(4)
    allreduce_tallies(t, None)
    assert world_info() == (0, 1) and torch.equal(t, torch.ones(4))


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher: the parent starts two fresh ranks under torch.distributed.run (gloo,
    --dry-run: no transport, this box has no GPU), they meet, exchange once per step and rank 0's line comes back"""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '0', '--photons', '1001',
                        '--scaling', 'strong', '--backend', 'gloo', '--dry-run'], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['rccl_ranks'] == 2 and line['scaling'] == 'strong' and line['data'] == 'dry-run'
    assert line['value'] is None                                     # not a measurement
    assert line['config']['photon_ids_covered'] == 3*1001            # the two shares of every step add up


def test_bench_refuses_more_gpus_than_the_box_has():
    """a --gpus N the machine cannot serve must fail loudly instead of reporting an N-GPU number from fewer devices"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '64', '--steps', '1'], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')})
    assert r.returncode != 0 and 'refusing' in r.stderr
