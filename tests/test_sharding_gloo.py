"""CPU, world_size 2, gloo: the N > 1 path -- contiguous shards, no data-path collective, one
all-gather of first-step controls; gathered result == unsharded result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B_total, ragged, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, 'igt-mpc-int_amd'), os.path.join(root, 'oracle')):
        sys.path.insert(0, p)
    import np_oracle as O
    from igtmpc.scenarios import make_batch
    from igtmpc.sharding import allgather_controls, first_controls, shard_range
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        full = make_batch(B_total, dtype=np.float64)
        lo, hi = shard_range(B_total, rank, world)
        P = O.Params()
        # each rank solves ONLY its shard (the oracle stands in for the GPU here: no GPU in CPU tests)
        r = O.solve_batch(full['x0'][lo:hi], full['u_prev'][lo:hi], full['kparams'][lo:hi], full['flags'][lo:hi],
                          full['obs_xy'][lo:hi], None, None, P, C=64)
        u0 = first_controls(torch.from_numpy(np.nan_to_num(r['u'], nan=-9.0)))
        g = allgather_controls(u0, B_total=None if ragged else B_total)
        # the exchange's self-check (bench.py runs it untimed after the timed region): accepts the gathered vector, refuses
        # one with a flipped bit in the OTHER rank's block, one with two rows swapped, and a truncated one -- on every rank
        from igtmpc.sharding import verify_gathered
        good = verify_gathered(u0, g)
        olo, ohi = shard_range(B_total, 1 - rank, world)
        flipped = g.clone()
        flipped[olo, 0] = torch.nextafter(flipped[olo, 0], torch.tensor(1e9, dtype=flipped.dtype))
        swapped = g.clone()
        swapped[[olo, ohi - 1]] = swapped[[ohi - 1, olo]]
        verdicts = (good['ok'], verify_gathered(u0, flipped)['ok'], verify_gathered(u0, swapped)['ok'],
                    verify_gathered(u0, g[:-1])['ok'], good['ranks_seen'], good['per_rank_B_local'])
        q.put((rank, (g.numpy(), verdicts)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('B_total,ragged', [(16, False), (13, True)])
def test_two_rank_allgather_equals_unsharded(B_total, ragged):
    import np_oracle as O
    from igtmpc.scenarios import make_batch
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B_total, ragged, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = make_batch(B_total, dtype=np.float64)
    ref = O.solve_batch(full['x0'], full['u_prev'], full['kparams'], full['flags'], full['obs_xy'], None, None,
                        O.Params(), C=64)
    want = np.nan_to_num(ref['u'], nan=-9.0)[:, :, 0]
    assert np.array_equal(got[0][0], want) and np.array_equal(got[1][0], want)
    sizes = [B_total - B_total // 2, B_total // 2]
    for r in range(2):
        assert got[r][1] == (True, False, False, False, 2, sizes), got[r][1]
