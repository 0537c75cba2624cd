"""-m gpu : the regime bench.py's headline is measured in -- F handles on F streams, igt_set_concurrency(F), solves
genuinely overlapped on the device -- gives the answers of the same batches solved alone.

With F >= 3 the persistent search kernels take one wave per SIMD and two of them are co-resident (include/igtmpc.h
igt_set_concurrency); every handle owns its workspace (partials, queue counters, value-net list) and every lane its
output buffers.  A race between co-resident kernels, or between a lane's search and its own emit / value kernels, would
show here as a differing bit: every lane's x, u, cost, argmin, status of every round is compared with
  (1) the same batch solved alone on a fresh handle with the whole device to itself -- bit for bit, and
  (2) the float64 oracle on a fixed 256-scenario subsample (f64: 1e-9, f32: 1e-5 with the float32 set-asides).
The lanes cover f64 and f32, lattice and tracking candidates and one gt_mpc lane (terminal value network), each lane with
its own batches (two per lane, alternating between rounds), B = 4096 per solve as in BASELINE configs[1].
"""
import numpy as np
import pytest

from helpers import F32_EPS, REL_TOL, oracle_params
from test_gpu_fullsize import _check_subsample, _cinf, _net

pytestmark = pytest.mark.gpu

B = 4096
ROUNDS = 4                       # >= 3 rounds of solves in flight per lane, none of them waited for
KEYS = ('x', 'u', 'cost', 'argmin', 'status')
# (dtype, candidate family, value net of scenario sc or 0); lane q takes LANES[q % 5]
LANES = [('f64', 'lattice', 0), ('f64', 'track', 0), ('f32', 'lattice', 0), ('f64', 'lattice', 1), ('f32', 'track', 0)]


def _solver(igt, golden_dir, dtype, cand, gt, concurrency):
    s = igt.BatchSolver(dtype=dtype, cand_mode=cand, cost_mode='value_net' if gt else 'progress')
    s.set_cinf(*_cinf())
    if gt:
        s.set_value_net(layers=_net(golden_dir, gt), Wn=np.eye(6), mu_f=np.zeros(6), sigma_t=1.0, mu_t=0.0)
    s.set_concurrency(concurrency)
    return s


def _device_args(torch, b, gt):
    keys = ['x0', 'u_prev', 'kparams', 'flags', 'obs_xy'] + (['tv_sv', 'enc'] if gt else [])
    return [torch.from_numpy(b[k].view(np.int32) if b[k].dtype == np.uint32 else b[k]).cuda() for k in keys]


def _empty_out(torch, dtype):
    td = torch.float64 if dtype == 'f64' else torch.float32
    return dict(x=torch.empty((B, 7, 21), dtype=td, device='cuda'), u=torch.empty((B, 2, 20), dtype=td, device='cuda'),
                cost=torch.empty((B,), dtype=td, device='cuda'), argmin=torch.empty((B,), dtype=torch.int32, device='cuda'),
                status=torch.empty((B,), dtype=torch.int32, device='cuda'))


@pytest.mark.parametrize('F', [3, 4, 8])
def test_overlapped_solves_equal_the_same_batches_solved_alone(golden_dir, F):
    import torch
    import igtmpc as igt
    from igtmpc.scenarios import make_batch
    igt.load_library()
    cfg = [LANES[q % len(LANES)] for q in range(F)]
    # two batches per lane, no two lanes share one (make_batch(offset=...) is the generator's per-shard path)
    host = [[make_batch(B, dtype=np.float64 if d == 'f64' else np.float32, offset=(2 * q + v + 1) * B) for v in range(2)]
            for q, (d, _, _) in enumerate(cfg)]
    dargs = [[_device_args(torch, host[q][v], cfg[q][2]) for v in range(2)] for q in range(F)]
    solvers = [_solver(igt, golden_dir, *cfg[q], concurrency=F) for q in range(F)]
    streams = [torch.cuda.Stream() for _ in range(F)]
    outs = [[_empty_out(torch, cfg[q][0]) for _ in range(ROUNDS)] for q in range(F)]
    # one eager solve per lane first: the workspaces grow on first use (a synchronising allocation), not while overlapped
    for q in range(F):
        solvers[q].solve(*dargs[q][0], out=outs[q][0])
    torch.cuda.synchronize()
    for q in range(F):
        for k in KEYS:
            outs[q][0][k].fill_(-7)
    base = torch.cuda.Event(enable_timing=True)
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(ROUNDS)] for _ in range(F)]
    base.record()
    for st in streams:
        st.wait_event(base)
    for r in range(ROUNDS):                  # nothing waits for anything: F x ROUNDS solves enqueued back to back
        for q in range(F):
            with torch.cuda.stream(streams[q]):
                ev[q][r][0].record()
                solvers[q].solve(*dargs[q][r % 2], out=outs[q][r])
                ev[q][r][1].record()
    torch.cuda.synchronize()
    # the solves did overlap on the device: spans (relative to `base`) of different lanes intersect
    span = [[(base.elapsed_time(a), base.elapsed_time(e)) for a, e in ev[q]] for q in range(F)]
    pairs = sum(1 for q in range(F) for p in range(q + 1, F) for a in span[q] for b_ in span[p] if a[0] < b_[1] and b_[0] < a[1])
    busy = sum(e - a for q in range(F) for a, e in span[q])
    wall = max(e for q in range(F) for _, e in span[q]) - min(a for q in range(F) for a, _ in span[q])
    print(f'F={F}: {pairs} overlapping pairs of solves from different lanes; summed spans {busy:.2f} ms in {wall:.2f} ms of wall time')
    assert pairs >= F * ROUNDS // 2, 'the lanes did not overlap: this test would not see a race'
    assert busy > 1.5 * wall
    got = [[{k: outs[q][r][k].cpu().numpy() for k in KEYS} for r in range(ROUNDS)] for q in range(F)]
    for s in solvers:
        s.close()
    for q, (dtype, cand, gt) in enumerate(cfg):
        with _solver(igt, golden_dir, dtype, cand, gt, concurrency=1) as solo:       # alone on the device, two waves per SIMD
            P = oracle_params(solo)
            alone = []
            for v in range(2):
                o = solo.solve(*dargs[q][v])
                torch.cuda.synchronize()
                alone.append({k: o[k].cpu().numpy() for k in KEYS})
        for r in range(ROUNDS):
            assert 0.5 < (got[q][r]['status'] == 0).mean() < 1.0
            for k in KEYS:
                assert np.array_equal(got[q][r][k], alone[r % 2][k], equal_nan=True), (F, q, cfg[q], r, k)
        # ... and the overlapped answers are the oracle's (round ROUNDS-1 of batch (ROUNDS-1) % 2)
        v = (ROUNDS - 1) % 2
        idx = np.sort(np.random.default_rng(5 + q).choice(B, 256, replace=False))
        tol, eps = (1e-9, 1e-9) if dtype == 'f64' else (REL_TOL, F32_EPS if cand == 'lattice' else 2e-5)
        net = dict(layers=_net(golden_dir, gt), Wn=np.eye(6), mu_f=np.zeros(6), sigma_t=1.0, mu_t=0.0) if gt else None
        _check_subsample(host[q][v], got[q][ROUNDS - 1], idx, P, tol, eps, net, cand)
