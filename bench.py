#!/usr/bin/env python3
"""bench.py -- MPC solves/s of the batched shooting solver (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

One "step" = one pass of the hot path over one batch of synthetic two-vehicle intersection
scenarios per GPU (SURVEY.md section 8d generator), inputs already resident in HBM:
    search kernel (C=256 candidates x N=20 steps x 4 RK4 sub-steps, cost, verdicts, arg-min)
  + emit kernel (winner trajectory / controls)
  + for N > 1: RCCL all-gather of the first-step controls u*[:, :, 0].
N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ...`:
one process per GPU, each solving its own contiguous shard (weak scaling, no data-path
collective).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd'))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector
FLOP_EQ_PER_SOLVE = 9.8e6      # 2.9 MFLOP + 343k transcendentals at 20 flop each


def host_cores():
    """CPU threads this process may actually use: affinity mask capped by the cgroup quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            with open(path) as f:
                tok = f.read().split()
            if path.endswith('cpu.max'):
                if tok[0] != 'max':
                    n = min(n, max(1, int(int(tok[0]) / int(tok[1]))))
            else:
                q = int(tok[0])
                if q > 0:
                    with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
                        n = min(n, max(1, q // int(f.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(batch, N, C, budget_s=12.0):
    """The oracle's C restatement (oracle/igt_oracle.c, kind "port") on this host's cores,
    on a bounded sample of the SAME workload.  Checker code timed as a baseline only."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import c_oracle as CO
    import np_oracle as O
    from igtmpc.cinf import cinf_halfplanes
    P = O.Params(N=N)
    A, b = cinf_halfplanes(dt=P.dt, jerk=P.jerk)
    f = lambda k, sl: np.asarray(batch[k][sl], dtype=np.float64)
    cores = host_cores()
    chunk = 64 * cores
    done, t_all = 0, 0.0
    # one untimed chunk to page in + spin up the OpenMP team
    sl = slice(0, min(chunk, len(batch['x0'])))
    CO.solve_batch(f('x0', sl), f('u_prev', sl), f('kparams', sl), batch['flags'][sl], f('obs_xy', sl), A, b, P, C=C,
                   nthreads=cores)
    B = len(batch['x0'])
    lo = 0
    while t_all < budget_s and lo < B:
        sl = slice(lo, min(lo + chunk, B))
        t0 = time.perf_counter()
        CO.solve_batch(f('x0', sl), f('u_prev', sl), f('kparams', sl), batch['flags'][sl], f('obs_xy', sl), A, b, P,
                       C=C, nthreads=cores)
        t_all += time.perf_counter() - t0
        done += sl.stop - sl.start
        lo = sl.stop
    # single-thread figure on a small slice
    sl = slice(0, min(128, B))
    t0 = time.perf_counter()
    CO.solve_batch(f('x0', sl), f('u_prev', sl), f('kparams', sl), batch['flags'][sl], f('obs_xy', sl), A, b, P, C=C,
                   nthreads=1)
    t1 = time.perf_counter() - t0
    return {'value': done / t_all, 'unit': 'solves/s', 'cores': cores, 'kind': 'port',
            'sample': f'first {done} scenarios of the same batch, float64 C restatement (oracle/igt_oracle.c), '
                      f'OpenMP over scenarios; reference mpc.py (CasADi/IPOPT) cannot run on this image',
            'single_thread_value': (sl.stop - sl.start) / t1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=4096, help='scenarios per GPU per step (BASELINE config 2)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--gt', type=int, default=0, metavar='SC',
                    help='gt_mpc cost with the shipped value net of scenario SC (1 or 3; BASELINE configs[4]); 0 = mpc cost')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # the exchange path can be rehearsed on one GPU with a single-rank process group (IGT_BENCH_FORCE_DIST=1 and the
    # torchrun environment variables): same streams, events and collective calls as with N ranks
    exchange = world > 1 or os.environ.get('IGT_BENCH_FORCE_DIST') == '1'
    if exchange:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank), rank=rank, world_size=world)
    n_gpus = world if world > 1 else 1
    dev = local_rank if world > 1 else 0
    torch.cuda.set_device(dev)

    from igtmpc import BatchSolver
    from igtmpc.cinf import cinf_halfplanes
    from igtmpc.scenarios import make_batch
    from igtmpc.sharding import allgather_controls, first_controls

    B, N, C = args.batch, 20, 256
    batch = make_batch(B, N=N, dtype=np.float32, offset=rank * B)
    dargs = [torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda(dev)
             for a in (batch['x0'], batch['u_prev'], batch['kparams'], batch['flags'], batch['obs_xy'])]
    solver = BatchSolver(N=N, C=C, n_obs=1, device=dev, dtype='f32', cost_mode='value_net' if args.gt else 'progress')
    solver.set_cinf(*cinf_halfplanes(dt=solver.params.dt, jerk=solver.params.jerk_limit))
    if args.gt:
        g = np.load(os.path.join(ROOT, 'tests', 'golden', 'value_net_golden.npz'))
        layers, i = [], 0
        while f'sc{args.gt}_W{i}' in g:
            layers.append((g[f'sc{args.gt}_W{i}'], g[f'sc{args.gt}_b{i}']))
            i += 1
        solver.set_value_net(layers)      # identity whitening: the reference's statistics are not shipped
        dargs += [torch.from_numpy(batch['tv_sv']).cuda(dev), torch.from_numpy(batch['enc']).cuda(dev)]
    out = solver.solve(*dargs)

    # The one exchange of the path: all-gather of the first-step controls u*[:, :, 0] so that every rank
    # holds the full action vector.  A rank's next step does not depend on the other ranks' controls (scenarios are
    # independent), so the exchange of step t runs on its own stream under the search pass of step t+1,
    # double-buffered; the timed region ends with a device-wide synchronize, i.e. with every exchange complete.
    if exchange:
        main = torch.cuda.current_stream(dev)
        comm = torch.cuda.Stream(dev)
        u0_buf = [torch.empty((B, 2), dtype=torch.float32, device=f'cuda:{dev}') for _ in range(2)]
        gathered = [torch.empty((B * world, 2), dtype=torch.float32, device=f'cuda:{dev}') for _ in range(2)]
        ev_ready = [torch.cuda.Event() for _ in range(2)]
        ev_free = [torch.cuda.Event() for _ in range(2)]
        for e in ev_free:
            e.record(comm)
    step_no = [0]

    def step():
        solver.solve(*dargs, out=out)
        if exchange:
            i = step_no[0] & 1
            step_no[0] += 1
            main.wait_event(ev_free[i])                    # the exchange two steps back has released buffer i
            u0_buf[i].copy_(out['u'][:, :, 0])
            ev_ready[i].record(main)
            with torch.cuda.stream(comm):
                comm.wait_event(ev_ready[i])
                dist.all_gather_into_tensor(gathered[i], u0_buf[i])
                ev_free[i].record(comm)
            return gathered[i]
        return None

    def fence():
        torch.cuda.synchronize(dev)
        if exchange:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if exchange:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f'cuda:{dev}')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel durations: HIP events recorded by the library on the launch stream, same workload
    solver.set_profiling(True)
    ks, ke = [], []
    for _ in range(min(args.steps, 50)):
        solver.solve(*dargs, out=out)
        a, e = solver.kernel_ms()
        ks.append(a)
        ke.append(e)
    solver.set_profiling(False)
    search_ms, emit_ms = float(np.mean(ks)), float(np.mean(ke))
    rd, wr = solver.algorithmic_bytes_per_solve()
    feasible = float((out['status'] == 0).float().mean().item())

    if rank == 0:
        value = B * n_gpus * args.steps / elapsed
        bytes_per_launch = (rd + 12) * B          # search kernel: reads inputs, writes cost/argmin/status
        ach = bytes_per_launch / (search_ms * 1e-3) / 1e9
        traffic, valu_busy = None, None      # PMC figures of the committed rocprofv3 run of this same command
        tpath = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
        if os.path.exists(tpath) and not args.gt:
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get('batch') == B:
                    traffic = tj.get('search_kernel_hbm_bytes_per_launch')
                    valu_busy = tj.get('derived', {}).get('simd_valu_busy_fraction')
            except Exception:
                traffic, valu_busy = None, None
        line = {
            'metric': 'mpc_solves_per_sec', 'value': value, 'unit': 'solves/s', 'n_gpus': n_gpus,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'batch={B} two-vehicle intersection scenarios per GPU, horizon N=20, '
                                   f'{C} lattice candidates, 4 RK4 sub-steps, Frenet bicycle model, C_inf terminal set '
                                   f'(BASELINE configs[1])',
                       'arithmetic': 'float32 stage derivatives + float64 state accumulators, cost and verdicts',
                       'parallelism': f'scenario shards x{n_gpus}, all-gather of u*[:, :, 0] on its own stream under the next step' if exchange else 'single GPU',
                       'cost': f'gt_mpc value net V_GT_sc{args.gt} ({len(layers) - 1} hidden layers, identity normalisation)' if args.gt else 'mpc progress cost',
                       'feasible_fraction': feasible},
            'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': ach / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel': 'search_fast_kernel (+ value kernels in gt mode)', 'kernel_ms': search_ms,
                         'algorithmic_bytes_per_solve': rd + 12,
                         'note': 'the path is FP32-VALU-bound (arithmetic intensity ~1e4 flop/B); see valu_roofline'},
            # the pipe that actually bounds the path: share of SIMD cycles issuing VALU work (PMC, committed profile of
            # this same command) -- null when no profile of this batch size is committed.  The algorithmic flop count of
            # SURVEY 8d is reported beside it for reference only: early exit, the closed-form sub-steps and the rotation
            # polynomials skip most of it, so its rate can exceed the FP32 vector peak.
            'valu_roofline': {'bound': 'valu_issue', 'achieved': valu_busy, 'peak': 1.0,
                              'unit': 'fraction of SIMD cycles issuing VALU instructions (SQ_ACTIVE_INST_VALU)',
                              'frac': valu_busy,
                              'algorithmic_tflop_eq_per_s': B / (search_ms * 1e-3) * FLOP_EQ_PER_SOLVE / 1e12,
                              'fp32_vector_peak_tflops': FP32_VALU_PEAK_TFLOPS, 'flop_eq_per_solve': FLOP_EQ_PER_SOLVE},
            'kernels_ms': {'search': search_ms, 'emit': emit_ms},
            'whole_solve_bytes': rd + wr,
        }
        if n_gpus == 1 and not args.no_cpu_baseline and not args.gt:
            line['cpu_baseline'] = cpu_baseline(batch, N, C)
        print(json.dumps(line), flush=True)
    solver.close()
    if exchange:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
