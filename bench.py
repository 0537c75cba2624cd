#!/usr/bin/env python3
"""bench.py -- MPC solves/s of the batched shooting solver (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--dtype f64|f32] [--gt SC] [--in-flight F]

One "step" = one pass of the hot path over one batch of synthetic two-vehicle intersection
scenarios per GPU (SURVEY.md section 8d generator), inputs already resident in HBM:
    search kernel (C=256 candidates x N=20 steps x 4 RK4 sub-steps, cost, verdicts, arg-min)
  + emit kernel (winner trajectory / controls)
  + for N > 1: RCCL all-gather of the first-step controls u*[:, :, 0].

The headline (`value`, `dtype`) is the float64 entry igt_solve_batch_f64 -- the reference's own precision
(kinematic_bicycle_model_frenet.py:70-127 and mpc.py are float64 end to end); the float32 entry
(float stage derivatives, double state accumulators) is timed in the same run and reported beside it as
`f32_path`.

Steps are pipelined (`--in-flight F`, default 4): step t is enqueued on handle / stream t mod F -- every handle owns
its workspace and its output buffers, the inputs are read-only -- so the emit pass (one roll-out of latency) and the
drain tail of one step's search overlap the search pass of the next instead of being exposed.  All K steps complete
inside the timed region; the one-solve-at-a-time figure (`--in-flight 1`) is measured in the same run and printed as
`one_solve_in_flight`.

N > 1: one process per GPU, each solving its own contiguous shard (weak scaling, no data-path collective).
The driver launches that as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`; a plain
`python bench.py --gpus N` (no torchrun variables) spawns exactly that command itself -- before this process
touches a GPU -- and fails loudly if it cannot.  Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd'))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SIMDS = 256 * 4                # 256 CUs x 4 SIMDs
VALU_ISSUE_CYCLES = 4          # a wave64 VALU instruction occupies its SIMD's 16 lanes for 4 cycles
PEAK_CLOCK_GHZ = 2.4           # MI355X_MICROARCH.md: peak engine clock (the issue capacity the fraction is taken of)
N_HORIZON, N_CAND = 20, 256


def default_batch(n_gpus):
    """Scenarios per GPU: BASELINE configs[1] (4096) up to 4 GPUs; configs[3] (262 144 sharded 8 ways = 32 768) at 8."""
    return 32768 if n_gpus >= 8 else 4096


def workload_name(B, n_gpus, gt):
    if gt:
        tag = 'BASELINE configs[4] (gt_mpc, batch=65536 per GPU)' if B == 65536 else 'custom gt_mpc batch'
    elif B == 4096:
        tag = 'BASELINE configs[1]' if n_gpus == 1 else f'BASELINE configs[1] per GPU x{n_gpus}'
    elif B == 65536 and n_gpus == 1:
        tag = 'BASELINE configs[2]'
    elif B == 32768 and n_gpus == 8:
        tag = 'BASELINE configs[3] (262144 scenarios sharded 8 ways)'
    else:
        tag = 'custom batch'
    return (f'{tag}: batch={B} two-vehicle intersection scenarios per GPU (all 8 sc variants tiled), horizon N={N_HORIZON}, '
            f'{N_CAND} lattice candidates, 4 RK4 sub-steps, Frenet bicycle model, C_inf terminal set')


def source_hash():
    """Hash of the kernel sources the loaded library was built from (profiles/ entries carry the hash they were
    collected at; a PMC figure is only quoted when it matches)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'igt-mpc-int_amd', 'csrc')
    for fn in sorted(os.listdir(d)):
        if fn.endswith(('.hip', '.h', '.inc')):
            with open(os.path.join(d, fn), 'rb') as f:
                h.update(fn.encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


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
    import numpy as np
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


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without torchrun variables: start the N ranks ourselves (the same command the driver
    uses), before this process has made any GPU call, and hand back the children's exit code.  Rank 0's JSON line goes
    straight to our stdout."""
    import torch
    if not args.rehearse_cpu:
        have = torch.cuda.device_count()         # counts devices without initialising the runtime
        if have < args.gpus:
            print(f'bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible; refusing to report a '
                  f'{args.gpus}-GPU line from fewer devices', file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
               IGT_BENCH_SPAWNED='1')
    try:
        return subprocess.run(cmd, env=env).returncode
    except OSError as e:
        print(f'bench.py: cannot spawn {args.gpus} ranks: {e}', file=sys.stderr)
        return 2


def rehearse_cpu(world, rank):
    """CPU rehearsal of the launch path (tests only: `--rehearse-cpu`): the ranks meet over gloo, all-gather a token and
    rank 0 reports how many ranks it saw.  No solve, no GPU."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    tok = torch.tensor([rank], dtype=torch.int64)
    got = [torch.zeros_like(tok) for _ in range(world)]
    dist.all_gather(got, tok)
    if rank == 0:
        print(json.dumps({'rehearsal': True, 'ranks_seen': dist.get_world_size(), 'backend': dist.get_backend(),
                          'tokens': [int(t.item()) for t in got],
                          'spawned_by_bench': os.environ.get('IGT_BENCH_SPAWNED') == '1'}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=None,
                    help='scenarios per GPU per step (default: 4096 = BASELINE configs[1]; 32768 = configs[3] at --gpus 8)')
    ap.add_argument('--dtype', default='f64', choices=['f64', 'f32'],
                    help="entry point behind the headline `value` (default f64: the reference's precision)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the other-precision block')
    ap.add_argument('--gt', type=int, default=0, metavar='SC',
                    help='gt_mpc cost with the shipped value net of scenario SC (1 or 3; BASELINE configs[4]); 0 = mpc cost')
    ap.add_argument('--in-flight', type=int, default=4, choices=[1, 2, 3, 4, 5, 6, 8],
                    help='solves in flight: step t runs on handle/stream t mod F (each handle owns its workspace and '
                         'output buffers), so the emit pass and the drain tail of one step overlap the search pass of the next')
    ap.add_argument('--rehearse-cpu', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error('--gpus must be >= 1')

    if 'WORLD_SIZE' not in os.environ:
        if args.gpus > 1:
            sys.exit(spawn_ranks(args, sys.argv[1:]))
        world, rank, local_rank = 1, 0, 0
    else:
        world = int(os.environ['WORLD_SIZE'])
        rank = int(os.environ.get('RANK', '0'))
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        if world != args.gpus and os.environ.get('IGT_BENCH_FORCE_DIST') != '1':
            print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}', file=sys.stderr)
            sys.exit(2)
    if args.rehearse_cpu:
        sys.exit(rehearse_cpu(world, rank))

    import numpy as np
    import torch
    import torch.distributed as dist

    # the exchange path can be rehearsed on one GPU with a single-rank process group (IGT_BENCH_FORCE_DIST=1 and the
    # torchrun environment variables): same streams, events and collective calls as with N ranks
    exchange = world > 1 or os.environ.get('IGT_BENCH_FORCE_DIST') == '1'
    backend = None
    if exchange:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank), rank=rank, world_size=world)
        backend = dist.get_backend()
        assert dist.get_world_size() == world
    n_gpus = world if world > 1 else 1
    dev = local_rank if world > 1 else 0
    torch.cuda.set_device(dev)

    from igtmpc import BatchSolver
    from igtmpc.cinf import cinf_halfplanes
    from igtmpc.scenarios import make_batch

    N, C = N_HORIZON, N_CAND
    B = args.batch if args.batch else default_batch(n_gpus)
    layers = []
    if args.gt:
        g = np.load(os.path.join(ROOT, 'tests', 'golden', 'value_net_golden.npz'))
        i = 0
        while f'sc{args.gt}_W{i}' in g:
            layers.append((g[f'sc{args.gt}_W{i}'], g[f'sc{args.gt}_b{i}']))
            i += 1

    def measure(dtype, Bm, steps, warmup, in_flight=None, cand_mode='lattice'):
        """W untimed + K timed steps of the `dtype` entry point at Bm scenarios per GPU, then per-kernel HIP events."""
        npdt = np.float64 if dtype == 'f64' else np.float32
        td = torch.float64 if dtype == 'f64' else torch.float32
        batch = make_batch(Bm, N=N, dtype=npdt, offset=rank * Bm)
        keys = ['x0', 'u_prev', 'kparams', 'flags', 'obs_xy'] + (['tv_sv', 'enc'] if args.gt else [])
        dargs = [torch.from_numpy(batch[k].view(np.int32) if batch[k].dtype == np.uint32 else batch[k]).cuda(dev) for k in keys]
        F = in_flight or args.in_flight
        solvers, outs, lanes = [], [], []
        for _ in range(F):
            sv = BatchSolver(N=N, C=C, n_obs=1, device=dev, dtype=dtype, cost_mode='value_net' if args.gt else 'progress',
                             cand_mode=cand_mode)
            sv.set_cinf(*cinf_halfplanes(dt=sv.params.dt, jerk=sv.params.jerk_limit))
            if args.gt:
                sv.set_value_net(layers)      # identity whitening: the reference's statistics are not shipped
            solvers.append(sv)
            outs.append(sv.solve(*dargs))
            lanes.append(torch.cuda.Stream(dev) if F > 1 else None)
        torch.cuda.synchronize(dev)
        solver, out = solvers[0], outs[0]

        # The one exchange of the path: all-gather of the first-step controls u*[:, :, 0] so that every rank
        # holds the full action vector.  A rank's next step does not depend on the other ranks' controls (scenarios are
        # independent), so the exchange of step t runs on its own stream under the search pass of step t+1,
        # double-buffered; the timed region ends with a device-wide synchronize, i.e. with every exchange complete.
        if exchange:
            comm = torch.cuda.Stream(dev)
            u0_buf = [torch.empty((Bm, 2), dtype=td, device=f'cuda:{dev}') for _ in range(2)]
            gathered = [torch.empty((Bm * world, 2), dtype=td, device=f'cuda:{dev}') for _ in range(2)]
            ev_ready = [torch.cuda.Event() for _ in range(2)]
            ev_free = [torch.cuda.Event() for _ in range(2)]
            for e in ev_free:
                e.record(comm)
        step_no = [0]

        def step():
            # step t on handle / stream t mod F; the inputs are read-only, every lane has its own outputs.  The staging
            # copy of u*[:, :, 0] runs on the lane's own stream right behind the solve (so the lane's next solve cannot
            # overwrite what is still being copied); the all-gather runs on the communication stream.
            q = step_no[0] % F
            lane = lanes[q] if F > 1 else torch.cuda.current_stream(dev)
            i = step_no[0] & 1
            step_no[0] += 1
            with torch.cuda.stream(lane):
                solvers[q].solve(*dargs, out=outs[q])
                if exchange:
                    lane.wait_event(ev_free[i])                # the exchange two steps back has released buffer i
                    u0_buf[i].copy_(outs[q]['u'][:, :, 0])
                    ev_ready[i].record(lane)
            if exchange:
                with torch.cuda.stream(comm):
                    comm.wait_event(ev_ready[i])
                    dist.all_gather_into_tensor(gathered[i], u0_buf[i])
                    ev_free[i].record(comm)

        def fence():
            torch.cuda.synchronize(dev)
            if exchange:
                dist.barrier()
            torch.cuda.synchronize(dev)

        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
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
        for _ in range(min(steps, 50)):
            solver.solve(*dargs, out=out)
            a, e = solver.kernel_ms()
            ks.append(a)
            ke.append(e)
        solver.set_profiling(False)
        rd, wr = solver.algorithmic_bytes_per_solve()
        res = dict(dtype=dtype, B=Bm, elapsed=elapsed, steps=steps, value=Bm * n_gpus * steps / elapsed,
                   ms_per_step=elapsed / steps * 1e3, search_ms=float(np.mean(ks)), emit_ms=float(np.mean(ke)),
                   rd=rd, wr=wr, feasible=float((out['status'] == 0).float().mean().item()), batch=batch)
        for sv in solvers:
            sv.close()
        return res

    head = measure(args.dtype, B, args.steps, args.warmup)
    other_dtype = 'f32' if args.dtype == 'f64' else 'f64'
    other = None if args.no_secondary else measure(other_dtype, B, args.steps, args.warmup)
    # the driver computes scaling efficiency from the per-N values; at N = 8 the per-GPU batch is configs[3]'s 32 768,
    # not configs[1]'s 4096, so the same-per-GPU-work figure is measured beside it in the same run
    same_work = None
    if n_gpus > 1 and B != default_batch(1) and not args.batch:
        same_work = measure(args.dtype, default_batch(1), args.steps, args.warmup)
    # one solve at a time (every step waits for the previous one's emit pass), for comparison with the pipelined headline
    serial = measure(args.dtype, B, args.steps, args.warmup, in_flight=1) if args.in_flight > 1 and not args.no_secondary else None
    # the candidate family the planner and the closed-loop driver use by default (state-feedback steering):
    # the headline stays on SURVEY 8d's lattice, this is the same batch through the family that gives the better answers
    tracking = None if args.no_secondary else measure(args.dtype, B, args.steps, args.warmup, cand_mode='track')

    if rank == 0:
        search_ms, emit_ms, rd = head['search_ms'], head['emit_ms'], head['rd']
        bytes_per_launch = (rd + 12) * B          # search kernel: reads inputs, writes cost/argmin/status partials
        ach = bytes_per_launch / (search_ms * 1e-3) / 1e9
        # PMC figures (HBM bytes, VALU instruction count per launch) are properties of this workload + this binary; they
        # come from the committed rocprofv3 --pmc passes of this same command and are quoted only when the profile was
        # collected at the kernel sources this library was built from -- otherwise null.
        traffic, valu_insts, pmc_src = None, None, None
        tpath = os.path.join(ROOT, 'profiles', f'r02_pmc_{args.dtype}{"_gt%d" % args.gt if args.gt else ""}_b{B}.json')
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get('source_hash') == source_hash() and tj.get('batch') == B:
                    traffic = tj.get('search_kernel_hbm_bytes_per_launch')
                    valu_insts = tj.get('derived', {}).get('valu_instructions_per_launch')
                    pmc_src = f'{os.path.relpath(tpath, ROOT)} (rocprofv3 --pmc passes of this command at source hash ' \
                              f'{tj.get("source_hash")}; not measured by this run)'
            except Exception:
                traffic, valu_insts, pmc_src = None, None, None
        clk_ghz = PEAK_CLOCK_GHZ
        valu = {'bound': 'valu_issue', 'unit': 'fraction of SIMD issue cycles',
                'definition': 'VALU wave-instructions per launch x 4 issue cycles / (1024 SIMDs x kernel cycles); kernel '
                              'cycles = live kernel_ms x peak shader clock',
                'valu_instructions_per_launch': valu_insts, 'kernel_ms': search_ms, 'shader_clock_ghz': clk_ghz,
                'achieved': None, 'peak': 1.0, 'frac': None, 'source': pmc_src}
        if valu_insts:
            valu['achieved'] = valu_insts * VALU_ISSUE_CYCLES / (SIMDS * search_ms * 1e-3 * clk_ghz * 1e9)
            valu['frac'] = valu['achieved']
            # the same instruction count against the whole pipelined step (what `value` is made of): the search kernel's
            # drain tail and the emit pass are overlapped by the next steps' search passes
            valu['frac_of_whole_step'] = valu_insts * VALU_ISSUE_CYCLES / (SIMDS * head['ms_per_step'] * 1e-3 * clk_ghz * 1e9)
        arith = {'f64': 'float64 throughout (stage derivatives, state, cost, verdicts)',
                 'f32': 'float32 stage derivatives + float64 state accumulators, cost and verdicts'}
        line = {
            'metric': 'mpc_solves_per_sec', 'value': head['value'], 'unit': 'solves/s', 'n_gpus': n_gpus,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': head['ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': workload_name(B, n_gpus, args.gt),
                       'per_gpu_batch': B, 'global_batch': B * n_gpus,
                       'entry_point': f'igt_solve_batch_{args.dtype}',
                       'arithmetic': arith[args.dtype],
                       'parallelism': (f'scenario shards x{n_gpus}, one process per GPU, all-gather of u*[:, :, 0] on its own '
                                       f'stream under the next step') if exchange else 'single GPU',
                       'ranks_seen': world if exchange else 1, 'backend': backend,
                       'solves_in_flight': args.in_flight,
                       'pipelining': (f'step t runs on handle / stream t mod {args.in_flight} (own workspace and output buffers each); '
                                      f'all {args.steps} steps complete inside the timed region') if args.in_flight > 1 else 'none',
                       'cost': f'gt_mpc value net V_GT_sc{args.gt} ({len(layers) - 1} hidden layers, identity normalisation)' if args.gt else 'mpc progress cost',
                       'feasible_fraction': head['feasible']},
            'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': ach / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': pmc_src,
                         'kernel': f'search kernel of igt_solve_batch_{args.dtype} (+ value kernels in gt mode)',
                         'kernel_ms': search_ms, 'algorithmic_bytes_per_solve': rd + 12,
                         'note': 'the path is VALU-issue-bound (arithmetic intensity ~1e4 flop/B); see valu_roofline'},
            'valu_roofline': valu,
            'kernels_ms': {'search': search_ms, 'emit': emit_ms},
            'whole_solve_bytes': rd + head['wr'],
            'precision': {'f64': 'igt_solve_batch_f64: <= 1e-9 of the float64 oracle on every trajectory (curvature '
                                 'break-point straddlers included)',
                          'f32': 'igt_solve_batch_f32: <= 1e-5*max(1,|ref|) except trajectories whose curvature switch is '
                                 'decided inside float32 noise (share measured in tests/test_gpu_parity.py)'},
        }
        if tracking is not None:
            line['tracking_family'] = {
                'value': tracking['value'], 'unit': 'solves/s', 'ms_per_step': tracking['ms_per_step'], 'dtype': args.dtype,
                'kernels_ms': {'search': tracking['search_ms'], 'emit': tracking['emit_ms']},
                'feasible_fraction': tracking['feasible'],
                'note': 'same batch, entry point and timing with cand_mode = IGT_CAND_TRACK (256 candidates, one pass): '
                        'the default family of MPC_Planner and igtmpc.evaluate'}
        if other is not None:
            line[f'{other_dtype}_path'] = {
                'value': other['value'], 'unit': 'solves/s', 'ms_per_step': other['ms_per_step'], 'dtype': other_dtype,
                'entry_point': f'igt_solve_batch_{other_dtype}', 'arithmetic': arith[other_dtype],
                'kernels_ms': {'search': other['search_ms'], 'emit': other['emit_ms']},
                'steps': args.steps, 'warmup': args.warmup, 'feasible_fraction': other['feasible'],
                'note': 'same workload, same run, timed exactly like the headline'}
        if serial is not None:
            line['one_solve_in_flight'] = {
                'value': serial['value'], 'ms_per_step': serial['ms_per_step'], 'dtype': args.dtype,
                'note': 'same workload and entry point with --in-flight 1: step t+1 is enqueued behind step t on one stream, '
                        'so every emit pass and every search drain tail is exposed'}
        if same_work is not None:
            line['same_per_gpu_work_as_n1'] = {
                'per_gpu_batch': same_work['B'], 'value': same_work['value'], 'ms_per_step': same_work['ms_per_step'],
                'note': 'BASELINE configs[1] batch per GPU, for a weak-scaling comparison against the N=1 line'}
        if n_gpus == 1 and not args.no_cpu_baseline and not args.gt:
            line['cpu_baseline'] = cpu_baseline(head['batch'], N, C)
        print(json.dumps(line), flush=True)
    if exchange:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
