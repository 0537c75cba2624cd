#!/usr/bin/env python3
"""bench.py -- MPC solves/s of the batched shooting solver (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--dtype f64|f32] [--gt SC] [--in-flight F]

One "step" = one pass of the hot path over one batch of synthetic two-vehicle intersection
scenarios per GPU (SURVEY.md section 8d generator), inputs already resident in HBM:
    search kernel (C=256 candidates x N=20 steps x 4 RK4 sub-steps, cost, verdicts, arg-min)
  + emit kernel (winner trajectory / controls)
  + for N > 1: RCCL all-gather of the first-step controls u*[:, :, 0].

The headline (`value`, `dtype`) is the float64 entry igt_solve_batch_f64 -- the reference's own precision
(kinematic_bicycle_model_frenet.py:70-127 and mpc.py are float64 end to end) -- on BASELINE configs[1]
(B = 4096 per GPU, SURVEY 8d's lattice candidates).  The same JSON line carries, timed in the same run:
  * `one_solve_in_flight`  -- the same workload with every step waiting for the previous one (what a closed MPC loop
                              sees), with the kernel durations the `roofline` block refers to;
  * `tracking_family`      -- the same batch through IGT_CAND_TRACK, the family MPC_Planner / igtmpc.evaluate use;
  * `f32_path`             -- the float32 entry;
  * `configs`              -- BASELINE configs[2] (B = 65 536) and configs[4] (gt_mpc, B = 65 536; lattice and tracking).

Steps are pipelined (`--in-flight F`, default 4): step t is enqueued on handle / stream t mod F -- every handle owns
its workspace and its output buffers, the inputs are read-only -- so the emit pass (one roll-out of latency) and the
drain tail of one step's search overlap the search pass of the next instead of being exposed.  All K steps complete
inside the timed region.  Before the W warm-up steps an untimed settle phase (`--settle-ms`, default 150 ms of the same
steps) lets the shader clock reach its working point: with it a 20-step and a 200-step call report the same rate.

N > 1: one process per GPU, each solving its own contiguous shard (weak scaling, no data-path collective).
The driver launches that as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`; a plain
`python bench.py --gpus N` (no torchrun variables) spawns exactly that command itself -- before this process
touches a GPU -- and fails loudly if it cannot.  Rank 0 prints ONE JSON line.
"""
import argparse
import gc
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
PEAK_CLOCK_GHZ = 2.4           # MI355X_MICROARCH.md: peak engine clock
N_HORIZON, N_CAND = 20, 256
PROFILE_ROUND = 'r04'          # profiles/<round>_pmc_*.json: the rocprofv3 --pmc passes of this command (tools/collect.sh)


def default_batch(n_gpus, gt=0):
    """Scenarios per GPU: BASELINE configs[1] (4096) up to 4 GPUs; configs[3] (262 144 sharded 8 ways = 32 768) at 8;
    configs[4] (gt_mpc) is quoted at 65 536 scenarios per GPU whatever the number of GPUs (weak scaling)."""
    if gt:
        return 65536
    return 32768 if n_gpus >= 8 else 4096


def workload_name(B, n_gpus, gt, cand='lattice'):
    if gt:
        tag = (f'BASELINE configs[4] (gt_mpc, batch=65536 per GPU{", x%d GPUs" % n_gpus if n_gpus > 1 else ""})'
               if B == 65536 else 'custom gt_mpc batch')
    elif B == 4096:
        tag = 'BASELINE configs[1]' if n_gpus == 1 else f'BASELINE configs[1] per GPU x{n_gpus}'
    elif B == 65536 and n_gpus == 1:
        tag = 'BASELINE configs[2]'
    elif B == 32768 and n_gpus == 8:
        tag = 'BASELINE configs[3] (262144 scenarios sharded 8 ways)'
    else:
        tag = 'custom batch'
    fam = {'lattice': f'{N_CAND} lattice candidates', 'track': f'{N_CAND} tracking candidates (IGT_CAND_TRACK)'}[cand]
    return (f'{tag}: batch={B} two-vehicle intersection scenarios per GPU (all 8 sc variants tiled), horizon N={N_HORIZON}, '
            f'{fam}, 4 RK4 sub-steps, Frenet bicycle model, C_inf terminal set')


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
    """The oracle's C restatement (oracle/igt_oracle.c, -O3 without fast-math, kind "port") on this host's cores, on a
    bounded sample of the SAME workload; the numpy oracle on a smaller slice beside it.
    Checker code timed as a baseline only."""
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
    # the numpy oracle (vectorised over scenarios x candidates, one process)
    sn = slice(0, min(48, B))
    t0 = time.perf_counter()
    O.solve_batch(f('x0', sn), f('u_prev', sn), f('kparams', sn), batch['flags'][sn], f('obs_xy', sn), A, b, P, C=C)
    tn = time.perf_counter() - t0
    return {'value': done / t_all, 'unit': 'solves/s', 'cores': cores, 'kind': 'port',
            'sample': f'first {done} scenarios of the same batch, float64 C restatement (oracle/igt_oracle.c, gcc -O3, no '
                      f'fast-math), OpenMP over scenarios; reference mpc.py (CasADi/IPOPT) cannot run on this image',
            'single_thread_value': (sl.stop - sl.start) / t1,
            'numpy_oracle_value': (sn.stop - sn.start) / tn,
            'numpy_oracle_sample': f'first {sn.stop - sn.start} scenarios, oracle/np_oracle.py, one process'}


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


def rehearse_cpu(args, world, rank):
    """CPU rehearsal of the launch path (tests only: `--rehearse-cpu`): the ranks meet over gloo, all-gather a token,
    take the MAX of a per-rank time like the timed region does, and rank 0 assembles the SAME JSON line the GPU run
    prints -- from placeholder measurements (every rate is null) -- so that the multi-GPU line, the gt_mpc one included,
    is exercised without a GPU.  No solve."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    tok = torch.tensor([rank], dtype=torch.int64)
    got = [torch.zeros_like(tok) for _ in range(world)]
    dist.all_gather(got, tok)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # the exchange's self-check (igtmpc.sharding.verify_gathered) on stand-in shards: every rank's "first-step controls" are a
    # function of its rank, NaN rows (infeasible scenarios) included; a corrupted copy must be refused on every rank
    from igtmpc.sharding import allgather_controls, verify_gathered
    Bl = 24
    u0 = torch.arange(Bl * 2, dtype=torch.float64).reshape(Bl, 2) * 0.37 + 1000.0 * rank
    u0[3::5] = float('nan')
    g = allgather_controls(u0, B_total=Bl * world)
    chk = verify_gathered(u0, g)
    bad = g.clone()
    bad[(Bl * (world - 1) + 1) % (Bl * world), 1] += 1e-9          # one bit of another rank's block
    chk_bad = verify_gathered(u0, bad) if world > 1 else dict(ok=False)
    if rank == 0:
        B = args.batch if args.batch else default_batch(world, args.gt)
        fake = lambda Bm: dict(dtype=args.dtype, B=Bm, elapsed=float(t.item()), steps=args.steps, value=None, ms_per_step=None,
                               search_ms=None, emit_ms=None, rd=0, wr=0, feasible=None, in_flight=args.in_flight,
                               lane_search_ms=None, lane_emit_ms=None, overlap_checked=None,
                               exchange=dict(chk, lanes=args.in_flight) if world > 1 else None)
        same = fake(default_batch(1)) if (world > 1 and B != default_batch(1) and not args.batch and not args.gt) else None
        line = assemble_line(args, fake(B), n_gpus=world, world=world, backend=dist.get_backend(), exchange=world > 1,
                             same_work=same, n_layers=3 if args.gt else 0)
        line.update({'rehearsal': True, 'backend': dist.get_backend(), 'corrupted_gather_refused': not chk_bad['ok'],
                     'tokens': [int(x.item()) for x in got], 'max_over_ranks_s': float(t.item()),
                     'spawned_by_bench': os.environ.get('IGT_BENCH_SPAWNED') == '1'})
        print(json.dumps(line), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


def pmc_profile(dtype, gt, B, cand='lattice'):
    """The committed rocprofv3 --pmc summary of this workload, if it was collected at the kernel sources the library is
    built from (otherwise None: a counter of another binary is not quoted)."""
    name = f'{PROFILE_ROUND}_pmc_{dtype}{"_gt%d" % gt if gt else ""}{"_track" if cand == "track" else ""}_b{B}.json'
    path = os.path.join(ROOT, 'profiles', name)
    if not os.path.exists(path):
        return None, None
    try:
        with open(path) as f:
            tj = json.load(f)
    except Exception:
        return None, None
    if tj.get('source_hash') != source_hash() or tj.get('batch') != B:
        return None, None
    return tj, (f'{os.path.relpath(path, ROOT)} (rocprofv3 --pmc passes of this command, one solve in flight, at source hash '
                f'{tj.get("source_hash")}; not measured by this run)')


def assemble_line(args, head, n_gpus, world, backend, exchange, other=None, serial=None, tracking=None, same_work=None,
                  configs=None, cpu=None, n_layers=0):
    """The one JSON line, from the measurements (a pure function: tests/test_host_logic.py builds the N = 8 gt_mpc line
    from placeholders)."""
    B = head['B']
    other_dtype = 'f32' if args.dtype == 'f64' else 'f64'
    # kernel durations of the roofline block: measured with ONE solve in flight (HIP events on the launch stream) --
    # `serial` when the headline is pipelined, the headline itself when it is not
    ref = serial if serial is not None else head
    search_ms, emit_ms, rd = ref['search_ms'], ref['emit_ms'], head['rd']
    bytes_per_launch = (rd + 12) * B          # search kernel: reads inputs, writes cost/argmin/status partials
    ach = bytes_per_launch / (search_ms * 1e-3) / 1e9 if search_ms else None
    tj, pmc_src = pmc_profile(args.dtype, args.gt, B)
    d = (tj or {}).get('derived', {})
    traffic = (tj or {}).get('search_kernel_hbm_bytes_per_launch')
    valu = {'bound': 'valu_busy', 'unit': 'fraction of SIMD cycles the vector ALU is busy',
            'definition': 'SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles), counters and shader clock '
                          '(GRBM_GUI_ACTIVE) from the same rocprofv3 --pmc passes of this command',
            'regime': 'one solve in flight (the counter passes serialise the kernels)',
            'achieved': d.get('simd_valu_busy_fraction'), 'peak': 1.0, 'frac': d.get('simd_valu_busy_fraction'),
            'valu_instructions_per_launch': d.get('valu_instructions_per_launch'),
            'valu_instructions_per_unit': d.get('valu_instructions_per_unit'),
            'busy_cycles_per_valu_instruction': d.get('busy_cycles_per_valu_instruction'),
            'mean_waves_resident_per_simd': d.get('mean_waves_resident_per_simd'),
            'shader_clock_ghz_measured': d.get('shader_clock_GHz_during_search'),
            'kernel_ms_under_profiler': d.get('search_kernel_ms'),
            'issue_fraction_at_4_cycles_per_instruction': d.get('valu_issue_fraction_4_cycles_per_instruction'),
            'microbenchmark': d.get('microbenchmark_cycles_per_instruction'),
            'source': pmc_src}
    if d.get('valu_instructions_per_launch') and head['ms_per_step']:
        # the same instruction count against the whole (pipelined) step `value` is made of, at the measured busy cycles
        # per instruction: how much of the step the vector ALUs would be busy if only that work existed
        clk = d.get('shader_clock_GHz_during_search') or PEAK_CLOCK_GHZ
        cpi = d.get('busy_cycles_per_valu_instruction') or 4.0
        valu['busy_fraction_of_whole_step'] = (d['valu_instructions_per_launch'] * cpi /
                                               (SIMDS * head['ms_per_step'] * 1e-3 * clk * 1e9))
    arith = {'f64': 'float64 throughout (stage derivatives, state, cost, verdicts)',
             'f32': 'float32 stage derivatives + float64 state accumulators, cost and verdicts'}
    F = head['in_flight']
    line = {
        'metric': 'mpc_solves_per_sec', 'value': head['value'], 'unit': 'solves/s', 'n_gpus': n_gpus,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': head['ms_per_step'],
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': workload_name(B, n_gpus, args.gt, args.cand),
                   'per_gpu_batch': B, 'global_batch': B * n_gpus,
                   'entry_point': f'igt_solve_batch_{args.dtype}',
                   'arithmetic': arith[args.dtype],
                   'parallelism': (f'scenario shards x{n_gpus}, one process per GPU, all-gather of u*[:, :, 0] on its own '
                                   f'stream under the next step') if exchange else 'single GPU',
                   'ranks_seen': world if exchange else 1, 'backend': backend,
                   'solves_in_flight': F,
                   'pipelining': (f'step t runs on handle / stream t mod {F} (own workspace and output buffers each); '
                                  f'all {args.steps} steps complete inside the timed region') if F > 1 else 'none',
                   'settle_ms': args.settle_ms,
                   'cost': (f'gt_mpc value net V_GT_sc{args.gt} ({n_layers - 1} hidden layers, igtmpc.shipped_value_net, identity '
                            f'normalisation)') if args.gt else 'mpc progress cost',
                   'feasible_fraction': head['feasible']},
        'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': ach / HBM_PEAK_GBS if ach else None, 'traffic': traffic, 'traffic_source': pmc_src,
                     'kernel': f'search kernel of igt_solve_batch_{args.dtype} (+ value kernels in gt mode)',
                     'kernel_ms': search_ms, 'algorithmic_bytes_per_solve': rd + 12,
                     'regime': 'one solve in flight: kernel_ms is the search kernel alone on the GPU (HIP events on its '
                               'launch stream), the figure `one_solve_in_flight.ms_per_step` is made of; the headline '
                               '`value` overlaps %d solves, see `pipelined`' % F if F > 1 else 'one solve in flight',
                     'note': 'BASELINE.json names the HBM roofline; the path is vector-ALU-bound (arithmetic intensity ~1e4 '
                             'flop/B) -- the bound that applies is `valu_roofline`'},
        'valu_roofline': valu,
        'kernels_ms': {'search': search_ms, 'emit': emit_ms, 'regime': 'one solve in flight'},
        'whole_solve_bytes': rd + head['wr'],
        'precision': {'f64': 'igt_solve_batch_f64: <= 1e-9 of the float64 oracle on every trajectory (curvature '
                             'break-point straddlers included)',
                      'f32': 'igt_solve_batch_f32: <= 1e-5*max(1,|ref|) except trajectories whose curvature switch is '
                             'decided inside float32 noise (share measured in tests/test_gpu_parity.py)'},
    }
    # self-checks of the timed regime (untimed, after it): lanes bit-equal to a solve alone; gathered vectors complete
    line['overlap_checked'] = head.get('overlap_checked')
    ex = head.get('exchange')
    line['ranks_seen'] = ex['ranks_seen'] if ex else 1
    if exchange:
        line['exchange_checked'] = bool(ex and ex['ok'])
        line['per_rank_B_local'] = ex['per_rank_B_local'] if ex else None
        line['config']['ranks_seen'] = ex['ranks_seen'] if ex else world
    if F > 1:
        line['pipelined'] = {
            'solves_in_flight': F, 'ms_per_step': head['ms_per_step'],
            'kernels_ms_while_overlapped': {'search': head.get('lane_search_ms'), 'emit': head.get('lane_emit_ms')},
            'note': 'durations of one lane\'s kernels (HIP events on the lane\'s stream) while the other lanes\' kernels share the '
                    'GPU: each kernel takes longer than alone, the step rate is higher because tails and emit passes overlap'}
    if tracking is not None:
        tt, tsrc = pmc_profile(args.dtype, args.gt, B, 'track')
        line['tracking_family'] = {
            'value': tracking['value'], 'unit': 'solves/s', 'ms_per_step': tracking['ms_per_step'], 'dtype': args.dtype,
            'solves_in_flight': tracking['in_flight'],
            'kernels_ms': {'search': tracking['search_ms'], 'emit': tracking['emit_ms'], 'regime': 'one solve in flight'},
            'feasible_fraction': tracking['feasible'],
            'valu_busy_fraction': ((tt or {}).get('derived', {}) or {}).get('simd_valu_busy_fraction'),
            'valu_instructions_per_unit': ((tt or {}).get('derived', {}) or {}).get('valu_instructions_per_unit'),
            'pmc_source': tsrc,
            'note': 'same batch, entry point and timing with cand_mode = IGT_CAND_TRACK (256 candidates, one pass): '
                    'the default family of MPC_Planner and igtmpc.evaluate'}
    if other is not None:
        line[f'{other_dtype}_path'] = {
            'value': other['value'], 'unit': 'solves/s', 'ms_per_step': other['ms_per_step'], 'dtype': other_dtype,
            'entry_point': f'igt_solve_batch_{other_dtype}', 'arithmetic': arith[other_dtype],
            'kernels_ms': {'search': other['search_ms'], 'emit': other['emit_ms'], 'regime': 'one solve in flight'},
            'steps': args.steps, 'warmup': args.warmup, 'feasible_fraction': other['feasible'],
            'note': 'same workload, same run, timed exactly like the headline'}
    if serial is not None:
        line['one_solve_in_flight'] = {
            'value': serial['value'], 'ms_per_step': serial['ms_per_step'], 'dtype': args.dtype,
            'kernels_ms': {'search': serial['search_ms'], 'emit': serial['emit_ms']},
            'note': 'same workload and entry point with --in-flight 1: step t+1 is enqueued behind step t on one stream, '
                    'so every emit pass and every search drain tail is exposed -- what a closed MPC loop, where step t+1 '
                    'needs step t, gets'}
    if same_work is not None:
        line['same_per_gpu_work_as_n1'] = {
            'per_gpu_batch': same_work['B'], 'value': same_work['value'], 'ms_per_step': same_work['ms_per_step'],
            'note': 'BASELINE configs[1] batch per GPU, for a weak-scaling comparison against the N=1 line'}
    if configs:
        line['configs'] = configs
    if cpu is not None:
        line['cpu_baseline'] = cpu
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=None,
                    help='scenarios per GPU per step (default: 4096 = BASELINE configs[1]; 32768 = configs[3] at --gpus 8; '
                         '65536 = configs[4] with --gt)')
    ap.add_argument('--dtype', default='f64', choices=['f64', 'f32'],
                    help="entry point behind the headline `value` (default f64: the reference's precision)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the other-precision / serial / tracking blocks')
    ap.add_argument('--no-configs', action='store_true', help='skip the configs[2] / configs[4] sub-records of the default run')
    ap.add_argument('--gt', type=int, default=0, metavar='SC',
                    help='gt_mpc cost with the shipped value net of scenario SC (1..8; BASELINE configs[4]); 0 = mpc cost')
    ap.add_argument('--cand', default='lattice', choices=['lattice', 'track'], help='candidate family of the headline')
    ap.add_argument('--in-flight', type=int, default=4, choices=[1, 2, 3, 4, 5, 6, 8],
                    help='solves in flight: step t runs on handle/stream t mod F (each handle owns its workspace and '
                         'output buffers), so the emit pass and the drain tail of one step overlap the search pass of the next')
    ap.add_argument('--settle-ms', type=float, default=150.0,
                    help='untimed steps of the same work before the warm-up, until this much wall time has passed')
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
        sys.exit(rehearse_cpu(args, world, rank))

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

    from igtmpc import BatchSolver, shipped_value_net
    from igtmpc.cinf import cinf_halfplanes
    from igtmpc.scenarios import make_batch

    N, C = N_HORIZON, N_CAND
    B = args.batch if args.batch else default_batch(n_gpus, args.gt)
    batches = {}
    checks_log = []          # one entry per measurement: the untimed self-checks of the regime it timed

    def measure(dtype, Bm, steps, warmup, in_flight=None, cand_mode='lattice', gt=None, settle_ms=None):
        """settle + W untimed + K timed steps of the `dtype` entry point at Bm scenarios per GPU, then per-kernel HIP
        events (one solve at a time, and -- pipelined runs -- with the lanes overlapping)."""
        gt = args.gt if gt is None else gt
        npdt = np.float64 if dtype == 'f64' else np.float32
        td = torch.float64 if dtype == 'f64' else torch.float32
        if (Bm, dtype) not in batches:
            batches.clear()                                  # one batch resident at a time (65 536 x f64 is ~50 MB on the host)
            batches[(Bm, dtype)] = make_batch(Bm, N=N, dtype=npdt, offset=rank * Bm)
        batch = batches[(Bm, dtype)]
        layers = shipped_value_net(gt)['layers'] if gt else []      # the weights the package ships (identity whitening:
        keys = ['x0', 'u_prev', 'kparams', 'flags', 'obs_xy'] + (['tv_sv', 'enc'] if gt else [])   # the statistics are not shipped)
        dargs = [torch.from_numpy(batch[k].view(np.int32) if batch[k].dtype == np.uint32 else batch[k]).cuda(dev) for k in keys]
        F = in_flight or args.in_flight
        solvers, outs, lanes = [], [], []
        for _ in range(F):
            sv = BatchSolver(N=N, C=C, n_obs=1, device=dev, dtype=dtype, cost_mode='value_net' if gt else 'progress',
                             cand_mode=cand_mode)
            sv.set_cinf(*cinf_halfplanes(dt=sv.params.dt, jerk=sv.params.jerk_limit))
            sv.set_concurrency(F)                         # F solves in flight: the search kernels share the wave slots in pairs
            if gt:
                sv.set_value_net(layers)
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
            # one staging buffer and one gathered vector per lane: a lane's exchange is ordered behind its own solve and
            # ahead of its next one by the lane's stream alone
            u0_buf = [torch.empty((Bm, 2), dtype=td, device=f'cuda:{dev}') for _ in range(F)]
            gathered = [torch.empty((Bm * world, 2), dtype=td, device=f'cuda:{dev}') for _ in range(F)]
        # the self-check behind the timed region reads every lane's outputs: a lane the overlapped steps (settle, warm-up, timed:
        # all of them run F in flight) never wrote would show this sentinel.  Filled HERE, ahead of the settle phase, not between
        # it and the warm-up steps: the first torch kernel of a process takes the host milliseconds to load, the device idles
        # and clocks down meanwhile, and W = 5 warm-up steps (1 ms) do not bring the clock back -- the 20-step region behind them
        # then runs 5 % slower (profiles/r04_idle_gap_before_warmup.txt; same box, alternating: 22.5 against 21.3 M solves/s)
        for o in outs:
            o['argmin'].fill_(-5)
        if exchange:
            for g in gathered:
                g.fill_(-5.0)
        torch.cuda.synchronize(dev)
        step_no = [0]
        drain_hint = os.environ.get('IGT_BENCH_NO_DRAIN_HINT') != '1'      # A/B switch (tools only)

        conc = [F] * F                    # what each handle was last told about the solves in flight

        def step(in_flight_now=None, ev=None):
            # in_flight_now: how many solves are in flight once this one is enqueued, when the caller knows (the timed loop does:
            # it enqueues exactly K steps and waits, so the last F - 1 steps see the pipeline drain) -- igt_set_concurrency is told,
            # as include/igtmpc.h asks of a caller that overlaps solves (below 3 in flight a search takes two waves per SIMD again
            # and the emit pass rolls in pieces: a solve with the device mostly to itself)
            # step t on handle / stream t mod F; the inputs are read-only, every lane has its own outputs and its own exchange
            # buffers.  The staging copy of u*[:, :, 0] and the all-gather are enqueued behind the lane's solve (the collective
            # itself runs on the process group's stream, which waits for the lane and which the lane waits for).  The search
            # kernels are persistent -- every wave slot stays taken until a search pass drains -- so each small kernel of
            # another stream gets on the chip at the next drain: a lane's copy and collective add latency to that lane, not
            # device time, and the other lanes' solves cover it.
            q = step_no[0] % F
            lane = lanes[q] if lanes[q] is not None else torch.cuda.current_stream(dev)
            step_no[0] += 1
            want = F if in_flight_now is None else max(1, min(F, in_flight_now))
            if want != conc[q] and drain_hint:
                solvers[q].set_concurrency(want)
                conc[q] = want
            with torch.cuda.stream(lane):
                solvers[q].solve(*dargs, out=outs[q])
                if exchange:
                    u0_buf[q].copy_(outs[q]['u'][:, :, 0])
                    dist.all_gather_into_tensor(gathered[q], u0_buf[q])
                if ev is not None:
                    ev.record()

        def fence():
            torch.cuda.synchronize(dev)
            if exchange:
                dist.barrier()
            torch.cuda.synchronize(dev)

        # settle: the same steps, untimed, until the clocks have reached their working point (every rank runs the same
        # number: the count is agreed through the fence below, not by each rank's own clock -- an all-gather per step)
        settle = args.settle_ms if settle_ms is None else settle_ms
        n_settle = 0
        if settle > 0:
            if exchange:                       # the first collective sets the communicator up: not a step time
                step(); torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            step(); torch.cuda.synchronize(dev)
            per = max(time.perf_counter() - t0, 1e-5)
            n_settle = int(min(2000, max(F, settle * 1e-3 / per)))
            if exchange:
                ns = torch.tensor([n_settle], dtype=torch.int64, device=f'cuda:{dev}')
                dist.all_reduce(ns, op=dist.ReduceOp.MAX)
                n_settle = int(ns.item())
            for _ in range(n_settle):
                step()
        for _ in range(warmup):
            step()
        fence()
        evs = None
        if os.environ.get('IGT_BENCH_EVENTS') == '1':      # tools only: when each timed step completes (tools/region_events.py)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
            evs[steps].record()
        t0 = time.perf_counter()
        for i in range(steps):
            step(in_flight_now=steps - i, ev=evs[i] if evs else None)      # the last F - 1 steps: fewer solves behind this one than lanes
        issued = time.perf_counter() - t0          # host time to enqueue the timed steps (close to `elapsed`: host-bound)
        fence()
        elapsed = time.perf_counter() - t0
        if evs:
            print(f'[events] {dtype} {cand_mode} F={F}: wall {elapsed * 1e3:.3f} ms, issued in {issued * 1e3:.3f}; steps complete at ' +
                  ' '.join(f'{evs[steps].elapsed_time(e):.2f}' for e in evs[:steps]), file=sys.stderr, flush=True)
        if exchange:
            t = torch.tensor([elapsed], dtype=torch.float64, device=f'cuda:{dev}')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        # ---- untimed self-check of the regime just timed: every lane's outputs -- written by solves
        # that overlapped the other lanes' -- must be, bit for bit, what ONE solve of the same batch gives with the device to
        # itself; and every lane's gathered action vector must hold every rank's u*[:, :, 0] (checksums all-reduced from the
        # shards, igtmpc.sharding.verify_gathered).  A line whose check fails is not printed (main()).
        torch.cuda.synchronize(dev)
        solvers[0].set_concurrency(1)
        alone = solvers[0].solve(*dargs)                   # fresh output buffers, current stream, nothing else in flight
        torch.cuda.synchronize(dev)
        solvers[0].set_concurrency(F)
        bits = lambda t: t.contiguous().view(torch.int64 if t.element_size() == 8 else torch.int32)
        overlap_checked = all(bool(torch.equal(bits(o[k]), bits(alone[k]))) for o in outs for k in ('x', 'u', 'cost', 'argmin', 'status'))
        exch = None
        if exchange:
            from igtmpc.sharding import verify_gathered
            u0_alone = alone['u'][:, :, 0].contiguous()
            res = [verify_gathered(u0_alone, g) for g in gathered]
            exch = dict(ok=all(r['ok'] for r in res), ranks_seen=res[0]['ranks_seen'],
                        per_rank_B_local=res[0]['per_rank_B_local'], lanes=F)
        feasible = float((alone['status'] == 0).float().mean().item())
        del alone
        checks_log.append(dict(dtype=dtype, B=Bm, in_flight=F, cand=cand_mode, gt=gt, overlap_checked=overlap_checked, exchange=exch))

        # per-kernel durations: HIP events recorded by the library on the launch stream, same workload.
        # (1) pipelined runs: every lane profiles its own kernels while the other lanes' kernels overlap them
        lane_s, lane_e = None, None
        if F > 1:
            for sv in solvers:
                sv.set_concurrency(F)
                sv.set_profiling(True)
            ls, le = [], []
            n_prof = max(4 * F, min(steps, 48))
            for t_ in range(n_prof):
                q = t_ % F
                if t_ >= F:
                    a, e = solvers[q].kernel_ms()
                    ls.append(a); le.append(e)
                with torch.cuda.stream(lanes[q]):
                    solvers[q].solve(*dargs, out=outs[q])
            torch.cuda.synchronize(dev)
            for sv in solvers:
                sv.set_profiling(False)
            lane_s, lane_e = float(np.mean(ls)), float(np.mean(le))
        # (2) one solve at a time (with the whole device to itself: two waves per SIMD again)
        solver.set_concurrency(1)
        solver.set_profiling(True)
        ks, ke = [], []
        for _ in range(max(8, min(steps, 50))):
            solver.solve(*dargs, out=out)
            a, e = solver.kernel_ms()
            ks.append(a)
            ke.append(e)
        solver.set_profiling(False)
        rd, wr = solver.algorithmic_bytes_per_solve()
        res = dict(dtype=dtype, B=Bm, elapsed=elapsed, steps=steps, value=Bm * n_gpus * steps / elapsed,
                   ms_per_step=elapsed / steps * 1e3, search_ms=float(np.mean(ks)), emit_ms=float(np.mean(ke)),
                   lane_search_ms=lane_s, lane_emit_ms=lane_e, in_flight=F, settle_steps=n_settle, host_issue_ms=issued / steps * 1e3,
                   rd=rd, wr=wr, feasible=feasible, batch=batch, overlap_checked=overlap_checked, exchange=exch)
        # ordered teardown ("exit-time abort"): nothing of this measurement is left to interpreter or
        # static destructors -- device idle, exchange buffers and lane streams dropped, handles destroyed, cache released
        torch.cuda.synchronize(dev)
        if exchange:
            del u0_buf, gathered
        for sv in solvers:
            sv.close()
        del dargs, outs, out, solver, solvers, lanes
        gc.collect()
        torch.cuda.empty_cache()
        return res

    head = measure(args.dtype, B, args.steps, args.warmup, cand_mode=args.cand)
    cpu = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline and not args.gt:
        cpu = cpu_baseline(head['batch'], N, C)       # before the other measurements replace the resident batch
    other_dtype = 'f32' if args.dtype == 'f64' else 'f64'
    # one solve at a time (every step waits for the previous one's emit pass), for comparison with the pipelined headline
    serial = measure(args.dtype, B, args.steps, args.warmup, in_flight=1, cand_mode=args.cand) \
        if args.in_flight > 1 and not args.no_secondary else None
    # the candidate family the planner and the closed-loop driver use by default (state-feedback steering):
    # the headline stays on SURVEY 8d's lattice, this is the same batch through the family that gives the better answers
    tracking = None if (args.no_secondary or args.cand == 'track') else \
        measure(args.dtype, B, args.steps, args.warmup, cand_mode='track')
    # the driver computes scaling efficiency from the per-N values; at N = 8 the per-GPU batch is configs[3]'s 32 768,
    # not configs[1]'s 4096, so the same-per-GPU-work figure is measured beside it in the same run
    same_work = None
    if n_gpus > 1 and B != default_batch(1) and not args.batch and not args.gt:
        same_work = measure(args.dtype, default_batch(1), args.steps, args.warmup, cand_mode=args.cand)
    other = None if args.no_secondary else measure(other_dtype, B, args.steps, args.warmup, cand_mode=args.cand)

    # the other single-GPU BASELINE configurations, driver-timed in the same run (default invocation only):
    # configs[2] = 65 536 scenarios; configs[4] = gt_mpc at 65 536 (V_GT_sc1; lattice and tracking candidates)
    configs = []
    if n_gpus == 1 and not args.no_configs and not args.batch and not args.gt and args.cand == 'lattice':
        ks, kw = max(8, min(args.steps, 16)), 3
        for name, kwm in (('configs[2]', dict(Bm=65536, cand_mode='lattice', gt=0)),
                          ('configs[4]', dict(Bm=65536, cand_mode='lattice', gt=1)),
                          ('configs[4] through the tracking family', dict(Bm=65536, cand_mode='track', gt=1)),
                          ('configs[4] with V_GT_sc3 (3 hidden layers)', dict(Bm=65536, cand_mode='lattice', gt=3))):
            m = measure(args.dtype, kwm['Bm'], ks, kw, in_flight=1, cand_mode=kwm['cand_mode'], gt=kwm['gt'], settle_ms=100.0)
            configs.append({'config': name, 'workload': workload_name(kwm['Bm'], 1, kwm['gt'], kwm['cand_mode']),
                            'value': m['value'], 'unit': 'solves/s', 'ms_per_step': m['ms_per_step'], 'dtype': args.dtype,
                            'steps': ks, 'warmup': kw, 'solves_in_flight': 1,
                            'kernels_ms': {'search_incl_value_net': m['search_ms'], 'emit': m['emit_ms']},
                            'feasible_fraction': m['feasible']})

    # a `value` measured in a regime whose answers were not confirmed is not printed (every rank learns of any rank's failure:
    # verify_gathered reduces its flag; the lane check is reduced here)
    bad = [m for m in checks_log if m['overlap_checked'] is not True or (exchange and not (m['exchange'] and m['exchange']['ok']))]
    n_bad = len(bad)
    if exchange:
        nb = torch.tensor([n_bad], dtype=torch.int64, device=f'cuda:{dev}')
        dist.all_reduce(nb, op=dist.ReduceOp.MAX)
        n_bad = int(nb.item())
    if n_bad:
        if rank == 0:
            print('bench.py: the self-check of the timed regime FAILED (overlapped lanes differ from a solve alone, or a gathered '
                  'action vector is incomplete) -- no line is printed: ' +
                  '; '.join(f"{m['dtype']} {m['cand']} gt={m['gt']} B={m['B']} F={m['in_flight']} overlap_checked={m['overlap_checked']} "
                            f"exchange={m['exchange']}" for m in bad),
                  file=sys.stderr, flush=True)
        if exchange:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(3)
    if rank == 0:
        n_layers = len(shipped_value_net(args.gt)['layers']) if args.gt else 0
        line = assemble_line(args, head, n_gpus, world, backend, exchange, other=other, serial=serial, tracking=tracking,
                             same_work=same_work, configs=configs, cpu=cpu, n_layers=n_layers)
        line['config']['settle_steps'] = head['settle_steps']
        line['host_issue_ms_per_step'] = head.get('host_issue_ms')      # host time to enqueue one step; near ms_per_step: host-bound
        print(json.dumps(line), flush=True)
    # ordered exit: measurements dropped, device idle, process group destroyed -- before the interpreter starts tearing down
    del head, serial, tracking, same_work, other, batches
    gc.collect()
    torch.cuda.synchronize(dev)
    if exchange:
        dist.barrier()
        torch.cuda.synchronize(dev)
        dist.destroy_process_group()
    torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
