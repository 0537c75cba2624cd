"""CPU: host-side logic -- route geometry, scenario generator, C_inf, C-ABI surface."""
import ctypes as ct
import json
import re
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_route_tables_match_reference_constants(golden_dir):
    from igtmpc import routes as R
    with open(f'{golden_dir}/route_constants.json') as f:
        ref = json.load(f)
    for r, k in ref.items():
        rid = R.ROUTE_ID[r]
        p0 = R.frenet2global(rid, 0.0)
        assert np.allclose(p0, [k['x0'], k['y0']], atol=1e-12)
        assert R.psi_ref(rid, 0.0) == k['heading0'] and R.psi_ref(rid, 999.0) == k['headingN']
        kp = R.kparams(rid)
        if k['straight']:
            assert np.isinf(kp[0]) and kp[2] == 0
        else:
            assert kp[0] == k['b0'] and kp[1] == k['b1'] and kp[2] == k['Kv']
            # SURVEY 8a-2 table
            assert k['b0'] == (19.3 if r in R.LEFT else 10.7)
            assert abs(k['radius'] - (8.6 if r in R.LEFT else 11.4)) < 1e-9
            assert np.sign(k['Kv']) == (1 if r in R.LEFT else -1)


def test_frenet2global_is_arc_length_parametrised():
    from igtmpc import routes as R
    s = np.linspace(0, 60, 6001)
    for r in R.ROUTES:
        rid = np.full(s.shape, R.ROUTE_ID[r])
        xy = R.frenet2global(rid, s)
        d = np.hypot(*np.diff(xy, axis=0).T)
        # unit speed except the reference's own ~4 mm jump where it freezes the end coordinate
        assert np.abs(d - 0.01).max() < 1.2e-3
        if r in R.STRAIGHT:
            assert np.abs(d - 0.01).max() < 1e-12


def test_filter_preds_moves_obstacles_behind_ego():
    from igtmpc import routes as R
    obs = np.zeros((2, 1, 2, 5))
    obs[0, 0, 0] = 5.0      # ahead of an ego at origin heading +x
    obs[1, 0, 0] = -5.0     # behind
    out = R.filter_preds(np.zeros((2, 2)), np.zeros(2), obs)
    assert np.array_equal(out[0], obs[0]) and (out[1] == -20).all()


def test_scenario_tables():
    from igtmpc import routes as R
    assert len(R.SCENARIO_ROUTES) == 8
    for sc, pairs in enumerate(R.SCENARIO_ROUTES, start=1):
        for p in pairs:
            assert R.scenario_of(p) == sc
            e = R.scenario_encoding_sign(p, sc)
            assert sorted(e) == [-sc, sc]
    # utils.py:84-139 spot checks
    assert R.scenario_encoding_sign(('13', '23'), 1) == (1, -1)
    assert R.scenario_encoding_sign(('12', '42'), 1) == (-1, 1)      # '42' counts as origin 0
    assert R.scenario_encoding_sign(('12', '32'), 4) == (4, -4)      # left first
    assert R.scenario_encoding_sign(('14', '24'), 5) == (-5, 5)      # straight second


def test_make_batch_is_deterministic_and_well_formed():
    from igtmpc.scenarios import make_batch
    a, b = make_batch(130), make_batch(130)
    for k in a:
        assert np.array_equal(a[k], b[k])
    assert a['x0'].shape == (130, 7) and a['obs_xy'].shape == (130, 1, 2, 21) and a['flags'].dtype == np.uint32
    assert set(a['sc']) == set(range(1, 9))
    assert np.isfinite(a['x0']).all() and np.isfinite(a['obs_xy']).all()
    # shards differ from each other but are reproducible
    c = make_batch(130, offset=130)
    assert not np.array_equal(a['x0'], c['x0'])
    assert np.array_equal(c['x0'], make_batch(130, offset=130)['x0'])
    # 64-combination tiling continues across shards
    assert c['sc'][0] == (130 % 64) // 8 + 1


def test_cinf_properties():
    """PARITY UNPINNED vs polytope (absent): pinned by what defines the set (utils.py:588-627)."""
    from igtmpc.cinf import control_invariant_set
    A, b, V, it = control_invariant_set()
    assert it < 100 and 40 < len(b) < 120
    assert np.allclose(np.linalg.norm(A, axis=1), 1.0)
    # inside X (mpc.py:88-95)
    assert V[:, 0].min() >= -1 - 1e-9 and V[:, 0].max() <= 5 + 1e-9 and V[:, 1].max() <= 3 + 1e-9
    # numbers recorded by the survey's throw-away run (SURVEY section 7 hard part 5)
    assert abs(V[:, 1].min() - (-3.2416)) < 1e-4
    amax = min((bb - n0 * 4.5) / n1 for (n0, n1), bb in zip(A, b) if n1 > 1e-12)
    assert abs(amax - 0.9045) < 1e-4
    # control invariance: every vertex can be kept inside with an admissible input
    Am, r = np.array([[1, 0.1], [0, 1.0]]), 0.09
    for x in V:
        y = Am @ x
        best = min((A @ (y + np.array([0, u])) - b).max() for u in np.linspace(-r, r, 37))
        assert best < 1e-9
    # maximality (spot): points just outside a curved facet but inside X leave X under every input
    rng = np.random.default_rng(1)
    for _ in range(20):
        i = rng.integers(len(b))
        p = (V[i] + V[(i + 1) % len(V)]) / 2 + 1e-3 * A[i]
        if not (-1 <= p[0] <= 5 and -4 <= p[1] <= 3):
            continue
        # greedy best response: steer a towards the side that keeps v in range
        x, ok = p.copy(), True
        for _ in range(200):
            u = r if x[1] < 0 else -r
            x = Am @ x + np.array([0, u])
            if not (-1 - 1e-9 <= x[0] <= 5 + 1e-9 and -4 <= x[1] <= 3):
                ok = False
                break
        assert not ok
    # another discretisation converges too
    A2, b2, _, it2 = control_invariant_set(dt=0.05)
    assert it2 < 200 and len(b2) > len(b)


def test_shard_ranges_partition_the_batch():
    from igtmpc.sharding import shard_range
    for B in (0, 1, 7, 4096, 262144, 1000):
        for w in (1, 2, 3, 8):
            r = [shard_range(B, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == B
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads (no GPU needed) and exports exactly what include/igtmpc.h declares."""
    from igtmpc import _lib as L
    hdr = open(os.path.join(ROOT, 'include', 'igtmpc.h')).read()
    declared = set(re.findall(r'\b(igt_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    lib = L.load()
    for name in declared:
        assert hasattr(lib, name)
    dev = L.load(dev=True)                                        # the developer build exports the same ABI
    assert dev is not lib and all(hasattr(dev, name) for name in declared)
    assert lib.igt_version() == 201 == int(re.search(r"#define IGT_VERSION (\d+)", hdr).group(1))
    p = L.igt_params()
    assert lib.igt_params_default(ct.byref(p)) == 0
    # the numbers MPC_Planner.__init__ hard-codes (mpc.py:45-62)
    assert (p.N, p.n_rk4, p.C, p.n_obs) == (20, 4, 256, 1)
    assert (p.v_min, p.v_max, p.a_min, p.a_max, p.df_max) == (0, 5, -4, 3, 1)
    assert (p.jerk_limit, p.steer_rate_limit, p.ey_lim, p.d_min, p.w_u) == (0.9, 0.7, 0.2, 5.6, 0.05)
    assert abs(p.l_r - 2.235) < 1e-15 and ct.sizeof(L.igt_params) == 24 + 14 * 8 + 8 + 5 * 8 and p.refine_iters == 0
    assert (p.track_ke, p.track_span, p.track_beta_lim, p.track_env, p.track_vcap) == (0.3, 0.1, 0.7, 1.0, 1.0)
    # struct layout agrees with the header's field order
    fields = re.search(r'typedef struct igt_params \{(.*?)\} igt_params;', hdr, re.S).group(1)
    names = re.findall(r'\b(?:int32_t|double)\s+([^;]+);', fields)
    flat = [n.strip() for grp in names for n in grp.split(',')]
    assert flat == [f[0] for f in L.igt_params._fields_]


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No CPU fallback: without libigtmpc.so the product raises ImportError naming the build command."""
    from igtmpc import _lib as L
    monkeypatch.setattr(L, '_libs', {})
    monkeypatch.setattr(L, 'LIB_PATH', str(tmp_path / 'libigtmpc.so'))
    with pytest.raises(ImportError, match='build'):
        L.load()
    import igtmpc
    with pytest.raises(ImportError):
        igtmpc.BatchSolver()


# ----------------------------------------------------------------------------- bench.py launch path (no GPU needed)
def _run_bench(*argv, env_extra=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, 'bench.py'), *argv], env=env, capture_output=True,
                          text=True, timeout=240)


def test_bench_gpus_2_without_torchrun_variables_spawns_two_ranks():
    """`python bench.py --gpus 2` with no torchrun variables must start 2 ranks itself (never silently run one):
    rehearsed on CPU over gloo -- the ranks meet, all-gather a token, and rank 0 reports what it saw."""
    import json
    r = _run_bench('--gpus', '2', '--rehearse-cpu')
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['ranks_seen'] == 2 and line['tokens'] == [0, 1] and line['spawned_by_bench'] is True
    # the exchange's self-check ran over gloo: complete vector accepted on every rank, a one-bit corruption of another
    # rank's block refused
    assert line['exchange_checked'] is True and line['per_rank_B_local'] == [24, 24] and line['corrupted_gather_refused'] is True


def test_bench_multi_gpu_gt_line_is_assembled_without_a_gpu():
    """`bench.py --gpus N --gt 1` must emit the configs[4] line.  (a) two gloo ranks spawned by
    bench.py itself assemble the real line from placeholder measurements; (b) the N = 8 line through the same pure
    function: workload name, per-GPU batch 65 536, ranks seen, backend, no CPU baseline, null PMC fields."""
    import argparse
    import importlib.util
    import json
    r = _run_bench('--gpus', '2', '--gt', '1', '--rehearse-cpu')
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['ranks_seen'] == 2 and line['n_gpus'] == 2 and line['max_over_ranks_s'] == 2.0
    assert 'configs[4]' in line['config']['workload'] and line['config']['per_gpu_batch'] == 65536
    assert line['config']['global_batch'] == 131072 and line['config']['ranks_seen'] == 2 and line['config']['backend'] == 'gloo'
    assert 'V_GT_sc1' in line['config']['cost'] and 'cpu_baseline' not in line and line['scaling'] == 'weak'
    spec = importlib.util.spec_from_file_location('bench_mod3', os.path.join(ROOT, 'bench.py'))
    bm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bm)
    for gt, B in ((1, 65536), (0, 32768)):
        args = argparse.Namespace(dtype='f64', gt=gt, cand='lattice', steps=20, warmup=5, settle_ms=150.0, in_flight=4)
        m = lambda Bm, F=4: dict(dtype='f64', B=Bm, elapsed=1.0, steps=20, value=Bm * 8 * 20 / 1.0, ms_per_step=50.0, search_ms=3.6,
                                 emit_ms=0.13, lane_search_ms=4.0, lane_emit_ms=0.2, rd=452, wr=1512, feasible=0.78, in_flight=F)
        line = bm.assemble_line(args, m(B), n_gpus=8, world=8, backend='nccl', exchange=True, serial=m(B, 1),
                                same_work=None if gt else m(4096), n_layers=3 if gt else 0)
        assert line['n_gpus'] == 8 and line['config']['ranks_seen'] == 8 and line['config']['global_batch'] == 8 * B
        assert ('configs[4]' if gt else 'configs[3]') in line['config']['workload']
        assert (line.get('same_per_gpu_work_as_n1') is None) == bool(gt)
        assert line['roofline']['bound'] == 'hbm' and line['roofline']['kernel_ms'] == 3.6
        assert abs(line['roofline']['achieved'] - (452 + 12) * B / 3.6e-3 / 1e9) < 1e-9
        assert 'one solve in flight' in line['roofline']['regime'] and line['pipelined']['solves_in_flight'] == 4
        assert line['valu_roofline']['bound'] == 'valu_busy'
        # placeholders carry no checks: the keys are there and say so (main() refuses to print such a line on a GPU)
        assert line['overlap_checked'] is None and line['exchange_checked'] is False
        ok = dict(m(B), overlap_checked=True, exchange=dict(ok=True, ranks_seen=8, per_rank_B_local=[B] * 8, lanes=4))
        line = bm.assemble_line(args, ok, n_gpus=8, world=8, backend='nccl', exchange=True, serial=m(B, 1), n_layers=3 if gt else 0)
        assert line['overlap_checked'] is True and line['exchange_checked'] is True and line['ranks_seen'] == 8
        assert line['per_rank_B_local'] == [B] * 8
        json.dumps(line)


def test_bench_refuses_a_multi_gpu_line_it_cannot_measure():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('needs a host with fewer than 2 GPUs')
    r = _run_bench('--gpus', '2')
    assert r.returncode != 0 and 'refusing' in r.stderr and not r.stdout.strip()
    r = _run_bench('--gpus', '2', env_extra={'WORLD_SIZE': '4', 'RANK': '0'})      # launcher and flag disagree
    assert r.returncode != 0 and 'WORLD_SIZE=4' in r.stderr


def test_curvature_from_callable_recovers_exact_breakpoints():
    """Any callable K(s) of the reference's shape (evaluate.py:384-402 builds a casadi Function) is probed for
    (b0, b1, Kv): grid + bisection to the last representable s -- the exact float64 numbers the callable compares."""
    from igtmpc.vehicle import Curvature
    from igtmpc import routes as R
    for r in R.ROUTES:
        b0, b1, kv = (float(q) for q in R.kparams(R.ROUTE_ID[r]))
        K = (lambda b0, b1, kv: (lambda s: (kv if s >= b0 else 0.0) - (kv if s >= b1 else 0.0)))(b0, b1, kv)
        got = Curvature.from_callable(K).kparams
        assert got == ((b0, b1, kv) if kv != 0.0 else (float('inf'), float('inf'), 0.0)), r
        assert Curvature.from_callable(K) is Curvature.from_callable(K)           # cached on the callable
    with pytest.raises(ValueError):
        Curvature.from_callable(lambda s: 0.1)
    with pytest.raises(ValueError):
        Curvature.from_callable(lambda s: 0.1 * (int(s) % 2))


def test_bench_names_the_configuration_it_runs():
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bm)
    assert [bm.default_batch(n) for n in (1, 2, 4, 8)] == [4096, 4096, 4096, 32768]
    assert [bm.default_batch(n, 1) for n in (1, 8)] == [65536, 65536]          # configs[4]: 65 536 per GPU at any N
    assert 'configs[4]' in bm.workload_name(65536, 8, 1) and 'x8 GPUs' in bm.workload_name(65536, 8, 1)
    assert 'tracking candidates' in bm.workload_name(4096, 1, 0, 'track')
    assert 'configs[1]' in bm.workload_name(4096, 1, 0) and 'batch=4096' in bm.workload_name(4096, 1, 0)
    assert 'configs[2]' in bm.workload_name(65536, 1, 0)
    assert 'configs[3]' in bm.workload_name(32768, 8, 0) and '262144' in bm.workload_name(32768, 8, 0)
    assert 'configs[4]' in bm.workload_name(65536, 1, 1)
    assert 'custom' in bm.workload_name(12345, 1, 0) and 'configs' not in bm.workload_name(12345, 1, 0).split(':')[0].replace('custom batch', '')
    assert len(bm.source_hash()) == 16


def test_public_header_is_plain_c(tmp_path):
    """include/igtmpc.h is the drop-in boundary: it must compile as C99 (and C++) on its own -- plain pointers and sizes,
    no torch / HIP types in the signatures."""
    import shutil
    import subprocess
    if not shutil.which('gcc'):
        pytest.skip('no gcc')
    src = tmp_path / 'hdr.c'
    src.write_text('#include "igtmpc.h"\nint main(void) { igt_params p; return (int)sizeof(p) == 0; }\n')
    inc = os.path.join(ROOT, 'include')
    for cmd in (['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-fsyntax-only'], ['g++', '-std=c++11', '-x', 'c++', '-fsyntax-only']):
        r = subprocess.run(cmd + ['-I', inc, str(src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    hdr = open(os.path.join(inc, 'igtmpc.h')).read()
    assert re.findall(r'#include\s*[<"]([^>"]+)', hdr) == ['stdint.h']          # nothing but <stdint.h> is pulled in


def _build_c_caller(out):
    import shutil
    import subprocess
    if not shutil.which('gcc') or not os.path.exists('/opt/rocm/lib/libamdhip64.so'):
        pytest.skip('needs gcc and the ROCm runtime library')
    libdir = os.path.join(ROOT, 'igt-mpc-int_amd', 'igtmpc')
    cmd = ['gcc', '-std=c99', '-O2', '-Wall', '-Wextra', '-Werror', '-I', os.path.join(ROOT, 'include'),
           os.path.join(ROOT, 'examples', 'c_caller.c'), '-o', str(out), '-L', libdir, '-ligtmpc', f'-Wl,-rpath,{libdir}',
           '-L/opt/rocm/lib', '-lamdhip64', '-lm']
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_plain_c_program_links_against_the_abi(tmp_path):
    """examples/c_caller.c -- a C99 program with no Python and no torch -- compiles and links against libigtmpc.so
    through include/igtmpc.h alone (it is run on the GPU box by tests/test_gpu_api.py)."""
    exe = _build_c_caller(tmp_path / 'c_caller')
    assert os.path.getsize(exe) > 0


def test_design_cites_profiles_that_exist_and_belong_together():
    """DESIGN.md section 7 promises that every number comes from a file under profiles/ made at one kernel-source hash:
    every cited file exists, and every PMC summary of the round carries the hash DESIGN.md names."""
    import glob
    import warnings
    text = open(os.path.join(ROOT, 'DESIGN.md')).read() + open(os.path.join(ROOT, 'README.md')).read()
    cited = set(re.findall(r'`(profiles/[A-Za-z0-9_./-]+\.(?:json|csv|txt))`', text))
    assert len(cited) >= 15, cited
    missing = [c for c in cited if not os.path.exists(os.path.join(ROOT, c))]
    assert not missing, missing
    m = re.search(r'kernel-source hash `([0-9a-f]{16})`', text)
    assert m, 'DESIGN.md must name the kernel-source hash of its profiles'
    pm = glob.glob(os.path.join(ROOT, 'profiles', 'r04_pmc_*.json'))      # this round's counter summaries
    assert len(pm) >= 6
    for f in pm:
        assert json.load(open(f))['source_hash'] == m.group(1), f
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod2', os.path.join(ROOT, 'bench.py'))
    bm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bm)
    if bm.source_hash() != m.group(1):      # kernels changed since the profiles were collected: bench.py prints null PMC fields
        warnings.warn(f'profiles/ were collected at {m.group(1)}, the kernel sources are now {bm.source_hash()}: '
                      f're-run tools/collect.sh + tools/publish.py')


def test_value_net_from_config_reads_the_reference_yaml_shape(tmp_path):
    """mpc.py:72-74, 108-124 / evaluate.py:191: nn_config_dir is a sc{n}_config.yaml naming the checkpoint; the shipped
    weights of that scenario are returned, identity statistics are announced, a wrong architecture is refused."""
    import warnings
    from igtmpc.planner import value_net_from_config
    from igtmpc.value_nets import shipped_value_net
    d = tmp_path / 'configs'
    d.mkdir()
    for sc, nl in ((1, 2), (3, 3)):
        (d / f'sc{sc}_config.yaml').write_text(f'data_path: /game_theoretic_NN/dataset/processed_sc{sc}.pkl\n'
                                               f'model_path: /game_theoretic_NN/models/V_GT_sc{sc}.pt\nN: 10\n'
                                               f'include_route: False\nhidden_size: 128\nnum_layers: {nl}\ninput_size: 6\n')
        with pytest.warns(UserWarning, match='identity statistics'):
            net, include_route = value_net_from_config(str(d / f'sc{sc}_config.yaml'))
        want = shipped_value_net(sc)['layers']
        assert not include_route and len(net['layers']) == nl + 1
        assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(net['layers'], want))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        net, _ = value_net_from_config('/not/there/game_theoretic_NN/configs/sc6_config.yaml')      # by name
        assert len(net['layers']) == 4
        (d / 'sc2_config.yaml').write_text('model_path: /game_theoretic_NN/models/V_GT_sc2.pt\nnum_layers: 3\nhidden_size: 128\n')
        with pytest.raises(ValueError, match='num_layers'):
            value_net_from_config(str(d / 'sc2_config.yaml'))
        with pytest.raises(ValueError):
            value_net_from_config(str(tmp_path / 'whatever.yaml'))


def test_oracle_gt_loop_rules():
    """oracle/closed_loop.py gt mode (evaluate.py:171, 207-210, 232): (0, 0) initial inputs, warm starts counted from t = 2,
    and a first-step forecast that differs from the mpc branch's."""
    import closed_loop as CL
    import np_oracle as O
    from igtmpc import routes as R
    from igtmpc.cinf import cinf_halfplanes
    from igtmpc.evaluate import initial_states
    from igtmpc.value_nets import shipped_value_net
    pair = R.SCENARIO_ROUTES[0][0]
    x, _ = initial_states(np.random.default_rng(2026), [pair])
    x[0, :, 5] = 2.0
    net = dict(shipped_value_net(1), Wn=np.eye(6), mu_f=np.zeros(6), sigma_t=1.0, mu_t=0.0)
    P = O.Params(N=20)
    g = CL.run_episode(x[0], pair, P, cinf_halfplanes(), M_sim=4, cand_mode='track', eval_mode='gt_mpc', net=net)
    m = CL.run_episode(x[0], pair, P, cinf_halfplanes(), M_sim=4, cand_mode='track')
    assert g['events']['warm'] == 2 * 2 and m['events']['warm'] == 2 * 3          # t = 2, 3 vs t = 1, 2, 3
    assert np.abs(g['u_data'][0::2, 0]).max() <= 0.09 + 1e-12                      # |a_0 - 0| <= dt * jerk from u_prev = (0, 0)
    assert not np.array_equal(g['u_data'], m['u_data'])
    assert CL.scenario_encoding(pair) in ([1, -1], [-1, 1])


@pytest.mark.parametrize('case', ['lattice', 'table_all', 'no_obs_no_cinf', 'two_obs', 'empty'])
def test_c_oracle_is_clean_under_asan_and_ubsan(tmp_path, case):
    """SURVEY section 5 (sanitizers): oracle/igt_oracle.c built with -fsanitize=address,undefined (host; the pool has no
    GPU sanitizer) runs exactly-sized heap inputs without a report and returns, bit for bit, what the ordinary -O3 build
    returns -- lattice and table candidates, 0 / 1 / 2 obstacles, with and without C_inf, an empty batch."""
    import subprocess
    import c_oracle as CO
    import np_oracle as O
    from igtmpc.cinf import cinf_halfplanes
    from igtmpc.scenarios import make_batch
    here = os.path.join(ROOT, 'oracle')
    subprocess.run(['make', '-C', here, 'asan'], check=True, capture_output=True)
    B, N, C = (0 if case == 'empty' else 12), 20, 64
    n_obs = {'no_obs_no_cinf': 0, 'two_obs': 2}.get(case, 1)
    A, b = (np.zeros((0, 2)), np.zeros(0)) if case == 'no_obs_no_cinf' else cinf_halfplanes()
    bt = make_batch(max(B, 1), N=N, dtype=np.float64)
    x0, up, kp, fl = bt['x0'][:B], bt['u_prev'][:B], bt['kparams'][:B], bt['flags'][:B]
    obs = np.concatenate([bt['obs_xy'][:B]] * 2, axis=1)[:, :n_obs] if n_obs else np.zeros((B, 0, 2, N + 1))
    table = case == 'table_all'
    U = None
    if table:
        U = O.candidates_lattice(bt['u_prev'][:1], O.Params(N=N), C)[0] + 0.001
    hdr = np.array([B, N, 4, C, n_obs, len(b), int(table), int(table)], dtype=np.int32)
    with open(tmp_path / 'in.bin', 'wb') as f:
        f.write(hdr.tobytes())
        for a in (x0, up, kp, obs, A, b) + ((U,) if table else ()):
            f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
        f.write(np.ascontiguousarray(fl, dtype=np.uint32).tobytes())
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    r = subprocess.run([os.path.join(here, 'asan_driver'), str(tmp_path / 'in.bin'), str(tmp_path / 'out.bin')],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0 and 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr, r.stderr[-2000:]
    raw = open(tmp_path / 'out.bin', 'rb').read()
    P = O.Params(N=N)
    cinf = (None, None) if len(b) == 0 else (A, b)
    if table:
        ref = CO.rollout_all(x0, up, kp, fl, obs, *cinf, P, C=C, U=U, nthreads=2)
        want = b''.join(ref[k].tobytes() for k in ('X', 'U', 'cost', 'viol'))
    else:
        ref = CO.solve_batch(x0, up, kp, fl, obs, *cinf, P, C=C, nthreads=2) if B else None
        want = b''.join(ref[k].tobytes() for k in ('x', 'u', 'cost', 'argmin', 'status')) if B else b''
    assert raw == want


def test_c_abi_argument_validation_without_a_gpu():
    """The C ABI's argument checks that come before any device call (no GPU here): null pointers and invalid
    parameters return an IGT_E_* code with a message -- no crash, no exception across the boundary."""
    from igtmpc import _lib as L
    lib = L.load()
    assert lib.igt_params_default(None) != 0 and lib.igt_last_error()
    p = L.igt_params()
    assert lib.igt_params_default(ct.byref(p)) == 0
    h = ct.c_void_p()
    for field, bad in (('N', 0), ('C', 100), ('n_rk4', 0), ('n_obs', -1), ('dt', -0.1), ('cand_mode', 99)):
        q = L.igt_params()
        lib.igt_params_default(ct.byref(q))
        setattr(q, field, bad)
        assert lib.igt_create(ct.byref(q), 0, ct.byref(h)) != 0, field
        assert lib.igt_last_error()
    assert lib.igt_create(None, 0, ct.byref(h)) != 0
    assert lib.igt_create(ct.byref(p), 0, None) != 0
    for name in ('igt_destroy', 'igt_comm_destroy'):
        getattr(lib, name)(None)                                     # tolerated or refused, never a crash
    assert lib.igt_set_cinf(None, None, None, 0) != 0
    assert lib.igt_comm_init(None, 1, 0, None) != 0
    assert lib.igt_allgather_controls_f64(None, 4, None, None, None) != 0
    assert lib.igt_set_profiling(None, 1) != 0


def test_closed_loop_envelope_scale_follows_the_horizon():
    from igtmpc.evaluate import auto_track_env
    assert auto_track_env(40, 0.1) == 1.0 and auto_track_env(20, 0.1) == 0.5 and auto_track_env(10, 0.1) == 0.25
    assert auto_track_env(5, 0.1) == 0.25 and auto_track_env(80, 0.1) == 1.0


def test_shipped_library_is_built_without_the_developer_kernels(monkeypatch):
    """the 3-waves-per-SIMD builds, the oracle-order float64 kernels and the literal north_star mapping
    are compiled into libigtmpc_dev.so only; the shipped library refuses the IGT_DEV_FLAGS that select them (checked
    before any device call), and the Python face picks the developer library when such a flag is set."""
    from igtmpc import _lib as L
    shipped, dev = L.load(False), L.load(True)
    find = lambda path, needle: needle in open(path, 'rb').read()
    for sym in (b'search_f64_kernel_o3', b'search_fast_kernel_o3', b'search_literal_f64_kernel'):
        assert find(L.LIB_PATH_DEV, sym) and not find(L.LIB_PATH, sym), sym
    assert os.path.getsize(L.LIB_PATH) < os.path.getsize(L.LIB_PATH_DEV)
    p = L.igt_params()
    shipped.igt_params_default(ct.byref(p))
    h = ct.c_void_p()
    for flag in ('32', '1024', '2048', str(1024 | 4)):
        monkeypatch.setenv('IGT_DEV_FLAGS', flag)
        assert shipped.igt_create(ct.byref(p), 0, ct.byref(h)) == -1      # IGT_E_INVALID
        assert b'libigtmpc_dev.so' in shipped.igt_last_error()
        assert L.wants_dev_kernels() and L.load() is dev
    monkeypatch.setenv('IGT_DEV_FLAGS', '4')                       # switches of the production kernels stay available
    assert not L.wants_dev_kernels() and L.load() is shipped


def test_save_results_leaves_the_reference_drivers_run_directory(tmp_path):
    """igtmpc.evaluate.save_results (evaluate.py:319-369, 578-640): directory names, file names, array layouts and csv columns
    of the reference's driver -- cl_traj.pkl is [episodes, 7 M, T+1] so that generate_video.py:191 `cl_traj[index]` gets one
    episode's [7 M, T+1]; a second call on the same run directory appends episodes as the reference's iterations do."""
    import csv
    import pickle
    from igtmpc.evaluate import save_results
    rng = np.random.default_rng(0)
    E, M, T = 3, 2, 5
    r = dict(x_data=rng.normal(size=(E, 7 * M, T + 1)), u_data=rng.normal(size=(E, 2 * M, T)),
             infeasible_ratio=rng.uniform(size=(E, M)), deadlock=np.array([True, False, False]),
             routes=[('13', '24'), ('24', '31'), ('31', '42')], solve_ms=np.array([0.5, 0.25, 0.25, 0.25, 0.25]))
    base = str(tmp_path) + '/'
    for mode, sub, stats, cols in (('mpc', '/mpc', 'eval_stats.csv', ['avg_sol_times', 'std_solve_times', 'infeasible_ratio', 'deadlock']),
                                   ('gt_mpc', '/game_mpc/evaluation', 'stats.csv',
                                    ['NN_query_time', 'avg_sol_times', 'std_solve_times', 'infeasible_ratio', 'deadlock'])):
        run = save_results(r, base, mode, 3, policy={'type': 'MPC', 'N': 40}, timenow='20261004_000000')
        assert run == base + mode + '_sc3_seed2026_20261004_000000'          # evaluate.py:325: plain concatenation
        assert os.path.isdir(run + sub.replace('/evaluation', '') + '/evaluation_videos')
        save_results(r, base, mode, 3, policy={'type': 'MPC', 'N': 40}, timenow='20261004_000000')
        with open(run + sub + '/cl_traj.pkl', 'rb') as f:
            cl = pickle.load(f)
        with open(run + sub + '/u_cl.pkl', 'rb') as f:
            ucl = pickle.load(f)
        assert cl.shape == (2 * E, 7 * M, T + 1) and ucl.shape == (2 * E, 2 * M, T)
        assert np.array_equal(cl[:E], r['x_data']) and np.array_equal(cl[E:], r['x_data']) and np.array_equal(ucl[E:], r['u_data'])
        with open(run + sub + '/' + stats, newline='') as f:
            rows = list(csv.DictReader(f))
        assert list(rows[0].keys()) == cols and len(rows) == 2 * E
        assert [row['deadlock'] for row in rows[:E]] == ['True', 'False', 'False']
        assert rows[1]['infeasible_ratio'] == str(np.asarray(r['infeasible_ratio'][1]))      # numpy's text, as the reference's rows
        assert abs(float(rows[0]['avg_sol_times'].strip('[]').split()[0]) - 0.3e-3) < 1e-12   # seconds
        import yaml
        assert yaml.safe_load(open(run + sub + '/mpc.yaml'))['N'] == 40
        if mode == 'mpc':
            with open(run + sub + '/evaluation_data.pkl', 'rb') as f:
                d = pickle.load(f)
            assert d['x_cl'].shape == cl.shape and d['routes'].shape == (2 * E, M) and d['deadlock'].shape == (2 * E, 1)
            assert d['N'] == 40 and d['agent_types'] == ['CAV', 'CAV'] and d['initial_agents'].shape == (2 * E, M, 7)
    # a run directory with pickles this package did not write is refused, not unpickled
    foreign = tmp_path / 'foreign' / 'mpc_sc3_seed2026_t' / 'mpc'
    foreign.mkdir(parents=True)
    (foreign / 'cl_traj.pkl').write_bytes(b'cos\nsystem\n(S\'echo unpickled > ' + str(tmp_path / 'pwned').encode() + b'\'\ntR.')
    with pytest.raises(FileExistsError, match='does not'):
        save_results(r, str(tmp_path / 'foreign') + '/', 'mpc', 3, timenow='t')
    assert not (tmp_path / 'pwned').exists()


def test_load_reference_configs_reads_the_reference_files_and_refuses_another_intersection(tmp_path):
    """igtmpc.evaluate.load_reference_configs: mpc.yaml / fourwayint.yaml with the reference's keys (values as shipped,
    mpc.yaml:1-15, fourwayint.yaml:1-32, restated here as data) -> run_closed_loop keywords; prediction_type is matched the way
    evaluate.py:76-81 matches it; a road the frozen route tables were not generated for, a third agent, obca or an unknown
    prediction type are refused."""
    import yaml
    from igtmpc.evaluate import load_reference_configs
    pol = {'type': 'MPC', 'NN_type': 't+N', 'input_sequence_length': 5, 'N': 40, 'dt': 0.1, 'a_min': -4, 'a_max': 3,
           'v_min': -1.0, 'v_max': 5, 'prediction_type': 'constant_acceleration', 'collision_avoidance_type': 'circle'}
    env = {'render_fps': 10, 'dt': 0.1, 'road_width': 11.4, 'width_buffer': 0.2, 'road_length': 50, 'fillet_radius': 8.4,
           'ca_radius': 2.8, 'num_agents': 2, 'v0': 0, 'v_des': 5, 'width': 2.0, 'length': 4.47, 'l_r': 2.235, 'l_f': 2.235,
           'a_max': 3.0, 'a_min': -4.0, 'v_max': 20.0, 'v_min': -2.0, 'steering_min': -0.6, 'steering_max': 0.6}

    def files(p, e):
        pp, ee = tmp_path / 'mpc.yaml', tmp_path / 'fourwayint.yaml'
        pp.write_text(yaml.safe_dump(p))
        ee.write_text(yaml.safe_dump(e))
        return str(pp), str(ee)

    kw, policy = load_reference_configs(*files(pol, env))
    assert kw == {'constant_speed': False, 'N': 40, 'a_min_policy': -4.0, 'dt': 0.1, 'v0': 0.0,
                  'limits': {'l_r': 2.235, 'l_f': 2.235}}
    assert policy == pol
    assert load_reference_configs(*files(dict(pol, prediction_type='constant_speed', N=10, a_min=-2.5), env))[0] == {
        'constant_speed': True, 'N': 10, 'a_min_policy': -2.5, 'dt': 0.1, 'v0': 0.0, 'limits': {'l_r': 2.235, 'l_f': 2.235}}
    assert load_reference_configs(None, None) == ({}, {})
    for bad_p, bad_e in ((dict(pol, collision_avoidance_type='obca'), env), (dict(pol, prediction_type='kalman'), env),
                         (dict(pol, type='RL'), env), (pol, dict(env, road_width=12.0)), (pol, dict(env, num_agents=3)),
                         (pol, dict(env, ca_radius=3.0)), (pol, dict(env, v_max=10.0)), (dict(pol, dt=0.05), env)):
        with pytest.raises(ValueError):
            load_reference_configs(*files(bad_p, bad_e))
