"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI (ctypes), against the CPU oracle
on the same seeded inputs, against the committed golden fixtures, and -- at the benchmark's full size -- through
size-independent properties.

Tolerances
  f64 precision ....... decisions (collision flags, terminated, truncated, reward, reset samples) identical to the oracle;
                        positions / velocities within 1e-9 (same operations in the same order, contraction off; only libm's
                        log/sin/cos in the noise draws differ in the last bits)
  f32 precision ....... observations within 1e-4 of the oracle (north_star's tolerance; measured: ~1e-6), decisions identical
                        wherever the oracle's smallest decision margin exceeds 1e-5; the fraction of envs below that margin
                        is bounded
"""
import os

import numpy as np
import pytest

from conftest import load_json, load_npz, record_metric

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gpr():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import gymnasium_planar_robotics_amd as g
    g._native.load()  # fails loudly if libgpr_hip.so is missing
    return g


def _base_env(gpr, layout, M=1, cp=None, tile_hs=0.12, **kw):
    return gpr.VecBenchmarkPlanningEnv(np.asarray(layout), M, num_envs=kw.pop('num_envs', 1), collision_params=cp,
                                       tile_params={'size': np.array([tile_hs, tile_hs, 0.0176])}, precision='f64', **kw)


# ----------------------------------------------------------------------------------------------------------------------
# stand-alone operators: bit-exact against the golden fixtures (and therefore against the reference)
# ----------------------------------------------------------------------------------------------------------------------
def test_geometry_golden(gpr):
    G = gpr.geometry_2D_utils
    g = load_json('geometry_ref_tests.json')
    for c in g['segments']:
        assert np.array_equal(G.check_line_segments_intersect(c['p1'], c['p2'], c['q1'], c['q2']).cpu().numpy(), c['expected'])
    for c in g['rectangles']:
        out = G.check_rectangles_intersect(c['qpos_r1'], c['qpos_r2'], c['size_r1'], c['size_r2']).cpu().numpy()
        assert np.array_equal(out, c['expected'])
    z = load_npz('geometry_fuzz.npz')
    s = z['segs']
    assert np.array_equal(G.check_line_segments_intersect(s[:, 0:2], s[:, 2:4], s[:, 4:6], s[:, 6:8]).cpu().numpy(), z['seg_out'])
    assert np.array_equal(G.get_2D_rect_vertices(z['r1'], z['s1']).cpu().numpy(), z['verts_r1'])  # bit-exact vertices
    assert np.array_equal(G.check_rectangles_intersect(z['r1'], z['r2'], z['s1'], z['s2']).cpu().numpy(), z['rect_out'])
    # empty batch
    e = np.zeros((0, 2))
    assert G.check_line_segments_intersect(e, e, e, e).shape == (0,)


def test_qpos_is_valid_golden(gpr):
    cases = load_json('qpos_is_valid_ref_tests.json')
    for i, case in enumerate(cases):
        cp = case['collision_params']
        env = _base_env(gpr, case['layout'], 1, cp)
        c_size = case['c_size'] if case['c_size'] is not None else env.c_size
        out = env.qpos_is_valid(case['qpos'], c_size, case['add_safety_offset']).cpu().numpy()
        assert np.array_equal(out, case['expected']), f'reference known-answer case {i}'
        env.close()
    z = load_npz('qpos_is_valid_fuzz.npz')
    total = 0
    for i in range(z['out'].shape[0]):
        lay_p = z['layout'][i]
        tx = int((lay_p[:, 0] >= 0).sum()); ty = int((lay_p[0, :] >= 0).sum())
        box = bool(z['shape'][i])
        cs = z['c_size'][i] if box else z['c_size'][i][:, 0]
        cp = {'shape': 'box' if box else 'circle', 'size': cs[0] if box else float(cs[0]), 'offset': float(z['offset'][i]),
              'offset_wall': float(z['offset_wall'][i])}
        env = _base_env(gpr, lay_p[:tx, :ty], 1, cp, tile_hs=float(z['tile_half_size'][i]))
        out = env.qpos_is_valid(z['qpos'][i], cs, bool(z['add_safety_offset'][i])).cpu().numpy()
        assert np.array_equal(out, z['out'][i]), f'fuzz config {i}'
        total += out.size
        env.close()
    assert total > 20000


def test_mover_collision_golden(gpr):
    z = load_npz('mover_collision_fuzz.npz')
    envs = {}
    for i in range(z['res'].shape[0]):
        M = int(z['M'][i]); box = bool(z['shape'][i]); off = float(z['offset'][i])
        key = (M, box, off)
        if key not in envs:
            cp = {'shape': 'box', 'size': np.array([0.0775, 0.0775]), 'offset': off} if box else {'shape': 'circle', 'offset': off}
            envs[key] = _base_env(gpr, np.ones((10, 10)), M, cp)
        cs = z['c_size'][i][:M] if box else z['c_size'][i][:M, 0]
        res = bool(envs[key].check_mover_collision(z['qpos'][i][:M], cs, bool(z['add'][i])))
        assert res == bool(z['res'][i]), f'case {i}'


def test_clamp_and_reward_golden(gpr):
    env = _base_env(gpr, np.ones((9, 9)), 1)
    z = load_npz('ensure_max_dyn_val.npz')
    for mv in np.unique(z['maxv']):
        m = z['maxv'] == mv
        vals, ders = env.ensure_max_dyn_val(z['cur'][m], float(mv), z['der'][m])
        assert np.array_equal(vals.cpu().numpy(), z['vals'][m]) and np.array_equal(ders.cpu().numpy(), z['ders'][m])
    z = load_npz('reward_cases.npz')
    for M in (1, 2, 4, 8):
        e = _base_env(gpr, np.ones((9, 9)), M)
        info = {'mover_collision': z[f'M{M}_mc'], 'wall_collision': z[f'M{M}_wc']}
        assert np.array_equal(e.compute_reward(z[f'M{M}_ag'], z[f'M{M}_dg'], info).cpu().numpy(), z[f'M{M}_reward'])
        assert np.array_equal(e.compute_terminated(z[f'M{M}_ag'], z[f'M{M}_dg'], info).cpu().numpy(), z[f'M{M}_terminated'])
        assert not e.compute_truncated(torch.zeros(5, 2 * M), torch.zeros(5, 2 * M)).any()
        # the reference's own calling conventions (PL:666-688): an ARRAY of per-transition info dicts (what SB3's HER buffer passes),
        # a list of them, and ONE info dict of scalars (-> a scalar, the reference's reward[0])
        n = 300
        dicts = np.array([{'mover_collision': bool(z[f'M{M}_mc'][i]), 'wall_collision': bool(z[f'M{M}_wc'][i]), 'is_success': False}
                          for i in range(n)])
        for form in (dicts, list(dicts)):
            assert np.array_equal(e.compute_reward(z[f'M{M}_ag'][:n], z[f'M{M}_dg'][:n], form).cpu().numpy(), z[f'M{M}_reward'][:n])
            assert np.array_equal(e.compute_terminated(z[f'M{M}_ag'][:n], z[f'M{M}_dg'][:n], form).cpu().numpy(), z[f'M{M}_terminated'][:n])
        for i in range(5):
            r1 = e.compute_reward(z[f'M{M}_ag'][i], z[f'M{M}_dg'][i], dicts[i])
            assert r1.ndim == 0 and float(r1) == z[f'M{M}_reward'][i]
            assert bool(e.compute_terminated(z[f'M{M}_ag'][i], z[f'M{M}_dg'][i], dicts[i])) == bool(z[f'M{M}_terminated'][i])


def test_operators_large_fuzz_vs_oracle(gpr, oracle):
    """beyond the 22k reference-labelled cases: 1.2 M wall checks on random layouts with holes, a third of them snapped onto
    tile edges / onto "circle touches the edge" positions (the f64 operators must agree with the oracle bit for bit there
    too), plus 200k rectangle pairs and segment pairs around touching configurations"""
    rng = np.random.default_rng(2024)
    total = 0
    for it in range(12):
        tx, ty = int(rng.integers(1, 8)), int(rng.integers(1, 8))
        layout = (rng.random((tx, ty)) < 0.8).astype(np.int8)
        layout[rng.integers(tx), rng.integers(ty)] = 1
        box = it % 2 == 1
        hs = float(rng.choice([0.12, 0.1]))
        off, offw = float(rng.choice([0.0, 0.003])), float(rng.choice([0.0, 0.002]))
        n = 100_000
        qpos = np.zeros((n, 7)); qpos[:, 3] = 1.0
        qpos[:, 0] = rng.uniform(-0.05, 2 * hs * tx + 0.05, n); qpos[:, 1] = rng.uniform(-0.05, 2 * hs * ty + 0.05, n)
        if box:
            size = np.array([0.07, 0.05])
            yaw = rng.uniform(-np.pi, np.pi, n)
            yaw[: n // 4] = rng.choice([0.0, np.pi / 2, np.pi, -np.pi / 2], n // 4)
            qpos[:, 3] = np.cos(yaw / 2); qpos[:, 6] = np.sin(yaw / 2)
            cs, r = size, size.max()
        else:
            cs, r = 0.11 * hs / 0.12, 0.11 * hs / 0.12
        # snap a third of the positions to k * tile_width (+/- r, +/- safety offsets, +/- 1 ulp-ish)
        k = n // 3
        for d, t in ((0, tx), (1, ty)):
            edge = 2 * hs * rng.integers(0, t + 1, k)
            qpos[:k, d] = edge + rng.choice([0.0, r, -r, r + offw, -(r + offw), r + offw + off, -(r + offw + off)], k) + rng.choice([0.0, 1e-16, -1e-16, 1e-9, -1e-9], k)
        cp = {'shape': 'box' if box else 'circle', 'size': cs, 'offset': off, 'offset_wall': offw}
        env = _base_env(gpr, layout, 1, cp, tile_hs=hs)
        ol = oracle.OracleLayout(layout, (hs, hs), 'box' if box else 'circle', off, offw)
        for add in (False, True):
            got = env.qpos_is_valid(qpos, cs, add).cpu().numpy()
            want = ol.qpos_is_valid(qpos, cs, add)
            # the reference raises for a centre above no tile (BE:514); the engine reports "invalid" and the oracle -1
            assert np.array_equal(got != 0, want > 0), (it, add, int((got.astype(bool) != (want > 0)).sum()))
            total += n
        env.close()
    assert total >= 1_200_000
    G = gpr.geometry_2D_utils
    n = 200_000
    r1 = np.zeros((n, 7)); r2 = np.zeros((n, 7))
    for r in (r1, r2):
        yaw = rng.uniform(-np.pi, np.pi, n); yaw[: n // 4] = 0.0
        r[:, 3] = np.cos(yaw / 2); r[:, 6] = np.sin(yaw / 2)
    r1[:, :2] = rng.uniform(0, 1, (n, 2))
    s1 = rng.uniform(0.03, 0.12, (n, 2)); s2 = rng.uniform(0.03, 0.12, (n, 2))
    r2[:, :2] = r1[:, :2] + rng.uniform(-0.3, 0.3, (n, 2))
    r2[: n // 4, 0] = r1[: n // 4, 0] + s1[: n // 4, 0] + s2[: n // 4, 0]; r2[: n // 4, 1] = r1[: n // 4, 1]  # axis-aligned, exactly touching
    assert np.array_equal(G.check_rectangles_intersect(r1, r2, s1, s2).cpu().numpy(), oracle.check_rectangles_intersect(r1, r2, s1, s2))
    assert np.array_equal(G.get_2D_rect_vertices(r1, s1).cpu().numpy(), oracle.get_2D_rect_vertices(r1, s1))
    p1 = rng.uniform(0, 1, (n, 2)); p2 = p1 + rng.uniform(-0.3, 0.3, (n, 2)); q1 = rng.uniform(0, 1, (n, 2)); q2 = q1 + rng.uniform(-0.3, 0.3, (n, 2))
    q1[: n // 5] = p1[: n // 5] + (p2[: n // 5] - p1[: n // 5]) * rng.uniform(0, 1, (n // 5, 1))  # q1 on the segment p1p2
    q2[n // 5: n // 4] = p2[n // 5: n // 4]                                                      # shared end point
    assert np.array_equal(G.check_line_segments_intersect(p1, p2, q1, q2).cpu().numpy(), oracle.check_line_segments_intersect(p1, p2, q1, q2))


# ----------------------------------------------------------------------------------------------------------------------
# closed-form trajectories (what the reference's CI asserts MuJoCo equals)
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('precision,tol', [('f64', 1e-12), ('f32_f64acc', 1e-5), ('f32', 5e-5)])
def test_closed_form_trajectories(gpr, precision, tol):
    z = load_npz('trajectories.npz')
    for i in range(int(z['n'])):
        M = int(z[f'{i}_M']); mode = str(z[f'{i}_mode']); ncyc = int(z[f'{i}_num_cycles'])
        layout = np.ones((9, 9)) if str(z[f'{i}_env']) == 'planning' else np.ones((3, 3))
        env = gpr.VecBenchmarkPlanningEnv(layout, M, num_envs=1, std_noise=0.0, num_cycles=ncyc, v_max=0.01, a_max=0.2, j_max=150.0,
                                          learn_jerk=(mode == 'jerk'), max_episode_steps=None, autoreset_mode='disabled', precision=precision)
        env.reset(seed=0)
        env.pos[0] = torch.as_tensor(z[f'{i}_starts'].reshape(-1), dtype=env.dtype)
        env.goals[0] = 0.3
        act = torch.as_tensor(z[f'{i}_deriv'][None, :], dtype=env.dtype, device=env.device)
        pos, vel = [], []
        for _ in range(100):
            env.step(act)
            pos.append(env.pos[0].double().cpu().numpy()); vel.append(env.vel[0].double().cpu().numpy())
        assert np.abs(np.array(pos) - z[f'{i}_pos']).max() < tol, (i, precision)
        assert np.abs(np.array(vel) - z[f'{i}_vel']).max() < tol, (i, precision)
        env.close()


# ----------------------------------------------------------------------------------------------------------------------
# step / reset against the oracle
# ----------------------------------------------------------------------------------------------------------------------
HOLES = np.array([[1, 1, 1, 1, 1], [1, 1, 0, 1, 1], [1, 1, 1, 1, 1], [1, 0, 1, 1, 1], [1, 1, 1, 1, 0]])
CONFIGS = [
    # layout, M, kwargs
    (np.ones((3, 3)), 1, {}),
    (np.ones((6, 6)), 4, {}),
    (np.ones((6, 6)), 4, {'std_noise': 0.0}),
    (np.ones((5, 4)), 2, {'learn_jerk': True}),
    (HOLES, 3, {'collision_params': {'shape': 'circle', 'size': 0.1, 'offset': 0.005, 'offset_wall': 0.002}}),
    (np.ones((8, 8)), 8, {'std_noise': np.array([1e-5, 2e-5, 0.0])}),
    (np.ones((10, 10)), 16, {'num_cycles': 10}),
    (np.ones((6, 6)), 4, {'autoreset_mode': 'next_step'}),
    (np.ones((6, 6)), 5, {'autoreset_mode': 'disabled', 'max_episode_steps': 7}),
    # 'box' collision shape: rotated-rectangle wall check (vertices, missing-corner tests) and rectangle-rectangle mover check
    (np.ones((5, 5)), 3, {'collision_params': {'shape': 'box', 'size': np.array([0.0775, 0.0775]), 'offset': 0.002}}),
    (HOLES, 4, {'collision_params': {'shape': 'box', 'size': np.array([0.06, 0.09]), 'offset_wall': 0.001}, 'std_noise': 0.0}),
    (np.ones((4, 7)), 8, {'collision_params': {'shape': 'box', 'size': np.array([0.0775, 0.0775])}, 'learn_jerk': True, 'num_cycles': 20}),
]


def _make_pair(gpr, oracle, layout, M, E, precision, kw, env_id_offset=0):
    kw = dict(kw)
    autoreset = kw.pop('autoreset_mode', 'same_step')
    mes = kw.pop('max_episode_steps', 50)
    genv = gpr.VecBenchmarkPlanningEnv(layout, M, num_envs=E, precision=precision, autoreset_mode=autoreset, max_episode_steps=mes,
                                       env_id_offset=env_id_offset, **kw)
    okw = dict(kw)
    okw.setdefault('std_noise', 1e-5)
    oenv = oracle.OraclePlanningEnv(layout, M, E, autoreset=autoreset, max_episode_steps=mes, storage_f32=(precision != 'f64'),
                                    env_id_offset=env_id_offset, num_threads=min(16, os.cpu_count() or 1) if E >= 1024 else 1, **okw)
    return genv, oenv


def _g(t):
    return t.double().cpu().numpy() if t.dtype in (torch.float32, torch.float64) else t.cpu().numpy()


FLAG_KEYS = ('terminated', 'truncated', 'is_success', 'mover_collision', 'wall_collision')


def _compare(step, gout, oout, genv, oenv, tol, margin_min, stats):
    obs, rew, term, trunc, info = gout
    got = {'observation': _g(obs['observation']), 'achieved_goal': _g(obs['achieved_goal']), 'desired_goal': _g(obs['desired_goal']),
           'reward': _g(rew), 'terminated': _g(term), 'truncated': _g(trunc), 'is_success': _g(info['is_success']),
           'mover_collision': _g(info['mover_collision']), 'wall_collision': _g(info['wall_collision'])}
    safe = oout['margin'] > margin_min  # envs whose every decision had room
    stats['unsafe'] += int((~safe).sum()); stats['total'] += safe.size
    for k in FLAG_KEYS:
        bad = (got[k] != oout[k]) & safe
        assert not bad.any(), f'step {step}: {k} differs in envs {np.nonzero(bad)[0][:8]} (margins {oout["margin"][bad][:8]})'
    agree = np.ones(safe.size, dtype=bool)
    for k in FLAG_KEYS:
        agree &= got[k] == oout[k]
    stats['mismatch'] += int((~agree).sum())
    ok = safe & agree
    assert np.array_equal(got['reward'][ok], oout['reward'][ok]), f'step {step}: reward'
    M2 = 2 * genv.num_movers
    for k in ('observation', 'achieved_goal', 'desired_goal'):
        d = np.abs(got[k][ok] - oout[k][ok])
        stats['max_err'] = max(stats['max_err'], float(d.max()) if d.size else 0.0)
        assert (d < tol).all(), f'step {step}: {k} max err {d.max()}'
    for name, gt, ot in (('pos', genv.pos, oenv.pos), ('vel', genv.vel, oenv.vel), ('goal', genv.goals, oenv.goal)):
        d = np.abs(_g(gt)[ok] - ot[ok])
        assert (d < tol).all(), f'step {step}: state {name} max err {d.max()}'
    assert np.array_equal(_g(genv.elapsed_steps)[ok], oenv.elapsed[ok]), f'step {step}: elapsed'
    return ok


def _sync_from_oracle(genv, oenv):
    """teacher forcing: the next GPU step starts from the oracle's state, so one boundary flip cannot cascade"""
    dev, dt = genv.device, genv.dtype
    genv.pos.copy_(torch.as_tensor(oenv.pos, dtype=dt, device=dev)); genv.vel.copy_(torch.as_tensor(oenv.vel, dtype=dt, device=dev))
    genv.acc.copy_(torch.as_tensor(oenv.acc, dtype=dt, device=dev)); genv.goals.copy_(torch.as_tensor(oenv.goal, dtype=dt, device=dev))
    genv.elapsed_steps.copy_(torch.as_tensor(oenv.elapsed, device=dev))


@pytest.mark.parametrize('cfg_i', range(len(CONFIGS)))
@pytest.mark.parametrize('precision', ['f64', 'f32_f64acc', 'f32'])
def test_step_reset_vs_oracle(gpr, oracle, cfg_i, precision):
    layout, M, kw = CONFIGS[cfg_i]
    E = 300 if M <= 8 else 96  # not a multiple of the 64 / LPE envs a wavefront holds: exercises the ragged tail
    genv, oenv = _make_pair(gpr, oracle, layout, M, E, precision, kw, env_id_offset=1000)
    tol, margin_min = {'f64': (1e-9, 1e-12), 'f32_f64acc': (2e-6, 2e-7), 'f32': (1e-4, 1e-5)}[precision]
    stats = {'unsafe': 0, 'total': 0, 'max_err': 0.0, 'mismatch': 0}
    g0 = genv.reset(seed=1234 + cfg_i)
    o0 = oenv.reset(seed=1234 + cfg_i)
    zeros = torch.zeros(E, dtype=genv.dtype, device=genv.device)
    _compare(-1, (g0[0], zeros, zeros.bool(), zeros.bool(), g0[1]), o0, genv, oenv, tol, margin_min, stats)
    rng = np.random.default_rng(cfg_i)
    lim = genv.j_max if genv.learn_jerk else genv.a_max
    n_steps = 60
    n_term = 0
    for step in range(n_steps):
        a = rng.uniform(-1.2 * lim, 1.2 * lim, size=(E, 2 * M)).astype(np.float32).astype(np.float64)  # also exercises the clip
        if step % 7 == 3:
            a[: E // 3] *= 0.05  # some gentle actions so that not every episode ends in a crash
        _sync_from_oracle(genv, oenv)
        gout = genv.step(torch.as_tensor(a, dtype=genv.dtype, device=genv.device))
        oout = oenv.step(a)
        _compare(step, gout, oout, genv, oenv, tol, margin_min, stats)
        n_term += int(oout['terminated'].sum())
        if kw.get('autoreset_mode') == 'disabled' and step % 10 == 9:
            mask = (oout['terminated'] | oout['truncated'])
            genv.reset(options={'mask': torch.as_tensor(mask)}); oenv.reset(mask=mask)
    assert n_term > 0, 'the scenario never terminated an episode: nothing was tested'
    # every flag disagreement was already shown to sit on a decision with less than margin_min of room (_compare); here:
    # such disagreements are rare, and impossible in f64
    frac_mismatch = stats['mismatch'] / stats['total']
    assert frac_mismatch <= {'f64': 0.0, 'f32_f64acc': 2e-4, 'f32': 3e-3}[precision], frac_mismatch
    print(f'cfg {cfg_i} {precision}: max obs err {stats["max_err"]:.3e}, env-steps with a flag mismatch {frac_mismatch:.2e} '
          f'(below margin: {stats["unsafe"] / stats["total"]:.2e})')
    record_metric(f'step_reset_vs_oracle[cfg{cfg_i}-{precision}]', max_obs_err=stats['max_err'], tol=tol, frac_flag_mismatch=frac_mismatch,
                  frac_below_margin=stats['unsafe'] / stats['total'], env_steps=stats['total'])
    genv.close()


FREE_RUNNING_F64 = [
    (np.ones((6, 6)), 4, {}),
    (HOLES, 3, {'collision_params': {'shape': 'circle', 'size': 0.1, 'offset': 0.005, 'offset_wall': 0.002}}),  # table-driven wall test
    (np.ones((5, 5)), 3, {'collision_params': {'shape': 'box', 'size': np.array([0.0775, 0.0775]), 'offset': 0.002}}),  # rectangle tests
    (np.ones((6, 6)), 4, {'learn_jerk': True}),                                                         # two literal clamps per cycle
    (np.ones((8, 8)), 8, {'std_noise': np.array([1e-4, 2e-4, 0.0])}),                                    # wide noise bands, 8 lanes per env
    (np.ones((10, 10)), 16, {'E': 2048, 'steps': 60}),   # BASELINE config 5's shape: parallel reset trials (~45 rejections per reset), 16 lanes per env
    (np.ones((8, 8)), 8, {'E': 4096, 'steps': 60}),      # BASELINE config 4's shape
    # the shared searches of the 16-lane kernels (a block's wavefronts take rounds of four trials from one counter per search): a trial bound
    # that is not a multiple of a round and that most searches run into (the last candidate stands: PL:369 with a bound), and a padded
    # group (12 movers in 16 lanes) on a layout so tight that a search takes hundreds of rounds
    (np.ones((10, 10)), 16, {'E': 1024, 'steps': 40, 'max_reset_trials': 6}),
    (np.ones((7, 7)), 12, {'E': 1024, 'steps': 40}),
]


@pytest.mark.parametrize('cfg_i', range(len(FREE_RUNNING_F64)))
def test_free_running_f64_matches_oracle_exactly(gpr, oracle, cfg_i):
    """no teacher forcing: 120 steps, every episode boundary, every auto-reset sample must agree with the oracle"""
    layout, M, kw = FREE_RUNNING_F64[cfg_i]
    kw = dict(kw)
    E, n_steps = kw.pop('E', 256), kw.pop('steps', 120)
    genv, oenv = _make_pair(gpr, oracle, layout, M, E, 'f64', kw)
    genv.reset(seed=7); oenv.reset(seed=7)
    rng = np.random.default_rng(3)
    lim = genv.j_max if genv.learn_jerk else genv.a_max
    n_done = 0
    for step in range(n_steps):
        a = rng.uniform(-lim, lim, size=(E, 2 * M))
        obs, rew, term, trunc, info = genv.step(torch.as_tensor(a, device=genv.device))
        o = oenv.step(a)
        assert np.array_equal(_g(term), o['terminated']) and np.array_equal(_g(trunc), o['truncated']), step
        assert np.array_equal(_g(rew), o['reward']), step
        assert np.abs(_g(obs['achieved_goal']) - o['achieved_goal']).max() < 1e-9, step
        assert np.abs(_g(obs['desired_goal']) - o['desired_goal']).max() < 1e-12, step
        done = o['terminated'] | o['truncated']
        fo = info['final_obs']
        assert np.abs(_g(fo['achieved_goal'])[done] - o['final_achieved_goal'][done]).max(initial=0) < 1e-9
        assert np.abs(_g(fo['observation'])[done] - o['final_observation'][done]).max(initial=0) < 1e-9
        n_done += int(done.sum())
    record_metric(f'free_running_f64[{cfg_i}: {M} movers x {E} envs]', env_steps=E * n_steps, auto_resets_compared=n_done, mismatches=0)
    assert n_done > E // 2
    genv.close()


@pytest.mark.parametrize('M,n,E', [(4, 6, 4096), (8, 8, 4096), (16, 10, 2048), (4, 'box', 1024), (3, 'holes', 1024)])
def test_free_running_f32_follows_the_oracle_within_1e_4(gpr, oracle, M, n, E):
    """The north-star statement, in the benchmark's precision and without teacher forcing: same seed, same actions, 100 steps
    (two full episodes, every auto-reset in between); every env is followed until the first env-step in which the float engine
    and the fp64 oracle decide differently -- which may only happen where the oracle itself had less than 1e-5 of room --
    and up to there every observation agrees within 1e-4."""
    layout, kw = (np.ones((n, n)), {}) if isinstance(n, int) else \
        (np.ones((6, 6)), {'collision_params': {'shape': 'box', 'size': np.array([0.0775, 0.0775]), 'offset': 0.002}}) if n == 'box' else \
        (HOLES, {'collision_params': {'shape': 'circle', 'size': 0.1, 'offset': 0.005, 'offset_wall': 0.002}})
    genv, oenv = _make_pair(gpr, oracle, layout, M, E, 'f32', kw)
    genv.reset(seed=21); oenv.reset(seed=21)
    rng = np.random.default_rng(8)
    alive = np.ones(E, dtype=bool)
    worst, n_steps, n_resets = 0.0, 0, 0
    for step in range(100):
        a = rng.uniform(-10, 10, size=(E, 2 * M)).astype(np.float32).astype(np.float64)
        obs, rew, term, trunc, info = genv.step(torch.as_tensor(a, dtype=genv.dtype, device=genv.device))
        o = oenv.step(a)
        same = np.ones(E, dtype=bool)
        for k, gv in (('terminated', term), ('truncated', trunc), ('is_success', info['is_success']),
                      ('mover_collision', info['mover_collision']), ('wall_collision', info['wall_collision'])):
            same &= _g(gv) == o[k]
        err = np.zeros(E)
        for k in ('observation', 'achieved_goal', 'desired_goal'):
            err = np.maximum(err, np.abs(_g(obs[k]) - o[k]).max(axis=1))
        same &= (err < 1e-4) & (_g(rew) == o['reward'])
        split = alive & ~same
        # leaving the oracle's history (another decision, or the same collision found one cycle apart) needs a decision on the edge
        assert (o['margin'][split] < 1e-5).all(), (step, o['margin'][split], err[split])
        alive &= same
        worst = max(worst, float(err[alive].max(initial=0.0)))
        n_steps += int(alive.sum()); n_resets += int((o['terminated'] | o['truncated'])[alive].sum())
    assert alive.mean() > 0.99 and n_resets > E and worst < 1e-5, (alive.mean(), n_resets, worst)  # (measured: nobody leaves, worst 4.3e-6)
    record_metric(f'free_running_f32[M{M}{"" if isinstance(n, int) else "-" + n}]', frac_envs_following_the_oracle=float(alive.mean()), max_obs_err=worst, tol=1e-4,
                  env_steps_followed=n_steps, auto_resets_followed=n_resets)
    genv.close()


def test_sharding_invariance_and_determinism(gpr):
    """envs [0,E) as one handle == two handles with env_id_offset 0 and E/2 (no cross-env dependence, streams keyed by
    global env id); and the same seed reproduces the same rollout bit for bit"""
    layout, M, E = np.ones((6, 6)), 4, 512
    mk = lambda n, off: gpr.VecBenchmarkPlanningEnv(layout, M, num_envs=n, env_id_offset=off, precision='f32')  # noqa: E731
    whole, lo, hi, again = mk(E, 0), mk(E // 2, 0), mk(E // 2, E // 2), mk(E, 0)
    for e in (whole, lo, hi, again):
        e.reset(seed=99)
    g = torch.Generator(device='cuda').manual_seed(0)
    for step in range(30):
        a = (torch.rand(E, 2 * M, device='cuda', generator=g) * 20 - 10)
        ow = whole.step(a); oa = again.step(a)
        ol = lo.step(a[: E // 2]); oh = hi.step(a[E // 2:])
        for k in ('observation', 'achieved_goal', 'desired_goal'):
            assert torch.equal(ow[0][k], oa[0][k])
            assert torch.equal(ow[0][k], torch.cat([ol[0][k], oh[0][k]]))
        assert torch.equal(ow[1], torch.cat([ol[1], oh[1]])) and torch.equal(ow[2], torch.cat([ol[2], oh[2]]))
        assert torch.equal(ow[3], torch.cat([ol[3], oh[3]]))


def test_free_box_blocks_match_cycle_by_cycle(gpr, monkeypatch):
    """Layouts with holes: the block loop tests the wall against a per-lane free box (a rectangle of existing tiles around the start
    tile) and replays flagged blocks through the table-driven test; GPR_EAGER_NOISE=1 runs that test in every cycle. A layout of
    diagonal holes, an L-shaped corridor and a missing corner, circles just under half a tile: both must agree bit for bit, and
    wall collisions next to the holes must actually occur."""
    layout = np.array([[1, 1, 1, 1, 1, 0], [1, 0, 1, 1, 1, 1], [1, 1, 0, 1, 1, 1], [1, 1, 1, 1, 0, 1], [0, 1, 1, 1, 1, 1], [1, 1, 1, 0, 1, 1]])
    for M, size in ((4, 0.11), (2, 0.1195), (3, 0.06)):
        E = 16384
        kw = dict(num_envs=E, std_noise=np.array([2e-5, 1e-5, 0.0]), collision_params={'shape': 'circle', 'size': size, 'offset': 0.0, 'offset_wall': 0.0})
        monkeypatch.delenv('GPR_EAGER_NOISE', raising=False)
        blocks = gpr.VecBenchmarkPlanningEnv(layout, M, **kw)
        monkeypatch.setenv('GPR_EAGER_NOISE', '1')
        cycles = gpr.VecBenchmarkPlanningEnv(layout, M, **kw)
        monkeypatch.delenv('GPR_EAGER_NOISE', raising=False)
        blocks.reset(seed=5); cycles.reset(seed=5)
        assert torch.equal(blocks.pos, cycles.pos)
        g = torch.Generator(device='cuda').manual_seed(9)
        n_wall = 0
        for step in range(40):
            a = (torch.rand(E, 2 * M, device='cuda', generator=g) * 2 - 1) * blocks.a_max
            ob = blocks.step(a); oc = cycles.step(a)
            for k in ('observation', 'achieved_goal', 'desired_goal'):
                assert torch.equal(ob[0][k], oc[0][k]), (M, step, k)
            assert torch.equal(ob[1], oc[1]) and torch.equal(ob[2], oc[2]) and torch.equal(ob[3], oc[3])
            assert torch.equal(ob[4]['wall_collision'], oc[4]['wall_collision']) and torch.equal(ob[4]['mover_collision'], oc[4]['mover_collision'])
            assert torch.equal(blocks.pos, cycles.pos) and torch.equal(blocks.vel, cycles.vel)
            n_wall += int(ob[4]['wall_collision'].sum())
        assert n_wall > 1000, n_wall
        record_metric(f'free_box_blocks[M{M}-r{size}]', env_steps=40 * E, wall_collisions=n_wall, mismatches=0)
        blocks.close(); cycles.close()


def test_lazy_noise_is_exact(gpr, monkeypatch):
    """The kernels skip a noise draw when no possible value of it could change the outcome of the comparison it feeds.
    GPR_EAGER_NOISE=1 (read at env creation) draws everything: both must produce bit-identical rollouts."""
    box = {'shape': 'box', 'size': np.array([0.0775, 0.0775]), 'offset': 0.001}
    for layout, M, prec, cp, ncyc in ((np.ones((6, 6)), 4, 'f32', None, 40), (HOLES, 3, 'f32', None, 40), (np.ones((5, 5)), 2, 'f64', None, 40),
                                      (np.ones((4, 6)), 2, 'f32', None, 40), (HOLES, 4, 'f32', box, 40), (np.ones((4, 4)), 3, 'f64', box, 40),
                                      (np.ones((6, 6)), 4, 'f32', box, 40),     # box shape on a grid without holes: fast wall / pair pre-tests
                                      (np.ones((6, 6)), 4, 'f32', None, 100),   # more cycles than an LDS row of readings holds
                                      (np.ones((8, 8)), 1, 'f32', None, 40),    # one mover: no pair check at all
                                      (np.ones((10, 10)), 16, 'f32', None, 7),  # 16 lanes per env, odd cycle count
                                      # cycle counts that are no multiple of a block (8 cycles, 10 in thin launches): the tail goes cycle by cycle;
                                      # a negative count = the same with whole wavefronts (set_wave_budget(0)): the other block length
                                      (np.ones((6, 6)), 4, 'f32', None, 37), (np.ones((6, 6)), 4, 'f32', None, -37),
                                      (np.ones((8, 8)), 8, 'f32', None, 29), (np.ones((8, 8)), 8, 'f32', None, -29)):
        whole, ncyc = ncyc < 0, abs(ncyc)
        E = 2048 if prec == 'f32' else 256
        kw = dict(num_envs=E, precision=prec, std_noise=np.array([2e-5, 1e-5, 0.0]), learn_jerk=(prec == 'f64'), collision_params=cp,
                  num_cycles=ncyc)
        monkeypatch.delenv('GPR_EAGER_NOISE', raising=False)
        lazy = gpr.VecBenchmarkPlanningEnv(layout, M, **kw)
        monkeypatch.setenv('GPR_EAGER_NOISE', '1')
        eager = gpr.VecBenchmarkPlanningEnv(layout, M, **kw)
        monkeypatch.delenv('GPR_EAGER_NOISE', raising=False)
        if whole:
            lazy.set_wave_budget(0); eager.set_wave_budget(0)
        lazy.reset(seed=11); eager.reset(seed=11)
        g = torch.Generator(device='cuda').manual_seed(5)
        lim = lazy.j_max if lazy.learn_jerk else lazy.a_max
        n_done = 0
        for step in range(80):
            a = ((torch.rand(E, 2 * M, device='cuda', generator=g) * 2 - 1) * lim).to(lazy.dtype)
            if step % 3 == 0:
                a = a.sign() * lim  # saturate: drives movers into the speed clamp, where the velocity noise matters
            ol = lazy.step(a); oe = eager.step(a)
            for k in ('observation', 'achieved_goal', 'desired_goal'):
                assert torch.equal(ol[0][k], oe[0][k]), (step, k)
            assert torch.equal(ol[1], oe[1]) and torch.equal(ol[2], oe[2]) and torch.equal(ol[3], oe[3])
            assert torch.equal(lazy.pos, eager.pos) and torch.equal(lazy.vel, eager.vel)
            n_done += int(ol[2].sum())
        assert n_done > 0
        lazy.close(); eager.close()


@pytest.mark.parametrize('M,precision,kw', [(1, 'f32', {}), (3, 'f64', {}), (4, 'f32', {}), (8, 'f32', {}), (16, 'f32', {}), (7, 'f64', {}), (15, 'f64', {}),
                                            (16, 'f64', {}), (4, 'f32', {'autoreset_mode': 'next_step'}), (5, 'f32_f64acc', {'autoreset_mode': 'disabled'}),
                                            (4, 'f32', {'learn_jerk': True})])
def test_parallel_env_agent_views(gpr, M, precision, kw):
    """PettingZoo-parallel layer: the per-agent views -- written by the step kernel's own epilogue (one launch per step()) -- against a
    plain NumPy statement of their documented layout (engine-defined, the reference has no multi-agent semantics), against the
    stand-alone gather kernel over the stored outputs (bit for bit), and consistency with the single-agent outputs."""
    E = 200
    n = max(6, int(np.ceil(np.sqrt(M))) * 3)
    penv = gpr.VecBenchmarkPlanningParallelEnv(np.ones((n, n)), M, num_envs=E, precision=precision, **kw)
    same_step = kw.get('autoreset_mode', 'same_step') == 'same_step'
    lim = penv.env.j_max if penv.env.learn_jerk else penv.env.a_max
    assert penv.agents == [f'mover_{i}' for i in range(M)] and penv.observation_space('mover_0').shape == (4 * M + 2,)
    obs, infos = penv.reset(seed=2)
    g = torch.Generator(device='cuda').manual_seed(0)
    n_trunc = 0
    for step in range(60):  # past the TimeLimit (50): truncated envs are auto-reset in the same call
        # (the first half of the envs never accelerates: those envs live to the TimeLimit)
        still = (torch.arange(E, device='cuda') < E // 2).unsqueeze(1)
        act = {a: ((torch.rand(E, 2, device='cuda', generator=g) * 2 - 1) * lim).masked_fill(still, 0.0) for a in penv.agents}
        obs, rew, term, trunc, infos = penv.step(act)
        e = penv.env
        if True:  # the stand-alone gather over the stored outputs writes the same bytes
            fused_o, fused_r = penv._agent_obs.clone(), penv._agent_reward.clone()
            penv._agent_obs.fill_(-7.0); penv._agent_reward.fill_(-7.0)
            penv._views()
            assert torch.equal(fused_o, penv._agent_obs) and torch.equal(fused_r, penv._agent_reward), step
        v = e._obs['observation'].double().cpu().numpy().reshape(E, -1)[:, :2 * M].reshape(E, M, 2); p = e._obs['achieved_goal'].double().cpu().numpy().reshape(E, M, 2)
        gl = e._obs['desired_goal'].double().cpu().numpy().reshape(E, M, 2); r = e._reward.double().cpu().numpy()
        # the reward of an env that ended in this call belongs to its TERMINAL state (info['final_obs']), not to the first
        # observation of the new episode that same-step autoreset returns
        done = (e._terminated | e._truncated).bool().cpu().numpy() & same_step
        n_trunc += int(e._truncated.sum())
        fp = e._final['achieved_goal'].double().cpu().numpy().reshape(E, M, 2); fg = e._final['desired_goal'].double().cpu().numpy().reshape(E, M, 2)
        rp = np.where(done[:, None, None], fp, p); rg = np.where(done[:, None, None], fg, gl)
        if not same_step and kw.get('autoreset_mode') == 'disabled' and step == 55:
            penv.reset(seed=step)  # (nobody resets a finished env in this mode)
            continue
        tot = np.zeros(E)
        for i, a in enumerate(penv.agents):
            exp = [v[:, i], p[:, i], gl[:, i]]
            for j in range(M):
                if j != i:
                    exp += [p[:, j] - p[:, i], v[:, j] - v[:, i]]
            exp = np.concatenate(exp, axis=1)
            got = obs[a].double().cpu().numpy()
            assert np.abs(got - exp).max() < (1e-12 if precision == 'f64' else 1e-6), (step, a)
            shared = (r == 50) | (r == -50)
            own = np.where(np.linalg.norm(rp[:, i] - rg[:, i], axis=1) <= 0.1, 0.0, -1.0)
            if kw.get('autoreset_mode') == 'next_step':  # a call that only reset the env (step count 0 afterwards): reward 0 for every agent
                own = np.where(e.elapsed_steps.cpu().numpy() == 0, 0.0, own)
            assert np.array_equal(rew[a].double().cpu().numpy(), np.where(shared, r, own))
            tot += np.where(shared, 0.0, own)
            assert term[a] is term['mover_0'] and torch.equal(term[a], e._terminated.view(torch.bool))
        assert np.array_equal(tot[~shared], r[~shared])  # the per-agent rewards add up to the env reward (PL:530), truncated envs included
    assert n_trunc > 0
    assert penv.state().shape[0] == E
    penv.close()


def test_step_n_writes_agent_views_per_step(gpr):
    """gpr_step_n with agent_observation_dev / agent_reward_dev bound: row i holds what the i-th step() + gather would have returned"""
    E, M, T = 1500, 4, 55
    lay = np.ones((6, 6))
    a = gpr.VecBenchmarkPlanningParallelEnv(lay, M, num_envs=E); b = gpr.VecBenchmarkPlanningEnv(lay, M, num_envs=E)
    a.reset(seed=3); b.reset(seed=3)
    g = torch.Generator(device='cuda').manual_seed(4)
    acts = torch.rand(T, E, 2 * M, device='cuda', generator=g) * 20 - 10
    roll = b.step_n(acts, b.alloc_rollout(T, agent_views=True))
    for t in range(T):
        a.step(acts[t])
        assert torch.equal(a._agent_obs, roll['agent_observation'][t]) and torch.equal(a._agent_reward, roll['agent_reward'][t]), t
    a.close(); b.close()


# ----------------------------------------------------------------------------------------------------------------------
# full benchmark size: size-independent properties
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,layout_n,E', [(4, 6, 65536), (8, 8, 8192), (16, 10, 4096)])
def test_full_size_properties(gpr, M, layout_n, E):
    env = gpr.VecBenchmarkPlanningEnv(np.ones((layout_n, layout_n)), M, num_envs=E, precision='f32')
    obs, info = env.reset(seed=5)
    r, mgd = 0.11, env.min_goal_dist
    hi = layout_n * 0.24

    def check_fresh(mask):
        """freshly reset envs: starts collision free with the reset's margins, goals valid and spaced, zero velocity"""
        p = env.pos[mask].reshape(-1, M, 2); gl = env.goals[mask].reshape(-1, M, 2)
        assert (env.vel[mask] == 0).all() and (env.elapsed_steps[mask] == 0).all()
        for t, mind in ((p, 2 * r), (gl, mgd)):
            assert (t >= r - 1e-6).all() and (t <= hi - r + 1e-6).all()
            d = (t[:, :, None, :] - t[:, None, :, :]).norm(dim=-1) + torch.eye(M, device=t.device) * 10
            assert d.min() >= mind - 1e-5

    check_fresh(torch.ones(E, dtype=torch.bool, device=env.device))
    # a start sampled within ~sigma of the safety margin can read as a collision through the noisy check of BE:1799-1805
    assert info['wall_collision'].float().mean() < 1e-3 and info['mover_collision'].float().mean() < 1e-3
    g = torch.Generator(device='cuda').manual_seed(1)
    n_done = 0
    for step in range(55):
        a = torch.rand(E, 2 * M, device='cuda', generator=g) * 20 - 10
        obs, rew, term, trunc, info = env.step(a)
        coll = info['wall_collision'] | info['mover_collision']
        # reward / terminated / info algebra (PL:478, PL:519-531, PL:597)
        assert ((rew == -50) == coll).all()
        assert ((rew == 50) == info['is_success']).all()
        assert (term == (coll | info['is_success'])).all()
        assert ((rew <= 0) | (rew == 50)).all() and (rew[~term] >= -M).all() and (rew[~term] == rew[~term].round()).all()
        done = term | trunc
        n_done += int(done.sum())
        check_fresh(done)  # same-step autoreset
        fo = info['final_obs']
        # terminal observations of crashed envs really are in collision (up to the noise), others are not
        ag = fo['achieved_goal'][coll].reshape(-1, M, 2)
        if ag.numel():
            wall = ((ag < r + 2e-4) | (ag > hi - r - 2e-4)).any(dim=-1).any(dim=-1)
            d = (ag[:, :, None, :] - ag[:, None, :, :]).norm(dim=-1) + torch.eye(M, device=ag.device) * 10
            assert (wall | (d.min(dim=-1).values.min(dim=-1).values <= 2 * r + 2e-4)).all()
        # speed limit (PL:631-645) and positions on the grid
        assert (env.vel.reshape(E, M, 2).norm(dim=-1) <= env.v_max + 1e-4).all()  # the clamp acts on the noisy reading (PL:430,442)
        live = ~done
        pl = env.pos[live]
        assert (pl > r - 1e-4).all() and (pl < hi - r + 1e-4).all()
        assert (trunc <= done).all()
    assert n_done > E // 4
    # observation noise: achieved_goal - pos ~ N(0, 1e-5)
    d = (obs['achieved_goal'] - env.pos).flatten().double()
    assert abs(float(d.mean())) < 3e-7 and 0.9e-5 < float(d.std()) < 1.1e-5
    assert float((d.abs() > 3e-5).double().mean()) == pytest.approx(0.0027, abs=0.0015)


# ----------------------------------------------------------------------------------------------------------------------
# engine additions behind the C ABI: resumable random streams, episode counters, large layouts
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('precision', ['f32', 'f64'])
def test_checkpoint_restores_bit_exactly(gpr, precision):
    """state_dict() = the state tensors + gpr_get_rng_state(); a fresh handle that loads it continues the run bit for bit
    (noise draws, auto-reset samples and all)."""
    E, M = 2048, 4
    mk = lambda: gpr.VecBenchmarkPlanningEnv(np.ones((6, 6)), M, num_envs=E, precision=precision)  # noqa: E731
    a = mk()
    a.reset(seed=11)
    g = torch.Generator(device='cuda').manual_seed(3)
    acts = [(torch.rand(E, 2 * M, device='cuda', generator=g) * 20 - 10).to(a.dtype) for _ in range(40)]
    for t in range(20):
        a.step(acts[t])
    sd = a.state_dict()
    assert sd['rng_tick'] == 21 and sd['rng_seed'] == 11  # one reset + 20 steps
    import io
    buf = io.BytesIO(); torch.save(sd, buf); buf.seek(0)
    b = mk()
    b.load_state_dict(torch.load(buf))
    for t in range(20, 40):
        oa, ra, ta, ua, ia = a.step(acts[t])
        ob, rb, tb, ub, ib = b.step(acts[t])
        for k in oa:
            assert torch.equal(oa[k], ob[k]), (t, k)
        assert torch.equal(ra, rb) and torch.equal(ta, tb) and torch.equal(ua, ub)
        assert torch.equal(a.pos, b.pos) and torch.equal(a.goals, b.goals) and torch.equal(a.elapsed_steps, b.elapsed_steps)
    assert int((ta | ua).sum()) >= 0 and a.get_rng_state() == b.get_rng_state() == (11, 41)
    a.close(); b.close()


@pytest.mark.parametrize('M,n,E', [(4, 6, 5000), (16, 10, 5000), (4, 6, 2705), (1, 3, 10900)])
def test_episode_counters_match_the_flags(gpr, M, n, E):
    """the on-device counters (per-wavefront popcounts, reduced by gpr_read_counters) equal the sums of the returned flags.
    2 705 x 4 lanes and 10 900 x 1 lane: 170.06 wavefronts hold envs = 4 080 bytes of rows, but the grid's last block brings 172
    wavefronts, whose rows end behind the 4 KB line -- the table is sized from the grid."""
    env = gpr.VecBenchmarkPlanningEnv(np.ones((n, n)), M, num_envs=E)
    env.reset(seed=4)
    env.enable_episode_counters()
    g = torch.Generator(device='cuda').manual_seed(9)
    want = np.zeros(6, dtype=np.int64)
    for t in range(60):
        a = torch.rand(E, 2 * M, device='cuda', generator=g) * 20 - 10
        a[:E // 5] = 0.0  # (envs that never accelerate live to the TimeLimit)
        obs, r, term, trunc, info = env.step(a)
        want += np.array([int((term | trunc).sum()), int(term.sum()), int(trunc.sum()), int(info['is_success'].sum()),
                          int(info['wall_collision'].sum()), int(info['mover_collision'].sum())])
        if t == 30:
            assert np.array_equal(env.read_episode_counters().cpu().numpy()[:6], want)  # totals accumulate across reads
    tot = env.read_episode_counters().cpu().numpy()
    assert np.array_equal(tot[:6], want) and tot[6] == 0 and tot[7] == 0  # (no search exhausted; clipped actions are not being counted)
    assert want[0] > E and want[2] > 0 and want[4] > 0
    assert np.array_equal(env.all_reduce_episode_counters().cpu().numpy()[:6], want)  # single process: the same numbers
    env.close()


@pytest.mark.parametrize('M,n,E,kw', [(4, 6, 1000, {}), (3, 5, 777, {'precision': 'f64'}), (8, 8, 900, {}), (16, 10, 300, {}), (1, 3, 2000, {}),
                                      (4, 6, 640, {'collision_params': {'shape': 'box', 'size': np.array([0.0775, 0.0775])}}), (2, 4, 1500, {'autoreset_mode': 'next_step'})])
def test_thin_wavefronts_equal_whole_wavefronts(gpr, M, n, E, kw):
    """gpr_set_wave_budget: a small batch spread over more wavefronts (fewer envs in each; down to one env per wavefront here) returns bit
    for bit what whole wavefronts return -- observations, rewards, flags, state, every auto-reset, and the episode counters' totals"""
    lay = np.ones((n, n))
    envs = []
    for budget in (0, 512, 65536):
        e = gpr.VecBenchmarkPlanningEnv(lay, M, num_envs=E, **kw)
        e.set_wave_budget(budget)
        e.enable_episode_counters()
        e.reset(seed=21)
        envs.append(e)
    g = torch.Generator(device='cuda').manual_seed(3)
    n_done = 0
    for t in range(50):
        a = (torch.rand(E, 2 * M, device='cuda', generator=g) * 20 - 10).to(envs[0].dtype)
        res = [e.step(a) for e in envs]
        for r in res[1:]:
            for k in res[0][0]:
                assert torch.equal(res[0][0][k], r[0][k]), (t, k)
            assert torch.equal(res[0][1], r[1]) and torch.equal(res[0][2], r[2]) and torch.equal(res[0][3], r[3]), t
            for k in ('is_success', 'mover_collision', 'wall_collision'):
                assert torch.equal(res[0][4][k], r[4][k]), (t, k)
        n_done += int((res[0][2] | res[0][3]).sum())
    assert n_done > E // 2
    # gpr_step_n under the same budgets (its steps' parameter blocks carry the spread): 12 more steps in one launch each
    acts = (torch.rand(12, E, 2 * M, device='cuda', generator=g) * 20 - 10).to(envs[0].dtype)
    rolls = [e.step_n(acts, e.alloc_rollout(12)) for e in envs]
    for r in rolls[1:]:
        for k in rolls[0]:
            if isinstance(rolls[0][k], torch.Tensor):
                assert torch.equal(rolls[0][k], r[k]), k
    n_done += int((rolls[0]['terminated'].bool() | rolls[0]['truncated'].bool()).sum())
    tot = [e.read_episode_counters().cpu().numpy() for e in envs]
    for e in envs[1:]:
        assert torch.equal(envs[0].pos, e.pos) and torch.equal(envs[0].vel, e.vel) and torch.equal(envs[0].goals, e.goals)
        assert torch.equal(envs[0].elapsed_steps, e.elapsed_steps)
    assert np.array_equal(tot[0], tot[1]) and np.array_equal(tot[0], tot[2]) and tot[0][0] == n_done
    # the budget is refused once the counter table is too small for it, and a NULL handle is an error
    small = gpr.VecBenchmarkPlanningEnv(lay, M, num_envs=64)
    small.set_wave_budget(0)
    small.enable_episode_counters()
    with pytest.raises(Exception):
        small.set_wave_budget(65536)
    small.close()
    for e in envs:
        e.close()


def test_thin_wavefronts_with_agent_views(gpr):
    """the per-agent views of the step kernel's epilogue (rows staged per wavefront) under thin wavefronts: the PettingZoo-parallel env at a
    small batch (spread by default) against the same env with whole wavefronts"""
    E, M = 500, 6
    lay = np.ones((7, 7))
    a_env = gpr.VecBenchmarkPlanningParallelEnv(lay, M, num_envs=E)
    b_env = gpr.VecBenchmarkPlanningParallelEnv(lay, M, num_envs=E)
    b_env.env.set_wave_budget(0)
    a_env.reset(seed=8); b_env.reset(seed=8)
    g = torch.Generator(device='cuda').manual_seed(4)
    for t in range(30):
        act = torch.rand(E, 2 * M, device='cuda', generator=g) * 20 - 10
        acts = {ag: act[:, 2 * i:2 * i + 2] for i, ag in enumerate(a_env.agents)}
        ra, rb = a_env.step(acts), b_env.step(acts)
        for part_a, part_b in zip(ra[:4], rb[:4]):
            for ag in a_env.agents:
                assert torch.equal(torch.as_tensor(part_a[ag]), torch.as_tensor(part_b[ag])), (t, ag)
    a_env.close(); b_env.close()


def test_large_layout_tables_fit(gpr, oracle):
    """a 64 x 64 layout with holes (GPR_MAX_TILES): the f64 tile tables (~47 KB) plus the kernels' static LDS stay launchable, and the
    verdicts still match the oracle (table bytes vs the 64 KB default LDS limit)"""
    rng = np.random.default_rng(0)
    layout = (rng.uniform(size=(64, 64)) > 0.1).astype(np.int8)
    layout[:8, :8] = 1
    M, E = 2, 64
    kw = dict(std_noise=0.0, max_reset_trials=200)
    env = gpr.VecBenchmarkPlanningEnv(layout, M, num_envs=E, precision='f64', **kw)
    ref = oracle.OraclePlanningEnv(layout, M, E, **kw)
    env.reset(seed=1); ref.reset(seed=1)
    assert np.array_equal(env.pos.cpu().numpy(), ref.pos)
    a = rng.uniform(-10, 10, size=(E, 2 * M))
    obs, r, term, trunc, info = env.step(torch.as_tensor(a, device='cuda'))
    o2 = ref.step(a)
    assert np.array_equal(term.cpu().numpy(), o2['terminated']) and np.array_equal(info['wall_collision'].cpu().numpy(), o2['wall_collision'])
    assert np.abs(obs['achieved_goal'].cpu().numpy() - o2['achieved_goal']).max() < 1e-12
    env.close()


def test_bench_two_ranks_on_one_gpu(gpr):
    """`python bench.py --gpus 2` with no launcher around it: the parent spawns the two ranks (before touching the GPU), gloo
    rendezvous on 127.0.0.1, both ranks on cuda:0; rank 0's line carries the weak and the strong shape and the flush extra"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--single-device', '--envs-per-gpu', '4096',
                        '--steps', '30', '--warmup', '10', '--min-seconds', '0.02', '--extra-seconds', '0.02', '--regions', '3', '--no-cpu-baseline', '--launch-timeout', '200', '--gather-timeout', '60',
                        '--watchdog', '180'], env=env, capture_output=True, text=True, timeout=240)   # (a hang ends with the ranks' stacks in stderr)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['value'] > 0 and line['steps'] == 30
    c = line['config']
    assert c['strong_scaling']['envs_total'] == 4096 and c['strong_scaling']['envs_per_gpu'] == 2048
    assert c['strong_scaling']['rollout_launch']['env_steps_per_s'] > 0
    assert 0 < c['done_fraction_per_step'] < 0.5
    # (3 regions with the repeat count that first reached 0.02 s: the later ones may fall a few per cent short of it)
    assert c['timed_seconds'] >= 0.05 and c['timed_steps'] % 30 == 0 and c['timed_regions'] == 3 and len(c['headline_spread']['region_seconds']) == 3
    assert line['strong_value'] == c['strong_scaling']['env_steps_per_s'] > 0
    # north_star's collective is in the DEFAULT line: one all-gather of the [50, E/N, .] rollout buffers per flush, timed
    g = c['rollout_gather']
    assert g['status'] == 'ok' and g['env_steps_per_s'] > 0 and g['minibatch']['env_steps_per_s'] > 0 and g['counters_only']['env_steps_per_s'] > 0
    assert g['bytes_per_rank_per_flush'] > 50 * 2048 * 100
    assert c['distributed'] == {'backend': 'gloo', 'world_size_seen_by_backend': 2}
    assert line['roofline']['launches_in_flight'] in (1, 2, 4) and line['roofline']['envs_per_launch'] * line['roofline']['launches_in_flight'] == 4096
    # (4 096 envs per rank leave SIMDs idle: thin launches, i.e. the k_step_thin instance of the step kernel)
    assert line['roofline']['kernel'] == 'gpr::k_step_thin<float, float, 4, false, true, true, 0>'


def test_bench_four_ranks_on_one_gpu(gpr):
    """the launcher, the sharding and all three flush collectives at the largest world size ONE GPU box allows (its process guard admits
    six GPU processes: this test process + 4 ranks; world size 8 runs over gloo on the CPU, tests/test_host_cpu.py)"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '4', '--backend', 'gloo', '--single-device', '--envs-per-gpu', '4096',
                        '--steps', '20', '--warmup', '5', '--min-seconds', '0.02', '--extra-seconds', '0.02', '--regions', '2', '--sub-batches', '1',
                        '--no-cpu-baseline', '--launch-timeout', '250', '--gather-timeout', '90', '--watchdog', '230'],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    c = line['config']
    assert line['n_gpus'] == 4 and c['distributed'] == {'backend': 'gloo', 'world_size_seen_by_backend': 4}
    assert c['strong_scaling']['envs_total'] == 4096 and c['strong_scaling']['envs_per_gpu'] == 1024 and line['strong_value'] > 0
    g = c['rollout_gather']
    assert g['status'] == 'ok' and g['bytes_per_rank_per_flush'] == 50 * 1024 * (4 * 4 * 8 + 4 + 2)
    assert g['env_steps_per_s'] > 0 and g['minibatch']['env_steps_per_s'] > 0 and g['counters_only']['env_steps_per_s'] > 0


def test_bench_dead_rank_fails_fast(gpr):
    """a rank that dies before the rendezvous must not leave its peer waiting: the self-launcher polls its children, stops the others
    and returns non-zero within seconds, without a JSON line"""
    import os, subprocess, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--single-device', '--envs-per-gpu', '1024',
                        '--steps', '10', '--warmup', '2', '--no-cpu-baseline', '--fail-rank', '1', '--launch-timeout', '200', '--watchdog', '180'],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and time.monotonic() - t0 < 120
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert 'rank 1' in r.stderr


@pytest.mark.parametrize('K,kw', [(2, {}), (4, {'precision': 'f64'}), (4, {'autoreset_mode': 'next_step'})])
def test_sub_batches_equal_the_whole_batch(gpr, K, kw):
    """VecBenchmarkPlanningEnvSubBatched (K handles over contiguous env ranges, K streams) returns bit for bit what one handle over the
    whole batch returns -- through step() (events), and through step_all_into (one gpr_step_many call, outputs bound into slices)"""
    E, M, T = 4096, 4, 60
    lay = np.ones((6, 6))
    whole = gpr.VecBenchmarkPlanningEnv(lay, M, num_envs=E, **kw)
    parts = gpr.VecBenchmarkPlanningEnvSubBatched(lay, M, num_envs=E, num_sub_batches=K, **kw)
    K = parts.num_sub_batches  # (as many as there are streams that really overlap)
    assert K >= 2 and parts.stream_probe['tried'] >= K, parts.stream_probe
    ow, _ = whole.reset(seed=5); op, _ = parts.reset(seed=5)
    for k in ow:
        assert torch.equal(ow[k], op[k]), k
    g = torch.Generator(device='cuda').manual_seed(1)
    acts = (torch.rand(T, E, 2 * M, device='cuda', generator=g) * 20 - 10).to(whole.dtype)
    n_done = 0
    for t in range(T // 2):
        a = whole.step(acts[t]); b = parts.step(acts[t])
        for k in a[0]:
            assert torch.equal(a[0][k], b[0][k]), (t, k)
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
        for k in ('is_success', 'mover_collision', 'wall_collision'):
            assert torch.equal(a[4][k], b[4][k]), (t, k)
        n_done += int((a[2] | a[3]).sum())
    assert n_done > E // 4
    # the bench's path: one C-ABI call per env-step, outputs land in slices of [E, .] tensors, no events
    z = lambda *s, dtype=whole.dtype: torch.zeros(*s, dtype=dtype, device='cuda')  # noqa: E731
    outs = dict(observation=z(E, whole.obs_dim), achieved_goal=z(E, 2 * M), desired_goal=z(E, 2 * M), reward=z(E), terminated=z(E, dtype=torch.uint8),
                truncated=z(E, dtype=torch.uint8))
    bound = parts.bind_outputs(**outs)
    torch.cuda.synchronize()
    for t in range(T // 2, T):
        a = whole.step(acts[t])
        parts.step_all_into(parts.pack_actions(acts[t]), bound)
        parts.synchronize()
        for k in a[0]:
            assert torch.equal(a[0][k], outs[k]), (t, k)
        assert torch.equal(a[1], outs['reward']) and torch.equal(a[2], outs['terminated'].bool()) and torch.equal(a[3], outs['truncated'].bool())
    for k in range(K):
        sl = parts.sub_slice(k)
        assert torch.equal(whole.pos[sl], parts.sub(k).pos) and torch.equal(whole.goals[sl], parts.sub(k).goals) and torch.equal(whole.elapsed_steps[sl], parts.sub(k).elapsed_steps)
    whole.close(); parts.close()


def test_sub_batched_reset_is_ordered_before_the_next_steps(gpr):
    """reset() runs the sub-batches' reset kernels on the CALLER's stream; step_all_into launches on the sub-batches' own streams without
    events. reset() therefore ends by making those streams wait: reset -> step with NO host synchronisation in between must
    equal the single batch, every time (a large batch, so that the reset kernels are still running when the step launches arrive)."""
    E, M = 65536, 4
    lay = np.ones((6, 6))
    whole = gpr.VecBenchmarkPlanningEnv(lay, M, num_envs=E)
    parts = gpr.VecBenchmarkPlanningEnvSubBatched(lay, M, num_envs=E, num_sub_batches=4)
    g = torch.Generator(device='cuda').manual_seed(2)
    acts = (torch.rand(3, E, 2 * M, device='cuda', generator=g) * 20 - 10)
    z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device='cuda')  # noqa: E731
    outs = dict(observation=z(E, whole.obs_dim), achieved_goal=z(E, 2 * M), desired_goal=z(E, 2 * M), reward=z(E), terminated=z(E, dtype=torch.uint8),
                truncated=z(E, dtype=torch.uint8))
    bound = parts.bind_outputs(**outs)
    packed = [parts.pack_actions(acts[i]) for i in range(3)]
    torch.cuda.synchronize()
    for rep in range(8):
        whole.reset(seed=100 + rep)
        for i in range(3):
            a = whole.step(acts[i])
        want = {k: v.clone() for k, v in a[0].items()}
        parts.reset(seed=100 + rep)          # no synchronisation from here ...
        for i in range(3):
            parts.step_all_into(packed[i], bound)
        parts.synchronize()                  # ... to here (stream order only)
        for k in want:
            assert torch.equal(want[k], outs[k]), (rep, k)
        sd = parts.state_dict()
        parts.load_state_dict(sd)            # the same for a restored checkpoint: copies on the caller's stream, steps on the sub-batches'
        parts.step_all_into(packed[0], bound)
        parts.synchronize()
        b = whole.step(acts[0])
        assert torch.equal(b[0]['achieved_goal'], outs['achieved_goal']), rep
    whole.close(); parts.close()


def test_parallel_env_with_sub_batches_equals_one_launch(gpr):
    """VecBenchmarkPlanningParallelEnv(num_sub_batches=K): per-agent observations / rewards / flags bit for bit those of the one-launch
    env, through step() (dict of agents) and through the bound outputs + step_all_into (bench.py's config-5 rows)"""
    E, M, n = 2048, 16, 10
    lay = np.ones((n, n))
    one = gpr.VecBenchmarkPlanningParallelEnv(lay, M, num_envs=E)
    sub = gpr.VecBenchmarkPlanningParallelEnv(lay, M, num_envs=E, num_sub_batches=4)
    assert sub.num_sub_batches >= 2
    oa, _ = one.reset(seed=9); ob, _ = sub.reset(seed=9)
    for a in one.agents:
        assert torch.equal(oa[a], ob[a]), a
    g = torch.Generator(device='cuda').manual_seed(3)
    n_done = 0
    for t in range(30):
        act = (torch.rand(E, 2 * M, device='cuda', generator=g) * 20 - 10)
        ra = one.step(act); rb = sub.step(act)
        for a in one.agents:
            assert torch.equal(ra[0][a], rb[0][a]) and torch.equal(ra[1][a], rb[1][a]), (t, a)
        assert torch.equal(ra[2]['mover_0'], rb[2]['mover_0']) and torch.equal(ra[3]['mover_0'], rb[3]['mover_0'])
        n_done += int((ra[2]['mover_0'] | ra[3]['mover_0']).sum())
    assert n_done > E
    assert torch.equal(one.state(), sub.state())
    # getters and diagnostics of the sub-batched env = those of the one-launch env (noise streams are keyed by global env id; the call counters agree)
    names = [one.agents[3], one.agents[0], one.agents[15]]
    for nm in ('qpos', 'qvel', 'qacc'):
        assert torch.equal(getattr(one.env, f'get_mover_{nm}_arr')(names), getattr(sub.env, f'get_mover_{nm}_arr')(names)), nm
    assert torch.equal(one.env.get_mover_qpos('mover_5', add_noise=True), sub.env.get_mover_qpos('mover_5', add_noise=True))
    assert one.env.diagnostics() == sub.env.diagnostics()
    # bound outputs: the views land in caller-owned tensors
    z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device='cuda')  # noqa: E731
    outs = dict(observation=z(E, 2 * M), achieved_goal=z(E, 2 * M), desired_goal=z(E, 2 * M), reward=z(E), terminated=z(E, dtype=torch.uint8),
                truncated=z(E, dtype=torch.uint8), agent_observation=z(E, M, 4 * M + 2), agent_reward=z(E, M))
    bound = sub.bind_outputs(**outs)
    for t in range(5):
        act = (torch.rand(E, 2 * M, device='cuda', generator=g) * 20 - 10)
        ra = one.step(act)
        sub.env.wait_for_caller()
        sub.env.step_all_into(sub.env.pack_actions(act), bound)
        sub.env.synchronize()
        assert torch.equal(one._agent_obs, outs['agent_observation']) and torch.equal(one._agent_reward, outs['agent_reward']), t
    one.close(); sub.close()


def test_step_many_refuses_a_handle_twice(gpr):
    """two launches of ONE handle would run on two threads and race on its call counter"""
    import ctypes as C
    env = gpr.VecBenchmarkPlanningEnv(np.ones((6, 6)), 4, num_envs=256)
    env.reset(seed=0)
    nat, lib = gpr._native, gpr._native.load()
    act = torch.zeros(256, 8, device='cuda')
    handles = (C.c_void_p * 2)(env._handle, env._handle)
    acts = (C.c_void_p * 2)(act.data_ptr(), act.data_ptr())
    outs = (C.POINTER(nat.GprStepOut) * 2)(C.pointer(env._out), C.pointer(env._out))
    st = torch.cuda.current_stream().cuda_stream
    streams = (C.c_void_p * 2)(st, st)
    seed, tick = env.get_rng_state()
    rc = lib.gpr_step_many(2, handles, acts, outs, streams)
    assert rc == nat.E_INVALID and b'twice' in lib.gpr_last_error(env._handle)
    assert env.get_rng_state() == (seed, tick), 'a refused call must not have launched anything'
    env.close()


# ----------------------------------------------------------------------------------------------------------------------
# fixtures taken from the imported reference METHODS (tests/golden/make_golden.py, round 2): observation orderings (PL:536-573)
# and the per-cycle control decision (PL:420-450), through the C ABI
# ----------------------------------------------------------------------------------------------------------------------
def _fixed_state_env(gpr, M, E, **kw):
    env = gpr.VecBenchmarkPlanningEnv(np.ones((6, 6)), M, num_envs=E, precision='f64', std_noise=0.0, autoreset_mode='disabled',
                                      max_episode_steps=None, **kw)
    env.reset(seed=0)
    return env


@pytest.mark.parametrize('mode', ['acc', 'jerk'])
def test_obs_ordering_matches_reference_get_obs(gpr, mode):
    z = load_npz('obs_ordering.npz')
    for M in (1, 3, 4):
        t = f'{mode}_M{M}'
        env = _fixed_state_env(gpr, M, 1, learn_jerk=(mode == 'jerk'), num_cycles=0)
        f = lambda k: torch.as_tensor(z[f'{t}_{k}'].reshape(1, -1), device=env.device)  # noqa: E731
        env.pos.copy_(f('pos')); env.vel.copy_(f('vel')); env.acc.copy_(f('acc')); env.goals.copy_(f('goals'))
        obs, *_ = env.step(torch.zeros(1, 2 * M, dtype=env.dtype, device=env.device))
        for k in ('observation', 'achieved_goal', 'desired_goal'):
            assert np.array_equal(obs[k].cpu().numpy()[0], z[f'{t}_{k}']), (t, k)  # a num_cycles = 0 step only assembles the observation
        env.close()


@pytest.mark.parametrize('mode', ['acc', 'jerk'])
def test_control_decision_matches_reference_step_callback(gpr, mode):
    """one control cycle (num_cycles = 1): the ctrl the reference hands to MuJoCo, recovered from the state change of the step"""
    z = load_npz('step_callback.npz')
    vel, acc, act, ctrl = z[f'{mode}_vel'], z[f'{mode}_acc'], z[f'{mode}_action'], z[f'{mode}_ctrl']
    n, M, dt = vel.shape[0], vel.shape[1], float(z['dt'])
    env = _fixed_state_env(gpr, M, n, learn_jerk=(mode == 'jerk'), num_cycles=1)
    pos = np.tile(np.array([[0.40, 0.40, 1.00, 1.00]]), (n, 1))
    env.pos.copy_(torch.as_tensor(pos, device=env.device))
    env.vel.copy_(torch.as_tensor(vel.reshape(n, -1), device=env.device)); env.acc.copy_(torch.as_tensor(acc.reshape(n, -1), device=env.device))
    before = (env.acc if mode == 'jerk' else env.vel).clone()
    env.step(torch.as_tensor(act, device=env.device))
    got = (((env.acc if mode == 'jerk' else env.vel) - before) / dt).cpu().numpy().reshape(n, M, 2)  # act' = act + dt jerk / v' = v + dt acc
    err = np.abs(got - ctrl).max()
    record_metric(f'control_decision[{mode}]', max_ctrl_err=err, tol=1e-8 * (100.0 if mode == 'jerk' else 10.0))
    assert err < 1e-8 * (100.0 if mode == 'jerk' else 10.0), err
    env.close()


@pytest.mark.parametrize('M,n,E,kw', [(4, 6, 20000, {}), (4, 6, 3000, {'autoreset_mode': 'next_step'}), (16, 10, 1500, {}),
                                      (4, 6, 3000, {'precision': 'f64'}), (3, 5, 2000, {'collision_params': {'shape': 'box', 'size': np.array([0.0775, 0.0775])}})])
def test_step_n_equals_n_steps(gpr, M, n, E, kw):
    """gpr_step_n (T steps in one launch, every block walking through its steps alone) returns bit for bit what T calls of
    gpr_step return: every output row, the final state, the position of the random streams"""
    T = 57  # past the TimeLimit: truncations, collisions and auto-resets inside the launch
    lay = np.ones((n, n))
    a = gpr.VecBenchmarkPlanningEnv(lay, M, num_envs=E, **kw); b = gpr.VecBenchmarkPlanningEnv(lay, M, num_envs=E, **kw)
    a.reset(seed=6); b.reset(seed=6)
    g = torch.Generator(device='cuda').manual_seed(2)
    acts = (torch.rand(T, E, 2 * M, device='cuda', generator=g) * 20 - 10).to(a.dtype)
    acts[:, : E // 4] *= 0.02  # (some envs live to the TimeLimit)
    roll = b.step_n(acts, b.alloc_rollout(T))
    n_done = 0
    for t in range(T):
        obs, rew, term, trunc, info = a.step(acts[t])
        for k in ('observation', 'achieved_goal', 'desired_goal'):
            assert torch.equal(obs[k], roll[k][t]), (t, k)
        assert torch.equal(rew, roll['reward'][t]) and torch.equal(term, roll['terminated'][t].bool()) and torch.equal(trunc, roll['truncated'][t].bool())
        for k in ('is_success', 'mover_collision', 'wall_collision'):
            assert torch.equal(info[k], roll[k][t].bool()), (t, k)
        done = term | trunc
        n_done += int(done.sum())
        if a.autoreset_mode == 'same_step' and bool(done.any()):
            for k in ('observation', 'achieved_goal', 'desired_goal'):
                assert torch.equal(info['final_obs'][k][done], roll['final_' + k][t][done]), (t, k)
    assert n_done > E // 2
    for k in ('pos', 'vel', 'goals', 'elapsed_steps'):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert a.get_rng_state() == b.get_rng_state()
    a.close(); b.close()


# ----------------------------------------------------------------------------------------------------------------------
# round 3: the step() / reset() TEMPLATES as the reference's methods execute them (tests/golden/step_template.npz, reset_template.npz:
# BasicPlanarRoboticsSingleAgentEnv.step / .reset and BenchmarkPlanningEnv._reset_callback called as written on a stand-in world),
# through the C ABI. tests/test_oracle_golden.py holds the oracle to the same fixtures bit for bit.
# ----------------------------------------------------------------------------------------------------------------------
def _template_cp(z, tag):
    box = bool(z[f'{tag}_shape'])
    return {'shape': 'box' if box else 'circle', 'size': z[f'{tag}_size'] if box else float(z[f'{tag}_size'][0]),
            'offset': float(z[f'{tag}_offset']), 'offset_wall': float(z[f'{tag}_offset_wall'])}


@pytest.mark.parametrize('precision', ['f64', 'f32_f64acc', 'f32'])
def test_step_template_matches_reference_step(gpr, precision):
    """the cycle loop's break, the state / flags / observation of the colliding cycle, reward, terminated, info, the clip (BE:1835-1950)"""
    z = load_npz('step_template.npz')
    exact = precision == 'f64'
    worst, n_cmp, n_flag_off = 0.0, 0, 0
    for tag in z['tags']:
        M, jerk, E = int(z[f'{tag}_M']), bool(z[f'{tag}_jerk']), z[f'{tag}_starts'].shape[0]
        env = gpr.VecBenchmarkPlanningEnv(z[f'{tag}_layout'], M, num_envs=E, precision=precision, std_noise=0.0, autoreset_mode='disabled',
                                          max_episode_steps=None, learn_jerk=jerk, num_cycles=int(z[f'{tag}_num_cycles']),
                                          collision_params=_template_cp(z, tag))
        env.reset(seed=0)
        dev = lambda a: torch.as_tensor(np.ascontiguousarray(a).reshape(E, -1), device=env.device).to(env.dtype)  # noqa: E731
        env.pos.copy_(dev(z[f'{tag}_starts'])); env.vel.copy_(dev(z[f'{tag}_v0'])); env.acc.copy_(dev(z[f'{tag}_a0'])); env.goals.copy_(dev(z[f'{tag}_goals']))
        follow = np.ones(E, dtype=bool)  # envs that are valid in the fixture and (float builds) have decided like the reference so far
        for t in range(z[f'{tag}_actions'].shape[0]):
            obs, rew, term, trunc, info = env.step(dev(z[f'{tag}_actions'][t]))
            follow &= z[f'{tag}_valid'][t]
            got = {'reward': rew, 'terminated': term, 'truncated': trunc, **{k: info[k] for k in ('is_success', 'mover_collision', 'wall_collision')}}
            same = np.ones(E, dtype=bool)
            for k, v in got.items():
                same &= v.double().cpu().numpy() == z[f'{tag}_{k}'][t].astype(np.float64)
            if exact:
                assert same[follow].all(), (tag, t)
            n_flag_off += int((~same & follow).sum()); n_cmp += int(follow.sum())
            follow &= same
            for k, ref in (('observation', obs['observation']), ('achieved_goal', obs['achieved_goal']), ('desired_goal', obs['desired_goal']),
                           ('pos', env.pos), ('vel', env.vel)) + ((('acc', env.acc),) if jerk else ()):
                d = np.abs(ref.double().cpu().numpy() - z[f'{tag}_{k}'][t])[follow]
                worst = max(worst, float(d.max()) if d.size else 0.0)
        env.close()
    record_metric(f'step_template[{precision}]', max_abs_err=worst, env_steps_compared=n_cmp, env_steps_deciding_differently=n_flag_off,
                  tol=1e-12 if exact else 1e-4)
    assert worst < (1e-12 if exact else 1e-4), worst
    assert n_flag_off <= (0 if exact else 0.01 * n_cmp), (n_flag_off, n_cmp)  # float builds: a scenario within float rounding of a threshold


@pytest.mark.parametrize('precision', ['f64', 'f32'])
def test_reset_template_matches_reference_reset(gpr, oracle, precision):
    """reset() + _reset_callback as written on a scripted np_random.uniform, replayed through gpr_reset_candidates: the first valid set of
    the stream wins, the goal loop goes on from there, goals are not checked against starts (PL:355-418, BE:1770-1833)"""
    z = load_npz('reset_template.npz')
    for tag in z['tags']:
        M, cand = int(z[f'{tag}_M']), z[f'{tag}_cand']
        E = cand.shape[0]
        env = gpr.VecBenchmarkPlanningEnv(z[f'{tag}_layout'], M, num_envs=E, precision=precision, std_noise=0.0, autoreset_mode='disabled',
                                          max_episode_steps=None, collision_params=_template_cp(z, tag))
        stream = np.ascontiguousarray(np.swapaxes(cand, 0, 1))
        obs, info = env.reset(seed=0, options={'candidates': stream})
        if precision == 'f64':
            assert np.array_equal(info['candidates_used'].cpu().numpy(), z[f'{tag}_n_used']), tag
            assert np.array_equal(env.pos.cpu().numpy(), z[f'{tag}_starts']) and np.array_equal(env.goals.cpu().numpy(), z[f'{tag}_goals']), tag
            for k in ('observation', 'achieved_goal', 'desired_goal'):
                assert np.array_equal(obs[k].cpu().numpy(), z[f'{tag}_{k}']), (tag, k)
            for k in ('is_success', 'mover_collision', 'wall_collision'):
                assert np.array_equal(info[k].cpu().numpy(), z[f'{tag}_{k}']), (tag, k)
        else:  # float storage: the candidates themselves are rounded to float; the oracle with float storage is the yardstick
            ref = oracle.OraclePlanningEnv(z[f'{tag}_layout'], M, E, std_noise=0.0, collision_params=_template_cp(z, tag), autoreset='disabled',
                                           max_episode_steps=None, storage_f32=True)
            o, n_used = ref.reset_candidates(stream.astype(np.float32).astype(np.float64))
            same = info['candidates_used'].cpu().numpy() == n_used
            assert same.mean() > 0.9, (tag, same.mean())  # (a float-rounded candidate can sit on a threshold)
            assert np.array_equal(env.pos.double().cpu().numpy()[same], ref.pos[same]) and np.array_equal(env.goals.double().cpu().numpy()[same], ref.goal[same])
        assert not env.vel.any() and not env.elapsed_steps.any()
        env.close()
    # a stream that runs out: the last set stands (as max_reset_trials does), for the start loop and for the goal loop
    env = gpr.VecBenchmarkPlanningEnv(np.ones((4, 4)), 2, num_envs=3, precision=precision, std_noise=0.0)
    bad = np.array([[0.3, 0.3], [0.3, 0.3]]); ok1 = np.array([[0.3, 0.3], [0.7, 0.7]]); ok2 = np.array([[0.5, 0.2], [0.2, 0.6]])
    stream = np.stack([np.stack([bad, ok1, ok1]), np.stack([bad, bad, ok2]), np.stack([bad, bad, bad])])  # [n_cand = 3][E = 3][M][2]
    obs, info = env.reset(seed=0, options={'candidates': stream})
    want_pos = np.stack([bad, ok1, ok1]).reshape(3, -1); want_goal = np.stack([bad, bad, ok2]).reshape(3, -1)
    want_pos[0] = bad.reshape(-1); want_goal[0] = bad.reshape(-1)      # env 0: nothing valid -> start = last set, goal = last set
    want_pos[1] = ok1.reshape(-1); want_goal[1] = bad.reshape(-1)      # env 1: start accepted as set 0, goals bad, bad -> the last stands
    want_pos[2] = ok1.reshape(-1); want_goal[2] = ok2.reshape(-1)      # env 2: start = set 0, goal: set 1 (ok2)
    want_pos[0] = bad.reshape(-1)
    got_used = info['candidates_used'].cpu().numpy()
    assert np.allclose(env.pos.double().cpu().numpy()[1:], want_pos[1:], atol=1e-6) and np.allclose(env.goals.double().cpu().numpy()[1:], want_goal[1:], atol=1e-6)
    assert np.allclose(env.pos.double().cpu().numpy()[0], bad.reshape(-1), atol=1e-6) and np.allclose(env.goals.double().cpu().numpy()[0], bad.reshape(-1), atol=1e-6)
    assert got_used.tolist() == [3, 3, 2], got_used
    env.close()


# ----------------------------------------------------------------------------------------------------------------------
# round 4: the public state getters (BE:801-855, 1244-1290) and the reference's warnings as device counts
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('precision', ['f64', 'f32'])
def test_state_getters_match_the_reference_getters(gpr, oracle, precision):
    """get_mover_qpos / qvel / qacc (+ _arr) against getters.npz -- the reference's getters called as written on the stand-in world: the z
    convention (body z minus half height = the gap above the tile, BE:821-824), row order = order of the names, noise on EVERY component
    with std_noise[0] / [1] / [2]; the noisy read bit for bit against the oracle's statement of the engine's stream (f64)."""
    z = load_npz('getters.npz')
    tol = 1e-15 if precision == 'f64' else 1e-6
    for tag in z['tags']:
        M, order = int(z[f'{tag}_M']), [int(i) for i in z[f'{tag}_order']]
        sn = z[f'{tag}_std_noise']
        mp = {'size': z[f'{tag}_mover_size'][0] if tag.startswith('cyl') else z[f'{tag}_mover_size']}
        if tag.startswith('cyl'):
            mp['shape'] = 'cylinder'
        E = 64
        env = gpr.VecBenchmarkPlanningEnv(np.ones((4, 4)), M, num_envs=E, precision=precision, std_noise=sn, mover_params=mp,
                                          initial_mover_zpos=float(z[f'{tag}_initial_mover_zpos']), autoreset_mode='disabled')
        env.reset(seed=4)
        for k, t in (('pos', env.pos), ('vel', env.vel), ('acc', env.acc)):
            t.copy_(torch.as_tensor(z[f'{tag}_{k}'].reshape(1, 2 * M), dtype=env.dtype).expand(E, 2 * M))
        names = [env.mover_names[i] for i in order]
        for which, nm, nc in ((0, 'qpos', 7), (1, 'qvel', 6), (2, 'qacc', 6)):
            arr = getattr(env, f'get_mover_{nm}_arr')(names)
            assert arr.shape == (E, M, nc) and arr.dtype == env.dtype
            assert np.abs(arr.double().cpu().numpy() - z[f'{tag}_{nm}'][None]).max() <= tol, (tag, nm)
            one = getattr(env, f'get_mover_{nm}')(names[1 % M])
            assert torch.equal(one, arr[:, 1 % M])
            # noise: every component, sigma = std_noise[which] (the reference: clean + std_noise[which] * unit normal, all components)
            assert np.allclose(z[f'{tag}_{nm}_noisy'] - z[f'{tag}_{nm}'], sn[which] * z[f'{tag}_{nm}_units'], atol=1e-15)
            big = gpr.VecBenchmarkPlanningEnv(np.ones((4, 4)), M, num_envs=8192, precision=precision, std_noise=sn, autoreset_mode='disabled')
            big.reset(seed=1)
            d = (getattr(big, f'get_mover_{nm}_arr')(big.mover_names, add_noise=True) - getattr(big, f'get_mover_{nm}_arr')(big.mover_names)).double()
            sd = d.reshape(-1, nc).std(dim=0).cpu().numpy()
            assert np.all(np.abs(sd / sn[which] - 1.0) < 0.03), (tag, nm, sd)
            assert abs(float(d.mean())) < 0.02 * sn[which]
            big.close()
        if precision == 'f64':  # the engine's noise stream, bit for bit (kind 5 of gpr-rng-v2)
            o = oracle.OraclePlanningEnv(np.ones((4, 4)), M, E, std_noise=sn)
            o.pos[:], o.vel[:], o.acc[:] = env.pos.cpu().numpy(), env.vel.cpu().numpy(), env.acc.cpu().numpy()
            o.seed, o.tick = env.get_rng_state()
            for which, nm in ((0, 'qpos'), (1, 'qvel'), (2, 'qacc')):
                want = o.get_mover_state(which, order, env.initial_mover_zpos, add_noise=True)
                got = getattr(env, f'get_mover_{nm}_arr')(names, add_noise=True).cpu().numpy()
                assert np.abs(got - want).max() < 1e-15, (tag, nm)
            assert env.get_rng_state() == (o.seed, o.tick), 'a noisy read advances the call counter by one'
        with pytest.raises(ValueError):
            env.get_mover_qpos('mover_99')
        env.close()


def test_exhausted_reset_searches_and_clipped_actions_are_counted(gpr, oracle):
    """The reference never gives up a reset search and warns every 100 trials (PL:371-375, 397-401); it warns about an action outside the
    action space before clipping it (BE:1871-1873). The engine bounds the search and clips silently on the device -- and COUNTS both:
    diagnostics(), counters 6 / 7, and a host warning from reset() / warn_diagnostics() / close()."""
    import warnings
    # 12 movers of radius 0.11 on 7 x 7 tiles need hundreds of trials: with max_reset_trials = 3 nearly every search is exhausted
    E, M = 512, 12
    env = gpr.VecBenchmarkPlanningEnv(np.ones((7, 7)), M, num_envs=E, max_reset_trials=3, precision='f64')
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        env.reset(seed=0)
    assert any('found no valid start / goal configuration' in str(x.message) for x in w), [str(x.message) for x in w]
    d = env.diagnostics()
    o = oracle.OraclePlanningEnv(np.ones((7, 7)), M, E, max_reset_trials=3)
    o.reset(seed=0)
    # an exhausted search leaves an invalid set: every env whose start set collides (safety offsets, noise-free) had its start search exhausted
    q = np.zeros((E * M, 7)); q[:, :2] = env.pos.cpu().numpy().reshape(-1, 2); q[:, 3] = 1.0
    bad_wall = ~env.qpos_is_valid(q, 0.11, add_safety_offset=True).bool().cpu().numpy().reshape(E, M).all(axis=1)
    bad_pair = env.check_mover_collision(q.reshape(E, M, 7), 0.11, add_safety_offset=True).cpu().numpy()
    n_bad_start = int((bad_wall | bad_pair).sum())
    assert n_bad_start > E // 2 and d['reset_exhausted'] >= n_bad_start, (d, n_bad_start)
    assert d['reset_exhausted'] <= 2 * E and d['actions_clipped'] == 0
    # a generous bound: nothing is exhausted, nothing is reported
    ok = gpr.VecBenchmarkPlanningEnv(np.ones((6, 6)), 4, num_envs=4096)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        ok.reset(seed=0)
        ok.count_clipped_actions()
        a = torch.zeros(4096, 8, device='cuda')
        a[:100, 3] = 10.5; a[50:200, 6] = -11.0; a[300, 0] = 10.0  # (exactly a_max is inside the action space)
        ok.enable_episode_counters()
        for _ in range(3):
            ok.step(a)
        tot = ok.read_episode_counters().cpu().numpy()
        ok.warn_diagnostics()
    msgs = [str(x.message) for x in w]
    assert not any('found no valid' in m for m in msgs) and any('600 env-step(s) had an action outside the action space' in m for m in msgs), msgs
    assert ok.diagnostics() == {'reset_exhausted': 0, 'actions_clipped': 600} and tot[7] == 600 and tot[6] == 0 and len(tot) == 8
    ok.count_clipped_actions(False)
    ok.step(a)
    assert ok.diagnostics()['actions_clipped'] == 600
    env.close(); ok.close()


@pytest.mark.parametrize('M,n,precision', [(4, 3, 'f64'), (12, 7, 'f64'), (16, 10, 'f32'), (8, 5, 'f64')])
def test_exhausted_searches_of_auto_resets_are_counted_exactly(gpr, M, n, precision):
    """GPR_CNT_RESET_EXHAUSTED through the step kernels' own reset paths (the block's queue for up to 8 lanes per env, the shared searches of
    the 16-lane kernels, gpr_step_n's in-wavefront resets): a search that finds a valid set within max_reset_trials returns a VALID set, an
    exhausted one keeps its last, INVALID candidate -- so the counter must equal the number of invalid start sets + invalid goal sets among
    the envs that were reset, which the stand-alone operators can count from the outside."""
    E, trials = 1024, 6
    env = gpr.VecBenchmarkPlanningEnv(np.ones((n, n)), M, num_envs=E, precision=precision, max_reset_trials=trials, std_noise=0.0, warn_on_reset=False)
    env.reset(seed=7)

    def invalid_sets(xy):  # [K, M, 2] float64 -> (wall | pair with the safety offsets, PL:379-383), (wall | closer than min_goal_dist, PL:403-413)
        K = xy.shape[0]
        q = np.zeros((K * M, 7)); q[:, :2] = xy.reshape(-1, 2); q[:, 3] = 1.0
        wall = ~env.qpos_is_valid(q, 0.11, add_safety_offset=True).bool().cpu().numpy().reshape(K, M).all(axis=1)
        pair = env.check_mover_collision(q.reshape(K, M, 7), 0.11, add_safety_offset=True).cpu().numpy()
        d = np.linalg.norm(xy[:, :, None, :] - xy[:, None, :, :], axis=-1) + np.eye(M)[None] * 10.0
        return wall | pair, wall | (d.min(axis=(1, 2)) < env.min_goal_dist)

    base = env.diagnostics()['reset_exhausted']
    s_bad, g_bad = invalid_sets(env.pos.double().cpu().numpy().reshape(E, M, 2)), invalid_sets(env.goals.double().cpu().numpy().reshape(E, M, 2))
    want = int(s_bad[0].sum() + g_bad[1].sum())
    slack = 0 if precision == 'f64' else 3   # (float builds compare squared distances in float: a set within an ulp of a limit may be judged differently here)
    assert abs(base - want) <= slack and (want > 0 or M == 4), (base, want)
    g = torch.Generator(device='cuda').manual_seed(3)
    total = base
    n_reset = 0
    for t in range(12):
        a = (torch.rand(E, 2 * M, device='cuda', generator=g) * 20 - 10).to(env.dtype)
        obs, rew, term, trunc, info = env.step(a)
        done = (term | trunc).cpu().numpy()
        n_reset += int(done.sum())
        s_bad = invalid_sets(env.pos.double().cpu().numpy().reshape(E, M, 2)[done])[0] if done.any() else np.zeros(0, bool)
        g_bad = invalid_sets(env.goals.double().cpu().numpy().reshape(E, M, 2)[done])[1] if done.any() else np.zeros(0, bool)
        total += int(s_bad.sum() + g_bad.sum())
        assert abs(env.diagnostics()['reset_exhausted'] - total) <= slack * (t + 2), (t, env.diagnostics(), total)
    assert n_reset > E // 4
    # the same through gpr_step_n (resets inside the wavefront that stepped the env)
    T = 6
    acts = (torch.rand(T, E, 2 * M, device='cuda', generator=g) * 20 - 10).to(env.dtype)
    roll = env.alloc_rollout(T)
    before = env.diagnostics()['reset_exhausted']
    env.step_n(acts, roll)
    after = env.diagnostics()['reset_exhausted']
    n_done = int(((roll['terminated'] | roll['truncated']) != 0).sum())
    assert n_done > 0 and after >= before and (M == 4 or after > before)
    env.close()
