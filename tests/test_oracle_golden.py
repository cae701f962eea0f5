"""CPU oracle vs the golden fixtures produced by the reference's own NumPy code (tests/golden/make_golden.py).

This is what PINS the oracle: every known-answer case of the reference's tests/test_basic_env.py and
tests/test_geometry_2D_utils.py, the closed-form trajectories of tests/test_benchmark_planning_env.py /
tests/test_benchmark_pushing_env.py, plus fuzz labelled by the reference functions themselves.
"""
import numpy as np
import pytest

from conftest import load_json, load_npz


def _layout_from_case(o, case):
    cp = case['collision_params']
    return o.OracleLayout(case['layout'], shape=cp.get('shape', 'circle'), offset=cp.get('offset', 0.0),
                          offset_wall=cp.get('offset_wall', 0.0))


def test_qpos_is_valid_reference_known_answers(oracle):
    cases = load_json('qpos_is_valid_ref_tests.json')
    assert len(cases) == 100
    for i, case in enumerate(cases):
        lay = _layout_from_case(oracle, case)
        cp = case['collision_params']
        c_size = case['c_size'] if case['c_size'] is not None else cp.get('size', 0.11)
        out = lay.qpos_is_valid(case['qpos'], c_size, case['add_safety_offset'])
        assert np.array_equal(out, case['expected']), f'case {i}'
        assert np.array_equal(out, case['reference_output']), f'case {i}'


def test_qpos_is_valid_fuzz(oracle):
    z = load_npz('qpos_is_valid_fuzz.npz')
    n_cfg = z['out'].shape[0]
    total = 0
    for i in range(n_cfg):
        lay_p = z['layout'][i]
        tx = int((lay_p[:, 0] >= 0).sum()); ty = int((lay_p[0, :] >= 0).sum())
        hs = float(z['tile_half_size'][i])
        box = bool(z['shape'][i])
        lay = oracle.OracleLayout(lay_p[:tx, :ty], (hs, hs), 'box' if box else 'circle', float(z['offset'][i]), float(z['offset_wall'][i]))
        cs = z['c_size'][i] if box else z['c_size'][i][:, 0]
        out = lay.qpos_is_valid(z['qpos'][i], cs, bool(z['add_safety_offset'][i]))
        assert np.array_equal(out, z['out'][i]), f'config {i}: {np.nonzero(out != z["out"][i])}'
        total += out.size
    assert total > 20000


def test_geometry_reference_known_answers(oracle):
    g = load_json('geometry_ref_tests.json')
    n = 0
    for c in g['segments']:
        out = oracle.check_line_segments_intersect(c['p1'], c['p2'], c['q1'], c['q2'])
        assert np.array_equal(out, c['expected'])
        n += out.size
    assert n == 17
    n = 0
    for c in g['rectangles']:
        out = oracle.check_rectangles_intersect(c['qpos_r1'], c['qpos_r2'], c['size_r1'], c['size_r2'])
        assert np.array_equal(out, c['expected'])
        v = oracle.get_2D_rect_vertices(c['qpos_r1'], c['size_r1'])
        if v.shape[0] > 1:
            assert np.array_equal(v, c['vertices_r1'])  # bit-exact incl. the float32 quaternion normalisation
        else:
            # a batch of ONE rectangle goes through BLAS in NumPy (FMA chain) and may differ in the last bit from the
            # batched (N > 1) arithmetic the oracle follows; see oracle/gpr_oracle.c rect_vertices
            assert np.abs(v - c['vertices_r1']).max() <= 2.8e-17
        n += out.size
    assert n == 17


def test_geometry_fuzz(oracle):
    z = load_npz('geometry_fuzz.npz')
    s = z['segs']
    out = oracle.check_line_segments_intersect(s[:, 0:2], s[:, 2:4], s[:, 4:6], s[:, 6:8])
    assert np.array_equal(out, z['seg_out'])
    v = oracle.get_2D_rect_vertices(z['r1'], z['s1'])
    assert np.array_equal(v, z['verts_r1'])
    out = oracle.check_rectangles_intersect(z['r1'], z['r2'], z['s1'], z['s2'])
    assert np.array_equal(out, z['rect_out'])


def test_mover_collision_fuzz(oracle):
    z = load_npz('mover_collision_fuzz.npz')
    for i in range(z['res'].shape[0]):
        M = int(z['M'][i]); box = bool(z['shape'][i])
        lay = oracle.OracleLayout(np.ones((10, 10)), shape='box' if box else 'circle', offset=float(z['offset'][i]))
        cs = z['c_size'][i][:M] if box else z['c_size'][i][:M, 0]
        res = lay.check_mover_collision(z['qpos'][i][:M], cs, bool(z['add'][i]))
        assert res == bool(z['res'][i]), f'case {i}'


def test_ensure_max_dyn_val(oracle):
    z = load_npz('ensure_max_dyn_val.npz')
    vals, ders = oracle.ensure_max_dyn_val(z['cur'], z['maxv'], z['der'], float(z['cycle_time']))
    assert np.array_equal(vals, z['vals'])
    assert np.array_equal(ders, z['ders'])


def test_reward_cases(oracle):
    z = load_npz('reward_cases.npz')
    for M in (1, 2, 4, 8):
        rew, term, succ = oracle.compute_reward(z[f'M{M}_ag'], z[f'M{M}_dg'], z[f'M{M}_mc'], z[f'M{M}_wc'], M, float(z['threshold_pos']))
        assert np.array_equal(rew, z[f'M{M}_reward'])
        assert np.array_equal(term, z[f'M{M}_terminated'])
        assert np.array_equal(succ, z[f'M{M}_is_success'])


def test_constants(oracle):
    cases = load_json('constants.json')
    for c in cases:
        cp = c['collision_params']
        lay = oracle.OracleLayout(c['layout'], shape=cp.get('shape', 'circle'), offset=cp.get('offset', 0.0),
                                  offset_wall=cp.get('offset_wall', 0.0))
        xc, yc = lay.tile_centres()
        assert np.array_equal(xc, c['x_pos_tiles'][:, 0]) and np.array_equal(yc, c['y_pos_tiles'][0, :])
        for which, key in enumerate(('idx_3x3', 'idx_2x2_bl', 'idx_2x2_br', 'idx_2x2_tl', 'idx_2x2_tr')):
            ix, iy = lay.tile_indices_mask(which)
            assert np.array_equal(ix, c[key][0]) and np.array_equal(iy, c[key][1]), key
        mn, mx, mgd, gh = lay.planning_constants(cp.get('size', 0.11))
        assert np.array_equal(mn, c['min_xy_pos']) and np.array_equal(mx, c['max_xy_pos'])
        assert mgd == c['min_goal_dist']
        assert np.array_equal(gh, c['goal_space_high'][:2])


def test_reset_predicates(oracle):
    z = load_npz('reset_predicates.npz')
    for ci in range(int(z['n_cfg'])):
        M = int(z[f'{ci}_M']); box = bool(z[f'{ci}_shape'])
        lay = oracle.OracleLayout(z[f'{ci}_layout'], shape='box' if box else 'circle', offset=float(z[f'{ci}_offset']),
                                  offset_wall=float(z[f'{ci}_offset_wall']))
        size = z[f'{ci}_size'] if box else float(z[f'{ci}_size'][0])
        cand = z[f'{ci}_cand']
        for i in range(cand.shape[0]):
            q = np.zeros((M, 7)); q[:, 2] = 0.003; q[:, 3] = 1; q[:, :2] = cand[i]
            valid = lay.qpos_is_valid(q, size, True)
            if (z[f'{ci}_valid'][i] >= 0).all():  # -1: the reference's own sanity assert fired (BE:650), no defined answer
                assert np.array_equal(valid, z[f'{ci}_valid'][i])
            assert lay.check_mover_collision(q, size, True) == bool(z[f'{ci}_collision'][i])


@pytest.mark.parametrize('storage_f32', [False])
def test_closed_form_trajectories(oracle, storage_f32):
    """The oracle's env step (sigma = 0, far from walls) against the closed-form loops the reference's CI asserts MuJoCo
    equals, with the reference's own tolerance (np.allclose defaults: rtol 1e-5, atol 1e-8) -- and a much tighter one."""
    z = load_npz('trajectories.npz')
    n = int(z['n'])
    assert n == 56
    for i in range(n):
        M = int(z[f'{i}_M']); mode = str(z[f'{i}_mode']); ncyc = int(z[f'{i}_num_cycles'])
        layout = np.ones((9, 9)) if str(z[f'{i}_env']) == 'planning' else np.ones((3, 3))
        env = oracle.OraclePlanningEnv(layout, M, 1, std_noise=0.0, num_cycles=ncyc, v_max=0.01, a_max=0.2, j_max=150.0,
                                       learn_jerk=(mode == 'jerk'), max_episode_steps=0, autoreset='disabled', storage_f32=storage_f32)
        env.reset(seed=0)
        env.pos[0] = z[f'{i}_starts'].reshape(-1)
        env.goal[0] = 0.3  # irrelevant
        pos = []; vel = []; acc = []
        for step in range(100):
            env.step(z[f'{i}_deriv'][None, :])
            pos.append(env.pos[0].copy()); vel.append(env.vel[0].copy()); acc.append(env.acc[0].copy())
        pos, vel, acc = np.array(pos), np.array(vel), np.array(acc)
        assert np.allclose(pos, z[f'{i}_pos']) and np.allclose(vel, z[f'{i}_vel']) and np.allclose(acc, z[f'{i}_acc']), i
        assert np.abs(pos - z[f'{i}_pos']).max() < 1e-12 and np.abs(vel - z[f'{i}_vel']).max() < 1e-13, i
        assert np.abs(acc - z[f'{i}_acc']).max() < 1e-10, i


# ----------------------------------------------------------------------------------------------------------------------
# round 2: observation orderings (PL:536-573) and the per-cycle control decision (PL:420-450), from the reference's methods
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('mode', ['acc', 'jerk'])
def test_obs_ordering(oracle, mode):
    z = load_npz('obs_ordering.npz')
    for M in (1, 3, 4):
        t = f'{mode}_M{M}'
        env = oracle.OraclePlanningEnv(np.ones((6, 6)), M, 1, std_noise=0.0, num_cycles=0, learn_jerk=(mode == 'jerk'), autoreset='disabled',
                                       max_episode_steps=None)
        env.pos[:] = z[f'{t}_pos'].reshape(1, -1); env.vel[:] = z[f'{t}_vel'].reshape(1, -1)
        env.acc[:] = z[f'{t}_acc'].reshape(1, -1); env.goal[:] = z[f'{t}_goals'].reshape(1, -1)
        o = env.step(np.zeros((1, 2 * M)))
        for k in ('observation', 'achieved_goal', 'desired_goal'):
            assert np.array_equal(o[k][0], z[f'{t}_{k}']), (t, k)


@pytest.mark.parametrize('mode', ['acc', 'jerk'])
def test_control_decision(oracle, mode):
    z = load_npz('step_callback.npz')
    vel, acc, act, ctrl = z[f'{mode}_vel'], z[f'{mode}_acc'], z[f'{mode}_action'], z[f'{mode}_ctrl']
    n, M, dt = vel.shape[0], vel.shape[1], float(z['dt'])
    env = oracle.OraclePlanningEnv(np.ones((6, 6)), M, n, std_noise=0.0, num_cycles=1, learn_jerk=(mode == 'jerk'), autoreset='disabled',
                                   max_episode_steps=None)
    env.pos[:] = np.array([0.40, 0.40, 1.00, 1.00]); env.vel[:] = vel.reshape(n, -1); env.acc[:] = acc.reshape(n, -1)
    before = (env.acc if mode == 'jerk' else env.vel).copy()
    env.step(act)
    got = (((env.acc if mode == 'jerk' else env.vel) - before) / dt).reshape(n, M, 2)
    assert np.abs(got - ctrl).max() < 1e-8 * (100.0 if mode == 'jerk' else 10.0)
    assert (np.abs(ctrl.reshape(n, -1) - act).max(axis=1) > 1e-9).mean() > 0.4  # the fixture really exercises the clamps


# ----------------------------------------------------------------------------------------------------------------------
# round 3: the step() / reset() TEMPLATES (BE:1770-1950, PL:355-418) and the pushing env's own methods (PU:373-607), from the reference's
# methods called as written on a stand-in world (tests/golden/make_golden.py: World) -- control flow, not mj_step
# ----------------------------------------------------------------------------------------------------------------------
def template_cp(z, tag):
    box = bool(z[f'{tag}_shape'])
    return {'shape': 'box' if box else 'circle', 'size': z[f'{tag}_size'] if box else float(z[f'{tag}_size'][0]),
            'offset': float(z[f'{tag}_offset']), 'offset_wall': float(z[f'{tag}_offset_wall'])}


STEP_KEYS = ('observation', 'achieved_goal', 'desired_goal', 'reward', 'terminated', 'truncated', 'is_success', 'mover_collision',
             'wall_collision')


def test_step_template(oracle):
    """step() as written: the cycle loop breaks at the first collision and returns the state, flags and observation of that cycle
    (BE:1879-1905), reward -50 / +50 / -(M - n_reached), terminated, info; actions outside the action space are clipped (BE:1871-1873).
    T consecutive calls per scenario, also past a termination (the template does not care). Bit for bit."""
    z = load_npz('step_template.npz')
    seen = dict(wall=0, pair=0, success=0, partial=0)
    for tag in z['tags']:
        M, jerk, E = int(z[f'{tag}_M']), bool(z[f'{tag}_jerk']), z[f'{tag}_starts'].shape[0]
        env = oracle.OraclePlanningEnv(z[f'{tag}_layout'], M, E, std_noise=0.0, num_cycles=int(z[f'{tag}_num_cycles']), learn_jerk=jerk,
                                       collision_params=template_cp(z, tag), autoreset='disabled', max_episode_steps=None)
        env.pos[:] = z[f'{tag}_starts'].reshape(E, -1); env.vel[:] = z[f'{tag}_v0'].reshape(E, -1); env.acc[:] = z[f'{tag}_a0'].reshape(E, -1)
        env.goal[:] = z[f'{tag}_goals'].reshape(E, -1)
        alive = np.ones(E, dtype=bool)
        for t in range(z[f'{tag}_actions'].shape[0]):
            o = env.step(z[f'{tag}_actions'][t])
            alive &= z[f'{tag}_valid'][t]  # (a mover above no tile at all: the reference asserts, BE:514 -- nothing to compare from then on)
            for k in STEP_KEYS:
                assert np.array_equal(o[k][alive], z[f'{tag}_{k}'][t][alive]), (tag, t, k)
            assert np.array_equal(env.pos[alive], z[f'{tag}_pos'][t][alive]) and np.array_equal(env.vel[alive], z[f'{tag}_vel'][t][alive]), (tag, t)
            assert np.array_equal(env.acc[alive], z[f'{tag}_acc'][t][alive]), (tag, t)
            seen['wall'] += int(o['wall_collision'][alive].sum()); seen['pair'] += int(o['mover_collision'][alive].sum())
            seen['success'] += int(o['is_success'][alive].sum()); seen['partial'] += int(((o['reward'] > -M) & (o['reward'] < 0))[alive].sum())
    assert min(seen.values()) > 100, seen  # every exit of the template is exercised


def test_reset_template(oracle):
    """reset() + _reset_callback as written on a scripted np_random.uniform: the first valid start set of the stream wins, the goal loop
    goes on from there, goals are not checked against the starts; initial flags, observation and info (BE:1797-1831). Bit for bit."""
    z = load_npz('reset_template.npz')
    for tag in z['tags']:
        M, cand = int(z[f'{tag}_M']), z[f'{tag}_cand']
        E = cand.shape[0]
        env = oracle.OraclePlanningEnv(z[f'{tag}_layout'], M, E, std_noise=0.0, collision_params=template_cp(z, tag), autoreset='disabled',
                                       max_episode_steps=None)
        assert np.array_equal(env.min_xy, z[f'{tag}_uniform_low'][0]) and np.array_equal(env.max_xy, z[f'{tag}_uniform_high'][0])  # PL:377
        o, n_used = env.reset_candidates(np.swapaxes(cand, 0, 1))
        assert np.array_equal(n_used, z[f'{tag}_n_used']), tag
        assert np.array_equal(env.pos, z[f'{tag}_starts']) and np.array_equal(env.goal, z[f'{tag}_goals']), tag
        assert not env.vel.any() and not env.acc.any() and not env.elapsed.any()
        for k in ('observation', 'achieved_goal', 'desired_goal', 'is_success', 'mover_collision', 'wall_collision'):
            assert np.array_equal(o[k], z[f'{tag}_{k}']), (tag, k)
        assert (z[f'{tag}_n_used'] > 2).mean() > 0.5  # most resets reject at least one set
        assert np.all(env.goal == env.pos, axis=1).sum() >= 5  # "goals == starts" accepted (goals unchecked against starts)


def push_reward_mover_positions(ag, wc):
    """a mover position per env that is out of the object's reach: over the wall margin (x = 0.05) where the case wants a wall collision,
    else in the corner of the safe area farthest from the object"""
    far = np.where(ag < 0.36, 0.60, 0.12)
    hit = np.stack([np.full(len(ag), 0.05), far[:, 1]], axis=1)
    return np.where(wc[:, None], hit, far)


def test_pushing_methods(oracle):
    """BenchmarkPushingEnv._get_obs / compute_reward / compute_terminated / reset() + _reset_callback as written (PU:373-607)"""
    z = load_npz('pushing_methods.npz')
    for mode in ('acc', 'jerk'):
        t = f'obs_{mode}'
        n = z[f'{t}_pos'].shape[0]
        env = oracle.OraclePushingEnv(n, std_noise=0.0, object_noise=0.0, num_cycles=0, learn_jerk=(mode == 'jerk'), autoreset='disabled',
                                      max_episode_steps=None)
        env.pos[:] = z[f'{t}_pos']; env.vel[:] = z[f'{t}_vel']; env.acc[:] = z[f'{t}_acc']; env.object[:, :2] = z[f'{t}_object']; env.goal[:] = z[f'{t}_goal']
        o = env.step(np.zeros((n, 2)))
        for k in ('observation', 'achieved_goal', 'desired_goal'):
            assert np.array_equal(o[k], z[f'{t}_{k}']), (t, k)
    # reward / terminated through the observation path: the object sits at `ag`, the goal at `dg`, the wall flag comes from a mover
    # parked outside the grid's safe area (wall collision) or well inside it
    ag, dg, wc = z['reward_ag'], z['reward_dg'], z['reward_wc']
    n = ag.shape[0]
    env = oracle.OraclePushingEnv(n, std_noise=0.0, object_noise=0.0, num_cycles=1, autoreset='disabled', max_episode_steps=None,
                                  threshold_pos=float(z['reward_threshold_pos']))
    env.pos[:] = push_reward_mover_positions(ag, wc); env.object[:, :2] = ag; env.goal[:] = dg
    o = env.step(np.zeros((n, 2)))
    assert np.array_equal(o['wall_collision'], wc) and np.array_equal(o['achieved_goal'], ag)
    assert np.array_equal(o['reward'], z['reward_reward']) and np.array_equal(o['terminated'], z['reward_terminated'])
    assert set(np.unique(z['reward_reward'])) == {-50.0, -1.0, 0.0}
    # reset(): scripted object candidates, with / without the two options (PU:389-391, 412-414)
    cand = z['reset_object_cand']; E = cand.shape[0]
    opt_s, opt_g = z['reset_opt_start'], z['reset_opt_goal']
    for has_s in (False, True):
        for has_g in (False, True):
            sel = (~np.isnan(opt_s[:, 0]) == has_s) & (~np.isnan(opt_g[:, 0]) == has_g)
            idx = np.nonzero(sel)[0]
            assert len(idx) >= 10
            env = oracle.OraclePushingEnv(len(idx), std_noise=0.0, object_noise=0.0, autoreset='disabled', max_episode_steps=None)
            assert np.array_equal(env.min_xy, z['reset_mover_low']) and np.array_equal(env.max_xy, z['reset_mover_high'])
            assert np.array_equal(env.obj_min, z['reset_object_low']) and np.array_equal(env.obj_max, z['reset_object_high'])
            assert env.min_mo_dist == float(z['reset_min_mo_dist'])
            # what is not given as an option was drawn by the (scripted) generator: hand the draws in through the same door
            ms = np.where(has_s, opt_s[idx], z['reset_mover_draw'][idx]); og = np.where(has_g, opt_g[idx], z['reset_goal_draw'][idx])
            o, n_used = env.reset_candidates(np.swapaxes(cand[idx], 0, 1), mover_start_xy_pos=ms, object_goal_xy_pos=og)
            assert np.array_equal(n_used, z['reset_n_object'][idx]), (has_s, has_g)
            assert np.array_equal(env.pos, z['reset_mover_start'][idx]) and np.array_equal(env.object[:, :2], z['reset_object_start'][idx])
            assert np.array_equal(env.goal, z['reset_object_goal'][idx])
            for k in ('observation', 'achieved_goal', 'desired_goal', 'wall_collision'):
                assert np.array_equal(o[k], z[f'reset_{k}'][idx]), k
    assert (z['reset_n_object'] >= 4).all()  # the three near candidates and (mostly) the one at min_mo_dist are rejected (PU:407: strictly greater)
