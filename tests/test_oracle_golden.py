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
