"""Generate the golden fixtures under tests/golden/ from the reference mounted at /root/reference.

Run ONCE in the build container (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is written is DATA ONLY (inputs + the outputs the reference's own NumPy code returns for them, plus the
known-answer cases the reference's own tests hold):

  qpos_is_valid_ref_tests.json   the 100 cases of tests/test_basic_env.py (inputs, expected, reference output)
  qpos_is_valid_fuzz.npz         random layouts / poses / shapes labelled by BasicPlanarRoboticsEnv.qpos_is_valid
  geometry_ref_tests.json        the segment / rectangle cases of tests/test_geometry_2D_utils.py
  geometry_fuzz.npz              random + snapped segments / rectangles labelled by geometry_2D_utils
  mover_collision_fuzz.npz       check_mover_collision over M in {2,4,8,16}, circle / box, touching distances
  ensure_max_dyn_val.npz         clamp I/O incl. norm == max
  trajectories.npz               the closed-form loops of tests/test_benchmark_planning_env.py and
                                 tests/test_benchmark_pushing_env.py (56 parametrisations) evaluated with the
                                 reference's own ensure_max_dyn_val -- what the reference CI asserts MuJoCo equals
  reward_cases.npz               compute_reward / compute_terminated on batches incl. dist == threshold
  constants.json                 ctor-time tables (tile centres, 3x3 / 2x2 index lists, min/max xy, min_goal_dist)
  reset_predicates.npz           start / goal candidate sets with the accept decisions of _reset_callback's checks
  obs_ordering.npz               BenchmarkPlanningEnv._get_obs (PL:536-573) called on the imported class with its three state getters
                                 replaced by given arrays: the observation / achieved_goal / desired_goal orderings, acc and jerk mode
  step_callback.npz              BenchmarkPlanningEnv._mujoco_step_callback (PL:420-450) called as written, per cycle: given
                                 velocities / accelerations / actions -> the ctrl values it hands to MuJoCo (acc and jerk mode,
                                 incl. the "velocity clamp overrides the jerk" branch PL:437-440)

MuJoCo itself is not installed, so nothing here depends on mj_step (see _import_reference.py).
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _import_reference import REFERENCE_ROOT, import_reference  # noqa: E402

import_reference()
from gymnasium_planar_robotics.envs.basic_envs import BasicPlanarRoboticsEnv  # noqa: E402
from gymnasium_planar_robotics.envs.manipulation.benchmark_pushing_env import BenchmarkPushingEnv  # noqa: E402
from gymnasium_planar_robotics.envs.planning.benchmark_planning_env import BenchmarkPlanningEnv  # noqa: E402
from gymnasium_planar_robotics.utils import geometry_2D_utils  # noqa: E402


def load_ref_test_module(fname: str):
    spec = importlib.util.spec_from_file_location('ref_' + fname[:-3], os.path.join(REFERENCE_ROOT, 'tests', fname))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def param_values(fn):
    for mark in fn.pytestmark:
        if mark.name == 'parametrize':
            return mark.args[1]
    raise RuntimeError('no parametrize mark')


def jsonable(x):
    if isinstance(x, np.ndarray):
        return {'__nd__': x.tolist(), 'dtype': str(x.dtype)}
    if isinstance(x, (np.floating, np.integer, np.bool_)):
        return x.item()
    if isinstance(x, dict):
        return {k: jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    return x


def make_env(layout, collision_params, num_movers=1):
    return BasicPlanarRoboticsEnv(
        layout_tiles=np.asarray(layout), num_movers=num_movers, collision_params=collision_params, render_mode=None
    )


# ----------------------------------------------------------------------------------------------------------
# 1. qpos_is_valid: the reference's own 100 known-answer cases
# ----------------------------------------------------------------------------------------------------------
def gen_qpos_ref_tests():
    mod = load_ref_test_module('test_basic_env.py')
    cases = []
    for layout, qpos, cparams, add_off, expected, c_size in param_values(mod.test_mover_position_is_valid_check):
        env = make_env(layout, cparams)
        out = env.qpos_is_valid(qpos, c_size=c_size if c_size is not None else env.c_size, add_safety_offset=add_off)
        assert (out == expected).all()
        cases.append(
            dict(
                layout=jsonable(np.asarray(layout)),
                qpos=jsonable(np.asarray(qpos, dtype=np.float64)),
                collision_params=jsonable(cparams),
                add_safety_offset=bool(add_off),
                expected=jsonable(np.asarray(expected)),
                reference_output=jsonable(np.asarray(out)),
                c_size=jsonable(c_size),
            )
        )
    with open(os.path.join(HERE, 'qpos_is_valid_ref_tests.json'), 'w') as f:
        json.dump(cases, f)
    print('qpos_is_valid_ref_tests:', len(cases))


# ----------------------------------------------------------------------------------------------------------
# 2. qpos_is_valid fuzz
# ----------------------------------------------------------------------------------------------------------
def random_layout(rng, max_t=6):
    tx, ty = rng.integers(1, max_t + 1, size=2)
    while True:
        p = rng.choice([0.5, 0.75, 0.9, 1.0])
        lay = (rng.random((tx, ty)) < p).astype(np.int8)
        if lay.sum() > 0:
            return lay


def random_quats(rng, n):
    """z-axis rotations: exact 0/45/90/180 degrees (as the reference's tests write them), random yaw, and a few
    slightly off-axis / unnormalised ones."""
    kind = rng.integers(0, 6, size=n)
    q = np.zeros((n, 4))
    yaw = rng.uniform(-np.pi, np.pi, size=n)
    q[:, 0] = np.cos(yaw / 2)
    q[:, 3] = np.sin(yaw / 2)
    q[kind == 0] = [1, 0, 0, 0]
    q[kind == 1] = [0.9238795, 0, 0, 0.3826834]
    q[kind == 2] = [0.7071068, 0, 0, 0.7071068]
    m = kind == 3
    q[m] *= rng.uniform(0.5, 2.0, size=(m.sum(), 1))  # unnormalised input (unit_vector renormalises, in float32)
    m = kind == 4
    q[m, 1:3] = rng.normal(0, 1e-5, size=(m.sum(), 2))  # sensor-noise sized tilt
    return q


def gen_qpos_fuzz(n_cfg=500, n_q=48, seed=20240601):
    rng = np.random.default_rng(seed)
    rec = dict(layout=[], shape=[], c_size=[], per_qpos=[], offset=[], offset_wall=[], add=[], qpos=[], out=[], tile=[])
    n_assert = 0
    for cfg_i in range(n_cfg):
        lay = random_layout(rng)
        tx, ty = lay.shape
        box = bool(rng.integers(0, 2))
        tile_hs = 0.12 if rng.random() < 0.8 else float(rng.choice([0.1, 0.15]))
        tile_params = {'size': np.array([tile_hs, tile_hs, 0.0176])}
        offset = float(rng.choice([0.0, 0.0, 0.001, 0.01]))
        offset_wall = float(rng.choice([0.0, 0.0, 0.001, 0.005]))
        per_qpos = bool(rng.random() < 0.3)
        if box:
            base = np.array([rng.choice([0.0775, 0.08, 0.05, 0.02, 0.1]), rng.choice([0.0775, 0.08, 0.05, 0.02, 0.1])])
            cp = {'shape': 'box', 'size': base, 'offset': offset, 'offset_wall': offset_wall}
            c_size = rng.choice([0.02, 0.05, 0.0775, 0.08, 0.1], size=(n_q, 2)) if per_qpos else base
        else:
            base = float(rng.choice([0.11, 0.1, 0.05, 0.08, 0.119]))
            cp = {'shape': 'circle', 'size': base, 'offset': offset, 'offset_wall': offset_wall}
            c_size = rng.choice([0.05, 0.08, 0.1, 0.11], size=(n_q,)) if per_qpos else base
        env = BasicPlanarRoboticsEnv(
            layout_tiles=lay, num_movers=1, collision_params=cp, tile_params=tile_params, render_mode=None
        )
        w = 2 * tile_hs
        qpos = np.zeros((n_q, 7))
        qpos[:, 0] = rng.uniform(0, tx * w, size=n_q)
        qpos[:, 1] = rng.uniform(0, ty * w, size=n_q)
        # snap some coordinates onto decision boundaries: tile edges, edge +- size, tile centres
        snap = rng.random((n_q, 2)) < 0.35
        for k in range(n_q):
            for d, nt in ((0, tx), (1, ty)):
                if snap[k, d]:
                    edge = rng.integers(0, nt + 1) * w
                    cs = (c_size[k] if per_qpos else c_size)
                    cs = float(np.atleast_1d(cs)[d if box else 0])
                    delta = rng.choice([0.0, cs, -cs, cs + offset_wall, -(cs + offset_wall), tile_hs, -tile_hs, 1e-9, -1e-9])
                    qpos[k, d] = min(max(edge + delta, 0.0), nt * w)
        qpos[:, 2] = 0.003
        qpos[:, 3:] = random_quats(rng, n_q) if box else np.tile([1.0, 0, 0, 0], (n_q, 1))
        add = bool(rng.integers(0, 2))
        try:
            out = env.qpos_is_valid(qpos, c_size=c_size, add_safety_offset=add)
        except AssertionError:
            n_assert += 1  # reference's own sanity asserts (oversized movers etc.): not a defined answer, skip
            continue
        rec['layout'].append(np.pad(lay, ((0, 6 - tx), (0, 6 - ty)), constant_values=-1))
        rec['shape'].append(int(box))
        cs_arr = np.zeros((n_q, 2))
        if box:
            cs_arr[:] = c_size
        else:
            cs_arr[:, 0] = c_size
        rec['c_size'].append(cs_arr)
        rec['per_qpos'].append(per_qpos)
        rec['offset'].append(offset)
        rec['offset_wall'].append(offset_wall)
        rec['add'].append(add)
        rec['qpos'].append(qpos)
        rec['out'].append(np.asarray(out, dtype=np.int8))
        rec['tile'].append(tile_hs)
    np.savez_compressed(
        os.path.join(HERE, 'qpos_is_valid_fuzz.npz'),
        layout=np.array(rec['layout'], dtype=np.int8),
        shape=np.array(rec['shape'], dtype=np.int8),
        c_size=np.array(rec['c_size']),
        per_qpos=np.array(rec['per_qpos']),
        offset=np.array(rec['offset']),
        offset_wall=np.array(rec['offset_wall']),
        add_safety_offset=np.array(rec['add']),
        qpos=np.array(rec['qpos']),
        out=np.array(rec['out']),
        tile_half_size=np.array(rec['tile']),
    )
    print('qpos_is_valid_fuzz: configs', len(rec['out']), 'skipped (reference assert)', n_assert,
          'valid frac', float(np.mean(rec['out'])))


# ----------------------------------------------------------------------------------------------------------
# 3. geometry
# ----------------------------------------------------------------------------------------------------------
def gen_geometry():
    mod = load_ref_test_module('test_geometry_2D_utils.py')
    seg_cases = []
    for p1, p2, q1, q2, exp in param_values(mod.test_line_segments_intersect_check):
        p1, p2, q1, q2 = (np.asarray(a, dtype=np.float64) for a in (p1, p2, q1, q2))
        out = geometry_2D_utils.check_line_segments_intersect(p1=p1, p2=p2, q1=q1, q2=q2)
        assert (out == exp).all()
        seg_cases.append(dict(p1=jsonable(p1), p2=jsonable(p2), q1=jsonable(q1), q2=jsonable(q2), expected=jsonable(np.asarray(exp))))
    rect_cases = []
    for r1, r2, s1, s2, exp in param_values(mod.test_rectangles_intersect_check):
        r1, r2, s1, s2 = (np.asarray(a, dtype=np.float64) for a in (r1, r2, s1, s2))
        out = geometry_2D_utils.check_rectangles_intersect(qpos_r1=r1, qpos_r2=r2, size_r1=s1, size_r2=s2)
        assert (out == exp).all()
        verts1 = geometry_2D_utils.get_2D_rect_vertices(qpos=r1, size=s1)
        rect_cases.append(dict(qpos_r1=jsonable(r1), qpos_r2=jsonable(r2), size_r1=jsonable(s1), size_r2=jsonable(s2),
                               expected=jsonable(np.asarray(exp)), vertices_r1=jsonable(verts1)))
    with open(os.path.join(HERE, 'geometry_ref_tests.json'), 'w') as f:
        json.dump(dict(segments=seg_cases, rectangles=rect_cases), f)
    print('geometry_ref_tests: segments', sum(len(c['expected']['__nd__']) for c in seg_cases),
          'rectangles', sum(len(c['expected']['__nd__']) for c in rect_cases))

    rng = np.random.default_rng(77)
    n = 6000
    # segments on a coarse lattice (many collinear / touching / shared-endpoint cases) + continuous ones
    lat = rng.integers(-3, 4, size=(n, 8)) * 0.25
    cont = rng.uniform(-1, 1, size=(n, 8))
    mix = np.where(rng.random((n, 8)) < 0.5, lat, cont)
    segs = np.concatenate([lat, cont, mix], axis=0)
    p1, p2, q1, q2 = segs[:, 0:2], segs[:, 2:4], segs[:, 4:6], segs[:, 6:8]
    seg_out = geometry_2D_utils.check_line_segments_intersect(p1=p1, p2=p2, q1=q1, q2=q2)

    m = 6000
    r1 = np.zeros((m, 7))
    r2 = np.zeros((m, 7))
    r1[:, :2] = rng.uniform(-0.3, 0.3, size=(m, 2))
    r2[:, :2] = rng.uniform(-0.3, 0.3, size=(m, 2))
    r1[:, 3:] = random_quats(rng, m)
    r2[:, 3:] = random_quats(rng, m)
    s1 = rng.choice([0.02, 0.05, 0.0775, 0.08, 0.12], size=(m, 2))
    s2 = rng.choice([0.02, 0.05, 0.0775, 0.08, 0.12], size=(m, 2))
    # axis-aligned exactly touching / nearly touching pairs
    k = m // 4
    r1[:k, 3:] = [1, 0, 0, 0]
    r2[:k, 3:] = [1, 0, 0, 0]
    r2[:k, 0] = r1[:k, 0] + (s1[:k, 0] + s2[:k, 0]) * rng.choice([-1, 1], size=k) + rng.choice([0, 1e-9, -1e-9, 1e-6, -1e-6, 1e-3], size=k)
    r2[:k, 1] = r1[:k, 1] + rng.uniform(-1, 1, size=k) * (s1[:k, 1] + s2[:k, 1])
    rect_out = geometry_2D_utils.check_rectangles_intersect(qpos_r1=r1, qpos_r2=r2, size_r1=s1, size_r2=s2)
    verts = geometry_2D_utils.get_2D_rect_vertices(qpos=r1, size=s1)
    np.savez_compressed(os.path.join(HERE, 'geometry_fuzz.npz'), segs=segs, seg_out=seg_out, r1=r1, r2=r2, s1=s1, s2=s2,
                        rect_out=rect_out, verts_r1=verts)
    print('geometry_fuzz: seg true frac', seg_out.mean(), 'rect true frac', rect_out.mean())


# ----------------------------------------------------------------------------------------------------------
# 4. check_mover_collision
# ----------------------------------------------------------------------------------------------------------
def gen_mover_collision(seed=5):
    rng = np.random.default_rng(seed)
    out = dict(M=[], shape=[], qpos=[], c_size=[], per_mover=[], offset=[], add=[], res=[])
    for M in (2, 4, 8, 16):
        for rep in range(150):
            box = bool(rep % 2)
            per_mover = bool(rng.random() < 0.3)
            offset = float(rng.choice([0.0, 0.01]))
            if box:
                base = np.array([0.0775, 0.0775]) if rng.random() < 0.5 else rng.choice([0.05, 0.08, 0.1], size=2)
                cp = {'shape': 'box', 'size': base, 'offset': offset}
                c_size = rng.choice([0.05, 0.0775, 0.1], size=(M, 2)) if per_mover else base
            else:
                base = 0.11
                cp = {'shape': 'circle', 'size': base, 'offset': offset}
                c_size = rng.choice([0.08, 0.11, 0.12], size=(M,)) if per_mover else base
            env = make_env(np.ones((10, 10)), cp, num_movers=M)
            qpos = np.zeros((M, 7))
            spread = rng.choice([0.6, 1.2, 2.2])
            qpos[:, :2] = rng.uniform(0.2, 0.2 + spread, size=(M, 2))
            qpos[:, 2] = 0.003
            qpos[:, 3:] = random_quats(rng, M) if box else np.tile([1.0, 0, 0, 0], (M, 1))
            if rng.random() < 0.4:  # put mover 1 exactly (or almost) at touching distance from mover 0 (3-4-5 triangle)
                ci = float(np.atleast_1d(c_size[0] if per_mover else c_size)[0])
                cj = float(np.atleast_1d(c_size[1] if per_mover else c_size)[0])
                add_now = offset if rep % 3 == 0 else 0.0
                d = ci + cj + 2 * add_now + rng.choice([0.0, 1e-12, -1e-12, 1e-7, -1e-7])
                qpos[1, :2] = qpos[0, :2] + np.array([0.6, 0.8]) * d
            add = bool(rep % 3 == 0)
            res = env.check_mover_collision(mover_names=env.mover_names, c_size=c_size, add_safety_offset=add, mover_qpos=qpos)
            cs = np.zeros((16, 2))
            if box:
                cs[:M] = c_size
            else:
                cs[:M, 0] = c_size
            qp = np.zeros((16, 7))
            qp[:M] = qpos
            out['M'].append(M); out['shape'].append(int(box)); out['qpos'].append(qp); out['c_size'].append(cs)
            out['per_mover'].append(per_mover); out['offset'].append(offset); out['add'].append(add); out['res'].append(bool(res))
    np.savez_compressed(os.path.join(HERE, 'mover_collision_fuzz.npz'), **{k: np.array(v) for k, v in out.items()})
    print('mover_collision_fuzz:', len(out['res']), 'collision frac', float(np.mean(out['res'])))


# ----------------------------------------------------------------------------------------------------------
# 5. ensure_max_dyn_val + closed-form trajectories
# ----------------------------------------------------------------------------------------------------------
def planning_env(**kw):
    args = dict(layout_tiles=np.ones((9, 9)), num_movers=1, show_2D_plot=False, mover_colors_2D_plot=[], std_noise=0.0,
                render_mode=None)
    args.update(kw)
    return BenchmarkPlanningEnv(**args)


def gen_clamp(seed=11):
    rng = np.random.default_rng(seed)
    env = planning_env()
    n = 4000
    cur = rng.normal(0, 1.2, size=(n, 2))
    der = rng.uniform(-10, 10, size=(n, 2))
    maxv = rng.choice([2.0, 0.01, 0.2, 10.0], size=n)
    cur[: n // 4] *= 0.01
    # exact boundary: cycle_time*der + cur == (max, 0)
    der[:50] = 0.0
    cur[:50, 0] = maxv[:50]
    cur[:50, 1] = 0.0
    vals = np.zeros((n, 2))
    ders = np.zeros((n, 2))
    for i in range(n):  # the reference's vectorised form only broadcasts for one clamped row at a time
        v, d = env.ensure_max_dyn_val(cur[i], float(maxv[i]), der[i])
        vals[i], ders[i] = v[0], d[0]
    np.savez_compressed(os.path.join(HERE, 'ensure_max_dyn_val.npz'), cur=cur, der=der, maxv=maxv, vals=vals, ders=ders,
                        cycle_time=env.cycle_time)
    print('ensure_max_dyn_val:', n)


def closed_form(env, mode, starts, deriv_arr, num_cycles, num_steps=100):
    """The manual loop of the reference's actuator tests (tests/test_benchmark_planning_env.py:71-93,185-204), per
    mover, using the reference's own ensure_max_dyn_val."""
    M = starts.shape[0]
    dt = env.cycle_time
    pos = np.zeros((num_steps, 2 * M)); vel = np.zeros((num_steps, 2 * M)); acc = np.zeros((num_steps, 2 * M))
    for m in range(M):
        p = starts[m].copy(); v = np.zeros(2); a = np.zeros(2)
        for step in range(num_steps):
            for _ in range(num_cycles):
                d = deriv_arr[2 * m: 2 * m + 2].copy()
                if mode == 'jerk':
                    next_a, _ = env.ensure_max_dyn_val(a, env.a_max, d)
                else:
                    next_a = d
                v, a_tmp = env.ensure_max_dyn_val(v, env.v_max, next_a)
                a = a_tmp.copy()
                p = dt * v + p
            pos[step, 2 * m: 2 * m + 2] = p.flatten(); vel[step, 2 * m: 2 * m + 2] = v.flatten(); acc[step, 2 * m: 2 * m + 2] = a.flatten()
    return pos, vel, acc


def gen_trajectories():
    mp = load_ref_test_module('test_benchmark_planning_env.py')
    mu = load_ref_test_module('test_benchmark_pushing_env.py')
    rec = []

    def arr(val, tx, ty, M):
        if tx and ty:
            return np.array([val / 2, val / 2] * M)
        if tx:
            return np.array([val, 0] * M)
        if ty:
            return np.array([0, val] * M)
        return np.zeros(2 * M)

    for mode, fn in (('jerk', mp.test_jerk_actuator), ('acc', mp.test_acceleration_actuator)):
        for M, mass, val, ncyc, tx, ty in param_values(fn):
            env = planning_env(num_movers=M, mover_params={'mass': mass}, num_cycles=ncyc, v_max=0.01, a_max=0.2, j_max=150,
                               learn_jerk=(mode == 'jerk'))
            starts = np.array([[1.2, 1.2]]) if M == 1 else np.array([[0.96, 0.96], [1.2, 1.2]])
            d = arr(val, tx, ty, M)
            pos, vel, acc = closed_form(env, mode, starts, d, ncyc)
            rec.append(dict(env='planning', mode=mode, M=M, val=float(val), num_cycles=int(ncyc), starts=starts, deriv=d,
                            pos=pos, vel=vel, acc=acc))
    for mode, fn in (('jerk', mu.test_jerk_actuator), ('acc', mu.test_acceleration_actuator)):
        for mass, val, ncyc, tx, ty in param_values(fn):
            # BenchmarkPushingEnv's ctor needs a live MuJoCo model (impedance controller); its ensure_max_dyn_val
            # (benchmark_pushing_env.py:614-649) only reads self.cycle_time, so call it unbound on a bare namespace.
            env = types.SimpleNamespace(cycle_time=0.001, v_max=0.01, a_max=0.2, j_max=150)
            env.ensure_max_dyn_val = lambda c, m, d, _e=env: BenchmarkPushingEnv.ensure_max_dyn_val(_e, c, m, d)
            starts = np.array([[0.48, 0.48]])
            d = arr(val, tx, ty, 1)
            pos, vel, acc = closed_form(env, mode, starts, d, ncyc)
            rec.append(dict(env='pushing', mode=mode, M=1, val=float(val), num_cycles=int(ncyc), starts=starts, deriv=d,
                            pos=pos, vel=vel, acc=acc))
    flat = {}
    for i, r in enumerate(rec):
        for k, v in r.items():
            flat[f'{i}_{k}'] = np.asarray(v)
    flat['n'] = np.array(len(rec))
    flat['v_max'] = np.array(0.01); flat['a_max'] = np.array(0.2); flat['j_max'] = np.array(150.0); flat['dt'] = np.array(0.001)
    np.savez_compressed(os.path.join(HERE, 'trajectories.npz'), **flat)
    print('trajectories:', len(rec))


# ----------------------------------------------------------------------------------------------------------
# 6. reward / terminated
# ----------------------------------------------------------------------------------------------------------
def gen_reward(seed=9):
    rng = np.random.default_rng(seed)
    out = {}
    for M in (1, 2, 4, 8):
        env = planning_env(num_movers=M, layout_tiles=np.ones((9, 9)))
        B = 600
        dg = rng.uniform(0.2, 1.9, size=(B, 2 * M))
        ag = dg + rng.normal(0, 0.08, size=(B, 2 * M))
        near = rng.random(B) < 0.4
        ag[near] = dg[near] + rng.normal(0, 0.03, size=(near.sum(), 2 * M))
        # exact threshold distance (3-4-5 scaled): 0.1 * (0.6, 0.8)
        ag[:40, 0:2] = dg[:40, 0:2] + np.array([0.06, 0.08])
        mc = rng.random(B) < 0.15
        wc = rng.random(B) < 0.15
        info = np.array([{'mover_collision': bool(a), 'wall_collision': bool(b)} for a, b in zip(mc, wc)])
        rew = env.compute_reward(ag, dg, info)
        term = env.compute_terminated(ag, dg, info)
        succ = np.array([env._get_info(bool(mc[i]), bool(wc[i]), False, ag[i], dg[i])['is_success'] for i in range(B)])
        # single-env path too
        r1 = np.array([env.compute_reward(ag[i], dg[i], info[i]) for i in range(20)])
        assert np.array_equal(r1, rew[:20])
        out.update({f'M{M}_ag': ag, f'M{M}_dg': dg, f'M{M}_mc': mc, f'M{M}_wc': wc, f'M{M}_reward': rew, f'M{M}_terminated': term,
                    f'M{M}_is_success': succ})
    out['threshold_pos'] = np.array(0.1)
    np.savez_compressed(os.path.join(HERE, 'reward_cases.npz'), **out)
    print('reward_cases done')


# ----------------------------------------------------------------------------------------------------------
# 7. ctor-time constants
# ----------------------------------------------------------------------------------------------------------
def gen_constants():
    rng = np.random.default_rng(3)
    cases = []
    layouts = [np.ones((3, 3)), np.ones((6, 6)), np.ones((8, 8)), np.ones((9, 9)), np.ones((10, 10)), np.ones((1, 1)), np.ones((2, 5)),
               np.array([[1, 1], [1, 0]]), np.array([[1, 0, 1], [1, 1, 1], [1, 1, 1]]),
               np.array([[1, 1, 1, 1], [1, 1, 1, 1], [1, 1, 0, 1], [1, 1, 1, 1], [0, 1, 1, 1]])]
    for _ in range(6):
        layouts.append(random_layout(rng, 7))
    for lay in layouts:
        for cp in ({'shape': 'circle'}, {'shape': 'circle', 'size': 0.1, 'offset': 0.01, 'offset_wall': 0.002},
                   {'shape': 'box', 'size': np.array([0.0775, 0.0775]), 'offset': 0.01}):
            lay = np.asarray(lay)
            if lay.shape[0] * lay.shape[1] < 1:
                continue
            env = BenchmarkPlanningEnv(layout_tiles=lay, num_movers=1, show_2D_plot=False, render_mode=None, collision_params=cp)
            cases.append(dict(
                layout=jsonable(lay.astype(np.int8)), collision_params=jsonable(cp),
                x_pos_tiles=jsonable(env.x_pos_tiles), y_pos_tiles=jsonable(env.y_pos_tiles),
                idx_3x3=jsonable(np.stack([env.idx_x_tiles_3x3, env.idx_y_tiles_3x3]).astype(np.int64)),
                idx_2x2_bl=jsonable(np.stack([env.idx_x_tiles_2x2_bl, env.idx_y_tiles_2x2_bl]).astype(np.int64)),
                idx_2x2_br=jsonable(np.stack([env.idx_x_tiles_2x2_br, env.idx_y_tiles_2x2_br]).astype(np.int64)),
                idx_2x2_tl=jsonable(np.stack([env.idx_x_tiles_2x2_tl, env.idx_y_tiles_2x2_tl]).astype(np.int64)),
                idx_2x2_tr=jsonable(np.stack([env.idx_x_tiles_2x2_tr, env.idx_y_tiles_2x2_tr]).astype(np.int64)),
                min_xy_pos=jsonable(np.asarray(env.min_xy_pos, dtype=np.float64)), max_xy_pos=jsonable(np.asarray(env.max_xy_pos, dtype=np.float64)),
                min_goal_dist=float(env.min_goal_dist),
                goal_space_high=jsonable(env.observation_space['desired_goal'].high),
                action_low=jsonable(env.action_space.low), action_high=jsonable(env.action_space.high),
                cycle_time=float(env.cycle_time), num_cycles=int(env.num_cycles), v_max=env.v_max, a_max=env.a_max, j_max=env.j_max,
                threshold_pos=env.threshold_pos, reward_success=env.reward_success,
                std_noise=jsonable(env.std_noise), mover_names=list(env.mover_names),
            ))
    with open(os.path.join(HERE, 'constants.json'), 'w') as f:
        json.dump(cases, f)
    print('constants:', len(cases))


# ----------------------------------------------------------------------------------------------------------
# 8. reset predicates (accept / reject decisions of _reset_callback's checks on candidate sets)
# ----------------------------------------------------------------------------------------------------------
def gen_reset_predicates(seed=21):
    rng = np.random.default_rng(seed)
    out = {}
    cfgs = [(np.ones((3, 3)), 1, {}), (np.ones((6, 6)), 4, {}), (np.ones((8, 8)), 8, {}), (np.ones((10, 10)), 16, {}),
            (np.array([[1, 1, 1, 1], [1, 1, 0, 1], [1, 1, 1, 1], [1, 0, 1, 1]]), 3, {'offset': 0.01, 'offset_wall': 0.002}),
            (np.ones((4, 4)), 4, {'shape': 'box', 'size': np.array([0.0775, 0.0775]), 'offset': 0.005})]
    for ci, (lay, M, cp) in enumerate(cfgs):
        env = BenchmarkPlanningEnv(layout_tiles=lay, num_movers=M, show_2D_plot=False, render_mode=None, collision_params=cp)
        n = 400
        cand = rng.uniform(low=env.min_xy_pos, high=env.max_xy_pos, size=(n, M, 2))
        valid = np.zeros((n, M), dtype=np.int8); coll = np.zeros(n, dtype=bool); goal_ok = np.zeros(n, dtype=bool)
        for i in range(n):
            q = np.zeros((M, 7)); q[:, 2] = env.initial_mover_zpos; q[:, 3] = 1; q[:, :2] = cand[i]
            try:
                valid[i] = env.qpos_is_valid(qpos=q, c_size=env.c_size, add_safety_offset=True)
            except AssertionError:
                valid[i] = -1
            coll[i] = env.check_mover_collision(mover_names=env.mover_names, c_size=env.c_size, add_safety_offset=True, mover_qpos=q)
            ok = True
            for a in range(M):
                for b in range(a + 1, M):
                    if np.linalg.norm(q[a, :2] - q[b, :2], ord=2) < env.min_goal_dist:
                        ok = False
            goal_ok[i] = ok
        out.update({f'{ci}_layout': np.asarray(lay, dtype=np.int8), f'{ci}_M': np.array(M), f'{ci}_cand': cand, f'{ci}_valid': valid,
                    f'{ci}_collision': coll, f'{ci}_goal_dist_ok': goal_ok, f'{ci}_min_xy': np.asarray(env.min_xy_pos, dtype=np.float64),
                    f'{ci}_max_xy': np.asarray(env.max_xy_pos, dtype=np.float64), f'{ci}_min_goal_dist': np.array(env.min_goal_dist),
                    f'{ci}_shape': np.array(int(cp.get('shape', 'circle') == 'box')),
                    f'{ci}_size': np.atleast_1d(np.asarray(cp.get('size', 0.11), dtype=np.float64)),
                    f'{ci}_offset': np.array(cp.get('offset', 0.0)), f'{ci}_offset_wall': np.array(cp.get('offset_wall', 0.0))})
    out['n_cfg'] = np.array(len(cfgs))
    np.savez_compressed(os.path.join(HERE, 'reset_predicates.npz'), **out)
    print('reset_predicates done')


# ----------------------------------------------------------------------------------------------------------
# 9. _get_obs orderings and the per-cycle control decision, from the imported class methods themselves
# ----------------------------------------------------------------------------------------------------------
def gen_obs_ordering(seed=31):
    rng = np.random.default_rng(seed)
    out = {}
    for jerk in (False, True):
        for M in (1, 3, 4):
            env = planning_env(num_movers=M, learn_jerk=jerk, layout_tiles=np.ones((6, 6)))
            pos = rng.uniform(0.2, 1.2, size=(M, 7)); vel = rng.uniform(-2, 2, size=(M, 6)); acc = rng.uniform(-10, 10, size=(M, 6))
            goals = rng.uniform(0.2, 1.2, size=(M, 2))
            env.get_mover_qpos_arr = lambda mover_names, add_noise=False, _a=pos: _a.copy()
            env.get_mover_qvel_arr = lambda mover_names, add_noise=False, _a=vel: _a.copy()
            env.get_mover_qacc_arr = lambda mover_names, add_noise=False, _a=acc: _a.copy()
            env.goals = goals.copy()
            o = BenchmarkPlanningEnv._get_obs(env)
            tag = f'{"jerk" if jerk else "acc"}_M{M}'
            out[tag + '_pos'] = pos[:, :2]; out[tag + '_vel'] = vel[:, :2]; out[tag + '_acc'] = acc[:, :2]; out[tag + '_goals'] = goals
            for k in ('observation', 'achieved_goal', 'desired_goal'):
                out[tag + '_' + k] = np.asarray(o[k])
    np.savez_compressed(os.path.join(HERE, 'obs_ordering.npz'), **out)
    print('obs_ordering.npz:', len(out), 'arrays')


def gen_step_callback(seed=41, n=400):
    import gymnasium_planar_robotics.envs.planning.benchmark_planning_env as PLMOD
    rng = np.random.default_rng(seed)
    out = {}
    for jerk in (False, True):
        M = 2
        env = planning_env(num_movers=M, learn_jerk=jerk, layout_tiles=np.ones((6, 6)), v_max=2.0, a_max=10.0, j_max=100.0)
        lim = 100.0 if jerk else 10.0
        # speeds / accelerations from "far inside" to "at / beyond the limit" so that every branch of PL:432-443 is taken
        def polar(mags):
            ang = rng.uniform(0, 2 * np.pi, size=(n, M))
            r = rng.choice(mags, size=(n, M)) * rng.uniform(0.97, 1.03, size=(n, M))
            return np.stack([r * np.cos(ang), r * np.sin(ang)], axis=-1)

        vel = polar([0.3, 1.2, 1.95, 2.0, 2.0, 2.05])
        acc = polar([1.0, 6.0, 9.8, 10.0, 10.2])
        act = rng.uniform(-lim, lim, size=(n, M * 2))
        ctrl = np.zeros((n, M, 2))
        captured = {}

        def fake_set(model, data, actuator_name, value, _c=captured):
            _c[actuator_name] = float(value)

        orig = PLMOD.mujoco_utils.set_actuator_ctrl
        PLMOD.mujoco_utils.set_actuator_ctrl = fake_set
        try:
            for i in range(n):
                env.get_mover_qvel = lambda mover_name, add_noise=False, _i=i: np.concatenate([vel[_i, env.mover_names.index(mover_name)], np.zeros(4)])
                env.get_mover_qacc = lambda mover_name, add_noise=False, _i=i: np.concatenate([acc[_i, env.mover_names.index(mover_name)], np.zeros(4)])
                captured.clear()
                BenchmarkPlanningEnv._mujoco_step_callback(env, act[i].copy())
                for m in range(M):
                    ctrl[i, m, 0] = captured[env.mover_actuator_x_names[m]]
                    ctrl[i, m, 1] = captured[env.mover_actuator_y_names[m]]
        finally:
            PLMOD.mujoco_utils.set_actuator_ctrl = orig
        tag = 'jerk' if jerk else 'acc'
        out[tag + '_vel'] = vel; out[tag + '_acc'] = acc; out[tag + '_action'] = act; out[tag + '_ctrl'] = ctrl
        changed = np.abs(ctrl.reshape(n, -1) - act).max(axis=1) > 1e-12
        print(f'step_callback {tag}: {int(changed.sum())} of {n} cases leave the action changed')
    out['dt'] = np.array(0.001); out['v_max'] = np.array(2.0); out['a_max'] = np.array(10.0); out['j_max'] = np.array(100.0)
    np.savez_compressed(os.path.join(HERE, 'step_callback.npz'), **out)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'round2':  # only the fixtures added in round 2 (the others are unchanged)
        gen_obs_ordering()
        gen_step_callback()
        sys.exit(0)
    gen_qpos_ref_tests()
    gen_qpos_fuzz()
    gen_geometry()
    gen_mover_collision()
    gen_clamp()
    gen_trajectories()
    gen_reward()
    gen_constants()
    gen_reset_predicates()
    gen_obs_ordering()
    gen_step_callback()
