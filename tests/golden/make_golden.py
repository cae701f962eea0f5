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

  step_template.npz              BasicPlanarRoboticsSingleAgentEnv.step (BE:1835-1950) called AS WRITTEN on BenchmarkPlanningEnv objects, sigma = 0,
                                 with the MuJoCo calls replaced by a stand-in "world" (see World below): pins the TEMPLATE's control flow --
                                 clip, the cycle loop with its break, the flags / state / observation of the colliding cycle, reward,
                                 terminated, info -- NOT mj_step itself (a4: that is trajectories.npz)
  reset_template.npz             BasicPlanarRoboticsSingleAgentEnv.reset (BE:1770-1833) + BenchmarkPlanningEnv._reset_callback (PL:355-418) called
                                 AS WRITTEN with np_random.uniform scripted to hand out a prescribed candidate sequence: "the first valid
                                 set in order wins; starts, then goals; goals are not checked against starts" (row a14)
  pushing_methods.npz            BenchmarkPushingEnv._get_obs (PU:529-576), compute_reward / compute_terminated (PU:457-527) on arrays of info
                                 dicts, _reset_callback (PU:373-417) incl. the `options`, with scripted np_random.uniform -- as written (a16)

  getters.npz                    get_mover_qpos / get_mover_qvel / get_mover_qacc and the _arr forms (BE:801-855, 1244-1290) as written on the
                                 stand-in world: z convention, row order, and the additive noise on every component (unit normals scripted)

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


# ----------------------------------------------------------------------------------------------------------
# 10. the step() / reset() TEMPLATES as written, on a stand-in world (round 3)
# ----------------------------------------------------------------------------------------------------------
class World:
    """What the reference's methods see instead of MuJoCo's model / data while the template fixtures are taken.

    It holds the free-joint state of M movers (+ optionally one object) and answers exactly the calls the methods under test make:
    mujoco_utils.get_joint_qpos / qvel / qacc, mujoco_utils.set_actuator_ctrl, mujoco.mj_step, mujoco.mj_forward. Its mj_step is the
    closed form the reference's own tests assert MuJoCo equals for this MJCF (tests/test_benchmark_planning_env.py:86-121,199-231):
    qacc = ctrl (jerk mode: act += dt * ctrl, qacc = act), v += dt * qacc, p += dt * v. It is a STUB: it pins which calls the
    templates make in which order and what they do with the answers (rows a1 / a14), not MuJoCo's integrator (row a4)."""

    def __init__(self, env, jerk):
        self.env, self.jerk, self.M = env, jerk, env.num_movers
        self.dt = 0.001
        self.load(np.zeros((self.M, 2)))
        self.object_qpos = np.array([0.12, 0.36, 0.02, 1.0, 0.0, 0.0, 0.0])

    def load(self, start_xy):  # what compiling the MJCF of reload_model yields: bodies at their start poses, at rest (BE:1069)
        M = self.M
        self.qpos = np.zeros((M, 7)); self.qvel = np.zeros((M, 6)); self.qacc = np.zeros((M, 6)); self.act = np.zeros((M, 2))
        self.qpos[:, :2] = np.asarray(start_xy, dtype=np.float64).reshape(M, 2)
        self.qpos[:, 2] = self.env.initial_mover_zpos + self.env.mover_size[:, 2]
        self.qpos[:, 3] = 1.0
        self.ctrl = {}

    def _idx(self, joint_name):
        return self.env.mover_joint_names.index(joint_name)

    def install(self):
        import mujoco
        from gymnasium_planar_robotics.utils import mujoco_utils
        self._saved = (mujoco_utils.get_joint_qpos, mujoco_utils.get_joint_qvel, mujoco_utils.get_joint_qacc, mujoco_utils.set_actuator_ctrl,
                       mujoco.mj_step, mujoco.mj_forward)
        obj = getattr(self.env, 'object_joint_name', None)
        mujoco_utils.get_joint_qpos = lambda model, data, name: (self.object_qpos.copy() if name == obj else self.qpos[self._idx(name)].copy())
        mujoco_utils.get_joint_qvel = lambda model, data, name: self.qvel[self._idx(name)].copy()
        mujoco_utils.get_joint_qacc = lambda model, data, name: self.qacc[self._idx(name)].copy()
        mujoco_utils.set_actuator_ctrl = lambda model, data, actuator_name, value: self.ctrl.__setitem__(actuator_name, float(value))
        mujoco.mj_step = lambda model, data, nstep=1: self.mj_step()
        mujoco.mj_forward = lambda model, data: None
        orig_reload = self.env.reload_model

        def reload_and_load(mover_start_xy_pos=None, mover_goal_xy_pos=None, **kw):
            if mover_goal_xy_pos is None:
                orig_reload(mover_start_xy_pos=mover_start_xy_pos, **kw)
            else:
                orig_reload(mover_start_xy_pos=mover_start_xy_pos, mover_goal_xy_pos=mover_goal_xy_pos, **kw)
            self.load(mover_start_xy_pos)

        self.env.reload_model = reload_and_load
        return self

    def uninstall(self):
        import mujoco
        from gymnasium_planar_robotics.utils import mujoco_utils
        (mujoco_utils.get_joint_qpos, mujoco_utils.get_joint_qvel, mujoco_utils.get_joint_qacc, mujoco_utils.set_actuator_ctrl, mujoco.mj_step,
         mujoco.mj_forward) = self._saved

    def mj_step(self):
        dt, env = self.dt, self.env
        for i in range(self.M):
            u = np.array([self.ctrl[env.mover_actuator_x_names[i]], self.ctrl[env.mover_actuator_y_names[i]]])
            if self.jerk:
                self.act[i] = self.act[i] + dt * u
                qacc = self.act[i].copy()
            else:
                qacc = u
            self.qacc[i, :2] = qacc
            self.qvel[i, :2] = self.qvel[i, :2] + dt * qacc
            self.qpos[i, :2] = self.qpos[i, :2] + dt * self.qvel[i, :2]


def reference_valid_sets(env, world, seed0):
    """(starts, goals) accepted by the reference's own unscripted _reset_callback. On layouts with holes qpos_is_valid can trip its own
    sanity assertion (BE:650, mask_valid > 1 for some positions near two missing tiles): such seeds are skipped."""
    for r in range(seed0, seed0 + 1000):
        env.np_random = np.random.default_rng(r)
        try:
            BenchmarkPlanningEnv._reset_callback(env, None)
        except AssertionError:
            continue
        return world.qpos[:, :2].copy(), env.goals.copy()
    raise RuntimeError('no assertion-free reset found')


def reference_accepts_without_assert(env, xy):
    q = np.zeros((env.num_movers, 7)); q[:, 2] = env.initial_mover_zpos; q[:, 3] = 1; q[:, :2] = xy
    try:
        env.qpos_is_valid(qpos=q, c_size=env.c_size, add_safety_offset=True)
        return True
    except AssertionError:
        return False


class ScriptedUniform:
    """np_random stand-in whose uniform() hands out a prescribed sequence of samples and remembers how it was asked"""

    def __init__(self, samples):
        self.samples, self.calls = [np.asarray(x, dtype=np.float64) for x in samples], []

    def uniform(self, low, high, size=None):
        self.calls.append((np.array(low, dtype=np.float64), np.array(high, dtype=np.float64), tuple(np.atleast_1d(size))))
        x = self.samples[len(self.calls) - 1]
        assert x.shape == tuple(np.atleast_1d(size)), (x.shape, size)
        return x.copy()


STEP_TEMPLATE_CONFIGS = [
    # tag, layout, M, jerk, collision_params, num_cycles
    ('acc_M1', np.ones((3, 3)), 1, False, {}, 40),
    ('acc_M2', np.ones((4, 4)), 2, False, {}, 40),
    ('acc_M4', np.ones((6, 6)), 4, False, {}, 40),
    ('jerk_M2', np.ones((4, 4)), 2, True, {}, 40),
    ('acc_M3_holes', np.array([[1, 1, 1, 1], [1, 1, 0, 1], [1, 1, 1, 1], [1, 0, 1, 1]]), 3, False, {'offset': 0.01, 'offset_wall': 0.002}, 25),
    ('acc_M2_box', np.ones((4, 4)), 2, False, {'shape': 'box', 'size': np.array([0.0775, 0.0775])}, 40),
]


def gen_step_template(seed=51, E=48, T=6):
    """E scenarios x T consecutive step() calls per configuration. Scenarios are built so that every exit of the template is taken:
    a wall collision in some cycle k (start near a wall, moving towards it), a mover collision in some cycle (two movers on a
    collision course), all goals reached (+50), some reached (-(M - n)), none, and actions outside the action space (clipped)."""
    from gymnasium_planar_robotics.envs.basic_envs import BasicPlanarRoboticsSingleAgentEnv
    rng = np.random.default_rng(seed)
    out = {'tags': np.array([c[0] for c in STEP_TEMPLATE_CONFIGS])}
    for tag, layout, M, jerk, cp, ncyc in STEP_TEMPLATE_CONFIGS:
        env = planning_env(layout_tiles=layout, num_movers=M, learn_jerk=jerk, collision_params=cp, num_cycles=ncyc, std_noise=0.0)
        world = World(env, jerk).install()
        lim = env.j_max if jerk else env.a_max
        od = 4 * M if jerk else 2 * M
        starts = np.zeros((E, M, 2)); goals = np.zeros((E, M, 2)); v0 = np.zeros((E, M, 2)); a0 = np.zeros((E, M, 2)); acts = np.zeros((T, E, 2 * M))
        res = dict(observation=np.zeros((T, E, od)), achieved_goal=np.zeros((T, E, 2 * M)), desired_goal=np.zeros((T, E, 2 * M)),
                   reward=np.zeros((T, E)), terminated=np.zeros((T, E), dtype=bool), truncated=np.zeros((T, E), dtype=bool),
                   is_success=np.zeros((T, E), dtype=bool), mover_collision=np.zeros((T, E), dtype=bool), wall_collision=np.zeros((T, E), dtype=bool),
                   pos=np.zeros((T, E, 2 * M)), vel=np.zeros((T, E, 2 * M)), acc=np.zeros((T, E, 2 * M)), valid=np.zeros((T, E), dtype=bool))
        try:
            for e in range(E):
                # a valid, collision-free start set drawn by the reference's own reset loop (scripted uniform not needed here)
                st, gl = reference_valid_sets(env, world, 1000 * (1000 * seed + e))
                kind = e % 6
                env.goals = gl.copy()
                vv = np.zeros((M, 2)); aa = np.zeros((M, 2))
                act = rng.uniform(-lim, lim, size=(T, 2 * M))
                if kind == 0:    # heading for a wall: full speed towards the nearest outer edge
                    hi = np.array([layout.shape[0], layout.shape[1]]) * 0.24
                    for i in range(M):
                        d = np.argmin(np.concatenate([st[i], hi - st[i]]))
                        dirv = np.array([[-1, 0], [0, -1], [1, 0], [0, 1]][d], dtype=np.float64)
                        vv[i] = dirv * rng.uniform(1.0, 2.0)
                        act[:, 2 * i:2 * i + 2] = dirv * lim * rng.uniform(0.3, 1.0)
                elif kind == 1 and M > 1:  # movers 0 and 1 on a collision course
                    d = st[1] - st[0]; n = d / np.linalg.norm(d)
                    vv[0] = n * 1.9; vv[1] = -n * 1.9
                    act[:, 0:2] = n * lim * 0.5; act[:, 2:4] = -n * lim * 0.5
                elif kind == 2:  # every mover starts inside its goal radius, at rest, tiny actions: +50 at once (goals unchecked against starts)
                    gl = st + rng.uniform(-0.05, 0.05, size=(M, 2))
                    act = rng.uniform(-0.02, 0.02, size=(T, 2 * M)) * lim
                elif kind == 3:  # some reached, some not
                    gl = st + rng.uniform(-0.05, 0.05, size=(M, 2))
                    far = rng.integers(0, M)
                    gl[far] = env.goals[far] if np.linalg.norm(env.goals[far] - st[far]) > 0.2 else st[far] + np.array([0.3, 0.0]) * (1 if st[far, 0] < 0.4 else -1)
                    act = rng.uniform(-0.02, 0.02, size=(T, 2 * M)) * lim
                elif kind == 4:  # actions far outside the action space (BE:1871-1873: clipped), speed clamp active
                    act = rng.uniform(-5 * lim, 5 * lim, size=(T, 2 * M))
                    vv = rng.uniform(-1.3, 1.3, size=(M, 2))
                else:            # free flight with some initial velocity
                    vv = rng.uniform(-0.8, 0.8, size=(M, 2))
                if jerk:
                    aa = rng.uniform(-6, 6, size=(M, 2))
                starts[e], goals[e], v0[e], a0[e], acts[:, e] = st, gl, vv, aa, act
                world.load(st); world.qvel[:, :2] = vv; world.qacc[:, :2] = aa; world.act[:] = aa
                env.goals = gl.copy()
                for t in range(T):
                    try:
                        o, r, term, trunc, info = BasicPlanarRoboticsSingleAgentEnv.step(env, act[t].copy())
                    except AssertionError:
                        break  # a mover above no tile at all: the reference asserts (BE:514); the engine reports a wall collision
                    res['valid'][t, e] = True
                    res['observation'][t, e] = o['observation']; res['achieved_goal'][t, e] = o['achieved_goal']; res['desired_goal'][t, e] = o['desired_goal']
                    res['reward'][t, e] = r; res['terminated'][t, e] = bool(term); res['truncated'][t, e] = bool(trunc)
                    res['is_success'][t, e] = bool(info['is_success']); res['mover_collision'][t, e] = bool(info['mover_collision'])
                    res['wall_collision'][t, e] = bool(info['wall_collision'])
                    res['pos'][t, e] = world.qpos[:, :2].reshape(-1); res['vel'][t, e] = world.qvel[:, :2].reshape(-1); res['acc'][t, e] = world.qacc[:, :2].reshape(-1)
        finally:
            world.uninstall()
        v = res['valid']
        print(f'step_template {tag}: {int(v.sum())} valid env-steps, wall {int(res["wall_collision"][v].sum())}, pair {int(res["mover_collision"][v].sum())}, '
              f'+50 {int((res["reward"][v] == 50).sum())}, partial {int(((res["reward"][v] > -M) & (res["reward"][v] < 0)).sum())}, '
              f'none {int((res["reward"][v] == -M).sum())}')
        out[tag + '_layout'] = np.asarray(layout, dtype=np.int8); out[tag + '_M'] = np.array(M); out[tag + '_jerk'] = np.array(jerk)
        out[tag + '_num_cycles'] = np.array(ncyc); out[tag + '_shape'] = np.array(int(cp.get('shape', 'circle') == 'box'))
        out[tag + '_size'] = np.atleast_1d(np.asarray(cp.get('size', 0.11), dtype=np.float64))
        out[tag + '_offset'] = np.array(cp.get('offset', 0.0)); out[tag + '_offset_wall'] = np.array(cp.get('offset_wall', 0.0))
        out[tag + '_starts'] = starts; out[tag + '_goals'] = goals; out[tag + '_v0'] = v0; out[tag + '_a0'] = a0; out[tag + '_actions'] = acts
        for k, a in res.items():
            out[f'{tag}_{k}'] = a
    np.savez_compressed(os.path.join(HERE, 'step_template.npz'), **out)


RESET_TEMPLATE_CONFIGS = [
    ('M4', np.ones((6, 6)), 4, {}),
    ('M8', np.ones((5, 5)), 8, {}),
    ('M3_holes', np.array([[1, 1, 1, 1], [1, 1, 0, 1], [1, 1, 1, 1], [1, 0, 1, 1]]), 3, {'offset': 0.01, 'offset_wall': 0.002}),
    ('M2_box', np.ones((3, 3)), 2, {'shape': 'box', 'size': np.array([0.0775, 0.0775]), 'offset': 0.005}),
]


def gen_reset_template(seed=61, E=40, K=32):
    """E scripted resets per configuration. reset() is called as written with np_random.uniform scripted to hand out ONE prescribed
    stream of K candidate sets (the start loop consumes from it until it accepts, the goal loop goes on from there -- exactly how the
    two loops share np_random); the fixture holds the stream and what reset() made of it: how many sets it consumed, the accepted
    starts and goals, the initial observation and info. Candidate sets include positions outside the sampling box (a scripted sampler
    can return them; the loops must reject them through qpos_is_valid), overlapping movers, goals closer than min_goal_dist, and a goal
    set that equals the accepted start set (accepted: goals are not checked against starts). A set known to be valid sits at a random
    place of the stream's first half and valid goal sets fill its tail, so both loops end inside the script."""
    from gymnasium_planar_robotics.envs.basic_envs import BasicPlanarRoboticsSingleAgentEnv
    rng = np.random.default_rng(seed)
    out = {'tags': np.array([c[0] for c in RESET_TEMPLATE_CONFIGS])}
    for tag, layout, M, cp in RESET_TEMPLATE_CONFIGS:
        env = planning_env(layout_tiles=layout, num_movers=M, collision_params=cp, std_noise=0.0)
        world = World(env, False).install()
        cand = np.zeros((E, K, M, 2)); n_used = np.zeros(E, dtype=np.int32)
        st = np.zeros((E, 2 * M)); gl = np.zeros((E, 2 * M))
        obs = np.zeros((E, 2 * M)); ag = np.zeros((E, 2 * M)); dg = np.zeros((E, 2 * M))
        succ = np.zeros(E, dtype=bool); mc = np.zeros(E, dtype=bool); wc = np.zeros(E, dtype=bool)
        lows = np.zeros((E, 2)); highs = np.zeros((E, 2))
        try:
            for e in range(E):
                good_s, good_g = reference_valid_sets(env, world, 1000 * (7000 + 100 * seed + e))  # from the reference's own, unscripted loop
                span = env.max_xy_pos - env.min_xy_pos
                for k in range(K):  # squeezed into a corner of the sampling box (many overlaps) or spread out; some leave the box
                    cand[e, k] = env.min_xy_pos + rng.uniform(size=(M, 2)) * span * rng.choice([0.3, 0.6, 1.0, 1.0])
                    if rng.uniform() < 0.2:
                        i = rng.integers(0, M)
                        cand[e, k, i] = np.clip(cand[e, k, i] - rng.choice([[0.3, 0.0], [0.0, 0.3]]), 0.001, None)  # (still above a cell: BE:514 asserts otherwise)
                    while not reference_accepts_without_assert(env, cand[e, k]):  # (BE:650 can fire near two missing tiles: not a case to pin)
                        cand[e, k] = env.min_xy_pos + rng.uniform(size=(M, 2)) * span
                ks = 0 if e % 5 == 0 else int(rng.integers(0, K // 2))
                cand[e, ks] = good_s
                if e % 5 == 0:
                    cand[e, 1] = good_s  # the goal loop's first candidate is the accepted start set itself
                cand[e, K - K // 4:] = good_g
                script = ScriptedUniform(list(cand[e]))
                env.np_random = script
                # reset(seed=None): gym.Env.reset leaves np_random alone; the template then calls _reset_callback, mj_forward, the two
                # initial checks, _get_obs, _get_info -- all as written
                o, info = BasicPlanarRoboticsSingleAgentEnv.reset(env, seed=None, options=None)
                n_used[e] = len(script.calls)
                st[e] = world.qpos[:, :2].reshape(-1); gl[e] = env.goals.reshape(-1)
                obs[e], ag[e], dg[e] = o['observation'], o['achieved_goal'], o['desired_goal']
                succ[e], mc[e], wc[e] = bool(info['is_success']), bool(info['mover_collision']), bool(info['wall_collision'])
                lows[e], highs[e] = script.calls[0][0], script.calls[0][1]
                assert all(c[2] == (M, 2) and np.array_equal(c[0], lows[e]) and np.array_equal(c[1], highs[e]) for c in script.calls)
                assert np.array_equal(gl[e], cand[e, n_used[e] - 1].reshape(-1))
        finally:
            world.uninstall()
        print(f'reset_template {tag}: sets consumed {n_used.min()}..{n_used.max()} (mean {n_used.mean():.1f}), goals == starts accepted in '
              f'{int(np.sum(np.all(gl == st, axis=1)))} resets, initial flags set in {int((mc | wc).sum())}')
        out[tag + '_layout'] = np.asarray(layout, dtype=np.int8); out[tag + '_M'] = np.array(M)
        out[tag + '_shape'] = np.array(int(cp.get('shape', 'circle') == 'box')); out[tag + '_size'] = np.atleast_1d(np.asarray(cp.get('size', 0.11), dtype=np.float64))
        out[tag + '_offset'] = np.array(cp.get('offset', 0.0)); out[tag + '_offset_wall'] = np.array(cp.get('offset_wall', 0.0))
        out[tag + '_cand'] = cand; out[tag + '_n_used'] = n_used
        out[tag + '_starts'] = st; out[tag + '_goals'] = gl; out[tag + '_observation'] = obs; out[tag + '_achieved_goal'] = ag; out[tag + '_desired_goal'] = dg
        out[tag + '_is_success'] = succ; out[tag + '_mover_collision'] = mc; out[tag + '_wall_collision'] = wc
        out[tag + '_uniform_low'] = lows; out[tag + '_uniform_high'] = highs
    np.savez_compressed(os.path.join(HERE, 'reset_template.npz'), **out)


PUSH_KW = dict(render_mode=None, std_noise=0.0)


def gen_pushing_methods(seed=71):
    """BenchmarkPushingEnv's own methods as written (row a16; everything except MuJoCo's contact solve and the impedance hold):
    _get_obs (acc + jerk), compute_reward / compute_terminated on arrays of info dicts incl. dist == threshold, and
    reset() + _reset_callback with a scripted np_random.uniform, with and without the options."""
    from gymnasium_planar_robotics.envs.basic_envs import BasicPlanarRoboticsSingleAgentEnv
    rng = np.random.default_rng(seed)
    out = {}
    # ---- _get_obs ----
    for jerk in (False, True):
        env = BenchmarkPushingEnv(learn_jerk=jerk, **PUSH_KW)
        env.object_noise_xy_pos = 0.0
        world = World(env, jerk).install()
        n = 40
        pos = rng.uniform(0.12, 0.6, size=(n, 2)); vel = rng.uniform(-2, 2, size=(n, 2)); acc = rng.uniform(-10, 10, size=(n, 2))
        obj = rng.uniform(0.2, 0.5, size=(n, 2)); goal = rng.uniform(0.2, 0.5, size=(n, 2))
        obs = np.zeros((n, 6 if jerk else 4)); ag = np.zeros((n, 2)); dg = np.zeros((n, 2))
        try:
            for i in range(n):
                world.qpos[0, :2] = pos[i]; world.qvel[0, :2] = vel[i]; world.qacc[0, :2] = acc[i]; world.object_qpos[:2] = obj[i]
                env.object_xy_goal_pos = goal[i].copy()
                o = BenchmarkPushingEnv._get_obs(env)
                obs[i], ag[i], dg[i] = o['observation'], o['achieved_goal'], o['desired_goal']
        finally:
            world.uninstall()
        tag = 'obs_jerk' if jerk else 'obs_acc'
        out.update({tag + '_pos': pos, tag + '_vel': vel, tag + '_acc': acc, tag + '_object': obj, tag + '_goal': goal,
                    tag + '_observation': obs, tag + '_achieved_goal': ag, tag + '_desired_goal': dg})
    # ---- compute_reward / compute_terminated ----
    env = BenchmarkPushingEnv(**PUSH_KW)
    n = 600
    ag = rng.uniform(0.2, 0.5, size=(n, 2)); dg = ag + rng.uniform(-0.08, 0.08, size=(n, 2))
    ang = rng.uniform(0, 2 * np.pi, size=100)
    dg[:100] = ag[:100] + env.threshold_pos * np.stack([np.cos(ang), np.sin(ang)], axis=1)  # dist == threshold up to rounding: either side
    dg[100:120] = ag[100:120] + np.array([env.threshold_pos, 0.0])
    wcol = rng.uniform(size=n) < 0.3
    infos = np.array([{'is_success': False, 'mover_collision': False, 'wall_collision': bool(w)} for w in wcol])
    rew = np.asarray(env.compute_reward(achieved_goal=ag, desired_goal=dg, info=infos), dtype=np.float64)
    term = np.asarray(env.compute_terminated(achieved_goal=ag, desired_goal=dg, info=infos), dtype=bool)
    r1 = np.array([env.compute_reward(achieved_goal=ag[i], desired_goal=dg[i], info=infos[i]) for i in range(20)])
    assert np.array_equal(r1, rew[:20])
    out.update({'reward_ag': ag, 'reward_dg': dg, 'reward_wc': wcol, 'reward_reward': rew, 'reward_terminated': term,
                'reward_threshold_pos': np.array(env.threshold_pos)})
    # ---- reset() + _reset_callback, scripted ----
    E, K = 60, 24
    env = BenchmarkPushingEnv(**PUSH_KW)
    env.object_noise_xy_pos = 0.0
    world = World(env, False).install()
    mover_draw = np.zeros((E, 2)); obj_cand = np.zeros((E, K, 2)); goal_draw = np.zeros((E, 2))
    opt_start = np.full((E, 2), np.nan); opt_goal = np.full((E, 2), np.nan)
    n_obj = np.zeros(E, dtype=np.int32)
    mstart = np.zeros((E, 2)); ostart = np.zeros((E, 2)); ogoal = np.zeros((E, 2)); obs = np.zeros((E, 4)); ag = np.zeros((E, 2)); dg = np.zeros((E, 2))
    succ = np.zeros(E, dtype=bool); wc = np.zeros(E, dtype=bool)
    bounds = {}
    try:
        for e in range(E):
            mover_draw[e] = rng.uniform(env.min_xy_pos, env.max_xy_pos)
            goal_draw[e] = rng.uniform(env.object_min_xy_pos, env.object_max_xy_pos)
            options = {}
            if e % 4 == 1 or e % 4 == 3:
                opt_start[e] = rng.uniform(env.min_xy_pos, env.max_xy_pos); options['mover_start_xy_pos'] = opt_start[e].copy()
            if e % 4 == 2 or e % 4 == 3:
                opt_goal[e] = rng.uniform(env.object_min_xy_pos, env.object_max_xy_pos); options['object_goal_xy_pos'] = opt_goal[e].copy()
            mv = opt_start[e] if 'mover_start_xy_pos' in options else mover_draw[e]
            # object candidates: some within min_mo_dist of the mover (rejected, PU:407: strictly greater), one exactly at a distance
            # computed as min_mo_dist (either side of the rounding), the last ones far away
            for k in range(K):
                obj_cand[e, k] = rng.uniform(env.object_min_xy_pos, env.object_max_xy_pos)
            near = mv + rng.uniform(-0.08, 0.08, size=(3, 2))
            obj_cand[e, :3] = near
            a = rng.uniform(0, 2 * np.pi)
            obj_cand[e, 3] = mv + env.min_mo_dist * np.array([np.cos(a), np.sin(a)])
            far = np.where(mv < 0.33, env.object_max_xy_pos, env.object_min_xy_pos)
            obj_cand[e, K - 4:] = far
            script = ScriptedUniform([mover_draw[e].reshape(1, 2)] + [c.reshape(1, 2) for c in obj_cand[e]])
            # the goal draw comes after the object loop: the script hands it out as "the next sample", whatever the loop consumed

            class Stream:
                def __init__(self):
                    self.calls, self.n_obj = [], 0

                def uniform(self, low, high, size=None):
                    size_t = tuple(np.atleast_1d(size))
                    self.calls.append((np.array(low, dtype=np.float64), np.array(high, dtype=np.float64), size_t))
                    if len(self.calls) == 1:
                        return mover_draw[e].reshape(size_t).copy()        # PU:389-391 (drawn even when the option is given)
                    if size_t == (2,):
                        return goal_draw[e].copy()                          # PU:412-414 (drawn even when the option is given)
                    self.n_obj += 1
                    return obj_cand[e, self.n_obj - 1].reshape(size_t).copy()  # PU:403-405

            stream = Stream()
            env.np_random = stream
            o, info = BasicPlanarRoboticsSingleAgentEnv.reset(env, seed=None, options=options if options else None)
            n_obj[e] = stream.n_obj
            mstart[e] = world.qpos[0, :2]; ostart[e] = np.asarray(env.object_xy_start_pos).reshape(-1); ogoal[e] = np.asarray(env.object_xy_goal_pos).reshape(-1)
            world.object_qpos[:2] = ostart[e]  # (what compiling reload_model's MJCF yields for the object body)
            o = BenchmarkPushingEnv._get_obs(env)
            obs[e], ag[e], dg[e] = o['observation'], o['achieved_goal'], o['desired_goal']
            succ[e], wc[e] = bool(info['is_success']), bool(info['wall_collision'])
            assert not info['mover_collision']
            bounds = {'mover_low': stream.calls[0][0], 'mover_high': stream.calls[0][1], 'object_low': stream.calls[1][0], 'object_high': stream.calls[1][1]}
            assert stream.calls[-1][2] == (2,) and len(stream.calls) == 2 + n_obj[e]
    finally:
        world.uninstall()
    print(f'pushing reset: object trials {n_obj.min()}..{n_obj.max()} (mean {n_obj.mean():.1f}); initial wall flags {int(wc.sum())}')
    out.update({'reset_mover_draw': mover_draw, 'reset_object_cand': obj_cand, 'reset_goal_draw': goal_draw, 'reset_opt_start': opt_start,
                'reset_opt_goal': opt_goal, 'reset_n_object': n_obj, 'reset_mover_start': mstart, 'reset_object_start': ostart,
                'reset_object_goal': ogoal, 'reset_observation': obs, 'reset_achieved_goal': ag, 'reset_desired_goal': dg,
                'reset_wall_collision': wc, 'reset_min_mo_dist': np.array(env.min_mo_dist),
                **{'reset_' + k: v for k, v in bounds.items()}})
    np.savez_compressed(os.path.join(HERE, 'pushing_methods.npz'), **out)


# ----------------------------------------------------------------------------------------------------------
# 12. the public state getters as written (round 4): get_mover_qpos / qvel / qacc and the _arr forms (BE:801-855, 1244-1290)
# ----------------------------------------------------------------------------------------------------------
class ScriptedNormal:
    """rng_noise stand-in: normal(loc, scale, size) = loc + scale * the next `size` prescribed unit normals (records every call)"""

    def __init__(self, units):
        self.units, self.k, self.calls = np.asarray(units, dtype=np.float64).reshape(-1), 0, []

    def normal(self, loc=0.0, scale=1.0, size=None):
        n = int(np.prod(size)) if size is not None else 1
        u = self.units[self.k:self.k + n]
        self.k += n
        self.calls.append((float(scale), n))
        return loc + scale * u.reshape(size if size is not None else ())


def gen_getters(seed=81):
    """The reference's getters called as written on the stand-in world: the z convention (body z minus the mover's half height, box index 2 /
    cylinder index 1, BE:821-824), the row order of the _arr forms (= the order of the names passed), and what add_noise adds (std_noise[0] /
    [1] / [2] times a unit normal on EVERY component of qpos / qvel / qacc, BE:828, 841, 855) -- the unit normals are scripted."""
    rng = np.random.default_rng(seed)
    out = {}
    cases = [('box4', dict(num_movers=4), None), ('box3_sizes', dict(num_movers=3, mover_params={'size': np.array([[0.07, 0.07, 0.006], [0.05, 0.06, 0.004], [0.0775, 0.0775, 0.01]])}), None),
             ('cyl2', dict(num_movers=2, mover_params={'shape': 'cylinder', 'size': np.array([0.07, 0.005, 0.001])}), None)]
    for tag, kw, _ in cases:
        env = planning_env(layout_tiles=np.ones((4, 4)), std_noise=np.array([1e-3, 2e-3, 3e-3]), **kw)
        M = env.num_movers
        world = World(env, False).install()
        try:
            xy = rng.uniform(0.2, 0.7, size=(M, 2))
            world.load(xy)
            if env.mover_shape == 'cylinder':  # (World.load puts box movers at their body height; a cylinder's height is size[1], BE:1070)
                world.qpos[:, 2] = env.initial_mover_zpos + env.mover_size[:, 1]
            world.qvel[:, :2] = rng.uniform(-1, 1, size=(M, 2)); world.qacc[:, :2] = rng.uniform(-5, 5, size=(M, 2))
            order = list(rng.permutation(M))
            names = [env.mover_names[i] for i in order]
            res = {}
            for which, fn_arr, fn_one, nc in ((0, env.get_mover_qpos_arr, env.get_mover_qpos, 7), (1, env.get_mover_qvel_arr, env.get_mover_qvel, 6),
                                              (2, env.get_mover_qacc_arr, env.get_mover_qacc, 6)):
                env.rng_noise = ScriptedNormal(np.zeros(1000))
                clean = fn_arr(names, add_noise=False)
                units = rng.standard_normal((M, nc))
                env.rng_noise = sn = ScriptedNormal(units)
                noisy = fn_arr(names, add_noise=True)
                assert all(c == (float(env.std_noise[which]), nc) for c in sn.calls) and len(sn.calls) == M
                env.rng_noise = ScriptedNormal(np.zeros(16))
                one = fn_one(env.mover_names[order[0]], add_noise=False)
                assert np.array_equal(one, clean[0])
                res[which] = (clean, noisy, units)
            out[f'{tag}_M'] = np.array(M); out[f'{tag}_order'] = np.array(order)
            out[f'{tag}_initial_mover_zpos'] = np.array(env.initial_mover_zpos); out[f'{tag}_std_noise'] = env.std_noise.copy()
            out[f'{tag}_pos'] = world.qpos[:, :2].copy(); out[f'{tag}_vel'] = world.qvel[:, :2].copy(); out[f'{tag}_acc'] = world.qacc[:, :2].copy()
            out[f'{tag}_mover_size'] = env.mover_size.copy()
            for which, nm in ((0, 'qpos'), (1, 'qvel'), (2, 'qacc')):
                out[f'{tag}_{nm}'] = res[which][0]; out[f'{tag}_{nm}_noisy'] = res[which][1]; out[f'{tag}_{nm}_units'] = res[which][2]
        finally:
            world.uninstall()
    out['tags'] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(HERE, 'getters.npz'), **out)
    print('getters:', ', '.join(f'{c[0]}' for c in cases))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'round4':  # only the fixture added in round 4
        gen_getters()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'round3':  # only the fixtures added in round 3
        gen_step_template()
        gen_reset_template()
        gen_pushing_methods()
        sys.exit(0)
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
    gen_step_template()
    gen_reset_template()
    gen_pushing_methods()
    gen_getters()
