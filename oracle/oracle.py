"""ctypes binding of the CPU oracle (oracle/libgpr_oracle.so). TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libgpr_oracle.so')
SAN_LIB = os.environ.get('GPR_ORACLE_SANITIZED_LIB')  # tests/test_host_cpu.py: the same oracle built with -fsanitize=address,undefined
MAX_MOVERS = 16


def build(force: bool = False) -> str:
    src = [os.path.join(HERE, f) for f in ('gpr_oracle.c', 'gpr_oracle.h')]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        subprocess.check_call(['make', '-C', HERE, '-B', 'libgpr_oracle.so'], stdout=subprocess.DEVNULL)
    return LIB_PATH


class Layout(C.Structure):
    _fields_ = [('num_tiles_x', C.c_int32), ('num_tiles_y', C.c_int32), ('layout', C.POINTER(C.c_int8)),
                ('tile_hx', C.c_double), ('tile_hy', C.c_double), ('c_shape', C.c_int32), ('c_offset', C.c_double),
                ('c_offset_wall', C.c_double)]


class EnvCfg(C.Structure):
    _fields_ = [('lay', Layout), ('num_movers', C.c_int32), ('c_size', C.c_double * (MAX_MOVERS * 2)), ('dt', C.c_double),
                ('num_cycles', C.c_int32), ('v_max', C.c_double), ('a_max', C.c_double), ('j_max', C.c_double),
                ('learn_jerk', C.c_int32), ('threshold_pos', C.c_double), ('std_noise', C.c_double * 3),
                ('reward_success', C.c_double), ('max_episode_steps', C.c_int32), ('min_xy', C.c_double * 2),
                ('max_xy', C.c_double * 2), ('min_goal_dist', C.c_double), ('max_reset_trials', C.c_int32),
                ('storage_f32', C.c_int32)]


class StepOut(C.Structure):
    _fields_ = [('observation', C.c_void_p), ('achieved_goal', C.c_void_p), ('desired_goal', C.c_void_p), ('reward', C.c_void_p),
                ('terminated', C.c_void_p), ('truncated', C.c_void_p), ('is_success', C.c_void_p), ('mover_collision', C.c_void_p),
                ('wall_collision', C.c_void_p), ('final_observation', C.c_void_p), ('final_achieved_goal', C.c_void_p),
                ('final_desired_goal', C.c_void_p), ('margin', C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if SAN_LIB:
            _lib = C.CDLL(SAN_LIB)
        else:
            build()
            _lib = C.CDLL(LIB_PATH)
        _lib.gpro_qpos_is_valid.restype = C.c_int64
        _lib.gpro_check_mover_collision.restype = C.c_int32
        _lib.gpro_tile_indices_mask.restype = C.c_int32
        _lib.gpro_rng_key.restype = C.c_uint32
        _lib.gpro_rng_draw.restype = C.c_uint32
        _lib.gpro_rng_uniform.restype = C.c_double
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OracleLayout:
    """Holds a gpro_layout (and keeps the layout array alive)."""

    def __init__(self, layout_tiles, tile_half_size=(0.12, 0.12), shape='circle', offset=0.0, offset_wall=0.0):
        self.layout = np.ascontiguousarray(np.asarray(layout_tiles), dtype=np.int8)
        assert self.layout.ndim == 2
        self.c = Layout(self.layout.shape[0], self.layout.shape[1], self.layout.ctypes.data_as(C.POINTER(C.c_int8)),
                        float(tile_half_size[0]), float(tile_half_size[1]), 0 if shape == 'circle' else 1, float(offset),
                        float(offset_wall))
        self.shape = shape

    # ---- ctor tables ------------------------------------------------------------------------------------
    def tile_centres(self):
        tx, ty = self.layout.shape
        xc = np.zeros(tx); yc = np.zeros(ty)
        lib().gpro_tile_centres(C.c_int32(tx), C.c_double(self.c.tile_hx), _p(xc))
        lib().gpro_tile_centres(C.c_int32(ty), C.c_double(self.c.tile_hy), _p(yc))
        return xc, yc

    def tile_indices_mask(self, which: int):
        cap = self.layout.size + 1
        ix = np.zeros(cap, dtype=np.int32); iy = np.zeros(cap, dtype=np.int32)
        n = lib().gpro_tile_indices_mask(C.byref(self.c), C.c_int32(which), C.c_int32(cap), _p(ix), _p(iy))
        return ix[:n].copy(), iy[:n].copy()

    def planning_constants(self, c_size):
        cs = np.zeros(2); cs[:] = np.atleast_1d(np.asarray(c_size, dtype=np.float64))[:2] if np.ndim(c_size) else c_size
        mn = np.zeros(2); mx = np.zeros(2); gh = np.zeros(2); mgd = C.c_double(0)
        lib().gpro_planning_constants(C.byref(self.c), _p(cs), _p(mn), _p(mx), C.byref(mgd), _p(gh))
        return mn, mx, mgd.value, gh

    # ---- checks -------------------------------------------------------------------------------------------
    def c_size_arr(self, c_size, n):
        """Expand a reference-style c_size (float | (n,) | (2,) | (n,2)) to the oracle's [n][2] array."""
        out = np.zeros((n, 2))
        cs = np.asarray(c_size, dtype=np.float64)
        if self.shape == 'circle':
            out[:, 0] = cs if cs.ndim == 0 else cs.reshape(n)
        else:
            out[:] = cs if cs.shape == (2,) else cs.reshape(n, 2)
        return out

    def qpos_is_valid(self, qpos, c_size, add_safety_offset=False, return_margin=False):
        qpos = _f64(qpos); n = qpos.shape[0]
        cs = self.c_size_arr(c_size, n)
        out = np.zeros(n, dtype=np.int32)
        margin = np.zeros(n) if return_margin else None
        n_off = lib().gpro_qpos_is_valid(C.byref(self.c), C.c_int64(n), _p(qpos), _p(cs), C.c_int32(int(add_safety_offset)), _p(out),
                                         _p(margin))
        return (out, margin, n_off) if return_margin else out

    def check_mover_collision(self, qpos, c_size, add_safety_offset=False):
        qpos = _f64(qpos); m = qpos.shape[0]
        cs = self.c_size_arr(c_size, m)
        return bool(lib().gpro_check_mover_collision(C.byref(self.c), C.c_int32(m), _p(qpos), _p(cs), C.c_int32(int(add_safety_offset)), None))


def check_line_segments_intersect(p1, p2, q1, q2):
    p1, p2, q1, q2 = (_f64(a) for a in (p1, p2, q1, q2))
    out = np.zeros(p1.shape[0], dtype=np.uint8)
    lib().gpro_check_line_segments_intersect(C.c_int64(p1.shape[0]), _p(p1), _p(p2), _p(q1), _p(q2), _p(out))
    return out.astype(bool)


def get_2D_rect_vertices(qpos, size):
    qpos, size = _f64(qpos), _f64(size)
    out = np.zeros((qpos.shape[0], 2, 4))
    lib().gpro_get_2D_rect_vertices(C.c_int64(qpos.shape[0]), _p(qpos), _p(size), _p(out))
    return out


def check_rectangles_intersect(qpos_r1, qpos_r2, size_r1, size_r2):
    a, b, c, d = (_f64(x) for x in (qpos_r1, qpos_r2, size_r1, size_r2))
    out = np.zeros(a.shape[0], dtype=np.uint8)
    lib().gpro_check_rectangles_intersect(C.c_int64(a.shape[0]), _p(a), _p(b), _p(c), _p(d), _p(out))
    return out.astype(bool)


def ensure_max_dyn_val(cur, max_value, deriv, dt):
    cur, deriv = _f64(cur).reshape(-1, 2), _f64(deriv).reshape(-1, 2)
    n = cur.shape[0]
    mv = np.broadcast_to(np.asarray(max_value, dtype=np.float64), (n,)).copy()
    vals = np.zeros((n, 2)); ders = np.zeros((n, 2))
    lib().gpro_ensure_max_dyn_val(C.c_int64(n), _p(cur), _p(mv), _p(deriv), C.c_double(dt), _p(vals), _p(ders))
    return vals, ders


def compute_reward(achieved, desired, mover_collision, wall_collision, num_movers, threshold_pos=0.1, reward_success=50.0):
    ag, dg = _f64(achieved).reshape(-1, 2 * num_movers), _f64(desired).reshape(-1, 2 * num_movers)
    b = ag.shape[0]
    mc = np.ascontiguousarray(mover_collision, dtype=np.uint8).reshape(b)
    wc = np.ascontiguousarray(wall_collision, dtype=np.uint8).reshape(b)
    rew = np.zeros(b); term = np.zeros(b, dtype=np.uint8); succ = np.zeros(b, dtype=np.uint8)
    lib().gpro_compute_reward(C.c_int64(b), C.c_int32(num_movers), _p(ag), _p(dg), _p(mc), _p(wc), C.c_double(threshold_pos),
                              C.c_double(reward_success), _p(rew), _p(term), _p(succ))
    return rew, term.astype(bool), succ.astype(bool)


def rng_key(seed, tick, kind, env_id, mover):
    return int(lib().gpro_rng_key(C.c_uint64(seed), C.c_uint64(tick), C.c_uint32(kind), C.c_uint64(env_id), C.c_uint32(mover)))


def rng_draws(key, n):
    return np.array([lib().gpro_rng_draw(C.c_uint32(key), C.c_uint32(k)) for k in range(n)], dtype=np.uint32)


def rng_normals(key, n_pairs):
    out = np.zeros((n_pairs, 2)); a = C.c_double(); b = C.c_double()
    for q in range(n_pairs):
        lib().gpro_rng_normal_pair(C.c_uint32(key), C.c_uint32(q), C.byref(a), C.byref(b))
        out[q] = (a.value, b.value)
    return out


class OraclePlanningEnv:
    """Batched BenchmarkPlanningEnv on the CPU oracle, mirroring the engine's semantics (same RNG spec, same autoreset modes)."""

    def __init__(self, layout_tiles, num_movers, num_envs, tile_half_size=(0.12, 0.12), std_noise=1e-5, num_cycles=40,
                 collision_params=None, v_max=2.0, a_max=10.0, j_max=100.0, learn_jerk=False, threshold_pos=0.1, max_episode_steps=50,
                 autoreset='same_step', env_id_offset=0, max_reset_trials=10000, storage_f32=False, dt=0.001, num_threads=1):
        cp = dict(collision_params or {})
        self.shape = cp.get('shape', 'circle')
        self.lay = OracleLayout(layout_tiles, tile_half_size, self.shape, cp.get('offset', 0.0), cp.get('offset_wall', 0.0))
        self.M, self.E = int(num_movers), int(num_envs)
        size = cp.get('size', 0.11)
        cs = self.lay.c_size_arr(size, self.M)
        base = cs[0] if self.shape == 'box' else cs[0, 0]
        mn, mx, mgd, gh = self.lay.planning_constants(base)
        sn = np.broadcast_to(np.asarray(std_noise, dtype=np.float64), (3,))
        self.cfg = EnvCfg()
        self.cfg.lay = self.lay.c
        self.cfg.num_movers = self.M
        for i in range(self.M):
            self.cfg.c_size[2 * i], self.cfg.c_size[2 * i + 1] = cs[i, 0], cs[i, 1]
        self.cfg.dt, self.cfg.num_cycles = dt, num_cycles
        self.cfg.v_max, self.cfg.a_max, self.cfg.j_max = v_max, a_max, j_max
        self.cfg.learn_jerk, self.cfg.threshold_pos = int(learn_jerk), threshold_pos
        for i in range(3):
            self.cfg.std_noise[i] = sn[i]
        self.cfg.reward_success = 50.0
        self.cfg.max_episode_steps = max_episode_steps if max_episode_steps else 0
        self.cfg.min_xy[0], self.cfg.min_xy[1], self.cfg.max_xy[0], self.cfg.max_xy[1] = mn[0], mn[1], mx[0], mx[1]
        self.cfg.min_goal_dist = mgd
        self.cfg.max_reset_trials = max_reset_trials
        self.cfg.storage_f32 = int(storage_f32)
        self.min_xy, self.max_xy, self.min_goal_dist = mn, mx, mgd
        self.autoreset = {'disabled': 0, 'same_step': 1, 'next_step': 2}[autoreset]
        self.env_id_offset = env_id_offset
        self.num_threads = num_threads
        E, M = self.E, self.M
        self.obs_dim = 4 * M if learn_jerk else 2 * M
        self.pos = np.zeros((E, 2 * M)); self.vel = np.zeros((E, 2 * M)); self.acc = np.zeros((E, 2 * M)); self.goal = np.zeros((E, 2 * M))
        self.elapsed = np.zeros(E, dtype=np.int32)
        self.seed, self.tick = 0, 0
        self.o = dict(observation=np.zeros((E, self.obs_dim)), achieved_goal=np.zeros((E, 2 * M)), desired_goal=np.zeros((E, 2 * M)),
                      reward=np.zeros(E), terminated=np.zeros(E, dtype=np.uint8), truncated=np.zeros(E, dtype=np.uint8),
                      is_success=np.zeros(E, dtype=np.uint8), mover_collision=np.zeros(E, dtype=np.uint8),
                      wall_collision=np.zeros(E, dtype=np.uint8), final_observation=np.zeros((E, self.obs_dim)),
                      final_achieved_goal=np.zeros((E, 2 * M)), final_desired_goal=np.zeros((E, 2 * M)), margin=np.zeros(E))
        self.out = StepOut(*[_p(self.o[k]) for k, _ in StepOut._fields_])

    def _result(self):
        return {k: (v.astype(bool) if v.dtype == np.uint8 else v.copy()) for k, v in self.o.items()}

    def reset(self, seed=None, mask=None):
        if seed is not None:
            self.seed, self.tick = int(seed), 0
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        rc = lib().gpro_planning_reset(C.byref(self.cfg), C.c_int64(self.E), C.c_uint64(self.seed), C.c_uint64(self.tick),
                                       C.c_int64(self.env_id_offset), _p(m), _p(self.pos), _p(self.vel), _p(self.acc), _p(self.goal),
                                       _p(self.elapsed), C.byref(self.out))
        assert rc == 0
        self.tick += 1
        return self._result()

    def reset_candidates(self, cand, mask=None):
        """reset() on a scripted candidate stream: cand [n_cand, E, M, 2] (or [n_cand, E, 2M]); -> (result, sets consumed per env)"""
        cand = np.ascontiguousarray(np.asarray(cand, dtype=np.float64).reshape(-1, self.E, 2 * self.M))
        n_used = np.zeros(self.E, dtype=np.int32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        rc = lib().gpro_planning_reset_candidates(C.byref(self.cfg), C.c_int64(self.E), C.c_uint64(self.seed), C.c_uint64(self.tick),
                                                  C.c_int64(self.env_id_offset), _p(m), C.c_int32(cand.shape[0]), _p(cand), _p(n_used),
                                                  _p(self.pos), _p(self.vel), _p(self.acc), _p(self.goal), _p(self.elapsed), C.byref(self.out))
        assert rc == 0
        self.tick += 1
        return self._result(), n_used

    def get_mover_state(self, which, mover_idx, z_report, add_noise=False, mover_rot=None):
        """get_mover_qpos / qvel / qacc (+ _arr) of every env (BE:801-855, 1244-1290): which 0 / 1 / 2 -> [E, n_sel, 7 / 6 / 6]; a noisy read
        consumes one call-counter tick like the engine's"""
        idx = np.ascontiguousarray(mover_idx, dtype=np.int32)
        out = np.zeros((self.E, len(idx), 7 if which == 0 else 6))
        rot = None if mover_rot is None else _f64(mover_rot)
        lib().gpro_get_mover_state(C.byref(self.cfg), C.c_int64(self.E), C.c_uint64(self.seed), C.c_uint64(self.tick), C.c_int64(self.env_id_offset),
                                   C.c_int32(which), C.c_int32(len(idx)), _p(idx), C.c_double(z_report), C.c_int32(int(add_noise)), _p(self.pos),
                                   _p(self.vel), _p(self.acc), _p(rot), _p(out))
        if add_noise:
            self.tick += 1
        return out

    def step(self, action):
        a = _f64(action).reshape(self.E, 2 * self.M)
        args = [C.byref(self.cfg), C.c_int64(self.E), C.c_uint64(self.seed), C.c_uint64(self.tick), C.c_int64(self.env_id_offset), _p(a),
                _p(self.pos), _p(self.vel), _p(self.acc), _p(self.goal), _p(self.elapsed), C.c_int32(self.autoreset), C.byref(self.out)]
        if self.num_threads > 1:
            rc = lib().gpro_planning_step_mt(*args, C.c_int32(self.num_threads))
        else:
            rc = lib().gpro_planning_step(*args)
        assert rc == 0
        self.tick += 1
        return self._result()


class PushParams(C.Structure):
    _fields_ = [('mover_mass', C.c_double), ('mover_half', C.c_double * 2), ('object_half', C.c_double), ('object_mass', C.c_double),
                ('object_damping', C.c_double), ('friction_tile', C.c_double), ('friction_contact', C.c_double),
                ('friction_torsional', C.c_double), ('gravity', C.c_double), ('rot_stiffness', C.c_double), ('contact_timeconst', C.c_double),
                ('object_noise', C.c_double), ('min_mo_dist', C.c_double), ('obj_min_xy', C.c_double * 2), ('obj_max_xy', C.c_double * 2)]


class OraclePushingEnv:
    """Batched BenchmarkPushingEnv on the CPU oracle (mover control / obs / reward pinned by the reference; contact physics is the
    engine's own planar model gpr-push-v3 -- parity unpinned, see oracle/gpr_oracle.h). num_movers = 1 is the reference's env,
    2 the engine-defined variant of BASELINE.json's config 3."""

    def __init__(self, num_envs, mover_mass=1.24, mover_half=(0.0775, 0.0775), std_noise=1e-5, num_cycles=40, collision_params=None,
                 v_max=2.0, a_max=10.0, j_max=100.0, learn_jerk=False, threshold_pos=0.05, max_episode_steps=50, autoreset='same_step',
                 env_id_offset=0, max_reset_trials=10000, storage_f32=False, dt=0.001, object_noise=1e-5, num_movers=1, push_overrides=None):
        cp = dict(collision_params or {})
        assert cp.get('shape', 'circle') == 'circle' and num_movers in (1, 2)
        base = OraclePlanningEnv(np.ones((3, 3)), num_movers, num_envs, std_noise=std_noise, num_cycles=num_cycles, collision_params=cp, v_max=v_max,
                                 a_max=a_max, j_max=j_max, learn_jerk=learn_jerk, threshold_pos=threshold_pos,
                                 max_episode_steps=max_episode_steps, autoreset=autoreset, env_id_offset=env_id_offset,
                                 max_reset_trials=max_reset_trials, storage_f32=storage_f32, dt=dt)
        self.base, self.cfg, self.E, self.NM = base, base.cfg, int(num_envs), int(num_movers)
        self.autoreset, self.env_id_offset = base.autoreset, env_id_offset
        c_size, off, off_w = cp.get('size', 0.11), cp.get('offset', 0.0), cp.get('offset_wall', 0.0)
        margin = c_size + off_w + off
        self.min_xy, self.max_xy = base.min_xy, base.max_xy
        self.obj_min, self.obj_max = self.min_xy + margin, self.max_xy - margin  # PU:257-258
        obj_half = 0.07 / 2
        self.min_mo_dist = max(float(np.linalg.norm(obj_half + np.asarray(mover_half), ord=2)), c_size + off)  # PU:279-283
        self.pp = PushParams(mover_mass, (C.c_double * 2)(*mover_half), obj_half, 0.01, 0.01, 1.0, 1.0, 0.005, 9.81, 0.1, 0.02, object_noise,
                             self.min_mo_dist, (C.c_double * 2)(*self.obj_min), (C.c_double * 2)(*self.obj_max))
        for k, v in (push_overrides or {}).items():  # (tests of the contact model's invariants switch friction / damping off)
            setattr(self.pp, {'mover_half_size': 'mover_half', 'object_half_size': 'object_half'}.get(k, k), v)
        E, NM = self.E, self.NM
        self.obs_dim = (6 if learn_jerk else 4) * NM
        self.pos = np.zeros((E, 2 * NM)); self.vel = np.zeros((E, 2 * NM)); self.acc = np.zeros((E, 2 * NM)); self.rot = np.zeros((E, 2 * NM))
        self.object = np.zeros((E, 6)); self.goal = np.zeros((E, 2)); self.elapsed = np.zeros(E, dtype=np.int32)
        self.seed, self.tick = 0, 0
        self.o = dict(observation=np.zeros((E, self.obs_dim)), achieved_goal=np.zeros((E, 2)), desired_goal=np.zeros((E, 2)), reward=np.zeros(E),
                      terminated=np.zeros(E, dtype=np.uint8), truncated=np.zeros(E, dtype=np.uint8), is_success=np.zeros(E, dtype=np.uint8),
                      mover_collision=np.zeros(E, dtype=np.uint8), wall_collision=np.zeros(E, dtype=np.uint8),
                      final_observation=np.zeros((E, self.obs_dim)), final_achieved_goal=np.zeros((E, 2)), final_desired_goal=np.zeros((E, 2)),
                      margin=np.zeros(E))
        self.out = StepOut(*[_p(self.o[k]) for k, _ in StepOut._fields_])

    def _result(self):
        return {k: (v.astype(bool) if v.dtype == np.uint8 else v.copy()) for k, v in self.o.items()}

    def reset(self, seed=None, mask=None, mover_start_xy_pos=None, object_goal_xy_pos=None):
        if seed is not None:
            self.seed, self.tick = int(seed), 0
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        ms = None if mover_start_xy_pos is None else np.ascontiguousarray(np.broadcast_to(np.asarray(mover_start_xy_pos, dtype=np.float64).reshape(-1, 2 * self.NM), (self.E, 2 * self.NM)))
        og = None if object_goal_xy_pos is None else np.ascontiguousarray(np.broadcast_to(np.asarray(object_goal_xy_pos, dtype=np.float64).reshape(-1, 2), (self.E, 2)))
        rc = lib().gpro_pushing_reset(C.byref(self.cfg), C.byref(self.pp), C.c_int64(self.E), C.c_uint64(self.seed), C.c_uint64(self.tick),
                                      C.c_int64(self.env_id_offset), _p(m), _p(ms), _p(og), _p(self.pos), _p(self.vel), _p(self.acc), _p(self.rot),
                                      _p(self.object), _p(self.goal), _p(self.elapsed), C.byref(self.out))
        assert rc == 0
        self.tick += 1
        return self._result()

    def reset_candidates(self, object_cand, mask=None, mover_start_xy_pos=None, object_goal_xy_pos=None):
        """reset() with the object's start candidates scripted: object_cand [n_cand, E, 2]; -> (result, candidates consumed per env)"""
        cand = np.ascontiguousarray(np.asarray(object_cand, dtype=np.float64).reshape(-1, self.E, 2))
        n_used = np.zeros(self.E, dtype=np.int32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        ms = None if mover_start_xy_pos is None else np.ascontiguousarray(np.broadcast_to(np.asarray(mover_start_xy_pos, dtype=np.float64).reshape(-1, 2 * self.NM), (self.E, 2 * self.NM)))
        og = None if object_goal_xy_pos is None else np.ascontiguousarray(np.broadcast_to(np.asarray(object_goal_xy_pos, dtype=np.float64).reshape(-1, 2), (self.E, 2)))
        rc = lib().gpro_pushing_reset_candidates(C.byref(self.cfg), C.byref(self.pp), C.c_int64(self.E), C.c_uint64(self.seed), C.c_uint64(self.tick),
                                                 C.c_int64(self.env_id_offset), _p(m), _p(ms), _p(og), C.c_int32(cand.shape[0]), _p(cand), _p(n_used),
                                                 _p(self.pos), _p(self.vel), _p(self.acc), _p(self.rot), _p(self.object), _p(self.goal),
                                                 _p(self.elapsed), C.byref(self.out))
        assert rc == 0
        self.tick += 1
        return self._result(), n_used

    def step(self, action):
        a = _f64(action).reshape(self.E, 2 * self.NM)
        rc = lib().gpro_pushing_step(C.byref(self.cfg), C.byref(self.pp), C.c_int64(self.E), C.c_uint64(self.seed), C.c_uint64(self.tick),
                                     C.c_int64(self.env_id_offset), _p(a), _p(self.pos), _p(self.vel), _p(self.acc), _p(self.rot),
                                     _p(self.object), _p(self.goal), _p(self.elapsed), C.c_int32(self.autoreset), C.byref(self.out))
        assert rc == 0
        self.tick += 1
        return self._result()
