"""ctypes binding of libgpr_hip.so (C ABI: include/gpr.h). There is NO fallback: if the library is missing the
import of anything that computes fails loudly -- build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C gymnasium-planar-robotics_amd/csrc`."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GPR_LIB_PATH') or os.path.join(HERE, 'libgpr_hip.so')  # (override: A/B timing of two builds)
MAX_MOVERS = 16

OK, E_INVALID, E_HIP, E_UNBOUND, E_UNSUPPORTED = 0, -1, -2, -3, -4
SHAPE = {'circle': 0, 'box': 1}
AUTORESET = {'disabled': 0, 'same_step': 1, 'next_step': 2}
PRECISION = {'f32': 0, 'f32_f64acc': 1, 'f64': 2}
COUNTERS = ('episodes', 'terminated', 'truncated', 'success', 'wall_collision', 'mover_collision', 'reset_exhausted', 'actions_clipped')  # GPR_CNT_*

EXPORTS = (
    'gpr_version', 'gpr_last_error', 'gpr_create', 'gpr_destroy', 'gpr_bind_state', 'gpr_seed', 'gpr_get_rng_state', 'gpr_set_rng_state',
    'gpr_reset', 'gpr_reset_candidates', 'gpr_step', 'gpr_step_many', 'gpr_streams_overlap', 'gpr_set_wave_budget', 'gpr_step_n', 'gpr_enable_counters', 'gpr_read_counters', 'gpr_count_clipped_actions', 'gpr_read_diagnostics', 'gpr_get_mover_state',
    'gpr_compute_reward', 'gpr_agent_views', 'gpr_push_configure', 'gpr_push_reset', 'gpr_push_reset_opts', 'gpr_push_reset_candidates', 'gpr_push_step', 'gpr_qpos_is_valid', 'gpr_check_mover_collision', 'gpr_check_line_segments_intersect',
    'gpr_get_2D_rect_vertices', 'gpr_check_rectangles_intersect', 'gpr_ensure_max_dyn_val',
)


class GprConfig(C.Structure):
    _fields_ = [
        ('num_envs', C.c_int64), ('env_id_offset', C.c_int64), ('num_movers', C.c_int32), ('num_tiles_x', C.c_int32),
        ('num_tiles_y', C.c_int32), ('layout_tiles', C.POINTER(C.c_int8)), ('tile_half_size', C.c_double * 2),
        ('tile_x_centres', C.POINTER(C.c_double)), ('tile_y_centres', C.POINTER(C.c_double)), ('c_shape', C.c_int32),
        ('c_size', C.POINTER(C.c_double)), ('c_offset', C.c_double), ('c_offset_wall', C.c_double), ('cycle_time', C.c_double),
        ('num_cycles', C.c_int32), ('v_max', C.c_double), ('a_max', C.c_double), ('j_max', C.c_double), ('learn_jerk', C.c_int32),
        ('threshold_pos', C.c_double), ('std_noise', C.c_double * 3), ('max_episode_steps', C.c_int32), ('autoreset_mode', C.c_int32),
        ('min_xy_pos', C.c_double * 2), ('max_xy_pos', C.c_double * 2), ('min_goal_dist', C.c_double), ('max_reset_trials', C.c_int32),
        ('precision', C.c_int32),
    ]


class GprState(C.Structure):
    _fields_ = [('pos_dev', C.c_void_p), ('vel_dev', C.c_void_p), ('acc_dev', C.c_void_p), ('goal_dev', C.c_void_p),
                ('elapsed_dev', C.c_void_p)]


class GprStepOut(C.Structure):
    _fields_ = [('observation_dev', C.c_void_p), ('achieved_goal_dev', C.c_void_p), ('desired_goal_dev', C.c_void_p),
                ('reward_dev', C.c_void_p), ('terminated_dev', C.c_void_p), ('truncated_dev', C.c_void_p), ('is_success_dev', C.c_void_p),
                ('mover_collision_dev', C.c_void_p), ('wall_collision_dev', C.c_void_p), ('final_observation_dev', C.c_void_p),
                ('final_achieved_goal_dev', C.c_void_p), ('final_desired_goal_dev', C.c_void_p), ('agent_observation_dev', C.c_void_p),
                ('agent_reward_dev', C.c_void_p)]


class GprPushParams(C.Structure):
    _fields_ = [('mover_mass', C.c_double), ('mover_half_size', C.c_double * 2), ('object_half_size', C.c_double), ('object_mass', C.c_double),
                ('object_damping', C.c_double), ('friction_tile', C.c_double), ('friction_contact', C.c_double),
                ('friction_torsional', C.c_double), ('gravity', C.c_double), ('rot_stiffness', C.c_double), ('contact_timeconst', C.c_double),
                ('object_noise', C.c_double), ('min_mo_dist', C.c_double), ('obj_min_xy', C.c_double * 2), ('obj_max_xy', C.c_double * 2)]


class GprPushState(C.Structure):
    _fields_ = [('mover_rot_dev', C.c_void_p), ('object_dev', C.c_void_p)]


class GprError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f'libgpr_hip error {code}: {message}')
        self.code = code


_lib = None


def load() -> C.CDLL:
    """Load libgpr_hip.so and declare every entry point of include/gpr.h. Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} not found: the HIP engine is not built and there is no CPU fallback. '
            'Run `python -c "import __graft_entry__ as g; g.build()"` (hipcc --offload-arch=gfx950).'
        )
    lib = C.CDLL(LIB_PATH)
    vp, i64, i32, u64, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_uint64, C.c_double
    lib.gpr_version.restype = C.c_char_p
    lib.gpr_version.argtypes = []
    lib.gpr_last_error.restype = C.c_char_p
    lib.gpr_last_error.argtypes = [vp]
    lib.gpr_create.argtypes = [C.POINTER(GprConfig), C.POINTER(vp)]
    lib.gpr_destroy.restype = None
    lib.gpr_destroy.argtypes = [vp]
    lib.gpr_bind_state.argtypes = [vp, C.POINTER(GprState)]
    lib.gpr_seed.argtypes = [vp, u64]
    lib.gpr_reset.argtypes = [vp, vp, C.POINTER(GprStepOut), vp]
    lib.gpr_reset_candidates.argtypes = [vp, vp, i32, vp, vp, C.POINTER(GprStepOut), vp]
    lib.gpr_step.argtypes = [vp, vp, C.POINTER(GprStepOut), vp]
    lib.gpr_step_many.argtypes = [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.POINTER(GprStepOut)), C.POINTER(vp)]
    lib.gpr_streams_overlap.argtypes = [vp, vp, C.POINTER(i32)]
    lib.gpr_set_wave_budget.argtypes = [vp, i32]
    lib.gpr_step_n.argtypes = [vp, i32, vp, C.POINTER(GprStepOut), vp]
    lib.gpr_compute_reward.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp]
    lib.gpr_agent_views.argtypes = [vp, C.POINTER(GprStepOut), vp, vp, vp]
    lib.gpr_get_rng_state.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    lib.gpr_set_rng_state.argtypes = [vp, u64, u64]
    lib.gpr_enable_counters.argtypes = [vp, i32]
    lib.gpr_read_counters.argtypes = [vp, vp, vp]
    lib.gpr_count_clipped_actions.argtypes = [vp, i32]
    lib.gpr_read_diagnostics.argtypes = [vp, C.POINTER(u64), vp]
    lib.gpr_get_mover_state.argtypes = [vp, i32, i32, vp, dbl, i32, vp, vp]
    lib.gpr_push_configure.argtypes = [vp, C.POINTER(GprPushParams), C.POINTER(GprPushState)]
    lib.gpr_push_reset.argtypes = [vp, vp, C.POINTER(GprStepOut), vp]
    lib.gpr_push_step.argtypes = [vp, vp, C.POINTER(GprStepOut), vp]
    lib.gpr_push_reset_opts.argtypes = [vp, vp, vp, vp, C.POINTER(GprStepOut), vp]
    lib.gpr_push_reset_candidates.argtypes = [vp, vp, vp, vp, i32, vp, vp, C.POINTER(GprStepOut), vp]
    lib.gpr_qpos_is_valid.argtypes = [vp, i64, vp, vp, i32, vp, vp]
    lib.gpr_check_mover_collision.argtypes = [vp, i64, vp, vp, i32, vp, vp]
    lib.gpr_check_line_segments_intersect.argtypes = [i64, vp, vp, vp, vp, vp, vp]
    lib.gpr_get_2D_rect_vertices.argtypes = [i64, vp, vp, vp, vp]
    lib.gpr_check_rectangles_intersect.argtypes = [i64, vp, vp, vp, vp, vp, vp]
    lib.gpr_ensure_max_dyn_val.argtypes = [i64, vp, vp, vp, dbl, vp, vp, vp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ('gpr_version', 'gpr_last_error', 'gpr_destroy'):
            fn.restype = C.c_int
    _lib = lib
    return lib


def check(rc: int, handle=None) -> None:
    if rc != OK:
        msg = load().gpr_last_error(handle)
        raise GprError(rc, msg.decode() if msg else '')


def current_stream_ptr(device) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream
