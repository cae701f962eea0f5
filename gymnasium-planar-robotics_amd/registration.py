"""Registration of the vectorised envs under the reference's Gymnasium ids (gymnasium_planar_robotics/__init__.py:21-41:
``BenchmarkPlanningEnv-v0`` and ``BenchmarkPushingEnv-v0``, both with ``max_episode_steps=50``) -- only when Gymnasium is
importable; the engine itself does not depend on it.

The ids are registered as VECTOR entry points (``gymnasium.make_vec(id, num_envs=..., vectorization_mode='vector_entry_point')``
constructs the batched HIP env directly; there is no single-env class to wrap), under the namespace ``gpr_amd/`` always and under
the reference's bare id only if nobody -- e.g. the reference package itself -- has registered it already.
"""
from __future__ import annotations

from . import spaces

VECTOR_ENTRY_POINTS = {
    'BenchmarkPlanningEnv-v0': 'gymnasium_planar_robotics_amd.envs.planning.benchmark_planning_env:VecBenchmarkPlanningEnv',
    'BenchmarkPushingEnv-v0': 'gymnasium_planar_robotics_amd.envs.manipulation.benchmark_pushing_env:VecBenchmarkPushingEnv',
}
MAX_EPISODE_STEPS = 50
NAMESPACE = 'gpr_amd'


def register_gymnasium_envs() -> list[str]:
    """-> the ids registered by this call ([] when Gymnasium is absent)."""
    if not spaces.HAVE_GYMNASIUM:
        return []
    from gymnasium.envs.registration import register, registry
    done = []
    for env_id, entry in VECTOR_ENTRY_POINTS.items():
        for full in (f'{NAMESPACE}/{env_id}', env_id):
            if full in registry:
                continue
            # the TimeLimit of the registered ids lives in the step kernel (the classes' max_episode_steps keyword, default 50)
            register(id=full, vector_entry_point=entry, kwargs={'max_episode_steps': MAX_EPISODE_STEPS})
            done.append(full)
    return done
