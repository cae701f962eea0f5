"""MI355X-native vectorised planar-robotics environments: a drop-in for the per-env hot path of
ubi-coro/gymnasium-planar-robotics (BenchmarkPlanningEnv step()/reset(), collision checks, geometry_2D_utils), batched and
executed by hand-written HIP kernels (gfx950) behind the C ABI of include/gpr.h. See DESIGN.md."""
__version__ = '0.1.0'

from . import _native, distributed, registration, spaces
from .envs.basic_envs import VecBasicPlanarRoboticsEnv
from .envs.manipulation.benchmark_pushing_env import VecBenchmarkPushingEnv
from .envs.planning.benchmark_planning_env import VecBenchmarkPlanningEnv
from .envs.planning.parallel_planning_env import VecBenchmarkPlanningParallelEnv
from .envs.planning.sub_batched_env import VecBenchmarkPlanningEnvSubBatched
from .utils import geometry_2D_utils

registered_ids = registration.register_gymnasium_envs()  # [] unless gymnasium is importable

__all__ = ['VecBasicPlanarRoboticsEnv', 'VecBenchmarkPlanningEnv', 'VecBenchmarkPlanningParallelEnv', 'VecBenchmarkPlanningEnvSubBatched', 'VecBenchmarkPushingEnv', 'distributed', 'geometry_2D_utils', 'spaces', '_native']
