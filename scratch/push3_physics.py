"""gpr-push-v3 on the CPU oracle: the analytic answers the GPU tests hold the engine to (scratch, round 4)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle

g, mu = 9.81, 1.0
# 1. free sliding: an object launched at v0 stops after v0^2 / (2 mu g) (+ the joint damping's share, small)
for v0 in (0.2, 0.5, 1.0):
    e = oracle.OraclePushingEnv(1, std_noise=0.0, autoreset='disabled', max_episode_steps=None, object_noise=0.0, push_overrides={'object_damping': 0.0})
    e.reset(seed=0)
    e.pos[:] = [0.6, 0.6]; e.object[:] = 0; e.object[0, :2] = [0.15, 0.36]; e.object[0, 3] = v0
    x0 = e.object[0, 0]
    for _ in range(10):
        e.step(np.zeros((1, 2)))
    print('slide v0', v0, 'dist', e.object[0, 0] - x0, 'analytic', v0 ** 2 / (2 * mu * g), 'end v', e.object[0, 3])
# 2. steady push: mover at constant speed pushes the object face-on; object speed -> mover speed, penetration bounded
e = oracle.OraclePushingEnv(1, std_noise=0.0, autoreset='disabled', max_episode_steps=None, object_noise=0.0, v_max=0.2)
e.reset(seed=0)
e.pos[:] = [0.2, 0.36]; e.vel[:] = [0.2, 0.0]; e.object[:] = 0; e.object[0, :2] = [0.2 + 0.0775 + 0.035 + 0.002, 0.36]
for t in range(8):
    r = e.step(np.array([[10.0, 0.0]]))
    gap = e.object[0, 0] - e.pos[0, 0] - (0.0775 + 0.035)
    print('push step', t, 'mover v', e.vel[0], 'obj v', e.object[0, 3:6], 'gap', gap, 'yaw', e.object[0, 2], e.rot[0])
# 3. off-centre push: the object's centre is above the mover's face centre line by 2 cm, mover narrower? (mover half 0.0775 > object half 0.035:
# the whole object face is on the mover's face) -> pure translation expected; push with the mover's CORNER region instead: object offset by 0.07
e = oracle.OraclePushingEnv(1, std_noise=0.0, autoreset='disabled', max_episode_steps=None, object_noise=0.0, v_max=0.2)
e.reset(seed=0)
e.pos[:] = [0.2, 0.30]; e.vel[:] = [0.2, 0.0]; e.object[:] = 0; e.object[0, :2] = [0.2 + 0.0775 + 0.035 + 0.002, 0.30 + 0.09]
for t in range(8):
    e.step(np.array([[10.0, 0.0]]))
    print('offc step', t, 'obj', e.object[0], 'mover yaw', e.rot[0])
