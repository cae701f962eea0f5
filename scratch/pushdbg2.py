import torch, numpy as np, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import gymnasium_planar_robotics_amd as gpr
from oracle import oracle
E=333
genv = gpr.VecBenchmarkPushingEnv(num_envs=E, precision='f32', autoreset_mode='next_step', env_id_offset=77, std_noise=0.0)
oenv = oracle.OraclePushingEnv(E, autoreset='next_step', storage_f32=True, env_id_offset=77, std_noise=0.0)
genv.reset(seed=3); oenv.reset(seed=3)
rng = np.random.default_rng(1); lim = 10.0
_g = lambda t: t.double().cpu().numpy() if t.dtype.is_floating_point else t.cpu().numpy()
for step in range(70):
    to_obj = oenv.object[:, :2] - oenv.pos
    a = np.where(rng.random((E, 1)) < 0.7, to_obj / (np.linalg.norm(to_obj, axis=1, keepdims=True) + 1e-9) * lim * 0.6, rng.uniform(-lim, lim, size=(E, 2)))
    a = a.astype(np.float32).astype(np.float64)
    for gt, ot in ((genv.pos, oenv.pos), (genv.vel, oenv.vel), (genv.acc, oenv.acc), (genv.goals, oenv.goal), (genv.mover_rot, oenv.rot), (genv.object, oenv.object)):
        gt.copy_(torch.as_tensor(ot, dtype=genv.dtype, device=genv.device))
    genv.elapsed_steps.copy_(torch.as_tensor(oenv.elapsed, device=genv.device))
    pre = (oenv.pos.copy(), oenv.vel.copy(), oenv.rot.copy(), oenv.object.copy())
    obs, rew, term, trunc, info = genv.step(torch.as_tensor(a, dtype=genv.dtype, device=genv.device))
    o = oenv.step(a)
    err = np.abs(_g(obs['observation']) - o['observation']).max(axis=1)
    b = int(err.argmax())
    if err[b] > 1e-4:
        print('step', step, 'env', b, 'err', err[b], 'margin', o['margin'][b], 'term', o['terminated'][b], _g(term)[b])
        print(' pre pos', pre[0][b], 'vel', pre[1][b], 'rot', pre[2][b], 'obj', pre[3][b], 'a', a[b])
        print(' gpu obs', _g(obs['observation'])[b], 'rot', _g(genv.mover_rot)[b], 'obj', _g(genv.object)[b])
        print(' ora obs', o['observation'][b], 'rot', oenv.rot[b], 'obj', oenv.object[b])
