import sys, time, numpy as np, torch
sys.path.insert(0,'.')
import gymnasium_planar_robotics_amd as gpr
def run(E, M, tiles, sigma, steps=300, precision='f32'):
    env=gpr.VecBenchmarkPlanningEnv(np.ones((tiles,tiles)),M,num_envs=E,std_noise=sigma,precision=precision)
    env.reset(seed=0)
    acts=[(torch.rand(E,2*M,device='cuda')*20-10).to(env.dtype) for _ in range(8)]
    for i in range(20): env.step_into(acts[i%8], env._out)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): env.step_into(acts[i%8], env._out)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/steps
    print(f'E={E:7d} M={M:2d} sigma={sigma:g} {precision}: {us:8.2f} us/step  {E/us*1e6:.3e} env-steps/s  {us*1e3/E*1.0:.4f} ns/env', flush=True)
    env.close()
for E in (4096, 16384, 65536, 262144, 1048576):
    for s in (0.0, 1e-5):
        run(E,4,6,s)
run(65536,4,6,1e-5,precision='f32_f64acc'); run(65536,4,6,1e-5,precision='f64')
run(8192,8,8,1e-5); run(65536,8,8,1e-5); run(4096,16,10,1e-5); run(32768,16,10,1e-5); run(65536,1,3,1e-5); run(65536,2,4,1e-5)
