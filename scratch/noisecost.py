import sys, os, numpy as np, torch
sys.path.insert(0,'.')
import gymnasium_planar_robotics_amd as gpr
def run(sigma, eager=False, E=65536, M=4, steps=400, amax=10.0, label=''):
    if eager: os.environ['GPR_EAGER_NOISE']='1'
    else: os.environ.pop('GPR_EAGER_NOISE',None)
    env=gpr.VecBenchmarkPlanningEnv(np.ones((6,6)),M,num_envs=E,std_noise=sigma)
    env.reset(seed=0)
    acts=[(torch.rand(E,2*M,device='cuda')*2*amax-amax).to(env.dtype) for _ in range(8)]
    for i in range(60): env.step_into(acts[i%8], env._out)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): env.step_into(acts[i%8], env._out)
    e1.record(); torch.cuda.synchronize()
    print(f'{label:28s} sigma={sigma:g} eager={eager}: {e0.elapsed_time(e1)*1e3/steps:7.2f} us/step', flush=True)
    env.close()
run(0.0, label='no noise kernel')
run(1e-30, label='noise kernel, ~never drawn')
run(1e-5, label='noise kernel lazy')
run(1e-5, eager=True, label='noise kernel eager')
run(1e-5, amax=1.0, label='lazy, gentle actions')
run(0.0, amax=1.0, label='no noise, gentle actions')
