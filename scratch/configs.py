# timings of the BASELINE.json configurations other than the headline (parity-test cases, not bench lines)
import sys, numpy as np, torch
sys.path.insert(0,'.')
import gymnasium_planar_robotics_amd as gpr
def timeit(fn, steps=300, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/steps
def planning(E,M,tiles,**kw):
    env=gpr.VecBenchmarkPlanningEnv(np.ones((tiles,tiles)),M,num_envs=E,**kw); env.reset(seed=0)
    acts=[(torch.rand(E,2*M,device='cuda')*20-10).to(env.dtype) for _ in range(8)]
    i=[0]
    def f(): env.step(acts[i[0]%8]); i[0]+=1
    us=timeit(f); print(f'planning E={E} M={M} {kw}: {us:.1f} us/step {E/us*1e6:.3e} env-steps/s', flush=True); env.close()
def pushing(E,**kw):
    env=gpr.VecBenchmarkPushingEnv(num_envs=E,**kw); env.reset(seed=0)
    acts=[(torch.rand(E,2,device='cuda')*20-10).to(env.dtype) for _ in range(8)]
    i=[0]
    def f(): env.step(acts[i[0]%8]); i[0]+=1
    us=timeit(f); print(f'pushing  E={E} {kw}: {us:.1f} us/step {E/us*1e6:.3e} env-steps/s', flush=True); env.close()
def parallel(E,M,tiles):
    env=gpr.VecBenchmarkPlanningParallelEnv(np.ones((tiles,tiles)),M,num_envs=E); env.reset(seed=0)
    acts=[{a:(torch.rand(E,2,device='cuda')*20-10) for a in env.agents} for _ in range(4)]
    i=[0]
    def f(): env.step(acts[i[0]%4]); i[0]+=1
    us=timeit(f,steps=100,warm=10); print(f'parallel E={E} M={M}: {us:.1f} us/step {E/us*1e6:.3e} env-steps/s', flush=True); env.close()
planning(4096,4,6); planning(8192,8,8); planning(4096,16,10)
pushing(4096); pushing(65536); pushing(4096,precision='f64')
parallel(4096,16,10)
