import sys; sys.argv=['x']
import numpy as np, torch
sys.path.insert(0,'.')
import gymnasium_planar_robotics_amd as gpr
def run(E,M,tiles,cp,sigma=1e-5,steps=200,**kw):
    env=gpr.VecBenchmarkPlanningEnv(np.ones((tiles,tiles)),M,num_envs=E,std_noise=sigma,collision_params=cp,**kw); env.reset(seed=0)
    acts=[(torch.rand(E,2*M,device='cuda')*20-10).to(env.dtype) for _ in range(8)]
    for i in range(30): env.step_into(acts[i%8], env._out)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): env.step_into(acts[i%8], env._out)
    e1.record(); torch.cuda.synchronize()
    print(f'E={E} M={M} shape={cp["shape"] if cp else "circle"} sigma={sigma} {kw}: {e0.elapsed_time(e1)*1e3/steps:.1f} us/step', flush=True); env.close()
box={'shape':'box','size':np.array([0.0775,0.0775]),'offset':0.0}
run(65536,4,6,box); run(65536,4,6,box,sigma=0.0); run(4096,4,6,box)
holes=np.ones((6,6)); 
run(65536,1,6,box,sigma=0.0); run(65536,2,6,box,sigma=0.0); run(65536,4,6,box,sigma=0.0, num_cycles=0)
run(65536,4,6,box,sigma=0.0, autoreset_mode='disabled', max_episode_steps=None)
run(65536,4,6,box,sigma=0.0, num_cycles=10)
