import sys, numpy as np, torch
sys.path.insert(0,'.')
import gymnasium_planar_robotics_amd as gpr
def run(E, ncyc, mes=50, sigma=1e-5, auto='same_step', steps=200):
    env=gpr.VecBenchmarkPushingEnv(num_envs=E,num_cycles=ncyc,max_episode_steps=mes,std_noise=sigma,autoreset_mode=auto); env.reset(seed=0)
    acts=[(torch.rand(E,2,device='cuda')*20-10) for _ in range(8)]
    for i in range(30): env.step(acts[i%8])
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): env.step(acts[i%8])
    e1.record(); torch.cuda.synchronize()
    done=float((env._terminated.float().mean()+env._truncated.float().mean()).item())
    print(f'push E={E} ncyc={ncyc} mes={mes} sigma={sigma} auto={auto}: {e0.elapsed_time(e1)*1e3/steps:.1f} us/step done {done:.3f}', flush=True); env.close()
run(65536,0); run(65536,10); run(65536,40); run(65536,40,sigma=0.0); run(65536,40,mes=None,auto='disabled'); run(4096,40); run(4096,0)
