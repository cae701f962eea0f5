import sys, time, numpy as np, torch
sys.path.insert(0,'.')
import gymnasium_planar_robotics_amd as gpr
E,M=65536,4
def run(K, sigma=1e-5, steps=2000):
    envs=[gpr.VecBenchmarkPlanningEnv(np.ones((6,6)),M,num_envs=E//K,std_noise=sigma,env_id_offset=k*(E//K)) for k in range(K)]
    streams=[torch.cuda.Stream() for _ in range(K)]
    acts=[[(torch.rand(E//K,2*M,device='cuda')*20-10) for _ in range(8)] for _ in range(K)]
    for e in envs: e.reset(seed=0)
    torch.cuda.synchronize()
    def go(n):
        for i in range(n):
            for k in range(K):
                with torch.cuda.stream(streams[k]):
                    envs[k].step_into(acts[k][i%8], envs[k]._out)
    go(100); torch.cuda.synchronize()
    t0=time.perf_counter(); go(steps); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(f'K={K} sub-batches on {K} streams, sigma={sigma}: {dt/steps*1e6:.2f} us per full step  {E*steps/dt:.3e} env-steps/s', flush=True)
    for e in envs: e.close()
for K in (1,2,4):
    run(K); run(K, 0.0)
