import sys, numpy as np, torch
sys.path.insert(0,'.')
import gymnasium_planar_robotics_amd as gpr
E,M=65536,4
env=gpr.VecBenchmarkPlanningEnv(np.ones((6,6)),M,num_envs=E)
env.reset(seed=0)
g=torch.Generator(device='cuda').manual_seed(1)
pv=[];pw=[];pp=[]
for s in range(150):
    a=torch.rand(E,2*M,device='cuda',generator=g)*20-10
    env.step(a)
    if s>=50:
        v=env.vel.reshape(E,M,2); p=env.pos.reshape(E,M,2)
        sp=(v+0.04*a.reshape(E,M,2)).norm(dim=-1)   # crude: speed reachable within next step
        pv.append(float((v.norm(dim=-1)>1.9).float().mean()))
        m=torch.minimum(torch.minimum(p[...,0]-0.11, 1.44-0.11-p[...,0]), torch.minimum(p[...,1]-0.11,1.44-0.11-p[...,1]))
        pw.append(float((m.abs()<5.7e-5).float().mean()))
        d=(p[:,:,None,:]-p[:,None,:,:]).norm(dim=-1)+torch.eye(M,device='cuda')*10
        pp.append(float(((d.min(-1).values-0.22).abs()<1.2e-4).float().mean()))
pv,pw,pp=np.mean(pv),np.mean(pw),np.mean(pp)
print('frac movers with |v|>1.9:',pv,' -> wave prob',1-(1-pv)**64)
print('frac movers within band of wall:',pw,' -> wave prob',1-(1-pw)**64)
print('frac movers within band of pair thr:',pp,' -> wave prob',1-(1-pp)**64)
print('mean speed', float(env.vel.reshape(E,M,2).norm(dim=-1).mean()), 'mean elapsed', float(env.elapsed_steps.float().mean()))
