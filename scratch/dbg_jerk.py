import numpy as np, torch, sys
sys.path.insert(0,'.')
import gymnasium_planar_robotics_amd as gpr
from oracle import oracle
layout=np.ones((5,4)); M=2; E=300
for ncyc in (40,):
    genv=gpr.VecBenchmarkPlanningEnv(layout,M,num_envs=E,precision='f32',learn_jerk=True,env_id_offset=1000,num_cycles=ncyc)
    oenv=oracle.OraclePlanningEnv(layout,M,E,learn_jerk=True,storage_f32=True,env_id_offset=1000,num_cycles=ncyc)
    genv.reset(seed=1237); oenv.reset(seed=1237)
    rng=np.random.default_rng(3)
    for step in range(60):
        a=rng.uniform(-120,120,size=(E,2*M)).astype(np.float32).astype(np.float64)
        if step%7==3: a[:E//3]*=0.05
        dev,dt=genv.device,genv.dtype
        genv.pos.copy_(torch.as_tensor(oenv.pos,dtype=dt,device=dev)); genv.vel.copy_(torch.as_tensor(oenv.vel,dtype=dt,device=dev))
        genv.acc.copy_(torch.as_tensor(oenv.acc,dtype=dt,device=dev)); genv.goals.copy_(torch.as_tensor(oenv.goal,dtype=dt,device=dev))
        genv.elapsed_steps.copy_(torch.as_tensor(oenv.elapsed,device=dev))
        v0=oenv.vel.copy(); a0=oenv.acc.copy()
        obs,rew,term,trunc,info=genv.step(torch.as_tensor(a,dtype=dt,device=dev))
        o=oenv.step(a)
        d=np.abs(obs['observation'].double().cpu().numpy()-o['observation'])
        same=(term.cpu().numpy()==o['terminated'])
        d[~same]=0
        e,k=np.unravel_index(d.argmax(),d.shape)
        if d.max()>1e-3:
            print('step',step,'env',e,'comp',k,'err',d.max(),'gpu',obs['observation'][e].cpu().numpy(),'orc',o['observation'][e],'\n v0',v0[e],'a0',a0[e],'act',a[e],'term',o['terminated'][e],'el',oenv.elapsed[e])
