import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import gymnasium_planar_robotics_amd as gpr
from oracle import oracle
M,n,E=4,6,4096
genv = gpr.VecBenchmarkPlanningEnv(np.ones((n,n)), M, num_envs=E, precision='f32')
oenv = oracle.OraclePlanningEnv(np.ones((n,n)), M, E, storage_f32=True, std_noise=1e-5)
genv.reset(seed=21); oenv.reset(seed=21)
rng=np.random.default_rng(8)
_g=lambda t: t.double().cpu().numpy() if t.dtype.is_floating_point else t.cpu().numpy()
alive=np.ones(E,bool); nsplit=0; nmarg=0
for step in range(100):
    a=rng.uniform(-10,10,size=(E,2*M)).astype(np.float32).astype(np.float64)
    obs,rew,term,trunc,info=genv.step(torch.as_tensor(a,dtype=genv.dtype,device=genv.device))
    o=oenv.step(a)
    same=np.ones(E,bool)
    for k,gv in (('terminated',term),('truncated',trunc),('is_success',info['is_success']),('mover_collision',info['mover_collision']),('wall_collision',info['wall_collision'])):
        same&=_g(gv)==o[k]
    split=alive&~same; nsplit+=split.sum()
    alive&=same
    err=np.abs(_g(obs['achieved_goal'])-o['achieved_goal']).max(axis=1)
    bad=alive&(err>1e-4)
    lowm=alive&(o['margin']<=1e-6); nmarg+=lowm.sum()
    if step<12 or step%10==0: print(step,'alive',alive.sum(),'split',split.sum(),'lowmargin',lowm.sum(),'obs>1e-4 among alive',bad.sum(),'max err alive',err[alive].max(), 'margins q', np.quantile(o['margin'],[0.001,0.01,0.1]))
    alive&=~bad
print(nsplit,nmarg)
