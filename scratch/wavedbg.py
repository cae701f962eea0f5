import sys, os, numpy as np, torch
os.environ['GPR_DBG']='1'
sys.path.insert(0,'.')
import gymnasium_planar_robotics_amd as gpr
E,M=65536,4
env=gpr.VecBenchmarkPlanningEnv(np.ones((6,6)),M,num_envs=E,std_noise=float(sys.argv[1]))
env.reset(seed=0)
acts=[(torch.rand(E,2*M,device='cuda')*20-10).to(env.dtype) for _ in range(8)]
for i in range(150): env.step_into(acts[i%8], env._out)
torch.cuda.synchronize(); env.close()
d=np.fromfile('gpurun_out/wavedbg.bin',dtype=np.uint32).reshape(4096,8).astype(np.float64)
loop,wait,rst,clamp,near,nres,t0=d[:,0],d[:,1],d[:,2],d[:,3],d[:,4],d[:,5],d[:,6]
print('sigma',sys.argv[1],'(last step) units: s_memtime ticks')
print('start spread (t0 max-min):', t0.max()-t0.min())
for name,x in (('until barrier',loop),('barrier wait',wait),('reset pass',rst),('total',loop+wait+rst)):
    print(f'{name:14s} mean {x.mean():8.0f}  p50 {np.median(x):8.0f}  p99 {np.quantile(x,0.99):8.0f}  max {x.max():8.0f}')
end=t0-t0.min()+loop+wait+rst
print('end time (rel. first start): p50 %.0f max %.0f'%(np.median(end), end.max()))
for lo,hi in ((0,0),(1,10),(11,39),(40,40)):
    m=(clamp>=lo)&(clamp<=hi)
    if m.any(): print(f'clamp cycles {lo}-{hi}: {m.mean()*100:5.1f}% of waves, loop mean {loop[m].mean():.0f}')
w0=(np.arange(4096)%4==0)
print('wave 0 of block (does resets) reset pass mean %.0f; others %.0f; n_reset mean %.1f'%(rst[w0].mean(), rst[~w0].mean(), nres[w0].mean()))
print('corr loop~clamp %.2f loop~near %.2f'%(np.corrcoef(loop,clamp)[0,1], np.corrcoef(loop,near)[0,1]))
