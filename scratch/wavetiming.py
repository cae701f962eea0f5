# per-wavefront clocks of the step kernel (needs: make -C gymnasium-planar-robotics_amd/csrc clean && make ... EXTRA=-DGPR_WAVE_TIMING)
import sys, os, numpy as np, torch
os.environ['GPR_WAVE_TIMING'] = 'gpurun_out/wave_timing.bin'
sys.path.insert(0, '.')
import gymnasium_planar_robotics_amd as gpr
E, M = 65536, 4
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-5
cp = {'shape': 'box', 'size': np.array([0.0775, 0.0775])} if (len(sys.argv) > 2 and sys.argv[2] == 'box') else None
env = gpr.VecBenchmarkPlanningEnv(np.ones((6, 6)), M, num_envs=E, std_noise=sigma, collision_params=cp)
env.reset(seed=0)
acts = [(torch.rand(E, 2 * M, device='cuda') * 20 - 10).to(env.dtype) for _ in range(8)]
for i in range(150): env.step_into(acts[i % 8], env._out)
torch.cuda.synchronize(); env.close()
d = np.fromfile('gpurun_out/wave_timing.bin', dtype=np.uint32).reshape(-1, 8).astype(np.float64)
cyc, serve, clamp, near, served = d[:, 0], d[:, 1], d[:, 2], d[:, 3], d[:, 4]
print(f'sigma {sigma}: clocks per wavefront in the last of 150 steps ({len(d)} wavefronts)')
for name, x in (('cycles + own stores', cyc), ('queue service / wait', serve), ('total', cyc + serve)):
    print(f'  {name:22s} mean {x.mean():8.0f}  p50 {np.median(x):8.0f}  p99 {np.quantile(x, 0.99):8.0f}  max {x.max():8.0f}')
blk = (cyc + serve).reshape(-1, 4)
print(f'  per block: last wavefront done after mean {blk.max(axis=1).mean():.0f}, max {blk.max():.0f}; slowest cycles per block mean {cyc.reshape(-1,4).max(axis=1).mean():.0f}')
for lo, hi in ((0, 0), (1, 10), (11, 39), (40, 40)):
    m = (clamp >= lo) & (clamp <= hi)
    if m.any(): print(f'  wavefronts clamping in {lo}-{hi} cycles: {m.mean() * 100:5.1f} %, cycles phase mean {cyc[m].mean():.0f}')
print(f'  envs served per wavefront: mean {served.mean():.1f}; near-collision cycles per wavefront: mean {near.mean():.1f}')
rows, direct, due = d[:, 5], d[:, 6], d[:, 7]
top = np.argsort(-cyc)[:40]
print('  the slowest wavefronts (cycles phase): clocks / clamp cycles / clamp looks / near cycles / LDS rows built / per-cycle draws')
for i in top[:8]: print(f'    {cyc[i]:7.0f} {clamp[i]:3.0f} {due[i]:3.0f} {near[i]:3.0f} {rows[i]:2.0f} {direct[i]:3.0f}')
A = np.stack([np.ones_like(cyc), clamp, near, rows, direct, due - clamp], axis=1)
coef, *_ = np.linalg.lstsq(A, cyc, rcond=None)
print('  least squares: cycles phase = %.0f + %.0f per clamp cycle + %.0f per near cycle + %.0f per row built + %.0f per per-cycle draw + %.0f per idle clamp look' % tuple(coef))
