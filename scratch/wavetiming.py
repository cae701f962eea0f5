# per-wavefront clocks of the step kernel (needs: make -C gymnasium-planar-robotics_amd/csrc clean && make ... EXTRA=-DGPR_WAVE_TIMING)
import sys, os, numpy as np, torch
os.environ['GPR_WAVE_TIMING'] = 'gpurun_out/wave_timing.bin'
sys.path.insert(0, '.')
import gymnasium_planar_robotics_amd as gpr
E, M = int(os.environ.get('WT_ENVS', 65536)), int(os.environ.get('WT_MOVERS', 4))
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-5
cp = {'shape': 'box', 'size': np.array([0.0775, 0.0775])} if (len(sys.argv) > 2 and sys.argv[2] == 'box') else None
layout = np.ones(({1:3,2:4,4:6,8:8,16:10}[M],) * 2)
if os.environ.get('WT_HOLES'): layout[2, 2] = 0; layout[3, 4] = 0
SUB = int(os.environ.get('WT_SUB', 1))
if SUB > 1:  # bench.py's way: WT_SUB sub-batches on their own streams, one gpr_step_many call per step (the dump is the LAST sub-batch's kernel)
    import argparse, bench
    args = argparse.Namespace(movers=M, tiles=layout.shape[0], precision=os.environ.get('PREC', 'f32'), steps=300, warmup=60, min_seconds=0.1, extra_seconds=0.1, regions=1)
    p = bench.Pass(args, 0, 0, 1, envs=E, sub=SUB, std_noise=sigma, collision_params=cp)
    p.do_steps(600); torch.cuda.synchronize(); p.close()
else:
    env = gpr.VecBenchmarkPlanningEnv(layout, M, num_envs=E, std_noise=sigma, collision_params=cp, precision=os.environ.get('PREC', 'f32'))
    env.reset(seed=0)
    env.elapsed_steps.copy_(torch.randint(0, 50, (E,), device='cuda', dtype=torch.int32))
    acts = [(torch.rand(E, 2 * M, device='cuda') * 20 - 10).to(env.dtype) for _ in range(8)]
    for i in range(150): env.step_into(acts[i % 8], env._out)
    torch.cuda.synchronize(); env.close()
d = np.fromfile('gpurun_out/wave_timing.bin', dtype=np.uint32).reshape(-1, 16)
f = d.astype(np.float64)
tot, serve, clamp, near, served, rows, direct, due = (f[:, i] for i in range(8))
pro, loop, epi, exact = f[:, 8], f[:, 9], f[:, 10], f[:, 11]
print(f'sigma {sigma}: clocks per wavefront in the last of 150 steps ({len(d)} wavefronts)')
def stat(name, x): print(f'  {name:30s} mean {x.mean():8.0f}  p50 {np.median(x):8.0f}  p90 {np.quantile(x, 0.9):8.0f}  p99 {np.quantile(x, 0.99):8.0f}  max {x.max():8.0f}')
stat('prologue (loads, setup)', pro); stat('cycles loop', loop); stat('observe + stores', epi); stat('queue service / wait', serve); stat('total', tot + serve)
t0 = d[:, 12].astype(np.int64); t1 = d[:, 13].astype(np.int64)
base = t0.min(); st = (t0 - base) & 0xFFFFFFFF; en = (t1 - base) & 0xFFFFFFFF
stat('start after first wavefront', st.astype(np.float64)); stat('end after first start', en.astype(np.float64))
simd = (d[:, 15].astype(np.int64) & 0xF) << 16 | (d[:, 14].astype(np.int64) & 0xFF30)
u, inv, cnt = np.unique(simd, return_inverse=True, return_counts=True)
print(f'  SIMDs used {len(u)}; wavefronts per SIMD: min {cnt.min()} max {cnt.max()}')
mk = np.zeros(len(u)); np.maximum.at(mk, inv, en.astype(np.float64))
stat('per-SIMD finish', mk)
for lo, hi in ((0, 0), (1, 10), (11, 39), (40, 40)):
    m = (exact >= lo) & (exact <= hi)
    if m.any(): print(f'  wavefronts with {lo}-{hi} cycles on the exact path: {m.mean() * 100:5.1f} %, loop mean {loop[m].mean():.0f}; clamping block-cycles mean {clamp[m].mean():.1f}')
for lo, hi in ((0, 0), (1, 39), (40, 400)):
    m = (clamp >= lo) & (clamp <= hi) & (exact == 0)
    if m.any(): print(f'  all-fast wavefronts with {lo}-{hi} clamp cycles: {m.mean() * 100:5.1f} %, loop mean {loop[m].mean():.0f}')
print(f'  envs served per wavefront: mean {served.mean():.1f}; replayed blocks per wavefront: mean {near.mean():.2f}; exact cycles mean {exact.mean():.1f}')
top = np.argsort(-(tot + serve))[:10]
print('  the slowest wavefronts: start / prologue / loop / epilogue / queue | clamp cycles / exact cycles / replays / rows / served')
for i in top: print(f'    {st[i]:6d} {pro[i]:6.0f} {loop[i]:7.0f} {epi[i]:6.0f} {serve[i]:7.0f} | {clamp[i]:3.0f} {exact[i]:3.0f} {near[i]:3.0f} {rows[i]:2.0f} {served[i]:3.0f}')
A = np.stack([np.ones_like(loop), clamp, exact, rows, near], axis=1)
coef, *_ = np.linalg.lstsq(A, loop, rcond=None)
print('  least squares: loop = %.0f + %.0f per clamp block-cycle + %.0f per exact cycle + %.0f per row built + %.0f per replayed block' % tuple(coef))
res = loop - A @ coef
stat('residual of the fit', res)
# does a wavefront's residual follow the load of the other wavefronts on its SIMD?
load = np.zeros(len(u)); np.add.at(load, inv, loop)
others = load[inv] - loop
print('  correlation(residual, loop clocks of the other wavefronts on the SIMD) = %.2f' % np.corrcoef(res, others)[0, 1])
worst = np.argsort(-res)[:6]
print('  largest residuals: loop / predicted / clamp / exact / replays / rows / direct / due / others-on-SIMD mean loop')
for i in worst: print(f'    {loop[i]:7.0f} {(A @ coef)[i]:7.0f} {clamp[i]:3.0f} {exact[i]:3.0f} {near[i]:3.0f} {rows[i]:2.0f} {direct[i]:3.0f} {due[i]:3.0f} {others[i] / 3:7.0f}')
