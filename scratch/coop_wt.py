# per-wavefront clocks of the 16-mover step (timing builds; GPR_LIB_PATH selects): where the queue service goes
import sys, os, numpy as np, torch
os.environ['GPR_WAVE_TIMING'] = 'gpurun_out/wave_timing.bin'
sys.path.insert(0, '.')
import gymnasium_planar_robotics_amd as gpr
E = int(os.environ.get('WT_ENVS', 4096))
env = gpr.VecBenchmarkPlanningEnv(np.ones((10, 10)), 16, num_envs=E); env.reset(seed=0)
env.elapsed_steps.copy_(torch.randint(0, 50, (E,), device='cuda', dtype=torch.int32))
acts = [(torch.rand(E, 32, device='cuda') * 20 - 10) for _ in range(8)]
for i in range(150): env.step_into(acts[i % 8], env._out)
torch.cuda.synchronize(); env.close()
f = np.fromfile('gpurun_out/wave_timing.bin', dtype=np.uint32).reshape(-1, 16).astype(np.float64)
def stat(name, x): print(f'  {name:34s} mean {x.mean():8.0f}  p50 {np.median(x):8.0f}  p90 {np.quantile(x, 0.9):8.0f}  p99 {np.quantile(x, 0.99):8.0f}  max {x.max():8.0f}')
print(os.path.basename(os.environ.get('GPR_LIB_PATH', 'default')), E, 'envs x 16 movers, clocks (100 MHz s_memtime ticks x ?) per wavefront, last step')
stat('cycles + own stores', f[:, 0]); stat('queue service / wait', f[:, 1]); stat('resets formed', f[:, 4])
if 'nocoop' not in os.environ.get('GPR_LIB_PATH', ''):
    stat('  scanning', f[:, 2]); stat('  in searches', f[:, 3]); stat('  forming resets', f[:, 5]); stat('  polls with nothing to do', f[:, 6]); stat('  searches joined', f[:, 7]); stat('  of which to help', f[:, 11])
t0 = f[:, 12]; t1 = f[:, 13]; print('  kernel span (first start to last end):', ((t1 - t0.min()) % 2**32).max())
# per block (4 consecutive wavefronts): is the queue service balanced inside the block, and how do the blocks compare?
blk = (f[:, 0] + f[:, 1]).reshape(-1, 4)
srv = f[:, 1].reshape(-1, 4)
print('  per block: total clocks of its slowest wavefront'); stat('    slowest wavefront of the block', blk.max(axis=1)); stat('    mean wavefront of the block', blk.mean(axis=1))
stat('    spread inside the block (max - min of queue service)', srv.max(axis=1) - srv.min(axis=1))
formed = f[:, 4].reshape(-1, 4).sum(axis=1); stat('    resets formed per block', formed)
c = np.corrcoef(formed, blk.max(axis=1))[0, 1]; print(f'    correlation(resets of the block, its slowest wavefront) = {c:.2f}')
if 'nocoop' not in os.environ.get('GPR_LIB_PATH', ''):
    ss = f[:, 3].reshape(-1, 4).sum(axis=1); stat('    search clocks per block (sum over its wavefronts)', ss); stat('    search clocks per reset', ss / np.maximum(formed, 1))
