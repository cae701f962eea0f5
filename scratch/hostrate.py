# is a per-launch step at small batches bound by the host's call path? time step_into (Python -> ctypes -> gpr_step -> launch) at
# several batch sizes, and the bare C call in a tight ctypes loop
import sys, time, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
import gymnasium_planar_robotics_amd as gpr
for E in (64, 1024, 4096, 8192, 16384):
    env = gpr.VecBenchmarkPlanningEnv(np.ones((6, 6)), 4, num_envs=E); env.reset(seed=0)
    env.elapsed_steps.copy_(torch.randint(0, 50, (E,), device='cuda', dtype=torch.int32))
    acts = [torch.rand(E, 8, device='cuda') * 20 - 10 for _ in range(16)]
    a = acts[0]
    for i in range(200): env.step_into(acts[i % 16], env._out)
    torch.cuda.synchronize()
    n = 4000
    t0 = time.perf_counter()
    for i in range(n): env.step_into(acts[i % 16], env._out)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    lib, h = gpr._native.load(), env._handle
    stream = torch.cuda.current_stream().cuda_stream
    out = C.byref(env._out); aps = [C.c_void_p(x.data_ptr()) for x in acts]
    t0 = time.perf_counter()
    for i in range(n): lib.gpr_step(h, aps[i % 16], out, stream)
    t_c_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_c_all = time.perf_counter() - t0
    print(f'E={E:6d}: step_into issue {t_issue / n * 1e6:.2f} us/call, done {t_all / n * 1e6:.2f} us/step | bare gpr_step via ctypes: issue {t_c_issue / n * 1e6:.2f}, done {t_c_all / n * 1e6:.2f}', flush=True)
    env.close()
