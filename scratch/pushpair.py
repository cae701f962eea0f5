# two-mover pushing kernel: one lane per (mover, object) pair (GPR_PUSH_PAIR=1) vs both pairs in every lane (=0), by batch size and lanes per env
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
import gymnasium_planar_robotics_amd as gpr
for E in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '4096 16384 65536 131072 262144').split()]:
    env = gpr.VecBenchmarkPushingEnv(num_envs=E, num_movers=2); env.reset(seed=0)
    env.elapsed_steps.copy_(torch.randint(0, 50, (E,), device='cuda', dtype=torch.int32))
    acts = [(torch.rand(E, 4, device='cuda') * 20 - 10) for _ in range(8)]
    res = []
    for pair, lanes in (('0', ''), ('1', ''), ('1', '2'), ('1', '4'), ('1', '8'), ('1', '16'), ('', '')):
        if lanes and int(lanes) * E > 64 * 1024 * 8: continue
        for k, v in (('GPR_PUSH_PAIR', pair), ('GPR_PUSH_LANES', lanes)):
            if v: os.environ[k] = v
            else: os.environ.pop(k, None)
        for i in range(20): env.step(acts[i % 8])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(60): env.step(acts[i % 8])
        e1.record(); torch.cuda.synchronize()
        res.append(f'pair={pair or "auto"} lanes={lanes or "auto"}: {e0.elapsed_time(e1) / 60 * 1e3:.1f}')
    print(f'f32 pushing 2 movers, {E:6d} envs, us/step | ' + ' | '.join(res), flush=True)
    env.close()
