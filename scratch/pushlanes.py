# step time of the pushing kernel by lanes per env (GPR_PUSH_LANES, read at every launch) and batch size; '' = the library's own choice
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
import gymnasium_planar_robotics_amd as gpr
prec = sys.argv[1] if len(sys.argv) > 1 else 'f32'
for NM in (1, 2):
    for E in (4096, 16384, 65536):
        env = gpr.VecBenchmarkPushingEnv(num_envs=E, num_movers=NM, precision=prec); env.reset(seed=0)
        env.elapsed_steps.copy_(torch.randint(0, 50, (E,), device='cuda', dtype=torch.int32))
        acts = [(torch.rand(E, 2 * NM, device='cuda') * 20 - 10).to(env.dtype) for _ in range(8)]
        res = []
        for lanes in ('', '1', '2', '4', '8', '16', '32', '64'):
            if lanes and int(lanes) * E > 64 * 1024 * 8: continue
            if lanes: os.environ['GPR_PUSH_LANES'] = lanes
            else: os.environ.pop('GPR_PUSH_LANES', None)
            for i in range(20): env.step(acts[i % 8])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(60): env.step(acts[i % 8])
            e1.record(); torch.cuda.synchronize()
            res.append(f'{lanes or "auto"}: {e0.elapsed_time(e1) / 60 * 1e3:.1f}')
        print(f'{prec} pushing {NM} mover(s), {E:6d} envs, us/step by lanes per env | ' + ' | '.join(res), flush=True)
        env.close()
