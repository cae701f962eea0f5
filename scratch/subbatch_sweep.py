# how many sub-batches (handles + streams, one gpr_step_many call per env-step) should a rank's 65 536 envs be stepped as?
import json, subprocess, sys
for prec in ('f32', 'f64'):
    for S in (1, 2, 4, 8, 16):
        r = subprocess.run([sys.executable, 'bench.py', '--no-extras', '--no-cpu-baseline', '--sub-batches', str(S), '--precision', prec,
                            '--steps', '1000' if prec == 'f32' else '300', '--min-seconds', '0.3'], capture_output=True, text=True)
        try:
            d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
            print(prec, 'S=%2d' % S, 'value %.3e' % d['value'], 'ms_per_step %.5f' % d['ms_per_step'], 'gpu step_us %.2f' % d['roofline']['kernel_us'], 'frac %.4f' % d['roofline']['frac'], flush=True)
        except Exception as e:
            print(prec, S, 'failed', r.stderr[-800:], flush=True)
