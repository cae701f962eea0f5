# condenses the rocprofv3 output of scratch/profile_round4.sh into the files committed under profiles/
import csv, collections, glob, json, sys, os
O = sys.argv[1]
out_dir = os.path.join(O, 'summary'); os.makedirs(out_dir, exist_ok=True)

def short(name):
    return name.replace('void ', '').replace('(gpr::KParams)', '').split('(')[0][:110]

rows = []
for tag in ('trace_bench', 'trace_bench_s1', 'trace_bench_f64', 'trace_all'):
    for f in sorted(glob.glob(f'{O}/{tag}/*/*_kernel_stats.csv')):
        for r in csv.DictReader(open(f)):
            if 'gpr::' in r['Name'] or 'k_spin' in r['Name']:
                rows.append((tag, short(r['Name']), int(r['Calls']), float(r['AverageNs']), float(r['MinNs']), float(r['MaxNs'])))
with open(f'{out_dir}/round4_kernel_stats.csv', 'w') as f:
    f.write('source,kernel,calls,avg_ns,min_ns,max_ns\n')
    for r in rows: f.write('%s,"%s",%d,%.0f,%.0f,%.0f\n' % r)

def pmc(tag, pick=('k_step', 'k_push', 'k_agent', 'k_reset', 'k_qpos', 'k_mover', 'k_seg', 'k_rect', 'k_clamp', 'k_reward', 'k_counter')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(f'{O}/{tag}/*/*_counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            if any(p in r['Kernel_Name'] for p in pick):
                agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return agg

def second_half_mean(vals):
    h = vals[len(vals) // 2:]
    return sum(h) / max(1, len(h))

tags = [t for t in ('pmc_fetch_s4', 'pmc_write_s4', 'pmc_sq_s4', 'pmc_fetch_s1', 'pmc_write_s1', 'pmc_sq_s1', 'pmc_fetch_f64', 'pmc_write_f64', 'pmc_sq_f64',
                    'pmc_sq_all', 'pmc_fetch_all', 'pmc_write_all')]
with open(f'{out_dir}/round4_pmc_summary.txt', 'w') as f:
    for tag in tags:
        f.write(f'== rocprofv3 --pmc pass {tag}: mean per dispatch (second half of the dispatches of each kernel) ==\n')
        for k, v in sorted(pmc(tag).items()):
            f.write(k + '\n')
            waves = second_half_mean(v['SQ_WAVES']) if 'SQ_WAVES' in v else None
            for c, vals in sorted(v.items()):
                m = second_half_mean(vals)
                f.write('   %-22s %.6g%s\n' % (c, m, (' (per wavefront %.1f)' % (m / waves)) if waves else ''))

# per (kernel, envs per launch): HBM traffic and the VALU counters bench.py reads
CASES = [('s4', 'gpr::k_step<float, float, 4, false, true, true, 0>', 16384, 305), ('s1', 'gpr::k_step<float, float, 4, false, true, true, 0>', 65536, 305),
         ('f64', 'gpr::k_step<double, double, 4, false, true, true, 0>', 16384, 597)]
traffic, valu = [], []
for tag, kernel, envs, B in CASES:
    def mean_of(pass_tag, counter):
        for k, v in pmc(pass_tag).items():
            if kernel.replace('gpr::', '') in k and counter in v:
                return second_half_mean(v[counter])
        return None
    fs, ws = mean_of(f'pmc_fetch_{tag}', 'FETCH_SIZE'), mean_of(f'pmc_write_{tag}', 'WRITE_SIZE')
    if fs is not None and ws is not None:
        traffic.append({'kernel': kernel, 'envs': envs, 'movers': 4, 'FETCH_SIZE_KiB': fs, 'WRITE_SIZE_KiB': ws, 'hbm_bytes_per_launch': (2 * fs + ws) * 1024,
                        'algorithmic_bytes_per_launch': B * envs,
                        'note': 'FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md section HBM); separate --pmc passes; mean over the '
                                'second half of the dispatches of `python bench.py --steps 40 --warmup 60 --no-extras` at this launch size'})
    sq = {c: mean_of(f'pmc_sq_{tag}', c) for c in ('SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_ACTIVE_INST_VALU', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY')}
    if sq['SQ_WAVES']:
        valu.append(dict(kernel=kernel, envs=envs, movers=4, **sq, note='rocprofv3 --pmc (SQ counters in one pass), mean per dispatch over the second half of the dispatches; '
                                                                        'SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md)'))
json.dump(traffic, open(f'{out_dir}/round4_hbm_traffic.json', 'w'), indent=1)
json.dump(valu, open(f'{out_dir}/round4_valu.json', 'w'), indent=1)
print(open(f'{out_dir}/round4_kernel_stats.csv').read())
print(json.dumps(valu, indent=1)[:1500])
