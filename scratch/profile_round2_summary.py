# condenses the rocprofv3 output of scratch/profile_round2.sh into the files committed under profiles/
import csv, collections, glob, json, sys, os
O = sys.argv[1]
out_dir = os.path.join(O, 'summary'); os.makedirs(out_dir, exist_ok=True)

def short(name):
    return name.replace('void ', '').replace('(gpr::KParams)', '').split('(')[0][:110]

# 1/2: kernel stats
rows = []
for tag in ('trace_bench', 'trace_all'):
    for f in sorted(glob.glob(f'{O}/{tag}/*/*_kernel_stats.csv')):
        for r in csv.DictReader(open(f)):
            if 'gpr::' in r['Name']:
                rows.append((tag, short(r['Name']), int(r['Calls']), float(r['AverageNs']), float(r['MinNs']), float(r['MaxNs'])))
with open(f'{out_dir}/round2_kernel_stats.csv', 'w') as f:
    f.write('source,kernel,calls,avg_ns,min_ns,max_ns\n')
    for r in rows: f.write('%s,"%s",%d,%.0f,%.0f,%.0f\n' % r)

def pmc(tag, pick=('k_step', 'k_push', 'k_agent', 'k_reset', 'k_qpos', 'k_mover', 'k_seg', 'k_rect', 'k_clamp', 'k_reward', 'k_counter')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(f'{O}/{tag}/*/*_counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            if any(p in r['Kernel_Name'] for p in pick):
                agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return agg

with open(f'{out_dir}/round2_pmc_summary.txt', 'w') as f:
    for tag in ('pmc_fetch', 'pmc_write', 'pmc_tcc', 'pmc_sq_all', 'pmc_fetch_all', 'pmc_write_all'):
        f.write(f'== rocprofv3 --pmc pass {tag}: mean per dispatch (second half of the dispatches of each kernel) ==\n')
        for k, v in sorted(pmc(tag).items()):
            f.write(k + '\n')
            waves = None
            if 'SQ_WAVES' in v:
                w = v['SQ_WAVES']; waves = sum(w[len(w) // 2:]) / max(1, len(w) - len(w) // 2)
            for c, vals in sorted(v.items()):
                h = vals[len(vals) // 2:]; m = sum(h) / len(h)
                f.write('   %-22s %.6g%s\n' % (c, m, (' (per wavefront %.1f)' % (m / waves)) if waves else ''))

# headline HBM traffic: FETCH_SIZE (KiB, doubled on gfx950 as MI355X_MICROARCH.md prescribes) + WRITE_SIZE (KiB)
def mean_of(tag, counter, kernel='k_step<float, float, 4, false, true, true, 0>'):
    for k, v in pmc(tag).items():
        if kernel in k and counter in v:
            h = v[counter][len(v[counter]) // 2:]
            return sum(h) / len(h)
    return None
fs, ws = mean_of('pmc_fetch', 'FETCH_SIZE'), mean_of('pmc_write', 'WRITE_SIZE')
if fs is not None and ws is not None:
    d = {'kernel': 'gpr::k_step<float, float, 4, false, true, true, 0>', 'envs': 65536, 'movers': 4, 'FETCH_SIZE_KiB': fs, 'WRITE_SIZE_KiB': ws,
         'hbm_bytes_per_launch': (2 * fs + ws) * 1024, 'algorithmic_bytes_per_launch': 305 * 65536,
         'note': 'FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md section HBM); separate --pmc passes; '
                 'mean over the second half of 100 dispatches of `python bench.py --steps 40 --warmup 60 --no-extras`'}
    rd, wr = mean_of('pmc_tcc', 'TCC_EA0_RDREQ_sum'), mean_of('pmc_tcc', 'TCC_EA0_WRREQ_sum')
    if rd is not None: d['TCC_EA0_RDREQ'] = rd; d['TCC_EA0_WRREQ'] = wr
    json.dump(d, open(f'{out_dir}/round2_hbm_traffic.json', 'w'), indent=1)
print(open(f'{out_dir}/round2_kernel_stats.csv').read())
