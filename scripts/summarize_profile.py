"""Condense a gpu_profile.sh run into the small files that get committed under profiles/."""
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(src, 'summary')
os.makedirs(out, exist_ok=True)


def find(sub, pat):
    fs = glob.glob(os.path.join(src, sub, '**', pat), recursive=True)
    return fs[0] if fs else None


def short(name):
    name = name.split('(')[0]
    if 'rocprim' in name:
        return 'rocprim::radix_sort_onesweep'
    return name.replace('void ', '').strip()


# kernel stats (rocprofv3 --kernel-trace --stats)
st = find('trace', '*kernel_stats.csv')
rows = []
if st:
    with open(st) as f:
        for r in csv.DictReader(f):
            rows.append(dict(kernel=short(r['Name']), calls=int(r['Calls']), total_ns=int(r['TotalDurationNs']),
                             avg_ns=float(r['AverageNs']), pct=float(r['Percentage']), min_ns=int(r['MinNs']),
                             max_ns=int(r['MaxNs'])))
with open(os.path.join(out, f'{tag}_kernel_stats.csv'), 'w') as f:
    f.write('kernel,calls,total_ns,avg_ns,pct,min_ns,max_ns\n')
    for r in rows:
        f.write('"{kernel}",{calls},{total_ns},{avg_ns:.1f},{pct},{min_ns},{max_ns}\n'.format(**r))


def counter(sub, cname, kernel):
    fn = find(sub, '*counter_collection.csv')
    vals = []
    if not fn:
        return vals
    with open(fn) as f:
        for r in csv.DictReader(f):
            if kernel in r.get('Kernel_Name', '') and r.get('Counter_Name') == cname:
                vals.append(float(r['Counter_Value']))
    return vals


# one time step = the flux kernel, plus -- in the split store form (float32 default) -- the streaming expansion of the
# four derived planes: bench.py's HIP events bracket whatever a step launches, so that is what the traffic is summed over.
# bench.py runs the float64 workload and then the float32 one in the same process: the flux kernels are told apart by
# their template argument, the expansion kernel belongs to whichever dtype runs the split form.
bj = json.load(open(os.path.join(src, 'bench_trace.json')))
c = bj['config']
res = {}
legs = [(bj['dtype'], bj['roofline'])]
if 'f32' in bj and bj['dtype'] == 'f64':
    legs.append(('f32', bj['f32']['roofline']))
for dtype, roof in legs:
    kname = 'k_flux<double' if dtype == 'f64' else 'k_flux<float'
    fetch, write = counter('pmc_fetch', 'FETCH_SIZE', kname), counter('pmc_write', 'WRITE_SIZE', kname)
    if not (fetch and write):
        continue
    split = roof['avg_ms_by_kernel'].get('nf::k_expand_planes', 0) > 0
    xfetch = counter('pmc_fetch', 'FETCH_SIZE', 'k_expand_planes') if split else []
    xwrite = counter('pmc_write', 'WRITE_SIZE', 'k_expand_planes') if split else []
    f_kb, w_kb = sum(fetch) / len(fetch), sum(write) / len(write)
    xf_kb = sum(xfetch) / len(xfetch) if xfetch else 0.0
    xw_kb = sum(xwrite) / len(xwrite) if xwrite else 0.0
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the
    # bytes of a wide coalesced (16 B/lane) streaming read -> double it; WRITE_SIZE is exact for 16 B/lane stores.
    hbm = 2.0 * f_kb * 1024 + w_kb * 1024 + 2.0 * xf_kb * 1024 + xw_kb * 1024
    key = f"{c['nx']}x{c['ny']}x{c['nz']}x{c['nt_global']}_{dtype}"
    res[key] = dict(kernel='nf::k_flux + nf::k_expand_planes' if split else 'nf::k_flux', launches_sampled=[len(fetch), len(write)],
                    FETCH_SIZE_KiB_avg=f_kb, WRITE_SIZE_KiB_avg=w_kb, expand_FETCH_SIZE_KiB_avg=xf_kb,
                    expand_WRITE_SIZE_KiB_avg=xw_kb, fetch_correction='x2 (gfx950, 16 B/lane coalesced stream)',
                    hbm_bytes_per_launch=hbm,
                    algorithmic_bytes_per_launch=roof['algorithmic_bytes_per_unit'] * c['nz'] * c['ny'] * c['nx'])
with open(os.path.join(out, 'pmc_traffic.json'), 'w') as f:
    json.dump(res, f, indent=1)
for fn in ('bench_trace.json',):
    with open(os.path.join(src, fn)) as f, open(os.path.join(out, f'{tag}_{fn}'), 'w') as g:
        g.write(f.read())
print(json.dumps(res, indent=1))
for r in rows[:8]:
    print(r)
