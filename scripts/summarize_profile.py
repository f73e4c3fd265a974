"""Condense a gpu_profile.sh run into the small files that get committed under profiles/."""
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
commit = sys.argv[3] if len(sys.argv) > 3 else None       # the commit the profiled snapshot was taken at (the box has no .git)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402  (flux_source_sha16: fingerprint of the kernel sources these counters belong to)
measured_at = {'commit': commit, 'flux_source_sha16': bench.flux_source_sha16(), 'tag': tag}
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
# ... and of the file-ingest kernels, from the same kind of pass over tools/inflate_rate.py (7 kinds of streams, 4 + 256 streams
# each: the averages mix the kinds -- per-kind durations are in <tag>_numbers.md / pmc_traffic.json -> ingest)
ingest_rows = []
st2 = find('ingest_trace', '*kernel_stats.csv')
if st2:
    with open(st2) as f:
        for r in csv.DictReader(f):
            if 'k_inflate' in r['Name'] or 'k_place' in r['Name']:
                ingest_rows.append(dict(kernel=short(r['Name']) + ' [tools/inflate_rate.py]', calls=int(r['Calls']),
                                        total_ns=int(r['TotalDurationNs']), avg_ns=float(r['AverageNs']),
                                        pct=float(r['Percentage']), min_ns=int(r['MinNs']), max_ns=int(r['MaxNs'])))
with open(os.path.join(out, f'{tag}_kernel_stats.csv'), 'w') as f:
    f.write(f'# rocprofv3 --kernel-trace --stats at commit {commit}, flux kernel sources {measured_at["flux_source_sha16"]}\n')
    f.write('kernel,calls,total_ns,avg_ns,pct,min_ns,max_ns\n')
    for r in rows + ingest_rows:
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

# ---- BASELINE config C3 (bench.py --only-c3): the one-field flux kernel, counters from its own two passes
for dtype, kname, es in (('f64', 'k_flux_field<double', 8), ('f32', 'k_flux_field<float', 4)):
    fetch, write = counter('c3_fetch', 'FETCH_SIZE', kname), counter('c3_write', 'WRITE_SIZE', kname)
    if fetch and write:
        f_kb, w_kb = sum(fetch) / len(fetch), sum(write) / len(write)
        res[f'1440x1021x75x1_{dtype}'] = dict(kernel='nf::k_flux_field', launches_sampled=[len(fetch), len(write)],
                                             FETCH_SIZE_KiB_avg=f_kb, WRITE_SIZE_KiB_avg=w_kb,
                                             fetch_correction='x2 (gfx950, 16 B/lane coalesced stream)',
                                             hbm_bytes_per_launch=2.0 * f_kb * 1024 + w_kb * 1024,
                                             algorithmic_bytes_per_launch=(2 * es + 64.0 / 75) * 75 * 1021 * 1440)

# ---- K3 (nf::k_gather_segscan: both dtypes' passes share it): measured bytes against the algorithmic 80 B per record
# (40 B of record + 4 x 8 B gathered + 8 B out).  The record stream is read 16 B per lane (counted at half by FETCH_SIZE on
# gfx950), the gathers are 8-B accesses that pull whole 64-B requests (counted in full): the true figure lies between the
# raw and the doubled counter; both are recorded.
nrec = c.get('weight_entries', 0) // 4
kf, kw = counter('pmc_fetch', 'FETCH_SIZE', 'k_gather_segscan'), counter('pmc_write', 'WRITE_SIZE', 'k_gather_segscan')
if kf and nrec:       # (rocprofv3 leaves a kernel out of the WRITE_SIZE file when the counter reads 0: run sums go out non-temporal)
    f_kb, w_kb = sum(kf) / len(kf), (sum(kw) / len(kw) if kw else 0.0)
    alg = 80.0 * nrec
    res['k_gather_segscan'] = dict(kernel='nf::k_gather_segscan', records=nrec, launches_sampled=[len(kf), len(kw)],
                                   FETCH_SIZE_KiB_avg=f_kb, WRITE_SIZE_KiB_avg=w_kb,
                                   hbm_bytes_per_launch_raw=f_kb * 1024 + w_kb * 1024,
                                   hbm_bytes_per_launch_fetch_x2=2 * f_kb * 1024 + w_kb * 1024,
                                   algorithmic_bytes_per_launch=alg, record_stream_bytes=40.0 * nrec,
                                   overfetch_raw=(f_kb * 1024 + w_kb * 1024) / alg,
                                   overfetch_fetch_x2=(2 * f_kb * 1024 + w_kb * 1024) / alg)


# ---- file ingest (tools/inflate_rate.py): per case two launches (4 streams to warm up, 256 measured)
def per_launch(sub, cname, kernel):
    fn = find(sub, '*counter_collection.csv')
    acc = {}
    if not fn:
        return []
    with open(fn) as f:
        for r in csv.DictReader(f):
            if kernel in r.get('Kernel_Name', '') and r.get('Counter_Name') == cname:
                acc.setdefault(int(r['Dispatch_Id']), 0.0)
                acc[int(r['Dispatch_Id'])] += float(r['Counter_Value'])
    return [acc[k] for k in sorted(acc)]


def durations(sub, kernel):
    fn = find(sub, '*kernel_trace.csv')
    if not fn:
        return []
    with open(fn) as f:
        rows_ = [r for r in csv.DictReader(f) if kernel in r['Kernel_Name']]
    return [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows_]


CASES = ['plane0 stored', 'plane2 literals+short matches', 'plane3 long matches', 'whole level (5.88 MB, 4 planes)',
         'literals, 4-bit codes', 'noise (stored)', 'matches of 8 bytes']
ingest = {}
dur_i, dur_p = durations('ingest_trace', 'k_inflate'), durations('ingest_trace', 'k_place')
fi, wi = per_launch('ingest_fetch', 'FETCH_SIZE', 'k_inflate'), per_launch('ingest_write', 'WRITE_SIZE', 'k_inflate')
fp, wp = per_launch('ingest_fetch', 'FETCH_SIZE', 'k_place'), per_launch('ingest_write', 'WRITE_SIZE', 'k_place')
log = os.path.join(src, 'ingest_trace.log')
sizes = []
if os.path.exists(log):
    for line in open(log):
        if ' bytes from ' in line:
            a = line.rsplit(':', 1)[1].split()
            sizes.append((int(a[0]), int(a[3])))
for k, name in enumerate(CASES):
    j = 2 * k + 1                        # the 256-stream launch of the case
    if j >= len(dur_i) or k >= len(sizes):
        break
    out_b, in_b = sizes[k]
    e = dict(streams=256, decoded_bytes_per_stream=out_b, compressed_bytes_per_stream=in_b, k_inflate_ms=dur_i[j],
             k_place_ms=dur_p[j] if j < len(dur_p) else None, MB_per_s_per_stream=out_b / dur_i[j] / 1e3)
    if j < len(fi) and j < len(wi):
        e.update(k_inflate_FETCH_SIZE_KiB=fi[j], k_inflate_WRITE_SIZE_KiB=wi[j],
                 k_inflate_bytes_algorithmic=256.0 * (in_b + out_b))
    if j < len(fp) and j < len(wp):
        e.update(k_place_FETCH_SIZE_KiB=fp[j], k_place_WRITE_SIZE_KiB=wp[j], k_place_bytes_algorithmic=512.0 * out_b)
    ingest[name] = e
if ingest:
    res['ingest'] = ingest
res['measured_at'] = measured_at
with open(os.path.join(out, 'pmc_traffic.json'), 'w') as f:
    json.dump(res, f, indent=1)

# ---- the numbers profiles/README.md and DESIGN.md quote, straight from the files above
def stat(kernel_sub):
    for r in rows:
        if kernel_sub in r['kernel']:
            return r
    return None


lines = [f'# {tag}: numbers quoted in profiles/README.md and DESIGN.md (generated by scripts/summarize_profile.py)', '',
         f'Profiled snapshot: commit `{commit}`, flux kernel sources `{measured_at["flux_source_sha16"]}` (`bench.flux_source_sha16()`).', '']
lines.append('| kernel (rocprofv3 --kernel-trace --stats of `bench.py --steps 5 --warmup 1 --no-cpu`) | calls | avg ms |')
lines.append('|---|---|---|')
for sub in ('k_flux<double', 'k_flux<float', 'k_expand_planes', 'k_gather_segscan', 'k_finalize_seg', 'k_finalize_tr', 'k_geometry',
            'k_clip', 'k_uv'):
    r = stat(sub)
    if r:
        lines.append(f"| `{r['kernel'][:70]}` | {r['calls']} | {r['avg_ns'] / 1e6:.4f} |")
# BASELINE config C3 beside the headline: its own trace pass (bench.py --only-c3), events against the CSV
c3_csv = find('c3_trace', '*kernel_stats.csv')
c3_json = os.path.join(src, 'bench_c3.json')
if c3_csv and os.path.exists(c3_json):
    with open(c3_json) as f:
        c3 = json.loads([l for l in f if l.startswith('{')][0])['c3']
    with open(c3_csv) as f:
        c3rows = {short(r['Name']): r for r in csv.DictReader(f)}
    lines.append('')
    lines.append('| kernel (`bench.py --only-c3 --steps 20`: 1440 x 1021 x 75, one time step) | calls | avg ms (CSV) | HIP events in that run | frac of 8 TB/s (CSV) | HBM traffic (two --pmc passes) |')
    lines.append('|---|---|---|---|---|---|')
    for dtype, key in (('f64', 'k_flux_field<double'), ('f32', 'k_flux_field<float')):
        hit = [r for n, r in c3rows.items() if key in n]
        if hit and dtype in c3:
            t = float(hit[0]['AverageNs']) / 1e6
            alg = c3[dtype]['algorithmic_bytes_per_unit'] * 75 * 1021 * 1440
            pm = res.get(f'1440x1021x75x1_{dtype}', {}).get('hbm_bytes_per_launch')
            lines.append(f"| `nf::{key}, ...>` | {hit[0]['Calls']} | {t:.4f} | {c3[dtype]['k1_ms']:.4f} | {alg / (t * 1e-3) / 8e12:.4f} |" +
                         (f" {pm / 1e9:.4f} GB vs {alg / 1e9:.4f} GB algorithmic ({pm / alg:.3f}) |" if pm else ' |'))
lines.append('')
for dtype, roof in legs:
    s_ = 8 if dtype == 'f64' else 4
    alg = roof['algorithmic_bytes_per_unit'] * c['nz'] * c['ny'] * c['nx']
    kname = 'k_flux<double' if dtype == 'f64' else 'k_flux<float'
    r = stat(kname)
    x = stat('k_expand_planes') if roof['avg_ms_by_kernel'].get('nf::k_expand_planes', 0) > 0 else None
    t_csv = (r['avg_ns'] + (x['avg_ns'] if x else 0)) / 1e6 if r else None
    lines.append(f"* {dtype}: algorithmic {alg / 1e9:.4f} GB per launch; HIP events in the profiled run {roof['avg_launch_ms']:.4f} ms -> "
                 f"frac {roof['frac']:.4f}; kernel-stats CSV {t_csv:.4f} ms -> frac {alg / (t_csv * 1e-3) / 8e12:.4f}; "
                 f"PMC traffic {res.get(str(c['nx']) + 'x' + str(c['ny']) + 'x' + str(c['nz']) + 'x' + str(c['nt_global']) + '_' + dtype, {}).get('hbm_bytes_per_launch', float('nan')) / 1e9:.4f} GB"
                 if t_csv else f'* {dtype}: no kernel stats')
if 'k_gather_segscan' in res:
    k3 = res['k_gather_segscan']
    r = stat('k_gather_segscan')
    lines.append(f"* K3 `k_gather_segscan`: {k3['records']} records, algorithmic {k3['algorithmic_bytes_per_launch'] / 1e6:.1f} MB "
                 f"(record stream {k3['record_stream_bytes'] / 1e6:.1f} MB); FETCH_SIZE {k3['FETCH_SIZE_KiB_avg'] * 1024 / 1e6:.1f} MB raw, "
                 f"WRITE_SIZE {k3['WRITE_SIZE_KiB_avg'] * 1024 / 1e6:.1f} MB -> over-fetch {k3['overfetch_raw']:.2f} (raw) .. "
                 f"{k3['overfetch_fetch_x2']:.2f} (FETCH_SIZE doubled); {r['avg_ns'] / 1e3:.1f} us per launch -> "
                 f"{k3['algorithmic_bytes_per_launch'] / r['avg_ns']:.0f} GB/s algorithmic, "
                 f"{k3['hbm_bytes_per_launch_raw'] / r['avg_ns']:.0f} .. {k3['hbm_bytes_per_launch_fetch_x2'] / r['avg_ns']:.0f} GB/s measured" if r else '')
for name, e in ingest.items():
    extra = ''
    if 'k_inflate_FETCH_SIZE_KiB' in e:
        extra = (f"; k_inflate FETCH_SIZE {e['k_inflate_FETCH_SIZE_KiB'] * 1024 / 1e6:.0f} MB raw + WRITE_SIZE "
                 f"{e['k_inflate_WRITE_SIZE_KiB'] * 1024 / 1e6:.0f} MB vs {e['k_inflate_bytes_algorithmic'] / 1e6:.0f} MB in + out")
    if 'k_place_FETCH_SIZE_KiB' in e:
        extra += (f"; k_place FETCH_SIZE {e['k_place_FETCH_SIZE_KiB'] * 1024 / 1e6:.0f} MB raw + WRITE_SIZE "
                  f"{e['k_place_WRITE_SIZE_KiB'] * 1024 / 1e6:.0f} MB vs {e['k_place_bytes_algorithmic'] / 1e6:.0f} MB")
    lines.append(f"* ingest, {name}: k_inflate {e['k_inflate_ms']:.2f} ms for 256 streams = one stream at {e['MB_per_s_per_stream']:.1f} MB/s"
                 + (f", k_place {e['k_place_ms']:.2f} ms" if e.get('k_place_ms') else '') + extra)
with open(os.path.join(out, f'{tag}_numbers.md'), 'w') as f:
    f.write('\n'.join(lines) + '\n')
print('\n'.join(lines))
for fn in ('bench_trace.json',):
    with open(os.path.join(src, fn)) as f, open(os.path.join(out, f'{tag}_{fn}'), 'w') as g:
        g.write(f.read())
print(json.dumps(res, indent=1))
for r in rows[:8]:
    print(r)
