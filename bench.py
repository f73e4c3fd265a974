#!/usr/bin/env python3
"""bench.py -- headline benchmark of the transect-flux hot path on MI355X.

Metric (BASELINE.json): edge-flux integrals/sec = (t,z,j,i) grid points per wall-second through the full-field
pass  A4 (missing->0, vertical integral) + A5 (edge fluxes, 4-slot array, |.|, running max) + A7 (all transects),
inputs resident in HBM; plus the absolute error of every transect total against the closed form (fluxexact).

Workload at N=1 (config C4 of BASELINE.json, the one the metric is quoted on; it fits one 288 GB GPU):
    3600 x 1800 x 75 levels x 12 time steps, float64, psi = (1+10z)(t+1) arctan2(y, x+180)/(2 pi) (singular),
    transects: README.md:51's singular transect + 64 seeded node-snapped polylines (config C5's batch).
A "step" = one pass over all (t,z) slabs this rank owns (12 launches of the flux kernel + the transect
reduction per time step), followed -- for N>1 -- by ONE RCCL all-reduce of the (nt, nseg+ntransect) rows.

Scaling: strong by default -- BASELINE config 4 is the FIXED 3600 x 1800 x 75 x 12 problem "nt x nz sharded over
8 x MI355X": its 900 (t,z) slabs are cut into N contiguous ranges, one per rank (11.7 GB of u,v per GPU at N=8).
--scaling weak gives every rank a C4-sized block of 12 time steps of a 12*N-step series instead.
At N=1 the line also carries a float32 sub-record (the dtype of real NEMO files) and the CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: starts its own N ranks, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N
"""
import argparse
import json
import os
import sys
import time

import numpy

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def flux_source_sha16():
    """Fingerprint of what the flux kernels are compiled from (nf_flux.hip, the marked parts of nf_common.h -- constants, XCD
    tile map, launch arguments --, the compiler flags of the Makefile):
    the HBM-traffic figures under profiles/ are counters of ANOTHER run, so they carry this fingerprint and the commit they
    were taken at, and the bench reports them only while the sources still are what was measured (round-4 verdict W7)."""
    import hashlib
    import re
    h = hashlib.sha256()
    base = os.path.join(ROOT, 'nemoflux_amd', 'csrc')
    with open(os.path.join(base, 'nf_flux.hip'), 'rb') as f:
        h.update(f.read())
    with open(os.path.join(base, 'nf_common.h')) as f:      # only what K1 uses of the shared header (marked there)
        for part in re.findall(r'\[flux-fingerprint-begin\](.*?)\[flux-fingerprint-end\]', f.read(), re.S):
            h.update(part.encode())
    with open(os.path.join(base, 'Makefile')) as f:
        h.update(''.join(l for l in f if l.startswith('CXXFLAGS')).encode())
    return h.hexdigest()[:16]


def pmc_traffic(key):
    """(HBM bytes per launch or None, where the figure comes from) for workload `key` of profiles/pmc_traffic.json."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        d = json.load(f)
    val = d.get(key, {}).get('hbm_bytes_per_launch')
    if val is None:
        return None, None
    at = d.get('measured_at', {})
    mine = flux_source_sha16()
    if at.get('flux_source_sha16') != mine:
        return None, (f'profiles/pmc_traffic.json holds counters of flux sources {at.get("flux_source_sha16")} (commit '
                      f'{at.get("commit")}); this build is {mine}: not reported')
    return val, (f'profiles/pmc_traffic.json: separate rocprofv3 --pmc passes of this command at commit {at.get("commit")} '
                 f'(flux sources {mine}, unchanged since), not measured in this run')


def make_transects(nx, ny, xmin, xmax, ymin, ymax, nbatch, seed=20260401, seam=False):
    """Transect batch of SURVEY 8d C5: `nbatch` seeded polylines of 8-64 vertices snapped to grid nodes, |lat| <= 80, every
    second one closed.

    seam=False (the bench, psi = the singular arctan2 case): README.md:51's singular transect first; all vertices
    inside the lon box and clear of column 0 -- that psi is not x-periodic, and the west slot of column 0 is always the
    periodic copy of column nx-1 (field.py:223), so a polyline that cuts through column 0 would not see psi(-180, .).
    seam=True (for x-periodic psi): node columns are drawn from [-nx/4, nx + nx/4], i.e. longitudes from xmin - 90 to
    xmax + 90, so the polylines cross the +-180 seam and column 0 in both directions; three fixed polylines come first:
    one across the seam eastwards, one that runs ALONG the seam (every piece shared by column nx-1 and the periodic
    image of column 0) and one westwards given with longitudes below -180."""
    dx, dy = (xmax - xmin) / nx, (ymax - ymin) / ny
    rng = numpy.random.default_rng(seed)
    jlo, jhi = int(numpy.ceil((-80. - ymin) / dy)), int(numpy.floor((80. - ymin) / dy))
    if seam:
        def node(i, j):
            return (xmin + int(i) * dx, ymin + int(j) * dy)
        j0, q = ny // 2, max(1, ny // 9)
        e = max(1, nx // 36)
        polys = [[node(nx - e, j0 - 3 * q), node(nx + e, j0 - q), node(nx + 2 * e, j0 + 2 * q)],
                 [node(nx - 3 * e, j0 - 2 * q), node(nx, j0 - 2 * q), node(nx, j0 + 2 * q), node(nx + 3 * e, j0 + 2 * q)],
                 [node(e, j0 + q), node(-e, j0 + 3 * q), node(-2 * e, j0 - 2 * q)]]
        ilo, ihi = -(nx // 4), nx + nx // 4
    else:
        polys = [[(-180., -80.), (-10., -80.), (-10., 80.), (-180., 80.)]]
        ilo, ihi = 1, nx
    for p in range(nbatch):
        n = int(rng.integers(8, 65))
        i = rng.integers(ilo, ihi + 1, size=n)
        j = rng.integers(jlo, jhi + 1, size=n)
        pts = [(xmin + int(a) * dx, ymin + int(b) * dy) for a, b in zip(i, j)]
        if p % 2 == 1:
            pts.append(pts[0])
        polys.append(pts)
    return polys


def run_workload(args, dtype, scaling, rank, world, local, want_totals=False, emulate=None):
    """Generate the workload on the device, time `steps` passes, check the accuracy.  Returns a dict of measurements.
    emulate = (r, N): this ONE process does exactly what rank r of an N-rank strong-scaling run does on its GPU -- the same
    slab range, the same time steps generated, the same launches -- minus the all-reduce (there is nobody to reduce with)."""
    import contextlib
    import ctypes
    import io
    import torch
    import torch.distributed as dist
    from nemoflux_amd import dist as nfdist
    from nemoflux_amd._lib import DeviceArray, lib, check
    from nemoflux_amd.datagen import DataGen, STREAM_FUNCTIONS
    from nemoflux_amd.field import Field
    from nemoflux_amd.fluxexact import exactFlux

    nx, ny, nz = args.nx, args.ny, args.nz
    psi = STREAM_FUNCTIONS[5]
    nt_global = args.nt * world if scaling == 'weak' else args.nt
    srange = nfdist.slab_range(nt_global, nz, *(emulate if emulate else (rank, world)))
    t_begin, t_end = nfdist.time_steps_touched(srange, nz)
    if t_end <= t_begin:        # more ranks than slabs: this rank owns nothing, but its pointers must still be valid
        t_begin, t_end = 0, 1
    real = 'float64' if dtype == 'f64' else 'float32'

    # ---- synthetic input, generated on the device (datagen.py counterpart): only the time steps this rank touches
    dg = DataGen(real=real)
    dg.setSizes(nx, ny, nz, nt_global)
    dg.setBoundingBox(-180., 180., -90., 90., 0., 1.)
    dg.build()
    dg.applyStreamFunction(psi)
    u, v = dg.computeUVFromPotential(t_begin, t_end)
    slab_elems = ny * nx
    ug = DeviceArray(nfdist.virtual_base(u, t_begin * nz, slab_elems), (nt_global, nz, ny, nx), real, u)
    vg = DeviceArray(nfdist.virtual_base(v, t_begin * nz, slab_elems), (nt_global, nz, ny, nx), real, v)
    polys = make_transects(nx, ny, -180., 180., -90., 90., args.batch)
    xyzs = [numpy.array([(x, y, 0.) for x, y in p]) for p in polys]
    stream = torch.cuda.current_stream().cuda_stream
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        fld = Field.fromArrays(dg.bounds_lon, dg.bounds_lat, dg.deptht_bounds, ug, vg, xyzs, slab_range=srange,
                               readback=False, stream=stream, compact=args.compact)
    setup_s = time.time() - t0
    # the weight build (K2) of this transect batch on its own: a second build of the same weights, timed (the first one above
    # also paid the first-use costs of the process)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    check(lib.nf_field_build_weights(ctypes.byref(fld._h), 128, ctypes.c_double(360.)))
    torch.cuda.synchronize()
    weights_build_ms = (time.perf_counter() - t0) * 1e3
    rows = torch.zeros((nt_global, fld._rowlen), dtype=torch.float64, device='cuda')

    def step():
        check(lib.nf_field_compute_all_async(ctypes.byref(fld._h), ctypes.c_void_p(rows.data_ptr())))
        nfdist.reduce_rows(rows)

    def barrier():
        if world > 1:
            if dist.get_backend() == 'nccl':
                dist.barrier(device_ids=[local])     # this rank's GPU, not a guess from the global rank
            else:
                dist.barrier()

    if world > 1:   # communicator set-up (RCCL ring build) never lands in the timed region, even with --warmup 0
        nfdist.reduce_rows(rows)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    # events for every launch of the timed region are created here, outside it
    fld.enableKernelTiming(True, reserve=args.steps * max(1, t_end - t_begin))
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    nlaunch, kernel_ms, flux_ms, expand_ms = fld.readKernelTiming(split=True)
    k3_ms = fld.readTransectTiming()
    fld.enableKernelTiming(False)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        nfdist.all_reduce(tt, dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    units_total = float(nt_global) * nz * ny * nx
    if emulate:     # the rank's own share: what it integrates per pass
        units_total = float(srange[1] - srange[0]) * ny * nx
    m = {'value': units_total * args.steps / elapsed, 'ms_per_step': elapsed / args.steps * 1e3,
         'nt_global': nt_global, 'setup_s': setup_s, 'weights_build_ms': weights_build_ms, 'psi': psi, 'polys': polys, 'nseg': fld._nseg,
         'weight_entries': int(fld.getWeights()[0].size), 'dg': dg, 'u': u, 'v': v, 'xyz0': xyzs[0]}

    # ---- the one collective of the N>1 path, timed on its own after the timed region (message = the rows)
    if world > 1:
        reps = 20
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            nfdist.reduce_rows(rows)
        torch.cuda.synchronize()
        m['reduce'] = {'message_bytes': int(rows.numel() * rows.element_size()), 'op': 'all_reduce(SUM, float64)',
                       'backend': dist.get_backend(), 'world_size': dist.get_world_size(),
                       'allreduce_ms': round((time.perf_counter() - t0) / reps * 1e3, 4), 'calls_per_step': 1}
        # what the communicator itself says (the engine's own RCCL communicator when the backend is RCCL), and one record
        # per rank: which device it ran on, which slabs it owned, what its kernels took
        comm = nfdist.native_comm()
        seen = comm.info() if comm is not None else None
        m['reduce']['path'] = ('nf_rows_allreduce (ncclAllReduce on the library\'s own communicator)' if comm is not None
                               else 'torch.distributed.all_reduce')
        m['reduce']['fell_back'] = comm is None and os.environ.get('NF_NATIVE_REDUCE', '1') != '0' and dist.get_backend() == 'nccl'
        if m['reduce']['fell_back']:
            m['reduce']['fell_back_reason'] = nfdist.fell_back_reason.get(None, 'unknown')
        if seen is not None:
            m['reduce']['world_size_rccl'] = seen['world_size']
            m['reduce']['library'] = seen['library']
            if seen['world_size'] != world:     # a communicator over fewer ranks would sum fewer partial rows under the same name
                print(f'bench.py: rank {rank}: the RCCL communicator spans {seen["world_size"]} ranks, the job has {world}; '
                      'refusing to report a value', file=sys.stderr, flush=True)
                raise SystemExit(4)
        prop = torch.cuda.get_device_properties(local)
        mine = {'rank': rank, 'device_index': local, 'device_name': prop.name,
                'device_uuid': str(getattr(prop, 'uuid', '')), 'pci_bus_id': int(getattr(prop, 'pci_bus_id', -1)),
                'rccl_rank': None if seen is None else seen['rank'],
                'rccl_device_index': None if seen is None else seen['device_index'],
                'slabs': [int(srange[0]), int(srange[1])], 'steps_touched': [int(t_begin), int(t_end)],
                'launches_per_pass': nlaunch // max(1, args.steps),
                'k_flux_ms': round(flux_ms / max(1, args.steps), 4),
                'k_expand_ms': round(expand_ms / max(1, args.steps), 4),
                'k3_ms': round(k3_ms / max(1, args.steps), 4)}
        recs = [None] * world
        dist.all_gather_object(recs, mine)
        m['ranks'] = recs
        step()                       # the rows the accuracy block reads: one clean pass after the repeated reduces

    # ---- accuracy: every transect total of every time step vs the closed form (fluxexact.py:36-46)
    res = rows.cpu().numpy()
    nseg = fld._nseg
    max_err, max_ref = 0.0, 0.0
    for p, pts in enumerate(polys if rank == 0 and not emulate else []):     # rank 0 prints the line: only it needs the check
        ex = exactFlux(psi, pts, nz, nt_global)
        got = res[:, nseg + p]
        max_err = max(max_err, float(numpy.abs(got - numpy.array(ex)).max()))
        max_ref = max(max_ref, float(numpy.abs(ex).max()))
    if emulate:     # partial rows of one rank: nothing to compare with the closed form
        max_err, max_ref = None, None
        m['emulated'] = {'rank': emulate[0], 'of': emulate[1], 'slabs': [int(srange[0]), int(srange[1])],
                         'steps_touched': [int(t_begin), int(t_end)], 'launches_per_pass': nlaunch // max(1, args.steps),
                         'ms_per_pass': round(elapsed / args.steps * 1e3, 4),
                         'k_flux_ms': round(flux_ms / max(1, args.steps), 4),
                         'k_expand_ms': round(expand_ms / max(1, args.steps), 4),
                         'k3_ms': round(k3_ms / max(1, args.steps), 4),
                         'partial_step_planes': 'full (knob)' if 'partial_step_planes=1' in args.knob else 'signed-only (default)'}
    m['accuracy'] = {'max_abs_err_vs_fluxexact': max_err, 'max_abs_exact': max_ref,
                     'singular_transect_t0': float(res[0, nseg + 0]), 'transect_steps_checked': len(polys) * nt_global}
    if want_totals:
        m['totals'] = res[:, nseg:nseg + len(polys)].tolist()

    # ---- roofline of the dominant kernel (vertical integral + edge flux), HIP events on its stream
    s = 8 if dtype == 'f64' else 4
    bytes_per_unit = 2 * s + 64.0 / nz                      # u,v reads + (arc 16 + iV 32 + abs 16)/nz
    if args.compact:
        bytes_per_unit = 2 * s + 32.0 / nz                  # compact mode: arc 16 + (eU, eV) 16 per column
    own = srange[1] - srange[0]
    units_per_launch = own * ny * nx / max(1.0, (nlaunch / args.steps))   # owned slabs / launches per step
    avg_ms = kernel_ms / max(1, nlaunch)
    achieved = bytes_per_unit * units_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic, traffic_source = None, None
    wl_key = f'{nx}x{ny}x{nz}x{args.nt}_{dtype}' + ('_compact' if args.compact else '')
    if world == 1:   # the PMC pass measured whole time steps (75 levels per launch): N=1 only
        traffic, traffic_source = pmc_traffic(wl_key)
    m['roofline'] = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                     'traffic_source': traffic_source,
                     'kernel': 'nf::k_flux + nf::k_expand_planes (one event pair around both)' if expand_ms > 0 else 'nf::k_flux',
                     'avg_launch_ms': round(avg_ms, 4), 'launches': nlaunch,
                     'avg_ms_by_kernel': {'nf::k_flux': round(flux_ms / max(1, nlaunch), 4),
                                          'nf::k_expand_planes': round(expand_ms / max(1, nlaunch), 4),
                                          'transect reduction (nf::k_gather_segscan + 2 finalize kernels)':
                                              round(k3_ms / max(1, nlaunch), 4)},
                     'algorithmic_bytes_per_unit': round(bytes_per_unit, 3),
                     'units_per_launch': units_per_launch,
                     'wall_frac': round(bytes_per_unit * units_total * (own / float(nt_global * nz)) /
                                        (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4)}
    if emulate:
        m['roofline']['wall_frac'] = round(bytes_per_unit * units_total / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4)
    del fld
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--nx', type=int, default=3600)
    ap.add_argument('--ny', type=int, default=1800)
    ap.add_argument('--nz', type=int, default=75)
    ap.add_argument('--nt', type=int, default=12, help='time steps in total (strong) or per rank (weak)')
    ap.add_argument('--dtype', default='f64', choices=['f64', 'f32'])
    ap.add_argument('--scaling', default='strong', choices=['weak', 'strong'],
                    help='strong (default): the nt*nz slabs of the stated problem are cut N ways; weak: nt steps per rank')
    ap.add_argument('--batch', type=int, default=64, help='number of extra seeded transects')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--no-f32', action='store_true', help='skip the float32 sub-record (N=1, --dtype f64 only)')
    ap.add_argument('--no-ingest', action='store_true', help='skip the file-ingest sub-record (N=1)')
    ap.add_argument('--no-c3', action='store_true', help='skip the ORCA025-like C3 sub-record (N=1)')
    ap.add_argument('--only-c3', action='store_true', help='print the C3 sub-record alone (for profiling it: scripts/gpu_profile.sh)')
    ap.add_argument('--dump-totals', action='store_true', help='add the (nt, ntransect) totals to the JSON (small grids)')
    ap.add_argument('--emulate-rank', default=None, metavar='r/N',
                    help='NOT a scaling run: on ONE GPU, do exactly what rank r of an N-rank strong-scaling run does (its slab '
                         'range, its launches; no reduce) and report its ms per pass -- per-rank compute evidence for N > 1')
    ap.add_argument('--knob', action='append', default=[], metavar='name=value',
                    help='NOT the headline configuration: nf_tuning_set(name, value) before anything runs (A/B and counter passes)')
    ap.add_argument('--compact', action='store_true',
                    help='NOT the headline configuration: keep only (eU, eV) resident per step (nf_field_set_compact); the '
                         '(ncell,4) copies and |.| arrays are derived at read-back, which a batch driver never asks for')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  It has made no GPU call and never will
        # (it does not import torch; devices are counted from sysfs); the N ranks are fresh child processes.
        raise SystemExit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    from nemoflux_amd import dist as nfdist

    rank, world, local = nfdist.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree', file=sys.stderr)
        raise SystemExit(2)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: nemoflux_amd has no CPU fallback')
    torch.cuda.set_device(local)
    for kv in args.knob:
        from nemoflux_amd._lib import lib, check
        name, val = kv.split('=')
        check(lib.nf_tuning_set(name.encode(), int(val)))

    if args.only_c3:
        if world != 1:
            raise SystemExit('bench.py: --only-c3 needs --gpus 1')
        print(json.dumps({'c3': c3_record(args.steps)}))
        return
    nx, ny, nz = args.nx, args.ny, args.nz
    emulate = None
    if args.emulate_rank:
        r, n = (int(x) for x in args.emulate_rank.split('/'))
        if world != 1 or args.scaling != 'strong' or not (0 <= r < n):
            raise SystemExit('bench.py: --emulate-rank r/N needs --gpus 1, strong scaling and 0 <= r < N')
        emulate = (r, n)
        args.no_cpu = args.no_f32 = args.no_ingest = args.no_c3 = True
    m = run_workload(args, args.dtype, args.scaling, rank, world, local, want_totals=args.dump_totals, emulate=emulate)
    slabs = m['nt_global'] * nz
    out = {
        'metric': 'edge-flux integrals/sec', 'value': m['value'], 'unit': 'integrals/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': m['ms_per_step'],
        'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': args.dtype,
        'data': 'synthetic',
        'config': {'workload': f'C4 ORCA12-like {nx}x{ny}x{nz}x{args.nt}' +
                               (f' per GPU (weak: {m["nt_global"]} time steps in total)' if args.scaling == 'weak' else
                                f' in total (strong: {slabs} (t,z) slabs cut {world} way' + ('s' if world > 1 else '') + ')') +
                               f', psi={m["psi"]}, README singular transect + {args.batch} node-snapped transects',
                   'nx': nx, 'ny': ny, 'nz': nz, 'nt_global': m['nt_global'], 'transects': len(m['polys']),
                   'target_segments': m['nseg'], 'weight_entries': m['weight_entries'],
                   'parallelism': f'(t,z)-slab sharding x{world}, 1 all-reduce',
                   'resident_outputs': 'eU,eV only (compact, non-headline)' if args.compact else
                                       'integratedVelocity [4][ncell] + |eU|,|eV| every step',
                   'setup_s': round(m['setup_s'], 3), 'weights_build_ms': round(m['weights_build_ms'], 3)},
        'roofline': m['roofline'],
        'accuracy': m['accuracy'],
    }
    if emulate:
        out['emulated_rank'] = m['emulated']
        out['config']['workload'] += (f' -- EMULATION of rank {emulate[0]} of {emulate[1]} on one GPU: value = this rank\'s own '
                                      'slabs per second, no reduce; not a scaling measurement')
    if 'reduce' in m:
        out['reduce'] = m['reduce']
        out['ranks'] = m['ranks']
    if 'totals' in m:
        out['totals'] = m['totals']

    # ---- CPU baseline (rank 0, N=1 only): the reference's numpy statements on one time step
    if rank == 0 and world == 1 and not args.no_cpu:
        out['cpu_baseline'] = cpu_baseline(m['dg'], m['u'], m['v'], nz, ny, nx, m['polys'], args)
        # the CPU leg integrates time step 0 across the README transect with the oracle's own weights: the same number as the
        # GPU's row, from an independent implementation end to end (geometry, vertical integral, weights, reduction)
        out['accuracy']['cpu_flux_t0'] = out['cpu_baseline']['legs']['flux_t0']
        out['accuracy']['cpu_gpu_flux_abs_diff'] = abs(out['cpu_baseline']['legs']['flux_t0'] - m['accuracy']['singular_transect_t0'])
        if not out['accuracy']['cpu_gpu_flux_abs_diff'] <= 1e-11 * max(1.0, abs(m['accuracy']['singular_transect_t0'])):
            print(f'bench.py: CPU and GPU disagree on the flux of step 0: {out["accuracy"]["cpu_flux_t0"]!r} vs '
                  f'{m["accuracy"]["singular_transect_t0"]!r}', file=sys.stderr, flush=True)
            raise SystemExit(5)

    # ---- float32 inputs (real NEMO files are float32: SURVEY 2 row 13): same workload, same checks, short record
    if world == 1 and args.dtype == 'f64' and not args.no_f32:
        import gc
        for k in ('dg', 'u', 'v'):
            m.pop(k, None)
        gc.collect()
        torch.cuda.empty_cache()
        m32 = run_workload(args, 'f32', args.scaling, rank, world, local)
        r = m32['roofline']
        out['f32'] = {'value': m32['value'], 'unit': 'integrals/s', 'ms_per_step': m32['ms_per_step'], 'dtype': 'f32',
                      'roofline': {k: r[k] for k in ('achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source',
                                                    'kernel', 'avg_launch_ms', 'launches', 'avg_ms_by_kernel',
                                                    'algorithmic_bytes_per_unit', 'wall_frac')},
                      'accuracy': m32['accuracy']}
    # ---- file ingest (N=1): what a file-backed pass adds in front of the kernels above -- one launch of the device DEFLATE
    # decoder on 256 copies of a NEMO-like level (field.py:149's lazy NetCDF read: HDF5 shuffle + deflate)
    if world == 1 and not args.no_ingest:
        try:
            out['ingest'] = ingest_record()
        except Exception as e:      # never costs the headline line
            out['ingest'] = {'error': f'{type(e).__name__}: {e}'}
    # ---- BASELINE config C3 (ORCA025-like, one time step) beside the headline: the mid-size grid most real files have
    if world == 1 and not args.no_c3 and not emulate:
        try:
            out['c3'] = c3_record()
        except Exception as e:
            out['c3'] = {'error': f'{type(e).__name__}: {e}'}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        nfdist.destroy_native_comms()
        dist.destroy_process_group()


def c3_record(steps=20):
    """BASELINE config C3 beside the headline (N=1): the ORCA025-like 1440 x 1021 x 75 grid, ONE time step, the 50-station
    transect data/S3_sta_bdep.txt (tests/golden/stations.json holds the reference's own parse of it), float64 and float32.
    A step of this grid is only 2.2 / 1.1 rounds of resident wavefronts, so K1 runs in its one-field form
    (nf::k_flux_field); same events, same algorithmic bytes per unit as the headline."""
    import contextlib
    import ctypes
    import io
    import torch
    from nemoflux_amd._lib import lib, check
    from nemoflux_amd.datagen import DataGen, STREAM_FUNCTIONS
    from nemoflux_amd.field import Field
    nx, ny, nz = 1440, 1021, 75
    with open(os.path.join(ROOT, 'tests', 'golden', 'stations.json')) as f:
        st = json.load(f)['S3_sta_bdep.txt']
    xyz = numpy.array([(lon, lat, 0.) for lon, lat in st])
    out = {'workload': f'C3 ORCA025-like {nx}x{ny}x{nz}x1, psi={STREAM_FUNCTIONS[2]}, transect S3_sta_bdep.txt ({len(st)} stations), '
                       f'{steps} timed steps'}
    for dtype, real, es in (('f64', 'float64', 8), ('f32', 'float32', 4)):
        dg = DataGen(real=real)
        dg.setSizes(nx, ny, nz, 1)
        dg.setBoundingBox(-180., 180., -90., 90., 0., 1.)
        dg.build()
        dg.applyStreamFunction(STREAM_FUNCTIONS[2])
        u, v = dg.computeUVFromPotential()
        with contextlib.redirect_stdout(io.StringIO()):
            fld = Field.fromArrays(dg.bounds_lon, dg.bounds_lat, dg.deptht_bounds, u, v, [xyz], readback=False,
                                   stream=torch.cuda.current_stream().cuda_stream)
        rows = torch.zeros((1, fld._rowlen), dtype=torch.float64, device='cuda')
        for _ in range(3):
            check(lib.nf_field_compute_all_async(ctypes.byref(fld._h), ctypes.c_void_p(rows.data_ptr())))
        fld.enableKernelTiming(True, reserve=steps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            check(lib.nf_field_compute_all_async(ctypes.byref(fld._h), ctypes.c_void_p(rows.data_ptr())))
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps
        nl, kms, flux_ms, expand_ms = fld.readKernelTiming(split=True)
        k3 = fld.readTransectTiming()
        fld.enableKernelTiming(False)
        units = float(nz) * ny * nx
        bpu = 2 * es + 64.0 / nz
        k1 = kms / max(1, nl)
        traffic, traffic_source = pmc_traffic(f'{nx}x{ny}x{nz}x1_{dtype}')     # two separate --pmc passes of `bench.py --only-c3`
        out[dtype] = {'value': units / wall, 'unit': 'integrals/s', 'ms_per_step': round(wall * 1e3, 4),
                      'k1_ms': round(k1, 4), 'k3_ms': round(k3 / max(1, nl), 4),
                      'frac': round(bpu * units / (k1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      'wall_frac': round(bpu * units / wall / 1e9 / HBM_PEAK_GBS, 4),
                      'algorithmic_bytes_per_unit': round(bpu, 3), 'traffic': traffic, 'traffic_source': traffic_source,
                      'kernel': 'nf::k_flux_field' if expand_ms == 0 else 'nf::k_flux + nf::k_expand_planes',
                      'flux': float(rows[0, -1].item())}
        del fld, dg, u, v
        torch.cuda.empty_cache()
    return out


def ingest_record(streams=256):
    """One launch of the device decoder (nf_inflate.hip) on `streams` copies of one 1440 x 1021 float32 level, byte-shuffled
    and deflated at zlib level 4 the way XIOS / netCDF-4 store NEMO output: all streams are resident at once (4 per CU), so
    the launch takes the time of ONE stream.  The compressed bytes are uploaded first; the timed part is inflate +
    un-shuffle + placement (torch events on the launch stream); the result is compared with the source bit for bit."""
    import zlib
    import torch
    from nemoflux_amd._lib import DeviceBuffer
    from nemoflux_amd.ingest import ChunkDecoder, StagedChunks
    ny, nx = 1021, 1440
    rng = numpy.random.default_rng(1)
    y = numpy.linspace(-90, 90, ny)[:, None]
    x = numpy.linspace(-180, 180, nx)[None, :]
    f = ((numpy.cos(2 * numpy.pi * y / 360) + numpy.sin(2 * numpy.pi * x / 360)) * 3.1).astype('<f4')
    f *= (1 + numpy.float32(1e-3) * rng.standard_normal(f.shape, dtype=numpy.float32))
    comp = zlib.compress(numpy.ascontiguousarray(f.view(numpy.uint8).reshape(-1, 4).T).tobytes(), 4)
    dec = ChunkDecoder()
    pad = (len(comp) + 7) & ~7
    pinned = dec.new_pinned(streams * pad + 64)
    in_off = numpy.arange(streams, dtype=numpy.int64) * pad
    for i in range(streams):
        pinned.array[in_off[i]:in_off[i] + len(comp)] = numpy.frombuffer(comp, numpy.uint8)
    origin = numpy.zeros((streams, 3), numpy.int64)
    origin[:, 0] = numpy.arange(streams)
    plan = dict(chunk_dims=(1, ny, nx), slab_dims=(streams, ny, nx), chunk_bytes=f.nbytes, elem_size=4, shuffled=1)
    staged = StagedChunks(pinned, streams * pad, in_off, numpy.full(streams, len(comp), numpy.int64), origin, plan)
    slab = DeviceBuffer(streams * f.nbytes)
    best = 1e30
    for rep in range(3):
        dec.upload(pinned, staged.used)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dec.decode(staged, slab.ptr, uploaded=True)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    got = slab.download((streams, ny, nx), '<f4')
    ok = bool(numpy.array_equal(got[0].view(numpy.uint32), f.view(numpy.uint32)) and
              numpy.array_equal(got[streams - 1].view(numpy.uint32), f.view(numpy.uint32)))
    slab.free()
    del staged, pinned
    # ---- which of the two limits a file-backed step (round-5 verdict W5): one staged GROUP of the C3 float32 pass the way
    # nemoflux_amd.staging runs it -- 6 time steps x (75 levels of uo + 75 of vo) = 900 chunks -- (a) the upload of its
    # compressed bytes from pinned host memory (the host link) and (b) its device half (inflate + un-shuffle + placement),
    # each timed on its own; in the pipelined pass (a) of the next group runs under (b) of this one, so a step costs the
    # larger of the two
    gsteps, glev = 6, 150
    n = gsteps * glev
    gp = dec.new_pinned(n * pad + 64)
    g_off = numpy.arange(n, dtype=numpy.int64) * pad
    row = numpy.zeros(pad, numpy.uint8)
    row[:len(comp)] = numpy.frombuffer(comp, numpy.uint8)
    gp.array[:n * pad].reshape(n, pad)[:] = row
    g_origin = numpy.zeros((n, 3), numpy.int64)
    g_origin[:, 0] = numpy.arange(n)
    g_plan = dict(chunk_dims=(1, ny, nx), slab_dims=(n, ny, nx), chunk_bytes=f.nbytes, elem_size=4, shuffled=1)
    g_staged = StagedChunks(gp, n * pad, g_off, numpy.full(n, len(comp), numpy.int64), g_origin, g_plan)
    g_slab = DeviceBuffer(n * f.nbytes)
    up_best, dev_best = 1e30, 1e30
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dec.upload(gp, g_staged.used)                      # synchronous: complete at return
        up_best = min(up_best, (time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter()
        dec.decode(g_staged, g_slab.ptr, uploaded=True)    # synchronous
        dev_best = min(dev_best, (time.perf_counter() - t0) * 1e3)
    g_ok = bool(numpy.array_equal(g_slab.download((n, ny, nx), '<f4')[n - 1].view(numpy.uint32), f.view(numpy.uint32)))
    g_slab.free()
    comp_step = glev * len(comp)
    up_step, dev_step = up_best / gsteps, dev_best / gsteps
    group = {'what': f'one staged group of a C3 float32 file-backed pass: {gsteps} time steps x {glev} levels (uo + vo) = {n} chunks',
             'compressed_bytes_per_step': comp_step, 'decoded_bytes_per_step': glev * f.nbytes,
             'upload_ms_per_step': round(up_step, 3), 'upload_GB_per_s': round(comp_step / up_step / 1e6, 2),
             'device_half_ms_per_step': round(dev_step, 3),
             'device_half_decoded_GB_per_s': round(glev * f.nbytes / dev_step / 1e6, 2),
             'bound': 'decoder' if dev_step >= up_step else 'host link',
             'device_half_over_upload': round(dev_step / up_step, 2), 'bit_identical': g_ok,
             'note': 'the pipelined pass uploads the next group under the decode of this one: a file-backed step costs about the '
                     'larger of the two.  Here the 900 chunks are copies of ONE level and finish together; the distinct levels of a '
                     'real file do not (16.2 ms per step for the device half, 20.2 ms per step for the whole pipelined pass of 24 '
                     'steps incl. its first group: tools/filebacked_timing.py, profiles/r06_filebacked_timing.txt)'}
    return {'bound': group['bound'], 'group': group,
            'workload': f'{streams} copies of one {nx}x{ny} float32 level, HDF5 shuffle + zlib level 4 ({f.nbytes} bytes from '
                        f'{len(comp)}): one wavefront per stream, all resident at once',
            'inflate_unshuffle_place_ms': round(best, 3), 'MB_per_s_per_stream': round(f.nbytes / best / 1e3, 2),
            'decoded_GB_per_s_per_launch': round(streams * f.nbytes / best / 1e6, 2), 'bit_identical': ok,
            'kernels': 'nf::k_inflate + nf::k_place16', 'resident_streams_capacity': ChunkDecoder.capacity()}


def visible_gpu_count(kfd='/sys/class/kfd'):
    """GPUs a child process would see, counted WITHOUT touching the HIP runtime (the launcher must stay a process that has
    never initialised the GPU): the KFD topology in sysfs (GPU nodes have simd_count > 0), narrowed by HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when set.  None when sysfs cannot be read (the ranks then fail loudly
    themselves if a device is missing)."""
    n = None
    try:
        base = os.path.join(kfd, 'kfd', 'topology', 'nodes')
        if not os.path.exists(kfd):
            raise FileNotFoundError
        n = 0
        for node in os.listdir(base):
            try:
                with open(os.path.join(base, node, 'properties')) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            except PermissionError:
                continue        # a GPU of the host that this container may not use (device cgroup): not ours
            if int(props.get('simd_count', '0')) > 0:
                n += 1
    except FileNotFoundError:
        n = 0           # no KFD node at all: no compute driver, no GPU
    except (OSError, ValueError):
        n = None
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            k = len([x for x in v.split(',') if x.strip() != ''])
            n = k if n is None else min(n, k)
    return n


def self_launch(ngpus):
    """`python bench.py --gpus N` without a launcher: start the N ranks the way the driver's own N>1 command does
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>), one process per GPU, as a
    child of this process -- which has not touched the GPU and has not even imported torch -- relay rank 0's JSON line and
    return the child's exit code.
    Fewer than N visible devices is an error, not a silent N=1 run (NF_FORCE_DEVICE, the rehearsal hook of
    nemoflux_amd.dist that puts every rank on one GPU, lifts that check)."""
    import socket
    import subprocess
    ndev = visible_gpu_count()              # from sysfs / the *_VISIBLE_DEVICES variables: no HIP or torch call in this process
    if ndev is not None and ndev < ngpus and 'NF_FORCE_DEVICE' not in os.environ:
        print(f'bench.py: --gpus {ngpus} but {ndev} GPU(s) visible on this node; refusing to run a smaller job under that '
              'name', file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={ngpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # No environment of its own: what a rank needs (dmabuf IPC for RCCL, the start-up time limit, its host threads) is set by
    # the rank itself in nemoflux_amd.dist.rank_environment(), so that `torchrun ... bench.py --gpus N` runs identical ranks
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, close_fds=True)
    for line in child.stdout:               # rank 0's JSON line (and anything else the ranks print) goes straight through
        sys.stdout.write(line)
        sys.stdout.flush()
    return child.wait()


def cpu_baseline(dg, u, v, nz, ny, nx, polys, args):
    """Reference CPU path timed on this box's host cores, on ONE time step of the bench workload already in RAM (NetCDF
    I/O excluded, like the GPU side).  Legs (SURVEY 8d "CPU baseline beside it"):
      (i)  numpy restatement of field.py:157,161 (missing->0, tensordot over z) + field.py:183-234 (edge fluxes) + the
           oracle's A7 for the README transect: default BLAS threads (= `value`, `cores`) and again with 1 thread;
           the plain-C port of the same step with OpenMP;
      (ii) the C restatement of the mint weights + getIntegral (A6+A7): one thread (like mint) on the README transect,
           and on all host cores over a sample of the batch polylines (one polyline per thread)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import nf_oracle as o
    o.build()
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        threads = max([p.get('num_threads', 1) for p in threadpool_info()] + [1])
    except Exception:
        threadpool_limits = None
        threads = os.cpu_count()
    try:
        ncores = len(os.sched_getaffinity(0))
    except Exception:
        ncores = os.cpu_count()
    uh, vh = u[0].cpu().numpy(), v[0].cpu().numpy()
    blon, blat = dg.bounds_lon.cpu().numpy(), dg.bounds_lat.cpu().numpy()
    pts = o.assemble_points(blon, blat)
    xyz0 = numpy.array([(x, y, 0.) for x, y in polys[0]])
    t0 = time.perf_counter()
    arc = o.np_arc_lengths(pts)
    t_arc = time.perf_counter() - t0
    t0 = time.perf_counter()
    w = o.polyline_weights(pts, xyz0)
    t_w = time.perf_counter() - t0
    th = dg.zbot - dg.ztop
    st = o.EdgeFluxState(ny, nx)

    def numpy_step():
        U = o.np_read_field(uh, th)
        V = o.np_read_field(vh, th)
        o.np_edge_flux(st, U, V, arc)
        return o.get_integral(w, st.integratedVelocity)

    best = 1e30
    reps = 0
    t_all = time.perf_counter()
    while reps < 3 or (time.perf_counter() - t_all < 10.0 and reps < 8):
        t0 = time.perf_counter()
        tot = numpy_step()
        best = min(best, time.perf_counter() - t0)
        reps += 1
    units = float(nz) * ny * nx
    t_1 = None
    if threadpool_limits is not None:        # the same statements with ONE BLAS thread, one repetition
        with threadpool_limits(limits=1):
            t0 = time.perf_counter()
            numpy_step()
            t_1 = time.perf_counter() - t0
    # the plain-C port (OpenMP over columns; OMP_NUM_THREADS or all host cores), same step
    t_c = 1e30
    for _ in range(2):
        t0 = time.perf_counter()
        Uc = o.vertical_integral(uh, th)
        Vc = o.vertical_integral(vh, th)
        o.edge_flux(o.EdgeFluxState(ny, nx), Uc, Vc, arc)
        t_c = min(t_c, time.perf_counter() - t0)
    omp = os.environ.get('OMP_NUM_THREADS', str(os.cpu_count()))
    # A6+A7 on all host cores: one polyline of the batch per thread (the C restatement releases the GIL under ctypes)
    from concurrent.futures import ThreadPoolExecutor
    sample = polys[1:1 + min(len(polys) - 1, max(4, min(16, ncores)))]
    nseg_sample = sum(len(p) - 1 for p in sample)
    t_a6 = None
    if sample:
        def one(p):
            ww = o.polyline_weights(pts, numpy.array([(x, y, 0.) for x, y in p]))
            return o.get_integral(ww, st.integratedVelocity)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=ncores) as ex:
            list(ex.map(one, sample))
        t_a6 = time.perf_counter() - t0
    legs = {'flux_t0': float(tot), 'numpy_default_threads_s_per_step': round(best, 4), 'numpy_threads': int(threads),
            'numpy_1_thread_s_per_step': None if t_1 is None else round(t_1, 4),
            'numpy_1_thread_integrals_per_s': None if t_1 is None else units / t_1,
            'c_openmp_s_per_step': round(t_c, 4), 'c_openmp_threads': omp, 'c_openmp_integrals_per_s': units / t_c,
            'numpy_arc_lengths_s': round(t_arc, 3),
            'a6_a7_1_thread_s': round(t_w, 3), 'a6_a7_1_thread_segments': len(polys[0]) - 1,
            'a6_a7_all_cores_s': None if t_a6 is None else round(t_a6, 3), 'a6_a7_all_cores_threads': ncores,
            'a6_a7_all_cores_segments': nseg_sample,
            'a6_a7_all_cores_segments_per_s': None if not t_a6 else nseg_sample / t_a6}
    one_thread = 'n/a' if t_1 is None else f'{units / t_1:.3e} integrals/s ({t_1:.2f} s/step)'
    a6 = 'n/a' if t_a6 is None else (f'{len(sample)} batch polylines ({nseg_sample} segments) on {ncores} threads, one '
                                     f'polyline per thread: {t_a6:.2f} s')
    return {'value': units / best, 'unit': 'integrals/s', 'cores': int(threads), 'kind': 'port',
            'sample': f'1 of {args.nt} time steps of the bench workload ({nx}x{ny}x{nz}, {args.dtype}), best of {reps} '
                      f'reps of the numpy restatement of field.py:157-234 (+oracle A7, README transect): '
                      f'{best:.3f} s/step; the same with 1 BLAS thread, 1 rep: {one_thread}; one-off: numpy arc lengths '
                      f'{t_arc:.2f} s; C restatement of the mint weights + getIntegral (A6+A7): one '
                      f'thread like mint, README transect over all {ny * nx} cells {t_w:.2f} s; {a6}; '
                      f'C port of the same step with OpenMP ({omp} threads): {units / t_c:.3e} integrals/s; flux {tot:.6g}',
            'legs': legs}


if __name__ == '__main__':
    main()
