"""CPU suite, part 3: the N>1 path.  world_size-2 gloo processes each integrate their (t,z) slab range (with
the oracle standing in for the GPU engine, which needs a card) and ONE all_reduce of the rows must reproduce
the single-process result -- the property nemoflux_amd.dist relies on."""
import os
import sys

import numpy
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden, transect_xyz


def test_slab_ranges_partition_everything():
    from nemoflux_amd.dist import slab_range, time_steps_touched
    for nt, nz, world in [(12, 75, 8), (12, 75, 7), (1, 1, 2), (20, 10, 3), (5, 3, 16)]:
        cuts = [slab_range(nt, nz, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == nt * nz
        assert all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
        sizes = [e - b for b, e in cuts]
        assert max(sizes) - min(sizes) <= 1
        for b, e in cuts:
            tb, te = time_steps_touched((b, e), nz)
            assert (e <= b and tb == te == 0) or (tb * nz <= b and e <= te * nz)
    assert slab_range(12, 75, 3, 8) == (337, 450)


def _partial_rows(oracle, g, m, srange, weights):
    """Rows [nt][ntransect] from the slabs in srange only (zeros elsewhere), the way the engine computes them."""
    nt, nz, ny, nx = m['nt'], m['nz'], m['ny'], m['nx']
    rows = numpy.zeros((nt, len(weights)))
    for t in range(nt):
        lo, hi = max(t * nz, srange[0]), min((t + 1) * nz, srange[1])
        if hi <= lo:
            continue
        z0, z1 = lo - t * nz, hi - t * nz
        U = oracle.vertical_integral(g['u'][t][z0:z1], g['thickness'][z0:z1], m['fill_value'])
        V = oracle.vertical_integral(g['v'][t][z0:z1], g['thickness'][z0:z1], m['fill_value'])
        st = oracle.EdgeFluxState(ny, nx)
        oracle.edge_flux(st, U, V, g['arcLengths'], m['sverdrup'])
        rows[t] = [oracle.get_integral(w, st.integratedVelocity) for w in weights]
    return rows


def _worker(rank, world, port, name, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import json
    import nf_oracle as oracle
    from nemoflux_amd import dist as nfdist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = nfdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    m = [c for c in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'cases.json'))) if c['name'] == name][0]
    g = load_golden(name)
    pts = oracle.assemble_points(g['bounds_lon'], g['bounds_lat'])
    weights = [oracle.polyline_weights(pts, transect_xyz(t['points'])) for t in m['transects'].values()]
    srange = nfdist.slab_range(m['nt'], m['nz'], rank, world)
    rows = torch.from_numpy(_partial_rows(oracle, g, m, srange, weights))
    nfdist.reduce_rows(rows)
    if rank == 0:
        numpy.save(out_path, rows.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize('name', ['rot36_zt', 'def36_zt'])
def test_two_rank_gloo_reduce_matches_single(name, oracle, cases, tmp_path):
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'rows.npy')
    mp.spawn(_worker, args=(2, port, name, out), nprocs=2, join=True)
    got = numpy.load(out)
    m = [c for c in cases if c['name'] == name][0]
    g = load_golden(name)
    pts = oracle.assemble_points(g['bounds_lon'], g['bounds_lat'])
    weights = [oracle.polyline_weights(pts, transect_xyz(t['points'])) for t in m['transects'].values()]
    full = _partial_rows(oracle, g, m, (0, m['nt'] * m['nz']), weights)
    # the reduction order differs from the single-process sum: 1e-13 relative, not bitwise
    assert numpy.allclose(got, full, rtol=1e-13, atol=1e-13 * numpy.abs(full).max())
    # nz = 3, nt = 2 cut in two: rank 0 owns (t0: z0..2), rank 1 owns (t1: z0..2) for rot36 -> also try uneven
    assert got.shape == (m['nt'], len(m['transects']))


def test_bench_transects_are_seeded_and_clear_of_column_zero():
    """bench.py's batch (config C5): deterministic, node-snapped, |lat| <= 80, inside the lon box and clear of column 0
    (whose west slot is the periodic copy of column nx-1, field.py:223 -- psi of the bench is not x-periodic)."""
    sys.path.insert(0, ROOT)
    import bench
    a = bench.make_transects(3600, 1800, -180., 180., -90., 90., 64)
    b = bench.make_transects(3600, 1800, -180., 180., -90., 90., 64)
    assert a == b and len(a) == 65 and a[0][0] == (-180., -80.)
    dx = 0.1
    for k, poly in enumerate(a[1:]):
        xy = numpy.array(poly)
        assert 8 <= len(poly) <= 65
        assert xy[:, 0].min() >= -180. + dx - 1e-9 and xy[:, 0].max() <= 180. + 1e-9
        assert numpy.abs(xy[:, 1]).max() <= 80. + 1e-9
        assert numpy.allclose(numpy.round((xy[:, 0] + 180.) / dx) * dx - 180., xy[:, 0], atol=1e-9)   # on nodes
        assert (poly[0] == poly[-1]) == (k % 2 == 1)                                              # half are closed


def test_bench_refuses_more_gpus_than_the_node_has():
    """--gpus N with fewer than N devices visible (and no rehearsal hook) exits non-zero with a message -- it never runs a
    smaller job under the larger name."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('NF_FORCE_DEVICE', 'WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '64', '--steps', '1'], env=env,
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 2 and 'GPU(s) visible' in r.stderr and not r.stdout.strip()
