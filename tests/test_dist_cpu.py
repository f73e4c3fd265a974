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
    # the cut by whole time steps (full-field outputs per step): step boundaries only, everything covered, balanced in steps
    from nemoflux_amd.dist import slab_range_by_steps
    for nt, nz, world in [(12, 75, 8), (12, 75, 5), (3, 4, 8), (20, 10, 3)]:
        cuts = [slab_range_by_steps(nt, nz, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == nt * nz and all(a[1] == b[0] for a, b in zip(cuts[:-1], cuts[1:]))
        assert all(b % nz == 0 and e % nz == 0 for b, e in cuts)
        steps = [(e - b) // nz for b, e in cuts]
        assert max(steps) - min(steps) <= 1
    assert [slab_range_by_steps(12, 75, r, 8) for r in (0, 3, 7)] == [(0, 75), (300, 450), (750, 900)]


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


@pytest.mark.parametrize('name,world', [('rot36_zt', 2), ('def36_zt', 2), ('rot36_zt', 8)])
def test_gloo_reduce_matches_single(name, world, oracle, cases, tmp_path):
    """world = 8 on rot36_zt (nt * nz = 6 slabs): the job size of BASELINE config 4 with MORE ranks than slabs -- two ranks own
    nothing and contribute exact zeros."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'rows.npy')
    mp.spawn(_worker, args=(world, port, name, out), nprocs=world, join=True)
    got = numpy.load(out)
    m = [c for c in cases if c['name'] == name][0]
    g = load_golden(name)
    pts = oracle.assemble_points(g['bounds_lon'], g['bounds_lat'])
    weights = [oracle.polyline_weights(pts, transect_xyz(t['points'])) for t in m['transects'].values()]
    full = _partial_rows(oracle, g, m, (0, m['nt'] * m['nz']), weights)
    # the reduction order differs from the single-process sum: 1e-13 relative, not bitwise -- relative to what is
    # added up: a closed loop's total is ~0 while the per-slab contributions of a cut in z are not
    from nemoflux_amd.dist import slab_range
    parts = [_partial_rows(oracle, g, m, slab_range(m['nt'], m['nz'], r, world), weights) for r in range(world)]
    if world > m['nt'] * m['nz']:
        assert sum(1 for q in parts if not q.any()) >= world - m['nt'] * m['nz']       # ranks that own nothing add zeros
    scale = 0.          # sum |w f| of the full step: what the closed loop's ~0 is the difference of
    for t in range(m['nt']):
        st = oracle.EdgeFluxState(m['ny'], m['nx'])
        oracle.edge_flux(st, oracle.vertical_integral(g['u'][t], g['thickness'], m['fill_value']),
                         oracle.vertical_integral(g['v'][t], g['thickness'], m['fill_value']), g['arcLengths'], m['sverdrup'])
        scale = max([scale] + [numpy.abs(w.weight * st.integratedVelocity.reshape(-1)[w.cell_edge]).sum() for w in weights])
    assert numpy.abs(got - full).max() <= 1e-13 * scale
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


def _start_ranks(world, env_extra, timeout=120):
    """`world` processes of tools/dist_startup_probe.py over gloo; returns [(returncode, stdout, stderr)] by rank."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK=str(r),
                   WORLD_SIZE=str(world), NF_NATIVE_REDUCE='rehearse', **env_extra)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tools', 'dist_startup_probe.py')], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    out = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            so, se = p.communicate()
            so += '\n<killed by the test: still running>'
        out.append((p.returncode, so, se))
    return out


def test_multi_gpu_start_up_fails_soft():
    """Round-3 verdict W6.  ncclCommInitRank is collective: a rank that fails BEFORE it (no device, librccl missing) used to
    leave the others waiting inside it.  Now every rank first runs the non-collective nf_rccl_preflight, the ranks agree with
    one MIN all-reduce, and only then is the communicator created.  Control flow rehearsed over gloo: rank 1's preflight
    fails (NF_TEST_FAIL_COMM=1) -> BOTH ranks fall back to torch.distributed together, say so on stderr, and the reduce
    gives the right sums.  (On this CPU box rank 0's own preflight fails too -- no GPU -- which is the same path.)"""
    res = _start_ranks(2, {'NF_TEST_FAIL_COMM': '1', 'NF_DIST_TIMEOUT_S': '60'})
    for r, (rc, so, se) in enumerate(res):
        assert rc == 0, (r, so[-1500:], se[-3000:])
        assert f'rank {r}: native False rows [3.0, 0.5]' in so
        assert 'native RCCL communicator unavailable' in se and 'reducing through torch.distributed' in se
    assert 'injected preflight failure on rank 1' in res[1][2]


def test_multi_gpu_start_up_is_bounded():
    """A rank that never reaches the agreement must end the job, not hang it: the waiting rank gives up after
    NF_DIST_TIMEOUT_S with a message and a non-zero exit code (a launcher then ends the other ranks)."""
    import time
    t0 = time.time()
    res = _start_ranks(2, {'NF_TEST_HANG_COMM': '1', 'NF_TEST_HANG_SECONDS': '25', 'NF_DIST_TIMEOUT_S': '5'}, timeout=90)
    rc0, so0, se0 = res[0]
    assert rc0 != 0 and 'rows' not in so0, (so0, se0[-2000:])
    assert 'did not finish within 5 s' in se0 or 'imed out' in se0 or 'timeout' in se0.lower(), se0[-3000:]
    assert time.time() - t0 < 80


def test_visible_gpu_count_reads_the_kfd_topology(tmp_path, monkeypatch):
    """bench.py's launcher counts GPUs without touching HIP: the KFD topology in sysfs (round-4 verdict W6).  A fake tree with
    8 GPU nodes and 1 CPU node (simd_count 0), one GPU node's properties unreadable (a GPU of the host that the container's
    device cgroup hides): 7; narrowed by HIP_VISIBLE_DEVICES; 0 without a KFD node; None when sysfs cannot be read at all."""
    import builtins
    sys.path.insert(0, ROOT)
    import bench
    for k in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        monkeypatch.delenv(k, raising=False)
    nodes = tmp_path / 'kfd' / 'kfd' / 'topology' / 'nodes'
    for n in range(9):
        d = nodes / str(n)
        d.mkdir(parents=True)
        simd = 0 if n == 0 else 1024
        (d / 'properties').write_text(f'cpu_cores_count {96 if n == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\ngfx_target_version 90500\n')
    hidden = str(nodes / '5' / 'properties')
    real_open = builtins.open

    def guarded(path, *a, **kw):      # the test may run as root, which chmod does not stop
        if str(path) == hidden:
            raise PermissionError(13, 'Operation not permitted', hidden)
        return real_open(path, *a, **kw)
    monkeypatch.setattr(bench, 'open', guarded, raising=False)
    kfd = str(tmp_path / 'kfd')
    assert bench.visible_gpu_count(kfd) == 7
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,1,2,3')
    assert bench.visible_gpu_count(kfd) == 4
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,1,2,3,4,5,6,7,8,9')
    assert bench.visible_gpu_count(kfd) == 7                       # the variable cannot add devices
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '')
    assert bench.visible_gpu_count(kfd) == 0
    monkeypatch.delenv('HIP_VISIBLE_DEVICES')
    assert bench.visible_gpu_count(str(tmp_path / 'no_such_kfd')) == 0
    (nodes / '3' / 'properties').write_text('simd_count many\n')   # unparsable: the count is unknown, not a guess
    assert bench.visible_gpu_count(kfd) is None
    monkeypatch.setenv('ROCR_VISIBLE_DEVICES', '0,1')
    assert bench.visible_gpu_count(kfd) == 2


def test_rank_environment_is_set_by_the_rank_itself(monkeypatch):
    """Both ways of starting N ranks (`python bench.py --gpus N` and `torchrun ... bench.py --gpus N`) run the same ranks: what
    a rank needs in its environment is set by nemoflux_amd.dist.rank_environment(), first thing in init_from_env, and the
    launcher adds nothing of its own.  Values already present win."""
    import inspect
    sys.path.insert(0, ROOT)
    import bench
    from nemoflux_amd import dist as nfdist
    for k in ('HSA_ENABLE_IPC_MODE_LEGACY', 'NF_DIST_TIMEOUT_S', 'WORLD_SIZE'):
        monkeypatch.delenv(k, raising=False)
    nfdist.rank_environment()
    assert os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and os.environ['NF_DIST_TIMEOUT_S'] == '300'
    monkeypatch.setenv('NF_DIST_TIMEOUT_S', '17')
    nfdist.rank_environment()
    assert os.environ['NF_DIST_TIMEOUT_S'] == '17' and nfdist.startup_timeout_s() == 17.0
    assert nfdist.collective_timeout_s() == 1800.0                 # the run-time bound is a different one
    src = inspect.getsource(bench.self_launch)
    assert 'setdefault' not in src and 'env=' not in src           # the launcher hands its own environment down unchanged
    assert inspect.getsource(nfdist.init_from_env).split('rank_environment()')[0].count('torch.cuda') == 0


def test_rank_refuses_over_subscription_before_the_rendezvous(monkeypatch, capsys):
    """Round-5 verdict W4b.  Under `torchrun --nproc-per-node 8` on a node where a rank sees 4 GPUs, ranks used to be mapped
    `local % ndev` -- two per device -- and fail later inside RCCL.  Now the rank refuses at start, before init_process_group,
    with exit code 2 and a message.  Not refused: one visible device per rank (a launcher that sets HIP_VISIBLE_DEVICES for
    each rank: index 0), the NF_FORCE_DEVICE rehearsal hook, and as many devices as ranks."""
    import torch
    from nemoflux_amd import dist as nfdist
    for k in ('NF_FORCE_DEVICE', 'LOCAL_WORLD_SIZE', 'WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '8')
    assert nfdist._device_for_local_rank(5, 8) == 5                      # one device per rank
    assert nfdist._device_for_local_rank(5, 1) == 0                      # HIP_VISIBLE_DEVICES per rank
    assert nfdist._device_for_local_rank(5, 0) == 5                      # no GPU at all (CPU tests over gloo)
    with pytest.raises(SystemExit) as e:
        nfdist._device_for_local_rank(1, 4)                              # 8 ranks, 4 devices: even a rank that "fits" refuses
    assert e.value.code == 2
    assert '8 ranks on this node but 4 GPUs visible' in capsys.readouterr().err
    monkeypatch.delenv('LOCAL_WORLD_SIZE')
    with pytest.raises(SystemExit):
        nfdist._device_for_local_rank(6, 4)                              # no LOCAL_WORLD_SIZE (another launcher): index out of range
    assert nfdist._device_for_local_rank(3, 4) == 3
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '8')
    monkeypatch.setenv('NF_FORCE_DEVICE', '0')
    assert nfdist._device_for_local_rank(0, 4) == 0                      # the rehearsal hook lifts the check
    monkeypatch.delenv('NF_FORCE_DEVICE')
    # through init_from_env, with a node that shows 4 devices: exits before any process group exists
    called = []
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 4)
    monkeypatch.setattr(torch.cuda, 'set_device', lambda d: called.append(('set_device', d)))
    monkeypatch.setattr(nfdist.dist, 'init_process_group', lambda *a, **kw: called.append('init_process_group'))
    monkeypatch.setenv('WORLD_SIZE', '8')
    monkeypatch.setenv('RANK', '6')
    monkeypatch.setenv('LOCAL_RANK', '6')
    with pytest.raises(SystemExit) as e:
        nfdist.init_from_env()
    assert e.value.code == 2 and called == []
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '4')
    monkeypatch.setenv('WORLD_SIZE', '4')
    monkeypatch.setenv('RANK', '3')
    monkeypatch.setenv('LOCAL_RANK', '3')
    assert nfdist.init_from_env() == (3, 4, 3)
    assert called == [('set_device', 3), 'init_process_group']


def _agree_worker(rank, world, port, out_dir):
    import json
    from nemoflux_amd import dist as nfdist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), NF_DIST_TIMEOUT_S='60')
    nfdist.init_from_env(backend='gloo')
    res = {}
    res['all_yes'] = nfdist._agree(1)
    res['one_no'] = nfdist._agree(0 if rank == 2 else 1)
    res['again_yes'] = nfdist._agree(1)                       # the sequence number keeps the calls apart
    # two sub-groups with the same first rank, last rank and size: {0,1,3} and {0,2,3} (their keys collided before round 6)
    ga = dist.new_group([0, 1, 3], backend='gloo')
    gb = dist.new_group([0, 2, 3], backend='gloo')
    if rank in (0, 1, 3):
        res['group_a'] = nfdist._agree(0 if rank == 1 else 1, ga)
    if rank in (0, 2, 3):
        res['group_b'] = nfdist._agree(1, gb)
    if rank in (0, 1, 3):
        res['group_a_2'] = nfdist._agree(1, ga)
    with open(os.path.join(out_dir, f'agree{rank}.json'), 'w') as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


def test_agreement_through_the_store_with_subgroups(tmp_path):
    """nemoflux_amd.dist._agree (the MIN over ranks that precedes ncclCommInitRank) through the rendezvous store, 4 gloo
    ranks: mixed flags, repeated calls, and two sub-groups that share first rank, last rank and size -- a rank in both must
    not read the other group's flag (round-5 advisor)."""
    import json
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    mp.spawn(_agree_worker, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    res = [json.load(open(tmp_path / f'agree{r}.json')) for r in range(4)]
    for r in range(4):
        assert (res[r]['all_yes'], res[r]['one_no'], res[r]['again_yes']) == (1, 0, 1), (r, res[r])
    for r in (0, 1, 3):
        assert res[r]['group_a'] == 0 and res[r]['group_a_2'] == 1, (r, res[r])
    for r in (0, 2, 3):
        assert res[r]['group_b'] == 1, (r, res[r])
