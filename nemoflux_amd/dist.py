"""Multi-GPU layer: (t,z) slab sharding + ONE reduce of the per-segment / per-transect rows.

The reference has no distributed code; this is the MI355X-native addition of SURVEY.md 8e: every
(t,z) slab contributes additively and independently to every transect total, so the flattened slab index
s = t*nz + z is cut into contiguous ranges, one per rank (one process per GPU); each rank integrates only its
slabs (nf_field_set_slab_range) and the (nt, nseg+ntransect) float64 rows are summed with a single
all-reduce over RCCL / xGMI.  The message is <= ~0.4 MB (latency-bound), so no bucketing is needed.

On GPUs ('nccl' backend = RCCL) the reduce is the C ABI's own: nf_rows_allreduce (ncclAllReduce(sum, ncclDouble) in
csrc/nf_reduce.hip) on a communicator the library creates with nf_rccl_comm_init -- torch.distributed only carries the
128-byte unique id from rank 0 to the others, the way a plain-C client would carry it over MPI or a socket -- after every
rank passed nf_rccl_preflight and the ranks agreed on that (start-up fails soft: see native_comm).  gloo (CPU
tests, and the rehearsal of N ranks on a one-GPU box) goes through torch.distributed's all_reduce.  NF_NATIVE_REDUCE=0
keeps torch.distributed's RCCL call on GPUs too.
"""
import ctypes
import os
import sys

import torch
import torch.distributed as dist


def slab_range(nt, nz, rank, world):
    """Contiguous share of the nt*nz slabs for `rank` of `world` (balanced to within one slab)."""
    total = nt * nz
    return (rank * total) // world, ((rank + 1) * total) // world


def slab_range_by_steps(nt, nz, rank, world):
    """The cut for callers that want FULL-FIELD outputs per time step (integratedVelocity, the |.| arrays): whole time steps
    per rank (balanced to within one step), so every step's planes are complete on the rank that owns it (SURVEY.md 8e:
    "when full-field outputs are requested, shard by t").  Same (begin, end) slab convention as slab_range; with fewer steps
    than ranks the last ranks own nothing.  The transect totals need the same single all-reduce either way."""
    t0, t1 = (rank * nt) // world, ((rank + 1) * nt) // world
    return t0 * nz, t1 * nz


def time_steps_touched(srange, nz):
    """[t_begin, t_end) of the time steps that contain at least one owned slab."""
    b, e = srange
    if e <= b:
        return 0, 0
    return b // nz, (e - 1) // nz + 1


RANK_THREADS = 4      # host threads of one rank of a multi-process run (NF_RANK_THREADS)


def rank_environment():
    """What every rank of a multi-process run needs in its environment BEFORE its first GPU call, whoever started it
    (`python bench.py --gpus N`, which starts its own ranks, or `torchrun ... bench.py --gpus N`: the ranks are identical
    either way -- round-4 verdict W6).  `import torch` and loading libnemoflux_amd.so do not initialise HIP, so this is
    early enough when it runs first thing in init_from_env.
      HSA_ENABLE_IPC_MODE_LEGACY=0  dmabuf IPC: what RCCL needs between the processes of one host (read at hsa_init)
      NF_DIST_TIMEOUT_S=300         a rank that cannot join ends the job with a message (_Deadline), it does not hang it
    Values already present win.  The host threads of a rank are set through torch (the OpenMP runtime has read
    OMP_NUM_THREADS when torch was imported; torchrun exports OMP_NUM_THREADS=1 for N > 1 on both paths)."""
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.setdefault('NF_DIST_TIMEOUT_S', '300')
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        try:
            torch.set_num_threads(max(1, int(os.environ.get('NF_RANK_THREADS', RANK_THREADS))))
        except (RuntimeError, ValueError):
            pass


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun) if WORLD_SIZE > 1.
    Returns (rank, world, local_rank)."""
    rank_environment()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if 'NF_FORCE_DEVICE' in os.environ:      # rehearsal of N ranks on a one-GPU box (with NF_DIST_BACKEND=gloo)
        local = int(os.environ['NF_FORCE_DEVICE'])
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    local = _device_for_local_rank(local, ndev)
    backend = backend or os.environ.get('NF_DIST_BACKEND')
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        import datetime
        # Two different bounds (round-4 advisor): a rank that never arrives must not hang the job -- the rendezvous gives up
        # after startup_timeout_s() (_Deadline) --, while the collectives of a running job keep their own, longer limit
        timeout = datetime.timedelta(seconds=collective_timeout_s())
        with _Deadline(startup_timeout_s(), 'the torch.distributed rendezvous (init_process_group)'):
            if backend == 'nccl':     # RCCL: bind the communicator to this rank's GPU at init (one process per GPU)
                torch.cuda.set_device(local)
                dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout,
                                        device_id=torch.device('cuda', local))
            else:
                dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


class OverSubscribed(SystemExit):
    """More ranks on this node than GPUs this rank can see: refused at rank start (exit code 2)."""


def _device_for_local_rank(local, ndev):
    """The device index of this rank, or a refusal BEFORE init_process_group (round-5 verdict W4b).

    One process per GPU: with LOCAL_WORLD_SIZE ranks on this node (torchrun exports it) and fewer devices visible to this rank,
    two ranks would land on one device and fail later, inside RCCL (duplicate device in ncclCommInitRank) -- so the rank says
    so now and exits non-zero; torch.distributed.run then ends the others.  Two set-ups are not over-subscription:
      * ndev == 1: a launcher that gives every rank its own device through HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES -- every
        rank then sees exactly one device, index 0;
      * NF_FORCE_DEVICE: the rehearsal hook (N ranks sharing one GPU over gloo), handled by the caller."""
    if not ndev or 'NF_FORCE_DEVICE' in os.environ:
        return local
    try:
        local_world = int(os.environ.get('LOCAL_WORLD_SIZE', '0'))
    except ValueError:
        local_world = 0
    if ndev == 1:
        return 0
    if local_world > ndev or local >= ndev:
        msg = (f'nemoflux_amd.dist: rank {os.environ.get("RANK", "?")} (local rank {local}): '
               f'{local_world or "more than " + str(ndev)} ranks on this node but {ndev} GPUs visible; one process per GPU -- '
               'start at most that many ranks, give every rank its own HIP_VISIBLE_DEVICES, or set NF_FORCE_DEVICE (with '
               'NF_DIST_BACKEND=gloo) for a rehearsal on one GPU')
        print('# ' + msg, file=sys.stderr, flush=True)
        raise OverSubscribed(2)
    return local


def all_reduce(tensor, op=None, group=None):
    """In-place all_reduce that also works for an HBM tensor over a host-only backend: RCCL ('nccl') reduces HBM tensors
    directly over xGMI; gloo (CPU tests, and the rehearsal of N ranks on a one-GPU box) gets a host copy."""
    op = dist.ReduceOp.SUM if op is None else op
    if tensor.is_cuda and dist.get_backend(group) != 'nccl':
        host = tensor.cpu()
        dist.all_reduce(host, op=op, group=group)
        tensor.copy_(host)
    else:
        dist.all_reduce(tensor, op=op, group=group)
    return tensor


class _Deadline(object):
    """Bounds a collective start-up step: if the block has not finished after `seconds`, the process says what it was
    waiting for on stderr and exits with code 3 -- the launcher (torch.distributed.run) then ends the other ranks.  A rank
    that cannot join must end the job, not hang it.  It exits; it never re-executes anything."""

    def __init__(self, seconds, what):
        self.seconds, self.what = float(seconds), what

    def __enter__(self):
        import threading

        def expire():
            print(f'# nemoflux_amd.dist: rank {os.environ.get("RANK", "?")}: {self.what} did not finish within '
                  f'{self.seconds:.0f} s (another rank is missing or stuck); ending this rank', file=sys.stderr, flush=True)
            os._exit(3)
        self._timer = threading.Timer(self.seconds, expire)
        self._timer.daemon = True
        self._timer.start()
        return self

    def __exit__(self, *exc):
        self._timer.cancel()
        return False


def startup_timeout_s():
    """Seconds a rank waits in a collective start-up step before it gives up (NF_DIST_TIMEOUT_S, default 300)."""
    return float(os.environ.get('NF_DIST_TIMEOUT_S', '300'))


def collective_timeout_s():
    """Seconds a torch.distributed collective of a RUNNING job may take (NF_DIST_COLLECTIVE_TIMEOUT_S, default 1800 --
    torch's own default for gloo; the start-up steps have their own, shorter bound: startup_timeout_s)."""
    return float(os.environ.get('NF_DIST_COLLECTIVE_TIMEOUT_S', '1800'))


_agree_seq = {}


def _agree(flag, group=None):
    """MIN over the ranks of an int flag, through the rendezvous STORE (host-side key/value, TCP): a rank whose GPU turned
    out to be unusable can still say so -- on the 'nccl' backend an all_reduce would need a device tensor on exactly the
    device that failed, the rank would raise instead of reporting 0 and the others would wait for it (round-4 advisor).
    Every rank of the group calls this the same number of times (the sequence number is part of the key).  Falls back to an
    all_reduce on the group's backend when the process group has no store to offer."""
    flag = int(flag)
    try:
        store = dist.distributed_c10d._get_default_store()
        ranks = list(dist.get_process_group_ranks(group if group is not None else dist.group.WORLD))
        me = dist.get_rank()
    except Exception:
        store = None
    if store is None:
        try:
            dev = 'cuda' if dist.get_backend(group) == 'nccl' else 'cpu'
            t = torch.tensor([flag], dtype=torch.int32, device=dev)
        except Exception:       # no usable device on this rank: it must still take part, with the answer "no"
            t = torch.tensor([0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return int(t.item())
    import datetime
    # the key names the group by ALL of its ranks (two sub-groups may share first rank, last rank and size: {0,1,3} and
    # {0,2,3} -- round-5 advisor), and carries the group's own call counter
    import hashlib
    tag = hashlib.sha1(','.join(str(r) for r in ranks).encode()).hexdigest()[:16]
    seq = _agree_seq[tag] = _agree_seq.get(tag, 0) + 1
    prefix = f'nemoflux_amd/agree/{tag}/{seq}'
    store.set(f'{prefix}/{me}', str(flag))
    keys = [f'{prefix}/{r}' for r in ranks]
    store.wait(keys, datetime.timedelta(seconds=startup_timeout_s()))
    return min(int(store.get(k)) for k in keys)


class NativeComm(object):
    """The engine's own RCCL communicator over the ranks of a torch.distributed group (nf_rccl_* of the C ABI).

    ncclCommInitRank is collective, so nothing that can fail on ONE rank may sit between the ranks' agreement and that call:
    `preflight` (not collective: librccl resolves, the rank has a usable device, rank 0 has the unique id) runs first on
    every rank, the outcomes are combined with one MIN all-reduce over torch.distributed, and only when every rank passed
    do all of them call `connect`."""

    def __init__(self):
        from ._lib import lib, check
        self._lib, self._check = lib, check
        self.ptr = None
        self._ident = None

    def preflight(self, group=None):
        """This rank's share of the start-up that needs no other rank.  Raises on failure."""
        rank = dist.get_rank(group)
        inject = os.environ.get('NF_TEST_FAIL_COMM')          # test hook: the named rank fails here
        if inject is not None and int(inject) == rank:
            raise RuntimeError(f'NF_TEST_FAIL_COMM={inject}: injected preflight failure on rank {rank}')
        dev = ctypes.c_int(-1)
        self._check(self._lib.nf_rccl_preflight(ctypes.byref(dev)))
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            self._check(self._lib.nf_rccl_unique_id(buf))
            self._ident = buf.raw
        return dev.value

    def connect(self, group=None):
        """Collective: every rank of the group, after all of them passed preflight."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        ident = [self._ident]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(ident, src=src, group=group)
        ptr = ctypes.c_void_p()
        self._check(self._lib.nf_rccl_comm_init(ctypes.byref(ptr), world, ctypes.c_char_p(ident[0]), rank))
        self.ptr = ptr

    def all_reduce_sum(self, rows):
        assert rows.is_cuda and rows.dtype == torch.float64 and rows.is_contiguous()
        self._check(self._lib.nf_rows_allreduce(self.ptr, ctypes.c_void_p(rows.data_ptr()), rows.numel(),
                                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def info(self):
        """What the communicator says about itself: {'world_size', 'rank', 'device_index', 'library'}."""
        n, r, d = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self._check(self._lib.nf_rccl_comm_info(self.ptr, ctypes.byref(n), ctypes.byref(r), ctypes.byref(d)))
        path = ctypes.create_string_buffer(512)
        self._check(self._lib.nf_rccl_library(path, 512))
        return {'world_size': n.value, 'rank': r.value, 'device_index': d.value, 'library': path.value.decode()}

    def destroy(self):
        if getattr(self, 'ptr', None):
            torch.cuda.synchronize()
            self._lib.nf_rccl_comm_destroy(self.ptr)
            self.ptr = None


_native = {}   # group -> NativeComm, or False when it could not be created on every rank
fell_back_reason = {}   # group -> why the ranks reduce through torch.distributed instead (bench.py: reduce.fell_back)


def _fallback(group, err, comm=None):
    if comm is not None:
        try:
            comm.destroy()
        except Exception:
            pass
    if err or dist.get_rank(group) == 0:
        print(f'# nemoflux_amd.dist: native RCCL communicator unavailable ({err or "failed on another rank"}); '
              'reducing through torch.distributed', file=sys.stderr, flush=True)
    _native[group] = False
    fell_back_reason[group] = err or 'failed on another rank'
    return None


def native_comm(group=None):
    """The NativeComm of `group` (created at first use, collectively), or None: not the 'nccl' backend, switched off with
    NF_NATIVE_REDUCE=0, or the communicator could not be created on every rank (reported once on stderr; the ranks then
    all reduce through torch.distributed).  NF_NATIVE_REDUCE=rehearse runs the start-up protocol up to the agreement on
    any backend (gloo on CPU, or N ranks sharing one GPU): the control flow of a failed start-up can be tested without RCCL.
    Every collective step is bounded by startup_timeout_s(): a rank that waits longer ends the job with a message."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    mode = os.environ.get('NF_NATIVE_REDUCE', '1')
    is_rccl = dist.get_backend(group) == 'nccl'
    if mode == '0' or not (is_rccl or mode == 'rehearse'):
        return None
    if group in _native:
        return _native[group] or None
    comm, err = None, ''
    # 1. what each rank can check on its own -- no rank raises, the ranks must agree on the path they take
    try:
        comm = NativeComm()
        comm.preflight(group)
    except Exception as e:
        err = f'{type(e).__name__}: {e}'
    fell_back_reason.pop(group, None)
    # 2. agreement BEFORE the collective ncclCommInitRank (a rank that failed above would leave the others inside it)
    with _Deadline(startup_timeout_s(), 'the agreement before nf_rccl_comm_init'):
        all_ok = _agree(0 if err else 1, group)
    if not all_ok:
        return _fallback(group, err, comm)
    if not is_rccl:      # rehearsal on gloo: N ranks on one device cannot form an RCCL communicator (duplicate device)
        return _fallback(group, 'rehearsal on the %s backend: no RCCL communicator is attempted' % dist.get_backend(group), comm)
    # 3. every rank is able: create the communicator together, then one known-answer all-reduce before it carries real rows
    with _Deadline(startup_timeout_s(), 'nf_rccl_comm_init (ncclCommInitRank over all ranks)'):
        try:
            comm.connect(group)
            world, rank = dist.get_world_size(group), dist.get_rank(group)
            probe = torch.tensor([rank + 1.0, 0.5], dtype=torch.float64, device='cuda')
            comm.all_reduce_sum(probe)
            torch.cuda.synchronize()
            if probe.tolist() != [world * (world + 1) / 2.0, 0.5 * world]:
                err = f'self-test gave {probe.tolist()}'
        except Exception as e:
            err = f'{type(e).__name__}: {e}'
        all_ok = _agree(0 if err else 1, group)
    if not all_ok:
        return _fallback(group, err, comm)
    _native[group] = comm
    return comm


def destroy_native_comms():
    for comm in _native.values():
        if comm:
            comm.destroy()
    _native.clear()


def reduce_rows(rows, group=None):
    """Sum the per-rank partial rows in place (all ranks get the totals).  One collective."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        comm = native_comm(group) if rows.is_cuda and rows.dtype == torch.float64 and rows.is_contiguous() else None
        if comm is not None:
            comm.all_reduce_sum(rows)
        else:
            all_reduce(rows, dist.ReduceOp.SUM, group)
    return rows


def virtual_base(tensor, first_slab, slab_elems):
    """HBM address such that address + s*slab_bytes is slab s of the GLOBAL (nt,nz,ny,nx) array, when
    `tensor` holds only the slabs from `first_slab` on.  The engine never dereferences slabs it does not own."""
    return tensor.data_ptr() - first_slab * slab_elems * tensor.element_size()
