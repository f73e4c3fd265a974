"""Multi-GPU layer: (t,z) slab sharding + ONE reduce of the per-segment / per-transect rows.

The reference has no distributed code; this is the MI355X-native addition of SURVEY.md 8e: every
(t,z) slab contributes additively and independently to every transect total, so the flattened slab index
s = t*nz + z is cut into contiguous ranges, one per rank (one process per GPU); each rank integrates only its
slabs (nf_field_set_slab_range) and the (nt, nseg+ntransect) float64 rows are summed with a single
torch.distributed all_reduce -- RCCL over xGMI on GPUs ('nccl' backend), gloo on CPU for tests.  The message is
<= ~0.4 MB (latency-bound), so no bucketing is needed.
"""
import os

import torch
import torch.distributed as dist


def slab_range(nt, nz, rank, world):
    """Contiguous share of the nt*nz slabs for `rank` of `world` (balanced to within one slab)."""
    total = nt * nz
    return (rank * total) // world, ((rank + 1) * total) // world


def time_steps_touched(srange, nz):
    """[t_begin, t_end) of the time steps that contain at least one owned slab."""
    b, e = srange
    if e <= b:
        return 0, 0
    return b // nz, (e - 1) // nz + 1


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun) if WORLD_SIZE > 1.
    Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if 'NF_FORCE_DEVICE' in os.environ:      # rehearsal of N ranks on a one-GPU box (with NF_DIST_BACKEND=gloo)
        local = int(os.environ['NF_FORCE_DEVICE'])
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev and local >= ndev:   # launcher that restricts every rank to its own device (HIP_VISIBLE_DEVICES per rank)
        local %= ndev
    backend = backend or os.environ.get('NF_DIST_BACKEND')
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':     # RCCL: bind the communicator to this rank's GPU at init (one process per GPU)
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


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


def reduce_rows(rows, group=None):
    """Sum the per-rank partial rows in place (all ranks get the totals).  One collective."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        all_reduce(rows, dist.ReduceOp.SUM, group)
    return rows


def virtual_base(tensor, first_slab, slab_elems):
    """HBM address such that address + s*slab_bytes is slab s of the GLOBAL (nt,nz,ny,nx) array, when
    `tensor` holds only the slabs from `first_slab` on.  The engine never dereferences slabs it does not own."""
    return tensor.data_ptr() - first_slab * slab_elems * tensor.element_size()
