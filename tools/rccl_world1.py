"""One-rank RCCL smoke: the exact init / all_reduce / barrier calls bench.py makes at N>1, on one GPU (world_size 1).
Run on the GPU box: MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python tools/rccl_world1.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
torch.cuda.set_device(0)
dist.init_process_group(backend='nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
rows = torch.arange(12., dtype=torch.float64, device='cuda').reshape(3, 4)
dist.all_reduce(rows, op=dist.ReduceOp.SUM)
dist.barrier(device_ids=[0])
tt = torch.tensor([1.5], dtype=torch.float64, device='cuda')
dist.all_reduce(tt, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
assert rows.sum().item() == 66.0 and tt.item() == 1.5
# the engine's own communicator over the same group: created collectively (id broadcast, comm_init, agreement, known-answer
# probe), then the rows summed by nf_rows_allreduce -- exactly what nemoflux_amd.dist.reduce_rows does at N > 1
from nemoflux_amd import dist as nfdist
comm = nfdist.native_comm()
assert comm is not None, 'native RCCL communicator could not be created'
info = comm.info()
assert info['world_size'] == 1 and info['rank'] == 0 and info['device_index'] == 0 and 'librccl' in info['library'], info
before = rows.clone()
comm.all_reduce_sum(rows)
torch.cuda.synchronize()
assert torch.equal(rows, before)
print('native communicator:', info)
nfdist.destroy_native_comms()
dist.destroy_process_group()
print('rccl world-1 OK', dist.is_nccl_available())
