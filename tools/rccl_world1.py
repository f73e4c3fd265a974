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
dist.destroy_process_group()
print('rccl world-1 OK', dist.is_nccl_available())
