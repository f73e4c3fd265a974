"""One rank of a gloo job that walks nemoflux_amd.dist's multi-GPU start-up (preflight -> agreement -> [RCCL communicator]
-> reduce) without needing a GPU: tests/test_dist_cpu.py starts N of these with NF_NATIVE_REDUCE=rehearse and the fault
hooks NF_TEST_FAIL_COMM=<rank> (that rank's preflight fails) / NF_TEST_HANG_COMM=<rank> (that rank never reaches the
agreement).  Prints 'rows <sum>' when the reduce went through."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
from nemoflux_amd import dist as nfdist  # noqa: E402

rank, world, _ = nfdist.init_from_env(backend='gloo')
if os.environ.get('NF_TEST_HANG_COMM') == str(rank):
    time.sleep(float(os.environ.get('NF_TEST_HANG_SECONDS', '120')))     # this rank never arrives
comm = nfdist.native_comm()
rows = torch.tensor([rank + 1.0, 0.25], dtype=torch.float64)
nfdist.reduce_rows(rows)
print(f'rank {rank}: native {comm is not None} rows {rows.tolist()}', flush=True)
nfdist.destroy_native_comms()
torch.distributed.destroy_process_group()
