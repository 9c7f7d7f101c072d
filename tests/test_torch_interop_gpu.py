"""The sharded driver on one real GPU: torch's HIP stream handed to the context, the context's
device buffers wrapped as torch tensors (no copies), and the RCCL collectives of both exchange
schemes run with world_size 1 (the only multi-rank-free way to exercise them on a 1-GPU box).
Runs in a subprocess so that torch initialises the GPU before the library does, as bench.py does."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import nbody_amd as nb
from nbody_amd.shard import ShardedSim, HipShardBackend
torch.cuda.set_device(0)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n = 65536
m, pos, vel = nb.gen_plummer(n, seed=3)
ref = nb.NBodySim(n, law=nb.LAW_PLUMMER); ref.upload(m, pos, vel); ref.step_allpairs(2); rp, rv, ra = ref.download(); ref.close()
be = HipShardBackend(n, 0, n, 0, law=nb.LAW_PLUMMER)
assert be.pair_sharding
be.upload(m, pos, vel)
sim = ShardedSim(be, n, 0, 1)
# world == 1 skips the collectives; drive them by hand exactly as ShardedSim does for world > 1
assert be.torch_stream.cuda_stream != 0 and be.sim.stream() == be.torch_stream.cuda_stream
for _ in range(2):
    be.forces_partial(0, 1)
    with torch.cuda.stream(be.torch_stream):
        part = be.acc_partial()
        assert part.is_cuda and part.shape == (n, 4) and part.data_ptr() == be.sim.device_buffer(nb.BUF_ACC_PARTIAL)[0]
        dist.all_reduce(part)
    be.integrate_partial()
    with torch.cuda.stream(be.torch_stream):
        full = be.posm()
        own = full[0:n].clone()
        dist.all_gather_into_tensor(full, own)
torch.cuda.synchronize()
p, v, a = be.download()
assert np.array_equal(p, rp) and np.array_equal(v, rv) and np.array_equal(a, ra)
assert np.array_equal(be.posm().cpu().numpy()[:, :3], p[:, :3])
# the same two steps with the exchange inside the library (torch only carries the communicator's id);
# torch's own RCCL is already in the process and is the copy the library binds
be2 = HipShardBackend(n, 0, n, 0, law=nb.LAW_PLUMMER)
be2.upload(m, pos, vel)
sim2 = ShardedSim(be2, n, 0, 1, library_comm=True)
sim2.step_allpairs(2)
p2, v2, a2 = be2.download()
assert np.array_equal(p2, rp) and np.array_equal(v2, rv) and np.array_equal(a2, ra)
be3 = HipShardBackend(n, 0, n, 0, law=nb.LAW_PLUMMER, bh_ranked=True)
be3.upload(m, pos, vel)
ShardedSim(be3, n, 0, 1, library_comm=True).step_bh(2)
ref = nb.NBodySim(n, law=nb.LAW_PLUMMER); ref.upload(m, pos, vel); ref.step_bh(2); bp, bv, ba = ref.download(); ref.close()
p3, v3, a3 = be3.download()
assert np.array_equal(p3, bp) and np.array_equal(v3, bv) and np.array_equal(a3, ba)
dist.destroy_process_group()
print("INTEROP_OK")
'''


def test_sharded_backend_on_one_gpu(tmp_path):
    script = tmp_path / "interop.py"
    script.write_text(SCRIPT)
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=300)
    assert "INTEROP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
