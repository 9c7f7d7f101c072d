"""Per-rank cost of a sharded Barnes-Hut step on ONE GPU: the share of one rank of a world of W
(replicated build over all N bodies, traversal of N/W targets).  The exchange is replaced by a
device copy of the other ranks' (unmoved) bodies, so no communication is timed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import nbody_amd as nb
from nbody_amd.shard import shard_range, HipShardBackend
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
m, p, v = nb.gen_plummer(n, seed=42)
for world in (1, 2, 4, 8):
    for rank in sorted({0, world - 1}):
        b, c = shard_range(n, rank, world)
        be = HipShardBackend(n, b, c, 0, law=nb.LAW_PLUMMER)
        be.upload(m, p, v)
        def step():
            prev = be.posm()
            be.step("bh")
            new = be.posm()
            if b: new[:b].copy_(prev[:b])
            if b + c < n: new[b + c:].copy_(prev[b + c:])
        for _ in range(2): step()
        torch.cuda.synchronize()
        be.sim.set_kernel_timing(True); be.sim.kernel_time(bh=True)
        t0 = time.perf_counter()
        k = 10
        for _ in range(k): step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / k
        wms, _ = be.sim.kernel_time(bh=True)
        be.sim.close()
        print(f"n={n} world={world} rank={rank}: {dt*1e3:.3f} ms/step (walk kernel {wms:.3f} ms)", flush=True)
