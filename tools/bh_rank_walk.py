"""Traversal kernel time of one rank's run of the sorted order (N = 16M, worlds of 8, 4 and 2): how the traversal
scales when a rank walks a part of a replicated tree."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import nbody_amd as nb
from nbody_amd.shard import shard_range
n = 1 << 24
m, p, v = nb.gen_plummer(n, seed=42)
with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
    sim.upload(m, p, v)
    sim.step_bh(2)
    for world in (8, 4, 2):
        for rank in (0, world // 2):
            b, c = shard_range(n, rank, world)
            sim.set_kernel_timing(True); sim.kernel_time(bh=True)
            for _ in range(6):
                sim.bh_forces_ranked(b, c)
            sim.sync()
            wms, _ = sim.kernel_time(bh=True)
            sim.set_kernel_timing(False)
            print(f"world={world} rank={rank}: walk kernel {wms:.3f} ms", flush=True)
