"""Per-rank cost of a replicated-state Barnes-Hut step on ONE GPU: the share of one rank of a
world of W (octree build over all N bodies + traversal of the rank's run of the sorted order,
then un-ranking + integration of all N bodies).  The all-gather between the two is not timed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb
from nbody_amd.shard import shard_range
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
m, p, v = nb.gen_plummer(n, seed=42)
with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
    sim.upload(m, p, v)
    sim.step_bh(2)
    for world in (1, 2, 4, 8):
        for rank in range(world):
            b, c = shard_range(n, rank, world)
            tf = ti = 0.0
            k = 4
            sim.set_kernel_timing(True); sim.kernel_time(bh=True)
            for _ in range(k):
                sim.sync(); t0 = time.perf_counter()
                sim.bh_forces_ranked(b, c)
                sim.sync(); tf += time.perf_counter() - t0
            wms, _ = sim.kernel_time(bh=True)
            sim.set_kernel_timing(False)
            for _ in range(k):
                sim.bh_forces_ranked(0, n)       # what the all-gather would deliver
                sim.sync(); t0 = time.perf_counter()
                sim.integrate_ranked()
                sim.sync(); ti += time.perf_counter() - t0
            print(f"n={n} world={world} rank={rank}: build+walk {tf/k*1e3:.3f} ms (walk kernel {wms:.3f})  "
                  f"unrank+integrate {ti/k*1e3:.3f} ms", flush=True)
