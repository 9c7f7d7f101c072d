"""Per-rank cost of a pair-sharded all-pairs step on ONE GPU: rank r's share of a world of W
(its pair tiles, reduction of its partial rows, integration of its targets).  No exchange."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb
from nbody_amd.shard import shard_range
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
m, p, v = nb.gen_plummer(n, seed=42)
for world in [int(w) for w in os.environ.get("WORLDS", "1,2,4,8").split(",")]:
    for rank in sorted({0, world // 2, world - 1}, reverse=True):
        b, c = shard_range(n, rank, world)
        with nb.NBodySim(n, law=nb.LAW_PLUMMER, shard_begin=b, shard_count=c) as sim:
            sim.upload(m, p, v)
            sim.allpairs_partial(rank, world); sim.integrate_partial(); sim.sync()
            sim.set_kernel_timing(True); sim.kernel_time()
            k = 3
            t0 = time.perf_counter()
            for _ in range(k):
                sim.allpairs_partial(rank, world); sim.integrate_partial()
            sim.sync()
            dt = (time.perf_counter() - t0) / k
            kms, _ = sim.kernel_time()
        print(f"n={n} world={world} rank={rank}: {dt*1e3:.3f} ms/step (pair kernel {kms:.3f} ms) -> x{world} ideal {161.0/world:.2f}", flush=True)
