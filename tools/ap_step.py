"""All-pairs step time (PLUMMER law) at the given sizes: wall time per step over a queued run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nb

for n, steps in ((65536, 200), (131072, 100), (1 << 20, 5)):
    m, p, v = nb.gen_plummer(n, seed=42)
    with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
        sim.upload(m, p, v)
        sim.step_allpairs(2); sim.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.time(); sim.step_allpairs(steps); sim.sync()
            best = min(best, (time.time() - t0) / steps)
        pos, _, _ = sim.download()
    print(f"n={n}: {best * 1e3:.4f} ms/step  {n * n / best:.4e} interactions/s  pos checksum {np.abs(pos[:, :3]).sum(dtype=np.float64):.9e}", flush=True)
