"""Long runs: thousands of Barnes-Hut steps (tree rebuilt every step, capacity growth, clamping,
collapse toward the centre) and a few hundred all-pairs steps; positions must stay finite and
the error flag clear."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb
n = 200000
m, p, v = nb.gen_plummer(n, seed=5)          # cold start: the sphere collapses and rebounds
for law in (nb.LAW_PLUMMER, nb.LAW_WGSL_BIAS):
    with nb.NBodySim(n, law=law) as sim:
        sim.upload(m, p, v)
        t0 = time.time()
        for chunk in range(10):
            sim.step_bh(300)
            pos, vel, acc = sim.download()
            c = sim.counters()
            assert np.isfinite(pos).all() and np.isfinite(vel).all(), "non-finite state"
            assert pos[:, :3].min() >= 0.0 and pos[:, :3].max() <= 250.0
            print(f"law {law} after {(chunk + 1) * 300} steps: nodes {c['octree_nodes']}, "
                  f"r_median {np.median(np.linalg.norm(pos[:, :3] - 125.0, axis=1)):.2f}, {time.time() - t0:.1f} s", flush=True)
n2 = 65536
m, p, v = nb.gen_plummer(n2, seed=6, equilibrium=True)
with nb.NBodySim(n2, law=nb.LAW_PLUMMER) as sim:
    sim.upload(m, p, v)
    k0, u0 = sim.energy()
    sim.step_allpairs(500)
    k1, u1 = sim.energy()
    pos, _, _ = sim.download()
    assert np.isfinite(pos).all()
    print(f"all-pairs N={n2}, 500 steps: energy {k0 + u0:.6e} -> {k1 + u1:.6e}")
print("SOAK_OK")
