"""Barnes-Hut on one GPU beyond the benchmark sizes: builds and traverses N = argv[1] Plummer
bodies (default 64M), times the step, and checks the accelerations of a few sample bodies against
the exact f64 sum over all bodies (the traversal's truncation error at theta = 0.5 is ~1e-3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t0 = time.time()
m, p, v = nb.gen_plummer(n, seed=42)
print(f"n={n}: generated in {time.time() - t0:.1f} s", flush=True)
G, eps2 = 0.0066743, 1.0
with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
    sim.upload(m, p, v)
    sim.build_octree(); sim.sync()
    acc = sim.accel_bh()
    print(f"nodes {sim.counters()['octree_nodes']}  finite {bool(np.isfinite(acc[:, :3]).all())}", flush=True)
    pick = np.linspace(0, n - 1, samples).astype(np.int64)
    worst = 0.0
    for i in pick:
        ex = np.zeros(3)
        for b in range(0, n, 1 << 24):
            d = p[b:b + (1 << 24), :3].astype(np.float64) - p[i, :3].astype(np.float64)
            s = (d * d).sum(1) + eps2
            ex += (d * (G * m[b:b + (1 << 24)].astype(np.float64) / (s * np.sqrt(s)))[:, None]).sum(0)
        err = np.linalg.norm(acc[i, :3] - ex) / np.linalg.norm(ex)
        worst = max(worst, err)
        print(f"  body {i}: relative error {err:.2e}", flush=True)
    print(f"worst relative error {worst:.2e}", flush=True)
    sim.step_bh(1); sim.sync()
    sim.set_kernel_timing(True)
    k = 3
    t0 = time.time()
    sim.step_bh(k); sim.sync()
    ts = (time.time() - t0) / k
    wms, _ = sim.kernel_time(bh=True)
    bms, _ = sim.kernel_time(build=True)
    print(f"n={n}: step {ts*1e3:.2f} ms ({1/ts:.2f} steps/s)  walk {wms:.2f} ms  build {bms:.2f} ms", flush=True)
