"""Octree build and traversal times for a pile of 100,000 bodies in ONE depth-cap cell (and for as many
coincident bodies in the clamp corner), in both leaf-list modes: what the bounds of
tests/test_robust_gpu.py are set from.  usage: python tools/pile_time.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb

n_pile, n_rest = 100000, 3000
n = n_pile + n_rest
for coincident in (False, True):
    m, p = nb.gen_refcube(n, seed=9)
    cell = 250.0 / 65536
    if coincident:
        p[:n_pile, :3] = 250.0
    else:
        rng = np.random.default_rng(1)
        base = np.float32(cell) * np.array([20000, 31000, 40000], np.float32)
        p[:n_pile, :3] = base + (rng.random((n_pile, 3)) * 0.9 * cell + 0.05 * cell).astype(np.float32)
    for mode in (0, 1):
        with nb.NBodySim(n, law=nb.LAW_PLUMMER, leaflist_mode=mode) as sim:
            sim.upload(m, p)
            sim.build_octree(); sim.accel_bh()
            sim.set_kernel_timing(True)
            sim.kernel_time(bh=True); sim.kernel_time(build=True)
            for _ in range(3):
                sim.build_octree()
                sim.accel_bh()
            w, _ = sim.kernel_time(bh=True)
            b, _ = sim.kernel_time(build=True)
        print(json.dumps({"coincident": coincident, "leaflist_mode": mode, "build_ms": b, "walk_ms": w}), flush=True)
