"""Barnes-Hut traversal and build timing at the given sizes (NBODY_HIP_LIB selects a measurement build)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nb

def run(n, law, dataset="plummer", reps=5):
    if dataset == "plummer":
        m, p, v = nb.gen_plummer(n, seed=42)
    else:
        m, p = nb.gen_refcube(n, seed=42); v = None
    with nb.NBodySim(n, law=law) as sim:
        sim.upload(m, p, v)
        sim.build_octree(); sim.sync()
        a = sim.accel_bh()
        sim.set_kernel_timing(True)
        for _ in range(reps):
            sim.bh_forces_ranked(0, n)
        sim.sync()
        kms, cnt = sim.kernel_time(bh=True)
        bms, _ = sim.kernel_time(build=True)
    print(f"n={n} law={law} {dataset}: walk {kms:.3f} ms build {bms:.3f} ms  acc checksum {np.abs(a[:, :3]).sum(dtype=np.float64):.9e} "
          f"nan={int(np.isnan(a).sum())}", flush=True)

if __name__ == "__main__":
    sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [65536, 1 << 20]
    for n in sizes:
        run(n, 2)
    run(sizes[-1], 1)
    run(1 << 20, 2, dataset="refcube")
