"""Barnes-Hut step at N = 1M and 16M: wall time per step over a queued run against the event-timed build and
traversal inside it (what is left is integration, launch gaps and the host's wait for the level sizes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb
for n, steps in ((1 << 20, 200), (1 << 24, 20)):
    m, p, v = nb.gen_plummer(n, seed=42)
    with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
        sim.upload(m, p, v)
        sim.step_bh(3); sim.sync()
        t0 = time.perf_counter(); sim.step_bh(steps); sim.sync()
        wall = (time.perf_counter() - t0) / steps * 1e3
        sim.set_kernel_timing(True); sim.kernel_time(bh=True); sim.kernel_time(build=True)
        sim.step_bh(steps); sim.sync()
        w, _ = sim.kernel_time(bh=True); b, _ = sim.kernel_time(build=True)
    print(f"n={n}: {wall:.3f} ms/step (untimed run); traversal {w:.3f} + build {b:.3f} = {w + b:.3f} ms", flush=True)
