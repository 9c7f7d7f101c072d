"""Times the Barnes-Hut step pieces (build, walk) and reports traversal statistics."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nb

def run(n, law, steps=3, dataset="plummer"):
    if dataset == "plummer":
        m, p, v = nb.gen_plummer(n, seed=42)
    else:
        m, p = nb.gen_refcube(n, seed=42); v = None
    with nb.NBodySim(n, law=law) as sim:
        sim.upload(m, p, v)
        sim.build_octree(); sim.sync()
        t0 = time.time()
        for _ in range(steps):
            sim.build_octree()
        sim.sync()
        tb = (time.time() - t0) / steps
        sim.set_counting(True)
        sim.accel_bh()
        c = sim.counters()
        sim.set_counting(False)
        sim.set_kernel_timing(True)
        t0 = time.time()
        sim.step_bh(steps); sim.sync()
        ts = (time.time() - t0) / steps
        kms, cnt = sim.kernel_time(bh=True)
    inter = c["bh_leaf_interactions"] + c["bh_aggregate_interactions"]
    print(f"n={n} law={law} {dataset}: nodes={c['octree_nodes']} build {tb*1e3:.3f} ms  walk kernel {kms:.3f} ms  step {ts*1e3:.3f} ms "
          f"({1/ts:.2f} steps/s)  visits/body {c['bh_node_visits']/n:.1f}  inter/body {inter/n:.1f}  wave visits {c['bh_wave_node_visits']} "
          f"({c['bh_wave_node_visits']*64/max(1,c['bh_node_visits']):.2f}x lane visits)  {inter/(kms*1e-3)/1e9:.2f} Ginter/s  max_stack {c['bh_max_stack']}", flush=True)

if __name__ == "__main__":
    for n in (65536, 1 << 20):
        for law in (2, 1):
            run(n, law)
    run(1 << 20, 2, dataset="refcube")
    if len(sys.argv) > 1:
        run(1 << 24, 2, steps=2)
