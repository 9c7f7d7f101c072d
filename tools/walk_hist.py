"""Wavefront node visits of the traversal by the number of lanes each is made for (counting build;
NBODY_WALK_HIST=1 prints the 16 bins of 4 lanes to stderr).  usage: NBODY_WALK_HIST=1 python tools/walk_hist.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
m, p, v = nb.gen_plummer(n, seed=42)
with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
    sim.upload(m, p, v)
    sim.build_octree()
    sim.set_counting(True)
    sim.accel_bh()
    print(sim.counters())
