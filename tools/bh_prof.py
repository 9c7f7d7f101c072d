import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb
n = 1 << 20
law = int(sys.argv[1]) if len(sys.argv) > 1 else 2
m, p, v = nb.gen_plummer(n, seed=42)
with nb.NBodySim(n, law=law) as sim:
    sim.upload(m, p, v)
    sim.build_octree()
    for _ in range(3):
        sim.accel_bh()
