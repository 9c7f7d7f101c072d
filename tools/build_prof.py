import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
m, p, v = nb.gen_plummer(n, seed=42)
with nb.NBodySim(n, law=2) as sim:
    sim.upload(m, p, v)
    for _ in range(3):
        sim.build_octree()
    sim.sync()
