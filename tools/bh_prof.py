"""A few octree builds and traversals for rocprofv3 (kernel trace or one --pmc pass)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
law = int(sys.argv[2]) if len(sys.argv) > 2 else 2
m, p, v = nb.gen_plummer(n, seed=42)
with nb.NBodySim(n, law=law) as sim:
    sim.upload(m, p, v)
    sim.step_bh(3)                     # build + traversal (targets grouped along the Hilbert curve) + integration, three times
    sim.sync()
