"""A few all-pairs steps with the ordered-pair kernels (the pair kernel switched off), for one rocprofv3 --pmc pass:
PLUMMER at N = 262,144 and the reference's law at N = 65,536."""
import os, sys
os.environ["NBODY_AP_SYM"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb
for n, law in ((262144, nb.LAW_PLUMMER), (65536, nb.LAW_WGSL_BIAS)):
    m, p, v = nb.gen_plummer(n, seed=42)
    with nb.NBodySim(n, law=law) as sim:
        sim.upload(m, p, v)
        sim.step_allpairs(3)
        sim.sync()
