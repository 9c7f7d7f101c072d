"""Free device memory before and after many create / step / destroy cycles."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb
hip = C.CDLL("libamdhip64.so")
def free_bytes():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value
n = 70000
m, p, v = nb.gen_plummer(n, seed=1)
def cycle(**cfg):
    with nb.NBodySim(n, **cfg) as sim:
        sim.upload(m, p, v)
        sim.step_allpairs(1)
        sharded = "shard_count" in cfg
        sim.step_bh(1)
        sim.download_octree()
        if not sharded:
            sim.step_bh(2)
            sim.bh_forces_ranked(0, n); sim.integrate_ranked()
        sim.download()
cycle(law=2)
f0 = free_bytes()
for rounds in (10, 40, 80):
    for i in range(rounds):
        cycle(law=2, quadrupole=i % 2, leaflist_mode=1)
        cycle(law=1)
        cycle(law=2, shard_begin=100, shard_count=30000)
    f1 = free_bytes()
    print("after %d more rounds: free %d, %d bytes less than at the start" % (rounds, f1, f0 - f1), flush=True)
assert abs(f0 - f1) < 64 << 20
