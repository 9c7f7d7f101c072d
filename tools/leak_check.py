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
def domain_cycle():
    """Domains switched on, viewed from two of them, regrown by a pile, switched off, all within one context."""
    with nb.NBodySim(n, law=2) as sim:
        sim.upload(m, p, v)
        for g, levels in ((8, 3), (5, 2), (0, 0), (32, 4)):
            sim.bh_set_domains(g, levels)
            sim.step_bh(1)
            if g:
                sim.bh_set_domain_view(g - 1)
                sim.step_bh(1)
                sim.bh_set_domain_view(0)
                sim.build_octree()
                sim.bh_domain_stats()
                sim.bh_set_domain_view(-1)
        q = p.copy()
        q[:, :3] = 60.0 + 0.001 * (q[:, :3] - 125.0)      # everybody into one depth-3 cell: one domain outgrows its room
        sim.upload(m, q, v)
        sim.step_bh(1)
cycle(law=2)
domain_cycle()
f0 = free_bytes()
for rounds in (10, 40, 80):
    for i in range(rounds):
        cycle(law=2, quadrupole=i % 2, leaflist_mode=1)
        cycle(law=1)
        cycle(law=2, shard_begin=100, shard_count=30000)
        if i % 4 == 0:
            domain_cycle()
    f1 = free_bytes()
    print("after %d more rounds: free %d, %d bytes less than at the start" % (rounds, f1, f0 - f1), flush=True)
assert abs(f0 - f1) < 64 << 20
