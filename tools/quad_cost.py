import sys, time; sys.path.insert(0, '.')
import nbody_amd as nb
n = 1 << 20
m, p, v = nb.gen_plummer(n, seed=42)
for q in (0, 1):
    with nb.NBodySim(n, law=2, quadrupole=q, leaflist_mode=1) as sim:
        sim.upload(m, p, v); sim.step_bh(3); sim.set_kernel_timing(True); sim.kernel_time(bh=True); sim.kernel_time(build=True)
        sim.sync(); t0 = time.perf_counter(); sim.step_bh(20); sim.sync(); dt = (time.perf_counter() - t0) / 20
        print("quadrupole", q, "step %.3f ms walk %.3f build %.3f" % (dt * 1e3, sim.kernel_time(bh=True)[0], sim.kernel_time(build=True)[0]))
