"""Total-energy drift over 1,000 steps (dt = 0.1, the reference's velocity-Verlet update): the
HIP path must drift no more than the reference arithmetic (the oracle) does on the same bodies."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_energy_drift_1000_steps_no_worse_than_reference(orc):
    import nbody_amd as nb
    n, steps = 2048, 1000
    m, pos, vel = nb.gen_plummer(n, seed=7, equilibrium=True)
    z = np.zeros((n, 4), np.float32)
    k0, u0 = orc.energy(pos, vel, m)
    e0 = k0 + u0
    # reference arithmetic
    rp, rv, ra = orc.step_wgsl_allpairs(orc.LAW_PLUMMER, m, pos, vel, z, steps)
    kr, ur = orc.energy(rp, rv, m)
    drift_ref = abs((kr + ur - e0) / e0)
    # all-pairs on the GPU
    with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
        sim.upload(m, pos, vel)
        sim.step_allpairs(steps)
        kg, ug = sim.energy()
        p, v, a = sim.download()
    drift_gpu = abs((kg + ug - e0) / e0)
    kk, uu = orc.energy(p, v, m)          # the device-side energy sum agrees with the host's
    assert kg == pytest.approx(kk, rel=1e-9) and ug == pytest.approx(uu, rel=1e-9)
    assert drift_ref < 0.2                 # the reference update itself at this dt, in f32 state
    assert drift_gpu <= 1.05 * drift_ref + 1e-6, (drift_gpu, drift_ref)
    # Barnes-Hut on the GPU: the monopole approximation adds its own error; it must stay bounded
    with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
        sim.upload(m, pos, vel)
        sim.step_bh(steps)
        kb, ub = sim.energy()
    drift_bh = abs((kb + ub - e0) / e0)
    assert drift_bh < 2.0 * drift_ref + 0.02, (drift_bh, drift_ref)
    print(f"energy drift over {steps} steps: reference {drift_ref:.3e}, HIP all-pairs {drift_gpu:.3e}, HIP Barnes-Hut {drift_bh:.3e}")


def test_leapfrog_is_the_formula_and_conserves_energy(orc):
    """The opt-in leapfrog integrator (no counterpart in the reference): bit-equal to the plain
    formula given the library's own accelerations, and far less energy drift than the reference
    update over the same 1,000 steps."""
    import nbody_amd as nb
    n, steps, dt = 2048, 1000, np.float32(0.1)
    m, pos, vel = nb.gen_plummer(n, seed=7, equilibrium=True)
    with nb.NBodySim(n, law=nb.LAW_PLUMMER, integrator=nb.INT_LEAPFROG) as sim:
        sim.upload(m, pos, vel)
        # two steps by hand: v += a * (dt/2 then dt); x += v * dt, in f32 without contraction
        p, v = pos.copy(), vel.copy()
        for kick in (np.float32(0.5) * dt, dt):
            a = sim.accel_allpairs()
            v[:, :3] = v[:, :3] + a[:, :3] * kick
            p[:, :3] = p[:, :3] + v[:, :3] * dt
            sim.step_allpairs(1)
            gp, gv, ga = sim.download()
            assert np.array_equal(gp[:, :3], p[:, :3]) and np.array_equal(gv[:, :3], v[:, :3])
            assert np.array_equal(ga[:, :3], a[:, :3])
        sim.step_allpairs(steps - 2)
        gp, gv, _ = sim.download()
        a = sim.accel_allpairs()
    k0, u0 = orc.energy(pos, vel, m)
    v_sync = gv.copy()
    v_sync[:, :3] = gv[:, :3] + a[:, :3] * (np.float32(0.5) * dt)   # bring the half-step velocity onto the step
    k1, u1 = orc.energy(gp, v_sync, m)
    drift_leapfrog = abs((k1 + u1 - k0 - u0) / (k0 + u0))
    z = np.zeros((n, 4), np.float32)
    rp, rv, _ = orc.step_wgsl_allpairs(orc.LAW_PLUMMER, m, pos, vel, z, steps)
    kr, ur = orc.energy(rp, rv, m)
    drift_ref = abs((kr + ur - k0 - u0) / (k0 + u0))
    print(f"energy drift over {steps} steps: reference update {drift_ref:.3e}, leapfrog {drift_leapfrog:.3e}")
    assert drift_leapfrog < 0.1 * drift_ref, (drift_leapfrog, drift_ref)
