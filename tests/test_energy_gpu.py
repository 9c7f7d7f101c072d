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
