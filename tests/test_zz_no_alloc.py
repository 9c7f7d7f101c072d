"""nbody_accel_allpairs_noalloc (no allocation, fixed device scratch) against nbody_accel_allpairs.

Both run the same kernels with the same slicing, so the accelerations must be bit-equal.  Two
calls in a row, the second with a smaller input: the scratch keeps the first call's contents.
"""
import numpy as np
import pytest

import nbody_amd as nb

pytestmark = pytest.mark.gpu


def _both(n, law, seed):
    m, p = nb.gen_refcube(n, seed=seed)
    rng = np.random.default_rng(seed)
    v = np.zeros((n, 4), np.float32)
    v[:, :3] = rng.normal(size=(n, 3)).astype(np.float32)
    with nb.NBodySim(n, law=law) as sim:
        sim.upload(m, p, v)
        want = sim.accel_allpairs()
        posm, _ = sim.device_buffer(nb.BUF_POSM)
        vel, _ = sim.device_buffer(nb.BUF_VEL)
        out, _ = sim.device_buffer(nb.BUF_ACC_PARTIAL)
        nb.accel_allpairs_noalloc(posm, vel, n, law, sim.cfg.G, sim.cfg.eps2, out, sim.stream())
        sim.sync()
        got = sim.read_buffer(nb.BUF_ACC_PARTIAL)
    return want, got


@pytest.mark.parametrize("law", [nb.LAW_PLUMMER, nb.LAW_WGSL_BIAS, nb.LAW_CPU_R6])
def test_noalloc_matches_the_allocating_entry_point_twice_in_a_row(law):
    for n, seed in ((20000, 5), (5003, 6)):  # the second call is smaller: stale partial sums must not leak in
        want, got = _both(n, law, seed)
        assert np.isfinite(got).all()
        assert np.array_equal(want, got), f"law {law} n {n}: max diff {np.abs(want - got).max()}"


def test_noalloc_rejects_what_does_not_fit():
    with nb.NBodySim(64) as sim:
        posm, _ = sim.device_buffer(nb.BUF_POSM)
        out, _ = sim.device_buffer(nb.BUF_ACC_PARTIAL)
        with pytest.raises(nb.NBodyError) as e:
            nb.accel_allpairs_noalloc(posm, None, 131073, nb.LAW_PLUMMER, 1.0, 1.0, out, sim.stream())
        assert e.value.code == -5 and "131072" in str(e.value)
        with pytest.raises(nb.NBodyError) as e:
            nb.accel_allpairs_noalloc(posm, None, 64, nb.LAW_WGSL_BIAS, 1.0, 1.0, out, sim.stream())
        assert e.value.code == -1
