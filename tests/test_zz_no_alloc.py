"""nbody_accel_allpairs_noalloc (no allocation, fixed device scratch) against nbody_accel_allpairs.

Up to N = 16,384 both run the same kernels with the same slicing, so the accelerations must be
bit-equal.  Two calls in a row, the second with a smaller input: the scratch keeps the first call's
contents.  Above that the fixed 16 MiB scratch caps the number of source slices (N = 100,000: 10
instead of 42), and for the PLUMMER law from N = 65,536 the allocating entry point runs the pair
kernel: then the two agree to rounding, 2e-6 relative per body.
"""
import numpy as np
import pytest

import nbody_amd as nb

pytestmark = pytest.mark.gpu


def _both(n, law, seed, plummer=False):
    m, p = nb.gen_plummer(n, seed=seed)[:2] if plummer else nb.gen_refcube(n, seed=seed)
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


@pytest.mark.parametrize("law,n", [(nb.LAW_WGSL_BIAS, 100000), (nb.LAW_PLUMMER, 70000), (nb.LAW_PLUMMER, 131072)])
def test_noalloc_where_the_slicing_differs_agrees_to_rounding(law, n):
    """The scratch caps the slices (WGSL_BIAS, N = 100,000), or the allocating side is the pair kernel
    (PLUMMER from N = 65,536; N = 131,072 is the largest input the entry point takes)."""
    want, got = _both(n, law, 7, plummer=True)
    assert np.isfinite(got).all()
    scale = np.linalg.norm(want[:, :3].astype(np.float64), axis=1)
    err = np.linalg.norm(got[:, :3].astype(np.float64) - want[:, :3], axis=1)
    assert (err <= 2e-6 * scale).all(), float((err / scale).max())
    want2, got2 = _both(5003, law, 8)            # and a smaller call right after it, bit-equal again
    assert np.array_equal(want2, got2)


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
