"""The benchmark configurations at their full sizes (N = 1M all-pairs and Barnes-Hut, N = 16M
Barnes-Hut), where the scalar oracle would take hours: checked through properties that do not
depend on the size.

* accelerations of sample bodies against the exact f64 sums over all N sources, restated here
  with numpy from src/nbody.wgsl:56-81: 1e-5 relative for all-pairs (the f32 kernel's rounding),
  1e-2 for Barnes-Hut at theta = 0.5 (the truncation error of the reference's own acceptance rule);
* Newton's third law: sum of m a vanishes for the law that is symmetric in the pair;
* the sorted order is a permutation with non-decreasing keys, equal keys in body-index order
  (the insertion order of src/octree_maxdepth.rs:39-48 within a depth-cap cell);
* every exported node aggregates its children: box = union of theirs (exact), range = longest side
  (src/octree_maxdepth.rs:99-101), mass = sum of theirs; the leaves hold every body;
* traversing the sorted order in two runs gives the bits of traversing it whole.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = np.float64(np.float32(0.0066743))
SAMPLES = 12


def exact_accel(i, m, pos, vel=None, chunk=1 << 22):
    """f64 sum over all sources of body i's acceleration: PLUMMER law, or with vel the WGSL law
    with its acos/pow factor (src/nbody.wgsl:56-81).  The self term is zero in both."""
    p = pos[i, :3].astype(np.float64)
    out = np.zeros(3)
    if vel is not None:
        v = vel[i, :3].astype(np.float64)
        vn = np.sqrt((v * v).sum())
    for b in range(0, len(m), chunk):
        d = pos[b:b + chunk, :3].astype(np.float64) - p
        r2 = (d * d).sum(1)
        s = r2 + 1.0
        w = G * m[b:b + chunk].astype(np.float64) / (s * np.sqrt(s))
        if vel is not None:
            w = w * np.arccos((d @ v) / (np.sqrt(r2) * vn + 1.0)) ** 0.15
        out += (d * w[:, None]).sum(0)
    return out


def sample_errors(acc, m, pos, vel=None, seed=0):
    rng = np.random.default_rng(seed)
    picks = rng.choice(len(m), SAMPLES, replace=False)
    errs = []
    for i in picks:
        ex = exact_accel(i, m, pos, vel)
        errs.append(np.linalg.norm(acc[i, :3].astype(np.float64) - ex) / np.linalg.norm(ex))
    return np.array(errs)


@pytest.fixture(scope="module")
def plummer_1m():
    import nbody_amd as nb
    return nb.gen_plummer(1 << 20, seed=42, equilibrium=True)


def test_allpairs_1m_plummer_law(plummer_1m):
    import nbody_amd as nb
    m, pos, vel = plummer_1m
    n = len(m)
    with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
        sim.upload(m, pos, vel)
        acc = sim.accel_allpairs()
        again = sim.accel_allpairs()
    assert np.array_equal(acc, again), "the pair kernel must be reproducible bit for bit"
    assert np.isfinite(acc).all() and not acc[:, 3].any()
    errs = sample_errors(acc, m, pos)
    assert errs.max() < 1e-5, errs
    ma = acc[:, :3].astype(np.float64) * m[:, None].astype(np.float64)
    assert np.abs(ma.sum(0)).max() < 1e-6 * np.abs(ma).sum(0).min()


def test_allpairs_1m_wgsl_law(plummer_1m):
    import nbody_amd as nb
    m, pos, vel = plummer_1m
    n = len(m)
    with nb.NBodySim(n, law=nb.LAW_WGSL_BIAS) as sim:
        sim.upload(m, pos, vel)
        acc = sim.accel_allpairs()
    assert np.isfinite(acc).all()
    errs = sample_errors(acc, m, pos, vel, seed=1)
    assert errs.max() < 1e-5, errs


def check_sorted_order(keys, order, n):
    assert np.array_equal(np.sort(order), np.arange(n, dtype=np.uint32)), "the sorted order must be a permutation"
    ks = keys[order]
    step = np.diff(ks.astype(np.int64))
    assert (step >= 0).all(), "keys must be non-decreasing along the sorted order"
    tie = step == 0
    assert (np.diff(order.astype(np.int64))[tie] > 0).all(), "equal keys must stay in body-index order"


def test_barnes_hut_1m_tree_and_forces(plummer_1m):
    import nbody_amd as nb
    m, pos, vel = plummer_1m
    n = len(m)
    with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
        sim.upload(m, pos, vel)
        sim.build_octree()
        keys, depth, order = sim.morton_keys()
        nodes = sim.download_octree()
        acc = sim.accel_bh()
        sim.bh_forces_ranked(0, n // 2)
        sim.bh_forces_ranked(n // 2, n - n // 2)
        ranked = sim.read_buffer(nb.BUF_ACC_PARTIAL)
    check_sorted_order(keys, order, n)
    assert np.array_equal(ranked[:, :3], acc[order][:, :3])

    typ = nodes["node_type"]
    interior = typ == nb.NODE_INTERIOR
    assert not (typ == nb.NODE_DUMMY).any()
    if not (typ == nb.NODE_LEAFLIST).any():
        assert (typ == nb.NODE_LEAFBODY).sum() == n, "every body is one leaf"
    child = nodes["child"][interior].astype(np.int64)
    has = child != 0
    assert (has.sum(1) >= 1).all()
    lo = np.where(has[..., None], nodes["pos_min"][child], np.inf).min(1)
    hi = np.where(has[..., None], nodes["pos_max"][child], -np.inf).max(1)
    assert np.array_equal(lo, nodes["pos_min"][interior]) and np.array_equal(hi, nodes["pos_max"][interior])
    assert np.array_equal(nodes["range"][interior], (hi - lo).max(1))
    msum = np.where(has, nodes["total_mass"][child].astype(np.float64), 0.0).sum(1)
    assert np.allclose(msum, nodes["total_mass"][interior], rtol=2e-6, atol=0)
    assert np.array_equal(nodes["pos_min"][0], pos[:, :3].min(0)) and np.array_equal(nodes["pos_max"][0], pos[:, :3].max(0))
    assert abs(float(nodes["total_mass"][0]) - m.astype(np.float64).sum()) < 1e-6 * m.astype(np.float64).sum()
    com = (pos[:, :3].astype(np.float64) * m[:, None]).sum(0) / m.astype(np.float64).sum()
    assert np.abs(nodes["com"][0] - com).max() < 5e-4

    errs = sample_errors(acc, m, pos, seed=2)
    assert errs.max() < 1e-2 and np.median(errs) < 3e-3, errs


def test_barnes_hut_16m_sorted_order_and_forces():
    import nbody_amd as nb
    n = 1 << 24
    m, pos, vel = nb.gen_plummer(n, seed=42)
    with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
        sim.upload(m, pos, vel)
        sim.build_octree()
        keys, depth, order = sim.morton_keys()
        acc = sim.accel_bh()
        sim.bh_forces_ranked(0, n // 2)
        sim.bh_forces_ranked(n // 2, n - n // 2)
        ranked = sim.read_buffer(nb.BUF_ACC_PARTIAL)
        nodes_total = sim.counters()["octree_nodes"]
    check_sorted_order(keys, order, n)
    assert n < nodes_total < 2 * n
    assert 1 <= depth.min() and depth.max() <= 16
    assert np.isfinite(acc).all()
    assert np.array_equal(ranked[:, :3], acc[order][:, :3])
    errs = sample_errors(acc, m, pos, seed=3)
    assert errs.max() < 1e-2 and np.median(errs) < 3e-3, errs
