"""Parity at the sizes the benchmark is quoted on, against the oracle itself rather than samples.

At N = 1,048,576 the oracle evaluates 4,096 targets (four ranges that cover the first and the last
superblock of the pair kernel, a superblock boundary and a source-tile boundary) against all N
sources, in f64 and as the f32 sum in the reference's order.  Tolerance: the task's 1e-5 relative
for f32 with a different summation order.  Measured (profiles/r02_parity.json): the GPU result is
within 3e-7 of the f64 evaluation there, while the reference-order f32 sum itself has drifted 5e-5
from it -- so the distance to f64 is what is held tight (1e-6), and the distance to the f32
reference is held to the reference's own distance from f64 plus 1e-5.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LAWS = {"plummer": 2, "wgsl_bias": 1}


@pytest.fixture(scope="module")
def par():
    from oracle import parity
    return parity


def _acc(n, law, seed=42):
    import nbody_amd as nb
    m, pos, vel = nb.gen_plummer(n, seed=seed, equilibrium=True)
    with nb.NBodySim(n, law=law) as sim:
        sim.upload(m, pos, vel)
        acc = sim.accel_allpairs()
        path = sim.counters()["allpairs_path"]
    return m, pos, vel, acc, path


def _check(st, tight):
    assert st["within_bound"], st
    assert st["vs_f64"]["max"] < tight, st
    # never further from the f32 reference than the reference is from f64, plus the tolerance
    assert st["vs_f32_reference_order"]["max"] < st["f32_reference_order_vs_f64"]["max"] + 1e-5, st


@pytest.mark.parametrize("law", ["plummer", "wgsl_bias"])
def test_allpairs_1m_four_target_ranges_against_the_oracle(par, law):
    n = 1 << 20
    m, pos, vel, acc, path = _acc(n, LAWS[law])
    assert path == (2 if law == "plummer" else 1)
    ranges = par.slices_of(n)
    assert sum(c for _, c in ranges) == 4096 and ranges[0][0] == 0 and ranges[-1][0] + ranges[-1][1] == n
    _check(par.allpairs_stats(LAWS[law], m, pos, vel, acc, ranges), 1e-6)


def test_allpairs_padded_last_superblock_at_scale(par):
    """N = 1,048,576 + 777: the last superblock of the pair kernel is mostly padding bodies."""
    n = (1 << 20) + 777
    m, pos, vel, acc, path = _acc(n, LAWS["plummer"])
    assert path == 2
    ranges = par.slices_of(n) + [(n - 777 - 64, 128)]   # across the start of the partial superblock as well
    _check(par.allpairs_stats(LAWS["plummer"], m, pos, vel, acc, ranges), 1e-6)


def test_allpairs_65536_reference_law_every_body(par):
    """Config 2 with the reference's own law (src/nbody.wgsl:54-81), all 65,536 targets."""
    n = 65536
    m, pos, vel, acc, path = _acc(n, LAWS["wgsl_bias"])
    _check(par.allpairs_stats(LAWS["wgsl_bias"], m, pos, vel, acc, [(0, n)]), 1e-6)


def test_pair_parts_sum_to_the_whole_at_1m():
    """Config 3's arithmetic on one GPU: the eight ranks' nbody_allpairs_partial at N = 1,048,576
    (each on a context with that rank's target range, as in production), summed, against the
    unsharded pair kernel."""
    import nbody_amd as nb
    from nbody_amd.shard import shard_range
    n, world = 1 << 20, 8
    m, pos, vel = nb.gen_plummer(n, seed=42)
    with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
        sim.upload(m, pos, vel)
        whole = sim.accel_allpairs()[:, :3].astype(np.float64)
    total = np.zeros((n, 3))
    for rank in range(world):
        b, c = shard_range(n, rank, world)
        with nb.NBodySim(n, law=nb.LAW_PLUMMER, shard_begin=b, shard_count=c) as sim:
            sim.upload(m, pos, vel)
            sim.allpairs_partial(rank, world)
            total += sim.read_buffer(nb.BUF_ACC_PARTIAL)[:, :3]
    err = np.linalg.norm(total - whole, axis=1) / np.linalg.norm(whole, axis=1)
    assert err.max() < 2e-6, float(err.max())


@pytest.mark.parametrize("n,sample", [(1 << 20, 4096), (1 << 24, 2048)])
@pytest.mark.parametrize("law", ["plummer", "wgsl_bias"])
def test_barnes_hut_traversal_against_the_oracle_on_the_same_tree(law, n, sample):
    """Configs 4 and 5: the GPU-built octree is downloaded and the oracle's per-thread traversal
    (src/nbodybh.wgsl:63-161) walks the SAME tree bytes for a run of targets in the middle of the
    body range (the centre of the sphere, where a body's ~1,600 terms nearly cancel).  Per body:
    within 1e-5 of the f32 reference walk up to that walk's own rounding error against its f64
    evaluation (the bound of tests/test_octree_gpu.py) and an absolute rounding floor of 3e-6 of
    the sample's median acceleration -- f32 sums of terms far larger than their resultant cannot be
    held to 1e-5 of the resultant by either side: the reference walk itself is up to 1.6e-5 from
    its own f64 evaluation on these bodies.  The 99th percentile is held to the plain 1e-5."""
    import nbody_amd as nb
    from oracle import oracle as orc
    m, pos, vel = nb.gen_plummer(n, seed=42, equilibrium=True)
    with nb.NBodySim(n, law=LAWS[law]) as sim:
        sim.upload(m, pos, vel)
        sim.build_octree()
        nodes = sim.download_octree().view(orc.NODE_DTYPE)
        acc = sim.accel_bh()
    b = (n // 2) // 64 * 64
    r32, _ = orc.bh_accel(LAWS[law], pos, vel, nodes, i_begin=b, i_count=sample)
    r64 = orc.bh_accel_f64(LAWS[law], pos, vel, nodes, i_begin=b, i_count=sample)
    r64 = r64[b:b + sample] if len(r64) == n else r64
    a = acc[b:b + sample, :3].astype(np.float64)
    r32 = r32[b:b + sample, :3].astype(np.float64)
    scale = np.linalg.norm(r64, axis=1)
    e_gpu = np.linalg.norm(a - r64, axis=1)
    e_ref = np.linalg.norm(r32 - r64, axis=1)
    bound = 1e-5 * scale + e_ref + 3e-6 * np.median(scale)
    bad = e_gpu > bound
    assert not bad.any(), (int(bad.sum()), float((e_gpu / scale).max()))
    e32 = np.linalg.norm(a - r32, axis=1)
    assert not (e32 > bound + e_ref).any(), float((e32 / scale).max())
    assert np.quantile(e32 / np.linalg.norm(r32, axis=1), 0.99) < 1e-5
