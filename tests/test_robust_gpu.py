"""Degenerate inputs and call orders: piles of bodies in one depth-cap cell, coincident bodies in
the clamp corner, and the exchange buffer's ownership between a force call and its integration."""
import time

import numpy as np
import pytest

import nbody_amd as nb

pytestmark = pytest.mark.gpu


def _pile(n_pile, n_rest, coincident):
    """n_pile bodies inside ONE depth-16 cell (or all at the clamp corner) plus a uniform rest."""
    m, p = nb.gen_refcube(n_pile + n_rest, seed=9)
    cell = 250.0 / 65536
    if coincident:
        p[:n_pile, :3] = 250.0  # where src/nbodybh.wgsl:167 piles escapers up
    else:
        rng = np.random.default_rng(1)
        base = np.float32(cell) * np.array([20000, 31000, 40000], np.float32)
        p[:n_pile, :3] = base + (rng.random((n_pile, 3)) * 0.9 * cell + 0.05 * cell).astype(np.float32)
    return m, p


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("coincident", [False, True])
def test_pile_in_one_cell_builds_and_walks_in_bounded_time(orc, mode, coincident):
    n_pile, n_rest = 100000, 3000
    n = n_pile + n_rest
    m, p = _pile(n_pile, n_rest, coincident)
    # counts of a few targets inside the pile and outside it, against the oracle's traversal of the same tree bytes
    targets = [(0, 64), (n_pile - 64, 128), (n - 64, 64)]
    with nb.NBodySim(n, law=nb.LAW_PLUMMER, leaflist_mode=mode) as sim:
        sim.upload(m, p)
        sim.build_octree(); sim.accel_bh()
        sim.set_kernel_timing(True)
        sim.kernel_time(build=True); sim.kernel_time(bh=True)
        sim.build_octree()
        nodes = sim.download_octree()
        keys, depth, _ = sim.morton_keys()
        acc = sim.accel_bh()
        t_build, t_walk = sim.kernel_time(build=True)[0], sim.kernel_time(bh=True)[0]
        sim.set_kernel_timing(False)
        sim.set_counting(True)
        sim.accel_bh()
        c_all = sim.counters()
    # Measured (tools/pile_time.py): build 14.4 ms -- the cell's aggregate is a chain of 100,000 dependent f32
    # roundings, ~0.14 us each with the cell's bodies loaded 64 at a time (50 ms with one load per step) --
    # and traversal 10.8 ms (every body of the pile meets every member of the list); three times that.
    assert t_build < 45.0 and t_walk < 35.0, (t_build, t_walk)
    assert np.isfinite(acc[n_pile:]).all()
    assert c_all["bh_max_stack"] <= 8 * 18
    # the pile is one LEAFLIST node with the reference's member count (literal: k - 1 prefix aggregates)
    lists = nodes[nodes["node_type"] == nb.NODE_LEAFLIST]
    assert lists["child"][:, 1].max() == (n_pile if mode else n_pile - 1)
    assert len(np.unique(keys[:n_pile])) == 1 and (depth[:n_pile] == 16).all()
    ref_nodes, _, _, _ = orc.build_octree(p, m, leaflist_mode=mode)
    assert len(ref_nodes) == len(nodes)
    onodes = nodes.view(orc.NODE_DTYPE)
    for b, cnt in targets:
        ref, rc = orc.bh_accel(orc.LAW_PLUMMER, p, np.zeros_like(p), onodes, i_begin=b, i_count=cnt)
        with nb.NBodySim(n, law=nb.LAW_PLUMMER, leaflist_mode=mode, shard_begin=b, shard_count=cnt) as sim:
            sim.upload(m, p)
            sim.build_octree()
            sim.set_counting(True)
            got = sim.accel_bh()
            c = sim.counters()
        assert (c["bh_node_visits"], c["bh_leaf_interactions"], c["bh_aggregate_interactions"]) == \
            (rc["visits"], rc["leaf"], rc["aggregate"]), (b, cnt)
        r, g = ref[b:b + cnt, :3], got[b:b + cnt, :3]
        ok = np.isfinite(r).all(axis=1)  # coincident bodies with eps2 > 0 stay finite; keep the mask anyway
        scale = np.linalg.norm(r[ok], axis=1)
        assert (np.linalg.norm(g[ok] - r[ok], axis=1) <= 2e-4 * scale + 1e-7).all()


def test_exchange_buffer_is_not_touched_by_transfers_and_stale_forces_are_refused():
    n = 70000
    m, p, v = nb.gen_plummer(n, seed=4)
    with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
        sim.upload(m, p, v)
        with pytest.raises(nb.NBodyError) as e:
            sim.integrate_partial()  # nothing to integrate yet
        assert e.value.code == -4
        sim.allpairs_partial(0, 1)
        before = sim.read_buffer(nb.BUF_ACC_PARTIAL)
        sim.download()            # positions pass through a buffer of their own
        sim.upload_alt(p, v)
        assert np.array_equal(before, sim.read_buffer(nb.BUF_ACC_PARTIAL))
        sim.integrate_partial()
        with pytest.raises(nb.NBodyError):
            sim.integrate_partial()  # consumed
        sim.allpairs_partial(0, 1)
        sim.accel_allpairs()      # overwrites the exchange buffer
        with pytest.raises(nb.NBodyError) as e:
            sim.integrate_partial()
        assert e.value.code == -4
    with nb.NBodySim(5000, law=nb.LAW_PLUMMER) as sim:
        m, p = nb.gen_refcube(5000)
        sim.upload(m, p)
        with pytest.raises(nb.NBodyError) as e:
            sim.integrate_ranked()
        assert e.value.code == -4
        sim.bh_forces_ranked(0, 5000)
        sim.integrate_ranked()


@pytest.mark.parametrize("law", [nb.LAW_PLUMMER, nb.LAW_WGSL_BIAS])
def test_a_nan_position_ends_in_nan_accelerations_not_in_an_endless_traversal(orc, law):
    """A body with a NaN coordinate makes every distance to its cells NaN: no cell is accepted, all are
    opened down to the body's own leaf, whose term is NaN -- for every target, as in the reference's
    traversal (src/nbodybh.wgsl:76-102 interacts with a LEAFBODY whatever the distance).  The
    traversal must end (a leaf record is never opened, an empty group never pushed) and agree with
    the oracle's on the same tree bytes about which results are NaN."""
    n = 3000
    m, p = nb.gen_refcube(n, seed=4)
    p[1234, 1] = np.nan
    v = np.zeros((n, 4), np.float32)
    with nb.NBodySim(n, law=law) as sim:
        sim.upload(m, p, v)
        sim.build_octree()
        nodes = sim.download_octree()
        sim.set_counting(True)
        acc = sim.accel_bh()
        c = sim.counters()
        sim.step_bh(2)          # and a step on top of it returns as well
        sim.sync()
    ref, rc = orc.bh_accel(law, p, v, np.ascontiguousarray(nodes.view(orc.NODE_DTYPE)))
    assert np.array_equal(np.isnan(acc[:, :3]), np.isnan(ref[:, :3]))
    assert np.isnan(acc[:, :3]).any()
    assert c["bh_node_visits"] == rc["visits"]


def test_an_uploaded_interior_node_without_children_is_refused(orc):
    n = 500
    m, p = nb.gen_refcube(n, seed=6)
    nodes, _, _, _ = orc.build_octree(p, m)
    bad = nodes.copy()
    k = int(np.flatnonzero(bad["node_type"] == nb.NODE_INTERIOR)[-1])
    bad["child"][k, :] = 0
    with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
        sim.upload(m, p)
        with pytest.raises(nb.NBodyError) as e:
            sim.upload_octree(bad.view(nb.NODE_DTYPE))
        assert e.value.code == -1 and "no children" in str(e.value)
        sim.upload_octree(nodes.view(nb.NODE_DTYPE))   # the intact tree still goes in
        assert np.isfinite(sim.accel_bh()).all()


def test_bodies_piled_up_by_the_clamp_are_walked_like_the_reference(orc):
    """A run long enough for the Barnes-Hut clamp (src/nbodybh.wgsl:167) to pile escapers on the faces
    and in a corner of the world cube: bodies with equal coordinates share depth-cap cells and end in
    leaf lists.  On the tree bytes of that state the counting traversal must make the oracle's visits and
    interactions, for targets inside the pile, on a face and in the bulk."""
    n = 12000
    m, p = nb.gen_refcube(n, seed=21)
    v = np.zeros((n, 4), np.float32)
    rng = np.random.default_rng(3)
    v[:3000, :3] = 120.0 + 30.0 * rng.random((3000, 3)).astype(np.float32)      # towards the corner (250, 250, 250)
    v[3000:5000, 0] = -150.0                                                     # towards the face x = 0
    with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
        sim.upload(m, p, v)
        sim.step_bh(25)
        pos, vel, _ = sim.download()
        assert np.isfinite(pos).all()
        corner = np.flatnonzero((pos[:, :3] == 250.0).all(axis=1))
        face = np.flatnonzero((pos[:, 0] == 0.0) & ~(pos[:, 1:3] == 250.0).any(axis=1))
        assert len(corner) >= 2900 and len(face) >= 1900, (len(corner), len(face))
        sim.build_octree()
        nodes = sim.download_octree()
        lists = nodes[nodes["node_type"] == nb.NODE_LEAFLIST]
        assert lists["child"][:, 1].max() >= len(corner) - 1       # the corner pile is ONE leaf list (literal mode: k - 1 members)
        acc = sim.accel_bh()
    onodes = np.ascontiguousarray(nodes.view(orc.NODE_DTYPE))
    bulk = np.setdiff1d(np.arange(n), np.concatenate([corner, face]))
    for what, idx in (("corner pile", corner[:96]), ("face", face[:96]), ("bulk", bulk[:96])):
        tp, tv = np.ascontiguousarray(pos[idx]), np.ascontiguousarray(vel[idx])
        ref, rc = orc.bh_accel(orc.LAW_PLUMMER, tp, tv, onodes)
        got = np.zeros(3, np.int64)
        for b in idx:
            with nb.NBodySim(n, law=nb.LAW_PLUMMER, shard_begin=int(b), shard_count=1) as one:
                one.upload(m, pos, vel)
                one.build_octree()
                one.set_counting(True)
                one.accel_bh()
                c = one.counters()
            got += (c["bh_node_visits"], c["bh_leaf_interactions"], c["bh_aggregate_interactions"])
            if got[0] > 0 and b == idx[7]:
                break                      # eight single-target contexts per set are enough for the counts
        _, rc8 = orc.bh_accel(orc.LAW_PLUMMER, tp[:8], tv[:8], onodes)
        assert tuple(got) == (rc8["visits"], rc8["leaf"], rc8["aggregate"]), what
        r = ref[:, :3].astype(np.float64)
        g = acc[idx, :3].astype(np.float64)
        assert (np.linalg.norm(g - r, axis=1) <= 2e-4 * np.linalg.norm(r, axis=1) + 1e-9).all(), what
