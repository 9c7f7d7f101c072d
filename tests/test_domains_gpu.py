"""Domain-decomposed octree build (nbody_bh_set_domains; no counterpart in the reference, which builds
one tree on one host: src/octree_maxdepth.rs:26-61): every domain's tree from its own bodies, the cells
above depth L rebuilt from the domains' cells.  The merged tree must be the one tree over all bodies:
accelerations bit for bit, the same visits and interactions per traversal, the same number of nodes."""
import numpy as np
import pytest

import nbody_amd as nb

pytestmark = pytest.mark.gpu


def _bodies(kind, n):
    if kind == "plummer":
        m, p, _ = nb.gen_plummer(n, seed=11)
    else:
        m, p = nb.gen_refcube(n, seed=11)
        if kind == "clumps":   # a third of the bodies in three tight clumps, one of them inside a single depth-16 cell
            rng = np.random.default_rng(5)
            k = n // 9
            for c, (centre, width) in enumerate((((60.0, 70.0, 80.0), 2.0), ((190.0, 60.0, 130.0), 0.05),
                                                 ((125.0, 125.0, 125.0), 250.0 / 65536 * 0.4))):
                base = np.array(centre, np.float32)
                if c == 2:
                    base = np.float32(250.0 / 65536) * np.floor(base / np.float32(250.0 / 65536)) + np.float32(250.0 / 65536 * 0.3)
                p[c * k:(c + 1) * k, :3] = base + (rng.random((k, 3)) * width).astype(np.float32)
    v = np.zeros((n, 4), np.float32)
    v[:, :3] = 0.2 * (nb.gen_refcube(n, seed=12)[1][:, :3] / 125.0 - 1.0)
    return m, p, v


def _run(sim, steps):
    sim.build_octree()
    nodes = sim.counters()["octree_nodes"]
    acc = sim.accel_bh()
    sim.set_counting(True)
    sim.accel_bh()
    c = sim.counters()
    sim.set_counting(False)
    sim.step_bh(steps)
    return nodes, acc, (c["bh_node_visits"], c["bh_leaf_interactions"], c["bh_aggregate_interactions"]), sim.download()


@pytest.mark.parametrize("kind,n", [("plummer", 200003), ("refcube", 50000), ("clumps", 90000)])
@pytest.mark.parametrize("domains,top_levels", [(2, 1), (8, 3), (5, 4), (32, 5)])
@pytest.mark.parametrize("law", [nb.LAW_PLUMMER, nb.LAW_WGSL_BIAS])
def test_merged_tree_is_the_one_tree(kind, n, domains, top_levels, law):
    if law == nb.LAW_WGSL_BIAS and (kind != "plummer" or domains != 8):
        pytest.skip("the reference's law on one configuration is enough: the tree does not depend on the law")
    m, p, v = _bodies(kind, n)
    with nb.NBodySim(n, law=law) as sim:
        sim.upload(m, p, v)
        nodes0, acc0, cnt0, state0 = _run(sim, 3)
    with nb.NBodySim(n, law=law) as sim:
        sim.upload(m, p, v)
        sim.bh_set_domains(domains, top_levels)
        nodes1, acc1, cnt1, state1 = _run(sim, 3)
        st = sim.bh_domain_stats()
        with pytest.raises(nb.NBodyError):
            sim.download_octree()
        sim.bh_set_domains(0, 0)               # and back to the one tree
        sim.upload(m, p, v)
        nodes2, acc2, cnt2, state2 = _run(sim, 3)
    assert np.isfinite(acc0).all()
    assert np.array_equal(acc0, acc1) and cnt0 == cnt1
    for a, b in zip(state0, state1):
        assert np.array_equal(a, b)
    assert np.array_equal(acc0, acc2) and cnt0 == cnt2 and nodes0 == nodes2
    assert st["domains"] == domains and st["top_levels"] == top_levels
    assert sum(st["bodies"]) == n


def test_domain_stats_after_a_counting_traversal():
    n, domains = 300000, 8
    m, p, v = _bodies("plummer", n)
    with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
        sim.upload(m, p, v)
        sim.build_octree()
        whole = sim.counters()["octree_nodes"]
    with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
        sim.upload(m, p, v)
        sim.bh_set_domains(domains, 4)
        sim.build_octree()
        sim.set_counting(True)
        sim.accel_bh()
        st = sim.bh_domain_stats()
    assert st["merged_nodes"] == whole                      # node for node the one tree
    assert sum(st["bodies"]) == n and min(st["bodies"]) > 0
    assert max(st["bodies"]) < 2.0 * n / domains            # runs of whole cells: even within a cell's bodies
    assert st["top_nodes"] <= 2 * st["top_entries"] + 64 and st["top_entries"] <= 8 ** 4 + 64
    assert 0 < st["top_nodes_visited"] <= st["top_nodes"]
    for k in range(domains):
        others = sum(st["nodes"]) - st["nodes"][k]
        assert 0 < st["imported_nodes"][k] < others         # a domain needs some of the others' nodes, never all
        assert st["build_ms"][k] > 0


def test_domains_are_refused_where_they_do_not_apply():
    with nb.NBodySim(1000, shard_begin=0, shard_count=500) as sim:
        with pytest.raises(nb.NBodyError):
            sim.bh_set_domains(4, 3)
    with nb.NBodySim(1000, max_depth=3) as sim:
        with pytest.raises(nb.NBodyError):
            sim.bh_set_domains(4, 3)                          # top_levels must stay below max_depth
        sim.bh_set_domains(4, 2)
    with nb.NBodySim(1000) as sim:
        with pytest.raises(nb.NBodyError):
            sim.bh_set_domains(33, 3)
        with pytest.raises(nb.NBodyError):
            sim.bh_domain_stats()                             # nothing set
        with pytest.raises(nb.NBodyError):
            sim.bh_set_domain_view(0)                         # a view needs domains
        sim.bh_set_domains(4, 3)
        with pytest.raises(nb.NBodyError):
            sim.bh_set_domain_view(4)                         # domains are 0..3
        sim.bh_set_domain_view(3)
        sim.bh_set_domain_view(-1)


def test_domains_stay_bit_equal_over_a_long_run():
    """A cold sphere collapses and rebounds: the domains' populations shift, bodies reach the clamp, the
    grouping of the traversal's targets is renewed now and then -- 400 steps with and without domains end
    in the same bits."""
    n = 60000
    m, p, v = nb.gen_plummer(n, seed=5)
    out = []
    for g in (0, 6):
        with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
            sim.upload(m, p, v)
            if g:
                sim.bh_set_domains(g, 3)
            sim.step_bh(400)
            out.append(sim.download())
    for a, b in zip(*out):
        assert np.isfinite(a).all() and np.array_equal(a, b)


@pytest.mark.parametrize("n", [1, 2, 9, 70, 700])
def test_fewer_bodies_than_domains_can_hold(n):
    """Most domains are empty, the top tree has a handful of entries (or one)."""
    m, p, v = _bodies("refcube", n)
    with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
        sim.upload(m, p, v)
        nodes0, acc0, cnt0, state0 = _run(sim, 5)
    with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
        sim.upload(m, p, v)
        sim.bh_set_domains(8, 3)
        sim.build_octree()
        st = sim.bh_domain_stats()
        nodes1, acc1, cnt1, state1 = _run(sim, 5)
    assert np.isfinite(acc0).all() and np.array_equal(acc0, acc1) and cnt0 == cnt1
    for a, b in zip(state0, state1):
        assert np.array_equal(a, b)
    assert sum(st["bodies"]) == n and st["merged_nodes"] == nodes0


@pytest.mark.parametrize("leaflist_mode", [0, 1])
def test_one_domain_holds_everything(leaflist_mode):
    """All bodies inside one depth-3 cell (and a tenth of them inside one depth-16 cell): one domain gets them
    all, outgrows the room a domain is given at first and is given more; the others stay empty."""
    n = 50000
    rng = np.random.default_rng(3)
    m = np.ones(n, np.float32)
    p = np.zeros((n, 4), np.float32)
    p[:, :3] = (40.0 + 20.0 * rng.random((n, 3))).astype(np.float32)       # world 250: depth-3 cells are 31.25 wide
    cell = np.float32(250.0 / 65536)
    p[: n // 10, :3] = (cell * np.floor(np.float32(50.0) / cell) + cell * (0.2 + 0.6 * rng.random((n // 10, 3)))).astype(np.float32)
    p[:, 3] = m
    v = np.zeros((n, 4), np.float32)
    out = []
    for g in (0, 8):
        with nb.NBodySim(n, law=nb.LAW_PLUMMER, leaflist_mode=leaflist_mode) as sim:
            sim.upload(m, p, v)
            if g:
                sim.bh_set_domains(g, 3)
                sim.build_octree()
                st = sim.bh_domain_stats()
            out.append(_run(sim, 2))
    (nodes0, acc0, cnt0, state0), (nodes1, acc1, cnt1, state1) = out
    assert np.isfinite(acc0).all() and np.array_equal(acc0, acc1) and cnt0 == cnt1
    for a, b in zip(state0, state1):
        assert np.array_equal(a, b)
    assert max(st["bodies"]) == n and st["merged_nodes"] == nodes0


def test_a_given_target_grouping_is_kept_with_domains():
    n = 40000
    m, p, v = _bodies("plummer", n)
    order = np.random.default_rng(9).permutation(n).astype(np.uint32)
    acc = []
    for g in (0, 4):
        with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
            sim.upload(m, p, v)
            if g:
                sim.bh_set_domains(g, 2)
            sim.build_octree()
            sim.bh_set_target_order(order)
            acc.append(sim.accel_bh())
    assert np.isfinite(acc[0]).all() and np.array_equal(acc[0], acc[1])


def _domain_bodies(sim, st):
    """Body indices of every domain: the merged sorted order, cut at the domains' populations."""
    _, _, order = sim.morton_keys(with_depth=False)
    cuts = np.concatenate(([0], np.cumsum(st["bodies"]))).astype(np.int64)
    return [order[cuts[k]:cuts[k + 1]] for k in range(st["domains"])]


@pytest.mark.parametrize("kind,n,domains,top_levels,law", [
    ("plummer", 200003, 8, 3, nb.LAW_PLUMMER), ("plummer", 60000, 8, 4, nb.LAW_WGSL_BIAS),
    ("refcube", 50000, 5, 2, nb.LAW_PLUMMER), ("clumps", 90000, 8, 3, nb.LAW_PLUMMER),
    ("clumps", 90000, 3, 1, nb.LAW_WGSL_BIAS), ("refcube", 300, 8, 3, nb.LAW_PLUMMER)])
def test_essential_trees_serve_their_domain(kind, n, domains, top_levels, law):
    """The merged tree as ONE domain's GPU would hold it -- its own tree whole, of the others only what
    they would send (nbody_bh_set_domain_view) -- gives that domain's bodies the one tree's accelerations
    bit for bit, with the same visits and interactions, from fewer nodes."""
    m, p, v = _bodies(kind, n)
    with nb.NBodySim(n, law=law) as sim:
        sim.upload(m, p, v)
        sim.build_octree()
        acc0 = sim.accel_bh()
    assert np.isfinite(acc0).all()
    with nb.NBodySim(n, law=law) as sim:
        sim.upload(m, p, v)
        sim.bh_set_domains(domains, top_levels)
        sim.build_octree()
        whole = sim.bh_domain_stats()
        for d in range(domains):
            sim.bh_set_domain_view(d)
            sim.build_octree()
            st = sim.bh_domain_stats()
            mine = _domain_bodies(sim, st)[d]
            acc = sim.accel_bh()
            assert st["view_domain"] == d and st["bodies"] == whole["bodies"]
            assert np.array_equal(acc[mine], acc0[mine]), d
            assert st["exported_nodes"][d] == 0
            for k in range(domains):
                assert st["exported_nodes"][k] <= st["nodes"][k]
            if n >= 50000:
                assert st["merged_nodes"] < whole["merged_nodes"]
        sim.bh_set_domain_view(-1)
        sim.build_octree()
        assert np.array_equal(sim.accel_bh(), acc0)
