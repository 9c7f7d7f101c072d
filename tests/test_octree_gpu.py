"""Barnes-Hut parity: GPU octree build vs the insertion-order oracle (src/octree_maxdepth.rs,
src/octree.rs), and the per-wavefront traversal vs the per-thread walk of src/nbodybh.wgsl on
identical tree bytes.

Bit-exact: Morton keys, body -> leaf depth, the set of cells, node types, bounds, range, leaf-list
members (literal mode).  Mass / centre of mass of interior nodes: rounding only (the reference
accumulates in insertion order, the GPU bottom-up).  Accelerations: 1e-5 relative per body."""
import numpy as np
import pytest

from conftest import body_rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-5


def check_walk(orc, law, pos, vel, nodes, acc_gpu, what):
    """1e-5 relative per body.  The f32 reference walk is itself only known to its own rounding
    error e_ref against the float64 evaluation of the same node sequence (large for
    cancellation-dominated bodies), so the bound on the distance to float64 is 1e-5*|a| + e_ref:
    everything within 1e-5 of the reference satisfies it, by the triangle inequality.  The GPU adds
    the same terms in another order (a node's children together; subtrees that few lanes open,
    last), whose own f32 rounding is priced as in the all-pairs tests: 4 eps x the sum of the terms'
    magnitudes."""
    ref32, _ = orc.bh_accel(law, pos, vel, nodes)
    ref64, mag = orc.bh_accel_f64(law, pos, vel, nodes, with_abs=True)
    e_gpu = np.linalg.norm(acc_gpu[:, :3].astype(np.float64) - ref64, axis=1)
    e_ref = np.linalg.norm(ref32[:, :3].astype(np.float64) - ref64, axis=1)
    scale = np.linalg.norm(ref64, axis=1)
    bad = e_gpu > TOL * scale + e_ref + 4 * np.finfo(np.float32).eps * mag + 1e-30
    assert not bad.any(), f"{what}: {bad.sum()} bodies off, worst {np.max(e_gpu / scale):.3e}"
    assert np.quantile(body_rel_err(acc_gpu, ref32), 0.99) < TOL, what


def p4(rows):
    a = np.zeros((len(rows), 4), np.float32)
    a[:, :3] = np.asarray(rows, np.float32)
    return a


def canon(nodes, T_LEAFLIST=2, T_INTERIOR=3):
    """{path: node} with path = tuple of octants from the root; list members get (..., 'L', j)."""
    out = {}
    stack = [(0, ())]
    while stack:
        idx, path = stack.pop()
        nd = nodes[idx]
        out[path] = nd
        if nd["node_type"] == T_INTERIOR:
            for k in range(8):
                c = int(nd["child"][k])
                if c:
                    stack.append((c, path + (k,)))
        elif nd["node_type"] == T_LEAFLIST:
            for j in range(int(nd["child"][1])):
                out[path + ("L", j)] = nodes[int(nd["child"][0]) + j]
    return out


def exact_aggregates(pos, m, keys, depth, D):
    """{path: (mass, com, count)} in float64 for every cell on the way to each body's leaf."""
    agg = {}
    p64 = pos[:, :3].astype(np.float64)
    m64 = m.astype(np.float64)
    for b in range(len(m)):
        k = int(keys[b])
        digits = tuple((k >> (3 * (D - 1 - l))) & 7 for l in range(D))
        for d in range(int(depth[b]) + 1):
            path = digits[:d]
            a = agg.get(path)
            if a is None:
                agg[path] = [m64[b], m64[b] * p64[b], 1]
            else:
                a[0] += m64[b]
                a[1] = a[1] + m64[b] * p64[b]
                a[2] += 1
    return {k: (v[0], v[1] / v[0], v[2]) for k, v in agg.items()}


def compare_trees(gpu_nodes, ref_nodes, pos, m, keys, depth, D=16):
    """Same cells, types, bounds, range and list members as the insertion-order build.  Interior
    mass / centre of mass: the reference's running f32 sums drift with the number of inserts
    (6e-5 relative for 20,000 equal masses), so both builds are held to the float64 aggregates --
    the GPU's to rounding, the reference's to its drift."""
    g, r = canon(gpu_nodes), canon(ref_nodes)
    exact = exact_aggregates(pos, m, keys, depth, D)
    assert len(gpu_nodes) == len(ref_nodes)
    assert set(g) == set(r), "cell sets differ"
    for path, rn in r.items():
        gn = g[path]
        assert gn["node_type"] == rn["node_type"], path
        is_member = "L" in path
        if rn["node_type"] in (1, 2):
            # bodies, list members and list aggregates follow the reference arithmetic exactly
            assert np.array_equal(gn["com"], rn["com"]) and gn["total_mass"] == rn["total_mass"], path
        else:
            M, com, cnt = exact[path]
            assert abs(gn["total_mass"] - M) <= 1e-6 * M, path
            assert np.abs(gn["com"] - com).max() <= 1e-6 * 250, path
            drift = 1e-6 + 6e-8 * cnt
            assert abs(rn["total_mass"] - M) <= drift * M, path
            assert np.abs(rn["com"] - com).max() <= drift * 250, path
        assert np.array_equal(gn["pos_min"], rn["pos_min"]) and np.array_equal(gn["pos_max"], rn["pos_max"]), path
        assert gn["range"] == rn["range"], path
        if not is_member:
            assert gn["depth"] == len(path)
        if rn["node_type"] == 2:
            assert gn["child"][1] == rn["child"][1]


DATASETS = ["refcube250", "refcube4096", "plummer20000", "edges", "clustered"]


def dataset(orc, name):
    if name.startswith("refcube"):
        m, pos = orc.gen_refcube(int(name[7:]), seed=3)
    elif name.startswith("plummer"):
        m, pos = orc.gen_plummer(int(name[7:]), seed=42)
    elif name == "edges":
        # positions on cell boundaries, the world faces, the centre planes and outside the cube
        vals = [0.0, 250.0, 125.0, 62.5, 187.5, 124.99999, 125.00001, 250.0 / 65536 * 7, -3.0, 260.0, 31.25, 1e-30]
        rows = [(x, y, z) for x in vals for y in vals[:6] for z in (vals[2], vals[7], vals[8])]
        pos = p4(rows)
        m = np.linspace(0.5, 8, len(rows)).astype(np.float32)
    else:  # tight clumps that force deep chains and depth-16 leaf lists
        rng = np.random.default_rng(7)
        cell = 250.0 / 65536
        centres = rng.uniform(20, 230, (40, 3))
        rows = []
        for c in centres:
            k = rng.integers(2, 6)
            rows += [c + rng.uniform(0, 0.3 * cell, 3) for _ in range(k)]       # same depth-16 cell (mostly)
            rows += [c + rng.uniform(0, 40 * cell, 3) for _ in range(3)]        # deep but separable
        rows += list(rng.uniform(0, 250, (300, 3)))
        pos = p4(rows)
        m = rng.uniform(0.25, 64, len(rows)).astype(np.float32)
    return m, pos


@pytest.mark.parametrize("name", DATASETS)
def test_keys_and_leaf_assignment_bit_exact(orc, name):
    import nbody_amd as nb
    m, pos = dataset(orc, name)
    n = len(m)
    with nb.NBodySim(n) as sim:
        sim.upload(m, pos)
        sim.build_octree()
        keys, depth, order = sim.morton_keys()
    assert np.array_equal(keys, orc.morton_keys(pos))
    ref_nodes, leaf, dep, _ = orc.build_octree(pos, m)
    assert np.array_equal(depth, dep)
    # sorted order: keys ascending, equal keys in body-index order (stable)
    ks = keys[order]
    assert np.all(ks[1:] >= ks[:-1])
    same = ks[1:] == ks[:-1]
    assert np.all(order[1:][same] > order[:-1][same])
    assert sorted(order) == list(range(n))
    # the dormant unbounded builder (src/octree.rs) assigns the same leaves whenever it stays
    # within 16 levels
    try:
        _, leaf1, dep1, md = orc.build_octree(pos, m, variant=1)
    except RuntimeError:
        return  # coincident-cell bodies: src/octree.rs does not terminate / exceeds the cap
    if dep1.max() <= 16 and not (dep == 16).any():
        assert np.array_equal(depth, dep1)
        assert md == dep1.max() + 1 or n == 1


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("name", DATASETS)
def test_tree_structure_matches_insertion_order_build(orc, name, mode):
    import nbody_amd as nb
    m, pos = dataset(orc, name)
    with nb.NBodySim(len(m), leaflist_mode=mode) as sim:
        sim.upload(m, pos)
        sim.build_octree()
        nodes = sim.download_octree()
        keys, depth, _ = sim.morton_keys()
        assert sim.counters()["octree_nodes"] == len(nodes)
    ref_nodes, _, _, _ = orc.build_octree(pos, m, leaflist_mode=mode)
    compare_trees(nodes, ref_nodes, pos, m, keys, depth)


@pytest.mark.parametrize("max_depth", [1, 3, 8])
def test_shallow_depth_caps_make_leaf_lists(orc, max_depth):
    import nbody_amd as nb
    m, pos = orc.gen_refcube(600, seed=9)
    for mode in (0, 1):
        with nb.NBodySim(600, max_depth=max_depth, leaflist_mode=mode) as sim:
            sim.upload(m, pos)
            sim.build_octree()
            nodes = sim.download_octree()
            keys, depth, _ = sim.morton_keys()
        ref_nodes, _, dep, _ = orc.build_octree(pos, m, max_depth=max_depth, leaflist_mode=mode)
        compare_trees(nodes, ref_nodes, pos, m, keys, depth, D=max_depth)
        assert np.array_equal(depth, dep)
        assert np.array_equal(keys, orc.morton_keys(pos, levels=max_depth))


@pytest.mark.parametrize("world", [64.0, 1000.0, 3.0])
def test_other_world_sizes(orc, world):
    """The root cell is configuration (src/octree_maxdepth.rs:28-29 has 250): keys, leaves and
    the tree for other extents, including one that is not a power of two times anything."""
    import nbody_amd as nb
    m, pos = orc.gen_refcube(3000, seed=21)
    pos = pos.copy()
    pos[:, :3] = (pos[:, :3] * np.float32(world / 250.0)).astype(np.float32)
    with nb.NBodySim(len(m), world=world) as sim:
        sim.upload(m, pos)
        sim.build_octree()
        keys, depth, _ = sim.morton_keys()
        nodes = sim.download_octree()
    assert np.array_equal(keys, orc.morton_keys(pos, world=world))
    ref_nodes, _, dep, _ = orc.build_octree(pos, m, world=world)
    assert np.array_equal(depth, dep)
    compare_trees(nodes, ref_nodes, pos, m, keys, depth)


def test_bodies_outside_the_root_cell(orc):
    """The all-pairs step does not clamp (src/nbody.wgsl:132-138), so a tree may be asked for
    bodies outside [0, world]^3: the reference's descent only compares with cell centres, and
    such bodies end up in the outermost cells; same keys, same tree."""
    import nbody_amd as nb
    m, pos = orc.gen_refcube(2000, seed=33)
    pos = pos.copy()
    pos[:, :3] = (pos[:, :3] - 125.0) * np.float32(2.4) + 125.0        # [-55, 305]
    assert (pos[:, :3] < 0).any() and (pos[:, :3] > 250).any()
    with nb.NBodySim(len(m)) as sim:
        sim.upload(m, pos)
        sim.build_octree()
        keys, depth, _ = sim.morton_keys()
        nodes = sim.download_octree()
        acc = sim.accel_bh()
    assert np.array_equal(keys, orc.morton_keys(pos))
    ref_nodes, _, dep, _ = orc.build_octree(pos, m)
    assert np.array_equal(depth, dep)
    compare_trees(nodes, ref_nodes, pos, m, keys, depth)
    check_walk(orc, 1, pos, np.zeros_like(pos), nodes.view(orc.NODE_DTYPE), acc, "bodies outside the root cell")


def test_tiny_trees(orc):
    import nbody_amd as nb
    for rows in ([[10, 10, 10]], [[10, 10, 10], [200, 200, 200]], [[1, 1, 1], [20, 20, 20]],
                 [[5, 5, 5], [5, 5, 5]], [[5, 5, 5], [5, 5, 5], [5, 5, 5], [100, 5, 5]]):
        pos = p4(rows)
        m = np.arange(1, len(rows) + 1, dtype=np.float32)
        with nb.NBodySim(len(rows)) as sim:
            sim.upload(m, pos)
            sim.build_octree()
            nodes = sim.download_octree()
            keys, depth, _ = sim.morton_keys()
            acc = sim.accel_bh()
        ref_nodes, _, _, _ = orc.build_octree(pos, m)
        compare_trees(nodes, ref_nodes, pos, m, keys, depth)
        ref, _ = orc.bh_accel(orc.LAW_WGSL_BIAS, pos, np.zeros_like(pos), ref_nodes)
        assert np.allclose(acc[:, :3], ref[:, :3], rtol=1e-5, atol=1e-12)


@pytest.mark.parametrize("law", [1, 2])
@pytest.mark.parametrize("name", ["refcube4096", "plummer20000", "clustered"])
def test_walk_matches_reference_walk_on_identical_tree(orc, name, law):
    """Walk parity isolated from build parity: both sides traverse the same node bytes."""
    import nbody_amd as nb
    m, pos = dataset(orc, name)
    n = len(m)
    vel = np.zeros((n, 4), np.float32)
    vel[:, :3] = 0.4 * (orc.gen_refcube(n, seed=77)[1][:, :3] / 125.0 - 1.0)
    ref_nodes, _, _, _ = orc.build_octree(pos, m)
    ref, cnt = orc.bh_accel(law, pos, vel, ref_nodes)
    with nb.NBodySim(n, law=law) as sim:
        sim.upload(m, pos, vel)
        # (a) the oracle's tree, reference allocation-order numbering, uploaded as the reference does
        sim.upload_octree(ref_nodes)
        sim.set_counting(True)
        acc = sim.accel_bh()
        c = sim.counters()
        assert (c["bh_node_visits"], c["bh_leaf_interactions"], c["bh_aggregate_interactions"]) == \
            (cnt["visits"], cnt["leaf"], cnt["aggregate"])
        assert c["bh_max_stack"] <= 8 * 17
        sim.set_counting(False)
        acc2 = sim.accel_bh()
        assert np.array_equal(acc, acc2)
        check_walk(orc, law, pos, vel, ref_nodes, acc, "uploaded tree")
        # (b) the GPU-built tree: the oracle walks the downloaded bytes
        sim.build_octree()
        nodes = sim.download_octree()
        acc = sim.accel_bh()
    check_walk(orc, law, pos, vel, nodes.view(orc.NODE_DTYPE), acc, "GPU-built tree")
    # the two trees differ only by rounding of interior centres: a few MAC decisions may flip
    err = body_rel_err(acc, ref)
    assert np.quantile(err, 0.99) < 1e-4 and err.max() < 2e-2


@pytest.mark.parametrize("law", [1, 2])
@pytest.mark.parametrize("theta,eps2", [(0.3, 1.0), (1.0, 1.0), (0.5, 1e-6), (0.5, 1e-4), (0.5, 100.0), (0.8, 0.01)])
def test_walk_decisions_for_other_opening_angles_and_softenings(orc, law, theta, eps2):
    """The per-node acceptance thresholds are prepared from theta (and, under the PLUMMER law,
    carry the softening): the decisions must stay the reference's for any of them, seen as equal
    visit / interaction counts on the same tree bytes, and accelerations within tolerance."""
    import nbody_amd as nb
    m, pos = dataset(orc, "clustered" if eps2 < 1.0 else "plummer20000")
    n = len(m)
    vel = np.zeros((n, 4), np.float32)
    vel[:, :3] = 0.4 * (orc.gen_refcube(n, seed=78)[1][:, :3] / 125.0 - 1.0)
    with nb.NBodySim(n, law=law, theta=theta, eps2=eps2) as sim:
        sim.upload(m, pos, vel)
        sim.build_octree()
        nodes = sim.download_octree().view(orc.NODE_DTYPE)
        sim.set_counting(True)
        acc = sim.accel_bh()
        c = sim.counters()
        sim.set_counting(False)
        fast = sim.accel_bh()            # the non-counting build of the kernel: same bits
    ref, cnt = orc.bh_accel(law, pos, vel, nodes, eps2=eps2, theta=theta)
    assert (c["bh_node_visits"], c["bh_leaf_interactions"], c["bh_aggregate_interactions"]) == \
        (cnt["visits"], cnt["leaf"], cnt["aggregate"])
    assert np.array_equal(acc, fast)
    ref64 = orc.bh_accel_f64(law, pos, vel, nodes, eps2=eps2, theta=theta)
    e_gpu = np.linalg.norm(acc[:, :3].astype(np.float64) - ref64, axis=1)
    e_ref = np.linalg.norm(ref[:, :3].astype(np.float64) - ref64, axis=1)
    scale = np.linalg.norm(ref64, axis=1)
    # unsoftened close pairs amplify the f32 rounding of the separation itself: both sides are
    # held to the same bound relative to the f64 evaluation of the same node sequence
    assert (e_gpu <= 1e-5 * scale + 4 * e_ref + 1e-30).all(), float(np.max(e_gpu / scale))


def test_theta_zero_walk_equals_allpairs(orc):
    import nbody_amd as nb
    m, pos = orc.gen_refcube(1500, seed=12)
    with nb.NBodySim(1500, law=2, theta=0.0) as sim:
        sim.upload(m, pos)
        sim.build_octree()
        bh = sim.accel_bh()
        ap = sim.accel_allpairs()
    assert body_rel_err(bh, ap).max() < TOL


def test_bh_steps_match_oracle(orc):
    """src/nbody_gpu_bh.rs:445-527: build, walk, Verlet + clamp, swap -- 10 steps."""
    import nbody_amd as nb
    n = 2000
    m, pos = orc.gen_refcube(n, seed=21)
    z = np.zeros((n, 4), np.float32)
    # a body outside the cube is clamped onto its face by the first step
    pos[0, :3] = (251.0, -1.0, 100.0)
    with nb.NBodySim(n, law=1) as sim:
        sim.upload(m, pos)
        sim.step_bh(1)
        p1, v1, a1 = sim.download()
        sim.step_bh(9)
        p10, v10, a10 = sim.download()
    rp, rv, ra = orc.step_bh(1, m, pos, z, z, 1)
    assert np.array_equal(p1[0, :3], [250.0, 0.0, 100.0]) and np.array_equal(rp[0, :3], p1[0, :3])
    assert np.quantile(body_rel_err(a1, ra), 0.99) < 1e-4
    rp, rv, ra = orc.step_bh(1, m, pos, z, z, 10)
    assert np.abs(p10 - rp).max() < 1e-3
    assert np.quantile(body_rel_err(v10, rv), 0.99) < 1e-3


def test_sharded_walk_covers_the_whole(orc):
    import nbody_amd as nb
    n = 3000
    m, pos = orc.gen_plummer(n, seed=5)
    with nb.NBodySim(n, law=2) as sim:
        sim.upload(m, pos)
        sim.build_octree()
        whole = sim.accel_bh()
    parts = np.zeros_like(whole)
    for b, c in ((0, 1000), (1000, 2000)):
        with nb.NBodySim(n, law=2, shard_begin=b, shard_count=c) as sim:
            sim.upload(m, pos)
            sim.build_octree()
            a = sim.accel_bh()
            assert np.all(a[:b] == 0) and np.all(a[b + c:] == 0)
            parts += a
    assert np.array_equal(parts, whole)


def test_upload_rejects_bad_tree(orc):
    import nbody_amd as nb
    m, pos = orc.gen_refcube(10, seed=1)
    nodes, _, _, _ = orc.build_octree(pos, m)
    bad = nodes.copy()
    bad[0]["child"][0] = len(nodes) + 5
    with nb.NBodySim(10) as sim:
        sim.upload(m, pos)
        with pytest.raises(nb.NBodyError):
            sim.upload_octree(bad)
        with pytest.raises(nb.NBodyError):
            sim.accel_bh()          # no tree yet


def test_step_bh_refuses_the_cpu_binary_configuration():
    """src/nbodybh.wgsl has neither the r^-6 law nor the cross-buffer Euler update of
    src/nbody_cpu.rs: asking for them is an error, not a silently different step."""
    import nbody_amd as nb
    m, pos = nb.gen_refcube(100, seed=1)
    for cfg in (dict(law=nb.LAW_CPU_R6), dict(law=nb.LAW_WGSL_BIAS, integrator=nb.INT_CPU_EULER_XBUF)):
        with nb.NBodySim(100, **cfg) as sim:
            sim.upload(m, pos)
            with pytest.raises(nb.NBodyError):
                sim.step_bh(1)
            with pytest.raises(nb.NBodyError):
                sim.bh_forces_ranked(0, 100)


def test_ranked_forces_and_integration_equal_a_plain_step(orc):
    """The replicated-state multi-GPU scheme on one context: traversal by runs of the sorted order
    into the rank-ordered buffer, then un-ranking + integration of all bodies == nbody_step_bh."""
    import nbody_amd as nb
    n = 5003
    m, pos, vel = nb.gen_plummer(n, seed=9, equilibrium=True)
    with nb.NBodySim(n, law=1) as ref:
        ref.upload(m, pos, vel)
        ref.step_bh(2)
        rp, rv, ra = ref.download()
        _, _, order = ref.morton_keys()     # of the octree the second step built
    with nb.NBodySim(n, law=1) as sim:
        sim.upload(m, pos, vel)
        for step in range(2):
            for b, c in ((0, 1700), (1700, 3303), (5003, 0)):     # uneven runs, an empty one
                sim.bh_forces_ranked(b, c)
            if step == 1:
                ranked = sim.read_buffer(nb.BUF_ACC_PARTIAL)
            sim.integrate_ranked()
        p, v, a = sim.download()
    assert np.array_equal(p, rp) and np.array_equal(v, rv) and np.array_equal(a, ra)
    # the exchanged buffer holds, at rank r, the acceleration of the r-th body of the sorted order
    assert np.array_equal(ranked[:, :3], a[order][:, :3])
    with nb.NBodySim(n, law=1, shard_begin=0, shard_count=n // 2) as sharded:
        sharded.upload(m, pos, vel)
        with pytest.raises(nb.NBodyError):
            sharded.bh_forces_ranked(0, 10)       # needs a context that holds all bodies
    with nb.NBodySim(n, law=1) as sim:
        sim.upload(m, pos, vel)
        with pytest.raises(nb.NBodyError):
            sim.bh_forces_ranked(n - 5, 10)       # run outside the bodies
        with pytest.raises(nb.NBodyError):
            sim.integrate_ranked()                # no octree built yet


def _comb_tree(levels):
    """An uploaded tree that keeps seven sibling groups pending per level: every interior node has
    eight interior children, seven of them stubs with one leaf body each, the eighth (octant 7,
    visited first) continuing downwards.  Ranges are huge, so a traversal opens everything."""
    import nbody_amd as nb
    nodes = []

    def new(node_type, com=(0.0, 0.0, 0.0), mass=0.0, rng=0.0):
        nd = np.zeros((), nb.NODE_DTYPE)
        nd["node_type"], nd["com"], nd["total_mass"], nd["range"] = node_type, com, mass, rng
        nd["pos_min"], nd["pos_max"] = com, com
        nodes.append(nd)
        return len(nodes) - 1

    root = new(nb.NODE_INTERIOR, (100.0, 100.0, 100.0), 1.0, 1e6)
    cur, leaves = root, []
    for lvl in range(levels):
        kids = []
        for k in range(8):
            c = new(nb.NODE_INTERIOR, (100.0 + lvl, 100.0 + k, 100.0), 1.0, 1e6)
            kids.append(c)
        nodes[cur]["child"] = kids
        for k in range(7):
            leaf = new(nb.NODE_LEAFBODY, (100.0 + lvl, 100.0 + k, 101.0), 1.0 + 0.01 * (lvl * 7 + k))
            nodes[kids[k]]["child"][0] = leaf
            leaves.append(leaf)
        cur = kids[7]
    last = new(nb.NODE_LEAFBODY, (90.0, 90.0, 90.0), 2.0)
    nodes[cur]["child"][0] = last
    return np.array(nodes, nb.NODE_DTYPE)


def test_traversal_stack_overflow_is_an_error_not_a_wrong_answer(orc):
    """The per-wavefront stack holds 160 pending groups, plenty for any 16-level octree; a deeper
    uploaded tree that needs more must fail loudly at the next blocking call."""
    import nbody_amd as nb
    n = 64
    m, pos = orc.gen_refcube(n, seed=2)
    vel = np.zeros((n, 4), np.float32)
    ok = _comb_tree(12)                       # 7 x 12 pending groups: fits
    with nb.NBodySim(n, law=2) as sim:
        sim.upload(m, pos)
        sim.upload_octree(ok)
        a = sim.accel_bh()
    ra, _ = orc.bh_accel(2, pos, vel, ok)
    assert body_rel_err(a, ra).max() < TOL
    deep = _comb_tree(40)                     # 7 x 40 pending groups: does not
    with nb.NBodySim(n, law=2) as sim:
        sim.upload(m, pos)
        sim.upload_octree(deep)
        with pytest.raises(nb.NBodyError, match="stack"):
            sim.accel_bh()


def test_quadrupole_option_tightens_the_force_error(orc):
    """Opt-in quadrupole moments (no counterpart in the reference): same tree, same acceptance
    decisions, and the error against the direct sum drops by a large factor."""
    import nbody_amd as nb
    n = 20000
    m, pos, _ = nb.gen_plummer(n, seed=13)
    vel = np.zeros((n, 4), np.float32)
    exact = orc.accel_allpairs_f64(orc.LAW_PLUMMER, pos, vel, m)

    def run(quadrupole):
        with nb.NBodySim(n, law=nb.LAW_PLUMMER, quadrupole=quadrupole, leaflist_mode=1) as sim:
            sim.upload(m, pos)
            sim.build_octree()
            sim.set_counting(True)
            a = sim.accel_bh()
            return a, sim.counters()

    a0, c0 = run(0)
    a1, c1 = run(1)
    for k in ("bh_node_visits", "bh_leaf_interactions", "bh_aggregate_interactions", "octree_nodes"):
        assert c0[k] == c1[k]
    scale = np.linalg.norm(exact, axis=1)
    e0 = np.linalg.norm(a0[:, :3] - exact, axis=1) / scale
    e1 = np.linalg.norm(a1[:, :3] - exact, axis=1) / scale
    print(f"median / 99th percentile relative force error: monopole {np.median(e0):.2e} / {np.quantile(e0, 0.99):.2e}, "
          f"with quadrupoles {np.median(e1):.2e} / {np.quantile(e1, 0.99):.2e}")
    # cells are accepted up to range/distance = 0.5, so the next multipole order gains a factor ~2
    assert np.median(e1) < 0.65 * np.median(e0)
    assert np.quantile(e1, 0.99) < 0.65 * np.quantile(e0, 0.99)
    with pytest.raises(nb.NBodyError):
        nb.NBodySim(n, law=nb.LAW_WGSL_BIAS, quadrupole=1)


def test_quadrupole_of_a_far_cluster_matches_the_direct_sum(orc):
    """One tight cluster seen from far away: monopole error ~ (s/r)^2, with the quadrupole ~ (s/r)^3."""
    import nbody_amd as nb
    rng = np.random.default_rng(3)
    k = 64
    pos = np.zeros((k + 1, 4), np.float32)
    pos[:k, :3] = 60.0 + rng.uniform(-1.0, 1.0, (k, 3)) * np.array([1.0, 0.4, 0.2])   # an elongated clump
    pos[k, :3] = (200.0, 180.0, 190.0)                                               # the far target
    m = np.concatenate([rng.uniform(0.5, 8.0, k), [1.0]]).astype(np.float32)
    vel = np.zeros_like(pos)
    exact = orc.accel_allpairs_f64(orc.LAW_PLUMMER, pos, vel, m)[k]
    err = []
    for quadrupole in (0, 1):
        with nb.NBodySim(k + 1, law=nb.LAW_PLUMMER, quadrupole=quadrupole, leaflist_mode=1) as sim:
            sim.upload(m, pos)
            sim.build_octree()
            a = sim.accel_bh()[k, :3]
        err.append(np.linalg.norm(a - exact) / np.linalg.norm(exact))
    print('far cluster, relative error: monopole %.2e, with quadrupole %.2e' % tuple(err))
    assert err[1] < 0.05 * err[0] and err[1] < 5e-6, err


@pytest.mark.parametrize("name", ["refcube250", "refcube4096", "plummer20000"])
def test_uncapped_octree_rs_variant(orc, name):
    """max_depth = 0: the dormant builder of src/octree.rs -- same cells, is_leaf at offset 84,
    max_depth on the root (src/octree.rs:36,58), as long as 16 levels separate all bodies."""
    import nbody_amd as nb
    m, pos = dataset(orc, name)
    ref_nodes, leaf, dep, md = orc.build_octree(pos, m, variant=1)
    assert dep.max() < 16
    with nb.NBodySim(len(m), max_depth=0) as sim:
        sim.upload(m, pos)
        sim.build_octree()
        nodes = sim.download_octree()
        keys, depth, _ = sim.morton_keys()
    assert np.array_equal(depth, dep)
    assert len(nodes) == len(ref_nodes)
    assert nodes[0]["depth"] == md == ref_nodes[0]["max_depth"]          # the max_depth field of the root
    assert not nodes[1:]["depth"].any()

    def cells(ns):
        out, stack = {}, [(0, ())]
        while stack:
            idx, path = stack.pop()
            out[path] = ns[idx]
            if not ns[idx]["node_type"]:                              # is_leaf == 0
                stack += [(int(c), path + (k,)) for k, c in enumerate(ns[idx]["child"]) if c]
        return out

    g, r = cells(nodes), cells(ref_nodes)
    assert set(g) == set(r)
    for path, rn in r.items():
        gn = g[path]
        assert gn["node_type"] == rn["node_type"]
        assert np.array_equal(gn["pos_min"], rn["pos_min"]) and np.array_equal(gn["pos_max"], rn["pos_max"])
        assert gn["range"] == rn["range"]
        if rn["node_type"]:
            assert np.array_equal(gn["com"], rn["com"]) and gn["total_mass"] == rn["total_mass"]
        else:
            assert abs(gn["total_mass"] - rn["total_mass"]) <= 2e-4 * rn["total_mass"]


def test_uncapped_variant_refuses_what_16_levels_cannot_separate(orc):
    import nbody_amd as nb
    pos = p4([[10, 10, 10], [10, 10, 10 + 250.0 / 65536 / 8], [200, 100, 50]])
    m = np.ones(3, np.float32)
    with nb.NBodySim(3, max_depth=0) as sim:
        sim.upload(m, pos)
        with pytest.raises(nb.NBodyError, match="level-16"):
            sim.build_octree()


def test_uncapped_variant_refuses_massless_bodies(orc):
    """src/octree.rs:49 takes a node of total mass 0 for an empty one (a massless body is replaced by the next
    body to reach its node); neither the library nor the oracle reproduces that: both refuse.  The capped
    builder tells empty nodes by their type and takes massless bodies."""
    import nbody_amd as nb
    pos = p4([[10, 10, 10], [100, 40, 30], [200, 100, 50]])
    m = np.array([1, 0, 2], np.float32)
    with nb.NBodySim(3, max_depth=0) as sim:
        with pytest.raises(nb.NBodyError, match="mass 0"):
            sim.upload(m, pos)
    with pytest.raises(ValueError):
        orc.build_octree(pos, m, variant=1)
    with nb.NBodySim(3) as sim:
        sim.upload(m, pos)
        sim.build_octree()
        ref, _, _, _ = orc.build_octree(pos, m, variant=0)
        assert sim.counters()["octree_nodes"] == len(ref)
