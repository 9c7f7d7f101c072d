"""The rule by which a domain chooses what to export of its octree (nbody_amd/csrc/octree.hip: k_let_box,
let_reach, k_let_mark), checked on the CPU against the reference's own tree and traversal: on the oracle's
insertion-order tree (src/octree_maxdepth.rs) every node that a body of ANOTHER domain visits under
src/nbodybh.wgsl:104-106 must be among the children of the nodes the rule marks.  The rule is restated here
in numpy (same boxes, same thresholds and margins); the GPU tests check that the kernels' exports reproduce the
one tree's accelerations bit for bit, this one that the rule itself is a superset on a tree it did not build."""
import numpy as np
import pytest

T_LEAFBODY, T_LEAFLIST, T_INTERIOR = 1, 2, 3
D = 16
f32 = np.float32


def _tree_geometry(nodes):
    """Depth and cell prefix (3 bits per level) of every node, from the root down."""
    n = len(nodes)
    depth = np.full(n, -1, np.int64)
    prefix = np.zeros(n, np.int64)
    depth[0] = 0
    for p in range(n):                      # children are allocated after their parents
        if depth[p] < 0:
            continue
        t = int(nodes["node_type"][p])
        if t == T_INTERIOR:
            for k in range(8):
                c = int(nodes["child"][p][k])
                if c:
                    depth[c] = depth[p] + 1
                    prefix[c] = prefix[p] * 8 + k
        elif t == T_LEAFLIST:
            first, cnt = int(nodes["child"][p][0]), int(nodes["child"][p][1])
            depth[first:first + cnt] = depth[p] + 1
            prefix[first:first + cnt] = prefix[p] * 8   # (members lie in their list's cell)
    return depth, prefix


def _children(nodes, p):
    t = int(nodes["node_type"][p])
    if t == T_INTERIOR:
        return [int(c) for c in nodes["child"][p] if c]
    if t == T_LEAFLIST:
        first, cnt = int(nodes["child"][p][0]), int(nodes["child"][p][1])
        return list(range(first, first + cnt))
    return []


def _partition(keys, n, G, L):
    """Runs of whole depth-L cells in Morton order, a cell going where its middle body would fall."""
    cell = (keys >> np.uint64(3 * (D - L))).astype(np.int64)
    counts = np.bincount(cell, minlength=8 ** L)
    before = np.concatenate(([0], np.cumsum(counts)[:-1]))
    cell_dom = np.minimum(G - 1, ((before + counts // 2) * G) // n)
    return cell, cell_dom


def _boxes(pos, keys, members):
    """Box 0 around all of a domain's bodies, then one per depth-2 cell holding some of them."""
    if not len(members):
        return []
    p = pos[members, :3]
    out = [(p.min(axis=0), p.max(axis=0))]
    c2 = (keys[members] >> np.uint64(3 * (D - 2))).astype(np.int64)
    for c in np.unique(c2):
        q = p[c2 == c]
        out.append((q.min(axis=0), q.max(axis=0)))
    return out


def _reach(box, com, r2_accept, soft):
    lo, hi = box
    d = np.maximum(np.maximum(lo - com, com - hi), f32(0))
    d2 = f32(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
    return not (d2 * f32(0.9999) > (r2_accept - soft) * f32(1.0001) + soft * f32(1e-5))


@pytest.mark.parametrize("dataset,n,G,L,soft", [("plummer", 3000, 4, 2, 0.0), ("plummer", 4000, 8, 3, 1.0),
                                                 ("refcube", 2500, 5, 2, 0.0), ("clump", 2000, 3, 1, 0.0)])
def test_exports_cover_what_other_domains_visit(orc, dataset, n, G, L, soft):
    theta = f32(0.5)
    if dataset == "plummer":
        m, pos = orc.gen_plummer(n, seed=8)[:2]
    else:
        m, pos = orc.gen_refcube(n, seed=8)
        if dataset == "clump":                   # a fifth of the bodies inside one depth-16 cell: a leaf list
            cell = f32(250.0 / 65536)
            base = cell * np.floor(f32(70.0) / cell)
            rng = np.random.default_rng(1)
            pos[: n // 5, :3] = (base + cell * (0.1 + 0.8 * rng.random((n // 5, 3)))).astype(np.float32)
    nodes, _, _, _ = orc.build_octree(pos, m, variant=0)
    keys = orc.morton_keys(pos)
    depth, prefix = _tree_geometry(nodes)
    cell, cell_dom = _partition(keys, n, G, L)
    body_dom = cell_dom[cell]
    members = [np.nonzero(body_dom == k)[0] for k in range(G)]
    # the domain a node's subtree lies in (depth >= L), or -1 for the part of the tree every domain holds
    owner = np.full(len(nodes), -1, np.int64)
    deep = depth >= L
    owner[deep] = cell_dom[prefix[deep] >> (3 * (depth[deep] - L))]
    boxes = [_boxes(pos, keys, members[k]) for k in range(G)]
    # the rule: children of every node of domain r that some other domain's boxes reach
    soft32 = f32(soft)
    accept = f32(1.0) / (theta * theta * (f32(1.0) - f32(4e-6)))
    exported = np.zeros(len(nodes), bool)
    for p in range(len(nodes)):
        if owner[p] < 0 or int(nodes["node_type"][p]) == T_LEAFBODY:
            continue
        rg = f32(nodes["range"][p])
        r2_accept = rg * rg * accept + soft32 * (f32(1.0) + f32(1e-6))
        com = nodes["com"][p]
        hit = False
        for s in range(G):
            if s == owner[p] or not boxes[s] or not _reach(boxes[s][0], com, r2_accept, soft32):
                continue
            if any(_reach(b, com, r2_accept, soft32) for b in boxes[s][1:]):
                hit = True
                break
        if hit:
            exported[_children(nodes, p)] = True
    # the reference's traversal, all bodies of a domain at once: which nodes does some body of it visit?
    kids = [_children(nodes, p) for p in range(len(nodes))]
    com_all, rng_all, typ = nodes["com"], nodes["range"], nodes["node_type"]
    missing = 0
    needed = 0
    for s in range(G):
        ps = pos[members[s], :3]
        if not len(ps):
            continue
        reach_mask = {0: np.ones(len(ps), bool)}
        for p in range(len(nodes)):
            mask = reach_mask.pop(p, None)
            if mask is None or not mask.any():
                continue
            if owner[p] >= 0 and owner[p] != s and depth[p] > L:
                needed += 1
                missing += 0 if exported[p] else 1
            if typ[p] == T_LEAFBODY:
                continue
            dlt = ps - com_all[p]
            r = np.sqrt(((dlt[:, 0] * dlt[:, 0]) + (dlt[:, 1] * dlt[:, 1])) + (dlt[:, 2] * dlt[:, 2]), dtype=np.float32)
            with np.errstate(divide="ignore", invalid="ignore"):
                accept_here = (rng_all[p] / r) < theta          # src/nbodybh.wgsl:104-106
            opened = mask & ~accept_here
            if opened.any():
                for c in kids[p]:
                    reach_mask[c] = reach_mask[c] | opened if c in reach_mask else opened.copy()
    assert needed > 0 and missing == 0
    below = int(((owner >= 0) & (depth > L)).sum())
    assert exported.sum() <= below
    if dataset == "plummer" and n >= 4000:
        assert exported.sum() < below           # and it does leave something behind
