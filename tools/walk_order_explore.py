"""How the grouping of targets into wavefronts changes the traversal's wavefront node visits and time
(nbody_bh_set_target_order; results do not depend on the grouping).  Candidates: the Morton order
(default), a Hilbert order derived from the Morton keys, and either of them with a wavefront ended
early where the next body leaves the cell that is `delta` levels above the first body's "natural" cell
(the deepest cell around it holding >= 64 bodies).
usage: python tools/walk_order_explore.py [N] [law]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import nbody_amd as nb

EMPTY = 0xFFFFFFFF


def coords_from_keys(keys, bits=16):
    x = np.zeros(len(keys), np.uint32)
    y = np.zeros(len(keys), np.uint32)
    z = np.zeros(len(keys), np.uint32)
    for l in range(bits):                       # level l (0 = top) is the triple at bits 3 (bits - 1 - l)
        sh = np.uint64(3 * (bits - 1 - l))
        d = ((keys >> sh) & np.uint64(7)).astype(np.uint32)
        x = (x << 1) | ((d >> 2) & 1)
        y = (y << 1) | ((d >> 1) & 1)
        z = (z << 1) | (d & 1)
    return x, y, z


def hilbert_index(x, y, z, bits=16):
    """Skilling's transform: the Hilbert index (3 * bits bits) of integer coordinates; every octree
    cell is a contiguous run of the curve."""
    X = [x.astype(np.uint32).copy(), y.astype(np.uint32).copy(), z.astype(np.uint32).copy()]
    Q = np.uint32(1 << (bits - 1))
    while Q > 1:
        P = np.uint32(Q - 1)
        for i in range(3):
            hit = (X[i] & Q) != 0
            X[0] = np.where(hit, X[0] ^ P, X[0])
            t = np.where(hit, np.uint32(0), (X[0] ^ X[i]) & P)
            X[0] = X[0] ^ t
            X[i] = X[i] ^ t
        Q = np.uint32(Q >> 1)
    X[1] ^= X[0]
    X[2] ^= X[1]
    t = np.zeros_like(X[0])
    Q = np.uint32(1 << (bits - 1))
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ np.uint32(Q - 1), t)
        Q = np.uint32(Q >> 1)
    for i in range(3):
        X[i] ^= t
    h = np.zeros(len(x), np.uint64)
    for b in range(bits - 1, -1, -1):
        for i in range(3):
            h = (h << np.uint64(1)) | ((X[i] >> np.uint32(b)) & 1).astype(np.uint64)
    return h


def cell_ranges(skeys, level, bits=16):
    """For bodies in an order in which every cell is contiguous: [start, end) of each body's level-`level` cell."""
    cid = skeys >> np.uint64(3 * (bits - level))
    edge = np.flatnonzero(np.concatenate(([True], cid[1:] != cid[:-1], [True])))
    run = np.repeat(np.arange(len(edge) - 1), np.diff(edge))
    return edge[run], edge[run + 1]


def with_breaks(order_bodies, cell_keys, delta, bits=16, group=64):
    """Greedy wavefronts over `order_bodies`; cell_keys[r] = a key of the body at position r whose top 3 l
    bits name its level-l cell (Morton or Hilbert alike).  A wavefront ends where the next body is
    outside the cell `delta` levels above the first body's natural cell."""
    n = len(order_bodies)
    starts, ends = [], []
    for l in range(bits + 1):
        s, e = cell_ranges(cell_keys, l, bits)
        starts.append(s)
        ends.append(e)
    size = np.stack([e - s for s, e in zip(starts, ends)])          # [level][body]
    natural = (size >= group).sum(axis=0) - 1                          # deepest level with >= group bodies
    out = []
    r = 0
    waves = 0
    while r < n:
        l = max(int(natural[r]) - delta, 0)
        stop = min(r + group, int(ends[l][r]))
        chunk = order_bodies[r:stop]
        out.append(chunk)
        if len(chunk) < group:
            out.append(np.full(group - len(chunk), EMPTY, np.uint32))
        waves += 1
        r = stop
    return np.concatenate(out).astype(np.uint32), waves


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    law = {"plummer": nb.LAW_PLUMMER, "wgsl_bias": nb.LAW_WGSL_BIAS}[sys.argv[2] if len(sys.argv) > 2 else "plummer"]
    m, p, v = nb.gen_plummer(n, seed=42)
    with nb.NBodySim(n, law=law) as sim:
        sim.upload(m, p, v)
        sim.build_octree()
        keys, depth, order = sim.morton_keys()
        skeys = keys[order]
        ref = None

        def measure(label, ordr, waves=None):
            nonlocal ref
            sim.bh_set_target_order(ordr)
            sim.set_counting(True)
            acc = sim.accel_bh()
            c = sim.counters()
            sim.set_counting(False)
            if ref is None:
                ref = acc
            same = bool(np.array_equal(ref, acc))
            sim.accel_bh()
            sim.accel_bh()                       # second traversal of the same order: longest-first launch order in use
            sim.set_kernel_timing(True)
            sim.kernel_time(bh=True)
            for _ in range(5):
                sim.accel_bh()
            ms, _ = sim.kernel_time(bh=True)
            sim.set_kernel_timing(False)
            w = waves if waves is not None else (n + 63) // 64
            print(json.dumps({"order": label, "n": n, "wavefronts": w, "wave_node_visits": c["bh_wave_node_visits"],
                              "node_visits": c["bh_node_visits"],
                              "lane_use": c["bh_node_visits"] / (64.0 * c["bh_wave_node_visits"]),
                              "walk_ms": ms, "bit_equal_to_default": same}), flush=True)

        measure("morton (default)", None)
        t0 = time.time()
        x, y, z = coords_from_keys(skeys)
        h = hilbert_index(x, y, z)
        perm = np.argsort(h, kind="stable")
        horder = order[perm]
        hkeys = h[perm]
        print(json.dumps({"hilbert_host_s": time.time() - t0}), flush=True)
        measure("hilbert", horder)
        for delta in (0, 1, 2, 3):
            o, w = with_breaks(order, skeys, delta)
            measure(f"morton, breaks delta={delta}", o, w)
        for delta in (1, 2, 3):
            o, w = with_breaks(horder, hkeys, delta)
            measure(f"hilbert, breaks delta={delta}", o, w)
        # Hilbert order at reduced resolution (top levels only), Morton within: what a cheap re-sort would give
        for lv in (6, 8, 10, 12):
            hk = h >> np.uint64(3 * (16 - lv))
            pr = np.argsort(hk, kind="stable")
            measure(f"hilbert top {lv} levels, morton below", order[pr])


if __name__ == "__main__":
    main()
