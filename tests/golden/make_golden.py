#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/nbody_oracle.cpp).

The reference ships no fixtures and cannot be built here, so these vectors pin the ORACLE (and
through it the HIP path) against regressions; they are not outputs of the reference binaries.
Run from the repository root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def bodies(n, seed):
    m, pos = orc.gen_refcube(n, seed=seed)
    vel = np.zeros((n, 4), np.float32)
    vel[:, :3] = 0.3 * (orc.gen_refcube(n, seed=seed + 1000)[1][:, :3] / 125.0 - 1.0)
    return m, pos, vel


def main():
    for n, seeds in ((64, (1, 2, 3)), (250, (1, 2, 3)), (4096, (1,))):
        for seed in seeds:
            m, pos, vel = bodies(n, seed)
            z = np.zeros((n, 4), np.float32)
            d = {"masses": m, "pos": pos, "vel": vel}
            for name, law in (("cpu_r6", 0), ("wgsl_bias", 1), ("plummer", 2)):
                d[f"acc_{name}"] = orc.accel_allpairs(law, pos, vel, m)
                d[f"acc64_{name}"] = orc.accel_allpairs_f64(law, pos, vel, m)
                d[f"abs_{name}"] = orc.accel_allpairs_abs(law, pos, vel, m)
            nodes, leaf, dep, _ = orc.build_octree(pos, m)
            d["tree_nodes"] = nodes.view(np.uint8).reshape(-1, 96)
            d["leaf_depth"] = dep
            d["keys"] = orc.morton_keys(pos)
            for name, law in (("wgsl_bias", 1), ("plummer", 2)):
                d[f"bh_acc_{name}"], _ = orc.bh_accel(law, pos, vel, nodes)
            # src/nbody_cpu.rs: 100 steps (N = 4,096 is the reference's CPU configuration at its own size),
            # both buffer sets started from the same positions
            r = orc.step_cpu(m, pos, z, z, pos, z, z, 100)
            d["cpu100_pos"], d["cpu100_vel"] = r[0], r[1]
            if n <= 250:
                p, v, a = orc.step_wgsl_allpairs(1, m, pos, vel, z, 20)
                d["wgsl20_pos"], d["wgsl20_vel"], d["wgsl20_acc"] = p, v, a
                p, v, a = orc.step_bh(1, m, pos, vel, z, 10)
                d["bh10_pos"], d["bh10_vel"] = p, v
            np.savez_compressed(os.path.join(OUT, f"refcube_n{n}_s{seed}.npz"), **d)
            print("wrote", f"refcube_n{n}_s{seed}.npz", len(nodes), "nodes")


if __name__ == "__main__":
    main()
