"""Committed golden vectors (tests/golden/*.npz, produced by tests/golden/make_golden.py from the
CPU oracle).  CPU part: the oracle still reproduces them bit for bit.  GPU part: the HIP path
matches them to the stated tolerances."""
import glob
import os

import numpy as np
import pytest

from conftest import body_rel_err

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))
LAWS = (("cpu_r6", 0), ("wgsl_bias", 1), ("plummer", 2))


def test_fixtures_present():
    assert len(GOLD) == 7


@pytest.mark.parametrize("path", GOLD, ids=os.path.basename)
def test_oracle_reproduces_golden(orc, path):
    g = np.load(path)
    m, pos, vel = g["masses"], g["pos"], g["vel"]
    m2, pos2 = orc.gen_refcube(len(m), seed=int(path.split("_s")[1].split(".")[0]))
    assert np.array_equal(m, m2) and np.array_equal(pos, pos2)
    for name, law in LAWS:
        assert np.array_equal(orc.accel_allpairs(law, pos, vel, m), g[f"acc_{name}"])
    nodes, _, dep, _ = orc.build_octree(pos, m)
    assert np.array_equal(nodes.view(np.uint8).reshape(-1, 96), g["tree_nodes"])
    assert np.array_equal(dep, g["leaf_depth"]) and np.array_equal(orc.morton_keys(pos), g["keys"])
    for name, law in LAWS[1:]:
        assert np.array_equal(orc.bh_accel(law, pos, vel, nodes)[0], g[f"bh_acc_{name}"])
    if "cpu100_pos" in g:
        z = np.zeros_like(pos)
        r = orc.step_cpu(m, pos, z, z, pos, z, z, 100)
        assert np.array_equal(r[0], g["cpu100_pos"]) and np.array_equal(r[1], g["cpu100_vel"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=os.path.basename)
def test_hip_matches_golden(path):
    import nbody_amd as nb
    g = np.load(path)
    m, pos, vel = g["masses"], g["pos"], g["vel"]
    n = len(m)
    z = np.zeros_like(pos)
    for name, law in LAWS:
        with nb.NBodySim(n, law=law) as sim:
            sim.upload(m, pos, vel)
            acc = sim.accel_allpairs()
        ref64 = g[f"acc64_{name}"]
        e = np.linalg.norm(acc[:, :3] - ref64, axis=1)
        e_ref = np.linalg.norm(g[f"acc_{name}"][:, :3] - ref64, axis=1)
        # 1e-5 relative, up to the oracle's own rounding (e_ref) and the f32 floor of the sum
        floor = 4 * np.finfo(np.float32).eps * g[f"abs_{name}"]
        assert np.all(e <= 1e-5 * np.linalg.norm(ref64, axis=1) + e_ref + floor)
    nodes = np.ascontiguousarray(g["tree_nodes"]).view(nb.NODE_DTYPE).reshape(-1)
    for name, law in LAWS[1:]:
        with nb.NBodySim(n, law=law) as sim:
            sim.upload(m, pos, vel)
            sim.upload_octree(nodes)                    # the reference-numbered tree
            acc = sim.accel_bh()
            assert np.quantile(body_rel_err(acc, g[f"bh_acc_{name}"]), 0.99) < 1e-5
            assert body_rel_err(acc, g[f"bh_acc_{name}"]).max() < 1e-4
            sim.build_octree()
            keys, depth, _ = sim.morton_keys()
            assert np.array_equal(keys, g["keys"]) and np.array_equal(depth, g["leaf_depth"])
            assert len(sim.download_octree()) == len(nodes)
    if n == 4096:
        # Configuration 1 at its own size: 100 steps of src/nbody_cpu.rs:319-360.  The unsoftened r^-5
        # law throws bodies out through close encounters (the oracle's state reaches |pos| ~ 1e14), and
        # such an encounter amplifies any rounding difference without bound, so the state is compared
        # where it is comparable: every body finite on both sides, the median body bit-close, nine
        # bodies in ten within 1e-3 (measured: median 0, 93 % within 1e-3, 84 % within 1e-4 of their
        # displacement; tools/cpu100_check.py).
        with nb.NBodySim(n, law=0, integrator=0) as sim:
            sim.upload(m, pos)
            sim.upload_alt(pos)
            sim.step_allpairs(100)
            p, v, a = sim.download()
        d = np.linalg.norm((p - g["cpu100_pos"])[:, :3].astype(np.float64), axis=1)
        assert np.isfinite(p).all() and np.isfinite(g["cpu100_pos"]).all()
        assert np.median(d) <= 1e-6 and (d < 1e-3).mean() >= 0.9, (float(np.median(d)), float((d < 1e-3).mean()))
    if "wgsl20_pos" in g:
        with nb.NBodySim(n, law=1) as sim:
            sim.upload(m, pos, vel)
            sim.step_allpairs(20)
            p, v, a = sim.download()
        assert np.abs(p - g["wgsl20_pos"]).max() < 1e-4
        assert body_rel_err(v, g["wgsl20_vel"]).max() < 1e-4
        with nb.NBodySim(n, law=0, integrator=0) as sim:
            sim.upload(m, pos)
            sim.upload_alt(pos)
            sim.step_allpairs(100)
            p, v, a = sim.download()
        assert np.abs(p - g["cpu100_pos"]).max() < 1e-3
        with nb.NBodySim(n, law=1) as sim:
            sim.upload(m, pos, vel)
            sim.step_bh(10)
            p, v, a = sim.download()
        assert np.abs(p - g["bh10_pos"]).max() < 1e-3
