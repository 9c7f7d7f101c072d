"""Traversal time and wavefront steps for the Morton and Hilbert target orders with the library
NBODY_HIP_LIB names (measurement builds with another -DNBODY_WALK_DEFER_LANES: how many lanes at most
may open a node for it to be left to those lanes' own walks).
usage: NBODY_HIP_LIB=tools/variants/libnbody_hip_d6.so python tools/walk_defer_sweep.py [N] [law]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nbody_amd as nb
from walk_order_explore import coords_from_keys, hilbert_index

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
law = {"plummer": nb.LAW_PLUMMER, "wgsl_bias": nb.LAW_WGSL_BIAS}[sys.argv[2] if len(sys.argv) > 2 else "plummer"]
m, p, v = nb.gen_plummer(n, seed=42)
with nb.NBodySim(n, law=law) as sim:
    sim.upload(m, p, v)
    sim.build_octree()
    keys, depth, order = sim.morton_keys()
    h = hilbert_index(*coords_from_keys(keys[order]))
    horder = order[np.argsort(h, kind="stable")]
    ref = None
    for label, o in (("library default", None), ("morton", order), ("hilbert (host, full depth)", horder)):
        sim.bh_set_target_order(o)
        sim.set_counting(True)
        acc = sim.accel_bh()
        c = sim.counters()
        sim.set_counting(False)
        if ref is None:
            ref = acc
        err = float((np.linalg.norm((acc - ref)[:, :3], axis=1) / np.linalg.norm(ref[:, :3], axis=1)).max())
        sim.accel_bh(); sim.accel_bh()
        sim.set_kernel_timing(True)
        sim.kernel_time(bh=True)
        for _ in range(5):
            sim.accel_bh()
        ms, _ = sim.kernel_time(bh=True)
        sim.set_kernel_timing(False)
        print(json.dumps({"lib": os.path.basename(os.environ.get("NBODY_HIP_LIB", "default")), "order": label, "n": n,
                          "wave_steps": c["bh_wave_node_visits"], "node_visits": c["bh_node_visits"],
                          "interactions": c["bh_leaf_interactions"] + c["bh_aggregate_interactions"],
                          "walk_ms": ms, "max_rel_diff_to_first": err}), flush=True)
