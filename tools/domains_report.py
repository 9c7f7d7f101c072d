"""The domain-decomposed octree build against the one tree, at a benchmark size: per-domain build time
(what one GPU of G would spend), the top tree, the nodes each domain would import, and that nothing
about the result changes (accelerations bit for bit, visits and interactions).
usage: python tools/domains_report.py [N] [domains] [top_levels] [law]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
L = int(sys.argv[3]) if len(sys.argv) > 3 else 4
law = {"plummer": nb.LAW_PLUMMER, "wgsl_bias": nb.LAW_WGSL_BIAS}[sys.argv[4] if len(sys.argv) > 4 else "plummer"]
m, p, v = nb.gen_plummer(n, seed=42)


def measure(sim, reps=5):
    sim.build_octree(); sim.accel_bh()
    sim.set_kernel_timing(True)
    sim.kernel_time(build=True); sim.kernel_time(bh=True)
    for _ in range(reps):
        sim.build_octree()
        acc = sim.accel_bh()
    b, _ = sim.kernel_time(build=True)
    w, _ = sim.kernel_time(bh=True)
    sim.set_kernel_timing(False)
    sim.set_counting(True)
    sim.accel_bh()
    c = sim.counters()
    sim.set_counting(False)
    return acc, b, w, c


with nb.NBodySim(n, law=law) as sim:
    sim.upload(m, p, v)
    acc0, b0, w0, c0 = measure(sim)
with nb.NBodySim(n, law=law) as sim:
    sim.upload(m, p, v)
    sim.bh_set_domains(G, L)
    acc1, b1, w1, c1 = measure(sim)
    st = sim.bh_domain_stats()
keys = ("bh_node_visits", "bh_leaf_interactions", "bh_aggregate_interactions")
out = {"n_bodies": n, "domains": G, "top_levels": L, "law": "plummer" if law == nb.LAW_PLUMMER else "wgsl_bias",
       "one_tree": {"build_ms": b0, "walk_ms": w0, "nodes": c0["octree_nodes"]},
       "domain_build": {"whole_build_ms_all_domains_one_after_the_other": b1, "walk_ms": w1,
                        "per_domain_build_ms": st["build_ms"], "partition_ms": st["partition_ms"], "merge_ms": st["merge_ms"],
                        "bodies": st["bodies"], "nodes": st["nodes"], "top_entries": st["top_entries"], "top_nodes": st["top_nodes"],
                        "merged_nodes": st["merged_nodes"], "imported_nodes": st["imported_nodes"],
                        "imported_MB_per_step": [x * 36 / 1e6 for x in st["imported_nodes"]],
                        "top_nodes_visited": st["top_nodes_visited"]},
       "accelerations_bit_equal": bool(np.array_equal(acc0, acc1)),
       "counts_equal": all(c0[k] == c1[k] for k in keys), "merged_nodes_equal_one_tree": st["merged_nodes"] == c0["octree_nodes"],
       "slowest_domain_over_one_tree_build": max(st["build_ms"]) / b0}
print(json.dumps(out), flush=True)
