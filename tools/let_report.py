"""Essential trees on one GPU: for every domain, the merged tree as its GPU would hold it
(nbody_bh_set_domain_view): what the other domains export, how that compares with the nodes its
targets actually visit, and what one rank of a distributed step does, timed piece by piece on the
unshared GPU: its own build, its own export, putting the received nodes and the top tree together,
and the traversal for its own bodies.  One JSON object.
usage: python tools/let_report.py [n] [domains] [top_levels]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nb


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    domains = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    top_levels = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    m, p, v = nb.gen_plummer(n, seed=42)
    with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
        sim.upload(m, p, v)
        sim.build_octree()
        acc_one = sim.accel_bh()
        sim.set_kernel_timing(True)
        sim.build_octree(); sim.accel_bh(); sim.kernel_time(bh=True); sim.kernel_time(build=True)
        for _ in range(3):
            sim.build_octree()
            sim.accel_bh()
        one_walk, one_build = sim.kernel_time(bh=True)[0], sim.kernel_time(build=True)[0]
        sim.set_kernel_timing(False)
    with nb.NBodySim(n, law=nb.LAW_PLUMMER) as sim:
        sim.upload(m, p, v)
        sim.bh_set_domains(domains, top_levels)
        sim.build_octree()
        sim.set_counting(True)
        sim.accel_bh()
        sim.set_counting(False)
        whole = sim.bh_domain_stats()
        out = {"n": n, "domains": domains, "top_levels": top_levels, "one_tree_build_ms": one_build, "one_tree_walk_ms": one_walk,
               "bodies": whole["bodies"], "nodes": whole["nodes"], "merged_nodes_whole": whole["merged_nodes"],
               "visited_of_others": whole["imported_nodes"], "partition_ms": whole["partition_ms"], "views": []}
        cuts = np.concatenate(([0], np.cumsum(whole["bodies"]))).astype(np.int64)
        sim.set_kernel_timing(True)
        for d in range(domains):
            sim.bh_set_domain_view(d)
            sim.build_octree()
            st = sim.bh_domain_stats()
            _, _, order = sim.morton_keys(with_depth=False)
            mine = order[cuts[d]:cuts[d + 1]]
            acc = sim.accel_bh()
            same = bool(np.array_equal(acc[mine], acc_one[mine]))
            # the traversal of this domain's own run of the sorted order, over its own merged tree
            sim.bh_forces_ranked(int(cuts[d]), int(cuts[d + 1] - cuts[d]))
            sim.kernel_time(bh=True)
            for _ in range(3):
                sim.bh_forces_ranked(int(cuts[d]), int(cuts[d + 1] - cuts[d]))
            walk_ms = sim.kernel_time(bh=True)[0]
            st = sim.bh_domain_stats()
            exports = [st["export_ms"][k] for k in range(domains) if k != d and st["bodies"][k]]
            received = sum(st["exported_nodes"])
            out["views"].append({
                "domain": d, "bodies": st["bodies"][d], "own_nodes": st["nodes"][d], "exported_by_others": st["exported_nodes"],
                "received_nodes": received, "received_MB": received * 36 / 1e6, "merged_nodes": st["merged_nodes"],
                "own_build_ms": st["build_ms"][d], "export_ms_each": float(np.mean(exports)) if exports else 0.0,
                # everything between the trees and the finished top tree, less the exports the others would have made
                "assemble_ms": st["merge_ms"] - float(np.sum(exports)), "own_walk_ms": walk_ms,
                "accelerations_of_its_bodies_bit_equal_to_one_tree": same})
        vs = out["views"]
        worst = max(vs, key=lambda x: x["own_build_ms"] + x["export_ms_each"] + x["assemble_ms"] + x["own_walk_ms"])
        out["slowest_rank"] = {
            "domain": worst["domain"],
            "ms": {"partition": whole["partition_ms"], "own_build": worst["own_build_ms"], "own_export": worst["export_ms_each"],
                   "assemble": worst["assemble_ms"], "own_walk": worst["own_walk_ms"]},
            "sum_ms": whole["partition_ms"] + worst["own_build_ms"] + worst["export_ms_each"] + worst["assemble_ms"] + worst["own_walk_ms"],
            "one_gpu_build_plus_walk_ms": one_build + one_walk,
            "note": "kernel work of one rank, measured on an unshared GPU; the exchanges themselves (received_MB of nodes, "
                    "16 B per body of accelerations) and the un-grouping + integration of all bodies come on top"}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
