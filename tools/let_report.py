"""Essential trees on one GPU: for every domain, the merged tree as its GPU would hold it
(nbody_bh_set_domain_view): what the other domains export, how that compares with the nodes its
targets actually visit, and the time of a build and a traversal of its own bodies' share."""
import json
import os
import sys
import time

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
        sim.bh_set_domains(domains, top_levels)
        sim.build_octree()
        sim.set_counting(True)
        sim.accel_bh()
        sim.set_counting(False)
        whole = sim.bh_domain_stats()
        out = {"n": n, "domains": domains, "top_levels": top_levels, "bodies": whole["bodies"], "nodes": whole["nodes"],
               "merged_nodes_whole": whole["merged_nodes"], "visited_of_others": whole["imported_nodes"], "views": []}
        for d in range(domains):
            sim.bh_set_domain_view(d)
            sim.build_octree()
            st = sim.bh_domain_stats()
            out["views"].append({"domain": d, "exported_by_others": st["exported_nodes"], "received_nodes": sum(st["exported_nodes"]),
                                 "received_MB": sum(st["exported_nodes"]) * 36 / 1e6, "merged_nodes": st["merged_nodes"],
                                 "assemble_ms": st["merge_ms"]})
        print(json.dumps(out))


if __name__ == "__main__":
    main()
