"""Turns the raw output of tools/collect_profiles.sh <tag> and tools/pmc_walk.sh N <tag> (under gpurun_out/)
into the summaries kept in profiles/ (argv: tag, output prefix such as r02)."""
import json, os, shutil, sys
tag, out = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
for name in ("bench_line.json", "bench_kernel_stats.csv", "bench_kernels_by_grid.json"):
    shutil.copy(os.path.join(src, name), os.path.join(dst, f"{out}_{name}"))

def traffic(fetch_file, write_file):
    """HBM bytes per kernel: FETCH_SIZE doubled (gfx950 correction), WRITE_SIZE as is; counter unit KB."""
    f, w = json.load(open(fetch_file)), json.load(open(write_file))
    res = {}
    for k in f:
        n = f[k]["launches"]
        fs = f[k].get("FETCH_SIZE", 0.0) / n
        ws = w.get(k, {}).get("WRITE_SIZE", 0.0) / max(w.get(k, {}).get("launches", n), 1)
        res[k] = {"FETCH_SIZE_KB_per_launch": fs, "WRITE_SIZE_KB_per_launch": ws,
                  "hbm_bytes_per_launch": (2.0 * fs + ws) * 1024.0, "launches": n}
    return res

ap = {"command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 --no-extras",
      "n_bodies": 1048576, "n_gpus": 1, "law": "plummer",
      "kernels": traffic(os.path.join(src, "pmc_allpairs_FETCH_SIZE.json"), os.path.join(src, "pmc_allpairs_WRITE_SIZE.json")),
      "note": "FETCH_SIZE doubled per the gfx950 correction for 16-B-per-lane coalesced reads; WRITE_SIZE taken as is; counter unit KB"}
json.dump(ap, open(os.path.join(dst, f"{out}_pmc_allpairs.json"), "w"), indent=1)

bh = {"command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) -- python3 tools/bh_prof.py N  (3 builds + 3 traversals)",
      "note": "per-launch averages; FETCH_SIZE doubled (gfx950 correction), WRITE_SIZE as is; counter unit KB"}
for n in (1048576, 16777216):
    for suffix in ("", "_wgsl_bias"):
        ff = os.path.join(src, f"pmc_bh_{n}{suffix}_FETCH_SIZE.json")
        if not os.path.exists(ff):
            continue
        t = traffic(ff, os.path.join(src, f"pmc_bh_{n}{suffix}_WRITE_SIZE.json"))
        # one build = every k_* launch that is not the traversal, its launch order (k_walk_order), the Hilbert
        # grouping of its targets (k_hilbert and the 32-bit-key sort passes, made once per 16 builds), the upload or the
        # integration of the step tools/bh_prof.py takes
        not_build = ("k_walk", "k_pack_posm", "k_hilbert", "k_rs_hist<unsigned int>", "k_rs_scatter<unsigned int>", "k_integrate")
        build = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in t.items()
                    if k.startswith("k_") and not k.startswith(not_build)) / t["k_morton"]["launches"]
        t["octree_build_bytes_per_build"] = build  # (tools/bh_prof.py builds twice per iteration: k_morton runs once per build)
        bh[f"n_{n}{suffix}"] = t
json.dump(bh, open(os.path.join(dst, f"{out}_pmc_bh.json"), "w"), indent=1)

line = json.load(open(os.path.join(src, "bench_line.json")))
visits = {1048576: line["barnes_hut"]["wave_node_visits"], 16777216: line["barnes_hut_16m"]["wave_node_visits"]}
sq = {"source": "rocprofv3 --pmc <8 SQ counters> (one run per counter set, no tracing) over `python3 tools/bh_prof.py N` (tools/pmc_walk.sh); per-launch averages of k_walk<PLUMMER>",
      "counter_sets": ["SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY",
                       "SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVES",
                       "GRBM_GUI_ACTIVE"]}
for n in (1048576, 16777216):
    c = {}
    for part in "abc":
        d = json.load(open(os.path.join(root, "gpurun_out", f"pmc_{tag}_{n}_{part}.json")))
        k = [x for x in d if x.startswith("k_walk<")][0]
        for name, val in d[k].items():
            if name != "launches":
                c[name] = val / d[k]["launches"]
    v = visits[n]
    simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0   # kernel cycles x SIMDs
    c["derived"] = {"wave_node_visits_per_launch": v, "kernel_cycles (GRBM_GUI_ACTIVE / 8 XCDs)": c["GRBM_GUI_ACTIVE"] / 8.0,
                    "valu_per_wave_visit": c["SQ_INSTS_VALU"] / v, "salu_per_wave_visit": c["SQ_INSTS_SALU"] / v,
                    "branch_per_wave_visit": c["SQ_INSTS_BRANCH"] / v, "smem_per_wave_visit": c["SQ_INSTS_SMEM"] / v,
                    "lds_per_wave_visit": c["SQ_INSTS_LDS"] / v, "simd_cycles_per_wave_visit": simd_cycles / v,
                    "resident_waves_per_simd": 4.0 * c["SQ_WAVE_CYCLES"] / simd_cycles,  # (the counter ticks once per four cycles)
                    "scalar_unit_instructions_per_cu_cycle": (c["SQ_INSTS_SALU"] + c["SQ_INSTS_BRANCH"] + c["SQ_INSTS_SMEM"]) / (c["GRBM_GUI_ACTIVE"] / 8.0 * 256.0)}
    sq[f"n_{n}"] = c
json.dump(sq, open(os.path.join(dst, f"{out}_pmc_walk_sq.json"), "w"), indent=1)
print("written", out)
