"""Per configuration and law: worst and 99.9th-percentile relative error of the GPU accelerations
against the oracle's f64 evaluation and against its f32 sum in the reference's order.  Writes JSON
(argv[1], default gpurun_out/parity.json); the committed copy is profiles/r02_parity.json."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_amd as nb
from oracle import oracle as orc, parity

out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity.json"
report = {"note": "relative error = ||a_gpu - a_ref|| / ||a_ref|| per body; oracle = oracle/nbody_oracle.cpp (CPU restatement "
                  "of src/nbody.wgsl / src/nbody_cpu.rs / src/nbodybh.wgsl); f32 reference order = one f32 accumulator over "
                  "the sources in index order, as the reference sums", "configs": {}}
LAWS = {"plummer": nb.LAW_PLUMMER, "wgsl_bias": nb.LAW_WGSL_BIAS, "cpu_r6": nb.LAW_CPU_R6}


def allpairs(name, n, laws, dataset, full=False):
    m, pos, vel = nb.gen_plummer(n, seed=42, equilibrium=True) if dataset == "plummer" else (*nb.gen_refcube(n, seed=42), None)
    if vel is None:
        vel = np.zeros((n, 4), np.float32)
    ranges = [(0, n)] if full else parity.slices_of(n)
    for law in laws:
        t0 = time.time()
        with nb.NBodySim(n, law=LAWS[law]) as sim:
            sim.upload(m, pos, vel)
            acc = sim.accel_allpairs()
            path = sim.counters()["allpairs_path"]
        st = parity.allpairs_stats(LAWS[law], m, pos, vel, acc, ranges)
        st.update({"n_bodies": n, "dataset": dataset, "kernel": "pair (unordered)" if path == 2 else "plain (ordered)",
                   "target_ranges": ranges})
        report["configs"][f"{name}/{law}"] = st
        print(f"{name}/{law}: vs f64 max {st['vs_f64']['max']:.2e} p99.9 {st['vs_f64']['p999']:.2e}; vs f32 ref order max "
              f"{st['vs_f32_reference_order']['max']:.2e} p99.9 {st['vs_f32_reference_order']['p999']:.2e}  ({time.time() - t0:.0f} s)",
              flush=True)


def barnes_hut(name, n, laws, sample):
    """The traversal against the oracle's traversal of the SAME tree bytes, over a range of targets."""
    m, pos, vel = nb.gen_plummer(n, seed=42, equilibrium=True)
    for law in laws:
        with nb.NBodySim(n, law=LAWS[law]) as sim:
            sim.upload(m, pos, vel)
            sim.build_octree()
            nodes = sim.download_octree().view(orc.NODE_DTYPE)
            acc = sim.accel_bh()
        b = (n // 2) // 64 * 64
        r32, cnt = orc.bh_accel(LAWS[law], pos, vel, nodes, i_begin=b, i_count=sample)
        r64 = orc.bh_accel_f64(LAWS[law], pos, vel, nodes, i_begin=b, i_count=sample)
        r64 = r64[b:b + sample] if len(r64) == n else r64
        a = acc[b:b + sample, :3].astype(np.float64)
        r32 = r32[b:b + sample, :3].astype(np.float64)
        scale = np.linalg.norm(r64, axis=1)
        q = lambda x: {"max": float(x.max()), "p999": float(np.quantile(x, 0.999)), "median": float(np.median(x))}
        st = {"n_bodies": n, "dataset": "plummer", "targets": sample, "theta": 0.5,
              "vs_f64_same_nodes": q(np.linalg.norm(a - r64, axis=1) / scale),
              "vs_f32_reference_order": q(np.linalg.norm(a - r32, axis=1) / np.linalg.norm(r32, axis=1)),
              "f32_reference_order_vs_f64": q(np.linalg.norm(r32 - r64, axis=1) / scale)}
        report["configs"][f"{name}/{law}"] = st
        print(f"{name}/{law}: vs f64 max {st['vs_f64_same_nodes']['max']:.2e}; vs f32 ref order max {st['vs_f32_reference_order']['max']:.2e}", flush=True)


allpairs("config1 all-pairs N=4096 refcube", 4096, ["cpu_r6", "wgsl_bias", "plummer"], "refcube", full=True)
allpairs("config2 all-pairs N=65536", 65536, ["wgsl_bias", "plummer"], "plummer", full=True)
allpairs("config3 all-pairs N=1048576 (4 target ranges)", 1 << 20, ["wgsl_bias", "plummer"], "plummer")
allpairs("all-pairs N=1048576+777 (padded last superblock)", (1 << 20) + 777, ["plummer"], "plummer")
barnes_hut("config4 Barnes-Hut N=1048576", 1 << 20, ["wgsl_bias", "plummer"], 8192)
barnes_hut("config5 Barnes-Hut N=16777216", 1 << 24, ["wgsl_bias", "plummer"], 4096)
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(report, open(out_path, "w"), indent=1)
