#!/usr/bin/env python3
"""Headline benchmark: all-pairs gravity step, N = 1,048,576 Plummer-sphere bodies, on N GPUs.

One "step" = forces of all N sources on this rank's targets + velocity-Verlet integration +
(N > 1 GPUs) the all-gather of the new positions.  Inputs are resident in HBM before timing.
Prints ONE JSON line (rank 0).  Extra legs, outside the timed region: the reference's own WGSL
law (acos/pow bias) on the same bodies, Barnes-Hut steps at N = 1M and N = 16M (sharded the same
way over the GPUs), and the CPU baseline (the oracle's restatement of
src/nbody_cpu.rs timed on the host cores, rank 0 at --gpus 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_INTERACTION = {"plummer": 20, "wgsl_bias": 46, "cpu_r6": 20}
PEAK_FP32_TFLOPS = 157.3  # MI355X FP32 vector peak (256 CU x 256 flop/clk x 2.4 GHz)
FLOP_PER_CLOCK = 256 * 256  # whole chip, FP32 vector
# what the pair kernel executes per UNORDERED pair (equal masses): 3 sub, 3 fma for r^2, rsq,
# 2 mul for r^-3, 3 fma on the targets, 3 mul/fma for the reaction = 24 flop (26 with unequal masses)
EXECUTED_FLOP_PER_PAIR = 24
PEAK_HBM_GBPS = 8000.0
N_SIMDS = 256 * 4
NOMINAL_CLOCK_MHZ = 2400.0
# SIMD cycles one wavefront node visit of the traversal costs when nothing but its own instruction
# sequence is in the way: that sequence run alone, records as constant scalar operands, 6-8 resident
# wavefronts per SIMD (tools/ubench_walkmix.hip, sequence A; profiles/r02_ubench_walkmix.txt: 56-61)
WALK_ISSUE_CYCLES_PER_VISIT = {"plummer": 56.0}


def profile_traffic(pattern, pick):
    """HBM bytes per launch measured offline with rocprofv3 --pmc (FETCH_SIZE and WRITE_SIZE in
    separate passes, summaries under profiles/): PMC counters cannot be read from inside this
    process.  The newest summary that has the entry wins.  Returns (bytes, file) or (None, None)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            v = pick(json.load(open(path)))
        except (OSError, KeyError, ValueError, TypeError, IndexError):
            v = None
        if v is not None:
            return float(v), "profiles/" + os.path.basename(path)
    return None, None


def kernel_entry(kernels, name):
    """The entry of a kernel in a per-kernel summary, whatever its template arguments."""
    for k, v in kernels.items():
        if k == name or k.startswith(name + "<"):
            return v
    raise KeyError(name)


def cpu_baseline(seconds_budget=10.0):
    """src/nbody_cpu.rs step (oracle port, OpenMP over targets like the rayon loop) at the
    reference CPU config: refcube N = 4096, 100 steps, repeated for about seconds_budget."""
    from oracle import oracle as orc
    n, steps = 4096, 100
    m, pos = orc.gen_refcube(n, seed=42)
    z = np.zeros((n, 4), np.float32)
    threads = orc.num_threads()
    orc.step_cpu(m, pos, z, z, pos, z, z, 2)   # warm the thread pool
    reps, t0 = 0, time.time()
    while reps == 0 or time.time() - t0 < seconds_budget:
        orc.step_cpu(m, pos, z, z, pos, z, z, steps)
        reps += 1
    dt = time.time() - t0
    inter = float(n) * (n - 1) * steps * reps
    return {"value": inter / dt, "unit": "interactions/s", "cores": threads, "kind": "port",
            "steps_per_s": steps * reps / dt,
            "sample": f"refcube N={n}, {reps} x {steps} steps of the nbody_cpu.rs law (r^-5, unsoftened) + cross-buffer "
                      f"Euler, {threads} OpenMP threads, {dt:.1f} s"}


def cpu_baseline_bh(masses, pos):
    """The reference's Barnes-Hut host work and shader on the CPU (oracle port): the insertion-order
    octree build of src/octree_maxdepth.rs (one thread, as in the reference) on the bench's N = 1M
    bodies, and the traversal of src/nbodybh.wgsl for a bounded sample of the bodies."""
    from oracle import oracle as orc
    n = len(masses)
    t0 = time.time()
    nodes, _, _, _ = orc.build_octree(pos, masses)
    build_s = time.time() - t0
    sample = min(n, 40000)
    vel = np.zeros((n, 4), np.float32)
    t0 = time.time()
    _, cnt = orc.bh_accel(orc.LAW_PLUMMER, pos, vel, nodes, i_begin=0, i_count=sample)
    walk_s = time.time() - t0
    inter = cnt["leaf"] + cnt["aggregate"]
    return {"octree_build_ms": build_s * 1e3, "octree_nodes": int(len(nodes)), "build_threads": 1,
            "walk_interactions_per_s": inter / walk_s, "walk_threads": orc.num_threads(), "kind": "port",
            "sample": f"build: all N={n} Plummer bodies; traversal: the first {sample} bodies, PLUMMER law, theta 0.5"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--bodies", dest="n", type=int, default=1 << 20,
                    help="body count (use --bodies under torch.distributed.run, whose own parser claims --n)")
    ap.add_argument("--law", default="plummer", choices=["plummer", "wgsl_bias"])
    ap.add_argument("--no-extras", action="store_true", help="skip the WGSL-law, Barnes-Hut and CPU-baseline legs")
    ap.add_argument("--bh-balance-every", type=int, default=0,
                    help="Barnes-Hut legs on several GPUs: every K steps move the ranks' runs of the sorted order after "
                         "their measured traversal times (0: equal runs; a centred Plummer sphere is balanced as it is)")
    ap.add_argument("--bh-domains", type=int, default=0,
                    help="with --comm library and several GPUs: Barnes-Hut with one domain per rank, each rank building "
                         "its own domain's octree and the ranks exchanging essential trees; the value is the depth of the "
                         "cells the bodies are dealt by (5 at N = 16M).  0: the replicated tree")
    ap.add_argument("--comm", default="torch", choices=["torch", "library"],
                    help="who runs the per-step exchange on several GPUs: torch.distributed collectives issued from "
                         "Python (default), or the library's own RCCL communicator (nbody_comm_init)")
    ap.add_argument("--no-config2", action="store_true",
                    help="skip the all-pairs legs at N=65536 (the reference's own GPU scale); in a kernel trace of this "
                         "command they are told from the headline launches by their grid size")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import nbody_amd as nb
    from nbody_amd.shard import ShardedSim, HipShardBackend, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # Rehearsal on a one-GPU box: NBODY_BENCH_BACKEND=gloo NBODY_BENCH_DEVICE=0 lets several ranks
    # share device 0 (RCCL refuses that); the numbers then mean nothing, the code path is the same.
    backend = os.environ.get("NBODY_BENCH_BACKEND", "nccl")
    if "NBODY_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["NBODY_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    n = args.n
    law = {"plummer": nb.LAW_PLUMMER, "wgsl_bias": nb.LAW_WGSL_BIAS}[args.law]
    begin, count = shard_range(n, rank, world)
    masses, pos, vel = nb.gen_plummer(n, seed=42)   # every rank generates the same bodies

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_run(law_id, steps, warmup):
        be = HipShardBackend(n, begin, count, local_rank, law=law_id)
        be.upload(masses, pos, vel)
        sim = ShardedSim(be, n, rank, world, library_comm=args.comm == "library")
        sim.step_allpairs(warmup)
        be.sim.set_kernel_timing(True)
        be.sim.kernel_time()
        barrier()
        t0 = time.perf_counter()
        sim.step_allpairs(steps)
        barrier()
        dt = time.perf_counter() - t0
        kms, launches = be.sim.kernel_time()
        clock = be.sim.kernel_clock()
        path = be.sim.counters()["allpairs_path"]
        agree = True
        if world > 1:
            t = torch.tensor([dt, kms], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, kms = float(t[0]), float(t[1])
            # every rank must hold the same gathered positions
            with torch.cuda.stream(be.torch_stream):
                cs = be.posm().to(torch.float64).sum().reshape(1)
            torch.cuda.synchronize()
            lo, hi = cs.clone(), cs.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            agree = bool(lo.item() == hi.item()) and bool(torch.isfinite(cs).item())
        be.sim.close()
        return dt, kms, launches, agree, clock, path

    pair = world > 1 and args.law == "plummer" and n >= HipShardBackend.PAIR_SHARD_MIN_N
    scheme = "single" if world == 1 else (f"pair-shard x{world}" if pair else f"target-shard x{world}")
    scheme_text = ("" if world == 1 else
                   ": pair tiles dealt over the ranks, per-step all-reduce of partial accelerations + all-gather of positions"
                   if pair else ": targets split by index range, per-step all-gather of positions")
    dt, kernel_ms, launches, agree, clock_mhz, path = timed_run(law, args.steps, args.warmup)
    interactions = float(n) * n * args.steps
    flop = FLOP_PER_INTERACTION[args.law]
    # the dominant kernel: one launch covers this rank's share of the N x N interactions (its
    # targets x all sources, or its 1/world share of the pair tiles)
    per_launch = float(n) * n / world
    achieved = per_launch * flop / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    out = {
        "metric": "all-pairs body-body interactions/s at N=" + ("1M" if n == 1 << 20 else str(n)),
        "value": interactions / dt,
        "unit": "interactions/s",
        "n_gpus": args.gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "steps_per_s": args.steps / dt,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"all-pairs direct sum, N={n} Plummer sphere (a=12.5, equal masses, seed 42), "
                               f"law={args.law}, velocity-Verlet dt=0.1, {args.gpus} GPU(s)" + scheme_text,
                   "n_bodies": n, "law": args.law, "parallelism": scheme},
        "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32_TFLOPS, "traffic": None,
                     "kernel": ("k_ap_sym" if path == 2 else "k_allpairs_plummer_pk" if args.law == "plummer" else "k_allpairs"),
                     "kernel_ms": kernel_ms, "launches": launches, "flop_per_interaction": flop,
                     "note": "FP32 vector (VALU) roofline. `achieved`/`frac` price the ALGORITHMIC work, 20 flop per "
                             "ordered interaction x N^2; k_ap_sym evaluates each unordered pair once (Newton's third "
                             "law), so what it EXECUTES is `executed_tflops`/`executed_frac` (24 flop per unordered "
                             "pair); `*_at_clock` price against the peak at the shader clock measured in the kernel"},
    }
    rf = out["roofline"]
    if path == 2:
        executed = per_launch / 2 * EXECUTED_FLOP_PER_PAIR / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        rf.update({"executed_flop_per_unordered_pair": EXECUTED_FLOP_PER_PAIR, "executed_tflops": executed,
                   "executed_frac": executed / PEAK_FP32_TFLOPS})
    else:
        rf.update({"executed_tflops": achieved, "executed_frac": achieved / PEAK_FP32_TFLOPS})
    if clock_mhz > 0:
        peak_clk = FLOP_PER_CLOCK * clock_mhz * 1e6 / 1e12
        rf.update({"shader_clock_mhz": clock_mhz, "peak_at_clock": peak_clk, "frac_at_clock": achieved / peak_clk,
                   "executed_frac_at_clock": rf["executed_tflops"] / peak_clk})
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process;
    # the value measured with rocprofv3 (separate FETCH_SIZE / WRITE_SIZE passes over this same
    # command) is kept under profiles/ and quoted when it is for this workload.
    def ap_pick(pmc):
        if (pmc["n_bodies"], pmc["law"]) != (n, args.law):
            return None
        b = kernel_entry(pmc["kernels"], rf["kernel"])["hbm_bytes_per_launch"]
        return b / world if pmc["n_gpus"] == 1 else None   # a rank's launch covers 1 / world of the pair work

    rf["traffic"], src = profile_traffic("r*_pmc_allpairs.json", ap_pick)
    if src:
        rf["traffic_source"] = src + " (rocprofv3 --pmc over `bench.py --no-extras`, offline, one GPU" + \
            ("; divided by the rank count)" if world > 1 else ")")
    if not args.no_extras and args.law == "plummer":
        # the reference's shader law (src/nbody.wgsl:54-81) on the same bodies, fewer steps
        wsteps = max(1, min(2, args.steps))
        wdt, wk, _, wagree, _, _ = timed_run(nb.LAW_WGSL_BIAS, wsteps, 1)
        agree = agree and wagree
        wach = per_launch * FLOP_PER_INTERACTION["wgsl_bias"] / (wk * 1e-3) / 1e12 if wk > 0 else 0.0
        out["wgsl_bias_law"] = {"value": float(n) * n * wsteps / wdt, "unit": "interactions/s",
                                "ms_per_step": wdt / wsteps * 1e3, "achieved_tflops": wach,
                                "frac": wach / PEAK_FP32_TFLOPS, "steps": wsteps}

    def small_leg(nsmall, law_id, steps):
        """All-pairs at the reference's own GPU scale (one GPU, resident inputs)."""
        m_, p_, v_ = nb.gen_plummer(nsmall, seed=42)
        with nb.NBodySim(nsmall, law=law_id) as s_:
            s_.upload(m_, p_, v_)
            s_.step_allpairs(3)
            s_.set_kernel_timing(True)
            s_.sync()
            t0 = time.perf_counter()
            s_.step_allpairs(steps)
            s_.sync()
            sdt = time.perf_counter() - t0
            skm, _ = s_.kernel_time()
            spath = s_.counters()["allpairs_path"]
        fl = FLOP_PER_INTERACTION["plummer" if law_id == nb.LAW_PLUMMER else "wgsl_bias"]
        ach = float(nsmall) * nsmall * fl / (skm * 1e-3) / 1e12 if skm > 0 else 0.0
        return {"interactions_per_s": float(nsmall) * nsmall * steps / sdt, "ms_per_step": sdt / steps * 1e3, "steps": steps,
                "kernel": "k_ap_sym" if spath == 2 else "plain", "kernel_ms": skm, "achieved_tflops": ach,
                "frac": ach / PEAK_FP32_TFLOPS, "flop_per_interaction": fl}

    def ordered_pair_leg(nbodies, steps):
        """The ordered-pair PLUMMER kernel (what sharded contexts, small N and the no-allocation entry point
        run) at the headline size, the pair kernel switched off for the duration of the leg."""
        old = os.environ.get("NBODY_AP_SYM")
        os.environ["NBODY_AP_SYM"] = "0"
        try:
            r = small_leg(nbodies, nb.LAW_PLUMMER, steps)
        finally:
            if old is None:
                del os.environ["NBODY_AP_SYM"]
            else:
                os.environ["NBODY_AP_SYM"] = old
        r["kernel"] = "k_allpairs_plummer_pk (one evaluation per ordered pair; equal masses applied once per target)"
        return r

    def bh_leg(nbodies, bsteps, bh_law=None):
        """Barnes-Hut step, theta = 0.5, depth cap 16.  With several GPUs every rank holds all bodies:
        replicated octree build, traversal of this rank's run of the sorted (Morton) order,
        all-gather of the accelerations (16 B/body), integration + clamp of all bodies."""
        m_, p_, v_ = (masses, pos, vel) if nbodies == n else nb.gen_plummer(nbodies, seed=42)
        b_, c_ = shard_range(nbodies, rank, world)
        be = HipShardBackend(nbodies, b_, c_, local_rank, law=nb.LAW_PLUMMER if bh_law is None else bh_law, bh_ranked=True)
        be.upload(m_, p_, v_)
        sim = ShardedSim(be, nbodies, rank, world, bh_balance_every=args.bh_balance_every,
                         library_comm=args.comm == "library")
        distributed = bool(args.bh_domains) and world > 1 and args.comm == "library"
        if distributed:
            be.sim.bh_set_domains(world, args.bh_domains)
        sim.step_bh(2)
        be.sim.set_counting(True)                        # counting build of the traversal kernel, own targets only
        if world == 1:
            be.sim.build_octree()
            be.sim.accel_bh()                            # (the grouping of the timed steps: targets along the Hilbert curve)
        else:
            be.sim.bh_forces_ranked(*sim.bh_ranges[rank])
        c = be.sim.counters()
        be.sim.set_counting(False)
        be.sim.set_kernel_timing(True)
        be.sim.kernel_time(bh=True)
        be.sim.kernel_time(build=True)
        barrier()
        t0 = time.perf_counter()
        sim.step_bh(bsteps)
        barrier()
        bdt = time.perf_counter() - t0
        wms, _ = be.sim.kernel_time(bh=True)
        bms, _ = be.sim.kernel_time(build=True)
        same = True
        if world > 1:   # replicated state: every rank must have computed the same positions
            with torch.cuda.stream(be.torch_stream):
                cs = be.posm().to(torch.float64).sum().reshape(1)
            torch.cuda.synchronize()
            lo, hi = cs.clone(), cs.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            same = bool(lo.item() == hi.item()) and bool(torch.isfinite(cs).item())
        be.sim.close()
        tot = torch.tensor([c["bh_leaf_interactions"] + c["bh_aggregate_interactions"], c["bh_node_visits"],
                            c["bh_wave_node_visits"]], device="cuda", dtype=torch.float64)
        mx = torch.tensor([bdt, wms, bms], device="cuda", dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        inter, visits, wave_visits = (float(x) for x in tot)
        bdt, wms, bms = float(mx[0]), float(mx[1]), float(mx[2])
        # bytes the traversal asks for: 32 B per wavefront node visit + per body 3 x 16 B read, 48 B written
        walk_bytes = (32.0 * wave_visits + nbodies * (16 + 16 + 16 + 48)) / world
        # algorithmic bytes of one (replicated) build: bodies read, keys + indices written, six
        # sort passes of 12 B in and out, bodies read again for the nodes, 32 B records written and
        # read once more by the aggregate passes
        build_bytes = nbodies * (16.0 + 12 + 144 + 16) + c["octree_nodes"] * 64.0
        law_name = "plummer" if bh_law in (None, nb.LAW_PLUMMER) else "wgsl_bias"
        pkey = f"n_{nbodies}" + ("" if law_name == "plummer" else "_wgsl_bias")
        walk_traffic, walk_src = profile_traffic(
            "r*_pmc_bh.json", lambda d: kernel_entry(d[pkey], "k_walk")["hbm_bytes_per_launch"] / world)
        build_traffic, build_src = profile_traffic("r*_pmc_bh.json", lambda d: d[pkey]["octree_build_bytes_per_build"])
        visits_per_s = wave_visits / world / (wms * 1e-3) if wms > 0 else 0.0
        issue = {"bound": "issue", "unit": "wavefront node visits/s", "achieved": visits_per_s,
                 "simd_cycles_per_visit": N_SIMDS * NOMINAL_CLOCK_MHZ * 1e6 / visits_per_s if visits_per_s > 0 else 0.0,
                 "note": "this law's per-visit instruction sequence (six transcendentals per term) has not been timed in "
                         "isolation, so no ceiling is quoted for it; the PLUMMER legs carry one"}
        if law_name in WALK_ISSUE_CYCLES_PER_VISIT:
            cyc = WALK_ISSUE_CYCLES_PER_VISIT[law_name]
            peak_visits = N_SIMDS * NOMINAL_CLOCK_MHZ * 1e6 / cyc
            issue = {"bound": "issue", "achieved": visits_per_s, "peak": peak_visits, "unit": "wavefront node visits/s",
                     "frac": visits_per_s / peak_visits, "cycles_per_visit_alone": cyc,
                     "simd_cycles_per_visit": N_SIMDS * NOMINAL_CLOCK_MHZ * 1e6 / visits_per_s if visits_per_s > 0 else None,
                     "clock_mhz": NOMINAL_CLOCK_MHZ, "simds": N_SIMDS,
                     "note": "the ceiling the traversal actually has: wavefront node visits x the SIMD cycles one visit's "
                             "instruction sequence takes alone (no memory, 6-8 wavefronts per SIMD: "
                             "profiles/r02_ubench_walkmix.txt, sequence A) / (1,024 SIMDs x the nominal clock)"}
        return {"n_bodies": nbodies, "n_gpus": world, "law": law_name,
                "tree": "one domain per rank, essential trees exchanged" if distributed else "replicated" if world > 1 else "one",
                "ranks_hold_identical_positions": same, "theta": 0.5, "max_depth": 16, "steps": bsteps,
                "steps_per_s": bsteps / bdt, "ms_per_step": bdt / bsteps * 1e3, "walk_kernel_ms": wms,
                "build_ms": bms,
                "build_roofline": {"bound": "hbm", "achieved": build_bytes / (bms * 1e-3) / 1e9 if bms > 0 else None,
                                   "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                   "frac": build_bytes / (bms * 1e-3) / 1e9 / PEAK_HBM_GBPS if bms > 0 else None,
                                   "bytes_per_build": build_bytes, "traffic": build_traffic, "traffic_source": build_src},
                "octree_nodes": c["octree_nodes"], "interactions_per_step": inter,
                "interactions_per_s": inter * bsteps / bdt, "node_visits_per_body": visits / nbodies,
                "wave_node_visits": wave_visits,
                "lane_utilisation": visits / (64.0 * wave_visits) if wave_visits > 0 else None,
                "walk_roofline": {"bound": "hbm", "achieved": walk_bytes / (wms * 1e-3) / 1e9 if wms > 0 else None,
                                  "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                  "frac": walk_bytes / (wms * 1e-3) / 1e9 / PEAK_HBM_GBPS if wms > 0 else None,
                                  "traffic": walk_traffic, "traffic_source": walk_src, "kernel": "k_walk", "kernel_ms": wms,
                                  "bytes_per_launch": walk_bytes,
                                  "note": "algorithmic bytes = 32 B x wavefront node visits + 96 B per body, per GPU.  The "
                                          "records are served by L2 / Infinity Cache; what limits the kernel is the issue "
                                          "rate of its instruction sequence: see issue_roofline"},
                "issue_roofline": issue,
                "equivalent_allpairs_interactions_per_s": float(nbodies) * nbodies * bsteps / bdt}

    def domains_leg(nbodies, domains, top_levels):
        """The domain-decomposed octree build on this one GPU (every domain's tree from its own bodies, the
        top of the tree rebuilt from the domains' cells): what one GPU of `domains` would spend building,
        and how many nodes of the others' trees its targets visit -- against the one tree over all bodies."""
        m_, p_, v_ = (masses, pos, vel) if nbodies == n else nb.gen_plummer(nbodies, seed=42)
        res = {}
        for label, g in (("one_tree", 0), ("domains", domains)):
            with nb.NBodySim(nbodies, law=nb.LAW_PLUMMER) as s_:
                s_.upload(m_, p_, v_)
                if g:
                    s_.bh_set_domains(g, top_levels)
                s_.build_octree()
                s_.accel_bh()
                s_.set_kernel_timing(True)
                s_.kernel_time(build=True)
                for _ in range(3):
                    s_.build_octree()
                bms, _ = s_.kernel_time(build=True)
                s_.set_kernel_timing(False)
                s_.set_counting(True)
                acc_ = s_.accel_bh()
                c_ = s_.counters()
                res[label] = (bms, acc_, c_, s_.bh_domain_stats() if g else None)
                if g:
                    # one rank's part of a distributed step, rehearsed for domain 0: the merged tree holds its own
                    # tree and only what the other domains would send it (their essential-tree exports)
                    s_.set_counting(False)
                    s_.bh_set_domain_view(0)
                    s_.build_octree()
                    vs = s_.bh_domain_stats()
                    n0 = vs["bodies"][0]
                    _, _, order_ = s_.morton_keys(with_depth=False)
                    mine_ = order_[:n0]
                    acc_v = s_.accel_bh()
                    s_.set_kernel_timing(True)
                    s_.bh_forces_ranked(0, n0)
                    s_.kernel_time(bh=True)
                    for _ in range(2):
                        s_.bh_forces_ranked(0, n0)
                    wms, _ = s_.kernel_time(bh=True)
                    vs = s_.bh_domain_stats()
                    ex = [vs["export_ms"][k] for k in range(1, g) if vs["bodies"][k]]
                    view = {"domain": 0, "bodies": n0, "own_nodes": vs["nodes"][0], "exported_by_others": vs["exported_nodes"],
                            "received_MB_per_step": sum(vs["exported_nodes"]) * 36 / 1e6, "merged_nodes": vs["merged_nodes"],
                            "own_build_ms": vs["build_ms"][0], "export_ms_each": float(np.mean(ex)) if ex else 0.0,
                            "assemble_ms": vs["merge_ms"] - float(np.sum(ex)), "own_walk_ms": wms,
                            "its_bodies_bit_equal_to_one_tree": bool(np.array_equal(acc_v[mine_], res["one_tree"][1][mine_]))}
        (b0, a0, c0, _), (b1, a1, c1, st) = res["one_tree"], res["domains"]
        same = bool(np.array_equal(a0, a1)) and all(c0[k] == c1[k] for k in
                                                     ("bh_node_visits", "bh_leaf_interactions", "bh_aggregate_interactions"))
        return {"n_bodies": nbodies, "domains": domains, "top_levels": top_levels, "one_tree_build_ms": b0,
                "per_domain_build_ms": st["build_ms"], "slowest_domain_over_one_tree": max(st["build_ms"]) / b0,
                "partition_ms": st["partition_ms"], "merge_ms": st["merge_ms"], "bodies": st["bodies"],
                "top_nodes": st["top_nodes"], "merged_nodes": st["merged_nodes"], "one_tree_nodes": c0["octree_nodes"],
                "imported_nodes": st["imported_nodes"], "imported_MB_per_step": [x * 36 / 1e6 for x in st["imported_nodes"]],
                "results_identical_to_one_tree": same, "essential_tree_view": view,
                "note": "logical domains on ONE GPU, built one after the other; per_domain_build_ms is each domain's own "
                        "gather + build, i.e. what a GPU holding only that domain would spend (projected, not a multi-GPU run); "
                        "essential_tree_view: one rank's part of the distributed step (nbody_step_bh with a communicator), "
                        "rehearsed on this GPU with nbody_bh_set_domain_view"}

    def extra(key, fn):
        """An extra leg must not cost the headline line.  (With several ranks a failure cannot be
        contained -- the others would wait in a collective -- so there it propagates.)"""
        if world > 1:
            out[key] = fn()
            return
        try:
            out[key] = fn()
        except Exception as e:
            out[key] = {"error": f"{type(e).__name__}: {e}"}

    if not args.no_config2 and not args.no_extras and world == 1:
        extra("allpairs_n65536", lambda: {"plummer": small_leg(65536, nb.LAW_PLUMMER, 50),
                                          "wgsl_bias": small_leg(65536, nb.LAW_WGSL_BIAS, 10)})
    if not args.no_config2 and not args.no_extras and world == 1 and args.law == "plummer":
        extra("allpairs_ordered_pair_kernel", lambda: ordered_pair_leg(n, 2))
    if not args.no_extras and args.law == "plummer":
        extra("barnes_hut", lambda: bh_leg(n, 20))
        # the reference's traversal shader applies its acos/pow bias too (src/nbodybh.wgsl:90-101)
        extra("barnes_hut_wgsl_bias_law", lambda: bh_leg(n, 10, nb.LAW_WGSL_BIAS))
        if n == 1 << 20:
            extra("barnes_hut_16m", lambda: bh_leg(1 << 24, 5))
            extra("barnes_hut_16m_wgsl_bias_law", lambda: bh_leg(1 << 24, 3, nb.LAW_WGSL_BIAS))
            if world == 1:
                extra("barnes_hut_16m_domains", lambda: domains_leg(1 << 24, 8, 5))
    if rank == 0 and args.gpus == 1 and not args.no_extras:
        try:
            out["cpu_baseline"] = cpu_baseline()
            if args.law == "plummer":
                out["cpu_baseline_barnes_hut"] = cpu_baseline_bh(masses, pos)
        except Exception as e:  # the oracle is a checker, never a dependency of the timed path
            out.setdefault("cpu_baseline", {"error": str(e)})
            out["cpu_baseline_error"] = str(e)
    if world > 1:
        out["ranks_hold_identical_positions"] = agree
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
