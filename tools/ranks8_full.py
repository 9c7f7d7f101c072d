"""Configurations 3 and 5 in their 8-rank form on ONE GPU, through the C ABI: host/nbody_headless --ranks 8
(one process per rank, the library's own exchange, collectives over the shared-memory test transport of
tests/transport_standin because RCCL refuses two ranks on one device) against the one-rank run of the
same binary.  Prints one JSON line per case.  usage: python tools/ranks8_full.py [outdir]"""
import json, os, subprocess, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "host", "nbody_headless")
TRANSPORT = os.path.join(ROOT, "tests", "transport_standin", "libshm_transport.so")
TMP = sys.argv[1] if len(sys.argv) > 1 else "/tmp"


def run(tag, ranks, args):
    dump = os.path.join(TMP, f"ranks8_{tag}.state")
    env = dict(os.environ)
    cmd = [EXE] + args + ["--report", "0", "--state", dump]
    if ranks > 1:
        env["NBODY_RCCL_LIB"] = TRANSPORT
        cmd += ["--ranks", str(ranks), "--devices", "1"]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    if r.returncode != 0:
        raise SystemExit(f"{tag}: rc {r.returncode}\n{r.stdout}\n{r.stderr}")
    raw = np.fromfile(dump, np.uint8)
    os.remove(dump)
    n = int(raw[:16].view(np.uint32)[1])
    state = raw[16:].view(np.float32).reshape(3, n, 4)          # final positions, velocities, accelerations
    tail = [l for l in r.stdout.strip().splitlines() if not l.startswith("frame ")]
    return state, " | ".join(tail[-2:]), time.time() - t0


CASES = [
    ("allpairs_plummer_1m", ["--mode", "gpu", "--law", "plummer", "--dataset", "plummer", "--n", "1048576", "--frames", "2"], 2e-6),
    ("allpairs_wgsl_bias_100003", ["--mode", "gpu", "--dataset", "plummer", "--n", "100003", "--frames", "2"], 2e-6),
    ("barnes_hut_16m", ["--mode", "bh", "--law", "plummer", "--dataset", "plummer", "--n", "16777216", "--frames", "2"], 0.0),
    ("barnes_hut_wgsl_bias_1m_odd", ["--mode", "bh", "--dataset", "plummer", "--n", "1048577", "--frames", "3"], 0.0),
    # every rank builds its own domain's tree only; essential trees exchanged (one rank: the one tree)
    ("barnes_hut_16m_domains", ["--mode", "bh", "--law", "plummer", "--dataset", "plummer", "--n", "16777216", "--frames", "2"], 0.0, ["--domains", "5"]),
    ("barnes_hut_wgsl_bias_1m_odd_domains", ["--mode", "bh", "--dataset", "plummer", "--n", "1048577", "--frames", "3"], 0.0, ["--domains", "4"]),
]
for case in CASES:
    tag, args, tol = case[:3]
    extra = case[3] if len(case) > 3 else []
    one, line1, t1 = run(tag + "_r1", 1, args)
    many, line8, t8 = run(tag + "_r8", 8, args + extra)
    own = one.shape[1] if "bh" in args[1] else -(-one.shape[1] // 8)   # all-pairs shards keep vel / acc on their rank
    a1, a8 = one[2][:own, :3].astype(np.float64), many[2][:own, :3].astype(np.float64)
    rel = float((np.linalg.norm(a8 - a1, axis=1) / np.linalg.norm(a1, axis=1)).max())
    out = {"case": tag, "args": " ".join(args), "eight_rank_args": " ".join(extra), "ranks": 8, "bit_equal_state": bool(np.array_equal(one[:, :own], many[:, :own]) and np.array_equal(one[0], many[0])),
           "compared": f"positions of all bodies; velocities and accelerations of rank 0's {own} targets",
           "max_rel_acceleration_difference": rel, "max_abs_position_difference": float(np.abs(many[0] - one[0]).max()),
           "tolerance": tol, "ok": bool(np.array_equal(one, many)) if tol == 0.0 else bool(rel <= tol),
           "finite": bool(np.isfinite(many).all()), "one_rank": line1, "eight_ranks": line8,
           "wall_s": [round(t1, 1), round(t8, 1)]}
    print(json.dumps(out), flush=True)
