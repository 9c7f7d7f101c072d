"""A long distributed Barnes-Hut run (one domain per rank, essential trees exchanged) against the one-rank,
one-tree run of the same binary: 8 ranks sharing one GPU over the shared-memory test transport, a cold sphere
that collapses, rebounds and reaches the clamp.  usage: python tools/ranks_soak.py [n] [frames] [ranks] [top_levels]"""
import json, os, subprocess, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "host", "nbody_headless")
TRANSPORT = os.path.join(ROOT, "tests", "transport_standin", "libshm_transport.so")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50021
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
ranks = int(sys.argv[3]) if len(sys.argv) > 3 else 8
levels = int(sys.argv[4]) if len(sys.argv) > 4 else 3


def run(tag, extra, env):
    state = f"/tmp/ranks_soak_{tag}.state"
    cmd = [EXE, "--mode", "bh", "--law", "plummer", "--dataset", "plummer", "--n", str(n), "--frames", str(frames),
           "--report", "0", "--seed", "11", "--state", state] + extra
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
    if r.returncode:
        raise SystemExit(f"{tag}: rc {r.returncode}\n{r.stdout}\n{r.stderr}")
    raw = np.fromfile(state, np.uint8)
    os.remove(state)
    return raw[16:].view(np.float32).reshape(3, n, 4), r.stdout.strip().splitlines()[-2:], time.time() - t0


one, l1, t1 = run("one", [], dict(os.environ))
many, l8, t8 = run("many", ["--ranks", str(ranks), "--devices", "1", "--domains", str(levels)],
                   dict(os.environ, NBODY_RCCL_LIB=TRANSPORT))
r = np.linalg.norm(one[0][:, :3] - 125.0, axis=1)
print(json.dumps({"n": n, "frames": frames, "ranks": ranks, "top_levels": levels, "bit_equal": bool(np.array_equal(one, many)),
                  "finite": bool(np.isfinite(many).all()), "median_radius_at_end": float(np.median(r)),
                  "bodies_on_the_clamp": int(((one[0][:, :3] <= 0) | (one[0][:, :3] >= 250)).any(axis=1).sum()),
                  "one_rank": l1, "ranks_run": l8, "wall_s": [round(t1, 1), round(t8, 1)]}))
