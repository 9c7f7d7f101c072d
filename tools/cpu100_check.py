"""Configuration 1 at its own size on the HIP path: 100 steps of the nbody_cpu.rs law and update at
N = 4,096 against the committed oracle state (tests/golden/refcube_n4096_s1.npz): how the differences
are distributed (the unsoftened r^-5 law ejects bodies through close encounters, which amplify
rounding differences without bound).  usage: python tools/cpu100_check.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd as nb
g = np.load(os.path.join(ROOT, "tests", "golden", "refcube_n4096_s1.npz"))
m, pos = g["masses"], g["pos"]
n = len(m)
with nb.NBodySim(n, law=0, integrator=0) as sim:
    sim.upload(m, pos)
    sim.upload_alt(pos)
    sim.step_allpairs(100)
    p, v, a = sim.download()
ref = g["cpu100_pos"]
d = np.linalg.norm((p - ref)[:, :3].astype(np.float64), axis=1)
disp = np.linalg.norm((ref - pos)[:, :3].astype(np.float64), axis=1)
rel = d / np.maximum(disp, 1e-30)
q = lambda x: {k: float(np.quantile(x, f)) for k, f in (("median", .5), ("p90", .9), ("p99", .99), ("p999", .999), ("max", 1.0))}
print(json.dumps({"abs": q(d), "relative_to_displacement": q(rel), "displacement": q(disp),
                  "within_1e-3_abs": float((d < 1e-3).mean()), "within_1e-4_rel": float((rel < 1e-4).mean()),
                  "finite": bool(np.isfinite(p).all())}))
