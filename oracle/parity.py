"""Error statistics of computed accelerations against the oracle, for the parity tests and for
tools/parity_report.py.  TEST INFRASTRUCTURE ONLY (like everything under oracle/): never imported
by the nbody_amd package."""
import numpy as np

from . import oracle as orc


def slices_of(n, superblock=4096, tile=64, width=1024):
    """Target ranges that between them cover the structures of the pair kernel: the first and the
    last superblock, a superblock boundary, a source-tile boundary in the middle, and the tail of
    the reduction (the last, possibly partial, superblock)."""
    mid = (n // 2) // tile * tile
    cand = [(0, width), (superblock - width // 2, width), (mid - width // 2, width), (max(0, n - width), width)]
    out = []
    for b, c in cand:
        b = max(0, min(b, n - 1))
        c = min(c, n - b)
        if c > 0:
            out.append((b, c))
    return out


def allpairs_stats(law, m, pos, vel, acc, ranges, G=0.0066743, eps2=1.0):
    """Relative errors ||a - ref|| / ||ref|| of acc (all bodies) over the given target ranges,
    against the f64 evaluation and against the f32 sum in the reference's order."""
    e64, e32, eref, bound_ok = [], [], [], []
    for b, c in ranges:
        r64 = orc.accel_allpairs_f64(law, pos, vel, m, G=G, eps2=eps2, i_begin=b, i_count=c)
        r32 = orc.accel_allpairs(law, pos, vel, m, G=G, eps2=eps2, i_begin=b, i_count=c)[b:b + c, :3].astype(np.float64)
        floor = 4 * np.finfo(np.float32).eps * orc.accel_allpairs_abs(law, pos, vel, m, G=G, eps2=eps2, i_begin=b, i_count=c)
        a = acc[b:b + c, :3].astype(np.float64)
        scale = np.linalg.norm(r64, axis=1)
        d64 = np.linalg.norm(a - r64, axis=1)
        dref = np.linalg.norm(r32 - r64, axis=1)
        e64.append(d64 / scale)
        eref.append(dref / scale)
        e32.append(np.linalg.norm(a - r32, axis=1) / np.linalg.norm(r32, axis=1))
        bound_ok.append(d64 <= 1e-5 * scale + dref + floor + 1e-30)
    e64, e32, eref = (np.concatenate(x) for x in (e64, e32, eref))
    q = lambda x: {"max": float(x.max()), "p999": float(np.quantile(x, 0.999)), "median": float(np.median(x))}
    return {"targets": int(len(e64)), "vs_f64": q(e64), "vs_f32_reference_order": q(e32),
            "f32_reference_order_vs_f64": q(eref), "within_bound": bool(np.concatenate(bound_ok).all())}
