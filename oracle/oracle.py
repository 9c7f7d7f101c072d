"""ctypes wrapper of the CPU oracle (oracle/liboracle.so) plus a numpy restatement of the dataset
generators.  TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the nbody_amd package."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

LAW_CPU_R6, LAW_WGSL_BIAS, LAW_PLUMMER = 0, 1, 2
T_DUMMY, T_LEAFBODY, T_LEAFLIST, T_INTERIOR = 0, 1, 2, 3

NODE_DTYPE = np.dtype({
    "names": ["com", "pos_min", "pos_max", "range", "total_mass", "child", "node_type", "max_depth"],
    "formats": [(np.float32, 3), (np.float32, 3), (np.float32, 3), np.float32, np.float32, (np.uint32, 8), np.uint32,
                np.uint32],
    "offsets": [0, 16, 32, 44, 48, 52, 84, 88],
    "itemsize": 96,
})

_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_morton_key.restype = C.c_uint64
        _lib.orc_morton_key.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32]
        _lib.orc_build_octree.restype = C.c_int64
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt=np.float32):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


def accel_allpairs(law, pos16, vel16, masses, G=0.0066743, eps2=1.0, i_begin=0, i_count=None, accum64=False):
    pos16, vel16, masses = _c(pos16), _c(vel16), _c(masses)
    n = len(masses)
    i_count = n - i_begin if i_count is None else i_count
    out = np.zeros((n, 4), np.float32)
    lib().orc_accel_allpairs(C.c_int(law), C.c_int64(n), _p(pos16), _p(vel16), _p(masses), C.c_float(G), C.c_float(eps2),
                             C.c_int64(i_begin), C.c_int64(i_count), C.c_int(int(accum64)), _p(out))
    return out


def accel_allpairs_f64(law, pos16, vel16, masses, G=0.0066743, eps2=1.0, i_begin=0, i_count=None):
    pos16, vel16, masses = _c(pos16), _c(vel16), _c(masses)
    n = len(masses)
    i_count = n - i_begin if i_count is None else i_count
    out = np.zeros((i_count, 3), np.float64)
    lib().orc_accel_allpairs_f64(C.c_int(law), C.c_int64(n), _p(pos16), _p(vel16), _p(masses), C.c_double(float(np.float32(G))),
                                 C.c_double(eps2), C.c_int64(i_begin), C.c_int64(i_count), _p(out))
    return out


def accel_allpairs_abs(law, pos16, vel16, masses, G=0.0066743, eps2=1.0, i_begin=0, i_count=None):
    """Per body, the sum of the norms of its source terms (float64): the rounding-error scale."""
    pos16, vel16, masses = _c(pos16), _c(vel16), _c(masses)
    n = len(masses)
    i_count = n - i_begin if i_count is None else i_count
    out = np.zeros(i_count, np.float64)
    lib().orc_accel_allpairs_abs(C.c_int(law), C.c_int64(n), _p(pos16), _p(vel16), _p(masses), C.c_double(float(np.float32(G))),
                                 C.c_double(eps2), C.c_int64(i_begin), C.c_int64(i_count), _p(out))
    return out


def step_cpu(masses, p1, v1, a1, p2, v2, a2, nsteps):
    """src/nbody_cpu.rs:321-360, nsteps times.  Returns the six arrays after the steps; the first
    three are the set the next step would read."""
    arrs = [np.array(_c(x), copy=True) for x in (p1, v1, a1, p2, v2, a2)]
    masses = _c(masses)
    lib().orc_step_cpu(C.c_int64(len(masses)), C.c_int(nsteps), _p(masses), *[_p(a) for a in arrs])
    return arrs


def step_wgsl_allpairs(law, masses, pos, vel, acc, nsteps, G=0.0066743, eps2=1.0, dt=0.1):
    pos, vel, acc = (np.array(_c(x), copy=True) for x in (pos, vel, acc))
    masses = _c(masses)
    lib().orc_step_wgsl_allpairs(C.c_int(law), C.c_int64(len(masses)), C.c_int(nsteps), _p(masses), C.c_float(G),
                                 C.c_float(eps2), C.c_float(dt), _p(pos), _p(vel), _p(acc))
    return pos, vel, acc


def integrate_wgsl(pos, vel, acc_in, acc_new, dt=0.1, clamp_world=0.0):
    pos, vel, acc_in, acc_new = (_c(x) for x in (pos, vel, acc_in, acc_new))
    n = len(pos)
    po, vo, ao = (np.zeros((n, 4), np.float32) for _ in range(3))
    lib().orc_integrate_wgsl(C.c_int64(n), C.c_float(dt), C.c_float(clamp_world), _p(pos), _p(vel), _p(acc_in), _p(acc_new),
                             _p(po), _p(vo), _p(ao))
    return po, vo, ao


def build_octree(pos16, masses, variant=0, world=250.0, max_depth=16, leaflist_mode=0):
    """Returns (nodes, body_leaf, body_depth, max_depth_value)."""
    pos16, masses = _c(pos16), _c(masses)
    n = len(masses)
    cap = 4 * n + 64
    while True:
        nodes = np.zeros(cap, NODE_DTYPE)
        leaf = np.zeros(n, np.uint32)
        dep = np.zeros(n, np.uint8)
        md = C.c_uint32(0)
        cnt = lib().orc_build_octree(C.c_int(variant), C.c_int64(n), _p(pos16), _p(masses), C.c_float(world),
                                     C.c_uint32(max_depth), C.c_int(leaflist_mode), _p(nodes), C.c_int64(cap), _p(leaf),
                                     _p(dep), C.byref(md))
        if cnt == -1:
            cap *= 4
            continue
        if cnt == -3:
            raise ValueError("the octree.rs variant is not restated for bodies of mass 0 (src/octree.rs:49 takes their node for an empty one)")
        if cnt < 0:
            raise RuntimeError(f"oracle octree build failed: {cnt}")
        # byte-wise copy: numpy does not promise to copy the padding bytes of a structured dtype
        raw = nodes.view(np.uint8).reshape(-1, 96)[:cnt].copy()
        return raw.view(NODE_DTYPE).reshape(-1), leaf, dep, md.value


def morton_keys(pos16, world=250.0, levels=16):
    pos16 = _c(pos16)
    keys = np.zeros(len(pos16), np.uint64)
    lib().orc_morton_keys(C.c_int64(len(pos16)), _p(pos16), C.c_float(world), C.c_uint32(levels), _p(keys))
    return keys


def bh_accel(law, pos16, vel16, nodes, G=0.0066743, eps2=1.0, theta=0.5, i_begin=0, i_count=None, stack_cap=1 << 20):
    """src/nbodybh.wgsl:63-161 per body.  Returns (acc16, counters dict)."""
    pos16, vel16 = _c(pos16), _c(vel16)
    nodes = np.ascontiguousarray(nodes)
    n = len(pos16)
    i_count = n - i_begin if i_count is None else i_count
    out = np.zeros((n, 4), np.float32)
    cnt = np.zeros(4, np.int64)
    rc = lib().orc_bh_accel(C.c_int(law), C.c_int64(n), _p(pos16), _p(vel16), _p(nodes), C.c_float(G), C.c_float(eps2),
                            C.c_float(theta), C.c_int64(i_begin), C.c_int64(i_count), C.c_int64(stack_cap), _p(out), _p(cnt))
    if rc:
        raise RuntimeError(f"oracle traversal failed: {rc}")
    return out, dict(visits=int(cnt[0]), leaf=int(cnt[1]), aggregate=int(cnt[2]), max_stack=int(cnt[3]))


def bh_accel_f64(law, pos16, vel16, nodes, G=0.0066743, eps2=1.0, theta=0.5, i_begin=0, i_count=None, with_abs=False):
    """The same node sequence as bh_accel, terms evaluated and summed in float64.  with_abs: also the sum
    of the terms' magnitudes per body (the scale of any f32 summation order's rounding error)."""
    pos16, vel16 = _c(pos16), _c(vel16)
    nodes = np.ascontiguousarray(nodes)
    n = len(pos16)
    i_count = n - i_begin if i_count is None else i_count
    out = np.zeros((i_count, 3), np.float64)
    mag = np.zeros(i_count, np.float64) if with_abs else None
    lib().orc_bh_accel_f64(C.c_int(law), C.c_int64(n), _p(pos16), _p(vel16), _p(nodes), C.c_double(float(np.float32(G))),
                           C.c_double(eps2), C.c_float(theta), C.c_int64(i_begin), C.c_int64(i_count), _p(out), _p(mag))
    return (out, mag) if with_abs else out


def step_bh(law, masses, pos, vel, acc, nsteps, G=0.0066743, eps2=1.0, dt=0.1, theta=0.5, world=250.0, max_depth=16,
            leaflist_mode=0):
    pos, vel, acc = (np.array(_c(x), copy=True) for x in (pos, vel, acc))
    masses = _c(masses)
    rc = lib().orc_step_bh(C.c_int(law), C.c_int64(len(masses)), C.c_int(nsteps), _p(masses), C.c_float(G), C.c_float(eps2),
                           C.c_float(dt), C.c_float(theta), C.c_float(world), C.c_uint32(max_depth), C.c_int(leaflist_mode),
                           _p(pos), _p(vel), _p(acc))
    if rc:
        raise RuntimeError(f"oracle BH step failed: {rc}")
    return pos, vel, acc


def energy(pos16, vel16, masses, G=0.0066743, eps2=1.0):
    pos16, vel16, masses = _c(pos16), _c(vel16), _c(masses)
    k, u = C.c_double(), C.c_double()
    lib().orc_energy(C.c_int64(len(masses)), _p(pos16), _p(vel16), _p(masses), C.c_double(float(np.float32(G))),
                     C.c_double(eps2), C.byref(k), C.byref(u))
    return k.value, u.value


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(t):
    lib().orc_set_num_threads(C.c_int(t))


# ---- dataset generators, numpy restatement (checks the library's C++ generators) ---------------
_M64 = (1 << 64) - 1


def _mix64(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return x


def draw(seed, idx, k):
    """Draw k of bodies idx: uniform [0,1) with 24 bits, float64."""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = np.uint64(seed) + (np.uint64(64) * idx + np.uint64(k + 1)) * np.uint64(0x9E3779B97F4A7C15)
        x = _mix64(x)
    return (x >> np.uint64(40)).astype(np.float64) / 16777216.0


def gen_refcube(n, seed=42, world=250.0, first=0):
    idx = np.arange(first, first + n, dtype=np.uint64)
    lo = np.float32(world) / np.float32(5.0)
    span = np.float32(4.0) * lo - lo
    u = [draw(seed, idx, k).astype(np.float32) for k in range(5)]
    pos = np.zeros((n, 4), np.float32)
    for c in range(3):
        pos[:, c] = lo + span * u[c]
    m = (np.float32(0.5) + np.float32(7.5) * u[3]) * (np.float32(0.5) + np.float32(7.5) * u[4])
    return m.astype(np.float32), pos


def gen_plummer(n, seed=42, world=250.0, a=12.5, rmax=100.0, total_mass=4515.625, first=0, n_total=None):
    n_total = n_total or n
    idx = np.arange(first, first + n, dtype=np.uint64)
    rad = np.full(n, float(a))
    done = np.zeros(n, bool)
    for k in range(20):
        u = draw(seed, idx, k)
        with np.errstate(divide="ignore", invalid="ignore"):
            t = np.power(u, -2.0 / 3.0) - 1.0
            cand = float(a) / np.sqrt(t)
        ok = (~done) & (u > 0) & (t > 0) & (cand <= rmax)
        rad[ok] = cand[ok]
        done |= ok
    z = 1.0 - 2.0 * draw(seed, idx, 20)
    phi = 2.0 * np.pi * draw(seed, idx, 21)
    s = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    c = 0.5 * world
    pos = np.zeros((n, 4), np.float32)
    pos[:, 0] = (c + rad * s * np.cos(phi)).astype(np.float32)
    pos[:, 1] = (c + rad * s * np.sin(phi)).astype(np.float32)
    pos[:, 2] = (c + rad * z).astype(np.float32)
    m = np.full(n, np.float32(total_mass) / np.float32(n_total), np.float32)
    return m, pos
