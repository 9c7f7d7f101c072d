"""The C-ABI library loads and exports every symbol include/nbody_hip.h declares (no GPU needed);
host-side generators agree with the numpy restatement."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nbody_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nbody_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from nbody_amd import _lib
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"libnbody_hip.so lacks {s}"
    # the Python binding table covers the header exactly
    assert sorted(_lib.SYMBOLS) == syms


def test_struct_layouts_match_header():
    from nbody_amd import _lib, NODE_DTYPE
    assert NODE_DTYPE.itemsize == 96
    assert [NODE_DTYPE.fields[k][1] for k in ("com", "pos_min", "pos_max", "range", "total_mass", "child", "node_type")] \
        == [0, 16, 32, 44, 48, 52, 84]
    assert C.sizeof(_lib.NBodyConfig) == 80
    assert C.sizeof(_lib.NBodyCounters) == 128
    lib = _lib.load()
    cfg = _lib.NBodyConfig()
    lib.nbody_config_default(C.byref(cfg))
    # constants of the reference: src/nbody.wgsl:13-15, src/nbodybh.wgsl:61, src/octree_maxdepth.rs:7
    assert (cfg.n_bodies, cfg.max_depth) == (250, 16)
    assert cfg.G == pytest.approx(0.0066743) and cfg.dt == pytest.approx(0.1)
    assert cfg.eps2 == 1.0 and cfg.theta == 0.5 and cfg.world == 250.0


def test_create_rejects_bad_config_without_gpu():
    from nbody_amd import _lib
    lib = _lib.load()
    cfg = _lib.NBodyConfig()
    lib.nbody_config_default(C.byref(cfg))
    h = C.c_void_p()
    cfg.n_bodies = 0
    assert lib.nbody_create(C.byref(h), C.byref(cfg)) == -1
    assert b"n_bodies" in lib.nbody_last_error(None)
    cfg.n_bodies = 10
    cfg.max_depth = 17
    assert lib.nbody_create(C.byref(h), C.byref(cfg)) == -1
    cfg.max_depth = 16
    cfg.shard_begin = 8
    cfg.shard_count = 5
    assert lib.nbody_create(C.byref(h), C.byref(cfg)) == -1
    cfg.shard_begin = 2
    cfg.shard_count = 2 ** 64 - 1  # begin + count wraps around to 1
    assert lib.nbody_create(C.byref(h), C.byref(cfg)) == -1
    assert b"shard range" in lib.nbody_last_error(None)
    assert lib.nbody_create(None, None) == -1
    # the allocation-free entry point refuses before it touches a device
    assert lib.nbody_accel_allpairs_noalloc(None, None, 10, 2, 1.0, 1.0, None, None) == -1
    assert b"noalloc" in lib.nbody_last_error(None)
    assert lib.nbody_comm_init(None, None, 0, 1) == -1


def test_generators_match_numpy(orc):
    import nbody_amd as nb
    for seed in (1, 42):
        m, p = nb.gen_refcube(4096, seed=seed)
        m2, p2 = orc.gen_refcube(4096, seed=seed)
        assert np.array_equal(m, m2) and np.array_equal(p, p2)
        assert p[:, :3].min() >= 50 and p[:, :3].max() <= 200 and m.min() >= 0.25 and m.max() <= 64
    # a sub-range equals the slice of the whole (shards generate their own bodies)
    m, p = nb.gen_refcube(100, seed=9, first=1000)
    m2, p2 = nb.gen_refcube(1100, seed=9)
    assert np.array_equal(m, m2[1000:]) and np.array_equal(p, p2[1000:])
    m, p, v = nb.gen_plummer(20000, seed=42)
    m2, p2 = orc.gen_plummer(20000, seed=42)
    assert np.array_equal(m, m2)
    # f64 libm vs numpy may differ in the last bit before the f32 cast
    assert np.abs(p - p2).max() <= 2e-5
    r = np.linalg.norm(p[:, :3] - 125.0, axis=1)
    assert r.max() <= 100.001 and np.all(v == 0)
    # half-mass radius of a Plummer sphere is 1.305 a (slightly less with the cut at 8a)
    assert 1.2 * 12.5 < np.median(r) < 1.35 * 12.5
    assert m.sum() == pytest.approx(4515.625, rel=1e-4)
    m, p, v = nb.gen_plummer(20000, seed=42, equilibrium=True)
    speed = np.linalg.norm(v[:, :3], axis=1)
    vesc = np.sqrt(2 * 0.0066743 * 4515.625 / 12.5) * (1 + (r / 12.5) ** 2) ** -0.25
    assert np.all(speed <= vesc * 1.0001) and speed.mean() > 0.2 * vesc.mean()


def test_rust_bindings_cover_the_header():
    """rust/nbody-hip-sys cannot be compiled in this image; at least it must bind every function
    of the header, each with the header's number of parameters, and mirror the struct fields."""
    text = open(os.path.join(ROOT, "include", "nbody_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    rust = open(os.path.join(ROOT, "rust", "nbody-hip-sys", "src", "lib.rs")).read()
    rust = re.sub(r"//[^\n]*", "", rust)

    def nparams(args):
        args = args.strip()
        return 0 if args in ("", "void") else args.count(",") + 1

    header = {m.group(1): nparams(m.group(2)) for m in re.finditer(r"\b(nbody_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text)}
    bound = {m.group(1): nparams(m.group(2)) for m in re.finditer(r"pub fn (nbody_[a-z0-9_]+)\s*\(([^)]*)\)", rust)}
    assert set(header) == set(declared_symbols())
    assert bound == header
    # struct fields in declaration order (the C names, except G which Rust spells g)
    cfg_c = re.search(r"typedef struct nbody_config \{(.*?)\} nbody_config;", text, re.S).group(1)
    cfg_fields = [f.lower() for f in re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*;", cfg_c)]
    cfg_rust = re.search(r"pub struct nbody_config \{(.*?)\}", rust, re.S).group(1)
    assert re.findall(r"pub ([a-z0-9_]+):", cfg_rust) == cfg_fields
    for name in ("NBODY_LAW_PLUMMER", "NBODY_INT_LEAPFROG", "NBODY_BUF_ACC_PARTIAL", "NBODY_ERR_NOMEM", "NBODY_NODE_INTERIOR"):
        c_val = int(re.search(name + r"\s*=\s*(-?\d+)", text).group(1))
        r_val = int(re.search(r"pub const " + name + r": [a-z0-9_]+ = (-?\d+);", rust).group(1))
        assert c_val == r_val, name


def test_test_transport_exports_what_the_library_binds():
    """tests/transport_standin/libshm_transport.so (test infrastructure: the collectives of several ranks
    on one GPU) must offer every RCCL entry point nbody_amd/csrc/comm.hip looks up."""
    path = os.path.join(ROOT, "tests", "transport_standin", "libshm_transport.so")
    if not os.path.exists(path):
        pytest.fail("the test transport is not built (run __graft_entry__.build())")
    lib = C.CDLL(path)
    src = open(os.path.join(ROOT, "nbody_amd", "csrc", "comm.hip")).read()
    bound = sorted(set(re.findall(r'"(nccl[A-Za-z]+)"', src)))
    assert len(bound) == 9, bound
    for name in bound:
        assert hasattr(lib, name), name
