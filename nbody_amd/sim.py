"""Host-side mirror of the reference's three binaries' step blocks, over the C ABI.

`NBodySim` owns one nbody_ctx.  Method names follow the reference's structure:
  step_allpairs  <-> Event::MainEventsCleared of src/nbody_gpu.rs:418-451 / src/nbody_cpu.rs:319-360
  step_bh        <-> Event::MainEventsCleared of src/nbody_gpu_bh.rs:445-527
  build_octree   <-> OctreeNode::new_tree, src/octree_maxdepth.rs:26
Arrays are numpy float32 with the reference's 16-byte vec3 stride: shape (N, 4).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import NBodyConfig, NBodyCounters

LAW_CPU_R6, LAW_WGSL_BIAS, LAW_PLUMMER = 0, 1, 2
INT_CPU_EULER_XBUF, INT_WGSL_VERLET, INT_LEAPFROG = 0, 1, 2
BUF_POSM, BUF_VEL, BUF_ACC, BUF_OCTREE, BUF_ACC_PARTIAL = 0, 1, 2, 3, 4
NODE_DUMMY, NODE_LEAFBODY, NODE_LEAFLIST, NODE_INTERIOR = 0, 1, 2, 3

# the 96-byte node of src/nbodybh.wgsl:1-9 as a numpy record
NODE_DTYPE = np.dtype({
    "names": ["com", "pos_min", "pos_max", "range", "total_mass", "child", "node_type", "depth"],
    "formats": [(np.float32, 3), (np.float32, 3), (np.float32, 3), np.float32, np.float32, (np.uint32, 8), np.uint32,
                np.uint32],
    "offsets": [0, 16, 32, 44, 48, 52, 84, 88],
    "itemsize": 96,
})


class NBodyError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"nbody_hip error {code}: {msg}")
        self.code = code


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32x4(a, n, name):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.shape != (n, 4):
        raise ValueError(f"{name} must have shape ({n}, 4), got {a.shape}")
    return a


def default_config(n_bodies, **kw):
    lib = _lib.load()
    cfg = NBodyConfig()
    lib.nbody_config_default(C.byref(cfg))
    cfg.n_bodies = n_bodies
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise TypeError(f"unknown config field {k}")
        setattr(cfg, k, v)
    return cfg


def gen_refcube(n, seed=42, world=250.0, first=0):
    """Masses (n,) and positions (n,4) following src/nbody_gpu.rs:61-81 with a seeded stream."""
    lib = _lib.load()
    m = np.empty(n, np.float32)
    p = np.empty((n, 4), np.float32)
    rc = lib.nbody_gen_refcube(seed, first, n, world, _ptr(m), _ptr(p))
    if rc:
        raise NBodyError(rc, "nbody_gen_refcube")
    return m, p


def gen_plummer(n, seed=42, world=250.0, a=12.5, rmax=100.0, total_mass=4515.625, G=0.0066743, equilibrium=False,
                first=0, n_total=None):
    lib = _lib.load()
    m = np.empty(n, np.float32)
    p = np.empty((n, 4), np.float32)
    v = np.empty((n, 4), np.float32)
    rc = lib.nbody_gen_plummer(seed, first, n, n_total or n, world, a, rmax, total_mass, G, int(equilibrium), _ptr(m),
                               _ptr(p), _ptr(v))
    if rc:
        raise NBodyError(rc, "nbody_gen_plummer")
    return m, p, v


def accel_allpairs_noalloc(dev_posm, dev_vel16, n_bodies, law, G, eps2, dev_acc16_out, stream=None):
    """nbody_accel_allpairs_noalloc on raw device addresses (ints): allocates nothing, does not block."""
    lib = _lib.load()
    rc = lib.nbody_accel_allpairs_noalloc(dev_posm, dev_vel16, n_bodies, law, G, eps2, dev_acc16_out, stream)
    if rc:
        raise NBodyError(rc, lib.nbody_last_error(None).decode())


def comm_unique_id():
    """128 bytes identifying a new RCCL communicator (call on one rank, hand them to the others)."""
    lib = _lib.load()
    buf = C.create_string_buffer(128)
    rc = lib.nbody_comm_unique_id(buf)
    if rc:
        raise NBodyError(rc, lib.nbody_last_error(None).decode())
    return buf.raw


class NBodySim:
    def __init__(self, n_bodies, **cfg):
        self.lib = _lib.load()
        self.cfg = default_config(n_bodies, **cfg)
        self.n = n_bodies
        h = C.c_void_p()
        rc = self.lib.nbody_create(C.byref(h), C.byref(self.cfg))
        if rc:
            raise NBodyError(rc, self.lib.nbody_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.nbody_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        if rc:
            raise NBodyError(rc, self.lib.nbody_last_error(self.h).decode())

    # -- buffers -------------------------------------------------------------------------------
    def upload(self, masses, pos16, vel16=None, acc16=None):
        m = np.ascontiguousarray(masses, dtype=np.float32)
        if m.shape != (self.n,):
            raise ValueError("masses must have shape (n,)")
        p, v, a = (_f32x4(x, self.n, s) for x, s in ((pos16, "pos16"), (vel16, "vel16"), (acc16, "acc16")))
        self._ck(self.lib.nbody_upload(self.h, _ptr(m), _ptr(p), _ptr(v), _ptr(a)))

    def upload_alt(self, pos16=None, vel16=None, acc16=None):
        p, v, a = (_f32x4(x, self.n, s) for x, s in ((pos16, "pos16"), (vel16, "vel16"), (acc16, "acc16")))
        self._ck(self.lib.nbody_upload_alt(self.h, _ptr(p), _ptr(v), _ptr(a)))

    def download(self, alt=False):
        p, v, a = (np.empty((self.n, 4), np.float32) for _ in range(3))
        fn = self.lib.nbody_download_alt if alt else self.lib.nbody_download
        self._ck(fn(self.h, _ptr(p), _ptr(v), _ptr(a)))
        return p, v, a

    @staticmethod
    def comm_unique_id():
        return comm_unique_id()

    def comm_init(self, unique_id, rank, world):
        """Attach an RCCL communicator (collective over the ranks): step_allpairs / step_bh then exchange
        inside the library."""
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        self._ck(self.lib.nbody_comm_init(self.h, C.create_string_buffer(unique_id, 128), rank, world))

    # -- steps ---------------------------------------------------------------------------------
    def step_allpairs(self, nsteps=1):
        self._ck(self.lib.nbody_step_allpairs(self.h, nsteps))

    def step_bh(self, nsteps=1):
        self._ck(self.lib.nbody_step_bh(self.h, nsteps))

    def allpairs_partial(self, part, nparts):
        self._ck(self.lib.nbody_allpairs_partial(self.h, part, nparts))

    def integrate_partial(self):
        self._ck(self.lib.nbody_integrate_partial(self.h))

    def bh_forces_ranked(self, rank_begin, rank_count):
        """Octree build + traversal for the bodies at sorted ranks [rank_begin, rank_begin + rank_count):
        accelerations to BUF_ACC_PARTIAL[rank]."""
        self._ck(self.lib.nbody_bh_forces_ranked(self.h, rank_begin, rank_count))

    def integrate_ranked(self):
        """BUF_ACC_PARTIAL (by sorted rank, all N) -> body order, integrate + clamp all bodies, swap."""
        self._ck(self.lib.nbody_integrate_ranked(self.h))

    def accel_allpairs(self):
        a = np.zeros((self.n, 4), np.float32)
        self._ck(self.lib.nbody_accel_allpairs(self.h, _ptr(a)))
        return a

    def accel_bh(self):
        a = np.zeros((self.n, 4), np.float32)
        self._ck(self.lib.nbody_accel_bh(self.h, _ptr(a)))
        return a

    def sync(self):
        self._ck(self.lib.nbody_sync(self.h))

    # -- octree --------------------------------------------------------------------------------
    def build_octree(self):
        self._ck(self.lib.nbody_build_octree(self.h))

    def download_octree(self):
        cnt = C.c_uint64()
        self._ck(self.lib.nbody_download_octree(self.h, None, 0, C.byref(cnt)))
        nodes = np.zeros(cnt.value, NODE_DTYPE)
        self._ck(self.lib.nbody_download_octree(self.h, _ptr(nodes), cnt.value, C.byref(cnt)))
        return nodes

    def upload_octree(self, nodes):
        nodes = np.ascontiguousarray(nodes)
        if nodes.dtype.itemsize != 96:
            raise ValueError("octree nodes must be 96-byte records")
        self._ck(self.lib.nbody_upload_octree(self.h, _ptr(nodes), len(nodes)))

    def morton_keys(self, with_depth=True):
        keys = np.empty(self.n, np.uint64)
        depth = np.empty(self.n, np.uint8) if with_depth else None  # (not kept for a domain-decomposed tree)
        order = np.empty(self.n, np.uint32)
        self._ck(self.lib.nbody_morton_keys(self.h, _ptr(keys), _ptr(depth), _ptr(order)))
        return keys, depth, order

    def bh_set_target_order(self, order):
        """Grouping of the traversal's targets into wavefronts (64 consecutive entries each): body indices,
        0xffffffff = empty lane; None restores the library's own order.  Results do not depend on it."""
        if order is None:
            self._ck(self.lib.nbody_bh_set_target_order(self.h, None, 0))
            return
        order = np.ascontiguousarray(order, dtype=np.uint32)
        self._ck(self.lib.nbody_bh_set_target_order(self.h, _ptr(order), len(order)))

    def bh_set_domains(self, domains, top_levels=4):
        """Domain-decomposed octree build (domains <= 1: off): every domain's tree from its own bodies,
        the cells above depth top_levels rebuilt from the domains' cells.  Results do not change."""
        self._ck(self.lib.nbody_bh_set_domains(self.h, domains, top_levels))

    def bh_set_domain_view(self, domain):
        """One-GPU rehearsal of the essential-tree exchange: later builds assemble the merged tree as the GPU
        of `domain` would hold it (its own tree whole, of the others what they would send); only that domain's
        bodies are served by it.  domain < 0: off."""
        self._ck(self.lib.nbody_bh_set_domain_view(self.h, domain))

    def bh_domain_stats(self):
        st = _lib.NBodyDomainStats()
        self._ck(self.lib.nbody_bh_domain_stats(self.h, C.byref(st)))
        g = st.domains
        per_domain = ("bodies", "nodes", "imported_nodes", "build_ms", "exported_nodes", "export_ms")
        out = {f: getattr(st, f) for f, _ in _lib.NBodyDomainStats._fields_ if f not in per_domain}
        for f in per_domain:
            out[f] = list(getattr(st, f))[:g]
        return out

    # -- measurement ---------------------------------------------------------------------------
    def energy(self):
        k, u = C.c_double(), C.c_double()
        self._ck(self.lib.nbody_energy(self.h, C.byref(k), C.byref(u)))
        return k.value, u.value

    def counters(self):
        c = NBodyCounters()
        self._ck(self.lib.nbody_counters(self.h, C.byref(c)))
        return {f: getattr(c, f) for f, _ in NBodyCounters._fields_ if f != "reserved"}

    def counters_reset(self):
        self._ck(self.lib.nbody_counters_reset(self.h))

    def set_counting(self, on):
        self._ck(self.lib.nbody_set_counting(self.h, int(on)))

    def set_kernel_timing(self, on):
        self._ck(self.lib.nbody_set_kernel_timing(self.h, int(on)))

    def kernel_time(self, bh=False, build=False):
        """(average ms, launches) since the last call: the all-pairs force kernel, with bh the
        traversal kernel, with build one whole octree build."""
        ms, cnt = C.c_double(), C.c_uint64()
        self._ck(self.lib.nbody_kernel_time(self.h, 2 if build else int(bool(bh)), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def kernel_clock(self):
        """Shader clock (MHz) held during the last pair-kernel launch made with kernel timing on (0: none)."""
        mhz = C.c_double()
        self._ck(self.lib.nbody_kernel_clock(self.h, C.byref(mhz)))
        return mhz.value

    def device_buffer(self, which):
        p, b = C.c_void_p(), C.c_size_t()
        self._ck(self.lib.nbody_device_buffer(self.h, which, C.byref(p), C.byref(b)))
        return p.value, b.value

    def read_buffer(self, which):
        """Host copy of one of the f32[N][4] device buffers (BUF_POSM/VEL/ACC/ACC_PARTIAL)."""
        out = np.empty((self.n, 4), np.float32)
        self._ck(self.lib.nbody_read_buffer(self.h, which, _ptr(out), out.nbytes))
        return out

    def stream(self):
        return self.lib.nbody_stream(self.h)
