"""nbody_amd: the gravity step of JMS55/nbody (all-pairs and Barnes-Hut) on MI355X.

The compute path is libnbody_hip.so (hand-written HIP for gfx950) behind the C ABI of
include/nbody_hip.h; this package is the thin host side.  There is no CPU fallback.
"""
from .sim import (NBodySim, accel_allpairs_noalloc, comm_unique_id, NBodyError, default_config, gen_refcube, gen_plummer, NODE_DTYPE, LAW_CPU_R6,
                  LAW_WGSL_BIAS, LAW_PLUMMER, INT_CPU_EULER_XBUF, INT_WGSL_VERLET, INT_LEAPFROG, BUF_POSM, BUF_VEL, BUF_ACC,
                  BUF_OCTREE, BUF_ACC_PARTIAL, NODE_DUMMY, NODE_LEAFBODY, NODE_LEAFLIST, NODE_INTERIOR)

__all__ = [n for n in dir() if not n.startswith("_")]
