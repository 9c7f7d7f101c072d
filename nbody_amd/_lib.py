"""ctypes binding of libnbody_hip.so (the C ABI declared in include/nbody_hip.h).

The library is the product: there is no Python or CPU fallback.  Importing this module fails
loudly when the shared library has not been built (run `python -c "import __graft_entry__ as g;
g.build()"` or `make -C nbody_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NBODY_HIP_LIB") or os.path.join(_HERE, "libnbody_hip.so")  # the override is for measurement builds


class NBodyConfig(C.Structure):
    _fields_ = [
        ("n_bodies", C.c_uint64),
        ("law", C.c_int32),
        ("integrator", C.c_int32),
        ("G", C.c_float),
        ("dt", C.c_float),
        ("eps2", C.c_float),
        ("theta", C.c_float),
        ("world", C.c_float),
        ("max_depth", C.c_uint32),
        ("leaflist_mode", C.c_int32),
        ("shard_begin", C.c_uint64),
        ("shard_count", C.c_uint64),
        ("device", C.c_int32),
        ("quadrupole", C.c_int32),
        ("stream", C.c_void_p),
    ]


class NBodyCounters(C.Structure):
    _fields_ = [
        ("steps_allpairs", C.c_uint64),
        ("steps_bh", C.c_uint64),
        ("interactions_allpairs", C.c_uint64),
        ("bh_node_visits", C.c_uint64),
        ("bh_leaf_interactions", C.c_uint64),
        ("bh_aggregate_interactions", C.c_uint64),
        ("bh_wave_node_visits", C.c_uint64),
        ("octree_nodes", C.c_uint64),
        ("bh_max_stack", C.c_uint64),
        ("allpairs_path", C.c_uint64),
        ("reserved", C.c_uint64 * 6),
    ]


class NBodyDomainStats(C.Structure):
    _fields_ = [
        ("domains", C.c_uint32),
        ("top_levels", C.c_uint32),
        ("top_entries", C.c_uint64),
        ("top_nodes", C.c_uint64),
        ("merged_nodes", C.c_uint64),
        ("top_nodes_visited", C.c_uint64),
        ("partition_ms", C.c_double),
        ("merge_ms", C.c_double),
        ("bodies", C.c_uint64 * 32),
        ("nodes", C.c_uint64 * 32),
        ("imported_nodes", C.c_uint64 * 32),
        ("build_ms", C.c_double * 32),
        ("view_domain", C.c_int32),
        ("distributed", C.c_int32),
        ("exported_nodes", C.c_uint64 * 32),
        ("export_ms", C.c_double * 32),
    ]


# every symbol include/nbody_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_F = C.POINTER(C.c_float)
SYMBOLS = {
    "nbody_version": (C.c_char_p, []),
    "nbody_last_error": (C.c_char_p, [_P]),
    "nbody_config_default": (None, [C.POINTER(NBodyConfig)]),
    "nbody_create": (C.c_int, [C.POINTER(_P), C.POINTER(NBodyConfig)]),
    "nbody_destroy": (None, [_P]),
    "nbody_upload": (C.c_int, [_P, _P, _P, _P, _P]),
    "nbody_upload_alt": (C.c_int, [_P, _P, _P, _P]),
    "nbody_step_allpairs": (C.c_int, [_P, C.c_int]),
    "nbody_step_bh": (C.c_int, [_P, C.c_int]),
    "nbody_allpairs_partial": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "nbody_integrate_partial": (C.c_int, [_P]),
    "nbody_bh_forces_ranked": (C.c_int, [_P, C.c_uint64, C.c_uint64]),
    "nbody_integrate_ranked": (C.c_int, [_P]),
    "nbody_comm_unique_id": (C.c_int, [_P]),
    "nbody_comm_init": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "nbody_comm_sorted_run": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "nbody_comm_exchange_plan": (C.c_int, [C.c_uint64, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "nbody_accel_allpairs": (C.c_int, [_P, _P]),
    "nbody_accel_allpairs_noalloc": (C.c_int, [_P, _P, C.c_uint64, C.c_int, C.c_float, C.c_float, _P, _P]),
    "nbody_accel_bh": (C.c_int, [_P, _P]),
    "nbody_build_octree": (C.c_int, [_P]),
    "nbody_download_octree": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "nbody_upload_octree": (C.c_int, [_P, _P, C.c_uint64]),
    "nbody_morton_keys": (C.c_int, [_P, _P, _P, _P]),
    "nbody_bh_set_target_order": (C.c_int, [_P, _P, C.c_uint64]),
    "nbody_bh_set_domains": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "nbody_bh_domain_stats": (C.c_int, [_P, C.POINTER(NBodyDomainStats)]),
    "nbody_bh_set_domain_view": (C.c_int, [_P, C.c_int32]),
    "nbody_download": (C.c_int, [_P, _P, _P, _P]),
    "nbody_download_alt": (C.c_int, [_P, _P, _P, _P]),
    "nbody_sync": (C.c_int, [_P]),
    "nbody_energy": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "nbody_counters": (C.c_int, [_P, C.POINTER(NBodyCounters)]),
    "nbody_counters_reset": (C.c_int, [_P]),
    "nbody_set_counting": (C.c_int, [_P, C.c_int]),
    "nbody_device_buffer": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "nbody_read_buffer": (C.c_int, [_P, C.c_int, _P, C.c_size_t]),
    "nbody_stream": (_P, [_P]),
    "nbody_kernel_time": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "nbody_set_kernel_timing": (C.c_int, [_P, C.c_int]),
    "nbody_kernel_clock": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "nbody_gen_refcube": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, _P, _P]),
    "nbody_gen_plummer": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, C.c_float, C.c_float,
                                    C.c_float, C.c_float, C.c_int, _P, _P, _P]),
}

_lib = None


def load():
    """Load libnbody_hip.so and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP library is not built. There is no CPU fallback; "
            "build it with `make -C nbody_amd/csrc` (hipcc, --offload-arch=gfx950).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
