"""The arithmetic behind the in-library exchange (nbody_amd/csrc/comm.hip), without a device: which
run of the sorted order a rank traverses, and how a set of ranges is exchanged (one in-place
all-gather needs begin[k] == k * count[0]; unequal or empty ranges go by broadcasts)."""
import ctypes as C

import numpy as np
import pytest


def _lib():
    from nbody_amd import _lib
    return _lib.load()


def sorted_runs(n, world):
    lib = _lib()
    out = []
    for k in range(world):
        b, c = C.c_uint64(), C.c_uint64()
        assert lib.nbody_comm_sorted_run(n, world, k, C.byref(b), C.byref(c)) == 0
        out.append((b.value, c.value))
    return out


def plan(n, ranges):
    lib = _lib()
    b = (C.c_uint64 * len(ranges))(*[r[0] for r in ranges])
    c = (C.c_uint64 * len(ranges))(*[r[1] for r in ranges])
    return lib.nbody_comm_exchange_plan(n, len(ranges), b, c)


ALLGATHER, BROADCASTS, REPLICATED, INVALID = 1, 2, 3, -1


@pytest.mark.parametrize("n", [1, 63, 64, 65, 300, 511, 512, 4096, 100037, 1 << 20, (1 << 24) + 777])
@pytest.mark.parametrize("world", [1, 2, 3, 7, 8])
def test_sorted_runs_tile_the_order_in_whole_wavefronts(n, world):
    runs = sorted_runs(n, world)
    nxt = 0
    for k, (b, c) in enumerate(runs):
        assert b == nxt                      # contiguous, in rank order
        nxt = b + c
        if k + 1 < world:
            assert (b + c) % 64 == 0         # a boundary never splits a wavefront's 64 targets
            assert b + c == (n * (k + 1) // world) // 64 * 64
    assert nxt == n
    p = plan(n, runs)
    counts = {c for _, c in runs}
    if len(counts) == 1:
        assert p == ALLGATHER
        assert all(b == k * runs[0][1] for k, (b, _) in enumerate(runs))   # what the in-place all-gather assumes
    else:
        assert p == BROADCASTS


def test_small_inputs_leave_ranks_without_targets():
    runs = sorted_runs(300, 8)
    assert [c for _, c in runs] == [0, 64, 0, 64, 0, 64, 64, 44]
    assert plan(300, runs) == BROADCASTS
    assert sorted_runs(10, 4) == [(0, 0), (0, 0), (0, 0), (0, 10)]


def test_exchange_plan_classifies_ranges():
    assert plan(1000, [(0, 250), (250, 250), (500, 250), (750, 250)]) == ALLGATHER
    assert plan(1001, [(0, 251), (251, 250), (501, 250), (751, 250)]) == BROADCASTS
    assert plan(10, [(0, 10)]) == ALLGATHER
    assert plan(10, [(0, 10), (0, 10), (0, 10)]) == REPLICATED
    assert plan(10, [(0, 5), (5, 0), (5, 5)]) == BROADCASTS            # an empty rank in the middle
    assert plan(10, [(0, 5), (6, 4)]) == INVALID                        # gap
    assert plan(10, [(0, 6), (5, 5)]) == INVALID                        # overlap
    assert plan(10, [(5, 5), (0, 5)]) == INVALID                        # out of rank order
    assert plan(10, [(0, 5), (5, 4)]) == INVALID                        # does not reach n
    assert plan(10, [(0, 5), (5, 2 ** 64 - 1)]) == INVALID              # begin + count wraps
    lib = _lib()
    b, c = C.c_uint64(), C.c_uint64()
    assert lib.nbody_comm_sorted_run(10, 0, 0, C.byref(b), C.byref(c)) == -1
    assert lib.nbody_comm_sorted_run(10, 2, 2, C.byref(b), C.byref(c)) == -1


def test_python_shard_ranges_agree_with_the_plan():
    """nbody_amd.shard.shard_range (the host-side split bench.py and the headless host use for
    all-pairs targets) always gives ranges the library accepts."""
    from nbody_amd.shard import shard_range
    for n in (8, 9, 1000, 1001, 1 << 20, 70001):
        for world in (1, 2, 3, 8):
            ranges = [shard_range(n, r, world) for r in range(world)]
            assert plan(n, ranges) == (ALLGATHER if n % world == 0 else BROADCASTS)
