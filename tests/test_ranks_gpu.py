"""The library's own multi-rank exchange (nbody_comm_init; nbody_amd/csrc/comm.hip) with MORE THAN ONE
rank: `host/nbody_headless --ranks R` forks R processes that share the box's one GPU.  RCCL refuses
two ranks on one device, so the collectives go through tests/transport_standin (test infrastructure
selected with NBODY_RCCL_LIB: host-staged exchange over POSIX shared memory, ordered on the stream);
everything else -- who owns which targets, the in-place all-gather offsets, the grouped broadcasts of
unequal shards, the pair kernel's work dealt over the ranks and its all-reduce, the runs of the
sorted order with empty ranks -- is the product's code, and rank 0's frames must be those of a
one-rank run (Barnes-Hut: bit for bit; all-pairs: to the rounding of a different slicing)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "host", "nbody_headless")
TRANSPORT = os.path.join(ROOT, "tests", "transport_standin", "libshm_transport.so")


def read_state(path):
    """Final positions, velocities and accelerations of a run (--state)."""
    raw = np.fromfile(path, np.uint8)
    assert raw[:4].tobytes() == b"NBS1"
    n = int(raw[:16].view(np.uint32)[1])
    return raw[16:].view(np.float32).reshape(3, n, 4)


def run(tmp_path, tag, ranks, args, timeout=300):
    if not os.path.exists(EXE) or not os.path.exists(TRANSPORT):
        pytest.fail("host/nbody_headless or the test transport is not built (run __graft_entry__.build())")
    state = tmp_path / f"{tag}.state"
    env = dict(os.environ)
    cmd = [EXE] + args + ["--report", "0", "--state", str(state)]
    if ranks > 1:
        env["NBODY_RCCL_LIB"] = TRANSPORT
        env["NBODY_SHM_SLOT_KB"] = "192"    # slots smaller than most messages: the chunked path runs too
        cmd += ["--ranks", str(ranks), "--devices", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    return read_state(state), r.stdout


def assert_close(many, one, ranks, tol=2e-6):
    """All-pairs shards: velocities and accelerations never leave their rank (the laws need only the
    target's own velocity), so rank 0's state holds them for ITS targets, the first n / ranks bodies;
    positions are gathered and compared for all bodies.  Accelerations per body within `tol`."""
    assert np.isfinite(one).all() and np.isfinite(many).all()
    n = one.shape[1]
    own = n // ranks + (1 if n % ranks else 0)
    a1, a8 = one[2][:own, :3].astype(np.float64), many[2][:own, :3].astype(np.float64)
    scale = np.linalg.norm(a1, axis=1)
    err = np.linalg.norm(a8 - a1, axis=1)
    assert scale.min() > 0 and (err <= tol * scale).all(), float((err / scale).max())
    assert np.allclose(many[1][:own], one[1][:own], rtol=4 * tol, atol=1e-9)
    assert np.abs(one[1][:own]).max() > 0
    assert np.allclose(many[0], one[0], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("ranks", [2, 8])
def test_barnes_hut_ranks_reproduce_one_rank_bit_for_bit(tmp_path, ranks):
    """Replicated build, traversal by runs of the sorted order, gather of the accelerations by rank:
    N not a multiple of 64 * ranks (unequal runs -> broadcasts)."""
    args = ["--mode", "bh", "--n", "100037", "--frames", "3", "--seed", "5"]
    one, _ = run(tmp_path, "one", 1, args)
    many, _ = run(tmp_path, f"r{ranks}", ranks, args)
    assert one.shape == (3, 100037, 4) and np.isfinite(one).all() and np.abs(one[2]).max() > 0
    assert np.array_equal(one, many)


def test_barnes_hut_equal_runs_use_the_in_place_all_gather(tmp_path):
    """N = 64 * 8 * k: every run has the same length, so the exchange is ONE in-place all-gather
    at offset rank * count."""
    args = ["--mode", "bh", "--law", "plummer", "--dataset", "plummer", "--n", str(64 * 8 * 200), "--frames", "2",
            "--seed", "42"]
    one, _ = run(tmp_path, "one", 1, args)
    many, _ = run(tmp_path, "r8", 8, args)
    assert np.array_equal(one, many)


def test_barnes_hut_with_fewer_wavefronts_than_ranks(tmp_path):
    """300 bodies over 8 ranks: runs of 0, 64, 0, 64, 0, 64, 64 and 44 targets."""
    args = ["--mode", "bh", "--n", "300", "--frames", "4", "--seed", "9"]
    one, _ = run(tmp_path, "one", 1, args)
    many, _ = run(tmp_path, "r8", 8, args)
    assert np.array_equal(one, many)


@pytest.mark.parametrize("ranks,n,law,dataset,top_levels", [
    (8, 100037, "plummer", "plummer", 3), (8, 60000, "wgsl_bias", "plummer", 4), (2, 30011, "wgsl_bias", "refcube", 1),
    (5, 50000, "plummer", "refcube", 2), (8, 300, "plummer", "refcube", 3)])
def test_barnes_hut_domains_exchange_essential_trees(tmp_path, ranks, n, law, dataset, top_levels):
    """--domains: every rank builds the octree of ITS domain of bodies only; bounding boxes, exported nodes
    and top-tree entries are exchanged (grouped broadcasts of unequal parts), every rank traverses for its
    own domain, the accelerations are exchanged domain by domain.  Rank 0's state is the one-rank,
    one-tree run's, bit for bit (300 bodies over 8 ranks: some domains hold a handful of bodies)."""
    args = ["--mode", "bh", "--law", law, "--dataset", dataset, "--n", str(n), "--frames", "3", "--seed", "7"]
    one, _ = run(tmp_path, "one", 1, args)
    many, out = run(tmp_path, f"d{ranks}", ranks, args + ["--domains", str(top_levels)])
    assert f"domains: {ranks} of depth-{top_levels} cells, distributed" in out
    assert np.isfinite(one).all() and np.abs(one[2]).max() > 0
    assert np.array_equal(one, many)


def test_barnes_hut_domains_over_a_long_run(tmp_path):
    """A cold sphere collapses and rebounds over 300 frames: the domains' populations and shapes change, bodies
    reach the clamp; four ranks, each with its own domain's tree and the others' exports, stay on the
    one-rank run's bits."""
    args = ["--mode", "bh", "--law", "plummer", "--dataset", "plummer", "--n", "20011", "--frames", "300", "--seed", "3"]
    one, _ = run(tmp_path, "one", 1, args)
    many, _ = run(tmp_path, "d4", 4, args + ["--domains", "2"])
    assert np.isfinite(one).all() and np.array_equal(one, many)


def test_one_rank_with_domains_is_the_one_tree(tmp_path):
    args = ["--mode", "bh", "--n", "40000", "--frames", "3", "--seed", "4"]
    one, _ = run(tmp_path, "one", 1, args)
    dom, _ = run(tmp_path, "dom", 1, args + ["--domains", "3"])
    assert np.array_equal(one, dom)


@pytest.mark.parametrize("ranks,n", [(3, 5003), (8, 5003), (4, 4096)])
def test_allpairs_target_shards_wgsl_law(tmp_path, ranks, n):
    """The reference's law, targets split by index range: unequal shards (grouped broadcasts) and
    equal ones (in-place all-gather)."""
    args = ["--mode", "gpu", "--n", str(n), "--frames", "3", "--seed", "3"]
    one, _ = run(tmp_path, "one", 1, args)
    many, _ = run(tmp_path, f"r{ranks}", ranks, args)
    assert_close(many, one, ranks)


def test_allpairs_pair_kernel_dealt_over_the_ranks(tmp_path):
    """PLUMMER from N = 65,536: every rank evaluates its share of the unordered pairs, the
    contributions are all-reduced, every rank integrates its own targets, positions are gathered."""
    args = ["--mode", "gpu", "--law", "plummer", "--dataset", "plummer", "--n", "70001", "--frames", "2", "--seed", "42"]
    one, out1 = run(tmp_path, "one", 1, args)
    many, out8 = run(tmp_path, "r8", 8, args)
    assert "all-pairs path 2" in out1 and "all-pairs path 2" in out8      # the pair kernel on both sides
    assert_close(many, one, 8)


def test_cpu_law_cross_buffer_step_in_shards(tmp_path):
    """src/nbody_cpu.rs's law and update, one frame (as shipped its second frame overflows: the
    second buffer set starts at the origin, src/nbody_cpu.rs:58)."""
    args = ["--mode", "cpu", "--n", "2001", "--frames", "1", "--seed", "2"]
    one, _ = run(tmp_path, "one", 1, args)
    many, _ = run(tmp_path, "r3", 3, args)
    assert_close(many, one, 3)


def test_more_ranks_than_devices_is_refused_without_a_transport(tmp_path):
    env = {k: v for k, v in os.environ.items() if k != "NBODY_RCCL_LIB"}
    r = subprocess.run([EXE, "--mode", "bh", "--n", "1000", "--frames", "1", "--ranks", "2", "--devices", "1"],
                       capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 2 and "one GPU per rank" in r.stderr


def test_a_failing_rank_ends_the_run(tmp_path):
    """A rank that cannot start (here: a transport library that does not exist) must not leave the
    others waiting: the supervisor ends them and reports failure."""
    env = dict(os.environ, NBODY_RCCL_LIB=str(tmp_path / "no_such_library.so"))
    r = subprocess.run([EXE, "--mode", "bh", "--n", "1000", "--frames", "1", "--ranks", "3", "--devices", "1"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 1
    assert not [f for f in os.listdir("/tmp") if f.startswith("nbody_headless_id.")]
