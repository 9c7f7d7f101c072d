"""The in-library RCCL exchange (nbody_comm_init): with a communicator attached the step entry
points run their collectives themselves.  A gpurun box has one GPU, so the communicator has one
rank; every code path (pair-dealt all-pairs with its all-reduce, position gather, ranked Barnes-Hut
with its gather) still runs through RCCL and must reproduce the single-context results."""
import numpy as np
import pytest

import nbody_amd as nb

pytestmark = pytest.mark.gpu


def _run(n, law, steps, bh, comm, seed=3, **cfg):
    m, p = nb.gen_refcube(n, seed=seed)
    rng = np.random.default_rng(seed)
    v = np.zeros((n, 4), np.float32)
    v[:, :3] = 0.3 * rng.normal(size=(n, 3)).astype(np.float32)
    with nb.NBodySim(n, law=law, **cfg) as sim:
        if comm:
            sim.comm_init(nb.comm_unique_id(), 0, 1)
        sim.upload(m, p, v)
        (sim.step_bh if bh else sim.step_allpairs)(steps)
        out = sim.download()
        c = sim.counters()
    return out, c


@pytest.mark.parametrize("law", [nb.LAW_WGSL_BIAS, nb.LAW_CPU_R6])
def test_allpairs_target_shard_path_with_communicator(law):
    (p0, v0, a0), _ = _run(5000, law, 3, False, False)
    (p1, v1, a1), c = _run(5000, law, 3, False, True)
    assert np.isfinite(p0).all()
    assert np.array_equal(p0, p1) and np.array_equal(v0, v1) and np.array_equal(a0, a1)
    assert c["steps_allpairs"] == 3


def test_allpairs_pair_dealt_path_with_communicator():
    """PLUMMER from N = 65,536: nbody_allpairs_partial(rank, world) + all-reduce + own integration."""
    n = 70001
    (p0, v0, a0), c0 = _run(n, nb.LAW_PLUMMER, 2, False, False)
    (p1, v1, a1), c1 = _run(n, nb.LAW_PLUMMER, 2, False, True)
    assert c0["allpairs_path"] == 2 and c1["allpairs_path"] == 2
    scale = np.linalg.norm(a0[:, :3], axis=1)
    err = np.linalg.norm(a1[:, :3] - a0[:, :3], axis=1)
    assert (err <= 2e-6 * scale + 1e-12).all(), float((err / scale).max())
    assert np.allclose(p0, p1, rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("law", [nb.LAW_PLUMMER, nb.LAW_WGSL_BIAS])
def test_bh_ranked_path_with_communicator_is_bit_exact(law):
    (p0, v0, a0), _ = _run(20000, law, 3, True, False)
    (p1, v1, a1), c = _run(20000, law, 3, True, True)
    assert np.array_equal(p0, p1) and np.array_equal(v0, v1) and np.array_equal(a0, a1)
    assert c["steps_bh"] == 3


def test_communicator_misuse_is_refused():
    with nb.NBodySim(1000, shard_begin=0, shard_count=400) as sim:
        with pytest.raises(nb.NBodyError) as e:  # one rank whose range does not cover the bodies
            sim.comm_init(nb.comm_unique_id(), 0, 1)
        assert e.value.code == -1
    with nb.NBodySim(1000) as sim:
        sim.comm_init(nb.comm_unique_id(), 0, 1)
        with pytest.raises(nb.NBodyError) as e:
            sim.comm_init(nb.comm_unique_id(), 0, 1)
        assert e.value.code == -4
        with pytest.raises(nb.NBodyError):
            sim.comm_init(nb.comm_unique_id(), 1, 1)
