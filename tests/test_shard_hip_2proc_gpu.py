"""Two ranks of the sharded driver on ONE real GPU (both processes use device 0; the exchange
runs over gloo, staged through the host, because RCCL refuses two ranks on one device).  Exercises, with the HIP backend
and real inter-process collectives, both all-pairs schemes (target-sharded and pair-sharded) and
the sharded Barnes-Hut step, and compares every rank's bodies with an unsharded run of the same
library: bit-exact for Barnes-Hut (per-body traversal order does not depend on the sharding), to
rounding for all-pairs (the split of the sums follows the launch shape and the rank count)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1])
rank, world = int(sys.argv[2]), int(sys.argv[3])
T0 = time.time()
def note(*a):
    # progress goes to a file when the caller names one (a silent multi-minute test looks hung)
    path = os.environ.get("NBODY_TEST_PROGRESS")
    if path:
        with open(f"{path}.{rank}", "a") as f:
            print(f"[{time.time() - T0:6.1f}s]", *a, file=f, flush=True)
note("start")
import numpy as np, torch, torch.distributed as dist
note("torch imported")
import nbody_amd as nb
from nbody_amd.shard import ShardedSim, HipShardBackend, shard_range
torch.cuda.set_device(0)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[4])
dist.init_process_group("gloo", rank=rank, world_size=world)
note("process group up")

def run(n, law, kind, steps, dataset, ranked=False, balance=0):
    import time
    t0 = time.time()
    try:
        return run_(n, law, kind, steps, dataset, ranked, balance)
    finally:
        note("case", n, law, kind, f"{time.time() - t0:.1f} s")

def run_(n, law, kind, steps, dataset, ranked, balance):
    m, pos, vel = dataset(n)
    ref = nb.NBodySim(n, law=law)
    ref.upload(m, pos, vel)
    (ref.step_bh if kind == "bh" else ref.step_allpairs)(steps)
    rp, rv, ra = ref.download()
    ref.close()
    b, c = shard_range(n, rank, world)
    be = HipShardBackend(n, b, c, 0, law=law, bh_ranked=ranked)
    be.upload(m, pos, vel)
    sim = ShardedSim(be, n, rank, world, bh_balance_every=balance)
    (sim.step_bh if kind == "bh" else sim.step_allpairs)(steps)
    torch.cuda.synchronize()
    if balance:
        # measured traversal times moved the boundaries (or left them); they still tile the order
        r = sim.bh_ranges
        assert r[0][0] == 0 and sum(c for _, c in r) == n and all(b % 64 == 0 for b, _ in r), r
        note("balanced ranges", r)
    p, v, a = be.download()
    pair = be.pair_sharding and kind != "bh"
    be.sim.close()
    own = slice(0, n) if ranked else slice(b, b + c)   # replicated state: every body is this rank's
    if kind != "bh":
        # the summation is split differently (launch shape follows the target count; pair tiles
        # are dealt over the ranks), so equality is to rounding
        scale = np.abs(ra[:, :3]).max(axis=1, keepdims=True)
        err = (np.abs(a[own, :3] - ra[own, :3]) / (scale[own] + 1e-30)).max(axis=1)
        assert np.all(err <= 2e-5), (n, law, kind, "acc", float(err.max()), int((err > 2e-5).sum()), int(np.argmax(err)) + b)
        assert np.allclose(p[:, :3], rp[:, :3], rtol=1e-6, atol=1e-5), (n, law, kind, "gathered pos")
    else:
        assert np.array_equal(a[own], ra[own]), (n, law, kind, "acc")
        assert np.array_equal(v[own], rv[own]), (n, law, kind, "vel")
        assert np.array_equal(p[own, :3], rp[own, :3]), (n, law, kind, "own pos")
        assert np.array_equal(p[:, :3], rp[:, :3]), (n, law, kind, "gathered pos")
    return pair

plummer = lambda n: nb.gen_plummer(n, seed=5)
def refcube(n):
    m, p = nb.gen_refcube(n, seed=5)
    return m, p, None
assert not run(4097, nb.LAW_WGSL_BIAS, "allpairs", 3, refcube)     # ragged split: unequal counts, windowed all-gather
assert not run(8192, nb.LAW_PLUMMER, "allpairs", 3, plummer)       # below the pair-sharding size
assert run(65536, nb.LAW_PLUMMER, "allpairs", 2, plummer)          # pair tiles dealt over the ranks
assert not run(20000, nb.LAW_WGSL_BIAS, "bh", 3, refcube)          # replicated tree, sharded walk
assert not run(30001, nb.LAW_PLUMMER, "bh", 3, plummer)
assert not run(20000, nb.LAW_WGSL_BIAS, "bh", 3, refcube, ranked=True)   # replicated state, traversal split by sorted rank
assert not run(30001, nb.LAW_PLUMMER, "bh", 3, plummer, ranked=True)
assert not run(30001, nb.LAW_PLUMMER, "bh", 7, plummer, ranked=True, balance=2)   # boundaries follow the measured times
dist.barrier()
dist.destroy_process_group()
print("RANK_OK", rank)
'''


def test_two_ranks_share_one_gpu(tmp_path):
    import torch  # noqa: F401  (pages the library in once, before two children do it at the same time)
    script = tmp_path / "two_ranks.py"
    script.write_text(SCRIPT)
    port = str(29600 + os.getpid() % 300)
    env = dict(os.environ, GLOO_SOCKET_IFNAME="lo")  # both ranks are on this host
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", port], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=300))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (so, se) in enumerate(outs):
        assert f"RANK_OK {r}" in so, so[-2000:] + se[-4000:]
