"""The multi-GPU path on CPU: world_size-2 (and 3, uneven) gloo processes run the sharded driver
(nbody_amd.shard.ShardedSim) with a test-only backend that computes with the CPU oracle.  What is
under test is the product's exchange logic: shard ranges, the per-step all-gather of the packed
positions, and that sharded stepping reproduces unsharded stepping bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class OracleShardBackend:
    """Test stand-in for HipShardBackend: same interface, oracle arithmetic on host tensors."""

    def __init__(self, n, begin, count, law, pair_sharding=False, bh_ranked=False):
        from oracle import oracle as orc
        self.orc, self.n, self.b, self.c, self.law = orc, n, begin, count, law
        self.pair_sharding = pair_sharding
        self.bh_ranked = bh_ranked
        self.partial = torch.zeros(n, 4)
        self.full = torch.zeros(n, 4)          # packed {x,y,z,m}: the buffer the ranks all-gather
        self.vel = np.zeros((n, 4), np.float32)
        self.acc = np.zeros((n, 4), np.float32)

    def upload(self, masses, pos16, vel16=None):
        self.m = np.asarray(masses, np.float32)
        f = self.full.numpy()
        f[:, :3] = pos16[:, :3]
        f[:, 3] = self.m
        if vel16 is not None:
            self.vel[:] = vel16

    def posm(self):
        return self.full

    # pair-sharded scheme: this part's contribution to every body's acceleration.  Any partition
    # of the pair set whose parts sum to the whole exercises the exchange; here part p takes the
    # sources j with j % nparts == p.
    def forces_partial(self, part, nparts):
        f = self.full.numpy()
        pos = np.zeros((self.n, 4), np.float32)
        pos[:, :3] = f[:, :3]
        sel = np.arange(part, self.n, nparts)
        m = np.zeros_like(self.m)
        m[sel] = self.m[sel]                 # the other sources weigh nothing in this part
        a = self.orc.accel_allpairs(self.law, pos, self.vel, m)
        self.partial.numpy()[:] = a

    def acc_partial(self):
        return self.partial

    def integrate_partial(self):
        orc, s = self.orc, slice(self.b, self.b + self.c)
        f = self.full.numpy()
        pos = np.zeros((self.n, 4), np.float32)
        pos[:, :3] = f[:, :3]
        a = self.partial.numpy()
        po, vo, ao = orc.integrate_wgsl(pos[s], self.vel[s], self.acc[s], a[s])
        f[s, :3] = po[:, :3]
        self.vel[s], self.acc[s] = vo, ao

    # replicated-state Barnes-Hut: accelerations of the bodies at sorted ranks [b, b + c) into
    # partial[rank]; after the all-gather every rank integrates ALL bodies
    def bh_forces_ranked(self, begin=None, count=None):
        if begin is not None:
            self.b, self.c = begin, count
        orc = self.orc
        f = self.full.numpy()
        pos = np.zeros((self.n, 4), np.float32)
        pos[:, :3] = f[:, :3]
        self.order = np.argsort(orc.morton_keys(pos), kind="stable")
        nodes, _, _, _ = orc.build_octree(pos, self.m)
        a, _ = orc.bh_accel(self.law, pos, self.vel, nodes)
        own = slice(self.b, self.b + self.c)
        self.partial.numpy()[own] = a[self.order[own]]

    def bh_cost(self):
        """A made-up cost for the balancing test: ranks in the first half of the sorted order are
        three times as expensive as the rest."""
        r = np.arange(self.b, self.b + self.c)
        return float(np.where(r < self.n // 2, 3.0, 1.0).sum())

    def integrate_ranked(self):
        orc = self.orc
        f = self.full.numpy()
        pos = np.zeros((self.n, 4), np.float32)
        pos[:, :3] = f[:, :3]
        a = np.zeros((self.n, 4), np.float32)
        a[self.order] = self.partial.numpy()
        po, vo, ao = orc.integrate_wgsl(pos, self.vel, self.acc, a, clamp_world=250.0)
        f[:, :3] = po[:, :3]
        self.vel[:], self.acc[:] = vo, ao

    def step(self, kind):
        orc, b, c = self.orc, self.b, self.c
        f = self.full.numpy()
        pos = np.zeros((self.n, 4), np.float32)
        pos[:, :3] = f[:, :3]
        if kind == "bh":
            nodes, _, _, _ = orc.build_octree(pos, self.m)          # replicated tree
            a, _ = orc.bh_accel(self.law, pos, self.vel, nodes, i_begin=b, i_count=c)
            clamp = 250.0
        else:
            a = orc.accel_allpairs(self.law, pos, self.vel, self.m, i_begin=b, i_count=c)
            clamp = 0.0
        s = slice(b, b + c)
        po, vo, ao = orc.integrate_wgsl(pos[s], self.vel[s], self.acc[s], a[s], clamp_world=clamp)
        f[s, :3] = po[:, :3]          # only the own range changes; the exchange brings the rest
        self.vel[s], self.acc[s] = vo, ao


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, law, kind, steps, out_dir, pair=False, ranked=False, balance=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nbody_amd.shard import ShardedSim, shard_range
        from oracle import oracle as orc
        orc.set_num_threads(2)
        m, pos = orc.gen_refcube(n, seed=8)
        vel = np.zeros((n, 4), np.float32)
        vel[:, :3] = 0.2 * (orc.gen_refcube(n, seed=9)[1][:, :3] / 125.0 - 1.0)
        b, c = shard_range(n, rank, world)
        be = OracleShardBackend(n, b, c, law, pair_sharding=pair, bh_ranked=ranked)
        be.upload(m, pos, vel)
        sim = ShardedSim(be, n, bh_balance_every=balance)
        assert (sim.begin, sim.count) == (b, c)
        (sim.step_bh if kind == "bh" else sim.step_allpairs)(steps)
        np.save(os.path.join(out_dir, f"posm_{rank}.npy"), be.full.numpy())
        np.save(os.path.join(out_dir, f"vel_{rank}.npy"), be.vel[b:b + c])
        np.save(os.path.join(out_dir, f"ranges_{rank}.npy"), np.array(sim.bh_ranges))
    finally:
        dist.destroy_process_group()


def _run(world, n, law, kind, steps, tmp_path, pair=False, ranked=False, balance=0):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, law, kind, steps, str(tmp_path), pair, ranked, balance), nprocs=world, join=True)
    from nbody_amd.shard import shard_range
    full = [np.load(tmp_path / f"posm_{r}.npy") for r in range(world)]
    vel = np.zeros((n, 4), np.float32)
    for r in range(world):
        b, c = shard_range(n, r, world)
        vel[b:b + c] = np.load(tmp_path / f"vel_{r}.npy")
    return full, vel


def test_shard_ranges_partition():
    from nbody_amd.shard import shard_range
    for n, w in ((10, 3), (1 << 20, 8), (7, 8), (1000, 1)):
        r = [shard_range(n, k, w) for k in range(w)]
        assert r[0][0] == 0 and sum(c for _, c in r) == n
        for (b0, c0), (b1, _) in zip(r, r[1:]):
            assert b0 + c0 == b1
        assert max(c for _, c in r) - min(c for _, c in r) <= 1


@pytest.mark.parametrize("world,n", [(2, 512), (3, 301)])
def test_sharded_allpairs_equals_unsharded(orc, tmp_path, world, n):
    steps, law = 3, orc.LAW_WGSL_BIAS
    full, vel = _run(world, n, law, "allpairs", steps, tmp_path)
    m, pos = orc.gen_refcube(n, seed=8)
    v0 = np.zeros((n, 4), np.float32)
    v0[:, :3] = 0.2 * (orc.gen_refcube(n, seed=9)[1][:, :3] / 125.0 - 1.0)
    rp, rv, ra = orc.step_wgsl_allpairs(law, m, pos, v0, np.zeros_like(pos), steps)
    for f in full:                       # every rank ends with every body's position and mass
        assert np.array_equal(f[:, :3], rp[:, :3]) and np.array_equal(f[:, 3], m)
    assert np.array_equal(vel, rv)


def test_sharded_bh_equals_unsharded(orc, tmp_path):
    world, n, steps, law = 2, 400, 2, orc.LAW_PLUMMER
    full, vel = _run(world, n, law, "bh", steps, tmp_path)
    m, pos = orc.gen_refcube(n, seed=8)
    v0 = np.zeros((n, 4), np.float32)
    v0[:, :3] = 0.2 * (orc.gen_refcube(n, seed=9)[1][:, :3] / 125.0 - 1.0)
    rp, rv, ra = orc.step_bh(law, m, pos, v0, np.zeros_like(pos), steps)
    for f in full:
        assert np.array_equal(f[:, :3], rp[:, :3])
    assert np.array_equal(vel, rv)


@pytest.mark.parametrize("world,n", [(2, 400), (3, 301)])
def test_ranked_bh_equals_unsharded(orc, tmp_path, world, n):
    """Replicated state, traversal split by sorted rank, accelerations all-gathered: every rank
    ends with the unsharded run's state, bit for bit."""
    steps, law = 2, orc.LAW_WGSL_BIAS
    full, vel = _run(world, n, law, "bh", steps, tmp_path, ranked=True)
    m, pos = orc.gen_refcube(n, seed=8)
    v0 = np.zeros((n, 4), np.float32)
    v0[:, :3] = 0.2 * (orc.gen_refcube(n, seed=9)[1][:, :3] / 125.0 - 1.0)
    rp, rv, ra = orc.step_bh(law, m, pos, v0, np.zeros_like(pos), steps)
    for f in full:
        assert np.array_equal(f[:, :3], rp[:, :3])
    assert np.array_equal(vel, rv)      # assembled from each rank's own slice of its replicated copy


def test_balance_ranges():
    from nbody_amd.shard import balance_ranges, shard_range
    n = 1 << 20
    eq = [shard_range(n, r, 4) for r in range(4)]
    assert balance_ranges(eq, [1.0, 1.0, 1.0, 1.0], n) == eq
    assert balance_ranges(eq, [0.0, 0.0, 0.0, 0.0], n) == eq            # nothing measured yet
    got = balance_ranges(eq, [2.0, 2.0, 1.0, 1.0], n)
    assert got == [(0, 196608), (196608, 196608), (393216, 262144), (655360, 393216)]
    for ranges, costs, total in ((eq, [5.0, 0.1, 0.1, 7.0], n), ([(0, 100), (100, 0), (100, 200)], [1.0, 0.0, 4.0], 300)):
        new = balance_ranges(ranges, costs, total, quantum=1 if total == 300 else 64)
        assert new[0][0] == 0 and sum(c for _, c in new) == total
        for (b0, c0), (b1, _) in zip(new, new[1:]):
            assert b0 + c0 == b1 and c0 >= 0
    # a second round on the re-measured costs of a piecewise-constant density leaves the boundaries alone
    dens = lambda b, c: 2.0 * max(0, min(b + c, n // 2) - b) + 1.0 * max(0, b + c - max(b, n // 2))
    r1 = balance_ranges(eq, [dens(b, c) for b, c in eq], n)
    r2 = balance_ranges(r1, [dens(b, c) for b, c in r1], n)
    assert r1 == r2 and abs(dens(*r1[0]) - dens(*r1[3])) <= 2 * 64


def test_ranked_bh_with_rebalancing_equals_unsharded(orc, tmp_path):
    """Boundaries that move between steps (uneven runs, the padded all-gather) change nothing in
    the result; all ranks derive the same boundaries."""
    world, n, steps, law = 3, 700, 3, orc.LAW_PLUMMER
    full, vel = _run(world, n, law, "bh", steps, tmp_path, ranked=True, balance=1)
    m, pos = orc.gen_refcube(n, seed=8)
    v0 = np.zeros((n, 4), np.float32)
    v0[:, :3] = 0.2 * (orc.gen_refcube(n, seed=9)[1][:, :3] / 125.0 - 1.0)
    rp, rv, ra = orc.step_bh(law, m, pos, v0, np.zeros_like(pos), steps)
    for f in full:
        assert np.array_equal(f[:, :3], rp[:, :3])
    ranges = [np.load(tmp_path / f"ranges_{r}.npy") for r in range(world)]
    assert all(np.array_equal(ranges[0], x) for x in ranges)
    from nbody_amd.shard import shard_range
    assert ranges[0].tolist() != [list(shard_range(n, r, world)) for r in range(world)]     # they did move
    assert ranges[0][:, 1].sum() == n and (ranges[0][1:, 0] % 64 == 0).all()
    # the expensive half got the shorter runs
    assert ranges[0][0, 1] < ranges[0][2, 1]


def test_pair_sharded_allpairs_matches_unsharded(orc, tmp_path):
    """The two-collective scheme (all-reduce of partial accelerations, then all-gather of
    positions): the parts' sums differ from the unsharded sum only by f32 summation order."""
    world, n, steps, law = 2, 384, 3, orc.LAW_PLUMMER
    full, vel = _run(world, n, law, "allpairs", steps, tmp_path, pair=True)
    m, pos = orc.gen_refcube(n, seed=8)
    v0 = np.zeros((n, 4), np.float32)
    v0[:, :3] = 0.2 * (orc.gen_refcube(n, seed=9)[1][:, :3] / 125.0 - 1.0)
    rp, rv, ra = orc.step_wgsl_allpairs(law, m, pos, v0, np.zeros_like(pos), steps)
    assert np.array_equal(full[0], full[1])           # both ranks hold the same gathered state
    assert np.abs(full[0][:, :3] - rp[:, :3]).max() < 1e-5
    assert np.abs(vel - rv).max() < 1e-7
