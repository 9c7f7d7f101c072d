"""Target-sharded stepping across the GPUs of one node (one process per GPU).

The reference is single-device.  Here the targets are split into contiguous equal ranges, one
per rank; every rank integrates its own bodies against ALL sources, then the ranks all-gather
the packed {x,y,z,m} positions (16 bytes per body) so that each holds the sources of the next
step.  Velocities and accelerations never leave their rank: the force laws only need the
target's own velocity (src/nbody.wgsl:74-76).

Barnes-Hut uses REPLICATED state instead (`HipShardBackend(..., bh_ranked=True)`): every rank
holds all N bodies and builds the same octree; rank r traverses the r-th run of the SORTED
(Morton) order, a spatially compact set of targets whose lanes share their tree paths; the ranks
all-gather the accelerations in sorted order (16 bytes per body, the same volume as positions
would be) and every rank integrates all bodies, reaching bit-identical states.  Splitting the
traversal by body index instead (the all-pairs split) costs 1.7x more traversal time per rank at
8 ranks, because a rank's 64-lane groups then span 8x the volume.

The exchange is one collective per step (`all_gather_into_tensor`, RCCL over xGMI on GPUs); the
compute backend is injected so that the exchange logic can be exercised with gloo on CPU.

PLUMMER all-pairs at large N uses a second scheme, because the single-GPU kernel evaluates each
unordered pair once and a target-sharded rank could not: the PAIR TILES of all bodies are dealt
out over the ranks, each rank produces its contribution to every body's acceleration
(16 bytes per body), the ranks all-reduce that buffer, integrate their own targets and
all-gather the positions as before: two collectives per step.
"""
import contextlib

import numpy as np
import torch
import torch.distributed as dist


def balance_ranges(ranges, costs, n, quantum=64):
    """Contiguous ranges of equal estimated cost.  `costs[r]` is what rank r measured for its
    current range `ranges[r] = (begin, count)`; within a range the cost is taken as evenly spread.
    Boundaries are rounded to `quantum` (64 keeps every wavefront's 64 targets together, so the
    traversal results do not depend on where the boundaries fall)."""
    world = len(ranges)
    total = float(sum(costs))
    segs = [(b, c, float(t)) for (b, c), t in zip(ranges, costs) if c > 0 and t > 0]
    if total <= 0 or not segs:
        return [shard_range(n, r, world) for r in range(world)]
    total = sum(t for _, _, t in segs)
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        acc, x = 0.0, float(n)
        for b, c, t in segs:
            if acc + t >= target:
                x = b + (target - acc) / t * c
                break
            acc += t
        xi = int(round(x / quantum)) * quantum
        bounds.append(min(max(xi, bounds[-1]), n))
    bounds.append(n)
    return [(bounds[r], bounds[r + 1] - bounds[r]) for r in range(world)]


def shard_range(n, rank, world):
    """Contiguous near-equal split: the first n % world ranks get one body more."""
    base, extra = divmod(n, world)
    begin = rank * base + min(rank, extra)
    return begin, base + (1 if rank < extra else 0)


class _DevicePtr:
    """Exposes a raw device pointer to torch through __cuda_array_interface__."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n, 4), "typestr": "<f4", "data": (ptr, False), "version": 3,
                                         "strides": None}


class HipShardBackend:
    """One rank's bodies on one MI355X through the C ABI (nbody_amd.NBodySim)."""

    PAIR_SHARD_MIN_N = 65536

    def __init__(self, n, begin, count, device_index, bh_ranked=False, **cfg):
        from .sim import NBodySim, BUF_POSM, BUF_ACC_PARTIAL, LAW_PLUMMER, INT_WGSL_VERLET, INT_CPU_EULER_XBUF
        torch.cuda.set_device(device_index)
        self.bh_ranked = bh_ranked
        self.begin, self.count = begin, count
        if bh_ranked:  # replicated state: the context holds (and integrates) all bodies
            cfg = dict(cfg, shard_begin=0, shard_count=n)
        else:
            cfg = dict(cfg, shard_begin=begin, shard_count=count)
        self._buf_posm = BUF_POSM
        self._buf_partial = BUF_ACC_PARTIAL
        self.pair_sharding = (cfg.get("law") == LAW_PLUMMER and cfg.get("integrator", INT_WGSL_VERLET) != INT_CPU_EULER_XBUF
                              and n >= self.PAIR_SHARD_MIN_N)
        # One explicit stream carries the library's kernels AND the collectives' stream
        # dependencies.  (A null stream handle would not do: the library takes it as "create a
        # stream of your own", which torch then knows nothing about.)
        self.torch_stream = torch.cuda.Stream(device=device_index)
        self.sim = NBodySim(n, device=device_index, stream=self.torch_stream.cuda_stream, **cfg)
        self.n = n
        self._views = {}
        self._timing = False

    def upload(self, masses, pos16, vel16=None):
        self.sim.upload(masses, pos16, vel16)

    def step(self, kind):
        (self.sim.step_bh if kind == "bh" else self.sim.step_allpairs)(1)

    def _view(self, which):
        ptr, _ = self.sim.device_buffer(which)
        if ptr not in self._views:
            self._views[ptr] = torch.as_tensor(_DevicePtr(ptr, self.n), device="cuda")
        return self._views[ptr]

    def posm(self):
        return self._view(self._buf_posm)

    def forces_partial(self, part, nparts):
        self.sim.allpairs_partial(part, nparts)

    def acc_partial(self):
        return self._view(self._buf_partial)

    def integrate_partial(self):
        self.sim.integrate_partial()

    def bh_forces_ranked(self, begin=None, count=None):
        self.sim.bh_forces_ranked(self.begin if begin is None else begin, self.count if count is None else count)

    def bh_cost(self):
        """Average traversal-kernel time (ms) of this rank since the last call; blocks on the stream."""
        if not self._timing:
            self.sim.set_kernel_timing(True)
            self._timing = True
            return 0.0
        return self.sim.kernel_time(bh=True)[0]

    def integrate_ranked(self):
        self.sim.integrate_ranked()

    def download(self):
        return self.sim.download()

    def sync(self):
        self.sim.sync()


class ShardedSim:
    def __init__(self, backend, n, rank=None, world=None, group=None, bh_balance_every=0, library_comm=False):
        """library_comm: the exchange runs inside the library (nbody_comm_init: RCCL on the context's
        stream); this class then only hands the communicator's id round and forwards the steps.
        bh_balance_every = K > 0 (replicated-state Barnes-Hut only): every K steps the ranks
        exchange their measured traversal times and move the boundaries of their runs of the
        sorted order so that the times even out (for bodies that are not spread symmetrically
        over the Morton order; a centred Plummer sphere is balanced within 2 % as it is)."""
        self.backend = backend
        self.n = n
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.begin, self.count = shard_range(n, self.rank, self.world)
        self._nccl = dist.get_backend(group) == "nccl" if dist.is_initialized() else False
        self.ranges = [shard_range(n, r, self.world) for r in range(self.world)]
        self.bh_ranges = list(self.ranges)       # runs of the sorted order, one per rank
        self.bh_balance_every = bh_balance_every
        self._bh_steps = 0
        self._stage = None
        self.library_comm = library_comm
        if library_comm:
            if bh_balance_every:
                raise ValueError("the in-library exchange splits the sorted order evenly: no bh_balance_every")
            # one rank makes the id, the host side (this process group) carries it to the others
            ids = [backend.sim.comm_unique_id() if self.rank == 0 else None]
            if self.world > 1:
                dist.broadcast_object_list(ids, src=0, group=group)
            backend.sim.comm_init(ids[0], self.rank, self.world)

    def _stream(self):
        """Collectives and tensor ops must be enqueued on the stream the backend computes on."""
        s = getattr(self.backend, "torch_stream", None)
        return torch.cuda.stream(s) if s is not None else contextlib.nullcontext()

    def exchange(self):
        """All-gather the current positions: every rank's own range reaches all the others."""
        if self.world == 1:
            return
        with self._stream():
            self._exchange()

    def _exchange(self):
        self._gather_ranges(self.backend.posm(), self.ranges)

    def _all_gather(self, out, own):
        """all_gather_into_tensor; device tensors over a host-only backend (tests: gloo) are staged
        through the host."""
        if out.is_cuda and not self._nccl:
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host, own.cpu(), group=self.group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, own, group=self.group)

    def _gather_ranges(self, full, ranges):
        """full[begin_r : begin_r + count_r] of every rank r reaches all ranks (in place).  One
        equal-sized all-gather in every case: ranks with unequal counts contribute a window of
        max(count) rows that contains their range."""
        counts = [c for _, c in ranges]
        b, c = ranges[self.rank]
        if min(counts) == max(counts):
            # a private copy of the own range (16 B/body): input and output never alias
            self._all_gather(full, full[b:b + c].clone())
            return
        cmax = max(counts)
        window = lambda br: min(br, self.n - cmax)       # start of the cmax-row window of a range
        if self._stage is None or self._stage.shape[0] != self.world * cmax or self._stage.device != full.device:
            self._stage = torch.empty((self.world * cmax,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
        self._all_gather(self._stage, full[window(b):window(b) + cmax])
        for r, (br, cr) in enumerate(ranges):
            if r == self.rank or cr == 0:
                continue
            off = r * cmax + (br - window(br))
            full[br:br + cr].copy_(self._stage[off:off + cr])

    def _sum_partials(self):
        part = self.backend.acc_partial()
        if part.is_cuda and not self._nccl:
            host = part.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            part.copy_(host)
        else:
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)

    def step_allpairs(self, nsteps=1):
        if self.world > 1 and getattr(self.backend, "bh_ranked", False):
            raise ValueError("a bh_ranked backend holds all bodies on every rank: it is for step_bh only")
        if self.library_comm:
            self.backend.sim.step_allpairs(nsteps)
            return
        pair = self.world > 1 and getattr(self.backend, "pair_sharding", False)
        for _ in range(nsteps):
            if pair:
                # every rank: its share of the pair tiles -> contributions to ALL bodies; sum them
                self.backend.forces_partial(self.rank, self.world)
                with self._stream():
                    self._sum_partials()
                self.backend.integrate_partial()
            else:
                self.backend.step("allpairs")
            self.exchange()

    def step_bh(self, nsteps=1):
        ranked = getattr(self.backend, "bh_ranked", False)
        if self.library_comm:
            if not ranked:
                raise ValueError("the in-library Barnes-Hut exchange needs a bh_ranked backend (every rank holds all bodies)")
            self.backend.sim.step_bh(nsteps)
            return
        for _ in range(nsteps):
            if ranked and self.world > 1:
                if self.bh_balance_every and self._bh_steps % self.bh_balance_every == 0:
                    self._rebalance_bh()
                self.backend.bh_forces_ranked(*self.bh_ranges[self.rank])
                with self._stream():
                    self._gather_ranges(self.backend.acc_partial(), self.bh_ranges)
                self.backend.integrate_ranked()
                self._bh_steps += 1
            else:
                self.backend.step("bh")
                self.exchange()

    def _rebalance_bh(self):
        """Every rank learns every rank's traversal time and derives the same new boundaries."""
        cost = float(self.backend.bh_cost())
        dev = "cuda" if (self._nccl and torch.cuda.is_available()) else "cpu"
        with self._stream() if dev == "cuda" else contextlib.nullcontext():
            mine = torch.tensor([cost], dtype=torch.float64, device=dev)
            every = torch.empty(self.world, dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(every, mine, group=self.group)
            costs = every.cpu().tolist()
        if min(costs) > 0:       # (the first call only switches the measurement on)
            self.bh_ranges = balance_ranges(self.bh_ranges, costs, self.n)
