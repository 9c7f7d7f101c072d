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

    def bh_forces_ranked(self):
        self.sim.bh_forces_ranked(self.begin, self.count)

    def integrate_ranked(self):
        self.sim.integrate_ranked()

    def download(self):
        return self.sim.download()

    def sync(self):
        self.sim.sync()


class ShardedSim:
    def __init__(self, backend, n, rank=None, world=None, group=None):
        self.backend = backend
        self.n = n
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.begin, self.count = shard_range(n, self.rank, self.world)
        self._nccl = dist.get_backend(group) == "nccl" if dist.is_initialized() else False

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
        self._gather_ranges(self.backend.posm())

    def _gather_ranges(self, full):
        """full[begin_r : begin_r + count_r] of every rank r reaches all ranks (in place)."""
        if self.n % self.world == 0:
            # a private copy of the own range (16 B/body): input and output never alias
            own = full[self.begin:self.begin + self.count].clone()
            if full.is_cuda and not self._nccl:
                # device tensors over a host-only backend (tests: gloo): stage through the host
                host = torch.empty(full.shape, dtype=full.dtype)
                dist.all_gather_into_tensor(host, own.cpu(), group=self.group)
                full.copy_(host)
            else:
                dist.all_gather_into_tensor(full, own, group=self.group)
        else:
            for r in range(self.world):
                b, c = shard_range(self.n, r, self.world)
                src = dist.get_global_rank(self.group, r) if self.group else r
                part = full[b:b + c]
                if full.is_cuda and not self._nccl:
                    host = part.cpu()
                    dist.broadcast(host, src=src, group=self.group)
                    if r != self.rank:
                        part.copy_(host)
                else:
                    dist.broadcast(part, src=src, group=self.group)

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
        for _ in range(nsteps):
            if ranked and self.world > 1:
                self.backend.bh_forces_ranked()
                with self._stream():
                    self._gather_ranges(self.backend.acc_partial())
                self.backend.integrate_ranked()
            else:
                self.backend.step("bh")
                self.exchange()
