"""Sharded Propagate::run across the GPUs of one node.

The product path is INSIDE the engine (include/mvskit_engine.h, "multi-GPU through the C ABI"): every rank creates its
engine with shard_index = rank / shard_count = world, the ranks share an RCCL communicator (mvs_engine_comm_init) and
mvs_engine_propagate exchanges and commits after each colour pass on the engine's own stream.  `EngineExchange` below
only distributes the communicator id over torch.distributed.  The classes further down stage the exchange through the
host instead: they serve the CPU tests (oracle-backed engines over gloo) and rehearsals in which several ranks share
one GPU, which RCCL refuses.


Every rank holds the whole patch pool and all image pyramids (98 MB for 12 x 1080p), sweeps the views
`rank, rank + world, ...` (mvs_config.view_begin / view_stride) and, after each colour pass, exchanges what
it created: the new patch records (128 B each, ordered (view, cell, sequence)) and the ids of the patches
it evicted.  One all-gather of counts, one padded all-gather of records, one of kill ids -- over RCCL/xGMI
when the tensors live on the GPU, over gloo in the CPU tests.  Every rank then commits the union in view
order, so all pools stay identical and the result does not depend on the number of ranks.

The `engine` argument only needs the pass / export / commit methods; tests drive this module with an
oracle-backed stand-in on CPU, bench.py with mvskit_amd.engine.Engine on GPUs.
"""
from __future__ import annotations

import numpy as np

RECORD_BYTES = 128


def owner_of(view: int, world: int) -> int:
    return view % world


def merge_in_view_order(per_rank_records, per_rank_view_counts, nviews, world):
    """Concatenate the ranks' exports into the global (view, cell, creation) order.  Every rank's export is already in
    that order over the cells it swept; a view is either owned by one rank (sharding by whole views) or cut into
    contiguous cell ranges held by consecutive ranks (sharding by job range), so within a view the ranks' pieces follow
    each other in rank order.
    per_rank_records[r]: uint8 array/tensor [n_r, 128]; per_rank_view_counts[r]: int array [nviews]."""
    offsets = [0] * world
    parts = []
    for v in range(nviews):
        for r in range(world):
            c = int(per_rank_view_counts[r][v])
            if c:
                parts.append(per_rank_records[r][offsets[r]: offsets[r] + c])
                offsets[r] += c
    return parts


class HostExchange:
    """Exchange through host numpy buffers + torch.distributed (gloo).  Used by the CPU tests."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def exchange(self, new_recs: np.ndarray, per_view: np.ndarray, kills: np.ndarray, nviews: int):
        import torch

        dist, world = self.dist, self.world
        counts = torch.zeros(world, nviews + 2, dtype=torch.int64)
        mine = torch.zeros(nviews + 2, dtype=torch.int64)
        mine[:nviews] = torch.as_tensor(per_view.astype(np.int64))
        mine[nviews] = new_recs.shape[0]
        mine[nviews + 1] = kills.shape[0]
        lst = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(lst, mine, group=self.group)
        counts = torch.stack(lst)
        max_new, max_kill = int(counts[:, nviews].max()), int(counts[:, nviews + 1].max())
        rb = int(new_recs.dtype.itemsize) if new_recs.dtype.itemsize > 1 else (new_recs.shape[1] if new_recs.ndim == 2 else RECORD_BYTES)  # 128, or 192 (64-view library)
        rec_bytes = np.zeros((max(max_new, 1), rb), dtype=np.uint8)
        if new_recs.shape[0]:
            rec_bytes[: new_recs.shape[0]] = new_recs.view(np.uint8).reshape(-1, rb)
        g = [torch.zeros(rec_bytes.shape, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(g, torch.from_numpy(rec_bytes), group=self.group)
        kpad = np.full(max(max_kill, 1), -1, dtype=np.int32)
        kpad[: kills.shape[0]] = kills
        gk = [torch.zeros(kpad.shape, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(gk, torch.from_numpy(kpad), group=self.group)
        parts = merge_in_view_order([t.numpy() for t in g], [counts[r, :nviews].numpy() for r in range(world)], nviews, world)
        allrec = np.concatenate(parts) if parts else np.zeros((0, rb), np.uint8)
        allkill = np.concatenate([gk[r].numpy()[: int(counts[r, nviews + 1])] for r in range(world)]) if max_kill else np.zeros(0, np.int32)
        return allrec, allkill


def sharded_propagate_host(engine, iter_index: int, exchange: HostExchange, nviews: int, patch_dtype):
    """Propagate::run(iter) with host-side exchange; `engine` exposes engine_pass / export_new / export_kills / commit."""
    totals = None
    for p in range(2):
        c = engine.engine_pass(iter_index, p)
        new, per_view = engine.export_new()
        kills = engine.export_kills()
        allrec, allkill = exchange.exchange(new, per_view, kills, nviews)
        engine.commit(allrec.view(patch_dtype).reshape(-1), allkill)
        totals = c if totals is None else {k: totals[k] + c[k] for k in c}
    return totals


class HostStaged:
    """Adapter giving an `Engine` the host-side pass/export/commit interface `sharded_propagate_host` drives (the
    exchange then runs over any torch.distributed backend on CPU tensors, e.g. gloo when several ranks share one GPU)."""

    def __init__(self, engine, device):
        import torch

        self.torch, self.e, self.device = torch, engine, device

    def engine_pass(self, it, p):
        return self.e.engine_pass(it, p)

    def export_new(self):
        torch = self.torch
        n_new, n_kill, per_view = self.e.export_counts()
        rec = torch.zeros(max(n_new, 1), self.e.dtype.itemsize, dtype=torch.uint8, device=self.device)  # 128 bytes, 192 with the 64-view library
        kil = torch.full((max(n_kill, 1),), -1, dtype=torch.int32, device=self.device)
        torch.cuda.synchronize(self.device)
        self.e.export_device(rec.data_ptr(), rec.shape[0], kil.data_ptr(), kil.shape[0])
        self._kills = kil[:n_kill].cpu().numpy()
        return rec[:n_new].cpu().numpy(), per_view

    def export_kills(self):
        return self._kills

    def commit(self, recs, kills):
        torch = self.torch
        r = torch.from_numpy(np.ascontiguousarray(recs).view(np.uint8).reshape(-1, self.e.dtype.itemsize).copy()).to(self.device)
        k = torch.from_numpy(np.ascontiguousarray(kills, dtype=np.int32).copy()).to(self.device)
        torch.cuda.synchronize(self.device)
        self.e.commit_device(r.data_ptr(), r.shape[0], k.data_ptr(), k.shape[0])


class EngineExchange:
    """The in-engine exchange: an RCCL communicator of the engine's own (created from an id that rank 0 generates and
    torch.distributed broadcasts), all-gather + commit inside mvs_engine_propagate."""

    def __init__(self, engine, device, group=None):
        import torch
        import torch.distributed as dist

        self.e = engine
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        uid = torch.zeros(engine.COMM_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            uid = torch.frombuffer(bytearray(engine.comm_unique_id()), dtype=torch.uint8).clone()
        backend = dist.get_backend(group)
        buf = uid.to(device) if backend == "nccl" else uid
        dist.broadcast(buf, src=0, group=group)
        engine.comm_init(bytes(buf.cpu().numpy().tobytes()), rank, world)
        self.description = ("in-engine RCCL exchange on the engine's stream: ncclAllGather of the counts, then every rank's block of 128-byte patch records "
                            "and kill ids broadcast in place behind the pool (grouped ncclBroadcast = all-gather-v)")
        self.last_timing = {}

    def propagate(self, iter_index: int):
        c = self.e.propagate(iter_index)
        self.last_timing = self.e.timing()
        return c


class HostStagedExchange:
    """Rehearsal path for ranks that share one GPU (RCCL refuses two ranks on a device): pass / export to device buffers /
    host copy / all-gather over gloo / commit_device.  Same call shape as EngineExchange."""

    def __init__(self, engine, device, nviews, group=None):
        self.e, self.nviews, self.dtype = engine, nviews, engine.dtype
        self.staged = HostStaged(engine, device)
        self.ex = HostExchange(group)
        self.last_timing = {}

    def propagate(self, iter_index: int):
        import time

        t = {"index_ms": 0.0, "sweep_ms": 0.0, "commit_ms": 0.0, "sweep_launches": 0, "exchange_ms": 0.0, "exchange_bytes": 0}
        totals = None
        for p in range(2):
            c = self.staged.engine_pass(iter_index, p)
            tt = self.e.timing()
            for k in ("index_ms", "sweep_ms", "sweep_launches"):
                t[k] += tt[k]
            t0 = time.perf_counter()
            new, per_view = self.staged.export_new()
            kills = self.staged.export_kills()
            allrec, allkill = self.ex.exchange(new, per_view, kills, self.nviews)
            t["exchange_ms"] += 1000.0 * (time.perf_counter() - t0)
            t["exchange_bytes"] += int(allrec.size + 4 * allkill.size)
            self.staged.commit(allrec.view(self.dtype).reshape(-1), allkill)
            t["commit_ms"] += self.e.timing()["commit_ms"]
            totals = c if totals is None else {k: totals[k] + c[k] for k in c}
        self.last_timing = t
        return totals
