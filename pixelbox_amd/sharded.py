"""Row-sharded `semantic_hashes` index over the GPUs of one node: one process per GPU.

The scan shards naturally (rows are independent): rank r owns the contiguous row
range [r*ceil(N/G), (r+1)*ceil(N/G)) with its slice of the image_ids; a query runs on every shard
(`pb_index_search_packed`), the per-shard top-k lists are exchanged with ONE all-gather per query batch
(torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests) and every rank
runs the same G-way merge (`pb_topk_merge_packed`, (dist, image_id) order).  The message is nq * (2k+1) * 8 B
per rank (25.7 KB for 16 queries, k = 100): latency-bound, so queries are batched per collective.

Reference API being replaced: Engine::query_by_image_hash_from_image (engine.rs:363-396), one table.

This module is the LAUNCHER-SIDE form (bench.py under torchrun: one process per GPU, the driver's contract).  The product
form for a single-process host such as the reference is in the library itself: `pb_sharded_create(device_ids, n, ...)` /
`pb_sharded_search` (pixelbox_amd/csrc/pb_sharded.hip: worker thread per shard, ncclCommInitAll + ncclAllGather, the same
device merge kernel), bound as capi.ShardedIndexC.
"""
from __future__ import annotations

import numpy as np

from . import capi


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    per = (n_total + world - 1) // world
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)


class ShardedIndex:
    def __init__(self, dim: int, capacity_total: int, rank: int = 0, world: int = 1, device: int = 0, group=None):
        self.dim, self.rank, self.world, self.group = dim, rank, world, group
        self.capacity_total = capacity_total
        lo, hi = shard_range(capacity_total, rank, world)
        self.row_lo, self.row_hi = lo, hi
        self.device = device
        self.index = self._make_shard(dim, max(hi - lo, 1), device)
        self._torch = None
        if world > 1 or group is not None:
            import torch

            self._torch = torch

    def _make_shard(self, dim: int, rows: int, device: int):
        """This rank's shard: the HIP index.  (The CPU tests of the collective + merge plumbing subclass this class in the
        TEST and return a stand-in with the same .load / .search; nothing in this package can produce anything else.)"""
        return capi.Index(dim, rows, device)

    # -- store ---------------------------------------------------------------------------------------------
    def fill_synthetic(self, seed: int, first_id: int = 1):
        """Rows [row_lo, row_hi) of the synthetic stream, generated on this rank's GPU (no host staging)."""
        n = self.row_hi - self.row_lo
        if n:
            self.index.fill_synthetic(seed, self.row_lo, n, first_id + self.row_lo)

    def load(self, image_ids: np.ndarray, rows: np.ndarray):
        """Every rank passes the whole table (ascending ids); each keeps its contiguous slice."""
        lo, hi = shard_range(len(image_ids), self.rank, self.world)
        self.index.load(image_ids[lo:hi], rows[lo:hi])

    # -- query ---------------------------------------------------------------------------------------------
    def search(self, queries: np.ndarray, k: int = 100, max_dist: float = 1e3):
        """-> (ids [nq, k] int64, dist [nq, k] f32, count [nq]) of the GLOBAL top-k, identical on every rank."""
        queries = np.ascontiguousarray(queries, dtype=np.uint8).reshape(-1, self.dim)
        nq = queries.shape[0]
        if self.world == 1 and self.group is None:
            return self.index.search(queries, k, max_dist)
        torch = self._torch
        import torch.distributed as dist

        backend = dist.get_backend(self.group)
        on_gpu = backend == "nccl"
        dev = torch.device("cuda", self.device) if on_gpu else torch.device("cpu")
        # per-rank message: int64 [nq, 2k+1] = ids | dist bits | count
        packed = torch.empty((nq, 2 * k + 1), dtype=torch.int64, device=dev)
        if on_gpu and isinstance(self.index, capi.Index):
            self.index.search_packed(queries, k, max_dist, packed.data_ptr())  # written on the device, synchronised
        else:
            l_ids, l_dist, l_cnt = self.index.search(queries, k, max_dist)
            packed.copy_(torch.from_numpy(capi.pack_results(l_ids[:, :k], l_dist[:, :k], l_cnt)))
        gathered = torch.empty((self.world * nq, 2 * k + 1), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(gathered, packed, group=self.group)
        if on_gpu:
            # merged on the device by the kernel pb_sharded_search uses (pb_merge_kernels.h); only the k results travel
            torch.cuda.current_stream(dev).synchronize()
            return capi.topk_merge_packed_device(self.device, gathered.data_ptr(), self.world, nq, k)
        g = gathered.numpy().reshape(self.world, nq, 2 * k + 1)
        return capi.topk_merge_packed(g, k)
