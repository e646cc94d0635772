#!/usr/bin/env python3
"""What ONE MI355X can measure of bench.py's multi-GPU scan step, and the prediction the first real N-GPU run is to be held to
(no multi-GPU node was available to any round).

A step = 64 queries answered over the whole table by N ranks (bench.py --gpus N; pixelbox_amd/sharded.py):
    t_step(N) = t_shard(rows per rank) + t_allgather(N) + t_merge(N) + t_host
Measured here, on one GPU:
  * t_shard(R)   ShardedIndex.search's shard part, `pb_index_search_packed` of 64 queries over R rows, R = 10M / N for the strong
                 leg (10M rows in all) and R = 10M for the weak leg (10M rows PER GPU), median of the steps after a warm-up;
  * t_merge(N)   `pb_topk_merge_packed_device` over N gathered lists of 64 x 201 x 8 B (the real kernel, N = 1, 2, 4, 8);
  * t_ag(1)      `all_gather_into_tensor` at world size 1 through RCCL (its launch floor);
  * t_host       whole ShardedIndex.search (world 1, collective forced) minus the three parts above.
NOT measurable here, and the one assumption of the model: the N-rank all-gather over xGMI.  The message is 102 912 B per rank; an
RCCL ring all-gather makes N - 1 steps, each a hop latency plus the message at one link's rate (xGMI: ~45-50 GB/s usable per
direction per link -> ~2.2 us for 103 KB), so  t_ag(N) = t_ag(1) + (N - 1) * (hop + 103 KB / 45 GB/s)  with hop = 5 us (low) and
15 us (high).  Both bounds are printed; the prediction table uses the HIGH one.

    python3 profiles/scaling_model.py > profiles/r05_scaling_model.txt
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ["RANK"] = "0"
os.environ["WORLD_SIZE"] = "1"
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
from pixelbox_amd import capi, synth
from pixelbox_amd.sharded import ShardedIndex

dev = torch.device("cuda", 0)
K, B, D = 100, 64, 256
STEPS = int(os.environ.get("PB_SCALE_STEPS", "12"))
qs = [synth.fill_synthetic(synth.SEED_QUERY, s * B * D, B * D).reshape(B, D) for s in range(STEPS + 2)]


def med(f, n=STEPS, warm=2):
    for i in range(warm):
        f(i)
    torch.cuda.synchronize()
    ts = []
    for i in range(n):
        t0 = time.perf_counter()
        f(warm + i)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), float(min(ts)), float(max(ts))


packed = torch.empty((B, 2 * K + 1), dtype=torch.int64, device=dev)
shard_ms = {}
host_ms = None
for rows in (10_000_000, 5_000_000, 2_500_000, 1_250_000):
    ix = capi.Index(D, rows)
    ix.fill_synthetic(synth.SEED_INDEX, 0, rows, 1)
    ix.set_option(capi.PB_OPT_SEARCH_PATH, 2)
    m, lo, hi = med(lambda i: ix.search_packed(qs[i], K, 1e3, packed.data_ptr()))
    shard_ms[rows] = m
    print(f"t_shard  {rows:>10d} rows: {m:8.3f} ms per 64 queries (min {lo:.3f}, max {hi:.3f})  = {B * rows * D / (m * 1e-3) / 1e12:.2f} TB/s of table bytes")
    if rows == 1_250_000:
        sh = ShardedIndex(D, rows, 0, 1, 0, group=dist.group.WORLD)
        sh.index = ix
        whole, _, _ = med(lambda i: sh.search(qs[i], K, 1e3))
        host_whole = whole
    del ix

gath = torch.empty((B, 2 * K + 1), dtype=torch.int64, device=dev)
ag1, _, _ = med(lambda i: dist.all_gather_into_tensor(gath, packed), n=50)
print(f"t_ag(1)  all_gather_into_tensor, world 1 (RCCL launch floor): {ag1 * 1e3:7.1f} us")
merge_ms = {}
for n in (1, 2, 4, 8):
    g = packed.repeat(n, 1).contiguous()
    m, _, _ = med(lambda i: capi.topk_merge_packed_device(0, g.data_ptr(), n, B, K), n=50)
    merge_ms[n] = m
    print(f"t_merge({n}) pb_topk_merge_packed_device over {n} lists of 64 x 201 x 8 B (results to the host): {m * 1e3:7.1f} us")
t_host = max(0.0, host_whole - shard_ms[1_250_000] - ag1 - merge_ms[1])
print(f"t_host   ShardedIndex.search (1.25M rows, world 1, collective forced) {host_whole:.3f} ms - parts = {t_host * 1e3:7.1f} us")

MSG = B * (2 * K + 1) * 8


def t_ag(n, hop_us):
    return ag1 + (n - 1) * (hop_us + MSG / 45e9 * 1e6) * 1e-3


print()
print(f"model: t_step(N) = t_shard + t_ag(N) + t_merge(N) + t_host;  t_ag(N) = t_ag(1) + (N - 1) (hop + {MSG} B / 45 GB/s), hop = 5 us (low) / 15 us (high)")
print("strong leg (10M rows in all; bench.py `value`):")
t1 = shard_ms[10_000_000]
print(f"  N = 1: {t1:8.3f} ms per step = {B / t1 * 1e3:9.1f} q/s (measured: one index, no exchange)")
for n, rows in ((2, 5_000_000), (4, 2_500_000), (8, 1_250_000)):
    lo = shard_ms[rows] + t_ag(n, 5) + merge_ms[n] + t_host
    hi = shard_ms[rows] + t_ag(n, 15) + merge_ms[n] + t_host
    print(f"  N = {n}: {hi:8.3f} ms per step ({lo:.3f} with the low hop) = {B / hi * 1e3:9.1f} q/s predicted, "
          f"speed-up {t1 / hi:.2f}x, efficiency {t1 / hi / n:.2f}  [shard {shard_ms[rows]:.3f} + all-gather {t_ag(n, 15):.3f} + merge {merge_ms[n]:.3f} + host {t_host:.3f}]")
print("weak leg (10M rows PER GPU; bench.py `weak_scaling.value` = queries/s over the N x 10M-row table):")
for n in (2, 4, 8):
    hi = t1 + t_ag(n, 15) + merge_ms[n] + t_host
    print(f"  N = {n}: {hi:8.3f} ms per step = {B / hi * 1e3:9.1f} q/s over {n} x 10M rows, {n * B * 10_000_000 * D / (hi * 1e-3) / 1e12:6.2f} TB/s of table bytes over all GPUs, "
          f"efficiency {t1 / hi:.3f} of N x the one-GPU stream")
dist.destroy_process_group()
