#!/usr/bin/env python3
"""bench.py's embedding-like table leg by itself (10M rows of quantise(tanh(0.5 N(0,1))), queries = rows of the table): PB_NO_SEED=1 for the
looped launch without the sample-seeded thresholds."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

class A: pass
a = A(); a.dim = 256; a.k = 100; a.queries = 64; a.rows = int(os.environ.get("PB_PROBE_ROWS", "10000000")); a.max_dist = 1e3; a.settle_seconds = float(os.environ.get("PB_SETTLE", "0")); a.settle_min_seconds = 1.5
print(json.dumps(bench.bench_clustered(a, torch, 0)))
