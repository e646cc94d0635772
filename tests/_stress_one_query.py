"""Stress for intermittent mismatches of one-query calls (not collected by pytest): the table of
test_one_query_calls_hand_out_the_table_tail_by_tickets[2100001], every form of the launch, many calls, all compared with
the oracle's answer.  argv: calls per form (default 300)"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

from oracle import capi as oracle
from pixelbox_amd import capi

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = int(os.environ.get("PB_STRESS_ROWS", "2100001"))
rng = np.random.default_rng(4242 + n)
rows = rng.integers(0, 256, size=(n, 256), dtype=np.uint8)
ids = np.arange(n, dtype=np.int64) * 3 + 1
q = rng.integers(0, 256, size=256, dtype=np.uint8)
spots = np.unique(np.concatenate([np.arange(min(n, 8)), n - 1 - np.arange(min(n, 40)),
                                  (n * 7 // 8 + np.arange(-20, 20) * 32) % n, rng.integers(0, n, size=60)]))
for j, r in enumerate(spots):
    rows[r] = q
    rows[r, j % 256] ^= np.uint8(1 + j % 7)
qs = np.concatenate([q[None, :], rng.integers(0, 256, size=(3, 256), dtype=np.uint8)])
want = [oracle.scan_topk(x, rows, ids, 100, 1e3) for x in qs]
for form in ("default", "PB_FORCE_STEAL", "PB_FORCE_TAIL_TICKETS", "PB_STATIC_TAIL"):
    if form != "default":
        os.environ[form] = "1"
    ix = capi.Index(256, n)
    ix.load(ids, rows)
    bad = 0
    for rep in range(calls):
        qi = rep % len(qs)
        gi, gd, gc = ix.search(qs[qi:qi + 1], 100, 1e3)
        c = int(gc[0])
        if c != len(want[qi][0]) or not np.array_equal(gi[0, :c], want[qi][0]) or not np.array_equal(gd[0, :c].view(np.uint32), want[qi][1].view(np.uint32)):
            bad += 1
            if bad <= 3:
                w = want[qi][0]
                diff = [i for i in range(min(c, len(w))) if gi[0, i] != w[i]]
                print(f"  {form}: call {rep} query {qi}: count {c} vs {len(w)}, first differing places {diff[:8]}, got {gi[0, diff[:4]] if diff else ''} want {w[diff[:4]] if diff else ''}", flush=True)
    st = ix.stats()
    print(f"{form:24s} rows {n}: {bad} of {calls} calls differ from the oracle; certified {st.fast_path}/{st.queries}, stamp time-outs {st.stamp_timeouts}", flush=True)
    del ix
    if form != "default":
        del os.environ[form]
