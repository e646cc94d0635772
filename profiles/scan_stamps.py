"""Where the fixed part of a one-query call over a small table goes: wall-clock stamps (s_memrealtime, 100 MHz) inside
k_scan_filter<ARGQ> and k_select_rescore.  Needs a library built with -DPB_SCAN_STAMP
(PB_EXTRA_HIPCC_FLAGS=-DPB_SCAN_STAMP, both when building and when running).  argv: rows (default 1000000)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np

from pixelbox_amd import capi, synth

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ix = capi.Index(256, rows)
ix.fill_synthetic(synth.SEED_INDEX, 0, rows, 1)
ix.set_option(capi.PB_OPT_SEARCH_PATH, 2)
q = synth.fill_synthetic(synth.SEED_QUERY + 5, 0, 64 * 256).reshape(64, 256)
for i in range(5):
    ix.search(q[i:i + 1], 100, 1e3)
L = capi.lib()
L.pb_debug_scan_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
nf = 512 * 16 * 8
bf = np.zeros(nf, dtype=np.uint64)
bs = np.zeros(16, dtype=np.uint64)
names = ["wave starts", "prologue done", "first tile evaluated", "streaming done", "wave list sorted / left the tree", "after barrier", "workgroup list written"]
acc = []
walls = []
for i in range(32):
    call, *_ = ix.prepared_search(q[i:i + 1], 100, 1e3)
    L.pb_debug_scan_stamps(bf.ctypes.data, bs.ctypes.data, 1)
    t0 = time.perf_counter()
    call()
    walls.append((time.perf_counter() - t0) * 1e6)
    L.pb_debug_scan_stamps(bf.ctypes.data, bs.ctypes.data, 0)
    st = bf.reshape(-1, 8).astype(np.float64)
    live = st[:, 0] > 0
    st = st[live]
    base = st[:, 0].min()
    row = []
    for s in range(7):
        v = st[:, s]
        v = v[v > 0]
        if v.size == 0:  # a slot this form of the kernel does not pass (the arrival tree has no barrier)
            row += [float("nan")] * 3
            continue
        row += [(v.min() - base) / 100.0, (np.median(v) - base) / 100.0, (v.max() - base) / 100.0]
    sel = (bs[:8].astype(np.float64) - base) / 100.0
    n_cand = int(bs[8])
    acc.append(row + sel.tolist() + [float(live.sum())])
    if i == 31 and os.environ.get("PB_STAMP_DETAIL"):
        full = bf.reshape(-1, 8, 8).astype(np.float64)  # [workgroup][wave][slot]
        n_wg = int(live.sum()) // 8
        fin = (full[:n_wg, :, 3].max(axis=1) - base) / 100.0  # the workgroup's last wave leaves the streaming loop
        print("  streaming done per workgroup, mean by XCD (blockIdx % 8):", " ".join(f"{fin[x::8].mean():.1f}" for x in range(8)))
        print("  ... by blockIdx / 32:", " ".join(f"{fin[32 * g:32 * g + 32].mean():.1f}" for g in range(n_wg // 32)))
        print("  ... slowest 16 workgroups:", " ".join(f"{b}:{fin[b]:.1f}" for b in np.argsort(-fin)[:16]))
a = np.nanmedian(np.array(acc), axis=0)
print(f"rows {rows}: {int(a[-1])} waves; us after the first wave's start, median over 32 calls (min / median / max over waves); call wall {np.median(walls):.1f} us")
for s in range(7):
    print(f"  {names[s]:24s} {a[3 * s]:7.2f} {a[3 * s + 1]:7.2f} {a[3 * s + 2]:7.2f}")
sn = ["k_select_rescore starts", "lists + headers loaded", "lower bound found", "candidates chosen", "candidates re-scored", "ordered + results stored", "fence + barrier", "header + stamp stored"]
for s in range(8):
    print(f"  {sn[s]:24s} {a[21 + s]:7.2f}")
print(f"  candidates of the last call: {n_cand}; lower bound {np.array([bs[9]], dtype=np.uint64).astype(np.uint32).view(np.float32)[0]:.6f}, cut {np.array([bs[10]], dtype=np.uint64).astype(np.uint32).view(np.float32)[0]:.6f}")
