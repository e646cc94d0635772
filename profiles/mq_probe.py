#!/usr/bin/env python3
"""Concurrent-query burst probe: 10M x 256 synthetic table, one 1024-query call; prints wall and the collect-kernel time.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split (sample passes / pick_tau / collect / rescore)."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from pixelbox_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ix = capi.Index(256, n)
ix.fill_synthetic(synth.SEED_INDEX, 0, n, 1)
q = synth.fill_synthetic(synth.SEED_QUERY + 1, 0, nq * 256).reshape(nq, 256)
ix.search(q[:128], 100, 1e3)
for rep in range(3):
    ix.stats(reset=True)
    ix.set_option(capi.PB_OPT_PROFILE, 1)
    t0 = time.perf_counter()
    ix.search(q, 100, 1e3)
    dt = time.perf_counter() - t0
    st = ix.stats()
    print(f"burst {nq}: {dt * 1e3:.3f} ms wall, {nq / dt:.0f} q/s; collect kernel {st.profiled_ms:.3f} ms; certified {st.fast_path} fallback {st.fallback}")
# a -DPB_MQ_STAMP build (profiles/build_scan_ablation.sh): s_memtime clocks of the collect kernel's phases, summed over its waves
import ctypes  # noqa: E402

if hasattr(capi.lib(), "pb_debug_mq_stamps"):
    out = (ctypes.c_ulonglong * 8)()
    capi.lib().pb_debug_mq_stamps(out, 1)
    c, s, b, nsteps = out[0], out[1], out[2], max(out[3], 1)
    print(f"stamps per wave-step (100 MHz ticks): compute {c / nsteps:.1f} stage {s / nsteps:.1f} barrier wait {b / nsteps:.1f}  ({nsteps} wave-steps)")
