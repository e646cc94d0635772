import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pixelbox_amd import capi, synth
n = int(os.environ.get("ROWS", "10000000")); d = 256; B = 64
ix = capi.Index(d, n); ix.fill_synthetic(synth.SEED_INDEX, 0, n, 1)
q = synth.fill_synthetic(synth.SEED_QUERY, 0, 8 * B * d).reshape(8, B, d)
ix.set_option(capi.PB_OPT_SEARCH_PATH, 2)
ref = None
for mode, waves, wgcu, var in ((0, 8, 1, 0), (2, 8, 1, 0), (2, 8, 1, 8), (2, 8, 1, 2), (2, 8, 1, 4), (2, 8, 2, 4), (2, 4, 2, 0), (2, 8, 1, 0), (2, 8, 1, 8)):
    ix.set_option(8, mode); ix.set_option(6, waves); ix.set_option(5, wgcu); ix.set_option(4, var)
    ix.search(q[0], 100, 1e3)
    ix.stats(reset=True); ix.set_option(capi.PB_OPT_PROFILE, 1)
    t0 = time.perf_counter()
    outs = [ix.search(q[s], 100, 1e3) for s in range(1, 8)]
    dt = time.perf_counter() - t0
    ix.set_option(capi.PB_OPT_PROFILE, 0)
    st = ix.stats()
    if ref is None: ref = outs
    same = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2], b[2]) for a, b in zip(outs, ref))
    print(f"rows {n} mode {mode} waves {waves} wg/cu {wgcu} variant {var}: {7 * B / dt:9.1f} q/s, {st.profiled_ms / (7 * B) * 1e3:7.2f} us per query pass = {st.profiled_bytes / (st.profiled_ms * 1e-3) / 1e9:7.1f} GB/s, fast {st.fast_path} fallback {st.fallback}, same results {same}")
