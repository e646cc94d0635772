"""Is a 'slow table'  a property of its memory or of the moment?  One 10M-row table, the 64-pass looped filter launch
timed again and again for a few seconds from process start.  PB_HOG_GB=n: before anything else this process allocates n GB in 1 GB
blocks and frees them again (what the process before this one on the box did by exiting)."""
import ctypes as C, os, sys, time
T0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pixelbox_amd import capi, synth

hog = float(os.environ.get("PB_HOG_GB", "0"))
if hog > 0:
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    bl = []
    for i in range(int(hog)):
        p_ = C.c_void_p()
        if hip.hipMalloc(C.byref(p_), 1 << 30) != 0:
            break
        if os.environ.get("PB_HOG_TOUCH"):
            hip.hipMemset(p_, 1, 1 << 30)
        bl.append(p_)
    hip.hipDeviceSynchronize()
    for b in bl:
        hip.hipFree(b)
    print(f"hog: {len(bl)} GB allocated and freed at t = {time.perf_counter() - T0:.2f} s", flush=True)
if os.environ.get("PB_DRAIN"):
    # does ALLOCATING (never touching) the free memory make the driver finish its scrubbing first?
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    fr, tot = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(fr), C.byref(tot))
    t_d = time.perf_counter()
    bl = []
    for i in range(max(0, (fr.value >> 30) - 8)):
        p_ = C.c_void_p()
        if hip.hipMalloc(C.byref(p_), 1 << 30) != 0:
            break
        bl.append(p_)
    hip.hipDeviceSynchronize()
    for b in bl:
        hip.hipFree(b)
    print(f"drain: {len(bl)} GB of {fr.value >> 30} free allocated untouched and freed in {time.perf_counter() - t_d:.2f} s", flush=True)
rows = 10_000_000
ix = capi.Index(256, rows)
ix.fill_synthetic(synth.SEED_INDEX, 0, rows, 1)
ix.set_option(capi.PB_OPT_SEARCH_PATH, 2)
q = synth.fill_synthetic(synth.SEED_QUERY, 0, 2 * 64 * 256).reshape(2, 64, 256)
ix.search(q[0], 100, 1e3)
out = []
for it in range(int(os.environ.get("PB_TIMELINE_N", "60"))):
    ix.stats(reset=True); ix.set_option(capi.PB_OPT_PROFILE, 1)
    ix.search(q[it & 1], 100, 1e3)
    st = ix.stats(); ix.set_option(capi.PB_OPT_PROFILE, 0)
    out.append((time.perf_counter() - T0, st.profiled_ms / st.profiled_launches))
print(" ".join(f"{t:.2f}s:{ms:.2f}" for t, ms in out))
