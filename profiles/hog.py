"""Writes PB_HOG_GB (default 150) GB of device memory in 1 GB blocks and exits: what a memory-heavy job leaves the NEXT process
on the box with (the driver scrubs the released memory while that process already runs; profiles/r05_placement.txt)."""
import ctypes as C, os
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
n = 0
for i in range(int(os.environ.get("PB_HOG_GB", "150"))):
    p = C.c_void_p()
    if hip.hipMalloc(C.byref(p), 1 << 30) != 0:
        break
    hip.hipMemset(p, 1, 1 << 30)
    n += 1
hip.hipDeviceSynchronize()
print(f"hog: wrote {n} GB, exiting")
