"""Build the HIP shared library in-tree: pixelbox_amd/libpixelbox_hip.so (gfx950 only).

hipcc cross-compiles without a GPU.  -ffp-contract=off: the exact re-scoring kernels and the host-side
query fold must round every multiply and add separately, like the reference's Rust f32 code
(engine.rs:572-588) -- the top-k order depends on it.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpixelbox_hip.so")
SOURCES = ["pb_scan.hip", "pb_embed.hip"]
HEADERS = ["pb_common.h", "pb_scan_kernels.h", "pb_embed_kernels.h", os.path.join("..", "..", "include", "pixelbox_hip.h")]


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs = []
    for s in srcs:
        o = os.path.splitext(s)[0] + ".o"
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-fno-fast-math", "-Wall", "-Wno-unused-function", "-c", s, "-o", o]
        cmd[1:1] = os.environ.get("PB_EXTRA_HIPCC_FLAGS", "").split()  # kernel experiments (-D...), empty by default
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
