"""Build the HIP shared library in-tree: pixelbox_amd/libpixelbox_hip.so (gfx950 only).

hipcc cross-compiles without a GPU.  -ffp-contract=off: the exact re-scoring kernels and the host-side
query fold must round every multiply and add separately, like the reference's Rust f32 code
(engine.rs:572-588) -- the top-k order depends on it.

The gate is a CONTENT hash, not an mtime: `libpixelbox_hip.so.stamp` records the sha256 of every source,
header and the compile flags the library was built from; build() recompiles whenever the tree's hash differs
and always prints which of the two happened, so a log reader can tell whether the shipped binary is the
source's.  The translation units compile in parallel.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpixelbox_hip.so")
STAMP = LIB + ".stamp"
SOURCES = ["pb_scan.hip", "pb_embed.hip", "pb_gemm_p3.hip", "pb_sharded.hip", "pb_phash.hip"]
HEADERS = ["pb_common.h", "pb_scan_kernels.h", "pb_embed_common.h", "pb_embed_kernels.h", "pb_gemm_p3.h", "pb_p3_common.h", "pb_gemm_p3_launch.h", "pb_front_band.h", "pb_block_small.h", "pb_phash_kernels.h", "pb_merge_kernels.h",
           os.path.join("..", "..", "include", "pixelbox_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall",
         "-Wno-unused-function"]
# RCCL (the sharded index's all-gather, pb_sharded.hip) is dlopen'ed at pb_sharded_create: a single-GPU host never loads it
LINK = ["-ldl"]


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def _extra() -> list[str]:
    return os.environ.get("PB_EXTRA_HIPCC_FLAGS", "").split()  # kernel experiments (-D...), empty by default


def source_hash() -> str:
    h = hashlib.sha256()
    h.update(" ".join(FLAGS + _extra() + LINK).encode())
    for rel in SOURCES + HEADERS:
        p = os.path.join(CSRC, rel)
        if os.path.exists(p):
            h.update(rel.encode())
            with open(p, "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def needs_build() -> bool:
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def _compile(src: str, verbose: bool) -> str:
    obj = os.path.splitext(src)[0] + ".o"
    cmd = [_hipcc()] + _extra() + FLAGS + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    want = source_hash()
    if not force and not needs_build():
        print(f"pixelbox_amd.build: {os.path.basename(LIB)} is up to date with the sources (sha256 {want[:16]})", file=sys.stderr)
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with ThreadPoolExecutor(max_workers=min(5, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, verbose), srcs))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + LINK
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(want + "\n")
    print(f"pixelbox_amd.build: compiled {len(srcs)} translation units for gfx950 -> {os.path.basename(LIB)} "
          f"(sha256 of sources + flags {want[:16]})", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--if-needed" not in sys.argv, verbose=True))
