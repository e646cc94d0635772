"""Weight blob (PBXW0001) for the embed network, and a seeded synthetic generator.

The reference loads `models/image_similarity.onnx` (src/image_hashes/efficientnet.rs:5,12), which is
git-ignored and absent (.gitignore:6).  Its content is fixed by resources/train.py:30-46,167-174:
torchvision EfficientNet-B0 `features` -> AdaptiveAvgPool2d(1) -> Flatten -> Linear(1280, D) -> Tanh,
exported in eval mode with constant folding, so BatchNorm is already folded into each conv
(w' = w*gamma/sqrt(var+eps), b' = beta - mean*gamma/sqrt(var+eps)).  The blob stores w', b' directly.

Layout (little-endian):
    bytes 0..7    magic "PBXW0001"
    u32 H, u32 W, u32 D, u32 n_tensors
    u64 n_floats
    f32[n_floats] tensors, in this order, each conv weight in torch OIHW order:
        stem.w [32,3,3,3]  stem.b [32]
        per MBConv block:  (expand.w [E,Cin] expand.b [E])   -- absent when expand ratio == 1
                           dw.w [E,k,k] dw.b [E]
                           se_reduce.w [S,E] se_reduce.b [S]      S = max(1, Cin // 4)
                           se_expand.w [E,S] se_expand.b [E]
                           project.w [Cout,E] project.b [Cout]
        head.w [1280,320] head.b [1280]
        fc.w [D,1280] fc.b [D]
"""
from __future__ import annotations

import struct
from dataclasses import dataclass

import numpy as np

MAGIC = b"PBXW0001"
HEADER_BYTES = 32

# (expand_ratio, kernel, stride, cin, cout, repeats) -- torchvision efficientnet_b0
STAGES = [
    (1, 3, 1, 32, 16, 1),
    (6, 3, 2, 16, 24, 2),
    (6, 5, 2, 24, 40, 2),
    (6, 3, 2, 40, 80, 3),
    (6, 5, 1, 80, 112, 3),
    (6, 5, 2, 112, 192, 4),
    (6, 3, 1, 192, 320, 1),
]


@dataclass(frozen=True)
class Block:
    cin: int
    cout: int
    expanded: int
    squeeze: int
    kernel: int
    stride: int
    has_expand: bool
    residual: bool


def blocks() -> list[Block]:
    out = []
    for ratio, k, s, cin, cout, reps in STAGES:
        for r in range(reps):
            ci = cin if r == 0 else cout
            st = s if r == 0 else 1
            out.append(Block(ci, cout, ci * ratio, max(1, ci // 4), k, st, ratio != 1, st == 1 and ci == cout))
    return out


def tensor_specs(d: int) -> list[tuple[str, tuple[int, ...], str]]:
    """(name, shape, role) for every tensor, in blob order. role in {silu, linear, dw, se_r, se_e, bias, fc}."""
    specs: list[tuple[str, tuple[int, ...], str]] = [("stem.w", (32, 3, 3, 3), "silu"), ("stem.b", (32,), "bias")]
    for i, b in enumerate(blocks()):
        p = f"b{i}."
        if b.has_expand:
            specs += [(p + "expand.w", (b.expanded, b.cin), "silu"), (p + "expand.b", (b.expanded,), "bias")]
        specs += [(p + "dw.w", (b.expanded, b.kernel, b.kernel), "dw"), (p + "dw.b", (b.expanded,), "bias")]
        specs += [(p + "se_reduce.w", (b.squeeze, b.expanded), "se_r"), (p + "se_reduce.b", (b.squeeze,), "bias")]
        specs += [(p + "se_expand.w", (b.expanded, b.squeeze), "se_e"), (p + "se_expand.b", (b.expanded,), "se_bias")]
        specs += [(p + "project.w", (b.cout, b.expanded), "linear"), (p + "project.b", (b.cout,), "bias")]
    specs += [("head.w", (1280, 320), "silu"), ("head.b", (1280,), "bias")]
    specs += [("fc.w", (d, 1280), "fc"), ("fc.b", (d,), "bias")]
    return specs


def n_floats(d: int) -> int:
    return sum(int(np.prod(s)) for _, s, _ in tensor_specs(d))


# ---- seeded synthetic weights (no ONNX file exists here; SURVEY.md section 7 "Weights") ------------
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def _splitmix64_at(seed: int, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (idx.astype(np.uint64) + np.uint64(1)) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _uniform_pm1(seed: int, start: int, n: int) -> np.ndarray:
    """n floats in [-1, 1): (top 24 bits of splitmix64 / 2^23) - 1, exact in f32."""
    z = _splitmix64_at(seed, np.arange(start, start + n, dtype=np.uint64))
    u = (z >> np.uint64(40)).astype(np.float32)  # < 2^24, exact
    return (u * np.float32(2.0 ** -23) - np.float32(1.0)).astype(np.float32)


# Raw init: U(-a, a) with a = sqrt(3 / fan_in) (unit-variance-preserving), depthwise filters get +1.0 on
# the centre tap (trained depthwise filters are centre-heavy; it also keeps the signal alive on the
# 1x1..4x4 maps of small test resolutions, where most taps fall into the zero padding).
# _CALIB: per-weight-tensor multipliers from a one-off LSUV-style pass (tests/golden/calibrate_weights.py)
# standing in for the BatchNorm statistics a trained, BN-folded export carries; constants, so the
# generator stays integer-PRNG + one f32 multiply and reproduces bit-identically anywhere.
_CALIB: list[float] | None = [
    2.66, 0.714, 1.05, 1.38, 1.05, 1.17, 1.31, 0.737, 1.59, 1.03, 1.16, 1.3, 0.948, 1.58, 0.949, 0.805, 1.24,
    1.24, 1.11, 1.57, 1.24, 1.02, 0.719, 1.52, 1.11, 0.883, 1.12, 1.16, 1.52, 0.914, 1.22, 1.1, 1.29, 1.84,
    0.995, 0.836, 1.31, 1.19, 1.3, 1.27, 0.749, 1.17, 1.27, 1.57, 1.14, 1.27, 1.16, 1.15, 1.68, 1.13, 0.893,
    1.23, 1.23, 1.36, 1.25, 0.71, 1.11, 1.25, 1.63, 1.22, 1.24, 1.18, 1.25, 1.41, 1.19, 0.877, 1.21, 1.2,
    1.46, 1.16, 0.747, 1.19, 1.36, 1.37, 1.17, 0.616, 1.2, 1.32, 1.29, 1.14, 1.27, 1.12,
]


def synthetic_blob(seed: int = 0x5EED0005, h: int = 128, w: int = 128, d: int = 256, fc_gain: float = 1.0) -> bytes:
    """Deterministic (integer PRNG) random-init weights in PBXW0001 format.

    fc_gain multiplies the final Linear's weights (one exact f32 multiply per weight; 1.0 = the blob every fixture uses).
    The end-to-end leg of bench.py uses the structured `synth.synthetic_scenes` images, whose pooled features vary less
    from image to image than the brightness-window images the calibration was made on; a gain of 3 restores outputs that
    fill (-1, 1) (a trained, BN-folded model does this by itself)."""
    specs = tensor_specs(d)
    total = n_floats(d)
    out = np.empty(total, dtype=np.float32)
    pos = 0
    wi = 0
    for name, shape, role in specs:
        n = int(np.prod(shape))
        r = _uniform_pm1(seed, pos, n)
        if role == "bias":
            t = r * np.float32(0.05)
        elif role == "se_bias":
            t = r * np.float32(0.05) + np.float32(1.0)  # gates centred on sigmoid(1) ~ 0.73
        else:
            fan_in = int(np.prod(shape[1:]))
            a = np.float32(np.sqrt(np.float32(3.0) / np.float32(fan_in)))
            t = (r * a).astype(np.float32)
            if role == "dw":
                t = t.reshape(shape).copy()
                t[:, shape[1] // 2, shape[2] // 2] += np.float32(1.0)
                t = t.reshape(-1)
            if _CALIB is not None:
                t = (t * np.float32(_CALIB[wi])).astype(np.float32)
            if role == "fc" and fc_gain != 1.0:
                t = (t * np.float32(fc_gain)).astype(np.float32)
            wi += 1
        out[pos : pos + n] = t.astype(np.float32)
        pos += n
    hdr = MAGIC + struct.pack("<IIIIQ", h, w, d, len(specs), total)
    assert len(hdr) == HEADER_BYTES
    return hdr + out.astype("<f4").tobytes()


def parse_blob(blob: bytes) -> tuple[int, int, int, dict[str, np.ndarray]]:
    """-> (H, W, D, {name: array}) ; raises ValueError on a malformed blob."""
    if len(blob) < HEADER_BYTES or blob[:8] != MAGIC:
        raise ValueError("not a PBXW0001 blob")
    h, w, d, nt, nf = struct.unpack("<IIIIQ", blob[8:HEADER_BYTES])
    specs = tensor_specs(d)
    if nt != len(specs) or nf != n_floats(d) or len(blob) != HEADER_BYTES + 4 * nf:
        raise ValueError("PBXW0001 blob: size/tensor-count mismatch")
    flat = np.frombuffer(blob, dtype="<f4", offset=HEADER_BYTES)
    out, pos = {}, 0
    for name, shape, _ in specs:
        n = int(np.prod(shape))
        out[name] = flat[pos : pos + n].reshape(shape)
        pos += n
    return h, w, d, out
