"""Synthetic inputs for tests and bench (host side, numpy): a counter-based splitmix64 byte stream.

Same definition as the device generator `pb_fill_synthetic` (include/pixelbox_hip.h) and the oracle's
`pbo_fill_synthetic`: byte j of stream `seed` is byte (j & 7), little-endian, of
mix(seed + (j/8 + 1) * 0x9E3779B97F4A7C15)  ("fixed-seed integer PRNG").
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)

# Seeds named in SURVEY.md section 8(d)
SEED_INDEX = 0x5EED0002
SEED_QUERY = 0x5EED0003
SEED_IMAGES = 0x5EED0004
SEED_WEIGHTS = 0x5EED0005


def splitmix64_at(seed: int, word_index: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (word_index.astype(np.uint64) + np.uint64(1)) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def fill_synthetic(seed: int, byte_offset: int, nbytes: int) -> np.ndarray:
    """uint8[nbytes]: bytes [byte_offset, byte_offset + nbytes) of stream `seed`."""
    w0 = byte_offset >> 3
    w1 = (byte_offset + nbytes + 7) >> 3
    z = splitmix64_at(seed, np.arange(w0, w1, dtype=np.uint64))
    b = z.astype("<u8").view(np.uint8)
    s = byte_offset - (w0 << 3)
    return b[s : s + nbytes].copy()


def synthetic_images(seed: int, start: int, n: int, h: int, w: int) -> np.ndarray:
    """uint8[n, h, w, 3] RGB images start..start+n of the synthetic image stream.

    Pure i.i.d. noise images all pool to the same embedding, so each image gets its own per-channel
    brightness window: px = lo + ((noise * span) >> 8), with (lo, span) drawn per (image, channel)
    from stream seed ^ 0xC0FFEE.  Integer arithmetic only -> identical everywhere.
    """
    per = h * w * 3
    noise = fill_synthetic(seed, start * per, n * per).reshape(n, h, w, 3).astype(np.uint32)
    z = splitmix64_at(seed ^ 0xC0FFEE, np.arange(start * 3, (start + n) * 3, dtype=np.uint64)).reshape(n, 1, 1, 3)
    lo = (z & np.uint64(0xFF)).astype(np.uint32) * 3 // 4  # 0..191
    span = ((z >> np.uint64(8)) & np.uint64(0xFF)).astype(np.uint32) // 4 + 1  # 1..64
    return np.minimum(lo + ((noise * span) >> 8), 255).astype(np.uint8)


def synthetic_scenes(seed: int, start: int, n: int, h: int, w: int, grid: int = 4) -> np.ndarray:
    """uint8[n, h, w, 3]: images start..start+n of the STRUCTURED synthetic stream (bench.py's end-to-end leg).

    `synthetic_images` gives an image 6 numbers of its own (a brightness window per channel), so a million of them fall onto
    a six-dimensional family of hashes with ~40 % exact duplicates.  Here every cell of a grid x grid partition of the image
    has its own window per channel (grid * grid * 6 numbers per image: 96 at grid 4): px = lo + ((noise * span) >> 8) with
    (lo, span) drawn per (image, cell, channel) from stream seed ^ 0xC0FFEE at index ((image * grid + cy) * grid + cx) * 3 + c.
    Same noise stream as synthetic_images; integer arithmetic only (the device generator pb_fill_synthetic_scenes and this
    function give identical bytes).  h and w must be multiples of grid.
    """
    per = h * w * 3
    noise = fill_synthetic(seed, start * per, n * per).reshape(n, h, w, 3).astype(np.uint32)
    cells = grid * grid
    z = splitmix64_at(seed ^ 0xC0FFEE, np.arange(start * cells * 3, (start + n) * cells * 3, dtype=np.uint64)).reshape(n, grid, grid, 3)
    lo = (z & np.uint64(0xFF)).astype(np.uint32) * 3 // 4  # 0..191
    span = ((z >> np.uint64(8)) & np.uint64(0xFF)).astype(np.uint32) // 4 + 1  # 1..64
    lo = np.repeat(np.repeat(lo, h // grid, axis=1), w // grid, axis=2)
    span = np.repeat(np.repeat(span, h // grid, axis=1), w // grid, axis=2)
    return np.minimum(lo + ((noise * span) >> 8), 255).astype(np.uint8)
