"""numpy restatement of the reference's scan arithmetic -- CPU ORACLE, TEST INFRASTRUCTURE ONLY.

Independent of oracle/pb_oracle.c (the two are cross-checked in tests/test_oracle.py).
Used here, in the build container, to generate the committed golden vectors under
tests/golden/ (see tests/golden/gen_golden.py), including a run of the reference's
*literal SQL* (engine.rs:375-381) through Python's sqlite3 with this restatement
registered as the `cosine_distance` scalar function (engine.rs:608-622).

Nothing in the product path (pixelbox_amd/) imports this module.
"""
from __future__ import annotations

import sqlite3

import numpy as np

F32 = np.float32


def dequant_lut() -> np.ndarray:
    """engine.rs:576 -- ((v as f32 / 255.0) * 2.0) - 1.0 for v in 0..255 (f32 at every step)."""
    v = np.arange(256, dtype=np.float32)
    t = (v / F32(255.0)).astype(np.float32)
    t = (t * F32(2.0)).astype(np.float32)
    return (t - F32(1.0)).astype(np.float32)


_LUT = dequant_lut()


def _fold(products: np.ndarray) -> np.float32:
    """Left-to-right f32 fold from 0.0 (Iterator::fold(0f32, ..)); cumsum in f32 is sequential."""
    if products.size == 0:
        return F32(0.0)
    return np.cumsum(products.astype(np.float32), dtype=np.float32)[-1]


def cosine_similarity(a: bytes | np.ndarray, b: bytes | np.ndarray):
    """engine.rs:575-586; returns (cs, degenerate)."""
    a = np.frombuffer(bytes(a), dtype=np.uint8) if not isinstance(a, np.ndarray) else a
    b = np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else b
    xa = _LUT[a]
    xb = _LUT[b]
    sa = _fold((xa * xa).astype(np.float32))
    sb = _fold((xb * xb).astype(np.float32))
    mag = F32(np.sqrt(sa, dtype=np.float32) * np.sqrt(sb, dtype=np.float32))
    if mag < F32(1e-6):
        return F32(0.0), True
    n = min(len(xa), len(xb))
    dot = _fold((xa[:n] * xb[:n]).astype(np.float32))
    return F32(dot / mag), False


def cosine_distance(a, b) -> np.float32:
    """engine.rs:572-588."""
    cs, degenerate = cosine_similarity(a, b)
    if degenerate:
        return F32(0.0)
    m = cs if cs > F32(1e-6) else F32(1e-6)  # f32::max, NaN-ignoring
    return F32(F32(F32(1.0) / m) - F32(1.0))


def cosine_distance_rows(query: np.ndarray, rows: np.ndarray) -> np.ndarray:
    """Vectorised over rows [n, d]; bit-identical to cosine_distance row by row (d == len(query))."""
    q = _LUT[query]
    xb = _LUT[rows]  # [n, d]
    sa = _fold((q * q).astype(np.float32))
    sb = np.cumsum((xb * xb).astype(np.float32), axis=1, dtype=np.float32)[:, -1]
    mag = (np.sqrt(sa, dtype=np.float32) * np.sqrt(sb, dtype=np.float32)).astype(np.float32)
    dot = np.cumsum((q[None, :] * xb).astype(np.float32), axis=1, dtype=np.float32)[:, -1]
    with np.errstate(divide="ignore", invalid="ignore"):
        cs = (dot / mag).astype(np.float32)
    m = np.where(cs > F32(1e-6), cs, F32(1e-6)).astype(np.float32)
    dist = ((F32(1.0) / m).astype(np.float32) - F32(1.0)).astype(np.float32)
    return np.where(mag < F32(1e-6), F32(0.0), dist).astype(np.float32)


def quantize(f: np.ndarray) -> np.ndarray:
    """efficientnet.rs:39 -- 128u8.saturating_add_signed((f*128).max(-128).min(128) as i8)."""
    f = np.asarray(f, dtype=np.float32)
    t = (f * F32(128.0)).astype(np.float32)
    t = np.where(np.isnan(t), F32(-128.0), np.maximum(t, F32(-128.0)))
    t = np.minimum(t, F32(128.0))
    i = np.clip(np.trunc(t), -128, 127).astype(np.int32)  # `as i8`: truncate, saturate
    return np.clip(128 + i, 0, 255).astype(np.uint8)


# -- the reference's SQL, verbatim in structure (engine.rs:48, :63-77, :375-381) -----------------
SCHEMA = [
    "CREATE TABLE images (id INTEGER PRIMARY KEY, filename TEXT, path TEXT UNIQUE, "
    "image_width INTEGER, image_height INTEGER, thumbnail BLOB)",
    "CREATE TABLE semantic_hashes (image_id INTEGER PRIMARY KEY, hash BLOB)",
]
QUERY_SQL = """
    SELECT images.id, semantic_hashes.hash, cosine_distance(?, semantic_hashes.hash) AS dist
    FROM semantic_hashes
    INNER JOIN images images ON images.id = semantic_hashes.image_id
    WHERE dist < ?
    ORDER BY dist ASC
    LIMIT 100"""


def sqlite_reference_query(query: np.ndarray, rows: np.ndarray, ids: np.ndarray, max_dist: float):
    """Run the reference's literal query plan through sqlite3 with the restated UDF.

    Returns (ids int64[m], dists float32[m]) in SQLite's output order (m <= 100).
    """
    conn = sqlite3.connect(":memory:")
    for s in SCHEMA:
        conn.execute(s)
    conn.create_function(
        "cosine_distance", 2, lambda a, b: float(cosine_distance(a, b)), deterministic=True
    )
    conn.executemany(
        "INSERT OR IGNORE INTO images (id, filename, path, image_width, image_height, thumbnail) "
        "VALUES (?, ?, ?, 128, 128, NULL)",
        [(int(i), f"f{int(i)}", f"/p/{int(i)}") for i in ids],
    )
    conn.executemany(
        "INSERT OR IGNORE INTO semantic_hashes (image_id, hash) VALUES (?, ?)",
        [(int(i), rows[j].tobytes()) for j, i in enumerate(ids)],
    )
    out = conn.execute(QUERY_SQL, (query.tobytes(), float(max_dist))).fetchall()
    conn.close()
    got_ids = np.array([r[0] for r in out], dtype=np.int64)
    got_d = np.array([r[2] for r in out], dtype=np.float64).astype(np.float32)
    return got_ids, got_d


def scan_topk(query, rows, ids, k=100, max_dist=1e3):
    """(dist asc, id asc) top-k with the f64 `dist < max_dist` filter -- the documented rule."""
    d = cosine_distance_rows(query, rows)
    keep = np.nonzero(d.astype(np.float64) < float(max_dist))[0]
    order = np.lexsort((ids[keep], d[keep]))[:k]
    sel = keep[order]
    return ids[sel].astype(np.int64), d[sel]


# -- synthetic data (same generator as oracle/pb_oracle.c and the device kernel) -----------------
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix64_at(seed: int, word_index: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (word_index.astype(np.uint64) + np.uint64(1)) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def fill_synthetic(seed: int, byte_offset: int, nbytes: int) -> np.ndarray:
    w0 = byte_offset >> 3
    w1 = (byte_offset + nbytes + 7) >> 3
    z = splitmix64_at(seed, np.arange(w0, w1, dtype=np.uint64))
    b = z.astype("<u8").view(np.uint8)
    s = byte_offset - (w0 << 3)
    return b[s : s + nbytes].copy()
