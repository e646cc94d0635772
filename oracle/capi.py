"""ctypes loader for oracle/libpb_oracle.so -- CPU ORACLE, TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from
pixelbox_amd/.  `build()` compiles the C restatement with gcc (oracle/Makefile).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpb_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("pb_oracle.c", "pb_oracle_effnet.c", "pb_oracle_effnet_f64.c", "pb_oracle_effnet_body.h", "pb_oracle_resize.c", "pb_oracle_phash.c", "pb_oracle_sqlite.c", "pb_oracle.h")]
    stale = not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpb_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, f32p, i64p = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int64)
        L.pbo_dequant_lut.argtypes = [f32p]
        L.pbo_cosine_distance.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t]
        L.pbo_cosine_distance.restype = C.c_float
        L.pbo_cosine_similarity.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t]
        L.pbo_cosine_similarity.restype = C.c_float
        L.pbo_byte_distance.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t]
        L.pbo_byte_distance.restype = C.c_float
        L.pbo_hamming_distance.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t]
        L.pbo_hamming_distance.restype = C.c_float
        L.pbo_quantize.argtypes = [f32p, C.c_size_t, u8p]
        L.pbo_scan_topk.argtypes = [u8p, u8p, i64p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, i64p, f32p]
        L.pbo_scan_topk.restype = C.c_size_t
        L.pbo_scan_topk_metric.argtypes = [C.c_int, u8p, u8p, i64p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, i64p, f32p]
        L.pbo_scan_topk_metric.restype = C.c_size_t
        L.pbo_scan_all.argtypes = [u8p, u8p, C.c_size_t, C.c_size_t, f32p]
        L.pbo_fill_synthetic.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, u8p]
        L.pbo_effnet_forward.argtypes = [u8p, C.c_size_t, u8p, f32p]
        L.pbo_effnet_forward.restype = C.c_int
        L.pbo_mlhash_batch.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, C.c_int, u8p, f32p]
        L.pbo_mlhash_batch.restype = C.c_int
        L.pbo_effnet_batch_f64.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
        L.pbo_effnet_batch_f64.restype = C.c_int
        L.pbo_resize_to_fill_rgb8.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u8p]
        L.pbo_resize_to_fill_rgb8.restype = C.c_int
        L.pbo_resize_dimensions_fill.argtypes = [C.c_uint32] * 4 + [C.POINTER(C.c_uint32)] * 2
        L.pbo_resize_dimensions_fit.argtypes = [C.c_uint32] * 4 + [C.POINTER(C.c_uint32)] * 2
        L.pbo_gaussian_kernel.argtypes = [C.c_float]
        L.pbo_gaussian_kernel.restype = C.c_float
        L.pbo_phash_rgb8.argtypes = [u8p, C.c_uint32, C.c_uint32, u8p, C.POINTER(C.c_uint32), u8p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.pbo_phash_rgb8.restype = C.c_int
        L.pbo_sqlite_scan.argtypes = [u8p, i64p, C.c_size_t, C.c_size_t, u8p, C.c_size_t, C.c_size_t, C.c_double, i64p, f32p,
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
        L.pbo_sqlite_scan.restype = C.c_int
        _lib = L
    return _lib


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i64(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def dequant_lut() -> np.ndarray:
    out = np.empty(256, dtype=np.float32)
    lib().pbo_dequant_lut(_f32(out))
    return out


def _bytes(a) -> np.ndarray:
    return np.ascontiguousarray(np.frombuffer(bytes(a), dtype=np.uint8) if not isinstance(a, np.ndarray) else a, dtype=np.uint8)


def cosine_distance(a, b) -> np.float32:
    a, b = _bytes(a), _bytes(b)
    return np.float32(lib().pbo_cosine_distance(_u8(a), a.size, _u8(b), b.size))


def cosine_similarity(a, b) -> np.float32:
    a, b = _bytes(a), _bytes(b)
    return np.float32(lib().pbo_cosine_similarity(_u8(a), a.size, _u8(b), b.size))


def byte_distance(a, b) -> np.float32:
    a, b = _bytes(a), _bytes(b)
    return np.float32(lib().pbo_byte_distance(_u8(a), a.size, _u8(b), b.size))


def hamming_distance(a, b) -> np.float32:
    a, b = _bytes(a), _bytes(b)
    return np.float32(lib().pbo_hamming_distance(_u8(a), a.size, _u8(b), b.size))


def quantize(f) -> np.ndarray:
    f = np.ascontiguousarray(f, dtype=np.float32)
    out = np.empty(f.shape, dtype=np.uint8)
    lib().pbo_quantize(_f32(f), f.size, _u8(out))
    return out


def scan_topk(query, rows, ids=None, k=100, max_dist=1e3):
    query = np.ascontiguousarray(query, dtype=np.uint8)
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    n, d = rows.shape if rows.ndim == 2 else (0, query.size)
    assert query.size == d
    if ids is not None:
        ids = np.ascontiguousarray(ids, dtype=np.int64)
    out_ids = np.empty(k, dtype=np.int64)
    out_d = np.empty(k, dtype=np.float32)
    cnt = lib().pbo_scan_topk(_u8(query), _u8(rows), _i64(ids) if ids is not None else None, n, d, k,
                              float(max_dist), _i64(out_ids), _f32(out_d))
    return out_ids[:cnt].copy(), out_d[:cnt].copy()


def scan_topk_metric(metric, query, rows, ids=None, k=100, max_dist=1e3):
    query = np.ascontiguousarray(query, dtype=np.uint8)
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    n, d = rows.shape
    if ids is not None:
        ids = np.ascontiguousarray(ids, dtype=np.int64)
    out_ids = np.empty(k, dtype=np.int64)
    out_d = np.empty(k, dtype=np.float32)
    cnt = lib().pbo_scan_topk_metric(metric, _u8(query), _u8(rows), _i64(ids) if ids is not None else None, n, d, k,
                                     float(max_dist), _i64(out_ids), _f32(out_d))
    return out_ids[:cnt].copy(), out_d[:cnt].copy()


def scan_all(query, rows) -> np.ndarray:
    query = np.ascontiguousarray(query, dtype=np.uint8)
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    n, d = rows.shape
    out = np.empty(n, dtype=np.float32)
    lib().pbo_scan_all(_u8(query), _u8(rows), n, d, _f32(out))
    return out


def fill_synthetic(seed: int, byte_offset: int, nbytes: int) -> np.ndarray:
    out = np.empty(nbytes, dtype=np.uint8)
    lib().pbo_fill_synthetic(seed, byte_offset, nbytes, _u8(out))
    return out


def effnet_forward(blob: bytes, img: np.ndarray, d: int) -> np.ndarray:
    b = np.frombuffer(blob, dtype=np.uint8)
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty(d, dtype=np.float32)
    rc = lib().pbo_effnet_forward(_u8(b), b.size, _u8(img), _f32(out))
    if rc:
        raise RuntimeError(f"pbo_effnet_forward rc={rc}")
    return out


def mlhash_batch(blob: bytes, imgs: np.ndarray, d: int, nthreads: int = 4, want_f32: bool = True):
    b = np.frombuffer(blob, dtype=np.uint8)
    imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
    n = imgs.shape[0]
    out = np.empty((n, d), dtype=np.uint8)
    f = np.empty((n, d), dtype=np.float32) if want_f32 else None
    rc = lib().pbo_mlhash_batch(_u8(b), b.size, _u8(imgs), n, nthreads, _u8(out), _f32(f) if want_f32 else None)
    if rc:
        raise RuntimeError(f"pbo_mlhash_batch rc={rc}")
    return out, f


def effnet_batch_f64(blob: bytes, imgs: np.ndarray, d: int, nthreads: int = 4) -> np.ndarray:
    """The embed network in f64 (pb_oracle_effnet_f64.c): [n, d] tanh outputs -- the third point under the 1e-5 bar."""
    b = np.frombuffer(blob, dtype=np.uint8)
    imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
    n = imgs.shape[0]
    out = np.empty((n, d), dtype=np.float64)
    rc = lib().pbo_effnet_batch_f64(_u8(b), b.size, _u8(imgs), n, nthreads, out.ctypes.data_as(C.POINTER(C.c_double)))
    if rc:
        raise RuntimeError(f"pbo_effnet_batch_f64 rc={rc}")
    return out


def resize_to_fill(img: np.ndarray, nw: int, nh: int) -> np.ndarray:
    """image 0.25.x `resize_to_fill(nw, nh, Triangle)` of an RGB8 image [h, w, 3] (restated, unpinned) -> [nh, nw, 3]."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    out = np.empty((nh, nw, 3), dtype=np.uint8)
    if lib().pbo_resize_to_fill_rgb8(_u8(img), w, h, nw, nh, _u8(out)):
        raise RuntimeError("pbo_resize_to_fill_rgb8: empty image")
    return out


def resize_dimensions_fill(w: int, h: int, nw: int, nh: int) -> tuple[int, int]:
    a, b = C.c_uint32(0), C.c_uint32(0)
    lib().pbo_resize_dimensions_fill(w, h, nw, nh, C.byref(a), C.byref(b))
    return a.value, b.value


def phash(rgb: np.ndarray, want_small: bool = False):
    """phash.rs:3-22 for an RGB8 image [h, w, 3] -> hash bytes (<= 32); with want_small also the resized image."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = rgb.shape[:2]
    out = np.zeros(32, dtype=np.uint8)
    small = np.zeros(16 * 16 * 3, dtype=np.uint8)
    n, sw, sh = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    rc = lib().pbo_phash_rgb8(_u8(rgb), w, h, _u8(out), C.byref(n), _u8(small), C.byref(sw), C.byref(sh))
    if rc != 0:
        raise ValueError("empty image")
    if want_small:
        return out[: n.value].copy(), small[: sw.value * sh.value * 3].reshape(sh.value, sw.value, 3).copy()
    return out[: n.value].copy()


def sqlite_scan(queries, rows, ids=None, max_dist=1e3):
    """The reference's query through a real SQLite (schema, registered cosine_distance UDF, literal SQL text, LIMIT 100).
    -> (ids [nq, 100], dist [nq, 100] f32, count [nq], seconds per query)."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    n, d = rows.shape
    q = np.ascontiguousarray(queries, dtype=np.uint8).reshape(-1, d)
    ids = np.arange(1, n + 1, dtype=np.int64) if ids is None else np.ascontiguousarray(ids, dtype=np.int64)
    nq, k = q.shape[0], 100
    out_ids = np.zeros((nq, k), dtype=np.int64)
    out_d = np.zeros((nq, k), dtype=np.float32)
    cnt = np.zeros(nq, dtype=np.uint32)
    secs = C.c_double(0.0)
    rc = lib().pbo_sqlite_scan(_u8(rows), _i64(ids), n, d, _u8(q), nq, k, float(max_dist), _i64(out_ids), _f32(out_d),
                               cnt.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(secs))
    if rc == -1:
        raise RuntimeError("libsqlite3.so.0 is not loadable")
    if rc != 0:
        raise RuntimeError("SQL error (see stderr)")
    return out_ids, out_d, cnt, secs.value
