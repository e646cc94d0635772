"""ctypes binding of include/pixelbox_hip.h -- the test / bench harness' view of the C ABI.

There is no CPU fallback here: if the HIP shared library is missing or there is no GPU, the calls
raise (PixelboxError / OSError).  The CPU oracle under oracle/ is never imported from this package.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PIXELBOX_LIB: another build of the same library (kernel ablations, profiles/*); the default is the in-tree build
LIB_PATH = os.environ.get("PIXELBOX_LIB") or os.path.join(_HERE, "libpixelbox_hip.so")

PB_OK = 0
PB_MAX_K = 256
PB_OPT_SEARCH_PATH = 1
PB_OPT_PROFILE = 2
PB_OPT_STREAM = 3
PB_OPT_MQ_MIN_QUERIES = 9
PB_OPT_MQ_WG_PER_CU = 10
PB_OPT_MQ_PER_CHUNK = 11
PB_OPT_APPEND_ASYNC = 12
PB_OPT_EXACT_QN = 13
PB_OPT_SECOND_CHANCE = 14
PB_OPT_EMBED_STREAM = 3  # pb_embed_set_option: stream handle to launch on (0 = the embedder's own)
PB_OPT_EMBED_STAGE_BYTES = 5
PB_OPT_EMBED_FRONT_SUB = 6
PB_OPT_EMBED_DUAL = 7
PB_ERR_RANGE = -7  # embed: an activation left the domain of the fixed-point squeeze-excite sums (pixelbox_hip.h)  # pb_embed_set_option: bytes per staging slot of pb_embed_stage_*
PB_OPT_EMBED_ASYNC = 4   # pb_embed_set_option: 1 = pb_embed_batch_device returns with the forward pass queued (default 0: waits)
PB_OPT_SCAN_LAUNCH = 8  # 0: one launch per query; 1: queries side by side in one grid; 2 (default): one launch, queries one after the other
PB_METRIC_COSINE, PB_METRIC_BYTE, PB_METRIC_HAMMING = 0, 1, 2

# every symbol include/pixelbox_hip.h declares (tests/test_abi.py checks the header against this list
# and the built library against both)
SYMBOLS = [
    "pb_last_error", "pb_version", "pb_device_count",
    "pb_index_create", "pb_index_create_metric", "pb_index_destroy", "pb_index_size", "pb_index_dim", "pb_index_contains",
    "pb_sharded_create", "pb_sharded_destroy", "pb_sharded_info", "pb_sharded_size", "pb_sharded_load", "pb_sharded_append",
    "pb_sharded_search", "pb_sharded_append_device", "pb_sharded_shard_device", "pb_sharded_contains", "pb_sharded_fill_synthetic", "pb_sharded_set_option", "pb_sharded_get_stats", "pb_topk_merge_packed_device", "pb_index_append", "pb_index_append_device", "pb_index_load",
    "pb_index_search", "pb_index_search_device", "pb_index_search_packed", "pb_topk_merge_packed", "pb_topk_merge", "pb_index_read", "pb_index_fill_synthetic",
    "pb_index_set_option", "pb_index_get_stats",
    "pb_embed_create", "pb_embed_destroy", "pb_embed_info", "pb_embed_batch", "pb_embed_batch_device", "pb_embed_check_range", "pb_mlhash",
    "pb_mlhash_image", "pb_embed_batch_images", "pb_embed_batch_images_device", "pb_embed_stage_acquire", "pb_embed_stage_release", "pb_embed_stage_close",
    "pb_embed_stage_commit", "pb_embed_stage_abort", "pb_resize_to_fill",
    "pb_embed_set_option", "pb_pinned_alloc", "pb_pinned_free", "pb_embed_tune_ms", "pb_embed_get_tuning", "pb_embed_set_tuning", "pb_fill_synthetic", "pb_fill_synthetic_images", "pb_fill_synthetic_scenes",
    "pb_phash_create", "pb_phash_destroy", "pb_phash_image", "pb_phash_batch_images", "pb_phash_small_image",
]


class PixelboxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"pixelbox_hip error {code}: {msg}")
        self.code = code


class ScanStats(C.Structure):
    _fields_ = [("queries", C.c_uint64), ("fast_path", C.c_uint64), ("fallback", C.c_uint64),
                ("profiled_launches", C.c_uint64), ("profiled_ms", C.c_double), ("profiled_bytes", C.c_uint64),
                ("second_chance", C.c_uint64), ("stamp_timeouts", C.c_uint64)]


_lib = None


def lib():
    """Load libpixelbox_hip.so (built by pixelbox_amd.build / __graft_entry__.build). Fails loudly."""
    global _lib
    if _lib is None:
        if os.environ.get("PIXELBOX_NO_TORCH_PRELOAD") != "1":
            # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so).  If this library pulls in
            # /opt/rocm's copy first, a later `import torch` in the same process finds "No HIP GPUs".  Loading
            # torch first makes both share one runtime (same soname).  Harness-only concern: a Rust/C host
            # never has torch in its process.
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} is missing: run `python -m pixelbox_amd.build` (hipcc, gfx950). "
                          "There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        u8p, f32p, i64p, u32p, u64p = (C.POINTER(t) for t in (C.c_uint8, C.c_float, C.c_int64, C.c_uint32, C.c_uint64))
        vp = C.c_void_p
        L.pb_last_error.restype = C.c_char_p
        L.pb_device_count.argtypes = [C.POINTER(C.c_int)]
        L.pb_index_create.argtypes = [C.POINTER(vp), C.c_int, C.c_uint32, C.c_uint64]
        L.pb_index_create_metric.argtypes = [C.POINTER(vp), C.c_int, C.c_uint32, C.c_uint64, C.c_int]
        L.pb_index_destroy.argtypes = [vp]
        L.pb_index_size.argtypes = [vp, u64p]
        L.pb_index_dim.argtypes = [vp, u32p]
        L.pb_index_append.argtypes = [vp, i64p, u8p, C.c_uint64, u64p]
        L.pb_index_append_device.argtypes = [vp, i64p, vp, C.c_uint64]
        L.pb_index_load.argtypes = [vp, i64p, u8p, C.c_uint64]
        L.pb_index_search.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, C.c_double, i64p, f32p, u32p]
        L.pb_index_search_device.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, C.c_double, vp, vp, vp]
        L.pb_index_search_packed.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, C.c_double, vp]
        L.pb_topk_merge_packed.argtypes = [i64p, C.c_uint32, C.c_uint32, C.c_uint32, i64p, f32p, u32p]
        L.pb_topk_merge_packed_device.argtypes = [C.c_int, vp, C.c_uint32, C.c_uint32, C.c_uint32, i64p, f32p, u32p]
        L.pb_index_contains.argtypes = [vp, C.c_int64, C.POINTER(C.c_int)]
        L.pb_sharded_create.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int, C.c_uint32, C.c_uint64]
        L.pb_sharded_destroy.argtypes = [vp]
        L.pb_sharded_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), u64p]
        L.pb_sharded_size.argtypes = [vp, u64p, u64p]
        L.pb_sharded_load.argtypes = [vp, i64p, u8p, C.c_uint64]
        L.pb_sharded_append.argtypes = [vp, i64p, u8p, C.c_uint64, u64p]
        L.pb_sharded_search.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, C.c_double, i64p, f32p, u32p]
        L.pb_sharded_append_device.argtypes = [vp, C.c_int, i64p, vp, C.c_uint64]
        L.pb_sharded_shard_device.argtypes = [vp, C.c_int, C.POINTER(C.c_int)]
        L.pb_sharded_contains.argtypes = [vp, C.c_int64, C.POINTER(C.c_int)]
        L.pb_sharded_fill_synthetic.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_int64]
        L.pb_sharded_set_option.argtypes = [vp, C.c_int, C.c_int64]
        L.pb_sharded_get_stats.argtypes = [vp, C.POINTER(ScanStats), C.c_int]
        L.pb_topk_merge.argtypes = [i64p, f32p, u32p, C.c_uint32, C.c_uint32, C.c_uint32, i64p, f32p, u32p]
        L.pb_index_read.argtypes = [vp, C.c_uint64, C.c_uint64, i64p, u8p]
        L.pb_index_fill_synthetic.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int64]
        L.pb_index_set_option.argtypes = [vp, C.c_int, C.c_int64]
        L.pb_index_get_stats.argtypes = [vp, C.POINTER(ScanStats), C.c_int]
        L.pb_embed_create.argtypes = [C.POINTER(vp), C.c_int, C.c_char_p, C.c_size_t, C.c_uint32]
        L.pb_embed_destroy.argtypes = [vp]
        L.pb_embed_info.argtypes = [vp, u32p, u32p, u32p, u32p]
        L.pb_embed_batch.argtypes = [vp, u8p, C.c_uint32, u8p, f32p]
        L.pb_embed_batch_device.argtypes = [vp, vp, C.c_uint32, vp, vp]
        L.pb_embed_check_range.argtypes = [vp]
        L.pb_mlhash.argtypes = [vp, u8p, u8p, C.c_size_t]
        L.pb_mlhash_image.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, u8p, C.c_size_t]
        L.pb_embed_batch_images.argtypes = [vp, C.POINTER(u8p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32, u8p, C.POINTER(C.c_float)]
        L.pb_embed_batch_images_device.argtypes = [vp, C.POINTER(u8p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32, u8p, C.POINTER(vp)]
        L.pb_embed_stage_acquire.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(vp), C.POINTER(C.c_uint64)]
        L.pb_embed_stage_release.argtypes = [vp, C.c_uint64]
        L.pb_embed_stage_close.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(vp)]
        L.pb_embed_stage_commit.argtypes = [vp, u8p, C.POINTER(vp)]
        L.pb_embed_stage_abort.argtypes = [vp]
        L.pb_resize_to_fill.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, u8p]
        L.pb_embed_set_option.argtypes = [vp, C.c_int, C.c_int64]
        L.pb_pinned_alloc.argtypes = [C.POINTER(vp), C.c_size_t]
        L.pb_pinned_free.argtypes = [vp]
        L.pb_embed_tune_ms.argtypes = [vp, C.POINTER(C.c_double)]
        L.pb_embed_get_tuning.argtypes = [vp, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.pb_embed_set_tuning.argtypes = [vp, u8p, C.c_size_t]
        L.pb_phash_create.argtypes = [C.POINTER(vp), C.c_int]
        L.pb_phash_destroy.argtypes = [vp]
        L.pb_phash_image.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, u8p, C.c_size_t, u32p]
        L.pb_phash_batch_images.argtypes = [vp, C.POINTER(u8p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32, u8p, u32p]
        L.pb_phash_small_image.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, u8p, u32p, u32p]
        L.pb_fill_synthetic.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, vp]
        L.pb_fill_synthetic_images.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, vp]
        L.pb_fill_synthetic_scenes.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, vp]
        _lib = L
    return _lib


def _check(rc: int):
    if rc != PB_OK:
        raise PixelboxError(rc, (lib().pb_last_error() or b"").decode("utf-8", "replace"))


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().pb_device_count(C.byref(n))
    return n.value if rc == PB_OK else 0


def topk_merge(ids: np.ndarray, dist: np.ndarray, counts: np.ndarray, k: int):
    """ids/dist: [n_lists, stride]; counts: [n_lists] -> (ids[m], dist[m]), m <= k.  Host-only."""
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    n_lists, stride = ids.shape
    out_ids = np.empty(max(k, 1), dtype=np.int64)
    out_d = np.empty(max(k, 1), dtype=np.float32)
    cnt = C.c_uint32(0)
    _check(lib().pb_topk_merge(_p(ids, C.c_int64), _p(dist, C.c_float), _p(counts, C.c_uint32), n_lists, stride, k,
                               _p(out_ids, C.c_int64), _p(out_d, C.c_float), C.byref(cnt)))
    return out_ids[: cnt.value].copy(), out_d[: cnt.value].copy()


def topk_merge_packed(gathered: np.ndarray, k: int):
    """gathered: int64 [n_lists, nq, 2k+1] (host) -> (ids [nq, k], dist [nq, k], count [nq]).  Host-only."""
    g = np.ascontiguousarray(gathered, dtype=np.int64)
    n_lists, nq, row = g.shape
    assert row == 2 * k + 1
    ids = np.zeros((nq, k), dtype=np.int64)
    dist = np.zeros((nq, k), dtype=np.float32)
    cnt = np.zeros(nq, dtype=np.uint32)
    _check(lib().pb_topk_merge_packed(_p(g, C.c_int64), n_lists, nq, k, _p(ids, C.c_int64), _p(dist, C.c_float),
                                      _p(cnt, C.c_uint32)))
    return ids, dist, cnt


def pack_results(ids: np.ndarray, dist: np.ndarray, cnt: np.ndarray) -> np.ndarray:
    """Host-side equivalent of the device packing (used by the CPU tests of the collective path)."""
    nq, k = ids.shape
    out = np.zeros((nq, 2 * k + 1), dtype=np.int64)
    out[:, :k] = ids
    out[:, k : 2 * k] = np.ascontiguousarray(dist, dtype=np.float32).view(np.uint32).astype(np.int64)
    out[:, 2 * k] = cnt
    return out


class Index:
    """Device-resident `semantic_hashes` table (reference: engine.rs:48,109,251-256,363-396)."""

    def __init__(self, dim: int, capacity_rows: int, device: int = 0, metric: int = 0):
        self._h = C.c_void_p()
        self.dim = dim
        self.device = device
        _check(lib().pb_index_create_metric(C.byref(self._h), device, dim, capacity_rows, metric))

    def close(self):
        if self._h:
            lib().pb_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        n = C.c_uint64(0)
        _check(lib().pb_index_size(self._h, C.byref(n)))
        return n.value

    def append(self, image_ids, rows) -> int:
        ids = np.ascontiguousarray(image_ids, dtype=np.int64)
        rows = np.ascontiguousarray(rows, dtype=np.uint8).reshape(-1, self.dim)
        assert ids.shape[0] == rows.shape[0]
        stored = C.c_uint64(0)
        _check(lib().pb_index_append(self._h, _p(ids, C.c_int64), _p(rows, C.c_uint8), ids.shape[0], C.byref(stored)))
        return stored.value

    def append_device(self, image_ids, d_rows_ptr: int):
        """Rows already on the device (uint8[n][dim] at d_rows_ptr); ids: host array, ascending and beyond every stored id."""
        ids = np.ascontiguousarray(image_ids, dtype=np.int64)
        _check(lib().pb_index_append_device(self._h, _p(ids, C.c_int64), C.c_void_p(d_rows_ptr), ids.shape[0]))

    def load(self, image_ids, rows):
        ids = np.ascontiguousarray(image_ids, dtype=np.int64)
        rows = np.ascontiguousarray(rows, dtype=np.uint8).reshape(-1, self.dim)
        _check(lib().pb_index_load(self._h, _p(ids, C.c_int64), _p(rows, C.c_uint8), ids.shape[0]))

    def fill_synthetic(self, seed: int, first_row: int, n: int, first_id: int):
        _check(lib().pb_index_fill_synthetic(self._h, seed, first_row, n, first_id))

    def read(self, first: int, n: int):
        ids = np.empty(n, dtype=np.int64)
        rows = np.empty((n, self.dim), dtype=np.uint8)
        _check(lib().pb_index_read(self._h, first, n, _p(ids, C.c_int64), _p(rows, C.c_uint8)))
        return ids, rows

    def search(self, queries, k: int = 100, max_dist: float = 1e3):
        """-> (ids [nq, k] int64, dist [nq, k] f32, count [nq] u32); slots >= count are unspecified."""
        q = np.ascontiguousarray(queries, dtype=np.uint8).reshape(-1, self.dim)
        nq = q.shape[0]
        ids = np.zeros((nq, k), dtype=np.int64)
        dist = np.zeros((nq, k), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        _check(lib().pb_index_search(self._h, _p(q, C.c_uint8), nq, k, float(max_dist), _p(ids, C.c_int64),
                                     _p(dist, C.c_float), _p(cnt, C.c_uint32)))
        return ids, dist, cnt

    def prepared_search(self, queries, k: int = 100, max_dist: float = 1e3):
        """-> (call, ids, dist, count): `call()` is the bare pb_index_search C call on pre-converted arguments, writing into
        the returned arrays -- for timing the library call itself (what a compiled host pays), without this wrapper's
        array allocations and pointer conversions."""
        q = np.ascontiguousarray(queries, dtype=np.uint8).reshape(-1, self.dim)
        nq = q.shape[0]
        ids = np.zeros((nq, k), dtype=np.int64)
        dist = np.zeros((nq, k), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        fn, h = lib().pb_index_search, self._h
        args = (h, _p(q, C.c_uint8), nq, k, C.c_double(float(max_dist)), _p(ids, C.c_int64), _p(dist, C.c_float), _p(cnt, C.c_uint32))

        def call(_keep=(q,)):
            _check(fn(*args))

        return call, ids, dist, cnt

    def search_one(self, query, k: int = 100, max_dist: float = 1e3):
        ids, dist, cnt = self.search(np.asarray(query).reshape(1, -1), k, max_dist)
        return ids[0, : cnt[0]].copy(), dist[0, : cnt[0]].copy()

    def search_device(self, queries, k, max_dist, d_ids_ptr: int, d_dist_ptr: int, d_count_ptr: int):
        q = np.ascontiguousarray(queries, dtype=np.uint8).reshape(-1, self.dim)
        _check(lib().pb_index_search_device(self._h, _p(q, C.c_uint8), q.shape[0], k, float(max_dist),
                                            C.c_void_p(d_ids_ptr), C.c_void_p(d_dist_ptr), C.c_void_p(d_count_ptr)))

    def search_packed(self, queries, k, max_dist, d_packed_ptr: int):
        q = np.ascontiguousarray(queries, dtype=np.uint8).reshape(-1, self.dim)
        _check(lib().pb_index_search_packed(self._h, _p(q, C.c_uint8), q.shape[0], k, float(max_dist), C.c_void_p(d_packed_ptr)))

    def set_option(self, option: int, value: int):
        _check(lib().pb_index_set_option(self._h, option, value))

    def stats(self, reset: bool = False) -> ScanStats:
        s = ScanStats()
        _check(lib().pb_index_get_stats(self._h, C.byref(s), int(reset)))
        return s


class ShardedIndexC:
    """pb_sharded_*: the row-sharded table driven by ONE process through the C ABI (RCCL all-gather inside the library)."""

    def __init__(self, dim: int, capacity_rows: int, device_ids):
        self._h = C.c_void_p()
        self.dim = dim
        devs = (C.c_int * len(device_ids))(*device_ids)
        _check(lib().pb_sharded_create(C.byref(self._h), devs, len(device_ids), dim, capacity_rows))
        self.n_shards = len(device_ids)

    def close(self):
        if self._h:
            lib().pb_sharded_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        n, r, x = C.c_int(0), C.c_int(0), C.c_uint64(0)
        _check(lib().pb_sharded_info(self._h, C.byref(n), C.byref(r), C.byref(x)))
        return {"n_shards": n.value, "uses_rccl": bool(r.value), "n_exchanges": x.value}

    def sizes(self):
        tot = C.c_uint64(0)
        per = np.zeros(self.n_shards, dtype=np.uint64)
        _check(lib().pb_sharded_size(self._h, C.byref(tot), _p(per, C.c_uint64)))
        return tot.value, per

    def __len__(self):
        return self.sizes()[0]

    def load(self, image_ids, rows):
        ids = np.ascontiguousarray(image_ids, dtype=np.int64)
        rows = np.ascontiguousarray(rows, dtype=np.uint8).reshape(-1, self.dim)
        _check(lib().pb_sharded_load(self._h, _p(ids, C.c_int64), _p(rows, C.c_uint8), ids.shape[0]))

    def append(self, image_ids, rows) -> int:
        ids = np.ascontiguousarray(image_ids, dtype=np.int64)
        rows = np.ascontiguousarray(rows, dtype=np.uint8).reshape(-1, self.dim)
        stored = C.c_uint64(0)
        _check(lib().pb_sharded_append(self._h, _p(ids, C.c_int64), _p(rows, C.c_uint8), ids.shape[0], C.byref(stored)))
        return stored.value

    def append_device(self, shard: int, image_ids, d_rows_ptr: int):
        """rows already in the memory of the shard's GPU (an embedder's output): no host round trip"""
        ids = np.ascontiguousarray(image_ids, dtype=np.int64)
        _check(lib().pb_sharded_append_device(self._h, shard, _p(ids, C.c_int64), C.c_void_p(d_rows_ptr), ids.shape[0]))

    def contains(self, image_id: int) -> bool:
        f = C.c_int(0)
        _check(lib().pb_sharded_contains(self._h, int(image_id), C.byref(f)))
        return bool(f.value)

    def shard_device(self, shard: int) -> int:
        d = C.c_int(-1)
        _check(lib().pb_sharded_shard_device(self._h, shard, C.byref(d)))
        return d.value

    def fill_synthetic(self, seed: int, n: int, first_id: int = 1):
        _check(lib().pb_sharded_fill_synthetic(self._h, seed, n, first_id))

    def search(self, queries, k: int = 100, max_dist: float = 1e3):
        q = np.ascontiguousarray(queries, dtype=np.uint8).reshape(-1, self.dim)
        nq = q.shape[0]
        ids = np.zeros((nq, k), dtype=np.int64)
        dist = np.zeros((nq, k), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        _check(lib().pb_sharded_search(self._h, _p(q, C.c_uint8), nq, k, float(max_dist), _p(ids, C.c_int64), _p(dist, C.c_float),
                                       _p(cnt, C.c_uint32)))
        return ids, dist, cnt

    def set_option(self, option: int, value: int):
        _check(lib().pb_sharded_set_option(self._h, option, value))

    def stats(self, reset: bool = False) -> ScanStats:
        s = ScanStats()
        _check(lib().pb_sharded_get_stats(self._h, C.byref(s), int(reset)))
        return s


def topk_merge_packed_device(device: int, d_gathered_ptr: int, n_lists: int, nq: int, k: int):
    """Device-side merge of an all-gathered message block (int64 [n_lists, nq, 2k+1] in device memory) -> host arrays."""
    ids = np.zeros((nq, k), dtype=np.int64)
    dist = np.zeros((nq, k), dtype=np.float32)
    cnt = np.zeros(nq, dtype=np.uint32)
    _check(lib().pb_topk_merge_packed_device(device, C.c_void_p(d_gathered_ptr), n_lists, nq, k, _p(ids, C.c_int64),
                                             _p(dist, C.c_float), _p(cnt, C.c_uint32)))
    return ids, dist, cnt


class Embedder:
    """EfficientNet-B0 embedder behind `image_hashes::mlhash` (reference: efficientnet.rs:10-42)."""

    def __init__(self, weights_blob: bytes, max_batch: int = 512, device: int = 0):
        self._h = C.c_void_p()
        _check(lib().pb_embed_create(C.byref(self._h), device, weights_blob, len(weights_blob), max_batch))
        h, w, d, mb = (C.c_uint32(0) for _ in range(4))
        _check(lib().pb_embed_info(self._h, C.byref(h), C.byref(w), C.byref(d), C.byref(mb)))
        self.h, self.w, self.d, self.max_batch = h.value, w.value, d.value, mb.value
        self.device = device

    def close(self):
        if self._h:
            lib().pb_embed_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def embed(self, rgb: np.ndarray, want_f32: bool = True):
        """rgb: uint8 [n, H, W, 3] -> (u8 [n, D], f32 [n, D] or None)."""
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8).reshape(-1, self.h, self.w, 3)
        n = rgb.shape[0]
        out = np.empty((n, self.d), dtype=np.uint8)
        f = np.empty((n, self.d), dtype=np.float32) if want_f32 else None
        _check(lib().pb_embed_batch(self._h, _p(rgb, C.c_uint8), n, _p(out, C.c_uint8),
                                    _p(f, C.c_float) if want_f32 else None))
        return out, f

    def embed_device(self, d_rgb_ptr: int, n: int, d_out_u8_ptr: int, d_out_f32_ptr: int = 0):
        _check(lib().pb_embed_batch_device(self._h, C.c_void_p(d_rgb_ptr), n, C.c_void_p(d_out_u8_ptr),
                                           C.c_void_p(d_out_f32_ptr) if d_out_f32_ptr else None))

    def check_range(self):
        """PB_OPT_EMBED_ASYNC callers, after their stream wait: raises PB_ERR_RANGE if a completed forward pass left the SE sums' domain"""
        _check(lib().pb_embed_check_range(self._h))

    def tune_ms(self) -> float:
        """host milliseconds this embedder has spent timing kernel forms at first use"""
        v = C.c_double(0.0)
        _check(lib().pb_embed_tune_ms(self._h, C.byref(v)))
        return float(v.value)

    def get_tuning(self) -> bytes:
        n = C.c_size_t(0)
        _check(lib().pb_embed_get_tuning(self._h, None, 0, C.byref(n)))
        buf = np.zeros(max(int(n.value), 1), dtype=np.uint8)
        _check(lib().pb_embed_get_tuning(self._h, buf.ctypes.data_as(C.POINTER(C.c_uint8)), buf.size, C.byref(n)))
        return buf[: int(n.value)].tobytes()

    def set_tuning(self, data: bytes) -> None:
        buf = np.frombuffer(data, dtype=np.uint8)
        _check(lib().pb_embed_set_tuning(self._h, buf.ctypes.data_as(C.POINTER(C.c_uint8)), buf.size))

    def mlhash(self, rgb: np.ndarray) -> np.ndarray:
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8).reshape(self.h, self.w, 3)
        out = np.empty(self.d, dtype=np.uint8)
        _check(lib().pb_mlhash(self._h, _p(rgb, C.c_uint8), _p(out, C.c_uint8), out.size))
        return out

    def mlhash_image(self, rgb: np.ndarray) -> np.ndarray:
        """mlhash of an RGB8 image of ANY size [h, w, 3]: resize_to_fill(W, H, Triangle) on the GPU, then the network."""
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        h, w = rgb.shape[:2]
        out = np.empty(self.d, dtype=np.uint8)
        _check(lib().pb_mlhash_image(self._h, _p(rgb, C.c_uint8), w, h, _p(out, C.c_uint8), out.size))
        return out

    def embed_images(self, images, want_f32: bool = True):
        """images: list of RGB8 arrays [h_i, w_i, 3] of individual sizes -> (u8 [n, D], f32 [n, D] or None)."""
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        n = len(imgs)
        ptrs = (C.POINTER(C.c_uint8) * n)(*[_p(im, C.c_uint8) for im in imgs])
        ws = (C.c_uint32 * n)(*[im.shape[1] for im in imgs])
        hs = (C.c_uint32 * n)(*[im.shape[0] for im in imgs])
        u8 = np.empty((n, self.d), dtype=np.uint8)
        f = np.empty((n, self.d), dtype=np.float32) if want_f32 else None
        _check(lib().pb_embed_batch_images(self._h, ptrs, ws, hs, n, _p(u8, C.c_uint8), _p(f, C.c_float) if want_f32 else None))
        return u8, f

    @staticmethod
    def image_batch_args(images):
        """The (pointers, widths, heights, n, keep-alive list) a pb_embed_batch_images* call takes, built once for a list of RGB8 arrays."""
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        n = len(imgs)
        ptrs = (C.POINTER(C.c_uint8) * n)(*[_p(im, C.c_uint8) for im in imgs])
        ws = (C.c_uint32 * n)(*[im.shape[1] for im in imgs])
        hs = (C.c_uint32 * n)(*[im.shape[0] for im in imgs])
        return ptrs, ws, hs, n, imgs

    def embed_images_device(self, batch_args, host_copy: np.ndarray = None) -> int:
        """pb_embed_batch_images_device: resize + forward of n <= max_batch images of individual sizes; returns the DEVICE pointer of
        the uint8[n][D] hashes (the embedder's own buffer, valid until its next call); host_copy (uint8 [n, D]) receives a copy."""
        ptrs, ws, hs, n, _keep = batch_args
        d = C.c_void_p(0)
        _check(lib().pb_embed_batch_images_device(self._h, ptrs, ws, hs, n, _p(host_copy, C.c_uint8) if host_copy is not None else None, C.byref(d)))
        return int(d.value or 0)

    # -- staging slots (pb_embed_stage_*): the decoder writes its pixels into the embedder's pinned block
    def stage_acquire(self, w: int, h: int):
        """-> (numpy view [h, w, 3] of the room inside the pinned block, ticket), or None when the open batch is full (PB_STAGE_FULL)."""
        px, ticket = C.c_void_p(0), C.c_uint64(0)
        rc = lib().pb_embed_stage_acquire(self._h, w, h, C.byref(px), C.byref(ticket))
        if rc == 1:
            return None
        _check(rc)
        buf = (C.c_uint8 * (w * h * 3)).from_address(px.value)
        return np.frombuffer(buf, dtype=np.uint8).reshape(h, w, 3), int(ticket.value)

    def stage_release(self, ticket: int):
        _check(lib().pb_embed_stage_release(self._h, ticket))

    def stage_close(self):
        """-> (n, generation, widths, heights): the batch is closed; its pixel blocks stay valid until stage_commit returns."""
        n, gen = C.c_uint32(0), C.c_uint32(0)
        ws = (C.c_uint32 * self.max_batch)()
        hs = (C.c_uint32 * self.max_batch)()
        rc = lib().pb_embed_stage_close(self._h, C.byref(n), C.byref(gen), ws, hs, None)
        if rc == 2:  # PB_STAGE_ABORTED: stage_abort took the batch while this call waited for its writers
            return 0, 0, [], []
        _check(rc)
        return int(n.value), int(gen.value), list(ws[: n.value]), list(hs[: n.value])

    def stage_abort(self):
        """Discards the open batch and a batch closed and not committed (a cancelled or failed run calls this)."""
        _check(lib().pb_embed_stage_abort(self._h))

    def stage_commit(self, n: int, host_copy: bool = True):
        """-> (uint8 [n, D] hashes or None, device pointer of the same hashes)"""
        out = np.empty((n, self.d), dtype=np.uint8) if host_copy else None
        d = C.c_void_p(0)
        _check(lib().pb_embed_stage_commit(self._h, _p(out, C.c_uint8) if out is not None else None, C.byref(d)))
        return out, int(d.value or 0)

    def resize_to_fill(self, rgb: np.ndarray) -> np.ndarray:
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        h, w = rgb.shape[:2]
        out = np.empty((self.h, self.w, 3), dtype=np.uint8)
        _check(lib().pb_resize_to_fill(self._h, _p(rgb, C.c_uint8), w, h, _p(out, C.c_uint8)))
        return out

    def set_option(self, option: int, value: int):
        _check(lib().pb_embed_set_option(self._h, option, value))


class PHasher:
    """`image_hashes::phash` (reference: src/image_hashes/phash.rs:3-22) on the GPU."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        _check(lib().pb_phash_create(C.byref(self._h), device))

    def close(self):
        if self._h:
            lib().pb_phash_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def phash(self, rgb: np.ndarray) -> np.ndarray:
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        h, w = rgb.shape[:2]
        out = np.zeros(32, dtype=np.uint8)
        n = C.c_uint32(0)
        _check(lib().pb_phash_image(self._h, _p(rgb, C.c_uint8), w, h, _p(out, C.c_uint8), 32, C.byref(n)))
        return out[: n.value].copy()

    def phash_batch(self, images):
        """pb_phash_batch_images: list of RGB8 arrays of individual sizes -> list of hashes (32 bytes for a square image)."""
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        n = len(imgs)
        ptrs = (C.POINTER(C.c_uint8) * n)(*[_p(im, C.c_uint8) for im in imgs])
        ws = (C.c_uint32 * n)(*[im.shape[1] for im in imgs])
        hs = (C.c_uint32 * n)(*[im.shape[0] for im in imgs])
        out = np.zeros((n, 32), dtype=np.uint8)
        nb = np.zeros(n, dtype=np.uint32)
        _check(lib().pb_phash_batch_images(self._h, ptrs, ws, hs, n, _p(out, C.c_uint8), _p(nb, C.c_uint32)))
        return [out[i, : nb[i]].copy() for i in range(n)]

    def small_image(self, rgb: np.ndarray) -> np.ndarray:
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        h, w = rgb.shape[:2]
        out = np.zeros(16 * 16 * 3, dtype=np.uint8)
        ow, oh = C.c_uint32(0), C.c_uint32(0)
        _check(lib().pb_phash_small_image(self._h, _p(rgb, C.c_uint8), w, h, _p(out, C.c_uint8), C.byref(ow), C.byref(oh)))
        return out[: ow.value * oh.value * 3].reshape(oh.value, ow.value, 3).copy()


def fill_synthetic_device(device: int, seed: int, byte_offset: int, nbytes: int, d_ptr: int):
    _check(lib().pb_fill_synthetic(device, seed, byte_offset, nbytes, C.c_void_p(d_ptr)))


def fill_synthetic_scenes_device(device: int, seed: int, start: int, n: int, h: int, w: int, d_ptr: int, grid: int = 4):
    _check(lib().pb_fill_synthetic_scenes(device, seed, start, n, h, w, grid, C.c_void_p(d_ptr)))


class PinnedBuffer:
    """pb_pinned_alloc'ed host memory as a numpy uint8 array (`.array`): what a decoder would write its pixels into."""

    def __init__(self, nbytes: int):
        self._p = C.c_void_p()
        _check(lib().pb_pinned_alloc(C.byref(self._p), nbytes))
        self.array = np.ctypeslib.as_array(C.cast(self._p, C.POINTER(C.c_uint8)), shape=(nbytes,))

    def close(self):
        if self._p:
            self.array = None
            lib().pb_pinned_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fill_synthetic_images_device(device: int, seed: int, start: int, n: int, h: int, w: int, d_ptr: int):
    """Images [start, start + n) of synth.synthetic_images(seed, ...) written to device memory uint8[n][h][w][3]."""
    _check(lib().pb_fill_synthetic_images(device, seed, start, n, h, w, C.c_void_p(d_ptr)))
