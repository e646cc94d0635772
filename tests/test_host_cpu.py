"""CPU tests (no GPU): the C-ABI library loads and exports every symbol the header declares, the
host-side merge and sharding logic are correct, the multi-process query path works over gloo
(world_size 2), and the product path fails loudly -- never falls back -- without a GPU."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import capi as oracle
from pixelbox_amd import capi, sharded, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    return capi.device_count() > 0


def test_header_symbols_are_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "pixelbox_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pb_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    L = capi.lib()
    for s in declared:
        assert getattr(L, s) is not None
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH], text=True)
    exported = set(re.findall(r"\bT (pb_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported


def test_library_embeds_gfx950_code_objects():
    data = open(capi.LIB_PATH, "rb").read()
    assert b"gfx950" in data
    assert b"k_scan_filter" in data


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a box without a GPU")
def test_no_gpu_means_loud_failure_not_fallback():
    with pytest.raises(capi.PixelboxError) as ei:
        capi.Index(256, 100)
    assert ei.value.code == -2  # PB_ERR_HIP
    assert "hip" in str(ei.value).lower()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pixelbox_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "pb_oracle" not in src and "libpb_oracle" not in src, f


def _ref_merge(ids, dist, counts, k):
    allv = [(float(dist[g, i]), int(ids[g, i])) for g in range(ids.shape[0]) for i in range(counts[g])]
    allv.sort()
    return np.array([v[1] for v in allv[:k]], dtype=np.int64), np.array([v[0] for v in allv[:k]], dtype=np.float32)


def test_topk_merge_matches_sort():
    rng = np.random.default_rng(3)
    for n_lists, stride, k in ((1, 100, 100), (8, 100, 100), (8, 100, 7), (3, 5, 50), (4, 10, 0)):
        ids = np.zeros((n_lists, stride), dtype=np.int64)
        dist = np.zeros((n_lists, stride), dtype=np.float32)
        counts = rng.integers(0, stride + 1, size=n_lists).astype(np.uint32)
        base = 0
        for g in range(n_lists):
            c = int(counts[g])
            dd = np.sort(rng.choice(np.array([0.0, 0.5, 0.5, 1.25, 3.0, 999999.0, -1.1920929e-07], dtype=np.float32), size=c))
            ii = base + np.arange(c)
            # sorted by (dist, id) within the list
            order = np.lexsort((ii, dd))
            ids[g, :c], dist[g, :c] = ii[order], dd[order]
            base += 1000
        got_i, got_d = capi.topk_merge(ids, dist, counts, k)
        want_i, want_d = _ref_merge(ids, dist, counts, k)
        assert np.array_equal(got_i, want_i) and np.array_equal(got_d, want_d)


def test_shard_ranges_partition_the_table():
    for n in (0, 1, 7, 8, 9, 1000, 10_000_000):
        for world in (1, 2, 3, 4, 8):
            spans = [sharded.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            assert all(lo <= hi for lo, hi in spans)


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import torch, torch.distributed as dist
from oracle import capi as oracle
from pixelbox_amd import sharded, synth

class OracleShard:  # stands in for the HIP shard: the test exercises the collective + merge plumbing
    def __init__(self, d): self.d = d; self.rows = np.zeros((0, d), np.uint8); self.ids = np.zeros(0, np.int64)
    def load(self, ids, rows): self.ids, self.rows = np.asarray(ids), np.asarray(rows)
    def search(self, queries, k, max_dist):
        nq = len(queries)
        I = np.zeros((nq, k), np.int64); D = np.zeros((nq, k), np.float32); C = np.zeros(nq, np.uint32)
        for q in range(nq):
            i, dd = oracle.scan_topk(queries[q], self.rows, self.ids, k, max_dist) if len(self.ids) else (np.zeros(0, np.int64), np.zeros(0, np.float32))
            C[q] = len(i); I[q, :len(i)] = i; D[q, :len(i)] = dd
        return I, D, C

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
d, n = 64, 3001
rows = synth.fill_synthetic(synth.SEED_INDEX, 0, n * d).reshape(n, d).copy()
rows[1500] = rows[10]; rows[2999] = rows[10]            # ties across shards -> image_id order
ids = np.arange(n, dtype=np.int64) * 3 + 1
class OracleSharded(sharded.ShardedIndex):  # the launcher-side class with the HIP shard swapped for the stand-in (no GPU here)
    def _make_shard(self, dim, rows, device): return OracleShard(dim)

sh = OracleSharded(d, n, rank=rank, world=world, group=dist.group.WORLD)
sh.load(ids, rows)
queries = np.stack([rows[10], synth.fill_synthetic(synth.SEED_QUERY, 0, d), 255 - rows[10]])
for k, md in ((100, 1e3), (5, 1e3), (100, 2e6), (100, 1e-3)):
    I, D, C = sh.search(queries, k, md)
    for q in range(len(queries)):
        wi, wd = oracle.scan_topk(queries[q], rows, ids, k, md)
        assert C[q] == len(wi), (k, md, q, C[q], len(wi))
        assert np.array_equal(I[q, :C[q]], wi), (k, md, q)
        assert np.array_equal(D[q, :C[q]].view(np.uint32), wd.view(np.uint32)), (k, md, q)
dist.barrier()
dist.destroy_process_group()
print("RANK", rank, "OK")
"""


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_query_over_gloo(world, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    port = 29500 + (os.getpid() % 2000) + world
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for pp in procs:
                pp.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out[-2000:]
        assert f"RANK {r} OK" in out


def test_option_constants_of_the_python_binding_match_the_header():
    import re

    from pixelbox_amd import capi

    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "pixelbox_hip.h")).read()
    defs = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(PB_(?:OPT|METRIC|ERR)_[A-Z0-9_]+)\s+(-?\d+)\b", hdr)}
    assert defs, "no option constants found in the header"
    checked = 0
    for name, value in vars(capi).items():
        if name.startswith(("PB_OPT_", "PB_METRIC_")) and name in defs:
            assert defs[name] == value, (name, defs[name], value)
            checked += 1
    assert checked >= 8
    # status codes: an enum in the header, not #defines
    enum = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"\b(PB_(?:OK|ERR_[A-Z]+))\s*=\s*(-?\d+)", hdr))
    assert enum.get("PB_OK") == 0 and enum.get("PB_ERR_RANGE") == capi.PB_ERR_RANGE == -7 and len(set(enum.values())) == len(enum)


# ---- bench.py launch logic (`python bench.py --gpus 8` launched plainly must not exit) -------------
def _bench():
    import importlib

    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    return importlib.import_module("bench")


def test_bench_plain_launch_decision_and_commands():
    b = _bench()
    a8, a1 = b.parse(["--gpus", "8"]), b.parse(["--gpus", "1"])
    assert b.needs_plain_launch(a8, {})                                  # no launcher around it: become the parent
    assert not b.needs_plain_launch(a8, {"WORLD_SIZE": "8"})             # a rank under torch.distributed.run
    assert not b.needs_plain_launch(a1, {})                              # the N = 1 driver command runs in-process
    assert b.needs_plain_launch(a1, {"PIXELBOX_FORCE_SPAWN": "1"})       # one-GPU pre-flight of the same route
    assert not b.needs_plain_launch(b.parse(["--gpus", "8", "--in-library-leg"]), {})
    ranks, single = b.plain_launch_commands(a8, ["--gpus", "8", "--steps", "3"], 29999)
    assert ranks[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in ranks
    assert ranks[ranks.index("--master-addr") + 1] == "127.0.0.1" and ranks[ranks.index("--master-port") + 1] == "29999"
    assert ranks[-4:] == ["--gpus", "8", "--steps", "3"] and ranks[-5].endswith("bench.py")
    assert single[-1] == "--in-library-leg" and single[1].endswith("bench.py")
    assert b._last_json_line('RCCL banner\n{"a": 1}\ntrailing\n') == {"a": 1}
    assert b._last_json_line("no json here") is None


def test_bench_plain_launch_without_a_gpu_fails_loudly_not_silently(tmp_path):
    # here (no GPU) both children refuse to run; the parent must relay that as a non-zero exit and an empty stdout --
    # and must do so as a parent of CHILD processes (it never initialises the GPU itself, never execs)
    env = dict(os.environ, PIXELBOX_NO_IN_LIBRARY_LEG="1")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--rows", "1000"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert p.returncode != 0
    assert p.stdout.strip() == b""
    assert b"starting the ranks as a child" in p.stderr and b"needs a GPU" in p.stderr


def test_structured_synthetic_images_are_deterministic_windows_over_the_same_noise():
    a = synth.synthetic_scenes(synth.SEED_IMAGES, 5, 3, 32, 64, 4)
    b = synth.synthetic_scenes(synth.SEED_IMAGES, 0, 8, 32, 64, 4)[5:8]
    assert a.shape == (3, 32, 64, 3) and a.dtype == np.uint8 and np.array_equal(a, b)  # image i does not depend on the call's range
    # inside one cell the pixels of a channel stay inside that cell's window (lo .. lo + span), windows differ between cells
    cell = a[0, :8, :16, 0].astype(int)
    assert cell.max() - cell.min() <= 64
    assert len({(int(a[0, 8 * cy:8 * cy + 8, 16 * cx:16 * cx + 16, 0].min())) for cy in range(4) for cx in range(4)}) > 8
    # grid 1 = one window per (image, channel), drawn from another position of the window stream than synthetic_images uses
    assert synth.synthetic_scenes(synth.SEED_IMAGES, 0, 1, 32, 32, 1).shape == (1, 32, 32, 3)


# ---- INTEGRATION.md's Rust `extern "C"` block against include/pixelbox_hip.h (the block declared
# pb_sharded_shard_device with the wrong arity and nothing parsed it) ---------------------------------------------------
_RUST_SCALARS = {"c_int": "int", "c_char": "char", "c_void": "void", "u8": "uint8_t", "u32": "uint32_t", "u64": "uint64_t",
                 "i64": "int64_t", "f32": "float", "f64": "double", "usize": "size_t", "PbIndex": "pb_index",
                 "PbSharded": "pb_sharded", "PbEmbedder": "pb_embedder", "PbPhasher": "pb_phasher", "PbScanStats": "pb_scan_stats"}


def _rust_type(t):
    """`*const *mut u8` -> ('ptr', True, ('ptr', False, 'uint8_t')): pointer levels with the constness of the pointee"""
    t = t.strip()
    m = re.match(r"\*(const|mut)\s+(.*)$", t, flags=re.S)
    if m:
        return ("ptr", m.group(1) == "const", _rust_type(m.group(2)))
    assert t in _RUST_SCALARS, f"unknown Rust type {t!r}"
    return _RUST_SCALARS[t]


def _c_type(t):
    """`const uint8_t *const *` -> the same canonical form (east-const reading, right to left)"""
    toks = re.findall(r"[A-Za-z_][A-Za-z0-9_]*|\*", t)
    toks = [x for x in toks if x not in ("struct", "enum")]
    # base: everything before the first '*'; const may sit on either side of the type name
    n_base = toks.index("*") if "*" in toks else len(toks)
    base = [x for x in toks[:n_base] if x != "const"]
    assert len(base) == 1, (t, toks)
    cur, cur_const = base[0], "const" in toks[:n_base]
    i = n_base
    while i < len(toks):
        assert toks[i] == "*", (t, toks)
        cur = ("ptr", cur_const, cur)
        cur_const = i + 1 < len(toks) and toks[i + 1] == "const"
        i += 2 if cur_const else 1
    return cur


def _c_prototypes():
    hdr = open(os.path.join(ROOT, "include", "pixelbox_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdr = re.sub(r"^\s*#.*?(?<!\\)$", "", hdr, flags=re.M | re.S)  # directives (with continuation lines)
    protos = {}
    for ret, name, args in re.findall(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(pb_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", hdr):
        params = []
        if args.strip() != "void":
            for a in args.split(","):
                m = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*$", a.strip(), flags=re.S)  # strip the parameter name
                params.append(_c_type(m.group(1)))
        protos[name] = (_c_type(ret), params)
    return protos, hdr


def _rust_prototypes():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r'^extern "C" \{\n(.*?)^\}', md, flags=re.M | re.S)
    assert m, 'no extern "C" block in INTEGRATION.md'
    body = re.sub(r"//[^\n]*", "", m.group(1))
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    protos = {}
    for name, args, ret in re.findall(r"pub fn (pb_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+?))?\s*;", body, flags=re.S):
        params = [_rust_type(a.split(":", 1)[1]) for a in args.split(",") if a.strip()]
        protos[name] = (_rust_type(ret) if ret else "void", params)
    return protos, md


def test_integration_md_extern_block_matches_the_header():
    c, hdr = _c_prototypes()
    r, md = _rust_prototypes()
    assert len(c) > 40 and set(c) == set(capi.SYMBOLS)  # the parser saw the whole header
    assert set(r) == set(c), f"only in INTEGRATION.md: {sorted(set(r) - set(c))}; only in the header: {sorted(set(c) - set(r))}"
    for name in sorted(c):
        assert r[name][0] == c[name][0], f"{name}: return type {r[name][0]} (Rust) vs {c[name][0]} (C)"
        assert len(r[name][1]) == len(c[name][1]), f"{name}: {len(r[name][1])} parameters in INTEGRATION.md, {len(c[name][1])} in the header"
        for i, (a, b) in enumerate(zip(r[name][1], c[name][1])):
            assert a == b, f"{name}: parameter {i} is {a} in INTEGRATION.md, {b} in the header"
    # the one by-value struct of the ABI: field order and widths
    cs = re.search(r"typedef struct pb_scan_stats \{(.*?)\}", hdr, flags=re.S).group(1)
    c_fields = [(_c_type(t), n) for t, n in re.findall(r"([A-Za-z_][A-Za-z0-9_ ]*?)\s+([a-z_]+)\s*;", cs)]
    rs = re.search(r"pub struct PbScanStats \{(.*?)\}", md, flags=re.S).group(1)
    r_fields = [(_rust_type(t), n) for n, t in re.findall(r"pub ([a-z_]+)\s*:\s*([A-Za-z0-9_]+)", rs)]
    assert r_fields == c_fields and len(c_fields) >= 7, (r_fields, c_fields)


def test_the_abi_type_parsers_agree_on_known_spellings():
    assert _c_type("const uint8_t *const *") == _rust_type("*const *const u8") == ("ptr", True, ("ptr", True, "uint8_t"))
    assert _c_type("const uint8_t **") == _rust_type("*mut *const u8")
    assert _c_type("pb_index **") == _rust_type("*mut *mut PbIndex")
    assert _c_type("const pb_sharded *") == _rust_type("*const PbSharded") != _rust_type("*mut PbSharded")
    assert _c_type("int") == _rust_type("c_int") and _c_type("size_t") == _rust_type("usize") != _rust_type("u32")
