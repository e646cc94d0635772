#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/.  Run HERE (build container) only:

    python tests/golden/gen_golden.py

Sources of truth, in order of authority:
  1. the reference's own known answers (engine.rs:703-708, README.md:54, engine.rs:693-701) -- asserted
     below before anything is written;
  2. oracle/numpy_oracle.py (numpy-f32 restatement of engine.rs:572-588 / efficientnet.rs:39) driven
     through Python's sqlite3 with the reference's literal SQL (engine.rs:375-381) -> top-100 fixtures;
  3. for the embed network, torch-CPU conv2d on a hand-assembled EfficientNet-B0 (resources/train.py:30-46
     architecture; torchvision is not installed) with the seeded synthetic weights of
     pixelbox_amd/weights.py -- an independent f32 implementation standing in for tract-onnx, which is
     absent.  Embedding floats are therefore "parity unpinned" against the real reference.

The fixtures are data only (inputs by seed or value, expected outputs).
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import numpy_oracle as no  # noqa: E402
from pixelbox_amd import synth, weights as W  # noqa: E402

F32 = np.float32


def kats():
    # --- reference known answers -------------------------------------------------------------
    assert no.cosine_distance(bytes([255, 0]), bytes([255, 0])) < F32(1e-6)  # engine.rs:705
    assert no.cosine_distance(bytes([0, 255]), bytes([0, 255])) < F32(1e-6)  # engine.rs:706
    assert no.cosine_distance(bytes([255, 0]), bytes([0, 255])) > F32(2.0)  # engine.rs:707
    q = no.quantize(np.array([-1.0, 1.0, 0.0, 0.1], dtype=np.float32))
    assert q.tolist() == [0x00, 0xFF, 0x80, 0x8C], q  # README.md:54
    lut = no.dequant_lut()
    cos_cases = [
        ([255, 0], [255, 0]),
        ([0, 255], [0, 255]),
        ([255, 0], [0, 255]),
        ([128, 128], [128, 128]),
        ([127, 127, 127], [128, 128, 128]),
        ([0, 0, 0, 0], [255, 255, 255, 255]),
        ([200, 10, 77, 128, 254], [201, 12, 70, 127, 250]),
        ([1, 2, 3], [1, 2, 3, 4, 5]),  # mismatched lengths: zip-truncated dot, full norms (engine.rs:581,585)
        ([], []),  # magnitude < 1e-6 -> 0.0 (engine.rs:582-584)
        ([255] * 256, [255] * 255 + [254]),
    ]
    qvals = [-1.0, 1.0, 0.0, 0.1, -0.005, -0.0079, -0.999, 0.9921875, 0.99, 2.0, -2.0, 1e-9, -1e-9,
             0.0078125, -0.0078125, 0.00781, float("nan"), float("inf"), float("-inf"), 0.5, -0.5]
    out = {
        "source": "engine.rs:572-588,703-708; efficientnet.rs:39; README.md:54 (restated: oracle/numpy_oracle.py)",
        "lut_bits": [int(x) for x in lut.view(np.uint32)],
        "cosine": [
            {"a": a, "b": b, "dist_bits": int(np.float32(no.cosine_distance(bytes(a), bytes(b))).view(np.uint32)),
             "dist": float(no.cosine_distance(bytes(a), bytes(b)))}
            for a, b in cos_cases
        ],
        "quantize": [
            {"f_bits": int(np.float32(v).view(np.uint32)), "f": repr(v), "u8": int(no.quantize(np.array([v], dtype=np.float32))[0])}
            for v in qvals
        ],
        # engine.rs:693-701 -- the reference's own exact equalities
        "hamming": [
            {"a": [0], "b": [0xFF], "d": 1.0}, {"a": [0x0F], "b": [0xFF], "d": 0.5}, {"a": [0], "b": [0], "d": 0.0},
            {"a": [0b10101010], "b": [0b01010101], "d": 1.0},
            {"a": [0b10101010, 0b01010101], "b": [0b01010101, 0b10101010], "d": 1.0},
            {"a": [0xFF, 0x0F], "b": [0x0F, 0x0F], "d": 0.25},
        ],
    }
    with open(os.path.join(HERE, "kats.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("kats.json written; cosine KATs:", [c["dist"] for c in out["cosine"][:3]])


def clustered_rows(n: int, d: int, query: np.ndarray, rng: np.random.Generator) -> np.ndarray:
    """Embedding-like bytes: quantise(tanh(N(0,1)*0.5)), with planted exact/near duplicates of the query."""
    rows = no.quantize(np.tanh(rng.standard_normal((n, d)).astype(np.float32) * F32(0.5)))
    dup = rng.choice(n, size=40, replace=False)
    rows[dup[:12]] = query  # exact duplicates -> identical (possibly negative) distances, tie-break by id
    for j in dup[12:30]:  # near duplicates: a few bytes off by one
        r = query.copy()
        idx = rng.choice(d, size=int(rng.integers(1, 6)), replace=False)
        r[idx] = np.clip(r[idx].astype(np.int32) + rng.choice([-1, 1], size=idx.size), 0, 255).astype(np.uint8)
        rows[j] = r
    for j in dup[30:]:  # anti-correlated rows -> the 999999 plateau
        rows[j] = 255 - query
    return rows


def scan_fixtures():
    d = 256
    # (1) uniform random, inputs regenerable from seeds (SURVEY 8d config-2 seeds, 4096-row prefix)
    n = 4096
    rows = no.fill_synthetic(0x5EED0002, 0, n * d).reshape(n, d)
    query = no.fill_synthetic(0x5EED0003, 0, d)
    ids = np.arange(1, n + 1, dtype=np.int64) * 3 + 7  # non-trivial, increasing image ids
    cases = {}
    for name, md in (("md1e3", 1e3), ("md5", 5.0), ("md2e6", 2e6)):
        sid, sdist = no.sqlite_reference_query(query, rows, ids, md)
        oid, odist = no.scan_topk(query, rows, ids, 100, md)
        assert np.array_equal(sid, oid) and np.array_equal(sdist.view(np.uint32), odist.view(np.uint32)), name
        cases[name] = (md, sid, sdist)
    np.savez_compressed(
        os.path.join(HERE, "scan_uniform_4k.npz"),
        seed_rows=np.uint64(0x5EED0002), seed_query=np.uint64(0x5EED0003), n=np.int64(n), d=np.int64(d), ids=ids,
        **{f"{k}_max_dist": np.float64(v[0]) for k, v in cases.items()},
        **{f"{k}_ids": v[1] for k, v in cases.items()},
        **{f"{k}_dist": v[2] for k, v in cases.items()},
    )
    print("scan_uniform_4k: results per case", {k: len(v[1]) for k, v in cases.items()})

    # (1b) tiny table: fewer than 100 rows beat the 999999 plateau, so with max_dist > 999999 the tail of the
    # result is plateau rows in image_id order (SURVEY Appendix A "Query semantics")
    n1 = 150
    sid, sdist = no.sqlite_reference_query(query, rows[:n1], ids[:n1], 2e6)
    oid, odist = no.scan_topk(query, rows[:n1], ids[:n1], 100, 2e6)
    assert np.array_equal(sid, oid) and np.array_equal(sdist.view(np.uint32), odist.view(np.uint32))
    assert (sdist == F32(999999.0)).sum() > 5
    np.savez_compressed(os.path.join(HERE, "scan_uniform_150_plateau.npz"), seed_rows=np.uint64(0x5EED0002),
                        seed_query=np.uint64(0x5EED0003), n=np.int64(n1), d=np.int64(d), ids=ids[:n1],
                        max_dist=np.float64(2e6), out_ids=sid, out_dist=sdist)
    print("scan_uniform_150_plateau: plateau rows in result:", int((sdist == F32(999999.0)).sum()))

    # (2) clustered with duplicates / near-ties / plateau; rows stored (not seed-derivable)
    rng = np.random.default_rng(20261002)
    n2 = 2048
    q2 = no.quantize(np.tanh(rng.standard_normal(d).astype(np.float32) * F32(0.5)))
    rows2 = clustered_rows(n2, d, q2, rng)
    ids2 = np.sort(rng.choice(10 * n2, size=n2, replace=False)).astype(np.int64) + 1
    cases2 = {}
    for name, md in (("md1e3", 1e3), ("md1e-3", 1e-3), ("md2e6", 2e6)):
        sid, sdist = no.sqlite_reference_query(q2, rows2, ids2, md)
        oid, odist = no.scan_topk(q2, rows2, ids2, 100, md)
        assert np.array_equal(sid, oid) and np.array_equal(sdist.view(np.uint32), odist.view(np.uint32)), name
        cases2[name] = (md, sid, sdist)
    np.savez_compressed(
        os.path.join(HERE, "scan_clustered_2k.npz"), query=q2, rows=rows2, ids=ids2,
        **{f"{k}_max_dist": np.float64(v[0]) for k, v in cases2.items()},
        **{f"{k}_ids": v[1] for k, v in cases2.items()},
        **{f"{k}_dist": v[2] for k, v in cases2.items()},
    )
    print("scan_clustered_2k: results per case", {k: len(v[1]) for k, v in cases2.items()},
          "first dists", cases2["md1e3"][2][:14])


# ---- embed: torch-CPU stand-in for tract ------------------------------------------------------------
def torch_b0_forward(blob: bytes, imgs_u8: np.ndarray) -> np.ndarray:
    import torch
    import torch.nn.functional as F

    torch.set_num_threads(4)
    h, w, d, t = W.parse_blob(blob)
    T = {k: torch.from_numpy(v.copy()) for k, v in t.items()}
    # efficientnet.rs:19-29: NCHW, value = px as f32 / 255.0
    x = torch.from_numpy(imgs_u8.astype(np.float32)).div(255.0).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        x = F.silu(F.conv2d(x, T["stem.w"], T["stem.b"], stride=2, padding=1))
        for i, b in enumerate(W.blocks()):
            p = f"b{i}."
            y = x
            if b.has_expand:
                y = F.silu(F.conv2d(y, T[p + "expand.w"][:, :, None, None], T[p + "expand.b"]))
            y = F.silu(F.conv2d(y, T[p + "dw.w"][:, None], T[p + "dw.b"], stride=b.stride,
                                padding=(b.kernel - 1) // 2, groups=b.expanded))
            s = y.mean(dim=(2, 3), keepdim=True)
            s = F.silu(F.conv2d(s, T[p + "se_reduce.w"][:, :, None, None], T[p + "se_reduce.b"]))
            s = torch.sigmoid(F.conv2d(s, T[p + "se_expand.w"][:, :, None, None], T[p + "se_expand.b"]))
            y = y * s
            y = F.conv2d(y, T[p + "project.w"][:, :, None, None], T[p + "project.b"])
            x = x + y if b.residual else y
        x = F.silu(F.conv2d(x, T["head.w"][:, :, None, None], T["head.b"]))
        x = x.mean(dim=(2, 3))
        x = torch.tanh(F.linear(x, T["fc.w"], T["fc.b"]))
    return x.numpy().astype(np.float32)


def embed_fixtures():
    # one weight seed for every config (the _CALIB table in pixelbox_amd/weights.py belongs to it);
    # H, W, D vary: the reference is parametric in all three, non-square included.
    configs = [("embed_128_256", 128, 128, 256, 6), ("embed_64x96_16", 64, 96, 16, 6), ("embed_32_8", 32, 32, 8, 6)]
    for name, h, w, d, n in configs:
        seed = synth.SEED_WEIGHTS
        blob = W.synthetic_blob(seed, h, w, d)
        imgs = synth.synthetic_images(synth.SEED_IMAGES, 0, n, h, w)
        imgs[2] = 0  # flat black
        imgs[3] = 255  # flat white
        f = torch_b0_forward(blob, imgs)
        u8 = no.quantize(f)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), weights_seed=np.uint64(seed), image_seed=np.uint64(synth.SEED_IMAGES),
                            h=np.int64(h), w=np.int64(w), d=np.int64(d), n=np.int64(n), flat_black=np.int64(2), flat_white=np.int64(3),
                            out_f32=f, out_u8=u8)
        print(name, "f32 range", float(f.min()), float(f.max()), "std", float(f.std()),
              "distinct bytes", len(np.unique(u8)), "img0/img1 byte agreement", float((u8[0] == u8[1]).mean()))


if __name__ == "__main__":
    what = sys.argv[1:] or ["kats", "scan", "embed"]
    if "kats" in what:
        kats()
    if "scan" in what:
        scan_fixtures()
    if "embed" in what:
        embed_fixtures()
