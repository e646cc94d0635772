"""GPU test of the C++ host mirror (include/pixelbox_host.hpp): a C++ program written against the
reference's own names (mlhash, Engine::insert_image_from_memory, query_by_image_hash_from_image,
get_query_results) is compiled here against libpixelbox_hip.so and its output checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from oracle import capi as oracle
from pixelbox_amd import capi, synth
from pixelbox_amd import weights as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_mirror_end_to_end(tmp_path):
    n, h, w, d = 24, 64, 64, 32
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, h, w, d)
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 0, n, h, w)
    (tmp_path / "w.pbxw").write_bytes(blob)
    (tmp_path / "imgs.u8").write_bytes(imgs.tobytes())
    exe = tmp_path / "demo"
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_mirror_demo.cpp"), "-o", str(exe),
                           "-L", libdir, "-lpixelbox_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = tmp_path / "out.txt"
    subprocess.check_call([str(exe), str(tmp_path / "w.pbxw"), str(tmp_path / "imgs.u8"), str(n), str(out)])
    lines = out.read_text().splitlines()
    assert lines[0] == f"indexed {n}"
    hashes = np.array([list(bytes.fromhex(x)) for x in lines[-n:]], dtype=np.uint8)
    ref_u8, ref_f = oracle.mlhash_batch(blob, imgs, d, nthreads=4)
    from embed_tol import assert_bytes_match

    assert_bytes_match(hashes, ref_u8, ref_f)
    ids = np.arange(1, n + 1, dtype=np.int64)  # rowids 1..n in insertion order
    pos = 1
    for qi in (0, n // 2):
        hdr = lines[pos].split()
        assert hdr[:2] == ["query", str(qi)]
        cnt = int(hdr[3])
        want_ids, want_d = oracle.scan_topk(hashes[qi], hashes, ids, 100, 1e3)
        assert cnt == len(want_ids)
        for j in range(cnt):
            rid, dist, path = lines[pos + 1 + j].split()
            assert int(rid) == want_ids[j]
            assert np.float32(float(dist)) == want_d[j]
            assert path == f"/synthetic/img{int(rid) - 1}.png"
        pos += 1 + cnt
