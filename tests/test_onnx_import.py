"""CPU tests of the weights-only ONNX importer.  No real `image_similarity.onnx` exists here
(git-ignored in the reference), so the tests write ONNX files shaped like torch's opset-11 export of
resources/train.py's model (tests/onnx_writer.py) and require the importer to reproduce the PBXW0001 blob
bit for bit, and to reject anything that is not that architecture."""
import numpy as np
import pytest

from onnx_writer import export_like_torch
from pixelbox_amd import onnx_import, synth
from pixelbox_amd import weights as W


@pytest.mark.parametrize("h,w,d", [(128, 128, 256), (224, 224, 8), (64, 96, 16)])
@pytest.mark.parametrize("raw,gemm", [(True, True), (False, True), (True, False)])
def test_roundtrip_bit_exact(h, w, d, raw, gemm):
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, h, w, d)
    onnx_bytes = export_like_torch(blob, raw=raw, gemm=gemm)
    got = onnx_import.import_onnx(onnx_bytes)
    assert got == blob


def test_dynamic_input_shape_needs_explicit_size():
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 16)
    onnx_bytes = export_like_torch(blob)
    # same file, but the caller overrides the recorded size (train.py exports 224x224; README.md:58 says 128x128)
    got = onnx_import.import_onnx(onnx_bytes, h=96, w=160)
    hh, ww, d, t = W.parse_blob(got)
    assert (hh, ww, d) == (96, 160, 16)
    assert np.array_equal(t["fc.w"], W.parse_blob(blob)[3]["fc.w"])


def test_rejects_other_architectures():
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 16)
    with pytest.raises(onnx_import.OnnxImportError, match="BatchNormalization"):
        onnx_import.import_onnx(export_like_torch(blob, with_bn=True))
    with pytest.raises(onnx_import.OnnxImportError, match="Conv nodes"):
        onnx_import.import_onnx(export_like_torch(blob, drop_conv=True))
    with pytest.raises(onnx_import.OnnxImportError):
        onnx_import.import_onnx(b"\x08\x06garbage")
    # a conv with the wrong shape
    bad = bytearray(W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 16))
    t = export_like_torch(bytes(bad))
    t = t.replace(b"onnx::Conv_900", b"onnx::Conv_9XX", 1)  # break the stem's weight reference on its node
    with pytest.raises(onnx_import.OnnxImportError):
        onnx_import.import_onnx(t)


@pytest.mark.parametrize("order", ["reversed", 1, 2])
def test_node_order_in_the_file_does_not_matter(order):
    # Conv nodes are matched by dependency order, not by their position in the node list
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 64, 64, 16)
    assert onnx_import.import_onnx(export_like_torch(blob, node_order=order, inits_first=(order == 2))) == blob
    # a dangling input is reported, not mis-assigned
    broken = export_like_torch(blob, node_order=order).replace(b"conv2_m", b"conv2_X", 1)
    with pytest.raises(onnx_import.OnnxImportError):
        onnx_import.import_onnx(broken)


@pytest.mark.gpu
def test_imported_model_embeds_like_the_blob_it_came_from(tmp_path):
    # import -> pb_embed_create -> embed: the same bits as the embedder built from the original PBXW0001 blob, and the
    # oracle's values (the "user-moddable model" path, README.md:58, through the GPU)
    from oracle import capi as oracle
    from pixelbox_amd import capi

    from embed_tol import assert_bytes_match, assert_embeddings_close

    blob = W.synthetic_blob(synth.SEED_WEIGHTS + 5, 128, 128, 256)
    path = tmp_path / "image_similarity.onnx"
    path.write_bytes(export_like_torch(blob, node_order=3))
    out = tmp_path / "image_similarity.pbxw"
    assert onnx_import.main(["onnx_import", str(path), str(out)]) == 0
    imported = out.read_bytes()
    assert imported == blob
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 900, 6, 128, 128)
    u8_a, f_a = capi.Embedder(imported, max_batch=8).embed(imgs)
    u8_b, f_b = capi.Embedder(blob, max_batch=8).embed(imgs)
    assert np.array_equal(f_a.view(np.uint32), f_b.view(np.uint32)) and np.array_equal(u8_a, u8_b)
    ref_u8, ref_f = oracle.mlhash_batch(imported, imgs, 256, nthreads=4)
    assert_embeddings_close(f_a, ref_f)
    assert_bytes_match(u8_a, ref_u8, ref_f)
