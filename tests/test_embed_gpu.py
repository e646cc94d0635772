"""GPU parity tests for the embed half: HIP EfficientNet-B0 (f32 MFMA) through the C ABI vs the CPU oracle
(oracle/pb_oracle_effnet.c) and the torch-generated golden fixtures.

Bar (embedding floats are "parity unpinned" against tract, which is absent):
  floats within 1e-5 of the oracle on images whose outputs do not saturate (5e-5 on saturating ones, e.g.
  flat black/white, where pre-tanh values are ~10x larger and f32 rounding scales with them -- see
  tests/embed_tol.py); the u8 quantiser (efficientnet.rs:39) bit-exact on
  the GPU's own floats; bytes identical to the oracle's except where the oracle's float sits within the
  float tolerance of a k/128 truncation boundary."""
import os

import numpy as np
import pytest

from oracle import capi as oracle
from oracle import numpy_oracle as no
from pixelbox_amd import capi, synth
from pixelbox_amd import weights as W

pytestmark = pytest.mark.gpu

from embed_tol import assert_bytes_match, assert_embeddings_close  # noqa: E402  (tests/embed_tol.py)


@pytest.mark.parametrize("name", ["embed_128_256", "embed_64x96_16", "embed_32_8"])
def test_golden_fixture(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    h, w, d, n = int(g["h"]), int(g["w"]), int(g["d"]), int(g["n"])
    blob = W.synthetic_blob(int(g["weights_seed"]), h, w, d)
    imgs = synth.synthetic_images(int(g["image_seed"]), 0, n, h, w)
    fb, fw = int(g["flat_black"]), int(g["flat_white"])
    imgs[fb] = 0
    imgs[fw] = 255
    emb = capi.Embedder(blob, max_batch=8)
    assert (emb.h, emb.w, emb.d) == (h, w, d)
    u8, f = emb.embed(imgs)
    assert_embeddings_close(f, g["out_f32"])
    assert np.array_equal(u8, no.quantize(f))  # quantiser bit-exact on the device's own floats
    assert_bytes_match(u8, g["out_u8"], g["out_f32"])


@pytest.mark.parametrize("n,max_batch", [(1, 1), (3, 8), (37, 16), (64, 64)])
def test_batches_vs_oracle(n, max_batch):
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 100, n, 128, 128)
    emb = capi.Embedder(blob, max_batch=max_batch)  # n > max_batch exercises the chunk loop
    u8, f = emb.embed(imgs)
    ref_u8, ref_f = oracle.mlhash_batch(blob, imgs, 256, nthreads=8)
    assert_embeddings_close(f, ref_f)
    assert np.array_equal(u8, no.quantize(f))
    assert_bytes_match(u8, ref_u8, ref_f)
    # image i's embedding does not depend on its batch position or batch size
    u8_b, f_b = emb.embed(imgs[::-1].copy())
    assert np.array_equal(f_b[::-1], f) and np.array_equal(u8_b[::-1], u8)


def test_config3_batch_512_vs_oracle():
    # BASELINE.json configs[2]: the batch the embed rate is quoted on.  The tuner picks kernel forms per (layer,
    # row-count bucket), so batch 512 runs candidates no smaller batch sees: all 512 images against the CPU oracle
    # (1e-5 on the floats / byte rule), the quantiser bit-exact on the device's own floats, and 8 sampled images
    # re-embedded one per call give the same bits.
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 0, 512, 128, 128)
    emb = capi.Embedder(blob, max_batch=512)
    u8, f = emb.embed(imgs)
    ref_u8, ref_f = oracle.mlhash_batch(blob, imgs, 256, nthreads=8)
    assert_embeddings_close(f, ref_f)
    assert np.array_equal(u8, no.quantize(f))
    assert_bytes_match(u8, ref_u8, ref_f)
    one = capi.Embedder(blob, max_batch=1)
    for i in (0, 1, 63, 64, 200, 255, 256, 511):
        u8_1, f_1 = one.embed(imgs[i:i + 1])
        assert np.array_equal(f_1[0].view(np.uint32), f[i].view(np.uint32)), i
        assert np.array_equal(u8_1[0], u8[i]), i
    # the device-pointer entry point (what bench.py times) gives the same bits as the host-buffer one
    import torch

    d_img = torch.from_numpy(imgs).cuda()
    d_u8 = torch.zeros((512, 256), dtype=torch.uint8, device="cuda")
    emb.embed_device(d_img.data_ptr(), 512, d_u8.data_ptr())  # synchronous by default
    assert np.array_equal(d_u8.cpu().numpy(), u8)


@pytest.mark.parametrize("kind", ["bench", "parity"])
def test_hip_embedding_is_as_close_to_the_f64_value_as_the_oracle_is(kind):
    """The third point under the 1e-5 bar: the same network evaluated in f64 (oracle/pb_oracle_effnet_f64.c).

    Both f32 evaluations -- the oracle's naive loops and the HIP path's matrix-core sums -- approximate that value with
    independent rounding, so per image their errors are two draws from the same distribution: over bench.py's 512 images the
    ratio hip / oracle has median 1.0 and reaches 3 on individual images whose oracle draw happened to be small (measured,
    profiles/r05_embed_f64.txt), while the maxima and medians over a set agree to a few per cent.  Asserted, saturated images
    INCLUDED and no carve-out:
      * over the set: max|hip - f64| <= 1.5 max|oracle - f64|, median likewise;
      * per image:    max|hip - f64| <= 1.5 max|oracle - f64| + 1e-6 (the floor is ~8 ulp of an output near 1: what one f32
        evaluation of this network scatters by on an unsaturated image);
      * the same two over the saturated images alone (where tests/embed_tol.py widens the hip-vs-oracle bar to 5e-5: the
        measured worst case there is 1.1e-5 for the HIP path and 1.3e-5 for the oracle).
    bench: bench.py's images (the raw synthetic byte stream); parity: synth.synthetic_images (brightness windows; ~16 % saturate)."""
    n = 512
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    if kind == "bench":
        imgs = synth.fill_synthetic(synth.SEED_IMAGES, 0, n * 128 * 128 * 3).reshape(n, 128, 128, 3)
    else:
        imgs = synth.synthetic_images(synth.SEED_IMAGES, 0, n, 128, 128)
    nthr = min(64, os.cpu_count() or 8)
    emb = capi.Embedder(blob, max_batch=512)
    _, f = emb.embed(imgs)
    _, ref_f = oracle.mlhash_batch(blob, imgs, 256, nthreads=nthr)
    f64 = oracle.effnet_batch_f64(blob, imgs, 256, nthreads=nthr)
    e_hip = np.abs(f.astype(np.float64) - f64).max(axis=1)
    e_orc = np.abs(ref_f.astype(np.float64) - f64).max(axis=1)
    sat = np.abs(f64).max(axis=1) >= 0.999
    assert e_hip.max() <= 1.5 * e_orc.max(), (e_hip.max(), e_orc.max())
    assert np.median(e_hip) <= 1.5 * np.median(e_orc), (np.median(e_hip), np.median(e_orc))
    assert np.all(e_hip <= 1.5 * e_orc + 1e-6), (e_hip[e_hip > 1.5 * e_orc + 1e-6], e_orc[e_hip > 1.5 * e_orc + 1e-6])
    if kind == "parity":
        assert sat.sum() >= 32  # the set does exercise saturation
        assert e_hip[sat].max() <= 1.5 * e_orc[sat].max() and e_hip[sat].max() <= 5e-5, (e_hip[sat].max(), e_orc[sat].max())
    assert e_hip[~sat].max() <= 1e-5  # the absolute bar, against the true value, where no output saturates


def test_embed_device_then_append_device_on_default_streams():
    # The pipeline the header advertises (embed_batch_device -> pb_index_append_device) with NOTHING configured: the
    # embedder and the index each own a non-blocking stream.  pb_embed_batch_device waits for its stream by default,
    # so the hashes are complete when the index's copy (its own stream) and a null-stream copy read them.
    import torch

    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    n, nb = 96, 32
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 40, n, 128, 128)
    emb = capi.Embedder(blob, max_batch=nb)
    want_u8, _ = emb.embed(imgs)
    ix = capi.Index(256, n)
    d_img = torch.from_numpy(imgs).cuda()
    for lo in range(0, n, nb):
        d_u8 = torch.zeros((nb, 256), dtype=torch.uint8, device="cuda")
        emb.embed_device(d_img[lo:lo + nb].data_ptr(), nb, d_u8.data_ptr())
        ix.append_device(np.arange(lo + 1, lo + nb + 1, dtype=np.int64), d_u8.data_ptr())
        assert np.array_equal(d_u8.cpu().numpy(), want_u8[lo:lo + nb])  # plain copy on torch's stream
    ids, rows = ix.read(0, n)
    assert np.array_equal(ids, np.arange(1, n + 1)) and np.array_equal(rows, want_u8)
    # opt-in asynchronous form: correct when producer and consumer share a stream
    s = torch.cuda.Stream()
    emb.set_option(capi.PB_OPT_EMBED_STREAM, s.cuda_stream)
    emb.set_option(capi.PB_OPT_EMBED_ASYNC, 1)
    ix2 = capi.Index(256, n)
    ix2.set_option(capi.PB_OPT_STREAM, s.cuda_stream)
    ix2.set_option(capi.PB_OPT_APPEND_ASYNC, 1)
    bufs = []
    for lo in range(0, n, nb):
        d_u8 = torch.zeros((nb, 256), dtype=torch.uint8, device="cuda")
        bufs.append(d_u8)
        emb.embed_device(d_img[lo:lo + nb].data_ptr(), nb, d_u8.data_ptr())
        ix2.append_device(np.arange(lo + 1, lo + nb + 1, dtype=np.int64), d_u8.data_ptr())  # ids: a temporary, freed on return
    ids2, rows2 = ix2.read(0, n)
    assert np.array_equal(ids2, ids) and np.array_equal(rows2, want_u8)
    got = ix2.search(want_u8[:3], 5, 1e3)
    ref = ix.search(want_u8[:3], 5, 1e3)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1].view(np.uint32), ref[1].view(np.uint32))
    with pytest.raises(capi.PixelboxError):
        emb.set_option(capi.PB_OPT_EMBED_ASYNC, 2)


def test_hash_is_bitwise_independent_of_batch_size():
    # The reference hashes one image per call (efficientnet.rs:31-42), so an image has ONE hash.  Batching must
    # not change it: the autotuned kernel forms differ per batch size (GEMM tile shapes, depthwise strip / rolling
    # / fused-expand kernels, SE partial-sum tiling), and every one of them has to produce the same bits
    # (fixed k-order in the GEMMs, fixed tap order in the depthwise convs, fixed-point SE pooling sums).
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 300, 192, 128, 128)
    big = capi.Embedder(blob, max_batch=192)
    u8, f = big.embed(imgs)
    for mb in (1, 2, 7, 48):
        emb = capi.Embedder(blob, max_batch=mb)
        sub = imgs[: max(3 * mb + 1, 5)]
        u8_s, f_s = emb.embed(sub)
        assert np.array_equal(f_s.view(np.uint32), f[: len(sub)].view(np.uint32)), mb
        assert np.array_equal(u8_s, u8[: len(sub)]), mb


@pytest.mark.parametrize("sub", [16, 40])
def test_front_of_the_network_over_sub_batches_gives_the_same_bits(sub):
    # PB_OPT_EMBED_FRONT_SUB: stem .. block 4 run `sub` images at a time (ragged last group included), blocks 5-15 and
    # the tail over the whole batch.  A host-side loop over the same kernels: same bits as the one-pass form.
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 900, 100, 128, 128)
    emb = capi.Embedder(blob, max_batch=128)
    u8, f = emb.embed(imgs)
    emb.set_option(capi.PB_OPT_EMBED_FRONT_SUB, sub)
    u8_s, f_s = emb.embed(imgs)
    assert np.array_equal(f_s.view(np.uint32), f.view(np.uint32))
    assert np.array_equal(u8_s, u8)
    with pytest.raises(capi.PixelboxError):
        emb.set_option(capi.PB_OPT_EMBED_FRONT_SUB, 129)


@pytest.mark.parametrize("n", [64, 37, 2])
def test_two_half_batches_side_by_side_give_the_same_bits(n):
    # PB_OPT_EMBED_DUAL: from that many images on a forward runs as two halves, the second on a stream and a workspace of the
    # embedder's own (odd sizes: the halves differ by one image and fall back to one after the other when their tuning buckets
    # differ).  Same kernels per image: same bits; repeated calls reuse the second workspace behind the right events.
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 1200, n, 128, 128)
    emb = capi.Embedder(blob, max_batch=64)
    u8, f = emb.embed(imgs)
    emb.set_option(capi.PB_OPT_EMBED_DUAL, 2)
    for _ in range(3):
        u8_d, f_d = emb.embed(imgs)
        assert np.array_equal(f_d.view(np.uint32), f.view(np.uint32))
        assert np.array_equal(u8_d, u8)
    emb.set_option(capi.PB_OPT_EMBED_DUAL, 0)
    u8_s, _ = emb.embed(imgs[: max(1, n // 2)])
    assert np.array_equal(u8_s, u8[: max(1, n // 2)])


def test_mlhash_is_deterministic_like_the_reference_test():
    # efficientnet.rs:54-67: hamming_distance(mlhash(img), mlhash(img)) == 0
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    img = synth.synthetic_images(synth.SEED_IMAGES, 7, 1, 128, 128)[0]
    emb = capi.Embedder(blob, max_batch=4)
    a, b = emb.mlhash(img), emb.mlhash(img)
    assert oracle.hamming_distance(a, b) == 0.0
    u8, _ = emb.embed(img[None])
    assert np.array_equal(a, u8[0])
    ref_u8, ref_f = oracle.mlhash_batch(blob, img[None], 256, nthreads=1)
    assert_bytes_match(a[None], ref_u8, ref_f)


def test_reference_head_shape_224_latent8():
    # the reference HEAD constants: 224x224 input, latent 8 (efficientnet.rs:6-8, train.py:178-182)
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 224, 224, 8)
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 0, 2, 224, 224)
    emb = capi.Embedder(blob, max_batch=2)
    u8, f = emb.embed(imgs)
    ref_u8, ref_f = oracle.mlhash_batch(blob, imgs, 8, nthreads=2)
    assert_embeddings_close(f, ref_f)
    assert_bytes_match(u8, ref_u8, ref_f)


def test_device_pointer_entry_point():
    import torch

    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    n = 5
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 0, n, 128, 128)
    emb = capi.Embedder(blob, max_batch=8)
    ref_u8, ref_f = emb.embed(imgs)
    d_img = torch.from_numpy(imgs).cuda()
    d_u8 = torch.zeros((n, 256), dtype=torch.uint8, device="cuda")
    d_f = torch.zeros((n, 256), dtype=torch.float32, device="cuda")
    emb.set_option(capi.PB_OPT_STREAM, torch.cuda.current_stream().cuda_stream)
    emb.embed_device(d_img.data_ptr(), n, d_u8.data_ptr(), d_f.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_u8.cpu().numpy(), ref_u8) and np.array_equal(d_f.cpu().numpy(), ref_f)


def test_device_image_generator_matches_the_host_definition():
    import torch

    for (h, w, start, n) in [(128, 128, 0, 3), (128, 128, 1000003, 5), (32, 40, 7, 9)]:
        d = torch.zeros((n, h, w, 3), dtype=torch.uint8, device="cuda")
        capi.fill_synthetic_images_device(0, synth.SEED_IMAGES, start, n, h, w, d.data_ptr())
        assert np.array_equal(d.cpu().numpy(), synth.synthetic_images(synth.SEED_IMAGES, start, n, h, w))
    with pytest.raises(capi.PixelboxError):
        capi.fill_synthetic_images_device(0, 1, 0, 1, 3, 3, 1)  # 27 bytes per image: not a multiple of 8


def test_bad_blobs_fail_loudly():
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    for bad in (blob[:-4], b"XXXX" + blob[4:], blob[:40]):
        with pytest.raises(capi.PixelboxError) as ei:
            capi.Embedder(bad, max_batch=2)
        assert ei.value.code == -5  # PB_ERR_FORMAT
    emb = capi.Embedder(blob, max_batch=2)
    with pytest.raises(capi.PixelboxError):
        emb.embed_device(1, 3, 1)  # n > max_batch


@pytest.mark.parametrize("tensor", ["stem.w", "b2.dw.b", "b7.dw.b", "b13.dw.b"])
def test_activation_outside_the_fixed_point_domain_fails_loudly(tensor):
    # the squeeze-excite pooled sums are 2^-24 fixed point with 32-bit addends: |depthwise output| < 128 (pb_embed_common.h).  A
    # model that leaves the domain must get PB_ERR_RANGE from the call -- never a hash computed from a saturated sum -- whichever
    # kernel the layer runs in (stem + block 0; band front; small-map front; whole-block kernel), and the embedder stays usable.
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    h, w, d, tensors = W.parse_blob(blob)
    raw = bytearray(blob)
    pos = W.HEADER_BYTES
    for name, shape, _ in W.tensor_specs(d):
        n = int(np.prod(shape))
        if name == tensor:
            t = np.frombuffer(bytes(raw[pos : pos + 4 * n]), dtype="<f4").copy()
            t = t * np.float32(2000.0) if name == "stem.w" else t + np.float32(500.0)  # SiLU(x + 500) ~ x + 500
            raw[pos : pos + 4 * n] = t.astype("<f4").tobytes()
        pos += 4 * n
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 0, 5, 128, 128)
    for n_img in (1, 5):
        bad = capi.Embedder(bytes(raw), max_batch=8)
        with pytest.raises(capi.PixelboxError) as ei:
            bad.embed(imgs[:n_img])
        assert ei.value.code == capi.PB_ERR_RANGE, (tensor, n_img, str(ei.value))
        with pytest.raises(capi.PixelboxError) as ei:  # the entry points that resize on the GPU
            bad.mlhash_image(np.ascontiguousarray(imgs[0][:100, :90])) if n_img == 1 else bad.embed_images([imgs[i] for i in range(n_img)])
        assert ei.value.code == capi.PB_ERR_RANGE
    good = capi.Embedder(blob, max_batch=8)
    u8, _ = good.embed(imgs)
    with pytest.raises(capi.PixelboxError):
        bad.embed(imgs)
    u8b, _ = good.embed(imgs)  # another embedder's failure leaves this one alone
    assert np.array_equal(u8, u8b)


def test_queued_calls_leave_the_range_flag_to_pb_embed_check_range():
    # PB_OPT_EMBED_ASYNC: a queued pb_embed_batch_device call neither reports nor consumes the out-of-domain flag (its own work is still
    # in flight, and the flag it could see belongs to an earlier batch); the caller asks pb_embed_check_range after ITS stream wait
    # (ADVICE r5).  A good model's check is quiet; a bad batch is reported exactly once.
    import torch

    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    h, w, d, tensors = W.parse_blob(blob)
    raw = bytearray(blob)
    pos = W.HEADER_BYTES
    for name, shape, _ in W.tensor_specs(d):
        n = int(np.prod(shape))
        if name == "stem.w":
            t = np.frombuffer(bytes(raw[pos : pos + 4 * n]), dtype="<f4").copy() * np.float32(2000.0)
            raw[pos : pos + 4 * n] = t.astype("<f4").tobytes()
        pos += 4 * n
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 0, 8, 128, 128)
    d_img = torch.from_numpy(imgs).cuda()
    d_u8 = torch.zeros((8, 256), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    for model, is_bad in ((bytes(raw), True), (blob, False)):
        emb = capi.Embedder(model, max_batch=8)
        emb.set_option(capi.PB_OPT_EMBED_STREAM, s.cuda_stream)
        emb.set_option(capi.PB_OPT_EMBED_ASYNC, 1)
        emb.embed_device(d_img.data_ptr(), 8, d_u8.data_ptr())  # queued: no error from the call itself
        emb.embed_device(d_img.data_ptr(), 8, d_u8.data_ptr())  # nor from the next one
        s.synchronize()
        if is_bad:
            with pytest.raises(capi.PixelboxError) as ei:
                emb.check_range()
            assert ei.value.code == capi.PB_ERR_RANGE
        emb.check_range()  # consumed (or never raised)


def test_one_image_calls_replay_a_graph_and_keep_the_bits(monkeypatch):
    # pb_mlhash / pb_embed_batch with one image replay the whole forward as one hipGraph from the third quiet call on (pb_embed.hip,
    # one_image_graph: a dependent launch is 3.1 us on a stream, 1.9 us as a graph node).  Same bits as the plain launches of an
    # embedder created with PB_NO_GRAPH=1, for every image; the graph is dropped and re-captured when the picks are replaced; a batch
    # call in between does not disturb it; an out-of-domain model still fails the call.
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 4000, 12, 128, 128)
    monkeypatch.setenv("PB_NO_GRAPH", "1")
    plain = capi.Embedder(blob, max_batch=8)
    monkeypatch.delenv("PB_NO_GRAPH")
    g = capi.Embedder(blob, max_batch=8)
    want = [plain.embed(imgs[i : i + 1]) for i in range(12)]
    for rnd in range(2):
        for i in range(12):
            u8, f = g.embed(imgs[i : i + 1])
            assert np.array_equal(u8, want[i][0]) and np.array_equal(f.view(np.uint32), want[i][1].view(np.uint32)), (rnd, i)
            assert np.array_equal(g.mlhash(imgs[i]), want[i][0][0])
            assert np.array_equal(g.mlhash_image(imgs[i]), want[i][0][0])  # exact size: passes through the resize path, same graph for the forward
            odd = np.ascontiguousarray(imgs[i][:101, :77])
            assert np.array_equal(g.mlhash_image(odd), plain.mlhash_image(odd))
        u8b, fb = g.embed(imgs[:5])  # a batch call between one-image calls
        assert np.array_equal(u8b, np.concatenate([w[0] for w in want[:5]]))
        g.set_tuning(g.get_tuning())  # drops the graph: the next calls run plainly, then capture again
    ref_u8, ref_f = oracle.mlhash_batch(blob, imgs, 256, nthreads=8)
    assert_embeddings_close(np.concatenate([w[1] for w in want]), ref_f)
    assert_bytes_match(np.concatenate([w[0] for w in want]), ref_u8, ref_f)


def test_embed_then_search_end_to_end():
    # config 1 in miniature: embed synthetic images, store the hashes, query with image #0's hash -> id(#0) first
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    n = 200
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 0, n, 128, 128)
    emb = capi.Embedder(blob, max_batch=64)
    u8, _ = emb.embed(imgs, want_f32=False)
    ix = capi.Index(256, n)
    ids = np.arange(1, n + 1, dtype=np.int64)
    ix.append(ids, u8)
    got_ids, got_d = ix.search_one(u8[0])
    want_ids, want_d = oracle.scan_topk(u8[0], u8, ids, 100, 1e3)
    assert np.array_equal(got_ids, want_ids) and np.array_equal(got_d.view(np.uint32), want_d.view(np.uint32))
    assert got_ids[0] == 1 and got_d[0] <= 1e-6


@pytest.mark.parametrize("h,w,d,n", [(32, 64, 8, 5), (96, 32, 64, 3), (160, 128, 16, 2), (64, 64, 1024, 4), (256, 256, 32, 1)])
def test_other_input_sizes_and_latent_dims_vs_oracle(h, w, d, n):
    # map sizes from 128 x 128 down to 1 x 2: every kernel form's eligibility rules get exercised (rolling fused
    # front, small-map fused front on 16x16 / 8x8 / 4x4, strip / rolling / LDS depthwise, stem fusion)
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, h, w, d)
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 900, n, h, w)
    emb = capi.Embedder(blob, max_batch=4)
    u8, f = emb.embed(imgs)
    ref_u8, ref_f = oracle.mlhash_batch(blob, imgs, d, nthreads=8)
    assert_embeddings_close(f, ref_f)
    assert np.array_equal(u8, no.quantize(f))
    assert_bytes_match(u8, ref_u8, ref_f)
    big = capi.Embedder(blob, max_batch=64)
    u8b, fb = big.embed(np.concatenate([imgs] * 16))
    assert np.array_equal(fb[:n].view(np.uint32), f.view(np.uint32)) and np.array_equal(fb[-n:].view(np.uint32), f.view(np.uint32))


def test_piece_arithmetic_layers_meet_the_bars_and_are_what_runs(monkeypatch):
    # The project layers of blocks 4-15, the head conv and the Linear are P3 layers (pixelbox_amd/csrc/pb_gemm_p3.h): f32 operands
    # split exactly into three bf16 pieces, the six leading piece products accumulated in f32 by v_mfma_f32_16x16x32_bf16, in one
    # fixed order in every kernel form (tiled GEMM in all its shapes, one-wave form, the whole-block kernel's project phase).
    # The bars are those of the f32 forms; PB_NO_P3 (A/B switch, read at pb_embed_create) puts every layer back on the f32 MFMA
    # chain: that arithmetic meets the bars too, and its bits differ -- i.e. the default embedder really runs the piece arithmetic.
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 300, 24, 128, 128)
    emb = capi.Embedder(blob, max_batch=32)
    u8, f = emb.embed(imgs)
    ref_u8, ref_f = oracle.mlhash_batch(blob, imgs, 256, nthreads=8)
    assert_embeddings_close(f, ref_f)
    assert_bytes_match(u8, ref_u8, ref_f)
    one = capi.Embedder(blob, max_batch=1)
    for i in (0, 7, 23):  # the same bits for every batch size
        _, f1 = one.embed(imgs[i:i + 1])
        assert np.array_equal(f1.view(np.uint32), f[i:i + 1].view(np.uint32))
    monkeypatch.setenv("PB_NO_P3", "1")
    u8c, fc = capi.Embedder(blob, max_batch=32).embed(imgs)
    assert_embeddings_close(fc, ref_f)
    assert_bytes_match(u8c, ref_u8, ref_f)
    assert not np.array_equal(fc.view(np.uint32), f.view(np.uint32))
    assert np.abs(fc - f).max() < 1e-5


@pytest.mark.parametrize("pick", ["1", "2"])
def test_every_kernel_form_gives_the_same_bits(monkeypatch, pick):
    # The embedder times several kernel forms per layer and batch size (tile shapes of the GEMM, one-wave form, fused or
    # separate expand + depthwise, strip / rolling / LDS depthwise, operands in registers or streamed) and keeps the
    # fastest.  PB_TUNE_PICK=1 keeps the SLOWEST candidate instead, 2 a pseudo-random one: whatever mixture of forms
    # results, the embedding must not change in a single bit.
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 500, 12, 128, 128)
    ref = capi.Embedder(blob, max_batch=16)
    u8_ref, f_ref = ref.embed(imgs)
    monkeypatch.setenv("PB_TUNE_PICK", pick)
    for max_batch, n in ((16, 12), (4, 3), (1, 1)):
        other = capi.Embedder(blob, max_batch=max_batch)
        u8, f = other.embed(imgs[:n])
        assert np.array_equal(f.view(np.uint32), f_ref[:n].view(np.uint32))
        assert np.array_equal(u8, u8_ref[:n])


def test_stem_fused_with_first_depthwise_gives_the_same_bits(monkeypatch):
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 700, 6, 128, 128)
    _, f_ref = capi.Embedder(blob, max_batch=8).embed(imgs)
    monkeypatch.setenv("PB_NO_STEM_FUSION", "1")  # k_stem + k_dwconv instead of k_stem_dw
    _, f = capi.Embedder(blob, max_batch=8).embed(imgs)
    assert np.array_equal(f.view(np.uint32), f_ref.view(np.uint32))


def test_host_buffer_call_larger_than_max_batch_pipelines_its_chunks():
    # pb_embed_batch with n > max_batch: chunks go through the two-slot pipeline (input copy, forward and output copy of
    # consecutive chunks on three streams); every image must come out as from a one-chunk call, ragged last chunk included
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 900, 27, 128, 128)
    u8_ref, f_ref = capi.Embedder(blob, max_batch=32).embed(imgs)
    small = capi.Embedder(blob, max_batch=4)
    for _ in range(2):  # the slots are reused across calls
        u8, f = small.embed(imgs)
        assert np.array_equal(f.view(np.uint32), f_ref.view(np.uint32))
        assert np.array_equal(u8, u8_ref)
    u8, _ = small.embed(imgs[:9], want_f32=False)
    assert np.array_equal(u8, u8_ref[:9])


@pytest.mark.parametrize("switch", ["PB_NO_BLOCK_FUSION", "PB_NO_TAIL_FUSION", "PB_NO_GEMM_T", "PB_NO_GEMM_STREAM", "PB_NO_BAND", "PB_FORCE_BAND", "PB_FOLD_SE",
                                    "PB_STEM_RPP"])
def test_round3_kernel_forms_give_the_same_bits(monkeypatch, switch):
    # Each switch (read at pb_embed_create) takes one of round 3's kernel forms out of -- or forces it into -- the forward:
    # the whole-block kernel of the 4 x 4 maps, the pooling / tanh + quantiser epilogues, the fragment-ordered GEMM, the
    # streaming GEMM of the thin early project layers, the LDS-ring front kernel, the squeeze-excite tails; round 5: one stem row per phase in
    # k_stem_dw.  At a batch where the forms are in use (512: one workgroup of the
    # whole-block kernel per CU) and at small ones the embedding must not change in a single bit.
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    imgs = synth.synthetic_scenes(synth.SEED_IMAGES, 4000, 512, 128, 128)
    ref = capi.Embedder(blob, max_batch=512)
    u8_ref, f_ref = ref.embed(imgs)
    assert len(np.unique(u8_ref, axis=0)) > 500  # a real table, not one repeated hash
    monkeypatch.setenv(switch, "1")
    other = capi.Embedder(blob, max_batch=512)
    for n in (512, 65, 2):
        u8, f = other.embed(imgs[:n])
        assert np.array_equal(f.view(np.uint32), f_ref[:n].view(np.uint32)), (switch, n)
        assert np.array_equal(u8, u8_ref[:n])


def test_tuning_picks_can_be_saved_and_restored():
    # pb_embed_get_tuning / pb_embed_set_tuning: the kernel-form picks an embedder measured at first use, restored into a fresh
    # embedder (another max_batch), spare it the timing loops -- a lazily created model's first mlhash runs on whatever thread
    # asks for it (efficientnet.rs:10-14, engine.rs:352-361) -- and change no bit.
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 1200, 12, 128, 128)
    a = capi.Embedder(blob, max_batch=16)
    assert a.tune_ms() == 0.0
    u8_a, f_a = a.embed(imgs)
    _ = a.embed(imgs[:1])
    assert a.tune_ms() > 1.0  # the loops ran (tens of milliseconds)
    saved = a.get_tuning()
    assert saved[:4] == b"PBTN" and len(saved) > 24
    b = capi.Embedder(blob, max_batch=32)
    b.set_tuning(saved)
    u8_b, f_b = b.embed(imgs)
    u8_1, f_1 = b.embed(imgs[:1])
    assert b.tune_ms() == 0.0  # every (layer, batch bucket) it met was covered
    assert np.array_equal(f_b.view(np.uint32), f_a.view(np.uint32)) and np.array_equal(u8_b, u8_a)
    assert np.array_equal(f_1.view(np.uint32), f_a[:1].view(np.uint32))
    assert b.get_tuning() == saved
    # the picks of a full batch as well: forms only the large buckets choose (k_front_band workgroups walking several items)
    big = synth.synthetic_images(synth.SEED_IMAGES + 3, 1200, 512, 128, 128)
    c = capi.Embedder(blob, max_batch=512)
    u8_c, f_c = c.embed(big)
    d = capi.Embedder(blob, max_batch=512)
    d.set_tuning(c.get_tuning())
    u8_d, f_d = d.embed(big)
    assert d.tune_ms() == 0.0
    assert np.array_equal(f_d.view(np.uint32), f_c.view(np.uint32)) and np.array_equal(u8_d, u8_c)
    # a block of another model shape, a truncated block and garbage are refused whole
    other = capi.Embedder(W.synthetic_blob(synth.SEED_WEIGHTS, 96, 96, 64), max_batch=4)
    for bad in (saved, saved[:-8], b"PBTN" + bytes(40)):
        with pytest.raises(capi.PixelboxError) as ei:
            (other if bad is saved else b).set_tuning(bad)
        assert ei.value.code == -5  # PB_ERR_FORMAT
