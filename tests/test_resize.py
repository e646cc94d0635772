"""Pre-processing: `img.resize_to_fill(W, H, FilterType::Triangle)` (efficientnet.rs:20; image crate 0.25.x).

The crate is a third-party dependency that is not under the reference tree and no reference test pins a resized
pixel, so this step is PARITY UNPINNED: the CPU tests check the restatement (oracle/pb_oracle_resize.c) against
hand-computed values of the published algorithm and its invariants, the GPU tests check the HIP kernels against the
restatement bit for bit."""
import numpy as np
import pytest

from oracle import capi as oracle


def _ref_1d(src: np.ndarray, n_out: int) -> np.ndarray:
    """f32 triangle resampling of a 1-D f32 signal, written independently of the C code (numpy scalars)."""
    n_in = len(src)
    f = np.float32
    ratio = f(n_in) / f(n_out)
    sratio = f(1.0) if ratio < f(1.0) else ratio
    out = np.empty(n_out, dtype=np.float32)
    for o in range(n_out):
        inp = (f(o) + f(0.5)) * ratio
        left = int(np.floor(inp - sratio))
        left = min(max(left, 0), n_in - 1)
        right = int(np.ceil(inp + sratio))
        right = min(max(right, left + 1), n_in)
        inp = inp - f(0.5)
        ws = []
        s = f(0.0)
        for i in range(left, right):
            x = abs((f(i) - inp) / sratio)
            w = f(1.0) - x if x < f(1.0) else f(0.0)
            ws.append(f(w))
            s = f(s + w)
        t = f(0.0)
        for i, w in zip(range(left, right), ws):
            t = f(t + f(src[i] * f(w / s)))
        out[o] = t
    return out


def _ref_resize_exact(img: np.ndarray, nw: int, nh: int) -> np.ndarray:
    h, w, _ = img.shape
    if (nw, nh) == (w, h):
        return img.copy()
    tmp = np.empty((nh, w, 3), dtype=np.float32)
    for x in range(w):
        for c in range(3):
            tmp[:, x, c] = _ref_1d(img[:, x, c].astype(np.float32), nh)
    out = np.empty((nh, nw, 3), dtype=np.uint8)
    for y in range(nh):
        for c in range(3):
            t = np.clip(_ref_1d(tmp[y, :, c], nw), 0.0, 255.0)
            out[y, :, c] = np.floor(t + np.float32(0.5)).astype(np.uint8)  # round half away from zero, t >= 0
    return out


def test_resize_dimensions_cover_and_round():
    assert oracle.resize_dimensions_fill(128, 128, 128, 128) == (128, 128)
    assert oracle.resize_dimensions_fill(4000, 3000, 128, 128) == (171, 128)   # 4000 * 128/3000 = 170.67 -> 171
    assert oracle.resize_dimensions_fill(100, 300, 128, 128) == (128, 384)
    assert oracle.resize_dimensions_fill(640, 480, 224, 224) == (299, 224)
    assert oracle.resize_dimensions_fill(3, 1000, 128, 128) == (128, 42667)


def test_identity_constant_and_crop_position():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(128, 128, 3), dtype=np.uint8)
    assert np.array_equal(oracle.resize_to_fill(img, 128, 128), img)           # same size: copied, not resampled
    flat = np.full((300, 200, 3), 77, dtype=np.uint8)
    assert np.all(oracle.resize_to_fill(flat, 128, 128) == 77)                 # normalised weights
    # same width, taller: no resampling at all, rows (h - 128) / 2 .. are kept
    tall = rng.integers(0, 256, size=(200, 128, 3), dtype=np.uint8)
    assert np.array_equal(oracle.resize_to_fill(tall, 128, 128), tall[36:164])
    wide = rng.integers(0, 256, size=(128, 131, 3), dtype=np.uint8)
    assert np.array_equal(oracle.resize_to_fill(wide, 128, 128), wide[:, 1:129])


def test_hand_computed_two_to_one():
    # 4x2 -> 2x1-shaped target (nw = 2, nh = 1): ratio 2 both ways.  Output column 0 of a row [0, 90, 180, 255]:
    # centre 1.0 -> taps 0..2 with triangle weights 0.75, 0.75, 0.25 (sum 1.75): (0*0.75 + 90*0.75 + 180*0.25)/1.75 = 64.29
    row = np.array([0, 90, 180, 255], dtype=np.uint8)
    img = np.repeat(np.stack([row, row])[:, :, None], 3, axis=2)
    out = oracle.resize_to_fill(img, 2, 1)
    assert out.shape == (1, 2, 3)
    # column 1: centre 3.0 -> taps 1..3, weights 0.25, 0.75, 0.75: (22.5 + 135 + 191.25) / 1.75 = 199.29
    assert out[0, 0, 0] == 64 and out[0, 1, 0] == 199


@pytest.mark.parametrize("h,w,nh,nw", [(37, 53, 16, 16), (300, 200, 128, 128), (90, 250, 64, 96), (20, 20, 32, 32), (129, 128, 128, 128)])
def test_c_restatement_equals_independent_numpy_restatement(h, w, nh, nw):
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    w2, h2 = oracle.resize_dimensions_fill(w, h, nw, nh)
    mid = _ref_resize_exact(img, w2, h2)
    if nw * h2 > w2 * nh:
        cy = (h2 - nh) // 2
        want = mid[cy : cy + nh, :nw]
    else:
        cx = (w2 - nw) // 2
        want = mid[:nh, cx : cx + nw]
    assert np.array_equal(oracle.resize_to_fill(img, nw, nh), want)


# ---- GPU -------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("h,w", [(128, 128), (300, 200), (200, 300), (1080, 1920), (97, 4001), (64, 64), (129, 128), (128, 640)])
def test_gpu_resize_equals_the_restatement_bit_for_bit(h, w):
    from pixelbox_amd import capi, synth
    from pixelbox_amd import weights as W

    rng = np.random.default_rng(h + 7 * w)
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    emb = capi.Embedder(W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256), max_batch=4)
    assert np.array_equal(emb.resize_to_fill(img), oracle.resize_to_fill(img, 128, 128))


@pytest.mark.gpu
def test_mlhash_of_any_size_image_is_mlhash_of_the_preprocessed_image():
    from pixelbox_amd import capi, synth
    from pixelbox_amd import weights as W

    rng = np.random.default_rng(5)
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    emb = capi.Embedder(blob, max_batch=3)
    # smooth images (pure noise resizes to flat grey): low-resolution random fields, upsampled by pixel repetition
    imgs = []
    for (h, w) in [(240, 320), (128, 128), (500, 375), (1000, 300), (131, 257)]:
        base = rng.integers(0, 256, size=(h // 16 + 1, w // 16 + 1, 3), dtype=np.uint8)
        imgs.append(np.kron(base, np.ones((16, 16, 1), dtype=np.uint8))[:h, :w].copy())
    pre = np.stack([oracle.resize_to_fill(im, 128, 128) for im in imgs])
    want_u8, want_f = emb.embed(pre)
    got_u8, got_f = emb.embed_images(imgs)          # 5 images through a max_batch of 3: two chunks
    assert np.array_equal(got_u8, want_u8) and np.array_equal(got_f.view(np.uint32), want_f.view(np.uint32))
    assert np.array_equal(emb.mlhash_image(imgs[2]), want_u8[2])
    assert np.array_equal(emb.mlhash_image(imgs[1]), emb.mlhash(imgs[1]))  # already 128x128: untouched
    with pytest.raises(capi.PixelboxError):
        emb.mlhash_image(np.zeros((0, 5, 3), dtype=np.uint8))


@pytest.mark.gpu
def test_image_batches_span_staging_sub_batches_and_chunks_without_changing_a_bit():
    # pb_embed_batch_images[_device] prepare a batch in sub-batches of <= 48 MB of source pixels over two pinned staging slots
    # (pack on the host, one transfer, two resize launches over a descriptor array) and, beyond max_batch images, in chunks
    # whose staging overlaps the previous chunk's forward pass.  80 images of mixed sizes (camera frames, thumbnails, network-size
    # frames that are only cropped, a wide strip, a 1 x 1 image): 80 MB of sources -> several sub-batches, three
    # chunks of 32 -- every hash equals the hash of the image pre-processed on its own by the CPU restatement, and the device
    # form leaves the same bytes in the embedder's output buffer.
    import torch

    from pixelbox_amd import capi, synth
    from pixelbox_amd import weights as W

    rng = np.random.default_rng(11)
    sizes = [(480, 640), (128, 128), (256, 256), (1080, 1920), (100, 1000), (131, 257), (1, 1), (128, 300)]
    imgs = []
    for i in range(80):
        h, w = sizes[i % len(sizes)]
        base = rng.integers(0, 256, size=(h // 16 + 1, w // 16 + 1, 3), dtype=np.uint8)
        imgs.append(np.kron(base, np.ones((16, 16, 1), dtype=np.uint8))[:h, :w].copy())
    assert sum(im.size for im in imgs) > 48 << 20
    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 256)
    pre = np.stack([oracle.resize_to_fill(im, 128, 128) for im in imgs])
    big = capi.Embedder(blob, max_batch=128)
    want_u8, want_f = big.embed(pre)
    emb = capi.Embedder(blob, max_batch=32)
    for _ in range(2):  # the staging slots and events are reused across calls
        got_u8, got_f = emb.embed_images(imgs)
        assert np.array_equal(got_u8, want_u8) and np.array_equal(got_f.view(np.uint32), want_f.view(np.uint32))
    host = np.zeros((80, 256), dtype=np.uint8)
    d_ptr = big.embed_images_device(capi.Embedder.image_batch_args(imgs), host_copy=host)
    assert np.array_equal(host, want_u8)
    idx = capi.Index(256, 128)
    idx.append_device(np.arange(1, 81, dtype=np.int64), d_ptr)  # the device copy is what the index stores
    _, rows = idx.read(0, 80)
    assert np.array_equal(rows, want_u8)
    torch.cuda.synchronize()
    # an invalid image in the middle of a batch fails the call as a whole and leaves the embedder usable
    bad = list(imgs[:10])
    bad[5] = np.zeros((0, 7, 3), dtype=np.uint8)
    with pytest.raises(capi.PixelboxError):
        emb.embed_images(bad)
    got_u8, _ = emb.embed_images(imgs[:10])
    assert np.array_equal(got_u8, want_u8[:10])


@pytest.mark.gpu
def test_decoders_writing_into_the_staging_slots_give_the_batch_calls_bits():
    """pb_embed_stage_* (VERDICT r4 item 6): decode workers write their pixels straight into the embedder's pinned block -- no packing
    pass -- and the batch is closed and committed by one thread.  Eight writer threads, images of mixed sizes (one needing no
    resampling), three batches through both slots with the writers of batch i + 1 running while batch i is committed, a batch that
    fills up (PB_STAGE_FULL) and an empty close: every image's hash equals pb_embed_batch_images' for the same pixels."""
    import threading

    from pixelbox_amd import capi, synth
    from pixelbox_amd import weights as W

    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 64)
    emb = capi.Embedder(blob, max_batch=16)
    rng = np.random.default_rng(5)
    sizes = [(128, 128), (200, 150), (97, 311), (256, 256), (640, 480), (130, 128), (33, 500), (300, 300)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for i in range(40) for (w, h) in [sizes[i % len(sizes)]]]
    ref = emb.embed_images(imgs, want_f32=False)[0]
    assert emb.stage_close()[0] == 0  # nothing staged: an empty close is fine
    lock = threading.Lock()
    cond = threading.Condition(lock)
    next_img = [0]
    placed = {}      # (generation, position) -> image index
    full = [False]
    errors = []

    def writer():
        try:
            while True:
                with lock:
                    i = next_img[0]
                    if i >= len(imgs):
                        return
                    next_img[0] += 1
                h, w = imgs[i].shape[:2]
                while True:
                    got = emb.stage_acquire(w, h)
                    if got is not None:
                        break
                    with cond:  # the batch is full: the committing thread closes it
                        full[0] = True
                        cond.notify_all()
                        cond.wait(0.01)
                view, ticket = got
                view[...] = imgs[i]
                with lock:
                    placed[(ticket >> 32, ticket & 0xFFFF)] = i
                emb.stage_release(ticket)
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    threads = [threading.Thread(target=writer) for _ in range(8)]
    for t in threads:
        t.start()
    seen = 0
    batches = 0
    while seen < len(imgs):
        with cond:
            cond.wait_for(lambda: full[0] or next_img[0] >= len(imgs), timeout=0.05)
            full[0] = False
        n, gen, ws, hs = emb.stage_close()
        if n == 0:
            assert not errors, errors
            continue
        hashes, d_ptr = emb.stage_commit(n)
        assert d_ptr != 0
        with cond:
            cond.notify_all()
        for pos in range(n):
            i = placed[(gen, pos)]
            assert (ws[pos], hs[pos]) == (imgs[i].shape[1], imgs[i].shape[0])
            assert np.array_equal(hashes[pos], ref[i]), (gen, pos, i)
        seen += n
        batches += 1
    for t in threads:
        t.join()
    assert not errors, errors
    assert batches >= 3 and seen == len(imgs)


@pytest.mark.gpu
def test_fused_and_two_kernel_resize_give_the_same_bytes(monkeypatch):
    """k_resize_fused (both passes of an output row in one workgroup, the vertical sums in LDS) against k_resize_v + k_resize_h
    (PB_NO_RESIZE_FUSION=1), mixed sizes incl. no-resample, a tall source (> 64 vertical taps) and one wider than the fused form's LDS."""
    from pixelbox_amd import capi, synth
    from pixelbox_amd import weights as W

    blob = W.synthetic_blob(synth.SEED_WEIGHTS, 128, 128, 32)
    rng = np.random.default_rng(9)
    sizes = [(128, 128), (256, 256), (640, 480), (200, 3000), (131, 129), (9000, 140), (300, 5000)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for (w, h) in sizes]
    a = capi.Embedder(blob, max_batch=8)
    pre_a = [a.resize_to_fill(im) for im in imgs]
    u8_a, _ = a.embed_images(imgs, want_f32=False)
    monkeypatch.setenv("PB_NO_RESIZE_FUSION", "1")
    b = capi.Embedder(blob, max_batch=8)
    pre_b = [b.resize_to_fill(im) for im in imgs]
    u8_b, _ = b.embed_images(imgs, want_f32=False)
    for x, y, im in zip(pre_a, pre_b, imgs):
        assert np.array_equal(x, y), im.shape
        assert np.array_equal(x, oracle.resize_to_fill(im, 128, 128)), im.shape
    assert np.array_equal(u8_a, u8_b)
