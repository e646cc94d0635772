"""phash (src/image_hashes/phash.rs:3-22): the CPU restatement against the reference's known answer and an independent
numpy restatement (CPU tests), and the HIP path against the restatement bit for bit (GPU tests).

What is pinned: phash.rs:36-41 -- a flat white (square) image hashes to 32 zero bytes.  What is not: the image crate's
Gaussian resampling and luma arithmetic (crate absent from the reference tree; phash.rs:43-78 need test images that are
absent too) -- restated from the crate's published algorithm in oracle/pb_oracle_phash.c.
"""
import numpy as np
import pytest

from oracle import capi as oracle

f32 = np.float32


def _np_phash(img):
    """Independent numpy restatement (vectorised where the order of f32 additions allows)."""
    h, w = img.shape[:2]
    ratio = min(16.0 / w, 16.0 / h)
    w2, h2 = max(int(np.floor(w * ratio + 0.5)), 1), max(int(np.floor(h * ratio + 0.5)), 1)

    def weights(o, in_size, out_size):
        r = f32(in_size) / f32(out_size)
        sr = r if r >= 1 else f32(1.0)
        sup = f32(3.0) * sr
        inp = (f32(o) + f32(0.5)) * r
        left = min(max(int(np.floor(inp - sup)), 0), in_size - 1)
        right = min(max(int(np.ceil(inp + sup)), left + 1), in_size)
        inp = inp - f32(0.5)
        xs = (np.arange(left, right, dtype=np.float32) - inp) / sr
        norm = f32(1.0) / (np.sqrt(f32(2.0) * f32(np.pi)) * f32(0.5))
        ws = (norm * np.exp(-(xs * xs) / f32(0.5))).astype(np.float32)
        s = f32(0.0)
        for v in ws:
            s = f32(s + v)
        return left, (ws / s).astype(np.float32)

    if (w2, h2) == (w, h):
        small = img.copy()
    else:
        tmp = np.zeros((h2, w, 3), dtype=np.float32)
        for oy in range(h2):
            left, ws = weights(oy, h, h2)
            acc = np.zeros((w, 3), dtype=np.float32)
            for i, wt in enumerate(ws):
                acc = (acc + img[left + i].astype(np.float32) * wt).astype(np.float32)
            tmp[oy] = acc
        small = np.zeros((h2, w2, 3), dtype=np.uint8)
        for ox in range(w2):
            left, ws = weights(ox, w, w2)
            acc = np.zeros((h2, 3), dtype=np.float32)
            for i, wt in enumerate(ws):
                acc = (acc + tmp[:, left + i] * wt).astype(np.float32)
            t = np.clip(acc, 0, 255)
            small[:, ox] = np.where(t - np.floor(t) >= 0.5, np.floor(t) + 1, np.floor(t)).astype(np.uint8)
    flat = small.reshape(-1, 3).astype(np.uint32)
    grey = ((2126 * flat[:, 0] + 7152 * flat[:, 1] + 722 * flat[:, 2]) // 10000).astype(np.uint8)
    mean = (int(grey.sum()) // 256) & 0xFF
    nb = len(grey) // 8
    bits = (grey[: nb * 8] > mean).reshape(nb, 8)
    return (bits * (1 << np.arange(8))).sum(axis=1).astype(np.uint8), small


def _images():
    rng = np.random.default_rng(21)
    out = []
    for (h, w) in [(16, 16), (300, 300), (128, 128), (100, 160), (480, 31), (31, 480), (9, 16), (8, 8), (5, 3), (1, 1), (17, 16), (1000, 1500)]:
        kind = rng.integers(0, 3)
        if kind == 0:
            img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        elif kind == 1:  # smooth gradients + a bright blob: what photographs look like to a 16x16 average hash
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) * 127 // max(w + h - 2, 1))], axis=-1).astype(np.uint8)
        else:
            img = np.full((h, w, 3), rng.integers(0, 256), dtype=np.uint8)
            img[h // 3: h // 2 + 1, w // 4: w // 2 + 1] = rng.integers(0, 256, size=3, dtype=np.uint8)
        out.append(img)
    return out


def test_reference_kat_flat_white_phash_rs_36_41():
    # test_phash_flat_white: 32 zero bytes (a square image; any flat square image gives the same)
    for side in (16, 64, 300):
        assert oracle.phash(np.full((side, side, 3), 255, np.uint8)).tolist() == [0] * 32
    assert oracle.phash(np.full((40, 40, 3), 7, np.uint8)).tolist() == [0] * 32
    # the reference divides the grey sum by the CONSTANT 256 (phash.rs:10): a 4:3 flat white image is resized to 16 x 12,
    # its "mean" is 191, every pixel exceeds it and the 24 hash bytes are all ones -- restated as it is
    assert oracle.phash(np.full((300, 400, 3), 255, np.uint8)).tolist() == [255] * 24


def test_restatement_matches_the_independent_numpy_one():
    for img in _images():
        want_hash, want_small = _np_phash(img)
        got_hash, got_small = oracle.phash(img, want_small=True)
        assert got_small.shape == want_small.shape
        # the two restatements differ only through expf (libm vs numpy's SIMD exp): allow a last-bit weight difference to
        # move a resized byte by one, and the hash to differ only where such a byte sits at the threshold
        assert np.abs(got_small.astype(int) - want_small.astype(int)).max() <= 1
        if np.array_equal(got_small, want_small):
            assert np.array_equal(got_hash, want_hash)


def test_identical_images_hash_identically_and_bits_are_lsb_first():
    # phash.rs:43-50: hamming_distance(h, h) == 0; and the bit order of phash.rs:14-18
    img = np.zeros((16, 16, 3), np.uint8)
    img[0, 0] = 255  # grey[0] is the only pixel above the mean (0)
    h = oracle.phash(img)
    assert h[0] == 1 and not h[1:].any()
    img[0, 7] = 255
    assert oracle.phash(img)[0] == 0x81
    assert oracle.hamming_distance(h, h) == 0.0


@pytest.mark.gpu
def test_gpu_phash_matches_the_restatement_bit_for_bit():
    from pixelbox_amd import capi

    ph = capi.PHasher()
    for img in _images():
        want_hash, want_small = oracle.phash(img, want_small=True)
        assert np.array_equal(ph.small_image(img), want_small), img.shape
        assert np.array_equal(ph.phash(img), want_hash), img.shape
    assert ph.phash(np.full((300, 300, 3), 255, np.uint8)).tolist() == [0] * 32  # phash.rs:36-41
    with pytest.raises(capi.PixelboxError):
        ph.phash(np.zeros((0, 5, 3), np.uint8))  # empty image: PB_ERR_INVALID, no crash


@pytest.mark.gpu
def test_gpu_phash_of_a_batch_is_the_phash_of_each_image():
    # pb_phash_batch_images: the crawler's batch in one call (two launches per sub-batch over a descriptor array) -- every
    # hash, and its length (non-square images give fewer than 32 bytes, as in the reference), equals the restatement's;
    # 75 MB of sources: two staging sub-batches
    from pixelbox_amd import capi

    rng = np.random.default_rng(21)
    imgs = list(_images())
    for (h, w) in [(1080, 1920), (16, 16), (1, 1), (480, 640), (2000, 100), (17, 16)] * 10:
        base = rng.integers(0, 256, size=(h // 8 + 1, w // 8 + 1, 3), dtype=np.uint8)
        imgs.append(np.kron(base, np.ones((8, 8, 1), dtype=np.uint8))[:h, :w].copy())
    assert sum(im.size for im in imgs) > 64 << 20
    ph = capi.PHasher()
    got = ph.phash_batch(imgs)
    for im, g in zip(imgs, got):
        want = oracle.phash(im)
        assert np.array_equal(g, want), im.shape
    assert [len(x) for x in ph.phash_batch(imgs[:3])] == [len(oracle.phash(im)) for im in imgs[:3]]  # the blocks are reused
    with pytest.raises(capi.PixelboxError):
        ph.phash_batch([imgs[0], np.zeros((0, 5, 3), np.uint8)])
    assert np.array_equal(ph.phash(imgs[0]), oracle.phash(imgs[0]))  # the handle survives a failed batch


@pytest.mark.gpu
def test_gpu_phash_feeds_the_hamming_scan():
    # the `phashes` table (engine.rs:106-109) scanned with hamming_distance (engine.rs:594-604): a resized copy of an
    # image lands next to the original (phash.rs:52-55 asserts hamming < 0.0001 for a resized copy)
    from pixelbox_amd import capi

    rng = np.random.default_rng(8)
    ph = capi.PHasher()
    base = []
    for i in range(40):
        yy, xx = np.mgrid[0:96, 0:96]
        a, b, c = rng.integers(1, 9, size=3)
        img = np.stack([(np.sin(xx / a) * 100 + 128), (np.cos(yy / b) * 100 + 128), (np.sin((xx + yy) / c) * 100 + 128)], axis=-1)
        base.append(np.clip(img, 0, 255).astype(np.uint8))
    hashes = np.stack([ph.phash(im) for im in base])
    assert hashes.shape == (40, 32)
    ix = capi.Index(32, 64, metric=capi.PB_METRIC_HAMMING)
    ix.append(np.arange(1, 41, dtype=np.int64), hashes)
    big = np.repeat(np.repeat(base[13], 2, axis=0), 2, axis=1)  # the same picture at twice the size
    q = ph.phash(big)
    ids, dist = ix.search_one(q, 5, 1e3)
    assert ids[0] == 14
    want_ids, want_d = oracle.scan_topk_metric(capi.PB_METRIC_HAMMING, q, hashes, np.arange(1, 41, dtype=np.int64), 5, 1e3)
    assert np.array_equal(ids, want_ids) and np.array_equal(dist.view(np.uint32), want_d.view(np.uint32))
