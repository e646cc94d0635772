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
