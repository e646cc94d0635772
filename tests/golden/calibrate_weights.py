#!/usr/bin/env python3
"""One-off LSUV-style calibration of the synthetic weight generator (run HERE only).

A random-init 82-layer network without BatchNorm statistics either dies or explodes; a trained,
BN-folded network has O(1) activations everywhere.  This script walks the network once on a few
synthetic 128x128 images and picks, per weight tensor, a multiplier (rounded to 3 significant digits)
that gives the layer's pre-activation output a target standard deviation.  The resulting table is
pasted into pixelbox_amd/weights.py (_CALIB), so that synthetic_blob() stays a pure integer-PRNG +
f32-multiply function that reproduces bit-identically on any machine.

    python tests/golden/calibrate_weights.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import numpy_oracle as no  # noqa: E402
from pixelbox_amd import synth, weights as W  # noqa: E402


def round3(x: float) -> float:
    return float(f"{x:.3g}")


def main():
    seed, h, d = 0x5EED0005, 128, 256
    W._CALIB = None  # raw generator
    blob = W.synthetic_blob(seed, h, h, d)
    _, _, _, t = W.parse_blob(blob)
    T = {k: torch.from_numpy(v.copy()) for k, v in t.items()}
    n = 16
    imgs = synth.synthetic_images(synth.SEED_IMAGES, 0, n, h, h)
    x = torch.from_numpy(imgs.astype(np.float32)).div(255).permute(0, 3, 1, 2).contiguous()
    mult: dict[str, float] = {}

    def calib(name, pre, target):
        m = round3(target / float(pre.std()))
        mult[name] = m
        return m

    with torch.no_grad():
        def conv(name, inp, target, **kw):
            w = T[name + ".w"]
            w4 = w[:, None] if name.endswith("dw") else (w[:, :, None, None] if w.dim() == 2 else w)
            zero_b = torch.zeros_like(T[name + ".b"])
            pre = F.conv2d(inp, w4, zero_b, **kw)
            m = calib(name + ".w", pre, target)
            return F.conv2d(inp, w4 * m, T[name + ".b"], **kw)

        x = F.silu(conv("stem", x, 1.0, stride=2, padding=1))
        for i, b in enumerate(W.blocks()):
            p = f"b{i}."
            y = x
            if b.has_expand:
                y = F.silu(conv(p + "expand", y, 1.0))
            y = F.silu(conv(p + "dw", y, 1.0, stride=b.stride, padding=(b.kernel - 1) // 2, groups=b.expanded))
            s = y.mean(dim=(2, 3), keepdim=True)
            s = F.silu(conv(p + "se_reduce", s, 1.0))
            s = torch.sigmoid(conv(p + "se_expand", s, 1.0))
            y = y * s
            y = conv(p + "project", y, 0.8)
            x = x + y if b.residual else y
        x = F.silu(conv("head", x, 1.0))
        x = x.mean(dim=(2, 3))
        pre = F.linear(x, T["fc.w"], torch.zeros_like(T["fc.b"]))
        m = calib("fc.w", pre, 0.6)
        pre = F.linear(x, T["fc.w"] * m, T["fc.b"])
        print("pre-tanh std", float(pre.std()), "across images", float(pre.std(dim=0).mean()))
    names = [nme for nme, _, role in W.tensor_specs(d) if role not in ("bias", "se_bias")]
    print("_CALIB = [")
    line = "    "
    for nme in names:
        tok = f"{mult[nme]!r}, "
        if len(line) + len(tok) > 110:
            print(line.rstrip())
            line = "    "
        line += tok
    print(line.rstrip())
    print("]")


if __name__ == "__main__":
    main()
