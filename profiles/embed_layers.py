#!/usr/bin/env python3
"""Per-kernel table of ONE embed forward from a rocprofv3 --kernel-trace CSV of profiles/embed_probe.py: the last forward of
the largest batch, kernel by kernel in launch order, with each kernel's time, its algorithmic FLOPs (2 x
multiply-adds of the convolutions / Linear layers it computes, squeeze-excite included), the rate and the fraction of the
dense f32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md).

    python3 profiles/embed_layers.py <kt_kernel_trace.csv> [batch = 512] [mfma_busy.json]

Kernels are attributed to MBConv blocks by walking the network (pixelbox_amd.weights.blocks()) alongside the trace: a block
ends at its project GEMM (a gated k_gemm1x1 / k_gemm_t / k_gemm_p3) or at its k_block_small."""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelbox_amd import weights as W

PEAK_TF = 157.3


def main(path, batch):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "k_stem" in r["Kernel_Name"]]
    big = max(int(rows[i]["Grid_Size_X"]) * int(rows[i]["Grid_Size_Y"]) for i in idx)
    s = [i for i in idx if int(rows[i]["Grid_Size_X"]) * int(rows[i]["Grid_Size_Y"]) == big][-1]
    blocks = W.blocks()
    h = w = 64  # after the stem (128 x 128 input)
    geo = []
    for b in blocks:
        ho, wo = (h + b.stride - 1) // b.stride, (w + b.stride - 1) // b.stride
        geo.append((b, h, w, ho, wo))
        h, w = ho, wo
    stem_fl = 2 * 64 * 64 * 27 * 32
    bi = 0
    tail_gemms = 0
    tot_us = tot_fl = tot_gap = 0.0
    by = {}
    print(f"{'kernel':38s} {'block':>5s} {'us':>8s} {'GFLOP':>8s} {'TFLOP/s':>8s} {'of peak':>7s} {'gap us':>7s} {'MB':>7s}")
    prev_end = None
    for r in rows[s : s + 90]:
        n = r["Kernel_Name"]
        if "pbe::" not in n:
            break
        short = n.split("(")[0].replace("void pbe::", "").replace("pbe::", "")
        fam = short.split("<")[0]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
        fl, label = 0.0, ""
        by_ = 0.0  # algorithmic activation bytes per image this launch reads + writes (f32 maps; weights not counted)
        if fam in ("k_stem", "k_stem_dw"):
            fl, label = stem_fl, "stem"
            by_ = 128 * 128 * 3 + 64 * 64 * 32 * 4
            if fam == "k_stem_dw":  # + the first block's depthwise
                b, hh, ww, ho, wo = geo[0]
                fl += 2 * ho * wo * b.kernel * b.kernel * b.expanded
                by_ = 128 * 128 * 3 + ho * wo * b.expanded * 4
        elif bi < len(geo):
            b, hh, ww, ho, wo = geo[bi]
            label = f"b{bi}"
            f_exp = 2 * hh * ww * b.cin * b.expanded if b.has_expand else 0
            f_dw = 2 * ho * wo * b.kernel * b.kernel * b.expanded
            f_se = 2 * 2 * b.expanded * b.squeeze
            f_proj = 2 * ho * wo * b.expanded * b.cout
            gated = "true" in short
            if fam == "k_gemm_p3":  # <NR, MR, GATE, NW, EPI, DIRECT, KT>: the project layers are the gated ones
                gated = short.split("<")[1].split(",")[2].strip() == "true"
            resid = b.stride == 1 and b.cin == b.cout
            if fam == "k_block_small":
                fl = f_exp + f_dw + f_se + f_proj
                by_ = 4 * (hh * ww * b.cin + ho * wo * b.cout)
                bi += 1
            elif fam in ("k_front_band", "k_front_roll", "k_mbconv_small"):
                fl = f_exp + f_dw
                by_ = 4 * (hh * ww * b.cin + ho * wo * b.expanded)
            elif fam in ("k_dwconv", "k_dwconv_roll", "k_dwconv_lds"):
                fl = f_dw
                by_ = 4 * (hh * ww * b.expanded + ho * wo * b.expanded)
            elif fam == "k_se":
                fl = f_se
            elif fam == "k_gemm_stream" or (fam in ("k_gemm1x1", "k_gemm_t", "k_gemm_p3", "k_gemm_thin") and gated):
                fl = f_proj
                by_ = 4 * (ho * wo * b.expanded + ho * wo * b.cout * (2 if resid else 1))
                bi += 1
            elif fam in ("k_gemm1x1", "k_gemm_t", "k_gemm_p3", "k_gemm_thin"):
                fl = f_exp  # the expand GEMM of an unfused front
                by_ = 4 * (hh * ww * b.cin + hh * ww * b.expanded)
        else:
            label = "tail"
            if fam in ("k_gemm1x1", "k_gemm_t", "k_gemm_p3"):
                fl = 2 * 16 * 320 * 1280 if tail_gemms == 0 else 2 * 1280 * 256  # head conv (+ pool), then the Linear
                by_ = 4 * (16 * 320 + 1280) if tail_gemms == 0 else 4 * (1280 + 256) + 256
                tail_gemms += 1
        fl *= batch
        tot_us += dur
        tot_fl += fl
        by[fam] = by.get(fam, 0.0) + dur
        tf = fl / dur / 1e6 if dur > 0 else 0.0
        gap = (int(r["Start_Timestamp"]) - prev_end) / 1000 if prev_end is not None else 0.0  # idle time since the previous kernel ended
        prev_end = int(r["End_Timestamp"])
        tot_gap += gap
        print(f"{short[:38]:38s} {label:>5s} {dur:8.1f} {fl / 1e9:8.2f} {tf:8.1f} {tf / PEAK_TF:7.3f} {gap:7.1f} {by_ * batch / 1e6:7.1f}")
        if "k_tanh_quant" in n or ("k_gemm_t" in n and short.rstrip(">").endswith(", 2")):
            break
        if fam == "k_gemm_p3" and short.split("<")[1].split(",")[4].strip() == "2":  # the Linear + tanh + quantiser epilogue
            break
    tf = tot_fl / tot_us / 1e6
    print(f"{'total':38s} {'':>5s} {tot_us:8.1f} {tot_fl / 1e9:8.2f} {tf:8.1f} {tf / PEAK_TF:7.3f}   (sum of kernel durations; gaps between them {tot_gap:.1f} us; batch {batch})")
    print("per kernel family, us:", {k: round(v, 1) for k, v in sorted(by.items(), key=lambda kv: -kv[1])})


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 512)
