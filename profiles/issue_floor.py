#!/usr/bin/env python3
"""Issue-floor table of one batch-512 embed forward: per launch, what the kernel's own instruction stream
needs on the pipes it uses, against the time it took.

    python3 profiles/issue_floor.py layers.txt pmc1.txt pmc2.txt [pmc3.txt]

layers.txt = profiles/embed_layers.py (kernel trace: us per launch); pmc*.txt = profiles/pmc_last_forward.py of the counter passes
of profiles/embed_pmc_pass.sh (per-dispatch SQ counters of the same forward, one line per launch, same order).

Model (MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, one vector / matrix instruction stream per SIMD, 2.4 GHz peak clock):
  matrix pipe   SQ_VALU_MFMA_BUSY_CYCLES, cycles summed over the SIMDs (= 32 per v_mfma_f32_16x16x4_f32, 16 per
                v_mfma_f32_16x16x32_bf16: checked against the instruction counts of pass 3);
  vector issue  SQ_ACTIVE_INST_VALU x 4 (the counter is in quad-cycles; it includes the matrix instructions' issue slots and the
                8-cycle transcendentals at their real length -- measured, not priced from an ISA dump);
  LDS           SQ_LDS_IDX_ACTIVE (LDS-array cycles incl. bank conflicts, per CU: 256 of them, not 1024).
  HBM           the launch's algorithmic activation bytes (embed_layers.py's MB column: the maps it reads and writes, f32) / 8 TB/s.
  floor_max = max(matrix, vector, 4 x LDS, HBM) / (1024 SIMDs x 2.4 GHz): nothing overlaps worse than perfectly;
  floor_sum = (matrix + vector - the matrix instructions' own issue slots) / (1024 x 2.4 GHz): a wave's vector instructions and its
              OWN matrix instructions never overlap unless another wave of the SIMD fills the gap.
  of_max / of_sum = floor / measured: 1.0 = the kernel runs at that floor.  Also printed: the share of wave-cycles spent waiting
  (SQ_WAIT_ANY: s_waitcnt / barrier) and stalled at issue (SQ_WAIT_INST_ANY), waves resident per SIMD (SQ_WAVE_CYCLES x 4 / (1024 x clk x t))."""
import re
import sys

SIMDS, CUS, CLK = 1024, 256, 2.4e9


def parse_pmc(path):
    out = []
    for ln in open(path):
        m = re.match(r"(\S.*?)\s+grid=\s*(\d+)\s+(.*)", ln)
        if not m:
            continue
        d = {"name": m.group(1).strip()}
        for kv in m.group(3).split():
            k, v = kv.split("=")
            d[k] = float(v)
        out.append(d)
    return out


def parse_layers(path):
    out = []
    for ln in open(path):
        m = re.match(r"(k_\S.*?)\s+(stem|tail|b\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(-?[\d.]+)(?:\s+([\d.]+))?", ln)
        if m:
            out.append({"name": m.group(1).strip(), "block": m.group(2), "us": float(m.group(3)), "gflop": float(m.group(4)),
                        "mb": float(m.group(8)) if m.group(8) else 0.0})
    return out


def main():
    layers = parse_layers(sys.argv[1])
    passes = [parse_pmc(p) for p in sys.argv[2:]]
    n = len(layers)
    for i, p in enumerate(passes):
        if len(p) != n:
            print(f"# pass {i + 1}: {len(p)} dispatches against {n} launches in the trace -- matching by position from the start")
    print(f"{'kernel':34s} {'blk':>4s} {'us':>7s} {'mfma us':>8s} {'valu us':>8s} {'lds us':>7s} {'hbm us':>7s} {'max us':>7s} {'of_max':>6s} {'sum us':>7s} {'of_sum':>6s} "
          f"{'wait':>5s} {'stall':>5s} {'w/SIMD':>6s} {'trans%':>6s}")
    tot = {"us": 0.0, "max": 0.0, "sum": 0.0}
    fam = {}
    for i, L in enumerate(layers):
        c = {}
        for p in passes:
            if i < len(p):
                c.update({k: v for k, v in p[i].items() if k != "name"})
        t = L["us"] * 1e-6
        mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        valu = c.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0
        lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0) * 4.0  # per CU -> per-SIMD-equivalent units (x 4) so one denominator serves
        n_f32 = c.get("SQ_INSTS_VALU_MFMA_F32", 0.0)
        n_bf = c.get("SQ_INSTS_VALU_MFMA_BF16", 0.0)
        own_issue = (n_f32 + n_bf) * 4.0 if (n_f32 + n_bf) else 0.0
        den = SIMDS * CLK
        f_m, f_v, f_l = mfma / den * 1e6, valu / den * 1e6, lds / den * 1e6
        f_h = L["mb"] * 1e6 / 8.0e12 * 1e6
        f_max = max(f_m, f_v, f_l, f_h)
        f_sum = max((mfma + valu - own_issue) / den * 1e6, f_h)
        wc = c.get("SQ_WAVE_CYCLES", 0.0) * 4.0
        wait = c.get("SQ_WAIT_ANY", 0.0) * 4.0 / wc if wc else 0.0
        stall = c.get("SQ_WAIT_INST_ANY", 0.0) * 4.0 / wc if wc else 0.0
        occ = wc / (den * t) if t else 0.0
        trans = c.get("SQ_INSTS_VALU_TRANS_F32", 0.0) / c["SQ_INSTS_VALU"] * 100 if c.get("SQ_INSTS_VALU") else 0.0
        print(f"{L['name'][:34]:34s} {L['block']:>4s} {L['us']:7.1f} {f_m:8.1f} {f_v:8.1f} {f_l:7.1f} {f_h:7.1f} {f_max:7.1f} {f_max / L['us']:6.2f} {f_sum:7.1f} {f_sum / L['us']:6.2f} "
              f"{wait:5.2f} {stall:5.2f} {occ:6.2f} {trans:6.1f}")
        tot["us"] += L["us"]; tot["max"] += f_max; tot["sum"] += f_sum
        f = fam.setdefault(L["name"].split("<")[0], {"us": 0.0, "max": 0.0, "sum": 0.0, "n": 0})
        f["us"] += L["us"]; f["max"] += f_max; f["sum"] += f_sum; f["n"] += 1
    print(f"{'total':34s} {'':>4s} {tot['us']:7.1f} {'':8s} {'':8s} {'':7s} {'':7s} {tot['max']:7.1f} {tot['max'] / tot['us']:6.2f} {tot['sum']:7.1f} {tot['sum'] / tot['us']:6.2f}")
    print("\nper family: launches, us, floor_max us (of), floor_sum us (of)")
    for k, f in sorted(fam.items(), key=lambda kv: -kv[1]["us"]):
        print(f"  {k:18s} {f['n']:2d} {f['us']:7.1f}  {f['max']:7.1f} ({f['max'] / f['us']:.2f})  {f['sum']:7.1f} ({f['sum'] / f['us']:.2f})")


if __name__ == "__main__":
    main()
