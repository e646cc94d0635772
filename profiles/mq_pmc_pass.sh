#!/bin/bash
# Per-dispatch SQ counters of the burst collect kernel (k_scan_multi_wg) over profiles/mq_probe.py: three --pmc passes (the sets of
# embed_pmc_pass.sh) and an issue account per 16-row x 64-query wave-tile built from them.  usage (GPU box, repo root): bash profiles/mq_pmc_pass.sh [outdir]
O=$PWD/${1:-gpurun_out/mq_pmc}; R=$PWD; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p1 -o p -- python3 $R/profiles/mq_probe.py > /dev/null 2> $O/p1.err
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $O/p2 -o p -- python3 $R/profiles/mq_probe.py > /dev/null 2> $O/p2.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -o p -- python3 $R/profiles/mq_probe.py > /dev/null 2> $O/p3.err
python3 - $O <<'PY' > $O/mq_issue_account.txt 2>&1
import csv, glob, sys, collections
O = sys.argv[1]
tot = collections.OrderedDict()
for p in ("p1", "p2", "p3"):
    f = glob.glob(f"{O}/{p}/**/p_counter_collection.csv", recursive=True)
    if not f:
        print("no counters for pass", p); continue
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        if "k_scan_multi_wg" not in r["Kernel_Name"]:
            continue
        per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    if not per:
        print("no k_scan_multi_wg dispatch in pass", p); continue
    last = per[sorted(per)[-1]]  # the last 1 024-query launch of the probe
    tot.update(last)
for k, v in tot.items():
    print(f"{k:28s} {v:.6g}")
# per wave-tile: 10M rows / 16 rows x 16 groups of 64 queries = 10M wave-tiles per launch
WT = 10_000_000 / 16 * 16
print()
print("per 16-row x 64-query wave-tile (10 000 000 wave-tiles per launch):")
def per(k, scale=1.0):
    return tot.get(k, float('nan')) * scale / WT
print(f"  vector instructions issued (SQ_INSTS_VALU, MFMAs included)     {per('SQ_INSTS_VALU'):7.1f}")
print(f"  of them matrix (SQ_INSTS_VALU_MFMA_I8 / SQ_INSTS_MFMA)          {per('SQ_INSTS_VALU_MFMA_I8'):7.1f} / {per('SQ_INSTS_MFMA'):7.1f}")
print(f"  scalar (SQ_INSTS_SALU)                                         {per('SQ_INSTS_SALU'):7.1f}")
print(f"  LDS (SQ_INSTS_LDS; loads / stores)                             {per('SQ_INSTS_LDS'):7.1f}  ({per('SQ_INSTS_LDS_LOAD'):.1f} / {per('SQ_INSTS_LDS_STORE'):.1f})")
print(f"  vector memory reads                                            {per('SQ_INSTS_VMEM_RD'):7.2f}")
print(f"  clocks the wave's vector unit was issuing (SQ_ACTIVE_INST_VALU x 4) {per('SQ_ACTIVE_INST_VALU', 4):7.1f}")
print(f"  matrix pipe busy clocks (SQ_VALU_MFMA_BUSY_CYCLES)             {per('SQ_VALU_MFMA_BUSY_CYCLES'):7.1f}")
print(f"  LDS array clocks (SQ_LDS_IDX_ACTIVE), of them bank conflicts   {per('SQ_LDS_IDX_ACTIVE'):7.1f}, {per('SQ_LDS_BANK_CONFLICT'):.1f}")
print(f"  wave clocks (SQ_WAVE_CYCLES x 4), waiting (SQ_WAIT_INST_ANY x 4) {per('SQ_WAVE_CYCLES', 4):7.1f}, {per('SQ_WAIT_INST_ANY', 4):.1f}")
print("(counter units as in profiles/issue_floor.py: SQ_*_CYCLES / ACTIVE counters tick once per 4 clocks per wave or SIMD; GRBM_GUI_ACTIVE = the launch's clocks)")
PY
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS[A-Z_0-9]*" | sort -u > $O/sq_insts_counters.txt
rm -rf $O/p1 $O/p2 $O/p3
cd $R; cat $O/mq_issue_account.txt
