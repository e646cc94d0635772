"""Build-time checks on the generated gfx950 code (no GPU needed: hipcc cross-compiles here).

`fmac_shift` (pb_embed_kernels.h) issues `v_fmac_f32_dpp` through inline asm.  CDNA needs two wait states between a
VALU write of a VGPR and a DPP read of it, and five after a VALU write of EXEC; hipcc inserts them for the DPP moves
it generates itself but cannot see inside asm, so the kernels are written to keep the producers far away -- and this
test reads the device assembly to make sure the scheduler did not move one next to a consumer.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _written_vgprs(instr: str):
    m = re.match(r"(\S+)\s+(v\[(\d+):(\d+)\]|v(\d+))\b", instr)
    if not m or not m.group(1).startswith("v_"):
        return set()
    if m.group(3):
        return set(range(int(m.group(3)), int(m.group(4)) + 1))
    return {int(m.group(5))}


@pytest.mark.timeout(600)
def test_inline_asm_dpp_reads_keep_their_wait_states(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / "pb_embed.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                           "--cuda-device-only", "-S", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "pixelbox_amd", "csrc", "pb_embed.hip"), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    ins = [l.strip() for l in open(out) if re.match(r"^\s+[a-z]", l)]
    n = 0
    for k, t in enumerate(ins):
        m = re.match(r"v_fmac_f32_dpp v\d+, v(\d+), v\d+", t)
        if not m:
            continue
        n += 1
        src = int(m.group(1))
        for back in (1, 2):
            assert src not in _written_vgprs(ins[k - back]), f"DPP source v{src} written {back} instruction(s) before: {ins[k - back]} -> {t}"
        for back in range(1, 6):
            assert not ins[k - back].startswith("v_cmpx"), f"VALU write of EXEC {back} instruction(s) before a DPP read: {t}"
    assert n > 0, "fmac_shift no longer compiles to v_fmac_f32_dpp"


@pytest.mark.timeout(600)
def test_exact_pass_keeps_its_lds_reads_one_piece_ahead(tmp_path):
    # k_scan_exact_co issues the LDS reads of a 16-byte piece (row bytes + 4 QN broadcast reads of query values) through
    # inline asm, one piece ahead of their use (pb_scan_kernels.h).  Read the ISA: the piece loop of every instance has no
    # scratch access, exactly two waits, and each wait is followed at once by the next piece's 1 + 4 QN reads -- left to
    # the scheduler the prefetch is sunk back next to its use and every piece pays the LDS latency.
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / "pb_scan.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                           "--cuda-device-only", "-S", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "pixelbox_amd", "csrc", "pb_scan.hip"), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    s = open(out).read()
    found = 0
    for m in re.finditer(r"^(_ZN3pbk15k_scan_exact_coILi(\d)ELi(\d)E\w*):[^\n]*\n(.*?)\.end_amdhsa_kernel", s, re.S | re.M):
        qn = int(m.group(2))
        body = m.group(4)
        assert "scratch_" not in body
        blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
        loop = max(blocks, key=lambda b: len(re.findall(r"v_fma_f32|v_fmac_f32|v_pk_fma_f32", b)))
        loop = loop[: loop.index("s_cbranch_scc")]  # up to the back edge
        ops = [l.strip().split()[0] for l in loop.split("\n") if re.search(r"s_waitcnt lgkmcnt|ds_read_b", l)]
        want = ["s_waitcnt"] + ["ds_read_b128"] * (1 + 4 * qn)
        assert ops == want * 2, (m.group(1), ops)
        # the multiplies and adds of the fold are scalar instructions (asm): no packed pairs, no register shuffling
        assert len(re.findall(r"v_pk_mul_f32|v_pk_add_f32", loop)) == 0 and len(re.findall(r"\bv_add_f32 ", loop)) == 32 * qn
        found += 1
    assert found == 4
    # the four-query form (k_scan_exact_co4): 1 + 16 reads behind each wait
    found = 0
    for m in re.finditer(r"^(_ZN3pbk16k_scan_exact_co4ILi(\d)E\w*):[^\n]*\n(.*?)\.end_amdhsa_kernel", s, re.S | re.M):
        body = m.group(3)
        assert "scratch_" not in body
        blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
        loop = max(blocks, key=lambda b: len(re.findall(r"\bv_add_f32 ", b)))
        loop = loop[: loop.index("s_cbranch_scc")]
        ops = [l.strip().split()[0] for l in loop.split("\n") if re.search(r"s_waitcnt lgkmcnt|ds_read_b", l)]
        assert ops == (["s_waitcnt"] + ["ds_read_b128"] * 17) * 2, (m.group(1), ops)
        assert len(re.findall(r"\bv_add_f32 ", loop)) == 128 and len(re.findall(r"v_pk_mul_f32|v_pk_add_f32", loop)) == 0
        found += 1
    assert found == 2
