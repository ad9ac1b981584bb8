"""Hand-written assembly inside the kernels (csrc/lz_group.hip.h, csrc/zstd_chain.h, the walks in the three compressors) is opaque to
the compiler's hazard recogniser: the wait states it needs are counted by hand (`s_nop`).  This test reads the gfx950 listing the
compiler produces for every kernel file and checks, over ALL of it, the two rules that hand-written code in this tree relies on
(CDNA3 ISA guide, "manually inserted wait states"; ADVICE of round 5 asked for a disassembly check):
  * a v_readlane / v_writelane whose lane select is a scalar register needs 4 wait states behind a vector instruction that wrote
    that register (v_readlane, v_readfirstlane, v_cmp with a scalar destination, carry-outs);
  * a DPP instruction needs 2 wait states behind a vector instruction that wrote its source register.
It also pins what round 6 did about the group executor's loads: every inline-asm statement that issues a global load ends with its
own `s_waitcnt vmcnt(0)` (the outputs of such a statement are defined only behind the wait).  CPU only (hipcc cross-compiles)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "aocl-compression_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FILES = ["lz4_kernels.hip", "snappy_kernels.hip", "zstd_kernels.hip"]


@pytest.fixture(scope="module", params=FILES)
def listing(request, tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / (request.param + ".s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                        "-o", str(out), os.path.join(CSRC, request.param)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return request.param, open(out).read().split("\n")


INSTR = re.compile(r"^\t([a-z_0-9]+)\s*(.*?)\s*(?:;.*)?$")


def instructions(lines):
    """(mnemonic, operand text, line number) of every instruction line, with the inline-asm markers kept as pseudo entries"""
    out = []
    for n, l in enumerate(lines):
        if l.startswith("\t;;#ASMSTART"):
            out.append(("#asmstart", "", n)); continue
        if l.startswith("\t;;#ASMEND"):
            out.append(("#asmend", "", n)); continue
        m = INSTR.match(l)
        if not m or m.group(1).startswith(".") or l.startswith("\t."):
            continue
        out.append((m.group(1), m.group(2), n))
    return out


def sgprs(tok):
    """scalar registers an operand token names: s5 -> {5}, s[4:5] -> {4, 5}, vcc -> {'vcc'}"""
    tok = tok.strip()
    m = re.fullmatch(r"s(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    if tok in ("vcc", "vcc_lo", "vcc_hi"):
        return {"vcc"}
    return set()


def vgprs(tok):
    tok = tok.strip()
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def wait_states(mn, ops):
    if mn == "s_nop":
        return int(ops.split()[0], 0) + 1
    return 0 if mn.startswith("#") else 1


def lane_select_hazards(lines):
    ins = instructions(lines)
    bad = []
    for i, (mn, ops, n) in enumerate(ins):
        if mn not in ("v_readlane_b32", "v_writelane_b32"):
            continue
        parts = [p.strip() for p in ops.split(",")]
        sel = sgprs(parts[2]) if len(parts) >= 3 else set()
        if not sel:
            continue                                   # constant lane or m0
        ws, j = 0, i - 1
        while j >= 0 and ws < 4:
            pm, po, pn = ins[j]
            if pm.startswith("v_"):
                pp = [p.strip() for p in po.split(",")]
                written = sgprs(pp[0]) if pp else set()
                if pm.endswith("_co_u32_e32") or pm.endswith("_co_u32_e64") or pm.startswith("v_cmp") and pm.endswith("_e32"):
                    written |= {"vcc"} if pm.endswith("_e32") else (sgprs(pp[1]) if len(pp) > 1 else set())
                if written & sel:
                    bad.append((n + 1, lines[n].strip(), pn + 1, lines[pn].strip(), ws))
                    break
            ws += wait_states(pm, po)
            j -= 1
    return bad


def dpp_hazards(lines):
    ins = instructions(lines)
    bad = []
    for i, (mn, ops, n) in enumerate(ins):
        if not (mn.endswith("_dpp") or " row_" in ops or "quad_perm" in ops or "wave_sh" in ops or "row_bcast" in ops):
            continue
        parts = [p.strip() for p in re.split(r",|\s+(?=row_|quad_perm|wave_|bank_mask|row_mask|bound_ctrl)", ops) if p.strip()]
        src = vgprs(parts[1]) if len(parts) > 1 else set()   # the DPP-controlled operand is src0
        ws, j = 0, i - 1
        while j >= 0 and ws < 2:
            pm, po, pn = ins[j]
            if pm.startswith("v_") or pm.startswith("ds_read") or pm.startswith("global_load"):
                pp = [p.strip() for p in po.split(",")]
                if pm.startswith("v_") and pp and (vgprs(pp[0]) & src):
                    bad.append((n + 1, lines[n].strip(), pn + 1, lines[pn].strip(), ws))
                    break
            ws += wait_states(pm, po)
            j -= 1
    return bad


def unwaited_asm_loads(lines):
    ins = instructions(lines)
    bad, inside, start, loads, waited = [], False, 0, 0, True
    for mn, ops, n in ins:
        if mn == "#asmstart":
            inside, start, loads, waited = True, n, 0, True
        elif mn == "#asmend":
            if inside and loads and not waited:
                bad.append((start + 1, loads))
            inside = False
        elif inside:
            if mn.startswith("global_load") or mn.startswith("flat_load") or mn.startswith("buffer_load"):
                loads += 1; waited = False
            elif mn == "s_waitcnt" and "vmcnt(0)" in ops:
                waited = True
    return bad


def test_scalar_lane_select_has_its_four_wait_states(listing):
    name, lines = listing
    bad = lane_select_hazards(lines)
    assert not bad, f"{name}: lane select read too early behind the vector instruction that wrote it: {bad[:5]}"


def test_dpp_source_has_its_two_wait_states(listing):
    name, lines = listing
    bad = dpp_hazards(lines)
    assert not bad, f"{name}: DPP source read too early behind the vector instruction that wrote it: {bad[:5]}"


def test_inline_asm_that_loads_waits_for_its_loads(listing):
    name, lines = listing
    bad = unwaited_asm_loads(lines)
    assert not bad, f"{name}: inline-asm statements whose loads are not waited for inside the statement (line, loads): {bad[:5]}"


def test_the_checkers_see_what_they_are_for():
    """crafted listings: each rule violated once, and the same code with the wait states in place"""
    t = "\t"
    walk_bad = [t + "v_readlane_b32 s3, v4, s1", t + "s_cmp_lt_u32 s3, 64", t + "s_bitset1_b64 s[6:7], s1", t + "v_readlane_b32 s1, v4, s3"]
    walk_ok = walk_bad[:3] + [t + "s_nop 1", t + "v_readlane_b32 s1, v4, s3"]
    assert len(lane_select_hazards(walk_bad)) == 1 and not lane_select_hazards(walk_ok)
    cmp_bad = [t + "v_cmp_eq_u32_e64 s[8:9], v1, v2", t + "s_nop 1", t + "v_writelane_b32 v5, s2, s8"]
    assert len(lane_select_hazards(cmp_bad)) == 1
    dpp_bad = [t + "v_and_b32_e32 v6, 0xf1f, v5", t + "v_add_u32_dpp v9, v6, v6 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"]
    dpp_ok = [dpp_bad[0], t + "s_nop 1", dpp_bad[1]]
    dpp_ok2 = [dpp_bad[0], t + "v_bfe_u32 v7, v5, 8, 4", t + "v_bfe_u32 v8, v5, 12, 9", dpp_bad[1]]
    assert len(dpp_hazards(dpp_bad)) == 1 and not dpp_hazards(dpp_ok) and not dpp_hazards(dpp_ok2)
    asm_bad = [t + ";;#ASMSTART", t + "global_load_dwordx4 v[0:3], v4, s[2:3]", t + ";;#ASMEND", t + ";;#ASMSTART", t + "s_waitcnt vmcnt(0)", t + ";;#ASMEND"]
    asm_ok = asm_bad[:2] + [t + "s_waitcnt vmcnt(0)", t + ";;#ASMEND"]
    assert len(unwaited_asm_loads(asm_bad)) == 1 and not unwaited_asm_loads(asm_ok)
