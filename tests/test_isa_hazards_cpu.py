"""Static check of the hand-written DPP instructions in the built gfx950 code object.

The Cholesky panel of als_row_solve issues `v_fmac_f32_dpp` / `v_mul_f32_dpp ... row_newbcast:t` from inline asm
(collaborative-filtering_amd/csrc/row_solve.hip, panel_pivot / panel_trailing).  The compiler's hazard recogniser does not look
into asm, so the two wait states gfx9 requires between a VALU write of a VGPR and a DPP read of it are ordered
by hand (an `s_nop 1` tied to the operand).  This test disassembles libals_hip.so and verifies, for every such
instruction, that no VALU instruction wrote its DPP source within the two preceding wait states and that no
`v_cmpx` (VALU write of EXEC, five wait states before a DPP op) exists in the library at all.
No GPU needed: it reads the code object only.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "collaborative-filtering_amd", "csrc", "libals_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _disassemble(tmp_path):
    work = tmp_path / "obj"
    work.mkdir()
    shutil.copy(LIB, work / "lib.so")
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=work, check=True, stdout=subprocess.DEVNULL)
    out = []
    for f in sorted(os.listdir(work)):
        if "amdgcn" in f and "gfx950" in f and os.path.getsize(work / f) > 0:
            txt = subprocess.run([OBJDUMP, "-d", f], cwd=work, check=True, capture_output=True, text=True).stdout
            out.append(txt)
    return out


def _instructions(text):
    for line in text.splitlines():
        if not line.startswith("\t"):
            yield None                      # symbol header / blank: a boundary
            continue
        ins = line.split("//")[0].strip()
        if ins:
            yield ins


def _vgprs(operand):
    operand = operand.strip().lstrip("-|").rstrip("|")
    m = re.fullmatch(r"v(\d+)", operand)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def _check(text):
    """Number of hand-written DPP FMAs in one disassembly; asserts on a hazard."""
    ndpp = 0
    window = []                             # the preceding instructions of the same function
    for ins in _instructions(text):
        if ins is None:
            window = []
            continue
        assert not ins.startswith("v_cmpx"), "v_cmpx writes EXEC from the VALU: DPP ops need 5 wait states after it"
        if "row_newbcast" in ins:           # only the hand-written asm uses this DPP control
            ndpp += 1
            ops = ins.split(None, 1)[1].split(",")
            src = _vgprs(ops[1].split()[0])
            assert src, ins
            covered = 0
            for prev in reversed(window):
                if covered >= 2:
                    break
                if prev.startswith("v_"):
                    dst = _vgprs(prev.split(None, 1)[1].split(",")[0])
                    assert not (dst & src), f"DPP hazard: `{prev}` then `{ins}` within 2 wait states"
                m = re.match(r"s_nop (\d+)", prev)
                covered += int(m.group(1)) + 1 if m else 1
        window.append(ins)
        if len(window) > 8:
            window.pop(0)
    return ndpp


def test_checker_sees_a_hazard_and_accepts_the_ordered_form():
    dpp = "\tv_fmac_f32_dpp v5, v45, v46 row_newbcast:3 row_mask:0xf bank_mask:0xf// 0000: 0 0\n"
    ok = "f:\n\tv_mul_f32_e32 v45, v32, v46  // 0\n\ts_nop 1  // 0\n\tv_xor_b32_e32 v46, 0x80000000, v4 // 0\n" + dpp
    assert _check(ok) == 1
    far = "f:\n\tv_mul_f32_e32 v45, v32, v46  // 0\n\tv_mov_b32_e32 v1, v2 // 0\n\tv_mov_b32_e32 v3, v2 // 0\n" + dpp
    assert _check(far) == 1
    for bad in ("f:\n\tv_mul_f32_e32 v45, v32, v46  // 0\n" + dpp,
                "f:\n\tv_mul_f32_e32 v45, v32, v46  // 0\n\tv_mov_b32_e32 v1, v2 // 0\n" + dpp,
                "f:\n\tv_pk_mul_f32 v[44:45], v[2:3], v[4:5]  // 0\n\ts_nop 0 // 0\n" + dpp):
        with pytest.raises(AssertionError):
            _check(bad)


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm image not present")
def test_dpp_reads_are_two_wait_states_behind_their_valu_writes(tmp_path):
    if not os.path.exists(LIB):             # fresh checkout: the .so is git-ignored
        import __graft_entry__ as ge
        ge.build()
    texts = _disassemble(tmp_path)
    assert texts, "no gfx950 code object in libals_hip.so"
    ndpp = sum(_check(t) for t in texts)
    assert ndpp > 1000, f"expected the panel's DPP FMAs in the code object, found {ndpp}"
