"""Structural guard for the DPP read hazard of the leaf kernels (leaf_f64.hip: fmac_rowbcast / mov_rowbcast).

gfx90a+ needs 2 wait states between a VALU instruction that writes a VGPR and a DPP instruction that reads it through
its DPP operand (src0).  The `v_fmac_f64_dpp` / `v_mov_b64_dpp` of the leaf are inline asm, so the compiler's hazard
recogniser does not see them: the code relies on how the source is written (the broadcast operand is produced well
before the elimination instructions) and on `s_nop 1` inside mov_rowbcast.  This test disassembles the gfx950 code
object inside csrc/leaf_f64.o and checks every DPP instruction of the library build, so a toolchain update or an edit
that schedules a producer right in front of its DPP consumer fails here instead of corrupting factors silently
(only the numerical GPU tests would notice otherwise).  The factor it guards: gpmcmc.py:313 (pt.slinalg.cholesky)."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

LLVM = "/opt/rocm/lib/llvm/bin"
REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def _regs(tok):
    m = REG.search(tok)
    if not m:
        return set()
    if m.group(1) is not None:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(m.group(3))}


def _parse(text):
    """[(mnemonic, [operand tokens])] of the instruction lines of an llvm-objdump -d listing."""
    out = []
    for line in text.splitlines():
        line = line.split("//")[0].strip()
        if not line or line.endswith(":") or line.startswith(("Disassembly", "/", ".")) or "file format" in line:
            continue
        parts = line.split(None, 1)
        mnem = parts[0]
        if not re.match(r"^[a-z][a-z0-9_]*$", mnem):
            continue
        ops = [t.strip() for t in parts[1].split(",")] if len(parts) > 1 else []
        out.append((mnem, ops))
    return out


def dpp_hazards(text, need=2):
    """DPP instructions whose DPP operand (src0) is written by a VALU instruction fewer than `need` wait states earlier.
    Every instruction in between is one wait state, `s_nop N` is N + 1."""
    ins = _parse(text)
    bad = []
    for i, (mnem, ops) in enumerate(ins):
        if not mnem.endswith("_dpp") or len(ops) < 2:
            continue
        src0 = _regs(ops[1].split()[0])
        waited, j = 0, i - 1
        while j >= 0 and waited < need:
            pm, pops = ins[j]
            if pm == "s_nop":
                waited += int(pops[0], 0) + 1 if pops else 1
            else:
                if pm.startswith("v_") and pops and (_regs(pops[0].split()[0]) & src0):
                    bad.append((i, mnem, ops[1], j, pm, pops[0], waited))
                    break
                waited += 1
            j -= 1
    return bad


def test_checker_flags_a_broken_sequence_and_accepts_a_padded_one():
    broken = """
	v_mul_f64 v[22:23], v[36:37], -v[10:11]
	v_fmac_f64_dpp v[24:25], v[22:23], v[34:35] row_newbcast:1 row_mask:0xf bank_mask:0xf
"""
    one_between = """
	v_mul_f64 v[22:23], v[36:37], -v[10:11]
	v_add_f64 v[2:3], v[4:5], v[6:7]
	v_mov_b64_dpp v[24:25], v[22:23] row_newbcast:1 row_mask:0xf bank_mask:0xf
"""
    padded = """
	v_mul_f64 v[22:23], v[36:37], -v[10:11]
	s_nop 1
	v_fmac_f64_dpp v[24:25], v[22:23], v[34:35] row_newbcast:1 row_mask:0xf bank_mask:0xf
	v_mul_f64 v[34:35], v[36:37], -v[10:11]
	v_fmac_f64_dpp v[26:27], v[22:23], v[34:35] row_newbcast:2 row_mask:0xf bank_mask:0xf
"""
    assert len(dpp_hazards(broken)) == 1
    assert len(dpp_hazards(one_between)) == 1
    assert dpp_hazards(padded) == []  # the non-DPP operand (src1) may come straight from the previous instruction


def _device_listing(obj, tmp_path):
    local = str(tmp_path / os.path.basename(obj))
    shutil.copy(obj, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True)
    dev = [f for f in os.listdir(tmp_path) if "amdgcn" in f and f.startswith(os.path.basename(obj))]
    if not dev:
        return ""  # host-only translation unit (api_blocks.hip defines no kernels)
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", str(tmp_path / dev[0])], check=True,
                          capture_output=True, text=True).stdout


def test_no_dpp_instruction_of_the_library_reads_a_freshly_written_vgpr(tmp_path):
    csrc = os.path.join(ROOT, "andvaranaut_amd", "csrc")
    objs = sorted(f for f in os.listdir(csrc) if f.endswith(".o"))
    if "leaf_f64.o" not in objs or not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("csrc/*.o or llvm-objdump missing (run __graft_entry__.build() first)")
    for name in objs:
        text = _device_listing(os.path.join(csrc, name), tmp_path)
        if name == "leaf_f64.o":
            assert text, "no gfx950 code object inside leaf_f64.o"
            ndpp = sum(1 for m, _ in _parse(text) if m in ("v_fmac_f64_dpp", "v_mov_b64_dpp"))
            assert ndpp >= 100, f"expected the leaf's row-broadcast eliminations in the disassembly, found {ndpp}"
        bad = dpp_hazards(text)
        assert not bad, (f"{name}: {len(bad)} DPP reads within 2 wait states of the VALU write of their operand, "
                         f"first: {bad[:3]}")
