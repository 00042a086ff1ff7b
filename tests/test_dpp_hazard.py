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



# ---------------------------------------------------------------- control-flow aware form (round 4)
def dpp_hazards_cfg(text, need=2):
    """The same rule along EVERY path into a DPP instruction: the linear walk above only sees the instructions that precede
    it in the listing, but a DPP instruction that is (or closely follows) a branch target is also preceded at run time by
    the last instructions of every block that branches to it.  Walks backwards from each DPP instruction through the
    control-flow graph (fall-through and branch predecessors) until `need` wait states have been seen on that path."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_uninit_check as chk

    bad = []
    for name, insts in chk.parse_disassembly(text).items():
        addr2idx = {i.addr: k for k, i in enumerate(insts)}
        preds = {}  # instruction index -> indices of instructions that can execute right before it
        for k, i in enumerate(insts):
            falls = not (i.mn in ("s_branch", "s_endpgm", "s_setpc_b64"))
            if falls and k + 1 < len(insts):
                preds.setdefault(k + 1, []).append(k)
            if (i.mn == "s_branch" or i.mn.startswith("s_cbranch")) and i.target in addr2idx:
                preds.setdefault(addr2idx[i.target], []).append(k)
        for k, i in enumerate(insts):
            if not i.mn.endswith("_dpp") or len(i.ops) < 2:
                continue
            src0 = _regs(i.ops[1])
            stack, seen = [(p, 0) for p in preds.get(k, [])], set()
            while stack:
                j, waited = stack.pop()
                if (j, waited) in seen or waited >= need:
                    continue
                seen.add((j, waited))
                pj = insts[j]
                if pj.mn == "s_nop":
                    w = waited + (int(pj.mods.strip() or "0", 0) + 1)
                else:
                    if pj.mn.startswith("v_") and pj.ops and (_regs(pj.ops[0]) & src0):
                        bad.append((name, hex(i.addr), i.text, hex(pj.addr), pj.text, waited))
                        continue
                    w = waited + 1
                for q in preds.get(j, []):
                    stack.append((q, w))
    return bad


def test_cfg_checker_sees_a_hazard_through_a_branch_target():
    # the producer sits at the end of a block that BRANCHES to the DPP instruction; in the listing an unrelated, padded
    # block precedes the consumer, so the linear checker is blind to it
    def listing(pad):
        lines = ["0000000000001000 <k>:",
                 "\tv_mul_f64 v[22:23], v[36:37], -v[10:11]                    // 000000001000: D2810016 40021524"]
        addr = 0x1008
        if pad:
            lines.append(f"\ts_nop 1                                                    // {addr:012X}: BF800001")
            addr += 4
        lines.append(f"\ts_branch 3                                                 // {addr:012X}: BF820003 <k+{addr + 16 - 0x1000:#x}>")
        addr += 4
        lines.append(f"\tv_add_f64 v[2:3], v[4:5], v[6:7]                           // {addr:012X}: D2800002 00020D04")
        addr += 8
        lines.append(f"\ts_nop 1                                                    // {addr:012X}: BF800001")
        addr += 4
        lines.append(f"\tv_fmac_f64_dpp v[24:25], v[22:23], v[34:35] row_newbcast:1 row_mask:0xf bank_mask:0xf // {addr:012X}: 0830444A FF1522FA")
        addr += 8
        lines.append(f"\ts_endpgm                                                   // {addr:012X}: BF810000")
        return "\n".join(lines) + "\n"

    assert dpp_hazards(listing(False)) == []          # linear walk: v_add, s_nop 1 in front -> looks padded
    found = dpp_hazards_cfg(listing(False))           # along the branch: v_mul -> s_branch (one wait state) -> DPP read
    assert len(found) == 1 and found[0][4].startswith("v_mul_f64 v[22:23]"), found
    assert dpp_hazards_cfg(listing(True)) == []       # s_nop 1 + the branch itself: three wait states on that path

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
        bad = dpp_hazards_cfg(text)  # ... and along every control-flow path into a DPP instruction
        assert not bad, f"{name}: {len(bad)} DPP reads within 2 wait states of a VALU write on some path, first: {bad[:3]}"
