"""Structural guard against the compiler defect behind round 3's nondeterministic dLML/dX.

hipcc (ROCm 7.2.0, -O3) placed the VGPR -> AGPR spill copies of grad_x_kernel<4,1>'s accumulators in the exit block of a
divergent loop ABOVE the `s_or_b64 exec, exec, sN` that restores EXEC.  That loop exits through `s_cbranch_execz` only
when EXEC == 0, so the copies copied no lane and the reloads ~13k instructions later returned whatever the SIMD's
physical registers held from earlier waves: correct in a fresh process, garbage after other kernels had run.  The path it
sits on: the reference differentiates the same Cholesky w.r.t. warped inputs by autodiff (gpmcmc.py:211-279,1096-1165).

tools/isa_uninit_check.py disassembles a code object, builds each kernel's control-flow graph and runs a "definitely
written" dataflow in which the `s_cbranch_execz` edge skips the vector writes of the EXEC == 0 window.  Tests:
  * the checker flags the real failing listing (tests/golden/gradx41_round3_failing.dis.xz: llvm-objdump of the round-3
    object) and accepts the same listing with the eight copies moved below the EXEC restore -- the 72-byte reorder that
    made the failing binary immune to register poison on the GPU;
  * no kernel of ANY object of the library has such a read (every kernel, whatever its scratch / spill count)."""
import lzma
import os
import re
import sys

import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_uninit_check as chk  # noqa: E402


def _failing_listing():
    with lzma.open(os.path.join(GOLDEN, "gradx41_round3_failing.dis.xz"), "rt") as f:
        return f.read()


def test_checker_flags_round3_failing_kernel_and_accepts_the_reordered_block():
    text = _failing_listing()
    total, report = chk.check_text(text)
    (findings,) = report.values()
    agpr = sorted({w.split()[0] for _, w, cls in findings if cls == "agpr"})
    assert agpr == [f"a{i}" for i in range(8)], agpr   # the eight halves of acc[0][0..3]
    assert all(i.mn == "v_accvgpr_read_b32" for i, _, cls in findings if cls == "agpr")
    # move `s_or_b64 exec, exec, s[24:25]` (and the s_mov in front of it) above the eight v_accvgpr_write_b32
    lines = text.splitlines()
    w = [k for k, l in enumerate(lines) if "v_accvgpr_write_b32" in l]
    assert len(w) == 8 and w == list(range(w[0], w[0] + 8))
    restore = lines[w[-1] + 1 : w[-1] + 3]
    assert "s_mov_b32" in restore[0] and "s_or_b64 exec, exec" in restore[1]
    block = lines[w[0] : w[0] + 8]
    addrs = [re.search(r"// ([0-9A-F]+):", l).group(1) for l in block + restore]

    def readdress(ls):  # keep addresses ascending as in a real listing (8-byte VOP3P, 4-byte SOP)
        out, a = [], int(addrs[0], 16)
        for l in ls:
            out.append(re.sub(r"// [0-9A-F]+:", f"// {a:016X}:", l))
            a += 4 if l.lstrip().startswith("s_") else 8
        return out

    fixed = lines[: w[0]] + readdress([restore[1], restore[0]] + block) + lines[w[-1] + 3 :]
    total_fixed, _ = chk.check_text("\n".join(fixed))
    assert total > 0 and total_fixed == 0


def test_checker_on_a_minimal_hand_made_listing():
    bad = """
0000000000001000 <k>:
	v_mov_b32_e32 v5, 0                                        // 000000001000: 7E0A0280
	s_mov_b64 s[6:7], 0                                        // 000000001004: BE860180
	v_cmp_gt_i32_e32 vcc, 4, v0                                // 000000001008: 7D880084
	s_or_b64 s[6:7], vcc, s[6:7]                               // 00000000100C: 8786066A
	s_andn2_b64 exec, exec, s[6:7]                             // 000000001010: 89FE067E
	s_cbranch_execz 1                                          // 000000001014: BF880001 <k+0x1c>
	s_branch 65531                                             // 000000001018: BF82FFFB <k+0x8>
	v_accvgpr_write_b32 a0, v5                                 // 00000000101C: D3D94000 18000105
	s_or_b64 exec, exec, s[6:7]                                // 000000001024: 87FE067E
	v_accvgpr_read_b32 v6, a0                                  // 000000001028: D3D84006 18000100
	s_endpgm                                                   // 000000001030: BF810000
"""
    good = bad.replace("""	v_accvgpr_write_b32 a0, v5                                 // 00000000101C: D3D94000 18000105
	s_or_b64 exec, exec, s[6:7]                                // 000000001024: 87FE067E
""", """	s_or_b64 exec, exec, s[6:7]                                // 00000000101C: 87FE067E
	v_accvgpr_write_b32 a0, v5                                 // 000000001020: D3D94000 18000105
""")
    nbad, rep = chk.check_text(bad)
    assert nbad == 1 and rep["k"][0][2] == "agpr"
    assert chk.check_text(good)[0] == 0


def test_no_kernel_of_the_library_reads_state_that_was_saved_with_exec_zero():
    csrc = os.path.join(ROOT, "andvaranaut_amd", "csrc")
    objs = sorted(f for f in os.listdir(csrc) if f.endswith(".o"))
    if "grad_predict.o" not in objs or not os.path.exists(os.path.join(chk.LLVM, "llvm-objdump")):
        pytest.skip("csrc/*.o or llvm-objdump missing (run __graft_entry__.build() first)")
    nkernels = 0
    for name in objs:
        text = chk.disassemble(os.path.join(csrc, name))
        if not text:
            continue  # host-only translation unit
        total, report = chk.check_text(text)
        nkernels += len(report)
        bad = {k: [(hex(i.addr), i.text, w) for i, w, _ in v][:4] for k, v in report.items() if v}
        assert total == 0, f"{name}: registers / spill slots read after a save that ran with EXEC == 0: {bad}"
    assert nkernels >= 40, nkernels  # every kernel of the library was looked at (grad_x / grad_contract / predict_grad x NK, ...)
