#!/usr/bin/env python3
"""Static "read before write" check on gfx950 code objects (round 4; DESIGN.md section 5.6 / tests/test_isa_uninit.py).

A wave starts with whatever the previous wave left in its SGPRs / VGPRs / AGPRs and scratch: a register (or an SGPR-spill
lane of a VGPR, or a scratch slot) that is read on some path before anything wrote it makes the kernel's result depend on
the process's history -- exactly the signature of round 3's nondeterministic grad_x_kernel<4,1>, which was bit-stable in a
fresh process and wrong after other kernels had run.  This tool disassembles every kernel of an object / shared library,
builds its control-flow graph and runs a forward "definitely written" dataflow over

  * scalar registers s0..s105, vcc, m0 (exact: SALU / SMEM writes are unconditional),
  * SGPR-spill slots (VGPR, lane) of v_writelane_b32 / v_readlane_b32 with immediate lanes (exact),
  * accumulation registers a0..a255 (used as VGPR spill space by the compiler),
  * private-segment (scratch) dwords at constant offsets,
  * vector registers v0..v255 (approximate: a write under a partial EXEC mask counts as a write),

and reports every read that is not dominated by writes on ALL paths from the kernel entry.

usage: isa_uninit_check.py <file.o|file.so|file.dis> [--kernel SUBSTR] [--verbose] [--all-paths]
(default: only the findings caused by vector writes that execute with EXEC == 0; --all-paths adds the path-insensitive ones)
Exit code 1 if any kernel has findings.
"""
import re, subprocess, sys, os, tempfile
from collections import defaultdict

LLVM = "/opt/rocm/lib/llvm/bin"

NO_REG_OPS = re.compile(r"^(s_waitcnt|s_nop|s_barrier|s_endpgm|s_branch|s_cbranch|s_setprio|s_sleep|s_sethalt|s_trap|s_icache_inv|"
                        r"s_dcache|s_code_end|s_sendmsg|s_setreg_imm32|s_inst_prefetch|s_clause|s_waitcnt_|buffer_wbl2|buffer_inv|"
                        r"s_ttracedata|s_incperflevel|s_decperflevel|s_wakeup)")
ALL_SRC = re.compile(r"^(global_store|scratch_store|flat_store|buffer_store|ds_write|ds_store|s_cmp_|s_cmpk_|s_bitcmp|"
                     r"global_atomic_(?!.*_rtn)|ds_(add|sub|min|max|and|or|xor)_(?!rtn)|s_setreg_b32|s_store|v_cmpx|s_setpc|"
                     r"s_set_gpr_idx|s_cbranch_g_fork|s_rfe|exp\b|s_atc_probe|ds_gws|global_wb|global_inv)")
TWO_DST = re.compile(r"^(v_div_scale_f(32|64)|v_mad_u64_u32|v_mad_i64_i32|v_add_co_u32|v_sub_co_u32|v_subrev_co_u32|"
                     r"v_addc_co_u32|v_subb_co_u32|v_subbrev_co_u32|v_add_co_ci|s_swappc)")
DST_IS_SRC = re.compile(r"^(v_fmac_|v_mac_|v_dot\w*acc|s_addk_i32|s_mulk_i32|s_cmovk|s_cmov_|v_movrel|v_pk_fmac|s_bitset|"
                        r"v_writelane_b32_dummy)")


def regs_of(tok):
    """register names a single operand token covers"""
    t = tok.strip().lstrip("-").strip("|")
    t = re.sub(r"^(neg|abs|sext)\((.*)\)$", r"\2", t)
    m = re.match(r"^([sva])\[(\d+):(\d+)\]$", t)
    if m:
        return [f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    m = re.match(r"^([sva])(\d+)$", t)
    if m:
        return [t]
    if t in ("vcc", "vcc_lo", "vcc_hi"):
        return ["vcc"]
    if t == "m0":
        return ["m0"]
    return []


def split_ops(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    # drop modifiers (offset:.., row_newbcast:.., quad_perm:[..], op_sel:[..], glc, slc, nt, sc0, sc1 ...): they follow a space
    clean = []
    for o in out:
        first = o.split()[0] if o.split() else ""
        clean.append(first)
    return clean


class Inst:
    __slots__ = ("addr", "mn", "ops", "target", "text", "mods")

    def __init__(self, addr, mn, ops, target, text, mods):
        self.addr, self.mn, self.ops, self.target, self.text, self.mods = addr, mn, ops, target, text, mods


def parse_disassembly(text):
    """{kernel name: [Inst]} from llvm-objdump -d output (with // addr: encoding <sym+off> comments)"""
    kernels, cur, base = {}, None, 0
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]{8,16}) <([^>]+)>:", line)
        if m:
            base = int(m.group(1), 16)
            cur = kernels.setdefault(m.group(2), [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        body, _, comment = line.partition("//")
        body = body.strip()
        if not body:
            continue
        am = re.match(r"\s*([0-9A-Fa-f]+):", comment)
        if not am:
            continue
        addr = int(am.group(1), 16)
        parts = body.split(None, 1)
        mn = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        target = None
        tm = re.search(r"<[^>]*?\+0x([0-9a-fA-F]+)>", comment)
        if mn.startswith("s_cbranch") or mn == "s_branch":
            if tm:
                target = base + int(tm.group(1), 16)
            else:  # target is the symbol itself
                tm2 = re.search(r"<[^>+]+>", comment)
                target = base if tm2 else None
        cur.append(Inst(addr, mn, split_ops(rest) if not NO_REG_OPS.match(mn) else [], target, body, rest))
    return kernels


def defs_uses(i):
    """(defs, uses, lane_defs, lane_uses, scratch_defs, scratch_uses) of one instruction"""
    mn, ops = i.mn, i.ops
    d, u, ld, lu, sd, su = [], [], [], [], [], []
    if NO_REG_OPS.match(mn) or not ops:
        return d, u, ld, lu, sd, su
    if mn.startswith("v_writelane_b32"):
        u += regs_of(ops[1])
        lane = ops[2]
        if lane.isdigit():
            ld.append((ops[0], int(lane)))
        else:
            u += regs_of(lane)
        return d, u, ld, lu, sd, su
    if mn.startswith("v_readlane_b32"):
        d += regs_of(ops[0])
        lane = ops[2]
        if lane.isdigit():
            lu.append((ops[1], int(lane)))
        else:
            u += regs_of(lane) + regs_of(ops[1])
        return d, u, ld, lu, sd, su
    if mn.startswith("scratch_"):
        off = 0
        om = re.search(r"offset:(-?\d+)", i.mods)
        if om:
            off = int(om.group(1))
        width = {"dword": 1, "dwordx2": 2, "dwordx3": 3, "dwordx4": 4, "ubyte": 1, "sbyte": 1, "ushort": 1, "sshort": 1,
                 "byte": 1, "short": 1}.get(mn.split("_")[-1], 1)
        if mn.startswith("scratch_store"):
            # scratch_store data: vdata is ops[1]
            u += regs_of(ops[1])
            const = ops[0] == "off" and ops[2] == "off"
            u += regs_of(ops[0]) + regs_of(ops[2])
            (sd if const else d).extend([("scr", off + 4 * k) for k in range(width)] if const else [])
            if not const:
                sd.append(("scr_dyn", None))
        else:
            d += regs_of(ops[0])
            const = ops[1] == "off" and ops[2] == "off"
            u += regs_of(ops[1]) + regs_of(ops[2])
            if const:
                su += [("scr", off + 4 * k) for k in range(width)]
            else:
                su.append(("scr_dyn", None))
        return d, u, ld, lu, sd, su
    if ALL_SRC.match(mn):
        for o in ops:
            u += regs_of(o)
        return d, u, ld, lu, sd, su
    if re.match(r"^v_cmp_\w+_e32$", mn) or re.match(r"^v_cmp_\w+_(sdwa|dpp)$", mn):
        # vcc written implicitly (or named as first operand in some syntaxes)
        if ops and ops[0] in ("vcc", "vcc_lo"):
            ops = ops[1:]
        d.append("vcc")
        for o in ops:
            u += regs_of(o)
        return d, u, ld, lu, sd, su
    ndst = 2 if TWO_DST.match(mn) else 1
    for o in ops[:ndst]:
        d += regs_of(o)
    for o in ops[ndst:]:
        u += regs_of(o)
    if DST_IS_SRC.match(mn) or mn.startswith("v_mfma") and False:
        u += regs_of(ops[0])
    if mn.startswith("s_and_saveexec") or mn.startswith("s_or_saveexec") or mn.startswith("s_andn2_saveexec"):
        pass
    if mn.endswith("_e32") and (mn.startswith("v_cndmask") or mn.startswith("v_addc") or mn.startswith("v_subb")):
        u.append("vcc")
    if mn.startswith("v_div_fmas"):
        u.append("vcc")
    if mn.startswith("s_cbranch_vcc"):
        u.append("vcc")
    return d, u, ld, lu, sd, su


def is_exec_write(i):
    return bool(EXEC_WRITE.match(i.mn)) and (i.mn.startswith("v_cmpx") or "saveexec" in i.mn or (bool(i.ops) and i.ops[0] == "exec"))


def is_vector(i):
    return i.mn.startswith(("v_", "ds_", "scratch_", "global_", "flat_", "buffer_")) and not i.mn.startswith(
        ("v_readlane", "v_writelane", "v_readfirstlane"))


def analyse(insts, entry_defined, exec_aware=True):
    """Forward must-be-written dataflow; returns a list of (inst, description, register class) findings.

    EXEC-aware on one edge kind: `s_cbranch_execz T` is taken exactly when no lane is active, so on that edge the vector
    instructions from T up to the instruction that rewrites EXEC write nothing.  The edge is routed around that window:
    it enters the block that starts at the EXEC write and contributes only the window's scalar definitions."""
    addr2idx = {i.addr: k for k, i in enumerate(insts)}
    leaders = {0}
    window_end = {}  # index of an execz target -> index of the EXEC write that ends its EXEC == 0 window (or None)
    for k, i in enumerate(insts):
        if i.mn == "s_branch" or i.mn.startswith("s_cbranch") or i.mn in ("s_endpgm", "s_setpc_b64"):
            if k + 1 < len(insts):
                leaders.add(k + 1)
            if i.target is not None and i.target in addr2idx:
                leaders.add(addr2idx[i.target])
        if i.mn == "s_cbranch_execz" and i.target in addr2idx:
            t = addr2idx[i.target]
            e, has_vec = t, False
            zero = set()  # SGPRs that hold an EXEC mask saved INSIDE the window, i.e. zero
            while e < len(insts):
                j = insts[e]
                if is_exec_write(j):
                    # The window ends only at an EXEC write that can ENABLE lanes.  exec &= x cannot; s_and_saveexec saves the
                    # current (zero) mask into its destination, and OR-ing such a saved zero back cannot either.  (hipcc emits
                    # `s_cbranch_execz +1; s_branch +1` pairs at kernel entry and chains of s_and_saveexec / s_or_b64 exec
                    # around predicated loads: without this rule everything behind them looked unwritten.)
                    if j.mn.startswith("s_and_saveexec") or j.mn.startswith("s_andn2_saveexec"):
                        zero.update(regs_of(j.ops[0]))
                        has_vec = has_vec or False
                        e += 1
                        continue
                    if j.mn in ("s_and_b64", "s_andn2_b64", "s_and_b32", "s_andn2_b32") and j.ops and j.ops[0] == "exec" and "exec" in j.ops[1:]:
                        e += 1
                        continue
                    if j.mn in ("s_or_b64", "s_or_b32") and len(j.ops) == 3 and j.ops[0] == "exec" and \
                            all(o == "exec" or (regs_of(o) and set(regs_of(o)) <= zero) for o in j.ops[1:]):
                        e += 1
                        continue
                    break
                if not is_vector(j):
                    zero.difference_update(defs_uses(j)[0])  # a scalar write over a saved mask: no longer known to be zero
                if j.mn.startswith("s_cbranch") or j.mn in ("s_branch", "s_endpgm", "s_setpc_b64", "s_barrier"):
                    e = None
                    break
                has_vec = has_vec or is_vector(j)
                e += 1
            if exec_aware and e is not None and e < len(insts) and e > t and has_vec:
                window_end[t] = e
                leaders.add(e)
    leaders = sorted(leaders)
    bidx = {l: n for n, l in enumerate(leaders)}
    blocks = []
    for n, l in enumerate(leaders):
        e = leaders[n + 1] if n + 1 < len(leaders) else len(insts)
        blocks.append((l, e))
    du = [defs_uses(i) for i in insts]

    def scalar_gen(lo, hi):
        g = set()
        for k in range(lo, hi):
            if not is_vector(insts[k]):
                d, u, ld, lu, sd, su = du[k]
                g.update(d); g.update(ld)
        return g

    succ = defaultdict(list)       # n -> [(m, extra definitions carried by the edge)]
    for n, (l, e) in enumerate(blocks):
        last = insts[e - 1]
        if last.mn == "s_endpgm" or last.mn == "s_setpc_b64":
            continue
        if last.mn == "s_branch":
            if last.target in addr2idx:
                succ[n].append((bidx[addr2idx[last.target]], frozenset()))
            continue
        if last.mn.startswith("s_cbranch") and last.target in addr2idx:
            t = addr2idx[last.target]
            if last.mn == "s_cbranch_execz" and t in window_end:
                succ[n].append((bidx[window_end[t]], frozenset(scalar_gen(t, window_end[t]))))
            else:
                succ[n].append((bidx[t], frozenset()))
        if e < len(insts):
            succ[n].append((bidx[e], frozenset()))
    pred = defaultdict(list)
    for a, ss in succ.items():
        for b, extra in ss:
            pred[b].append((a, extra))
    gen = []
    for (l, e) in blocks:
        g = set()
        for k in range(l, e):
            d, u, ld, lu, sd, su = du[k]
            g.update(d); g.update(ld); g.update(sd)
        gen.append(g)
    TOP = None  # "everything" for unvisited blocks
    IN = [TOP] * len(blocks)
    OUT = [TOP] * len(blocks)
    IN[0] = set(entry_defined)
    work = [0]
    while work:
        n = work.pop()
        if n != 0:
            ins = [OUT[p] | extra for p, extra in pred[n] if OUT[p] is not TOP]
            if not ins:
                continue
            new_in = set.intersection(*ins) if len(ins) > 1 else set(ins[0])
            if IN[n] is not TOP and new_in == IN[n] and OUT[n] is not TOP:
                continue
            IN[n] = new_in
        out = IN[n] | gen[n]
        if OUT[n] is TOP or out != OUT[n]:
            OUT[n] = out
            work.extend(m for m, _ in succ[n])
    findings = []

    def where(n, l, r):
        lacking = [hex(insts[blocks[p][1] - 1].addr) for p, extra in pred[n] if OUT[p] is not TOP and r not in OUT[p] and r not in extra]
        return f"[block at {insts[l].addr:#x}; not written on the paths through {lacking[:4]}]"

    for n, (l, e) in enumerate(blocks):
        if IN[n] is TOP:
            continue  # unreachable
        cur = set(IN[n])
        for k in range(l, e):
            d, u, ld, lu, sd, su = du[k]
            for r in u:
                if r not in cur:
                    cls = "vgpr" if r.startswith("v") and r != "vcc" else "agpr" if r.startswith("a") else "sgpr"
                    findings.append((insts[k], f"{r}  {where(n, l, r)}", cls))
            for r in lu:
                if r not in cur:
                    findings.append((insts[k], f"SGPR-spill lane {r[1]} of {r[0]}  {where(n, l, r)}", "lane"))
            for r in su:
                if r[1] is not None and r not in cur and ("scr_dyn", None) not in cur:
                    findings.append((insts[k], f"scratch dword at offset {r[1]}", "scratch"))
            cur.update(d); cur.update(ld); cur.update(sd)
    return findings


EXEC_WRITE = re.compile(r"^(s_or_b64|s_mov_b64|s_and_b64|s_andn2_b64|s_xor_b64|s_or_saveexec_b64|s_and_saveexec_b64|"
                        r"s_andn2_saveexec_b64|s_xor_saveexec_b64|s_wqm_b64|s_cselect_b64|v_cmpx)")


def kernel_descriptors(path):
    """{kernel: (user_sgprs + system sgprs enabled, workitem id dims)} from the code object's notes (best effort)"""
    return {}


def disassemble(path):
    if path.endswith(".dis"):
        return open(path).read()
    tmp = tempfile.mkdtemp(prefix="isachk")
    data = open(path, "rb").read()
    co = path
    if b"__CLANG_OFFLOAD_BUNDLE__" in data:  # host object / shared library with an embedded fat binary
        local = os.path.join(tmp, os.path.basename(path))
        with open(local, "wb") as f:
            f.write(data)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, capture_output=True)
        dev = [f for f in os.listdir(tmp) if "amdgcn" in f and f.startswith(os.path.basename(path))]
        if not dev:
            return ""
        co = os.path.join(tmp, dev[0])
    return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", co], text=True)


def check_text(text, only=None, verbose=False, classes=("agpr", "lane", "scratch", "sgpr", "vgpr"), exec_only=True):
    """(number of findings in `classes`, {kernel: findings}).  VGPR findings are printed only with verbose: a vector register
    written under a partial EXEC mask and a path-insensitive join both look like "not written on every path" and are
    common in correct code; the exact classes (AGPR spill space, SGPR-spill lanes, scalar registers, constant-offset scratch
    slots) are what tests/test_isa_uninit.py asserts on."""
    kernels = parse_disassembly(text)
    total = 0
    report = {}
    for name, insts in kernels.items():
        if only and only not in name:
            continue
        if not insts or not any(i.mn == "s_endpgm" for i in insts):
            continue
        # ABI: s[0:1] kernarg pointer + up to s2..s5 workgroup ids / private-segment offset, v0 packed work-item ids
        entry = {f"s{k}" for k in range(0, 16)} | {"v0", "v1", "v2"}
        f = analyse(insts, entry, exec_aware=True)
        if exec_only:
            # keep what the EXEC == 0 windows cause: findings that a plain (EXEC-blind) control-flow graph does not produce
            plain = {(i.addr, w.split("  [")[0]) for i, w, c in analyse(insts, entry, exec_aware=False)}
            f = [x for x in f if (x[0].addr, x[1].split("  [")[0]) not in plain]
        hard = [x for x in f if x[2] in classes]
        report[name] = f
        total += len(hard)
        if hard or verbose:
            print(f"{name}: {len(insts)} instructions, {len(hard)} finding(s) in {classes}, {len(f) - len(hard)} other")
            seen = set()
            for inst, what, cls in (f if verbose else hard)[: (None if verbose else 16)]:
                key = (inst.addr, what)
                if key in seen:
                    continue
                seen.add(key)
                print(f"    {inst.addr:#x}: {inst.text:<56s} reads {what}")
    return total, report


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only = None
    if "--kernel" in sys.argv:
        only = sys.argv[sys.argv.index("--kernel") + 1]
        args = [a for a in args if a != only]
    total, _ = check_text(disassemble(args[0]), only, "--verbose" in sys.argv, exec_only="--all-paths" not in sys.argv)
    print("total findings:", total)
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
