#!/usr/bin/env python3
"""Build a libmi_gp variant whose grad_predict object comes from a PATCHED device assembly: assembly-level A/B experiments
(wait states, waitcnt, nops around spills ...) on ONE kernel without touching what the compiler generated elsewhere.
Written in round 4 to test hardware-hazard hypotheses for the nondeterministic grad_x_kernel<4,1> of round 3 -- every one
of them was refuted (all patched variants behaved like the unpatched build); the cause was a register-allocator spill
executed with EXEC == 0 (tools/isa_uninit_check.py).

  asm_variant.py <workdir> <name> [patch ...]
      env GX_KSTART  mangled-name prefix of the kernel to patch (default grad_x_kernel<4,1>)
          GX_TREE    tree whose other objects are linked (default: this repository)
          GX_OUT     output directory (default tools/exp/, git-ignored)

<workdir> holds the --save-temps output of `hipcc <library flags> --save-temps -c grad_predict.hip` (device .s, host .s).
The device code object is re-assembled from the patched .s, re-linked, re-bundled and embedded in the host object through
.incbin; libmi_gp_<name>.so is linked from it + the other objects of andvaranaut_amd/csrc/.

Patches: wait_after_scratch_store, wait_after_scratch_any, nop_after_scratch_sgpr, nop_before_scratch_sgpr,
nop_around_lane (v_readlane / v_writelane), nop_after_trans (rcp / rsq / sqrt / exp / log), wait_at_branches.
"""
import os, re, subprocess, sys

LLVM = "/opt/rocm/lib/llvm/bin"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TREE = os.environ.get("GX_TREE", REPO)
KSTART = os.environ.get("GX_KSTART", "_ZN4migp13grad_x_kernelILi4ELi1E")
KEND = ".Lfunc_end"


def sgpr_offset_scratch(line):
    s = line.strip()
    if not s.startswith("scratch_"):
        return False
    ops = [o.strip() for o in s.split(None, 1)[1].split(",")]
    return bool(re.match(r"s\d+", ops[-1].split()[0]))


def patch(lines, what):
    out = []
    for ln in lines:
        s = ln.strip()
        pre, post = [], []
        if what == "wait_after_scratch_store" and s.startswith("scratch_store"):
            post.append("\ts_waitcnt vmcnt(0)\n")
        elif what == "wait_after_scratch_any" and s.startswith("scratch_"):
            post.append("\ts_waitcnt vmcnt(0)\n")
        elif what == "nop_after_scratch_sgpr" and sgpr_offset_scratch(ln):
            post.append("\ts_nop 7\n")
        elif what == "nop_before_scratch_sgpr" and sgpr_offset_scratch(ln):
            pre.append("\ts_nop 7\n")
        elif what == "nop_around_lane" and (s.startswith("v_readlane_b32") or s.startswith("v_writelane_b32")):
            pre.append("\ts_nop 4\n"); post.append("\ts_nop 4\n")
        elif what == "nop_after_trans" and re.match(r"v_(rcp|rsq|sqrt|exp|log)_f(64|32)", s):
            post.append("\ts_nop 1\n")
        elif what == "wait_at_branches" and (re.match(r"s_c?branch", s) or re.match(r"\.LBB\d+_\d+:", s)):
            pre.append("\ts_waitcnt vmcnt(0) lgkmcnt(0)\n")
        out += pre + [ln] + post
    return out


def main():
    work, name, patches = sys.argv[1], sys.argv[2], sys.argv[3:]
    dev_s = [f for f in os.listdir(work) if f.endswith("gfx950.s")][0]
    host_s = [f for f in os.listdir(work) if f.endswith("host-x86_64-unknown-linux-gnu.s")][0]
    lines = open(os.path.join(work, dev_s)).readlines()
    a = next(i for i, l in enumerate(lines) if l.startswith(KSTART))
    b = next(i for i in range(a, len(lines)) if lines[i].startswith(KEND))
    body = lines[a + 1 : b]
    for p in patches:
        body = patch(body, p)
    outdir = os.path.join(work, "v_" + name)
    os.makedirs(outdir, exist_ok=True)
    ps = os.path.join(outdir, "dev.s")
    open(ps, "w").writelines(lines[: a + 1] + body + lines[b:])
    run = lambda *c: subprocess.check_call(list(c))
    run(f"{LLVM}/clang", "-cc1as", "-triple", "amdgcn-amd-amdhsa", "-filetype", "obj", "-target-cpu", "gfx950",
        "-mrelocation-model", "pic", "-mllvm", "-amdgpu-mfma-vgpr-form", "-o", f"{outdir}/dev.o", ps)
    run(f"{LLVM}/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", f"{outdir}/dev.out",
        f"{outdir}/dev.o")
    run(f"{LLVM}/clang-offload-bundler", "-type=o", "-bundle-align=4096",
        "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null",
        f"-input={outdir}/dev.out", f"-output={outdir}/dev.hipfb")
    hl = open(os.path.join(work, host_s)).readlines()
    fb = next(i for i, l in enumerate(hl) if l.lstrip().startswith(".asciz") and "__CLANG_OFFLOAD_BUNDLE__" in l)
    label = hl[fb - 1].strip().rstrip(":")
    hl[fb] = f'\t.incbin "{outdir}/dev.hipfb"\n'
    assert hl[fb + 1].strip().startswith(".size")
    hl[fb + 1] = f"\t.size\t{label}, {os.path.getsize(f'{outdir}/dev.hipfb')}\n"
    open(f"{outdir}/host.s", "w").writelines(hl)
    run(f"{LLVM}/clang", "-cc1as", "-triple", "x86_64-unknown-linux-gnu", "-filetype", "obj", "-target-cpu", "x86-64",
        "-mrelocation-model", "pic", "-o", f"{outdir}/grad_predict.o", f"{outdir}/host.s")
    csrc = os.path.join(TREE, "andvaranaut_amd", "csrc")
    others = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".o") and f != "grad_predict.o"]
    outd = os.environ.get("GX_OUT", os.path.join(REPO, "tools", "exp"))
    os.makedirs(outd, exist_ok=True)
    lib = os.path.join(outd, f"libmi_gp_{name}.so")
    run("/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", *others, f"{outdir}/grad_predict.o", "-o", lib)
    print("built", lib, "patches:", patches, "kernel body lines:", len(body))


if __name__ == "__main__":
    main()
