"""Register budgets of the tuned kernels, read from the code objects' metadata (no GPU needed).

Round 5 lost 4-14 % on every launch of the 64x64-tile GEMM kernel for half a day because a run-time branch added to it
(the k-segmented flush) changed its register allocation from 172 to 193 VGPRs: parity tests cannot see that, and a noisy
box-to-box clock hid it in the timings until a same-box A/B of two library builds showed it.  These bounds are what the
kernels were tuned with; a change that moves one of them should be a decision, not an accident."""
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_uninit_check as chk  # noqa: E402

CSRC = os.path.join(ROOT, "andvaranaut_amd", "csrc")
# kernel name (substring of the mangled symbol) -> (max VGPRs, spills allowed)
BUDGET = {
    "gemm_f64_kernel_sILb0ELb0ELb0E": (176, False),   # the plain 64x64-tile kernel, NT form: two workgroups per CU, tuned at 172
    "gemm_f64_kernel_sILb0ELb0ELb1E": (176, False),   # ... its k-flush instantiation
    "syrk_thin_kernelILi128E": (128, False),
    "syrk_thin_kernelILi256E": (192, False),
    "trsm_strip128_kernelILi1E": (224, False),        # (210 as tuned: two waves per SIMD)
    # round 6: the contraction is bound by LDS / exp latency -- one component without a rational quadratic runs three waves per SIMD
    # (140 VGPRs; 176 = two waves cost 0.7 ms of the N = 16384 LML + gradient), two components two waves (230)
    "grad_contract_kernelILi1ELb0E": (168, False),
    "grad_contract_kernelILi2ELb0E": (256, False),
}


def _metadata(obj):
    tmp = tempfile.mkdtemp(prefix="regchk")
    local = os.path.join(tmp, os.path.basename(obj))
    with open(local, "wb") as f:
        f.write(open(obj, "rb").read())
    subprocess.run([f"{chk.LLVM}/llvm-objdump", "--offloading", local], check=True, capture_output=True)
    dev = [f for f in os.listdir(tmp) if "amdgcn" in f and f.startswith(os.path.basename(obj))]
    if not dev:
        return ""
    return subprocess.check_output([f"{chk.LLVM}/llvm-readelf", "--notes", os.path.join(tmp, dev[0])], text=True)


def test_tuned_kernels_keep_their_register_budgets():
    objs = [os.path.join(CSRC, f) for f in ("gemm_f64.o", "thin_f64.o", "leaf_f64.o", "grad_predict.o")]
    if not all(os.path.exists(o) for o in objs) or not os.path.exists(os.path.join(chk.LLVM, "llvm-readelf")):
        pytest.skip("csrc/*.o or llvm-readelf missing (run __graft_entry__.build() first)")
    text = "".join(_metadata(o) for o in objs)
    seen = {}
    for block in text.split("- .agpr_count:")[1:]:
        name = re.search(r"^    \.name:\s+(\S+)", block, re.M)  # (the kernel's own entry: argument names sit deeper)
        vg = re.search(r"\.vgpr_count:\s+(\d+)", block)
        sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", block)
        if name and vg:
            seen[name.group(1)] = (int(vg.group(1)), int(sp.group(1)) if sp else 0)
    assert len(seen) >= 10, list(seen)[:5]
    for key, (vmax, spills_ok) in BUDGET.items():
        hits = {k: v for k, v in seen.items() if key in k}
        assert hits, (key, sorted(seen)[:20])
        for k, (vg, sp) in hits.items():
            assert vg <= vmax, (k, vg, vmax)
            assert spills_ok or sp == 0, (k, sp)
