"""The C-ABI library loads on a CPU-only box and exports every symbol include/mi_gp.h declares."""
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "mi_gp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_gp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from andvaranaut_amd import _lib

    lib = _lib.load()
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"libmi_gp.so does not export {n}"
    assert sorted(_lib.EXPORTS) == names, "andvaranaut_amd/_lib.py:EXPORTS is out of sync with include/mi_gp.h"


def test_library_exports_nothing_but_the_c_abi():
    """the C-ABI is the boundary -- no internal launcher (migp::launch_gemm_f64, chol_panel_blocks, ...), kernel
    handle object or C++ runtime template instantiation is linkable from outside (-fvisibility=hidden + csrc/libmi_gp.map)."""
    import shutil
    import subprocess

    import pytest

    from andvaranaut_amd import _lib

    nm = shutil.which("nm") or shutil.which("llvm-nm", path="/opt/rocm/lib/llvm/bin")
    if nm is None:
        pytest.skip("no nm")
    out = subprocess.check_output([nm, "-D", "--defined-only", _lib.LIB_PATH], text=True)
    syms = [line.split() for line in out.splitlines() if line.strip()]
    assert len(syms) >= 10
    names = sorted(s[-1] for s in syms)
    assert names == _declared(), sorted(set(names) ^ set(_declared()))
    assert all(s[-2] == "T" for s in syms), [s for s in syms if s[-2] != "T"]


def test_bad_arguments_are_reported_not_crashed():
    import ctypes

    from andvaranaut_amd import _lib

    lib = _lib.load()
    # no GPU needed: argument validation happens before any HIP call
    assert lib.mi_gp_gemm_f64(0, 1, 100, 128, 16, 1.0, None, 16, None, 16, 0.0, None, 128, 0, 0, 1, 0, 0, 0, None) == -1
    assert b"multiples" in lib.mi_gp_last_global_error()
    cfg = _lib.MiGpConfig()
    cfg.n, cfg.d, cfg.nkern = 0, 1, 1
    h = ctypes.c_void_p()
    assert lib.mi_gp_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"mi_gp_create" in lib.mi_gp_last_global_error()
    # the sharded-gradient blocks validate before touching the device too
    assert lib.mi_gp_trsm_block(None, 16, None, 0, 1, None, 16, 128, None) == -1
    assert lib.mi_gp_trmv_upper(None, 16, None, 4, None, None) == -1
    assert lib.mi_gp_grad_contract_block_scratch(1000, 512, 512, 10) == (16 - 8) * 8 * 10  # 16 tile rows, slab = columns 8..15
    # batched entry points: null handle / buffers
    assert lib.mi_gp_set_batch(None, None) == -1
    assert lib.mi_gp_lml_batch(None, 1, None, None, None) == -1
    assert lib.mi_gp_lml_grad_batch(None, 1, None, None, None, None) == -1
    ids = (ctypes.c_int * 4)(0, 0, 0, 0)
    assert lib.mi_gp_grad_contract_block(2, 1, ids, ids, None, None, 100, None, 16, 64, 0, 64, None, None, 0, None, None) == -1


def test_product_package_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "andvaranaut_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{f} mentions the oracle"
