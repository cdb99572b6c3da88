"""The C-ABI library loads and exports exactly what include/capdec.h declares (no compute, no GPU)."""
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "capdec.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(capdec_[a-z0-9_]+)\s*\(", src)))


def test_header_binding_and_library_agree():
    from imagecaption_amd import _capi
    declared = _declared()
    assert len(declared) >= 20
    assert sorted(_capi.SIGNATURES) == declared
    lib = _capi.lib()  # raises if the .so is missing or lacks a declared symbol
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.capdec_abi_version() == 2


def test_argument_errors_are_reported_without_a_gpu():
    from imagecaption_amd import _capi
    lib = _capi.lib()
    assert lib.capdec_gemm(0, 1, -1, 4, 4, None, 4, None, 4, None, 4, None, 0, None, 0, None) == -1
    assert b"capdec_gemm" in lib.capdec_last_error()
    assert lib.capdec_attn_fwd(None, 0, None, 0, None, 0, None, 0, None, 1, 1, 4, 4, 48, 0, 1.0, 0.0, 0, None) == -1
    assert b"head dim" in lib.capdec_last_error()
    # M*N >= 2^32 is refused only when the epilogue has dropout (32-bit mask index); here: refused for the null operands
    ep = _capi.Epilogue()
    ep.drop_p = 0.1
    import ctypes as C
    assert lib.capdec_gemm(0, 1, 1 << 17, 1 << 15, 8, 16, 8, 16, 8, 16, 1 << 15, C.byref(ep), 0, None, 0, None) == -1
    assert b"dropout" in lib.capdec_last_error()
    assert lib.capdec_gemm_workspace_bytes(512, 512, 5120) > 0
    assert lib.capdec_gemm_workspace_bytes(5120, 10000, 1024) == 0


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from imagecaption_amd import ops
    from imagecaption_amd._capi import CapdecError
    with pytest.raises(CapdecError, match="no CPU fallback"):
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4), 0, 1)
