"""The C-ABI library loads (no GPU needed) and exports every symbol include/mfhip.h declares."""
import ctypes
import os
import re

from reflecting_reality_amd import _build, hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "mfhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mf_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_loads():
    path = _build.build(verbose=False)
    assert os.path.exists(path)
    lib = hip.load()
    assert lib.mf_abi_version() == hip.ABI_VERSION


def test_every_declared_symbol_is_exported():
    lib = hip.load()
    declared = header_functions()
    assert declared, "no functions parsed from mfhip.h"
    assert sorted(hip.EXPORTS) == declared, "hip.EXPORTS is out of sync with include/mfhip.h"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in mfhip.h but not exported by libmfhip.so"


def test_descriptor_layouts_match():
    lib = hip.load()
    assert lib.mf_sizeof_gemm_desc() == ctypes.sizeof(hip.GemmDesc)
    assert lib.mf_sizeof_groupnorm_desc() == ctypes.sizeof(hip.GroupNormDesc)
    n = lib.mf_gemm_num_tiles()
    assert n >= 1
    bm, bn = ctypes.c_int(), ctypes.c_int()
    for t in range(1, n + 1):
        assert lib.mf_gemm_tile_shape(t, ctypes.byref(bm), ctypes.byref(bn)) == 0
        assert bm.value % 32 == 0 and bn.value % 32 == 0
    assert lib.mf_gemm_tile_shape(0, ctypes.byref(bm), ctypes.byref(bn)) != 0


def test_argument_errors_are_reported_without_a_gpu():
    lib = hip.load()
    assert lib.mf_gemm_conv(None, None) == -1
    assert b"null descriptor" in lib.mf_last_error()
    d = hip.GemmDesc()
    d.dtype = 7
    assert lib.mf_gemm_conv(ctypes.byref(d), None) == -1
    assert b"bad dtype" in lib.mf_last_error()


def test_ops_refuse_host_tensors():
    import pytest
    import torch
    with pytest.raises(hip.MfhipError):
        hip.silu_f32(torch.zeros(4))
