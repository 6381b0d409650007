"""CPU: the C-ABI library builds for gfx950, loads, and exports exactly what include/hvpr_amd.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "hvpr_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hvpr_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    from hvpr_amd import build
    return build.build()


def test_header_symbols_exported(built):
    L = ctypes.CDLL(built)
    names = _declared()
    assert len(names) >= 8
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/hvpr_amd.h but not exported"


def test_binding_table_matches_header(built):
    from hvpr_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    L = _lib.lib()
    from hvpr_amd import _lib
    assert L.hvpr_abi_version() == _lib.ABI_VERSION == 6
    assert L.hvpr_status_string(0) == b"ok"
    assert L.hvpr_status_string(-2).startswith(b"unsupported")


def test_workspace_queries_need_no_gpu(built):
    from hvpr_amd import _lib
    L = _lib.lib()
    assert L.hvpr_voxelize_workspace_bytes(1, 16384, 296, 248, 1) > 3 * 296 * 248 * 4
    assert L.hvpr_voxelize_workspace_bytes(0, 16384, 296, 248, 1) == 0
    assert L.hvpr_scatter_workspace_bytes(2, 296, 248) == 2 * 296 * 248 * 4


def test_no_cpu_fallback():
    import torch
    from hvpr_amd import kernels
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        kernels.memory_readout_fwd(torch.zeros(4, 64), torch.zeros(2000, 64), 20)
