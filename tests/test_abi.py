"""The C-ABI library loads on a CPU-only host and exports every symbol that
include/bz_abi.h declares; device entry points refuse to run without a GPU."""
import ctypes as C
import os
import re

import pytest

from betazero_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "bz_abi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bz_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound():
    L = _lib.lib()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in bz_abi.h but not exported"
    assert set(names) == set(_lib.ABI_SYMBOLS)
    assert L.bz_abi_version() == _lib.ABI_VERSION and L.bz_build_info() == b"product"


def test_struct_sizes_match_header():
    assert C.sizeof(_lib.EngineCfg) == 80
    assert C.sizeof(_lib.EngineLayout) == 21 * 8 + 8 + 3 * 8


def test_argument_validation_without_gpu():
    L = _lib.lib()
    out = C.c_uint64()
    assert L.bz_reversi_legal(0, 0, 5, C.byref(out)) == _lib.BZ_EINVAL
    assert b"size" in L.bz_last_error()
    assert L.bz_reversi_apply(0, 0, 8, 0, 0, C.byref(out), C.byref(out), None) == _lib.BZ_EILLEGAL_MOVE
    assert L.bz_engine_workspace_bytes(None) == -1
    cfg = _lib.EngineCfg(1, 4, 8, 0, 1.5, 0, 0, 1, 64, 0, 0, 0, 4, 0, 0.0, 0.0, 0)
    assert L.bz_engine_workspace_bytes(C.byref(cfg)) > 0


@pytest.mark.skipif(_lib.lib().bz_device_count() > 0, reason="CPU-only check")
def test_product_path_fails_loudly_without_gpu():
    from betazero_amd.engine import SelfPlayEngine
    with pytest.raises(RuntimeError, match="no HIP device"):
        SelfPlayEngine("ttt", 4, 8)
    with pytest.raises(RuntimeError, match="no HIP device"):
        import betazero_amd as bz
        bz.MCTSPlayer(1, sims=8).get_move(bz.TicTacToeBoard())
