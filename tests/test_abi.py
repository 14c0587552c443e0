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
    # the training structs, against the header itself (compiled by gcc): pointer structs are easy to get out of step
    import os
    import subprocess
    import tempfile
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "s.c")
        open(src, "w").write('#include <stdio.h>\n#include "bz_abi.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", '
                             "sizeof(bz_train_head_params), sizeof(bz_train_tensors), sizeof(bz_train_partials), sizeof(bz_train_batch), "
                             "sizeof(bz_train_adam), sizeof(bz_engine_cfg)); return 0; }\n")
        subprocess.check_call(["gcc", "-I", inc, "-o", os.path.join(d, "s"), src])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "s")]).split()]
    assert sizes == [C.sizeof(t) for t in (_lib.TrainHeadParams, _lib.TrainTensors, _lib.TrainPartials, _lib.TrainBatch, _lib.TrainAdam,
                                            _lib.EngineCfg)], sizes


def test_argument_validation_without_gpu():
    L = _lib.lib()
    out = C.c_uint64()
    assert L.bz_reversi_legal(0, 0, 9, C.byref(out)) == _lib.BZ_EINVAL and L.bz_reversi_legal(0, 0, 0, C.byref(out)) == _lib.BZ_EINVAL
    assert L.bz_reversi_legal(0, 0, 5, C.byref(out)) == _lib.BZ_OK  # sizes 1..8: every size whose cells fit (reversi_board.py:4-14 is generic)
    assert b"size" in L.bz_last_error()
    assert L.bz_reversi_apply(0, 0, 8, 0, 0, C.byref(out), C.byref(out), None) == _lib.BZ_EILLEGAL_MOVE
    assert L.bz_engine_workspace_bytes(None) == -1
    cfg = _lib.EngineCfg(1, 4, 8, 0, 1.5, 0, 0, 1, 64, 0, 0, 0, 4, 0, 0.0, 0.0, 0)
    assert L.bz_engine_workspace_bytes(C.byref(cfg)) > 0
    # the training entry points refuse shapes they are not built for, and null pointers, before anything is launched
    sizes = (C.c_int32 * 6)()
    assert L.bz_train_ends_sizes(96, 64, sizes) == _lib.BZ_EINVAL and b"64 or 128" in L.bz_last_error()
    assert L.bz_train_ends_sizes(64, 6, sizes) == _lib.BZ_EINVAL                      # the batch must be a multiple of 4
    assert L.bz_train_ends_sizes(128, 1024, sizes) == _lib.BZ_OK and list(sizes) == [512, 128 * 19, 256, 3 * 128 + 201, 128, 65 * 128 + 64 * 64]   # (heads: ... | loss, CE, MSE | the error word)
    assert L.bz_train_wgrad_bias_rows(128, 10) == 80 and L.bz_train_wgrad_bias_rows(96, 10) == 0
    assert L.bz_train_wf_bytes(96, 4) == -1 and L.bz_train_wf_bytes(128, 3) == -1      # width; odd number of conv layers
    assert L.bz_train_stem_fwd(None, 64, None, None, 64, None, None) == _lib.BZ_EINVAL
    assert L.bz_train_heads(None, None, 64, 64, 64, None, None, None, None, None, None, None) == _lib.BZ_EINVAL
    assert L.bz_train_heads_wgrad(None, None, None, 64, 65, None, None) == _lib.BZ_EINVAL and b"value_hidden" in L.bz_last_error()
    assert L.bz_train_finish(None, None, 64, 4, 64, 64, None, None, None) == _lib.BZ_EINVAL
    assert L.bz_train_pack_weights(None, 64, 4, None, None, None) == _lib.BZ_EINVAL


@pytest.mark.skipif(_lib.lib().bz_device_count() > 0, reason="CPU-only check")
def test_product_path_fails_loudly_without_gpu():
    from betazero_amd.engine import SelfPlayEngine
    with pytest.raises(RuntimeError, match="no HIP device"):
        SelfPlayEngine("ttt", 4, 8)
    with pytest.raises(RuntimeError, match="no HIP device"):
        import betazero_amd as bz
        bz.MCTSPlayer(1, sims=8).get_move(bz.TicTacToeBoard())


def test_engine_config_limits_are_refused_with_a_reason():
    """the tree's packed edge record holds node ids in 13 bits: sims <= BZ_ENGINE_MAX_SIMS (8189), <= 2045 with subtree
    reuse; the tic-tac-toe lane knob takes -1, 0, 1, 2, 4, 8.  Out-of-range values come back as -1 / BZ_EINVAL with a
    message that names the limit -- not as a silently corrupted tree."""
    L = _lib.lib()

    def ws(sims, flags=0, lanes=0, game=1):
        cfg = _lib.EngineCfg(game, 4, sims, 0, 1.5, 0, 0, 1, 64, 0, 0, 0, 4, flags, 0.0, 0.0, lanes)
        return L.bz_engine_workspace_bytes(C.byref(cfg))
    hdr = open(os.path.join(ROOT, "include", "bz_abi.h")).read()
    assert "#define BZ_ENGINE_MAX_SIMS 8189" in hdr and "#define BZ_ENGINE_MAX_SIMS_REUSE 2045" in hdr
    assert ws(8189) > 0 and ws(8190) == -1 and b"8189" in L.bz_last_error()
    assert ws(2045, _lib.ENGINE_REUSE_SUBTREE) > 0 and ws(2046, _lib.ENGINE_REUSE_SUBTREE) == -1
    for lanes in (-1, 0, 1, 2, 4, 8):
        assert ws(50, lanes=lanes, game=0) > 0
    for lanes in (3, 16, -2):
        assert ws(50, lanes=lanes, game=0) == -1
    assert ws(0) == -1 and ws(800) > 0
    # the workspace grows with sims as the layout says: nodes 32 B + 34 edges of 16 B per node (+ fixed per-game arrays)
    def ws64(sims):  # 64 games: every array is then a multiple of the 256-byte carving granule
        cfg = _lib.EngineCfg(1, 64, sims, 0, 1.5, 0, 0, 1, 64, 0, 0, 0, 64, 0, 0.0, 0.0, 0)
        return L.bz_engine_workspace_bytes(C.byref(cfg))
    assert (ws64(801) - ws64(800)) / 64 == 32 + 34 * 16
