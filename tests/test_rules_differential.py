"""Differential test of the product's host rule entry points (the host compilation of the
same __host__ __device__ functions the kernels run) against the CPU oracle on random,
mostly unreachable positions -- hypothesis-driven, CPU only."""
import ctypes as C

from hypothesis import given, settings, strategies as st

from betazero_amd import _lib
from oracle import oracle as orc

u64s = st.integers(min_value=0, max_value=2**64 - 1)


@settings(max_examples=3000, deadline=None)
@given(a=u64s, b=u64s, size=st.sampled_from([4, 6, 8]))
def test_reversi_legal_apply_game_over(a, b, size):
    valid = sum(1 << (8 * r + c) for r in range(size) for c in range(size))
    own, opp = a & valid & ~b, b & valid & ~a
    L = _lib.lib()
    out = C.c_uint64()
    assert L.bz_reversi_legal(own, opp, size, C.byref(out)) == 0
    legal = orc.reversi_legal(own, opp, size)
    assert out.value == legal
    over = C.c_int32()
    assert L.bz_reversi_game_over(own, opp, size, C.byref(over)) == 0
    assert bool(over.value) == orc.reversi_game_over(own, opp, size)
    m = legal
    k = 0
    while m and k < 4:  # a few legal moves + one illegal probe
        bit = (m & -m).bit_length() - 1
        m &= m - 1
        k += 1
        x, y, f = C.c_uint64(), C.c_uint64(), C.c_uint64()
        assert L.bz_reversi_apply(own, opp, size, bit >> 3, bit & 7, C.byref(x), C.byref(y), C.byref(f)) == 0
        assert (x.value, y.value, f.value) == orc.reversi_apply(own, opp, size, bit >> 3, bit & 7)
    for bit in (0, 9, 8 * (size - 1) + size - 1):
        if not legal >> bit & 1:
            x = C.c_uint64()
            assert L.bz_reversi_apply(own, opp, size, bit >> 3, bit & 7, C.byref(x), C.byref(x), None) == 2


@settings(max_examples=2000, deadline=None)
@given(x=st.integers(0, 511), o=st.integers(0, 511))
def test_ttt_rules(x, o):
    o &= ~x
    L = _lib.lib()
    lg, over, w = C.c_uint32(), C.c_int32(), C.c_int32()
    assert L.bz_ttt_legal(x, o, C.byref(lg)) == 0 and lg.value == orc.ttt_legal(x, o)
    assert L.bz_ttt_game_over(x, o, C.byref(over), C.byref(w)) == 0
    go, win = orc.ttt_game_over(x, o)
    assert bool(over.value) == go and (not go or w.value == win)
