// bz_rules.h -- board rules on bitboards, shared by host entry points and gfx950 kernels.
//
// Reversi: reversi_board.py:25-88 (is_valid_move / make_move / is_game_over /
// get_score / generate_possible_moves); bit = 8*row+col for sizes 4, 6 and 8
// (smaller boards sit in the top-left corner: cells outside are never stones,
// so rays stop there and the result is masked with valid(size)).
// Tic-tac-toe: tic_tac_toe_board.py:20-43; bit = 3*row+col.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BZ_HD __host__ __device__ __forceinline__

namespace bz {

typedef uint64_t u64;
typedef uint32_t u32;

constexpr u64 kInner = 0x7E7E7E7E7E7E7E7EULL;  // columns 1..6: stops east/west wrap
constexpr int kPass = 64;

BZ_HD int popc64(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}
BZ_HD int ctz64(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((unsigned long long)x) - 1;
#else
    return __builtin_ctzll(x);
#endif
}

BZ_HD u64 rev_valid(int size) {
    u64 row = (1ULL << size) - 1ULL;
    u64 m = 0;
    for (int r = 0; r < size; ++r) m |= row << (8 * r);
    return m;
}

// Bitboards as two 32-bit halves for the device code: 64-bit shifts by a constant 0 < s < 32 become two full-rate
// 32-bit operations (v_alignbit_b32 + a 32-bit shift) instead of the quarter-rate v_lshlrev_b64 / v_lshrrev_b64, on
// which the VALU-bound board kernels spent a quarter of their instructions (profiles/r02_pmc_env_*).  Keeping the
// halves apart through the whole parallel-prefix (rather than splitting per shift) stops the compiler from
// re-fusing them into 64-bit shifts.
struct B2 { u32 lo, hi; };
BZ_HD B2 b2(u64 x) { B2 r; r.lo = (u32)x; r.hi = (u32)(x >> 32); return r; }
BZ_HD u64 b2u(B2 x) { return ((u64)x.hi << 32) | (u64)x.lo; }
BZ_HD B2 operator&(B2 a, B2 b) { B2 r; r.lo = a.lo & b.lo; r.hi = a.hi & b.hi; return r; }
BZ_HD B2 operator|(B2 a, B2 b) { B2 r; r.lo = a.lo | b.lo; r.hi = a.hi | b.hi; return r; }
BZ_HD bool b2any(B2 a) { return (a.lo | a.hi) != 0u; }
template <int S> BZ_HD B2 shl(B2 x) {
    B2 r;
#if defined(__HIP_DEVICE_COMPILE__)
    r.hi = __builtin_amdgcn_alignbit(x.hi, x.lo, 32 - S);
#else
    r.hi = (x.hi << S) | (x.lo >> (32 - S));
#endif
    r.lo = x.lo << S;
    return r;
}
template <int S> BZ_HD B2 shr(B2 x) {
    B2 r;
#if defined(__HIP_DEVICE_COMPILE__)
    r.lo = __builtin_amdgcn_alignbit(x.hi, x.lo, S);
#else
    r.lo = (x.lo >> S) | (x.hi << (32 - S));
#endif
    r.hi = x.hi >> S;
    return r;
}

// one direction pair (shift S left / right), parallel-prefix over runs of <= 6
// opponent stones; o is opp pre-masked against wrap for this direction.
template <int S> BZ_HD B2 rev_moves_dir(B2 own, B2 o) {
    B2 fl = o & shl<S>(own), fr = o & shr<S>(own);
    fl = fl | (o & shl<S>(fl));      fr = fr | (o & shr<S>(fr));
    B2 pl = o & shl<S>(o),           pr = o & shr<S>(o);
    fl = fl | (pl & shl<2 * S>(fl)); fr = fr | (pr & shr<2 * S>(fr));
    fl = fl | (pl & shl<2 * S>(fl)); fr = fr | (pr & shr<2 * S>(fr));
    return shl<S>(fl) | shr<S>(fr);
}

// generate_possible_moves(player) as a mask; own = player's stones
BZ_HD u64 rev_legal(u64 own, u64 opp, u64 valid) {
    const B2 w = b2(own), o = b2(opp), oh = b2(opp & kInner);
    B2 m = rev_moves_dir<1>(w, oh) | rev_moves_dir<8>(w, o) | rev_moves_dir<7>(w, oh) | rev_moves_dir<9>(w, oh);
    return b2u(m) & ~(own | opp) & valid;
}
BZ_HD u64 rev_legal8(u64 own, u64 opp) { return rev_legal(own, opp, ~0ULL); }

// stones flipped by placing on bit m (m must be a legal cell); the 8 rays of
// make_move (reversi_board.py:49-58)
template <int S> BZ_HD B2 rev_flips_dir(B2 own, B2 o, B2 m) {
    // runs of <= 6 opponent stones next to m, parallel-prefix (2 + 2 + 2 cells)
    B2 fl = o & shl<S>(m), fr = o & shr<S>(m);
    fl = fl | (o & shl<S>(fl));      fr = fr | (o & shr<S>(fr));
    B2 pl = o & shl<S>(o),           pr = o & shr<S>(o);
    fl = fl | (pl & shl<2 * S>(fl)); fr = fr | (pr & shr<2 * S>(fr));
    fl = fl | (pl & shl<2 * S>(fl)); fr = fr | (pr & shr<2 * S>(fr));
    B2 out; out.lo = 0; out.hi = 0;
    if (b2any(shl<S>(fl) & own)) out = out | fl;
    if (b2any(shr<S>(fr) & own)) out = out | fr;
    return out;
}
BZ_HD u64 rev_flips(u64 own, u64 opp, u64 m) {
    const B2 w = b2(own), o = b2(opp), oh = b2(opp & kInner), mm = b2(m);
    return b2u(rev_flips_dir<1>(w, oh, mm) | rev_flips_dir<8>(w, o, mm) | rev_flips_dir<7>(w, oh, mm) |
               rev_flips_dir<9>(w, oh, mm));
}

// ---- tic-tac-toe
BZ_HD bool ttt_line(u32 s) {
    return ((s & 0x007) == 0x007) | ((s & 0x038) == 0x038) | ((s & 0x1C0) == 0x1C0) | ((s & 0x049) == 0x049) |
           ((s & 0x092) == 0x092) | ((s & 0x124) == 0x124) | ((s & 0x111) == 0x111) | ((s & 0x054) == 0x054);
}
// is_game_over: +1 first, then -1, then full board (tic_tac_toe_board.py:31-40)
BZ_HD bool ttt_over(u32 x, u32 o, int* winner) {
    if (ttt_line(x)) { *winner = 1; return true; }
    if (ttt_line(o)) { *winner = -1; return true; }
    *winner = 0;
    return ((x | o) & 0x1FF) == 0x1FF;
}

// ---- game traits used by the tree / self-play kernels (8x8 Reversi, 3x3 TTT)
// SIZE = 8 is the benchmark game; 6 and 4 are the reference's demo boards (reversi_gui.py:105,
// reversi_terminal.py:46, reversi_board.py:93).  All sizes share bit = 8*row+col and NA = 65.
template <int SIZE>
struct ReversiT {
    static constexpr int kGame = SIZE == 8 ? 1 : (SIZE == 6 ? 2 : 3), NA = 65, MAXCH = 34, MAXD = 128;
    static constexpr int GW = 16;  // lanes that serve one game in the tree kernels (mean branching 8.5, max 33)
    static constexpr u64 kValid = SIZE == 8 ? ~0ULL : (SIZE == 6 ? 0x00003F3F3F3F3F3FULL : 0x000000000F0F0F0FULL);
    static BZ_HD u64 legal(u64 own, u64 opp) { return rev_legal(own, opp, kValid); }
    // position after action a, seen by the next mover
    static BZ_HD void apply(u64 own, u64 opp, int a, u64* cown, u64* copp) {
        if (a == kPass) { *cown = opp; *copp = own; return; }
        u64 m = 1ULL << a, f = rev_flips(own, opp, m);
        *cown = opp & ~f;
        *copp = own | m | f;
    }
    // terminal test for a node whose mover has absolute colour to_move;
    // *tv = outcome for that mover
    static BZ_HD bool terminal(u64 own, u64 opp, int to_move, u64 legal_own, int* tv) {
        if (legal_own != 0 || rev_legal(opp, own, kValid) != 0) return false;
        int d = popc64(own) - popc64(opp);  // get_score, reversi_board.py:68-76
        *tv = d > 0 ? 1 : (d < 0 ? -1 : 0);
        return true;
    }
    static BZ_HD void start(u64* own, u64* opp) {  // reversi_board.py:9-11, +1 moves first
        constexpr int p = SIZE / 2 - 1;
        *own = (1ULL << (8 * p + p)) | (1ULL << (8 * (p + 1) + p + 1));
        *opp = (1ULL << (8 * p + p + 1)) | (1ULL << (8 * (p + 1) + p));
    }
};
typedef ReversiT<8> Reversi;
typedef ReversiT<6> Reversi6;
typedef ReversiT<4> Reversi4;
struct TicTacToe {
    static constexpr int kGame = 0, NA = 9, MAXCH = 9, MAXD = 16;
    static constexpr int GW = 4;   // 16 games per wave
    static BZ_HD u64 legal(u64 own, u64 opp) { return ~(own | opp) & 0x1FFULL; }
    static BZ_HD void apply(u64 own, u64 opp, int a, u64* cown, u64* copp) {
        *cown = opp;
        *copp = own | (1ULL << a);
    }
    static BZ_HD bool terminal(u64 own, u64 opp, int to_move, u64, int* tv) {
        u32 x = to_move == 1 ? (u32)own : (u32)opp, o = to_move == 1 ? (u32)opp : (u32)own;
        int w;
        if (!ttt_over(x, o, &w)) return false;
        *tv = w * to_move;
        return true;
    }
    static BZ_HD void start(u64* own, u64* opp) { *own = 0; *opp = 0; }
};

}  // namespace bz
