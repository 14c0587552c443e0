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

// three-input bit logic on 64-bit boards.  gfx950 has v_bitop3_b32 (any boolean function of three words in one
// instruction), but the compiler splits 64-bit and / or / andn into halves too late to form it (the env step's ISA showed
// v_and + v_or pairs throughout), so the device side asks for it per half; the host side is the plain expression.
#if defined(__HIP_DEVICE_COMPILE__)
template <unsigned TT> __device__ __forceinline__ u64 bitop3_64(u64 a, u64 b, u64 c) {
    const u32 lo = __builtin_amdgcn_bitop3_b32((u32)a, (u32)b, (u32)c, TT);
    const u32 hi = __builtin_amdgcn_bitop3_b32((u32)(a >> 32), (u32)(b >> 32), (u32)(c >> 32), TT);
    return ((u64)hi << 32) | lo;
}
BZ_HD u64 and_or(u64 a, u64 b, u64 c) { return bitop3_64<0xEA>(a, b, c); }     // (a & b) | c
BZ_HD u64 and3(u64 a, u64 b, u64 c) { return bitop3_64<0x80>(a, b, c); }       // a & b & c
BZ_HD u64 and_andn(u64 a, u64 b, u64 c) { return bitop3_64<0x40>(a, b, c); }   // a & b & ~c
BZ_HD u64 andn2(u64 a, u64 b, u64 c) { return bitop3_64<0x10>(a, b, c); }      // a & ~b & ~c
#else
BZ_HD u64 and_or(u64 a, u64 b, u64 c) { return (a & b) | c; }
BZ_HD u64 and3(u64 a, u64 b, u64 c) { return a & b & c; }
BZ_HD u64 and_andn(u64 a, u64 b, u64 c) { return a & b & ~c; }
BZ_HD u64 andn2(u64 a, u64 b, u64 c) { return a & ~b & ~c; }
#endif

BZ_HD u64 rev_valid(int size) {
    u64 row = (1ULL << size) - 1ULL;
    u64 m = 0;
    for (int r = 0; r < size; ++r) m |= row << (8 * r);
    return m;
}

// one direction pair (shift s left / right), parallel-prefix over runs of <= 6
// opponent stones; o is opp pre-masked against wrap for this direction.
BZ_HD u64 rev_moves_dir(u64 own, u64 o, int s) {
    u64 fl = o & (own << s), fr = o & (own >> s);
    fl = and_or(o, fl << s, fl);         fr = and_or(o, fr >> s, fr);
    u64 pl = o & (o << s),    pr = o & (o >> s);
    fl = and_or(pl, fl << (2 * s), fl);  fr = and_or(pr, fr >> (2 * s), fr);
    fl = and_or(pl, fl << (2 * s), fl);  fr = and_or(pr, fr >> (2 * s), fr);
    return (fl << s) | (fr >> s);
}

BZ_HD u64 brev64(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(x);
#else
    return __builtin_bitreverse64(x);
#endif
}

// the east / west pair by carry propagation (an addition carries in the +1 direction, which IS east): with oh = the
// opponent's stones in columns 1..6, x = the stones of oh whose west neighbour is the mover's, x + oh clears every run
// of opponent stones that starts at an x and sets the cell just past it -- the candidate (runs that start nowhere near
// the mover are left alone and masked off; a carry cannot leave its row because column 7 is never in oh).  West is east
// on the bit-reversed boards (bit i <-> 63 - i; the column mask is symmetric).  8 + 14 instructions against 2 x 24 for
// the parallel-prefix fill; the caller masks the result with the empty cells.
BZ_HD u64 rev_moves_ew(u64 own, u64 oh) {
    const u64 e = ((oh & (own << 1)) + oh) & ~oh;
    const u64 ro = brev64(oh), rp = brev64(own);
    const u64 w = ((ro & (rp << 1)) + ro) & ~ro;
    return e | brev64(w);
}

// generate_possible_moves(player) as a mask; own = player's stones
BZ_HD u64 rev_legal(u64 own, u64 opp, u64 valid) {
    u64 oh = opp & kInner;
    u64 m = rev_moves_ew(own, oh) | rev_moves_dir(own, opp, 8) | rev_moves_dir(own, oh, 7) |
            rev_moves_dir(own, oh, 9);
    return andn2(m, own, opp) & valid;
}
BZ_HD u64 rev_legal8(u64 own, u64 opp) { return rev_legal(own, opp, ~0ULL); }

// stones flipped by placing on bit m (m must be a legal cell); the 8 rays of
// make_move (reversi_board.py:49-58)
BZ_HD u64 rev_flips_dir(u64 own, u64 o, u64 m, int s) {
    // runs of <= 6 opponent stones next to m, parallel-prefix (2 + 2 + 2 cells)
    u64 fl = o & (m << s), fr = o & (m >> s);
    fl = and_or(o, fl << s, fl);         fr = and_or(o, fr >> s, fr);
    u64 pl = o & (o << s),    pr = o & (o >> s);
    fl = and_or(pl, fl << (2 * s), fl);  fr = and_or(pr, fr >> (2 * s), fr);
    fl = and_or(pl, fl << (2 * s), fl);  fr = and_or(pr, fr >> (2 * s), fr);
    u64 out = 0;
    if ((fl << s) & own) out |= fl;
    if ((fr >> s) & own) out |= fr;
    return out;
}
BZ_HD u64 rev_flips(u64 own, u64 opp, u64 m) {
    u64 oh = opp & kInner;
    return rev_flips_dir(own, oh, m, 1) | rev_flips_dir(own, opp, m, 8) | rev_flips_dir(own, oh, m, 7) |
           rev_flips_dir(own, oh, m, 9);
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
