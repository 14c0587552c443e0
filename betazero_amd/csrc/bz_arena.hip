// bz_arena.hip -- the reference's minimax players as batched gfx950 kernels (one game per lane) and as
// scalar host entry points of the same __host__ __device__ code: the strength yard-stick of the arena
// (SURVEY.md 8(f) row 3).
//
//  * Reversi: OptimalPlayer.minimax, src/reversi/players/reversi_players.py:41-69 -- depth-limited,
//    evaluation = stone difference for the player (evaluate_board :71-77), moves tried in
//    generate_possible_moves order (row-major = ascending bit), the FIRST move with a strictly better
//    score wins.  Its quirks are kept: there is no pass rule inside the search -- a node whose side to
//    move has no move but whose game is not over returns -inf (maximising) / +inf (minimising) with
//    move None; when every root move scores -inf the result is None and get_move falls back to
//    random.choice (:38-39) -- reported here as move = -1, the caller draws.
//  * Tic-tac-toe: OptimalPlayer.minimax, src/tic_tac_toe/players.py:41-70 -- full depth, scores +1/0/-1
//    for the player, first strictly better move in generate_possible_moves order (the random opening move
//    on the empty board, :35-36, is the caller's job: move = -2 is returned for an empty board).
//
// The recursion is an explicit stack (depth <= 8 Reversi / 9 TTT): lanes of a wave walk different trees,
// the loop structure is the same for all of them.
#include "bz_common.h"
#include "bz_rules.h"

using namespace bz;

namespace {

constexpr int kInf = 1000;       // stands for float("inf"): stone differences lie in [-64, 64]
constexpr int kMaxDepth = 8;        // the batched kernel's per-lane stack (the arena's opponent)
constexpr int kMaxDepthHost = 60;   // the scalar entry point: a game has at most 60 plies, so any depth the reference accepts

// `self` = stones of the minimax player, `other` = the opponent's.  Returns the root's best score;
// *best_move = bit index of the chosen move or -1 (None).  MAXD = the deepest max_depth this instance can hold.
template <int MAXD>
BZ_HD int rev_minimax(u64 self, u64 other, int max_depth, u64 valid, int* best_move) {
    u64 st_self[MAXD + 1], st_other[MAXD + 1], st_moves[MAXD + 1];
    int st_best[MAXD + 1], st_bm[MAXD + 1], st_cur[MAXD + 1];
    *best_move = -1;
    u64 ls = rev_legal(self, other, valid), lo = rev_legal(other, self, valid);
    if (max_depth <= 0 || (ls == 0 && lo == 0)) return popc64(self) - popc64(other);
    int d = 0;
    st_self[0] = self; st_other[0] = other; st_moves[0] = ls; st_best[0] = -kInf; st_bm[0] = -1; st_cur[0] = -1;
    for (;;) {
        const bool maxi = (d & 1) == 0;  // the player moves at even depth
        int score;
        bool have = false;
        if (st_moves[d] == 0) {          // node exhausted: hand its value to the parent
            if (d == 0) break;
            score = st_best[d];
            d--;
            have = true;
        } else {
            const u64 mv = st_moves[d] & (~st_moves[d] + 1);
            st_moves[d] &= st_moves[d] - 1;
            st_cur[d] = ctz64(mv);
            u64 cs = st_self[d], co = st_other[d];
            if (maxi) { u64 f = rev_flips(cs, co, mv); cs |= mv | f; co &= ~f; }
            else      { u64 f = rev_flips(co, cs, mv); co |= mv | f; cs &= ~f; }
            const u64 l_self = rev_legal(cs, co, valid), l_other = rev_legal(co, cs, valid);
            if (d + 1 >= max_depth || (l_self == 0 && l_other == 0)) {
                score = popc64(cs) - popc64(co);
                have = true;
            } else {
                d++;
                st_self[d] = cs; st_other[d] = co;
                st_moves[d] = (d & 1) == 0 ? l_self : l_other;
                st_best[d] = (d & 1) == 0 ? -kInf : kInf;
                st_bm[d] = -1; st_cur[d] = -1;
            }
        }
        if (have) {                      // `score` is the value of move st_cur[d] at node d
            const bool pmax = (d & 1) == 0;
            if (pmax ? score > st_best[d] : score < st_best[d]) { st_best[d] = score; st_bm[d] = st_cur[d]; }
        }
    }
    *best_move = st_bm[0];
    return st_best[0];
}

// is_game_over as the player scores it (players.py:42-49): the board tests the +1 (X) lines before the -1 (O)
// lines (tic_tac_toe_board.py:31-40), which only matters for unreachable positions holding both
BZ_HD bool ttt_result(u32 self, u32 other, bool self_is_x, int* score) {
    const bool ls = ttt_line(self), lo = ttt_line(other);
    if (ls && lo) { *score = self_is_x ? 1 : -1; return true; }
    if (ls) { *score = 1; return true; }
    if (lo) { *score = -1; return true; }
    *score = 0;
    return ((self | other) & 0x1FF) == 0x1FF;
}

// `self` / `other`: 9-bit stone masks of the minimax player and the opponent; the player is to move.
BZ_HD int ttt_minimax(u32 self, u32 other, bool self_is_x, int* best_move) {
    u32 st_self[10], st_other[10], st_moves[10];
    int st_best[10], st_bm[10], st_cur[10];
    *best_move = -1;
    int r0;
    if (ttt_result(self, other, self_is_x, &r0)) return r0;
    int d = 0;
    st_self[0] = self; st_other[0] = other; st_moves[0] = ~(self | other) & 0x1FFu; st_best[0] = -kInf; st_bm[0] = -1; st_cur[0] = -1;
    for (;;) {
        const bool maxi = (d & 1) == 0;
        int score;
        bool have = false;
        if (st_moves[d] == 0) {
            if (d == 0) break;
            score = st_best[d];
            d--;
            have = true;
        } else {
            const u32 mv = st_moves[d] & (~st_moves[d] + 1u);
            st_moves[d] &= st_moves[d] - 1u;
            st_cur[d] = ctz64((u64)mv);
            u32 cs = st_self[d], co = st_other[d];
            if (maxi) cs |= mv; else co |= mv;
            if (ttt_result(cs, co, self_is_x, &score)) have = true;
            else {
                d++;
                st_self[d] = cs; st_other[d] = co; st_moves[d] = ~(cs | co) & 0x1FFu;
                st_best[d] = (d & 1) == 0 ? -kInf : kInf;
                st_bm[d] = -1; st_cur[d] = -1;
            }
        }
        if (have) {
            const bool pmax = (d & 1) == 0;
            if (pmax ? score > st_best[d] : score < st_best[d]) { st_best[d] = score; st_bm[d] = st_cur[d]; }
        }
    }
    *best_move = st_bm[0];
    return st_best[0];
}

__global__ void __launch_bounds__(64) k_reversi_minimax(const u64* __restrict__ self, const u64* __restrict__ other,
                                                        const uint8_t* __restrict__ active, int64_t n, int depth, u64 valid,
                                                        int8_t* __restrict__ move, int16_t* __restrict__ score) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    int bm = -1, sc = 0;
    if (!active || active[i]) sc = rev_minimax<kMaxDepth>(self[i], other[i], depth, valid, &bm);
    move[i] = (int8_t)bm;
    score[i] = (int16_t)sc;
}

__global__ void __launch_bounds__(64) k_ttt_minimax(const uint16_t* __restrict__ self, const uint16_t* __restrict__ other,
                                                    const int8_t* __restrict__ symbol, const uint8_t* __restrict__ active, int64_t n,
                                                    int8_t* __restrict__ move, int16_t* __restrict__ score) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    int bm = -1, sc = 0;
    if (!active || active[i]) {
        u32 s = self[i] & 0x1FF, o = other[i] & 0x1FF;
        if ((s | o) == 0) bm = -2;  // empty board: the reference draws a random opening move
        else sc = ttt_minimax(s, o, symbol[i] == 1, &bm);
    }
    move[i] = (int8_t)bm;
    score[i] = (int16_t)sc;
}

bool size_ok(int32_t s) { return s == 4 || s == 6 || s == 8; }

}  // namespace

BZ_EXPORT int32_t bz_reversi_minimax(uint64_t self, uint64_t other, int32_t size, int32_t max_depth, int32_t* move,
                                     int32_t* score) {
    BZ_REQUIRE(size_ok(size) && max_depth >= 0 && move && score, "bz_reversi_minimax: size must be 4, 6 or 8 and max_depth >= 0");
    if (max_depth > kMaxDepthHost) max_depth = kMaxDepthHost;  // no line of play is longer: deeper limits search the same tree
    int bm;
    *score = rev_minimax<kMaxDepthHost>(self, other, max_depth, rev_valid(size), &bm);
    *move = bm;
    return BZ_OK;
}

BZ_EXPORT int32_t bz_ttt_minimax(uint32_t self, uint32_t other, int32_t symbol, int32_t* move, int32_t* score) {
    BZ_REQUIRE(move && score, "bz_ttt_minimax: null pointer");
    BZ_REQUIRE(((self & other) & 0x1FF) == 0, "bz_ttt_minimax: overlapping stones");
    int bm = -1;
    if (((self | other) & 0x1FF) == 0) { *move = -2; *score = 0; return BZ_OK; }
    *score = ttt_minimax(self & 0x1FF, other & 0x1FF, symbol == 1, &bm);
    *move = bm;
    return BZ_OK;
}

BZ_EXPORT int32_t bz_reversi_minimax_batch(const uint64_t* self, const uint64_t* other, const uint8_t* active, int64_t n,
                                           int32_t size, int32_t max_depth, int8_t* move, int16_t* score, void* stream) {
    BZ_REQUIRE(n >= 0 && self && other && move && score, "bz_reversi_minimax_batch: null pointer");
    BZ_REQUIRE(size_ok(size) && max_depth >= 0 && max_depth <= kMaxDepth,
               "bz_reversi_minimax_batch: size must be 4, 6 or 8 and 0 <= max_depth <= 8");
    if (n == 0) return BZ_OK;
    // 64-thread workgroups: every lane runs a whole search, so spread the games over as many CUs as possible
    hipLaunchKernelGGL(k_reversi_minimax, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, self, other,
                       active, n, (int)max_depth, rev_valid(size), move, score);
    BZ_LAUNCH_CHECK("k_reversi_minimax");
    return BZ_OK;
}

BZ_EXPORT int32_t bz_ttt_minimax_batch(const uint16_t* self, const uint16_t* other, const int8_t* symbol,
                                       const uint8_t* active, int64_t n, int8_t* move, int16_t* score, void* stream) {
    BZ_REQUIRE(n >= 0 && self && other && symbol && move && score, "bz_ttt_minimax_batch: null pointer");
    if (n == 0) return BZ_OK;
    hipLaunchKernelGGL(k_ttt_minimax, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, self, other, symbol,
                       active, n, move, score);
    BZ_LAUNCH_CHECK("k_ttt_minimax");
    return BZ_OK;
}
