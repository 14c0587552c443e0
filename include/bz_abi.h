/*
 * bz_abi.h -- C ABI of libbz_hip.so, the MI355X (gfx950) self-play engine that
 * drops in behind BetaZero's Python Game / Player API.
 *
 * The reference (whaiproject/BetaZero) is pure Python and has NO FFI: the seams
 * this ABI sits behind are duck-typed Python interfaces.  Each entry point
 * cites the reference interface it replaces (paths relative to the reference
 * root).  INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *  - extern "C", plain C types only; every function returns int32_t status
 *    (BZ_OK or a BZ_E* code) and never throws / aborts; bz_last_error() gives
 *    the thread-local message of the last failure.
 *  - Batched entry points take RAW DEVICE POINTERS (torch.Tensor.data_ptr()),
 *    element counts and a hipStream_t passed as void* (0 = default stream).
 *    They are asynchronous on that stream.  The library never allocates or
 *    frees device memory the caller sees: the caller passes a workspace sized
 *    by the matching *_workspace_bytes() query.
 *  - Bitboards: Reversi bit = 8*row+col for every board size (the rule entry
 *    points -- bz_reversi_legal / _apply / _game_over / _step_batch_sized --
 *    take size 1..8, every size of the reference's generic constructor,
 *    reversi_board.py:4-14, whose cells fit 64 bits; the engine's games and the
 *    arena are 4, 6 and 8); Tic-tac-toe bit = 3*row+col.  "own" = stones of the side to move,
 *    "opp" = the other side (the side-to-move canonical form of
 *    src/tic_tac_toe/SL/generate_training_games.py:17-18).
 *  - Action index = size*row+col as in the reference's CSV flattening
 *    (generate_training_games.py:42-43); on 8x8 that equals the bit index.
 *    Action 64 = "pass" exists only inside the search tree.
 */
#ifndef BZ_ABI_H
#define BZ_ABI_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BZ_ABI_VERSION 6

enum { BZ_OK = 0, BZ_EINVAL = 1, BZ_EILLEGAL_MOVE = 2, BZ_EHIP = 3, BZ_ENOMEM = 4, BZ_ENOGPU = 5,
       BZ_ESTATE = 6 };
/* REVERSI = 8x8 (the benchmark game); REVERSI6 / REVERSI4 = the reference's 6x6 and 4x4 demo boards (same
 * bit = 8*row+col, same 65 actions; the conv net serves them through the top-left corner of its 8x8 planes) */
enum { BZ_GAME_TTT = 0, BZ_GAME_REVERSI = 1, BZ_GAME_REVERSI6 = 2, BZ_GAME_REVERSI4 = 3 };
/* leaf evaluators: uniform priors + v=0 (BASELINE cfg 2), synthetic hash (P,v)
 * (tree-kernel parity runs), the conv net in exact-fp32 parity mode, the conv
 * net on bf16 MFMA (the product path), or caller-filled logits/value. */
enum { BZ_EVAL_UNIFORM = 0, BZ_EVAL_HASH = 1, BZ_EVAL_NET_F32 = 2, BZ_EVAL_NET_BF16 = 3,
       BZ_EVAL_EXTERNAL = 4, BZ_EVAL_NET_FP8 = 5 };
#define BZ_PASS_ACTION 64

int32_t bz_abi_version(void);
const char* bz_last_error(void);
/* the -D flags this library was compiled with ("product" for the shipped build; diagnostic
 * variants carry BZ_EXPERIMENT and are refused by the Python binding unless asked for) */
const char* bz_build_info(void);
/* number of visible HIP devices (0 on a CPU-only host); never fails */
int32_t bz_device_count(void);

/* ------------------------------------------------------------------------ */
/* Scalar rules (host side of the same __host__ __device__ rule functions    */
/* the kernels use).  Back the API-compatible single-board classes.          */
/* ------------------------------------------------------------------------ */
/* ReversiBoard.generate_possible_moves / is_valid_move
 *   src/reversi/game_logic/reversi_board.py:25-41, 87-88.  size 1..8.  Cells that hold neither
 *   side's stones but are not empty (the reference stores any `player` value and treats every
 *   non-zero cell as occupied, :26) are walls: leave them out of own/opp and clear them from
 *   *legal (rays stop at any cell that is not the opponent's, so nothing else changes). */
int32_t bz_reversi_legal(uint64_t own, uint64_t opp, int32_t size, uint64_t* legal);
/* ReversiBoard.make_move  reversi_board.py:43-59.  BZ_EILLEGAL_MOVE where the
 * reference raises ValueError("Invalid move").  Outputs are NOT swapped:
 * own_after = mover's stones after the move. */
int32_t bz_reversi_apply(uint64_t own, uint64_t opp, int32_t size, int32_t row, int32_t col,
                         uint64_t* own_after, uint64_t* opp_after, uint64_t* flips);
/* ReversiBoard.is_game_over  reversi_board.py:61-65 */
int32_t bz_reversi_game_over(uint64_t a, uint64_t b, int32_t size, int32_t* over);
/* ReversiBoard.get_score  reversi_board.py:67-85 (x = +1 stones, o = -1 stones) */
int32_t bz_reversi_score(uint64_t x, uint64_t o, int32_t* winner, int32_t* n_x, int32_t* n_o);
/* TicTacToeBoard.generate_possible_moves  src/tic_tac_toe/tic_tac_toe_board.py:42-43 */
int32_t bz_ttt_legal(uint32_t x, uint32_t o, uint32_t* legal);
/* TicTacToeBoard.make_move  tic_tac_toe_board.py:20-29 */
int32_t bz_ttt_apply(uint32_t own, uint32_t opp, int32_t row, int32_t col, uint32_t* own_after);
/* TicTacToeBoard.is_game_over  tic_tac_toe_board.py:31-40 (+1 tested before -1);
 * *winner is 1/-1/0 and meaningful only when *over != 0 */
int32_t bz_ttt_game_over(uint32_t x, uint32_t o, int32_t* over, int32_t* winner);

/* ------------------------------------------------------------------------ */
/* Batched board-env step (device).  One game per lane.                      */
/* Replaces one iteration of the turn loop in                                */
/*   ReversiTerminal.play  reversi_terminal.py:19-35                         */
/*   TicTacToeHeadless.play  src/tic_tac_toe/tic_tac_toe.py:16-27            */
/* for n games at once.                                                      */
/* ------------------------------------------------------------------------ */
/* status codes written by the step kernels */
enum { BZ_ST_RUNNING = 0, BZ_ST_TERMINAL = 1, BZ_ST_ILLEGAL = 2, BZ_ST_MUST_PASS = 3 };
/* in : own/opp of the mover, action[n] (0..63, or 64 = pass)
 * out: position seen by the NEXT mover (own_next = next mover's stones),
 *      legal_next = next mover's legal mask, status (above; on ILLEGAL the
 *      outputs repeat the inputs), winner = +1/-1/0 for the player who just
 *      moved (valid when TERMINAL).  BZ_ST_MUST_PASS: game not over but the
 *      next mover has no move (the driver flips the side, reversi_terminal.py:32). */
int32_t bz_reversi_step_batch(const uint64_t* own, const uint64_t* opp, const uint8_t* action, int64_t n,
                              uint64_t* own_next, uint64_t* opp_next, uint64_t* legal_next,
                              uint8_t* status, int8_t* winner, void* stream);
/* the same for the reference's smaller boards (size 1..8; bit = 8*row+col; actions are bit
 * indices, 64 = pass) */
int32_t bz_reversi_step_batch_sized(const uint64_t* own, const uint64_t* opp, const uint8_t* action, int64_t n,
                                    int32_t size, uint64_t* own_next, uint64_t* opp_next, uint64_t* legal_next,
                                    uint8_t* status, int8_t* winner, void* stream);
int32_t bz_reversi_legal_batch(const uint64_t* own, const uint64_t* opp, int64_t n, uint64_t* legal,
                               void* stream);
/* ReversiBoard.get_score  reversi_board.py:67-85 for n boards: x = +1 stones, o = -1 stones;
 * winner[n] = +1/-1/0, counts[n][2] = (n_x, n_o) */
int32_t bz_reversi_score_batch(const uint64_t* x, const uint64_t* o, int64_t n, int8_t* winner,
                               uint8_t* counts, void* stream);
/* Tic-tac-toe: to_move[n] = absolute colour (+1/-1) of the mover; winner is the
 * ABSOLUTE colour (+1/-1/0) exactly as TicTacToeBoard.is_game_over returns it. */
int32_t bz_ttt_step_batch(const uint16_t* own, const uint16_t* opp, const uint8_t* action,
                          const int8_t* to_move, int64_t n, uint16_t* own_next, uint16_t* opp_next,
                          uint16_t* legal_next, uint8_t* status, int8_t* winner, void* stream);

/* 8-fold symmetry augmentation of (s, pi) rows on the device -- the transforms of
 * TicTacToeDataset.expand_with_transforms, src/tic_tac_toe/SL/train.py:27-36, in the
 * reference's order (id, flip rows, flip cols, rot90 x1/x2/x3, transpose, flip-rows-then-
 * transpose; the last equals rot90 x3 under torch semantics, as in the reference).
 * Row 8*i+t of the outputs = transform t of input row i; pi is permuted with the board
 * (entries beyond size*size -- Reversi's pass -- stay).  key8 (optional, may be null) gets a
 * 64-bit content key of each output row for the dedupe of train.py:45-50. */
int32_t bz_augment_d4_batch(const uint64_t* own, const uint64_t* opp, const float* pi, int64_t n, int32_t size,
                            int32_t na, uint64_t* own8, uint64_t* opp8, float* pi8, uint64_t* key8, void* stream);

/* ------------------------------------------------------------------------ */
/* The reference's minimax players (the arena's yard-stick, SURVEY.md 8(f)   */
/* row 3), one game per lane, and as scalar host entry points.               */
/* ------------------------------------------------------------------------ */
/* OptimalPlayer.minimax  src/reversi/players/reversi_players.py:41-69 with evaluate_board :71-77:
 * depth-limited (any max_depth >= 0 for the scalar entry point; 0 <= max_depth <= 8 for the batched kernel, whose
 * stack lives in a lane's registers / scratch), stone difference for the player, moves in
 * generate_possible_moves order, first strictly better score wins, no pass rule inside the
 * search (a side without a move in an unfinished game scores -inf / +inf: the reference's
 * behaviour).  self = the player's stones, other = the opponent's; the player is to move.
 * *move = bit index 8*row+col, or -1 where the reference's best_move is None (get_move then
 * draws random.choice, :38-39 -- left to the caller); *score = the root value (+-1000 = +-inf). */
int32_t bz_reversi_minimax(uint64_t self, uint64_t other, int32_t size, int32_t max_depth, int32_t* move,
                           int32_t* score);
/* OptimalPlayer.minimax  src/tic_tac_toe/players.py:41-70 (full depth; +1/0/-1 for the player).
 * symbol = the player's colour (+1 = X), needed only for is_game_over's X-before-O test order.
 * *move = 3*row+col, -1 = None (finished board), -2 = empty board (the reference plays a random
 * opening move there, :35-36 -- left to the caller). */
int32_t bz_ttt_minimax(uint32_t self, uint32_t other, int32_t symbol, int32_t* move, int32_t* score);
/* batched: device arrays [n]; active (optional, may be null): 0 = skip the game (move -1) */
int32_t bz_reversi_minimax_batch(const uint64_t* self, const uint64_t* other, const uint8_t* active, int64_t n,
                                 int32_t size, int32_t max_depth, int8_t* move, int16_t* score, void* stream);
int32_t bz_ttt_minimax_batch(const uint16_t* self, const uint16_t* other, const int8_t* symbol,
                             const uint8_t* active, int64_t n, int8_t* move, int16_t* score, void* stream);

/* ------------------------------------------------------------------------ */
/* Policy/value net (build-authored architecture, SURVEY.md 8(d) "net";      */
/* the calling convention generalises AIPlayer.get_move players.py:84-98:    */
/* side-to-move canonical input, logits out, legality masked by the caller)  */
/* ------------------------------------------------------------------------ */
typedef struct bz_net bz_net;
/* flat fp32 parameter vector, torch layouts, in this order:
 *  stem.w[C][2][3][3] stem.b[C] { c1.w[C][C][3][3] c1.b[C] c2.w[C][C][3][3] c2.b[C] } x NB
 *  pol.w[2][C] pol.b[2] polfc.w[65][128] polfc.b[65]
 *  val.w[1][C] val.b[1] v1.w[VH][64] v1.b[VH] v2.w[1][VH] v2.b[1] */
int64_t bz_net_param_count(int32_t C, int32_t NB, int32_t VH);
int64_t bz_net_workspace_bytes(int32_t C, int32_t NB, int32_t VH, int32_t max_batch);
/* params_host: host pointer.  Repacks + uploads weights into the workspace
 * (synchronises the stream once). */
int32_t bz_net_create(int32_t C, int32_t NB, int32_t VH, int32_t max_batch, const float* params_host,
                      void* workspace, int64_t workspace_bytes, void* stream, bz_net** out);
int32_t bz_net_destroy(bz_net* net);
/* replace the weights of an existing net (same shape) -- after a training step */
int32_t bz_net_update(bz_net* net, const float* params_host, void* stream);
/* own/opp: device u64[n]; logits: device f32[n][65]; value: device f32[n].
 * _f32: exact parity mode (k-ordered fmaf chains == oracle bit for bit).
 * _bf16: MFMA path (bf16 activations/weights, fp32 accumulate); C = 64, 128 or 256
 *        (C = 32 nets run on the _f32 path only). */
int32_t bz_net_forward_f32(bz_net* net, const uint64_t* own, const uint64_t* opp, int32_t n,
                           float* logits, float* value, void* stream);
int32_t bz_net_forward_bf16(bz_net* net, const uint64_t* own, const uint64_t* opp, int32_t n,
                            float* logits, float* value, void* stream);
/* _fp8: e4m3 weights (per-output-channel power-of-two scale) and activations (x16) on the
 * MX-scaled 32x32x64 MFMA (BASELINE config 5); pass parameters fake-quantised by
 * betazero_amd/quant.py.  Needs C == 128. */
int32_t bz_net_forward_fp8(bz_net* net, const uint64_t* own, const uint64_t* opp, int32_t n,
                           float* logits, float* value, void* stream);

/* ------------------------------------------------------------------------ */
/* Batched MCTS self-play engine.  The plug-in point it fills is             */
/*   Player.get_move(board)  src/tic_tac_toe/players.py:6-9,                 */
/*   ReversiPlayer.get_move  src/reversi/players/reversi_players.py:5-8      */
/* (one search per call) and, batched, the game loops cited above plus the   */
/* example extraction of generate_training_games.py:12-38.                   */
/* ------------------------------------------------------------------------ */
typedef struct bz_engine bz_engine;
typedef struct bz_engine_cfg {
    int32_t game;        /* BZ_GAME_* */
    int32_t n_games;     /* concurrent game slots B on this GPU */
    int32_t sims;        /* simulations per move */
    int32_t eval_kind;   /* BZ_EVAL_* */
    float c_puct;        /* 1.5 in every BASELINE config */
    int32_t temp_moves;  /* moves_made < temp_moves -> sample ~ N (tau=1), else argmax N */
    int32_t openings;    /* Reversi: first 2 plies from the 12 fixed openings (game_id % 12) */
    int32_t rounds;      /* example-buffer depth: slot s plays games s, s+stride, ... (>=1) */
    int32_t t_max;       /* example rows per game (64 Reversi, 9 TTT) */
    int32_t stagger;     /* bench only: slot g starts its round-0 game pre-advanced by (g % stagger)
                          * pseudo-random plies so that completions are spread evenly (0 = off) */
    uint64_t seed;
    uint64_t game_id_base;   /* global id of slot 0, round 0 (= rank * n_games) */
    uint64_t game_id_stride; /* id distance between rounds (= world_size * n_games) */
    /* opt-in search features (all zero = the BASELINE configurations) */
    uint32_t flags;          /* BZ_ENGINE_* bits */
    float dirichlet_alpha;   /* 0 < alpha <= 1 when dirichlet_eps > 0 */
    float dirichlet_eps;     /* > 0: root priors P' = (1 - eps) P + eps Dirichlet(alpha), a fresh draw per
                              * search keyed by (seed, game id, moves made) -- DESIGN.md 3.9 */
    int32_t ttt_lanes;       /* tic-tac-toe fused search (synthetic evaluators, sims <= 120): lanes that serve one game --
                              * 0 = default (4), 1 / 2 / 4 / 8 = as given, -1 = the generic
                              * any-game fused kernel.  Results are identical for every setting. */
} bz_engine_cfg;
/* The tree's edge record packs (visits 14 bits | action | the child's edge count, terminal flag and value) and
 * (child id 13 bits | the child's first edge 19 bits) into two words, so a game's tree holds at most 8191 nodes:
 * sims <= 8189, or <= 2045 with BZ_ENGINE_REUSE_SUBTREE (the arena then holds 4 x (sims + 2) nodes).  Larger
 * values are refused with BZ_EINVAL. */
#define BZ_ENGINE_MAX_SIMS 8189
#define BZ_ENGINE_MAX_SIMS_REUSE 2045
/* keep the chosen child's subtree as the next search's tree; searches then go through the
 * step kernels for every evaluator */
#define BZ_ENGINE_REUSE_SUBTREE 1u
/* Evaluation cache (net evaluators; ignored with the synthetic / external evaluators and with BZ_ENGINE_REUSE_SUBTREE):
 * a leaf whose position was already evaluated earlier in the SAME search -- reached by another move order -- takes that
 * node's priors and value instead of an evaluator row.  The evaluator is a function of the position alone
 * (players.py:84-98: canonical planes in, logits out), so every result (visit counts, W, P, pi, moves) is bit for
 * bit what it is without the cache -- the tree is the same tree, only the repeated forward is not run.  The work counters
 * say how often: counters[8] = repeats served from the cache, counters[7] = rows the evaluator computed; their sum is
 * what counters[7] reads without the cache.  A table hit is confirmed against the stored node's position before use. */
#define BZ_ENGINE_EVAL_CACHE 2u
/* ... and across CONSECUTIVE searches of a slot (with BZ_ENGINE_EVAL_CACHE): the previous search's tree stays intact in a
 * second arena while the new one grows (the arenas alternate), and a leaf whose position that tree evaluated takes the
 * evaluation from there.  After a move the new root is a child of the old one, and a fresh search from it re-creates that
 * child's old subtree node for node (deterministic PUCT on the same evaluations), so about the played move's share of the
 * previous search's visits -- a quarter at cfg 3 -- never reaches the net again.  The new tree is still built from scratch
 * (this is NOT subtree reuse: no statistic is kept, DESIGN.md 3.10 stays an option of its own): results are bit for bit
 * those without any cache.  counters[9] = the part of counters[8] that came from the previous search.  Nothing is carried
 * over a change of weights: a search that follows bz_net_update / bz_engine_set_net starts with the in-search cache only. */
#define BZ_ENGINE_EVAL_CACHE_CARRY 4u

/* offsets (bytes, from the workspace base) of the caller-visible arrays */
typedef struct bz_engine_layout {
    int64_t ex_own, ex_opp;   /* u64 [rounds][B][t_max]  side-to-move canonical s */
    int64_t ex_pi;            /* f32 [rounds][B][t_max][NA]  pi = N/sum N            */
    int64_t ex_z;             /* i8  [rounds][B][t_max]  outcome for the mover       */
    int64_t ex_mover, ex_act; /* i8 / u8 [rounds][B][t_max]                          */
    int64_t ex_len;           /* i32 [rounds][B]  rows valid (-1 = game not finished) */
    int64_t ex_winner;        /* i8  [rounds][B]  absolute winner                    */
    int64_t root_N, root_W, root_P; /* u32/f32/f32 [B][NA] filled by bz_engine_root_stats */
    int64_t leaf_own, leaf_opp;     /* u64 [B]   positions awaiting evaluation       */
    int64_t leaf_kind;              /* u8  [B]   1 = needs (logits,value)             */
    int64_t logits, value;          /* f32 [B][NA], f32 [B]  evaluator outputs        */
    int64_t g_own, g_opp;           /* u64 [B] current positions                      */
    int64_t g_to_move, g_state;     /* i8 / u8 [B]  (state: 0 active, 1 finished)     */
    int64_t counters;               /* u64 [24] work counters: 0..8 used */
    int32_t na, t_max;
    /* The example arrays ex_own .. ex_winner are consecutive in the workspace and are followed by
     * a 256-byte header (ex_meta: u64 magic, game_id_base, game_id_stride, B, rounds, t_max, NA,
     * game, then the 8 array offsets relative to ex_begin).  [ex_begin, ex_begin + ex_bytes) is the
     * ONE contiguous, self-describing byte range that the iteration-end all-gather ships. */
    int64_t ex_begin, ex_bytes, ex_meta;
} bz_engine_layout;

int64_t bz_engine_workspace_bytes(const bz_engine_cfg* cfg);
int32_t bz_engine_create(const bz_engine_cfg* cfg, void* workspace, int64_t workspace_bytes,
                         bz_engine** out);
int32_t bz_engine_destroy(bz_engine* e);
int32_t bz_engine_get_layout(const bz_engine* e, bz_engine_layout* out);
int32_t bz_engine_set_net(bz_engine* e, bz_net* net);
/* test hook: set the count of searches begun so far (0 .. 2^19 - 3) -- the evaluation cache stamps its entries with it, cycling
 * through 1 .. 2^19 - 2; a test starts just below the wrap with this.  Nothing is carried over the jump. */
int32_t bz_engine_debug_set_search_seq(bz_engine* e, uint32_t seq);
/* start every slot at the game's start position (round 0) */
int32_t bz_engine_reset_games(bz_engine* e, void* stream);
/* load arbitrary root positions (MCTSPlayer.get_move, the arena, tests): device arrays [B];
 * to_move[g] = +1 / -1 (absolute colour of the mover) or 0 = leave slot g idle in this search */
int32_t bz_engine_set_roots(bz_engine* e, const uint64_t* own, const uint64_t* opp,
                            const int8_t* to_move, void* stream);
/* one full search (root expansion + cfg.sims simulations) for every active slot */
int32_t bz_engine_search(bz_engine* e, void* stream);
/* the search, step by step (BZ_EVAL_EXTERNAL callers fill logits/value between) */
int32_t bz_engine_root_begin(bz_engine* e, void* stream);   /* roots -> leaf buffers       */
/* sim_index = simulations already completed in this search (0, 1, 2, ...) */
int32_t bz_engine_select(bz_engine* e, uint32_t sim_index, void* stream); /* M2: PUCT walk + env step */
int32_t bz_engine_evaluate(bz_engine* e, void* stream);     /* run cfg.eval_kind on leaves */
int32_t bz_engine_expand_backup(bz_engine* e, void* stream);/* M3 + M4                     */
/* Dirichlet root noise (cfg.dirichlet_eps > 0; a no-op otherwise) on the priors of the expanded roots.  Step-API
 * callers run it once per search, after the expand_backup that follows root_begin and before select(0) -- the order
 * bz_engine_search uses. */
int32_t bz_engine_root_noise(bz_engine* e, void* stream);
/* copy root edge statistics into the root_N/W/P arrays, indexed by action */
int32_t bz_engine_root_stats(bz_engine* e, void* stream);
/* M5: pi, move choice, example row, env step, pass rule, terminal handling.
 * restart != 0: a finished slot starts its next game (next round) at once. */
int32_t bz_engine_play(bz_engine* e, int32_t restart, void* stream);
/* synchronises the stream; number of active slots / finished games so far.  error_flags (sticky until the next
 * reset_games / set_roots): 1 = a game's edge arena overflowed, 2 = a root position was already terminal, 4 = more example
 * rows than t_max, 8 = a walk deeper than the path buffer, 16 = the evaluator returned a non-finite logit or value (the
 * search results of that move are meaningless) */
#define BZ_ENGINE_ERR_EDGE_OVERFLOW 1
#define BZ_ENGINE_ERR_TERMINAL_ROOT 2
#define BZ_ENGINE_ERR_EXAMPLE_OVERFLOW 4
#define BZ_ENGINE_ERR_DEPTH 8
#define BZ_ENGINE_ERR_EVAL_NONFINITE 16
int32_t bz_engine_status(bz_engine* e, void* stream, int32_t* n_active, int64_t* games_finished,
                         int32_t* error_flags);
/* ------------------------------------------------------------------------ */
/* Packed examples: the rows of the FINISHED games only, compacted on the    */
/* device in (round, slot, ply) order -- the batched form of what            */
/* collect_game_data keeps, src/tic_tac_toe/SL/generate_training_games.py:   */
/* 30-36 (only complete games reach all_states / all_actions).  This is what */
/* the iteration-end all-gather ships (one fixed-capacity buffer per rank).  */
/*                                                                           */
/* Block = 256-byte header, then eight arrays of cap_rows elements, each     */
/* starting at a multiple of 256 bytes:                                      */
/*   header u64[32]: magic 0x425A50414B000001, n_rows, n_games, cap_rows,    */
/*                   NA, game, dropped_rows, bytes, offs[8]                  */
/*   own u64, opp u64, pi f32[NA], game id i64, z i8, mover i8, act u8,      */
/*   ply u8                                                                  */
/* dropped_rows = rows of finished games that did not fit (0 in a healthy    */
/* run; ~0 = an append met a block of another geometry); rows [0, n_rows)    */
/* of every array are valid.                                                 */
/* ------------------------------------------------------------------------ */
#define BZ_PACKED_MAGIC 0x425A50414B000001ULL
int64_t bz_examples_packed_bytes(int32_t na, int64_t cap_rows);
/* packed: caller-owned device block of >= bz_examples_packed_bytes(NA, cap_rows) bytes, 256-byte aligned.
 * append == 0: start the block (header + this engine's rows); append != 0: add this engine's rows behind
 * those already there (the second pipeline of a rank).  Asynchronous on `stream`; calls that fill one block
 * must be ordered (same stream, or events). */
int32_t bz_engine_pack_examples(bz_engine* e, void* packed, int64_t packed_bytes, int64_t cap_rows, int32_t append,
                                void* stream);

/* aliases under the names SURVEY.md 8(b) lists: select / expand+backup of one simulation, and
 * one whole move (search + play) for every active slot */
int32_t bz_mcts_select(bz_engine* e, uint32_t sim_index, void* stream);
int32_t bz_mcts_expand_backup(bz_engine* e, void* stream);
int32_t bz_selfplay_run(bz_engine* e, int32_t restart, void* stream);
/* one move (search + play) for n engines -- the pipelines of one GPU, engine i on streams[i] -- issued by ONE host
 * thread and interleaved simulation by simulation.  run_ahead_sims > 0 bounds how far the thread runs ahead of the
 * streams (it then sleeps on blocking-sync events instead of spinning on a full queue; 0 = unbounded, the behaviour of
 * bz_selfplay_run called engine by engine).  Results are those of bz_selfplay_run on every engine.  All engines must
 * search the same number of simulations; n <= 16. */
int32_t bz_engines_step(bz_engine* const* engines, void* const* streams, int32_t n, int32_t restart,
                        int32_t run_ahead_sims);
int32_t bz_engine_reset_counters(bz_engine* e, void* stream);
/* fold the kernels' per-wave counter slots into the layout's counters[16] array (async) */
int32_t bz_engine_sum_counters(bz_engine* e, void* stream);

/* ------------------------------------------------------------------------ */
/* Training step of the residual tower: hand-written  */
/* bf16 MFMA kernels for forward-with-saved-activations, backward-data and   */
/* backward-weights of its n_layers = 2 NB conv3x3 layers.  The loop they    */
/* serve has the shape of src/tic_tac_toe/SL/train.py:85-136 (forward, loss, */
/* backward, Adam step); stem, heads, losses and the optimiser are the next   */
/* section's entry points (or the caller's own); the fp32 master weights stay */
/* with the caller.  C = 64 or 128.                                           */
/*                                                                           */
/* Tensors (device, bf16 unless noted), n = batch (a multiple of             */
/* bz_train_positions_per_workgroup(C)):                                     */
/*   act[a], a = 0..L : [n][64 cells][C]  act[0] = the stem's output (after   */
/*                      its ReLU), act[a] = output of conv layer a - 1        */
/*   g[a],   a = 0..L : [n][64][C]  g[a] = d loss / d (pre-activation of      */
/*                      act[a]) for a >= 1; g[0] = d loss / d act[0]          */
/*   W (fp32)         : [L][C co][C ci][3][3]  torch Conv2d layout, stacked   */
/*   wf_fwd / wf_bwd  : bz_train_wf_bytes() each: the kernels' fragment-major */
/*                      weight streams (bz_train_pack_weights writes both)    */
/*   masks            : bz_train_mask_bytes(): the ReLU pattern of act[1..L]  */
/* ------------------------------------------------------------------------ */
int64_t bz_train_wf_bytes(int32_t C, int32_t n_layers);
int32_t bz_train_positions_per_workgroup(int32_t C);
int64_t bz_train_mask_bytes(int32_t C, int32_t n_layers, int32_t n);
int32_t bz_train_pack_weights(const float* W, int32_t C, int32_t n_layers, void* wf_fwd, void* wf_bwd, void* stream);
/* act0 = act[0]; acts_out = act[1..L] as [L][n][64][C]; bias fp32 [L][C] */
int32_t bz_train_tower_fwd(const void* act0, const void* wf_fwd, const float* bias, int32_t C, int32_t n_layers, int32_t n,
                           void* acts_out, void* masks, void* stream);
/* g_top = g[L]; gs_out = g[0..L-1] as [L][n][64][C]; zeros_c = C fp32 zeros */
int32_t bz_train_tower_bwd(const void* g_top, const void* wf_bwd, const float* zeros_c, const void* masks, int32_t C,
                           int32_t n_layers, int32_t n, void* gs_out, void* stream);
/* weight gradients: acts = act[0..L-1], gs = g[1..L], both [L][n][64][C].  partial (fp32) =
 * [L][splits][9 taps][C ci][C co] and db_partial (fp32) = [L][bz_train_wgrad_bias_rows(C, splits)][C]: the caller sums over
 * the second axis of both (bz_train_wgrad_splits()) and permutes the former to W's layout; the latter is the bias gradient
 * (sum of g[l + 1] over positions and cells).  (bz_train_finish does both.) */
int32_t bz_train_wgrad_splits(int32_t C, int32_t n_layers, int32_t n);
int32_t bz_train_wgrad_bias_rows(int32_t C, int32_t splits);
int32_t bz_train_wgrad(const void* acts, const void* gs, int32_t C, int32_t n_layers, int32_t n, int32_t splits, float* partial,
                       float* db_partial, void* stream);

/* ------------------------------------------------------------------------ */
/* The two ends of the same step (csrc/bz_train_ends.hip), so that a whole   */
/* step -- train.py:85-136's forward, loss, backward -- is 9 launches        */
/* (10 with the optimiser):                                                   */
/*   stem_fwd, pack_weights, tower_fwd, heads, tower_bwd, wgrad, stem_wgrad, */
/*   heads_wgrad, finish (+ the Adam update as a tenth launch).  The net is     */
/* betazero_amd/net.py's (SURVEY 8(d) "net"): every pointer below is one of   */
/* its parameter tensors (fp32, torch layout) or the gradient of one.         */
/* n = batch, a multiple of 4 (and of bz_train_positions_per_workgroup(C)).   */
/* ------------------------------------------------------------------------ */
typedef struct bz_train_head_params {   /* pol: Conv2d(C, 2, 1); polfc: Linear(128, 65); val: Conv2d(C, 1, 1); v1: Linear(64, VH); v2: Linear(VH, 1) */
    const float *pol_w, *pol_b, *polfc_w, *polfc_b, *val_w, *val_b, *v1_w, *v1_b, *v2_w, *v2_b;
} bz_train_head_params;
typedef struct bz_train_tensors {       /* one fp32 device pointer per parameter tensor of the net: the gradients, the parameters, an Adam moment */
    float *stem_w, *stem_b, *tower_w, *tower_b, *pol_w, *pol_b, *polfc_w, *polfc_b, *val_w, *val_b, *v1_w, *v1_b, *v2_w, *v2_b;
} bz_train_tensors;
typedef struct bz_train_partials {      /* the partial sums the kernels of one step leave behind */
    const float *tower, *tower_b;       /* bz_train_wgrad's partial / db_partial */
    const float *stem, *heads, *heads_w; /* bz_train_stem_wgrad's, bz_train_heads', bz_train_heads_wgrad's (bz_train_ends_sizes) */
    int32_t splits;                     /* bz_train_wgrad's splits */
} bz_train_partials;
/* The batch of a step.  This struct lives in DEVICE memory and is read by the kernels at launch time: a step captured
 * into a HIP graph keeps working when the data set's tensors are replaced or another batch is drawn -- the host rewrites
 * these 48 bytes (or just the idx array).  Batch position p is row idx[p] of the data set, or row p when idx is NULL.  An
 * index outside [0, n_rows) never becomes a fault -- the kernels read row 0 / n_rows - 1 in its place -- but it is an ERROR:
 * every such batch position is counted into the step's error word, losses[3] of bz_train_finish (the reference's
 * index_select-style gather, SL/train.py:102-108, raises on it).  The rows are what bz_engine_pack_examples / the example block hold: (s, pi, z) of
 * generate_training_games.py:12-23 in own/opp form. */
typedef struct bz_train_batch {
    const uint64_t *own, *opp;          /* [n_rows] bitboards, side-to-move canonical */
    const float* pi;                    /* [n_rows][65] */
    const int8_t* z;                    /* [n_rows] */
    const int64_t* idx;                 /* [n] or NULL */
    int64_t n_rows;
} bz_train_batch;
/* Adam behind bz_train_finish (torch.optim.Adam's arithmetic without weight decay / amsgrad: what train.py:87 constructs):
 * one more launch that updates all 14 parameter tensors from the gradients just written.  hyper = 16 bytes of DEVICE memory
 * {float learning rate; float steps done so far; float warm-up steps; float unused}: bz_train_finish advances the step
 * count to t = steps done + 1 and the update runs at rate lr * min(1, t / warm-up) (warm-up 0: lr), so a captured graph can
 * be replayed step after step with no host write in between; the caller writes the block to (re)start or change the rate. */
typedef struct bz_train_adam {
    float* hyper;
    float beta1, beta2, eps;
    bz_train_tensors p, m, v;           /* parameters (updated in place), first and second moments (zero before step 1) */
} bz_train_adam;
/* sizes[0..5] = number of partial vectors, floats per vector of: stem_wgrad, heads, heads_wgrad */
int32_t bz_train_ends_sizes(int32_t C, int32_t n, int32_t* sizes);
/* act0[pos][cell][c] (bf16) = relu(conv3x3(planes(own, opp)))[c][cell]: the bit planes never exist in memory */
int32_t bz_train_stem_fwd(const bz_train_batch* batch_dev, int32_t n, const float* stem_w, const float* stem_b, int32_t C, void* act0,
                          void* stream);
/* g0 = g[0] (d loss / d act[0], what bz_train_tower_bwd leaves there); partial [sizes[0]][sizes[1]] */
int32_t bz_train_stem_wgrad(const bz_train_batch* batch_dev, const void* act0, const void* g0, int32_t n, int32_t C, float* partial,
                            void* stream);
/* heads + losses, forward and backward: act_top = act[L]; loss = mean CE(pi, softmax(logits)) + mean (v - z)^2 over the
 * batch.  Writes g_top = g[L] (bf16 [n][64][C], the input of bz_train_tower_bwd), the per-position operands of the FC
 * weight gradients (hv fp32 [n][192], dl [n][65], dv1 [n][64]) and partial [sizes[2]][sizes[3]]. */
int32_t bz_train_heads(const void* act_top, const bz_train_batch* batch_dev, int32_t n, int32_t C, int32_t VH,
                       const bz_train_head_params* P, void* g_top, float* hv, float* dl, float* dv1, float* partial, void* stream);
/* partial [sizes[4]][sizes[5]] */
int32_t bz_train_heads_wgrad(const float* hv, const float* dl, const float* dv1, int32_t n, int32_t VH, float* partial, void* stream);
/* every partial sum -> the gradient tensors G (torch layouts); losses[4] = loss, policy CE, value MSE of the batch, and the
 * ERROR WORD: losses[3] += the number of batch positions of this step whose row index was out of range (bz_train_batch).
 * The error word accumulates over steps (a graph can be replayed many times before the host looks): the caller zeroes it,
 * and a value != 0 when the losses are read means some step trained on a wrong row.
 * opt != NULL: followed by the Adam update of every parameter (a second launch) */
int32_t bz_train_finish(const bz_train_partials* Q, const bz_train_tensors* G, int32_t C, int32_t n_layers, int32_t VH, int32_t n,
                        float* losses, const bz_train_adam* opt, void* stream);

/* ------------------------------------------------------------------------ */
/* In-library kernel timers: HIP events recorded on the launch stream around  */
/* each launch of the named kernel (off by default; bench.py turns them on).  */
/* ------------------------------------------------------------------------ */
enum { BZ_PROF_TOWER = 0, BZ_PROF_STEM = 1, BZ_PROF_HEADS = 2, BZ_PROF_SELECT = 3, BZ_PROF_EXPAND_BACKUP = 4,
       BZ_PROF_SEARCH_FUSED = 5, BZ_PROF_PLAY = 6, BZ_PROF_ENV_STEP = 7, BZ_PROF_N = 8 };
int32_t bz_profile_enable(int32_t on);
/* create the events for n_launches launches of a slot up front (keeps event creation out of a
 * timed loop); a slot records at most 262,144 launches between two resets */
int32_t bz_profile_reserve(int32_t slot, int64_t n_launches);
/* synchronises the device; launches = all launches seen, timed = launches that
 * carried events (capped), total_ms = sum of their durations */
int32_t bz_profile_read(int32_t slot, int64_t* launches, int64_t* timed, double* total_ms);
int32_t bz_profile_reset(void);
/* start/end (ms, relative to the slot's first event) of every timed launch; launches issued on
 * different streams may overlap in time */
int32_t bz_profile_intervals(int32_t slot, double* starts_ms, double* ends_ms, int64_t cap, int64_t* n);

/* Do two HIP streams run side by side?  Starts one single-wave kernel that waits spin_us microseconds on each of
 * them behind a common event and reports max(completion time) / spin_us, best of `reps`: ~1.0 = the streams
 * overlap, ~2.0 = the second waited for the first (e.g. they share a hardware queue).  The pipelined self-play
 * (two engines on two streams) picks its stream pair with it.  Synchronises both streams. */
int32_t bz_stream_overlap_probe(void* stream_a, void* stream_b, int32_t spin_us, int32_t reps, float* serial_ratio);

#ifdef __cplusplus
}
#endif
#endif /* BZ_ABI_H */
