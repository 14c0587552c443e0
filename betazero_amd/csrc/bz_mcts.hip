// bz_mcts.hip -- batched MCTS self-play engine for gfx950.
//
// G::GW lanes serve one game (Reversi 16, TTT 4).  Per game the tree is a bump-allocated AoS
// pair in HBM:
//   Node 32 B {own, opp, legal, edge0, info}     nodes[g][sims+2]
//   Edge 16 B {w0, W, P, w3}                     edges[g][(sims+2)*MAXCH]
//        w0 = N 14 | action 7 | child's edge count 6 | child terminal 1 | child's value 2
//        w3 = child id 13 | child's first edge 19
// One dwordx4 load brings an edge's (N, W, P) AND everything the walk needs to know about the child behind
// it, so a level of the PUCT walk is ONE dependent load (the children's edge block) instead of node-then-
// edges; the node array is read by the walk only for the position of the node it extends (issued beside the
// edge load, off the chain).  The select path is stored game-major as {edge index, w0, W} as seen at
// selection time, so the backup of the next launch is one coalesced load + one 8-byte store per edge (nothing
// else touches a game's tree in between), and the created leaf's legal mask / header go to per-game words, so
// the expansion does not re-read the node.  The words the tree step reads about a game sit in one 64-byte record
// (GameHot); the caller-visible per-game arrays (positions, state, leaf buffers) are SoA across games.  A simulation is two
// launches: k_tree_step ([expand + backup of the previous leaf] + [select of the next one]) and the evaluator;
// leaves that need the net are packed through a double-buffered device-side counter.
// Synthetic evaluators run the whole search in one launch (k_search_fused; k_search_fused_ttt for tic-tac-toe at
// sims <= 120: root edges in registers, child header packed into the edge word, path in LDS).  Opt-in: Dirichlet root
// noise (k_root_noise) and subtree reuse (two arenas, dev_reroot) -- DESIGN.md 3.9, 3.10.
//
// Float discipline: compiled with -ffp-contract=off; PUCT / softmax / backup use
// the single-rounding operation order of the oracle (oracle/bz_oracle.c), so
// visit counts, moves and W/P/pi are bit-identical to it.
//
// Reference anchors: turn loop + pass rule reversi_terminal.py:16-38;
// trajectory contract tic_tac_toe.py:13-34; canonical side-to-move states
// generate_training_games.py:12-23.  MCTS itself is build-authored (the
// reference has none).
#include <new>

#include <time.h>

#include "bz_common.h"
#include "bz_math.h"
#include "bz_rules.h"

using namespace bz;

namespace {

struct __attribute__((aligned(16))) Node { u64 own, opp, legal; u32 edge0, info; };
struct __attribute__((aligned(16))) Edge { u32 w0; float W; float P; u32 w3; };
static_assert(sizeof(Node) == 32 && sizeof(Edge) == 16, "layout");
// packed edge words (the limits they imply -- nodes per game <= kMaxNodes -- are checked by cfg_ok)
constexpr int kNBits = 14, kActShift = 14, kNchShift = 21, kTermShift = 27, kValShift = 28, kChildBits = 13;
constexpr u32 kNMask = (1u << kNBits) - 1u, kChildMask = (1u << kChildBits) - 1u;
constexpr int kMaxNodes = (1 << kChildBits) - 1;  // also bounds N (<= nodes) and the edge index (19 bits >= 8191 * 34)
__host__ __device__ __forceinline__ u32 e_N(u32 w0) { return w0 & kNMask; }
__host__ __device__ __forceinline__ int e_action(u32 w0) { return (int)((w0 >> kActShift) & 0x7Fu); }
__host__ __device__ __forceinline__ int e_nch(u32 w0) { return (int)((w0 >> kNchShift) & 0x3Fu); }
__host__ __device__ __forceinline__ bool e_term(u32 w0) { return ((w0 >> kTermShift) & 1u) != 0; }
__host__ __device__ __forceinline__ int e_val(u32 w0) { return (int)((w0 >> kValShift) & 3u) - 1; }
__host__ __device__ __forceinline__ u32 e_child(u32 w3) { return w3 & kChildMask; }
__host__ __device__ __forceinline__ u32 e_edge0(u32 w3) { return w3 >> kChildBits; }

constexpr u32 kTerm = 1u << 8;
// leaf_kind: NONE = slot idle; EVAL = the leaf awaits (logits, value); TERMINAL = backup of a terminal value;
// READY = nothing to expand or back up, select at once (a root whose subtree was kept from the previous move);
// COPY = the leaf's position was evaluated earlier in this search (evaluation cache): its priors and value are copied
// from that node instead of going through the evaluator again
enum { LEAF_NONE = 0, LEAF_EVAL = 1, LEAF_TERMINAL = 2, LEAF_READY = 3, LEAF_COPY = 4 };
enum { CNT_SIMS, CNT_PATH_NODES, CNT_CHILD_SCORED, CNT_EDGES_BACKED, CNT_EXPANDED, CNT_CHILD_WRITTEN,
       CNT_ENV_STEPS, CNT_NET_LEAVES, CNT_CACHE_HITS, CNT_CACHE_HITS_PREV, CNT_N };
constexpr int kCntWords = 24;  // u64 words of the counters block: CNT_N work counters, then (diagnostic builds) stamps at 16..23
// NEVAL[2]: packed-leaf counters, double-buffered by simulation parity (the tree step that
// packs into one buffer zeroes the other, so no extra reset launch is needed)
enum { FLAG_ERR = 0, FLAG_FINISHED = 1, FLAG_ACTIVE = 2, FLAG_NEVAL = 4, FLAG_N = 8 };
// ERR_EVAL_NONFINITE: the evaluator handed back a NaN / infinity (a diverged net, an external evaluator's bug): the
// softmax of such a row is NaN, every PUCT comparison with it is false and the search would quietly walk garbage
enum { ERR_EDGE_OVERFLOW = 1, ERR_TERMINAL_ROOT = 2, ERR_EXAMPLE_OVERFLOW = 4, ERR_DEPTH = 8, ERR_EVAL_NONFINITE = 16 };

struct __attribute__((aligned(16))) PathEnt { u32 eidx; u32 w0; float W; u32 pad; };

// the tree step's per-game words in ONE 64-byte line (they were nine arrays: nine cache lines, nine TLB lookups per game at
// the head of every launch, all of them missing the L2 after the net kernel has streamed its weights through it)
struct __attribute__((aligned(64))) GameHot {
    u64 leaf_legal; u32 leaf_node, leaf_info;   // the leaf awaiting expansion / backup: its legal mask, node id, header
    u32 depth, n_nodes, n_edges, leaf_slot;     // its depth; the tree's fill; the evaluator row of the leaf
    u32 root_n, root_base;                      // the root's child count; visits a kept subtree came with (subtree reuse)
    // evaluation cache: this search's generation; a COPY leaf's source node (bit 31: it lives in the PREVIOUS search's arena)
    // and that node's first edge; the generation of the last search this slot took part in and how many nodes the
    // previous search's arena holds for it (0: nothing to carry over)
    u32 tt_gen, copy_src, copy_e0, last_gen, prev_nodes, pad[1];
};
constexpr u32 kSrcPrev = 1u << 31;
static_assert(sizeof(GameHot) == 64, "layout");

struct EngineDev {
    int B, ncap, ecap, sims, na, t_max, rounds, temp_moves, openings, maxd, stagger;
    int compact;  // net evaluators: leaves needing evaluation are packed (c_own/c_opp/logits/value by slot)
    int reuse;    // BZ_ENGINE_REUSE_SUBTREE: two tree arenas; nodes/edges = this move's, *_alt = the previous move's
    Node* nodes_alt; Edge* edges_alt; u32* g_reuse;
    float c_puct, dir_alpha, dir_eps;  // dir_eps > 0: Dirichlet noise on the root priors
    u64 seed, id_base, id_stride;
    Node* nodes; Edge* edges;
    u64 *g_own, *g_opp; int8_t* g_to_move; uint8_t* g_state; int32_t *g_moves, *g_nex, *g_round, *g_passes;
    struct GameHot* hot;  // [B]
    PathEnt* path;        // [B][maxd], game-major
    uint8_t* leaf_kind; u64 *leaf_own, *leaf_opp, *c_own, *c_opp;
    float *logits, *value;
    u64 *ex_own, *ex_opp; float* ex_pi; int8_t *ex_z, *ex_mover; uint8_t* ex_act; int32_t* ex_len; int8_t* ex_winner;
    u32* root_N; float *root_W, *root_P;
    u64* counters; u64* cnt_slots; int n_cnt_slots; u32* flags;
    int32_t* pack_off;  // [rounds][B]: first row of a finished game in the packed example block (k_pack_scan)
    // evaluation cache (BZ_ENGINE_EVAL_CACHE, net evaluators): per game a hash table position -> node of its first
    // evaluation in the current search (tt_buckets buckets of 16 eight-byte entries) and every node's value
    // ecache == 2 (BZ_ENGINE_EVAL_CACHE_CARRY): the previous search's tree stays intact in the other arena (nodes_alt / edges_alt
    // / node_v_alt: the arenas alternate search by search) and its evaluations serve this search too
    int ecache, tt_buckets; u64* tt; float *node_v, *node_v_alt;
};

struct Cnt { u32 v[CNT_N]; };

// Diagnostic build only (betazero_amd.build.build_variant("treestamps", ["-DBZ_EXP_TREE_STAMPS"]), tools/exp_tree_stamps.py):
// shader-clock stamps between the phases of k_tree_step, summed over all waves into counters[16..23].  The product build
// compiles the empty struct away.
#if defined(BZ_EXP_TREE_STAMPS) || defined(BZ_EXP_NO_COOP_ENV) || defined(BZ_EXP_TT_WEAK_HASH)
#ifndef BZ_EXPERIMENT
#error "BZ_EXP_* are diagnostic options: build them through betazero_amd.build.build_variant()"
#endif
#endif
#ifdef BZ_EXP_TREE_STAMPS
struct Stamps {
    u64 last; u32 acc[8];
    __device__ __forceinline__ void start() { for (int k = 0; k < 8; ++k) acc[k] = 0; last = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void mark(int k) { const u64 now = __builtin_readcyclecounter(); acc[k] += (u32)(now - last); last = now; }
    __device__ __forceinline__ void flush(u64* counters) {
        if ((threadIdx.x & 63) == 0) {
            for (int k = 0; k < 7; ++k) atomicAdd(reinterpret_cast<unsigned long long*>(counters) + 16 + k, (unsigned long long)acc[k]);
            atomicAdd(reinterpret_cast<unsigned long long*>(counters) + 23, 1ULL);  // waves
        }
    }
};
#else
struct Stamps {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void flush(u64*) {}
};
#endif

// ---- lane exchange inside a lane group (kGW <= 16 lanes = at most one DPP row): butterfly partner for step O,
// valid when the steps run in ASCENDING order (1, 2, 4, 8): 1 and 2 are quad permutes, 4 / 8 the half-row / row
// mirrors (the lower levels are uniform by then, so the mirror image is the partner half).  DPP moves are VALU
// operations; the ds_bpermute a generic shuffle compiles to is an LDS round trip.
template <int O>
__device__ __forceinline__ int xchg_i(int x) {
    static_assert(O == 1 || O == 2 || O == 4 || O == 8, "butterfly step");
    constexpr int ctrl = O == 1 ? 0xB1 : (O == 2 ? 0x4E : (O == 4 ? 0x141 : 0x140));
    return __builtin_amdgcn_update_dpp(0, x, ctrl, 0xF, 0xF, true);
}
template <int O> __device__ __forceinline__ u32 xchg(u32 x) { return (u32)xchg_i<O>((int)x); }
template <int O> __device__ __forceinline__ int xchg(int x) { return xchg_i<O>(x); }
template <int O> __device__ __forceinline__ float xchg(float x) { return __int_as_float(xchg_i<O>(__float_as_int(x))); }
// value of the previous lane of the row (lane 0 of a row: 0)
__device__ __forceinline__ float row_shr1(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xF, 0xF, true));
}

// PUCT candidates: first maximum (lowest child index on ties) over the kGW lanes of a group
struct Cand { float sc; int i; u32 w0, w3; float W; };
template <int O>
__device__ __forceinline__ void cand_step(Cand& c) {
    const float s2 = xchg<O>(c.sc), W2 = xchg<O>(c.W); const int i2 = xchg<O>(c.i);
    const u32 a2 = xchg<O>(c.w0), b2 = xchg<O>(c.w3);
    const bool take = (s2 > c.sc) || (s2 == c.sc && i2 < c.i);
    if (take) { c.sc = s2; c.i = i2; c.w0 = a2; c.w3 = b2; c.W = W2; }
}
template <int kGW>
__device__ __forceinline__ void group_argmax(Cand& c) {
    if (kGW > 1) cand_step<1>(c);
    if (kGW > 2) cand_step<2>(c);
    if (kGW > 4) cand_step<4>(c);
    if (kGW > 8) cand_step<8>(c);
}
template <int kGW>
__device__ __forceinline__ float group_max(float m) {
    if (kGW > 1) { float m2 = xchg<1>(m); m = m2 > m ? m2 : m; }
    if (kGW > 2) { float m2 = xchg<2>(m); m = m2 > m ? m2 : m; }
    if (kGW > 4) { float m2 = xchg<4>(m); m = m2 > m ? m2 : m; }
    if (kGW > 8) { float m2 = xchg<8>(m); m = m2 > m ? m2 : m; }
    return m;
}

// counters are accumulated by the lead lane of every lane group and flushed per wave into that wave's own slot
// (1000+ waves hammering 8 shared addresses cost more than the tree walk itself): the leads' counts are summed
// across the groups, then lanes 0..7 each add ONE counter with an atomic that returns nothing -- nothing waits
// for it (eight load -> wait -> store sequences at the end of every wave were a fifth of the tree step's time).
// k_sum_counters folds the slots into counters[16] when the host asks.
template <int kGW>
__device__ __forceinline__ void cnt_flush(const EngineDev& E, Cnt& c) {
    const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = (int)(threadIdx.x & 63);
    u32 mine = 0;
#pragma unroll
    for (int k = 0; k < CNT_N; ++k) {
        u32 x = c.v[k];
        if (kGW >= 16) {  // few leads (lanes 0, kGW, ...): read them straight into scalar registers -- no LDS round trips
            u32 sum = 0;
#pragma unroll
            for (int l = 0; l < 64; l += kGW) sum += (u32)__builtin_amdgcn_readlane((int)x, l);
            x = sum;
        } else {
#pragma unroll
            for (int o = 32; o >= kGW; o >>= 1) x += __shfl_xor(x, o, 64);  // (lanes that are not leads hold zeros)
            x = (u32)__builtin_amdgcn_readfirstlane((int)x);                 // lane 0 is a lead: it holds the wave's sum
        }
        mine = lane == k ? x : mine;
    }
    if (lane < CNT_N && mine && wave < (u32)E.n_cnt_slots)
        __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(E.cnt_slots) + (size_t)wave * CNT_N + lane,
                               (unsigned long long)mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(256) k_sum_counters(EngineDev E) {
    __shared__ unsigned long long part[256];
    for (int k = 0; k < CNT_N; ++k) {
        unsigned long long acc = 0;
        for (int w = threadIdx.x; w < E.n_cnt_slots; w += 256) acc += E.cnt_slots[(size_t)w * CNT_N + k];
        part[threadIdx.x] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o]; __syncthreads(); }
        if (threadIdx.x == 0) E.counters[k] = part[0];
        __syncthreads();
    }
}

// Orders a lane group's earlier stores (edges / node header written by dev_expand, child node
// written by dev_select) before its later loads of the same addresses.  All communication is
// between lanes of ONE wave (a game's G::GW lanes divide 64), so WAVEFRONT scope is the exact
// scope: acq_rel keeps the compiler from moving the stores or the following loads across it,
// and the hardware executes one wave's vector-memory instructions to an address in issue order,
// so no s_waitcnt vmcnt(0) (a full store round trip, which a workgroup-scope fence costs) is needed.
__device__ __forceinline__ void group_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// where the select walk records its path: edge index + that edge's w0 (N and the child header) and W as the walk
// saw / left them, so that the backup is a pure store.  HBM, game-major (a game's entries are one coalesced load for
// its lane group), for the step-by-step kernels whose backup runs in the next launch; LDS for the fused search.
// (every lane of the group calls put() with the same arguments)
template <int kGW>
struct PathHbm {  // lane d of the group keeps entry d in registers during the walk -- no store, so nothing for the next
    PathEnt* p;   // level's load to queue behind -- and writes it at the end (one coalesced store per game); levels
    PathEnt mine; // beyond the group width (rare) go to memory directly
    __device__ __forceinline__ void put(int d, int sub, u32 eidx, u32 w0, float W) {
        if (d < kGW) { if (sub == d) { mine.eidx = eidx; mine.w0 = w0; mine.W = W; } }
        else if (sub == 0) { PathEnt e; e.eidx = eidx; e.w0 = w0; e.W = W; e.pad = 0; p[d] = e; }
    }
    __device__ __forceinline__ void flush(int sub, int depth) const { if (sub < depth && sub < kGW) p[sub] = mine; }
};
struct PathEntLds { u32 eidx; u32 w0; float W; };  // 12 bytes: LDS capacity decides how many games a CU holds
struct PathLds {
    PathEntLds* p;  // this game's MAXD entries
    __device__ __forceinline__ void put(int d, int sub, u32 eidx, u32 w0, float W) const {
        if (sub == 0) { PathEntLds e; e.eidx = eidx; e.w0 = w0; e.W = W; p[d] = e; }
    }
};

// logits source for the expansion of one leaf
struct LogitSrc {
    int kind; u64 h; const float* row;
    __device__ __forceinline__ float operator()(int a) const {
        return kind == BZ_EVAL_UNIFORM ? 0.0f : (kind == BZ_EVAL_HASH ? hash_logit(h, a) : row[a]);
    }
};

// ---- cooperative tree walk: G::GW lanes serve one game (Reversi 16 -> 4 games per wave, TTT 4 -> 16
// games per wave: the walk is a dependent chain per game, so games in flight = memory-level
// parallelism).  A level of the PUCT walk is one coalesced 256-byte load of up to 16 edges (the chosen
// edge's words say where the child's edges are, how many, and whether the child is terminal), the
// scores are computed one child per lane and reduced with 4 shuffle steps (first maximum, i.e.
// lowest action on ties); softmax terms are computed one legal move per lane, but SUMMED in
// ascending action order by a serial shuffle scan so that the result is bit-identical to the
// sequential spec; the backup updates one path edge per lane.

// ---- the env step of child creation, shared out over a lane group.  All kGW lanes of a group would otherwise
// compute the same 8 rays of make_move (reversi_board.py:49-58) and the same 8 rays of the child's legal mask
// (:25-41, :87-88) redundantly -- ~400 integer instructions per wave, on the walk's critical path.  Instead lane
// (sub & 7) takes ONE ray direction: directions 0..3 shift left by 1 / 8 / 7 / 9; 4..7 are their opposites,
// computed as the same left shifts on the bit-reversed boards (bit i <-> 63 - i maps a right shift onto a left
// shift; the column-wrap mask is symmetric), and the 8 partial masks are OR-ed across the lanes with DPP moves.
__device__ __forceinline__ u64 brev64(u64 x) { return ((u64)__brev((u32)x) << 32) | (u64)__brev((u32)(x >> 32)); }
__device__ __forceinline__ u64 or8(u64 x) {  // OR over the 8 lanes of a half row (lanes 8..15 of a 16-lane group mirror 0..7)
    u32 lo = (u32)x, hi = (u32)(x >> 32);
    lo |= xchg<1>(lo); hi |= xchg<1>(hi);
    lo |= xchg<2>(lo); hi |= xchg<2>(hi);
    lo |= xchg<4>(lo); hi |= xchg<4>(hi);
    return ((u64)hi << 32) | lo;
}
// one ray direction (left shift by s) of generate_possible_moves / of the flips of placing on m; o = opponent
// stones pre-masked against column wrap; runs of <= 6 stones by parallel prefix (2 + 2 + 2 cells)
__device__ __forceinline__ u64 ray_fill(u64 seed, u64 o, int s) {
    u64 fl = o & (seed << s);
    fl = and_or(o, fl << s, fl);
    const u64 pl = o & (o << s);
    fl = and_or(pl, fl << (2 * s), fl);
    fl = and_or(pl, fl << (2 * s), fl);
    return fl;
}
template <class G, bool kShare = (G::kGame != 0 && G::GW >= 8)>
struct CoopChild {  // tic-tac-toe / narrow groups: nothing to share out
    static __device__ __forceinline__ void run(u64 own, u64 opp, int act, int, u64* cown, u64* copp, u64* legal) {
        G::apply(own, opp, act, cown, copp);
        *legal = G::legal(*cown, *copp);
    }
};
template <class G>
struct CoopChild<G, true> {
    static __device__ __forceinline__ void run(u64 own, u64 opp, int act, int sub, u64* cown, u64* copp, u64* legal) {
        const int d = sub & 7, k = d & 3;
        const int s = k == 0 ? 1 : (k == 1 ? 8 : (k == 2 ? 7 : 9));
        const bool rv = d >= 4;
        const u64 wrap = k == 1 ? ~0ULL : kInner;
        u64 co = opp, cp = own;
        if (act != kPass) {
            const u64 m = 1ULL << act;
            const u64 a = rv ? brev64(own) : own, b = rv ? brev64(opp) : opp, mm = rv ? brev64(m) : m;
            u64 fl = ray_fill(mm, b & wrap, s);
            fl = ((fl << s) & a) ? fl : 0ULL;
            const u64 f = or8(rv ? brev64(fl) : fl);
            co = opp & ~f; cp = own | m | f;
        }
        const u64 a = rv ? brev64(co) : co, b = rv ? brev64(cp) : cp;
        const u64 mv = ray_fill(a, b & wrap, s) << s;
        *legal = or8(rv ? brev64(mv) : mv) & ~(co | cp) & G::kValid;
        *cown = co; *copp = cp;
    }
};

// what the walk starts from: the root's position, mover colour and child count (its edges start at index 0)
// `pre`: this lane's root edge (index sub) when the caller fetched the first kGW root edges ahead of the walk
struct RootRef { u64 own, opp; int tm; int n; u32 sumN; bool has_pre; Edge pre; u32 tt_gen, prev_nodes; };
// the node select created (valid when a child was created); src / src_e0: the node whose evaluation a COPY leaf shares
struct LeafPos { u64 own, opp, legal; u32 info, src, src_e0; };

// ---- evaluation cache.  A search reaches some positions by more than one move order (measured on the CPU oracle: 6 - 12 %
// of the 800 evaluations of a cfg-3 search, profiles/r05_leaf_duplication.json); the net is a function of the position
// alone, so the second node takes the first one's priors and value -- bit for bit what the evaluator would have returned
// -- and no evaluator row.  The TREE is unchanged: the repeat is a node of its own with its own statistics (exact
// sequential MCTS, the oracle's tree), only the evaluation is shared.  Per game a table of tt_buckets x 16 entries
// {tag 32 | generation 19 | node 13}; the bucket is one coalesced 128-byte load by the game's lanes; a tag match is
// CONFIRMED against the stored node's position (a false positive is impossible, whatever the hash does); entries of
// earlier searches (other generation) are free slots; a full bucket just means no insertion.
constexpr u32 kTtGenMax = (1u << 19) - 1u;
template <int kGW>
__device__ __forceinline__ u32 group_min_u32(u32 x) {
    if (kGW > 1) { const u32 y = xchg<1>(x); x = y < x ? y : x; }
    if (kGW > 2) { const u32 y = xchg<2>(x); x = y < x ? y : x; }
    if (kGW > 4) { const u32 y = xchg<4>(x); x = y < x ? y : x; }
    if (kGW > 8) { const u32 y = xchg<8>(x); x = y < x ? y : x; }
    return x;
}
// looks (own, opp) up.  A confirmed hit returns true with the source node -- flagged kSrcPrev when it is a node of the
// PREVIOUS search's tree (carry-over: generation gen - 1, arena nodes_alt, only ids below prev_nodes) -- and its first edge.
// A miss, and a carry-over hit too, inserts `new_id` for this generation (nodes 0 .. new_id - 1 of this search exist and are
// expanded; the new node will be by the time anyone finds it).  All lanes of the group return the same values.
template <int kGW>
__device__ __forceinline__ bool tt_lookup_insert(const EngineDev& E, int g, int sub, u64 own, u64 opp, u32 gen, u32 prev_nodes,
                                                 u32 new_id, u32& src, u32& src_e0) {
#ifdef BZ_EXP_TT_WEAK_HASH
    // TEST BUILD ONLY (tests/test_gpu_parity.py): every position has tag 0 and lives in one of two buckets -- tag matches that
    // are NOT the position and full buckets become the normal case; the results must still be those without any cache
    const u64 h = hash_pos(own, opp) & 1u;
#else
    const u64 h = hash_pos(own, opp);
#endif
    const u32 tag = (u32)(h >> 32);
    const u32 gen_prev = gen == 1u ? kTtGenMax - 1u : gen - 1u;  // (the host's counter cycles 1 .. 2^19 - 2)
    u64* bucket = E.tt + ((size_t)g * (size_t)E.tt_buckets + (size_t)(h & (u64)(E.tt_buckets - 1))) * 16;
    // (is-previous << 20 | slot << 16 | node) of the best matching slot this lane saw: this search's entries win
    u32 hit = ~0u, fre = ~0u;
#pragma unroll
    for (int s0 = 0; s0 < 16; s0 += kGW) {
        const int s = s0 + sub;
        const u64 e = bucket[s];
        const u32 meta = (u32)(e >> 32), egen = meta >> kChildBits;
        const bool cur = egen == gen, prv = prev_nodes != 0u && egen == gen_prev;
        if ((cur || prv) && (u32)e == tag) {
            const u32 key = ((prv ? 1u : 0u) << 20) | ((u32)s << 16) | (meta & kChildMask);
            hit = key < hit ? key : hit;
        }
        if (!cur && !prv && fre == ~0u) fre = (u32)s << 16;
    }
    hit = group_min_u32<kGW>(hit);
    fre = group_min_u32<kGW>(fre);
    bool found = false, from_prev = false;
    if (hit != ~0u) {
        const u32 x = hit & kChildMask;
        from_prev = (hit >> 20) != 0u;
        if (x != 0u && x < (from_prev ? prev_nodes : new_id)) {
            const Node xn = (from_prev ? E.nodes_alt : E.nodes)[(size_t)g * E.ncap + x];  // (one address for the whole group)
            found = xn.own == own && xn.opp == opp && !(xn.info & kTerm) && (xn.info & 0xFFu) != 0u;
            src = x | (from_prev ? kSrcPrev : 0u); src_e0 = xn.edge0;
        }
    }
    if ((!found || from_prev) && fre != ~0u) {
        const int s = (int)(fre >> 16);
        if (sub == s % kGW) bucket[s] = (u64)tag | ((u64)((gen << kChildBits) | new_id) << 32);
    }
    return found;
}

// M2: PUCT walk from the root; creates the child node behind the chosen unexpanded edge (env step:
// apply + legal + terminal).  All lanes of the group return the same values.
template <class G, class Sink>
__device__ __forceinline__ void dev_select(const EngineDev& E, int g, int sub, const RootRef& root, u32& n_nodes_g,
                                           u32& leaf, int& kind, int& depth_out, float& tval, Cnt& c,
                                           Sink& sink, LeafPos& lp, Stamps& st) {
    constexpr int kGW = G::GW;
    Node* nodes = E.nodes + (size_t)g * E.ncap;
    Edge* edges = E.edges + (size_t)g * E.ecap;
    // sum of the root's child visits == simulations done so far (+ the visits a kept subtree came with)
    u32 e0 = 0, sumN = root.sumN;
    int n = root.n, depth = 0;
    u64 pown = root.own, popp = root.opp;  // position of the node whose edges are being scored
    const bool lead = sub == 0;
    if (lead) { c.v[CNT_SIMS]++; c.v[CNT_PATH_NODES]++; }
    for (;;) {
        const float sq = fsqrt((float)(sumN > 1u ? sumN : 1u));
        const Edge* ed = edges + e0;
        float bests = -__builtin_inff(), bestW = 0.0f; int best = 0; u32 bestw0 = 0, bestw3 = 0;
        for (int base = 0; base < n; base += kGW) {
            Cand cd; cd.i = base + sub; cd.sc = -__builtin_inff(); cd.W = 0.0f; cd.w0 = 0; cd.w3 = 0;
            if (cd.i < n) {
                Edge e = root.pre;
                if (!(root.has_pre && depth == 0 && base == 0)) e = ed[cd.i];
                const u32 N = e_N(e.w0);
                float q = N > 0 ? fdiv(e.W, (float)N) : 0.0f;
                float u = E.c_puct * e.P;
                u = u * sq;
                u = fdiv(u, 1.0f + (float)N);
                cd.sc = q + u; cd.w0 = e.w0; cd.w3 = e.w3; cd.W = e.W;
            }
            group_argmax<kGW>(cd);
            if (cd.sc > bests) { bests = cd.sc; best = cd.i; bestw0 = cd.w0; bestw3 = cd.w3; bestW = cd.W; }
        }
        if (lead) c.v[CNT_CHILD_SCORED] += (u32)n;
        const u32 eidx = e0 + (u32)best;
        const u32 child = e_child(bestw3);
        if (child) {
            if (lead) c.v[CNT_PATH_NODES]++;
            if (depth < E.maxd) sink.put(depth, sub, eidx, bestw0, bestW);
            else if (lead) atomicOr(&E.flags[FLAG_ERR], ERR_DEPTH);
            depth++;
            if (e_term(bestw0)) { kind = LEAF_TERMINAL; leaf = child; tval = (float)e_val(bestw0); break; }
            // children's visits of X == visits of the edge into X minus the creating one
            e0 = e_edge0(bestw3); n = e_nch(bestw0); sumN = e_N(bestw0) - 1u;
            // X's position is needed only if the walk ends by extending X: the load goes out beside X's edge load
            const Node* xn = nodes + child;
            pown = xn->own; popp = xn->opp;
            continue;
        }
        st.mark(3);
        const int act = e_action(bestw0);
        u64 cown, copp, lg;
#ifdef BZ_EXP_NO_COOP_ENV
        G::apply(pown, popp, act, &cown, &copp);
        lg = G::legal(cown, copp);
#else
        CoopChild<G>::run(pown, popp, act, sub, &cown, &copp, &lg);
#endif
        const u32 id = n_nodes_g++;
        const int tm = (depth & 1) ? root.tm : -root.tm;  // child's mover: the colours alternate down the walk (passes included)
        int tv = 0;
        const bool term = G::terminal(cown, copp, tm, lg, &tv);
        lp.own = cown; lp.opp = copp; lp.legal = term ? 0 : lg;
        lp.info = (term ? kTerm : 0u) | ((u32)(tv + 1) << 9) | ((tm == 1 ? 1u : 0u) << 11);
        const u32 w0 = bestw0 | ((term ? 1u : 0u) << kTermShift) | ((u32)(tv + 1) << kValShift);
        if (lead) {
            Node ch;
            ch.own = cown; ch.opp = copp; ch.legal = lp.legal; ch.edge0 = 0; ch.info = lp.info;
            nodes[id] = ch;
            edges[eidx].w0 = w0; edges[eidx].w3 = id;  // (the child's first edge / edge count follow with its expansion)
            c.v[CNT_ENV_STEPS]++; c.v[CNT_PATH_NODES]++;
        }
        if (depth < E.maxd) sink.put(depth, sub, eidx, w0, bestW);
        else if (lead) atomicOr(&E.flags[FLAG_ERR], ERR_DEPTH);
        depth++;
        leaf = id; kind = term ? LEAF_TERMINAL : LEAF_EVAL; tval = (float)tv;
        if (E.ecache && !term) {  // evaluated before in this search? (wave-uniform branch: a kernel argument)
            u32 src = 0, src_e0 = 0;
            if (tt_lookup_insert<kGW>(E, g, sub, cown, copp, root.tt_gen, root.prev_nodes, id, src, src_e0)) { kind = LEAF_COPY; lp.src = src; lp.src_e0 = src_e0; }
        }
        st.mark(4);
        break;
    }
    depth_out = depth;
}

// k-th (0-based) set bit of m, or -1
__device__ __forceinline__ int nth_bit(u64 m, int k) {
    if (k >= popc64(m)) return -1;
    u32 w = (u32)m;
    int pos = 0, c = __popc(w);
    if (k >= c) { k -= c; w = (u32)(m >> 32); pos = 32; }
    c = __popc(w & 0xFFFFu); if (k >= c) { k -= c; w >>= 16; pos += 16; }
    c = __popc(w & 0xFFu);   if (k >= c) { k -= c; w >>= 8;  pos += 8; }
    c = __popc(w & 0xFu);    if (k >= c) { k -= c; w >>= 4;  pos += 4; }
    c = __popc(w & 0x3u);    if (k >= c) { k -= c; w >>= 2;  pos += 2; }
    return pos + (k >= (int)(w & 1u) ? 1 : 0);
}

// how dev_expand writes a fresh edge: the engine's packed record, or the private record of the tic-tac-toe
// fused search (plain N in word 0, its own child header in word 3)
struct EdgeFmtPacked { static __device__ __forceinline__ Edge make(float P, int a) { Edge e; e.w0 = (u32)a << kActShift; e.W = 0.0f; e.P = P; e.w3 = 0; return e; } };
struct EdgeFmtTttFused { static __device__ __forceinline__ Edge make(float P, int a) { Edge e; e.w0 = 0; e.W = 0.0f; e.P = P; e.w3 = (u32)a << 24; return e; } };

// M3: masked softmax over the legal actions (ascending), edges bump-allocated.  `legal` is the
// leaf's legal mask (known to the caller: the root's or the node select just created).
// `info` = the leaf's header word as created (child count 0): the header is rewritten, never re-read.
// Returns the number of edges written (they start at the old n_edges_g).
// kUniform: the caller's logits are all equal (the uniform evaluator, BASELINE cfg 2).  The spec's arithmetic then
// collapses without changing a bit: m = 0, every e = expf_spec(0) = 1.0 exactly, their sequential sum is the integer n
// exactly, P = 1 / float(n) by the same correctly rounded division -- no exponentials, no serial sum, no logits.
template <class G, int kGW = G::GW, bool kHeader = true, class Fmt = EdgeFmtPacked, bool kUniform = false>
__device__ __forceinline__ int dev_expand(const EngineDev& E, int g, int sub, u32 leaf, u64 legal, u32 info,
                                          const LogitSrc& ls, u32& n_edges_g, Cnt& c, Stamps& st) {
    constexpr int kCH = (G::MAXCH + kGW - 1) / kGW;
    Node* nd = E.nodes + (size_t)g * E.ncap + leaf;
    const u32 e0 = n_edges_g;
    Edge* ed = E.edges + (size_t)g * E.ecap + e0;
    const int room = E.ecap - (int)e0;
    int n = 0;
    if (legal == 0) {  // forced pass: one edge, P = 1
        if (room >= 1) {
            if (sub == 0) ed[0] = Fmt::make(1.0f, kPass);
            n = 1;
        }
    } else {
        n = popc64(legal);
        if (n > room) { if (sub == 0) atomicOr(&E.flags[FLAG_ERR], ERR_EDGE_OVERFLOW); n = room; }
        // this lane's moves: the sub-th, (sub+GW)-th, ... legal actions
        int a[kCH]; float x[kCH];
        float m = -__builtin_inff();
        u32 rest = (u32)legal;  // small boards (tic-tac-toe: 9 cells): walk the mask instead of a rank search per move
        if (G::NA <= 32)
            for (int j = 0; j < sub; ++j) rest &= rest - 1u;
#pragma unroll
        for (int k = 0; k < kCH; ++k) {
            if (G::NA <= 32) {  // this lane's k-th move = the lowest bit left; then skip the other lanes' kGW - 1 moves
                a[k] = (sub + kGW * k < n && rest) ? (int)__builtin_ctz(rest) : -1;
#pragma unroll
                for (int j = 0; j < kGW; ++j) rest &= rest - 1u;
            } else {
                a[k] = (sub + kGW * k < n) ? nth_bit(legal, sub + kGW * k) : -1;
            }
            if (!kUniform) {
                x[k] = a[k] >= 0 ? ls(a[k]) : -__builtin_inff();
                m = x[k] > m ? x[k] : m;
            }
        }
        if (!kUniform) m = group_max<kGW>(m);
        st.mark(1);  // (the evaluator's row has arrived)
        float ex[kCH];
#pragma unroll
        for (int k = 0; k < kCH; ++k) ex[k] = a[k] >= 0 ? (kUniform ? 1.0f : expf_spec(x[k] - m)) : 0.0f;
        // ascending-action serial sum (the spec's order): the partial sum travels up the group one lane per step
        // (lane t adds its own term to what lane t-1 holds), every step one DPP add; lanes past the last move hold
        // +0.0, which leaves a positive sum unchanged, so the chain runs its full length without tests.
        float s = kUniform ? (float)n : 0.0f;
#pragma unroll
        for (int k = 0; k < kCH; ++k) {
            if (!kUniform && kGW * k < n) {
                float part = s + ex[k];  // (lane 0's is the true partial sum; the others are overwritten below)
#pragma unroll
                for (int t = 1; t < kGW; ++t) {
                    const float cand = row_shr1(part) + ex[k];
                    part = sub == t ? cand : part;
                }
                s = __shfl(part, kGW - 1, kGW);
            }
        }
        if (sub == 0 && !(s >= 1.0f && s <= 3.0e38f)) atomicOr(&E.flags[FLAG_ERR], ERR_EVAL_NONFINITE);  // (the maximum's term is exactly 1)
        float pr[kCH];
#pragma unroll
        for (int k = 0; k < kCH; ++k) pr[k] = a[k] >= 0 ? fdiv(ex[k], s) : 0.0f;
#pragma unroll
        for (int k = 0; k < kCH; ++k)
            if (a[k] >= 0) ed[sub + kGW * k] = Fmt::make(pr[k], a[k]);
    }
    if (sub == 0) {
        if (kHeader) *reinterpret_cast<uint2*>(&nd->edge0) = make_uint2(e0, (info & ~0xFFu) | (u32)n);  // edge0, info: one 8-byte store
        c.v[CNT_EXPANDED]++;
        c.v[CNT_CHILD_WRITTEN] += (u32)n;
    }
    n_edges_g = e0 + (u32)n;
    return n;
}

// M3 for a leaf whose position was evaluated earlier in this search (evaluation cache): the same edges -- actions in
// ascending order, the priors the masked softmax gave the first time, bit for bit -- copied from that node's block;
// N = 0, W = 0, no child.  legal / info as for dev_expand.  Returns the number of edges written.
template <class G, int kGW = G::GW>
__device__ __forceinline__ int dev_expand_copy(const EngineDev& E, int g, int sub, u32 leaf, u64 legal, u32 info, u32 src_e0,
                                               bool from_prev, u32& n_edges_g, Cnt& c) {
    constexpr int kCH = (G::MAXCH + kGW - 1) / kGW;
    Node* nd = E.nodes + (size_t)g * E.ncap + leaf;
    const u32 e0 = n_edges_g;
    Edge* ed = E.edges + (size_t)g * E.ecap + e0;
    const Edge* from = (from_prev ? E.edges_alt : E.edges) + (size_t)g * E.ecap + src_e0;
    const int room = E.ecap - (int)e0;
    int n = legal == 0 ? 1 : popc64(legal);
    if (n > room) { if (sub == 0) atomicOr(&E.flags[FLAG_ERR], ERR_EDGE_OVERFLOW); n = room; }
    Edge got[kCH];
#pragma unroll
    for (int k = 0; k < kCH; ++k) if (sub + kGW * k < n) got[k] = from[sub + kGW * k];
#pragma unroll
    for (int k = 0; k < kCH; ++k) if (sub + kGW * k < n) ed[sub + kGW * k] = EdgeFmtPacked::make(got[k].P, e_action(got[k].w0));
    if (sub == 0) {
        *reinterpret_cast<uint2*>(&nd->edge0) = make_uint2(e0, (info & ~0xFFu) | (u32)n);
        c.v[CNT_EXPANDED]++;
        c.v[CNT_CHILD_WRITTEN] += (u32)n;
        c.v[CNT_CACHE_HITS]++;
        if (from_prev) c.v[CNT_CACHE_HITS_PREV]++;
    }
    n_edges_g = e0 + (u32)n;
    return n;
}

// M4: W is stored for the mover at the parent, so the sign flips every ply; one path edge per lane.
// N and W of every path edge were captured by the select walk (nothing else touches this game's tree
// in between), so the backup is one 8-byte store per path edge and reads nothing but the path.  The
// deepest edge leads to the leaf: when the leaf was expanded just now (exp_n > 0 edges from exp_e0),
// its edge count / first edge go into that edge's words with the same stores.
template <class PE>
__device__ __forceinline__ uint2 backup_edge(Edge* edges, const PE& pe, float val, bool deepest, u32 leaf, u32 exp_e0, int exp_n) {
    u32 w0 = pe.w0 + 1u;
    if (deepest && exp_n > 0) {
        w0 |= (u32)exp_n << kNchShift;
        edges[pe.eidx].w3 = leaf | (exp_e0 << kChildBits);
    }
    const uint2 wr = make_uint2(w0, __float_as_uint(pe.W + val));
    *reinterpret_cast<uint2*>(edges + pe.eidx) = wr;
    return wr;
}

template <int kGW, class PE>
__device__ __forceinline__ void dev_backup(const EngineDev& E, int g, int sub, int depth, float v, const PE* path,
                                           u32 leaf, u32 exp_e0, int exp_n, Cnt& c) {
    Edge* edges = E.edges + (size_t)g * E.ecap;
    const int dmax = depth < E.maxd ? depth : E.maxd;
    for (int d = sub; d < dmax; d += kGW) {
        const PE pe = path[d];
        const float val = ((dmax - 1 - d) & 1) ? v : -v;  // deepest edge gets -v
        backup_edge(edges, pe, val, d == depth - 1, leaf, exp_e0, exp_n);
    }
    if (sub == 0) c.v[CNT_EDGES_BACKED] += (u32)dmax;
}

template <class G>
__device__ __forceinline__ bool dev_root_init(const EngineDev& E, int g) {
    u64 own = E.g_own[g], opp = E.g_opp[g];
    int tm = E.g_to_move[g];
    u64 lg = G::legal(own, opp);
    int tv;
    if (G::terminal(own, opp, tm, lg, &tv)) {
        atomicOr(&E.flags[FLAG_ERR], ERR_TERMINAL_ROOT);
        return false;
    }
    Node r; r.own = own; r.opp = opp; r.legal = lg; r.edge0 = 0; r.info = (tm == 1 ? 1u : 0u) << 11;
    E.nodes[(size_t)g * E.ncap] = r;
    E.hot[g].leaf_legal = lg; E.hot[g].leaf_info = r.info;  // what the expansion of this "leaf" reads
    return true;
}

// start position (+ fixed two-ply openings) of the game with global id gid
template <class G>
__device__ __forceinline__ void dev_start_game(const EngineDev& E, int g, int round) {
    u64 own, opp;
    G::start(&own, &opp);
    int tm = 1, made = 0;
    if (G::kGame == 1 && E.openings) {  // the 12 two-ply openings are an 8x8 notion
        u64 gid = E.id_base + (u64)round * E.id_stride + (u64)g;
        int k = (int)(gid % 12ULL);
        int pick[2] = {k / 3, k % 3};
        for (int i = 0; i < 2; ++i) {
            u64 l = G::legal(own, opp);
            for (int j = 0; j < pick[i]; ++j) l &= l - 1;
            u64 c0, c1;
            G::apply(own, opp, ctz64(l), &c0, &c1);
            own = c0; opp = c1; tm = -tm; made++;
        }
    }
    E.g_own[g] = own; E.g_opp[g] = opp; E.g_to_move[g] = (int8_t)tm;
    E.g_moves[g] = made; E.g_nex[g] = 0; E.g_round[g] = round; E.g_passes[g] = 0; E.g_state[g] = 0;
    if (E.reuse) E.g_reuse[g] = 0;
    E.hot[g].root_base = 0;
}

template <class G>
__global__ void __launch_bounds__(256) k_reset_games(EngineDev E) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.B) return;
    dev_start_game<G>(E, g, 0);
    if (E.stagger > 0) {  // bench only: pre-advance by (g % stagger) pseudo-random legal plies
        u64 own = E.g_own[g], opp = E.g_opp[g];
        int tm = E.g_to_move[g], made = E.g_moves[g], want = g % E.stagger;
        bool ok = true;
        for (int k = 0; k < want && ok; ++k) {
            u64 lg = G::legal(own, opp);
            int tv;
            if (G::terminal(own, opp, tm, lg, &tv)) { ok = false; break; }
            if (lg == 0) { u64 t = own; own = opp; opp = t; tm = -tm; continue; }
            int pick = (int)(rng_draw(E.seed ^ 0x5AFEC0DEULL, (u64)g, (u64)k) % (u64)popc64(lg));
            for (int j = 0; j < pick; ++j) lg &= lg - 1;
            u64 c0, c1;
            G::apply(own, opp, ctz64(lg), &c0, &c1);
            own = c0; opp = c1; tm = -tm; made++;
        }
        u64 lg = G::legal(own, opp);
        int tv;
        if (ok && !G::terminal(own, opp, tm, lg, &tv)) {
            if (lg == 0) { u64 t = own; own = opp; opp = t; tm = -tm; }
            E.g_own[g] = own; E.g_opp[g] = opp; E.g_to_move[g] = (int8_t)tm; E.g_moves[g] = made;
        }
    }
    for (int r = 0; r < E.rounds; ++r) { E.ex_len[(size_t)r * E.B + g] = -1; E.ex_winner[(size_t)r * E.B + g] = 0; }
    if (g == 0) { E.flags[FLAG_ERR] = 0; E.flags[FLAG_FINISHED] = 0; E.flags[FLAG_NEVAL] = 0; E.flags[FLAG_NEVAL + 1] = 0; }
}

__global__ void __launch_bounds__(256) k_set_roots(EngineDev E, const u64* own, const u64* opp, const int8_t* tm) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.B) return;
    E.g_own[g] = own[g]; E.g_opp[g] = opp[g]; E.g_to_move[g] = tm[g];
    E.g_state[g] = tm[g] == 0 ? 1 : 0;  // to_move 0 = slot not in use (the arena searches a subset of its games)
    if (E.reuse) E.g_reuse[g] = 0;
    E.hot[g].root_base = 0;
    E.g_moves[g] = 0; E.g_nex[g] = 0; E.g_round[g] = 0; E.g_passes[g] = 0;
    if (g == 0) { E.flags[FLAG_ERR] = 0; E.flags[FLAG_FINISHED] = 0; E.flags[FLAG_NEVAL] = 0; E.flags[FLAG_NEVAL + 1] = 0; }
}

// Subtree reuse: copy the subtree below node `src_root` of the previous move's arena to the front of
// this move's arena, breadth first (Cheney): node 0 = the new root, a node's edges stay one contiguous block in
// ascending action order, child ids and edge0 are rewritten.  One lane per game; a few hundred nodes per move.
// (With noise on, k_root_noise then makes a fresh Dirichlet draw on the kept root's stored priors, as for a new root.)
template <class G>
__device__ __forceinline__ void dev_reroot(const EngineDev& E, int g, u32 src_root) {
    const Node* sn = E.nodes_alt + (size_t)g * E.ncap;
    const Edge* se = E.edges_alt + (size_t)g * E.ecap;
    Node* dn = E.nodes + (size_t)g * E.ncap;
    Edge* de = E.edges + (size_t)g * E.ecap;
    u32 n_dst = 1, e_dst = 0;
    dn[0] = sn[src_root];
    // nodes are laid out in breadth-first order and so are their edge blocks: the first edge of the node that gets
    // id k is the number of edges of all nodes before it, known the moment k is handed out
    u32 e_next = dn[0].info & 0xFFu;
    E.hot[g].root_n = e_next;
    for (u32 i = 0; i < n_dst; ++i) {
        Node nd = dn[i];  // edge0 still points into the source arena
        if (nd.info & kTerm) continue;
        const u32 n = nd.info & 0xFFu, s0 = nd.edge0;
        for (u32 j = 0; j < n; ++j) {
            Edge e = se[s0 + j];
            const u32 child = e_child(e.w3);
            if (child) {
                const u32 id = n_dst++;
                const Node cn = sn[child];
                dn[id] = cn;
                e.w3 = id | (e_next << kChildBits);
                e_next += cn.info & 0xFFu;  // (0 for terminal / not yet expanded nodes)
            }
            de[e_dst + j] = e;
        }
        dn[i].edge0 = e_dst;
        e_dst += n;
    }
    E.hot[g].n_nodes = n_dst; E.hot[g].n_edges = e_dst;
}

template <class G>
__global__ void __launch_bounds__(256) k_root_begin(EngineDev E, u32 tt_gen, int carry_ok) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.B) return;
    uint8_t kind = LEAF_NONE;
    if (E.ecache) {  // entries older than the previous search are free slots from now on
        const u32 gen_prev = tt_gen == 1u ? kTtGenMax - 1u : tt_gen - 1u;
        // carry-over: the other arena holds this slot's previous tree only if the slot took part in the previous search
        const bool carry = E.ecache == 2 && carry_ok && E.hot[g].last_gen == gen_prev;
        E.hot[g].prev_nodes = carry ? E.hot[g].n_nodes : 0u;
        E.hot[g].tt_gen = tt_gen;
        if (E.g_state[g] == 0) E.hot[g].last_gen = tt_gen;
    }
    const u32 keep = (E.reuse && E.g_state[g] == 0) ? E.g_reuse[g] : 0u;
    if (keep) {
        dev_reroot<G>(E, g, keep);
        kind = LEAF_READY;
        E.hot[g].leaf_node = 0; E.hot[g].depth = 0;
    } else if (E.g_state[g] == 0 && dev_root_init<G>(E, g)) {
        kind = LEAF_EVAL;
        E.hot[g].n_nodes = 1; E.hot[g].n_edges = 0; E.hot[g].leaf_node = 0; E.hot[g].depth = 0;
        E.hot[g].root_base = 0;
    }
    E.leaf_own[g] = E.g_own[g]; E.leaf_opp[g] = E.g_opp[g];
    E.leaf_kind[g] = kind;
    if (E.compact && kind == LEAF_EVAL) {
        u32 slot = atomicAdd(&E.flags[FLAG_NEVAL + 1], 1u);
        E.hot[g].leaf_slot = slot; E.c_own[slot] = E.g_own[g]; E.c_opp[slot] = E.g_opp[g];
    }
    if (g == 0) E.flags[FLAG_NEVAL] = 0;
}

// Dirichlet root noise, its own small kernel so that the sampler's registers stay out of the tree
// kernels (inlined into the expansion it took k_tree_step from 78 to 129 VGPRs and the cfg-2 kernel from 93 to 219):
// P' = (1 - eps) P + eps g / sum(g) over the root's edges in ascending action order, for every active game whose
// root is expanded -- freshly (by the expand-only tree step before this launch) or kept from the previous move.
__global__ void __launch_bounds__(64) k_root_noise(EngineDev E) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.B || E.g_state[g] != 0) return;
    const Node r = E.nodes[(size_t)g * E.ncap];
    const int n = (int)(r.info & 0xFFu);
    if (n == 0 || r.legal == 0) return;  // not expanded / a forced-pass root (P = 1 stays)
    Edge* ed = E.edges + (size_t)g * E.ecap + r.edge0;
    const u64 gid = E.id_base + (u64)E.g_round[g] * E.id_stride + (u64)g, ply = (u64)E.g_moves[g];
    float gs = 0.0f;
    for (int i = 0; i < n; ++i) gs = gs + gamma_spec(E.dir_alpha, E.seed, gid, ply, i);
    if (!(gs > 0.0f)) return;
    const float keep = 1.0f - E.dir_eps;
    for (int i = 0; i < n; ++i) {
        float t1 = keep * ed[i].P;
        float t2 = E.dir_eps * fdiv(gamma_spec(E.dir_alpha, E.seed, gid, ply, i), gs);
        ed[i].P = t1 + t2;
    }
}

// synthetic evaluators as a separate step (used by the step-by-step API)
template <class G>
__global__ void __launch_bounds__(256) k_eval_synth(EngineDev E, int eval_kind) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.B || E.leaf_kind[g] != LEAF_EVAL) return;
    u64 h = hash_pos(E.leaf_own[g], E.leaf_opp[g]);
    float* row = E.logits + (size_t)g * G::NA;
    for (int a = 0; a < G::NA; ++a) row[a] = eval_kind == BZ_EVAL_HASH ? hash_logit(h, a) : 0.0f;
    E.value[g] = eval_kind == BZ_EVAL_HASH ? hash_value(h) : 0.0f;
}

// One tree step for every game (16 lanes per game): [expand + backup of the previous leaf] and/or
// [select of the next leaf].  With a net in the loop a simulation is exactly two launches:
// this kernel and the net.  The kernel is a dependent-access chain per game (every game of the launch is
// in flight at once), so its time is the number of memory round trips on the longest chain: one for all
// per-game words and the path (independent loads), one for the evaluator's row, then one per level of the walk.
// keeps a loaded value "used" at this point of the program: the loads above it cannot be sunk into the branches
// that consume them, so they go out back to back and complete in ONE memory round trip
template <class T> __device__ __forceinline__ void pin(T& x) { asm volatile("" : "+v"(x)); }

template <class G>
__global__ void __launch_bounds__(256) k_tree_step(EngineDev E, int do_expand, int do_select, u32 sim_idx) {
    constexpr int kGW = G::GW;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = t / kGW, sub = t % kGW;
    Cnt c = {};
    Stamps st; st.start();
    // packed-leaf slots: ONE atomic on the shared counter per workgroup.  Same-address atomics are served one after the
    // other (~10 ns each): one per wave -- 1024 per launch at 4096 games -- was half of this kernel's time.
    __shared__ u32 s_need, s_base;
    if (threadIdx.x == 0) s_need = 0;
    __syncthreads();
    bool want_slot = false; u32 my_rank = 0; u64 slot_own = 0, slot_opp = 0;
    if (g < E.B) {
        PathEnt* path = E.path + (size_t)g * E.maxd;
        // ---- round trip 1: every per-game word this step can need + the first kGW path entries, all independent
        int kind = E.leaf_kind[g], state = E.g_state[g];
        const GameHot hot = E.hot[g];
        u32 leaf = hot.leaf_node, ninfo = hot.leaf_info, ne = hot.n_edges, nn = hot.n_nodes, row = E.compact ? hot.leaf_slot : (u32)g;
        int depth0 = (int)hot.depth, root_n = (int)hot.root_n, rtm = E.g_to_move[g];
        u64 nlegal = hot.leaf_legal, rown = E.g_own[g], ropp = E.g_opp[g];
        u32 root_base = E.reuse ? hot.root_base : 0u;
        PathEnt pe0 = path[sub];  // (maxd >= kGW for every game)
        // the walk's first load too: root edges 0..kGW-1 (the root's edges start at index 0; whatever this step's
        // backup / expansion changes in them is patched in registers below) -- their latency hides behind the
        // evaluator row's round trip and the softmax instead of heading the walk
        Edge* const edges_g = E.edges + (size_t)g * E.ecap;
        Edge re0 = edges_g[sub];
        pin(re0.w0); pin(re0.W); pin(re0.P); pin(re0.w3);
        bool pre_ok = true;
        pin(kind); pin(state); pin(leaf); pin(ninfo); pin(ne); pin(nn); pin(row); pin(depth0); pin(root_n); pin(rtm);
        pin(nlegal); pin(rown); pin(ropp); pin(root_base); pin(pe0.eidx); pin(pe0.w0); pin(pe0.W);
        const bool active = state == 0 && kind != LEAF_NONE;
        st.mark(0);
        if (do_expand && (kind == LEAF_EVAL || kind == LEAF_TERMINAL || kind == LEAF_COPY)) {
            float v;
            u32 e0 = ne; int n = 0;
            if (kind == LEAF_EVAL) {  // ---- round trip 2: the evaluator's row
                LogitSrc ls; ls.kind = BZ_EVAL_EXTERNAL; ls.h = 0; ls.row = E.logits + (size_t)row * G::NA;
                v = E.value[row];
                if (sub == 0 && !(v >= -3.0e38f && v <= 3.0e38f)) atomicOr(&E.flags[FLAG_ERR], ERR_EVAL_NONFINITE);
                n = dev_expand<G>(E, g, sub, leaf, nlegal, ninfo, ls, ne, c, st);
                if (sub == 0) {
                    E.hot[g].n_edges = ne; c.v[CNT_NET_LEAVES]++; if (leaf == 0) E.hot[g].root_n = (u32)n;
                    if (E.ecache) E.node_v[(size_t)g * E.ncap + leaf] = v;  // what a later repeat of this position copies
                }
                if (leaf == 0) { root_n = n; pre_ok = false; }  // the root's edges did not exist when re0 was fetched
            } else if (kind == LEAF_COPY) {  // ---- round trip 2: the first evaluation's edges and value (evaluation cache)
                const bool from_prev = (hot.copy_src & kSrcPrev) != 0u;
                v = (from_prev ? E.node_v_alt : E.node_v)[(size_t)g * E.ncap + (hot.copy_src & ~kSrcPrev)];
                n = dev_expand_copy<G>(E, g, sub, leaf, nlegal, ninfo, hot.copy_e0, from_prev, ne, c);
                if (sub == 0) { E.hot[g].n_edges = ne; E.node_v[(size_t)g * E.ncap + leaf] = v; }
            } else {
                v = (float)((int)((ninfo >> 9) & 3u) - 1);
            }
            // backup: path entries 0..kGW-1 are in registers already; deeper walks (rare) load theirs
            Edge* edges = edges_g;
            const int dmax = depth0 < E.maxd ? depth0 : E.maxd;
            uint2 wr = make_uint2(0u, 0u);
            if (sub < dmax) wr = backup_edge(edges, pe0, ((dmax - 1 - sub) & 1) ? v : -v, sub == depth0 - 1, leaf, e0, n);
            for (int d = sub + kGW; d < dmax; d += kGW)
                backup_edge(edges, path[d], ((dmax - 1 - d) & 1) ? v : -v, d == depth0 - 1, leaf, e0, n);
            if (sub == 0) c.v[CNT_EDGES_BACKED] += (u32)dmax;
            if (dmax >= 1) {  // the path's root edge (lane 0 wrote it): the same words into the lane that holds it in re0
                const u32 pe_e = (u32)__shfl((int)pe0.eidx, 0, kGW), w0n = (u32)__shfl((int)wr.x, 0, kGW), Wn = (u32)__shfl((int)wr.y, 0, kGW);
                if ((u32)sub == pe_e) {
                    re0.w0 = w0n; re0.W = __uint_as_float(Wn);
                    if (depth0 == 1 && n > 0) re0.w3 = leaf | (e0 << kChildBits);
                }
            }
        }
        st.mark(2);
        if (do_select) {
            uint8_t kind8 = LEAF_NONE;
            if (active) {
                group_fence();  // edges written above are read below
                u32 leaf2; int k2, depth; float tv;
                LeafPos lpos; lpos.own = 0; lpos.opp = 0; lpos.legal = 0; lpos.info = 0; lpos.src = 0; lpos.src_e0 = 0;
                RootRef root; root.own = rown; root.opp = ropp; root.tm = rtm; root.n = root_n;
                root.sumN = sim_idx + root_base; root.has_pre = pre_ok; root.pre = re0; root.tt_gen = hot.tt_gen; root.prev_nodes = hot.prev_nodes;
                PathHbm<kGW> sink; sink.p = path; sink.mine.eidx = 0; sink.mine.w0 = 0; sink.mine.W = 0.0f; sink.mine.pad = 0;
                dev_select<G>(E, g, sub, root, nn, leaf2, k2, depth, tv, c, sink, lpos, st);  // ---- one round trip per level
                sink.flush(sub, depth);
                if (sub == 0) {
                    E.hot[g].n_nodes = nn; E.hot[g].leaf_node = leaf2; E.hot[g].depth = (u32)depth;
                    if (k2 == LEAF_EVAL) {  // only an evaluated leaf's position is consumed (evaluator input)
                        E.leaf_own[g] = lpos.own; E.leaf_opp[g] = lpos.opp; E.hot[g].leaf_legal = lpos.legal; E.hot[g].leaf_info = lpos.info;
                        if (E.compact) { want_slot = true; my_rank = atomicAdd(&s_need, 1u); slot_own = lpos.own; slot_opp = lpos.opp; }
                    } else if (k2 == LEAF_COPY) {  // its evaluation exists already: no evaluator row, the next step copies it
                        E.hot[g].leaf_legal = lpos.legal; E.hot[g].leaf_info = lpos.info;
                        E.hot[g].copy_src = lpos.src; E.hot[g].copy_e0 = lpos.src_e0;
                    } else {  // terminal leaf (new or revisited): the backup needs its value only
                        E.hot[g].leaf_info = (u32)((int)tv + 1) << 9;
                    }
                }
                kind8 = (uint8_t)k2;
            }
            if (sub == 0) E.leaf_kind[g] = kind8;
            if (t == 0) E.flags[FLAG_NEVAL + ((sim_idx + 1u) & 1u)] = 0;  // the buffer the NEXT select packs into
        } else if (t == 0) {  // expand-only step (end of a search / step API): nothing is packed any more
            E.flags[FLAG_NEVAL] = 0; E.flags[FLAG_NEVAL + 1] = 0;
        }
    }
    if (do_select && E.compact) {  // (kernel arguments: the whole grid takes this branch or none of it)
        __syncthreads();
        if (threadIdx.x == 0) s_base = s_need ? atomicAdd(&E.flags[FLAG_NEVAL + (sim_idx & 1u)], s_need) : 0u;
        __syncthreads();
        if (want_slot) {
            const u32 slot = s_base + my_rank;
            E.hot[g].leaf_slot = slot; E.c_own[slot] = slot_own; E.c_opp[slot] = slot_opp;
        }
    }
    st.mark(5);
    cnt_flush<G::GW>(E, c);
    st.mark(6);
    st.flush(E.counters);
}

// whole search in one launch for the synthetic evaluators (BASELINE cfg 2):
// root expansion + sims x (select, expand, backup), no host round trip.  The dependent-access
// chain per simulation is what bounds this kernel (every game of the batch is in flight at once,
// one lane group per game), so it is kept as short as the data structure allows: the select path
// lives in LDS together with the N and W it saw (backup = stores only), the created leaf's position
// and legal mask come back from the walk in registers (expansion reads nothing), and the group's
// store -> load ordering is a wavefront-scope fence (no wait for store round trips).
template <class G>
__global__ void __launch_bounds__(256) k_search_fused(EngineDev E, int eval_kind) {
    constexpr int kGW = G::GW, kGPB = 256 / kGW;
    __shared__ PathEntLds s_path[kGPB][G::MAXD];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = t / kGW, sub = t % kGW;
    PathEntLds* mypath = s_path[threadIdx.x / kGW];
    Cnt c = {};
    if (g < E.B && E.g_state[g] == 0) {
        bool ok = true;
        if (sub == 0) ok = dev_root_init<G>(E, g);
        ok = __shfl((int)ok, 0, kGW) != 0;
        if (ok) {
            u32 nn = 1, ne = 0;
            LogitSrc ls; ls.kind = eval_kind; ls.row = nullptr;
            RootRef root; root.own = E.g_own[g]; root.opp = E.g_opp[g]; root.tm = E.g_to_move[g];
            root.has_pre = false; root.pre.w0 = 0; root.pre.W = 0.0f; root.pre.P = 0.0f; root.pre.w3 = 0;
            ls.h = hash_pos(root.own, root.opp);
            group_fence();
            Stamps st; st.start();
            root.n = dev_expand<G>(E, g, sub, 0, G::legal(root.own, root.opp), (root.tm == 1 ? 1u : 0u) << 11, ls, ne, c, st);
            if (sub == 0) E.hot[g].root_n = (u32)root.n;
            PathLds sink{mypath};
            for (int s = 0; s < E.sims; ++s) {
                group_fence();  // this group's stores -> its loads
                u32 leaf; int kind, depth; float v;
                LeafPos lp; lp.own = 0; lp.opp = 0; lp.legal = 0; lp.info = 0;
                root.sumN = (u32)s;
                dev_select<G>(E, g, sub, root, nn, leaf, kind, depth, v, c, sink, lp, st);
                u32 e0 = ne; int n = 0;
                if (kind == LEAF_EVAL) {
                    ls.h = hash_pos(lp.own, lp.opp);
                    n = dev_expand<G>(E, g, sub, leaf, lp.legal, lp.info, ls, ne, c, st);
                    v = eval_kind == BZ_EVAL_HASH ? hash_value(ls.h) : 0.0f;
                }
                group_fence();  // LDS path entries (lane 0) -> the lanes that back them up
                dev_backup<G::GW>(E, g, sub, depth, v, mypath, leaf, e0, n, c);
            }
            if (sub == 0) { E.hot[g].n_nodes = nn; E.hot[g].n_edges = ne; }
        }
    }
    cnt_flush<G::GW>(E, c);
}

// ---- Tic-tac-toe specialisation of the fused search (BASELINE cfg 2; sims <= 120).
// The walk is a dependent chain per game and every game of the batch is in flight at once, so the
// time of a launch is sims x (dependent memory round trips per simulation).  Three things cut the chain:
//  * the root's edges live in REGISTERS for the whole search (lane sub holds edges sub, sub+GW, ...): level 0
//    of every simulation reads nothing, its backup writes nothing; the edges go to HBM once, at the end;
//  * an edge's `ca` word also carries what the walk needs to know about the child -- its first edge, its
//    child count, terminal flag and value -- so a level is ONE load (the child's edges), never the child's
//    node first (TTT's numbers fit in ONE word: child 8 | edge0 12 | n 4 | action 4 | terminal 1 | value 2 in w3,
//    with w0 a plain visit count -- this kernel's private edge record, EdgeFmtTttFused);
//  * positions are carried down the walk by applying the actions (TTT: swap + one bit), so nodes below the
//    root are never loaded -- and, since nothing reads them, never stored.
// Results (root statistics, counters, examples) are bit-identical to the generic kernel and the oracle; the
// root edges are written back as the engine's packed record, which is all that k_play / k_root_stats read.
__device__ __forceinline__ u32 ttt_ca(u32 child, u32 edge0, u32 n, u32 act, bool term, int tv) {
    return child | (edge0 << 8) | (n << 20) | (act << 24) | ((term ? 1u : 0u) << 28) | ((u32)(tv + 1) << 29);
}
constexpr int kTttFusedMaxSims = 120;

__device__ __forceinline__ float puct_score(const Edge& e, float c_puct, float sq) {
    float q = e.w0 > 0 ? fdiv(e.W, (float)e.w0) : 0.0f;
    float u = c_puct * e.P;
    u = u * sq;
    u = fdiv(u, 1.0f + (float)e.w0);
    return q + u;
}

// kUniform = the uniform evaluator (cfg 2) at compile time: the kernel is at ~90 % of its SIMDs' issue slots, so the
// exponentials, the serial softmax sum and the position hash it does not need are time, not just instructions.
template <int kGW, bool kUniform>
__global__ void __launch_bounds__(256) k_search_fused_ttt(EngineDev E) {
    using G = TicTacToe;
    constexpr int eval_kind = kUniform ? BZ_EVAL_UNIFORM : BZ_EVAL_HASH;
    constexpr int kGPB = 256 / kGW, kCH = (G::MAXCH + kGW - 1) / kGW;
    __shared__ PathEntLds s_path[kGPB][G::MAXD];
    // sqrt(max(visits, 1)) for every visit count this kernel can see (sims <= 120): the correctly rounded square root is
    // ~20 instructions, the table one LDS read -- and holds the very values fsqrt() returns
    __shared__ float s_sqrt[128];
    if (threadIdx.x < 128) s_sqrt[threadIdx.x] = fsqrt((float)(threadIdx.x > 1 ? threadIdx.x : 1));
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = t / kGW, sub = t % kGW;
    PathEntLds* mypath = s_path[threadIdx.x / kGW];
    Cnt c = {};
    if (g < E.B && E.g_state[g] == 0) {
        bool ok = true;
        if (sub == 0) ok = dev_root_init<G>(E, g);
        ok = __shfl((int)ok, 0, kGW) != 0;
        if (ok) {
            Edge* edges = E.edges + (size_t)g * E.ecap;
            const bool lead = sub == 0;
            u32 nn = 1, ne = 0;
            LogitSrc ls; ls.kind = eval_kind; ls.row = nullptr;
            const u64 rown = E.g_own[g], ropp = E.g_opp[g];
            const int rtm = E.g_to_move[g];
            ls.h = kUniform ? 0 : hash_pos(rown, ropp);
            group_fence();
            Stamps st; st.start();
            dev_expand<G, kGW, true, EdgeFmtTttFused, kUniform>(E, g, sub, 0, G::legal(rown, ropp), (rtm == 1 ? 1u : 0u) << 11, ls, ne, c, st);
            group_fence();
            const int rn = (int)ne;  // the root's children (>= 1: a terminal root was refused above)
            Edge re[kCH];
            // Q = W / N and c_puct * P of the register-resident root edges are kept next to them: Q changes only for the
            // edge a simulation backs up (one correctly rounded division per simulation instead of one per root edge),
            // c_puct * P never.  The values are the ones puct_score would recompute, bit for bit.
            float rq[kCH], rcp[kCH];
#pragma unroll
            for (int k = 0; k < kCH; ++k) {
                const int i = sub + kGW * k;
                if (i < rn) re[k] = edges[i];
                else { re[k].w0 = 0; re[k].W = 0.0f; re[k].P = 0.0f; re[k].w3 = 0; }
                rq[k] = re[k].w0 > 0 ? fdiv(re[k].W, (float)re[k].w0) : 0.0f;
                rcp[k] = E.c_puct * re[k].P;
            }
            for (int s = 0; s < E.sims; ++s) {
                group_fence();  // this group's stores of the previous simulation -> its loads
                // ---- level 0 from registers
                // (first maximum = highest score, lowest child index on ties: every lane scans its own edges in ascending
                // order, then ONE exchange across the group's lanes -- the order of the comparisons does not matter)
                float bestW = 0.0f; int best = 0; u32 bestN = 0, bestca = 0;
                {
                    const float sq = s_sqrt[s];
                    Cand lb; lb.sc = -__builtin_inff(); lb.i = sub; lb.w0 = 0; lb.w3 = 0; lb.W = 0.0f;
#pragma unroll
                    for (int k = 0; k < kCH; ++k) {
                        const int i = sub + kGW * k;
                        if (i < rn) {  // puct_score with the cached terms
                            float u = rcp[k] * sq;
                            u = fdiv(u, 1.0f + (float)re[k].w0);
                            const float sc = rq[k] + u;
                            if (sc > lb.sc) { lb.sc = sc; lb.i = i; lb.w0 = re[k].w0; lb.w3 = re[k].w3; lb.W = re[k].W; }
                        }
                    }
                    group_argmax<kGW>(lb);
                    best = lb.i; bestN = lb.w0; bestca = lb.w3; bestW = lb.W;
                }
                if (lead) { c.v[CNT_SIMS]++; c.v[CNT_PATH_NODES]++; c.v[CNT_CHILD_SCORED] += (u32)rn; }
                const int i0 = best; const u32 N0 = bestN; const float W0 = bestW;
                int depth = 1;
                u64 own = rown, opp = ropp; int tm = rtm;
                u32 ca = bestca, parentN = bestN, pe_idx = (u32)best, new_ca = 0;
                bool made = false;
                float v = 0.0f;
                for (;;) {  // walk below the root: `ca` is the edge just chosen
                    const int act = (int)((ca >> 24) & 0xFu);
                    u64 cown, copp;
                    G::apply(own, opp, act, &cown, &copp);
                    own = cown; opp = copp; tm = -tm;
                    if (ca & 0xFFu) {  // the child exists
                        if (lead) c.v[CNT_PATH_NODES]++;
                        if ((ca >> 28) & 1u) { v = (float)((int)((ca >> 29) & 3u) - 1); break; }  // terminal, revisited
                        const u32 e0 = (ca >> 8) & 0xFFFu; const int n = (int)((ca >> 20) & 0xFu);
                        const u32 sumN = parentN - 1u;  // visits of the child's edges = visits into it minus the creating one
                        const float sq = s_sqrt[sumN & 127u];
                        const Edge* ed = edges + e0;
                        Cand lb; lb.sc = -__builtin_inff(); lb.i = sub; lb.w0 = 0; lb.w3 = 0; lb.W = 0.0f;
#pragma unroll
                        for (int k = 0; k < kCH; ++k) {
                            const int i = sub + kGW * k;
                            if (i < n) {
                                const Edge e = ed[i];
                                const float sc = puct_score(e, E.c_puct, sq);
                                if (sc > lb.sc) { lb.sc = sc; lb.i = i; lb.w0 = e.w0; lb.w3 = e.w3; lb.W = e.W; }
                            }
                        }
                        group_argmax<kGW>(lb);
                        best = lb.i; bestN = lb.w0; bestca = lb.w3; bestW = lb.W;
                        pe_idx = e0 + (u32)best;
                        if (lead) {
                            c.v[CNT_CHILD_SCORED] += (u32)n;
                            if (depth < E.maxd) { PathEntLds pe; pe.eidx = pe_idx; pe.w0 = bestN; pe.W = bestW; mypath[depth] = pe; }
                            else atomicOr(&E.flags[FLAG_ERR], ERR_DEPTH);
                        }
                        depth++;
                        ca = bestca; parentN = bestN;
                        continue;
                    }
                    // create the child behind the chosen edge (env step) and, unless terminal, expand it at once
                    const u32 id = nn++;
                    const u64 lg = G::legal(own, opp);
                    int tv = 0;
                    const bool term = G::terminal(own, opp, tm, lg, &tv);
                    if (lead) { c.v[CNT_ENV_STEPS]++; c.v[CNT_PATH_NODES]++; }
                    if (term) {
                        new_ca = ttt_ca(id, 0, 0, (u32)act, true, tv);
                        v = (float)tv;
                    } else {
                        ls.h = kUniform ? 0 : hash_pos(own, opp);
                        const u32 e0 = ne;
                        dev_expand<G, kGW, false, EdgeFmtTttFused, kUniform>(E, g, sub, id, lg, 0, ls, ne, c, st);
                        new_ca = ttt_ca(id, e0, ne - e0, (u32)act, false, 0);
                        v = eval_kind == BZ_EVAL_HASH ? hash_value(ls.h) : 0.0f;
                    }
                    made = true;
                    break;
                }
                if (made && depth > 1 && lead) edges[pe_idx].w3 = new_ca;  // (depth 1: the parent edge is a root register)
                // ---- backup: root edge in registers, deeper edges by one 8-byte store each from the LDS path
                const int dmax = depth < E.maxd ? depth : E.maxd;
                {
                    const float val0 = ((dmax - 1) & 1) ? v : -v;
#pragma unroll
                    for (int k = 0; k < kCH; ++k)
                        if (sub + kGW * k == i0) {
                            re[k].w0 = N0 + 1u; re[k].W = W0 + val0;
                            rq[k] = fdiv(re[k].W, (float)re[k].w0);
                            if (made && depth == 1) re[k].w3 = new_ca;
                        }
                }
                group_fence();  // the lead lane's LDS path entries -> the lanes that back them up
                for (int d = sub; d < dmax; d += kGW) {
                    if (d == 0) continue;
                    PathEntLds pe = mypath[d];
                    float val = ((dmax - 1 - d) & 1) ? v : -v;
                    *reinterpret_cast<uint2*>(edges + pe.eidx) = make_uint2(pe.w0 + 1u, __float_as_uint(pe.W + val));
                }
                if (lead) c.v[CNT_EDGES_BACKED] += (u32)dmax;
            }
            // root edges back to HBM as the engine's packed record (N | action, child id): what k_play / k_root_stats read
#pragma unroll
            for (int k = 0; k < kCH; ++k) {
                const int i = sub + kGW * k;
                if (i < rn) { Edge e = re[k]; e.w0 = e.w0 | (((e.w3 >> 24) & 0xFu) << kActShift); e.w3 = e.w3 & 0xFFu; edges[i] = e; }
            }
            if (lead) { E.hot[g].n_nodes = nn; E.hot[g].n_edges = ne; E.hot[g].root_n = (u32)rn; }
        }
    }
    cnt_flush<kGW>(E, c);
}

template <class G>
__global__ void __launch_bounds__(256) k_root_stats(EngineDev E) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.B) return;
    u32* rn = E.root_N + (size_t)g * G::NA; float* rw = E.root_W + (size_t)g * G::NA; float* rp = E.root_P + (size_t)g * G::NA;
    for (int a = 0; a < G::NA; ++a) { rn[a] = 0; rw[a] = 0.0f; rp[a] = 0.0f; }
    if (E.g_state[g] != 0) return;
    Node r = E.nodes[(size_t)g * E.ncap];
    const Edge* ed = E.edges + (size_t)g * E.ecap + r.edge0;
    for (int i = 0; i < (int)(r.info & 0xFFu); ++i) {
        Edge e = ed[i];
        int a = e_action(e.w0);
        rn[a] = e_N(e.w0); rw[a] = e.W; rp[a] = e.P;
    }
}

// M5 + the reference's turn loop: pi, move choice, example row, env step,
// pass rule (reversi_terminal.py:31-35), terminal handling, z back-fill.
template <class G>
__global__ void __launch_bounds__(256) k_play(EngineDev E, int restart) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.B || E.g_state[g] != 0) return;
    Node root = E.nodes[(size_t)g * E.ncap];
    const Edge* ed = E.edges + (size_t)g * E.ecap + root.edge0;
    int n = (int)(root.info & 0xFFu);
    u32 sumN = 0;
    for (int i = 0; i < n; ++i) sumN += e_N(ed[i].w0);
    int round = E.g_round[g], nex = E.g_nex[g], made = E.g_moves[g], tm = E.g_to_move[g];
    size_t rowbase = ((size_t)round * E.B + g) * E.t_max;
    if (nex >= E.t_max) { atomicOr(&E.flags[FLAG_ERR], ERR_EXAMPLE_OVERFLOW); E.g_state[g] = 1; return; }
    size_t row = rowbase + nex;
    float* pi = E.ex_pi + row * G::NA;
    for (int a = 0; a < G::NA; ++a) pi[a] = 0.0f;
    int pick = 0;
    if (made < E.temp_moves) {  // tau = 1: sample ~ N with the counter RNG (seed, game id, moves made)
        u64 gid = E.id_base + (u64)round * E.id_stride + (u64)g;
        u64 rr = rng_draw(E.seed, gid, (u64)made) % (u64)sumN, cum = 0;
        bool found = false;
        for (int i = 0; i < n; ++i) {
            const u32 w0 = ed[i].w0, N = e_N(w0);
            pi[e_action(w0)] = fdiv((float)N, (float)sumN);
            cum += N;
            if (!found && cum > rr) { pick = i; found = true; }
        }
    } else {  // tau = 0: argmax N, ties -> lowest action
        u32 bn = 0;
        for (int i = 0; i < n; ++i) {
            const u32 w0 = ed[i].w0, N = e_N(w0);
            pi[e_action(w0)] = fdiv((float)N, (float)sumN);
            if (N > bn) { bn = N; pick = i; }
        }
    }
    const Edge pk = ed[pick];
    int a = e_action(pk.w0);
    u32 keep_node = e_child(pk.w3), keep_N = e_N(pk.w0);  // subtree reuse: the chosen child and its visits
    E.ex_own[row] = root.own; E.ex_opp[row] = root.opp; E.ex_mover[row] = (int8_t)tm; E.ex_act[row] = (uint8_t)a;
    nex++;
    u64 own, opp;
    G::apply(root.own, root.opp, a, &own, &opp);
    tm = -tm; made++;
    u64 lg = G::legal(own, opp);
    int tv;
    if (G::terminal(own, opp, tm, lg, &tv)) {
        int w = tv * tm;
        for (int t = 0; t < nex; ++t) E.ex_z[rowbase + t] = (int8_t)(w * E.ex_mover[rowbase + t]);
        E.ex_len[(size_t)round * E.B + g] = nex;
        E.ex_winner[(size_t)round * E.B + g] = (int8_t)w;
        atomicAdd(&E.flags[FLAG_FINISHED], 1u);
        if (restart && round + 1 < E.rounds) { dev_start_game<G>(E, g, round + 1); return; }
        E.g_state[g] = 1;
    } else if (lg == 0) {  // the next mover cannot move: flip the side again
        u64 t = own; own = opp; opp = t; tm = -tm;
        E.g_passes[g]++;
        if (E.reuse && keep_node) {  // the kept root lies behind the child's only edge, the pass
            const Node cn = E.nodes[(size_t)g * E.ncap + keep_node];
            const Edge pe = E.edges[(size_t)g * E.ecap + cn.edge0];
            keep_node = e_child(pe.w3); keep_N = e_N(pe.w0);
        }
    }
    if (E.reuse) {  // keep the subtree iff it exists and the next search cannot outgrow the arena
        const bool ok = keep_node != 0 && keep_N + (u32)E.sims + 2u <= (u32)E.ncap;
        E.g_reuse[g] = ok ? keep_node : 0u;
        E.hot[g].root_base = ok ? keep_N - 1u : 0u;
    }
    E.g_own[g] = own; E.g_opp[g] = opp; E.g_to_move[g] = (int8_t)tm; E.g_moves[g] = made; E.g_nex[g] = nex;
}

__global__ void __launch_bounds__(256) k_count_active(EngineDev E) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    u32 act = (g < E.B && E.g_state[g] == 0) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) act += __shfl_xor(act, o, 64);
    if ((threadIdx.x & 63) == 0 && act) atomicAdd(&E.flags[FLAG_ACTIVE], act);
}


// ---- packed example block (bz_abi.h): the rows of the FINISHED games only, compacted in (round, slot, ply) order
// behind a 256-byte header -- what the iteration-end all-gather ships.  Two launches: a one-workgroup scan of the
// games' row counts (first row of every finished game -> pack_off), then one wave per game copies its rows.
struct PackedHdr {
    u64 magic, n_rows, n_games, cap_rows, na, game, dropped_rows, bytes;
    u64 offs[8];  // own, opp, pi, game id, z, mover, act, ply: byte offsets from the start of the block
    u64 pad[16];
};
static_assert(sizeof(PackedHdr) == 256, "packed example header");
constexpr u64 kPackedMagic = 0x425A50414B000001ULL;  // "BZPAK" + layout version 1
struct PackedLayout { int64_t offs[8]; int64_t total; };

__global__ void __launch_bounds__(1024) k_pack_scan(EngineDev E, PackedHdr* hdr, PackedLayout L, u64 cap, int append, int game) {
    __shared__ u32 wsum[16];
    __shared__ u32 s_games, s_rows_end;
    const int n = E.rounds * E.B, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool bad = false;
    u32 carry = 0, games0 = 0;
    u64 dropped0 = 0;
    if (append) {  // every thread reads the header before thread 0 rewrites it (barriers below)
        bad = hdr->magic != kPackedMagic || hdr->cap_rows != cap || hdr->na != (u64)E.na || hdr->bytes != (u64)L.total ||
              hdr->game != (u64)game;
        carry = (u32)hdr->n_rows; games0 = (u32)hdr->n_games; dropped0 = hdr->dropped_rows;
    }
    if (threadIdx.x == 0) { s_games = 0; s_rows_end = carry; }
    __syncthreads();
    u32 total_all = carry;
    for (int c0 = 0; c0 < n; c0 += 1024) {
        const int i = c0 + (int)threadIdx.x;
        const int len = i < n ? E.ex_len[i] : -1;
        const u32 v = len > 0 ? (u32)len : 0u;
        u32 x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const u32 y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        u32 woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const u32 sw = wsum[w]; tot += sw; if (w < wave) woff += sw; }
        const u32 first = total_all + woff + x - v;
        if (i < n) {
            const bool fits = len >= 0 && !bad && (u64)first + v <= cap;
            E.pack_off[i] = fits ? (int32_t)first : -1;
            if (fits) { atomicAdd(&s_games, 1u); atomicMax(&s_rows_end, first + v); }
        }
        total_all += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        hdr->magic = kPackedMagic; hdr->cap_rows = cap; hdr->na = (u64)E.na; hdr->bytes = (u64)L.total;
        hdr->game = (u64)game;
        for (int k = 0; k < 8; ++k) hdr->offs[k] = (u64)L.offs[k];
        hdr->n_rows = s_rows_end; hdr->n_games = games0 + s_games;
        hdr->dropped_rows = bad ? ~0ULL : dropped0 + (u64)(total_all - s_rows_end);
    }
}

__global__ void __launch_bounds__(256) k_pack_rows(EngineDev E, char* blk, PackedLayout L) {
    const int i = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= E.rounds * E.B) return;
    const int off = E.pack_off[i];
    if (off < 0) return;
    const int len = E.ex_len[i], round = i / E.B, g = i - round * E.B;
    const size_t src = (size_t)i * E.t_max;  // [round][slot][ply]
    u64* own = reinterpret_cast<u64*>(blk + L.offs[0]) + off; u64* opp = reinterpret_cast<u64*>(blk + L.offs[1]) + off;
    float* pi = reinterpret_cast<float*>(blk + L.offs[2]) + (size_t)off * E.na;
    int64_t* gid = reinterpret_cast<int64_t*>(blk + L.offs[3]) + off;
    int8_t* z = reinterpret_cast<int8_t*>(blk + L.offs[4]) + off; int8_t* mv = reinterpret_cast<int8_t*>(blk + L.offs[5]) + off;
    uint8_t* act = reinterpret_cast<uint8_t*>(blk + L.offs[6]) + off; uint8_t* ply = reinterpret_cast<uint8_t*>(blk + L.offs[7]) + off;
    const int64_t id = (int64_t)(E.id_base + (u64)round * E.id_stride + (u64)g);
    for (int t = lane; t < len; t += 64) {
        own[t] = E.ex_own[src + t]; opp[t] = E.ex_opp[src + t]; gid[t] = id; z[t] = E.ex_z[src + t];
        mv[t] = E.ex_mover[src + t]; act[t] = E.ex_act[src + t]; ply[t] = (uint8_t)t;
    }
    const float* sp = E.ex_pi + src * E.na;  // a game's rows are consecutive at both ends: one straight copy
    for (int k = lane; k < len * E.na; k += 64) pi[k] = sp[k];
}

}  // namespace

// ---------------------------------------------------------------- host side
struct bz_engine {
    bz_engine_cfg cfg;
    bz_engine_layout lay;
    EngineDev dev;
    bz_net* net;
    int64_t bytes;
    int pack_parity;  // which NEVAL buffer the last root_begin / select packed into
    int ttt_gw;       // lanes per game of the TTT-specialised fused search (cfg.ttt_lanes; 0 = the generic any-game kernel)
    uint32_t search_seq;  // searches begun so far: the evaluation cache's generation (entries of earlier searches are dead)
    uint64_t eval_epoch;  // bz_net_epoch of the weights the previous search evaluated with (carry-over needs the same ones)
    // bz_engines_step: ring of blocking-sync events that bounds how far the host thread runs ahead of this engine's
    // stream (created on first use)
    hipEvent_t ahead[4];
    int n_ahead;
};

namespace {
struct Carver {
    int64_t off = 0;
    int64_t take(int64_t bytes) { int64_t o = off; off += (bytes + 255) & ~int64_t(255); return o; }
};

struct Offsets {
    int64_t nodes, edges, nodes_alt, edges_alt, g_reuse, g_own, g_opp, g_to_move, g_state, g_moves, g_nex, g_round, g_passes, hot,
        path, leaf_kind, leaf_own, leaf_opp, c_own, c_opp, logits, value, ex_own, ex_opp, ex_pi, ex_z, ex_mover,
        ex_act, ex_len, ex_winner, ex_meta, root_N, root_W, root_P, counters, cnt_slots, flags, pack_off, tt, node_v, node_v_alt, total;
    int n_cnt_slots;
    int ncap, ecap, na, maxd;
    int ecache, tt_buckets;
};
inline bool net_eval(int ek) { return ek == BZ_EVAL_NET_F32 || ek == BZ_EVAL_NET_BF16 || ek == BZ_EVAL_NET_FP8; }

// nodes a game's arena holds: a fresh tree grows by one node per simulation; with subtree reuse a kept
// subtree + sims new nodes must fit
inline int64_t nodes_per_game(const bz_engine_cfg& c) { return ((c.flags & BZ_ENGINE_REUSE_SUBTREE) ? 4 : 1) * ((int64_t)c.sims + 2); }

bool cfg_ok(const bz_engine_cfg* c) {
    if (!(c && c->game >= BZ_GAME_TTT && c->game <= BZ_GAME_REVERSI4 && c->n_games > 0 && c->sims >= 1 &&
          c->eval_kind >= 0 && c->eval_kind <= BZ_EVAL_NET_FP8 && c->rounds >= 1 &&
          c->t_max >= 1 && c->dirichlet_eps >= 0.0f && c->dirichlet_eps <= 1.0f &&
          (c->dirichlet_eps == 0.0f || (c->dirichlet_alpha > 0.0f && c->dirichlet_alpha <= 1.0f)) &&
          (c->ttt_lanes == -1 || c->ttt_lanes == 0 || c->ttt_lanes == 1 || c->ttt_lanes == 2 || c->ttt_lanes == 4 ||
           c->ttt_lanes == 8)))
        return false;
    // the packed edge record holds node ids in 13 bits and visit counts in 14 (BZ_ENGINE_MAX_SIMS in bz_abi.h)
    return nodes_per_game(*c) <= kMaxNodes;
}

Offsets carve(const bz_engine_cfg& c) {
    Offsets o{};
    bool ttt = c.game == BZ_GAME_TTT;
    o.na = ttt ? TicTacToe::NA : Reversi::NA;
    o.maxd = ttt ? TicTacToe::MAXD : Reversi::MAXD;
    const bool reuse = (c.flags & BZ_ENGINE_REUSE_SUBTREE) != 0;
    o.ncap = (int)nodes_per_game(c);
    o.ecap = o.ncap * (ttt ? TicTacToe::MAXCH : Reversi::MAXCH);
    int64_t B = c.n_games, R = c.rounds, T = c.t_max;
    Carver k;
    o.nodes = k.take(B * o.ncap * (int64_t)sizeof(Node));
    o.edges = k.take(B * o.ecap * (int64_t)sizeof(Edge));
    // evaluation cache: net evaluators only (a synthetic evaluation costs less than the lookup), not with subtree reuse
    // (a kept subtree's nodes are not in the new search's table).  2 = with carry-over: a second arena, like subtree reuse
    o.ecache = ((c.flags & BZ_ENGINE_EVAL_CACHE) && net_eval(c.eval_kind) && !reuse) ? ((c.flags & BZ_ENGINE_EVAL_CACHE_CARRY) ? 2 : 1) : 0;
    const bool two = reuse || o.ecache == 2;
    o.nodes_alt = k.take(two ? B * o.ncap * (int64_t)sizeof(Node) : 0);
    o.edges_alt = k.take(two ? B * o.ecap * (int64_t)sizeof(Edge) : 0);
    o.g_reuse = k.take(reuse ? B * 4 : 0);
    o.g_own = k.take(B * 8); o.g_opp = k.take(B * 8); o.g_to_move = k.take(B); o.g_state = k.take(B);
    o.g_moves = k.take(B * 4); o.g_nex = k.take(B * 4); o.g_round = k.take(B * 4); o.g_passes = k.take(B * 4);
    o.hot = k.take(B * (int64_t)sizeof(GameHot));
    o.path = k.take((int64_t)o.maxd * B * (int64_t)sizeof(PathEnt));
    o.leaf_kind = k.take(B); o.leaf_own = k.take(B * 8); o.leaf_opp = k.take(B * 8);
    o.c_own = k.take(B * 8); o.c_opp = k.take(B * 8);
    o.logits = k.take(B * o.na * 4); o.value = k.take(B * 4);
    o.ex_own = k.take(R * B * T * 8); o.ex_opp = k.take(R * B * T * 8); o.ex_pi = k.take(R * B * T * o.na * 4);
    o.ex_z = k.take(R * B * T); o.ex_mover = k.take(R * B * T); o.ex_act = k.take(R * B * T);
    o.ex_len = k.take(R * B * 4); o.ex_winner = k.take(R * B);
    // ex_own .. ex_meta are consecutive: ONE byte range [ex_own, ex_meta + 256) is what the all-gather ships
    o.ex_meta = k.take(256);
    o.root_N = k.take(B * o.na * 4); o.root_W = k.take(B * o.na * 4); o.root_P = k.take(B * o.na * 4);
    o.counters = k.take(kCntWords * 8);
    o.n_cnt_slots = (int)((B * 16 + 63) / 64) + 4;  // one slot per wave of the widest (group) launch (<= 16 lanes per game)
    o.cnt_slots = k.take((int64_t)o.n_cnt_slots * CNT_N * 8);
    o.flags = k.take(FLAG_N * 4);
    o.pack_off = k.take(R * B * 4);
    // the table's buckets: a power of two, >= 2 slots per live node (with carry-over two searches' entries are live)
    o.tt_buckets = 16;
    while ((int64_t)o.tt_buckets * 16 < (o.ecache == 2 ? 4 : 2) * (int64_t)o.ncap) o.tt_buckets *= 2;
    o.tt = k.take(o.ecache ? B * o.tt_buckets * 16 * 8 : 0);
    o.node_v = k.take(o.ecache ? B * o.ncap * 4 : 0);
    o.node_v_alt = k.take(o.ecache == 2 ? B * o.ncap * 4 : 0);
    o.total = k.off;
    return o;
}

template <class T> T* at(void* base, int64_t off) { return reinterpret_cast<T*>(static_cast<char*>(base) + off); }
inline dim3 grid_of(int B) { return dim3((B + 255) / 256); }
inline dim3 grid_groups(int B, int gw) { return dim3(((size_t)B * gw + 255) / 256); }
}  // namespace

#define BZ_LAUNCH_ONE(G, KERNEL, grid, stream, ...) \
    hipLaunchKernelGGL(KERNEL<G>, grid, dim3(256), 0, (hipStream_t)(stream), __VA_ARGS__)
#define BZ_DISPATCH_IMPL(e, KERNEL, GRIDFN, stream, ...)                                                        \
    do {                                                                                                        \
        switch ((e)->cfg.game) {                                                                                \
        case BZ_GAME_TTT: BZ_LAUNCH_ONE(TicTacToe, KERNEL, GRIDFN((e)->dev.B, TicTacToe::GW), stream, __VA_ARGS__); break; \
        case BZ_GAME_REVERSI6: BZ_LAUNCH_ONE(Reversi6, KERNEL, GRIDFN((e)->dev.B, Reversi6::GW), stream, __VA_ARGS__); break; \
        case BZ_GAME_REVERSI4: BZ_LAUNCH_ONE(Reversi4, KERNEL, GRIDFN((e)->dev.B, Reversi4::GW), stream, __VA_ARGS__); break; \
        default: BZ_LAUNCH_ONE(Reversi, KERNEL, GRIDFN((e)->dev.B, Reversi::GW), stream, __VA_ARGS__); break;      \
        }                                                                                                       \
        BZ_LAUNCH_CHECK(#KERNEL);                                                                               \
    } while (0)
inline dim3 grid_lane(int B, int) { return grid_of(B); }
// one lane per game / G::GW lanes per game
#define BZ_DISPATCH(e, KERNEL, stream, ...) BZ_DISPATCH_IMPL(e, KERNEL, grid_lane, stream, __VA_ARGS__)
#define BZ_DISPATCH_G(e, KERNEL, stream, ...) BZ_DISPATCH_IMPL(e, KERNEL, grid_groups, stream, __VA_ARGS__)

const char* kBadCfg = "bad config (note: sims <= 8189, or <= 2045 with BZ_ENGINE_REUSE_SUBTREE -- the packed edge record, bz_abi.h)";

BZ_EXPORT int64_t bz_engine_workspace_bytes(const bz_engine_cfg* cfg) {
    if (!cfg_ok(cfg)) { set_error("bz_engine_workspace_bytes: %s", kBadCfg); return -1; }
    return carve(*cfg).total;
}

BZ_EXPORT int32_t bz_engine_create(const bz_engine_cfg* cfg, void* ws, int64_t bytes, bz_engine** out) {
    BZ_REQUIRE(ws && out, "bz_engine_create: null pointer");
    if (!cfg_ok(cfg)) { set_error("bz_engine_create: %s", kBadCfg); return BZ_EINVAL; }
    if (bz_device_count() <= 0) { set_error("bz_engine_create: no HIP device (the engine has no CPU path)"); return BZ_ENOGPU; }
    Offsets o = carve(*cfg);
    if (bytes < o.total) { set_error("bz_engine_create: workspace too small (%lld < %lld)", (long long)bytes, (long long)o.total); return BZ_ENOMEM; }
    BZ_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "bz_engine_create: workspace must be 256-byte aligned");
    bz_engine* e = new (std::nothrow) bz_engine();
    if (!e) { set_error("out of host memory"); return BZ_ENOMEM; }
    e->cfg = *cfg; e->net = nullptr; e->bytes = o.total; e->pack_parity = 1; e->n_ahead = 0; e->search_seq = 0; e->eval_epoch = 0;
    // measured on MI355X at 65,536 games x 50 sims: round 2 (profiles/r02_bench_ttt_gw*) 2 lanes 0.185 ms, 4 lanes 0.190 ms,
    // 8 lanes 0.294 ms per launch; round 3, after the kernel became issue-bound and lost a third of its instructions
    // (profiles/r03_bench_ttt_lanes.txt): 1 lane 0.162, 2 lanes 0.137, 4 lanes 0.134, 8 lanes 0.181 ms -> 4 lanes
    e->ttt_gw = cfg->ttt_lanes > 0 ? cfg->ttt_lanes : (cfg->ttt_lanes < 0 ? 0 : 4);
    EngineDev& d = e->dev;
    d.B = cfg->n_games; d.ncap = o.ncap; d.ecap = o.ecap; d.sims = cfg->sims; d.na = o.na; d.t_max = cfg->t_max;
    d.rounds = cfg->rounds; d.temp_moves = cfg->temp_moves; d.openings = cfg->openings; d.maxd = o.maxd;
    d.stagger = cfg->stagger;
    d.c_puct = cfg->c_puct; d.dir_alpha = cfg->dirichlet_alpha; d.dir_eps = cfg->dirichlet_eps; d.seed = cfg->seed; d.id_base = cfg->game_id_base; d.id_stride = cfg->game_id_stride;
    d.nodes = at<Node>(ws, o.nodes); d.edges = at<Edge>(ws, o.edges);
    d.reuse = (cfg->flags & BZ_ENGINE_REUSE_SUBTREE) ? 1 : 0;
    d.nodes_alt = at<Node>(ws, o.nodes_alt); d.edges_alt = at<Edge>(ws, o.edges_alt);
    d.g_reuse = at<u32>(ws, o.g_reuse);
    d.g_own = at<u64>(ws, o.g_own); d.g_opp = at<u64>(ws, o.g_opp); d.g_to_move = at<int8_t>(ws, o.g_to_move);
    d.g_state = at<uint8_t>(ws, o.g_state); d.g_moves = at<int32_t>(ws, o.g_moves); d.g_nex = at<int32_t>(ws, o.g_nex);
    d.g_round = at<int32_t>(ws, o.g_round); d.g_passes = at<int32_t>(ws, o.g_passes);
    d.hot = at<GameHot>(ws, o.hot); d.path = at<PathEnt>(ws, o.path);
    d.leaf_kind = at<uint8_t>(ws, o.leaf_kind);
    d.leaf_own = at<u64>(ws, o.leaf_own); d.leaf_opp = at<u64>(ws, o.leaf_opp);
    d.c_own = at<u64>(ws, o.c_own); d.c_opp = at<u64>(ws, o.c_opp);
    d.compact = net_eval(cfg->eval_kind) ? 1 : 0;
    d.ecache = o.ecache; d.tt_buckets = o.tt_buckets; d.tt = at<u64>(ws, o.tt); d.node_v = at<float>(ws, o.node_v); d.node_v_alt = at<float>(ws, o.node_v_alt);
    if (o.ecache) {  // generation 0 = never written
        hipError_t ce = hipMemset(d.tt, 0, (size_t)cfg->n_games * o.tt_buckets * 16 * 8);
        if (ce != hipSuccess) { delete e; return hip_fail(ce, "bz_engine_create: evaluation-cache clear"); }
    }
    d.logits = at<float>(ws, o.logits); d.value = at<float>(ws, o.value);
    d.ex_own = at<u64>(ws, o.ex_own); d.ex_opp = at<u64>(ws, o.ex_opp); d.ex_pi = at<float>(ws, o.ex_pi);
    d.ex_z = at<int8_t>(ws, o.ex_z); d.ex_mover = at<int8_t>(ws, o.ex_mover); d.ex_act = at<uint8_t>(ws, o.ex_act);
    d.ex_len = at<int32_t>(ws, o.ex_len); d.ex_winner = at<int8_t>(ws, o.ex_winner);
    d.root_N = at<u32>(ws, o.root_N); d.root_W = at<float>(ws, o.root_W); d.root_P = at<float>(ws, o.root_P);
    d.counters = at<u64>(ws, o.counters); d.flags = at<u32>(ws, o.flags);
    d.cnt_slots = at<u64>(ws, o.cnt_slots); d.n_cnt_slots = o.n_cnt_slots;
    d.pack_off = at<int32_t>(ws, o.pack_off);
    bz_engine_layout& l = e->lay;
    l.ex_own = o.ex_own; l.ex_opp = o.ex_opp; l.ex_pi = o.ex_pi; l.ex_z = o.ex_z; l.ex_mover = o.ex_mover;
    l.ex_act = o.ex_act; l.ex_len = o.ex_len; l.ex_winner = o.ex_winner; l.root_N = o.root_N; l.root_W = o.root_W;
    l.root_P = o.root_P; l.leaf_own = o.leaf_own; l.leaf_opp = o.leaf_opp; l.leaf_kind = o.leaf_kind;
    l.logits = o.logits; l.value = o.value; l.g_own = o.g_own; l.g_opp = o.g_opp; l.g_to_move = o.g_to_move;
    l.g_state = o.g_state; l.counters = o.counters; l.na = o.na; l.t_max = cfg->t_max;
    l.ex_begin = o.ex_own; l.ex_bytes = o.ex_meta + 256 - o.ex_own; l.ex_meta = o.ex_meta;
    {   // self-describing header of the example block (so a gathered block can be unpacked without its engine)
        uint64_t meta[32] = {0};
        meta[0] = 0x425A455841000002ULL;  // "BZEXA" + layout version 2
        meta[1] = cfg->game_id_base; meta[2] = cfg->game_id_stride; meta[3] = (uint64_t)cfg->n_games;
        meta[4] = (uint64_t)cfg->rounds; meta[5] = (uint64_t)cfg->t_max; meta[6] = (uint64_t)o.na; meta[7] = (uint64_t)cfg->game;
        const int64_t offs[8] = {o.ex_own, o.ex_opp, o.ex_pi, o.ex_z, o.ex_mover, o.ex_act, o.ex_len, o.ex_winner};
        for (int i = 0; i < 8; ++i) meta[8 + i] = (uint64_t)(offs[i] - o.ex_own);
        hipError_t me = hipMemcpy(at<char>(ws, o.ex_meta), meta, sizeof(meta), hipMemcpyHostToDevice);
        if (me != hipSuccess) { delete e; return hip_fail(me, "bz_engine_create: example header upload"); }
    }
    *out = e;
    return BZ_OK;
}

BZ_EXPORT int32_t bz_engine_destroy(bz_engine* e) {
    if (e) for (int i = 0; i < e->n_ahead; ++i) (void)hipEventDestroy(e->ahead[i]);
    delete e;
    return BZ_OK;
}

BZ_EXPORT int32_t bz_engine_get_layout(const bz_engine* e, bz_engine_layout* out) {
    BZ_REQUIRE(e && out, "bz_engine_get_layout: null pointer");
    *out = e->lay;
    return BZ_OK;
}

BZ_EXPORT int32_t bz_engine_set_net(bz_engine* e, bz_net* net) {
    BZ_REQUIRE(e, "bz_engine_set_net: null engine");
    e->net = net;
    return BZ_OK;
}

/* test hook: the number of searches begun so far, as the evaluation cache counts them (its generation stamp cycles through
 * 1 .. 2^19 - 2).  Lets a test start an engine just below the wrap instead of running half a million searches. */
BZ_EXPORT int32_t bz_engine_debug_set_search_seq(bz_engine* e, uint32_t seq) {
    BZ_REQUIRE(e, "null engine");
    BZ_REQUIRE(seq < kTtGenMax - 1u, "bz_engine_debug_set_search_seq: 0 <= seq <= 2^19 - 3");
    e->search_seq = seq;
    e->eval_epoch = 0;  // nothing is carried over from before the jump
    return BZ_OK;
}

BZ_EXPORT int32_t bz_engine_reset_counters(bz_engine* e, void* stream) {
    BZ_REQUIRE(e, "null engine");
    BZ_HIP(hipMemsetAsync(e->dev.counters, 0, kCntWords * 8, (hipStream_t)stream));
    BZ_HIP(hipMemsetAsync(e->dev.cnt_slots, 0, (size_t)e->dev.n_cnt_slots * CNT_N * 8, (hipStream_t)stream));
    return BZ_OK;
}

/* fold the per-wave counter slots into the counters[16] array of the layout (async) */
BZ_EXPORT int32_t bz_engine_sum_counters(bz_engine* e, void* stream) {
    BZ_REQUIRE(e, "null engine");
    hipLaunchKernelGGL(k_sum_counters, dim3(1), dim3(256), 0, (hipStream_t)stream, e->dev);
    BZ_LAUNCH_CHECK("k_sum_counters");
    return BZ_OK;
}

BZ_EXPORT int32_t bz_engine_reset_games(bz_engine* e, void* stream) {
    BZ_REQUIRE(e, "null engine");
    BZ_DISPATCH(e, k_reset_games, stream, e->dev);
    return BZ_OK;
}

BZ_EXPORT int32_t bz_engine_set_roots(bz_engine* e, const uint64_t* own, const uint64_t* opp, const int8_t* to_move,
                                      void* stream) {
    BZ_REQUIRE(e && own && opp && to_move, "bz_engine_set_roots: null pointer");
    hipLaunchKernelGGL(k_set_roots, grid_of(e->dev.B), dim3(256), 0, (hipStream_t)stream, e->dev, own, opp, to_move);
    BZ_LAUNCH_CHECK("k_set_roots");
    return BZ_OK;
}

BZ_EXPORT int32_t bz_engine_root_begin(bz_engine* e, void* stream) {
    BZ_REQUIRE(e, "null engine");
    e->search_seq = e->search_seq % (kTtGenMax - 1u) + 1u;  // 1 .. 2^19 - 2 (0 = never written)
    if (e->dev.ecache == 2) {  // the tree just searched stays intact in the other arena while the new one grows in this one
        Node* tn = e->dev.nodes; e->dev.nodes = e->dev.nodes_alt; e->dev.nodes_alt = tn;
        Edge* te = e->dev.edges; e->dev.edges = e->dev.edges_alt; e->dev.edges_alt = te;
        float* tv = e->dev.node_v; e->dev.node_v = e->dev.node_v_alt; e->dev.node_v_alt = tv;
    }
    // evaluations of the previous search are this search's only under the same weights (bz_net_update / bz_engine_set_net between
    // two searches of a live engine: nothing is carried over, exactly as if the slot had sat the previous search out)
    const uint64_t epoch = bz_net_epoch(e->net);
    const int carry_ok = epoch == e->eval_epoch;
    e->eval_epoch = epoch;
    BZ_DISPATCH(e, k_root_begin, stream, e->dev, e->search_seq, carry_ok);
    e->pack_parity = 1;
    return BZ_OK;
}

static int32_t tree_step(bz_engine* e, int do_expand, int do_select, uint32_t sim_idx, void* stream) {
    ProfScope ps(do_select ? BZ_PROF_SELECT : BZ_PROF_EXPAND_BACKUP, stream);
    BZ_DISPATCH_G(e, k_tree_step, stream, e->dev, do_expand, do_select, sim_idx);
    if (do_select) e->pack_parity = (int)(sim_idx & 1u);
    return BZ_OK;
}

/* sim_index = number of simulations already completed in this search (the root's visit sum) */
BZ_EXPORT int32_t bz_engine_select(bz_engine* e, uint32_t sim_index, void* stream) {
    BZ_REQUIRE(e, "null engine");
    return tree_step(e, 0, 1, sim_index, stream);
}

BZ_EXPORT int32_t bz_engine_evaluate(bz_engine* e, void* stream) {
    BZ_REQUIRE(e, "null engine");
    int ek = e->cfg.eval_kind;
    if (ek == BZ_EVAL_UNIFORM || ek == BZ_EVAL_HASH) {
        BZ_DISPATCH(e, k_eval_synth, stream, e->dev, ek);
        return BZ_OK;
    }
    if (ek == BZ_EVAL_EXTERNAL) return BZ_OK;
    BZ_REQUIRE(e->net, "bz_engine_evaluate: eval_kind needs a net (bz_engine_set_net)");
    // the net works on 8x8 planes; the reference's 6x6 / 4x4 boards live in their top-left corner (bit = 8*row+col for
    // every size), cells outside are never stones and never legal, so the same net serves them
    BZ_REQUIRE(e->cfg.game != BZ_GAME_TTT, "bz_engine_evaluate: the conv net serves the Reversi boards, not tic-tac-toe");
    // leaves were packed by select: evaluate only the first flags[NEVAL] slots (device-side count)
    return bz_net_forward_dev(e->net, ek == BZ_EVAL_NET_BF16 ? 1 : (ek == BZ_EVAL_NET_FP8 ? 2 : 0), e->dev.c_own, e->dev.c_opp, e->dev.B,
                              e->dev.flags + FLAG_NEVAL + e->pack_parity, e->dev.logits, e->dev.value, stream);
}

BZ_EXPORT int32_t bz_engine_expand_backup(bz_engine* e, void* stream) {
    BZ_REQUIRE(e, "null engine");
    return tree_step(e, 1, 0, 0, stream);
}

/* Dirichlet noise on the priors of every active slot's (expanded) root; a no-op when cfg.dirichlet_eps == 0.
 * Step-API order: root_begin, evaluate, expand_backup, root_noise, then select(0) ... -- what bz_engine_search does. */
BZ_EXPORT int32_t bz_engine_root_noise(bz_engine* e, void* stream) {
    BZ_REQUIRE(e, "null engine");
    if (!(e->dev.dir_eps > 0.0f)) return BZ_OK;
    hipLaunchKernelGGL(k_root_noise, dim3((e->dev.B + 63) / 64), dim3(64), 0, (hipStream_t)stream, e->dev);
    BZ_LAUNCH_CHECK("k_root_noise");
    return BZ_OK;
}

BZ_EXPORT int32_t bz_engine_search(bz_engine* e, void* stream) {
    BZ_REQUIRE(e, "null engine");
    int ek = e->cfg.eval_kind;
    const bool noise = e->dev.dir_eps > 0.0f;
    if ((ek == BZ_EVAL_UNIFORM || ek == BZ_EVAL_HASH) && !e->dev.reuse && !noise) {
        ProfScope ps(BZ_PROF_SEARCH_FUSED, stream);
        if (e->cfg.game == BZ_GAME_TTT && e->cfg.sims <= kTttFusedMaxSims && e->ttt_gw > 0) {
            const dim3 grid = grid_groups(e->dev.B, e->ttt_gw);
#define BZ_TTT_FUSED(GWN)                                                                                              \
    do {                                                                                                                \
        if (ek == BZ_EVAL_UNIFORM) hipLaunchKernelGGL((k_search_fused_ttt<GWN, true>), grid, dim3(256), 0, (hipStream_t)stream, e->dev); \
        else hipLaunchKernelGGL((k_search_fused_ttt<GWN, false>), grid, dim3(256), 0, (hipStream_t)stream, e->dev);     \
    } while (0)
            switch (e->ttt_gw) {
            case 1: BZ_TTT_FUSED(1); break;
            case 2: BZ_TTT_FUSED(2); break;
            case 8: BZ_TTT_FUSED(8); break;
            default: BZ_TTT_FUSED(4); break;
            }
#undef BZ_TTT_FUSED
            BZ_LAUNCH_CHECK("k_search_fused_ttt");
            return BZ_OK;
        }
        BZ_DISPATCH_G(e, k_search_fused, stream, e->dev, ek);
        return BZ_OK;
    }
    BZ_REQUIRE(ek != BZ_EVAL_EXTERNAL, "bz_engine_search: BZ_EVAL_EXTERNAL callers drive the step API");
    int32_t rc;
    if ((rc = bz_engine_root_begin(e, stream)) != BZ_OK) return rc;
    if ((rc = bz_engine_evaluate(e, stream)) != BZ_OK) return rc;
    if (noise) {  // expand the roots on their own, then draw the noise, then start selecting
        if ((rc = tree_step(e, 1, 0, 0, stream)) != BZ_OK) return rc;
        if ((rc = bz_engine_root_noise(e, stream)) != BZ_OK) return rc;
    }
    for (int s = 0; s < e->cfg.sims; ++s) {  // expand+backup of leaf s-1 (s = 0: the root) fused with select s
        if ((rc = tree_step(e, (noise && s == 0) ? 0 : 1, 1, (uint32_t)s, stream)) != BZ_OK) return rc;
        if ((rc = bz_engine_evaluate(e, stream)) != BZ_OK) return rc;
    }
    return tree_step(e, 1, 0, 0, stream);
}

BZ_EXPORT int32_t bz_engine_root_stats(bz_engine* e, void* stream) {
    BZ_REQUIRE(e, "null engine");
    BZ_DISPATCH(e, k_root_stats, stream, e->dev);
    return BZ_OK;
}

BZ_EXPORT int32_t bz_engine_play(bz_engine* e, int32_t restart, void* stream) {
    BZ_REQUIRE(e, "null engine");
    {
        ProfScope ps(BZ_PROF_PLAY, stream);
        BZ_DISPATCH(e, k_play, stream, e->dev, (int)restart);
    }
    if (e->dev.reuse) {  // the tree just searched becomes the source of the next root_begin's subtree copy
        Node* tn = e->dev.nodes; e->dev.nodes = e->dev.nodes_alt; e->dev.nodes_alt = tn;
        Edge* te = e->dev.edges; e->dev.edges = e->dev.edges_alt; e->dev.edges_alt = te;
    }
    return BZ_OK;
}

BZ_EXPORT int32_t bz_engine_status(bz_engine* e, void* stream, int32_t* n_active, int64_t* games_finished,
                                   int32_t* error_flags) {
    BZ_REQUIRE(e, "null engine");
    hipStream_t s = (hipStream_t)stream;
    BZ_HIP(hipMemsetAsync(e->dev.flags + FLAG_ACTIVE, 0, 4, s));
    hipLaunchKernelGGL(k_count_active, grid_of(e->dev.B), dim3(256), 0, s, e->dev);
    BZ_LAUNCH_CHECK("k_count_active");
    u32 h[FLAG_N];
    BZ_HIP(hipMemcpyAsync(h, e->dev.flags, sizeof(h), hipMemcpyDeviceToHost, s));
    BZ_HIP(hipStreamSynchronize(s));
    if (n_active) *n_active = (int32_t)h[FLAG_ACTIVE];
    if (games_finished) *games_finished = (int64_t)h[FLAG_FINISHED];
    if (error_flags) *error_flags = (int32_t)h[FLAG_ERR];
    return BZ_OK;
}



namespace {
PackedLayout packed_layout(int na, int64_t cap) {
    PackedLayout L{};
    Carver k;
    k.take(256);  // header first: a prefix of the block describes the rest
    const int64_t esz[8] = {8, 8, 4 * (int64_t)na, 8, 1, 1, 1, 1};
    for (int i = 0; i < 8; ++i) L.offs[i] = k.take(cap * esz[i]);
    L.total = k.off;
    return L;
}
}  // namespace

BZ_EXPORT int64_t bz_examples_packed_bytes(int32_t na, int64_t cap_rows) {
    if (na < 1 || na > 4096 || cap_rows < 1 || cap_rows > (int64_t(1) << 31) - 1) { set_error("bz_examples_packed_bytes: bad arguments"); return -1; }
    return packed_layout(na, cap_rows).total;
}

/* compact the finished games' rows of this engine into the caller's packed block (append != 0: behind the rows a
 * previous call -- another engine of the same geometry -- left there) */
BZ_EXPORT int32_t bz_engine_pack_examples(bz_engine* e, void* packed, int64_t packed_bytes, int64_t cap_rows, int32_t append,
                                          void* stream) {
    BZ_REQUIRE(e && packed, "bz_engine_pack_examples: null pointer");
    BZ_REQUIRE(cap_rows >= 1 && cap_rows <= (int64_t(1) << 31) - 1, "bz_engine_pack_examples: bad capacity");
    BZ_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 255) == 0, "bz_engine_pack_examples: the block must be 256-byte aligned");
    const PackedLayout L = packed_layout(e->dev.na, cap_rows);
    if (packed_bytes < L.total) { set_error("bz_engine_pack_examples: block too small (%lld < %lld)", (long long)packed_bytes, (long long)L.total); return BZ_ENOMEM; }
    hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, (hipStream_t)stream, e->dev, static_cast<PackedHdr*>(packed), L,
                       (u64)cap_rows, (int)(append != 0), (int)e->cfg.game);
    BZ_LAUNCH_CHECK("k_pack_scan");
    const int n = e->dev.rounds * e->dev.B;
    hipLaunchKernelGGL(k_pack_rows, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, e->dev, static_cast<char*>(packed), L);
    BZ_LAUNCH_CHECK("k_pack_rows");
    return BZ_OK;
}

/* names used by SURVEY.md 8(b) for the same entry points */
BZ_EXPORT int32_t bz_mcts_select(bz_engine* e, uint32_t sim_index, void* stream) { return bz_engine_select(e, sim_index, stream); }
BZ_EXPORT int32_t bz_mcts_expand_backup(bz_engine* e, void* stream) { return bz_engine_expand_backup(e, stream); }
/* one move for every active slot: search (root expansion + cfg.sims simulations) then play */
BZ_EXPORT int32_t bz_selfplay_run(bz_engine* e, int32_t restart, void* stream) {
    int32_t rc = bz_engine_search(e, stream);
    return rc != BZ_OK ? rc : bz_engine_play(e, restart, stream);
}

/* One move (search + play) for several engines at once -- the pipelines of one GPU, each on its own stream -- issued
 * by ONE host thread, interleaved simulation by simulation, so that every stream always has work queued whatever the
 * host's run-ahead.  run_ahead_sims > 0 bounds that run-ahead: every run_ahead_sims / 2 simulations the thread records
 * a blocking-sync event per stream and sleeps on the one recorded two marks earlier.  Without the bound the thread
 * queues launches until the runtime's queue is full and then spins there: measured 100 % of a core for the whole
 * timed region (plus a second runtime thread), which eight ranks on one host cannot afford. */
static int32_t ahead_mark(bz_engine* e, int idx, hipStream_t s) {
    while (e->n_ahead < 4) {
        BZ_HIP(hipEventCreateWithFlags(&e->ahead[e->n_ahead], hipEventBlockingSync | hipEventDisableTiming));
        e->n_ahead++;
    }
    BZ_HIP(hipEventRecord(e->ahead[idx & 3], s));
    if (idx >= 2) {
        // hipEventSynchronize spins on this stack even for hipEventBlockingSync events (measured: the thread stayed at
        // 100 % of a core), so the wait is a poll with real sleeps; a mark is many milliseconds of queued GPU work
        const timespec nap = {0, 100 * 1000};
        for (;;) {
            const hipError_t q = hipEventQuery(e->ahead[(idx - 2) & 3]);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) return hip_fail(q, "bz_engines_step: hipEventQuery");
            (void)hipGetLastError();  // hipErrorNotReady is sticky in hipGetLastError
            nanosleep(&nap, nullptr);
        }
    }
    return BZ_OK;
}

BZ_EXPORT int32_t bz_engines_step(bz_engine* const* engines, void* const* streams, int32_t n, int32_t restart,
                                  int32_t run_ahead_sims) {
    BZ_REQUIRE(engines && streams && n >= 1 && n <= 16 && run_ahead_sims >= 0, "bz_engines_step: bad arguments");
    bool stepwise = true;
    for (int i = 0; i < n; ++i) {
        BZ_REQUIRE(engines[i], "bz_engines_step: null engine");
        BZ_REQUIRE(engines[i]->cfg.sims == engines[0]->cfg.sims, "bz_engines_step: the engines must search the same number of simulations");
        const int ek = engines[i]->cfg.eval_kind;
        BZ_REQUIRE(ek != BZ_EVAL_EXTERNAL, "bz_engines_step: BZ_EVAL_EXTERNAL callers drive the step API");
        if ((ek == BZ_EVAL_UNIFORM || ek == BZ_EVAL_HASH) && !engines[i]->dev.reuse && !(engines[i]->dev.dir_eps > 0.0f)) stepwise = false;
    }
    int32_t rc;
    if (!stepwise) {  // a fused search is one launch per engine: nothing to interleave
        for (int i = 0; i < n; ++i)
            if ((rc = bz_selfplay_run(engines[i], restart, streams[i])) != BZ_OK) return rc;
        return BZ_OK;
    }
    for (int i = 0; i < n; ++i) {
        bz_engine* e = engines[i];
        if ((rc = bz_engine_root_begin(e, streams[i])) != BZ_OK) return rc;
        if ((rc = bz_engine_evaluate(e, streams[i])) != BZ_OK) return rc;
        if (e->dev.dir_eps > 0.0f) {
            if ((rc = tree_step(e, 1, 0, 0, streams[i])) != BZ_OK) return rc;
            if ((rc = bz_engine_root_noise(e, streams[i])) != BZ_OK) return rc;
        }
    }
    const int sims = engines[0]->cfg.sims, q = run_ahead_sims > 0 ? (run_ahead_sims >= 2 ? run_ahead_sims / 2 : 1) : 0;
    for (int s = 0; s < sims; ++s) {
        for (int i = 0; i < n; ++i) {
            bz_engine* e = engines[i];
            if ((rc = tree_step(e, (e->dev.dir_eps > 0.0f && s == 0) ? 0 : 1, 1, (uint32_t)s, streams[i])) != BZ_OK) return rc;
            if ((rc = bz_engine_evaluate(e, streams[i])) != BZ_OK) return rc;
        }
        if (q && (s + 1) % q == 0)
            for (int i = 0; i < n; ++i)
                if ((rc = ahead_mark(engines[i], (s + 1) / q - 1, (hipStream_t)streams[i])) != BZ_OK) return rc;
    }
    for (int i = 0; i < n; ++i) {
        if ((rc = tree_step(engines[i], 1, 0, 0, streams[i])) != BZ_OK) return rc;
        if ((rc = bz_engine_play(engines[i], restart, streams[i])) != BZ_OK) return rc;
    }
    return BZ_OK;
}
