// bz_env.hip -- scalar rule entry points (host) and the batched board-env step
// kernels (gfx950).  One game per lane: a wave steps 64 games, all loads and
// stores are coalesced 8-byte (Reversi) / 2-byte (TTT) streams, the rule code is
// pure 64-bit integer ALU (bz_rules.h) -- HBM-bound at 34+8 B per env step.
#include <stdarg.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "bz_common.h"
#include "bz_rules.h"

namespace bz {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace bz

namespace bz {
namespace {
// In-library kernel timers.  Every launch of a slot carries a start and an end event on its own
// stream; events are created up front by bz_profile_reserve() (so that nothing but the two
// hipEventRecord calls sits in a timed loop) and grown on demand otherwise.  A slot holds at most
// kProfCap launches -- bench.py's default run records about 32 k per slot.
constexpr int kProfCap = 1 << 18;
struct ProfSlot { std::vector<hipEvent_t> ev; int64_t launches = 0; int used = 0; };
ProfSlot g_prof[BZ_PROF_N];
std::atomic<bool> g_prof_on{false};
std::mutex g_prof_mu;  // launches may come from several host threads (one per stream)
bool prof_grow(ProfSlot& p, int n_launches) {
    while ((int)p.ev.size() < 2 * n_launches) {
        hipEvent_t a;
        if (hipEventCreate(&a) != hipSuccess) { (void)hipGetLastError(); return false; }
        p.ev.push_back(a);
    }
    return true;
}
}  // namespace
int prof_begin(int slot, hipStream_t s) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return -1;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfSlot& p = g_prof[slot];
    p.launches++;
    if (p.used >= kProfCap || !prof_grow(p, p.used + 1)) return -1;
    int idx = p.used++;
    (void)hipEventRecord(p.ev[2 * idx], s);
    return idx;
}
void prof_end(int slot, int idx, hipStream_t s) {
    if (idx < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_prof[slot].ev[2 * idx + 1], s);
}
}  // namespace bz

using namespace bz;

BZ_EXPORT int32_t bz_profile_enable(int32_t on) { bz::g_prof_on.store(on != 0); return BZ_OK; }
BZ_EXPORT int32_t bz_profile_reserve(int32_t slot, int64_t n_launches) {
    BZ_REQUIRE(slot >= 0 && slot < BZ_PROF_N && n_launches >= 0, "bz_profile_reserve: bad arguments");
    std::lock_guard<std::mutex> lk(bz::g_prof_mu);
    int n = (int)(n_launches < bz::kProfCap ? n_launches : bz::kProfCap);
    if (!bz::prof_grow(bz::g_prof[slot], n)) { set_error("bz_profile_reserve: hipEventCreate failed"); return BZ_EHIP; }
    return BZ_OK;
}
BZ_EXPORT int32_t bz_profile_reset(void) {
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lk(bz::g_prof_mu);
    for (auto& p : bz::g_prof) { p.launches = 0; p.used = 0; }
    return BZ_OK;
}
BZ_EXPORT int32_t bz_profile_read(int32_t slot, int64_t* launches, int64_t* timed, double* total_ms) {
    BZ_REQUIRE(slot >= 0 && slot < BZ_PROF_N && launches && timed && total_ms, "bz_profile_read: bad arguments");
    BZ_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(bz::g_prof_mu);
    bz::ProfSlot& p = bz::g_prof[slot];
    double tot = 0;
    for (int i = 0; i < p.used; ++i) {
        float ms = 0;
        BZ_HIP(hipEventElapsedTime(&ms, p.ev[2 * i], p.ev[2 * i + 1]));
        tot += ms;
    }
    *launches = p.launches; *timed = p.used; *total_ms = tot;
    return BZ_OK;
}

/* start/end of every timed launch of a slot, in ms relative to the slot's first recorded event
 * (launches on different streams may overlap; the caller can form the union) */
BZ_EXPORT int32_t bz_profile_intervals(int32_t slot, double* starts_ms, double* ends_ms, int64_t cap, int64_t* n) {
    BZ_REQUIRE(slot >= 0 && slot < BZ_PROF_N && starts_ms && ends_ms && n, "bz_profile_intervals: bad arguments");
    BZ_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(bz::g_prof_mu);
    bz::ProfSlot& p = bz::g_prof[slot];
    int64_t m = p.used < cap ? p.used : cap;
    for (int64_t i = 0; i < m; ++i) {
        float a = 0, b = 0;
        BZ_HIP(hipEventElapsedTime(&a, p.ev[0], p.ev[2 * i]));
        BZ_HIP(hipEventElapsedTime(&b, p.ev[0], p.ev[2 * i + 1]));
        starts_ms[i] = a; ends_ms[i] = b;
    }
    *n = m;
    return BZ_OK;
}

// ---- do two streams really run side by side?  (bz_stream_overlap_probe, bz_abi.h)
// One wave that waits for `ticks` of the 100-MHz wall clock (s_memrealtime); the iteration cap is the exit every
// wave reaches whatever the clock does.
__global__ void __launch_bounds__(64) k_spin(unsigned long long ticks, unsigned int* sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned int it = 0;
    while (wall_clock64() - t0 < ticks && it < (1u << 22)) { __builtin_amdgcn_s_sleep(8); ++it; }
    if (sink && it == 0xFFFFFFFFu) *sink = it;  // never true: keeps the loop observable
}

BZ_EXPORT int32_t bz_stream_overlap_probe(void* stream_a, void* stream_b, int32_t spin_us, int32_t reps, float* serial_ratio) {
    BZ_REQUIRE(serial_ratio && spin_us > 0 && spin_us <= 5000 && reps > 0 && reps <= 16, "bz_stream_overlap_probe: bad arguments");
    hipStream_t a = (hipStream_t)stream_a, b = (hipStream_t)stream_b;
    BZ_REQUIRE(a != b, "bz_stream_overlap_probe: the two streams are the same stream");
    hipEvent_t e0, ea, eb;
    BZ_HIP(hipEventCreate(&e0)); BZ_HIP(hipEventCreate(&ea)); BZ_HIP(hipEventCreate(&eb));
    const unsigned long long ticks = (unsigned long long)spin_us * 100ULL;
    float best = 1e30f;
    int32_t rc = BZ_OK;
    for (int r = 0; r <= reps && rc == BZ_OK; ++r) {  // (repetition 0 warms the code object up and is not counted)
        hipError_t e = hipEventRecord(e0, a);
        if (e == hipSuccess) e = hipStreamWaitEvent(b, e0, 0);  // both spins start behind the same point in time
        if (e == hipSuccess) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, ticks, (unsigned int*)nullptr); e = hipGetLastError(); }
        if (e == hipSuccess) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, ticks, (unsigned int*)nullptr); e = hipGetLastError(); }
        if (e == hipSuccess) e = hipEventRecord(ea, a);
        if (e == hipSuccess) e = hipEventRecord(eb, b);
        if (e == hipSuccess) e = hipEventSynchronize(ea);
        if (e == hipSuccess) e = hipEventSynchronize(eb);
        float ta = 0, tb = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ta, e0, ea);
        if (e == hipSuccess) e = hipEventElapsedTime(&tb, e0, eb);
        if (e != hipSuccess) { rc = bz::hip_fail(e, "bz_stream_overlap_probe"); break; }
        const float ratio = (ta > tb ? ta : tb) * 1000.0f / (float)spin_us;
        if (r > 0 && ratio < best) best = ratio;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
    if (rc == BZ_OK) *serial_ratio = best;
    return rc;
}

#ifndef BZ_BUILD_INFO
#define BZ_BUILD_INFO "unknown"
#endif
BZ_EXPORT const char* bz_build_info(void) { return BZ_BUILD_INFO; }
BZ_EXPORT int32_t bz_abi_version(void) { return BZ_ABI_VERSION; }
BZ_EXPORT const char* bz_last_error(void) { return bz::g_err; }
BZ_EXPORT int32_t bz_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// the rule entry points take every size whose cells fit bit = 8*row+col (reversi_board.py:4-14 is generic); the engine's
// games (bz_engine_cfg.game) and the arena stay 4 / 6 / 8
static bool size_ok(int32_t s) { return s >= 1 && s <= 8; }

BZ_EXPORT int32_t bz_reversi_legal(uint64_t own, uint64_t opp, int32_t size, uint64_t* legal) {
    BZ_REQUIRE(size_ok(size) && legal, "bz_reversi_legal: size must be 1..8");
    *legal = rev_legal(own, opp, rev_valid(size));
    return BZ_OK;
}
BZ_EXPORT int32_t bz_reversi_apply(uint64_t own, uint64_t opp, int32_t size, int32_t row, int32_t col,
                                   uint64_t* own_after, uint64_t* opp_after, uint64_t* flips) {
    BZ_REQUIRE(size_ok(size) && own_after && opp_after, "bz_reversi_apply: bad arguments");
    if (row < 0 || row >= size || col < 0 || col >= size) { set_error("Invalid move"); return BZ_EILLEGAL_MOVE; }
    u64 m = 1ULL << (8 * row + col);
    if (!(rev_legal(own, opp, rev_valid(size)) & m)) { set_error("Invalid move"); return BZ_EILLEGAL_MOVE; }
    u64 f = rev_flips(own, opp, m);
    *own_after = own | m | f;
    *opp_after = opp & ~f;
    if (flips) *flips = f;
    return BZ_OK;
}
BZ_EXPORT int32_t bz_reversi_game_over(uint64_t a, uint64_t b, int32_t size, int32_t* over) {
    BZ_REQUIRE(size_ok(size) && over, "bz_reversi_game_over: bad arguments");
    u64 v = rev_valid(size);
    *over = rev_legal(a, b, v) == 0 && rev_legal(b, a, v) == 0;
    return BZ_OK;
}
BZ_EXPORT int32_t bz_reversi_score(uint64_t x, uint64_t o, int32_t* winner, int32_t* n_x, int32_t* n_o) {
    BZ_REQUIRE(winner, "bz_reversi_score: bad arguments");
    int cx = popc64(x), co = popc64(o);
    *winner = cx > co ? 1 : (co > cx ? -1 : 0);
    if (n_x) *n_x = cx;
    if (n_o) *n_o = co;
    return BZ_OK;
}
BZ_EXPORT int32_t bz_ttt_legal(uint32_t x, uint32_t o, uint32_t* legal) {
    BZ_REQUIRE(legal, "bz_ttt_legal: bad arguments");
    *legal = ~(x | o) & 0x1FFu;
    return BZ_OK;
}
BZ_EXPORT int32_t bz_ttt_apply(uint32_t own, uint32_t opp, int32_t row, int32_t col, uint32_t* own_after) {
    BZ_REQUIRE(own_after, "bz_ttt_apply: bad arguments");
    if (row < 0 || row >= 3 || col < 0 || col >= 3 || ((own | opp) >> (3 * row + col) & 1u)) {
        set_error("Invalid move");
        return BZ_EILLEGAL_MOVE;
    }
    *own_after = own | (1u << (3 * row + col));
    return BZ_OK;
}
BZ_EXPORT int32_t bz_ttt_game_over(uint32_t x, uint32_t o, int32_t* over, int32_t* winner) {
    BZ_REQUIRE(over && winner, "bz_ttt_game_over: bad arguments");
    int w;
    *over = ttt_over(x & 0x1FF, o & 0x1FF, &w);
    *winner = w;
    return BZ_OK;
}

// ---------------------------------------------------------------- kernels
// The flips of make_move (reversi_board.py:49-58) by carry propagation, for the batched step kernel (which is bound by
// integer VALU issue, so instructions per step are what counts).  For a ray that runs towards HIGHER bit indices from
// the move square, with M = the ray's cells: (O | ~M) + 1 ripples a carry from bit 0 through every non-ray bit (all
// ones) and every opponent stone on the ray and stops at the first ray cell that holds no opponent stone; if the mover
// owns that cell the stones below it on the ray are the flips: 15 instructions per ray against ~28 for the
// parallel-prefix fill.  The four rays towards LOWER indices are the same computation on the bit-reversed boards from
// square 63 - a (bit i <-> 63 - i turns every ray around).  The 64 x 4 ray masks (with their complements, one 16-byte
// LDS read each) are built by the workgroup when the kernel starts.
struct RayEnt { u64 m, nm; };
__device__ __forceinline__ void build_ray_table(RayEnt* tab /* [64][4] */) {
    for (int e = threadIdx.x; e < 256; e += blockDim.x) {
        const int sq = e >> 2, d = e & 3, r0 = sq >> 3, c0 = sq & 7;
        const int dr = d == 0 ? 0 : 1, dc = d == 0 ? 1 : (d == 1 ? 0 : (d == 2 ? -1 : 1));  // +1, +8, +7, +9
        u64 m = 0;
        for (int r = r0 + dr, c = c0 + dc; r < 8 && c >= 0 && c < 8; r += dr, c += dc) m |= 1ULL << (8 * r + c);
        tab[e].m = m; tab[e].nm = ~m;
    }
}
__device__ __forceinline__ u64 ray_flips_up(u64 P, u64 O, RayEnt e) {
    const u64 of = and3((O | e.nm) + 1ULL, e.m, P);    // the mover's stone that closes the run, or 0
    const u64 x = of - 1ULL;                            // of == 0 -> all ones (bit 63 set); else the bits below it
    return and_andn(x, e.m, (u64)((int64_t)x >> 63));
}
// flips of placing on square a (the cell must be empty); me = mover's stones
__device__ __forceinline__ u64 rev_flips_carry(u64 me, u64 you, int a, const RayEnt* tab) {
    const RayEnt* up = tab + 4 * a;
    u64 f = ray_flips_up(me, you, up[0]) | ray_flips_up(me, you, up[1]) | ray_flips_up(me, you, up[2]) |
            ray_flips_up(me, you, up[3]);
    const u64 rme = ((u64)__brev((u32)me) << 32) | (u64)__brev((u32)(me >> 32));
    const u64 ryou = ((u64)__brev((u32)you) << 32) | (u64)__brev((u32)(you >> 32));
    const RayEnt* dn = tab + 4 * (63 - a);
    const u64 g = ray_flips_up(rme, ryou, dn[0]) | ray_flips_up(rme, ryou, dn[1]) | ray_flips_up(rme, ryou, dn[2]) |
                  ray_flips_up(rme, ryou, dn[3]);
    return f | ((u64)__brev((u32)g) << 32) | (u64)__brev((u32)(g >> 32));
}

// One env step, split in two so that the batch kernel can run the RARE part once per lane instead of once per game:
//  * step_main: the move itself and the next mover's legal mask (always needed).  A placement is legal iff the cell is
//    empty and it flips something, so the mover's own legal mask is not computed.  Returns true when one more legal
//    mask J = rev_legal(jx, jy) is needed to finish: to verify a pass (the mover must have no move: J = legal(me, you)),
//    or, when the next mover has no move, to tell MUST_PASS from TERMINAL (J = legal(copp, cown)).  For a pass the two
//    are the same mask.
//  * step_finish: the status / winner from J.
// With 4 games per lane and 64 lanes, "rare" (a per cent of the games) would otherwise mean "in nearly every wave, for
// every one of the 4 game slots": ~80 extra instructions per game.
struct StepOut { u64 cown, copp, nl; uint8_t st; int8_t w; };
// (round 5) ONE code path for the three kinds of action.  A pass, a placement and an illegal placement all end in
// "the legal mask of whoever moves in the resulting position", so the position is chosen with selects and rev_legal
// appears once (it was compiled three times -- pass / nothing flipped / placement -- and a wave in which one lane passes
// executed two of the copies for that game slot).
__device__ __forceinline__ bool step_main(u64 me, u64 you, int a, u64 valid, const RayEnt* tab, StepOut& o, u64& jx, u64& jy) {
    const bool pass = a == kPass;
    const u64 m = a < 64 ? (1ULL << (a & 63)) & valid : 0ULL;
    const u64 f = (m & ~(me | you)) ? rev_flips_carry(me, you, a & 63, tab) : 0ULL;   // (skipped by lanes that pass or hit a stone)
    const bool placed = f != 0;            // a placement is legal iff the cell is empty and it flips something
    const bool moved = pass || placed;     // the side changes; an illegal placement leaves the position as it was
    o.cown = moved ? (you & ~f) : me;      // (f = 0 for a pass)
    o.copp = moved ? (me | (placed ? m : 0ULL) | f) : you;
    o.nl = rev_legal(o.cown, o.copp, valid);
    o.st = placed ? BZ_ST_RUNNING : BZ_ST_ILLEGAL;   // (a pass is settled by step_finish)
    o.w = 0;
    // one more legal mask J finishes the rare cases: the mover's own (a pass is legal only if it is empty), or the mask
    // of the player who just placed a stone when the next mover has no move (MUST_PASS vs TERMINAL)
    jx = pass ? me : o.copp; jy = pass ? you : o.cown;
    return pass || (placed && o.nl == 0);
}
__device__ __forceinline__ void step_finish(u64 me, u64 you, int a, u64 J, StepOut& o) {
    if (a == kPass && J != 0) {  // the mover had a move: the pass is illegal, nothing changes
        o.cown = me; o.copp = you; o.nl = J; o.st = BZ_ST_ILLEGAL; o.w = 0;
        return;
    }
    o.st = BZ_ST_RUNNING;
    if (o.nl == 0) {
        if (J == 0) {
            o.st = BZ_ST_TERMINAL;
            const int d = popc64(o.copp) - popc64(o.cown);  // copp = the player who just moved
            o.w = (int8_t)(d > 0 ? 1 : (d < 0 ? -1 : 0));
        } else {
            o.st = BZ_ST_MUST_PASS;
        }
    }
}
__device__ __forceinline__ void reversi_step_one(u64 me, u64 you, int a, u64 valid, const RayEnt* tab, u64& cown, u64& copp,
                                                 u64& nl, uint8_t& st, int8_t& w) {
    StepOut o; u64 jx, jy;
    if (step_main(me, you, a, valid, tab, o, jx, jy)) step_finish(me, you, a, rev_legal(jx, jy, valid), o);
    cown = o.cown; copp = o.copp; nl = o.nl; st = o.st; w = o.w;
}

// 4 games per lane: 2 x 16-byte loads per bitboard array, 4-byte loads/stores of the byte arrays
__global__ void __launch_bounds__(256) k_reversi_step(const u64* __restrict__ own, const u64* __restrict__ opp,
                                                      const uint8_t* __restrict__ action, int64_t n, u64 valid,
                                                      u64* __restrict__ own_next, u64* __restrict__ opp_next,
                                                      u64* __restrict__ legal_next, uint8_t* __restrict__ status,
                                                      int8_t* __restrict__ winner) {
    __shared__ RayEnt tab[256];
    build_ray_table(tab);
    __syncthreads();
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const ulonglong2* o2 = reinterpret_cast<const ulonglong2*>(own) + 2 * i;
        const ulonglong2* p2 = reinterpret_cast<const ulonglong2*>(opp) + 2 * i;
        ulonglong2 oa = o2[0], ob = o2[1], pa = p2[0], pb = p2[1];
        uchar4 ac = reinterpret_cast<const uchar4*>(action)[i];
        u64 me[4] = {oa.x, oa.y, ob.x, ob.y}, you[4] = {pa.x, pa.y, pb.x, pb.y};
        int aa[4] = {ac.x, ac.y, ac.z, ac.w};
        StepOut o[4]; u64 jx[4], jy[4];
        unsigned need = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) need |= step_main(me[k], you[k], aa[k], valid, tab, o[k], jx[k], jy[k]) ? 1u << k : 0u;
        // the games that need one more legal mask, one per lane per round (most lanes: none; a second round is very rare)
        while (__builtin_amdgcn_ballot_w64(need != 0)) {
            const int k = need ? __ffs(need) - 1 : 0;
            const u64 x = k == 0 ? jx[0] : (k == 1 ? jx[1] : (k == 2 ? jx[2] : jx[3]));
            const u64 y = k == 0 ? jy[0] : (k == 1 ? jy[1] : (k == 2 ? jy[2] : jy[3]));
            const u64 J = rev_legal(x, y, valid);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (need && k == q) step_finish(me[q], you[q], aa[q], J, o[q]);
            need &= need - 1u;
        }
        ulonglong2* on2 = reinterpret_cast<ulonglong2*>(own_next) + 2 * i;
        ulonglong2* pn2 = reinterpret_cast<ulonglong2*>(opp_next) + 2 * i;
        ulonglong2* ln2 = reinterpret_cast<ulonglong2*>(legal_next) + 2 * i;
        on2[0] = make_ulonglong2(o[0].cown, o[1].cown); on2[1] = make_ulonglong2(o[2].cown, o[3].cown);
        pn2[0] = make_ulonglong2(o[0].copp, o[1].copp); pn2[1] = make_ulonglong2(o[2].copp, o[3].copp);
        ln2[0] = make_ulonglong2(o[0].nl, o[1].nl); ln2[1] = make_ulonglong2(o[2].nl, o[3].nl);
        reinterpret_cast<uchar4*>(status)[i] = make_uchar4(o[0].st, o[1].st, o[2].st, o[3].st);
        reinterpret_cast<char4*>(winner)[i] = make_char4(o[0].w, o[1].w, o[2].w, o[3].w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // ragged tail
        int64_t i = (n4 << 2) + threadIdx.x;
        u64 co, cp, nl; uint8_t st; int8_t w;
        reversi_step_one(own[i], opp[i], action[i], valid, tab, co, cp, nl, st, w);
        own_next[i] = co; opp_next[i] = cp; legal_next[i] = nl; status[i] = st; winner[i] = w;
    }
}

__global__ void __launch_bounds__(256) k_reversi_legal(const u64* __restrict__ own, const u64* __restrict__ opp,
                                                       int64_t n, u64* __restrict__ legal) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        legal[i] = rev_legal8(own[i], opp[i]);
}

// get_score for n boards (reversi_board.py:67-85): winner = sign(n_x - n_o) and the two counts
__global__ void __launch_bounds__(256) k_reversi_score(const u64* __restrict__ x, const u64* __restrict__ o, int64_t n,
                                                       int8_t* __restrict__ winner, uint8_t* __restrict__ counts) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int cx = popc64(x[i]), co = popc64(o[i]);
        winner[i] = (int8_t)(cx > co ? 1 : (co > cx ? -1 : 0));
        reinterpret_cast<uchar2*>(counts)[i] = make_uchar2((unsigned char)cx, (unsigned char)co);
    }
}

__global__ void __launch_bounds__(256) k_ttt_step(const uint16_t* __restrict__ own, const uint16_t* __restrict__ opp,
                                                  const uint8_t* __restrict__ action, const int8_t* __restrict__ to_move,
                                                  int64_t n, uint16_t* __restrict__ own_next,
                                                  uint16_t* __restrict__ opp_next, uint16_t* __restrict__ legal_next,
                                                  uint8_t* __restrict__ status, int8_t* __restrict__ winner) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        u32 me = own[i] & 0x1FF, you = opp[i] & 0x1FF;
        int a = action[i], tm = to_move[i];
        bool ok = a < 9 && !(((me | you) >> a) & 1u);
        u32 cown = me, copp = you;
        uint8_t st = BZ_ST_ILLEGAL;
        int w = 0;
        if (ok) {
            cown = you; copp = me | (1u << a);
            u32 x = tm == 1 ? copp : cown, o = tm == 1 ? cown : copp;
            st = ttt_over(x, o, &w) ? BZ_ST_TERMINAL : BZ_ST_RUNNING;
        }
        own_next[i] = (uint16_t)cown; opp_next[i] = (uint16_t)copp;
        legal_next[i] = (uint16_t)(~(cown | copp) & 0x1FF);
        status[i] = st; winner[i] = (int8_t)w;
    }
}

// ---------------------------------------------------------------- D4 augmentation (SL/train.py:27-36)
// out[r][c] = x[src(r, c)] for the reference's 8 transforms, in its order:
//  0 id, 1 flip(dims=[0]), 2 flip(dims=[1]), 3 rot90(1), 4 rot90(2), 5 rot90(3), 6 t(), 7 flip(dims=[0]).t()
// (7 equals 5 in the reference -- torch semantics -- and is kept as is; its dedupe removes the copy)
__host__ __device__ inline void d4_src(int t, int n, int r, int c, int* sr, int* sc) {
    switch (t) {
    case 0: *sr = r; *sc = c; break;
    case 1: *sr = n - 1 - r; *sc = c; break;
    case 2: *sr = r; *sc = n - 1 - c; break;
    case 3: *sr = c; *sc = n - 1 - r; break;
    case 4: *sr = n - 1 - r; *sc = n - 1 - c; break;
    case 6: *sr = c; *sc = r; break;
    default: *sr = n - 1 - c; *sc = r; break;  // 5 and 7
    }
}

// thread = (row i, transform t): bitboards permuted bit by bit, pi permuted through the same map
__global__ void __launch_bounds__(256) k_augment_d4(const u64* __restrict__ own, const u64* __restrict__ opp,
                                                    const float* __restrict__ pi, int64_t n, int size, int na,
                                                    u64* __restrict__ own8, u64* __restrict__ opp8,
                                                    float* __restrict__ pi8, u64* __restrict__ key8) {
    int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (id >= n * 8) return;
    int64_t i = id >> 3;
    int t = (int)(id & 7);
    const int stride = size == 3 ? 3 : 8;
    u64 a = own[i], b = opp[i], oa = 0, ob = 0;
    const float* src = pi + i * na;
    float* dst = pi8 + id * na;
    u64 h = 0x243F6A8885A308D3ULL;
    for (int r = 0; r < size; ++r)
        for (int c = 0; c < size; ++c) {
            int sr, sc;
            d4_src(t, size, r, c, &sr, &sc);
            int sb = stride * sr + sc, db = stride * r + c;
            oa |= ((a >> sb) & 1ULL) << db;
            ob |= ((b >> sb) & 1ULL) << db;
            float p = src[size * sr + sc];
            dst[size * r + c] = p;
            h = (h ^ (u64)__float_as_uint(p)) * 0x100000001B3ULL;
        }
    for (int k = size * size; k < na; ++k) {  // actions beyond the board (Reversi's pass) do not move
        float p = src[k];
        dst[k] = p;
        h = (h ^ (u64)__float_as_uint(p)) * 0x100000001B3ULL;
    }
    own8[id] = oa; opp8[id] = ob;
    if (key8) {  // 64-bit content key of (s, pi) for the dedupe pass
        h ^= oa * 0x9E3779B97F4A7C15ULL;
        h = (h ^ (h >> 29)) * 0xBF58476D1CE4E5B9ULL;
        h ^= ob * 0xC2B2AE3D27D4EB4FULL;
        h = (h ^ (h >> 32)) * 0x94D049BB133111EBULL;
        key8[id] = h ^ (h >> 31);
    }
}

static int grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));  // <= 256 CUs x 8 blocks, grid-stride the rest
}

BZ_EXPORT int32_t bz_reversi_step_batch_sized(const uint64_t* own, const uint64_t* opp, const uint8_t* action, int64_t n,
                                              int32_t size, uint64_t* own_next, uint64_t* opp_next, uint64_t* legal_next,
                                              uint8_t* status, int8_t* winner, void* stream) {
    BZ_REQUIRE(n >= 0 && own && opp && action && own_next && opp_next && legal_next && status && winner,
               "bz_reversi_step_batch: null pointer");
    BZ_REQUIRE(size_ok(size), "bz_reversi_step_batch: size must be 1..8");
    if (n == 0) return BZ_OK;
    BZ_REQUIRE((((uintptr_t)own | (uintptr_t)opp | (uintptr_t)own_next | (uintptr_t)opp_next | (uintptr_t)legal_next) & 15) == 0 &&
                   (((uintptr_t)action | (uintptr_t)status | (uintptr_t)winner) & 3) == 0,
               "bz_reversi_step_batch: arrays must be 16-byte (u64) / 4-byte (u8) aligned");
    ProfScope ps(BZ_PROF_ENV_STEP, stream);
    hipLaunchKernelGGL(k_reversi_step, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, own, opp, action, n,
                       rev_valid(size), own_next, opp_next, legal_next, status, winner);
    BZ_LAUNCH_CHECK("k_reversi_step");
    return BZ_OK;
}
BZ_EXPORT int32_t bz_reversi_step_batch(const uint64_t* own, const uint64_t* opp, const uint8_t* action, int64_t n,
                                        uint64_t* own_next, uint64_t* opp_next, uint64_t* legal_next,
                                        uint8_t* status, int8_t* winner, void* stream) {
    return bz_reversi_step_batch_sized(own, opp, action, n, 8, own_next, opp_next, legal_next, status, winner, stream);
}
BZ_EXPORT int32_t bz_reversi_legal_batch(const uint64_t* own, const uint64_t* opp, int64_t n, uint64_t* legal,
                                         void* stream) {
    BZ_REQUIRE(n >= 0 && own && opp && legal, "bz_reversi_legal_batch: null pointer");
    if (n == 0) return BZ_OK;
    hipLaunchKernelGGL(k_reversi_legal, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, own, opp, n, legal);
    BZ_LAUNCH_CHECK("k_reversi_legal");
    return BZ_OK;
}
BZ_EXPORT int32_t bz_reversi_score_batch(const uint64_t* x, const uint64_t* o, int64_t n, int8_t* winner, uint8_t* counts,
                                         void* stream) {
    BZ_REQUIRE(n >= 0 && x && o && winner && counts, "bz_reversi_score_batch: null pointer");
    BZ_REQUIRE(((uintptr_t)counts & 1) == 0, "bz_reversi_score_batch: counts must be 2-byte aligned");
    if (n == 0) return BZ_OK;
    hipLaunchKernelGGL(k_reversi_score, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, o, n, winner, counts);
    BZ_LAUNCH_CHECK("k_reversi_score");
    return BZ_OK;
}
BZ_EXPORT int32_t bz_ttt_step_batch(const uint16_t* own, const uint16_t* opp, const uint8_t* action,
                                    const int8_t* to_move, int64_t n, uint16_t* own_next, uint16_t* opp_next,
                                    uint16_t* legal_next, uint8_t* status, int8_t* winner, void* stream) {
    BZ_REQUIRE(n >= 0 && own && opp && action && to_move && own_next && opp_next && legal_next && status && winner,
               "bz_ttt_step_batch: null pointer");
    if (n == 0) return BZ_OK;
    hipLaunchKernelGGL(k_ttt_step, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, own, opp, action, to_move,
                       n, own_next, opp_next, legal_next, status, winner);
    BZ_LAUNCH_CHECK("k_ttt_step");
    return BZ_OK;
}

BZ_EXPORT int32_t bz_augment_d4_batch(const uint64_t* own, const uint64_t* opp, const float* pi, int64_t n, int32_t size,
                                      int32_t na, uint64_t* own8, uint64_t* opp8, float* pi8, uint64_t* key8,
                                      void* stream) {
    BZ_REQUIRE(n >= 0 && own && opp && pi && own8 && opp8 && pi8, "bz_augment_d4_batch: null pointer");
    BZ_REQUIRE((size == 3 || size == 8) && na >= size * size, "bz_augment_d4_batch: size must be 3 or 8, na >= size*size");
    if (n == 0) return BZ_OK;
    int64_t blocks = (n * 8 + 255) / 256;
    hipLaunchKernelGGL(k_augment_d4, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, own, opp, pi, n, (int)size,
                       (int)na, own8, opp8, pi8, key8);
    BZ_LAUNCH_CHECK("k_augment_d4");
    return BZ_OK;
}
