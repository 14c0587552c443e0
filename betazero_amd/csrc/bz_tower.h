// bz_tower.h -- the bf16 MFMA conv-tower machinery shared by the inference kernel (bz_net.hip: k_tower_bf16) and the
// training kernels (bz_train.hip): geometry of the LDS-resident positions (Tw<C, P>), the weight-fragment stream, the
// K-loop of one conv3x3 layer (conv_layer) and the inference epilogue.  Everything here is a template or a
// __forceinline__ device function, so both translation units instantiate their own copies.
#pragma once
#include "bz_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace bz_tower {
using namespace bz;
// ------------------------------------------------------------------ bf16 MFMA tower
constexpr int kTC = 128;                  // channels of the benchmark net (the fp8 tower serves only this width)

// Geometry of the fused bf16 net kernel for C = 64, 128 or 256 channels.  A workgroup of 4 waves keeps P positions
// resident in LDS and its accumulator tiles (MT M-tiles of 32 output channels x P positions x 64 cells) are dealt
// to the waves as MW M-tiles x PW positions each, i.e. NU = 2 PW tiles of 32 cells ("units") per M-tile.
// Throughput shape (P = 512 / C: 2 buffers x P x TILE = 146 KB of LDS for every width, one workgroup per CU):
//   C =  64: P = 8, wave w -> M-tile  w & 1,        positions 4 (w >> 1) .. +3   (MW 1, PW 4)
//   C = 128: P = 4, wave w -> M-tile  w,            positions 0 .. 3             (MW 1, PW 4)   <- the benchmark net
//   C = 256: P = 2, wave w -> M-tiles 2w, 2w + 1,   positions 0, 1               (MW 2, PW 2)
// Latency shape for small batches (P = 1; 2 at C = 64): one position per workgroup -- a batch of up to 256 positions
// then spreads over as many CUs instead of 4 positions sharing one (an interactive MCTSPlayer search is B = 1: the
// throughput shape would compute three padding positions for every real one).
//
// Which 32 cells make a unit (ROWT).  With four positions per wave a unit is ONE BOARD ROW of the four positions
// (lane r -> position r >> 3, column r & 7; unit u = row u) instead of four rows of one position.  A conv tap with row
// shift dy = -1 then reads nothing but zero padding for the whole of unit 0 (dy = +1: unit 7), and that MFMA -- and its
// LDS read -- is skipped: 6 of the 9 taps issue 7 MFMAs per k-step instead of 8, 8.3 % of the tower's matrix work.  The
// skipped products are exact zeros, so every output is bit-identical to the full sum.  (Padding along x cannot be
// skipped the same way: a 32-cell unit cannot be a board column AND a board row.)  Waves with fewer than four
// positions keep the position-major units (lane r -> row 4 nt + (r >> 3), column r & 7 of one position).
//
// LDS image of a position: 8 board rows at a pitch of 9 cells -- 8 squares and one all-zero cell, which is the x = 8
// halo of its row and the x = -1 halo of the next one -- plus one leading zero cell: 73 cells.  A tap's column shift
// is then a plain address offset for every lane (no per-lane halo test).  Rows -1 and 8 do not exist: row-tile units
// skip them, position-major units point the affected lanes at a zero cell.
// (Round 1 measured 2 positions x 2 workgroups per CU, a tile-major last tap and the 16x16x32 MFMA shape, round 2 the
// MW = 2 split at C = 128: all within +-1 % because the kernel sits on the power-limited clock -- DESIGN.md 5.)
template <int C_, int P_ = 512 / C_, bool M16_ = false> struct Tw {
    static_assert(C_ == 64 || C_ == 128 || C_ == 256, "the MFMA tower is built for 64, 128 or 256 channels");
    static constexpr int C = C_;
    // M16: the K-loop on v_mfma_f32_16x16x32_bf16 instead of 32x32x16 (the "16x16x32 path" at the end of this file):
    // the same LDS image, the same bytes per MAC from LDS and L2, the same cycles -- and a higher clock under load
    static constexpr bool M16 = M16_;
    static constexpr int KQ = C / 32;                  // M16: steps of 32 input channels per conv tap
    static constexpr int KC = C / 16;                  // k-steps (16 input channels) per conv tap
    static constexpr int MT = C / 32;                  // M-tiles (32 output channels)
    static constexpr int CELL = 2 * C;                 // bytes of one board cell (all channels, bf16)
    static constexpr int ROWC = 9;                     // cells per board-row pitch: 8 squares + 1 zero cell
    static constexpr int NCELL = 8 * ROWC + 1;         // + the zero cell in front of row 0
    static constexpr int TILE = NCELL * CELL;          // bytes of one position
    static constexpr int P = P_;                       // positions resident per workgroup
    static constexpr int BUF = P * TILE;
    static constexpr int LDS = 2 * BUF;
    static constexpr int MW = MT >= 8 ? 2 : 1;         // M-tiles per wave (MW = 2 at C = 128: +-0.4 %, three A/Bs)
    static constexpr int PW = MT * P / 4 / MW;         // positions per wave
    static_assert(MW * PW * 4 == MT * P && PW >= 1, "the (M-tile, position) units must split evenly over 4 waves");
    static constexpr int NG = MT / MW;                 // wave groups along M
    static constexpr int NU = 2 * PW;                  // 32-cell units per M-tile of a wave
#ifdef BZ_EXP_NO_ROWT  // diagnostic A/B (position-major units for every shape: no skipped MFMAs)
    static constexpr bool ROWT = false;
#else
    static constexpr bool ROWT = PW == 4;              // units are board rows across the wave's four positions
#endif
    static constexpr int KS = KC < 8 ? KC : 8;         // k-steps per weight-prefetch chunk (register set)
    // activation-fragment buffers: fetched NBUF - 1 k-steps ahead (3 buffers and 3 weight sets were A/B'd on the
    // row-tile kernel: no change -- neither LDS nor L2 latency is what the K-loop waits for)
    static constexpr int NBUF = 2;
    static constexpr int CPT = KC / KS;                // chunks per tap
    static constexpr int NCH = 9 * CPT;                // chunks per layer
    // register sets of weight fragments = how far ahead the weight stream is fetched (DEPTH - 1 chunks).  The
    // throughput shapes issue 8 MFMAs per k-step, so one chunk ahead is 2048+ cycles -- beyond the L2 latency; the
    // latency shapes issue 2, one chunk is only ~512 cycles, so they fetch two chunks ahead (three sets)
    static constexpr int DEPTH = MW * PW <= 2 ? 3 : 2;
    static_assert((2 * NCH) % DEPTH == 0, "a residual block must bring the register-set rotation back to set 0");
    static __device__ __forceinline__ int wt0(int w) { return (w % NG) * MW; }
    static __device__ __forceinline__ int pos0(int w) { return (w / NG) * PW; }
    // byte offset of cell (y, x) inside a position, x = -1 .. 8 (the ends are zero cells)
    static __device__ __forceinline__ constexpr int cell_at(int y, int x) { return (y * ROWC + x + 1) * CELL; }
    // XOR swizzle of the 16-byte chunk index inside a cell, chosen so that the 16 lanes of every ds_read_b128 lane
    // group (8 columns x 2 values of `sel`, any tap shift) hit 16 distinct 16-byte bank slots: 256-B and 512-B cells
    // span whole bank rows -> 4 bits from (x, sel); 128-B cells share a bank row in pairs (cell-index parity = x
    // parity within the group) -> 3 bits from (x >> 1, sel).  sel = the other lane coordinate of a unit: the
    // position (row-tile units) or the board row (position-major units).
    static __device__ __forceinline__ int sw(int sel, int xx) {
        return C == 64 ? (((xx & 7) >> 1) | ((sel & 1) << 2)) : ((xx & 7) | ((sel & 1) << 3));
    }
    // M16 (the 16x16x32 path) places chunk ch of a cell in column xx at 16-byte position sigma((ch + 2 xx) mod 16),
    // sigma(s) = s >> 1 | (s & 1) << 3 -- additive in the column, so that a tap's column shift moves every lane's slot by
    // the same amount: in a ds_read_b128 lane group the two positions of a column differ in the parity of the reader's
    // k-group (slots 2h, 2h + 1) and the eight columns in h, for EVERY tap (the XOR swizzle above pairs lanes across
    // k-groups differently for shifted taps: 41 % conflict cycles when the 16x16x32 path first ran on it); sigma keeps
    // the 8-byte epilogue stores of a 16-lane group on 8 distinct 16-byte positions mod 128 B (2-way, as before).
    static __device__ __forceinline__ int pos16(int ch, int xx) {
        const int s = (ch + 2 * (xx & 7)) & 15;
        return ((s >> 1) | ((s & 1) << 3)) << 4;
    }
    // byte offset of 16-byte chunk k of board cell c of the workgroup's position p, inside that position
    static __device__ __forceinline__ int cell_off(int p, int c, int k) {
        if constexpr (M16) return cell_at(c >> 3, c & 7) + pos16(k, c & 7);
        else return cell_at(c >> 3, c & 7) + ((k ^ sw(ROWT ? p : c >> 3, c & 7)) << 4);
    }
    // unit u, lane column r (0..31) -> position inside the wave and board cell
    static __device__ __forceinline__ int unit_pos(int u, int r) { return ROWT ? r >> 3 : u >> 1; }
    static __device__ __forceinline__ int unit_cell(int u, int r) { return ROWT ? 8 * u + (r & 7) : 32 * (u & 1) + r; }
    // LDS offset of that cell = lane_home(r) + unit_imm(u): a per-lane part and a compile-time part
    static __device__ __forceinline__ int lane_home(int r) {
        return ROWT ? (r >> 3) * TILE + cell_at(0, r & 7) : cell_at(r >> 3, r & 7);
    }
    static __device__ __forceinline__ constexpr int unit_imm(int u) {
        return ROWT ? u * ROWC * CELL : (u >> 1) * TILE + (u & 1) * 4 * ROWC * CELL;
    }
    // units that read at least one board row for a tap with row shift dy: [unit_lo, unit_hi)
    static __device__ __forceinline__ constexpr int unit_lo(int dy) { return ROWT && dy < 0 ? 1 : 0; }
    static __device__ __forceinline__ constexpr int unit_hi(int dy) { return ROWT && dy > 0 ? NU - 1 : NU; }
};

// Diagnostic build only (tools/exp_stamps.sh -> a separate libbz_hip.stamps.so, never the product .so)
#if defined(BZ_EXP_STAMPS_TAPS) && !defined(BZ_EXP_STAMPS)
#error "BZ_EXP_STAMPS_TAPS needs BZ_EXP_STAMPS"
#endif
#if defined(BZ_EXP_STAMPS) && !defined(BZ_EXPERIMENT)
#error "BZ_EXP_STAMPS is a diagnostic variant: build it through betazero_amd.build.build_variant()"
#endif
#if (defined(BZ_EXP_NOPS) || defined(BZ_EXP_NOP1) || defined(BZ_EXP_NO_ROWT) || defined(BZ_EXP_NO_LAYER_BARRIER) || defined(BZ_EXP_MFMA16) || defined(BZ_EXP_MFMA_AMAJOR) || defined(BZ_EXP_EPILOGUE_HALF)) && !defined(BZ_EXPERIMENT)
#error "BZ_EXP_NOPS is a diagnostic variant: build it through betazero_amd.build.build_variant()"
#endif
// weight-fragment loads of the bf16 tower.  Diagnostic option BZ_EXP_WEIGHTS_NT: non-temporal loads, to see whether the
// 3.5 MB of fragments can pass through each XCD's 4-MB L2 without evicting the tree that the next tree step walks
#ifdef BZ_EXP_WEIGHTS_NT
#ifndef BZ_EXPERIMENT
#error "BZ_EXP_WEIGHTS_NT is a diagnostic variant: build it through betazero_amd.build.build_variant()"
#endif
typedef unsigned bz_u32x4 __attribute__((ext_vector_type(4)));
#define BZ_WLOAD(p) __builtin_nontemporal_load(reinterpret_cast<const bz_u32x4*>(p))
#else
#define BZ_WLOAD(p) (*(p))
#endif
#ifdef BZ_EXP_STAMPS
// (one copy per translation unit; bz_debug_read in bz_net.hip reads the inference kernels' copy)
static __device__ unsigned long long g_dbg[8 * 4096];
#define BZ_STAMP(var) do { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); var = _t; } while (0)
#else
#define BZ_STAMP(var) do { } while (0)
#endif

// Per-lane LDS byte offsets of the B-operand chunk h of k-step 0 for conv tap TAP, relative to the wave's first
// position; load_b XORs the k-step in and adds the unit's compile-time offset.
//  * row-tile units: ONE offset (the lane's position and column x + dx; the halo columns are zero cells of the
//    layout); the row (u + dy) is part of the compile-time offset.
//  * position-major units: one offset per cell tile nt (rows 4 nt + (r >> 3) + dy); a lane whose row falls off the board
//    reads a zero cell.  The swizzle always comes from the UNCLAMPED coordinates, so a halo lane reads the slot that
//    its virtual cell would occupy and the 16 lanes of a ds_read_b128 group still hit 16 distinct slots.
template <class G, int TAP>
__device__ __forceinline__ void tap_off(int r, int h, int (&boff)[2]) {  // row-tile units
    constexpr int dx = TAP % 3 - 1;
    const int xx = (r & 7) + dx;
    boff[0] = (r >> 3) * G::TILE + G::cell_at(0, xx) + ((G::sw(r >> 3, xx) ^ h) << 4);
    boff[1] = 0;
}
template <class G>
__device__ __forceinline__ void tap_off_pm(int tap, int r, int h, int (&boff)[2]) {  // position-major units
    const int dy = tap / 3 - 1, dx = tap % 3 - 1, xx = (r & 7) + dx;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        int yy = 4 * nt + (r >> 3) + dy;
        bool inb = (unsigned)yy < 8u;
        // 128-B cells pair up in a bank row: take the zero cell (index 0 or 9) with the parity of the virtual cell
        int zero = G::C == 64 && ((yy + xx + 1) & 1) ? G::ROWC * G::CELL : 0;
        boff[nt] = (inb ? G::cell_at(yy, xx) : zero) + ((G::sw(yy, xx) ^ h) << 4);
    }
}
// activation fragments of k-step kc for the units tap TAP needs (position-major units: all of them, whatever TAP)
template <class G, int TAP>
__device__ __forceinline__ void load_b(bf16x8 (&b)[G::NU], const char* in, const int (&boff)[2], int kc) {
    constexpr int dy = TAP / 3 - 1;
#pragma unroll
    for (int u = G::unit_lo(dy); u < G::unit_hi(dy); ++u) {
        if constexpr (G::ROWT)
            b[u] = *reinterpret_cast<const bf16x8*>(in + (boff[0] ^ (kc << 5)) + G::unit_imm(u) + dy * G::ROWC * G::CELL);
        else  // chunk 2 kc + h: the XOR stays inside the cell
            b[u] = *reinterpret_cast<const bf16x8*>(in + (boff[u & 1] ^ (kc << 5)) + (u >> 1) * G::TILE);
    }
}
template <class G, int TAP>
__device__ __forceinline__ void mfma_units(f32x16 (&acc)[G::MW][G::NU], const bf16x8 (&a)[G::MW], const bf16x8 (&b)[G::NU],
                                           [[maybe_unused]] int kpar = 0) {
    constexpr int dy = TAP / 3 - 1;
#pragma unroll
    for (int mt = 0; mt < G::MW; ++mt)
#pragma unroll
        for (int u = G::unit_lo(dy); u < G::unit_hi(dy); ++u)
        {
#ifdef BZ_EXP_MFMA16
            // TIMING ONLY (the results are wrong): the same operand registers, LDS and weight traffic, but the matrix
            // work of a 32x32x16 MFMA issued as TWO v_mfma_f32_16x16x32_bf16 (same MACs, 2 x 16 cycles instead of 32),
            // accumulating into the quarters of the same 16 registers (even k-steps: quarters 0, 1; odd: 2, 3) -- what a
            // kernel built on the 16x16x32 shape with a 32 co x (2 x 16 cells) wave tile would issue per 1-KB weight
            // fragment.  tools/exp_ab_mfma16.sh: does the chip hold a higher clock on that shape (MI355X_MICROARCH.md,
            // DVFS item 7)?
            {
                f32x16& c = acc[mt][u];
                f32x4 q0 = kpar ? f32x4{c[8], c[9], c[10], c[11]} : f32x4{c[0], c[1], c[2], c[3]};
                f32x4 q1 = kpar ? f32x4{c[12], c[13], c[14], c[15]} : f32x4{c[4], c[5], c[6], c[7]};
                q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b[u], q0, 0, 0, 0);
                q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[u], a[mt], q1, 0, 0, 0);  // (operands swapped: the compiler merges two identical products into one)
                if (kpar) { c[8] = q0[0]; c[9] = q0[1]; c[10] = q0[2]; c[11] = q0[3]; c[12] = q1[0]; c[13] = q1[1]; c[14] = q1[2]; c[15] = q1[3]; }
                else { c[0] = q0[0]; c[1] = q0[1]; c[2] = q0[2]; c[3] = q0[3]; c[4] = q1[0]; c[5] = q1[1]; c[6] = q1[2]; c[7] = q1[3]; }
            }
#else
            acc[mt][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[u], acc[mt][u], 0, 0, 0);
#endif
#ifdef BZ_EXP_NOPS  // diagnostic duty sweep (tools/exp_duty_sweep.sh): BZ_EXP_NOPS x 8 idle issue cycles behind every MFMA
#pragma unroll
            for (int z = 0; z < BZ_EXP_NOPS; ++z) asm volatile("s_nop 7");
#endif
#ifdef BZ_EXP_NOP1  // finer steps: BZ_EXP_NOP1 x 1 idle issue cycle
#pragma unroll
            for (int z = 0; z < BZ_EXP_NOP1; ++z) asm volatile("s_nop 0");
#endif
        }
}

// One weight chunk CC = KS k-steps of up to 8 MFMAs (a whole conv tap at C <= 128, half a tap at C = 256); the tap is a
// compile-time constant, so the units it skips cost nothing.  The chunk's weight fragments are in register set S of
// DEPTH; the set freed by the previous chunk is filled for the chunk DEPTH - 1 ahead (coalesced 1 KB loads; the fragment
// stream is linear over chunks, taps and layers).  Activation fragments are double-buffered: the ds_read_b128
// of k-step k+1 are issued between the MFMAs of k-step k.  `in` points at the wave's first position; boff addresses
// this chunk's tap and is replaced by the next chunk's on exit.
template <class G> struct WSets { bf16x8 s[G::DEPTH][G::KS][G::MW]; };  // the weight-fragment register sets

template <int S, int CC, class G>
__device__ __forceinline__ void chunk_step(f32x16 (&acc)[G::MW][G::NU], WSets<G>& WS, const uint4*& ap, const char* in,
                                           int (&boff)[2], int r, int h, bf16x8 (&B)[G::NBUF][G::NU]) {
    constexpr int TAP = CC / G::CPT, kc0 = (CC % G::CPT) * G::KS;
    constexpr bool last = CC + 1 >= G::NCH;  // the layer's last chunk has no successor to read ahead for
    constexpr int TAP_N = last ? TAP : (CC + 1) / G::CPT, kc0_n = last ? 0 : ((CC + 1) % G::CPT) * G::KS;
    bf16x8 (&use)[G::KS][G::MW] = WS.s[S];
    bf16x8 (&nxt)[G::KS][G::MW] = WS.s[(S + G::DEPTH - 1) % G::DEPTH];  // the set the previous chunk has just freed
#pragma unroll
    for (int kc = 0; kc < G::KS; ++kc)
#pragma unroll
        for (int mt = 0; mt < G::MW; ++mt) nxt[kc][mt] = __builtin_bit_cast(bf16x8, BZ_WLOAD(&ap[(kc * G::MT + mt) * 64 + (unsigned)(32 * h + r)]));
    ap += G::KS * G::MT * 64;
    static_assert(G::ROWT, "compile-time taps are for row-tile units");
    int boff_n[2] = {boff[0], boff[1]};
    if constexpr (TAP_N % 3 != TAP % 3) tap_off<G, TAP_N>(r, h, boff_n);
    // k-step k of this chunk sits in buffer (base + k) % NBUF; the fragments of k-step k + D are fetched while k runs
    constexpr int NB = G::NBUF, D = NB - 1, base = (CC * G::KS) % NB;
#pragma unroll
    for (int k = 0; k < G::KS; ++k) {
        if (k + D < G::KS) load_b<G, TAP>(B[(base + k + D) % NB], in, boff, kc0 + k + D);
        else if constexpr (!last) load_b<G, TAP_N>(B[(base + k + D) % NB], in, boff_n, kc0_n + k + D - G::KS);  // next chunk
        mfma_units<G, TAP>(acc, use[k], B[(base + k) % NB], k & 1);
    }
    constexpr int NA = G::unit_hi(TAP / 3 - 1) - G::unit_lo(TAP / 3 - 1);
#pragma unroll
    for (int i = 0; i < G::KS; ++i) {
        if (G::MW == 1) {
#pragma unroll
            for (int j = 0; j < NA; ++j) {
#ifdef BZ_EXP_MFMA16
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // the unit's 2 half-size MFMAs
#else
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
#endif
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
            }
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read (weight prefetch)
        } else {
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                if (j & 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 2 VMEM reads per k-step
            }
        }
    }
    boff[0] = boff_n[0]; boff[1] = boff_n[1];
    // the layer is straight-line code now: keep one chunk per scheduling region (the group-barrier solver is
    // super-linear in the region size)
    __builtin_amdgcn_sched_barrier(0);
}
// The same for position-major units (nothing to skip): the tap is a run-time value, so a layer is a loop over
// chunks unrolled by DEPTH only (straight-line layers cost these shapes their register budget).
template <int S, class G>
__device__ __forceinline__ void chunk_step_pm(f32x16 (&acc)[G::MW][G::NU], WSets<G>& WS, const uint4*& ap, const char* in,
                                              int (&boff)[2], int kc0, int tap_n, int kc0_n, int r, int h,
                                              bf16x8 (&b0)[G::NU], bf16x8 (&b1)[G::NU]) {
    constexpr int ANY = 4;  // every tap uses all units
    bf16x8 (&use)[G::KS][G::MW] = WS.s[S];
    bf16x8 (&nxt)[G::KS][G::MW] = WS.s[(S + G::DEPTH - 1) % G::DEPTH];
#pragma unroll
    for (int kc = 0; kc < G::KS; ++kc)
#pragma unroll
        for (int mt = 0; mt < G::MW; ++mt) nxt[kc][mt] = __builtin_bit_cast(bf16x8, BZ_WLOAD(&ap[(kc * G::MT + mt) * 64 + (unsigned)(32 * h + r)]));
    ap += G::KS * G::MT * 64;
    int boff_n[2];
    tap_off_pm<G>(tap_n, r, h, boff_n);
#pragma unroll
    for (int k2 = 0; k2 < G::KS / 2; ++k2) {
        load_b<G, ANY>(b1, in, boff, kc0 + 2 * k2 + 1);
        mfma_units<G, ANY>(acc, use[2 * k2], b0);
        if (k2 < G::KS / 2 - 1) load_b<G, ANY>(b0, in, boff, kc0 + 2 * k2 + 2);
        else load_b<G, ANY>(b0, in, boff_n, kc0_n);  // first k-step of the next chunk
        mfma_units<G, ANY>(acc, use[2 * k2 + 1], b1);
    }
#pragma unroll
    for (int i = 0; i < G::KS; ++i) {
        if (G::MW == 1) {
#pragma unroll
            for (int j = 0; j < G::NU; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
            }
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read (weight prefetch)
        } else {
#pragma unroll
            for (int j = 0; j < G::NU; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                if (j & 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 2 VMEM reads per k-step
            }
        }
    }
    boff[0] = boff_n[0]; boff[1] = boff_n[1];
}
template <int S0, int CC, class G>
__device__ __forceinline__ void run_chunks(f32x16 (&acc)[G::MW][G::NU], WSets<G>& WS, const uint4*& ap, const char* in,
                                           int (&boff)[2], int r, int h, bf16x8 (&B)[G::NBUF][G::NU]) {
    if constexpr (CC < G::NCH) {  // the register-set index and the tap must be compile-time constants
#ifdef BZ_EXP_STAMPS_TAPS  // per-tap cycles of workgroup 0, wave 0 (each stamp drains the LDS queue: it perturbs the layer totals)
        unsigned long long c0, c1;
        BZ_STAMP(c0);
#endif
        chunk_step<(S0 + CC) % G::DEPTH, CC, G>(acc, WS, ap, in, boff, r, h, B);
#ifdef BZ_EXP_STAMPS_TAPS
        BZ_STAMP(c1);
        if (blockIdx.x == 0 && threadIdx.x == 0) { g_dbg[8 * 2048 + CC] += c1 - c0; g_dbg[8 * 2048 + 32 + CC] += 1; }
#endif
        run_chunks<S0, CC + 1, G>(acc, WS, ap, in, boff, r, h, B);
    }
}

// The inference epilogue (EpInfer below): +bias (+skip) -> ReLU -> bf16 -> LDS.  D[row = co][col = lane column r of unit u]: lane (r, h) register 4q+i of
// M-tile wt holds co = 32wt + 8q + 4h + i of that cell, i.e. 4 consecutive channels = one 8-byte store.
// `out` points at the wave's first position.
// the layer's biases for the lane's channels.  Fetched BEFORE the K-loop: issued at the head of the epilogue they cost
// it a full L2 round trip with nothing to overlap (the per-tap scheduling regions keep the compiler from hoisting them)
template <class G> struct Bias { f32x4 q[G::MW][4]; };
template <class G>
__device__ __forceinline__ void load_bias(Bias<G>& b, const float* __restrict__ bl, int wt0, int h) {
#pragma unroll
    for (int mt = 0; mt < G::MW; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) b.q[mt][q] = *reinterpret_cast<const f32x4*>(bl + 32 * (wt0 + mt) + 4 * h + 8 * q);
}
template <class G>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[G::MW][G::NU], char* out, bool second,
                                         const Bias<G>& bias, int wt0, int r, int h) {
    const int swz = G::sw(r >> 3, r & 7);
    // two opaque bases (even / odd units): every store offset is then a multiple of 512 B from its base, which lets
    // pairs of 8-byte stores (and skip loads) go out as one ds_write2st64_b64 / ds_read2st64_b64
    int home2[2] = {G::lane_home(r) + 8 * h, G::lane_home(r) + 8 * h + G::unit_imm(1)};
    asm volatile("" : "+v"(home2[0]), "+v"(home2[1]));
#pragma unroll
    for (int mt = 0; mt < G::MW; ++mt) {
        const int wt = wt0 + mt;
        const f32x4 (&bq)[4] = bias.q[mt];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int par = 0; par < 2; ++par)
#pragma unroll
                for (int u = par; u < G::NU; u += 2) {  // units of one parity back to back: their stores pair up
                    int off = G::unit_imm(u & ~1) + home2[par] + (((4 * wt + q) ^ swz) << 4);
                    f32x4 v = {acc[mt][u][4 * q], acc[mt][u][4 * q + 1], acc[mt][u][4 * q + 2], acc[mt][u][4 * q + 3]};
                    v = v + bq[q];
                    if (second) {  // conv2 of a block writes X in place: the skip is what it overwrites
                        bf16x4 sk = *reinterpret_cast<const bf16x4*>(out + off);
                        v = v + __builtin_convertvector(sk, f32x4);
                    }
                    // ReLU on the bf16 bit patterns: rounding keeps the sign, so max(int16 bits, 0) of the rounded value
                    // = the rounded max(v, 0) bit for bit (-0 -> +0 included), at two packed ops per four channels
                    // (two 2-element conversions: one v_cvt_pk_bf16_f32 each; the 4-element form converts every
                    // value on its own and packs with v_perm)
                    f32x2 vlo = {v[0], v[1]}, vhi = {v[2], v[3]};
                    s16x2 lo = __builtin_bit_cast(s16x2, __builtin_convertvector(vlo, bf16x2));
                    s16x2 hi = __builtin_bit_cast(s16x2, __builtin_convertvector(vhi, bf16x2));
                    lo = __builtin_elementwise_max(lo, (s16x2)(0));
                    hi = __builtin_elementwise_max(hi, (s16x2)(0));
                    *reinterpret_cast<uint2*>(out + off) = make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
                }
    }
}

// ------------------------------------------------------------------ the 16x16x32 path (Tw<C, P, true>)
// Same wave tile as the 32x32x16 path -- 32 output channels x the 8 board rows of 4 positions -- cut into 16 x 16 pieces:
// a unit's accumulator f32x16 holds four 16 x 16 tiles, quarter 2a + b = output channels 32 wt + 16 a .. + 15 of the
// unit's cells 16 b .. 16 b + 15 (positions 2 b and 2 b + 1 of the row).  Lane (c = lane & 15, g = lane >> 4):
//   A (weights)     : row co = 16 a + c, k = 8 g + j  -> one 1-KB fragment per (32 input channels, a)
//   B (activations) : k = 8 g + j of cell c of half b -> ds_read_b128 of chunk 4 kq + g of that cell
//   D quarter (a, b): channel 16 a + 4 g + i (i = register), cell 16 b + c
// A "sub-step" (kq, b) = 32 input channels x one half of every unit: 8 ds_read_b128 + 16 MFMAs (2 per unit: a = 0, 1),
// i.e. the (2 MFMA, 1 DS read) pattern; a tap is 2 KQ sub-steps and 2 KQ weight fragments -- the bytes per MAC from LDS
// and from L2 are those of the 32x32x16 path, and so are the cycles (an MFMA of this shape is 16 cycles for half the
// MACs).  What changes is the clock the chip holds: MI355X_MICROARCH.md (DVFS give-back, item 7) reports 1.12-1.15 x the
// FLOP/s for this shape at equal cycles; measured on this kernel 1.86 -> 2.00 GHz (profiles/r04_ab_tower_16x16.txt).
// The LDS image keeps its cells (9 per row, halo included) but places a cell's chunks by Tw::pos16 instead of the XOR
// swizzle: the 16 lanes of a ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...) are the 16 (position, column) cells of a
// half-unit, the two positions of a column read by k-groups of opposite parity -- pos16 keeps them on 16 distinct slots
// for every tap.
template <class G> struct WSets16 { bf16x8 s[2][G::KQ][2]; };  // two register sets x KQ steps x 2 channel halves

// the lane's B offsets for the KQ steps of tap TAP (half b = 0, board row 0): chunk 4 kq + g of the cell in column x + dx
template <class G, int TAP>
__device__ __forceinline__ void tap_off16(int c, int g, int (&boff)[G::KQ]) {
    constexpr int dx = TAP % 3 - 1;
    const int xx = (c & 7) + dx;
    const int cell = (c >> 3) * G::TILE + G::cell_at(0, xx);
#pragma unroll
    for (int kq = 0; kq < G::KQ; ++kq) boff[kq] = cell + G::pos16(4 * kq + g, xx);
}
template <class G, int TAP>
__device__ __forceinline__ void load_b16(bf16x8 (&b)[G::NU], const char* in, int boff, int half) {
    constexpr int dy = TAP / 3 - 1;
#pragma unroll
    for (int u = G::unit_lo(dy); u < G::unit_hi(dy); ++u)
        b[u] = *reinterpret_cast<const bf16x8*>(in + boff + half * 2 * G::TILE + G::unit_imm(u) + dy * G::ROWC * G::CELL);
}
template <int Q>
__device__ __forceinline__ void mfma16_quarter(f32x16& c, const bf16x8& a, const bf16x8& b) {
    f32x4 q = {c[4 * Q], c[4 * Q + 1], c[4 * Q + 2], c[4 * Q + 3]};
    q = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, q, 0, 0, 0);
    c[4 * Q] = q[0]; c[4 * Q + 1] = q[1]; c[4 * Q + 2] = q[2]; c[4 * Q + 3] = q[3];
}
template <class G, int TAP, int HALF>
__device__ __forceinline__ void mfma_units16(f32x16 (&acc)[G::MW][G::NU], const bf16x8 (&a)[2], const bf16x8 (&b)[G::NU]) {
    constexpr int dy = TAP / 3 - 1;
#ifdef BZ_EXP_MFMA_AMAJOR
    // diagnostic A/B (tools/exp_ab_mfma_order.sh): the same MFMAs, channel-half-major -- consecutive MFMAs then share the
    // WEIGHT operand and change the activation operand, instead of sharing the activations and alternating the weights.
    // Every accumulator quarter sees the same sequence of products: outputs are bit-identical.  The kernel is power-bound
    // (1320 W at 2.0 GHz on random data against 985 W at 2.4 GHz on zero weights, profiles/r01_power_clock_rocm_smi.txt), so
    // the question is whether one issue order costs less energy per MFMA than the other.
#pragma unroll
    for (int u = G::unit_lo(dy); u < G::unit_hi(dy); ++u) mfma16_quarter<HALF>(acc[0][u], a[0], b[u]);
#pragma unroll
    for (int u = G::unit_lo(dy); u < G::unit_hi(dy); ++u) mfma16_quarter<2 + HALF>(acc[0][u], a[1], b[u]);
#else
#pragma unroll
    for (int u = G::unit_lo(dy); u < G::unit_hi(dy); ++u) {
        mfma16_quarter<HALF>(acc[0][u], a[0], b[u]);      // a = 0: quarter b
        mfma16_quarter<2 + HALF>(acc[0][u], a[1], b[u]);  // a = 1: quarter 2 + b
    }
#endif
}
// one conv tap: 2 KQ sub-steps; the register set freed by the previous tap is filled for the next one
template <int S, int TAP, class G>
__device__ __forceinline__ void tap_step16(f32x16 (&acc)[G::MW][G::NU], WSets16<G>& WS, const uint4*& ap, const char* in,
                                           int (&boff)[G::KQ], int lane, bf16x8 (&B)[2][G::NU]) {
    constexpr bool last = TAP == 8;
    constexpr int TAP_N = last ? TAP : TAP + 1;
    bf16x8 (&use)[G::KQ][2] = WS.s[S];
    bf16x8 (&nxt)[G::KQ][2] = WS.s[S ^ 1];
#pragma unroll
    for (int kq = 0; kq < G::KQ; ++kq)
#pragma unroll
        for (int a = 0; a < 2; ++a) nxt[kq][a] = __builtin_bit_cast(bf16x8, BZ_WLOAD(&ap[(kq * G::MT * 2 + a) * 64 + (unsigned)lane]));
    ap += G::KQ * G::MT * 2 * 64;
    int boff_n[G::KQ];
#pragma unroll
    for (int kq = 0; kq < G::KQ; ++kq) boff_n[kq] = boff[kq];
    if constexpr (TAP_N % 3 != TAP % 3) tap_off16<G, TAP_N>(lane & 15, lane >> 4, boff_n);
    constexpr int NS = 2 * G::KQ;  // sub-steps; sub-step ss sits in buffer ss & 1 (NS is even: every tap starts in buffer 0)
#pragma unroll
    for (int ss = 0; ss < NS; ++ss) {
        if (ss + 1 < NS) load_b16<G, TAP>(B[(ss + 1) & 1], in, boff[(ss + 1) >> 1], (ss + 1) & 1);
        else if constexpr (!last) load_b16<G, TAP_N>(B[0], in, boff_n[0], 0);  // first sub-step of the next tap
        if (ss & 1) mfma_units16<G, TAP, 1>(acc, use[ss >> 1], B[1]);
        else mfma_units16<G, TAP, 0>(acc, use[ss >> 1], B[0]);
    }
    constexpr int NA = G::unit_hi(TAP / 3 - 1) - G::unit_lo(TAP / 3 - 1);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // the unit's 2 MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read (weight prefetch)
    }
#pragma unroll
    for (int kq = 0; kq < G::KQ; ++kq) boff[kq] = boff_n[kq];
    __builtin_amdgcn_sched_barrier(0);  // one tap per scheduling region
}
template <int S0, int TAP, class G>
__device__ __forceinline__ void run_taps16(f32x16 (&acc)[G::MW][G::NU], WSets16<G>& WS, const uint4*& ap, const char* in,
                                           int (&boff)[G::KQ], int lane, bf16x8 (&B)[2][G::NU]) {
    if constexpr (TAP < 9) {
        tap_step16<(S0 + TAP) & 1, TAP, G>(acc, WS, ap, in, boff, lane, B);
        run_taps16<S0, TAP + 1, G>(acc, WS, ap, in, boff, lane, B);
    }
}
// the layer's biases for the lane's channels 16 a + 4 g + i (kept in Bias<G>::q[0][a])
template <class G>
__device__ __forceinline__ void load_bias16(Bias<G>& b, const float* __restrict__ bl, int wt, int g) {
#pragma unroll
    for (int a = 0; a < 2; ++a) b.q[0][a] = *reinterpret_cast<const f32x4*>(bl + 32 * wt + 16 * a + 4 * g);
    b.q[0][2] = b.q[0][3] = (f32x4)(0.0f);
}
// +bias (+skip) -> ReLU -> bf16 -> LDS for the 16x16 quarters.  `out` points at the wave's first position.
template <class G>
__device__ __forceinline__ void epilogue16(f32x16 (&acc)[G::MW][G::NU], char* out, bool second, const Bias<G>& bias, int wt,
                                           int lane) {
    const int c = lane & 15, g = lane >> 4, x = c & 7, pl = c >> 3;
    // per-lane bases of unit 0 / unit 1 (even / odd rows) of the lane's position in half b = 0; + 8 (g & 1): the lane's 4
    // channels are the lower or upper half of a 16-byte chunk
    int home2[2] = {pl * G::TILE + G::cell_at(0, x) + 8 * (g & 1), pl * G::TILE + G::cell_at(1, x) + 8 * (g & 1)};
    asm volatile("" : "+v"(home2[0]), "+v"(home2[1]));
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int slot = G::pos16(4 * wt + 2 * a + (g >> 1), x);
        const f32x4 bq = bias.q[0][a];
#ifdef BZ_EXP_EPILOGUE_HALF  // TIMING ONLY (results are wrong): what hiding half of the epilogue behind MFMAs could gain at most
        constexpr int kB0 = 1;
#else
        constexpr int kB0 = 0;
#endif
#pragma unroll
        for (int b = kB0; b < 2; ++b)
#pragma unroll
            for (int par = 0; par < 2; ++par)
#pragma unroll
                for (int u = par; u < G::NU; u += 2) {
                    const int off = home2[par] + slot + b * 2 * G::TILE + (u & ~1) * G::ROWC * G::CELL;
                    const int q = 2 * a + b;
                    f32x4 v = {acc[0][u][4 * q], acc[0][u][4 * q + 1], acc[0][u][4 * q + 2], acc[0][u][4 * q + 3]};
                    v = v + bq;
                    if (second) {
                        bf16x4 sk = *reinterpret_cast<const bf16x4*>(out + off);
                        v = v + __builtin_convertvector(sk, f32x4);
                    }
                    f32x2 vlo = {v[0], v[1]}, vhi = {v[2], v[3]};
                    s16x2 lo = __builtin_bit_cast(s16x2, __builtin_convertvector(vlo, bf16x2));
                    s16x2 hi = __builtin_bit_cast(s16x2, __builtin_convertvector(vhi, bf16x2));
                    lo = __builtin_elementwise_max(lo, (s16x2)(0));
                    hi = __builtin_elementwise_max(hi, (s16x2)(0));
                    *reinterpret_cast<uint2*>(out + off) = make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
                }
    }
}

// One conv3x3 layer over the resident positions: LDS -> MFMA -> (+bias, +skip, ReLU) -> LDS.
// S0 = register set that holds chunk 0's weight fragments on entry; on exit it is set (S0 + NCH) % DEPTH.
// What happens to a layer's accumulators is a policy: EpInfer = the inference epilogue above; the training kernels
// (bz_train.hip) bring their own (ReLU masks out / masks in).  ep(acc, out, second, bias, wt0, r, h).
template <class G> struct EpInfer {
    __device__ __forceinline__ void operator()(f32x16 (&acc)[G::MW][G::NU], char* out, bool second, const Bias<G>& bias, int wt0,
                                               int r, int h) const {
        if constexpr (G::M16) epilogue16<G>(acc, out, second, bias, wt0, 32 * h + r);
        else epilogue<G>(acc, out, second, bias, wt0, r, h);
    }
};
// the weight-fragment register sets of a geometry
template <class G> struct WSetsOf { typedef WSets<G> type; };
template <int C_, int P_> struct WSetsOf<Tw<C_, P_, true>> { typedef WSets16<Tw<C_, P_, true>> type; };
template <int S0, class G, class EP = EpInfer<G>>
__device__ __forceinline__ void conv_layer(const char* in, char* out, bool second, const float* __restrict__ bl,
                                           typename WSetsOf<G>::type& WS, const uint4*& ap, int w, int r, int h,
                                           unsigned long long (&tacc)[4], const EP& ep = EP()) {
    [[maybe_unused]] unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    BZ_STAMP(t0);
    f32x16 acc[G::MW][G::NU];
#pragma unroll
    for (int mt = 0; mt < G::MW; ++mt)
#pragma unroll
        for (int u = 0; u < G::NU; ++u) acc[mt][u] = (f32x16)(0.0f);
    const int wpos = G::pos0(w) * G::TILE;
    in += wpos; out += wpos;
    Bias<G> bias;
    if constexpr (G::M16) load_bias16<G>(bias, bl, G::wt0(w), (32 * h + r) >> 4);
    else load_bias<G>(bias, bl, G::wt0(w), h);
    int boff[2];
    if constexpr (G::M16) {
        static_assert(G::ROWT && G::MW == 1 && G::C == 128, "the 16x16x32 path serves the row-tile shape of the 128-channel net");
        bf16x8 B[2][G::NU];
        const int lane = 32 * h + r;
        int boff16[G::KQ];
        tap_off16<G, 0>(lane & 15, lane >> 4, boff16);
        load_b16<G, 0>(B[0], in, boff16[0], 0);
        __builtin_amdgcn_sched_barrier(0);
        run_taps16<S0, 0, G>(acc, WS, ap, in, boff16, lane, B);
    } else if constexpr (G::ROWT) {
        bf16x8 B[G::NBUF][G::NU];
        tap_off<G, 0>(r, h, boff);
#pragma unroll
        for (int k = 0; k + 1 < G::NBUF; ++k) load_b<G, 0>(B[k], in, boff, k);
        // the prologue reads stay OUT of tap 0's scheduling region: inside it the (1 MFMA, 1 DS read) pattern pairs them
        // with the MFMAs that consume them -- every MFMA of the tap then waits for the read issued right before it
        __builtin_amdgcn_sched_barrier(0);
        run_chunks<S0, 0, G>(acc, WS, ap, in, boff, r, h, B);
    } else {
        bf16x8 b0[G::NU], b1[G::NU];
        tap_off_pm<G>(0, r, h, boff);
        load_b<G, 4>(b0, in, boff, 0);
        // chunk c covers tap c / CPT, k-steps (c % CPT) * KS ..; the chunk after the last one is a harmless re-read
        auto tap_of = [](int c) { c = c < G::NCH ? c : G::NCH - 1; return c / G::CPT; };
        auto kc0_of = [](int c) { c = c < G::NCH ? c : G::NCH - 1; return (c % G::CPT) * G::KS; };
        constexpr int D = G::DEPTH, R = G::NCH % D, C0 = G::NCH - R;
#define BZ_CHUNK(J, CC) chunk_step_pm<(S0 + (J)) % D, G>(acc, WS, ap, in, boff, kc0_of(CC), tap_of((CC) + 1), kc0_of((CC) + 1), r, h, b0, b1)
#pragma unroll 1
        for (int c = 0; c + D <= G::NCH; c += D) {  // the register-set index must be a compile-time constant: unroll by DEPTH
            BZ_CHUNK(0, c);
            BZ_CHUNK(1, c + 1);
            if constexpr (D >= 3) BZ_CHUNK(2, c + 2);
        }
        if constexpr (R >= 1) BZ_CHUNK(C0, C0);
        if constexpr (R >= 2) BZ_CHUNK(C0 + 1, C0 + 1);
#undef BZ_CHUNK
    }
    BZ_STAMP(t1);
    ep(acc, out, second, bias, G::wt0(w), r, h);
    BZ_STAMP(t2);
#ifndef BZ_EXP_NO_LAYER_BARRIER  // TIMING ONLY (results are wrong without it): the ceiling of any scheme that relaxes the
    __syncthreads();             // per-layer barrier (per-row ready counters, ...) -- tools/exp_ab_barrier.sh, DESIGN.md 5
#endif
    BZ_STAMP(t3);
    tacc[0] += t1 - t0; tacc[1] += t2 - t1; tacc[2] += t3 - t2;
}

}  // namespace bz_tower
