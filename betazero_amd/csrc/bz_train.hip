// bz_train.hip -- the residual tower's TRAINING step on gfx950: forward with saved activations, backward-data and
// backward-weights of its 2 NB conv3x3 layers as hand-written bf16 MFMA kernels (the loop they
// serve has the shape of src/tic_tac_toe/SL/train.py:85-136 -- forward, loss, backward, Adam step -- with this net in
// place of the reference's MLP).  Stem, heads, losses, the gradient reduction and the optimiser are the kernels of
// bz_train_ends.hip; fp32 master weights.
//
//  * k_train_fwd<G>:  the inference tower's K-loop and LDS image (bz_tower.h: P positions resident in LDS across all
//    layers, weights streamed fragment-major from L2), fed from HBM instead of the stem, with every layer's output copied
//    out (the weight gradients need them) and the ReLU pattern kept as 4 bits per (cell, 4 channels) in exactly the
//    epilogue's lane order -- 16 bytes per lane and layer.
//  * k_train_bwd<G>:  the SAME dataflow run backwards.  d(loss)/d(input) of a conv3x3 is a conv3x3 of the output
//    gradient with the taps mirrored and (co, ci) swapped, so a residual block's backward is again "conv X -> M, conv
//    M -> X in place + what it overwrites" (the skip's gradient); the epilogue multiplies by the saved ReLU bits instead
//    of adding a bias and clamping, and never reads an activation.  Each layer's gradient is copied out for the
//    weight gradients.
//  * k_train_wgrad<C>: dW[tap][ci][co] = sum over (position, cell) of act[cell + tap][ci] * g[cell][co]: a GEMM whose
//    K axis is the cell index, while both operands are stored [cell][channel] -- both MFMA operands come through
//    ds_read_b64_tr_b16 (the LDS transpose read of CDNA4; lane map verified on hardware by tools/probe/probe_tr_read.hip).
//    A workgroup owns one row of taps (dy) of one layer and a slice of the batch; partial sums go to a scratch array
//    that the caller reduces (a few MB).
#include <type_traits>

#include "bz_common.h"
#include "bz_tower.h"

using namespace bz;
using namespace bz_tower;

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;  // (native vector: arrays of HIP's uint4 struct end up in scratch)

namespace {

struct TrainArgs {
    int n, n_layers;            // positions (a multiple of G::P), conv layers (2 NB)
    const uint4* wf;            // weight fragments in the order the kernel consumes them (+ 2 taps of padding)
    const float* bias;          // forward: [L][C]; backward: C zeros
    const __bf16* in;           // forward: act[0]; backward: g[L]                       [n][64][C]
    __bf16* out;                // forward: act[1..L] (slot l - 1); backward: g[0..L-1]  [L][n][64][C]
    uint4* masks;               // [L][n / P][256]: ReLU bits of act[l + 1] in slot l (forward writes, backward reads)
};

// ---- HBM <-> LDS image of the workgroup's P positions ([pos][64 cells][C] bf16 <-> the swizzled halo layout)
template <class G>
__device__ __forceinline__ void load_tile(char* buf, const __bf16* src, int pos0, int tid) {
    constexpr int ZC = G::CELL / 16, N = G::P * 64 * ZC;
    const uint4* s = reinterpret_cast<const uint4*>(src) + (size_t)pos0 * 64 * ZC;
#pragma unroll 4
    for (int i = tid; i < N; i += 256) {
        const int k = i % ZC, c = (i / ZC) % 64, p = i / (ZC * 64);
        *reinterpret_cast<uint4*>(buf + p * G::TILE + G::cell_off(p, c, k)) = s[i];
    }
}
#if defined(BZ_EXP_COPY_AFTER_BARRIER) && !defined(BZ_EXPERIMENT)
#error "BZ_EXP_COPY_AFTER_BARRIER is a diagnostic variant (the round-4 placement of the copy-out, for A/B): build it through betazero_amd.build.build_variant()"
#endif
#if defined(BZ_EXP_NO_TRAIN_STORES) && !defined(BZ_EXPERIMENT)
#error "BZ_EXP_NO_TRAIN_STORES is a diagnostic variant (timing only: the kernels then store nothing): build it through betazero_amd.build.build_variant()"
#endif
template <class G>
__device__ __forceinline__ void store_tile(const char* buf, __bf16* dst, int pos0, int tid) {
#ifdef BZ_EXP_NO_TRAIN_STORES
    return;
#endif
    constexpr int ZC = G::CELL / 16, N = G::P * 64 * ZC;
    uint4* d = reinterpret_cast<uint4*>(dst) + (size_t)pos0 * 64 * ZC;
#pragma unroll 4
    for (int i = tid; i < N; i += 256) {
        const int k = i % ZC, c = (i / ZC) % 64, p = i / (ZC * 64);
        d[i] = *reinterpret_cast<const uint4*>(buf + p * G::TILE + G::cell_off(p, c, k));
    }
}

// ---- the training epilogues.  Lane (r, h) register 4q + i of unit u holds channel 32 wt + 8q + 4h + i of board
// cell (row u, column r & 7) of position r >> 3 (row-tile units).  Its ReLU bits for (q, u) are nibble 8q + u of a
// 128-bit word per lane: word q, bits 4u .. 4u + 3.
//   FWD: + bias (+ skip) -> ReLU -> bf16 -> LDS, and the bits (value > 0) into `bits`
//   BWD: (+ skip) -> x bits -> bf16 -> LDS   (no bias, no clamp; use_bits == false: the gradient leaves unmasked)
template <class G, bool BWD>
__device__ __forceinline__ void epilogue_train(f32x16 (&acc)[G::MW][G::NU], char* out, bool second, const Bias<G>& bias,
                                               int wt0, int r, int h, unsigned (&bits)[4], bool use_bits) {
    static_assert(G::MW == 1 && (G::NU == 8 || G::NU == 4), "the training kernels use one M-tile per wave and 4 or 2 positions per wave");
    const int swz = G::sw(r >> 3, r & 7);
    int home2[2] = {G::lane_home(r) + 8 * h, G::lane_home(r) + 8 * h + G::unit_imm(1)};
    asm volatile("" : "+v"(home2[0]), "+v"(home2[1]));
    const f32x4 (&bq)[4] = bias.q[0];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned word = BWD ? bits[q] : 0u;
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int u = par; u < G::NU; u += 2) {
                const int off = G::unit_imm(u & ~1) + home2[par] + (((4 * wt0 + q) ^ swz) << 4);
                f32x4 v = {acc[0][u][4 * q], acc[0][u][4 * q + 1], acc[0][u][4 * q + 2], acc[0][u][4 * q + 3]};
                if (!BWD) v = v + bq[q];
                if (second) {
                    bf16x4 sk = *reinterpret_cast<const bf16x4*>(out + off);
                    v = v + __builtin_convertvector(sk, f32x4);
                }
                if (BWD) {
                    const unsigned nib = use_bits ? (word >> (4 * u)) & 0xFu : 0xFu;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = (nib >> i) & 1u ? v[i] : 0.0f;
                }
                f32x2 vlo = {v[0], v[1]}, vhi = {v[2], v[3]};
                s16x2 lo = __builtin_bit_cast(s16x2, __builtin_convertvector(vlo, bf16x2));
                s16x2 hi = __builtin_bit_cast(s16x2, __builtin_convertvector(vhi, bf16x2));
                if (!BWD) {
                    lo = __builtin_elementwise_max(lo, (s16x2)(0));
                    hi = __builtin_elementwise_max(hi, (s16x2)(0));
                    const unsigned nib = (lo[0] != 0 ? 1u : 0u) | (lo[1] != 0 ? 2u : 0u) | (hi[0] != 0 ? 4u : 0u) | (hi[1] != 0 ? 8u : 0u);
                    word |= nib << (4 * u);
                }
                *reinterpret_cast<uint2*>(out + off) = make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
            }
        if (!BWD) bits[q] = word;
    }
}
// The same for the 16x16x32 path (Tw<128, 4, true>; lane map of bz_tower.h's epilogue16): lane (c, g) register 4 (2a + b) + i
// of unit u holds channel 32 wt + 16 a + 4 g + i of board cell (row u, column c & 7) of position 2 b + (c >> 3).  ReLU bits:
// word 2a + b, bits 4u .. 4u + 3 -- again 128 bits per lane and layer.
template <class G, bool BWD>
__device__ __forceinline__ void epilogue_train16(f32x16 (&acc)[G::MW][G::NU], char* out, bool second, const Bias<G>& bias, int wt,
                                                 int lane, unsigned (&bits)[4], bool use_bits) {
    static_assert(G::MW == 1 && G::NU == 8 && G::M16, "the 16x16x32 path serves the row-tile shape of the 128-channel net");
    const int c = lane & 15, g = lane >> 4, x = c & 7, pl = c >> 3;
    int home2[2] = {pl * G::TILE + G::cell_at(0, x) + 8 * (g & 1), pl * G::TILE + G::cell_at(1, x) + 8 * (g & 1)};
    asm volatile("" : "+v"(home2[0]), "+v"(home2[1]));
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int slot = G::pos16(4 * wt + 2 * a + (g >> 1), x);
        const f32x4 bq = bias.q[0][a];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int q = 2 * a + b;
            unsigned word = BWD ? bits[q] : 0u;
#pragma unroll
            for (int par = 0; par < 2; ++par)
#pragma unroll
                for (int u = par; u < G::NU; u += 2) {
                    const int off = home2[par] + slot + b * 2 * G::TILE + (u & ~1) * G::ROWC * G::CELL;
                    f32x4 v = {acc[0][u][4 * q], acc[0][u][4 * q + 1], acc[0][u][4 * q + 2], acc[0][u][4 * q + 3]};
                    if (!BWD) v = v + bq;
                    if (second) {
                        bf16x4 sk = *reinterpret_cast<const bf16x4*>(out + off);
                        v = v + __builtin_convertvector(sk, f32x4);
                    }
                    if (BWD) {
                        const unsigned nib = use_bits ? (word >> (4 * u)) & 0xFu : 0xFu;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = (nib >> i) & 1u ? v[i] : 0.0f;
                    }
                    f32x2 vlo = {v[0], v[1]}, vhi = {v[2], v[3]};
                    s16x2 lo = __builtin_bit_cast(s16x2, __builtin_convertvector(vlo, bf16x2));
                    s16x2 hi = __builtin_bit_cast(s16x2, __builtin_convertvector(vhi, bf16x2));
                    if (!BWD) {
                        lo = __builtin_elementwise_max(lo, (s16x2)(0));
                        hi = __builtin_elementwise_max(hi, (s16x2)(0));
                        const unsigned nib = (lo[0] != 0 ? 1u : 0u) | (lo[1] != 0 ? 2u : 0u) | (hi[0] != 0 ? 4u : 0u) | (hi[1] != 0 ? 8u : 0u);
                        word |= nib << (4 * u);
                    }
                    *reinterpret_cast<uint2*>(out + off) = make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
                }
            if (!BWD) bits[q] = word;
        }
    }
}

// The copy-out of a layer's result (LDS tile -> HBM, what the weight gradients read) rides in the NEXT layer's epilogue on the
// row-tile shapes: the tile is that layer's input, complete since the barrier and never written by its epilogue.  Issued
// right behind a layer's own barrier the 16 stores per lane sit in front of the next K-loop's weight loads (vmcnt retires in
// order); in the epilogue they have ~1.5 us without a vector-memory wait to land.  Same-box A/B (tools/exp_train_stores.sh,
// profiles/r04_exp_train_stores.txt): -1.5 % (128 channels) .. -3.4 / -6.5 % (64 channels x 8 positions, forward / backward);
// the half-tile shape measured +3 % on backward and keeps the copy behind the barrier.  The copy-out itself costs 10-20 % of
// these kernels wherever it is issued (the same A/B against a build that stores nothing): 64 KB per workgroup and layer
// through the CU's 64-B/clk path to L2, in step across all workgroups.
template <class G> constexpr bool kCopyInEpilogue =
#ifdef BZ_EXP_COPY_AFTER_BARRIER
    false;
#else
    G::ROWT;
#endif
// (Rejected, code removed: a layer's result stored to HBM from its own epilogue, straight from the registers that go to LDS --
// no copy pass at all, but 8 bytes per lane and store, 32-byte runs per cell: 10 % slower, profiles/r04_exp_epilogue_stores.txt;
// and the copy pass dealt to the next K-loop's sub-steps: 5 % slower, profiles/r04_exp_kloop_copy.txt.)
template <class G, bool BWD> struct EpTrain {
    unsigned* bits;   // the lane's 128 ReLU bits of this layer (FWD: written, BWD: read)
    bool use_bits;
    const char* copy_buf;   // the workgroup's LDS tile to copy out first (this layer's input) ...
    __bf16* copy_dst;       // ... to here (the workgroup's first position of the destination tensor), or null: nothing to copy
    int tid;
    __device__ __forceinline__ void operator()(f32x16 (&acc)[G::MW][G::NU], char* out, bool second, const Bias<G>& bias, int wt0,
                                               int r, int h) const {
        if (kCopyInEpilogue<G> && copy_dst) store_tile<G>(copy_buf, copy_dst, 0, tid);
        unsigned (&b)[4] = *reinterpret_cast<unsigned (*)[4]>(bits);
        if constexpr (G::M16) epilogue_train16<G, BWD>(acc, out, second, bias, wt0, 32 * h + r, b, use_bits);
        else epilogue_train<G, BWD>(acc, out, second, bias, wt0, r, h, b, use_bits);
    }
};

template <class G>
__device__ __forceinline__ void zero_halo(char* smem, int tid) {
    constexpr int ZC = G::CELL / 16;
    for (int i = tid; i < 2 * G::P * 9 * ZC; i += 256) {
        int k = i % ZC, j = (i / ZC) % 9, pb = i / (9 * ZC);
        *reinterpret_cast<uint4*>(smem + pb * G::TILE + j * G::ROWC * G::CELL + k * 16) = make_uint4(0, 0, 0, 0);
    }
}
template <class G>
__device__ __forceinline__ void weights_prologue(WSets16<G>& WS, const uint4*& ap, int lane) {  // tap 0 of the first layer: KQ steps x 2 channel halves
#pragma unroll
    for (int kq = 0; kq < G::KQ; ++kq)
#pragma unroll
        for (int a = 0; a < 2; ++a) WS.s[0][kq][a] = __builtin_bit_cast(bf16x8, ap[(kq * G::MT * 2 + a) * 64 + (unsigned)lane]);
    ap += G::KQ * G::MT * 2 * 64;
}
template <class G>
__device__ __forceinline__ void weights_prologue(WSets<G>& WS, const uint4*& ap, int lane) {
#pragma unroll
    for (int d = 0; d + 1 < G::DEPTH; ++d) {
#pragma unroll
        for (int kc = 0; kc < G::KS; ++kc)
#pragma unroll
            for (int mt = 0; mt < G::MW; ++mt) WS.s[d][kc][mt] = __builtin_bit_cast(bf16x8, ap[(kc * G::MT + mt) * 64 + (unsigned)lane]);
        ap += G::KS * G::MT * 64;
    }
}

// forward with saved activations: act[0] (HBM) -> act[1 .. L] (HBM) + ReLU bits
template <class G>
__global__ void __launch_bounds__(256, 1) k_train_fwd(TrainArgs T) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pos0 = blockIdx.x * G::P, r = lane & 31, h = lane >> 5;
    if (pos0 >= T.n) return;
    [[maybe_unused]] unsigned long long tacc[4] = {0, 0, 0, 0};
    char* bufX = smem;
    char* bufM = smem + G::BUF;
    zero_halo<G>(smem, tid);
    const uint4* ap = T.wf + (size_t)G::wt0(w) * (G::M16 ? 2 * 64 : 64);
    typename WSetsOf<G>::type WS;
    weights_prologue<G>(WS, ap, lane);
    load_tile<G>(bufX, T.in, pos0, tid);
    __syncthreads();
    const size_t slot = (size_t)T.n * 64 * G::C;                       // elements of one activation tensor
    __bf16* const mine = T.out + (size_t)pos0 * 64 * G::C;             // the workgroup's positions in the first output tensor
    const size_t mslot = (size_t)(T.n / G::P) * 256;                    // mask words of one layer
    uint4* mk = T.masks + (size_t)blockIdx.x * 256 + tid;
#pragma unroll 1
    for (int blk = 0; blk < T.n_layers / 2; ++blk) {
        unsigned bits[4];
        // (act[2 blk] = X, the previous block's result, leaves in this layer's epilogue; act[0] came from HBM)
        conv_layer<0, G>(bufX, bufM, false, T.bias + (size_t)(2 * blk) * G::C, WS, ap, w, r, h, tacc,
                         EpTrain<G, false>{bits, true, bufX, blk > 0 ? mine + (size_t)(2 * blk - 1) * slot : nullptr, tid});
        if constexpr (!kCopyInEpilogue<G>)
            if (__bf16* cd = blk > 0 ? mine + (size_t)(2 * blk - 1) * slot : nullptr) store_tile<G>(bufX, cd, 0, tid);
        mk[(size_t)(2 * blk) * mslot] = make_uint4(bits[0], bits[1], bits[2], bits[3]);
        conv_layer<G::NCH % G::DEPTH, G>(bufM, bufX, true, T.bias + (size_t)(2 * blk + 1) * G::C, WS, ap, w, r, h, tacc,
                                         EpTrain<G, false>{bits, true, bufM, mine + (size_t)(2 * blk) * slot, tid});
        if constexpr (!kCopyInEpilogue<G>)
            if (__bf16* cd = mine + (size_t)(2 * blk) * slot) store_tile<G>(bufM, cd, 0, tid);
        mk[(size_t)(2 * blk + 1) * mslot] = make_uint4(bits[0], bits[1], bits[2], bits[3]);
    }
    store_tile<G>(bufX, mine + (size_t)(T.n_layers - 1) * slot, 0, tid);
}

// backward-data: g[L] (HBM) -> g[L-1 .. 0] (HBM).  g[l] = d(loss)/d(pre-activation of act[l]) for l >= 1, g[0] =
// d(loss)/d(act[0]) (the stem's ReLU belongs to the caller).  Weight stream: layers L-1, L-2, .., 0, each with mirrored
// taps and (co, ci) swapped (k_pack_weights).
template <class G>
__global__ void __launch_bounds__(256, 1) k_train_bwd(TrainArgs T) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pos0 = blockIdx.x * G::P, r = lane & 31, h = lane >> 5;
    if (pos0 >= T.n) return;
    [[maybe_unused]] unsigned long long tacc[4] = {0, 0, 0, 0};
    char* bufX = smem;
    char* bufM = smem + G::BUF;
    zero_halo<G>(smem, tid);
    const uint4* ap = T.wf + (size_t)G::wt0(w) * (G::M16 ? 2 * 64 : 64);
    typename WSetsOf<G>::type WS;
    weights_prologue<G>(WS, ap, lane);
    load_tile<G>(bufX, T.in, pos0, tid);
    __syncthreads();
    const size_t slot = (size_t)T.n * 64 * G::C;
    __bf16* const mine = T.out + (size_t)pos0 * 64 * G::C;
    const size_t mslot = (size_t)(T.n / G::P) * 256;
    const uint4* mk = T.masks + (size_t)blockIdx.x * 256 + tid;
#pragma unroll 1
    for (int blk = T.n_layers / 2 - 1; blk >= 0; --blk) {
        // conv2 transposed: g[2 blk + 2] (X) -> g[2 blk + 1] (M), x ReLU bits of act[2 blk + 1] (mask slot 2 blk)
        uint4 m1 = mk[(size_t)(2 * blk) * mslot];
        unsigned b1[4] = {m1.x, m1.y, m1.z, m1.w};
        // (g[2 blk + 2] = X, the block above's result, leaves in this layer's epilogue; g[L] came from HBM)
        conv_layer<0, G>(bufX, bufM, false, T.bias, WS, ap, w, r, h, tacc,
                         EpTrain<G, true>{b1, true, bufX, blk < T.n_layers / 2 - 1 ? mine + (size_t)(2 * blk + 2) * slot : nullptr, tid});
        if constexpr (!kCopyInEpilogue<G>)
            if (__bf16* cd = blk < T.n_layers / 2 - 1 ? mine + (size_t)(2 * blk + 2) * slot : nullptr) store_tile<G>(bufX, cd, 0, tid);
        // conv1 transposed: g[2 blk + 1] (M) -> X in place, + g[2 blk + 2] (the skip's gradient = what it overwrites),
        // x ReLU bits of act[2 blk] (mask slot 2 blk - 1; the tower's input act[0] has none: its ReLU is the stem's)
        uint4 m0 = blk > 0 ? mk[(size_t)(2 * blk - 1) * mslot] : make_uint4(0, 0, 0, 0);
        unsigned b0[4] = {m0.x, m0.y, m0.z, m0.w};
        conv_layer<G::NCH % G::DEPTH, G>(bufM, bufX, true, T.bias, WS, ap, w, r, h, tacc,
                                         EpTrain<G, true>{b0, blk > 0, bufM, mine + (size_t)(2 * blk + 1) * slot, tid});
        if constexpr (!kCopyInEpilogue<G>)
            if (__bf16* cd = mine + (size_t)(2 * blk + 1) * slot) store_tile<G>(bufM, cd, 0, tid);
    }
    store_tile<G>(bufX, mine, 0, tid);   // g[0]
}

// ---- weights: torch layout fp32 W[l][co][ci][tap] -> the two fragment streams (bf16, round to nearest even)
//   forward : frag[((l * 9 + t) * KC + kc) * MT + mt][lane 32h + r][j] = W[l][co = 32mt + r][ci = 16kc + 8h + j][t]
//   backward: frag[(((L-1-l) * 9 + t) * KC + kc) * MT + mt][lane][j]   = W[l][co = 16kc + 8h + j][ci = 32mt + r][8 - t]
__global__ void __launch_bounds__(256) k_pack_weights(const float* __restrict__ W, uint4* __restrict__ wf_fwd,
                                                      uint4* __restrict__ wf_bwd, int L, int C) {
    const int KC = C / 16, MT = C / 32;
    const long long total = (long long)L * 9 * KC * MT * 64;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63), r = lane & 31, h = lane >> 5;
    long long rest = idx >> 6;
    const int mt = (int)(rest % MT); rest /= MT;
    const int kc = (int)(rest % KC); rest /= KC;
    const int t = (int)(rest % 9);
    const int l = (int)(rest / 9);
    const float* Wl = W + (size_t)l * C * C * 9;
    bf16x8 f, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * kc + 8 * h + j, m = 32 * mt + r;
        f[j] = (__bf16)Wl[((size_t)m * C + k) * 9 + t];
        b[j] = (__bf16)Wl[((size_t)k * C + m) * 9 + (8 - t)];
    }
    wf_fwd[idx] = __builtin_bit_cast(uint4, f);
    wf_bwd[((((long long)(L - 1 - l) * 9 + t) * KC + kc) * MT + mt) * 64 + lane] = __builtin_bit_cast(uint4, b);
}

// the same two streams in the 16x16x32 path's fragment order (bz_tower.h; 128 channels): lane = 16 g + c
//   forward : frag[((((l * 9 + t) * KQ + kq) * MT + wt) * 2 + a)][lane][j] = W[l][co = 32wt + 16a + c][ci = 32kq + 8g + j][t]
//   backward: frag[(((((L-1-l) * 9 + t) * KQ + kq) * MT + wt) * 2 + a)][lane][j] = W[l][co = 32kq + 8g + j][ci = 32wt + 16a + c][8 - t]
__global__ void __launch_bounds__(256) k_pack_weights16(const float* __restrict__ W, uint4* __restrict__ wf_fwd,
                                                        uint4* __restrict__ wf_bwd, int L, int C) {
    const int KQ = C / 32, MT = C / 32;
    const long long total = (long long)L * 9 * KQ * MT * 2 * 64;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63), c = lane & 15, g = lane >> 4;
    long long rest = idx >> 6;
    const int a = (int)(rest & 1); rest >>= 1;
    const int wt = (int)(rest % MT); rest /= MT;
    const int kq = (int)(rest % KQ); rest /= KQ;
    const int t = (int)(rest % 9);
    const int l = (int)(rest / 9);
    const float* Wl = W + (size_t)l * C * C * 9;
    bf16x8 f, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 32 * kq + 8 * g + j, m = 32 * wt + 16 * a + c;
        f[j] = (__bf16)Wl[((size_t)m * C + k) * 9 + t];
        b[j] = (__bf16)Wl[((size_t)k * C + m) * 9 + (8 - t)];
    }
    wf_fwd[idx] = __builtin_bit_cast(uint4, f);
    wf_bwd[((((((long long)(L - 1 - l) * 9 + t) * KQ + kq) * MT + wt) * 2 + a) * 64) + lane] = __builtin_bit_cast(uint4, b);
}

// ---- backward-weights
// LDS images for the transpose reads, both [cell][channel] with the 64-byte pieces of a cell XOR-swizzled by the cell
// index so that the four cells of a ds_read_b64_tr_b16 block (4 consecutive cells x 64 B per 32-lane half) fall on
// four different 64-byte bank windows: 256-B cells: piece ^= idx & 3; 128-B cells: piece ^= (idx >> 1) & 1.
//   activations: 10 rows x 9 cells + 1 (rows -1 .. 8, columns -1 .. 8; idx = 9 (y + 1) + x + 1), all C channels: the zero
//                ring makes every tap shift a plain address offset
//   gradients  : 64 cells (idx = 8 y + x) x the workgroup's 64 output channels
// A workgroup owns ALL NINE taps of one layer for 64 output channels (all of them at C = 64, one half at C = 128) and a
// slice of the batch: every activation tile is read once (twice at C = 128), every gradient tile once.  (The first
// version gave a workgroup one tap row and all channels: three workgroups re-read every tile -- 393 MB per step at C =
// 64, which bound the kernel: 76 us where the matrix work is 15.)
template <int C_> struct Wg {
    static constexpr int C = C_, MT = C / 32, CELL = 2 * C, ZC = CELL / 16;
    static constexpr int NH = C / 64;                       // workgroups that share a (layer, slice): one per 64 output channels
    static constexpr int NTW = 2;                           // N tiles (32 output channels) per wave: both of the workgroup's 64 channels
    static constexpr int GCELL = 128, GZC = 8;              // the workgroup's 64 output channels of a gradient cell
    static constexpr int P2 = 256 / C;                      // positions per stage
    // waves per workgroup.  The workgroup's 9 x MT x 2 accumulator tiles are dealt as (M tile) x (tap group), every wave
    // taking both N tiles (an activation fragment then feeds two MFMAs: 14 transpose reads per 10 MFMAs):
    //   C =  64: 4 waves = 2 M tiles x (taps 0-4 | taps 5-8)   -> 10 / 8 tiles per wave
    //   C = 128: 8 waves = 4 M tiles x (taps 0-4 | taps 5-8)   -> 10 / 8 tiles per wave, two waves per SIMD
    // (Round 4's first split -- C = 128: 4 waves x all 9 taps x 2 N tiles -- meant 288 accumulator registers per wave,
    // more than the 256 AGPRs: the compiler rotated three tiles through the VGPR file with 96 v_accvgpr moves per 18 MFMAs,
    // and the VALU issue slots those take are the MFMAs' own: 338 us where the matrix work is 110.  C = 64: 4 waves x 9 taps x
    // ONE N tile read 20 fragments per 9 MFMAs: LDS-bound.)
    static constexpr int NW = 2 * MT, NT = 64 * NW;
    static constexpr int NTG = 2 / NTW;                     // N tile groups (1)
    static constexpr int TGN = NW / (MT * NTG);             // tap groups (2)
    static constexpr int ACELLS = 91, A_TILE = ACELLS * CELL, G_TILE = 64 * GCELL;
    static constexpr int STAGE = P2 * (A_TILE + G_TILE);
    static constexpr int NLA = P2 * 64 * ZC / NT, NLG = P2 * 64 * GZC / NT;  // 16-byte loads per thread and stage
    static_assert(NLA * NT == P2 * 64 * ZC && NLG * NT == P2 * 64 * GZC && TGN * MT * NTG == NW, "the stage must split evenly over the threads");
    static constexpr int LDS = 2 * STAGE;
    static_assert(LDS <= 160 * 1024, "two stages must fit the CU's LDS");
    static __device__ __forceinline__ int a_swz(int idx) { return C == 128 ? (idx & 3) : ((idx >> 1) & 1); }
    static __device__ __forceinline__ int a_off(int p, int idx, int piece) { return p * A_TILE + idx * CELL + ((piece ^ a_swz(idx)) << 6); }
    static __device__ __forceinline__ int g_off(int p, int idx, int piece) {
        return P2 * A_TILE + p * G_TILE + idx * GCELL + ((piece ^ ((idx >> 1) & 1)) << 6);
    }
};

struct WgradArgs {
    const __bf16* acts;   // act[0 .. L-1]   [L][n][64][C]
    const __bf16* gs;     // g[1 .. L]       [L][n][64][C]
    float* partial;       // [L][S][9 taps][ci][co]
    float* db_partial;    // [L][2 S MT][co]: sums of g over the slice's (position, cell) -- one row per k half and per wave of the
                          // 4-tap group (each adds 8 / MT of a fragment's 8 cells): the bias gradient's partial sums
    int n, L, S;
};

__device__ __forceinline__ bf16x8 tr_pair(const char* lds, int a0, int a1) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + a1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// a stage's activations and gradients: HBM -> registers (issued a stage ahead) -> the LDS images
template <class G>
__device__ __forceinline__ void wg_fetch(u32x4 (&ra)[G::NLA], u32x4 (&rg)[G::NLG], const u32x4* A, const u32x4* Gr, int s, int half, int tid) {
    const size_t base = (size_t)s * G::P2 * 64 * G::ZC;
#pragma unroll
    for (int j = 0; j < G::NLA; ++j) ra[j] = A[base + tid + G::NT * j];
#pragma unroll
    for (int j = 0; j < G::NLG; ++j) {  // chunk k (0..7) of the workgroup's half of cell (p, c)
        const int i = tid + G::NT * j, k = i % G::GZC, pc = i / G::GZC;
        rg[j] = Gr[base + (size_t)pc * G::ZC + half * G::GZC + k];
    }
}
template <class G>
__device__ __forceinline__ void wg_stash(const u32x4 (&ra)[G::NLA], const u32x4 (&rg)[G::NLG], char* st, int tid) {
#pragma unroll
    for (int j = 0; j < G::NLA; ++j) {
        const int i = tid + G::NT * j, k = i % G::ZC, c = (i / G::ZC) % 64, p = i / (G::ZC * 64);
        const int ai = 9 * ((c >> 3) + 1) + (c & 7) + 1;
        *reinterpret_cast<u32x4*>(st + G::a_off(p, ai, k >> 2) + ((k & 3) << 4)) = ra[j];
    }
#pragma unroll
    for (int j = 0; j < G::NLG; ++j) {
        const int i = tid + G::NT * j, k = i % G::GZC, c = (i / G::GZC) % 64, p = i / (G::GZC * 64);
        *reinterpret_cast<u32x4*>(st + G::g_off(p, c, k >> 2) + ((k & 3) << 4)) = rg[j];
    }
}

// diagnostic A/B: how far the k-step loop of the weight-gradient kernel is unrolled (tools/exp_wgrad_bounds.sh)
#if (defined(BZ_EXP_WGRAD_NO_LDS_READS) || defined(BZ_EXP_WGRAD_NO_MFMA)) && !defined(BZ_EXPERIMENT)
#error "BZ_EXP_WGRAD_NO_LDS_READS / _NO_MFMA are diagnostic variants (timing only): build them through betazero_amd.build.build_variant()"
#endif
#if defined(BZ_EXP_WGRAD_NO_FETCH) && !defined(BZ_EXPERIMENT)
#error "BZ_EXP_WGRAD_NO_FETCH is a diagnostic variant: build it through betazero_amd.build.build_variant()"
#endif
#ifdef BZ_EXP_WGRAD_UNROLL
#ifndef BZ_EXPERIMENT
#error "BZ_EXP_WGRAD_UNROLL is a diagnostic variant: build it through betazero_amd.build.build_variant()"
#endif
#define BZ_WGRAD_KK_UNROLL BZ_EXP_WGRAD_UNROLL
#else
#define BZ_WGRAD_KK_UNROLL 1
#endif
// the stages of one workgroup for the taps [T0, T1) of one wave (all waves run the same number of stages and barriers)
template <int C, int T0, int T1>
__device__ __forceinline__ void wgrad_wave(const WgradArgs& T, char* smem, int tid, int lane, int mt, int nt0, int l, int split, int half,
                                           int s_begin, int s_end, bool do_bias) {
    typedef Wg<C> G;
    constexpr int NTAP = T1 - T0;
    const size_t slot = (size_t)T.n * 64 * G::ZC;  // uint4 per tensor
    const u32x4* A = reinterpret_cast<const u32x4*>(T.acts) + (size_t)l * slot;
    const u32x4* Gr = reinterpret_cast<const u32x4*>(T.gs) + (size_t)l * slot;
    u32x4 ra[G::NLA], rg[G::NLG];
    f32x16 acc[NTAP][G::NTW];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int nt = 0; nt < G::NTW; ++nt) acc[t][nt] = (f32x16)(0.0f);
    // the bias gradient rides along in the waves of the 4-tap group (two MFMAs per k-step fewer than the others: the stage
    // ends at a barrier, so work on the 5-tap waves is work on the critical path -- the first version gave all of it to ONE
    // of them, 32 VALU instructions and an early LDS wait per k-step).  A lane's gradient fragment is 8 cells of ONE output
    // channel: wave mt adds cells 8 mt / MT .. of every fragment, so the waves' sums add up to the channel's column sum (one
    // float per N tile and wave; the two k halves of a channel are two lanes)
    float bsum[G::NTW];
#pragma unroll
    for (int nt = 0; nt < G::NTW; ++nt) bsum[nt] = 0.0f;
    // lane roles in a transpose read: group g = lane >> 4 (g >> 1 = the MFMA's k half, g & 1 = which 16 of the tile's 32
    // channels), lane 4q + pp of the group supplies cell q of the block, channels 4pp .. 4pp + 3
    const int g = lane >> 4, hh = g >> 1, cg = g & 1, q = (lane >> 2) & 3, pp = lane & 3;
    const int inner = 32 * cg + 8 * pp;
    wg_fetch<G>(ra, rg, A, Gr, s_begin, half, tid);
    __syncthreads();  // the zero fill is complete
    wg_stash<G>(ra, rg, smem, tid);
    __syncthreads();
#pragma unroll 1
    for (int s = s_begin; s < s_end; ++s) {
        const char* st = smem + ((s - s_begin) & 1) * G::STAGE;
#ifndef BZ_EXP_WGRAD_NO_FETCH  // (diagnostic, timing only: every stage computes on the first stage's data -- what the kernel costs without its input stream)
        wg_fetch<G>(ra, rg, A, Gr, s + 1 < s_end ? s + 1 : s, half, tid);  // (the last stage re-reads itself: no branch around the registers)
#endif
#pragma unroll 1
        for (int p = 0; p < G::P2; ++p) {
#pragma unroll BZ_WGRAD_KK_UNROLL
            for (int kk = 0; kk < 4; ++kk) {  // k-step = board rows 2 kk (k half 0) and 2 kk + 1 (k half 1)
                const int y = 2 * kk + hh;
                bf16x8 bf[G::NTW];
#pragma unroll
                for (int nt = 0; nt < G::NTW; ++nt) {
                    const int i0 = 8 * y + q, i1 = i0 + 4;
                    bf[nt] = tr_pair(st, G::g_off(p, i0, nt0 + nt) + inner, G::g_off(p, i1, nt0 + nt) + inner);
                }
#pragma unroll
                for (int t = T0; t < T1; ++t) {  // tap t = 3 (dy + 1) + (dx + 1): the activations one row / one column over
                    const int i0 = 9 * (y + t / 3) + (q + t % 3 - 1) + 1, i1 = i0 + 4;
#ifdef BZ_EXP_WGRAD_NO_LDS_READS   // (timing only: every tap multiplies the gradient fragment with itself -- the loop without its activation reads)
                    const bf16x8 af = bf[t & 1];
#else
                    const bf16x8 af = tr_pair(st, G::a_off(p, i0, mt) + inner, G::a_off(p, i1, mt) + inner);
#endif
#ifdef BZ_EXP_WGRAD_NO_MFMA        // (timing only: the reads alone, folded into one accumulator so that they stay)
                    acc[t - T0][0][0] += (float)af[0] + (float)af[7];
#else
#pragma unroll
                    for (int nt = 0; nt < G::NTW; ++nt) acc[t - T0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[nt], acc[t - T0][nt], 0, 0, 0);
#endif
                }
                if (do_bias) {   // (behind the taps: the adds issue while the last MFMAs run, and nothing waits early for bf)
                    constexpr int JW = 8 / G::MT;
                    auto add = [&](auto first) {   // (mt is wave-uniform: one scalar branch, compile-time element numbers)
#pragma unroll
                        for (int nt = 0; nt < G::NTW; ++nt)
#pragma unroll
                            for (int j = 0; j < JW; ++j) bsum[nt] += (float)bf[nt][decltype(first)::value + j];
                    };
                    if (mt == 0) add(std::integral_constant<int, 0>());
                    else if (mt == 1) add(std::integral_constant<int, JW>());
                    else if constexpr (G::MT == 4) {
                        if (mt == 2) add(std::integral_constant<int, 2 * JW>());
                        else add(std::integral_constant<int, 3 * JW>());
                    }
                }
            }
        }
        wg_stash<G>(ra, rg, smem + ((s + 1 - s_begin) & 1) * G::STAGE, tid);
        __syncthreads();
    }
    // D[row = ci][col = co]: lane (r, h) register i holds ci = 32 mt + (i & 3) + 8 (i >> 2) + 4 h, co = 64 half + 32 nt + r
    const int r = lane & 31, h = lane >> 5;
    float* P = T.partial + (((size_t)l * T.S + split) * 9) * C * C;
#pragma unroll
    for (int t = T0; t < T1; ++t)
#pragma unroll
        for (int nt = 0; nt < G::NTW; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ci = 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * h, co = 64 * half + 32 * (nt0 + nt) + r;
                P[((size_t)t * C + ci) * C + co] = acc[t - T0][nt][i];
            }
    if (do_bias) {  // lane (r, h): channel r of the tile, k half h -> partial sums [L][2 S MT][co]
#pragma unroll
        for (int nt = 0; nt < G::NTW; ++nt)
            T.db_partial[((size_t)l * 2 * T.S * G::MT + (size_t)(2 * split + h) * G::MT + mt) * C + 64 * half + 32 * (nt0 + nt) + r] = bsum[nt];
    }
}

template <int C>
__global__ void __launch_bounds__(Wg<C>::NT, 1) k_train_wgrad(WgradArgs T) {
    typedef Wg<C> G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the NH workgroups of one (layer, batch slice) sit 8 workgroups apart: same XCD (workgroups go round-robin over the 8
    // XCDs), launched together -> the activations they both read meet in that XCD's L2
    const int b = blockIdx.x, item = (b / (8 * G::NH)) * 8 + (b & 7), half = (b >> 3) % G::NH;
    if (item >= T.L * T.S) return;
    const int l = item / T.S, split = item % T.S;
    // this slice's stages (P2 positions each)
    const int stages_all = T.n / G::P2;
    const int s_begin = (int)((long long)stages_all * split / T.S), s_end = (int)((long long)stages_all * (split + 1) / T.S);
    const int mt = w % G::MT, rest = w / G::MT, nt0 = (rest % G::NTG) * G::NTW, tg = rest / G::NTG;
    // zero both stages' activation images once: the loads below only ever write board cells
    for (int i = tid; i < 2 * G::P2 * G::A_TILE / 16; i += G::NT) {
        const int st = i / (G::P2 * G::A_TILE / 16), o = i % (G::P2 * G::A_TILE / 16);
        *reinterpret_cast<uint4*>(smem + st * G::STAGE + o * 16) = make_uint4(0, 0, 0, 0);
    }
    if (s_begin >= s_end) return;  // (block-uniform; cannot happen: splits <= stages)
    const bool do_bias = tg == G::TGN - 1;  // wave-uniform
    if constexpr (G::TGN == 1) {
        wgrad_wave<C, 0, 9>(T, smem, tid, lane, mt, nt0, l, split, half, s_begin, s_end, do_bias);
    } else {
        if (tg == 0) wgrad_wave<C, 0, 5>(T, smem, tid, lane, mt, nt0, l, split, half, s_begin, s_end, do_bias);
        else wgrad_wave<C, 5, 9>(T, smem, tid, lane, mt, nt0, l, split, half, s_begin, s_end, do_bias);
    }
}

bool train_shape_ok(int C, int L, int n) { return (C == 64 || C == 128) && L >= 2 && L % 2 == 0 && L <= 128 && n >= 1; }
}  // namespace

BZ_EXPORT int64_t bz_train_wf_bytes(int32_t C, int32_t n_layers) {
    if (!train_shape_ok(C, n_layers, 1)) { set_error("bz_train_wf_bytes: C must be 64 or 128, n_layers even"); return -1; }
    return ((int64_t)n_layers * 9 + 2) * (C / 16) * (C / 32) * 1024;
}
BZ_EXPORT int32_t bz_train_positions_per_workgroup(int32_t C) { return C == 64 ? Tw<64>::P : (C == 128 ? Tw<128>::P : 0); }
BZ_EXPORT int64_t bz_train_mask_bytes(int32_t C, int32_t n_layers, int32_t n) {
    const int P = bz_train_positions_per_workgroup(C);
    if (!P || n % P) { set_error("bz_train_mask_bytes: n must be a multiple of %d for C = %d", P, C); return -1; }
    return (int64_t)n_layers * (n / P) * 256 * 16 * (C == 64 ? 2 : 1);   // (C = 64: small batches run 4 positions per workgroup, twice the lanes)
}
/* rows of bz_train_wgrad's db_partial per layer */
BZ_EXPORT int32_t bz_train_wgrad_bias_rows(int32_t C, int32_t splits) { return (C == 64 || C == 128) && splits >= 1 ? 2 * splits * (C / 32) : 0; }
/* number of batch slices per layer of the weight-gradient kernel: n_layers * (C / 64) * S workgroups fill the chip once */
BZ_EXPORT int32_t bz_train_wgrad_splits(int32_t C, int32_t n_layers, int32_t n) {
    if (!train_shape_ok(C, n_layers, n)) return 0;
    int s = 256 / (n_layers * (C / 64));
    const int stages = n / (256 / C);
    if (s > stages) s = stages;
    return s < 1 ? 1 : s;
}

BZ_EXPORT int32_t bz_train_pack_weights(const float* W, int32_t C, int32_t n_layers, void* wf_fwd, void* wf_bwd, void* stream) {
    BZ_REQUIRE(W && wf_fwd && wf_bwd && train_shape_ok(C, n_layers, 1), "bz_train_pack_weights: bad arguments (C = 64 or 128, n_layers even)");
    const long long total = (long long)n_layers * 9 * (C / 16) * (C / 32) * 64;   // (the same count in either fragment order)
    if (C == 128)   // the 128-channel tower kernels run on the 16x16x32 path
        hipLaunchKernelGGL(k_pack_weights16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W,
                           static_cast<uint4*>(wf_fwd), static_cast<uint4*>(wf_bwd), n_layers, C);
    else
        hipLaunchKernelGGL(k_pack_weights, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W,
                           static_cast<uint4*>(wf_fwd), static_cast<uint4*>(wf_bwd), n_layers, C);
    BZ_LAUNCH_CHECK("k_pack_weights");
    return BZ_OK;
}

static int32_t train_tower(bool bwd, const void* in, const void* wf, const float* bias, int32_t C, int32_t L, int32_t n, void* out,
                           void* masks, void* stream) {
    BZ_REQUIRE(in && wf && bias && out && masks && train_shape_ok(C, L, n), "bz_train_tower: bad arguments (C = 64 or 128, n_layers even)");
    const int P = bz_train_positions_per_workgroup(C);
    BZ_REQUIRE(n % P == 0, "bz_train_tower: the batch must be a multiple of the positions per workgroup (8 at C = 64, 4 at C = 128)");
    if (bz_device_count() <= 0) { set_error("bz_train_tower: no HIP device (the training kernels have no CPU path)"); return BZ_ENOGPU; }
    TrainArgs T;
    T.n = n; T.n_layers = L; T.wf = static_cast<const uint4*>(wf); T.bias = bias; T.in = static_cast<const __bf16*>(in);
    T.out = static_cast<__bf16*>(out); T.masks = static_cast<uint4*>(masks);
    hipStream_t s = (hipStream_t)stream;
#define BZ_TRAIN_LAUNCH(KERNEL, GEOM)                                                                              \
    do {                                                                                                           \
        static unsigned done = 0;   /* per (kernel, device) */                                                   \
        if (hipError_t e_ = lds_attr_per_device(reinterpret_cast<const void*>(KERNEL<GEOM>), GEOM::LDS, &done); e_ != hipSuccess) \
            return hip_fail(e_, "hipFuncSetAttribute(training kernel)");                                           \
        hipLaunchKernelGGL(KERNEL<GEOM>, dim3(n / GEOM::P), dim3(256), GEOM::LDS, s, T);                           \
    } while (0)
    // C = 64 keeps 8 positions per workgroup: n / 8 workgroups.  Below ~3/4 of the chip's 256 CUs, halve the tile instead
    // (4 positions per workgroup, 2 per wave: twice the weight stream per position, but twice the CUs at work)
    typedef Tw<64, 4> Tw64Half;
    // 128 channels: the inference tower's 16x16x32 path (v_mfma_f32_16x16x32_bf16: same cycles, higher clock under load)
    typedef Tw<128, 4, true> Tw128M16;
    if (C == 64 && n / 8 < 192) { if (bwd) BZ_TRAIN_LAUNCH(k_train_bwd, Tw64Half); else BZ_TRAIN_LAUNCH(k_train_fwd, Tw64Half); }
    else if (C == 64) { if (bwd) BZ_TRAIN_LAUNCH(k_train_bwd, Tw<64>); else BZ_TRAIN_LAUNCH(k_train_fwd, Tw<64>); }
    else { if (bwd) BZ_TRAIN_LAUNCH(k_train_bwd, Tw128M16); else BZ_TRAIN_LAUNCH(k_train_fwd, Tw128M16); }
#undef BZ_TRAIN_LAUNCH
    BZ_LAUNCH_CHECK("k_train_fwd / k_train_bwd");
    return BZ_OK;
}

BZ_EXPORT int32_t bz_train_tower_fwd(const void* act0, const void* wf_fwd, const float* bias, int32_t C, int32_t n_layers, int32_t n,
                                     void* acts_out, void* masks, void* stream) {
    return train_tower(false, act0, wf_fwd, bias, C, n_layers, n, acts_out, masks, stream);
}
BZ_EXPORT int32_t bz_train_tower_bwd(const void* g_top, const void* wf_bwd, const float* zeros_c, const void* masks, int32_t C,
                                     int32_t n_layers, int32_t n, void* gs_out, void* stream) {
    return train_tower(true, g_top, wf_bwd, zeros_c, C, n_layers, n, gs_out, const_cast<void*>(masks), stream);
}

BZ_EXPORT int32_t bz_train_wgrad(const void* acts, const void* gs, int32_t C, int32_t n_layers, int32_t n, int32_t splits, float* partial,
                                 float* db_partial, void* stream) {
    BZ_REQUIRE(acts && gs && partial && db_partial && train_shape_ok(C, n_layers, n) && splits >= 1, "bz_train_wgrad: bad arguments");
    BZ_REQUIRE(n % (256 / C) == 0 && splits <= n / (256 / C), "bz_train_wgrad: the batch must be a multiple of 256 / C and hold at least `splits` stages");
    if (bz_device_count() <= 0) { set_error("bz_train_wgrad: no HIP device (the training kernels have no CPU path)"); return BZ_ENOGPU; }
    WgradArgs T;
    T.acts = static_cast<const __bf16*>(acts); T.gs = static_cast<const __bf16*>(gs); T.partial = partial; T.db_partial = db_partial; T.n = n; T.L = n_layers; T.S = splits;
    const int items = n_layers * splits, grid = ((items + 7) / 8) * 8 * (C / 64);
    hipStream_t s = (hipStream_t)stream;
    if (C == 64) {
        static unsigned done = 0;
        if (hipError_t e = lds_attr_per_device(reinterpret_cast<const void*>(k_train_wgrad<64>), Wg<64>::LDS, &done); e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_train_wgrad)");
        hipLaunchKernelGGL(k_train_wgrad<64>, dim3(grid), dim3(Wg<64>::NT), Wg<64>::LDS, s, T);
    } else {
        static unsigned done = 0;
        if (hipError_t e = lds_attr_per_device(reinterpret_cast<const void*>(k_train_wgrad<128>), Wg<128>::LDS, &done); e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_train_wgrad)");
        hipLaunchKernelGGL(k_train_wgrad<128>, dim3(grid), dim3(Wg<128>::NT), Wg<128>::LDS, s, T);
    }
    BZ_LAUNCH_CHECK("k_train_wgrad");
    return BZ_OK;
}
