// bz_net.hip -- policy/value conv net forward for 8x8 boards on gfx950.
//
// Architecture (build-authored, SURVEY.md 8(d) "net"): input 2 planes (own, opp)
// 8x8; stem conv3x3 2->C + ReLU; NB residual blocks (conv3x3, ReLU, conv3x3,
// +skip, ReLU); policy head conv1x1 C->2, ReLU, FC 128->65; value head conv1x1
// C->1, ReLU, FC 64->VH, ReLU, FC VH->1, tanh.  226.86 MFLOP per position at
// C=128, NB=6.  The calling convention generalises AIPlayer.get_move
// (src/tic_tac_toe/players.py:84-98): side-to-move canonical input, logits out.
//
// Two paths:
//  * bf16 (product): the residual tower is ONE kernel (k_tower_bf16).  A
//    workgroup of 4 waves keeps the activations of 4 positions resident in LDS
//    (2 x 73 KB, XOR-swizzled 256-B cells, board rows at a pitch of 9 cells
//    with in-row zero cells for the conv halo) across all 2*NB conv layers;
//    wave w owns output channels 32w..32w+31 of all 4 positions (8 accumulator
//    tiles of v_mfma_f32_32x32x16_bf16 = the 8 board rows of the 4 positions,
//    so the taps that shift by a row skip the tile that would read only
//    padding), streams its weight fragments straight from
//    L2 into registers in a fragment-major layout (one coalesced 1 KB load per
//    k-step, prefetched one tap = 8 k-steps ahead) and reads the activation
//    fragments from LDS with conflict-free ds_read_b128.  Activations never
//    touch HBM between the stem and the heads; one barrier per layer.
//  * f32 (parity): per-layer VALU kernels whose every accumulation is the
//    k-ordered fmaf chain of oracle/bz_oracle.c -> bit-identical to the oracle.
#include <math.h>
#include <stdlib.h>
#include <new>
#include <atomic>
#include <vector>

#include "bz_common.h"
#include "bz_math.h"
#include "bz_tower.h"

using namespace bz;

using namespace bz_tower;

struct bz_net {
    int C, NB, VH, max_batch;
    // parameters (device)
    float *stem_w, *stem_b;          // [9][2][C], [C]
    float *conv_w, *conv_b;          // [2NB][9][C][C] as [tap][ci][co]; [2NB][C]
    __bf16* conv_wf;                 // [2NB+pad][9][8][4][64][8] fragment-major (C == 128)
    __bf16 *stem_wf, *head_wf;       // [2][4][64][8], [8][64][8] fragments (C == 128)
    __bf16 *conv_wf16, *stem_wf16;   // C == 128: the 16x16x32 path's fragments, [2NB+pad][9][4 kq][4 wt][2 a][64][8], [4 wt][2 a][64][8]
    uint8_t *conv_wf8, *head_wf8;    // e4m3: [L*9+1][2][4][2][64][16], [2][2][64][16]
    float *dq8, *head_dq8, *ones;    // [L][128], [4], [128]
    float *pol_w, *pol_b, *polfc_wT, *polfc_b;  // [2][C], [2], [128][65], [65]
    float *val_w, *val_b, *v1_wT, *v1_b, *v2_w, *v2_b;  // [C], [1], [64][VH], [VH], [VH], [1]
    // activations (device)
    float *act_a, *act_b;            // f32 parity path: [max_batch][64][C] x 2 (the MFMA paths keep activations in LDS)
    void* ws_base;
    // the f32 parity path ping-pongs through act_a/act_b, which belong to the net: forwards issued on
    // different streams are ordered through this event (the MFMA paths have no such scratch)
    hipEvent_t f32_done;
    bool f32_used;
    // stamp of the parameter upload the weights came from, unique over all nets of the process: an engine's evaluation
    // cache carries nothing over from a search made with another stamp
    uint64_t epoch;
};

namespace {

// ------------------------------------------------------------------ stem
// thread = output channel, block = position.  x in {0,1}: fmaf(1,w,acc) == acc + w.
template <class OutT>
__global__ void k_stem(const u64* __restrict__ own, const u64* __restrict__ opp, int n, const u32* n_dev, int C,
                       const float* __restrict__ w, const float* __restrict__ b, OutT* __restrict__ out) {
    int pos = blockIdx.x, co = threadIdx.x;
    if (n_dev) n = (int)*n_dev;
    if (pos >= n || co >= C) return;
    u64 me = own[pos], you = opp[pos];
    float wr[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) wr[i] = w[i * C + co];
    float bias = b[co];
    for (int cell = 0; cell < 64; ++cell) {
        int y = cell >> 3, x = cell & 7;
        float acc = bias;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if (yy < 0 || yy > 7 || xx < 0 || xx > 7) continue;
            int c2 = yy * 8 + xx;
            if ((me >> c2) & 1ULL) acc = acc + wr[2 * t];
            if ((you >> c2) & 1ULL) acc = acc + wr[2 * t + 1];
        }
        acc = acc > 0.0f ? acc : 0.0f;
        out[((size_t)pos * 64 + cell) * C + co] = (OutT)acc;
    }
}

// ------------------------------------------------------------------ f32 conv (parity path)
// block = position, 256 threads; thread = (co, cell group); 8 cells per pass.
__global__ void __launch_bounds__(256) k_conv_f32(const float* __restrict__ in, const float* __restrict__ w,
                                                  const float* __restrict__ b, const float* skip, float* out, int n,
                                                  const u32* n_dev, int C) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* xs = reinterpret_cast<float*>(smem_raw);  // [64][C]
    int pos = blockIdx.x;
    if (n_dev) n = (int)*n_dev;
    if (pos >= n) return;
    const float* xin = in + (size_t)pos * 64 * C;
    for (int i = threadIdx.x; i < 64 * C; i += 256) xs[i] = xin[i];
    __syncthreads();
    int co = threadIdx.x % C, cg = threadIdx.x / C, groups = 256 / C, cells_per = 64 / groups;
    float bias = b[co];
    for (int c0 = cg * cells_per; c0 < (cg + 1) * cells_per; c0 += 8) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = bias;
        for (int t = 0; t < 9; ++t) {
            int dy = t / 3 - 1, dx = t % 3 - 1;
            int src[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int cell = c0 + j, yy = (cell >> 3) + dy, xx = (cell & 7) + dx;
                src[j] = (yy < 0 || yy > 7 || xx < 0 || xx > 7) ? -1 : (yy * 8 + xx) * C;
            }
            const float* wt = w + (size_t)t * C * C + co;
            for (int ci = 0; ci < C; ++ci) {
                float wv = wt[(size_t)ci * C];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (src[j] >= 0) acc[j] = __builtin_fmaf(xs[src[j] + ci], wv, acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            size_t o = ((size_t)pos * 64 + c0 + j) * C + co;
            float v = acc[j];
            if (skip) v = v + skip[o];
            out[o] = v > 0.0f ? v : 0.0f;
        }
    }
}

// ------------------------------------------------------------------ heads (both paths)
// block = position, 192 threads.  Every dot product is a sequential fmaf chain in
// the oracle's order, so with f32 activations the result is bit-identical.
template <class InT>
__global__ void __launch_bounds__(192) k_heads(const InT* __restrict__ act, int n, const u32* n_dev, int C, int VH,
                                               const float* __restrict__ pol_w, const float* __restrict__ pol_b,
                                               const float* __restrict__ polfc_wT, const float* __restrict__ polfc_b,
                                               const float* __restrict__ val_w, const float* __restrict__ val_b,
                                               const float* __restrict__ v1_wT, const float* __restrict__ v1_b,
                                               const float* __restrict__ v2_w, const float* __restrict__ v2_b,
                                               float* __restrict__ logits, float* __restrict__ value) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* xs = reinterpret_cast<float*>(smem_raw);  // [64][C+1]
    float* pf = xs + 64 * (C + 1);                   // [128]
    float* vf = pf + 128;                            // [64]
    float* vh = vf + 64;                             // [VH]
    int pos = blockIdx.x, tid = threadIdx.x;
    if (n_dev) n = (int)*n_dev;
    if (pos >= n) return;
    const InT* x = act + (size_t)pos * 64 * C;
    for (int i = tid; i < 64 * C; i += 192) xs[(i / C) * (C + 1) + (i % C)] = (float)x[i];
    __syncthreads();
    {   // conv1x1: threads 0..127 -> policy (j, cell); 128..191 -> value (cell)
        int cell = tid & 63, j = tid >> 6;
        const float* wj = j < 2 ? pol_w + (size_t)j * C : val_w;
        float acc = j < 2 ? pol_b[j] : val_b[0];
        const float* xi = xs + cell * (C + 1);
        for (int c = 0; c < C; ++c) acc = __builtin_fmaf(xi[c], wj[c], acc);
        acc = acc > 0.0f ? acc : 0.0f;
        if (j < 2) pf[j * 64 + cell] = acc; else vf[cell] = acc;
    }
    __syncthreads();
    if (tid < 65) {
        float acc = polfc_b[tid];
        for (int i = 0; i < 128; ++i) acc = __builtin_fmaf(pf[i], polfc_wT[i * 65 + tid], acc);
        logits[(size_t)pos * 65 + tid] = acc;
    } else if (tid >= 128 && tid - 128 < VH) {
        int h = tid - 128;
        float acc = v1_b[h];
        for (int i = 0; i < 64; ++i) acc = __builtin_fmaf(vf[i], v1_wT[i * VH + h], acc);
        vh[h] = acc > 0.0f ? acc : 0.0f;
    }
    __syncthreads();
    if (tid == 0) {
        float acc = v2_b[0];
        for (int h = 0; h < VH; ++h) acc = __builtin_fmaf(vh[h], v2_w[h], acc);
        value[pos] = tanhf_spec(acc);
    }
}

typedef Tw<64, 2> TwS64;     // latency shapes: the smallest P whose (M-tile, position) units fill 4 waves
typedef Tw<128, 1> TwS128;
typedef Tw<256, 1> TwS256;
typedef Tw<128, 4, true> TwM16;  // the benchmark net's throughput shape on v_mfma_f32_16x16x32_bf16 (bz_tower.h)
constexpr int kSmallBatch = 256;  // one workgroup per position and per CU up to here

struct TowerArgs {
    const u64 *own, *opp;            // [n] bitboards, side-to-move canonical
    const u32* n_dev;                // optional device-side count (<= n): workgroups beyond it exit at once
    int n, n_layers, VH;
    const uint4* wf;                 // tower weight fragments (see bz_net_create)
    const uint4 *wf16, *stem_wf16;   // the same for the 16x16x32 path: [l][t][kq][wt][a][lane], [wt][a][lane]
    const float* bias;               // [n_layers][128]
    const uint4* stem_wf;            // [2][4][64] fragments of the stem as a K=32 GEMM (k = 2*tap + plane)
    const float* stem_b;             // [128]
    const uint4* head_wf;            // [8][64] fragments: rows 0,1 = policy conv1x1, row 2 = value conv1x1
    const uint4 *wf8, *head_wf8;     // fp8 (e4m3) fragments for the MX-scaled 32x32x64 MFMA
    const float *dq8, *head_dq8, *ones;  // [n_layers][128] dequant factors 1/(s_w*16), [4], [128] x 1.0f
    const float *pol_b, *val_b;      // [2], [1]
    const float *polfc_wT, *polfc_b; // [128][65], [65]
    const float *v1_wT, *v1_b, *v2_w, *v2_b;  // [64][VH], [VH], [VH], [1]
    float *logits, *value;           // [n][65], [n]
};

// 3x3 neighbourhood of `cell` on bitboard b: tap t = 3 (dy + 1) + (dx + 1) at bit 8 (dy + 1) + (dx + 1); cells off the
// board read 0 (rows fall out of the 64-bit word, the two wrap-around columns are masked)
__device__ __forceinline__ unsigned nbhd(u64 b, int cell) {
    const int s = cell - 9;
    const u64 w = s >= 0 ? b >> s : b << -s;
    unsigned n = (unsigned)w & 0x00070707u;
    const int x = cell & 7;
    if (x == 0) n &= ~0x00010101u;  // column -1 would be the previous row's column 7
    if (x == 7) n &= ~0x00040404u;
    return n;
}
// stem input fragment: B[k][cell] with k = 2*tap + plane (k < 18), for the lane's k-group kbase = 16 KC + 8 h, from the
// two neighbourhood words: four taps x (own, opp) as bf16 0.0 / 1.0
template <int KC>
__device__ __forceinline__ bf16x8 stem_frag(unsigned n_own, unsigned n_opp, int h) {
    unsigned wd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        constexpr int none = 31;  // bit 31 of a neighbourhood word is always 0
        const int t0 = 8 * KC + j, t1 = t0 + 4;  // tap for h = 0 / h = 1
        const int b0 = t0 <= 8 ? (t0 / 3) * 8 + t0 % 3 : none, b1 = t1 <= 8 ? (t1 / 3) * 8 + t1 % 3 : none;
        const int bit = h ? b1 : b0;
        const unsigned o = (n_own >> bit) & 1u, p = (n_opp >> bit) & 1u;
        wd[j] = (o | (p << 16)) * 0x3F80u;  // bf16 1.0 in the low (own) / high (opp) half
    }
    uint4 u = make_uint4(wd[0], wd[1], wd[2], wd[3]);
    return __builtin_bit_cast(bf16x8, u);
}

// the same for the 16x16x32 MFMA: lane k-group g holds k = 8g .. 8g + 7, i.e. taps 4g .. 4g + 3 (own, opp)
__device__ __forceinline__ bf16x8 stem_frag16(unsigned n_own, unsigned n_opp, int g) {
    unsigned wd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = 4 * g + j;                                   // tap; valid up to 8
        const int bit = t <= 8 ? (t / 3) * 8 + t % 3 : 31;         // bit 31 of a neighbourhood word is always 0
        const unsigned o = (n_own >> bit) & 1u, p = (n_opp >> bit) & 1u;
        wd[j] = (o | (p << 16)) * 0x3F80u;
    }
    uint4 u = make_uint4(wd[0], wd[1], wd[2], wd[3]);
    return __builtin_bit_cast(bf16x8, u);
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// The whole net forward for P positions per workgroup: stem (MFMA, K = 18 padded to 32, fed from
// the bitboards) -> residual tower (activations resident in LDS) -> heads (conv1x1 by MFMA, the
// small FCs by one wave per position).  HBM traffic per position: 16 B in, 264 B out.
template <class G>
__global__ void __launch_bounds__(256, 1)
k_tower_bf16(TowerArgs T) {
    constexpr int C = G::C, P = G::P, PW = G::PW, MW = G::MW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pos0 = blockIdx.x * P;
    if (T.n_dev) T.n = (int)*T.n_dev;
    if (pos0 >= T.n) return;  // block-uniform, before any barrier
    [[maybe_unused]] unsigned long long tacc[4] = {0, 0, 0, 0}, tk0 = 0, tk1 = 0, tr0 = 0, tr1 = 0;
    BZ_STAMP(tk0);
#ifdef BZ_EXP_STAMPS
    tr0 = __builtin_amdgcn_s_memrealtime();
#endif
    char* bufX = smem;
    char* bufM = smem + G::BUF;
    const int r = lane & 31, h = lane >> 5;

    // ---- zero cells (conv halo: cell indices 0, 9, .., 72 of every position) of both buffers
    constexpr int ZC = G::CELL / 16;  // 16-byte chunks per cell
    for (int i = tid; i < 2 * P * 9 * ZC; i += 256) {
        int k = i % ZC, j = (i / ZC) % 9, pb = i / (9 * ZC);  // pb = buffer * P + position: BUF = P * TILE
        *reinterpret_cast<uint4*>(smem + pb * G::TILE + j * G::ROWC * G::CELL + k * 16) = make_uint4(0, 0, 0, 0);
    }
    // weight-fragment stream of this wave: k-step ks, M-tile mt -> wf[(ks * MT + mt) * 64 + lane], linear over layers
    const int wt0 = G::wt0(w), wp0 = G::pos0(w);
    // wave-uniform base (scalar registers) + lane: the loads take the SGPR-base addressing mode, so advancing the stream
    // costs scalar adds instead of 64-bit vector adds between the MFMAs
    const uint4* ap = G::M16 ? T.wf16 + (size_t)wt0 * 2 * 64 : T.wf + (size_t)wt0 * 64;
    typename WSetsOf<G>::type WS;
    if constexpr (G::M16) {  // tap 0 of the first layer: KQ steps x 2 channel halves
#pragma unroll
        for (int kq = 0; kq < G::KQ; ++kq)
#pragma unroll
            for (int a = 0; a < 2; ++a) WS.s[0][kq][a] = __builtin_bit_cast(bf16x8, BZ_WLOAD(&ap[(kq * G::MT * 2 + a) * 64 + (unsigned)lane]));
        ap += G::KQ * G::MT * 2 * 64;
    } else {
#pragma unroll
        for (int d = 0; d + 1 < G::DEPTH; ++d) {  // chunks 0 .. DEPTH - 2 of the first layer
#pragma unroll
            for (int kc = 0; kc < G::KS; ++kc)
#pragma unroll
                for (int mt = 0; mt < MW; ++mt) WS.s[d][kc][mt] = __builtin_bit_cast(bf16x8, BZ_WLOAD(&ap[(kc * G::MT + mt) * 64 + (unsigned)lane]));
            ap += G::KS * G::MT * 64;
        }
    }

    // ---- stem: conv3x3 2 -> C as a [C x 32] x [32 x 64] GEMM per position
    if constexpr (G::M16) {  // K = 32 is ONE 16x16x32 MFMA per quarter: lane (c, g) feeds cell c of half b with k = 8g ..
        f32x16 acc[MW][G::NU];
        Bias<G> bias;
        const int c = lane & 15, g = lane >> 4;
        load_bias16<G>(bias, T.stem_b, wt0, g);
        bf16x8 sa[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) sa[a] = __builtin_bit_cast(bf16x8, T.stem_wf16[(wt0 * 2 + a) * 64 + lane]);
        u64 own[2], opp[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            int pos = pos0 + wp0 + 2 * b + (c >> 3);
            pos = pos < T.n ? pos : T.n - 1;
            own[b] = T.own[pos]; opp[b] = T.opp[pos];
        }
#pragma unroll
        for (int u = 0; u < G::NU; ++u) {
            acc[0][u] = (f32x16)(0.0f);
            const int cell = 8 * u + (c & 7);
            const bf16x8 f0 = stem_frag16(nbhd(own[0], cell), nbhd(opp[0], cell), g);
            const bf16x8 f1 = stem_frag16(nbhd(own[1], cell), nbhd(opp[1], cell), g);
            mfma16_quarter<0>(acc[0][u], sa[0], f0);
            mfma16_quarter<1>(acc[0][u], sa[0], f1);
            mfma16_quarter<2>(acc[0][u], sa[1], f0);
            mfma16_quarter<3>(acc[0][u], sa[1], f1);
        }
        epilogue16<G>(acc, bufX + wp0 * G::TILE, false, bias, wt0, lane);
    } else {
        f32x16 acc[MW][G::NU];
        Bias<G> bias;
        load_bias<G>(bias, T.stem_b, wt0, h);
        bf16x8 sa[2][MW];
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
            for (int mt = 0; mt < MW; ++mt) sa[kc][mt] = __builtin_bit_cast(bf16x8, T.stem_wf[(kc * G::MT + wt0 + mt) * 64 + lane]);
        constexpr int NB = G::ROWT ? 1 : PW;  // row-tile units: every lane feeds ONE position (r >> 3) in all units
        u64 own[NB], opp[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int pos = pos0 + wp0 + (G::ROWT ? r >> 3 : i);
            pos = pos < T.n ? pos : T.n - 1;
            own[i] = T.own[pos]; opp[i] = T.opp[pos];
        }
#pragma unroll
        for (int u = 0; u < G::NU; ++u) {
            const int i = G::ROWT ? 0 : u >> 1;
            const int cell = G::unit_cell(u, r);
            const unsigned n_own = nbhd(own[i], cell), n_opp = nbhd(opp[i], cell);
            bf16x8 sf[2] = {stem_frag<0>(n_own, n_opp, h), stem_frag<1>(n_own, n_opp, h)};
#pragma unroll
            for (int mt = 0; mt < MW; ++mt) {
                acc[mt][u] = (f32x16)(0.0f);
#pragma unroll
                for (int kc = 0; kc < 2; ++kc)
                    acc[mt][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[kc][mt], sf[kc], acc[mt][u], 0, 0, 0);
            }
        }
        epilogue<G>(acc, bufX + wp0 * G::TILE, false, bias, wt0, r, h);
    }
    __syncthreads();

    // ---- tower: a residual block = conv1 (X -> M) + conv2 (M -> X in place, + skip X)
#pragma unroll 1
    for (int blk = 0; blk < T.n_layers / 2; ++blk) {
        conv_layer<0, G>(bufX, bufM, false, T.bias + (size_t)(2 * blk) * C, WS, ap, w, r, h, tacc);
        conv_layer<G::NCH % G::DEPTH, G>(bufM, bufX, true, T.bias + (size_t)(2 * blk + 1) * C, WS, ap, w, r, h, tacc);
    }
    BZ_STAMP(tk1);

    // ---- heads: wave w serves positions w, w + 4, ...  conv1x1 (policy 2 ch + value 1 ch) by MFMA against
    // the resident tile, then the FCs in fp32 with the position's 192 features staged in LDS (M is dead now).
    float* S = reinterpret_cast<float*>(bufM + w * 1024);  // [pf 128 | vf 64], one scratch per wave
    const float pb0 = T.pol_b[0], pb1 = T.pol_b[1], vb = T.val_b[0];
    bf16x8 hw[G::KC];  // all head-conv fragments in flight at once (one L2 round trip, not one per MFMA)
#pragma unroll
    for (int kc = 0; kc < G::KC; ++kc) hw[kc] = __builtin_bit_cast(bf16x8, T.head_wf[kc * 64 + lane]);
    for (int p = w; p < P && pos0 + p < T.n; p += 4) {
        const int pos = pos0 + p;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            f32x16 acc = (f32x16)(0.0f);
            const int cell = 32 * nt + r;
#pragma unroll
            for (int kc = 0; kc < G::KC; ++kc) {
                bf16x8 b = *reinterpret_cast<const bf16x8*>(bufX + p * G::TILE + G::cell_off(p, cell, 2 * kc + h));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hw[kc], b, acc, 0, 0, 0);
            }
            if (h == 0) {  // rows 0..2 of D live in registers 0..2 of lanes 0..31
                float a0 = acc[0] + pb0, a1 = acc[1] + pb1, a2 = acc[2] + vb;
                S[cell] = a0 > 0.0f ? a0 : 0.0f;
                S[64 + cell] = a1 > 0.0f ? a1 : 0.0f;
                S[128 + cell] = a2 > 0.0f ? a2 : 0.0f;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's own LDS writes have landed
        // policy FC 128 -> 65: lane a owns logit a; logit 64 (pass) is a wave reduction
        // (the fma chains below keep their order; the unroll factors only decide how many weight loads are in flight)
        float acc = T.polfc_b[lane], part = 0.0f;
#pragma unroll 8
        for (int i = 0; i < 128; i += 4) {
            f32x4 s4 = *reinterpret_cast<const f32x4*>(S + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_fmaf(s4[j], T.polfc_wT[(i + j) * 65 + lane], acc);
        }
        part = S[lane] * T.polfc_wT[lane * 65 + 64] + S[lane + 64] * T.polfc_wT[(lane + 64) * 65 + 64];
        part = wave_sum(part);
        T.logits[(size_t)pos * 65 + lane] = acc;
        if (lane == 0) T.logits[(size_t)pos * 65 + 64] = part + T.polfc_b[64];
        // value FC 64 -> VH -> 1, tanh
        float vh = 0.0f;
        if (lane < T.VH) {
            float a = T.v1_b[lane];
#pragma unroll 32
            for (int i = 0; i < 64; ++i) a = __builtin_fmaf(S[128 + i], T.v1_wT[i * T.VH + lane], a);
            vh = (a > 0.0f ? a : 0.0f) * T.v2_w[lane];
        }
        vh = wave_sum(vh);
        if (lane == 0) T.value[pos] = tanhf_spec(vh + T.v2_b[0]);
        __builtin_amdgcn_s_waitcnt(0xC07F);  // the scratch is reused for the wave's next position
    }
#ifdef BZ_EXP_STAMPS
    unsigned long long tk2; BZ_STAMP(tk2);
    tr1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && blockIdx.x < 4096) {
        unsigned long long* d = g_dbg + blockIdx.x * 8;
        d[0] = tacc[0]; d[1] = tacc[1]; d[2] = tacc[2]; d[3] = tk1 - tk0; d[4] = tk2 - tk0; d[5] = tr1 - tr0; d[6] = tk0; d[7] = tr0;
    }
#endif
}


// ====================================================================================
// fp8 variant (BASELINE config 5): same decomposition, tower on
// v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3, unit block scales: 2x the bf16 MFMA rate per
// clock, half the operand bytes).  Operand map verified on hardware with
// tools/probe/probe_mfma_fp8.hip: A lane l = row l&31, B lane l = col l&31, byte j of lane half h
// pairs with byte j of half h (we define k = 32h + j); D as for bf16.  Quantisation spec
// (betazero_amd/quant.py, oracle mode 2): weights e4m3(w * s_co) with a per-output-channel
// power-of-two scale, activations stored as e4m3(x * 16); the epilogue multiplies the fp32
// accumulator by 1/(s_co * 16), adds bias (+ skip / 16), ReLUs, clamps to 448/16 and converts.
// Units and LDS image as in the bf16 kernel's row-tile shape: the wave's eight 32-cell units are the eight BOARD ROWS
// of its four positions (lane r -> position r >> 3, column r & 7), so a tap with row shift dy = -1 skips unit 0 and
// dy = +1 skips unit 7 (their inputs are zero padding): 8.3 % fewer MFMAs, bit-identical sums.  A position is 73
// cells of 128 B (8 chunks of 16 B): 8 rows at a pitch of 9 cells (8 squares + a zero cell that is the x = 8 halo of
// its row and the x = -1 halo of the next) + one leading zero cell.  Chunk c of a cell sits at slot c ^ sw,
// sw = (x >> 1) | (position & 1) << 2; a 256-B bank row holds two cells (cell-index parity = x parity inside a
// ds_read_b128 lane group), so the group's 16 lanes (8 columns x 2 positions) hit 16 distinct slots for every tap.
// ====================================================================================
namespace f8 {
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kCell = 128, kRowC = 9;
constexpr int kTile = (8 * kRowC + 1) * kCell;  // 9,344
constexpr int kBuf = 4 * kTile;                 // 37,376
constexpr int kLds = 2 * kBuf;                  // 74,752: two workgroups per CU
constexpr float kActScale = 16.0f;
constexpr int kUnit = 0x7F7F7F7F;   // E8M0 127 = 2^0 in every byte

__device__ __forceinline__ constexpr int cell_at(int y, int x) { return (y * kRowC + x + 1) * kCell; }  // x = -1 .. 8
__device__ __forceinline__ int sw3(int p, int x) { return ((x & 7) >> 1) | ((p & 1) << 2); }
// per-lane offset (from the workgroup's first position) of chunk 2h of the cell in row 0, column x + DX of the lane's
// position; the caller XORs (ks << 6) in and adds the unit's row as a compile-time offset
template <int DX>
__device__ __forceinline__ int tap_off(int r, int h) {
    const int xx = (r & 7) + DX;
    return (r >> 3) * kTile + cell_at(0, xx) + (((2 * h) ^ sw3(r >> 3, xx)) << 4);
}
__device__ __forceinline__ v8i ld32(const char* p0, int off, int imm) {
    v4i lo = *reinterpret_cast<const v4i*>(p0 + off + imm);
    v4i hi = *reinterpret_cast<const v4i*>(p0 + (off ^ 16) + imm);
    v8i v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return v;
}
constexpr int unit_lo(int dy) { return dy < 0 ? 1 : 0; }
constexpr int unit_hi(int dy) { return dy > 0 ? 7 : 8; }
// activation fragments of one half-step (units 4 pp .. 4 pp + 3 = board rows) of k-step ks for tap TAP
template <int TAP>
__device__ __forceinline__ void load_b(v8i (&b)[4], const char* in, int boff, int ks, int pp) {
    constexpr int dy = TAP / 3 - 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int u = 4 * pp + j;
        if (u >= unit_lo(dy) && u < unit_hi(dy)) b[j] = ld32(in, boff ^ (ks << 6), (u + dy) * kRowC * kCell);
    }
}
#if defined(BZ_EXP_MFMA16_FP8) && !defined(BZ_EXPERIMENT)
#error "BZ_EXP_MFMA16_FP8 is a diagnostic variant: build it through betazero_amd.build.build_variant()"
#endif
template <int TAP>
__device__ __forceinline__ void mfma4(f32x16 (&acc)[8], const v8i& a, const v8i (&b)[4], int pp, [[maybe_unused]] int kpar = 0) {
    constexpr int dy = TAP / 3 - 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int u = 4 * pp + j;
        if (u >= unit_lo(dy) && u < unit_hi(dy)) {
#ifdef BZ_EXP_MFMA16_FP8
            // TIMING ONLY (wrong results): the 32x32x64 MFMA as two v_mfma_scale_f32_16x16x128_f8f6f4 on the same operand
            // registers (same MACs, 2 x 32 cycles) -- tools/exp_ab_mfma16.sh FP8=1; second product with swapped operands
            // so that the compiler cannot merge the two
            f32x16& c = acc[u];
            f32x4 q0 = kpar ? f32x4{c[8], c[9], c[10], c[11]} : f32x4{c[0], c[1], c[2], c[3]};
            f32x4 q1 = kpar ? f32x4{c[12], c[13], c[14], c[15]} : f32x4{c[4], c[5], c[6], c[7]};
            q0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b[j], q0, 0, 0, 0, kUnit, 0, kUnit);
            q1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b[j], a, q1, 0, 0, 0, kUnit, 0, kUnit);
            if (kpar) { c[8] = q0[0]; c[9] = q0[1]; c[10] = q0[2]; c[11] = q0[3]; c[12] = q1[0]; c[13] = q1[1]; c[14] = q1[2]; c[15] = q1[3]; }
            else { c[0] = q0[0]; c[1] = q0[1]; c[2] = q0[2]; c[3] = q0[3]; c[4] = q1[0]; c[5] = q1[1]; c[6] = q1[2]; c[7] = q1[3]; }
#else
            acc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b[j], acc[u], 0, 0, 0, kUnit, 0, kUnit);
#endif
        }
    }
}

// one conv tap (compile-time: the skipped units cost nothing) = 2 k-steps x 2 half-steps of up to 4 MFMAs
template <int S, int TAP>
__device__ __forceinline__ void tap_step(f32x16 (&acc)[8], v8i (&A0)[2], v8i (&A1)[2], const uint4*& ap, const char* in,
                                         int& boff, int r, int h, v8i (&b0)[4], v8i (&b1)[4]) {
    constexpr bool last = TAP == 8;
    constexpr int TAP_N = last ? TAP : TAP + 1;
    v8i (&use)[2] = S ? A1 : A0;
    v8i (&nxt)[2] = S ? A0 : A1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        uint4 lo = ap[((ks * 4) * 2 + 0) * 64 + (unsigned)(32 * h + r)], hi = ap[((ks * 4) * 2 + 1) * 64 + (unsigned)(32 * h + r)];
        v8i v = {(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        nxt[ks] = v;
    }
    ap += 2 * 4 * 2 * 64;
    int boff_n = boff;
    if constexpr (TAP_N % 3 != TAP % 3) boff_n = tap_off<TAP_N % 3 - 1>(r, h);
    load_b<TAP>(b1, in, boff, 0, 1);
    mfma4<TAP>(acc, use[0], b0, 0, 0);
    load_b<TAP>(b0, in, boff, 1, 0);
    mfma4<TAP>(acc, use[0], b1, 1, 0);
    load_b<TAP>(b1, in, boff, 1, 1);
    mfma4<TAP>(acc, use[1], b0, 0, 1);
    if constexpr (!last) load_b<TAP_N>(b0, in, boff_n, 0, 0);  // first half-step of the next tap
    mfma4<TAP>(acc, use[1], b1, 1, 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#ifdef BZ_EXP_MFMA16_FP8
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // the unit's 2 half-size MFMAs
#else
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
#endif
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // 2 DS reads
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read
    }
    boff = boff_n;
    __builtin_amdgcn_sched_barrier(0);  // one tap per scheduling region (straight-line layer)
}
template <int S0, int TAP>
__device__ __forceinline__ void run_taps(f32x16 (&acc)[8], v8i (&A0)[2], v8i (&A1)[2], const uint4*& ap, const char* in,
                                         int& boff, int r, int h, v8i (&b0)[4], v8i (&b1)[4]) {
    if constexpr (TAP < 9) {
        tap_step<(S0 + TAP) % 2, TAP>(acc, A0, A1, ap, in, boff, r, h, b0, b1);
        run_taps<S0, TAP + 1>(acc, A0, A1, ap, in, boff, r, h, b0, b1);
    }
}

// e4m3(16 * relu(acc * dq + bias (+ skip))) -> LDS, computed in the x16 domain: fma(acc, 16 dq, 16 bias)
// (+ the stored skip code, which already is 16 x), one v_med3 for ReLU + saturation, cvt_pk.
// lane (r, h) register 4q+i of unit u = channel 32w + 8q + 4h + i of board cell (row u, column r & 7) of position r >> 3
struct Scale { f32x4 dq[4], b[4]; };  // 16 x dequant factor and 16 x bias of the lane's channels
__device__ __forceinline__ void load_scale(Scale& sc, const float* __restrict__ dq, const float* __restrict__ bl, int w, int h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // fetched before the K-loop (see the bf16 kernel's load_bias)
        sc.dq[q] = *reinterpret_cast<const f32x4*>(dq + 32 * w + 4 * h + 8 * q) * kActScale;
        sc.b[q] = *reinterpret_cast<const f32x4*>(bl + 32 * w + 4 * h + 8 * q) * kActScale;
    }
}
__device__ __forceinline__ void epilogue(f32x16 (&acc)[8], char* out, bool second, const Scale& sc, int w, int r, int h) {
    const f32x4 (&dqv)[4] = sc.dq;
    const f32x4 (&bq)[4] = sc.b;
    const int sw = sw3(r >> 3, r & 7);
    // two opaque bases (even / odd rows): every store offset is then a multiple of 256 B from its base, so pairs of
    // 4-byte stores (and skip loads) go out as one ds_write2st64_b32 / ds_read2st64_b32
    int home2[2] = {(r >> 3) * kTile + cell_at(0, r & 7) + 4 * h, (r >> 3) * kTile + cell_at(1, r & 7) + 4 * h};
    asm volatile("" : "+v"(home2[0]), "+v"(home2[1]));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int chunk = 2 * w + (q >> 1);
        const f32x2 dlo = {dqv[q][0], dqv[q][1]}, dhi = {dqv[q][2], dqv[q][3]};
        const f32x2 blo = {bq[q][0], bq[q][1]}, bhi = {bq[q][2], bq[q][3]};
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            int base = home2[par] + ((chunk ^ sw) << 4) + 8 * (q & 1);
            asm volatile("" : "+v"(base));  // keeps the + 8 out of the offset field (st64 offsets count 256-B steps)
#pragma unroll
            for (int u = par; u < 8; u += 2) {  // rows of one parity back to back: their stores pair up
                const int off = base + (u & ~1) * kRowC * kCell;
                f32x2 lo = {acc[u][4 * q], acc[u][4 * q + 1]}, hi = {acc[u][4 * q + 2], acc[u][4 * q + 3]};
                lo = __builtin_elementwise_fma(lo, dlo, blo);  // packed fp32 fma: the same roundings as four scalar fmas
                hi = __builtin_elementwise_fma(hi, dhi, bhi);
                if (second) {
                    int sk = *reinterpret_cast<const int*>(out + off);
                    lo += __builtin_amdgcn_cvt_pk_f32_fp8(sk, false);
                    hi += __builtin_amdgcn_cvt_pk_f32_fp8(sk, true);
                }
                float v[4] = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_fmed3f(v[i], 0.0f, 448.0f);
                int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
                pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
                *reinterpret_cast<int*>(out + off) = pk;
            }
        }
    }
}

template <int S0>
__device__ __forceinline__ void conv_layer(const char* in, char* out, bool second, const float* __restrict__ dq,
                                           const float* __restrict__ bl, v8i (&A0)[2], v8i (&A1)[2], const uint4*& ap,
                                           int w, int r, int h, unsigned long long (&tacc)[4]) {
    [[maybe_unused]] unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    BZ_STAMP(t0);
    f32x16 acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = (f32x16)(0.0f);
    Scale sc;
    load_scale(sc, dq, bl, w, h);
    int boff = tap_off<-1>(r, h);
    v8i b0[4], b1[4];
    load_b<0>(b0, in, boff, 0, 0);
    __builtin_amdgcn_sched_barrier(0);  // keep the prologue reads out of tap 0's scheduling region (see the bf16 kernel)
    run_taps<S0, 0>(acc, A0, A1, ap, in, boff, r, h, b0, b1);  // 9 taps: the weight set alternates, S0 ^ (tap & 1)
    BZ_STAMP(t1);
    epilogue(acc, out, second, sc, w, r, h);
    BZ_STAMP(t2);
#ifndef BZ_EXP_NO_LAYER_BARRIER  // TIMING ONLY (results are wrong without it): the ceiling of any scheme that relaxes the
    __syncthreads();             // per-layer barrier (per-row ready counters, ...) -- tools/exp_ab_barrier.sh, DESIGN.md 5
#endif
    BZ_STAMP(t3);
    tacc[0] += t1 - t0; tacc[1] += t2 - t1; tacc[2] += t3 - t2;
}

// 74.8 KB of LDS per workgroup: two workgroups per CU hide each other's epilogues (launch bound 2 waves per SIMD)
__global__ void __launch_bounds__(256, 2) k_tower_fp8(TowerArgs T) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pos0 = blockIdx.x * 4;
    if (T.n_dev) T.n = (int)*T.n_dev;
    if (pos0 >= T.n) return;
    [[maybe_unused]] unsigned long long tacc[4] = {0, 0, 0, 0}, tk0 = 0, tk1 = 0, tr0 = 0, tr1 = 0;
    BZ_STAMP(tk0);
#ifdef BZ_EXP_STAMPS
    tr0 = __builtin_amdgcn_s_memrealtime();
#endif
    char* bufX = smem;
    char* bufM = smem + kBuf;
    const int r = lane & 31, h = lane >> 5;

    for (int i = tid; i < 2 * 4 * 9 * 8; i += 256) {  // zero cells 0, 9, .., 72 of 2 buffers x 4 positions (8 x 16 B each)
        int k = i & 7, j = (i >> 3) % 9, pb = i / 72;
        *reinterpret_cast<uint4*>(smem + pb * kTile + j * kRowC * kCell + k * 16) = make_uint4(0, 0, 0, 0);
    }
    // weight stream: tap t, k-step ks, co-tile w, 16-byte halves: wf8[(((t*2 + ks)*4 + w)*2 + half)*64 + lane]
    const uint4* ap = T.wf8 + (size_t)(w * 2) * 64;  // wave-uniform base + lane (SGPR-base addressing)
    v8i A0[2], A1[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        uint4 lo = ap[((ks * 4) * 2 + 0) * 64 + (unsigned)lane], hi = ap[((ks * 4) * 2 + 1) * 64 + (unsigned)lane];
        v8i v = {(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        A0[ks] = v;
    }
    ap += 2 * 4 * 2 * 64;

    // ---- stem (bf16 MFMA, exact 0/1 inputs) -> e4m3 activations
    {
        f32x16 acc[8];
        Scale sc;
        load_scale(sc, T.ones, T.stem_b, w, h);
        bf16x8 sa[2];
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) sa[kc] = __builtin_bit_cast(bf16x8, T.stem_wf[(kc * 4 + w) * 64 + lane]);
        int pos = pos0 + (r >> 3) < T.n ? pos0 + (r >> 3) : T.n - 1;  // every lane feeds ONE position in all units
        u64 own = T.own[pos], opp = T.opp[pos];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned n_own = nbhd(own, 8 * u + (r & 7)), n_opp = nbhd(opp, 8 * u + (r & 7));
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[0], stem_frag<0>(n_own, n_opp, h), (f32x16)(0.0f), 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[1], stem_frag<1>(n_own, n_opp, h), acc[u], 0, 0, 0);
        }
        epilogue(acc, bufX, false, sc, w, r, h);
    }
    __syncthreads();

#pragma unroll 1
    for (int blk = 0; blk < T.n_layers / 2; ++blk) {
        conv_layer<0>(bufX, bufM, false, T.dq8 + (size_t)(2 * blk) * kTC, T.bias + (size_t)(2 * blk) * kTC, A0, A1, ap, w, r, h, tacc);
        conv_layer<1>(bufM, bufX, true, T.dq8 + (size_t)(2 * blk + 1) * kTC, T.bias + (size_t)(2 * blk + 1) * kTC, A0, A1, ap, w,
                      r, h, tacc);
    }

    BZ_STAMP(tk1);
    // ---- heads: wave p serves position p (conv1x1 in fp8, FCs in fp32)
    if (pos0 + w < T.n) {
        const int p = w, pos = pos0 + w;
        float* S = reinterpret_cast<float*>(bufM + p * 1024);
        const float pb0 = T.pol_b[0], pb1 = T.pol_b[1], vb = T.val_b[0];
        const float d0 = T.head_dq8[0], d1 = T.head_dq8[1], d2 = T.head_dq8[2];
        v8i hw[2];  // both head-conv fragments in flight before the first MFMA
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 lo = T.head_wf8[(ks * 2 + 0) * 64 + lane], hi = T.head_wf8[(ks * 2 + 1) * 64 + lane];
            v8i a = {(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
            hw[ks] = a;
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            f32x16 acc = (f32x16)(0.0f);
            const int cell = 32 * nt + r;
            const int cb = cell_at(cell >> 3, cell & 7) + (((2 * h) ^ sw3(p, cell & 7)) << 4);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                v8i b = ld32(bufX + p * kTile, cb ^ (ks << 6), 0);
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(hw[ks], b, acc, 0, 0, 0, kUnit, 0, kUnit);
            }
            if (h == 0) {
                float a0 = acc[0] * d0 + pb0, a1 = acc[1] * d1 + pb1, a2 = acc[2] * d2 + vb;
                S[cell] = a0 > 0.0f ? a0 : 0.0f;
                S[64 + cell] = a1 > 0.0f ? a1 : 0.0f;
                S[128 + cell] = a2 > 0.0f ? a2 : 0.0f;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        // (the fma chains below keep their order; the unroll factors only decide how many weight loads are in flight)
        float acc = T.polfc_b[lane], part = 0.0f;
#pragma unroll 8
        for (int i = 0; i < 128; i += 4) {
            f32x4 s4 = *reinterpret_cast<const f32x4*>(S + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_fmaf(s4[j], T.polfc_wT[(i + j) * 65 + lane], acc);
        }
        part = S[lane] * T.polfc_wT[lane * 65 + 64] + S[lane + 64] * T.polfc_wT[(lane + 64) * 65 + 64];
        part = wave_sum(part);
        T.logits[(size_t)pos * 65 + lane] = acc;
        if (lane == 0) T.logits[(size_t)pos * 65 + 64] = part + T.polfc_b[64];
        float vh = 0.0f;
        if (lane < T.VH) {
            float a = T.v1_b[lane];
#pragma unroll 32
            for (int i = 0; i < 64; ++i) a = __builtin_fmaf(S[128 + i], T.v1_wT[i * T.VH + lane], a);
            vh = (a > 0.0f ? a : 0.0f) * T.v2_w[lane];
        }
        vh = wave_sum(vh);
        if (lane == 0) T.value[pos] = tanhf_spec(vh + T.v2_b[0]);
    }
#ifdef BZ_EXP_STAMPS
    unsigned long long tk2; BZ_STAMP(tk2);
    tr1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && blockIdx.x < 4096) {
        unsigned long long* d = g_dbg + blockIdx.x * 8;
        d[0] = tacc[0]; d[1] = tacc[1]; d[2] = tacc[2]; d[3] = tk1 - tk0; d[4] = tk2 - tk0; d[5] = tr1 - tr0; d[6] = tk0; d[7] = tr0;
    }
#endif
}
}  // namespace f8

// ------------------------------------------------------------------ host helpers
struct Carver {
    int64_t off = 0;
    int64_t take(int64_t bytes) { int64_t o = off; off += (bytes + 255) & ~int64_t(255); return o; }
};
struct NetOffsets {
    int64_t stem_w, stem_b, conv_w, conv_b, conv_wf, stem_wf, head_wf, conv_wf16, stem_wf16, conv_wf8, head_wf8, dq8, head_dq8, ones, pol_w, pol_b, polfc_wT, polfc_b, val_w, val_b, v1_wT, v1_b, v2_w,
        v2_b, act_a, act_b, total;
};
NetOffsets net_carve(int C, int NB, int VH, int mb) {
    NetOffsets o{};
    Carver k;
    int64_t L = 2 * NB, mbp = (mb + 7) & ~7;
    o.stem_w = k.take(18LL * C * 4); o.stem_b = k.take(C * 4LL);
    o.conv_w = k.take(L * 9 * C * C * 4); o.conv_b = k.take(L * C * 4);
    const bool mfma = C == 64 || C == 128 || C == 256;  // widths the fused bf16 MFMA kernel is built for
    const int64_t KC = C / 16, MT = C / 32;             // k-steps per tap, M-tiles (1-KB fragments: [kc][mt][64 lanes][8])
    o.conv_wf = k.take(mfma ? (L * 9 + 2) * KC * MT * 1024 : 0);  // + 2 taps: the prefetch runs up to two chunks past the end
    o.stem_wf = k.take(mfma ? 2 * MT * 1024 : 0); o.head_wf = k.take(mfma ? KC * 1024 : 0);
    o.conv_wf16 = k.take(C == kTC ? (L * 9 + 2) * (C / 32) * MT * 2 * 1024 : 0);  // (+ 2 taps: prefetch past the end)
    o.stem_wf16 = k.take(C == kTC ? MT * 2 * 1024 : 0);
    o.conv_wf8 = k.take(C == kTC ? (L * 9 + 1) * 2LL * 4 * 2 * 64 * 16 : 0); o.head_wf8 = k.take(2 * 2 * 64 * 16);
    o.dq8 = k.take((L + 1) * 128 * 4); o.head_dq8 = k.take(16); o.ones = k.take(128 * 4);
    o.pol_w = k.take(2LL * C * 4); o.pol_b = k.take(8); o.polfc_wT = k.take(128 * 65 * 4); o.polfc_b = k.take(65 * 4);
    o.val_w = k.take(C * 4LL); o.val_b = k.take(4); o.v1_wT = k.take(64LL * VH * 4); o.v1_b = k.take(VH * 4LL);
    o.v2_w = k.take(VH * 4LL); o.v2_b = k.take(4);
    o.act_a = k.take(mbp * 64 * C * 4); o.act_b = k.take(mbp * 64 * C * 4);
    o.total = k.off;
    return o;
}
bool shape_ok(int C, int NB, int VH, int mb) {
    return (C == 32 || C == 64 || C == 128 || C == 256) && NB >= 0 && NB <= 64 && VH >= 1 && VH <= 64 && mb >= 1;
}
uint16_t f2bf(float f) {
    u32 u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)(u >> 16);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
// OCP e4m3fn encode (RNE, saturating at 448)
uint8_t f2e4m3(float f) {
    uint8_t sign = f < 0.0f ? 0x80 : 0;
    float a = fabsf(f);
    if (!(a == a)) return 0x7F;
    if (a > 448.0f) a = 448.0f;
    if (a == 0.0f) return sign;
    int ex;
    (void)frexpf(a, &ex);
    int e = ex - 1 < -6 ? -6 : ex - 1;
    float q = nearbyintf(a / ldexpf(1.0f, e - 3)) * ldexpf(1.0f, e - 3);
    if (q > 448.0f) q = 448.0f;
    if (q == 0.0f) return sign;
    (void)frexpf(q, &ex);
    int e2 = ex - 1;
    if (e2 < -6) return sign | (uint8_t)nearbyintf(q * 512.0f);  // subnormal: mantissa * 2^-9
    int mant = (int)nearbyintf((q / ldexpf(1.0f, e2) - 1.0f) * 8.0f);
    return sign | (uint8_t)(((e2 + 7) << 3) | mant);
}
// largest power of two s with max|w| * s <= 448
float pow2_scale(float maxabs) {
    if (!(maxabs > 0.0f)) return 1.0f;
    int ex;
    (void)frexpf(448.0f / maxabs, &ex);
    return ldexpf(1.0f, ex - 1);
}
template <class T> T* at(void* base, int64_t off) { return reinterpret_cast<T*>(static_cast<char*>(base) + off); }
}  // namespace

// host repack of the flat torch-layout parameter vector into the kernels' layouts, one H2D copy
static std::atomic<uint64_t> g_param_epoch{0};

static int32_t upload_params(bz_net* n, const float* p, hipStream_t s) {
    n->epoch = ++g_param_epoch;
    const int C = n->C, NB = n->NB, VH = n->VH;
    NetOffsets o = net_carve(C, NB, VH, n->max_batch);
    void* ws = n->ws_base;
    int64_t param_bytes = o.act_a;
    std::vector<char> img((size_t)param_bytes, 0);
    auto F = [&](int64_t off) { return reinterpret_cast<float*>(img.data() + off); };
    int L = 2 * NB;
    const float* q = p;
    for (int co = 0; co < C; ++co)
        for (int ci = 0; ci < 2; ++ci)
            for (int t = 0; t < 9; ++t) F(o.stem_w)[(t * 2 + ci) * C + co] = q[(co * 2 + ci) * 9 + t];
    const bool mfma = C == 64 || C == 128 || C == 256;
    const int KC = C / 16, MT = C / 32;
    if (mfma) {  // stem as GEMM fragments: A[co][k], k = 2*tap + plane, zero for k >= 18
        uint16_t* sf = reinterpret_cast<uint16_t*>(img.data() + o.stem_wf);
        for (int kc = 0; kc < 2; ++kc)
            for (int mt = 0; mt < MT; ++mt)
                for (int ln = 0; ln < 64; ++ln)
                    for (int j = 0; j < 8; ++j) {
                        int co = 32 * mt + (ln & 31), k = 16 * kc + 8 * (ln >> 5) + j;
                        float v = k < 18 ? q[(co * 2 + (k & 1)) * 9 + (k >> 1)] : 0.0f;
                        sf[(((size_t)kc * MT + mt) * 64 + ln) * 8 + j] = f2bf(v);
                    }
    }
    if (C == kTC) {  // 16x16x32 path: A[co = 32wt + 16a + (lane & 15)][k = 8 (lane >> 4) + j]
        uint16_t* sf = reinterpret_cast<uint16_t*>(img.data() + o.stem_wf16);
        for (int wt = 0; wt < MT; ++wt)
            for (int a = 0; a < 2; ++a)
                for (int ln = 0; ln < 64; ++ln)
                    for (int j = 0; j < 8; ++j) {
                        int co = 32 * wt + 16 * a + (ln & 15), k = 8 * (ln >> 4) + j;
                        float v = k < 18 ? q[(co * 2 + (k & 1)) * 9 + (k >> 1)] : 0.0f;
                        sf[(((size_t)wt * 2 + a) * 64 + ln) * 8 + j] = f2bf(v);
                    }
    }
    q += (size_t)C * 18;
    for (int i = 0; i < C; ++i) F(o.stem_b)[i] = q[i];
    q += C;
    uint16_t* wf = reinterpret_cast<uint16_t*>(img.data() + o.conv_wf);
    for (int l = 0; l < L; ++l) {
        float* wl = F(o.conv_w) + (size_t)l * 9 * C * C;
        for (int co = 0; co < C; ++co)
            for (int ci = 0; ci < C; ++ci)
                for (int t = 0; t < 9; ++t) {
                    float v = q[((size_t)co * C + ci) * 9 + t];
                    wl[((size_t)t * C + ci) * C + co] = v;
                    if (mfma) {  // fragment-major: [l][t][kc][mt][lane = 32h + r][j]
                        int kc = ci >> 4, hh = (ci >> 3) & 1, j = ci & 7, mt = co >> 5, rr = co & 31;
                        size_t f = ((((size_t)l * 9 + t) * KC + kc) * MT + mt) * 64 + (hh * 32 + rr);
                        wf[f * 8 + j] = f2bf(v);
                    }
                    if (C == kTC) {  // 16x16x32 path: [l][t][kq][wt][a][lane = 16g + c][j]: co = 32wt + 16a + c, ci = 32kq + 8g + j
                        uint16_t* w16 = reinterpret_cast<uint16_t*>(img.data() + o.conv_wf16);
                        int kq = ci >> 5, g = (ci >> 3) & 3, j = ci & 7, wt = co >> 5, a = (co >> 4) & 1, c = co & 15;
                        size_t f = (((((size_t)l * 9 + t) * (C / 32) + kq) * MT + wt) * 2 + a) * 64 + (g * 16 + c);
                        w16[f * 8 + j] = f2bf(v);
                    }
                }
        q += (size_t)C * C * 9;
        for (int i = 0; i < C; ++i) F(o.conv_b)[(size_t)l * C + i] = q[i];
        q += C;
    }
    if (C == kTC) {  // fp8 fragments + per-output-channel dequant factors, from the repacked fp32 weights
        uint8_t* w8 = reinterpret_cast<uint8_t*>(img.data() + o.conv_wf8);
        for (int l = 0; l < L; ++l) {
            const float* wl = F(o.conv_w) + (size_t)l * 9 * C * C;  // [t][ci][co]
            for (int co = 0; co < C; ++co) {
                float mx = 0.0f;
                for (int t = 0; t < 9; ++t)
                    for (int ci = 0; ci < C; ++ci) mx = fmaxf(mx, fabsf(wl[((size_t)t * C + ci) * C + co]));
                const float sc = pow2_scale(mx);
                F(o.dq8)[(size_t)l * C + co] = 1.0f / (sc * 16.0f);
                for (int t = 0; t < 9; ++t)
                    for (int ci = 0; ci < C; ++ci) {
                        int ks = ci >> 6, hh = (ci >> 5) & 1, j = ci & 31, half = j >> 4, mt = co >> 5, rr = co & 31;
                        size_t f = ((((((size_t)l * 9 + t) * 2 + ks) * 4 + mt) * 2 + half) * 64 + (hh * 32 + rr)) * 16 + (j & 15);
                        w8[f] = f2e4m3(wl[((size_t)t * C + ci) * C + co] * sc);
                    }
            }
        }
        for (int i = 0; i < 128; ++i) F(o.ones)[i] = 1.0f;
    }
    for (int i = 0; i < 2 * C; ++i) F(o.pol_w)[i] = q[i];
    q += 2 * C;
    F(o.pol_b)[0] = q[0]; F(o.pol_b)[1] = q[1]; q += 2;
    for (int a = 0; a < 65; ++a)
        for (int i = 0; i < 128; ++i) F(o.polfc_wT)[i * 65 + a] = q[a * 128 + i];
    q += 65 * 128;
    for (int a = 0; a < 65; ++a) F(o.polfc_b)[a] = q[a];
    q += 65;
    for (int i = 0; i < C; ++i) F(o.val_w)[i] = q[i];
    q += C;
    F(o.val_b)[0] = q[0]; q += 1;
    for (int hh = 0; hh < VH; ++hh)
        for (int i = 0; i < 64; ++i) F(o.v1_wT)[i * VH + hh] = q[hh * 64 + i];
    q += (size_t)VH * 64;
    for (int i = 0; i < VH; ++i) F(o.v1_b)[i] = q[i];
    q += VH;
    for (int i = 0; i < VH; ++i) F(o.v2_w)[i] = q[i];
    q += VH;
    F(o.v2_b)[0] = q[0];
    if (mfma) {  // head conv1x1 fragments: rows 0,1 = policy channels, row 2 = value channel
        uint16_t* hf = reinterpret_cast<uint16_t*>(img.data() + o.head_wf);
        for (int kc = 0; kc < KC; ++kc)
            for (int ln = 0; ln < 64; ++ln)
                for (int j = 0; j < 8; ++j) {
                    int row = ln & 31, k = 16 * kc + 8 * (ln >> 5) + j;
                    float v = row < 2 ? F(o.pol_w)[row * C + k] : (row == 2 ? F(o.val_w)[k] : 0.0f);
                    hf[((size_t)kc * 64 + ln) * 8 + j] = f2bf(v);
                }
    }
    if (C == kTC) {
        uint8_t* h8 = reinterpret_cast<uint8_t*>(img.data() + o.head_wf8);
        float hs[3];
        for (int row = 0; row < 3; ++row) {
            float mx = 0.0f;
            for (int k = 0; k < C; ++k) mx = fmaxf(mx, fabsf(row < 2 ? F(o.pol_w)[row * C + k] : F(o.val_w)[k]));
            hs[row] = pow2_scale(mx);
            F(o.head_dq8)[row] = 1.0f / (hs[row] * 16.0f);
        }
        for (int ks = 0; ks < 2; ++ks)
            for (int ln = 0; ln < 64; ++ln)
                for (int j = 0; j < 32; ++j) {
                    int row = ln & 31, k = 64 * ks + 32 * (ln >> 5) + j;
                    float v = row < 2 ? F(o.pol_w)[row * C + k] * hs[row] : (row == 2 ? F(o.val_w)[k] * hs[2] : 0.0f);
                    h8[((size_t)(ks * 2 + (j >> 4)) * 64 + ln) * 16 + (j & 15)] = f2e4m3(v);
                }
    }
    hipError_t e1 = hipMemcpyAsync(ws, img.data(), (size_t)param_bytes, hipMemcpyHostToDevice, s);
    hipError_t e2 = e1 == hipSuccess ? hipStreamSynchronize(s) : e1;
    if (e2 != hipSuccess) return hip_fail(e2, "bz_net upload");
    return BZ_OK;
}

BZ_EXPORT int64_t bz_net_param_count(int32_t C, int32_t NB, int32_t VH) {
    int64_t n = (int64_t)C * 2 * 9 + C;
    n += (int64_t)NB * 2 * ((int64_t)C * C * 9 + C);
    n += 2LL * C + 2 + 65 * 128 + 65;
    n += (int64_t)C + 1 + (int64_t)VH * 64 + VH + VH + 1;
    return n;
}

BZ_EXPORT int64_t bz_net_workspace_bytes(int32_t C, int32_t NB, int32_t VH, int32_t max_batch) {
    if (!shape_ok(C, NB, VH, max_batch)) { set_error("bz_net_workspace_bytes: unsupported shape"); return -1; }
    return net_carve(C, NB, VH, max_batch).total;
}

BZ_EXPORT int32_t bz_net_create(int32_t C, int32_t NB, int32_t VH, int32_t max_batch, const float* p, void* ws,
                                int64_t bytes, void* stream, bz_net** out) {
    BZ_REQUIRE(shape_ok(C, NB, VH, max_batch) && p && ws && out, "bz_net_create: unsupported shape or null pointer");
    if (bz_device_count() <= 0) { set_error("bz_net_create: no HIP device (the net has no CPU path)"); return BZ_ENOGPU; }
    NetOffsets o = net_carve(C, NB, VH, max_batch);
    if (bytes < o.total) { set_error("bz_net_create: workspace too small"); return BZ_ENOMEM; }
    BZ_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "bz_net_create: workspace must be 256-byte aligned");
    bz_net* n = new (std::nothrow) bz_net();
    if (!n) { set_error("out of host memory"); return BZ_ENOMEM; }
    n->C = C; n->NB = NB; n->VH = VH; n->max_batch = max_batch;
    n->stem_w = at<float>(ws, o.stem_w); n->stem_b = at<float>(ws, o.stem_b);
    n->conv_w = at<float>(ws, o.conv_w); n->conv_b = at<float>(ws, o.conv_b);
    n->conv_wf = (C == 64 || C == 128 || C == 256) ? at<__bf16>(ws, o.conv_wf) : nullptr;
    n->stem_wf = at<__bf16>(ws, o.stem_wf); n->head_wf = at<__bf16>(ws, o.head_wf);
    n->conv_wf16 = C == kTC ? at<__bf16>(ws, o.conv_wf16) : nullptr; n->stem_wf16 = C == kTC ? at<__bf16>(ws, o.stem_wf16) : nullptr;
    n->conv_wf8 = C == kTC ? at<uint8_t>(ws, o.conv_wf8) : nullptr; n->head_wf8 = at<uint8_t>(ws, o.head_wf8);
    n->dq8 = at<float>(ws, o.dq8); n->head_dq8 = at<float>(ws, o.head_dq8); n->ones = at<float>(ws, o.ones);
    n->pol_w = at<float>(ws, o.pol_w); n->pol_b = at<float>(ws, o.pol_b);
    n->polfc_wT = at<float>(ws, o.polfc_wT); n->polfc_b = at<float>(ws, o.polfc_b);
    n->val_w = at<float>(ws, o.val_w); n->val_b = at<float>(ws, o.val_b);
    n->v1_wT = at<float>(ws, o.v1_wT); n->v1_b = at<float>(ws, o.v1_b);
    n->v2_w = at<float>(ws, o.v2_w); n->v2_b = at<float>(ws, o.v2_b);
    n->act_a = at<float>(ws, o.act_a); n->act_b = at<float>(ws, o.act_b);

    n->ws_base = ws;
    n->f32_used = false;
    if (hipEventCreateWithFlags(&n->f32_done, hipEventDisableTiming) != hipSuccess) {
        delete n; set_error("bz_net_create: hipEventCreate failed"); return BZ_EHIP;
    }
    int32_t urc = upload_params(n, p, (hipStream_t)stream);
    if (urc != BZ_OK) { delete n; return urc; }
    {
        hipError_t e3 = hipSuccess;
#define BZ_TOWER_LDS(GEOM) hipFuncSetAttribute(reinterpret_cast<const void*>(k_tower_bf16<GEOM>), hipFuncAttributeMaxDynamicSharedMemorySize, GEOM::LDS)
        if (C == 64) { e3 = BZ_TOWER_LDS(Tw<64>); if (e3 == hipSuccess) e3 = BZ_TOWER_LDS(TwS64); }
        if (C == 128) { e3 = BZ_TOWER_LDS(TwM16); if (e3 == hipSuccess) e3 = BZ_TOWER_LDS(TwS128); }
        if (C == 256) { e3 = BZ_TOWER_LDS(Tw<256>); if (e3 == hipSuccess) e3 = BZ_TOWER_LDS(TwS256); }  // TwS256: 73 KB
#undef BZ_TOWER_LDS
        if (e3 != hipSuccess) { delete n; return hip_fail(e3, "hipFuncSetAttribute(k_tower_bf16)"); }
        // the f32 parity kernels stage a whole position in LDS: above 64 KB at C = 256
        if (C == 256) {
            e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_heads<float>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)((64 * (C + 1) + 128 + 64 + 64) * sizeof(float)));
            if (e3 == hipSuccess)
                e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_f32), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(64 * C * sizeof(float)));
            if (e3 != hipSuccess) { delete n; return hip_fail(e3, "hipFuncSetAttribute(f32 parity kernels)"); }
        }
    }
    if (C == kTC) {
        hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(f8::k_tower_fp8), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 f8::kLds);
        if (e3 != hipSuccess) { delete n; return hip_fail(e3, "hipFuncSetAttribute(k_tower_fp8)"); }
    }
    *out = n;
    return BZ_OK;
}

BZ_EXPORT int32_t bz_net_destroy(bz_net* net) {
    if (net) (void)hipEventDestroy(net->f32_done);
    delete net;
    return BZ_OK;
}

BZ_EXPORT int32_t bz_net_update(bz_net* net, const float* params_host, void* stream) {
    BZ_REQUIRE(net && params_host, "bz_net_update: null pointer");
    // searches in flight on ANY stream must not see half-replaced weights: drain the device first
    BZ_HIP(hipDeviceSynchronize());
    return upload_params(net, params_host, (hipStream_t)stream);
}

#ifdef BZ_EXP_STAMPS
BZ_EXPORT int32_t bz_debug_read(void* dst, int64_t bytes) {
    BZ_HIP(hipDeviceSynchronize());
    BZ_HIP(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dbg), (size_t)bytes));
    return BZ_OK;
}
#endif

static int32_t forward_f32(bz_net* n, const uint64_t* own, const uint64_t* opp, int32_t cnt, const u32* n_dev,
                           float* logits, float* value, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int C = n->C;
    if (n->f32_used) BZ_HIP(hipStreamWaitEvent(s, n->f32_done, 0));
    hipLaunchKernelGGL(k_stem<float>, dim3(cnt), dim3(C), 0, s, own, opp, cnt, n_dev, C, n->stem_w, n->stem_b, n->act_a);
    BZ_LAUNCH_CHECK("k_stem<float>");
    size_t lds = (size_t)64 * C * sizeof(float);
    for (int blk = 0; blk < n->NB; ++blk) {
        const float* w1 = n->conv_w + (size_t)(2 * blk) * 9 * C * C;
        const float* w2 = n->conv_w + (size_t)(2 * blk + 1) * 9 * C * C;
        hipLaunchKernelGGL(k_conv_f32, dim3(cnt), dim3(256), lds, s, n->act_a, w1, n->conv_b + (size_t)(2 * blk) * C,
                           (const float*)nullptr, n->act_b, cnt, n_dev, C);
        BZ_LAUNCH_CHECK("k_conv_f32");
        hipLaunchKernelGGL(k_conv_f32, dim3(cnt), dim3(256), lds, s, n->act_b, w2,
                           n->conv_b + (size_t)(2 * blk + 1) * C, (const float*)n->act_a, n->act_a, cnt, n_dev, C);
        BZ_LAUNCH_CHECK("k_conv_f32");
    }
    size_t hl = (64 * (n->C + 1) + 128 + 64 + 64) * sizeof(float);
    hipLaunchKernelGGL(k_heads<float>, dim3(cnt), dim3(192), hl, s, n->act_a, cnt, n_dev, n->C, n->VH, n->pol_w, n->pol_b,
                       n->polfc_wT, n->polfc_b, n->val_w, n->val_b, n->v1_wT, n->v1_b, n->v2_w, n->v2_b, logits, value);
    BZ_LAUNCH_CHECK("k_heads<float>");
    BZ_HIP(hipEventRecord(n->f32_done, s));
    n->f32_used = true;
    return BZ_OK;
}

static int32_t forward_bf16(bz_net* n, const uint64_t* own, const uint64_t* opp, int32_t cnt, const u32* n_dev,
                            float* logits, float* value, void* stream, bool fp8) {
    hipStream_t s = (hipStream_t)stream;
    TowerArgs T;
    T.own = own; T.opp = opp; T.n_dev = n_dev; T.n = cnt; T.n_layers = 2 * n->NB; T.VH = n->VH;
    T.wf = reinterpret_cast<const uint4*>(n->conv_wf); T.bias = n->conv_b;
    T.stem_wf = reinterpret_cast<const uint4*>(n->stem_wf); T.stem_b = n->stem_b;
    T.wf16 = reinterpret_cast<const uint4*>(n->conv_wf16); T.stem_wf16 = reinterpret_cast<const uint4*>(n->stem_wf16);
    T.head_wf = reinterpret_cast<const uint4*>(n->head_wf); T.pol_b = n->pol_b; T.val_b = n->val_b;
    T.wf8 = reinterpret_cast<const uint4*>(n->conv_wf8); T.head_wf8 = reinterpret_cast<const uint4*>(n->head_wf8);
    T.dq8 = n->dq8; T.head_dq8 = n->head_dq8; T.ones = n->ones;
    T.polfc_wT = n->polfc_wT; T.polfc_b = n->polfc_b; T.v1_wT = n->v1_wT; T.v1_b = n->v1_b; T.v2_w = n->v2_w;
    T.v2_b = n->v2_b; T.logits = logits; T.value = value;
    {
        ProfScope ps(BZ_PROF_TOWER, stream);
        if (fp8) hipLaunchKernelGGL(f8::k_tower_fp8, dim3((cnt + 3) / 4), dim3(256), f8::kLds, s, T);
        else {
            // up to kSmallBatch positions: one (two at C = 64) per workgroup, i.e. per CU -- the latency shape
            const bool small = cnt <= kSmallBatch;
#define BZ_TOWER_LAUNCH(GEOM) hipLaunchKernelGGL(k_tower_bf16<GEOM>, dim3((cnt + GEOM::P - 1) / GEOM::P), dim3(256), GEOM::LDS, s, T)
            if (n->C == 64) { if (small) BZ_TOWER_LAUNCH(TwS64); else BZ_TOWER_LAUNCH(Tw<64>); }
            else if (n->C == 256) { if (small) BZ_TOWER_LAUNCH(TwS256); else BZ_TOWER_LAUNCH(Tw<256>); }
            else { if (small) BZ_TOWER_LAUNCH(TwS128); else BZ_TOWER_LAUNCH(TwM16); }
#undef BZ_TOWER_LAUNCH
        }
    }
    BZ_LAUNCH_CHECK("k_tower_bf16");
    return BZ_OK;
}

uint64_t bz_net_epoch(const bz_net* n) { return n ? n->epoch : 0; }

int32_t bz_net_forward_dev(bz_net* n, int bf16 /* 0 f32, 1 bf16, 2 fp8 */, const uint64_t* own, const uint64_t* opp, int32_t max_n,
                           const uint32_t* n_dev, float* logits, float* value, void* stream) {
    BZ_REQUIRE(n && own && opp && logits && value, "bz_net_forward: null pointer");
    BZ_REQUIRE(bf16 != 1 || n->C == 64 || n->C == 128 || n->C == 256,
               "bz_net_forward_bf16: the MFMA tower is built for 64, 128 or 256 channels");
    BZ_REQUIRE(bf16 != 2 || n->C == kTC, "bz_net_forward_fp8: the fp8 tower is built for C == 128");
    BZ_REQUIRE(max_n >= 0 && max_n <= n->max_batch, "bz_net_forward: batch exceeds max_batch");
    if (max_n == 0) return BZ_OK;
    return bf16 ? forward_bf16(n, own, opp, max_n, n_dev, logits, value, stream, bf16 == 2)
                : forward_f32(n, own, opp, max_n, n_dev, logits, value, stream);
}

BZ_EXPORT int32_t bz_net_forward_f32(bz_net* n, const uint64_t* own, const uint64_t* opp, int32_t cnt, float* logits,
                                     float* value, void* stream) {
    return bz_net_forward_dev(n, 0, own, opp, cnt, nullptr, logits, value, stream);
}
BZ_EXPORT int32_t bz_net_forward_bf16(bz_net* n, const uint64_t* own, const uint64_t* opp, int32_t cnt, float* logits,
                                      float* value, void* stream) {
    return bz_net_forward_dev(n, 1, own, opp, cnt, nullptr, logits, value, stream);
}
BZ_EXPORT int32_t bz_net_forward_fp8(bz_net* n, const uint64_t* own, const uint64_t* opp, int32_t cnt, float* logits,
                                     float* value, void* stream) {
    return bz_net_forward_dev(n, 2, own, opp, cnt, nullptr, logits, value, stream);
}
