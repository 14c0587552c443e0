// bz_train_ends.hip -- the two ENDS of the training step around the tower kernels of bz_train.hip (SURVEY.md 8(f) row 4;
// loop shape of src/tic_tac_toe/SL/train.py:85-136 -- forward, loss, backward, optimiser step):
//
//   bitboards --k_train_stem--> act[0] --(bz_train.hip: forward)--> act[L] --k_train_heads--> losses, g[L]
//   g[L] --(bz_train.hip: backward, weight gradients)--> g[0] --k_train_stem_wgrad--> d stem
//   k_train_heads_wgrad: the two FC weight gradients whose reduction axis is the batch;  k_train_finish: every partial
//   sum of the step -> the gradient tensors in torch's parameter layouts (one launch);  k_train_adam: the update.
//
// With these a whole step is 10 launches instead of ~120 short torch kernels (casts, gathers, 1x1 "convolutions" as
// skinny GEMMs, soft-max, reductions, a multi-tensor optimiser), which at batch 1024 cost 2.5x the three tower kernels.
// Everything here is HBM- / latency-bound small work: coalesced 16-byte accesses, LDS for the transposes, fp32
// arithmetic on the bf16 activations the tower kernels store.  No MFMA on purpose (the largest product is 65 x 128).
//
// The net (betazero_amd/net.py, SURVEY 8(d) "net"): stem conv3x3 2 -> C, ReLU | tower | policy: conv1x1 C -> 2, ReLU,
// FC 128 -> 65 | value: conv1x1 C -> 1, ReLU, FC 64 -> VH, ReLU, FC VH -> 1, tanh.  Loss = mean CE(pi, softmax) + mean
// (v - z)^2 (betazero_amd/train.py).
#include "bz_common.h"

using namespace bz;

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned pack2(float a, float b) {  // two floats -> two bf16 (round to nearest even), a in the low half
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// bit 9 * plane + tap of the result: plane `plane` (0 = own, 1 = opp) holds a stone at the tap's neighbour of `cell`
// (tap = 3 ky + kx, neighbour (y + ky - 1, x + kx - 1); off-board neighbours read as 0 = conv2d's zero padding)
__device__ __forceinline__ unsigned nbhd_bits(unsigned long long own, unsigned long long opp, int cell) {
    const int cy = cell >> 3, cx = cell & 7;
    unsigned m = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int y = cy + t / 3 - 1, x = cx + t % 3 - 1;
        const bool ok = (unsigned)y < 8u && (unsigned)x < 8u;
        const int s = (8 * y + x) & 63;
        m |= (ok ? (unsigned)((own >> s) & 1ull) : 0u) << t;
        m |= (ok ? (unsigned)((opp >> s) & 1ull) : 0u) << (9 + t);
    }
    return m;
}

// the batch of a step: a descriptor in DEVICE memory (bz_train_batch, bz_abi.h) read at launch time, so that a captured
// graph keeps working when the data set's tensors are replaced (the host rewrites 48 bytes).  Row of batch position p:
// idx[p] clamped into the data set (an index out of range must not become a fault), or p itself without an index.
// A clamped index is an ERROR of the caller, not a feature: k_train_heads counts the batch positions whose index was out
// of range (batch_row_bad) into the step's error word, losses[3] (bz_train_finish), which the host checks when it reads
// the losses -- the step trained on row 0 / n_rows - 1 in that position's place.
__device__ __forceinline__ long long batch_row_raw(const bz_train_batch& B, int p) { return B.idx ? B.idx[p] : (long long)p; }
__device__ __forceinline__ long long batch_row(const bz_train_batch& B, int p) {
    long long r = batch_row_raw(B, p);
    r = r < 0 ? 0 : r;
    return r < B.n_rows ? r : (B.n_rows > 0 ? B.n_rows - 1 : 0);
}
__device__ __forceinline__ bool batch_row_bad(const bz_train_batch& B, int p) {
    const long long r = batch_row_raw(B, p);
    return r < 0 || r >= B.n_rows;
}

// ---------------------------------------------------------------------------------------------------------------------
// stem forward: act0[pos][cell][c] = relu(b[c] + sum_k nbhd_k * w[c][k]), k = 9 plane + tap (torch's [C][2][3][3]).
// A workgroup takes 4 positions; a thread owns 8 channels (its 8 x 18 weights in registers) and walks the rows
// (position, cell), so that the lanes of a row write one contiguous 16-byte piece each.
template <int C>
__global__ __launch_bounds__(256) void k_train_stem(const bz_train_batch* __restrict__ batch, int n, const float* __restrict__ w,
                                                    const float* __restrict__ b, __bf16* __restrict__ act0) {
    constexpr int CG = C / 8, RPP = 256 / CG;
    __shared__ unsigned nb[256];
    const int tid = threadIdx.x, pos0 = blockIdx.x * 4;
    {
        const bz_train_batch B = *batch;
        const int p = pos0 + (tid >> 6);
        unsigned m = 0u;
        if (p < n) { const long long row = batch_row(B, p); m = nbhd_bits(B.own[row], B.opp[row], tid & 63); }
        nb[tid] = m;
    }
    const int cg = tid % CG, rs = tid / CG;
    float wr[8][18], br[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        br[j] = b[8 * cg + j];
#pragma unroll
        for (int k = 0; k < 18; ++k) wr[j][k] = w[(8 * cg + j) * 18 + k];
    }
    __syncthreads();
    for (int it = 0; it < CG; ++it) {
        const int row = rs + it * RPP, p = pos0 + (row >> 6);
        if (p >= n) break;
        const unsigned m = nb[row];
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = br[j];
#pragma unroll
        for (int k = 0; k < 18; ++k) {
            const float f = (float)((m >> k) & 1u);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(f, wr[j][k], acc[j]);
        }
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pack2(fmaxf(acc[2 * j], 0.0f), fmaxf(acc[2 * j + 1], 0.0f));
        *reinterpret_cast<u32x4*>(act0 + ((size_t)p * 64 + (row & 63)) * C + 8 * cg) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// stem weight gradient: d w[c][k] = sum over rows of nbhd_k * g0[row][c] * (act0[row][c] > 0), d b[c] the same with
// nbhd = 1 (g0 = d loss / d act[0], what the tower's backward leaves in g[0]).  A wave takes one position at a time; a
// lane owns a channel pair (32-bit loads) and -- at C = 64 -- one of two cells per pass; 2 x 19 accumulators per lane,
// folded across lanes / waves through LDS at the end.  partial[block][C * 18 | C]: weights in torch's layout, then biases.
template <int C>
__global__ __launch_bounds__(256) void k_train_stem_wgrad(const bz_train_batch* __restrict__ batch, const unsigned* __restrict__ act0,
                                                          const unsigned* __restrict__ g0, int n, float* __restrict__ partial) {
    constexpr int PAIRS = C / 2, CPW = 64 / PAIRS;   // lanes per cell; cells per wave pass (2 at C = 64, 1 at C = 128)
    __shared__ unsigned nb[256];
    __shared__ float red[256 * 38];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, pair = lane % PAIRS, sub = lane / PAIRS;
    const bz_train_batch B = *batch;
    float acc[2][19];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int k = 0; k < 19; ++k) acc[s][k] = 0.0f;
    // a wave takes HALF a position (32 cells) at a time: twice the waves of "one position per wave" -- two per SIMD at batch
    // 1024 -- for a loop that is a chain of global loads and ~80 VALU instructions per cell (29 -> 21 us at 128 channels)
    for (int grp = blockIdx.x; grp * 2 < n; grp += gridDim.x) {
        const int item = grp * 4 + wv, p = item >> 1, c0 = 32 * (item & 1);
        __syncthreads();
        {
            unsigned m = 0u;
            if (p < n && lane < 32) { const long long row = batch_row(B, p); m = nbhd_bits(B.own[row], B.opp[row], c0 + lane); }
            nb[tid] = m;
        }
        __syncthreads();
        if (p < n) {
#pragma unroll 4
            for (int it = 0; it < 32 / CPW; ++it) {
                const int cl = it * CPW + sub, cell = c0 + cl;
                const size_t at = ((size_t)p * 64 + cell) * PAIRS + pair;
                const unsigned a = act0[at], g = g0[at];
                const float glo = (a & 0x7fffu) ? bf_lo(g) : 0.0f, ghi = (a & 0x7fff0000u) ? bf_hi(g) : 0.0f;   // act0 >= 0: "> 0" = "not (+-)0"
                const unsigned m = nb[64 * wv + cl];
#pragma unroll
                for (int k = 0; k < 18; ++k) {
                    const float f = (float)((m >> k) & 1u);
                    acc[0][k] = fmaf(f, glo, acc[0][k]);
                    acc[1][k] = fmaf(f, ghi, acc[1][k]);
                }
                acc[0][18] += glo;
                acc[1][18] += ghi;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int k = 0; k < 19; ++k) red[tid * 38 + s * 19 + k] = acc[s][k];
    __syncthreads();
    float* out = partial + (size_t)blockIdx.x * (C * 19);
    for (int o = tid; o < C * 19; o += 256) {
        const int c = o / 19, k = o % 19;
        float s = 0.0f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4)
#pragma unroll
            for (int q = 0; q < CPW; ++q) s += red[(64 * w4 + (c >> 1) + PAIRS * q) * 38 + (c & 1) * 19 + k];
        out[k < 18 ? c * 18 + k : C * 18 + c] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// heads + losses, forward AND backward, one wave per position (4 positions per workgroup pass):
//   h[cell][j]  = relu(sum_c x[cell][c] hw[j][c] + hb[j])         j = 0, 1: policy planes, j = 2: value plane
//   logit[a]    = sum_i Wp[a][i] h_flat[i] + bp[a]                 i = 64 j + cell (torch's .flatten(1) of [2, 8, 8])
//   v           = tanh(sum_h w2[h] relu(sum_cell W1[h][cell] h[cell][2] + b1[h]) + b2)
//   loss        = mean_pos( -sum_a pi[a] log softmax(logit)[a] ) + mean_pos( (v - z)^2 )
// and back to g_top = d loss / d (pre-activation of act[L]) = (d loss / d x) * (x > 0), bf16, what the tower's backward
// takes.  Parameter gradients whose reduction runs inside a position (the two 1x1 convolutions) and all bias / loss sums
// accumulate in registers and leave as one partial vector per workgroup; the two FC weight gradients reduce over the
// batch and are left to k_train_heads_wgrad (this kernel writes their operands h, d logit, d v1 per position).
struct HeadArgs {
    const __bf16* x;          // act[L]  [n][64][C]
    const bz_train_batch* batch;   // pi [rows][65], z [rows] and the batch's row indices (device memory)
    int n, VH;
    float inv_n;
    const float *pol_w, *pol_b, *polfc_w, *polfc_b, *val_w, *val_b, *v1_w, *v1_b, *v2_w, *v2_b;
    __bf16* g_top;            // [n][64][C]
    float *hv, *dl, *dv1;     // [n][192], [n][65], [n][64]
    float* partial;           // [gridDim.x][3 C + 201]
};
// the partial vector: [0, 3C) d hw (pol.weight [2][C], then val.weight [C]) | 3: d hb | 65: d polfc.bias | 64: d v1.bias |
// 64: d v2.weight | 1: d v2.bias | 3: loss, CE, MSE | 1: batch positions whose row index was out of range (the error word)
template <int C> struct Hd {
    static constexpr int XS = C + 8;                         // bf16 per LDS row of x (16 bytes of padding: conflict-free 16-byte reads down a column)
    static constexpr int WT = 0, V1T = 128 * 65, HWS = V1T + 64 * 65, SHARED = HWS + 3 * C;   // floats
    static constexpr int W_X = 0, W_HV = 64 * XS / 2, W_DP = W_HV + 192, W_DL = W_DP + 256, W_DV1 = W_DL + 68, W_FLOATS = W_DV1 + 64;
    static constexpr int LDS = (SHARED + 4 * W_FLOATS) * 4;
    static constexpr int O_HB = 3 * C, O_PFB = O_HB + 3, O_V1B = O_PFB + 65, O_V2W = O_V1B + 64, O_V2B = O_V2W + 64, O_LOSS = O_V2B + 1,
                         NP = O_LOSS + 4;
};

template <int C>
__global__ __launch_bounds__(256) void k_train_heads(HeadArgs A) {
    typedef Hd<C> H;
    constexpr int ZC = C / 8, CPL = C / 64;
    extern __shared__ float lds[];
    float* Wt = lds + H::WT;      // [i][a], 65 floats per row: polfc.weight transposed
    float* V1t = lds + H::V1T;    // [cell][h], 65 per row: v1.weight transposed (rows h >= VH zero)
    float* hwS = lds + H::HWS;    // [3][C]
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    float* mine = lds + H::SHARED + wv * H::W_FLOATS;
    unsigned short* xs = reinterpret_cast<unsigned short*>(mine + H::W_X);   // [64 cells][XS] bf16
    float* hvS = mine + H::W_HV;   // [192]: h_flat
    float* dpS = mine + H::W_DP;   // [64 cells][4]: d (pre-activation of h[cell][j])
    float* dlS = mine + H::W_DL;   // [65]: d logit
    float* dv1S = mine + H::W_DV1; // [64]: d (pre-activation of v1)

    // a position's x tile travels HBM -> registers -> LDS, 16 bytes per lane and step (consecutive lanes consecutive
    // addresses), all ZC loads in flight at once and issued a whole position ahead: the first one before the weight tables
    // are staged, the next one while this one is computed (four loads at a time, issued when needed, cost four memory
    // latencies per position)
    u32x4 xr[ZC];
    auto fetch_x = [&](int pos) {
        const u32x4* src = reinterpret_cast<const u32x4*>(A.x) + (size_t)pos * 64 * ZC;
#pragma unroll
        for (int k = 0; k < ZC; ++k) xr[k] = src[lane + 64 * k];
    };
    if ((int)blockIdx.x * 4 < A.n) fetch_x(blockIdx.x * 4 + wv);
    // the weight tables, eight loads in flight per thread (with the prefetched x tile and targets: 32 -> 28 us at 128 channels,
    // 19.5 -> 17.7 at 64)
    for (int i0 = 0; i0 < 65 * 128; i0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + 256 * u + tid; v[u] = i < 65 * 128 ? A.polfc_w[i] : 0.0f; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + 256 * u + tid; if (i < 65 * 128) Wt[(i & 127) * 65 + (i >> 7)] = v[u]; }
    }
    for (int i0 = 0; i0 < 64 * 64; i0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + 256 * u + tid; v[u] = (i >> 6) < A.VH ? A.v1_w[i] : 0.0f; }   // (i = 64 h + cell)
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + 256 * u + tid; V1t[(i & 63) * 65 + (i >> 6)] = v[u]; }
    }
    for (int i = tid; i < 3 * C; i += 256) hwS[i] = i < 2 * C ? A.pol_w[i] : A.val_w[i - 2 * C];
    const bz_train_batch B = *A.batch;
    const float hb[3] = {A.pol_b[0], A.pol_b[1], A.val_b[0]};
    const float pfb = A.polfc_b[lane], pfb64 = A.polfc_b[64], v2b = A.v2_b[0];
    const float v1b = lane < A.VH ? A.v1_b[lane] : 0.0f, v2w = lane < A.VH ? A.v2_w[lane] : 0.0f;

    float acc_hw[3][CPL], acc_hb[3] = {0.0f, 0.0f, 0.0f}, acc_pfb = 0.0f, acc_pfb64 = 0.0f, acc_v1b = 0.0f, acc_v2w = 0.0f, acc_v2b = 0.0f,
          acc_ce = 0.0f, acc_mse = 0.0f, acc_bad = 0.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int q = 0; q < CPL; ++q) acc_hw[j][q] = 0.0f;

    for (int grp = blockIdx.x; grp * 4 < A.n; grp += gridDim.x) {   // (n is a multiple of 4: every wave of a pass has a position)
        const int pos = grp * 4 + wv;
        const long long row = batch_row(B, pos);
        acc_bad += batch_row_bad(B, pos) ? 1.0f : 0.0f;
        // (the targets are needed in steps B and C: asked for now, they arrive under step A)
        const float pa = B.pi[(size_t)row * 65 + lane], p64 = B.pi[(size_t)row * 65 + 64], zf = (float)B.z[row];
        __syncthreads();   // the previous pass's copy-out has read xs; (first pass: the weight tables are in place)
#pragma unroll
        for (int k = 0; k < ZC; ++k) { const int i = lane + 64 * k; *reinterpret_cast<u32x4*>(xs + (i / ZC) * H::XS + (i % ZC) * 8) = xr[k]; }
        if ((grp + (int)gridDim.x) * 4 < A.n) fetch_x((grp + gridDim.x) * 4 + wv);
        __syncthreads();
        // ---- A: the three 1x1 convolutions, lane = cell
        float hvv[3];
        {
            float d0 = hb[0], d1 = hb[1], d2 = hb[2];
#pragma unroll 2
            for (int k = 0; k < ZC; ++k) {
                const u32x4 xv = *reinterpret_cast<const u32x4*>(xs + lane * H::XS + 8 * k);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = bf_lo(xv[e]), hi = bf_hi(xv[e]);
                    const int c = 8 * k + 2 * e;
                    d0 = fmaf(lo, hwS[c], d0);         d0 = fmaf(hi, hwS[c + 1], d0);
                    d1 = fmaf(lo, hwS[C + c], d1);     d1 = fmaf(hi, hwS[C + c + 1], d1);
                    d2 = fmaf(lo, hwS[2 * C + c], d2); d2 = fmaf(hi, hwS[2 * C + c + 1], d2);
                }
            }
            hvv[0] = fmaxf(d0, 0.0f); hvv[1] = fmaxf(d1, 0.0f); hvv[2] = fmaxf(d2, 0.0f);
#pragma unroll
            for (int j = 0; j < 3; ++j) { hvS[64 * j + lane] = hvv[j]; A.hv[(size_t)pos * 192 + 64 * j + lane] = hvv[j]; }
        }
        __syncthreads();
        // ---- B: policy FC, lane = action (action 64 = pass: all lanes together), soft-max, CE and d logit
        {
            float s = pfb, t64 = 0.0f;
#pragma unroll 8
            for (int i = 0; i < 128; ++i) s = fmaf(Wt[i * 65 + lane], hvS[i], s);
            t64 = fmaf(Wt[lane * 65 + 64], hvS[lane], fmaf(Wt[(lane + 64) * 65 + 64], hvS[lane + 64], 0.0f));
            const float l64 = wave_sum(t64) + pfb64;
            const float m = fmaxf(wave_max(s), l64);
            const float e = expf(s - m), e64 = expf(l64 - m);
            const float sum = wave_sum(e) + e64, lse = m + logf(sum);
            const float spi = wave_sum(pa) + p64;
            const float ce = -(wave_sum(pa * (s - lse)) + p64 * (l64 - lse));
            const float da = (e / sum * spi - pa) * A.inv_n, d64 = (e64 / sum * spi - p64) * A.inv_n;
            dlS[lane] = da;
            A.dl[(size_t)pos * 65 + lane] = da;
            if (lane == 0) { dlS[64] = d64; A.dl[(size_t)pos * 65 + 64] = d64; }
            acc_pfb += da; acc_pfb64 += d64; acc_ce += ce * A.inv_n;
        }
        // ---- C: value head, lane = hidden unit
        {
            float t = v1b;
#pragma unroll 8
            for (int cell = 0; cell < 64; ++cell) t = fmaf(V1t[cell * 65 + lane], hvS[128 + cell], t);
            const float v1h = fmaxf(t, 0.0f);                       // (lanes >= VH: weights and bias are zero -> 0)
            const float v = tanhf(wave_sum(v2w * v1h) + v2b);
            const float diff = v - zf;
            const float dpre2 = 2.0f * diff * A.inv_n * (1.0f - v * v);
            const float dv1h = t > 0.0f ? dpre2 * v2w : 0.0f;
            dv1S[lane] = dv1h;
            A.dv1[(size_t)pos * 64 + lane] = dv1h;
            acc_v1b += dv1h; acc_v2w = fmaf(dpre2, v1h, acc_v2w); acc_v2b += dpre2; acc_mse = fmaf(diff * diff, A.inv_n, acc_mse);
        }
        __syncthreads();
        // ---- D: back through the two FCs to the three planes, lane = cell; through their ReLUs
        float dp[3];
        {
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
#pragma unroll 5
            for (int a = 0; a < 65; ++a) { const float d = dlS[a]; a0 = fmaf(d, Wt[lane * 65 + a], a0); a1 = fmaf(d, Wt[(64 + lane) * 65 + a], a1); }
#pragma unroll 8
            for (int h = 0; h < 64; ++h) a2 = fmaf(dv1S[h], V1t[lane * 65 + h], a2);
            dp[0] = hvv[0] > 0.0f ? a0 : 0.0f; dp[1] = hvv[1] > 0.0f ? a1 : 0.0f; dp[2] = hvv[2] > 0.0f ? a2 : 0.0f;
            *reinterpret_cast<f32x4*>(dpS + 4 * lane) = (f32x4){dp[0], dp[1], dp[2], 0.0f};
#pragma unroll
            for (int j = 0; j < 3; ++j) acc_hb[j] += dp[j];
        }
        __syncthreads();
        // ---- E: d hw[j][c] += sum_cell dp[cell][j] x[cell][c], lane = channel (c = lane + 64 q)
#pragma unroll 4
        for (int cell = 0; cell < 64; ++cell) {
            const f32x4 d = *reinterpret_cast<const f32x4*>(dpS + 4 * cell);
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const float xv = __uint_as_float((unsigned)xs[cell * H::XS + lane + 64 * q] << 16);
                acc_hw[0][q] = fmaf(d[0], xv, acc_hw[0][q]);
                acc_hw[1][q] = fmaf(d[1], xv, acc_hw[1][q]);
                acc_hw[2][q] = fmaf(d[2], xv, acc_hw[2][q]);
            }
        }
        __syncthreads();
        // ---- F: d x[cell][c] = sum_j dp[cell][j] hw[j][c], times (x > 0), in place over x, lane = cell
#pragma unroll 2
        for (int k = 0; k < ZC; ++k) {
            u32x4* at = reinterpret_cast<u32x4*>(xs + lane * H::XS + 8 * k);
            const u32x4 xv = *at;
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 8 * k + 2 * e;
                const float glo = fmaf(dp[0], hwS[c], fmaf(dp[1], hwS[C + c], dp[2] * hwS[2 * C + c]));
                const float ghi = fmaf(dp[0], hwS[c + 1], fmaf(dp[1], hwS[C + c + 1], dp[2] * hwS[2 * C + c + 1]));
                // x is a stored ReLU output: bf16 >= 0, so "> 0" is "not the bit pattern of +0" (and of -0, which a kernel may have stored)
                o[e] = pack2((xv[e] & 0x7fffu) ? glo : 0.0f, (xv[e] & 0x7fff0000u) ? ghi : 0.0f);
            }
            *at = o;
        }
        __syncthreads();
        {
            u32x4* dst = reinterpret_cast<u32x4*>(A.g_top) + (size_t)pos * 64 * ZC;
#pragma unroll 4
            for (int i = lane; i < 64 * ZC; i += 64) dst[i] = *reinterpret_cast<const u32x4*>(xs + (i / ZC) * H::XS + (i % ZC) * 8);
        }
    }
    // ---- the workgroup's partial vector (the weight tables are dead: their LDS holds the four waves' vectors)
    __syncthreads();
    float* red = lds + wv * H::NP;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int q = 0; q < CPL; ++q) red[j * C + lane + 64 * q] = acc_hw[j][q];
        const float s = wave_sum(acc_hb[j]);
        if (lane == 0) red[H::O_HB + j] = s;
    }
    red[H::O_PFB + lane] = acc_pfb;
    red[H::O_V1B + lane] = acc_v1b;
    red[H::O_V2W + lane] = acc_v2w;
    if (lane == 0) {
        red[H::O_PFB + 64] = acc_pfb64;
        red[H::O_V2B] = acc_v2b;
        red[H::O_LOSS] = acc_ce + acc_mse; red[H::O_LOSS + 1] = acc_ce; red[H::O_LOSS + 2] = acc_mse;
        red[H::O_LOSS + 3] = acc_bad;   // (wave-uniform: one position per wave and pass)
    }
    __syncthreads();
    for (int o = tid; o < H::NP; o += 256)
        A.partial[(size_t)blockIdx.x * H::NP + o] = (lds[o] + lds[H::NP + o]) + (lds[2 * H::NP + o] + lds[3 * H::NP + o]);
}

// ---------------------------------------------------------------------------------------------------------------------
// the FC weight gradients that reduce over the batch:  d polfc.weight[a][i] = sum_pos dl[pos][a] h[pos][i]  (65 x 128),
// d v1.weight[h][cell] = sum_pos dv1[pos][h] h[pos][128 + cell]  (VH x 64).  A workgroup stages 8 positions' operands in
// LDS at a time and keeps all 8320 + 64 VH outputs in registers (<= 49 per thread); partial[block][8320 + 64 * 64].
constexpr int kHeadWOut = 65 * 128 + 64 * 64, kHeadWPerThread = (kHeadWOut + 255) / 256;
__global__ __launch_bounds__(256) void k_train_heads_wgrad(const float* __restrict__ hv, const float* __restrict__ dl,
                                                           const float* __restrict__ dv1, int n, int VH, float* __restrict__ partial) {
    __shared__ float hS[8 * 192], dS[8 * 65], vS[8 * 64];
    const int tid = threadIdx.x, nout = 65 * 128 + 64 * VH;
    float acc[kHeadWPerThread];
#pragma unroll
    for (int m = 0; m < kHeadWPerThread; ++m) acc[m] = 0.0f;
    for (int p0 = blockIdx.x * 8; p0 < n; p0 += gridDim.x * 8) {
        const int np = min(8, n - p0);
        __syncthreads();
        for (int i = tid; i < 8 * 192; i += 256) hS[i] = i < np * 192 ? hv[(size_t)p0 * 192 + i] : 0.0f;
        for (int i = tid; i < 8 * 65; i += 256) dS[i] = i < np * 65 ? dl[(size_t)p0 * 65 + i] : 0.0f;
        for (int i = tid; i < 8 * 64; i += 256) vS[i] = i < np * 64 ? dv1[(size_t)p0 * 64 + i] : 0.0f;
        __syncthreads();
#pragma unroll
        for (int m = 0; m < kHeadWPerThread; ++m) {
            const int o = tid + 256 * m;   // (a wave's 64 outputs share a / h: the d reads are broadcasts, the h reads consecutive)
            if (o < 65 * 128) {
                const int a = o >> 7, i = o & 127;
#pragma unroll
                for (int p = 0; p < 8; ++p) acc[m] = fmaf(dS[p * 65 + a], hS[p * 192 + i], acc[m]);
            } else if (o < nout) {
                const int h = (o - 65 * 128) >> 6, cell = o & 63;
#pragma unroll
                for (int p = 0; p < 8; ++p) acc[m] = fmaf(vS[p * 64 + h], hS[p * 192 + 128 + cell], acc[m]);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < kHeadWPerThread; ++m) {
        const int o = tid + 256 * m;
        if (o < kHeadWOut) partial[(size_t)blockIdx.x * kHeadWOut + o] = acc[m];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// all partial sums of the step -> gradient tensors.  Workgroups [0, L * C): the tower's weight gradient, one (layer, ci)
// each: sum over the batch slices of partial[l][s][tap][ci][co] (reads coalesced over co), written in torch's
// [L][co][ci][3][3] through LDS (36-byte runs).  The rest: plain "dst[o] = sum_s src[s * stride + o]" jobs, 64 outputs
// per workgroup (four thread groups share an output's partial vectors).
struct ReduceJob {
    const float* src;
    float* dst;
    int count, parts, stride, inner, outer_stride;   // o = hi * inner + lo  ->  src[hi * outer_stride + s * stride + lo]
    int block0;
    int sticky_from;   // outputs o >= sticky_from are ADDED to dst instead of stored (the step's error word); INT_MAX: none
};
// Adam (Kingma & Ba; torch.optim.Adam's arithmetic, no weight decay / amsgrad -- what train.py:87 constructs) on one element
struct AdamScalars { float lr, beta1, beta2, eps, bc1, bc2_rsqrt; };   // bc1 = 1 - beta1^t, bc2_rsqrt = 1 / sqrt(1 - beta2^t)
__device__ __forceinline__ void adam_update(float g, float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, size_t at,
                                            const AdamScalars& a) {
    const float mn = fmaf(a.beta1, m[at], (1.0f - a.beta1) * g);
    const float vn = fmaf(a.beta2, v[at], (1.0f - a.beta2) * g * g);
    m[at] = mn;
    v[at] = vn;
    p[at] -= (a.lr / a.bc1) * mn / (sqrtf(vn) * a.bc2_rsqrt + a.eps);
}
constexpr int kMaxJobs = 20;
struct FinishArgs {
    const float* tw_partial;
    float* tw_grad;
    float* steps_done;            // the optimiser's step counter (device; null without one): advanced here, one launch ahead of its reader
    int L, S, C, n_jobs;
    ReduceJob job[kMaxJobs];
};
// sum of p[k * stride], k < parts, eight loads in flight (one after the other the loop waits a full memory latency per
// term: 67 us for this kernel instead of ~10)
__device__ __forceinline__ float strided_sum(const float* __restrict__ p, int parts, size_t stride) {
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int k = 0;
    for (; k + 8 <= parts; k += 8) {
        const float a0 = p[(k + 0) * stride], a1 = p[(k + 1) * stride], a2 = p[(k + 2) * stride], a3 = p[(k + 3) * stride];
        const float a4 = p[(k + 4) * stride], a5 = p[(k + 5) * stride], a6 = p[(k + 6) * stride], a7 = p[(k + 7) * stride];
        s0 += a0; s1 += a1; s2 += a2; s3 += a3;
        s0 += a4; s1 += a5; s2 += a6; s3 += a7;
    }
    for (; k < parts; ++k) s0 += p[k * stride];
    return (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(256) void k_train_finish(FinishArgs F) {
    __shared__ float vals[9 * 128];
    const int tid = threadIdx.x, C = F.C;
    if (F.steps_done && blockIdx.x == 0 && tid == 0) *F.steps_done += 1.0f;
    if ((int)blockIdx.x < F.L * C) {
        const int l = blockIdx.x / C, ci = blockIdx.x % C;
        for (int idx = tid; idx < 9 * C; idx += 256) {
            const int tap = idx / C, co = idx % C;
            const float* p = F.tw_partial + (((size_t)l * F.S * 9 + tap) * C + ci) * C + co;
            vals[idx] = strided_sum(p, F.S, (size_t)9 * C * C);
        }
        __syncthreads();
        for (int idx = tid; idx < 9 * C; idx += 256) {
            const int co = idx / 9, tap = idx % 9;
            F.tw_grad[(((size_t)l * C + co) * C + ci) * 9 + tap] = vals[tap * C + co];
        }
        return;
    }
    const int b = blockIdx.x - F.L * C;
    int j = 0;
    while (j + 1 < F.n_jobs && b >= F.job[j + 1].block0) ++j;
    // 64 outputs per workgroup, the partial vectors dealt to four thread groups (up to 512 of them per output: one thread
    // walking them all -- 64 rounds of 8 loads -- made this kernel 10 us longer when the stem's gradient went to 512 partials)
    const ReduceJob J = F.job[j];
    const int o = (b - J.block0) * 64 + (tid & 63), slice = tid >> 6;
    float sum = 0.0f;
    if (o < J.count && slice < J.parts) {
        const float* p = J.src + (size_t)(o / J.inner) * J.outer_stride + (o % J.inner) + (size_t)slice * J.stride;
        sum = strided_sum(p, (J.parts - slice + 3) / 4, (size_t)4 * J.stride);
    }
    vals[tid] = sum;
    __syncthreads();
    if (tid < 64 && o < J.count) {
        const float v = (vals[tid] + vals[tid + 64]) + (vals[tid + 128] + vals[tid + 192]);
        J.dst[o] = o >= J.sticky_from ? J.dst[o] + v : v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// the Adam update of all 14 parameter tensors in one launch, 1024 consecutive elements per workgroup (coalesced: inside
// k_train_finish the same update ran on the tower's transposed write pattern, 36-byte runs x 7 streams: 48 us instead of 6).
// hyper (device): {learning rate, step number t, warm-up steps}; the rate is lr * min(1, t / warm-up).  t is advanced by
// k_train_finish, the launch before this one: a counter advanced HERE needs a last-workgroup ticket -- one device-scope
// atomic and fence per workgroup on one address, which serialised the 1728 workgroups of the 128-channel net to 64 us.
struct AdamJob { float *p, *m, *v; const float* g; int count, block0; };
struct AdamArgs {
    const float* hyper;
    float beta1, beta2, eps;
    int n_jobs;
    AdamJob job[14];
};
__global__ __launch_bounds__(256) void k_train_adam(AdamArgs A) {
    const int tid = threadIdx.x;
    const float t = A.hyper[1], warm = A.hyper[2];
    AdamScalars ad;
    ad.lr = warm > 0.0f ? A.hyper[0] * fminf(1.0f, t / warm) : A.hyper[0];
    ad.beta1 = A.beta1; ad.beta2 = A.beta2; ad.eps = A.eps;
    ad.bc1 = 1.0f - powf(A.beta1, t);
    ad.bc2_rsqrt = 1.0f / sqrtf(1.0f - powf(A.beta2, t));
    int j = 0;
    while (j + 1 < A.n_jobs && (int)blockIdx.x >= A.job[j + 1].block0) ++j;
    const AdamJob J = A.job[j];
    const int base = ((int)blockIdx.x - J.block0) * 1024;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int o = base + 256 * e + tid;
        if (o < J.count) adam_update(J.g[o], J.p, J.m, J.v, (size_t)o, ad);
    }
}

bool ends_shape_ok(int C, int n) { return (C == 64 || C == 128) && n >= 4 && n % 4 == 0; }
int stem_blocks(int n) { const int g = (n + 1) / 2; return g < 512 ? g : 512; }   // k_train_stem_wgrad: 4 half-positions per workgroup pass
int heads_blocks(int n) { const int g = n / 4; return g < 256 ? g : 256; }
int heads_w_blocks(int n) { const int g = (n + 7) / 8; return g < 128 ? g : 128; }

}  // namespace

/* sizes[0..5] = workgroups (= partial vectors) and floats per vector of: stem weight gradient, heads, FC weight gradients */
BZ_EXPORT int32_t bz_train_ends_sizes(int32_t C, int32_t n, int32_t* sizes) {
    BZ_REQUIRE(sizes && ends_shape_ok(C, n), "bz_train_ends_sizes: C must be 64 or 128 and the batch a multiple of 4");
    sizes[0] = stem_blocks(n); sizes[1] = C * 19;
    sizes[2] = heads_blocks(n); sizes[3] = 3 * C + 201;
    sizes[4] = heads_w_blocks(n); sizes[5] = kHeadWOut;
    return BZ_OK;
}

BZ_EXPORT int32_t bz_train_stem_fwd(const bz_train_batch* batch_dev, int32_t n, const float* stem_w, const float* stem_b, int32_t C, void* act0,
                                    void* stream) {
    BZ_REQUIRE(batch_dev && stem_w && stem_b && act0 && ends_shape_ok(C, n), "bz_train_stem_fwd: bad arguments (C = 64 or 128, n a multiple of 4)");
    if (bz_device_count() <= 0) { set_error("bz_train_stem_fwd: no HIP device (the training kernels have no CPU path)"); return BZ_ENOGPU; }
    const dim3 grid((n + 3) / 4);
    if (C == 64) hipLaunchKernelGGL(k_train_stem<64>, grid, dim3(256), 0, (hipStream_t)stream, batch_dev, n, stem_w, stem_b, static_cast<__bf16*>(act0));
    else hipLaunchKernelGGL(k_train_stem<128>, grid, dim3(256), 0, (hipStream_t)stream, batch_dev, n, stem_w, stem_b, static_cast<__bf16*>(act0));
    BZ_LAUNCH_CHECK("k_train_stem");
    return BZ_OK;
}

BZ_EXPORT int32_t bz_train_stem_wgrad(const bz_train_batch* batch_dev, const void* act0, const void* g0, int32_t n, int32_t C, float* partial,
                                      void* stream) {
    BZ_REQUIRE(batch_dev && act0 && g0 && partial && ends_shape_ok(C, n), "bz_train_stem_wgrad: bad arguments (C = 64 or 128, n a multiple of 4)");
    if (bz_device_count() <= 0) { set_error("bz_train_stem_wgrad: no HIP device (the training kernels have no CPU path)"); return BZ_ENOGPU; }
    const dim3 grid(stem_blocks(n));
    if (C == 64) hipLaunchKernelGGL(k_train_stem_wgrad<64>, grid, dim3(256), 0, (hipStream_t)stream, batch_dev, static_cast<const unsigned*>(act0), static_cast<const unsigned*>(g0), n, partial);
    else hipLaunchKernelGGL(k_train_stem_wgrad<128>, grid, dim3(256), 0, (hipStream_t)stream, batch_dev, static_cast<const unsigned*>(act0), static_cast<const unsigned*>(g0), n, partial);
    BZ_LAUNCH_CHECK("k_train_stem_wgrad");
    return BZ_OK;
}

BZ_EXPORT int32_t bz_train_heads(const void* act_top, const bz_train_batch* batch_dev, int32_t n, int32_t C, int32_t VH,
                                 const bz_train_head_params* P, void* g_top, float* hv, float* dl, float* dv1, float* partial, void* stream) {
    BZ_REQUIRE(act_top && batch_dev && P && g_top && hv && dl && dv1 && partial && ends_shape_ok(C, n) && VH >= 1 && VH <= 64,
               "bz_train_heads: bad arguments (C = 64 or 128, n a multiple of 4, value_hidden <= 64)");
    BZ_REQUIRE(P->pol_w && P->pol_b && P->polfc_w && P->polfc_b && P->val_w && P->val_b && P->v1_w && P->v1_b && P->v2_w && P->v2_b,
               "bz_train_heads: a head parameter pointer is null");
    if (bz_device_count() <= 0) { set_error("bz_train_heads: no HIP device (the training kernels have no CPU path)"); return BZ_ENOGPU; }
    HeadArgs A;
    A.x = static_cast<const __bf16*>(act_top); A.batch = batch_dev; A.n = n; A.VH = VH; A.inv_n = 1.0f / (float)n;
    A.pol_w = P->pol_w; A.pol_b = P->pol_b; A.polfc_w = P->polfc_w; A.polfc_b = P->polfc_b; A.val_w = P->val_w; A.val_b = P->val_b;
    A.v1_w = P->v1_w; A.v1_b = P->v1_b; A.v2_w = P->v2_w; A.v2_b = P->v2_b;
    A.g_top = static_cast<__bf16*>(g_top); A.hv = hv; A.dl = dl; A.dv1 = dv1; A.partial = partial;
    const dim3 grid(heads_blocks(n));
    hipStream_t s = (hipStream_t)stream;
    if (C == 64) {
        static unsigned done = 0;
        if (hipError_t e = lds_attr_per_device(reinterpret_cast<const void*>(k_train_heads<64>), Hd<64>::LDS, &done); e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_train_heads)");
        hipLaunchKernelGGL(k_train_heads<64>, grid, dim3(256), Hd<64>::LDS, s, A);
    } else {
        static unsigned done = 0;
        if (hipError_t e = lds_attr_per_device(reinterpret_cast<const void*>(k_train_heads<128>), Hd<128>::LDS, &done); e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_train_heads)");
        hipLaunchKernelGGL(k_train_heads<128>, grid, dim3(256), Hd<128>::LDS, s, A);
    }
    BZ_LAUNCH_CHECK("k_train_heads");
    return BZ_OK;
}

BZ_EXPORT int32_t bz_train_heads_wgrad(const float* hv, const float* dl, const float* dv1, int32_t n, int32_t VH, float* partial, void* stream) {
    BZ_REQUIRE(hv && dl && dv1 && partial && n >= 1 && VH >= 1 && VH <= 64, "bz_train_heads_wgrad: bad arguments (value_hidden <= 64)");
    if (bz_device_count() <= 0) { set_error("bz_train_heads_wgrad: no HIP device (the training kernels have no CPU path)"); return BZ_ENOGPU; }
    hipLaunchKernelGGL(k_train_heads_wgrad, dim3(heads_w_blocks(n)), dim3(256), 0, (hipStream_t)stream, hv, dl, dv1, n, VH, partial);
    BZ_LAUNCH_CHECK("k_train_heads_wgrad");
    return BZ_OK;
}

namespace {
bool all_set(const bz_train_tensors* T) {
    return T->stem_w && T->stem_b && T->tower_w && T->tower_b && T->pol_w && T->pol_b && T->polfc_w && T->polfc_b && T->val_w && T->val_b &&
           T->v1_w && T->v1_b && T->v2_w && T->v2_b;
}
}  // namespace

BZ_EXPORT int32_t bz_train_finish(const bz_train_partials* Q, const bz_train_tensors* G, int32_t C, int32_t n_layers, int32_t VH, int32_t n,
                                  float* losses, const bz_train_adam* opt, void* stream) {
    BZ_REQUIRE(Q && G && losses && ends_shape_ok(C, n) && n_layers >= 2 && VH >= 1 && VH <= 64, "bz_train_finish: bad arguments");
    BZ_REQUIRE(Q->tower && Q->tower_b && Q->stem && Q->heads && Q->heads_w && Q->splits >= 1, "bz_train_finish: a partial-sum pointer is null");
    BZ_REQUIRE(all_set(G), "bz_train_finish: a gradient pointer is null");
    if (opt) {
        BZ_REQUIRE(opt->hyper && all_set(&opt->p) && all_set(&opt->m) && all_set(&opt->v), "bz_train_finish: the optimiser block has a null pointer");
        BZ_REQUIRE(opt->beta1 >= 0.0f && opt->beta1 < 1.0f && opt->beta2 >= 0.0f && opt->beta2 < 1.0f && opt->eps > 0.0f,
                   "bz_train_finish: Adam needs 0 <= beta < 1 and eps > 0");
    }
    if (bz_device_count() <= 0) { set_error("bz_train_finish: no HIP device (the training kernels have no CPU path)"); return BZ_ENOGPU; }
    FinishArgs F;
    F.tw_partial = Q->tower; F.tw_grad = G->tower_w; F.L = n_layers; F.S = Q->splits; F.C = C; F.n_jobs = 0;
    F.steps_done = opt ? opt->hyper + 1 : nullptr;
    int next_block = 0;
    // field = the tensor's slot in bz_train_tensors (null for the three loss values)
    auto add = [&](const float* src, float* bz_train_tensors::*field, float* dst_plain, int count, int parts, int stride, int inner, int outer_stride) {
        ReduceJob& J = F.job[F.n_jobs++];
        J.src = src; J.dst = field ? G->*field : dst_plain;
        J.count = count; J.parts = parts; J.stride = stride; J.inner = inner; J.outer_stride = outer_stride; J.block0 = next_block;
        J.sticky_from = 0x7fffffff;
        next_block += (count + 63) / 64;
    };
    typedef bz_train_tensors T;
    const int nS = stem_blocks(n), nH = heads_blocks(n), nW = heads_w_blocks(n), NP = 3 * C + 201;
    const int bias_rows = 2 * Q->splits * (C / 32);   // bz_train_wgrad_bias_rows()
    add(Q->tower_b, &T::tower_b, nullptr, n_layers * C, bias_rows, C, C, bias_rows * C);
    add(Q->stem, &T::stem_w, nullptr, C * 18, nS, C * 19, C * 18, 0);
    add(Q->stem + C * 18, &T::stem_b, nullptr, C, nS, C * 19, C, 0);
    add(Q->heads, &T::pol_w, nullptr, 2 * C, nH, NP, 2 * C, 0);
    add(Q->heads + 2 * C, &T::val_w, nullptr, C, nH, NP, C, 0);
    add(Q->heads + 3 * C, &T::pol_b, nullptr, 2, nH, NP, 2, 0);
    add(Q->heads + 3 * C + 2, &T::val_b, nullptr, 1, nH, NP, 1, 0);
    add(Q->heads + 3 * C + 3, &T::polfc_b, nullptr, 65, nH, NP, 65, 0);
    add(Q->heads + 3 * C + 68, &T::v1_b, nullptr, VH, nH, NP, VH, 0);
    add(Q->heads + 3 * C + 132, &T::v2_w, nullptr, VH, nH, NP, VH, 0);
    add(Q->heads + 3 * C + 196, &T::v2_b, nullptr, 1, nH, NP, 1, 0);
    add(Q->heads + 3 * C + 197, nullptr, losses, 4, nH, NP, 4, 0);
    F.job[F.n_jobs - 1].sticky_from = 3;   // losses[3], the error word, ACCUMULATES until the caller zeroes it
    add(Q->heads_w, &T::polfc_w, nullptr, 65 * 128, nW, kHeadWOut, 65 * 128, 0);
    add(Q->heads_w + 65 * 128, &T::v1_w, nullptr, VH * 64, nW, kHeadWOut, VH * 64, 0);
    hipLaunchKernelGGL(k_train_finish, dim3(n_layers * C + next_block), dim3(256), 0, (hipStream_t)stream, F);
    BZ_LAUNCH_CHECK("k_train_finish");
    if (opt) {
        AdamArgs A;
        A.hyper = opt->hyper; A.beta1 = opt->beta1; A.beta2 = opt->beta2; A.eps = opt->eps; A.n_jobs = 0;
        int blocks = 0;
        auto adam = [&](float* bz_train_tensors::*field, int count) {
            AdamJob& J = A.job[A.n_jobs++];
            J.p = opt->p.*field; J.m = opt->m.*field; J.v = opt->v.*field; J.g = G->*field; J.count = count; J.block0 = blocks;
            blocks += (count + 1023) / 1024;
        };
        adam(&T::tower_w, n_layers * C * C * 9); adam(&T::tower_b, n_layers * C); adam(&T::stem_w, C * 18); adam(&T::stem_b, C);
        adam(&T::pol_w, 2 * C); adam(&T::pol_b, 2); adam(&T::polfc_w, 65 * 128); adam(&T::polfc_b, 65); adam(&T::val_w, C); adam(&T::val_b, 1);
        adam(&T::v1_w, VH * 64); adam(&T::v1_b, VH); adam(&T::v2_w, VH); adam(&T::v2_b, 1);
        hipLaunchKernelGGL(k_train_adam, dim3(blocks), dim3(256), 0, (hipStream_t)stream, A);
        BZ_LAUNCH_CHECK("k_train_adam");
    }
    return BZ_OK;
}
