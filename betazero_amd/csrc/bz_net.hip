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
//    (2 x 65 KB, XOR-swizzled 256-B cells, +1 zero cell per position for the
//    conv halo) across all 2*NB conv layers; wave w owns output channels
//    32w..32w+31 of all 4 positions (8 accumulator tiles of
//    v_mfma_f32_32x32x16_bf16), streams its weight fragments straight from
//    L2 into registers in a fragment-major layout (one coalesced 1 KB load per
//    k-step, prefetched one tap = 8 k-steps ahead) and reads the activation
//    fragments from LDS with conflict-free ds_read_b128.  Activations never
//    touch HBM between the stem and the heads; one barrier per layer.
//  * f32 (parity): per-layer VALU kernels whose every accumulation is the
//    k-ordered fmaf chain of oracle/bz_oracle.c -> bit-identical to the oracle.
#include <math.h>
#include <new>
#include <vector>

#include "bz_common.h"
#include "bz_math.h"

using namespace bz;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct bz_net {
    int C, NB, VH, max_batch;
    // parameters (device)
    float *stem_w, *stem_b;          // [9][2][C], [C]
    float *conv_w, *conv_b;          // [2NB][9][C][C] as [tap][ci][co]; [2NB][C]
    __bf16* conv_wf;                 // [2NB+pad][9][8][4][64][8] fragment-major (C == 128)
    float *pol_w, *pol_b, *polfc_wT, *polfc_b;  // [2][C], [2], [128][65], [65]
    float *val_w, *val_b, *v1_wT, *v1_b, *v2_w, *v2_b;  // [C], [1], [64][VH], [VH], [VH], [1]
    // activations (device)
    float *act_a, *act_b;            // f32 path: [max_batch][64][C] x 2
    __bf16* act_h;                   // bf16 path: [max_batch][64][C]
};

namespace {

// ------------------------------------------------------------------ stem
// thread = output channel, block = position.  x in {0,1}: fmaf(1,w,acc) == acc + w.
template <class OutT>
__global__ void k_stem(const u64* __restrict__ own, const u64* __restrict__ opp, int n, int C,
                       const float* __restrict__ w, const float* __restrict__ b, OutT* __restrict__ out) {
    int pos = blockIdx.x, co = threadIdx.x;
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
                                                  int C) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* xs = reinterpret_cast<float*>(smem_raw);  // [64][C]
    int pos = blockIdx.x;
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
__global__ void __launch_bounds__(192) k_heads(const InT* __restrict__ act, int n, int C, int VH,
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

// ------------------------------------------------------------------ bf16 MFMA tower
constexpr int kTC = 128;                  // channels
constexpr int kPosPerWG = 4;
constexpr int kTileBytes = 65 * 256;      // 64 cells x 256 B + one zero cell
constexpr int kBufBytes = kPosPerWG * kTileBytes;
constexpr int kTowerLds = 2 * kBufBytes;  // 133,120 B
constexpr int kFragsPerLayer = 9 * 8 * 4 * 64;  // 16-byte fragments per layer

// byte offset of 16-byte chunk k of cell c inside a position tile (XOR swizzle:
// the 16 lanes of every ds_read_b128 lane group hit 16 distinct slots)
__device__ __forceinline__ int cell_off(int c, int k) { return c * 256 + ((k ^ (c & 15)) << 4); }

__global__ void __launch_bounds__(256, 1)
k_tower_bf16(__bf16* __restrict__ act, int n, int n_layers, const uint4* __restrict__ wf,
             const float* __restrict__ bias) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pos0 = blockIdx.x * kPosPerWG;
    char* bufX = smem;
    char* bufM = smem + kBufBytes;

    // ---- zero cells, then the 4 input tiles (coalesced 16-B chunks, swizzled LDS image)
    if (tid < 128) {
        int b = tid >> 6, p = (tid >> 4) & 3, k = tid & 15;
        *reinterpret_cast<uint4*>(smem + b * kBufBytes + p * kTileBytes + 64 * 256 + k * 16) = make_uint4(0, 0, 0, 0);
    }
    {
        const uint4* src = reinterpret_cast<const uint4*>(act) + (size_t)pos0 * 1024;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            int i = it * 256 + tid, p = i >> 10, c = (i >> 4) & 63, k = i & 15;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (pos0 + p < n) v = src[i];
            *reinterpret_cast<uint4*>(bufX + p * kTileBytes + cell_off(c, k)) = v;
        }
    }
    __syncthreads();

    const int r = lane & 31, h = lane >> 5;
    // weight-fragment stream of this wave: k-step ks -> wf[(ks*4 + w)*64 + lane], linear over layers
    const uint4* ap = wf + (size_t)w * 64 + lane;
    bf16x8 a_cur[8], a_nxt[8];
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) a_cur[kc] = __builtin_bit_cast(bf16x8, ap[(size_t)kc * 256]);
    ap += 8 * 256;

    for (int layer = 0; layer < n_layers; ++layer) {
        const bool second = layer & 1;  // conv2 of a block: in = M, out = X (in place), skip = X
        const char* in = second ? bufM : bufX;
        char* out = second ? bufX : bufM;
        f32x16 acc[kPosPerWG][2];
#pragma unroll
        for (int p = 0; p < kPosPerWG; ++p) { acc[p][0] = (f32x16)(0.0f); acc[p][1] = (f32x16)(0.0f); }

        for (int tap = 0; tap < 9; ++tap) {
            // next tap's (or next layer's first tap's) weight fragments; the buffer is padded by one tap
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) a_nxt[kc] = __builtin_bit_cast(bf16x8, ap[(size_t)kc * 256]);
            ap += 8 * 256;
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            int boff[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                int cell = 32 * nt + r, yy = (cell >> 3) + dy, xx = (cell & 7) + dx;
                bool inb = (unsigned)yy < 8u && (unsigned)xx < 8u;
                int c2 = inb ? yy * 8 + xx : 64;
                boff[nt] = c2 * 256 + (((c2 & 15) ^ h) << 4);
            }
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                bf16x8 bfr[kPosPerWG][2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const char* bp = in + (boff[nt] ^ (kc << 5));
#pragma unroll
                    for (int p = 0; p < kPosPerWG; ++p)
                        bfr[p][nt] = *reinterpret_cast<const bf16x8*>(bp + p * kTileBytes);
                }
#pragma unroll
                for (int p = 0; p < kPosPerWG; ++p)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[p][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[kc], bfr[p][nt], acc[p][nt], 0, 0, 0);
            }
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) a_cur[kc] = a_nxt[kc];
        }

        // ---- epilogue: +bias (+skip) -> ReLU -> bf16 -> LDS.  D[row = co][col = cell]:
        // lane (r, h) register 4q+i holds co = 32w + 8q + 4h + i of cell 32nt + r.
        f32x4 bq[4];
        const float* bl = bias + (size_t)layer * kTC + 32 * w + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4*>(bl + 8 * q);
#pragma unroll
        for (int p = 0; p < kPosPerWG; ++p)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                int cell = 32 * nt + r;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int off = p * kTileBytes + cell_off(cell, 4 * w + q) + 8 * h;
                    f32x4 v = {acc[p][nt][4 * q], acc[p][nt][4 * q + 1], acc[p][nt][4 * q + 2], acc[p][nt][4 * q + 3]};
                    v = v + bq[q];
                    if (second) {
                        bf16x4 s = *reinterpret_cast<const bf16x4*>(out + off);
                        v = v + __builtin_convertvector(s, f32x4);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.0f ? v[i] : 0.0f;
                    *reinterpret_cast<bf16x4*>(out + off) = __builtin_convertvector(v, bf16x4);
                }
            }
        __syncthreads();
    }

    // ---- result tile (always X after an even number of layers) back to HBM
    {
        uint4* dst = reinterpret_cast<uint4*>(act) + (size_t)pos0 * 1024;
        const char* res = (n_layers & 1) ? bufM : bufX;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            int i = it * 256 + tid, p = i >> 10, c = (i >> 4) & 63, k = i & 15;
            if (pos0 + p < n) dst[i] = *reinterpret_cast<const uint4*>(res + p * kTileBytes + cell_off(c, k));
        }
    }
}

// ------------------------------------------------------------------ host helpers
struct Carver {
    int64_t off = 0;
    int64_t take(int64_t bytes) { int64_t o = off; off += (bytes + 255) & ~int64_t(255); return o; }
};
struct NetOffsets {
    int64_t stem_w, stem_b, conv_w, conv_b, conv_wf, pol_w, pol_b, polfc_wT, polfc_b, val_w, val_b, v1_wT, v1_b, v2_w,
        v2_b, act_a, act_b, act_h, total;
};
NetOffsets net_carve(int C, int NB, int VH, int mb) {
    NetOffsets o{};
    Carver k;
    int64_t L = 2 * NB, mbp = (mb + 3) & ~3;
    o.stem_w = k.take(18LL * C * 4); o.stem_b = k.take(C * 4LL);
    o.conv_w = k.take(L * 9 * C * C * 4); o.conv_b = k.take(L * C * 4);
    o.conv_wf = k.take(C == kTC ? (L * 9 + 1) * 8LL * 4 * 64 * 16 : 0);
    o.pol_w = k.take(2LL * C * 4); o.pol_b = k.take(8); o.polfc_wT = k.take(128 * 65 * 4); o.polfc_b = k.take(65 * 4);
    o.val_w = k.take(C * 4LL); o.val_b = k.take(4); o.v1_wT = k.take(64LL * VH * 4); o.v1_b = k.take(VH * 4LL);
    o.v2_w = k.take(VH * 4LL); o.v2_b = k.take(4);
    o.act_a = k.take(mbp * 64 * C * 4); o.act_b = k.take(mbp * 64 * C * 4); o.act_h = k.take(mbp * 64 * C * 2);
    o.total = k.off;
    return o;
}
bool shape_ok(int C, int NB, int VH, int mb) {
    return (C == 32 || C == 64 || C == 128) && NB >= 0 && NB <= 64 && VH >= 1 && VH <= 64 && mb >= 1;
}
uint16_t f2bf(float f) {
    u32 u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)(u >> 16);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
template <class T> T* at(void* base, int64_t off) { return reinterpret_cast<T*>(static_cast<char*>(base) + off); }
}  // namespace

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
    n->conv_wf = C == kTC ? at<__bf16>(ws, o.conv_wf) : nullptr;
    n->pol_w = at<float>(ws, o.pol_w); n->pol_b = at<float>(ws, o.pol_b);
    n->polfc_wT = at<float>(ws, o.polfc_wT); n->polfc_b = at<float>(ws, o.polfc_b);
    n->val_w = at<float>(ws, o.val_w); n->val_b = at<float>(ws, o.val_b);
    n->v1_wT = at<float>(ws, o.v1_wT); n->v1_b = at<float>(ws, o.v1_b);
    n->v2_w = at<float>(ws, o.v2_w); n->v2_b = at<float>(ws, o.v2_b);
    n->act_a = at<float>(ws, o.act_a); n->act_b = at<float>(ws, o.act_b); n->act_h = at<__bf16>(ws, o.act_h);

    // ---- host repack into one staging image of the parameter region, one H2D copy
    int64_t param_bytes = o.act_a;
    std::vector<char> img((size_t)param_bytes, 0);
    auto F = [&](int64_t off) { return reinterpret_cast<float*>(img.data() + off); };
    int L = 2 * NB;
    const float* q = p;
    for (int co = 0; co < C; ++co)
        for (int ci = 0; ci < 2; ++ci)
            for (int t = 0; t < 9; ++t) F(o.stem_w)[(t * 2 + ci) * C + co] = q[(co * 2 + ci) * 9 + t];
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
                    if (C == kTC) {  // fragment-major: [l][t][kc][mt][lane = 32h + r][j]
                        int kc = ci >> 4, hh = (ci >> 3) & 1, j = ci & 7, mt = co >> 5, rr = co & 31;
                        size_t f = ((((size_t)l * 9 + t) * 8 + kc) * 4 + mt) * 64 + (hh * 32 + rr);
                        wf[f * 8 + j] = f2bf(v);
                    }
                }
        q += (size_t)C * C * 9;
        for (int i = 0; i < C; ++i) F(o.conv_b)[(size_t)l * C + i] = q[i];
        q += C;
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
    hipStream_t s = (hipStream_t)stream;
    hipError_t e1 = hipMemcpyAsync(ws, img.data(), (size_t)param_bytes, hipMemcpyHostToDevice, s);
    hipError_t e2 = e1 == hipSuccess ? hipStreamSynchronize(s) : e1;
    if (e2 != hipSuccess) { delete n; return hip_fail(e2, "bz_net_create upload"); }
    if (C == kTC) {
        hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_tower_bf16),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, kTowerLds);
        if (e3 != hipSuccess) { delete n; return hip_fail(e3, "hipFuncSetAttribute(k_tower_bf16)"); }
    }
    *out = n;
    return BZ_OK;
}

BZ_EXPORT int32_t bz_net_destroy(bz_net* net) { delete net; return BZ_OK; }

static int32_t launch_heads_f32(bz_net* n, int cnt, float* logits, float* value, hipStream_t s) {
    size_t lds = (64 * (n->C + 1) + 128 + 64 + 64) * sizeof(float);
    hipLaunchKernelGGL(k_heads<float>, dim3(cnt), dim3(192), lds, s, n->act_a, cnt, n->C, n->VH, n->pol_w, n->pol_b,
                       n->polfc_wT, n->polfc_b, n->val_w, n->val_b, n->v1_wT, n->v1_b, n->v2_w, n->v2_b, logits, value);
    BZ_LAUNCH_CHECK("k_heads<float>");
    return BZ_OK;
}

BZ_EXPORT int32_t bz_net_forward_f32(bz_net* n, const uint64_t* own, const uint64_t* opp, int32_t cnt, float* logits,
                                     float* value, void* stream) {
    BZ_REQUIRE(n && own && opp && logits && value, "bz_net_forward_f32: null pointer");
    BZ_REQUIRE(cnt >= 0 && cnt <= n->max_batch, "bz_net_forward_f32: batch exceeds max_batch");
    if (cnt == 0) return BZ_OK;
    hipStream_t s = (hipStream_t)stream;
    int C = n->C;
    hipLaunchKernelGGL(k_stem<float>, dim3(cnt), dim3(C), 0, s, own, opp, cnt, C, n->stem_w, n->stem_b, n->act_a);
    BZ_LAUNCH_CHECK("k_stem<float>");
    size_t lds = (size_t)64 * C * sizeof(float);
    for (int blk = 0; blk < n->NB; ++blk) {
        const float* w1 = n->conv_w + (size_t)(2 * blk) * 9 * C * C;
        const float* w2 = n->conv_w + (size_t)(2 * blk + 1) * 9 * C * C;
        hipLaunchKernelGGL(k_conv_f32, dim3(cnt), dim3(256), lds, s, n->act_a, w1, n->conv_b + (size_t)(2 * blk) * C,
                           (const float*)nullptr, n->act_b, cnt, C);
        BZ_LAUNCH_CHECK("k_conv_f32");
        hipLaunchKernelGGL(k_conv_f32, dim3(cnt), dim3(256), lds, s, n->act_b, w2,
                           n->conv_b + (size_t)(2 * blk + 1) * C, (const float*)n->act_a, n->act_a, cnt, C);
        BZ_LAUNCH_CHECK("k_conv_f32");
    }
    return launch_heads_f32(n, cnt, logits, value, s);
}

BZ_EXPORT int32_t bz_net_forward_bf16(bz_net* n, const uint64_t* own, const uint64_t* opp, int32_t cnt, float* logits,
                                      float* value, void* stream) {
    BZ_REQUIRE(n && own && opp && logits && value, "bz_net_forward_bf16: null pointer");
    BZ_REQUIRE(n->C == kTC, "bz_net_forward_bf16: the MFMA tower is built for C == 128");
    BZ_REQUIRE(cnt >= 0 && cnt <= n->max_batch, "bz_net_forward_bf16: batch exceeds max_batch");
    if (cnt == 0) return BZ_OK;
    hipStream_t s = (hipStream_t)stream;
    {
        ProfScope ps(BZ_PROF_STEM, stream);
        hipLaunchKernelGGL(k_stem<__bf16>, dim3(cnt), dim3(kTC), 0, s, own, opp, cnt, kTC, n->stem_w, n->stem_b, n->act_h);
    }
    BZ_LAUNCH_CHECK("k_stem<bf16>");
    if (n->NB > 0) {
        ProfScope ps(BZ_PROF_TOWER, stream);
        hipLaunchKernelGGL(k_tower_bf16, dim3((cnt + kPosPerWG - 1) / kPosPerWG), dim3(256), kTowerLds, s, n->act_h, cnt,
                           2 * n->NB, reinterpret_cast<const uint4*>(n->conv_wf), n->conv_b);
        BZ_LAUNCH_CHECK("k_tower_bf16");
    }
    size_t lds = (64 * (kTC + 1) + 128 + 64 + 64) * sizeof(float);
    ProfScope ps(BZ_PROF_HEADS, stream);
    hipLaunchKernelGGL(k_heads<__bf16>, dim3(cnt), dim3(192), lds, s, n->act_h, cnt, kTC, n->VH, n->pol_w, n->pol_b,
                       n->polfc_wT, n->polfc_b, n->val_w, n->val_b, n->v1_wT, n->v1_b, n->v2_w, n->v2_b, logits, value);
    BZ_LAUNCH_CHECK("k_heads<bf16>");
    return BZ_OK;
}
