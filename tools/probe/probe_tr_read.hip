// probe_tr_read.hip -- what ds_read_b64_tr_b16 (gfx950) returns, checked with unique data.  The wgrad kernel of
// bz_train.hip feeds both MFMA operands through it (activations and gradients are stored [cell][channel], the MFMA wants
// 8 consecutive CELLS of one channel per lane), so the lane map below is a correctness premise, not a detail.
// Expected (cdna_hip_programming.md T10): per group of 16 consecutive lanes, lane 4q + p supplies the address of row q,
// columns 4p .. 4p+3 of a 4 x 16 block of 16-bit elements; lane i receives column i of the 4 rows (row q in element q).
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe/probe_tr_read tools/probe/probe_tr_read.hip && tools/probe/probe_tr_read
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;

__global__ void k(s16x4* out, int pitch) {
    __shared__ __attribute__((aligned(16))) short img[64 * 80];
    for (int i = threadIdx.x; i < 64 * 80; i += 64) img[i] = (short)((i / pitch) * 64 + (i % pitch));  // value = row * 64 + col
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    const int row = 5 * g + 2 * q + 1, col = 16 * ((g + 1) & 3) + 4 * p;  // rows need not be adjacent; groups are independent
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + row * pitch + col));
    out[l] = v;
}

int main() {
    s16x4* d;
    s16x4 h[64];
    int bad = 0;
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) { printf("no device\n"); return 2; }
    for (int pitch : {64, 72}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, pitch);
        if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 2; }
        for (int l = 0; l < 64; ++l) {
            const int g = l >> 4, i = l & 15;
            for (int e = 0; e < 4; ++e) {
                const int want = (5 * g + 2 * e + 1) * 64 + 16 * ((g + 1) & 3) + i;
                if (h[l][e] != want) { if (bad < 8) printf("pitch %d lane %d elem %d: got %d (row %d col %d) want %d\n", pitch, l, e, h[l][e], h[l][e] / 64, h[l][e] % 64, want); ++bad; }
            }
        }
    }
    printf(bad ? "MISMATCH (%d)\n" : "ds_read_b64_tr_b16: lane 4q+p supplies row q cols 4p..4p+3; lane i receives column i, row q in element q -- confirmed (%d mismatches)\n", bad);
    return bad != 0;
}
