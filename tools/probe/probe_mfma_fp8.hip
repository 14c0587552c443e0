// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3, unit scales) on gfx950:
// (1) lane -> row/col map of A/B, (2) that byte (h, j) of A pairs with byte (h, j) of B,
// (3) value semantics (e4m3 products, E8M0 scale 127 = 1.0, other scales).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k_probe(const uint8_t* A, const uint8_t* B, float* D, int scale_a, int scale_b) {
    int l = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = ((const int*)(A + l * 32))[i]; b[i] = ((const int*)(B + l * 32))[i]; }
    v16f c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
    for (int i = 0; i < 16; ++i) D[l * 16 + i] = c[i];
}

int main() {
    uint8_t *A, *B; float* D;
    hipMallocManaged(&A, 64 * 32); hipMallocManaged(&B, 64 * 32); hipMallocManaged(&D, 64 * 16 * 4);
    const uint8_t ONE = 0x38, TWO = 0x40, THREE = 0x44;  // e4m3: 1.0, 2.0, 3.0
    auto run = [&](int sa, int sb) { hipLaunchKernelGGL(k_probe, 1, 64, 0, 0, A, B, D, sa, sb); hipDeviceSynchronize(); };
    auto Dat = [&](int row, int col) { int lane = (col & 31) + 32 * ((row >> 2) & 1); int reg = (row & 3) + 4 * (row >> 3); return D[lane * 16 + reg]; };
    // (1) row map of A: only lane la has ones -> which rows of D light up (B all ones)
    int bad = 0;
    for (int la = 0; la < 64; ++la) {
        memset(A, 0, 2048); memset(B, ONE, 2048);
        memset(A + la * 32, ONE, 32);
        run(127, 127);
        for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
            float e = (r == (la & 31)) ? 32.0f : 0.0f;
            if (Dat(r, c) != e) { if (bad++ < 5) printf("rowmap la=%d r=%d c=%d got %g exp %g\n", la, r, c, Dat(r, c), e); }
        }
    }
    printf("A lane->row = lane&31, 32 k-values per lane: %s\n", bad ? "FAIL" : "ok");
    bad = 0;
    for (int lb = 0; lb < 64; ++lb) {
        memset(B, 0, 2048); memset(A, ONE, 2048);
        memset(B + lb * 32, ONE, 32);
        run(127, 127);
        for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
            float e = (c == (lb & 31)) ? 32.0f : 0.0f;
            if (Dat(r, c) != e) { if (bad++ < 5) printf("colmap lb=%d r=%d c=%d got %g exp %g\n", lb, r, c, Dat(r, c), e); }
        }
    }
    printf("B lane->col = lane&31: %s\n", bad ? "FAIL" : "ok");
    // (2) k pairing: A byte (h0,j0) x B byte (h1,j1) nonzero iff equal
    bad = 0;
    for (int h0 = 0; h0 < 2; ++h0) for (int j0 = 0; j0 < 32; ++j0) {
        memset(A, 0, 2048);
        for (int l = 0; l < 32; ++l) A[(l + 32 * h0) * 32 + j0] = TWO;
        for (int h1 = 0; h1 < 2; ++h1) for (int j1 = 0; j1 < 32; ++j1) {
            memset(B, 0, 2048);
            for (int l = 0; l < 32; ++l) B[(l + 32 * h1) * 32 + j1] = THREE;
            run(127, 127);
            float e = (h0 == h1 && j0 == j1) ? 6.0f : 0.0f;
            if (Dat(5, 9) != e || Dat(31, 0) != e) { if (bad++ < 8) printf("pair (%d,%d)x(%d,%d) got %g exp %g\n", h0, j0, h1, j1, Dat(5, 9), e); }
        }
    }
    printf("k pairing A(h,j) <-> B(h,j): %s\n", bad ? "FAIL" : "ok");
    // (3) scales: 2^(sa-127) * 2^(sb-127)
    memset(A, ONE, 2048); memset(B, ONE, 2048);
    run(127, 127); printf("scale 127/127: D=%g (expect 64)\n", Dat(0, 0));
    run(128, 127); printf("scale 128/127: D=%g (expect 128 if lane scale applies to whole lane)\n", Dat(0, 0));
    run(127, 126); printf("scale 127/126: D=%g (expect 32)\n", Dat(0, 0));
    run(0x7F7F7F80, 127); printf("scale bytes 80,7f,7f,7f opsel0: D=%g\n", Dat(0, 0));
    return 0;
}
