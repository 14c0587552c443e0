// probe_mfma_issue.hip -- how many cycles does one v_mfma_f32_32x32x16_bf16 cost a single wave per SIMD when
// independent MFMAs are issued back to back, alone and interleaved with the other instructions of the tower's K-loop
// (ds_read_b128 + s_waitcnt, a global load per 8 MFMAs)?  One workgroup of 4 waves per CU, like the tower kernel.
//   hipcc --offload-arch=gfx950 -O3 -o probe_mfma_issue probe_mfma_issue.hip && ./probe_mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NACC>
__global__ void __launch_bounds__(256, 1) k(const uint4* __restrict__ w, float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 81920 / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80);
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int u = 0; u < NACC; ++u) acc[u] = (f32x16)(0.0f);
    bf16x8 a8[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) a8[ks] = __builtin_bit_cast(bf16x8, w[ks * 64 + lane]);
    bf16x8 b[NACC];
    const char* bp = smem + lane * 16;
    [[maybe_unused]] int boff = 0;
    if (MODE >= 7) {  // the tower's row-tile pattern: lane (r, h) -> position r >> 3, column r & 7, 256-B cells, XOR swizzle
        const int r = lane & 31, h = lane >> 5, p = r >> 3, x = r & 7;
        int sw = x | ((p & 1) << 3);
        if (MODE == 8) sw = (x << 1) | (p & 1);      // candidate: position bit in the LOW slot bit
        if (MODE == 9) sw = x | ((p >> 1) << 3);     // candidate: pair positions (0,1) / (2,3)
        boff = p * 18688 + (x + 1) * 256 + ((sw ^ h) << 4);
    }
#pragma unroll
    for (int u = 0; u < NACC; ++u) b[u] = *reinterpret_cast<const bf16x8*>(bp + u * 2304);
    const uint4* ap = w + lane;
    unsigned long long t0, t1;
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(w), 0, 1024 * 64 * 16 + 4096, 0x00020000);
    [[maybe_unused]] bf16x8 n8[8];
    [[maybe_unused]] bf16x8 extra[NACC];
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    auto kstep = [&](bf16x8& use, bf16x8& load_into, int it, int ks, int where) {
        if (MODE == 10) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
#pragma unroll
        for (int u = 0; u < NACC; ++u) {
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(use, b[u], acc[u], 0, 0, 0);
            if (MODE >= 1) b[u] = *reinterpret_cast<const bf16x8*>(MODE >= 7 ? smem + (boff ^ ((ks & 7) << 5)) + u * 2304 : bp + u * 2304 + ((ks & 7) << 5));
            if (MODE == 6) {  // a second, independent ds_read_b128 per MFMA: is the 1:1 ratio already the LDS's limit?
                bf16x8 e = *reinterpret_cast<const bf16x8*>(bp + 20480 + u * 2304 + ((ks & 7) << 5));
                extra[u] = e;
            }
            if (u == where) {
                if (MODE == 10) {
                    const uint4* sb = w + (size_t)((it * 8 + ks) & 1023) * 64;  // wave-uniform
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(load_into) : "v"(lane * 16), "s"(sb) : "memory");
                } else if (MODE == 5) load_into = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, ((it * 8 + ks) & 1023) * 1024, 0));
                else if (MODE >= 2 && MODE < 6) load_into = __builtin_bit_cast(bf16x8, ap[(size_t)((it * 8 + ks) & 1023) * 64]);
            }
        }
        if (MODE >= 1) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, MODE == 6 ? 2 : 1, 0);
                if (MODE >= 2 && (MODE < 6 || MODE == 10) && j == where) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
    };
    if (MODE == 3) {  // two register sets: the load never targets registers an MFMA in flight reads
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) n8[ks] = a8[ks];
#pragma unroll 1
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) kstep(a8[ks], n8[ks], it, ks, NACC - 1);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) kstep(n8[ks], a8[ks], it + 1, ks, NACC - 1);
        }
    } else {
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) kstep(a8[ks], a8[ks], it, ks, MODE == 4 ? 3 : NACC - 1);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.0f;
    if (MODE == 6) {
#pragma unroll
        for (int u = 0; u < NACC; ++u) s += (float)extra[u][0];
    }
#pragma unroll
    for (int u = 0; u < NACC; ++u) s += acc[u][0] + acc[u][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NACC>
void run(const char* name, const uint4* w, float* out, unsigned long long* cyc, int grid) {
    const int iters = 200;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, NACC>), hipFuncAttributeMaxDynamicSharedMemorySize, 149504);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<MODE, NACC>), dim3(grid), dim3(256), 149504, 0, w, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double m = 0;
    for (auto v : h) m += (double)v;
    m /= grid;
    // s_memtime counts at a fixed 100 MHz on gfx950?  report raw ticks per MFMA and let the reader compare the modes
    printf("%-44s units %d: %.2f ticks per MFMA (%.0f ticks, %d MFMAs per wave)\n", name, NACC, m / (iters * 8.0 * NACC), m, iters * 8 * NACC);
}

int main() {
    uint4* w; float* out; unsigned long long* cyc;
    const int grid = 256;
    hipMalloc(&w, 1024 * 64 * 16 + 4096); hipMemset(w, 0, 1024 * 64 * 16 + 4096);
    hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 8);
    run<0, 8>("MFMA only", w, out, cyc, grid);
    run<0, 7>("MFMA only", w, out, cyc, grid);
    run<1, 8>("MFMA + ds_read_b128 1:1", w, out, cyc, grid);
    run<1, 7>("MFMA + ds_read_b128 1:1", w, out, cyc, grid);
    run<2, 8>("+ weight load 8 k-steps ahead, same regs", w, out, cyc, grid);
    run<2, 7>("+ weight load 8 k-steps ahead, same regs", w, out, cyc, grid);
    run<3, 8>("+ weight load, two register sets", w, out, cyc, grid);
    run<3, 7>("+ weight load, two register sets", w, out, cyc, grid);
    run<4, 8>("+ weight load in the middle of the k-step", w, out, cyc, grid);
    run<5, 8>("+ weight load by buffer_load (SGPR offset)", w, out, cyc, grid);
    run<5, 7>("+ weight load by buffer_load (SGPR offset)", w, out, cyc, grid);
    run<10, 8>("+ weight load, SGPR base + 32-bit lane offset (asm)", w, out, cyc, grid);
    run<10, 7>("+ weight load, SGPR base + 32-bit lane offset (asm)", w, out, cyc, grid);
    run<6, 8>("MFMA + 2 ds_read_b128 per MFMA", w, out, cyc, grid);
    run<7, 8>("MFMA + ds_read 1:1, tower address pattern", w, out, cyc, grid);
    run<7, 7>("MFMA + ds_read 1:1, tower address pattern", w, out, cyc, grid);
    run<8, 8>("... swizzle candidate A", w, out, cyc, grid);
    run<9, 8>("... swizzle candidate B", w, out, cyc, grid);
    return 0;
}
