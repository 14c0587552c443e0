// probe_lds_rate.hip -- LDS read throughput per CU for the read forms the kernels use: cycles per wave-instruction of
// ds_read_b64_tr_b16 (the weight-gradient kernel's operand reads), ds_read_b64 and ds_read_b128 (the tower's), with 4 and 8
// waves of one workgroup issuing nothing else.  128 B/clk would be 4 cycles per b64 wave-read and 8 per b128.  Addresses:
// the wgrad kernel's pattern for the transpose read (lane 4q + p of a 16-lane group: cell q of a block, channels 4p..),
// consecutive 8 / 16 bytes per lane for the plain reads (conflict-free).
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe/probe_lds_rate tools/probe/probe_lds_rate.hip && tools/probe/probe_lds_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#define LDSP(T, p) ((__attribute__((address_space(3))) T*)(p))

template <int MODE>
__global__ void k(unsigned long long* cycles, unsigned* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int off;
    if (MODE == 0) {   // transpose read: group g of 16 lanes, lane 4q + pp -> cell q (256-B cells, 64-B piece per group swizzled by cell), channels 4pp..
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        off = w * 4096 + q * 256 + ((g ^ q) << 6) + 8 * pp;
    } else if (MODE == 1) off = w * 4096 + lane * 8;
    else off = w * 4096 + lane * 16;
    unsigned acc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(off));   // (the addresses are loop-invariant: keep the compiler from hoisting the plain reads)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int a = off + (u << 10);         // sixteen different 1-KB windows: independent reads, nothing to merge
            if (MODE == 0) { s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(s16x4, smem + a)); acc += (unsigned)v[0] + (unsigned)v[3]; }
            else if (MODE == 1) { u32x2 v = *LDSP(u32x2, smem + a); acc += v[0] + v[1]; }
            else { u32x4 v = *LDSP(u32x4, smem + a); acc += v[0] + v[3]; }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (lane == 0) cycles[blockIdx.x * 16 + w] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    unsigned long long *d, h[16 * 4];
    unsigned* sink;
    if (hipMalloc(&d, sizeof(h)) != hipSuccess || hipMalloc(&sink, 4 * 512 * 4) != hipSuccess) { printf("no device\n"); return 2; }
    const int iters = 2000;
    const char* names[3] = {"ds_read_b64_tr_b16", "ds_read_b64", "ds_read_b128"};
    for (int waves : {4, 8})
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * waves), 65536, 0, d, sink, iters);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves), 65536, 0, d, sink, iters);
                else hipLaunchKernelGGL(k<2>, dim3(1), dim3(64 * waves), 65536, 0, d, sink, iters);
            }
            if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 2; }
            double worst = 0;
            for (int w = 0; w < waves; ++w) worst = h[w] > worst ? (double)h[w] : worst;
            const double per_cu = worst / ((double)iters * 16 * waves);   // CU cycles per wave-instruction when all waves stream
            printf("%-20s %d waves: %.2f cycles per wave-instruction per CU (%.0f B/clk)\n", names[mode], waves, per_cu,
                   (mode == 2 ? 1024.0 : 512.0) / per_cu);
        }
    return 0;
}
