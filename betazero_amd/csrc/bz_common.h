// bz_common.h -- error plumbing shared by the translation units of libbz_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/bz_abi.h"

#define BZ_EXPORT extern "C" __attribute__((visibility("default")))

namespace bz {
void set_error(const char* fmt, ...);
inline int32_t hip_fail(hipError_t e, const char* what) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return BZ_EHIP;
}
// hipFuncAttributeMaxDynamicSharedMemorySize belongs to a (kernel, DEVICE) pair: a process that runs the kernel on a second
// GPU must set it there too.  `done` = one static word per call site, bit = device ordinal (devices >= 32: set every time).
inline hipError_t lds_attr_per_device(const void* kernel, int bytes, unsigned* done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 32 && ((*done >> dev) & 1u)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev >= 0 && dev < 32) *done |= 1u << dev;
    return e;
}
// kernel timers (bz_env.hip); no-ops unless bz_profile_enable(1)
int prof_begin(int slot, hipStream_t s);
void prof_end(int slot, int idx, hipStream_t s);
struct ProfScope {
    int slot, idx; hipStream_t s;
    ProfScope(int slot_, void* stream) : slot(slot_), s((hipStream_t)stream) { idx = prof_begin(slot, s); }
    ~ProfScope() { prof_end(slot, idx, s); }
};
}  // namespace bz

// bz_net.hip: which parameter upload the net's weights are from (changes with every bz_net_create / bz_net_update)
uint64_t bz_net_epoch(const bz_net* net);
// bz_net.hip: forward over the first *n_dev (device counter, <= max_n) positions; n_dev may be null
int32_t bz_net_forward_dev(bz_net* net, int bf16, const uint64_t* own, const uint64_t* opp, int32_t max_n,
                           const uint32_t* n_dev, float* logits, float* value, void* stream);

#define BZ_HIP(call)                                             \
    do {                                                         \
        hipError_t _e = (call);                                  \
        if (_e != hipSuccess) return bz::hip_fail(_e, #call);    \
    } while (0)
#define BZ_LAUNCH_CHECK(name)                                    \
    do {                                                         \
        hipError_t _e = hipGetLastError();                       \
        if (_e != hipSuccess) return bz::hip_fail(_e, name);     \
    } while (0)
#define BZ_REQUIRE(cond, msg)                                    \
    do {                                                         \
        if (!(cond)) { bz::set_error("%s", msg); return BZ_EINVAL; } \
    } while (0)
