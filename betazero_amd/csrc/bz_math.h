// bz_math.h -- deterministic float helpers and the synthetic evaluator / RNG of
// the MCTS spec.  Every float expression is a sequence of single
// IEEE-754 binary32 operations in a fixed order; translation units including this
// header are compiled with -ffp-contract=off so that nothing is fused.
#pragma once
#include "bz_rules.h"

namespace bz {

BZ_HD float f_from_bits(u32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(b);
#else
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
#endif
}

BZ_HD float fdiv(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fdiv_rn(a, b);
#else
    return a / b;
#endif
}
BZ_HD float fsqrt(float a) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fsqrt_rn(a);
#else
    return __builtin_sqrtf(a);
#endif
}

// exp(x), x <= 0.  mul and add are separate roundings.
BZ_HD float expf_spec(float x) {
    if (x < -87.0f) return 0.0f;
    float t = x * 1.44269504f;
    float n = __builtin_floorf(t + 0.5f);
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    float rr = r * r;
    p = p * rr;
    p = p + r;
    p = p + 1.0f;
    float scale = f_from_bits((u32)((int)n + 127) << 23);
    return p * scale;
}

BZ_HD float tanhf_spec(float x) {
    float a = __builtin_fabsf(x);
    float e = expf_spec(a * -2.0f);
    float t = fdiv(1.0f - e, 1.0f + e);
    return x < 0.0f ? -t : t;
}

BZ_HD u64 mix64(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33; return x;
}
BZ_HD u64 hash_pos(u64 own, u64 opp) {
    return mix64(own * 0x9E3779B97F4A7C15ULL ^ mix64(opp + 0x632BE59BD9B4E019ULL));
}
BZ_HD float hash_logit(u64 h, int a) {
    u64 q = mix64(h + (u64)a * 0xD6E8FEB86659FD93ULL);
    return (float)((int)(q >> 40) - (1 << 23)) * (1.0f / 4194304.0f);
}
BZ_HD float hash_value(u64 h) {
    u64 q = mix64(h ^ 0xA5A5A5A5A5A5A5A5ULL);
    return (float)((int)(q >> 40) - (1 << 23)) * (1.0f / 8388608.0f);
}
BZ_HD u64 rng_draw(u64 seed, u64 game_id, u64 ply) {
    u64 h = mix64(seed * 0x9E3779B97F4A7C15ULL + game_id);
    return mix64(h ^ (ply * 0xBF58476D1CE4E5B9ULL + 0x94D049BB133111EBULL));
}

}  // namespace bz
