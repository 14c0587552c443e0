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

// ---- Dirichlet root noise (opt-in).  Everything is float32 with one rounding per
// operation in the written order, so the C oracle, the numpy twin and the kernels agree bit for bit.
BZ_HD u32 f_to_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    u32 b;
    __builtin_memcpy(&b, &f, 4);
    return b;
#endif
}
// ln(x) for normal x > 0 (Cephes logf: m in [sqrt(1/2), sqrt(2)), degree-8 polynomial in m - 1)
BZ_HD float logf_spec(float x) {
    u32 b = f_to_bits(x);
    int e = (int)(b >> 23) - 127;
    float m = f_from_bits((b & 0x007FFFFFu) | 0x3F800000u);  // [1, 2)
    if (m > 1.41421356f) { m = m * 0.5f; e = e + 1; }
    float z = m - 1.0f;
    float p = 7.0376836292e-2f;
    p = p * z + -1.1514610310e-1f;
    p = p * z + 1.1676998740e-1f;
    p = p * z + -1.2420140846e-1f;
    p = p * z + 1.4249322787e-1f;
    p = p * z + -1.6668057665e-1f;
    p = p * z + 2.0000714765e-1f;
    p = p * z + -2.4999993993e-1f;
    p = p * z + 3.3333331174e-1f;
    float zz = z * z;
    float y = z * zz;
    y = y * p;
    float fe = (float)e;
    float t = fe * -2.12194440e-4f;
    y = y + t;
    t = 0.5f * zz;
    y = y - t;
    float r = z + y;
    t = fe * 0.693359375f;
    r = r + t;
    return r;
}
// uniform in (0, 1): 23 random bits + 1/2, exact in float32
BZ_HD float u01_spec(u64 bits) { return ((float)(u32)(bits >> 41) + 0.5f) * (1.0f / 8388608.0f); }
BZ_HD u64 rng_noise(u64 seed, u64 game_id, u64 ply, u64 idx) {
    u64 h = rng_draw(seed ^ 0xD1B54A32D192ED03ULL, game_id, ply);
    return mix64(h + idx * 0x9E3779B97F4A7C15ULL);
}
// Gamma(alpha, 1) variate for 0 < alpha <= 1 and root edge number `edge` (ascending action order).
// alpha = 1: exponential.  alpha < 1: Johnk's generator -- X = U^(1/alpha), Y = V^(1/(1-alpha)), accept when
// X + Y <= 1, return E X / (X + Y) with E exponential -- at most 16 attempts (acceptance > 0.78 per attempt;
// the 16th attempt's values are used regardless).
BZ_HD float gamma_spec(float alpha, u64 seed, u64 game_id, u64 ply, int edge) {
    const u64 base = (u64)edge * 64ULL;
    if (!(alpha < 1.0f)) return -logf_spec(u01_spec(rng_noise(seed, game_id, ply, base)));
    const float ia = fdiv(1.0f, alpha), ib = fdiv(1.0f, 1.0f - alpha);
    float x = 0.0f, s = 0.0f;
    u64 k = base;
    for (int t = 0; t < 16; ++t) {
        k = base + 3ULL * (u64)t;
        float lu = logf_spec(u01_spec(rng_noise(seed, game_id, ply, k)));
        float lv = logf_spec(u01_spec(rng_noise(seed, game_id, ply, k + 1)));
        lu = lu * ia;
        lv = lv * ib;
        x = expf_spec(lu < 0.0f ? lu : 0.0f);
        float y = expf_spec(lv < 0.0f ? lv : 0.0f);
        s = x + y;
        if (s <= 1.0f) break;
    }
    if (!(s > 0.0f)) return 0.0f;
    float e = -logf_spec(u01_spec(rng_noise(seed, game_id, ply, k + 2)));
    float g = e * x;
    return fdiv(g, s);
}

}  // namespace bz
