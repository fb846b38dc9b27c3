/*
 * css_synth.h -- counter-based synthetic data generator shared by the HIP
 * library, the CPU oracle and (restated in numpy) the Python test helpers.
 *
 * SURVEY.md 8(d) asks for a counter-based generator so that any row of any
 * synthetic index / weight tensor can be regenerated bit-identically on either
 * box.  Transcendentals (Box-Muller) differ in the last ulp between libm,
 * numpy and the GPU, so the "normal" variate here is an Irwin-Hall sum of the
 * four 16-bit lanes of one 64-bit hash: integer arithmetic plus one correctly
 * rounded fp32 multiply, hence bit-identical everywhere.
 *
 * value(seed, idx) ~ approx N(0,1)   (sum of 4 U{0..65535}, centred, scaled)
 */
#ifndef CSS_SYNTH_H
#define CSS_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define CSS_HD __host__ __device__ __forceinline__
#else
#define CSS_HD static inline
#endif

/* splitmix64 finaliser over (seed + golden * (idx + 1)) */
CSS_HD uint64_t css_mix64(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* 1 / sqrt(4 * (65536^2 - 1) / 12) rounded to fp32: 0x37DDB5BB = 2.6429e-05f */
#define CSS_SYNTH_SCALE 2.64290629e-05f

CSS_HD float css_synth_normal(uint64_t seed, uint64_t idx) {
    uint64_t h = css_mix64(seed, idx);
    int s = (int)(h & 0xFFFFu) + (int)((h >> 16) & 0xFFFFu) +
            (int)((h >> 32) & 0xFFFFu) + (int)(h >> 48) - 131070;
    return (float)s * CSS_SYNTH_SCALE;
}

/* uniform integer in [lo, hi) (hi - lo < 2^32) */
CSS_HD uint32_t css_synth_uint(uint64_t seed, uint64_t idx, uint32_t lo, uint32_t hi) {
    uint64_t h = css_mix64(seed, idx);
    return lo + (uint32_t)(((h >> 32) * (uint64_t)(hi - lo)) >> 32);
}

/* Tensor ids for the synthetic MPNet weights (seed' = seed ^ (tensor_id << 40)).
 * Layout of ids: see DESIGN.md "synthetic weights". */
CSS_HD uint64_t css_synth_tensor_seed(uint64_t seed, uint32_t tensor_id) {
    return seed ^ ((uint64_t)tensor_id << 40) ^ 0xC55E7E11ull;
}

#endif /* CSS_SYNTH_H */
