// Device helpers shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DEVI __device__ __forceinline__

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11) keyed by the seed; counter = (quad index within the
// lattice, global lattice index low, global lattice index high, step t).  Four outputs give four
// standard normals (two Box-Muller pairs) for four consecutive elements of one lattice, so a
// draw depends only on (seed, global lattice, element, t) and never on batch size or sharding.
// ---------------------------------------------------------------------------------------------
DEVI void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Uniform in (0,1) from the top 23 bits: (k + 1/2) * 2^-23 with k < 2^23 is exactly representable in float32
// (24 significant bits), so the value never rounds to 0 or 1 and every step of the grid is equally likely.
DEVI float u01(uint32_t r) { return ((float)(r >> 9) + 0.5f) * 1.1920928955078125e-07f; }

DEVI void normal4(uint64_t seed, int64_t lattice, uint32_t quad, int32_t t, float z[4])
{
    uint32_t r[4];
    philox4x32_10(quad, (uint32_t)lattice, (uint32_t)((uint64_t)lattice >> 32), (uint32_t)t, (uint32_t)seed,
                  (uint32_t)(seed >> 32), r);
    // Box-Muller on the hardware transcendentals: v_log_f32 is log2, v_sin/v_cos take the angle in
    // revolutions (so sin(2*pi*u) is one instruction); absolute error of a normal ~1e-6.
    const float nl2 = -1.3862943611198906f;  // -2 ln 2
    const float m0 = __builtin_amdgcn_sqrtf(nl2 * __builtin_amdgcn_logf(u01(r[0])));
    const float m1 = __builtin_amdgcn_sqrtf(nl2 * __builtin_amdgcn_logf(u01(r[2])));
    const float a0 = u01(r[1]), a1 = u01(r[3]);
    const float s0 = __builtin_amdgcn_sinf(a0), c0 = __builtin_amdgcn_cosf(a0);
    const float s1 = __builtin_amdgcn_sinf(a1), c1 = __builtin_amdgcn_cosf(a1);
    z[0] = m0 * c0; z[1] = m0 * s0; z[2] = m1 * c1; z[3] = m1 * s1;
}

// Reverse update x' = ((c1*x) + (beta*score)) + (z*sd): every product and sum rounded separately (no fused
// multiply-add), whatever the translation unit's contraction setting - the same value in every kernel and build.
DEVI float reverse_update(float c1, float x, float beta, float score, float z, float sd, bool add_noise)
{
#pragma clang fp contract(off)
    float v = c1 * x + beta * score;
    if (add_noise) v = v + z * sd;
    return v;
}
