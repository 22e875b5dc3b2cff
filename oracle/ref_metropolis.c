/*
 * TEST INFRASTRUCTURE — CPU restatement (plain C) of the build's checkerboard Metropolis
 * generator.  Only tests/ may load this library; the product never links or calls it.
 *
 * The reference's generator (b_pure_metoropolis_2d_ising.py:20-39, :63-105) flips one randomly
 * chosen spin at a time with numpy's global RNG, which cannot be reproduced off the interpreter;
 * the GPU generator keeps its acceptance rule (:35-38: flip when u < exp(-beta*dE),
 * dE = 2 s (J * sum_nn + h), periodic boundaries) but sweeps the two checkerboard colours in turn
 * with counter-based uniforms.  Same stationary distribution, different chain: parity with the
 * reference is statistical (Onsager energy / magnetisation), parity with this file is bit-exact.
 *
 * Definition shared with the kernel: colour c of sweep w updates the sites with (row + col) % 2 == c;
 * the k-th such site in row-major order draws component k % 4 of Philox4x32-10 with counter
 * (k / 4, chain lo, chain hi, 2*w + c) and the seed as key, u = ((r >> 9) + 0.5) * 2^-23; it flips
 * when u < table[(s+1)/2][(nn+4)/2], the float32 acceptance table built by the caller.
 */
#include <stdint.h>
#include <stddef.h>

void ref_philox4x32(const uint32_t c[4], const uint32_t k[2], uint32_t out[4]);

void ref_metropolis(int8_t *spins, int B, int L, const float *table /* [2][5] */, int n_sweeps, uint64_t seed,
                    int64_t first_global_index, int sweep0)
{
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        int8_t *x = spins + (size_t)b * L * L;
        const int64_t g = first_global_index + b;
        const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
        const int half = L / 2;
        for (int w = sweep0; w < sweep0 + n_sweeps; ++w)
            for (int c = 0; c < 2; ++c)
                for (int k = 0; k < L * half; ++k) {
                    const int r = k / half, col = 2 * (k % half) + ((r + c) & 1);
                    uint32_t ctr[4] = {(uint32_t)(k >> 2), (uint32_t)g, (uint32_t)((uint64_t)g >> 32), (uint32_t)(2 * w + c)}, out[4];
                    ref_philox4x32(ctr, key, out);
                    const float u = ((float)(out[k & 3] >> 9) + 0.5f) * 1.1920928955078125e-07f;
                    const int s = x[r * L + col];
                    const int nn = x[((r + 1) % L) * L + col] + x[((r + L - 1) % L) * L + col] + x[r * L + (col + 1) % L] +
                                   x[r * L + (col + L - 1) % L];
                    if (u < table[((s + 1) >> 1) * 5 + ((nn + 4) >> 1)]) x[r * L + col] = (int8_t)-s;
                }
    }
}
