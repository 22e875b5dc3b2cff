/*
 * TEST INFRASTRUCTURE — CPU restatement (plain C) of the reference's dense spectral score model
 * and its sampler.  Only tests/ may load this library; the product never links or calls it.
 *
 * PARITY UNPINNED by the reference's own tests (it has none; Keras/TensorFlow are not installed
 * here).  Pinned by the saved weights models/b_diffusion_model_2d_ising.keras, the
 * trained-objective known answer in tests/test_dense_oracle.py and a float64 PyTorch restatement.
 *
 * Follows b_diffusion_model_fft_2d_ising.py:
 *   score_model      :105-114  Sequential: Dense(440,tanh) x4, Dense(440,linear) on [x | t_norm]
 *   diffusion_backward :19-38  x <- (1 + 0.5 b)x + b*score + noise, input xt = concat([x, t/T])
 *
 * CANONICAL ARITHMETIC as in ref_cpu.c: one fmaf chain from +0 over ascending input index (the
 * appended t_norm is the last input), bias after; tanh(z) = copysign((1-e)/(1+e), z) with
 * e = cexp(-2|z|); update with every operation rounded separately.  Compile with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

float ref_cexp(float x);
void ref_update(float *x, const float *score, const float *z, float beta, size_t n);

float ref_ctanh(float z)
{
    const float a = fabsf(z);
    const float e = ref_cexp(-2.0f * a);
    const float r = (1.0f - e) / (1.0f + e);
    return copysignf(r, z);
}

/* w[i]: (in_i, out_i) Keras layout; act[i]: 1 = tanh, 0 = linear.  x (B, n) with n = in_0 - 1. */
void ref_mlp_forward(int n_layers, const int *in_dim, const int *out_dim, const int *act, const float *const *w,
                     const float *const *b, const float *x, float t_norm, float *score, int B)
{
    const int n = in_dim[0] - 1;
    int wmax = in_dim[0];
    for (int i = 0; i < n_layers; ++i) if (out_dim[i] > wmax) wmax = out_dim[i];
#pragma omp parallel for schedule(static)
    for (int s = 0; s < B; ++s) {
        float *u = (float *)malloc(2 * (size_t)wmax * sizeof(float)), *v = u + wmax;
        memcpy(u, x + (size_t)s * n, n * sizeof(float));
        u[n] = t_norm;
        for (int i = 0; i < n_layers; ++i) {
            for (int j = 0; j < out_dim[i]; ++j) {
                float acc = 0.0f;
                for (int k = 0; k < in_dim[i]; ++k) acc = fmaf(u[k], w[i][(size_t)k * out_dim[i] + j], acc);
                acc = acc + b[i][j];
                v[j] = act[i] ? ref_ctanh(acc) : acc;
            }
            float *t = u; u = v; v = t;
        }
        memcpy(score + (size_t)s * n, u, n * sizeof(float));
        free(u < v ? u : v);
    }
}

/* diffusion_backward (:19-38) with injected standard normals noise_z (T,B,n) indexed by t (NULL:
 * drift only).  traj (optional): (T+1,B,n) in the reference's return order: traj[T] = x_T,
 * traj[t] = state after step t, so traj[0] is the final state. */
void ref_mlp_sample(int n_layers, const int *in_dim, const int *out_dim, const int *act, const float *const *w,
                    const float *const *b, float *x, const float *betas, int T, const float *noise_z, float *traj, int B)
{
    const size_t n = (size_t)B * (in_dim[0] - 1);
    float *score = (float *)malloc(n * sizeof(float));
    if (traj) memcpy(traj + (size_t)T * n, x, n * sizeof(float));
    for (int t = T - 1; t >= 0; --t) {
        ref_mlp_forward(n_layers, in_dim, out_dim, act, w, b, x, (float)t / (float)T, score, B);
        ref_update(x, score, noise_z ? noise_z + (size_t)t * n : NULL, betas[t], n);
        if (traj) memcpy(traj + (size_t)t * n, x, n * sizeof(float));
    }
    free(score);
}
