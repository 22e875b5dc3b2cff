// Float32 kernels with a fixed ("canonical") arithmetic order, plus the small elementwise,
// noise and observable kernels shared by both precisions.
//
// Canonical order: every contraction is one float32 fma chain from +0 over ascending
// k = (ky*3 + kx)*Cin + c, bias added afterwards; swish(y) = y / (1 + cexp(-y)) with the
// polynomial cexp below; pooling (((a+b)+c)+d)*0.25; sampler update with every operation
// rounded separately.  v_mfma_f32_32x32x2_f32 computes exactly such a chain (k = 0 then
// k = 1 per instruction), so the matrix-core path reproduces a scalar fmaf loop bit for bit.
// This file must be compiled with -ffp-contract=off.
#include "idm_internal.h"
#include "idm_device.h"
#include "idm_canon_math.h"

#include <math.h>

#include <algorithm>

DEVI float epilogue(float acc, int co, const float *bias, const float *scale, const float *shift, int silu,
                    const float *tadd)
{
    float z = __fadd_rn(acc, bias[co]);
    if (scale) { z = __fmul_rn(z, scale[co]); z = __fadd_rn(z, shift[co]); }
    if (silu) z = csilu(z);
    if (tadd) z = __fadd_rn(z, tadd[co]);
    return z;
}

// ---------------------------------------------------------------------------------------------
// 3x3 conv, float32, v_mfma_f32_32x32x2_f32.  One wave = 32 pixels x all Cout (MT tiles of 32).
// Operand fetch straight from global memory (this is the parity path, not the fast one).
// Weights are packed [tap][c/8][mt][lane][4]: element s of lane l is W[tap][8*(c/8)+2s+(l>>5)][32mt+(l&31)].
// ---------------------------------------------------------------------------------------------
template <int MT, int MODE>
__global__ __launch_bounds__(256) void conv3x3_f32_kernel(ConvArgs a)
{
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t npix = a.B * a.H * a.W;
    if (tile * 32 >= npix) return;
    const int64_t p = tile * 32 + (lane & 31);
    const bool pv = p < npix;
    const int64_t pc = pv ? p : 0;
    const int x = (int)(pc % a.W);
    const int y = (int)((pc / a.W) % a.H);
    const int64_t b = pc / ((int64_t)a.W * a.H);
    const int h = lane >> 5;
    const int Cin = a.C0 + a.C1;
    const int ncg = Cin >> 3;
    const float *in0 = (const float *)a.in0;
    const float *in1 = (const float *)a.in1;
    const f32x4 *wp = (const f32x4 *)a.wpack + lane;

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;

    for (int tap = a.tap_lo; tap < a.tap_hi; ++tap) {
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        const bool inb = pv && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
        for (int cg = 0; cg < ncg; ++cg) {
            const int c = cg * 8;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.0f;
            if (inb) {
                if (MODE == CONV_PLAIN) {
                    const f32x4 *q = (const f32x4 *)(in0 + ((b * a.H + yy) * a.W + xx) * a.C0 + c);
                    const f32x4 q0 = q[0], q1 = q[1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = q0[j]; v[4 + j] = q1[j]; }
                } else if (MODE == CONV_POOL) {
                    const int H2 = 2 * a.H, W2 = 2 * a.W;
                    const float *q = in0 + ((b * H2 + 2 * yy) * W2 + 2 * xx) * a.C0 + c;
                    const f32x4 *q00 = (const f32x4 *)q, *q01 = (const f32x4 *)(q + a.C0);
                    const f32x4 *q10 = (const f32x4 *)(q + (int64_t)W2 * a.C0), *q11 = (const f32x4 *)(q + (int64_t)W2 * a.C0 + a.C0);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const f32x4 t0 = q00[u], t1 = q01[u], t2 = q10[u], t3 = q11[u];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            v[4 * u + j] = __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(t0[j], t1[j]), t2[j]), t3[j]), 0.25f);
                    }
                } else {
                    const f32x4 *q;
                    if (c < a.C0) {
                        const int Hh = a.H >> 1, Wh = a.W >> 1;
                        q = (const f32x4 *)(in0 + ((b * Hh + (yy >> 1)) * Wh + (xx >> 1)) * a.C0 + c);
                    } else {
                        q = (const f32x4 *)(in1 + ((b * a.H + yy) * a.W + xx) * a.C1 + (c - a.C0));
                    }
                    const f32x4 q0 = q[0], q1 = q[1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = q0[j]; v[4 + j] = q1[j]; }
                }
            }
            const f32x4 *w = wp + (int64_t)((tap - a.tap_lo) * ncg + cg) * MT * 64;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const f32x4 wv = w[m * 64];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float bv = h ? v[2 * s + 1] : v[2 * s];
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[s], bv, acc[m], 0, 0, 0);
                }
            }
        }
    }
    if (!pv) return;
    float *out = (float *)a.out + p * a.Cout;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float z = epilogue(acc[m][r], co, a.bias, a.scale, a.shift, a.silu, a.tadd);
            if (a.resid) z = __fadd_rn(z, a.resid[p * a.Cout + co]);
            out[co] = z;
        }
}

template <int MT>
static void launch_conv_f32_mt(const ConvArgs &a, hipStream_t s)
{
    const int64_t npix = a.B * a.H * a.W;
    const unsigned grid = (unsigned)((npix + 127) / 128);
    switch (a.mode) {
    case CONV_PLAIN: hipLaunchKernelGGL((conv3x3_f32_kernel<MT, CONV_PLAIN>), dim3(grid), dim3(256), 0, s, a); break;
    case CONV_POOL: hipLaunchKernelGGL((conv3x3_f32_kernel<MT, CONV_POOL>), dim3(grid), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((conv3x3_f32_kernel<MT, CONV_UPCAT>), dim3(grid), dim3(256), 0, s, a); break;
    }
}

void launch_conv_f32(const ConvArgs &a, hipStream_t s)
{
    switch (a.Cout / 32) {
    case 1: launch_conv_f32_mt<1>(a, s); break;
    case 2: launch_conv_f32_mt<2>(a, s); break;
    default: launch_conv_f32_mt<4>(a, s); break;
    }
}

// ---------------------------------------------------------------------------------------------
// First layer: Cin = C in {1,2}, K = 9C is too thin for matrix cores.  One thread per pixel,
// canonical chain, reads the unpadded state (B,H,W,C) and writes the padded map (B,Hp,Wp,32).
// ---------------------------------------------------------------------------------------------
template <typename OutT>
__global__ __launch_bounds__(256) void conv_first_kernel(const float *__restrict__ x, int64_t B, int H, int W, int Hp,
                                                         int Wp, int C, const float *__restrict__ w,
                                                         const float *__restrict__ bias, const float *__restrict__ scale,
                                                         const float *__restrict__ shift, OutT *__restrict__ out)
{
    __shared__ float ws[9 * 2 * 32];
    __shared__ float bs[3 * 32];
    for (int i = threadIdx.x; i < 9 * C * 32; i += 256) ws[i] = w[i];
    if (threadIdx.x < 32) {
        bs[threadIdx.x] = bias[threadIdx.x];
        bs[32 + threadIdx.x] = scale ? scale[threadIdx.x] : 1.0f;
        bs[64 + threadIdx.x] = shift ? shift[threadIdx.x] : 0.0f;
    }
    __syncthreads();
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= B * Hp * Wp) return;
    const int xo = (int)(p % Wp), yo = (int)((p / Wp) % Hp);
    const int64_t b = p / ((int64_t)Wp * Hp);
    float acc[32];
#pragma unroll
    for (int f = 0; f < 32; ++f) acc[f] = 0.0f;
    for (int tap = 0; tap < 9; ++tap) {
        const int yy = yo + tap / 3 - 1, xx = xo + tap % 3 - 1;
        // taps outside the padded map contribute nothing; taps in the zero padding contribute 0*w
        if (yy < 0 || yy >= Hp || xx < 0 || xx >= Wp) continue;
        for (int c = 0; c < C; ++c) {
            const float v = (yy < H && xx < W) ? x[((b * H + yy) * W + xx) * C + c] : 0.0f;
            const float *wr = ws + (tap * C + c) * 32;
#pragma unroll
            for (int f = 0; f < 32; ++f) acc[f] = __fmaf_rn(v, wr[f], acc[f]);
        }
    }
    OutT *o = out + p * 32;
#pragma unroll
    for (int f = 0; f < 32; ++f) {
        float z = __fadd_rn(acc[f], bs[f]);
        if (scale) { z = __fmul_rn(z, bs[32 + f]); z = __fadd_rn(z, bs[64 + f]); }
        o[f] = (OutT)csilu(z);
    }
}

void launch_conv_first(const float *x, int64_t B, int H, int W, int Hp, int Wp, int C, const float *w,
                       const float *bias, const float *scale, const float *shift, void *out, int F, bool out_bf16,
                       hipStream_t s)
{
    (void)F;
    const unsigned grid = (unsigned)((B * Hp * Wp + 255) / 256);
    if (out_bf16)
        hipLaunchKernelGGL(conv_first_kernel<__bf16>, dim3(grid), dim3(256), 0, s, x, B, H, W, Hp, Wp, C, w, bias, scale,
                           shift, (__bf16 *)out);
    else
        hipLaunchKernelGGL(conv_first_kernel<float>, dim3(grid), dim3(256), 0, s, x, B, H, W, Hp, Wp, C, w, bias, scale,
                           shift, (float *)out);
}

// Head: 1x1 conv F -> C (linear) on the float32 output of the last 3x3 layer, cropped to (H,W).
template <typename InT>
__global__ __launch_bounds__(256) void head_kernel(const InT *__restrict__ a9, int64_t B, int H, int W, int Hp, int Wp,
                                                   int F, int C, const float *__restrict__ w,
                                                   const float *__restrict__ bias, float *__restrict__ score)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= B * H * W) return;
    const int x = (int)(p % W), y = (int)((p / W) % H);
    const int64_t b = p / ((int64_t)W * H);
    const InT *ip = a9 + ((b * Hp + y) * Wp + x) * F;
    // 16-byte loads of the pixel's channel vector; the chain still runs over ascending channel index
    constexpr int V = 16 / sizeof(InT);
    typedef InT vec_t __attribute__((ext_vector_type(V)));
    float acc[2] = {0.0f, 0.0f};
    for (int f0 = 0; f0 < F; f0 += V) {
        const vec_t v = *(const vec_t *)(ip + f0);
#pragma unroll
        for (int j = 0; j < V; ++j)
            for (int ch = 0; ch < C; ++ch) acc[ch] = __fmaf_rn((float)v[j], w[(f0 + j) * C + ch], acc[ch]);
    }
    for (int ch = 0; ch < C; ++ch) score[p * C + ch] = __fadd_rn(acc[ch], bias[ch]);
}

void launch_head(const float *a9, int64_t B, int H, int W, int Hp, int Wp, int F, int C, const float *w,
                 const float *bias, float *score, hipStream_t s)
{
    const unsigned grid = (unsigned)((B * H * W + 255) / 256);
    hipLaunchKernelGGL(head_kernel<float>, dim3(grid), dim3(256), 0, s, a9, B, H, W, Hp, Wp, F, C, w, bias, score);
}

void launch_head_bf16(const void *a9, int64_t B, int H, int W, int Hp, int Wp, int F, int C, const float *w,
                      const float *bias, float *score, hipStream_t s)
{
    const unsigned grid = (unsigned)((B * H * W + 255) / 256);
    hipLaunchKernelGGL(head_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16 *)a9, B, H, W, Hp, Wp, F, C, w, bias,
                       score);
}

// x <- ((c1*x) + (beta*score)) + z*sqrt(beta).  One thread per quad of four consecutive elements.
__global__ __launch_bounds__(256) void update_kernel(float *__restrict__ x, const float *__restrict__ score,
                                                     const float *__restrict__ zin, float c1, float beta, float sd,
                                                     int64_t n_lattices, int64_t elems, int add_noise, PhiloxArgs ph)
{
    const int64_t nq = (elems + 3) >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_lattices * nq) return;
    const int64_t b = i / nq;
    const uint32_t q = (uint32_t)(i % nq);
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (add_noise && !zin) normal4(ph.seed, ph.first_global_index + b, q, ph.t, z);
    const int64_t base = b * elems + (int64_t)q * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if ((int64_t)q * 4 + j >= elems) break;
        const int64_t e = base + j;
        float v = __fadd_rn(__fmul_rn(c1, x[e]), __fmul_rn(beta, score[e]));
        if (add_noise) v = __fadd_rn(v, __fmul_rn(zin ? zin[e] : z[j], sd));
        x[e] = v;
    }
}

// Same update for elems % 4 == 0 and 16-byte aligned tensors: one float4 per operand per thread.
__global__ __launch_bounds__(256) void update4_kernel(float *__restrict__ x, const float *__restrict__ score,
                                                      const float *__restrict__ zin, float c1, float beta, float sd,
                                                      int64_t n_lattices, int64_t nq, int add_noise, PhiloxArgs ph)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_lattices * nq) return;
    f32x4 xv = ((const f32x4 *)x)[i];
    const f32x4 sv = ((const f32x4 *)score)[i];
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (add_noise) {
        if (zin) {
            const f32x4 zv = ((const f32x4 *)zin)[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] = zv[j];
        } else {
            const int64_t b = i / nq;
            normal4(ph.seed, ph.first_global_index + b, (uint32_t)(i - b * nq), ph.t, z);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = __fadd_rn(__fmul_rn(c1, xv[j]), __fmul_rn(beta, sv[j]));
        if (add_noise) v = __fadd_rn(v, __fmul_rn(z[j], sd));
        xv[j] = v;
    }
    ((f32x4 *)x)[i] = xv;
}

void launch_update(float *x, const float *score, const float *z, float c1, float beta, float sd, int64_t n_lattices, int64_t elems,
                   bool add_noise, const PhiloxArgs &ph, hipStream_t s)
{
    const int64_t nq = (elems + 3) >> 2;
    const unsigned grid = (unsigned)((n_lattices * nq + 255) / 256);
    // (c1, beta = the score coefficient, sd: rounded on the host, so that they do not depend on the device's
    // square-root implementation)
    const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(score) | reinterpret_cast<uintptr_t>(z);
    if ((elems & 3) == 0 && (al & 15) == 0) {
        hipLaunchKernelGGL(update4_kernel, dim3(grid), dim3(256), 0, s, x, score, z, (float)c1, beta, (float)sd, n_lattices, nq,
                           add_noise ? 1 : 0, ph);
        return;
    }
    hipLaunchKernelGGL(update_kernel, dim3(grid), dim3(256), 0, s, x, score, z, (float)c1, beta, (float)sd, n_lattices,
                       elems, add_noise ? 1 : 0, ph);
}

__global__ __launch_bounds__(256) void philox_normal_kernel(float *__restrict__ out, int64_t n_lattices, int64_t elems,
                                                            PhiloxArgs ph)
{
    const int64_t nq = (elems + 3) >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_lattices * nq) return;
    const int64_t b = i / nq;
    const uint32_t q = (uint32_t)(i % nq);
    float z[4];
    normal4(ph.seed, ph.first_global_index + b, q, ph.t, z);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if ((int64_t)q * 4 + j < elems) out[b * elems + (int64_t)q * 4 + j] = z[j];
}

void launch_philox_normal(float *out, int64_t n_lattices, int64_t elems, const PhiloxArgs &ph, hipStream_t s)
{
    const int64_t nq = (elems + 3) >> 2;
    const unsigned grid = (unsigned)((n_lattices * nq + 255) / 256);
    hipLaunchKernelGGL(philox_normal_kernel, dim3(grid), dim3(256), 0, s, out, n_lattices, elems, ph);
}

// Time-embedding table: row t = Dense2(Dense1(swish(Dense0(t/T)))) — one block per t, canonical chains.
__global__ __launch_bounds__(128) void time_table_kernel(int T, int n, int single, float tn_single,
                                                         const float *__restrict__ w0,
                                                         const float *__restrict__ b0, const float *__restrict__ w1,
                                                         const float *__restrict__ b1, const float *__restrict__ w2,
                                                         const float *__restrict__ b2, float *__restrict__ table)
{
    __shared__ float e0[128], e1[128];
    const int t = blockIdx.x, j = threadIdx.x;
    const float tn = single ? tn_single : __fdiv_rn((float)t, (float)T);
    if (j < n) e0[j] = csilu(__fadd_rn(__fmaf_rn(tn, w0[j], 0.0f), b0[j]));
    __syncthreads();
    if (j < n) {
        float acc = 0.0f;
        for (int i = 0; i < n; ++i) acc = __fmaf_rn(e0[i], w1[i * n + j], acc);
        e1[j] = __fadd_rn(acc, b1[j]);
    }
    __syncthreads();
    if (j < n) {
        float acc = 0.0f;
        for (int i = 0; i < n; ++i) acc = __fmaf_rn(e1[i], w2[i * n + j], acc);
        table[(int64_t)t * n + j] = __fadd_rn(acc, b2[j]);
    }
}

void launch_time_table(int T, int n, const float *w0, const float *b0, const float *w1, const float *b1,
                       const float *w2, const float *b2, float *table, hipStream_t s)
{
    hipLaunchKernelGGL(time_table_kernel, dim3(T), dim3(128), 0, s, T, n, 0, 0.0f, w0, b0, w1, b1, w2, b2, table);
}

void launch_time_row(float t_norm, int n, const float *w0, const float *b0, const float *w1, const float *b1,
                     const float *w2, const float *b2, float *row, hipStream_t s)
{
    hipLaunchKernelGGL(time_table_kernel, dim3(1), dim3(128), 0, s, 1, n, 1, t_norm, w0, b0, w1, b1, w2, b2, row);
}

__global__ __launch_bounds__(256) void sign_project_kernel(const float *__restrict__ x, int8_t *__restrict__ spins, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) spins[i] = x[i] > 0.0f ? 1 : -1;
}

// 16 elements per thread: four float4 loads, one 16-byte store.
__global__ __launch_bounds__(256) void sign_project16_kernel(const float *__restrict__ x, int8_t *__restrict__ spins, int64_t n16)
{
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const f32x4 *p = (const f32x4 *)x + i * 4;
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x4 v = p[k];
        unsigned int w = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) w |= (v[j] > 0.0f ? 0x01u : 0xffu) << (8 * j);
        o[k] = w;
    }
    ((u32x4 *)spins)[i] = o;
}

void launch_sign_project(const float *x, int8_t *spins, int64_t n, hipStream_t s)
{
    if ((n & 15) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(spins)) & 15) == 0) {
        hipLaunchKernelGGL(sign_project16_kernel, dim3((unsigned)((n / 16 + 255) / 256)), dim3(256), 0, s, x, spins, n / 16);
        return;
    }
    hipLaunchKernelGGL(sign_project_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, spins, n);
}

// Observables: one 64-lane wavefront per lattice.  Lanes stride over the sites, each adds
// s, s*s_down and s*s_right (periodic), then a wavefront shuffle reduction; lane 0 writes
// M, S and bumps the histograms / moments with device-scope integer atomics.
__global__ __launch_bounds__(256) void observe_kernel(const int8_t *__restrict__ spins, int64_t B, int L,
                                                      int32_t *__restrict__ M, int32_t *__restrict__ S,
                                                      unsigned long long *__restrict__ histM,
                                                      unsigned long long *__restrict__ histS,
                                                      long long *__restrict__ moments)
{
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int8_t *s = spins + b * L * L;
    int m = 0, bond = 0;
    for (int i = lane; i < L * L; i += 64) {
        const int y = i / L, x = i - y * L;
        const int v = s[i];
        const int d = s[(y + 1 == L ? 0 : y + 1) * L + x];
        const int r = s[y * L + (x + 1 == L ? 0 : x + 1)];
        m += v;
        bond += v * d + v * r;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        m += __shfl_down(m, off, 64);
        bond += __shfl_down(bond, off, 64);
    }
    if (lane == 0) {
        if (M) M[b] = m;
        if (S) S[b] = bond;
        if (histM) atomicAdd(&histM[(m + L * L) / 2], 1ull);
        if (histS) atomicAdd(&histS[(bond + 2 * L * L) / 2], 1ull);
        if (moments) {
            atomicAdd((unsigned long long *)&moments[0], 1ull);
            atomicAdd((unsigned long long *)&moments[1], (unsigned long long)(long long)m);
            atomicAdd((unsigned long long *)&moments[2], (unsigned long long)(long long)(m < 0 ? -m : m));
            atomicAdd((unsigned long long *)&moments[3], (unsigned long long)((long long)m * m));
            atomicAdd((unsigned long long *)&moments[4], (unsigned long long)(long long)bond);
            atomicAdd((unsigned long long *)&moments[5], (unsigned long long)((long long)bond * bond));
        }
    }
}

// Fast path for L % 16 == 0: persistent wavefronts, each lane takes 16 consecutive sites (one 16-byte
// load) and works on sign bits only: for +-1 spins  sum s = n - 2*popc(negative),  sum s*s' = n - 2*popc(s ^ s').
// The right neighbour's first byte comes from the neighbouring lane (row wrap: the row's first lane),
// the row below is a second (cache-resident) load.  Histograms are accumulated in LDS (L <= 64) and
// moments in registers, flushed once per workgroup.
__global__ __launch_bounds__(256) void observe_fast_kernel(const int8_t *__restrict__ spins, int64_t B, int L,
                                                           int32_t *__restrict__ M, int32_t *__restrict__ S,
                                                           unsigned long long *__restrict__ histM,
                                                           unsigned long long *__restrict__ histS,
                                                           long long *__restrict__ moments, int lds_hist)
{
    extern __shared__ unsigned int obs_hist[];   // [L*L+1 | 2*L*L+1] when lds_hist
    __shared__ long long wg_mom[4][6];
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = L * L, nchunk = n >> 4, cpr = L >> 4;   // chunks per lattice / per row
    const int nbM = n + 1, nbS = 2 * n + 1;
    if (lds_hist) {
        for (int i = threadIdx.x; i < nbM + nbS; i += 256) obs_hist[i] = 0u;
        __syncthreads();
    }
    long long mom[6] = {0, 0, 0, 0, 0, 0};
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < B; b += nwaves) {
        const u32x4 *src = (const u32x4 *)(spins + b * n);
        int neg = 0, diff = 0;
        for (int c0 = 0; c0 < nchunk; c0 += 64) {
            const int c = c0 + lane;
            const bool act = c < nchunk;
            const int cc = act ? c : 0;
            const int down = cc + cpr >= nchunk ? cc + cpr - nchunk : cc + cpr;
            const u32x4 v = src[cc], d = src[down];
            // first byte of the next chunk of the same row (wraps to the row's first chunk)
            const int xin = cc % cpr;
            const int src_lane = xin + 1 == cpr ? lane - (cpr - 1) : lane + 1;
            const unsigned int nb = (unsigned int)__shfl((int)v[0], src_lane, 64) & 0xffu;
            if (act) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned int nxt = j < 3 ? v[j + 1] : nb;
                    const unsigned int r = (v[j] >> 8) | (nxt << 24);
                    neg += __popc(v[j] & 0x80808080u);
                    diff += __popc((v[j] ^ d[j]) & 0x80808080u) + __popc((v[j] ^ r) & 0x80808080u);
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            neg += __shfl_down(neg, off, 64);
            diff += __shfl_down(diff, off, 64);
        }
        if (lane == 0) {
            const int m = n - 2 * neg, bond = 2 * n - 2 * diff;
            if (M) M[b] = m;
            if (S) S[b] = bond;
            if (lds_hist) {
                atomicAdd(&obs_hist[(m + n) >> 1], 1u);
                atomicAdd(&obs_hist[nbM + ((bond + 2 * n) >> 1)], 1u);
            } else {
                if (histM) atomicAdd(&histM[(m + n) >> 1], 1ull);
                if (histS) atomicAdd(&histS[(bond + 2 * n) >> 1], 1ull);
            }
            mom[0] += 1; mom[1] += m; mom[2] += m < 0 ? -m : m; mom[3] += (long long)m * m;
            mom[4] += bond; mom[5] += (long long)bond * bond;
        }
    }
    if (lane == 0)
        for (int k = 0; k < 6; ++k) wg_mom[wave][k] = mom[k];
    __syncthreads();
    if (moments && threadIdx.x < 6) {
        const long long t = wg_mom[0][threadIdx.x] + wg_mom[1][threadIdx.x] + wg_mom[2][threadIdx.x] + wg_mom[3][threadIdx.x];
        if (t != 0) atomicAdd((unsigned long long *)&moments[threadIdx.x], (unsigned long long)t);
    }
    if (lds_hist) {
        for (int i = threadIdx.x; i < nbM + nbS; i += 256) {
            const unsigned int v = obs_hist[i];
            if (v == 0u) continue;
            if (i < nbM) { if (histM) atomicAdd(&histM[i], (unsigned long long)v); }
            else if (histS) atomicAdd(&histS[i - nbM], (unsigned long long)v);
        }
    }
}

void launch_observe(const int8_t *spins, int64_t B, int L, int32_t *M, int32_t *S, unsigned long long *histM,
                    unsigned long long *histS, long long *moments, hipStream_t s)
{
    // rows must not straddle a wavefront pass of 64 chunks: L/16 a power of two (L = 16, 32, 64, 128, ...)
    if ((L & 15) == 0 && L <= 1024 && 64 % (L >> 4) == 0 && (reinterpret_cast<uintptr_t>(spins) & 15) == 0) {
        const int lds_hist = L <= 64 ? 1 : 0;
        const size_t lds = lds_hist ? (size_t)(3 * L * L + 2) * sizeof(unsigned int) : 0;
        const int64_t want = (B + 3) / 4;
        const unsigned grid = (unsigned)std::min<int64_t>(want, 1024);
        hipLaunchKernelGGL(observe_fast_kernel, dim3(grid), dim3(256), lds, s, spins, B, L, M, S, histM, histS, moments,
                           lds_hist);
        return;
    }
    hipLaunchKernelGGL(observe_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, spins, B, L, M, S, histM, histS,
                       moments);
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const __bf16 *__restrict__ in, float *__restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

void launch_cast_bf16_to_f32(const void *in, float *out, int64_t n, hipStream_t s)
{
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const __bf16 *)in, out, n);
}

// ---------------------------------------------------------------------------------------------
// irfft2 + sign projection for the spectral model (b_fft.py:12-17 then
// b_diffusion_model_fft_unet_2d_ising.py:214-217).  One workgroup per lattice, everything in LDS,
// float64 arithmetic.  Same order as numpy.fft.irfft2: a complex inverse transform along axis 0
// for each of the L/2+1 stored columns, then a complex-to-real transform along axis 1 that uses
// only the real parts of the k=0 and k=L/2 columns, normalised by 1/L^2.  The transforms are
// direct sums over a cos/sin table (L <= 128: O(L^3) per lattice, once per sample, negligible
// next to the 1000 UNet evaluations).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void irfft2_sign_kernel(const float *__restrict__ spec, int L, int8_t *__restrict__ spins,
                                                          double *__restrict__ real_out)
{
    extern __shared__ __attribute__((aligned(16))) char irf_smem[];
    const int K = L / 2 + 1;
    double *cs = (double *)irf_smem;          // cos(2 pi q / L), q < L
    double *sn = cs + L;                      // sin(2 pi q / L)
    double *xr = sn + L;                      // input  [m][k]
    double *xi = xr + L * K;
    double *yr = xi + L * K;                  // after axis 0 [n][k]
    double *yi = yr + L * K;
    const int64_t b = blockIdx.x;
    const float *src = spec + b * (int64_t)L * K * 2;
    for (int q = threadIdx.x; q < L; q += 256) {
        double s, c;
        sincospi(2.0 * q / L, &s, &c);
        cs[q] = c; sn[q] = s;
    }
    for (int i = threadIdx.x; i < L * K; i += 256) { xr[i] = src[2 * i]; xi[i] = src[2 * i + 1]; }
    __syncthreads();
    for (int i = threadIdx.x; i < L * K; i += 256) {
        const int n = i / K, k = i - n * K;
        double ar = 0.0, ai = 0.0;
        int q = 0;
        for (int m = 0; m < L; ++m) {
            const double c = cs[q], s = sn[q], vr = xr[m * K + k], vi = xi[m * K + k];
            ar += vr * c - vi * s;
            ai += vr * s + vi * c;
            q += n; if (q >= L) q -= L;
        }
        yr[i] = ar; yi[i] = ai;
    }
    __syncthreads();
    const double norm = 1.0 / ((double)L * L);
    for (int i = threadIdx.x; i < L * L; i += 256) {
        const int n = i / L, j = i - n * L;
        double acc = yr[n * K] + ((j & 1) ? -yr[n * K + L / 2] : yr[n * K + L / 2]);
        int q = j;
        for (int k = 1; k < L / 2; ++k) {
            acc += 2.0 * (yr[n * K + k] * cs[q] - yi[n * K + k] * sn[q]);
            q += j; if (q >= L) q -= L;
        }
        acc *= norm;
        if (real_out) real_out[b * (int64_t)L * L + i] = acc;
        spins[b * (int64_t)L * L + i] = acc > 0.0 ? 1 : -1;
    }
}

hipError_t launch_irfft2_sign(const float *spec, int64_t B, int L, int8_t *spins, double *real_out, hipStream_t s)
{
    const int K = L / 2 + 1;
    const size_t lds = (size_t)(2 * L + 4 * L * K) * sizeof(double);
    static thread_local bool configured[IDM_MAX_DEVICES] = {};
    const int slot = idm_device_slot();
    if (slot < 0) return hipErrorInvalidDevice;
    if (!configured[slot]) {
        hipError_t e = hipFuncSetAttribute((const void *)irfft2_sign_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        configured[slot] = true;
    }
    hipLaunchKernelGGL(irfft2_sign_kernel, dim3((unsigned)B), dim3(256), lds, s, spec, L, spins, real_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Checkerboard Metropolis generator (acceptance rule of b_pure_metoropolis_2d_ising.py:35-38).
// One workgroup per chain, the lattice lives in LDS for all sweeps.  Colour c of sweep w updates the
// sites with (row + col) % 2 == c; the k-th such site draws component k % 4 of Philox with counter
// (k / 4, chain lo, chain hi, 2w + c); it flips when u < table[(s+1)/2][(nn+4)/2].
// ---------------------------------------------------------------------------------------------
struct MetroTable { float t[10]; };

__global__ __launch_bounds__(256) void metropolis_kernel(int8_t *__restrict__ spins, int L, MetroTable tab, int n_sweeps,
                                                         PhiloxArgs ph, int sweep0)
{
    extern __shared__ int8_t lat[];
    const int64_t b = blockIdx.x;
    int8_t *x = spins + b * L * L;
    for (int i = threadIdx.x; i < L * L; i += 256) lat[i] = x[i];
    __syncthreads();
    const int64_t g = ph.first_global_index + b;
    const int half = L >> 1, ngroups = (L * half + 3) >> 2;
    for (int w = sweep0; w < sweep0 + n_sweeps; ++w)
        for (int c = 0; c < 2; ++c) {
            for (int q = threadIdx.x; q < ngroups; q += 256) {
                uint32_t r4[4];
                philox4x32_10((uint32_t)q, (uint32_t)g, (uint32_t)((uint64_t)g >> 32), (uint32_t)(2 * w + c), (uint32_t)ph.seed,
                              (uint32_t)(ph.seed >> 32), r4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = q * 4 + j;
                    if (k >= L * half) break;
                    const int r = k / half, col = 2 * (k - r * half) + ((r + c) & 1);
                    const int rp = r + 1 == L ? 0 : r + 1, rm = r == 0 ? L - 1 : r - 1;
                    const int cp = col + 1 == L ? 0 : col + 1, cm = col == 0 ? L - 1 : col - 1;
                    const int s = lat[r * L + col];
                    const int nn = lat[rp * L + col] + lat[rm * L + col] + lat[r * L + cp] + lat[r * L + cm];
                    if (u01(r4[j]) < tab.t[((s + 1) >> 1) * 5 + ((nn + 4) >> 1)]) lat[r * L + col] = (int8_t)-s;
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < L * L; i += 256) x[i] = lat[i];
}

void launch_metropolis(int8_t *spins, int64_t B, int L, const float *table10, int n_sweeps, const PhiloxArgs &ph, int sweep0,
                       hipStream_t s)
{
    MetroTable t;
    for (int i = 0; i < 10; ++i) t.t[i] = table10[i];
    hipLaunchKernelGGL(metropolis_kernel, dim3((unsigned)B), dim3(256), (size_t)L * L, s, spins, L, t, n_sweeps, ph, sweep0);
}
