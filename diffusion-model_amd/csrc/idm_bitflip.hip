// MNIST bit-flip diffusion sampler (mnist_diffusion/mnist_diffusion.py:43-52, :162-240) in the
// canonical float32 arithmetic (parity mode): residual UNet with GroupNorm, time vectors from a
// host-supplied sinusoidal embedding, XOR update with counter-based uniforms.
//
// Canonical order, in addition to the rules of idm_canon.hip: GroupNorm sums are 64 strided
// partial sums (element e = pixel*Cg + channel-in-group goes to partial e % 64, ascending e)
// folded by an xor butterfly 32,16,...,1; variance from fmaf(d,d,.) of centred values; rsqrt by
// a fixed bit-trick + 4 Newton steps; y = (x*a) + b with a = inv*gamma, b = (-mean*a) + beta.
// This file must be compiled with -ffp-contract=off.
#include "../../include/isingdiff.h"
#include "idm_internal.h"
#include "idm_device.h"
#include "idm_canon_math.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

namespace {

constexpr int NBLK = 5;
constexpr int kCin[NBLK] = {32, 32, 64, 192, 96};
constexpr int kWidth[NBLK] = {32, 64, 128, 64, 32};
constexpr int EMB = 32, TEMB = 128, TVEC = 320;

typedef _Float16 act16;
typedef __attribute__((ext_vector_type(4))) _Float16 a16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 a16x8;

DEVI float rsqrt_canon(float v)
{
    float r = __uint_as_float(0x5f3759dfu - (__float_as_uint(v) >> 1));
    const float h = __fmul_rn(0.5f, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float t = __fmul_rn(r, r);
        t = __fmaf_rn(-h, t, 1.5f);
        r = __fmul_rn(r, t);
    }
    return r;
}

// Activations between the kernels are float32 (parity mode; bf16 mode on shapes its fused kernels do not cover) or IEEE half
// (bf16 mode otherwise: every tensor is rounded once where it is produced, all arithmetic stays float32).  Half, not bf16:
// these tensors are rounded to bf16 again as convolution operands, and a bf16 store in between would double that rounding
// noise (measured: 1.4 x the error against float32 arithmetic); half's 11 bits add 1/64 of it.  Stores saturate at +-65504.
DEVI f32x4 ld4(const float *p) { return *(const f32x4 *)p; }
DEVI f32x4 ld4(const act16 *p)
{
    const a16x4 v = *(const a16x4 *)p;
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
DEVI void st4(float *p, f32x4 v) { *(f32x4 *)p = v; }
DEVI void st4(act16 *p, f32x4 v)
{
    a16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (act16)__builtin_amdgcn_fmed3f(v[j], -65504.0f, 65504.0f);
    *(a16x4 *)p = o;
}

struct TimeW {
    const float *d1_w, *d1_b, *d2_w, *d2_b;
    const float *t_w[NBLK], *t_b[NBLK];
};

// One block per embedding row: Dense(128,swish) -> Dense(128,swish) -> five Dense(width_i,swish).
__global__ __launch_bounds__(128) void bf_time_kernel(const float *__restrict__ emb, TimeW w, float *__restrict__ out)
{
    __shared__ float e[EMB], a[TEMB], b[TEMB];
    const int j = threadIdx.x;
    const float *row = emb + (size_t)blockIdx.x * EMB;
    if (j < EMB) e[j] = row[j];
    __syncthreads();
    float acc = 0.0f;
    for (int i = 0; i < EMB; ++i) acc = __fmaf_rn(e[i], w.d1_w[i * TEMB + j], acc);
    a[j] = csilu(__fadd_rn(acc, w.d1_b[j]));
    __syncthreads();
    acc = 0.0f;
    for (int i = 0; i < TEMB; ++i) acc = __fmaf_rn(a[i], w.d2_w[i * TEMB + j], acc);
    b[j] = csilu(__fadd_rn(acc, w.d2_b[j]));
    __syncthreads();
    int o = 0;
    for (int k = 0; k < NBLK; ++k) {
        const int n = kWidth[k];
        if (j < n) {
            acc = 0.0f;
            for (int i = 0; i < TEMB; ++i) acc = __fmaf_rn(b[i], w.t_w[k][i * n + j], acc);
            out[(size_t)blockIdx.x * TVEC + o + j] = csilu(__fadd_rn(acc, w.t_b[k][j]));
        }
        o += n;
    }
}

// Conv2D(32, kernel 1) on the 0/1 image (:207).
template <typename AT>
__global__ __launch_bounds__(256) void bf_in_kernel(const uint8_t *__restrict__ x, const float *__restrict__ w,
                                                    const float *__restrict__ b, AT *__restrict__ out, int64_t npix)
{
    // four consecutive channels per thread
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix * 8) return;
    const int c = (int)(i & 7) * 4;
    const float xv = (float)x[i >> 3];
    const f32x4 w4 = *(const f32x4 *)(w + c), b4 = *(const f32x4 *)(b + c);
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = __fadd_rn(__fmaf_rn(xv, w4[j], 0.0f), b4[j]);
    st4(out + i * 4, r);
}

// GroupNormalization(groups=8) statistics: one wavefront per (sample, group); writes the affine
// pair (a, b) of every channel of the group: y = (x*a) + b.
__global__ __launch_bounds__(256) void bf_gn_stats_kernel(const float *__restrict__ x, int64_t B, int HW, int C,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          float *__restrict__ ab)
{
    const int lane = threadIdx.x & 63;
    const int64_t wv = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= B * 8) return;
    const int64_t b = wv >> 3;
    const int g = (int)(wv & 7), Cg = C >> 3, N = HW * Cg;
    const float *xb = x + (size_t)b * HW * C + g * Cg;
    // element e = lane, lane + 64, ... of the group is (pixel e / Cg, channel e % Cg); the index pair is stepped
    // instead of divided (same elements, same order: each lane still adds its elements one by one)
    const int p0 = lane / Cg, j0 = lane - p0 * Cg, dp = 64 / Cg, dj = 64 - dp * Cg;
    float s = 0.0f;
    {
        int p = p0, j = j0;
#pragma unroll 4
        for (int e = lane; e < N; e += 64) {
            s = __fadd_rn(s, xb[(size_t)p * C + j]);
            j += dj; p += dp;
            if (j >= Cg) { j -= Cg; ++p; }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s = __fadd_rn(s, __shfl_xor(s, off, 64));
    const float mean = __fdiv_rn(s, (float)N);
    float q = 0.0f;
    {
        int p = p0, j = j0;
#pragma unroll 4
        for (int e = lane; e < N; e += 64) {
            const float d = __fsub_rn(xb[(size_t)p * C + j], mean);
            q = __fmaf_rn(d, d, q);
            j += dj; p += dp;
            if (j >= Cg) { j -= Cg; ++p; }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q = __fadd_rn(q, __shfl_xor(q, off, 64));
    const float var = __fdiv_rn(q, (float)N);
    const float inv = rsqrt_canon(__fadd_rn(var, 1e-3f));
    if (lane < Cg) {
        const int c = g * Cg + lane;
        const float a = __fmul_rn(inv, gamma[c]);
        ab[((size_t)b * C + c) * 2] = a;
        ab[((size_t)b * C + c) * 2 + 1] = __fadd_rn(__fmul_rn(-mean, a), beta[c]);
    }
}

// OutT = float (parity mode) or __bf16 (the operand of the bf16 convolution)
template <typename OutT>
__global__ __launch_bounds__(256) void bf_gn_apply_kernel(const float *__restrict__ x, const float *__restrict__ ab,
                                                          OutT *__restrict__ y, int64_t n, int HW, int C)
{
    // four consecutive channels of one pixel per thread (C is a multiple of 4)
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const int c = (int)(i % C);
    const int64_t b = i / ((int64_t)HW * C);
    const f32x4 xv = *(const f32x4 *)(x + i);
    const float *abp = ab + (b * C + c) * 2;
    const f32x4 ab0 = *(const f32x4 *)abp, ab1 = *(const f32x4 *)(abp + 4);   // (a, shift) pairs of channels c .. c+3
    const float a[4] = {ab0[0], ab0[2], ab1[0], ab1[2]}, sh[4] = {ab0[1], ab0[3], ab1[1], ab1[3]};
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = csilu(__fadd_rn(__fmul_rn(xv[k], a[k]), sh[k]));
    if constexpr (sizeof(OutT) == 4) {
        *(f32x4 *)((float *)y + i) = f32x4{r[0], r[1], r[2], r[3]};
    } else {
        *(bf16x4 *)((__bf16 *)y + i) = bf16x4{(__bf16)r[0], (__bf16)r[1], (__bf16)r[2], (__bf16)r[3]};
    }
}

// bf16 convolution mode: GroupNormalization(groups=8) + swish of one sample in ONE workgroup, the sample's values held in
// registers between the statistics and the apply (x is read once, 4 bytes in and 2 bytes out per value, instead of three
// reads and one write by the two kernels above).  Thread t owns quads t, t + THREADS, ... (a quad = four consecutive
// channels of a pixel); THREADS is a multiple of the C / 4 quads of a pixel, so all of a thread's quads are the same four
// channels, hence one group (a group is a multiple of four channels wide).  Mean, then variance of the centred values,
// float32, partial sums folded per group through LDS; swish through the hardware exp2 / rcp.  Not the parity mode's
// summation order: used only where the output is rounded to bf16.
// The input of a ResBlock that follows an upsampling stage, x = Concatenate([UpSampling2D(up), skip]) (:225-226, :230-231),
// read in place: up is (B, Hh, Wh, Cu), skip is (B, 2 Hh, 2 Wh, Cs).  up == nullptr: a plain tensor.
struct UpCatSrc {
    const void *up, *skip;   // float32 or bf16, as the kernel's activation type says
    int Hh, Wh, Cu, Cs;
};

// workgroups per CU the register budget is set for: packed bf16 input takes 2 NV registers, float32 input 4 NV
constexpr int gn_blocks_per_cu(int threads, int nv, bool packed)
{
    const int need = (packed ? 2 : 4) * nv + 48;
    int wps = 512 / need;
    wps = wps > 8 ? 8 : wps < 1 ? 1 : wps;
    const int b = wps * 4 / (threads / 64);
    return b < 1 ? 1 : b;
}

// STATS: y is not written; the (a, b) pairs of the sample's channels go to ab ((samples, C, 2) floats) for a convolution
// kernel that applies y = swish(x a + b) while it stages its input.
template <int THREADS, int NV, typename AT, bool STATS = false>
__global__ __launch_bounds__(THREADS, gn_blocks_per_cu(THREADS, NV, sizeof(AT) == 2)) void bf_gn_fused_bf16_kernel(
    const AT *__restrict__ x, UpCatSrc uc, int HW, int C, const float *__restrict__ gamma, const float *__restrict__ beta,
    __bf16 *__restrict__ y, float *__restrict__ ab)
{
    __shared__ float part[THREADS];
    __shared__ float stat[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Q = C >> 2, Qg = Q >> 3, nq = HW * Q;          // quads per pixel, per group and pixel, per sample
    const int c4 = tid % Q, grp = c4 / Qg;
    // 16-bit input stays packed in the registers (half as many as float32 values: twice the workgroups per CU, which is what
    // hides one sample's load -> reduce -> reduce -> store chain behind another's) and is widened at each of its uses
    constexpr bool PACKED = sizeof(AT) == 2;
    using VT = typename std::conditional<PACKED, a16x4, f32x4>::type;
    auto raw = [](const AT *p) { return *(const VT *)p; };
    // (the packed value passes through an empty asm at each use, or the compiler widens once and keeps the float32 copy)
    auto wide = [](const VT &p) {
        if constexpr (PACKED) {
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            u32x2 u = __builtin_bit_cast(u32x2, p);
            asm volatile("" : "+v"(u));
            const a16x4 h = __builtin_bit_cast(a16x4, u);
            return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        } else {
            return p;
        }
    };
    VT v[NV];
    const VT zero = {};
    if (uc.up) {
        // a thread's quads are the same four channels of pixels p0, p0 + dp, ...: either all from up or all from skip
        const int Wo = 2 * uc.Wh, dp = THREADS / Q, dy = dp / Wo, dx = dp - dy * Wo;
        const int p0 = tid / Q;
        int py = p0 / Wo, px = p0 - py * Wo;
        const bool from_up = 4 * c4 < uc.Cu;
        const AT *ub = (const AT *)uc.up + (size_t)blockIdx.x * uc.Hh * uc.Wh * uc.Cu + 4 * c4;
        const AT *sb = (const AT *)uc.skip + (size_t)blockIdx.x * HW * uc.Cs + (4 * c4 - uc.Cu);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int q = tid + k * THREADS;
            const AT *src = from_up ? ub + (size_t)((py >> 1) * uc.Wh + (px >> 1)) * uc.Cu : sb + (size_t)(py * Wo + px) * uc.Cs;
            v[k] = q < nq ? raw(src) : zero;
            py += dy;
            px += dx;
            if (px >= Wo) { px -= Wo; ++py; }
        }
    } else {
        const AT *xb = x + (size_t)blockIdx.x * HW * C;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int q = tid + k * THREADS;
            v[k] = q < nq ? raw(xb + (size_t)q * 4) : zero;
        }
    }
    // sum over the threads of each group: wavefront w folds groups w, w + waves, ...; entry e of group g is thread
    // (e / Qg) * Q + g * Qg + e % Qg
    auto group_sums = [&](float mine) {
        part[tid] = mine;
        __syncthreads();
        for (int g = wave; g < 8; g += THREADS / 64) {
            float s = 0.0f;
            for (int e = lane; e < THREADS / 8; e += 64) s += part[(e / Qg) * Q + g * Qg + e % Qg];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
            if (lane == 0) stat[g] = s;
        }
        __syncthreads();
        const float r = stat[grp];
        __syncthreads();
        return r;
    };
    const float n = (float)(HW * (C >> 3));
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const f32x4 w = wide(v[k]);
        s += (w[0] + w[1]) + (w[2] + w[3]);
    }
    const float mean = group_sums(s) / n;
    float qd = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (tid + k * THREADS < nq) {
            const f32x4 w = wide(v[k]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = w[j] - mean;
                qd = __builtin_fmaf(d, d, qd);
            }
        }
    }
    const float inv = rsqrt_canon(group_sums(qd) / n + 1e-3f);
    const f32x4 gm = *(const f32x4 *)(gamma + 4 * c4), bt = *(const f32x4 *)(beta + 4 * c4);
    float a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = inv * gm[j];
        b[j] = bt[j] - mean * a[j];
    }
    if constexpr (STATS) {
        if (tid < Q) {
            float *o = ab + ((size_t)blockIdx.x * C + 4 * c4) * 2;
            *(f32x4 *)o = f32x4{a[0], b[0], a[1], b[1]};
            *(f32x4 *)(o + 4) = f32x4{a[2], b[2], a[3], b[3]};
        }
        return;
    }
    __bf16 *yb = y + (size_t)blockIdx.x * HW * C;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int q = tid + k * THREADS;
        if (q < nq) {
            const f32x4 w = wide(v[k]);
            float r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = __builtin_fmaf(w[j], a[j], b[j]);
                r[j] = z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f));
            }
            *(bf16x4 *)(yb + (size_t)q * 4) = bf16x4{(__bf16)r[0], (__bf16)r[1], (__bf16)r[2], (__bf16)r[3]};
        }
    }
}

// Launches the fused kernel when a (threads, quads per thread) instance covers the sample; false: the caller takes the
// two-kernel path.
template <int THREADS, int NV>
bool gn_fused_try(const void *x, bool a16, const UpCatSrc &uc, int64_t B, int HW, int C, const float *gamma, const float *beta,
                  __bf16 *y, float *ab, hipStream_t s, bool probe)
{
    const int Q = C / 4;
    if (THREADS % Q != 0 || (int64_t)HW * Q > (int64_t)THREADS * NV) return false;
    if (probe) return true;
    if (a16)
        hipLaunchKernelGGL((bf_gn_fused_bf16_kernel<THREADS, NV, act16, true>), dim3((unsigned)B), dim3(THREADS), 0, s, (const act16 *)x,
                           uc, HW, C, gamma, beta, y, ab);
    else
        hipLaunchKernelGGL((bf_gn_fused_bf16_kernel<THREADS, NV, float>), dim3((unsigned)B), dim3(THREADS), 0, s, (const float *)x, uc, HW, C,
                           gamma, beta, y, ab);
    return true;
}

// probe: only tell whether an instance covers the sample
// a16: x (or uc's two tensors) holds half and only the statistics are taken: the (a, b) pairs go to ab, y is not written
bool gn_fused_bf16(const void *x, bool a16, const UpCatSrc &uc, int64_t B, int HW, int C, const float *gamma, const float *beta,
                   __bf16 *y, hipStream_t s, bool probe = false, float *ab = nullptr)
{
    if (C % 32 != 0 || B > 0x7fffffff) return false;   // eight groups of a multiple of four channels
#define IDM_GN_TRY(T, N) gn_fused_try<T, N>(x, a16, uc, B, HW, C, gamma, beta, y, ab, s, probe)
    return IDM_GN_TRY(256, 4) || IDM_GN_TRY(256, 8) || IDM_GN_TRY(256, 16) || IDM_GN_TRY(256, 28) || IDM_GN_TRY(768, 4) ||
           IDM_GN_TRY(768, 13) || IDM_GN_TRY(768, 25) || IDM_GN_TRY(1024, 20);
#undef IDM_GN_TRY
}

// out = (conv1x1(x) + bias) + h  (the ResBlock's projected shortcut, :188-192); w == nullptr: out = x + h.
__global__ __launch_bounds__(256) void bf_shortcut_kernel(const float *__restrict__ x, int Cin, const float *__restrict__ w,
                                                          const float *__restrict__ bias, const float *__restrict__ h,
                                                          float *__restrict__ out, int64_t npix, int Cout)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix * Cout) return;
    if (!w) {
        out[i] = __fadd_rn(x[i], h[i]);
        return;
    }
    const int co = (int)(i % Cout);
    const float *xp = x + (i / Cout) * Cin;
    float acc = 0.0f;
    for (int c = 0; c < Cin; ++c) acc = __fmaf_rn(xp[c], w[(size_t)c * Cout + co], acc);
    out[i] = __fadd_rn(__fadd_rn(acc, bias[co]), h[i]);
}

// bf16 convolution mode: the projected shortcut out = (conv1x1(x) + bias) + h on the bf16 matrix cores.  One wavefront =
// 32 pixels x all output channels; x is read as float32 (eight consecutive channels per lane and step) and rounded to bf16
// on the way into the MFMA; wpack: bf16 A fragments [c / 16][Cout / 32][lane][8] (the centre tap of the 3x3 packing).
// HBM-bound: x, h in and out out once each.  AT = act16: x, h and out hold half.
template <int MT, typename AT>
__global__ __launch_bounds__(256) void bf_shortcut_mfma_kernel(const AT *__restrict__ x, UpCatSrc uc, int Cin,
                                                               const bf16x8 *__restrict__ wpack, const float *__restrict__ bias,
                                                               const AT *__restrict__ hres, AT *__restrict__ out, int64_t npix)
{
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile * 32 >= npix) return;
    const int64_t p = tile * 32 + li;
    const bool pv = p < npix;
    const AT *xp = x + (pv ? p : 0) * Cin + 8 * h;
    // upsample-concat input (Cu a multiple of 16): channel groups below Cu come from up's pixel (y / 2, x / 2), the others
    // from skip; xs is biased so that xs + c is skip's channel c - Cu
    const AT *xu = nullptr, *xs = nullptr;
    if (uc.up) {
        const int64_t pp = pv ? p : 0;
        const int Wo = 2 * uc.Wh, HWo = 4 * uc.Hh * uc.Wh;
        const int64_t b = pp / HWo;
        const int r = (int)(pp - b * HWo), py = r / Wo, px = r - py * Wo;
        xu = (const AT *)uc.up + ((b * uc.Hh + (py >> 1)) * uc.Wh + (px >> 1)) * uc.Cu + 8 * h;
        xs = (const AT *)uc.skip + pp * uc.Cs - uc.Cu + 8 * h;
    }
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
    const int ncg = Cin >> 4;
    for (int cg = 0; cg < ncg; ++cg) {
        const AT *src = !uc.up ? xp + 16 * cg : (16 * cg < uc.Cu ? xu : xs) + 16 * cg;
        bf16x8 bv;
        if constexpr (sizeof(AT) == 4) {
            const f32x4 v0 = *(const f32x4 *)src, v1 = *(const f32x4 *)(src + 4);
            bv = bf16x8{(__bf16)v0[0], (__bf16)v0[1], (__bf16)v0[2], (__bf16)v0[3],
                        (__bf16)v1[0], (__bf16)v1[1], (__bf16)v1[2], (__bf16)v1[3]};
        } else {
            const a16x8 v = *(const a16x8 *)src;
#pragma unroll
            for (int j = 0; j < 8; ++j) bv[j] = (__bf16)(float)v[j];
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wpack[((size_t)cg * MT + m) * 64 + lane], bv, acc[m], 0, 0, 0);
    }
    if (!pv) return;
    const int Cout = MT * 32;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = m * 32 + 8 * g + 4 * h;
            const f32x4 b4 = *(const f32x4 *)(bias + co), r4 = ld4(hres + p * Cout + co);
            const f32x4 z = {acc[m][4 * g], acc[m][4 * g + 1], acc[m][4 * g + 2], acc[m][4 * g + 3]};
            st4(out + p * Cout + co, (z + b4) + r4);
        }
}

// MaxPooling2D(2) (:212, :216): four consecutive channels per thread (C is a multiple of 4)
template <typename AT>
__global__ __launch_bounds__(256) void bf_maxpool_kernel(const AT *__restrict__ in, AT *__restrict__ out, int64_t B, int H, int W,
                                                         int C)
{
    const int Ho = H >> 1, Wo = W >> 1;
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= B * Ho * Wo * C) return;
    const int c = (int)(i % C);
    const int x = (int)((i / C) % Wo), y = (int)((i / ((int64_t)C * Wo)) % Ho);
    const int64_t b = i / ((int64_t)C * Wo * Ho);
    const AT *p = in + ((b * H + 2 * y) * W + 2 * x) * C + c;
    const f32x4 v0 = ld4(p), v1 = ld4(p + C), v2 = ld4(p + (size_t)W * C), v3 = ld4(p + (size_t)W * C + C);
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = fmaxf(fmaxf(v0[j], v1[j]), fmaxf(v2[j], v3[j]));
    st4(out + i, r);
}

// out (B,2H,2W,Cu+Cs) = Concatenate([UpSampling2D(up), skip]) (:225-226, :230-231)
__global__ __launch_bounds__(256) void bf_upcat_kernel(const float *__restrict__ up, const float *__restrict__ skip,
                                                       float *__restrict__ out, int64_t B, int H, int W, int Cu, int Cs)
{
    // four consecutive channels per thread (Cu and Cs are multiples of 4: a quad never straddles the two parts)
    const int C = Cu + Cs, Ho = 2 * H, Wo = 2 * W;
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= B * Ho * Wo * C) return;
    const int c = (int)(i % C);
    const int x = (int)((i / C) % Wo), y = (int)((i / ((int64_t)C * Wo)) % Ho);
    const int64_t b = i / ((int64_t)C * Wo * Ho);
    const float *src = c < Cu ? up + ((b * H + (y >> 1)) * W + (x >> 1)) * Cu + c : skip + ((b * Ho + y) * Wo + x) * Cs + (c - Cu);
    *(f32x4 *)(out + i) = *(const f32x4 *)src;
}

// Conv2D(1, kernel 1, sigmoid) (:236)
template <typename AT>
__global__ __launch_bounds__(256) void bf_head_kernel(const AT *__restrict__ r, const float *__restrict__ w,
                                                      const float *__restrict__ bias, float *__restrict__ proba, int64_t npix)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const AT *p = r + i * 32;
    float acc = 0.0f;
#pragma unroll
    for (int c4 = 0; c4 < 32; c4 += 4) {
        const f32x4 v = ld4(p + c4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __fmaf_rn(v[j], w[c4 + j], acc);
    }
    acc = __fadd_rn(acc, bias[0]);
    proba[i] = __fdiv_rn(1.0f, __fadd_rn(1.0f, cexp(-acc)));
}

// half activations handed out as float32 (the optional ResBlock outputs of a forward call)
__global__ __launch_bounds__(256) void bf_widen_kernel(const act16 *__restrict__ in, float *__restrict__ out, int64_t n)
{
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i < n) st4(out + i, ld4(in + i));
}

// X ^= (u < p) (:49-50).  One thread per four pixels; u from the injected tensor or from
// Philox4x32-10 with counter (quad, sample lo, sample hi, t) and the seed as key.
__global__ __launch_bounds__(256) void bf_flip_kernel(uint8_t *__restrict__ x, const float *__restrict__ proba,
                                                      const float *__restrict__ u_in, int64_t B, int HW, PhiloxArgs ph)
{
    const int nq = (HW + 3) >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * nq) return;
    const int64_t b = i / nq;
    const int q = (int)(i - b * nq);
    float u[4];
    if (!u_in) {
        uint32_t r[4];
        const int64_t g = ph.first_global_index + b;
        philox4x32_10((uint32_t)q, (uint32_t)g, (uint32_t)((uint64_t)g >> 32), (uint32_t)ph.t, (uint32_t)ph.seed,
                      (uint32_t)(ph.seed >> 32), r);
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = u01(r[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = q * 4 + j;
        if (e >= HW) break;
        const int64_t k = b * HW + e;
        const float uu = u_in ? u_in[k] : u[j];
        x[k] ^= (uint8_t)(uu < proba[k]);
    }
}

inline unsigned blocks(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

struct idm_bitflip {
    int device = 0;
    const float *t[IDM_BITFLIP_NTENSOR] = {};   // device copies, Keras layouts
    void *c1_pack[NBLK] = {}, *c2_pack[NBLK] = {}, *s_pack[NBLK] = {};
    void *c1_pack16[NBLK] = {}, *c2_pack16[NBLK] = {};   // bf16 fragments of the 3x3 kernels
    void *s_pack16[NBLK] = {};                             // ... and of the 1x1 shortcut kernels
    int precision = IDM_FP32;
    float *emb_dev = nullptr;   // (T, EMB) scratch for the embedding rows
    int emb_cap = 0;
    float *tvec = nullptr;      // (T, TVEC)
    int tvec_cap = 0;
    char *ws = nullptr;
    size_t ws_bytes = 0;
    std::vector<void *> owned;
};

namespace {

#define BFCHK(expr)                                                                                    \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return idm_set_error(IDM_ERR_HIP, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
    } while (0)

int bf_upload(idm_bitflip *m, const float *host, size_t n, float **dev)
{
    BFCHK(hipMalloc((void **)dev, n * sizeof(float)));
    m->owned.push_back(*dev);
    BFCHK(hipMemcpy(*dev, host, n * sizeof(float), hipMemcpyHostToDevice));
    return IDM_OK;
}

// tensor index helpers (order documented in include/isingdiff.h)
enum { T_IN_W = 0, T_IN_B, T_D1_W, T_D1_B, T_D2_W, T_D2_B, T_BLK0 };
enum { K_T_W = 0, K_T_B, K_G1_G, K_G1_B, K_C1_W, K_C1_B, K_G2_G, K_G2_B, K_C2_W, K_C2_B, K_S_W, K_S_B, K_PER_BLK };
inline int tb(int blk, int k) { return T_BLK0 + blk * K_PER_BLK + k; }
constexpr int T_OUT_W = T_BLK0 + NBLK * K_PER_BLK, T_OUT_B = T_OUT_W + 1;

size_t tensor_elems(int idx)
{
    if (idx == T_IN_W || idx == T_IN_B) return 32;
    if (idx == T_D1_W) return (size_t)EMB * TEMB;
    if (idx == T_D2_W) return (size_t)TEMB * TEMB;
    if (idx == T_D1_B || idx == T_D2_B) return TEMB;
    if (idx == T_OUT_W) return 32;
    if (idx == T_OUT_B) return 1;
    const int blk = (idx - T_BLK0) / K_PER_BLK, k = (idx - T_BLK0) % K_PER_BLK;
    const size_t ci = kCin[blk], w = kWidth[blk];
    switch (k) {
    case K_T_W: return TEMB * w;
    case K_G1_G: case K_G1_B: return ci;
    case K_C1_W: return 9 * ci * w;
    case K_C2_W: return 9 * w * w;
    case K_S_W: return ci * w;
    default: return w;
    }
}

int ensure(idm_bitflip *m, size_t bytes)
{
    if (bytes <= m->ws_bytes) return IDM_OK;
    if (m->ws) (void)hipFree(m->ws);
    m->ws = nullptr; m->ws_bytes = 0;
    if (hipMalloc((void **)&m->ws, bytes) != hipSuccess) return idm_set_error(IDM_ERR_NOMEM, "bit-flip workspace allocation failed");
    m->ws_bytes = bytes;
    return IDM_OK;
}

// floats of workspace per sample (see forward_chunk)
size_t ws_floats(int H, int W)
{
    const size_t hw = (size_t)H * W;
    return hw * (32 + 32 + 16 + 8 + 96 + 96 + 32 + 8 + 16 + 32 + 1) + 2 * 192;
}

int time_vectors(idm_bitflip *m, const float *emb_host, int rows, hipStream_t s)
{
    if (rows > m->emb_cap) {
        float *e = nullptr, *tv = nullptr;
        BFCHK(hipMalloc((void **)&e, (size_t)rows * EMB * sizeof(float)));
        BFCHK(hipMalloc((void **)&tv, (size_t)rows * TVEC * sizeof(float)));
        m->owned.push_back(e); m->owned.push_back(tv);
        m->emb_dev = e; m->tvec = tv; m->emb_cap = rows; m->tvec_cap = rows;
    }
    BFCHK(hipMemcpyAsync(m->emb_dev, emb_host, (size_t)rows * EMB * sizeof(float), hipMemcpyHostToDevice, s));
    TimeW w;
    w.d1_w = m->t[T_D1_W]; w.d1_b = m->t[T_D1_B]; w.d2_w = m->t[T_D2_W]; w.d2_b = m->t[T_D2_B];
    for (int k = 0; k < NBLK; ++k) { w.t_w[k] = m->t[tb(k, K_T_W)]; w.t_b[k] = m->t[tb(k, K_T_B)]; }
    hipLaunchKernelGGL(bf_time_kernel, dim3(rows), dim3(128), 0, s, (const float *)m->emb_dev, w, m->tvec);
    BFCHK(hipGetLastError());
    return IDM_OK;
}

// y: float32, or bf16 in the same buffer when `half`
void gn_silu(const float *x, int64_t B, int HW, int C, const float *gamma, const float *beta, float *ab, float *y, bool half,
             hipStream_t s, const UpCatSrc &uc = UpCatSrc{})
{
    if (half && gn_fused_bf16(x, false, uc, B, HW, C, gamma, beta, (__bf16 *)y, s)) return;
    hipLaunchKernelGGL(bf_gn_stats_kernel, dim3((unsigned)((B * 8 + 3) / 4)), dim3(256), 0, s, x, B, HW, C, gamma, beta, ab);
    const int64_t n = B * HW * C;
    if (half)
        hipLaunchKernelGGL(bf_gn_apply_kernel<__bf16>, dim3(blocks(n / 4)), dim3(256), 0, s, x, (const float *)ab, (__bf16 *)y, n, HW, C);
    else
        hipLaunchKernelGGL(bf_gn_apply_kernel<float>, dim3(blocks(n / 4)), dim3(256), 0, s, x, (const float *)ab, y, n, HW, C);
}

// half: `in` holds bf16 and wpack bf16 fragments -> bf16 MFMA kernel (float32 accumulation and output)
// resid / res_out (bf16 mode only, may be null): when the LDS-tiled kernel covers the shape it writes
// res_out = (conv + bias) + resid instead of out and the function returns true (the block's identity shortcut rides in the
// convolution's epilogue); false: out holds the plain convolution.
bool conv3(const float *in, int64_t B, int H, int W, int Cin, const void *wpack, const float *bias, const float *tadd,
           float *out, int Cout, bool half, hipStream_t s, const float *resid = nullptr, float *res_out = nullptr)
{
    ConvArgs a = {};
    a.in0 = in; a.in1 = nullptr; a.C0 = Cin; a.C1 = 0; a.H = H; a.W = W; a.B = B; a.wpack = wpack; a.bias = bias;
    a.scale = nullptr; a.shift = nullptr; a.tadd = tadd; a.out = out; a.Cout = Cout; a.mode = CONV_PLAIN; a.silu = 0;
    a.out_f32 = 1;
    if (!half) { launch_conv_f32(a, s); return false; }
    if (resid && res_out) {
        a.resid = resid; a.out = res_out;
        if (launch_conv_tile_bf16(a, s)) return true;
        a.resid = nullptr; a.out = out;
    }
    if (!launch_conv_tile_bf16(a, s)) launch_conv_bf16(a, s);   // (wider maps and other channel counts: per-layer kernel)
    return false;
}

template <typename AT>
void shortcut_mfma(int mt, const void *x, const UpCatSrc &uc, int Cin, const bf16x8 *wp, const float *sb, const void *h, void *out,
                   int64_t npix, hipStream_t s)
{
    const unsigned grid = (unsigned)((npix + 127) / 128);
#define IDM_SC(M) hipLaunchKernelGGL((bf_shortcut_mfma_kernel<M, AT>), dim3(grid), dim3(256), 0, s, (const AT *)x, uc, Cin, wp, sb, (const AT *)h, (AT *)out, npix)
    switch (mt) {
    case 1: IDM_SC(1); break;
    case 2: IDM_SC(2); break;
    case 3: IDM_SC(3); break;
    default: IDM_SC(4); break;
    }
#undef IDM_SC
}

// The ResBlock on half activations with both GroupNorms folded into the convolutions: a statistics kernel leaves the (a, b)
// pairs of every (sample, channel), the convolution applies y = swish(x a + b) while it stages its tiles, so the normalised
// tensors never go through memory.  uc.up == nullptr: the input is the plain tensor x.  false (an instance is missing):
// nothing useful was written, the caller runs the unfolded kernels.
bool resblock_act16(idm_bitflip *m, int i, const float *x, const UpCatSrc &uc, int64_t B, int H, int W, const float *tvec, float *out,
                    float *tbuf, float *ab, hipStream_t s)
{
    const int Cin = kCin[i], Cw = kWidth[i], HW = H * W;
    auto folded = [&](const void *up, int Cu, const void *full, int Cf, const void *wpack, const float *bias, const float *tadd,
                      const float *resid, float *dst) {
        ConvArgs a = {};
        a.in0 = up; a.in1 = full; a.C0 = Cu; a.C1 = Cf; a.H = H; a.W = W; a.B = B; a.wpack = wpack; a.bias = bias; a.tadd = tadd;
        a.out = dst; a.Cout = Cw; a.mode = CONV_UPCAT; a.out_f32 = 1; a.io_f16 = 1; a.gn_ab = ab; a.resid = resid;
        return launch_conv_tile_bf16(a, s);
    };
    if (!gn_fused_bf16(x, true, uc, B, HW, Cin, m->t[tb(i, K_G1_G)], m->t[tb(i, K_G1_B)], nullptr, s, false, ab)) return false;
    if (!(uc.up ? folded(uc.up, uc.Cu, uc.skip, uc.Cs, m->c1_pack16[i], m->t[tb(i, K_C1_B)], tvec, nullptr, tbuf)
                : folded(nullptr, 0, x, Cin, m->c1_pack16[i], m->t[tb(i, K_C1_B)], tvec, nullptr, tbuf)))
        return false;
    if (!gn_fused_bf16(tbuf, true, UpCatSrc{}, B, HW, Cw, m->t[tb(i, K_G2_G)], m->t[tb(i, K_G2_B)], nullptr, s, false, ab)) return false;
    const bool identity = !m->s_pack16[i];
    // conv2 reads tbuf and, with a projected shortcut, writes it too: in place is safe only per workgroup, so that case
    // goes through `out` and the shortcut kernel adds in place there instead
    if (identity) return folded(nullptr, 0, tbuf, Cw, m->c2_pack16[i], m->t[tb(i, K_C2_B)], nullptr, x, out);
    if (!folded(nullptr, 0, tbuf, Cw, m->c2_pack16[i], m->t[tb(i, K_C2_B)], nullptr, nullptr, out)) return false;
    shortcut_mfma<act16>(Cw / 32, x, uc, Cin, (const bf16x8 *)m->s_pack16[i], m->t[tb(i, K_S_B)], out, out, B * HW, s);
    return true;
}

// ResBlock (:168-193).  ta: >= B*H*W*max(Cin,width) floats, tbuf: >= B*H*W*width floats.
// cat (optional): x = Concatenate([UpSampling2D(cat->up), cat->skip]).  In the bf16 mode the block's two readers of x (the
// first GroupNorm and the projected shortcut) read the two tensors in place when their fused kernels cover the shape;
// otherwise the concatenation is first written to x.
void resblock(idm_bitflip *m, int i, float *x, int64_t B, int H, int W, const float *tvec, float *out, float *ta,
              float *tbuf, float *ab, hipStream_t s, const UpCatSrc *cat = nullptr)
{
    const int Cin = kCin[i], Cw = kWidth[i], HW = H * W;
    const bool half = m->precision == IDM_BF16;
    const bool mfma_shortcut = half && m->s_pack16[i] && Cw % 32 == 0 && Cw <= 128 && Cin % 16 == 0;
    UpCatSrc uc = {};
    if (cat) {
        const bool in_place = mfma_shortcut && cat->Cu % 16 == 0 && cat->Cs % 4 == 0 &&
                              gn_fused_bf16(nullptr, false, *cat, B, HW, Cin, nullptr, nullptr, nullptr, s, true);
        if (in_place) uc = *cat;
        else
            hipLaunchKernelGGL(bf_upcat_kernel, dim3(blocks(B * HW * Cin / 4)), dim3(256), 0, s, (const float *)cat->up,
                               (const float *)cat->skip, x, B, cat->Hh, cat->Wh, cat->Cu, cat->Cs);
    }
    gn_silu(x, B, HW, Cin, m->t[tb(i, K_G1_G)], m->t[tb(i, K_G1_B)], ab, ta, half, s, uc);
    conv3(ta, B, H, W, Cin, half ? m->c1_pack16[i] : m->c1_pack[i], m->t[tb(i, K_C1_B)], tvec, tbuf, Cw, half, s);
    gn_silu(tbuf, B, HW, Cw, m->t[tb(i, K_G2_G)], m->t[tb(i, K_G2_B)], ab, ta, half, s);
    // identity shortcut (Cin == width, :188-192 take the other branch): out = x + h, in the bf16 mode inside the convolution
    const bool identity = !m->s_pack[i] && !(half && m->s_pack16[i]);
    if (conv3(ta, B, H, W, Cw, half ? m->c2_pack16[i] : m->c2_pack[i], m->t[tb(i, K_C2_B)], nullptr, tbuf, Cw, half, s,
              half && identity ? x : nullptr, out))
        return;
    const int64_t npix = B * HW;
    if (mfma_shortcut) {
        const bf16x8 *wp = (const bf16x8 *)m->s_pack16[i];
        const float *sb = m->t[tb(i, K_S_B)];
        shortcut_mfma<float>(Cw / 32, x, uc, Cin, wp, sb, tbuf, out, npix, s);
    } else if (m->s_pack[i]) {
        // projected shortcut (:188-192) on the matrix cores: the centre tap alone is a 1x1 convolution with the
        // same k-ordered chain; out = (chain + bias) + h
        ConvArgs a = {};
        a.in0 = x; a.C0 = Cin; a.C1 = 0; a.H = H; a.W = W; a.B = B; a.wpack = m->s_pack[i]; a.bias = m->t[tb(i, K_S_B)];
        a.out = out; a.Cout = Cw; a.mode = CONV_PLAIN; a.silu = 0; a.out_f32 = 1;
        a.tap_lo = 4; a.tap_hi = 5; a.resid = tbuf;
        launch_conv_f32(a, s);
    } else {
        hipLaunchKernelGGL(bf_shortcut_kernel, dim3(blocks(npix * Cw)), dim3(256), 0, s, x, Cin, (const float *)nullptr,
                           (const float *)nullptr, (const float *)tbuf, out, npix, Cw);
    }
}

// bf16 mode: may the activations between the kernels be half, with the GroupNorms folded into the convolutions
// (resblock_act16)?  Where those kernels cover every level: maps at most 32 wide for the LDS-tiled convolution, a one-workgroup
// GroupNorm instance for every (pixels, channels) pair, the MFMA kernel for every projected shortcut.  Elsewhere the mode
// runs on float32 activations (resblock).  ISINGDIFF_BITFLIP_ACT=f32 selects those everywhere (comparison runs).
bool act16_covers(const idm_bitflip *m, int64_t B, int H, int W)
{
    static const bool off = [] { const char *e = getenv("ISINGDIFF_BITFLIP_ACT"); return e && !strcmp(e, "f32"); }();
    if (off || m->precision != IDM_BF16 || W > 32 || W < 16 || H < 4) return false;
    for (int i = 0; i < NBLK; ++i) {
        const int sh = (i == 0 || i == 4) ? 0 : (i == 2 ? 2 : 1), HW = (H >> sh) * (W >> sh);
        if (!gn_fused_bf16(nullptr, true, UpCatSrc{}, B, HW, kCin[i], nullptr, nullptr, nullptr, nullptr, true) ||
            !gn_fused_bf16(nullptr, true, UpCatSrc{}, B, HW, kWidth[i], nullptr, nullptr, nullptr, nullptr, true))
            return false;
        if (!m->c1_pack16[i] || !m->c2_pack16[i]) return false;
        if (kCin[i] != kWidth[i] && !(m->s_pack16[i] && kWidth[i] % 32 == 0 && kWidth[i] <= 128 && kCin[i] % 16 == 0)) return false;
    }
    return true;
}

// Forward of Bc samples; tvec: this step's five time vectors (device); proba (Bc,H,W).
int forward_chunk(idm_bitflip *m, const uint8_t *x, const float *tvec, float *proba, int64_t Bc, int H, int W,
                  float *const *taps, int64_t tap_off, hipStream_t s)
{
    const size_t hw = (size_t)H * W;
    const bool a16 = act16_covers(m, Bc, H, W);
    float *p = (float *)m->ws;
    // (with half activations the buffers use half of their float32 size; cat and ta stay unused)
    auto take = [&](size_t per) { float *q = p; p += per * Bc; return q; };
    float *x0 = take(hw * 32), *r0 = take(hw * 32), *r1 = take(hw * 16), *r2 = take(hw * 8), *cat = take(hw * 96),
          *ta = take(hw * 96), *tbuf = take(hw * 32), *pl = take(hw * 8), *d3 = take(hw * 16), *d4 = take(hw * 32);
    float *ab = p;  // (Bc, <=192, 2)
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
    const float *tv[NBLK];
    int o = 0;
    for (int i = 0; i < NBLK; ++i) { tv[i] = tvec + o; o += kWidth[i]; }
    const UpCatSrc cat3 = {r2, r1, H4, W4, 128, 64}, cat4 = {d3, r0, H2, W2, 64, 32}, plain = {};
    if (a16) {
        auto maxpool = [&](const float *in, float *out, int h, int w, int C) {
            hipLaunchKernelGGL(bf_maxpool_kernel<act16>, dim3(blocks(Bc * (int64_t)(h / 2) * (w / 2) * C / 4)), dim3(256), 0, s,
                               (const act16 *)in, (act16 *)out, Bc, h, w, C);
        };
        hipLaunchKernelGGL(bf_in_kernel<act16>, dim3(blocks(Bc * hw * 8)), dim3(256), 0, s, x, m->t[T_IN_W], m->t[T_IN_B], (act16 *)x0,
                           (int64_t)(Bc * hw));
        bool ok = resblock_act16(m, 0, x0, plain, Bc, H, W, tv[0], r0, tbuf, ab, s);
        maxpool(r0, pl, H, W, 32);
        ok = ok && resblock_act16(m, 1, pl, plain, Bc, H2, W2, tv[1], r1, tbuf, ab, s);
        maxpool(r1, pl, H2, W2, 64);
        ok = ok && resblock_act16(m, 2, pl, plain, Bc, H4, W4, tv[2], r2, tbuf, ab, s);
        ok = ok && resblock_act16(m, 3, nullptr, cat3, Bc, H2, W2, tv[3], d3, tbuf, ab, s);
        ok = ok && resblock_act16(m, 4, nullptr, cat4, Bc, H, W, tv[4], d4, tbuf, ab, s);
        if (!ok) return idm_set_error(IDM_ERR_INVALID, "bit-flip model: a kernel of the half-activation path does not cover this shape");
        hipLaunchKernelGGL(bf_head_kernel<act16>, dim3(blocks(Bc * hw)), dim3(256), 0, s, (const act16 *)d4, m->t[T_OUT_W], m->t[T_OUT_B],
                           proba, (int64_t)(Bc * hw));
    } else {
        auto maxpool = [&](const float *in, float *out, int h, int w, int C) {
            hipLaunchKernelGGL(bf_maxpool_kernel<float>, dim3(blocks(Bc * (int64_t)(h / 2) * (w / 2) * C / 4)), dim3(256), 0, s, in, out,
                               Bc, h, w, C);
        };
        hipLaunchKernelGGL(bf_in_kernel<float>, dim3(blocks(Bc * hw * 8)), dim3(256), 0, s, x, m->t[T_IN_W], m->t[T_IN_B], x0,
                           (int64_t)(Bc * hw));
        resblock(m, 0, x0, Bc, H, W, tv[0], r0, ta, tbuf, ab, s);
        maxpool(r0, pl, H, W, 32);
        resblock(m, 1, pl, Bc, H2, W2, tv[1], r1, ta, tbuf, ab, s);
        maxpool(r1, pl, H2, W2, 64);
        resblock(m, 2, pl, Bc, H4, W4, tv[2], r2, ta, tbuf, ab, s);
        resblock(m, 3, cat, Bc, H2, W2, tv[3], d3, ta, tbuf, ab, s, &cat3);
        resblock(m, 4, cat, Bc, H, W, tv[4], d4, ta, tbuf, ab, s, &cat4);
        hipLaunchKernelGGL(bf_head_kernel<float>, dim3(blocks(Bc * hw)), dim3(256), 0, s, (const float *)d4, m->t[T_OUT_W], m->t[T_OUT_B],
                           proba, (int64_t)(Bc * hw));
    }
    if (taps) {
        const float *src[NBLK] = {r0, r1, r2, d3, d4};
        const size_t per[NBLK] = {hw * 32, hw * 16, hw * 8, hw * 16, hw * 32};
        for (int i = 0; i < NBLK; ++i) {
            if (!taps[i]) continue;
            float *dst = taps[i] + (size_t)tap_off * per[i];
            if (a16)
                hipLaunchKernelGGL(bf_widen_kernel, dim3(blocks((int64_t)(per[i] * Bc / 4))), dim3(256), 0, s, (const act16 *)src[i], dst,
                                   (int64_t)(per[i] * Bc));
            else
                BFCHK(hipMemcpyAsync(dst, src[i], per[i] * Bc * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
    }
    BFCHK(hipGetLastError());
    return IDM_OK;
}

int check_hw(int64_t B, int H, int W)
{
    if (B <= 0 || H <= 0 || W <= 0 || (H & 3) || (W & 3))
        return idm_set_error(IDM_ERR_INVALID, "bit-flip model: B > 0 and H, W multiples of 4 required (two 2x2 poolings)");
    return IDM_OK;
}

int64_t pick_chunk(int64_t B, int H, int W, int64_t want)
{
    // at most 8 GiB of workspace; without a caller's choice the chunks are equal (a short last chunk runs its kernels on a
    // fraction of the chip)
    const size_t per = (ws_floats(H, W) + (size_t)H * W) * sizeof(float);
    int64_t cap = (int64_t)((size_t)8 << 30) / (int64_t)per;
    if (cap < 1) cap = 1;
    if (want > 0) return std::min<int64_t>(std::min<int64_t>(want, cap), B);
    const int64_t n = (B + cap - 1) / cap;
    return (B + n - 1) / n;
}

}  // namespace

extern "C" {

int idm_bitflip_create(idm_bitflip **out, int device, const idm_bitflip_desc *d)
{
    if (!out || !d || d->n_tensors != IDM_BITFLIP_NTENSOR) return idm_set_error(IDM_ERR_INVALID, "bad bit-flip descriptor");
    BFCHK(hipSetDevice(device));
    idm_bitflip *m = new idm_bitflip();
    m->device = device;
    int rc = IDM_OK;
    for (int i = 0; i < IDM_BITFLIP_NTENSOR && rc == IDM_OK; ++i) {
        const int blk = (i - T_BLK0) / K_PER_BLK, k = (i - T_BLK0) % K_PER_BLK;
        const bool in_blk = i >= T_BLK0 && i < T_OUT_W;
        const bool optional = in_blk && (k == K_S_W || k == K_S_B) && kCin[blk] == kWidth[blk];
        if (!d->tensors[i]) {
            if (!optional) rc = idm_set_error(IDM_ERR_INVALID, "bit-flip descriptor: missing tensor");
            continue;
        }
        float *dev = nullptr;
        rc = bf_upload(m, d->tensors[i], tensor_elems(i), &dev);
        m->t[i] = dev;
        if (rc == IDM_OK && in_blk && (k == K_C1_W || k == K_C2_W)) {
            const int ci = k == K_C1_W ? kCin[blk] : kWidth[blk];
            std::vector<float> pk = idm_pack_weights_f32(d->tensors[i], ci, kWidth[blk]);
            float *pd = nullptr;
            rc = bf_upload(m, pk.data(), pk.size(), &pd);
            (k == K_C1_W ? m->c1_pack : m->c2_pack)[blk] = pd;
            if (rc == IDM_OK) {
                std::vector<uint16_t> ph = idm_pack_weights_bf16(d->tensors[i], ci, kWidth[blk]);
                void *hd = nullptr;
                if (hipMalloc(&hd, ph.size() * 2) != hipSuccess) rc = idm_set_error(IDM_ERR_NOMEM, "bit-flip weight allocation failed");
                else {
                    m->owned.push_back(hd);
                    (void)hipMemcpy(hd, ph.data(), ph.size() * 2, hipMemcpyHostToDevice);
                    (k == K_C1_W ? m->c1_pack16 : m->c2_pack16)[blk] = hd;
                }
            }
        }
        if (rc == IDM_OK && in_blk && k == K_S_W) {
            std::vector<float> pk = idm_pack_weights_f32_1x1(d->tensors[i], kCin[blk], kWidth[blk]);
            float *pd = nullptr;
            rc = bf_upload(m, pk.data(), pk.size(), &pd);
            m->s_pack[blk] = pd;
            if (rc == IDM_OK && kCin[blk] % 16 == 0) {
                // the centre tap of a 3x3 kernel that is zero elsewhere, packed like the 3x3 kernels: [tap][c/16][mt][lane][8]
                const int ci = kCin[blk], co = kWidth[blk];
                std::vector<float> k3((size_t)9 * ci * co, 0.0f);
                std::copy(d->tensors[i], d->tensors[i] + (size_t)ci * co, k3.begin() + (size_t)4 * ci * co);
                const std::vector<uint16_t> ph = idm_pack_weights_bf16(k3.data(), ci, co);
                const size_t per_tap = (size_t)ci * co;
                void *hd = nullptr;
                if (hipMalloc(&hd, per_tap * 2) != hipSuccess) rc = idm_set_error(IDM_ERR_NOMEM, "bit-flip weight allocation failed");
                else {
                    m->owned.push_back(hd);
                    (void)hipMemcpy(hd, ph.data() + 4 * per_tap, per_tap * 2, hipMemcpyHostToDevice);
                    m->s_pack16[blk] = hd;
                }
            }
        }
    }
    if (rc != IDM_OK) {
        idm_bitflip_destroy(m);
        return rc;
    }
    *out = m;
    return IDM_OK;
}

int idm_bitflip_set_precision(idm_bitflip *m, int precision)
{
    if (!m) return idm_set_error(IDM_ERR_INVALID, "null model");
    if (precision != IDM_FP32 && precision != IDM_BF16) return idm_set_error(IDM_ERR_INVALID, "unknown precision");
    m->precision = precision;
    return IDM_OK;
}

int idm_bitflip_destroy(idm_bitflip *m)
{
    if (!m) return IDM_OK;
    (void)hipSetDevice(m->device);
    for (void *p : m->owned) (void)hipFree(p);
    if (m->ws) (void)hipFree(m->ws);
    delete m;
    return IDM_OK;
}

int idm_bitflip_time(idm_bitflip *m, const float *emb_host, float *tvec_dev)
{
    if (!m || !emb_host || !tvec_dev) return idm_set_error(IDM_ERR_INVALID, "null argument");
    BFCHK(hipSetDevice(m->device));
    int rc = time_vectors(m, emb_host, 1, nullptr);
    if (rc != IDM_OK) return rc;
    BFCHK(hipMemcpyAsync(tvec_dev, m->tvec, TVEC * sizeof(float), hipMemcpyDeviceToDevice, nullptr));
    BFCHK(hipStreamSynchronize(nullptr));
    return IDM_OK;
}

int idm_bitflip_forward(idm_bitflip *m, const uint8_t *x_dev, const float *emb_host, float *proba_dev, int64_t B, int H,
                        int W, float *const *taps_dev, void *stream)
{
    if (!m || !x_dev || !emb_host || !proba_dev) return idm_set_error(IDM_ERR_INVALID, "null argument");
    int rc = check_hw(B, H, W);
    if (rc != IDM_OK) return rc;
    BFCHK(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    if ((rc = time_vectors(m, emb_host, 1, s)) != IDM_OK) return rc;
    const int64_t chunk = pick_chunk(B, H, W, 0);
    if ((rc = ensure(m, (size_t)chunk * ws_floats(H, W) * sizeof(float))) != IDM_OK) return rc;
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int64_t bc = std::min<int64_t>(chunk, B - b0);
        rc = forward_chunk(m, x_dev + (size_t)b0 * H * W, m->tvec, proba_dev + (size_t)b0 * H * W, bc, H, W, taps_dev, b0, s);
        if (rc != IDM_OK) return rc;
    }
    return IDM_OK;
}

int idm_bitflip_sample(idm_bitflip *m, const idm_bitflip_cfg *cfg, uint8_t *x_dev, const float *uniforms_dev,
                       uint8_t *traj_dev, int64_t B, int H, int W, void *stream)
{
    if (!m || !cfg || !x_dev || !cfg->emb_table_host || cfg->T <= 0) return idm_set_error(IDM_ERR_INVALID, "bad argument");
    int rc = check_hw(B, H, W);
    if (rc != IDM_OK) return rc;
    BFCHK(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    const int T = cfg->T;
    if ((rc = time_vectors(m, cfg->emb_table_host, T, s)) != IDM_OK) return rc;
    const int64_t chunk = pick_chunk(B, H, W, cfg->chunk);
    const size_t hw = (size_t)H * W;
    if ((rc = ensure(m, (size_t)chunk * (ws_floats(H, W) + hw) * sizeof(float))) != IDM_OK) return rc;
    float *proba = (float *)m->ws + (size_t)chunk * ws_floats(H, W);
    const size_t n = (size_t)B * hw;
    if (traj_dev) BFCHK(hipMemcpyAsync(traj_dev, x_dev, n, hipMemcpyDeviceToDevice, s));
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int64_t bc = std::min<int64_t>(chunk, B - b0);
        uint8_t *xc = x_dev + (size_t)b0 * hw;
        for (int t = T - 1; t >= 0; --t) {
            rc = forward_chunk(m, xc, m->tvec + (size_t)t * TVEC, proba, bc, H, W, nullptr, 0, s);
            if (rc != IDM_OK) return rc;
            PhiloxArgs ph = {cfg->seed, cfg->first_global_index + b0, t};
            const float *u = uniforms_dev ? uniforms_dev + (size_t)t * n + (size_t)b0 * hw : nullptr;
            hipLaunchKernelGGL(bf_flip_kernel, dim3(blocks(bc * (int64_t)((hw + 3) / 4))), dim3(256), 0, s, xc,
                               (const float *)proba, u, bc, (int)hw, ph);
            if (traj_dev)
                BFCHK(hipMemcpyAsync(traj_dev + (size_t)(T - t) * n + (size_t)b0 * hw, xc, (size_t)bc * hw,
                                     hipMemcpyDeviceToDevice, s));
        }
    }
    BFCHK(hipGetLastError());
    return IDM_OK;
}

}  // extern "C"
