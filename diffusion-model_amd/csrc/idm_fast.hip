// bf16 matrix-core kernels: activations and weights bfloat16, accumulation float32.
#include "idm_internal.h"

#include <algorithm>

#define DEVI __device__ __forceinline__

DEVI float fast_silu(float y) { return y / (1.0f + __expf(-y)); }

// ---------------------------------------------------------------------------------------------
// Generic per-layer 3x3 conv, v_mfma_f32_32x32x16_bf16, operands fetched straight from global
// memory.  One wave = 32 pixels x all Cout.  Used for shapes the fused kernels do not cover.
// Weights packed [tap][c/16][mt][lane][8]: element j of lane l is W[tap][16*(c/16)+8*(l>>5)+j][32mt+(l&31)].
// ---------------------------------------------------------------------------------------------
template <int MT, int MODE>
__global__ __launch_bounds__(256) void conv3x3_bf16_direct_kernel(ConvArgs a)
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
    const int ncg = Cin >> 4;
    const __bf16 *in0 = (const __bf16 *)a.in0;
    const __bf16 *in1 = (const __bf16 *)a.in1;
    const bf16x8 *wp = (const bf16x8 *)a.wpack + lane;

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;

    for (int tap = 0; tap < 9; ++tap) {
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        const bool inb = pv && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
        for (int cg = 0; cg < ncg; ++cg) {
            const int c = cg * 16 + 8 * h;
            bf16x8 bv;
#pragma unroll
            for (int j = 0; j < 8; ++j) bv[j] = (__bf16)0.0f;
            if (inb) {
                if (MODE == CONV_PLAIN) {
                    bv = *(const bf16x8 *)(in0 + ((b * a.H + yy) * a.W + xx) * a.C0 + c);
                } else if (MODE == CONV_POOL) {
                    const int H2 = 2 * a.H, W2 = 2 * a.W;
                    const __bf16 *q = in0 + ((b * H2 + 2 * yy) * W2 + 2 * xx) * a.C0 + c;
                    const bf16x8 t0 = *(const bf16x8 *)q, t1 = *(const bf16x8 *)(q + a.C0);
                    const bf16x8 t2 = *(const bf16x8 *)(q + (int64_t)W2 * a.C0);
                    const bf16x8 t3 = *(const bf16x8 *)(q + (int64_t)W2 * a.C0 + a.C0);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        bv[j] = (__bf16)((((float)t0[j] + (float)t1[j]) + (float)t2[j] + (float)t3[j]) * 0.25f);
                } else {
                    if (c < a.C0) {
                        const int Hh = a.H >> 1, Wh = a.W >> 1;
                        bv = *(const bf16x8 *)(in0 + ((b * Hh + (yy >> 1)) * Wh + (xx >> 1)) * a.C0 + c);
                    } else {
                        bv = *(const bf16x8 *)(in1 + ((b * a.H + yy) * a.W + xx) * a.C1 + (c - a.C0));
                    }
                }
            }
            const bf16x8 *w = wp + (int64_t)(tap * ncg + cg) * MT * 64;
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[m * 64], bv, acc[m], 0, 0, 0);
        }
    }
    if (!pv) return;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = m * 32 + 8 * g + 4 * h;
            float z[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[m][4 * g + j] + a.bias[co + j];
                if (a.scale) v = v * a.scale[co + j] + a.shift[co + j];
                if (a.silu) v = fast_silu(v);
                if (a.tadd) v += a.tadd[co + j];
                z[j] = v;
            }
            if (a.out_f32) {
                f32x4 o = {z[0], z[1], z[2], z[3]};
                *(f32x4 *)((float *)a.out + p * a.Cout + co) = o;
            } else {
                bf16x4 o = {(__bf16)z[0], (__bf16)z[1], (__bf16)z[2], (__bf16)z[3]};
                *(bf16x4 *)((__bf16 *)a.out + p * a.Cout + co) = o;
            }
        }
}

template <int MT>
static void launch_conv_bf16_mt(const ConvArgs &a, hipStream_t s)
{
    const int64_t npix = a.B * a.H * a.W;
    const unsigned grid = (unsigned)((npix + 127) / 128);
    switch (a.mode) {
    case CONV_PLAIN: hipLaunchKernelGGL((conv3x3_bf16_direct_kernel<MT, CONV_PLAIN>), dim3(grid), dim3(256), 0, s, a); break;
    case CONV_POOL: hipLaunchKernelGGL((conv3x3_bf16_direct_kernel<MT, CONV_POOL>), dim3(grid), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((conv3x3_bf16_direct_kernel<MT, CONV_UPCAT>), dim3(grid), dim3(256), 0, s, a); break;
    }
}

void launch_conv_bf16(const ConvArgs &a, hipStream_t s)
{
    switch (a.Cout / 32) {
    case 1: launch_conv_bf16_mt<1>(a, s); break;
    case 2: launch_conv_bf16_mt<2>(a, s); break;
    default: launch_conv_bf16_mt<4>(a, s); break;
    }
}

// ---------------------------------------------------------------------------------------------
// First layer (C -> 32 channels, C = 1 or 2) with bf16 output: reads the unpadded float32 state (B,H,W,C), writes the
// padded map (B,Hp,Wp,32).  One thread per pixel (the taps are read once), hardware exp / rcp in the activation, four
// 16-byte stores per pixel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_first_bf16_kernel(const float *__restrict__ x, int64_t B, int H, int W, int Hp, int Wp,
                                                              int C, const float *__restrict__ w, const float *__restrict__ bias,
                                                              const float *__restrict__ scale, const float *__restrict__ shift,
                                                              __bf16 *__restrict__ out, int transposed)
{
    // transposed: the map is processed with rows and columns exchanged - x is stored (B, W, H, C), and the caller passes
    // the kernel with its taps transposed - so that the output is the transposed map of the layer
    __shared__ __attribute__((aligned(16))) float ws[9 * 2 * 32];
    __shared__ __attribute__((aligned(16))) float bs[3 * 32];
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
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;   // outside the map, or in its zero padding
        for (int c = 0; c < C; ++c) {
            const float v = transposed ? x[((b * W + xx) * H + yy) * C + c] : x[((b * H + yy) * W + xx) * C + c];
            const float *wr = ws + (tap * C + c) * 32;
#pragma unroll
            for (int f = 0; f < 32; ++f) acc[f] = fmaf(v, wr[f], acc[f]);
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bf16x8 o;
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const float z = (acc[8 * g + f] + bs[8 * g + f]) * bs[32 + 8 * g + f] + bs[64 + 8 * g + f];
            o[f] = (__bf16)(z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f)));
        }
        *(bf16x8 *)(out + p * 32 + 8 * g) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// First layer of the two-channel (spectral) model on the matrix cores: 32 pixels x 32 couts per wavefront and step,
// K = 9 taps x 2 channels = 18 of the 32 rows of two v_mfma_f32_32x32x16_bf16 steps.  Both operands are split into
// bf16 (hi, lo) pairs and the three significant products are accumulated (hi*hi + hi*lo + lo*hi: relative error 2^-16, the
// state and the weights keep their float32 precision), six MFMAs per 32 pixels.  The kernel above spends 576 vector FMAs per
// pixel on the same sums; this one is bound by its 64 output bytes per pixel.
// wp: A fragments [part (hi, lo)][k-step][lane][8] (lane l: cout l & 31, k = 16 step + 8 (l >> 5) + j, k = 2 tap + channel),
// BatchNorm folded into them; biasf: the folded bias.  x: (B, H, W, 2), or (B, W, H, 2) when transposed (see above).
// ---------------------------------------------------------------------------------------------
// ROWTILE: Wp == W == 32 and the map transposed (the spectral model at L = 32): a tile is one row of one spectrum, so the
// row and the spectrum are wavefront-uniform (scalar arithmetic) and a lane's column is its index.
template <bool ROWTILE>
__global__ __launch_bounds__(256) void conv_first2_mfma_kernel(const float *__restrict__ x, int npix, int ntiles, int H, int W, int Hp,
                                                               int Wp, const bf16x8 *__restrict__ wp, const float *__restrict__ biasf,
                                                               __bf16 *__restrict__ out, int transposed)
{
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const bf16x8 ahi0 = wp[lane], ahi1 = wp[64 + lane], alo0 = wp[128 + lane], alo1 = wp[192 + lane];
    f32x4 kb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) kb[g] = *(const f32x4 *)(biasf + 8 * g + 4 * h);
    const f32x2 *x2 = (const f32x2 *)x;
    const int nwaves = (int)gridDim.x * 4;
    for (int tile = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); tile < ntiles; tile += nwaves) {
        const int p = tile * 32 + li;
        const bool valid = p < npix;
        // this lane's taps: 4 h .. 4 h + 3 (k-step 0) and, for h = 0, tap 8 (k-step 1)
        f32x2 v[5];
        if constexpr (ROWTILE) {
            const unsigned ut = (unsigned)__builtin_amdgcn_readfirstlane(tile);
            const unsigned b = ut / (unsigned)Hp, yo = ut - b * (unsigned)Hp;
            const f32x2 *xb = x2 + (size_t)b * 32u * (unsigned)H;       // spectrum b: element (column xx, row yy) at xx * H + yy
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int tap = i < 4 ? 4 * h + i : 8;
                const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
                const int yy = (int)yo + ky - 1, xx = li + kx - 1;
                const bool inb = (i < 4 || h == 0) && yy >= 0 && yy < H && xx >= 0 && xx < 32;
                v[i] = inb ? xb[xx * H + yy] : f32x2{0.f, 0.f};
            }
        } else {
            const unsigned pc = valid ? (unsigned)p : 0u;
            const unsigned t = pc / (unsigned)Wp, xo = pc - t * (unsigned)Wp, b = t / (unsigned)Hp, yo = t - b * (unsigned)Hp;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int tap = i < 4 ? 4 * h + i : 8;
                const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
                const int yy = (int)yo + ky - 1, xx = (int)xo + kx - 1;
                const bool inb = valid && (i < 4 || h == 0) && yy >= 0 && yy < H && xx >= 0 && xx < W;
                const unsigned idx = transposed ? (b * (unsigned)W + (unsigned)xx) * (unsigned)H + (unsigned)yy
                                                : (b * (unsigned)H + (unsigned)yy) * (unsigned)W + (unsigned)xx;
                v[i] = inb ? x2[idx] : f32x2{0.f, 0.f};
            }
        }
        bf16x8 bhi0, blo0, bhi1, blo1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = v[j >> 1][j & 1];
            const __bf16 hi = (__bf16)f;
            bhi0[j] = hi;
            blo0[j] = (__bf16)(f - (float)hi);
            const float f1 = j < 2 ? v[4][j] : 0.0f;
            const __bf16 hi1 = (__bf16)f1;
            bhi1[j] = hi1;
            blo1[j] = (__bf16)(f1 - (float)hi1);
        }
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = kb[i >> 2][i & 3];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo0, bhi0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi0, blo0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo1, bhi1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi1, blo1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi0, bhi0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi1, bhi1, acc, 0, 0, 0);
        if (valid) {
            __bf16 *o = out + (size_t)p * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 r;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float z = acc[4 * g + i];
                    r[i] = (__bf16)(z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f)));
                }
                *(bf16x4 *)(o + 8 * g) = r;
            }
        }
    }
}

bool launch_conv_first2_mfma(const float *x, int64_t B, int H, int W, int Hp, int Wp, const void *wpack, const float *biasf, void *out,
                             hipStream_t s, bool transposed)
{
    const int64_t npix = B * Hp * Wp;
    if (npix <= 0 || npix >= ((int64_t)1 << 31) - 64 || B * H * W >= ((int64_t)1 << 31)) return false;   // 32-bit pixel and element indices
    const int ntiles = (int)((npix + 31) / 32);
    const unsigned grid = (unsigned)std::min<int64_t>((ntiles + 3) / 4, 256 * 32);
    if (transposed && W == 32 && Wp == 32)
        hipLaunchKernelGGL(conv_first2_mfma_kernel<true>, dim3(grid), dim3(256), 0, s, x, (int)npix, ntiles, H, W, Hp, Wp, (const bf16x8 *)wpack,
                           biasf, (__bf16 *)out, 1);
    else
        hipLaunchKernelGGL(conv_first2_mfma_kernel<false>, dim3(grid), dim3(256), 0, s, x, (int)npix, ntiles, H, W, Hp, Wp, (const bf16x8 *)wpack,
                           biasf, (__bf16 *)out, transposed ? 1 : 0);
    return true;
}

void launch_conv_first_bf16(const float *x, int64_t B, int H, int W, int Hp, int Wp, int C, const float *w, const float *bias,
                            const float *scale, const float *shift, void *out, hipStream_t s, bool transposed)
{
    const unsigned grid = (unsigned)((B * Hp * Wp + 255) / 256);
    hipLaunchKernelGGL(conv_first_bf16_kernel, dim3(grid), dim3(256), 0, s, x, B, H, W, Hp, Wp, C, w, bias, scale, shift,
                       (__bf16 *)out, transposed ? 1 : 0);
}
