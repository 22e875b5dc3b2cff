// Fused two-convolution band kernels (bf16 MFMA, float32 accumulate).
//
// One workgroup (4 wavefronts) owns one band of R output rows of one lattice and runs
//     stage-in -> conv1 -> (LDS) -> conv2 -> epilogue
// without the intermediate map ever leaving the CU.  conv1 is evaluated on R+2 rows (one halo
// row each side, recomputed by the neighbouring band) so that conv2 sees exactly the zero-padded
// "same" convolution of the whole map.  The UNet's data movement is folded into the edges:
//     FIRST   conv1 reads the float32 state x (Cin = 1, the 9 taps are the K of one MFMA step,
//             x split into bf16 hi + lo so the state keeps ~16 mantissa bits)
//     UPCAT   conv1 reads Concatenate([UpSampling2D(a), skip]): nearest upsampling is an index
//             shift on the LDS tile of the half-resolution map
//     POOL    conv2's output is also written 2x2-average-pooled
//     HEAD    conv2's output goes through the 1x1 head conv and the reverse-diffusion update of x
//             (Philox noise drawn in-kernel) instead of being stored
//
// GEMM orientation: D[cout][pixel] = sum_k W[cout][k] * X[k][pixel] with v_mfma_f32_32x32x16_bf16;
// A = weights (fragment-packed in global memory, L1/L2 resident), B = activations read from LDS
// as 16-byte runs of 8 consecutive channels.  LDS activation tiles are [row][col][channel] with
// one zero column each side and a position stride of 2*C + 16 bytes: the 16-byte pad makes the
// 16-lane groups of ds_read_b128 hit distinct banks without any address swizzle, so every operand
// address is base + compile-time/scalar offset.
#include "idm_internal.h"
#include "idm_device.h"

namespace {

constexpr int WG = 256;

enum { M1_PLAIN = 0, M1_UPCAT = 1, M1_FIRST = 2 };
enum { E2_STORE = 0, E2_STORE_POOL = 1, E2_HEAD = 2 };

DEVI float fsilu(float y) { return y * __builtin_amdgcn_rcpf(1.0f + __expf(-y)); }

DEVI bf16x4 pack4(float a, float b, float c, float d)
{
    bf16x4 o = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
    return o;
}

// One convolution stage over `npix` output pixels (pixel p -> row r0 + p / W, column p % W of the
// stage's output region).  NT pixel tiles and MS cout tiles are accumulated together per wave.
//   PLAIN part: operand for tap (ky,kx), channel group cg at  tile + ((r+ky)*(W+2) + x+kx)*S + 32*cg
//   UP part:    the same on the half-resolution tile at row (r+ky)>>1, column (x+kx+1)>>1
template <int CA, int CB, bool UP, int COUT, int NT, int MS, class Epi>
DEVI void conv_stage(const char *lds_a, const char *lds_b, int W, int r0, int npix, const bf16x8 *__restrict__ wp,
                     int wave, int lane, Epi &&epi)
{
    constexpr int MTOT = COUT / 32, NSL = MTOT / MS;
    constexpr int SA = CA * 2 + 16, SB = CB * 2 + 16;
    constexpr int NCG_A = CA / 16, NCG_B = CB / 16, NCG = NCG_A + NCG_B;
    const int ntiles = (npix + 31) >> 5, ngroups = (ntiles + NT - 1) / NT;
    const int h = lane >> 5, li = lane & 31;
    const int Wp2 = W + 2, Wh2 = (W >> 1) + 2;
    wp += lane;

#pragma unroll 1
    for (int u = wave; u < ngroups * NSL; u += 4) {
        const int grp = u / NSL, sl = u - grp * NSL;
        int pr[NT], px[NT];
        bool valid[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int p = (grp * NT + n) * 32 + li;
            valid[n] = p < npix;
            const int pc = valid[n] ? p : 0;
            const int q = pc / W;
            pr[n] = q + r0;
            px[n] = pc - q * W;
        }
        f32x16 acc[MS][NT];
#pragma unroll
        for (int m = 0; m < MS; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

        // Weights of one tap (NCG x MS fragments) are fetched one tap ahead into a second
        // register set, so the global-load latency is covered by the previous tap's MFMAs.
        const bf16x8 *wsl = wp + sl * MS * 64;
        auto load_w = [&](bf16x8(&w)[NCG][MS], int tap) {
            const bf16x8 *wt = wsl + (size_t)(tap * NCG) * MTOT * 64;
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
                for (int m = 0; m < MS; ++m) w[cg][m] = wt[(cg * MTOT + m) * 64];
        };
        auto compute = [&](const bf16x8(&w)[NCG][MS], int tap) {
            const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
            if constexpr (CA > 0) {
                int addr[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    addr[n] = UP ? ((((pr[n] + ky) >> 1) * Wh2 + ((px[n] + kx + 1) >> 1)) * SA + h * 16)
                                 : (((pr[n] + ky) * Wp2 + px[n] + kx) * SA + h * 16);
#pragma unroll
                for (int cg = 0; cg < NCG_A; ++cg) {
                    bf16x8 bv[NT];
#pragma unroll
                    for (int n = 0; n < NT; ++n) bv[n] = *(const bf16x8 *)(lds_a + addr[n] + cg * 32);
#pragma unroll
                    for (int m = 0; m < MS; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[cg][m], bv[n], acc[m][n], 0, 0, 0);
                }
            }
            if constexpr (CB > 0) {
                int addr[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) addr[n] = ((pr[n] + ky) * Wp2 + px[n] + kx) * SB + h * 16;
#pragma unroll
                for (int cg = 0; cg < NCG_B; ++cg) {
                    bf16x8 bv[NT];
#pragma unroll
                    for (int n = 0; n < NT; ++n) bv[n] = *(const bf16x8 *)(lds_b + addr[n] + cg * 32);
#pragma unroll
                    for (int m = 0; m < MS; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[NCG_A + cg][m], bv[n], acc[m][n], 0, 0, 0);
                }
            }
        };
        bf16x8 w0[NCG][MS], w1[NCG][MS];
        load_w(w0, 0);
#pragma unroll 1
        for (int tap = 0; tap < 8; tap += 2) {
            load_w(w1, tap + 1);
            compute(w0, tap);
            load_w(w0, tap + 2);
            compute(w1, tap + 1);
        }
        compute(w0, 8);
#pragma unroll
        for (int m = 0; m < MS; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) epi(acc[m][n], sl * MS + m, pr[n], px[n], valid[n]);
    }
}

// First layer (Cin = 1): K = 9 taps in one MFMA step; x = hi + lo in bf16, two MFMAs per tile.
// xs: float32 tile [(rows)][W+2].  wfirst: one A fragment (lane l: cout l&31, k = 8*(l>>5)+j -> tap k).
template <class Epi>
DEVI void conv_first_stage(const float *xs, int W, int r0, int npix, const bf16x8 *__restrict__ wfirst, int wave,
                           int lane, Epi &&epi)
{
    const int ntiles = (npix + 31) >> 5;
    const int h = lane >> 5, li = lane & 31, Wp2 = W + 2;
    const bf16x8 a = wfirst[lane];
    for (int t = wave; t < ntiles; t += 4) {
        const int p = t * 32 + li;
        const bool valid = p < npix;
        const int pc = valid ? p : 0;
        const int r = pc / W + r0, x = pc - (r - r0) * W;
        const float *q = xs + r * Wp2 + x;
        float v[8];
        if (h == 0) {
            v[0] = q[0]; v[1] = q[1]; v[2] = q[2];
            v[3] = q[Wp2]; v[4] = q[Wp2 + 1]; v[5] = q[Wp2 + 2];
            v[6] = q[2 * Wp2]; v[7] = q[2 * Wp2 + 1];
        } else {
            v[0] = q[2 * Wp2 + 2];
#pragma unroll
            for (int j = 1; j < 8; ++j) v[j] = 0.0f;
        }
        bf16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            hi[j] = (__bf16)v[j];
            lo[j] = (__bf16)(v[j] - (float)hi[j]);
        }
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, lo, acc, 0, 0, 0);
        epi(acc, 0, r, x, valid);
    }
}

// Copies rows [g0, g0+nrows) x all W columns x C channels of a global NHWC bf16 map into an LDS
// tile whose row t holds global row g0+t, with one zero column each side; rows outside [0,Hmap)
// stay zero (the tile was cleared).  16 bytes per thread per step, coalesced.
template <int C>
DEVI void stage_in(char *tile, const __bf16 *__restrict__ src, int Hmap, int W, int g0, int nrows, int tid)
{
    constexpr int S = C * 2 + 16, CH = C / 8;
    const int total = nrows * W * CH;
    for (int i = tid; i < total; i += WG) {
        const int ch = i % CH, pos = i / CH;
        const int x = pos % W, t = pos / W;
        const int g = g0 + t;
        if (g < 0 || g >= Hmap) continue;
        const bf16x8 v = *(const bf16x8 *)(src + ((size_t)g * W + x) * C + ch * 8);
        *(bf16x8 *)(tile + ((t * (W + 2)) + x + 1) * S + ch * 16) = v;
    }
}

DEVI void lds_clear(char *p, int bytes, int tid)
{
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid * 16; i < bytes; i += WG * 16) *(f32x4 *)(p + i) = z;
}

}  // namespace

template <int MODE1, int CA, int CB, int C1, int C2, int EPI2, int NT1, int MS1, int NT2, int MS2>
__global__ __launch_bounds__(WG, 2) void fused2_kernel(FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5;
    const int64_t b = blockIdx.x / a.nbands;
    const int band = blockIdx.x - (int)b * a.nbands;
    const int W = a.W, H = a.H, R = a.R;
    const int y0 = band * R;
    constexpr int SA = CA * 2 + 16, SB = CB * 2 + 16, SM = C1 * 2 + 16, SO = C2 * 2 + 16;
    (void)SA; (void)SB;

    // LDS map: [input region | mid tile]; the output staging / score buffer reuses the input region.
    char *lds_in = smem;
    char *lds_mid = smem + a.lds_in_bytes;
    lds_clear(smem, a.lds_in_bytes + a.lds_mid_bytes, tid);
    __syncthreads();

    // ---- stage in: rows y0-2 .. y0+R+1 of conv1's input ----
    char *tile_a = lds_in, *tile_b = nullptr;
    if constexpr (MODE1 == M1_FIRST) {
        float *xs = (float *)lds_in;
        const float *x = (const float *)a.in_a + (size_t)b * a.Hx * a.Wx;
        const int total = (R + 4) * W;
        for (int i = tid; i < total; i += WG) {
            const int t = i / W, c = i - t * W, g = y0 - 2 + t;
            if (g >= 0 && g < a.Hx && c < a.Wx) xs[t * (W + 2) + c + 1] = x[(size_t)g * a.Wx + c];
        }
    } else if constexpr (MODE1 == M1_PLAIN) {
        stage_in<CA>(tile_a, (const __bf16 *)a.in_a + (size_t)b * H * W * CA, H, W, y0 - 2, R + 4, tid);
    } else {
        const int Hh = H >> 1, Wh = W >> 1;
        stage_in<CA>(tile_a, (const __bf16 *)a.in_a + (size_t)b * Hh * Wh * CA, Hh, Wh, (y0 >> 1) - 1, (R >> 1) + 2, tid);
        tile_b = lds_in + (((R >> 1) + 2) * (Wh + 2) * SA + 15) / 16 * 16;
        stage_in<CB>(tile_b, (const __bf16 *)a.in_b + (size_t)b * H * W * CB, H, W, y0 - 2, R + 4, tid);
    }
    __syncthreads();

    // ---- conv1 on rows y0-1 .. y0+R (clipped to the map) -> mid tile row (g - (y0-1)) ----
    {
        const int rlo = y0 == 0 ? 1 : 0;
        const int rhi = min(R + 2, H - y0 + 1);
        const int npix = (rhi - rlo) * W;
        auto epi1 = [&](const f32x16 &acc, int mt, int r, int x, bool valid) {
            if (!valid) return;
            char *dst = lds_mid + (r * (W + 2) + x + 1) * SM;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = mt * 32 + 8 * g + 4 * h;
                const f32x4 bv = *(const f32x4 *)(a.b1 + co);
                float z[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) z[j] = acc[4 * g + j] + bv[j];
                if (a.s1) {
                    const f32x4 sv = *(const f32x4 *)(a.s1 + co), hv = *(const f32x4 *)(a.h1 + co);
#pragma unroll
                    for (int j = 0; j < 4; ++j) z[j] = z[j] * sv[j] + hv[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) z[j] = fsilu(z[j]);
                if (a.tadd1) {
                    const f32x4 tv = *(const f32x4 *)(a.tadd1 + co);
#pragma unroll
                    for (int j = 0; j < 4; ++j) z[j] += tv[j];
                }
                *(bf16x4 *)(dst + co * 2) = pack4(z[0], z[1], z[2], z[3]);
            }
        };
        if constexpr (MODE1 == M1_FIRST)
            conv_first_stage((const float *)lds_in, W, rlo, npix, (const bf16x8 *)a.w1, wave, lane, epi1);
        else
            conv_stage<CA, CB, MODE1 == M1_UPCAT, C1, NT1, MS1>(tile_a, tile_b, W, rlo, npix, (const bf16x8 *)a.w1, wave,
                                                                lane, epi1);
    }
    __syncthreads();

    // ---- conv2 on rows y0 .. y0+R-1 (clipped) ----
    const int rows2 = min(R, H - y0);
    const int npix2 = rows2 * W;
    if constexpr (EPI2 == E2_HEAD) {
        float *score_s = (float *)lds_in;  // rows2 * W floats
        auto epi2 = [&](const f32x16 &acc, int mt, int r, int x, bool valid) {
            (void)mt;
            float part = 0.0f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = 8 * g + 4 * h;
                const f32x4 bv = *(const f32x4 *)(a.b2 + co), wv = *(const f32x4 *)(a.w_head + co);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float z = acc[4 * g + j] + bv[j];
                    if (a.s2) z = z * a.s2[co + j] + a.h2[co + j];
                    part += fsilu(z) * wv[j];
                }
            }
            part += __shfl_xor(part, 32, 64);
            if (valid && h == 0) score_s[r * W + x] = part + a.b_head[0];
        };
        conv_stage<C1, 0, false, C2, NT2, MS2>(lds_mid, nullptr, W, 0, npix2, (const bf16x8 *)a.w2, wave, lane, epi2);
        __syncthreads();
        // reverse-diffusion update of the band's rows of x (unpadded (Hx, Wx))
        const int rowsx = min(rows2, a.Hx - y0);
        float *xg = a.x + (size_t)b * a.Hx * a.Wx;
        const size_t lat_off = (size_t)b * a.Hx * a.Wx;
        if ((a.Wx & 3) == 0) {
            const int qrow = a.Wx >> 2;
            for (int i = tid; i < rowsx * qrow; i += WG) {
                const int r = i / qrow, xq = i - r * qrow;
                const int e = (y0 + r) * a.Wx + xq * 4;
                f32x4 sc;
#pragma unroll
                for (int j = 0; j < 4; ++j) sc[j] = score_s[r * W + xq * 4 + j];
                if (a.score_out) *(f32x4 *)(a.score_out + lat_off + e) = sc;
                if (!a.do_update) continue;
                float z[4] = {0.f, 0.f, 0.f, 0.f};
                if (a.add_noise) {
                    if (a.noise) {
                        const f32x4 zv = *(const f32x4 *)(a.noise + lat_off + e);
#pragma unroll
                        for (int j = 0; j < 4; ++j) z[j] = zv[j];
                    } else {
                        normal4(a.ph.seed, a.ph.first_global_index + b, (uint32_t)(e >> 2), a.ph.t, z);
                    }
                }
                f32x4 xv = *(f32x4 *)(xg + e);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = __fadd_rn(__fmul_rn(a.c1, xv[j]), __fmul_rn(a.beta, sc[j]));
                    if (a.add_noise) v = __fadd_rn(v, __fmul_rn(z[j], a.sd));
                    xv[j] = v;
                }
                *(f32x4 *)(xg + e) = xv;
            }
        } else {
            for (int i = tid; i < rowsx * a.Wx; i += WG) {
                const int r = i / a.Wx, xc = i - r * a.Wx;
                const int e = (y0 + r) * a.Wx + xc;
                const float sc = score_s[r * W + xc];
                if (a.score_out) a.score_out[lat_off + e] = sc;
                if (!a.do_update) continue;
                float zz = 0.0f;
                if (a.add_noise) {
                    if (a.noise) zz = a.noise[lat_off + e];
                    else {
                        float z[4];
                        normal4(a.ph.seed, a.ph.first_global_index + b, (uint32_t)(e >> 2), a.ph.t, z);
                        zz = z[e & 3];
                    }
                }
                float v = __fadd_rn(__fmul_rn(a.c1, xg[e]), __fmul_rn(a.beta, sc));
                if (a.add_noise) v = __fadd_rn(v, __fmul_rn(zz, a.sd));
                xg[e] = v;
            }
        }
    } else {
        char *stage = lds_in;  // [rows2][W] positions of SO bytes
        auto epi2 = [&](const f32x16 &acc, int mt, int r, int x, bool valid) {
            if (!valid) return;
            char *dst = stage + (r * W + x) * SO;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = mt * 32 + 8 * g + 4 * h;
                const f32x4 bv = *(const f32x4 *)(a.b2 + co);
                float z[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) z[j] = acc[4 * g + j] + bv[j];
                if (a.s2) {
                    const f32x4 sv = *(const f32x4 *)(a.s2 + co), hv = *(const f32x4 *)(a.h2 + co);
#pragma unroll
                    for (int j = 0; j < 4; ++j) z[j] = z[j] * sv[j] + hv[j];
                }
                *(bf16x4 *)(dst + co * 2) = pack4(fsilu(z[0]), fsilu(z[1]), fsilu(z[2]), fsilu(z[3]));
            }
        };
        conv_stage<C1, 0, false, C2, NT2, MS2>(lds_mid, nullptr, W, 0, npix2, (const bf16x8 *)a.w2, wave, lane, epi2);
        __syncthreads();
        constexpr int CH = C2 / 8;
        __bf16 *og = (__bf16 *)a.out + ((size_t)b * H + y0) * W * C2;
        for (int i = tid; i < npix2 * CH; i += WG) {
            const int ch = i % CH, pos = i / CH;
            *(bf16x8 *)(og + (size_t)pos * C2 + ch * 8) = *(const bf16x8 *)(stage + pos * SO + ch * 16);
        }
        if constexpr (EPI2 == E2_STORE_POOL) {
            const int Wh = W >> 1, rh = rows2 >> 1;
            __bf16 *pg = (__bf16 *)a.out_pool + ((size_t)b * (H >> 1) + (y0 >> 1)) * Wh * C2;
            for (int i = tid; i < rh * Wh * CH; i += WG) {
                const int ch = i % CH, pos = i / CH;
                const int xh = pos % Wh, yh = pos / Wh;
                const char *q = stage + ((2 * yh) * W + 2 * xh) * SO + ch * 16;
                const bf16x8 t0 = *(const bf16x8 *)q, t1 = *(const bf16x8 *)(q + SO);
                const bf16x8 t2 = *(const bf16x8 *)(q + W * SO), t3 = *(const bf16x8 *)(q + W * SO + SO);
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    o[j] = (__bf16)((((float)t0[j] + (float)t1[j]) + (float)t2[j] + (float)t3[j]) * 0.25f);
                *(bf16x8 *)(pg + (size_t)pos * C2 + ch * 8) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Host side: LDS sizing and launch.
// ---------------------------------------------------------------------------------------------
namespace {

struct KindInfo { int mode1, ca, cb, c1, c2, epi2; };
constexpr KindInfo kKinds[5] = {
    {M1_FIRST, 1, 0, 32, 32, E2_STORE_POOL},    // FK_A : x -> conv0 -> conv1 -> skip1, pooled
    {M1_PLAIN, 32, 0, 64, 64, E2_STORE_POOL},   // FK_M1: pooled -> conv2 -> conv3 -> skip2, pooled
    {M1_PLAIN, 64, 0, 128, 128, E2_STORE},      // FK_M2: pooled -> conv4 (+time) -> conv5
    {M1_UPCAT, 128, 64, 64, 64, E2_STORE},      // FK_M3: [up(conv5), skip2] -> conv6 -> conv7
    {M1_UPCAT, 64, 32, 32, 32, E2_HEAD},        // FK_B : [up(conv7), skip1] -> conv8 -> conv9 -> head -> update
};

inline int align16(int v) { return (v + 15) & ~15; }

}  // namespace

void fused_lds_bytes(int kind, int W, int R, int *in_bytes, int *mid_bytes)
{
    const KindInfo &k = kKinds[kind];
    const int sa = k.ca * 2 + 16, sb = k.cb * 2 + 16, sm = k.c1 * 2 + 16, so = k.c2 * 2 + 16;
    int in;
    if (k.mode1 == M1_FIRST) in = (R + 4) * (W + 2) * 4;
    else if (k.mode1 == M1_PLAIN) in = (R + 4) * (W + 2) * sa;
    else in = align16((R / 2 + 2) * (W / 2 + 2) * sa) + (R + 4) * (W + 2) * sb;
    const int stage = k.epi2 == E2_HEAD ? R * W * 4 : R * W * so;
    *in_bytes = align16(in > stage ? in : stage);
    *mid_bytes = align16((R + 2) * (W + 2) * sm);
}

template <class K>
static hipError_t launch_one(K kernel, const FusedArgs &a, int64_t B, hipStream_t s)
{
    const int lds = a.lds_in_bytes + a.lds_mid_bytes;
    static thread_local int configured = 0;
    if (configured < lds) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        configured = 160 * 1024;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)(B * a.nbands)), dim3(WG), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_fused(int kind, const FusedArgs &a, int64_t B, hipStream_t s)
{
    switch (kind) {
    case FK_A: return launch_one(fused2_kernel<M1_FIRST, 1, 0, 32, 32, E2_STORE_POOL, 1, 1, 2, 1>, a, B, s);
    case FK_M1: return launch_one(fused2_kernel<M1_PLAIN, 32, 0, 64, 64, E2_STORE_POOL, 2, 2, 2, 2>, a, B, s);
    case FK_M2: return launch_one(fused2_kernel<M1_PLAIN, 64, 0, 128, 128, E2_STORE, 2, 1, 2, 1>, a, B, s);
    case FK_M3: return launch_one(fused2_kernel<M1_UPCAT, 128, 64, 64, 64, E2_STORE, 3, 1, 2, 1>, a, B, s);
    default: return launch_one(fused2_kernel<M1_UPCAT, 64, 32, 32, 32, E2_HEAD, 3, 1, 2, 1>, a, B, s);
    }
}
