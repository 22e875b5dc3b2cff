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

#include <algorithm>
#include <type_traits>

namespace {

constexpr int WG = 256;

enum { M1_PLAIN = 0, M1_UPCAT = 1, M1_FIRST = 2 };
enum { E2_STORE = 0, E2_STORE_POOL = 1, E2_HEAD = 2 };

#ifdef IDM_ABL_NOEPI
DEVI float fsilu(float y) { return y; }
#else
DEVI float fsilu(float y) { return y * __builtin_amdgcn_rcpf(1.0f + __expf(-y)); }
#endif

DEVI bf16x4 pack4(float a, float b, float c, float d)
{
    bf16x4 o = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
    return o;
}

template <int I, int N, class F>
DEVI void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// One convolution stage over `npix` output pixels (pixel p -> row r0 + p / W, column p % W of the
// stage's output region).  NT pixel tiles and MS cout tiles are accumulated together per wave.
//   PLAIN part: operand for tap (ky,kx), channel group cg at  tile + ((r+ky)*(W+2) + x+kx)*S + 32*cg
//   UP part:    the same on the half-resolution tile at row (r+ky)>>1, column (x+kx+1)>>1
// The 9*NCG K-steps are fully unrolled so that every register index is static.  Activation
// fragments run PD steps ahead of the MFMAs in a small register ring (LDS latency), weights one
// whole tap ahead in a second register set (L2 latency); per-lane LDS addresses are sums of
// values computed once per tile, so the loop body holds no integer multiplies.
template <int CA, int CB, bool UP, int COUT, int NT, int MS, class Epi>
DEVI void conv_stage(const char *lds_a, const char *lds_b, int W, int r0, int npix, const bf16x8 *__restrict__ wp,
                     int wave, int lane, Epi &&epi)
{
    constexpr int MTOT = COUT / 32, NSL = MTOT / MS;
    constexpr int SA = CA * 2 + 16, SB = CB * 2 + 16;
    constexpr int NCG_A = CA / 16, NCG_B = CB / 16, NCG = NCG_A + NCG_B;
    constexpr int NSTEP = 9 * NCG;
    constexpr int PD = 2;
    constexpr int WC = 6 / MS;  // K-steps per weight chunk (divides 9*NCG for every layer shape)
    static_assert(NSTEP % WC == 0, "weight chunk must divide the K loop");
    const int ntiles = (npix + 31) >> 5, ngroups = (ntiles + NT - 1) / NT;
    const int h = lane >> 5, li = lane & 31;
    const int Wp2 = W + 2, Wh2 = (W >> 1) + 2;
    wp += lane;

#pragma unroll 1
    for (int u = wave; u < ngroups * NSL; u += 4) {
        const int grp = u / NSL, sl = u - grp * NSL;
        int pr[NT], px[NT];
        bool valid[NT];
        int base_a[NT], base_b[NT], row_a[3][NT], col_a[3][NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int p = (grp * NT + n) * 32 + li;
            valid[n] = p < npix;
            const int pc = valid[n] ? p : 0;
            const int q = pc / W;
            pr[n] = q + r0;
            px[n] = pc - q * W;
            if constexpr (UP) {
                const int rs = Wh2 * SA;
                row_a[0][n] = (pr[n] >> 1) * rs;
                row_a[1][n] = row_a[0][n] + (pr[n] & 1) * rs;
                row_a[2][n] = row_a[0][n] + rs;
                col_a[0][n] = ((px[n] + 1) >> 1) * SA + h * 16;
                col_a[1][n] = ((px[n] >> 1) + 1) * SA + h * 16;
                col_a[2][n] = col_a[0][n] + SA;
            } else {
                base_a[n] = (pr[n] * Wp2 + px[n]) * SA + h * 16;
            }
            base_b[n] = (pr[n] * Wp2 + px[n]) * SB + h * 16;
        }
        (void)base_a; (void)base_b; (void)row_a; (void)col_a;
        f32x16 acc[MS][NT];
#pragma unroll
        for (int m = 0; m < MS; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

        // The slice offset is made opaque to the optimiser: otherwise the weight loads are
        // loop-invariant across tile groups and get hoisted out of this loop into (far too many)
        // registers.  (Laundering the offset, not the pointer, keeps the loads global_load: a
        // laundered pointer degrades to flat_load, which also counts on lgkmcnt and turns every
        // LDS wait into a wait for the weight prefetch.)
        int woff = sl * MS * 64;
        asm volatile("" : "+v"(woff));
        const bf16x8 *wsl = wp + woff;
        bf16x8 wreg[2][WC][MS];
        bf16x8 ring[PD + 1][NT];
        auto load_w = [&](auto CHUNK) {
            constexpr int ch = decltype(CHUNK)::value;
            const bf16x8 *wt = wsl + (size_t)(ch * WC) * MTOT * 64;
#pragma unroll
            for (int k = 0; k < WC; ++k)
#pragma unroll
                for (int m = 0; m < MS; ++m) wreg[ch & 1][k][m] = wt[(k * MTOT + m) * 64];
        };
        auto read_b = [&](auto S) {
            constexpr int st = decltype(S)::value;
            constexpr int tap = st / NCG, cg = st % NCG, ky = tap / 3, kx = tap % 3, slot = st % (PD + 1);
            if constexpr (cg < NCG_A) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    int addr;
                    if constexpr (UP) addr = row_a[ky][n] + col_a[kx][n];
                    else addr = base_a[n] + (ky * Wp2 + kx) * SA;
                    ring[slot][n] = *(const bf16x8 *)(lds_a + addr + cg * 32);
                }
            } else {
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    ring[slot][n] = *(const bf16x8 *)(lds_b + base_b[n] + (ky * Wp2 + kx) * SB + (cg - NCG_A) * 32);
            }
        };
        load_w(std::integral_constant<int, 0>{});
        static_for<0, PD>([&](auto S) { read_b(S); });
        static_for<0, NSTEP>([&](auto S) {
            constexpr int st = decltype(S)::value;
            constexpr int ch = st / WC, k = st % WC;
            // pin the software pipeline: without this the scheduler hoists every load of the
            // unrolled K loop to the top and spills
            __builtin_amdgcn_sched_barrier(0);
#ifdef IDM_ABL_NOW
            if constexpr (k == 0 && ch == 0) load_w(std::integral_constant<int, 1>{});
#else
            if constexpr (k == 0 && (ch + 1) * WC < NSTEP) load_w(std::integral_constant<int, ch + 1>{});
#endif
#ifdef IDM_ABL_NOB
            if constexpr (st == 0) read_b(std::integral_constant<int, PD>{});
#else
            if constexpr (st + PD < NSTEP) read_b(std::integral_constant<int, st + PD>{});
#endif
#pragma unroll
            for (int m = 0; m < MS; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#ifdef IDM_ABL_NOMFMA
                    acc[m][n][0] += (float)wreg[ch & 1][k][m][0] + (float)ring[st % (PD + 1)][n][0];
#else
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[ch & 1][k][m], ring[st % (PD + 1)][n], acc[m][n], 0, 0, 0);
#endif
        });
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

// Register-held payload of one LDS input tile: NP 16-byte items per thread (item i = tid + k*WG
// covers row i / (W*C/8), 8 channels).  load() reads rows [g0, g0+nrows) of a global NHWC bf16 map
// (zeros outside [0,Hmap)); store() writes them into the tile (row t = global row g0+t) and clears
// the tile's two halo columns.  Splitting the two lets the loads of the next band fly while the
// current band is still being computed.
template <int C, int NP>
struct TileRegs {
    static constexpr int S = C * 2 + 16, CH = C / 8;
    bf16x8 v[NP];

    DEVI void load(const __bf16 *__restrict__ src, int Hmap, int W, int g0, int nrows, int tid)
    {
        const int ipr = W * CH;
        int row = tid / ipr, rem = tid - row * ipr;
        const int drow = WG / ipr, drem = WG - drow * ipr;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            bf16x8 z;
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.0f;
            const int g = g0 + row;
#ifndef IDM_ABL_NOSTAGE
            if (row < nrows && g >= 0 && g < Hmap) z = *(const bf16x8 *)(src + ((size_t)g * W + rem / CH) * C + (rem % CH) * 8);
#endif
            v[k] = z;
            rem += drem;
            row += drow;
            if (rem >= ipr) { rem -= ipr; ++row; }
        }
    }

    DEVI void store(char *tile, int W, int nrows, int tid) const
    {
        const int ipr = W * CH;
        int row = tid / ipr, rem = tid - row * ipr;
        const int drow = WG / ipr, drem = WG - drow * ipr;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            if (row < nrows) *(bf16x8 *)(tile + (row * (W + 2) + rem / CH + 1) * S + (rem % CH) * 16) = v[k];
            rem += drem;
            row += drow;
            if (rem >= ipr) { rem -= ipr; ++row; }
        }
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.0f;
        for (int i = tid; i < nrows * 2 * CH; i += WG) {
            const int ch = i % CH, side = (i / CH) & 1, r = i / (2 * CH);
            *(bf16x8 *)(tile + (r * (W + 2) + side * (W + 1)) * S + ch * 16) = z;
        }
    }
};

// Zeroes the halo columns of rows [0,nrows) and the whole of rows [z0,z1) of an LDS tile.
template <int C>
DEVI void clear_halo(char *tile, int W, int nrows, int z0, int z1, int tid)
{
    constexpr int S = C * 2 + 16, CH = C / 8;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < nrows * 2 * CH; i += WG) {
        const int ch = i % CH, side = (i / CH) & 1, r = i / (2 * CH);
        *(f32x4 *)(tile + (r * (W + 2) + side * (W + 1)) * S + ch * 16) = z;
    }
    if (z1 > z0) {
        char *p = tile + z0 * (W + 2) * S;
        const int bytes = (z1 - z0) * (W + 2) * S;  // S is a multiple of 16
        for (int i = tid * 16; i < bytes; i += WG * 16) *(f32x4 *)(p + i) = z;
    }
}

}  // namespace

// Persistent: each workgroup walks bands wg = blockIdx.x, blockIdx.x + gridDim.x, ...
template <int MODE1, int CA, int CB, int C1, int C2, int EPI2, int NT1, int MS1, int NT2, int MS2, int NPA, int NPB>
__global__ __launch_bounds__(WG, 2) void fused2_kernel(FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5;
    const int W = a.W, H = a.H, R = a.R;
    constexpr int SA = CA * 2 + 16, SB = CB * 2 + 16, SM = C1 * 2 + 16, SO = C2 * 2 + 16;
    constexpr int CAT = MODE1 == M1_FIRST ? 8 : CA, CBT = MODE1 == M1_UPCAT ? CB : 8;
    (void)SA; (void)SB; (void)SO;
    const int64_t total = a.B * a.nbands;

    // LDS map: [input region | mid tile]; the output staging / score buffer reuses the input region.
    char *lds_in = smem;
    char *lds_mid = smem + a.lds_in_bytes;
    char *tile_a = lds_in;
    char *tile_b = MODE1 == M1_UPCAT ? lds_in + ((((R >> 1) + 2) * ((W >> 1) + 2) * SA + 15) & ~15) : nullptr;

    TileRegs<CAT, MODE1 == M1_FIRST ? 1 : NPA> ra;
    TileRegs<CBT, MODE1 == M1_UPCAT ? NPB : 1> rb;
    float rx[MODE1 == M1_FIRST ? NPA : 1];
    (void)ra; (void)rb; (void)rx;

    auto pf_load = [&](int64_t wg) {
        const int64_t b = wg / a.nbands;
        const int y0 = (int)(wg - b * a.nbands) * R;
        if constexpr (MODE1 == M1_FIRST) {
            const float *x = (const float *)a.in_a + (size_t)b * a.Hx * a.Wx;
#pragma unroll
            for (int k = 0; k < NPA; ++k) {
                const int i = tid + k * WG, t = i / W, c = i - t * W, g = y0 - 2 + t;
                rx[k] = (t < R + 4 && g >= 0 && g < a.Hx && c < a.Wx) ? x[(size_t)g * a.Wx + c] : 0.0f;
            }
        } else if constexpr (MODE1 == M1_PLAIN) {
            ra.load((const __bf16 *)a.in_a + (size_t)b * H * W * CA, H, W, y0 - 2, R + 4, tid);
        } else {
            const int Hh = H >> 1, Wh = W >> 1;
            ra.load((const __bf16 *)a.in_a + (size_t)b * Hh * Wh * CA, Hh, Wh, (y0 >> 1) - 1, (R >> 1) + 2, tid);
            rb.load((const __bf16 *)a.in_b + (size_t)b * H * W * CB, H, W, y0 - 2, R + 4, tid);
        }
    };
    auto pf_store = [&]() {
        if constexpr (MODE1 == M1_FIRST) {
            float *xs = (float *)lds_in;
#pragma unroll
            for (int k = 0; k < NPA; ++k) {
                const int i = tid + k * WG, t = i / W, c = i - t * W;
                if (t < R + 4) xs[t * (W + 2) + c + 1] = rx[k];
            }
            for (int i = tid; i < (R + 4) * 2; i += WG) xs[(i >> 1) * (W + 2) + (i & 1) * (W + 1)] = 0.0f;
        } else if constexpr (MODE1 == M1_PLAIN) {
            ra.store(tile_a, W, R + 4, tid);
        } else {
            ra.store(tile_a, W >> 1, (R >> 1) + 2, tid);
            rb.store(tile_b, W, R + 4, tid);
        }
    };

    clear_halo<C1>(lds_mid, W, R + 2, 0, 0, tid);  // halo columns of the mid tile: never written afterwards
    int64_t wg = blockIdx.x;
    if (wg < total) pf_load(wg);

    for (; wg < total; wg += gridDim.x) {
        const int64_t b = wg / a.nbands;
        const int y0 = (int)(wg - b * a.nbands) * R;
        const int rlo = y0 == 0 ? 1 : 0;
        const int rhi = min(R + 2, H - y0 + 1);

        pf_store();
        // rows of the mid tile outside the map are conv2's zero padding (an earlier band may have left data there)
        if (rlo) clear_halo<C1>(lds_mid, W, 0, 0, 1, tid);
        if (rhi < R + 2) clear_halo<C1>(lds_mid, W, 0, rhi, R + 2, tid);
        __syncthreads();

        // ---- conv1 on rows y0-1 .. y0+R (clipped to the map) -> mid tile row (g - (y0-1)) ----
        {
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
                conv_stage<CA, CB, MODE1 == M1_UPCAT, C1, NT1, MS1>(tile_a, tile_b, W, rlo, npix, (const bf16x8 *)a.w1,
                                                                    wave, lane, epi1);
        }
        __syncthreads();

        // ---- conv2 on rows y0 .. y0+R-1 (clipped) ----
        const int rows2 = min(R, H - y0);
        const int npix2 = rows2 * W;
        // the last iteration reloads its own band (discarded): an unconditional load keeps the prefetch
        // registers dead across conv1/conv2 instead of loop-carried
        const int64_t wg_next = wg + gridDim.x < total ? wg + gridDim.x : wg;
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

            // This band's x is fetched BEFORE the next band's input: vector-memory results return in
            // issue order, so the update below then never waits for the prefetch.
#ifdef IDM_ABL_NOOUT
            const int rowsx = tid == 999 ? 1 : 0;
#else
            const int rowsx = min(rows2, a.Hx - y0);
#endif
            float *xg = a.x + (size_t)b * a.Hx * a.Wx;
            const size_t lat_off = (size_t)b * a.Hx * a.Wx;
            const int qrow = a.Wx >> 2;
            const bool quad_path = (a.Wx & 3) == 0 && rowsx * qrow <= WG;
            const int qr = tid / max(qrow, 1), qx = tid - qr * qrow;
            const int qe = (y0 + qr) * a.Wx + qx * 4;
            const bool qact = quad_path && tid < rowsx * qrow;
            f32x4 xv = {0.f, 0.f, 0.f, 0.f};
            if (qact && a.do_update) xv = *(const f32x4 *)(xg + qe);
            pf_load(wg_next);
            __syncthreads();
            if (quad_path) {
                if (qact) {
                    f32x4 sc;
#pragma unroll
                    for (int j = 0; j < 4; ++j) sc[j] = score_s[qr * W + qx * 4 + j];
                    if (a.score_out) *(f32x4 *)(a.score_out + lat_off + qe) = sc;
                    if (a.do_update) {
                        float z[4] = {0.f, 0.f, 0.f, 0.f};
                        if (a.add_noise) {
                            if (a.noise) {
                                const f32x4 zv = *(const f32x4 *)(a.noise + lat_off + qe);
#pragma unroll
                                for (int j = 0; j < 4; ++j) z[j] = zv[j];
                            } else {
                                normal4(a.ph.seed, a.ph.first_global_index + b, (uint32_t)(qe >> 2), a.ph.t, z);
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float v = __fadd_rn(__fmul_rn(a.c1, xv[j]), __fmul_rn(a.beta, sc[j]));
                            if (a.add_noise) v = __fadd_rn(v, __fmul_rn(z[j], a.sd));
                            xv[j] = v;
                        }
                        *(f32x4 *)(xg + qe) = xv;
                    }
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
            // next band's input: issued after this wave's last weight wait, so the HBM latency overlaps
            // the copy-out instead of stalling the weight pipeline (vector-memory results return in order)
            pf_load(wg_next);
            __syncthreads();
            constexpr int CH = C2 / 8;
            __bf16 *og = (__bf16 *)a.out + ((size_t)b * H + y0) * W * C2;
#ifdef IDM_ABL_NOOUT
            if (tid == 999)
#endif
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
        __syncthreads();  // staging / score consumed before the next band's input overwrites the region
    }
}

// ---------------------------------------------------------------------------------------------
// Host side: LDS sizing and launch.
// ---------------------------------------------------------------------------------------------
namespace {

// npa / npb: 16-byte items (FK_A: floats) per thread the kernel instance can hold in registers
// for the next band's input tiles.
struct KindInfo { int mode1, ca, cb, c1, c2, epi2, npa, npb; };
constexpr KindInfo kKinds[5] = {
    {M1_FIRST, 1, 0, 32, 32, E2_STORE_POOL, 2, 0},    // FK_A : x -> conv0 -> conv1 -> skip1, pooled
    {M1_PLAIN, 32, 0, 64, 64, E2_STORE_POOL, 5, 0},   // FK_M1: pooled -> conv2 -> conv3 -> skip2, pooled
    {M1_PLAIN, 64, 0, 128, 128, E2_STORE, 3, 0},      // FK_M2: pooled -> conv4 (+time) -> conv5
    {M1_UPCAT, 128, 64, 64, 64, E2_STORE, 3, 6},      // FK_M3: [up(conv5), skip2] -> conv6 -> conv7
    {M1_UPCAT, 64, 32, 32, 32, E2_HEAD, 3, 6},        // FK_B : [up(conv7), skip1] -> conv8 -> conv9 -> head -> update
};

inline int align16(int v) { return (v + 15) & ~15; }

}  // namespace

// True when the next band's input tiles fit the kernel instance's prefetch registers.
bool fused_band_fits(int kind, int W, int R)
{
    const KindInfo &k = kKinds[kind];
    if (k.mode1 == M1_FIRST) return (R + 4) * W <= k.npa * WG;
    if (k.mode1 == M1_PLAIN) return (R + 4) * W * (k.ca / 8) <= k.npa * WG;
    return (R / 2 + 2) * (W / 2) * (k.ca / 8) <= k.npa * WG && (R + 4) * W * (k.cb / 8) <= k.npb * WG;
}

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
    static thread_local bool configured = false;
    static thread_local int cus = 0, cached_lds = -1, cached_blocks = 0;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t prop;
        if ((e = hipGetDevice(&dev)) != hipSuccess || (e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
        cus = prop.multiProcessorCount;
        configured = true;
    }
    if (cached_lds != lds) {
        int n = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)kernel, WG, (size_t)lds);
        if (e != hipSuccess) return e;
        cached_blocks = n > 0 ? n : 1;
        cached_lds = lds;
    }
    const int64_t total = B * a.nbands;
    const int64_t grid = std::min<int64_t>(total, (int64_t)cus * cached_blocks);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(WG), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_fused(int kind, const FusedArgs &a, int64_t B, hipStream_t s)
{
    switch (kind) {
    case FK_A: return launch_one(fused2_kernel<M1_FIRST, 1, 0, 32, 32, E2_STORE_POOL, 1, 1, 2, 1, 2, 0>, a, B, s);
    case FK_M1: return launch_one(fused2_kernel<M1_PLAIN, 32, 0, 64, 64, E2_STORE_POOL, 2, 2, 2, 2, 5, 0>, a, B, s);
    case FK_M2: return launch_one(fused2_kernel<M1_PLAIN, 64, 0, 128, 128, E2_STORE, 2, 1, 2, 1, 3, 0>, a, B, s);
    case FK_M3: return launch_one(fused2_kernel<M1_UPCAT, 128, 64, 64, 64, E2_STORE, 3, 1, 2, 1, 3, 6>, a, B, s);
    default: return launch_one(fused2_kernel<M1_UPCAT, 64, 32, 32, 32, E2_HEAD, 3, 1, 2, 1, 3, 6>, a, B, s);
    }
}
