// LDS-tiled bf16 3x3 convolution for small maps (width <= 32) with a raw float32 output: out = conv(in) + bias (+ a
// per-channel vector), no activation.  This is the convolution of the MNIST bit-flip UNet's ResBlocks in its bf16 mode
// (GroupNorm + swish produce the bf16 input, the float32 output feeds the next GroupNorm's statistics and the residual
// sum; mnist_diffusion/mnist_diffusion.py:168-193), on the sliding-window MFMA stage the Ising kernels use: a workgroup
// stages R + 2 rows of one sample in LDS ([row][col][channel], zero halo), its four wavefronts split the R output rows
// (or, for maps narrower than 32, the 32-pixel groups of several rows) and each MFMA tile's accumulators leave as
// float32, 16 bytes per lane and channel quad.  One workgroup per (sample, band); weights come from L2 through the stage's
// register ring.
// GNUP instances (ConvArgs::io_f16 + gn_ab) work on IEEE-half activations: they take a ResBlock convolution's input as it
// is BEFORE the GroupNorm in front of it - a plain tensor or x = Concatenate([UpSampling2D(up), skip]) (:225-226, :230-231) -
// and apply y = swish(x a + b) with the (a, b) pairs of the sample's channels while they stage the tile, so the normalised
// tensor never goes through memory; output and residual are half too (stores saturate at +-65504).
#include "idm_conv_stage.h"

#include <algorithm>

namespace {

typedef _Float16 act16;
typedef __attribute__((ext_vector_type(4))) _Float16 a16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 a16x8;

#ifdef IDM_TC_TIMELINE   // timing build: clock stamps of wavefronts 0 and 3 of 64 mid-launch workgroups of the 96 -> 32 instance
__device__ long long g_tc[64][2][16];
#define TCS(i)                                                                                                  \
    do {                                                                                                        \
        if (CA == 96 && GNUP && lane == 0 && (wave == 0 || wave == 3) && blockIdx.x >= 4096 && blockIdx.x < 4160) \
            g_tc[blockIdx.x - 4096][wave == 3][i] = (i) == 15 ? (long long)wall_clock64() : (long long)clock64();  \
    } while (0)
#else
#define TCS(i) do { } while (0)
#endif

struct GnUpSrc {
    const act16 *skip;    // (B, H, W, CA - Cu); the kernel's `in` is up, (B, H / 2, W / 2, Cu)
    const float *ab;      // (B, CA, 2)
    int Cu;
};

template <int CA, int COUT, int NT, int MS, int WD, int NP, typename OT, bool GNUP = false>
__global__ __launch_bounds__(256, 2) void tile_conv_f32out_kernel(const __bf16 *__restrict__ in, const bf16x8 *__restrict__ w,
                                                              const float *__restrict__ bias, const float *__restrict__ tadd,
                                                              const OT *__restrict__ resid, OT *__restrict__ out, int H,
                                                              int W, int R, int nbands, GnUpSrc gu)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SA = CA * 2 + 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int64_t b = blockIdx.x / nbands;
    const int y0 = (int)(blockIdx.x % nbands) * R;
    const int rows = min(R, H - y0);
    char *tile = smem;
    TCS(0);
    TCS(15);
    float *cst = (float *)(smem + (((R + 2) * (W + 2) * SA + 15) & ~15));
    for (int i = tid; i < COUT; i += 256) {
        cst[i] = bias[i];
        cst[COUT + i] = tadd ? tadd[i] : 0.0f;
    }
    // (begin() only takes addresses and requests the first weight fragments: they arrive while the tile is staged)
    using Conv = ConvStage<CA, 0, false, COUT, NT, MS, WD, true>;
    Conv cs;
    const LaneGeo geo = make_geo(W, lane & 31);
    cs.begin(tile, nullptr, cst, cst + COUT, geo, 0, rows, w, wave, lane);
    TCS(1);
    if constexpr (GNUP) {
        // Items (tile pixel, eight channels).  A thread keeps ONE channel octet (tid % CH; 256 / CH pixels per step, the
        // 256 % CH last threads idle), so the (a, b) pairs of its channels sit in registers for all its pixels and whether it
        // reads up or skip is fixed: no table in LDS (clock stamps of the first version: four dependent LDS reads per item
        // cost about 1 000 cycles per item, 16 k of a workgroup's 39 k), no per-item channel arithmetic.  The pixel
        // (tile row t, column xc) is stepped from one item to the next.  Rounds of U items, software-pipelined: round r + 1's
        // loads are issued before round r is processed, so only the first round's latency is exposed (all NP at once would
        // cost the 24-item instances their second workgroup per CU).
        constexpr int CH = CA / 8, PPS = 256 / CH, U = 8, NR = (NP + U - 1) / U;
        const int Cu = gu.Cu, Cs = CA - Cu, Hh = H >> 1, Wh = W >> 1;
        const int ch = tid % CH, q0 = tid / CH;
        const bool act = q0 < PPS, from_up = 8 * ch < Cu;
        const act16 *ub = (const act16 *)in + (size_t)b * Hh * Wh * Cu + 8 * ch, *sb = gu.skip + (size_t)b * H * W * Cs + (8 * ch - Cu);
        f32x4 kab[4];   // (a, b) of channels 8 ch + 2 j, 8 ch + 2 j + 1
#pragma unroll
        for (int j = 0; j < 4; ++j) kab[j] = *(const f32x4 *)(gu.ab + (size_t)b * 2 * CA + 16 * ch + 4 * j);
        bf16x8 zero;
        a16x8 zraw;
#pragma unroll
        for (int j = 0; j < 8; ++j) { zero[j] = (__bf16)0.0f; zraw[j] = (act16)0.0f; }
        const int dt = PPS / W, dx = PPS - dt * W;
        auto step = [&](int &t, int &xc) {
            t += dt;
            xc += dx;
            if (xc >= W) { xc -= W; ++t; }
        };
        int tl = q0 / W, xl = q0 - tl * W;   // the next pixel to load
        int tp = tl, xp = xl;                 // the next pixel to process
        a16x8 raw[2][U];
        auto issue = [&](auto RR) {
            constexpr int r = decltype(RR)::value;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (r * U + u < NP) {
                    const int gy = y0 - 1 + tl;
                    const bool live = act && tl < rows + 2 && gy >= 0 && gy < H;
                    const act16 *src = from_up ? ub + (size_t)((gy >> 1) * Wh + (xl >> 1)) * Cu : sb + (size_t)(gy * W + xl) * Cs;
                    raw[r & 1][u] = live ? *(const a16x8 *)src : zraw;
                    step(tl, xl);
                }
            }
        };
        auto process = [&](auto RR) {
            constexpr int r = decltype(RR)::value;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (r * U + u < NP) {
                    const int gy = y0 - 1 + tp;
                    const bool inside = gy >= 0 && gy < H;
                    bf16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const f32x4 kk = kab[j >> 1];
                        const float z0 = __builtin_fmaf((float)raw[r & 1][u][j], kk[0], kk[1]),
                                    z1 = __builtin_fmaf((float)raw[r & 1][u][j + 1], kk[2], kk[3]);
                        o[j] = (__bf16)(z0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z0 * -1.4426950408889634f)));
                        o[j + 1] = (__bf16)(z1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z1 * -1.4426950408889634f)));
                    }
                    if (act && tp < rows + 2) *(bf16x8 *)(tile + (tp * (W + 2) + xp + 1) * SA + ch * 16) = inside ? o : zero;
                    step(tp, xp);
                }
            }
        };
        issue(std::integral_constant<int, 0>{});
        TCS(2);
        static_for<0, NR>([&](auto RR) {
            constexpr int r = decltype(RR)::value;
            if constexpr (r + 1 < NR) issue(std::integral_constant<int, r + 1>{});
            process(RR);
            TCS(3 + 2 * r);
        });
        clear_halo<CA, 256>(tile, W, rows + 2, 0, 0, tid);
        TCS(8);
    } else {
        TileRegs<CA, NP, 256> regs;
        regs.load(in + (size_t)b * H * W * CA, H, W, y0 - 1, rows + 2, tid);
        regs.store(tile, W, rows + 2, tid);
    }
    __syncthreads();
    TCS(9);
    OT *ob = out + ((size_t)b * H + y0) * W * COUT;
    // resid (same shape as out, may be null): added last, out = (conv + bias + vector) + resid - a ResBlock's identity shortcut
    const OT *rb = resid ? resid + ((size_t)b * H + y0) * W * COUT : nullptr;
    cs.run([&](const f32x16 &acc, const f32x4 *, const f32x4 *kt, int mt, int r, int x, bool valid) {
        if (!valid) return;
        const size_t o = ((size_t)r * W + x) * COUT + mt * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 z = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            z = z + kt[g];
            if constexpr (sizeof(OT) == 4) {
                if (rb) z = z + *(const f32x4 *)(rb + o + 8 * g);
                *(f32x4 *)(ob + o + 8 * g) = z;
            } else {
                if (rb) {
                    const a16x4 r4 = *(const a16x4 *)(rb + o + 8 * g);
                    z = z + f32x4{(float)r4[0], (float)r4[1], (float)r4[2], (float)r4[3]};
                }
                a16x4 o4;
#pragma unroll
                for (int j = 0; j < 4; ++j) o4[j] = (act16)__builtin_amdgcn_fmed3f(z[j], -65504.0f, 65504.0f);
                *(a16x4 *)(ob + o + 8 * g) = o4;
            }
        }
    }, [&](int i) { if (i == 0) TCS(10); if (i == 1) TCS(11); if (i == 2) TCS(12); });
    TCS(13);
#ifdef IDM_TC_TIMELINE
    __builtin_amdgcn_s_waitcnt(0);
    TCS(14);
#endif
}

template <int CA, int COUT, int NT, int MS, int WD, int NP, typename OT, bool GNUP = false>
bool launch_tile_as(const ConvArgs &a, hipStream_t s)
{
    static_assert(BIAS_IN_ACC, "the epilogue expects the bias in the accumulators");
    constexpr int SA = CA * 2 + 16;
    // the tallest band whose tile fits NP x 256 staged items and 76 KB of LDS (two workgroups per CU)
    int nb = 1, R = a.H;
    // (76 KB: two workgroups per CU; shorter bands for three or four - the one-slice instances have the registers for it -
    // measured no faster: 52 KB +0.3 %, 39 KB -2.6 %, 30 KB -13 % for the bit-flip sampler)
    constexpr int budget = 76 * 1024;
    auto fits = [&](int r) {
        // (GNUP: a thread stages one channel octet of 256 / (CA / 8) pixels per step)
        return (GNUP ? (int64_t)(r + 2) * a.W <= (int64_t)NP * (256 / (CA / 8)) : (int64_t)(r + 2) * a.W * (CA / 8) <= (int64_t)NP * 256) &&
               (r + 2) * (a.W + 2) * SA + 2 * COUT * 4 + 16 <= budget;
    };
    while (!fits(R)) {
        if (R == 1) return false;
        ++nb;
        R = (a.H + nb - 1) / nb;
    }
    nb = (a.H + R - 1) / R;
    if (a.B * nb > 0x7fffffff) return false;
    const int lds = (((R + 2) * (a.W + 2) * SA + 15) & ~15) + 2 * COUT * 4;
    auto kernel = tile_conv_f32out_kernel<CA, COUT, NT, MS, WD, NP, OT, GNUP>;
    static thread_local bool configured[IDM_MAX_DEVICES] = {};
    const int slot = idm_device_slot();
    if (slot < 0) return false;
    if (!configured[slot]) {
        if (hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) return false;
        configured[slot] = true;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)(a.B * nb)), dim3(256), lds, s, (const __bf16 *)a.in0, (const bf16x8 *)a.wpack, a.bias,
                       a.tadd, (const OT *)a.resid, (OT *)a.out, a.H, a.W, R, nb,
                       GnUpSrc{(const act16 *)a.in1, a.gn_ab, a.C0});
    return true;
}

template <int CA, int COUT, int NT, int MS, int WD, int NP>
bool launch_tile(const ConvArgs &a, hipStream_t s)
{
    return launch_tile_as<CA, COUT, NT, MS, WD, NP, float>(a, s);
}

}  // namespace

#ifdef IDM_TC_TIMELINE
extern "C" int idm_tc_timeline(long long *host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tc), sizeof(g_tc)); }
#endif

// true: launched.  false: shape not covered (the caller falls back to the per-layer kernel).
bool launch_conv_tile_bf16(const ConvArgs &a, hipStream_t s)
{
    if (a.mode == CONV_UPCAT && a.gn_ab && a.io_f16 && !a.silu && a.out_f32 && !a.scale && a.W <= 32 && a.W >= 4 && a.H >= 1 &&
        a.C0 % 8 == 0 && (a.C0 == 0 || !((a.W | a.H) & 1))) {
        // (C0 = 0: a plain tensor in in1)
        const int ci = a.C0 + a.C1, co = a.Cout;
        if (ci == 32 && co == 32) return launch_tile_as<32, 32, 3, 1, 2, 14, act16, true>(a, s);
        if (ci == 32 && co == 64) return launch_tile_as<32, 64, 2, 2, 2, 14, act16, true>(a, s);
        if (ci == 64 && co == 64) return launch_tile_as<64, 64, 2, 2, 2, 14, act16, true>(a, s);
        if (ci == 64 && co == 128) return launch_tile_as<64, 128, 2, 2, 2, 14, act16, true>(a, s);
        if (ci == 128 && co == 128) return launch_tile_as<128, 128, 2, 2, 2, 14, act16, true>(a, s);
        if (ci == 192 && co == 64) return launch_tile_as<192, 64, 2, 2, 2, 24, act16, true>(a, s);
        if (ci == 96 && co == 32) return launch_tile_as<96, 32, 3, 1, 2, 24, act16, true>(a, s);
        return false;
    }
    if (a.gn_ab || a.io_f16) return false;
    if (a.mode != CONV_PLAIN || a.C1 != 0 || a.silu || !a.out_f32 || a.scale || a.W > 32 || a.W < 4 || a.H < 1) return false;
    const int ci = a.C0, co = a.Cout;
    if (ci == 32 && co == 32) return launch_tile<32, 32, 3, 1, 2, 14>(a, s);
    if (ci == 32 && co == 64) return launch_tile<32, 64, 2, 2, 2, 14>(a, s);
    if (ci == 64 && co == 64) return launch_tile<64, 64, 2, 2, 2, 14>(a, s);
    if (ci == 64 && co == 128) return launch_tile<64, 128, 2, 2, 2, 14>(a, s);
    if (ci == 128 && co == 128) return launch_tile<128, 128, 2, 2, 2, 14>(a, s);
    if (ci == 192 && co == 64) return launch_tile<192, 64, 2, 2, 2, 24>(a, s);
    if (ci == 96 && co == 32) return launch_tile<96, 32, 3, 1, 2, 24>(a, s);
    return false;
}
