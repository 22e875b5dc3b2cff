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

struct GnUpSrc {
    const act16 *skip;    // (B, H, W, CA - Cu); the kernel's `in` is up, (B, H / 2, W / 2, Cu)
    const float *ab;      // (B, CA, 2)
    int Cu;
};

template <int CA, int COUT, int NT, int MS, int WD, int NP, typename OT, bool GNUP = false>
__global__ __launch_bounds__(256) void tile_conv_f32out_kernel(const __bf16 *__restrict__ in, const bf16x8 *__restrict__ w,
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
    if constexpr (GNUP) {
        constexpr int CH = CA / 8;
        const int Cu = gu.Cu, Cs = CA - Cu, Hh = H >> 1, Wh = W >> 1, ipr = W * CH, total = (rows + 2) * ipr;
        const act16 *ub = (const act16 *)in + (size_t)b * Hh * Wh * Cu, *sb = gu.skip + (size_t)b * H * W * Cs;
        bf16x8 zero;
        a16x8 zraw;
#pragma unroll
        for (int j = 0; j < 8; ++j) { zero[j] = (__bf16)0.0f; zraw[j] = (act16)0.0f; }
        // items (tile row, column, eight channels), in rounds of U per thread: a round's loads are all in flight before its
        // first item is used (all NP at once would cost the 24-item instances their second workgroup per CU)
        constexpr int U = 8;
        auto item = [&](int i, int &t, int &xc, int &ch) {
            t = i / ipr;
            const int rem = i - t * ipr;
            xc = rem / CH;
            ch = rem - xc * CH;
        };
        float *abt = cst + 2 * COUT;
        for (int i = tid; i < 2 * CA; i += 256) abt[i] = gu.ab[(size_t)b * 2 * CA + i];
        bool synced = false;
#pragma unroll
        for (int k0 = 0; k0 < NP; k0 += U) {
            a16x8 raw[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = tid + (k0 + u) * 256;
                int t, xc, ch;
                item(i, t, xc, ch);
                const int gy = y0 - 1 + t;
                const bool live = k0 + u < NP && i < total && gy >= 0 && gy < H;
                const act16 *src = 8 * ch < Cu ? ub + (size_t)((gy >> 1) * Wh + (xc >> 1)) * Cu + 8 * ch
                                               : sb + (size_t)(gy * W + xc) * Cs + (8 * ch - Cu);
                raw[u] = live ? *(const a16x8 *)src : zraw;
            }
            if (!synced) { __syncthreads(); synced = true; }   // the (a, b) table
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = tid + (k0 + u) * 256;
                if (k0 + u >= NP || i >= total) break;
                int t, xc, ch;
                item(i, t, xc, ch);
                const int gy = y0 - 1 + t;
                bf16x8 o = zero;
                if (gy >= 0 && gy < H) {
                    const float *abp = abt + 16 * ch;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const f32x4 kk = *(const f32x4 *)(abp + 2 * j);   // a, b of channels j and j + 1
                        const float z0 = __builtin_fmaf((float)raw[u][j], kk[0], kk[1]),
                                    z1 = __builtin_fmaf((float)raw[u][j + 1], kk[2], kk[3]);
                        o[j] = (__bf16)(z0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z0 * -1.4426950408889634f)));
                        o[j + 1] = (__bf16)(z1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z1 * -1.4426950408889634f)));
                    }
                }
                *(bf16x8 *)(tile + (t * (W + 2) + xc + 1) * SA + ch * 16) = o;
            }
        }
        clear_halo<CA, 256>(tile, W, rows + 2, 0, 0, tid);
    } else {
        TileRegs<CA, NP, 256> regs;
        regs.load(in + (size_t)b * H * W * CA, H, W, y0 - 1, rows + 2, tid);
        regs.store(tile, W, rows + 2, tid);
    }
    __syncthreads();
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
    });
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
        return (int64_t)(r + 2) * a.W * (CA / 8) <= (int64_t)NP * 256 &&
               (r + 2) * (a.W + 2) * SA + 2 * COUT * 4 + (GNUP ? 2 * CA * 4 : 0) + 16 <= budget;
    };
    while (!fits(R)) {
        if (R == 1) return false;
        ++nb;
        R = (a.H + nb - 1) / nb;
    }
    nb = (a.H + R - 1) / R;
    if (a.B * nb > 0x7fffffff) return false;
    const int lds = (((R + 2) * (a.W + 2) * SA + 15) & ~15) + 2 * COUT * 4 + (GNUP ? 2 * CA * 4 : 0);
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
