// LDS-tiled bf16 3x3 convolution for small maps (width <= 32) with a raw float32 output: out = conv(in) + bias (+ a
// per-channel vector), no activation.  This is the convolution of the MNIST bit-flip UNet's ResBlocks in its bf16 mode
// (GroupNorm + swish produce the bf16 input, the float32 output feeds the next GroupNorm's statistics and the residual
// sum; mnist_diffusion/mnist_diffusion.py:168-193), on the sliding-window MFMA stage the Ising kernels use: a workgroup
// stages R + 2 rows of one sample in LDS ([row][col][channel], zero halo), its four wavefronts split the R output rows
// (or, for maps narrower than 32, the 32-pixel groups of several rows) and each MFMA tile's accumulators leave as
// float32, 16 bytes per lane and channel quad.  One workgroup per (sample, band); weights come from L2 through the
// stage's register ring.
#include "idm_conv_stage.h"

#include <algorithm>

namespace {

template <int CA, int COUT, int NT, int MS, int WD, int NP>
__global__ __launch_bounds__(256) void tile_conv_f32out_kernel(const __bf16 *__restrict__ in, const bf16x8 *__restrict__ w,
                                                              const float *__restrict__ bias, const float *__restrict__ tadd,
                                                              const float *__restrict__ resid, float *__restrict__ out, int H,
                                                              int W, int R, int nbands)
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
    TileRegs<CA, NP, 256> regs;
    regs.load(in + (size_t)b * H * W * CA, H, W, y0 - 1, rows + 2, tid);
    regs.store(tile, W, rows + 2, tid);
    using Conv = ConvStage<CA, 0, false, COUT, NT, MS, WD, true>;
    Conv cs;
    const LaneGeo geo = make_geo(W, lane & 31);
    cs.begin(tile, nullptr, cst, cst + COUT, geo, 0, rows, w, wave, lane);
    __syncthreads();
    float *ob = out + ((size_t)b * H + y0) * W * COUT;
    // resid (same shape as out, may be null): added last, out = (conv + bias + vector) + resid - a ResBlock's identity shortcut
    const float *rb = resid ? resid + ((size_t)b * H + y0) * W * COUT : nullptr;
    cs.run([&](const f32x16 &acc, const f32x4 *, const f32x4 *kt, int mt, int r, int x, bool valid) {
        if (!valid) return;
        const size_t o = ((size_t)r * W + x) * COUT + mt * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 z = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            z = z + kt[g];
            if (rb) z = z + *(const f32x4 *)(rb + o + 8 * g);
            *(f32x4 *)(ob + o + 8 * g) = z;
        }
    });
}

template <int CA, int COUT, int NT, int MS, int WD, int NP>
bool launch_tile(const ConvArgs &a, hipStream_t s)
{
    static_assert(BIAS_IN_ACC, "the epilogue expects the bias in the accumulators");
    constexpr int SA = CA * 2 + 16;
    // the tallest band whose tile fits NP x 256 staged items and 76 KB of LDS (two workgroups per CU)
    int nb = 1, R = a.H;
    auto fits = [&](int r) {
        return (int64_t)(r + 2) * a.W * (CA / 8) <= (int64_t)NP * 256 && (r + 2) * (a.W + 2) * SA + 2 * COUT * 4 + 16 <= 76 * 1024;
    };
    while (!fits(R)) {
        if (R == 1) return false;
        ++nb;
        R = (a.H + nb - 1) / nb;
    }
    nb = (a.H + R - 1) / R;
    if (a.B * nb > 0x7fffffff) return false;
    const int lds = (((R + 2) * (a.W + 2) * SA + 15) & ~15) + 2 * COUT * 4;
    auto kernel = tile_conv_f32out_kernel<CA, COUT, NT, MS, WD, NP>;
    static thread_local bool configured[IDM_MAX_DEVICES] = {};
    const int slot = idm_device_slot();
    if (slot < 0) return false;
    if (!configured[slot]) {
        if (hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) return false;
        configured[slot] = true;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)(a.B * nb)), dim3(256), lds, s, (const __bf16 *)a.in0, (const bf16x8 *)a.wpack, a.bias,
                       a.tadd, a.resid, (float *)a.out, a.H, a.W, R, nb);
    return true;
}

}  // namespace

// true: launched.  false: shape not covered (the caller falls back to the per-layer kernel).
bool launch_conv_tile_bf16(const ConvArgs &a, hipStream_t s)
{
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
