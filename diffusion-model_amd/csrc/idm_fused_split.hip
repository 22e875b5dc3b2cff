// Kernels of the L = 32 pyramid's bf16 schedule that are built around how gfx950 issues instructions - a wavefront issues one
// instruction per ~8 cycles whatever its kind, so whatever an MFMA wavefront does outside its loops idles the matrix pipe
// unless another wavefront of the SIMD has MFMAs in flight:
//   fused_m3_split_kernel   conv6 + conv7, eight wavefronts: four run the MFMA loops and their epilogues, four (the second
//                           wavefront of each SIMD) the tile copies beside them (the tiles leave room for one workgroup per CU)
//   fused_b_copy_kernel     conv8 + conv9 + head + update, eight wavefronts likewise, two lattice streams in alternating slots
//   fused_b_pair_kernel     the same with the two streams merged into one MFMA loop per convolution (the default)
//   fused_m1_mf16_kernel    conv2 + conv3, four wavefronts, two workgroups per CU, on the 16x16x32 MFMA stage
//   fused_m2_pair_kernel    conv4 + conv5 likewise, two lattices per pass (a weight fragment feeds both)
// This translation unit is compiled with -fno-slp-vectorize: the copy wavefronts' arithmetic must stay on the plain
// (unpacked) float32 ALU, because beside a wavefront that streams MFMAs packed float32 instructions wait for the matrix
// pipe while plain ones do not (inline assembly is no alternative: the compiler does not see that it consumes a
// transcendental's result and leaves out the wait state that needs).
#include "idm_conv_stage.h"
#include "idm_conv16.h"

#include <algorithm>
#include <cstdlib>

#ifndef IDM_M3_MFMA16_WD
#define IDM_M3_MFMA16_WD 2   // supersteps of conv6's weights in flight ahead of their MFMAs (16x16x32 instance)
#endif

#ifdef IDM_TIMELINE
// fused_b_copy_kernel: [kind of wavefront][iteration][2 k] = arrival at barrier k + 1, [2 k + 1] = departure from it
// (wavefronts 0 and 4 of one workgroup).
__device__ long long g_tl3[2][64][16];
extern "C" int idm_debug_timeline3(long long *host)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tl3), sizeof(g_tl3));
}
#define TLS(i)                                                                                 \
    do {                                                                                       \
        if (blockIdx.x == 7 && lane == 0 && (wave == 0 || wave == 4) && tls_n < 64)            \
            g_tl3[wave == 4][tls_n][i] = (long long)clock64();                                 \
    } while (0)
#define BAR(k) do { TLS(2 * (k) - 2); __syncthreads(); TLS(2 * (k) - 1); } while (0)
#else
#define BAR(k) __syncthreads()
#endif

// ---------------------------------------------------------------------------------------------
// fused_M3 (conv6 + conv7 on the whole 16x16 map of the L = 32 pyramid) with the tile copies on wavefronts of their own.
//
// The four-wavefront kernel runs one workgroup per CU (its LDS tiles take 125 KB), i.e. one wavefront per SIMD, so its
// stage-in, prefetch issue and copy-out - 4.5 k of a lattice's 32.8 k cycles - run while the matrix pipe idles.  Here
// wavefronts 4-7 do those copies beside the MFMA wavefronts 0-3, which keep the convolutions and their epilogues:
//
//     MFMA wavefronts                         copy wavefronts
//     conv6 (lattice i): loop, epilogue       copy-out of lattice i-1 (staging tile -> global); prefetch of lattice i+1
//  B1 (input tiles free, mid tile complete)
//     conv7 (lattice i): loop, epilogue       stage-in of lattice i+1 (registers -> input tiles)
//  B2 (staging tile complete, input tiles staged)
//
// The staging tile therefore has room of its own (the four-wavefront kernel lays it over the input tiles): 79 + 46.7 +
// 36.9 KB of tiles and 1 KB of constants just fit the 160 KB of a CU.  Partition, epilogue arithmetic and results are
// those of fused2_kernel.
// ---------------------------------------------------------------------------------------------
// MF16: both convolutions run on v_mfma_f32_16x16x32_bf16 (ConvMfma16, idm_conv16.h) instead of the 32x32x16 stages: the
// input tiles and the mid tile then have a 32-byte pad per position and share one all-zero row between the map's top and
// bottom padding (tile_a: 9 rows, tile_b and mid: 17).  Results differ from the 32x32x16 kernel by summation order only.
template <int NT1, int WD1, int NT2, int WD2, bool DEFER, bool MF16 = false>
__global__ __launch_bounds__(512, 1) void fused_m3_split_kernel(FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int W = 16, H = 16, R = 16, CA = 128, CB = 64, C1 = 64, C2 = 64;
    constexpr int PADI = MF16 ? 32 : 16;
    constexpr int SA = CA * 2 + PADI, SB = CB * 2 + PADI, SM = C1 * 2 + PADI, SO = C2 * 2 + 16;
    constexpr int ROWS_A = MF16 ? R / 2 + 1 : R / 2 + 2, ROWS_B = MF16 ? R + 1 : R + 4, G0_B = MF16 ? -1 : -2;
    constexpr int ROWS_M = MF16 ? R + 1 : R + 2;
    constexpr int TILE_A = (ROWS_A * (W / 2 + 2) * SA + 15) & ~15, TILE_B = ROWS_B * (W + 2) * SB;
    constexpr int IN = TILE_A + TILE_B, MID = ROWS_M * (W + 2) * SM, STAGE = R * W * SO;
    static_assert(!(MF16 && DEFER), "the 16x16x32 instance finishes conv7's epilogue right after its loop");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool copier = wave >= 4;          // wave-uniform
    const int rw = wave & 3, rtid = tid & 255;
    const int h = lane >> 5;
    const int nlat = (int)a.B;

    char *tile_a = smem, *tile_b = smem + TILE_A, *mid = smem + IN, *stage = smem + IN + MID;
    float *cst = (float *)(smem + IN + MID + STAGE);
    const float *c_b1 = cst, *c_b2 = cst + C1;
    for (int i = tid; i < C1; i += 512) cst[i] = a.b1[i];
    for (int i = tid; i < C2; i += 512) cst[C1 + i] = a.b2[i];
    // the mid tile's halo columns and its padding rows (above and below the map; MF16: one shared row) are never written afterwards
    clear_halo<C1, 512, PADI>(mid, W, ROWS_M, 0, 1, tid);
    if constexpr (!MF16) clear_halo<C1, 512, PADI>(mid, W, 0, R + 1, R + 2, tid);
    const int first = (int)blockIdx.x, step = (int)gridDim.x;
    const LaneGeo geo = make_geo(W, lane & 31);

    if constexpr (MF16) if (!copier) {
        ConvMfma16<CA, CB, C1, WD1, SM, true> c6;
        ConvMfma16<0, C1, C2, WD2, SO, false> c7;
        const int sl = rw >> 1, j0 = (rw & 1) * 4;     // the partition of the 32x32x16 kernel
        c6.setup(tile_a, tile_b, c_b1, a.w1_alt, mid, lane, sl, j0);
        c7.setup(nullptr, mid, c_b2, a.w2_alt, stage, lane, sl, j0);
        if (first < nlat) c6.prologue();
        __syncthreads();                       // constants, halo, the first lattice's tiles
        for (int b = first; b < nlat; b += step) {
            c6.run();
            c7.prologue();
            __syncthreads();                   // B1
            c7.run();
            if (b + step < nlat) c6.prologue();
            __syncthreads();                   // B2
        }
        return;
    }
    if (!copier) {
        using Conv1 = ConvStage<CA, CB, true, C1, NT1, 1, WD1, false>;
        using Conv2 = ConvStage<C1, 0, false, C2, NT2, 1, WD2, false>;
        Conv1 cs1;
        Conv2 cs2;
        auto epi1 = [&](const f32x16 &acc, const f32x4 *kb, const f32x4 *, int mt, int r, int x, bool valid) {
            if (!valid) return;
            char *dst = mid + __mul24(__mul24(r, W + 2) + x + 1, SM) + (mt * 32 + 4 * h) * 2;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 z = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                z = silu4(BIAS_IN_ACC ? z : z + kb[g]);
                *(bf16x4 *)(dst + 16 * g) = pack4(z[0], z[1], z[2], z[3]);
            }
        };
        auto epi2 = [&](const f32x16 &acc, const f32x4 *kb, const f32x4 *, int mt, int r, int x, bool valid) {
            if (!valid) return;
            char *dst = stage + __mul24(__mul24(r, W) + x, SO) + (mt * 32 + 4 * h) * 2;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 z = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                z = silu4(BIAS_IN_ACC ? z : z + kb[g]);
                *(bf16x4 *)(dst + 16 * g) = pack4(z[0], z[1], z[2], z[3]);
            }
        };
        // Every lattice is the same work item: 2 cout slices x 8 two-row tiles per stage = 16 units, four vertically adjacent
        // tiles of one slice per wavefront (slice rw / 2, tiles 4 (rw % 2) ...), as ConvStage::run() would deal them.  The
        // stages are set up once; per lattice only the weight requests (before the barrier that precedes the stage) and
        // the piece itself remain, so the operand addressing is loop-invariant.
        static_assert(NT1 == 4 && NT2 == 4, "one four-tile piece per wavefront and stage");
        const int sl = rw >> 1, j0 = (rw & 1) * 4;
        cs1.begin(tile_a, tile_b, c_b1, nullptr, geo, 1, R, (const bf16x8 *)a.w1, rw, lane, 0, 4, true);
        cs2.begin(mid, nullptr, c_b2, nullptr, geo, 0, R, (const bf16x8 *)a.w2, rw, lane, 0, 4, true);
        cs1.set_weights((const bf16x8 *)a.w1, lane);
        cs2.set_weights((const bf16x8 *)a.w2, lane);
        cs1.set_slice(sl);
        cs2.set_slice(sl);
        auto nohook = [](int) {};
        if constexpr (DEFER) {
            // conv7's activation and store are deferred: its accumulators stay in registers and are finished, two values
            // per step, inside the supersteps of the next lattice's conv6 loop, on the plain float32 ALU (the staging tile
            // is then complete one slot later, and the copy wavefronts write it out during the conv7 slot)
            static_assert(BIAS_IN_ACC, "the deferred epilogue assumes the bias in the accumulators");
            f32x16 pd[4];
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int i = 0; i < 16; ++i) pd[n][i] = 0.0f;
            char *pdst = stage;
            float h0 = 0.f, h1 = 0.f;
            // chunk c of 32: values 2 (c & 1), 2 (c & 1) + 1 of quad (c >> 1) & 3 of pending tile c >> 3
            auto ride = [&](char *dst, auto C) {
                constexpr int c = decltype(C)::value, n = c >> 3, g = (c >> 1) & 3, hf = c & 1;
                const float s0 = silu1(pd[n][4 * g + 2 * hf]), s1 = silu1(pd[n][4 * g + 2 * hf + 1]);
                if constexpr (hf == 0) { h0 = s0; h1 = s1; }
                else *(bf16x4 *)(dst + n * (W * SO) + 16 * g) = pack4(h0, h1, s0, s1);
            };
            auto sink = [&](auto T, const f32x16 &acc, const f32x4 *, const f32x4 *, int mt, int r, int x, bool) {
                constexpr int n = decltype(T)::value;
                pd[n] = acc;
                if constexpr (n == 0) pdst = stage + __mul24(__mul24(r, W) + x, SO) + (mt * 32 + 4 * h) * 2;
            };
            bool pending = false;
            if (first < nlat) cs1.prologue();
            __syncthreads();                   // constants, halo, the first lattice's tiles
            for (int b = first; b < nlat; b += step) {
                char *dst = pdst;
                // (the first lattice's ride finishes the zeros above: nothing reads the staging tile before the second's)
                cs1.template piece<4>(j0, sl, 0, epi1, nohook, [&](auto ST) {
                    constexpr int st = decltype(ST)::value, NS = Conv1::NS;
                    static_for<32 * st / NS, 32 * (st + 1) / NS>([&](auto C) { ride(dst, C); });
                });
                cs2.prologue();
                __syncthreads();               // B1
                cs2.template piece<4>(j0, sl, 0, sink, nohook);
                pending = true;
                if (b + step < nlat) cs1.prologue();
                __syncthreads();               // B2
            }
            if (pending) {
                char *dst = pdst;
                static_for<0, 32>([&](auto C) { ride(dst, C); });
            }
            __syncthreads();                   // the last lattice's staging tile
            return;
        }
        if (first < nlat) cs1.prologue();
        __syncthreads();                       // constants, halo, the first lattice's tiles
        for (int b = first; b < nlat; b += step) {
            cs1.template piece<4>(j0, sl, 0, epi1, nohook);
            cs2.prologue();
            __syncthreads();                   // B1
            cs2.template piece<4>(j0, sl, 0, epi2, nohook);
            if (b + step < nlat) cs1.prologue();
            __syncthreads();                   // B2
        }
        return;
    }

    TileRegs<CA, 5, 256, PADI> ra;
    TileRegs<CB, MF16 ? 9 : 10, 256, PADI> rb;
    auto pf_load = [&](int b) {
        if (b >= nlat) return;
        int t = rtid;
        asm volatile("" : "+v"(t));
        __builtin_assume(t >= 0 && t < 256);
        ra.load((const __bf16 *)a.in_a + (size_t)b * (H / 2) * (W / 2) * CA, H / 2, W / 2, -1, ROWS_A, t);
        rb.load((const __bf16 *)a.in_b + (size_t)b * H * W * CB, H, W, G0_B, ROWS_B, t);
    };
    auto stage_in = [&](int b) {
        if (b >= nlat) return;
        int t = rtid;
        asm volatile("" : "+v"(t));
        __builtin_assume(t >= 0 && t < 256);
        ra.template store<true>(tile_a, W >> 1, ROWS_A, t);
        rb.template store<true>(tile_b, W, ROWS_B, t);
    };
    auto copy_out = [&](int b) {
        constexpr int CH = C2 / 8;
        __bf16 *og = (__bf16 *)a.out + (size_t)b * H * W * C2;
        for (int i = rtid; i < H * W * CH; i += 256) {
            const int ch = i % CH, pos = i / CH;
            *(bf16x8 *)(og + (size_t)pos * C2 + ch * 8) = *(const bf16x8 *)(stage + pos * SO + ch * 16);
        }
    };
    pf_load(first);
    stage_in(first);
    __syncthreads();                           // constants, halo, the first lattice's tiles
    if constexpr (DEFER) {
        for (int b = first; b < nlat; b += step) {
            pf_load(b + step);
            __syncthreads();                   // B1: the previous lattice's staging tile is complete
            stage_in(b + step);
            if (b != first) copy_out(b - step);
            __syncthreads();                   // B2
        }
        __syncthreads();                       // the last lattice's staging tile
        if (first < nlat) {
            int last = first;
            while (last + step < nlat) last += step;
            copy_out(last);
        }
        return;
    }
    for (int b = first; b < nlat; b += step) {
        if (b != first) copy_out(b - step);
        pf_load(b + step);
        __syncthreads();                       // B1
        stage_in(b + step);
        __syncthreads();                       // B2
    }
    if (first < nlat) {
        int last = first;
        while (last + step < nlat) last += step;
        copy_out(last);
    }
}

hipError_t launch_m3_split(const FusedArgs &a, int64_t B, hipStream_t s)
{
    // ISINGDIFF_M3_DEFER=0: conv7's epilogue right after its MFMA loop (comparison runs)
    static const bool defer = [] { const char *e = getenv("ISINGDIFF_M3_DEFER"); return !(e && e[0] == '0'); }();
    // conv6 and conv7 on v_mfma_f32_16x16x32_bf16 (the default: the same time for this kernel at a 0.2 GHz higher clock,
    // i.e. less energy, which the other kernels of the step get back as clock: +1.2 ... 1.9 % for the whole job);
    // ISINGDIFF_M3_MFMA16=0 selects the 32x32x16 stages (same values up to summation order; read once per process)
    static const bool mf16 = [] { const char *e = getenv("ISINGDIFF_M3_MFMA16"); return !(e && e[0] == '0'); }();
    const bool use16 = mf16 && a.w1_alt != nullptr && a.w2_alt != nullptr;
    auto kernel = use16 ? fused_m3_split_kernel<4, IDM_M3_MFMA16_WD, 4, IDM_M3_MFMA16_WD, false, true>
                  : defer ? fused_m3_split_kernel<4, 2, 4, 3, true> : fused_m3_split_kernel<4, 2, 4, 3, false>;
    constexpr int tile_a = ((16 / 2 + 2) * (16 / 2 + 2) * 272 + 15) & ~15, tile_b = 20 * 18 * 144, mid = 18 * 18 * 144,
                  stage = 16 * 16 * 144;
    constexpr int lds32 = tile_a + tile_b + mid + stage + (64 + 64) * 4;
    constexpr int lds16 = 9 * 10 * 288 + 17 * 18 * 160 + 17 * 18 * 160 + stage + (64 + 64) * 4;
    static_assert(lds32 <= 160 * 1024 && lds16 <= 160 * 1024, "the tiles must fit a CU");
    const int lds = use16 ? lds16 : lds32;
    static thread_local struct { bool configured = false; int cus = 0; } all_st[IDM_MAX_DEVICES][2];
    const int slot = idm_device_slot();
    if (slot < 0) return hipErrorInvalidDevice;
    auto &st = all_st[slot][use16 ? 1 : 0];
    if (!st.configured) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t prop;
        if ((e = hipGetDevice(&dev)) != hipSuccess || (e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
        st.cus = prop.multiProcessorCount;
        st.configured = true;
    }
    if (B >= ((int64_t)1 << 30)) return hipErrorInvalidValue;
    const int64_t grid = std::min<int64_t>(B, (int64_t)st.cus);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(512), lds, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// fused_M1 (pooled skip1 -> conv2 -> conv3 -> skip2 + 2x2 average pool, whole 16x16 map of the L = 32 pyramid) on
// v_mfma_f32_16x16x32_bf16: four wavefronts per workgroup, two workgroups per CU (one's epilogues and copies run beside
// the other's MFMA loops), every wavefront one ConvMfma16 run per convolution (32 couts x 8 rows).  LDS per workgroup:
// input tile 17 x 18 x 96 B, mid tile 17 x 18 x 160 B, constants - 77 KB; conv3's output is staged OVER the mid tile
// (plain [16][16] positions of 144 B), so
//     [registers -> input tile]
//  B1 input tile complete, and the previous lattice's copy-out has read the staging tile
//     clear the mid tile's zero row and halo columns (the staged output lay over them)
//     conv2: loop, epilogue -> mid tile
//  B2 mid tile complete
//     conv3: loop;  B2' every wavefront has issued its last MFMA on the mid tile;  request the next lattice's input;
//     epilogue -> staging tile
//  B3 staging tile complete
//     copy-out: skip2 and its pooled copy
// Same partition and epilogue arithmetic as the 32x32x16 whole-map instance of fused2_kernel; results differ from it by
// summation order only.
// ---------------------------------------------------------------------------------------------
// The same kernel serves conv4 (+ time add) + conv5 where THEIR maps are 16 x 16 (L = 64): CB = 64, C1 = C2 = 128, no pooled
// copy, eight wavefronts (four cout slices x two row sets) and one workgroup per CU (137 KB of tiles).
template <int CB, int C1, int C2, bool TADD, bool POOL, int WD1, int WD2>
__global__ __launch_bounds__(C1 * 4, C1 == 64 ? 2 : 1) void fused_m1_mf16_kernel(FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int W = 16, H = 16, NT = C1 * 4;      // threads: two wavefronts (row sets) per 32-cout slice
    static_assert(C1 == C2 && (C1 == 64 || C1 == 128), "one ConvMfma16 unit per wavefront and convolution");
    constexpr int SB = CB * 2 + 32, SM = C1 * 2 + 32, SO = C2 * 2 + 16;
    constexpr int ROWS = H + 1;                       // tile rows: the shared zero row, then the map's rows
    constexpr int TILE_B = ROWS * (W + 2) * SB, MID = ROWS * (W + 2) * SM;
    static_assert(H * W * SO <= MID, "the staging tile lies over the mid tile");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nlat = (int)a.B;
    char *tile_b = smem, *mid = smem + TILE_B, *stage = mid;
    float *cst = (float *)(smem + TILE_B + MID);
    for (int i = tid; i < C1; i += NT) {
        cst[i] = a.b1[i];
        if constexpr (TADD) cst[C1 + C2 + i] = a.tadd1[i];   // (a third array of constants only in this instance's LDS size)
    }
    for (int i = tid; i < C2; i += NT) cst[C1 + i] = a.b2[i];

    ConvMfma16<0, CB, C1, WD1, SM, true, false, TADD> c2;
    ConvMfma16<0, C1, C2, WD2, SO, false> c3;
    const int sl = wave >> 1, j0 = (wave & 1) * 4;
    c2.setup(nullptr, tile_b, cst, a.w1_alt, mid, lane, sl, j0);
    c2.tadd = cst + C1 + C2 + 32 * sl + 4 * (lane >> 4);
    c3.setup(nullptr, mid, cst + C1, a.w2_alt, stage, lane, sl, j0);

    TileRegs<CB, (ROWS * W * (CB / 8) + NT - 1) / NT, NT, 32> rb;
    auto pf_load = [&](int b) {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        __builtin_assume(t >= 0 && t < NT);
        rb.load((const __bf16 *)a.in_a + (size_t)b * H * W * CB, H, W, -1, ROWS, t);
    };
    const int first = (int)blockIdx.x, step = (int)gridDim.x;
    if (first < nlat) {
        pf_load(first);
        c2.prologue();
    }
    for (int b = first; b < nlat; b += step) {
        {
            int t = threadIdx.x;
            asm volatile("" : "+v"(t));
            __builtin_assume(t >= 0 && t < NT);
            rb.template store<true>(tile_b, W, ROWS, t);
            __syncthreads();                   // B1
            // (after the barrier: before it another wavefront may still be copying the previous lattice's staged output,
            // which lies over these rows and columns, out; conv2's epilogue writes the interior only, conv3 reads after B2)
            clear_halo<C1, NT, 32>(mid, W, ROWS, 0, 1, t);
        }
        c2.run();
        c3.prologue();
        __syncthreads();                       // B2
        // B2' between conv3's loop and its epilogue; the next lattice's input is requested there, so that its registers
        // are not live inside the MFMA loops (the last lattice requests its own input again, discarded: an unconditional
        // request keeps the registers from being loop-carried)
        c3.run([&] {
            __syncthreads();
            pf_load(b + step < nlat ? b + step : b);
        });
        if (b + step < nlat) c2.prologue();
        __syncthreads();                       // B3
        // (the thread index is made opaque per lattice: otherwise the copy-out's per-thread offsets are hoisted out of the
        // lattice loop and held - spilled - across the convolutions)
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        __builtin_assume(t >= 0 && t < NT);
        constexpr int CH = C2 / 8;
        __bf16 *og = (__bf16 *)a.out + (size_t)b * H * W * C2;
        for (int i = t; i < H * W * CH; i += NT) {
            const int ch = i % CH, pos = i / CH;
            *(bf16x8 *)(og + (size_t)pos * C2 + ch * 8) = *(const bf16x8 *)(stage + pos * SO + ch * 16);
        }
        if constexpr (POOL) {
            constexpr int Wh = W / 2, Hh = H / 2;
            __bf16 *pg = (__bf16 *)a.out_pool + (size_t)b * Hh * Wh * C2;
            for (int i = t; i < Hh * Wh * CH; i += NT) {
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

#ifndef IDM_M1_MFMA16_WD
#define IDM_M1_MFMA16_WD 2, 2   // supersteps of weights in flight ahead of their MFMAs: conv2 (three supersteps), conv3 (six)
#endif
// wide: the conv4 + conv5 instance (16 x 16 maps of 64 -> 128 -> 128 channels, L = 64)
hipError_t launch_m1_mf16(const FusedArgs &a, int64_t B, hipStream_t s, bool wide)
{
    auto kernel = wide ? fused_m1_mf16_kernel<64, 128, 128, true, false, IDM_M1_MFMA16_WD> : fused_m1_mf16_kernel<32, 64, 64, false, true, IDM_M1_MFMA16_WD>;
    constexpr int lds_m1 = 17 * 18 * 96 + 17 * 18 * 160 + (64 + 64) * 4;
    constexpr int lds_wide = 17 * 18 * 160 + 17 * 18 * 288 + 3 * 128 * 4;
    static_assert(2 * lds_m1 <= 160 * 1024 && lds_wide <= 160 * 1024, "two workgroups / one workgroup per CU");
    static thread_local struct { bool configured = false; int cus = 0; } all_st[IDM_MAX_DEVICES][2];   // one slot per kernel function
    const int slot = idm_device_slot();
    if (slot < 0) return hipErrorInvalidDevice;
    auto &st = all_st[slot][wide ? 1 : 0];
    if (!st.configured) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t prop;
        if ((e = hipGetDevice(&dev)) != hipSuccess || (e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
        st.cus = prop.multiProcessorCount;
        st.configured = true;
    }
    if (a.H != 16 || a.W != 16 || !a.w1_alt || !a.w2_alt || (wide && !a.tadd1)) return hipErrorInvalidValue;
    if (B >= ((int64_t)1 << 30)) return hipErrorInvalidValue;
    const int64_t grid = std::min<int64_t>(B, (int64_t)st.cus * (wide ? 1 : 2));
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(wide ? 512 : 256), wide ? lds_wide : lds_m1, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// fused_M2 (pooled skip2 -> conv4 (+ time embedding) -> conv5, whole 8x8 maps of the L = 32 pyramid) on
// v_mfma_f32_16x16x32_bf16, TWO LATTICES per pass: a wavefront owns one 32-cout slice of both maps (ConvMfma16, PAIR
// geometry), so a weight fragment feeds eight MFMAs where the one-lattice kernel (two 32-pixel tiles per slice) gets two
// out of it - the weight stream, which bounds that kernel, halves.  Four wavefronts per workgroup, two workgroups per CU.
// LDS per workgroup: 2 x 82 positions x 160 B input tiles, 2 x 82 x 288 B mid tiles, constants - 73 KB; conv5's output
// is staged over each lattice's mid tile (plain [8][8] positions of 272 B).  Barriers as in fused_m1_mf16_kernel.
// An odd last lattice is paired with itself and stored once.
// ---------------------------------------------------------------------------------------------
// HROWS: rows of the maps (8; 5 for the third level of the transposed spectral model, whose maps are 5 x 8)
template <int WD1, int WD2, int HROWS = 8>
__global__ __launch_bounds__(256, 2) void fused_m2_pair_kernel(FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int W = 8, H = HROWS, C0 = 64, C1 = 128, C2 = 128;
    constexpr int S0 = C0 * 2 + 32, SM = C1 * 2 + 32, SO = C2 * 2 + 16;
    using Conv4 = ConvMfma16<0, C0, C1, WD1, SM, true, true, true, HROWS>;
    using Conv5 = ConvMfma16<0, C1, C2, WD2, SO, false, true, false, HROWS>;
    constexpr int NPOS = Conv4::PAIR_POS, PITCH = Conv4::PITCH;
    constexpr int IN_LAT = NPOS * S0, MID_LAT = NPOS * SM;
    static_assert(8 * W * SO <= MID_LAT, "a lattice's staging tile ([8][8] positions) lies over its mid tile");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nlat = (int)a.B, npairs = (nlat + 1) >> 1;
    char *tile_in = smem, *mid = smem + 2 * IN_LAT;
    float *cst = (float *)(smem + 2 * IN_LAT + 2 * MID_LAT);
    for (int i = tid; i < C1; i += 256) {
        cst[i] = a.b1[i];
        cst[C1 + i] = a.tadd1[i];
    }
    for (int i = tid; i < C2; i += 256) cst[2 * C1 + i] = a.b2[i];
    // input tiles: only the interior is ever written; halo positions and the zero row stay zero
    {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid * 16; i < 2 * IN_LAT; i += 256 * 16) *(f32x4 *)(tile_in + i) = z;
    }

    Conv4 c4;
    Conv5 c5;
    c4.setup_pair(tile_in, IN_LAT, cst, cst + C1, a.w1_alt, mid, MID_LAT, lane, wave);
    c5.setup_pair(mid, MID_LAT, cst + 2 * C1, nullptr, a.w2_alt, mid, MID_LAT, lane, wave);

    // a pair's input: 2 lattices x H * 8 positions x 8 items of 8 channels
    constexpr int IT = H * W * (C0 / 8), NPI = (2 * IT + 255) / 256;   // items per lattice, per thread
    bf16x8 rin[NPI];
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in_a, (unsigned)std::min<int64_t>((int64_t)nlat * IT * 16, 0xFFFFFFFFll));
    auto pf_load = [&](int p) {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        __builtin_assume(t >= 0 && t < 256);
#pragma unroll
        for (int k = 0; k < NPI; ++k) {
            const int i = t + k * 256, g = i / IT, r = i - g * IT;
            const int lat = min(2 * p + g, nlat - 1);
            rin[k] = buf_load16(rs_in, i < 2 * IT ? (unsigned)((lat * IT + r) * 16) : 0x80000000u, 0);
        }
    };
    const int first = (int)blockIdx.x, step = (int)gridDim.x;
    if (first < npairs) {
        pf_load(first);
        c4.prologue();
    }
    __syncthreads();                           // constants, zeroed input tiles
    for (int p = first; p < npairs; p += step) {
        {
            int t = threadIdx.x;
            asm volatile("" : "+v"(t));
            __builtin_assume(t >= 0 && t < 256);
#pragma unroll
            for (int k = 0; k < NPI; ++k) {
                const int i = t + k * 256, g = i / IT, r = i - g * IT, pos = r >> 3, ch = r & 7;
                if (i < 2 * IT)
                    *(bf16x8 *)(tile_in + g * IN_LAT + (((pos >> 3) + 1) * PITCH + (pos & 7) + 1) * S0 + ch * 16) = rin[k];
            }
            __syncthreads();                   // B1: input tiles complete; the previous pair's copy-out has read the staging tiles
            // the mid tiles' zero row (positions 0 .. 9), the halo positions 9 r of rows 2 .. H, and everything from row H + 1
            // on (its halo position, and the rows below a map of fewer than eight rows): the staged output lay over them
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            constexpr int U = SM / 16;         // 16-byte units per position
            constexpr int NCL = 10 + (H - 1) + (NPOS - (H + 1) * PITCH);
            for (int i = t; i < 2 * NCL * U; i += 256) {
                const int g = i / (NCL * U), r = i - g * (NCL * U), q = r / U, u = r - q * U;
                const int pos = q < 10 ? q : q < 10 + (H - 1) ? (q - 8) * PITCH : (H + 1) * PITCH + (q - 10 - (H - 1));
                *(f32x4 *)(mid + g * MID_LAT + pos * SM + u * 16) = z;
            }
        }
        c4.run();
        c5.prologue();
        __syncthreads();                       // B2: mid tiles complete
        c5.run([&] {
            __syncthreads();                   // B2': every wavefront has issued its last MFMA on the mid tiles
            pf_load(p + step < npairs ? p + step : p);
        });
        if (p + step < npairs) c4.prologue();
        __syncthreads();                       // B3: staging tiles complete
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        __builtin_assume(t >= 0 && t < 256);
        constexpr int CH = C2 / 8;             // 16 items per position
        constexpr int OT = H * W * CH, NPO = (2 * OT + 255) / 256;
#pragma unroll
        for (int k = 0; k < NPO; ++k) {
            const int i = t + k * 256, g = i / OT, r = i - g * OT;
            const int lat = 2 * p + g;
            if (i < 2 * OT && lat < nlat)
                *(bf16x8 *)((__bf16 *)a.out + (size_t)lat * H * W * C2 + (size_t)r * 8) =
                    *(const bf16x8 *)(mid + g * MID_LAT + (r / CH) * SO + (r % CH) * 16);
        }
    }
}

#ifndef IDM_M2_MFMA16_WD
#define IDM_M2_MFMA16_WD 2, 2   // supersteps of weights in flight ahead of their MFMAs: conv4 (six supersteps), conv5 (twelve)
#endif
hipError_t launch_m2_pair(const FusedArgs &a, int64_t B, hipStream_t s)
{
    auto kernel = a.H == 5 ? fused_m2_pair_kernel<IDM_M2_MFMA16_WD, 5> : fused_m2_pair_kernel<IDM_M2_MFMA16_WD, 8>;
    constexpr int lds = 2 * 82 * 160 + 2 * 82 * 288 + 3 * 128 * 4;
    static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
    static thread_local struct { bool configured = false; int cus = 0; } all_st[IDM_MAX_DEVICES][2];   // one slot per kernel function
    const int slot = idm_device_slot();
    if (slot < 0) return hipErrorInvalidDevice;
    if ((a.H != 8 && a.H != 5) || a.W != 8 || !a.w1_alt || !a.w2_alt || !a.tadd1) return hipErrorInvalidValue;
    auto &st = all_st[slot][a.H == 5];
    if (!st.configured) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t prop;
        if ((e = hipGetDevice(&dev)) != hipSuccess || (e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
        st.cus = prop.multiProcessorCount;
        st.configured = true;
    }
    if (B >= ((int64_t)1 << 19)) return hipErrorInvalidValue;   // the input (8 KB per lattice) is addressed through one 4 GiB buffer resource
    const int64_t grid = std::min<int64_t>((B + 1) / 2, (int64_t)st.cus * 2);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// fused_B (conv8, conv9, 1x1 head, reverse update) for the L = 32 lattice: one workgroup per CU, two lattices
// ("streams" X and Y, each with its own input tiles, mid tile and score tile: 2 x 76 KB) in flight.
//
//     MFMA wavefronts 0-3                      copy wavefronts 4-7
//     conv8 (X): loop, epilogue -> mid X       state update of Y's previous band (score tile Y, noise draw), carry rows Y
//  B1
//     conv8 (Y)                                stage-in of X's next band; prefetch of Y's next band
//  B2
//     conv9 (X): loop, epilogue, head -> score stage-in of Y's next band; prefetch of X's band after next
//  B3
//     conv9 (Y)                                state update of X's band, carry rows X
//  B4
//
// Each stage's first weights are requested as soon as its previous run ends (arm()); the rows of a mid tile that lie
// outside the map are cleared by the copy wavefronts while conv8 of that stream runs (its epilogue does not write them).
// Partition, epilogue arithmetic, Philox counters and update are those of fused2_kernel: same results bit for bit.
// ---------------------------------------------------------------------------------------------
// TOP9 (with DEFER): the ninth conv8 row of a lattice's top band - which otherwise makes one MFMA wavefront's slot half
// as long again as the others' - is computed by the four copy wavefronts, one 16-pixel x 16-cout quadrant each, on
// v_mfma_f32_16x16x32_bf16 (27 MFMAs of 16 cycles per SIMD beside the MFMA wavefronts' two rows each).  That row's sums
// run in another order than the 32x32x16 tiles' (needs the 16x16x32 weight packing, w1_alt).
template <int NT1, int WD1, int NT2, int WD2, bool DEFER, bool TOP9 = false>
__global__ __launch_bounds__(512, 1) void fused_b_copy_kernel(FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int W = 32, H = 32, R = 8, CA = 64, CB = 32, C1 = 32, C2 = 32;
    constexpr int SA = CA * 2 + 16, SB = CB * 2 + 16, SM = C1 * 2 + 16;
    constexpr int TILE_A = ((R / 2 + 2) * (W / 2 + 2) * SA + 15) & ~15, TILE_B = (R + 4) * (W + 2) * SB;
    constexpr int IN = TILE_A + TILE_B, MID = (R + 2) * (W + 2) * SM, SCORE = R * W * 4, STREAM = IN + MID + SCORE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool copier = wave >= 4;          // wave-uniform
    const int rw = wave & 3, rtid = tid & 255;
    const int h = lane >> 5, li = lane & 31;
    const int nlat = (int)a.B;

    float *cst = (float *)(smem + 2 * STREAM);
    const float *c_b1 = cst, *c_b2 = cst + C1, *c_wh = cst + C1 + C2, *c_bh = cst + C1 + 2 * C2;
    for (int i = tid; i < C1; i += 512) cst[i] = a.b1[i];
    for (int i = tid; i < C2; i += 512) {
        cst[C1 + i] = a.b2[i];
        cst[C1 + C2 + i] = a.w_head[i];
    }
    if (tid == 0) cst[C1 + 2 * C2] = a.b_head[0];
    auto tile_a = [&](int s) { return smem + s * STREAM; };
    auto tile_b = [&](int s) { return smem + s * STREAM + TILE_A; };
    auto mid = [&](int s) { return smem + s * STREAM + IN; };
    auto score = [&](int s) { return (float *)(smem + s * STREAM + IN + MID); };
    for (int s = 0; s < 2; ++s) {
        clear_halo<C1, 512>(mid(s), W, R + 2, 0, 0, tid);                  // halo columns: never written afterwards
        clear_halo<CA, 512>(tile_a(s), W >> 1, (R >> 1) + 2, 0, 0, tid);
        clear_halo<CB, 512>(tile_b(s), W, R + 4, 0, 0, tid);
    }

    // Item n of stream s: band n % 4 of the workgroup's (n / 4)-th lattice of that stream.
    struct Item { int b, y0; bool act; };
    auto item_of = [&](int s, int n) {
        Item it;
        const int k = n >> 2;
        it.b = (int)blockIdx.x + (2 * k + s) * (int)gridDim.x;
        it.y0 = (n & 3) * R;
        it.act = n >= 0 && it.b < nlat;
        return it;
    };
    auto band_geo = [&](Item it, int &rlo, int &rhi) {
        rlo = it.y0 == 0 ? 1 : 2;
        rhi = min(R + 2, H - it.y0 + 1);
    };
    const int per = 2 * (int)gridDim.x;
    const int nx = (int)blockIdx.x < nlat ? (nlat - 1 - (int)blockIdx.x) / per + 1 : 0;
    const int niter = nx * 4;   // stream Y ends no later than stream X
    const LaneGeo geo = make_geo(W, li);

    if constexpr (DEFER) if (!copier) {
        // Deferred epilogues: a piece's accumulators stay in registers and are finished, two values per step, inside the
        // supersteps of the NEXT slot's MFMA loop (the other stream's conv8, or a conv9), on the plain float32 ALU:
        //     slot 1  conv8 (X)  +  head of Y's previous band -> score Y
        //     slot 2  conv8 (Y)  +  activation of conv8 (X)   -> mid X
        //     slot 3  conv9 (X)  +  activation of conv8 (Y)   -> mid Y
        //     slot 4  conv9 (Y)  +  head of X                 -> score X
        // (a wavefront's third tile, in a lattice's nine-row top band, is finished at once).  Stores of a stream without
        // work go to a scratch area behind the constants.
        static_assert(BIAS_IN_ACC && NT1 == 3 && NT2 == 2, "deferred epilogues assume the bias in the accumulators");
        // The deferred sinks below store both pending tiles of a piece unconditionally (to dst and dst + one row): they
        // ignore the `valid` flag ConvStage::piece hands them.  That is correct only while every tile of every piece lies
        // inside its band: one row per 32-pixel tile (W == 32, so TR == 1), EXACT stages, and conv8 pieces of exactly two
        // or three rows that never run past the band's rows.  Any other band geometry must carry `valid` with the
        // pending piece and send an invalid tile's stores to `dummy`.
        static_assert(W == 32 && R == 8, "the deferred sinks assume one row per tile and full eight-row bands");
        using Conv1 = ConvStage<CA, CB, true, C1, NT1, 1, WD1, false, true>;
        using Conv2 = ConvStage<C1, 0, false, C2, NT2, 1, WD2, true, true>;
        Conv1 cs1;
        Conv2 cs2, cs2b;
        cs1.set_weights((const bf16x8 *)a.w1, lane);
        cs2.set_weights((const bf16x8 *)a.w2, lane);
        cs2b.set_weights((const bf16x8 *)a.w2, lane);
        cs2.begin(mid(0), nullptr, c_b2, c_wh, geo, 0, R, (const bf16x8 *)a.w2, rw, lane, 0, 4, true);
        cs2b.begin(mid(1), nullptr, c_b2, c_wh, geo, 0, R, (const bf16x8 *)a.w2, rw, lane, 0, 4, true);
        cs2.set_slice(0);
        cs2b.set_slice(0);
        constexpr int ROWB = (W + 2) * SM;
        char *const dummy = smem + 2 * STREAM + 512;            // 3 KB behind the constants
        f32x4 kw[4];                                            // the head's weights of this lane's channels
        f32x16 pa0, pa1, pb0, pb1;                              // the two pending pieces: one drains while the other fills
#pragma unroll
        for (int i = 0; i < 16; ++i) pa0[i] = pa1[i] = pb0[i] = pb1[i] = 0.0f;
        char *d8 = dummy;                                       // where the pending conv8 tiles go
        float *d9 = (float *)dummy;                             // ... and the pending head values
        float h0 = 0.f, h1 = 0.f;
        f32x4 p4a = {0.f, 0.f, 0.f, 0.f};
        auto nohook = [](int) {};
        // chunk c of 16: values 2 (c & 1), 2 (c & 1) + 1 of quad (c >> 1) & 3 of pending tile c >> 3
        auto ride8 = [&](const f32x16 &t0, const f32x16 &t1, char *dst, auto C) {
            constexpr int c = decltype(C)::value, n = c >> 3, g = (c >> 1) & 3, hf = c & 1;
            const f32x16 &t = n ? t1 : t0;
#ifdef IDM_EXP_RIDE8_RAW   // timing experiment (wrong results): conv8 stores its pre-activations
            const float s0 = t[4 * g + 2 * hf], s1 = t[4 * g + 2 * hf + 1];
#else
            const float s0 = silu1(t[4 * g + 2 * hf]), s1 = silu1(t[4 * g + 2 * hf + 1]);
#endif
            if constexpr (hf == 0) { h0 = s0; h1 = s1; }
            else *(bf16x4 *)(dst + n * ROWB + 16 * g) = pack4(h0, h1, s0, s1);
        };
        auto ride9 = [&](const f32x16 &t0, const f32x16 &t1, float *dst, auto C) {
            constexpr int c = decltype(C)::value, n = c >> 3, g = (c >> 1) & 3, hf = c & 1;
            const f32x16 &t = n ? t1 : t0;
            const float s0 = silu1(t[4 * g + 2 * hf]), s1 = silu1(t[4 * g + 2 * hf + 1]);
            if constexpr (g == 0) p4a[2 * hf] = p4a[2 * hf + 1] = 0.0f;
            p4a[2 * hf] += s0 * kw[g][2 * hf];           // (the same expression, hence the same contraction, as the
            p4a[2 * hf + 1] += s1 * kw[g][2 * hf + 1];   // four-at-a-time form of the undeferred epilogue)
            if constexpr ((c & 7) == 7) {
                float part = (p4a[0] + p4a[1]) + (p4a[2] + p4a[3]);
                part += __shfl_xor(part, 32, 64);
                if (h == 0) dst[n * W] = part + c_bh[0];
            }
        };
        // the chunks of superstep st of NS: [16 st / NS, 16 (st + 1) / NS)
        auto chunks_of = [&](auto ST, auto NSC, auto &&f) {
            constexpr int st = decltype(ST)::value, NS = decltype(NSC)::value;
            constexpr int lo = 16 * st / NS, hi = 16 * (st + 1) / NS;
            static_for<lo, hi>(f);
        };
        auto flush8 = [&](const f32x16 &t0, const f32x16 &t1, char *dst) {
            static_for<0, 16>([&](auto C) { ride8(t0, t1, dst, C); });
        };
        auto flush9 = [&](const f32x16 &t0, const f32x16 &t1, float *dst) {
            static_for<0, 16>([&](auto C) { ride9(t0, t1, dst, C); });
        };
        // conv8 of stream s; `ride` finishes the pending piece; the new accumulators go to (q0, q1), their target to d8
#ifdef IDM_TIMELINE
        int tls_n = 0;
#endif
        auto conv8 = [&](int s, Item it, f32x16 &q0, f32x16 &q1, auto &&ride, auto &&flush) {
            if (it.act) {
                int rlo, rhi;
                band_geo(it, rlo, rhi);
                char *m = mid(s);
                // (TOP9: the top band's ninth row is a copy wavefront's)
                cs1.begin(tile_a(s), tile_b(s), c_b1, nullptr, geo, rlo, TOP9 ? min(rhi - rlo, R) : rhi - rlo, (const bf16x8 *)a.w1, rw, lane, 0, 4, true);
                char *nd = nullptr;
                auto sink = [&](auto T, const f32x16 &acc, const f32x4 *, const f32x4 *, int, int r, int x, bool) {
                    constexpr int n = decltype(T)::value;
                    char *dst = m + __mul24(__mul24(r, W + 2) + x + 1, SM) + 8 * h;
                    if constexpr (n == 0) { q0 = acc; nd = dst; }
                    else if constexpr (n == 1) q1 = acc;
                    else {
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            *(bf16x4 *)(dst + 16 * g) = pack4(silu1(acc[4 * g]), silu1(acc[4 * g + 1]), silu1(acc[4 * g + 2]), silu1(acc[4 * g + 3]));
                    }
                };
                auto inter = [&](auto ST) { chunks_of(ST, std::integral_constant<int, Conv1::NS>{}, ride); };
                // (the parity of the piece's first window row is known per band: the half-resolution tile's fragments are
                // then read once per half-resolution row)
#ifdef IDM_TIMELINE   // stamps 8-10: operands addressed / last MFMA issued / sinks done (stream X)
                auto tlhook = [&](int i) { if (s == 0 && i < 3) TLS(8 + i); };
#else
                auto &tlhook = nohook;
#endif
                if (cs1.f1 - cs1.f == 3) cs1.template piece<3>(cs1.f, 0, 0, sink, tlhook, inter);
                else if ((rlo + cs1.f) & 1) cs1.template piece<2, 1>(cs1.f, 0, 0, sink, tlhook, inter);
                else cs1.template piece<2, 0>(cs1.f, 0, 0, sink, tlhook, inter);
                d8 = nd;
            } else {
                flush();
                d8 = dummy;
            }
            cs1.arm();
        };
        auto conv9 = [&](auto &cs, int s, Item it, f32x16 &q0, f32x16 &q1, auto &&ride, auto &&flush) {
            if (it.act) {
                float *sc = score(s);
                float *nd = nullptr;
                auto sink = [&](auto T, const f32x16 &acc, const f32x4 *, const f32x4 *, int, int r, int x, bool) {
                    constexpr int n = decltype(T)::value;
                    if constexpr (n == 0) { q0 = acc; nd = sc + r * W + x; }
                    else q1 = acc;
                };
                auto inter = [&](auto ST) { chunks_of(ST, std::integral_constant<int, Conv2::NS>{}, ride); };
#ifdef IDM_TIMELINE   // stamps 11-13, as above for conv9 (stream X)
                auto tlhook = [&](int i) { if (s == 0 && i < 3) TLS(11 + i); };
#else
                auto &tlhook = nohook;
#endif
                cs.template piece<2>(2 * rw, 0, 0, sink, tlhook, inter);
                d9 = nd;
            } else {
                flush();
                d9 = (float *)dummy;
            }
        };
        cs1.arm();
        cs2.prologue();
        __syncthreads();                       // constants, halo columns, the first bands' tiles
#pragma unroll
        for (int q = 0; q < 4; ++q) kw[q] = *(const f32x4 *)(c_wh + 8 * q + 4 * h);
        for (int n = 0; n < niter; ++n) {
#ifdef IDM_TIMELINE
            tls_n = n;
#endif
            const Item X = item_of(0, n), Y = item_of(1, n);
            {   // slot 1: conv8 (X) -> (pa0, pa1); drains the head of Y's previous band (pb0, pb1)
                float *t9 = d9;
                conv8(0, X, pa0, pa1, [&](auto C) { ride9(pb0, pb1, t9, C); }, [&] { flush9(pb0, pb1, t9); });
            }
            BAR(1);
            {   // slot 2: conv8 (Y) -> (pb0, pb1); drains conv8 (X)
                char *t8 = d8;
                conv8(1, Y, pb0, pb1, [&](auto C) { ride8(pa0, pa1, t8, C); }, [&] { flush8(pa0, pa1, t8); });
            }
            BAR(2);
            {   // slot 3: conv9 (X) -> (pa0, pa1); drains conv8 (Y)
                char *t8 = d8;
                conv9(cs2, 0, X, pa0, pa1, [&](auto C) { ride8(pb0, pb1, t8, C); }, [&] { flush8(pb0, pb1, t8); });
                cs2b.prologue();
            }
            BAR(3);
            {   // slot 4: conv9 (Y) -> (pb0, pb1); drains the head of X
                float *t9 = d9;
                conv9(cs2b, 1, Y, pb0, pb1, [&](auto C) { ride9(pa0, pa1, t9, C); }, [&] { flush9(pa0, pa1, t9); });
                cs2.prologue();
            }
            BAR(4);
        }
        flush9(pb0, pb1, d9);                   // the head of Y's last band
        __syncthreads();
        return;
    }
    if (!copier) {
        using Conv1 = ConvStage<CA, CB, true, C1, NT1, 1, WD1, false, true>;
        using Conv2 = ConvStage<C1, 0, false, C2, NT2, 1, WD2, true, true>;
        Conv1 cs1;
        Conv2 cs2;
        cs1.set_weights((const bf16x8 *)a.w1, lane);
        cs2.set_weights((const bf16x8 *)a.w2, lane);
        auto conv8 = [&](int s, Item it) {
            if (it.act) {
                int rlo, rhi;
                band_geo(it, rlo, rhi);
                char *m = mid(s);
                cs1.begin(tile_a(s), tile_b(s), c_b1, nullptr, geo, rlo, rhi - rlo, (const bf16x8 *)a.w1, rw, lane, 0, 4, true);
                cs1.run([&](const f32x16 &acc, const f32x4 *kb, const f32x4 *, int mt, int r, int x, bool valid) {
                    if (!valid) return;
                    char *dst = m + __mul24(__mul24(r, W + 2) + x + 1, SM) + (mt * 32 + 4 * h) * 2;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 z = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                        z = silu4(BIAS_IN_ACC ? z : z + kb[g]);
                        *(bf16x4 *)(dst + 16 * g) = pack4(z[0], z[1], z[2], z[3]);
                    }
                });
            }
            cs1.arm();
        };
        // conv9 is the same work item in every band (rows 2 rw, 2 rw + 1 of a full eight-row band): set up once per stream
        // (the streams differ in their tiles only), so its operand addressing leaves the band loop
        static_assert(NT2 == 2, "one two-tile piece per wavefront");
        Conv2 cs2b;
        cs2b.set_weights((const bf16x8 *)a.w2, lane);
        cs2.begin(mid(0), nullptr, c_b2, c_wh, geo, 0, R, (const bf16x8 *)a.w2, rw, lane, 0, 4, true);
        cs2b.begin(mid(1), nullptr, c_b2, c_wh, geo, 0, R, (const bf16x8 *)a.w2, rw, lane, 0, 4, true);
        cs2.set_slice(0);
        cs2b.set_slice(0);
        auto nohook = [](int) {};
        auto conv9 = [&](auto &cs, int s, Item it) {
            if (it.act) {
                float *sc = score(s);
                cs.template piece<2>(2 * rw, 0, 0, [&](const f32x16 &acc, const f32x4 *kb, const f32x4 *kw, int, int r, int x, bool valid) {
                    f32x4 p4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 z = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                        p4 += silu4(BIAS_IN_ACC ? z : z + kb[g]) * kw[g];
                    }
                    float part = (p4[0] + p4[1]) + (p4[2] + p4[3]);
                    part += __shfl_xor(part, 32, 64);
                    if (valid && h == 0) sc[r * W + x] = part + c_bh[0];
                }, nohook);
            }
        };
        cs1.arm();
        cs2.prologue();
        __syncthreads();                       // constants, halo columns, the first bands' tiles
        for (int n = 0; n < niter; ++n) {
#ifdef IDM_TIMELINE
            const int tls_n = n;
#endif
            const Item X = item_of(0, n), Y = item_of(1, n);
            conv8(0, X);
            BAR(1);
            conv8(1, Y);
            BAR(2);
            conv9(cs2, 0, X);
            cs2b.prologue();                   // (the two objects' weight rings: each is refilled while the other runs)
            BAR(3);
            conv9(cs2b, 1, Y);
            cs2.prologue();
            BAR(4);
        }
        return;
    }

    // ------------------------------ copy wavefronts ------------------------------
    // one set of prefetch registers per stream: a stream's next band is requested two slots before it is staged, and
    // the requests are issued during the conv9 slots (beside conv8 they slow its weight stream)
    TileRegs<CA, 3, 256> ra[2];
    TileRegs<CB, 6, 256> rb[2];
    auto pf_load = [&](auto &qa, auto &qb, Item it) {
        if (!it.act) return;
        int t = rtid;
        asm volatile("" : "+v"(t));
        __builtin_assume(t >= 0 && t < 256);
        qa.load((const __bf16 *)a.in_a + (size_t)it.b * (H / 2) * (W / 2) * CA, H / 2, W / 2, (it.y0 >> 1) - 1, (R >> 1) + 2, t);
        qb.load((const __bf16 *)a.in_b + (size_t)it.b * H * W * CB, H, W, it.y0 - 2, R + 4, t);
    };
    auto stage_in = [&](auto &qa, auto &qb, int s, Item it) {
        if (!it.act) return;
        int t = rtid;
        asm volatile("" : "+v"(t));
        __builtin_assume(t >= 0 && t < 256);
        qa.template store<false>(tile_a(s), W >> 1, (R >> 1) + 2, t);
        qb.template store<false>(tile_b(s), W, R + 4, t);
    };
    // rows of the mid tile outside the map are conv9's zero padding
    auto clear_pad_rows = [&](int s, Item it) {
        if (!it.act) return;
        int rlo, rhi;
        band_geo(it, rlo, rhi);
        if (it.y0 == 0) clear_halo<C1, 256>(mid(s), W, 0, 0, 1, rtid);
        if (rhi < R + 2) clear_halo<C1, 256>(mid(s), W, 0, rhi, R + 2, rtid);
    };
    // The 64 float4 quads of a band (one Philox counter each) are one wavefront's worth of lanes: the copy wavefronts take
    // the bands in turn, so that the noise draw - mostly quarter-rate integer multiplies, which hold the vector ALU that the
    // MFMA wavefront of the SIMD issues through - costs each SIMD one band in four.
    f32x4 xq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, zq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    auto my_turn = [&](int s, int n) { return ((n + 2 * s) & 3) == rw; };
    auto quad_of = [&](Item it, int &qs, int64_t &off) {
        const int qr = lane >> 3, qx = lane & 7;
        qs = qr * W + qx * 4;
        off = (int64_t)it.b * (H * W) + (it.y0 + qr) * W + qx * 4;
    };
    auto x_fetch = [&](int s, Item it, int n) {
        if (!it.act || !a.do_update || !my_turn(s, n)) return;
        int qs;
        int64_t off;
        quad_of(it, qs, off);
        xq[s] = *(const f32x4 *)(a.x + off);
        if (a.add_noise && a.noise) zq[s] = *(const f32x4 *)(a.noise + off);
    };
    auto update = [&](int s, Item it, int n) {
        if (!it.act) return;
        if (my_turn(s, n)) {
            int qs;
            int64_t off;
            quad_of(it, qs, off);
            const f32x4 sc = *(const f32x4 *)(score(s) + qs);
            if (a.score_out) *(f32x4 *)(a.score_out + off) = sc;
            if (a.do_update) {
                float z[4] = {zq[s][0], zq[s][1], zq[s][2], zq[s][3]};
                if (a.add_noise && !a.noise) {
                    uint32_t k0 = (uint32_t)a.ph.seed, k1 = (uint32_t)(a.ph.seed >> 32);
                    normal4(((uint64_t)k1 << 32) | k0, a.ph.first_global_index + it.b, (uint32_t)((off - (int64_t)it.b * (H * W)) >> 2),
                            a.ph.t, z);
                }
                f32x4 xv = xq[s];
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[j] = reverse_update(a.c1, xv[j], a.beta, sc[j], z[j], a.sd, a.add_noise != 0);
                *(f32x4 *)(a.x + off) = xv;
            }
        }
        // rows R, R+1 of the mid tile are rows 0, 1 of this lattice's next band
        if (it.y0 + R < H) {
            char *m = mid(s);
            const int bytes = 2 * (W + 2) * SM;
            const char *src = m + R * (W + 2) * SM;
            for (int i = rtid * 16; i < bytes; i += 256 * 16) *(f32x4 *)(m + i) = *(const f32x4 *)(src + i);
        }
    };

    // TOP9: conv8 on the ninth row of stream s's top band (global row 8 = mid row 9; input rows 7, 8, 9 = rows 9, 10, 11 of
    // the full-resolution tile and rows 4, 5, 5 of the half-resolution one), activated and stored at once as the MFMA
    // wavefront's third tile was.  Copy wavefront rw: pixels 16 (rw & 1) .. + 15, couts 16 (rw >> 1) .. + 15.  Operand lane
    // maps of v_mfma_f32_16x16x32_bf16: lane l -> pixel / cout l & 15, channels 8 (l >> 4) .. + 7 of a 32-channel group
    // (three groups per tap: two of the upsampled map, one of skip1); D: pixel l & 15, couts 4 (l >> 4) .. + 3.
    auto top_row = [&](int s, Item it) {
        if constexpr (TOP9) {
            if (!it.act || it.y0 != 0) return;
            constexpr int PU = (W / 2 + 2) * SA, PB = (W + 2) * SB;
            const int q = lane & 15, cb = lane >> 4, hc = rw >> 1, x = 16 * (rw & 1) + q;
            const char *ta = tile_a(s) + cb * 16, *tb = tile_b(s) + (9 * (W + 2) + x) * SB + cb * 16;
            const __amdgpu_buffer_rsrc_t wr = make_rsrc(a.w1_alt, 9u * (CA + CB) * C1 * 2u);
            unsigned wl = (unsigned)(lane * 16 + hc * 1024);
            asm volatile("" : "+v"(wl));
            f32x4 acc = *(const f32x4 *)(c_b1 + 16 * hc + 4 * cb);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                bf16x8 wq[3][3], fq[3][3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int cg = 0; cg < 3; ++cg) {
                        wq[kx][cg] = buf_load16(wr, wl, (unsigned)((((ky * 3 + kx) * 3 + cg) * 2) * 1024));
                        fq[kx][cg] = cg < 2 ? *(const bf16x8 *)(ta + (ky == 0 ? 4 : 5) * PU + ((x + kx + 1) >> 1) * SA + cg * 64)
                                            : *(const bf16x8 *)(tb + ky * PB + kx * SB);
                    }
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int cg = 0; cg < 3; ++cg) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[kx][cg], fq[kx][cg], acc, 0, 0, 0);
            }
            *(bf16x4 *)(mid(s) + (9 * (W + 2) + x + 1) * SM + (16 * hc + 4 * cb) * 2) = pack4(silu1(acc[0]), silu1(acc[1]), silu1(acc[2]), silu1(acc[3]));
        }
    };
    static_assert(!TOP9 || (DEFER && BIAS_IN_ACC), "the ninth-row hand-over belongs to the deferred schedule");

    pf_load(ra[0], rb[0], item_of(0, 0));
    pf_load(ra[1], rb[1], item_of(1, 0));
    stage_in(ra[0], rb[0], 0, item_of(0, 0));
    stage_in(ra[1], rb[1], 1, item_of(1, 0));
    pf_load(ra[0], rb[0], item_of(0, 1));
    pf_load(ra[1], rb[1], item_of(1, 1));
    __syncthreads();                           // constants, halo columns, the first bands' tiles
    if constexpr (DEFER) {
        // the score tiles are complete one slot later than in the schedule below: X's after slot 4, Y's after slot 1
        for (int n = 0; n < niter; ++n) {
#ifdef IDM_TIMELINE
            const int tls_n = n;
#endif
            const Item X = item_of(0, n), Y = item_of(1, n);
            top_row(0, X);
            update(0, item_of(0, n - 1), n - 1);
            BAR(1);
            top_row(1, Y);
            update(1, item_of(1, n - 1), n - 1);
            stage_in(ra[0], rb[0], 0, item_of(0, n + 1));
            clear_pad_rows(0, X);              // (not in the slot of X's carry copy, which reads the rows this clears)
            BAR(2);
            stage_in(ra[1], rb[1], 1, item_of(1, n + 1));
            clear_pad_rows(1, Y);
            pf_load(ra[0], rb[0], item_of(0, n + 2));
            x_fetch(0, X, n);
            BAR(3);
            x_fetch(1, Y, n);
            pf_load(ra[1], rb[1], item_of(1, n + 2));
            BAR(4);
        }
        update(0, item_of(0, niter - 1), niter - 1);
        __syncthreads();
        update(1, item_of(1, niter - 1), niter - 1);
        return;
    }
    for (int n = 0; n < niter; ++n) {
#ifdef IDM_TIMELINE
        const int tls_n = n;
#endif
        const Item X = item_of(0, n), Y = item_of(1, n);
        update(1, item_of(1, n - 1), n - 1);
        clear_pad_rows(0, X);
        BAR(1);
        stage_in(ra[0], rb[0], 0, item_of(0, n + 1));
        clear_pad_rows(1, Y);
        x_fetch(0, X, n);
        BAR(2);
        stage_in(ra[1], rb[1], 1, item_of(1, n + 1));
        pf_load(ra[0], rb[0], item_of(0, n + 2));
        BAR(3);
        update(0, X, n);
        x_fetch(1, Y, n);
        pf_load(ra[1], rb[1], item_of(1, n + 2));
        BAR(4);
    }
    update(1, item_of(1, niter - 1), niter - 1);
}

// ---------------------------------------------------------------------------------------------
// fused_B with the two streams MERGED: conv8 and conv9 run on X's and Y's band in ONE MFMA loop each (ConvStage::piece in
// its two-lattice mode), so a weight fragment feeds four MFMAs instead of two and a pair of bands takes three barriers
// instead of four slots' set-up:
//
//     MFMA wavefronts 0-3                               copy wavefronts 4-7
//     conv8 (X, Y): loop; the heads of the previous      carry rows X, Y; fetch of the x quads to update
//     bands ride in it -> score tiles
//  B1 (every wavefront has issued its last MFMA on the input tiles; score tiles complete; carry done)
//     activation of conv8 (X, Y) -> mid tiles            state update of X's, Y's previous band; padding rows of the mid tiles
//  B2 (mid tiles complete)
//     conv9 (X, Y): loop -> pending head accumulators    stage-in of the next bands; prefetch of the bands after
//  B3
//
// conv8's activation runs exposed (no other loop to ride in), which the shared weights and the halved set-up outweigh.
// A wavefront's third conv8 row (nine-row top band) is a second, one-row piece after the first.  Tiles, partition of the
// rows, epilogue arithmetic, Philox counters and update are those of fused_b_copy_kernel: same results bit for bit.
// ---------------------------------------------------------------------------------------------
template <int WD1, int WD2, bool TOP9 = false>
__global__ __launch_bounds__(512, 1) void fused_b_pair_kernel(FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int W = 32, H = 32, R = 8, CA = 64, CB = 32, C1 = 32, C2 = 32;
    constexpr int SA = CA * 2 + 16, SB = CB * 2 + 16, SM = C1 * 2 + 16;
    constexpr int TILE_A = ((R / 2 + 2) * (W / 2 + 2) * SA + 15) & ~15, TILE_B = (R + 4) * (W + 2) * SB;
    constexpr int IN = TILE_A + TILE_B, MID = (R + 2) * (W + 2) * SM, SCORE = R * W * 4, STREAM = IN + MID + SCORE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool copier = wave >= 4;          // wave-uniform
    const int rw = wave & 3, rtid = tid & 255;
    const int h = lane >> 5, li = lane & 31;
    const int nlat = (int)a.B;

    float *cst = (float *)(smem + 2 * STREAM);
    const float *c_b1 = cst, *c_b2 = cst + C1, *c_wh = cst + C1 + C2, *c_bh = cst + C1 + 2 * C2;
    for (int i = tid; i < C1; i += 512) cst[i] = a.b1[i];
    for (int i = tid; i < C2; i += 512) {
        cst[C1 + i] = a.b2[i];
        cst[C1 + C2 + i] = a.w_head[i];
    }
    if (tid == 0) cst[C1 + 2 * C2] = a.b_head[0];
    auto tile_a = [&](int s) { return smem + s * STREAM; };
    auto tile_b = [&](int s) { return smem + s * STREAM + TILE_A; };
    auto mid = [&](int s) { return smem + s * STREAM + IN; };
    auto score = [&](int s) { return (float *)(smem + s * STREAM + IN + MID); };
    for (int s = 0; s < 2; ++s) {
        clear_halo<C1, 512>(mid(s), W, R + 2, 0, 0, tid);                  // halo columns: never written afterwards
        clear_halo<CA, 512>(tile_a(s), W >> 1, (R >> 1) + 2, 0, 0, tid);
        clear_halo<CB, 512>(tile_b(s), W, R + 4, 0, 0, tid);
    }

    // Item n of stream s: band n % 4 of the workgroup's (n / 4)-th lattice of that stream.
    struct Item { int b, y0; bool act; };
    auto item_of = [&](int s, int n) {
        Item it;
        const int k = n >> 2;
        it.b = (int)blockIdx.x + (2 * k + s) * (int)gridDim.x;
        it.y0 = (n & 3) * R;
        it.act = n >= 0 && it.b < nlat;
        return it;
    };
    auto band_geo = [&](Item it, int &rlo, int &rhi) {
        rlo = it.y0 == 0 ? 1 : 2;
        rhi = min(R + 2, H - it.y0 + 1);
    };
    const int per = 2 * (int)gridDim.x;
    const int nx = (int)blockIdx.x < nlat ? (nlat - 1 - (int)blockIdx.x) / per + 1 : 0;
    const int niter = nx * 4;   // stream Y ends no later than stream X
    const LaneGeo geo = make_geo(W, li);

    if (!copier) {
        static_assert(BIAS_IN_ACC && W == 32 && R == 8, "one row per tile, full eight-row bands, the bias in the accumulators");
        using Conv1 = ConvStage<CA, CB, true, C1, 2, 1, WD1, false, true>;
        using Conv2 = ConvStage<C1, 0, false, C2, 2, 1, WD2, true, true>;
        Conv1 cs1;
        Conv2 cs2;
        cs1.set_weights((const bf16x8 *)a.w1, lane);
        cs2.set_weights((const bf16x8 *)a.w2, lane);
        cs1.dual_off = STREAM;
        cs2.dual_off = STREAM;
        cs2.begin(mid(0), nullptr, c_b2, c_wh, geo, 0, R, (const bf16x8 *)a.w2, rw, lane, 0, 4, true);
        cs2.set_slice(0);
        constexpr int ROWB = (W + 2) * SM;
        char *const dummy = smem + 2 * STREAM + 512;            // 3 KB behind the constants: stores of a stream without work
        f32x4 kw[4];                                            // the head's weights of this lane's channels
        f32x16 ph[4];                                           // pending conv9 accumulators: rows 2 rw, 2 rw + 1 of X, then of Y
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) ph[n][i] = 0.0f;
        float *d9[2] = {(float *)dummy, (float *)dummy};       // where the pending head values of X / Y go
        f32x4 p4a = {0.f, 0.f, 0.f, 0.f};
        // chunk c of 32: values 2 (c & 1), 2 (c & 1) + 1 of quad (c >> 1) & 3 of pending tile c >> 3
        auto ride9 = [&](auto C) {
            constexpr int c = decltype(C)::value, n = c >> 3, g = (c >> 1) & 3, hf = c & 1;
            const f32x16 &t = ph[n];
            const float s0 = silu1(t[4 * g + 2 * hf]), s1 = silu1(t[4 * g + 2 * hf + 1]);
            if constexpr (g == 0) p4a[2 * hf] = p4a[2 * hf + 1] = 0.0f;
            p4a[2 * hf] += s0 * kw[g][2 * hf];           // (the same expression, hence the same contraction, as the
            p4a[2 * hf + 1] += s1 * kw[g][2 * hf + 1];   // four-at-a-time form of the undeferred epilogue)
            if constexpr ((c & 7) == 7) {
                float part = (p4a[0] + p4a[1]) + (p4a[2] + p4a[3]);
                part += __shfl_xor(part, 32, 64);
                if (h == 0) d9[n >> 1][(n & 1) * W] = part + c_bh[0];
            }
        };
        auto act_store = [&](const f32x16 &acc, char *dst) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(bf16x4 *)(dst + 16 * g) = pack4(silu1(acc[4 * g]), silu1(acc[4 * g + 1]), silu1(acc[4 * g + 2]), silu1(acc[4 * g + 3]));
        };
        cs1.arm();
        __syncthreads();                       // constants, halo columns, the first bands' tiles
#pragma unroll
        for (int q = 0; q < 4; ++q) kw[q] = *(const f32x4 *)(c_wh + 8 * q + 4 * h);
        for (int n = 0; n < niter; ++n) {
#ifdef IDM_TIMELINE
            const int tls_n = n;
#endif
            const Item X = item_of(0, n), Y = item_of(1, n);
            // ---- conv8 (X, Y); the pending heads ride in its loop; B1 between the loop and the activation ----
            {
                int rlo, rhi;
                band_geo(X, rlo, rhi);      // (X is active in every iteration; Y is in the same band of its lattice)
                // (TOP9: the top band's ninth row is the copy wavefronts')
                cs1.begin(tile_a(0), tile_b(0), c_b1, nullptr, geo, rlo, TOP9 ? min(rhi - rlo, R) : rhi - rlo, (const bf16x8 *)a.w1, rw, lane, 0, 4, true);
                char *const mx = mid(0), *const my = mid(1);
                f32x16 q[4];
                char *dx = nullptr;
                auto sink = [&](auto T, const f32x16 &acc, const f32x4 *, const f32x4 *, int, int r, int x, bool) {
                    constexpr int t = decltype(T)::value;
                    q[t] = acc;
                    if constexpr (t == 0) dx = (char *)(size_t)(__mul24(__mul24(r, W + 2) + x + 1, SM) + 8 * h);
                };
                auto inter = [&](auto ST) {
                    constexpr int st = decltype(ST)::value, NS = Conv1::NS;
                    static_for<32 * st / NS, 32 * (st + 1) / NS>(ride9);
                };
                auto hook = [&](int i) { if (i == 1) BAR(1); };
                if ((rlo + cs1.f) & 1) cs1.template piece<2, 1, true>(cs1.f, 0, 0, sink, hook, inter);
                else cs1.template piece<2, 0, true>(cs1.f, 0, 0, sink, hook, inter);
                const size_t off = (size_t)dx;
                const bool yact = Y.act;
                act_store(q[0], mx + off);
                act_store(q[1], mx + off + ROWB);
                act_store(q[2], yact ? my + off : dummy);
                act_store(q[3], yact ? my + off + ROWB : dummy);
                if (cs1.f1 - cs1.f == 3) {
                    // the nine-row top band gives one wavefront a third row: a one-row piece, finished at once
                    cs1.prologue();
                    size_t o3 = 0;
                    f32x16 q3[2];
                    auto sink3 = [&](auto T, const f32x16 &acc, const f32x4 *, const f32x4 *, int, int r, int x, bool) {
                        constexpr int t = decltype(T)::value;
                        q3[t] = acc;
                        if constexpr (t == 0) o3 = (size_t)(__mul24(__mul24(r, W + 2) + x + 1, SM) + 8 * h);
                    };
                    auto nohook = [](int) {};
                    if ((rlo + cs1.f) & 1) cs1.template piece<1, 1, true>(cs1.f + 2, 0, 0, sink3, nohook);
                    else cs1.template piece<1, 0, true>(cs1.f + 2, 0, 0, sink3, nohook);
                    act_store(q3[0], mx + o3);
                    act_store(q3[1], yact ? my + o3 : dummy);
                }
                cs1.arm();
                cs2.prologue();             // (here, not a slot earlier: its registers would be live across conv8's loop)
            }
            BAR(2);
            // ---- conv9 (X, Y) -> pending ----
            {
                float *sx = score(0), *sy = Y.act ? score(1) : (float *)dummy;
                auto sink = [&](auto T, const f32x16 &acc, const f32x4 *, const f32x4 *, int, int r, int x, bool) {
                    constexpr int t = decltype(T)::value;
                    ph[t] = acc;
                    if constexpr (t == 0) {
                        d9[0] = sx + r * W + x;
                        d9[1] = Y.act ? sy + r * W + x : (float *)dummy;
                    }
                };
                auto nohook = [](int) {};
                cs2.template piece<2, -1, true>(2 * rw, 0, 0, sink, nohook);
            }
            BAR(3);
        }
        static_for<0, 32>(ride9);              // the heads of the last bands
        __syncthreads();
        return;
    }

    // ------------------------------ copy wavefronts ------------------------------
    TileRegs<CA, 3, 256> ra[2];
    TileRegs<CB, 6, 256> rb[2];
    auto pf_load = [&](auto &qa, auto &qb, Item it) {
        if (!it.act) return;
        int t = rtid;
        asm volatile("" : "+v"(t));
        __builtin_assume(t >= 0 && t < 256);
        qa.load((const __bf16 *)a.in_a + (size_t)it.b * (H / 2) * (W / 2) * CA, H / 2, W / 2, (it.y0 >> 1) - 1, (R >> 1) + 2, t);
        qb.load((const __bf16 *)a.in_b + (size_t)it.b * H * W * CB, H, W, it.y0 - 2, R + 4, t);
    };
    auto stage_in = [&](auto &qa, auto &qb, int s, Item it) {
        if (!it.act) return;
        int t = rtid;
        asm volatile("" : "+v"(t));
        __builtin_assume(t >= 0 && t < 256);
        qa.template store<false>(tile_a(s), W >> 1, (R >> 1) + 2, t);
        qb.template store<false>(tile_b(s), W, R + 4, t);
    };
    // rows of the mid tile outside the map are conv9's zero padding
    auto clear_pad_rows = [&](int s, Item it) {
        if (!it.act) return;
        int rlo, rhi;
        band_geo(it, rlo, rhi);
        if (it.y0 == 0) clear_halo<C1, 256>(mid(s), W, 0, 0, 1, rtid);
        if (rhi < R + 2) clear_halo<C1, 256>(mid(s), W, 0, rhi, R + 2, rtid);
    };
    // rows R, R+1 of the mid tile are rows 0, 1 of the lattice's next band (it: the band just finished)
    auto carry = [&](int s, Item it) {
        if (!it.act || it.y0 + R >= H) return;
        char *m = mid(s);
        const int bytes = 2 * (W + 2) * SM;
        const char *src = m + R * (W + 2) * SM;
        for (int i = rtid * 16; i < bytes; i += 256 * 16) *(f32x4 *)(m + i) = *(const f32x4 *)(src + i);
    };
    // The 64 float4 quads of a band (one Philox counter each) are one wavefront's worth of lanes: the copy wavefronts take
    // the bands in turn (see fused_b_copy_kernel).
    f32x4 xq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, zq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    auto my_turn = [&](int s, int n) { return ((n + 2 * s) & 3) == rw; };
    auto quad_of = [&](Item it, int &qs, int64_t &off) {
        const int qr = lane >> 3, qx = lane & 7;
        qs = qr * W + qx * 4;
        off = (int64_t)it.b * (H * W) + (it.y0 + qr) * W + qx * 4;
    };
    auto x_fetch = [&](int s, Item it, int n) {
        if (!it.act || !a.do_update || !my_turn(s, n)) return;
        int qs;
        int64_t off;
        quad_of(it, qs, off);
        xq[s] = *(const f32x4 *)(a.x + off);
        if (a.add_noise && a.noise) zq[s] = *(const f32x4 *)(a.noise + off);
    };
    auto update = [&](int s, Item it, int n) {
        if (!it.act || !my_turn(s, n)) return;
        int qs;
        int64_t off;
        quad_of(it, qs, off);
        const f32x4 sc = *(const f32x4 *)(score(s) + qs);
        if (a.score_out) *(f32x4 *)(a.score_out + off) = sc;
        if (a.do_update) {
            float z[4] = {zq[s][0], zq[s][1], zq[s][2], zq[s][3]};
            if (a.add_noise && !a.noise) {
                uint32_t k0 = (uint32_t)a.ph.seed, k1 = (uint32_t)(a.ph.seed >> 32);
                normal4(((uint64_t)k1 << 32) | k0, a.ph.first_global_index + it.b, (uint32_t)((off - (int64_t)it.b * (H * W)) >> 2),
                        a.ph.t, z);
            }
            f32x4 xv = xq[s];
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[j] = reverse_update(a.c1, xv[j], a.beta, sc[j], z[j], a.sd, a.add_noise != 0);
            *(f32x4 *)(a.x + off) = xv;
        }
    };

    // TOP9, as in fused_b_copy_kernel: conv8 on the ninth row of stream s's top band (global row 8 = mid row 9; input rows 7, 8, 9 = rows 9, 10, 11 of
    // the full-resolution tile and rows 4, 5, 5 of the half-resolution one), activated and stored at once as the MFMA
    // wavefront's third tile was.  Copy wavefront rw: pixels 16 (rw & 1) .. + 15, couts 16 (rw >> 1) .. + 15.  Operand lane
    // maps of v_mfma_f32_16x16x32_bf16: lane l -> pixel / cout l & 15, channels 8 (l >> 4) .. + 7 of a 32-channel group
    // (three groups per tap: two of the upsampled map, one of skip1); D: pixel l & 15, couts 4 (l >> 4) .. + 3.
    auto top_row = [&](int s, Item it) {
        if constexpr (TOP9) {
            if (!it.act || it.y0 != 0) return;
            constexpr int PU = (W / 2 + 2) * SA, PB = (W + 2) * SB;
            const int q = lane & 15, cb = lane >> 4, hc = rw >> 1, x = 16 * (rw & 1) + q;
            const char *ta = tile_a(s) + cb * 16, *tb = tile_b(s) + (9 * (W + 2) + x) * SB + cb * 16;
            const __amdgpu_buffer_rsrc_t wr = make_rsrc(a.w1_alt, 9u * (CA + CB) * C1 * 2u);
            unsigned wl = (unsigned)(lane * 16 + hc * 1024);
            asm volatile("" : "+v"(wl));
            f32x4 acc = *(const f32x4 *)(c_b1 + 16 * hc + 4 * cb);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                bf16x8 wq[3][3], fq[3][3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int cg = 0; cg < 3; ++cg) {
                        wq[kx][cg] = buf_load16(wr, wl, (unsigned)((((ky * 3 + kx) * 3 + cg) * 2) * 1024));
                        fq[kx][cg] = cg < 2 ? *(const bf16x8 *)(ta + (ky == 0 ? 4 : 5) * PU + ((x + kx + 1) >> 1) * SA + cg * 64)
                                            : *(const bf16x8 *)(tb + ky * PB + kx * SB);
                    }
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int cg = 0; cg < 3; ++cg) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[kx][cg], fq[kx][cg], acc, 0, 0, 0);
            }
            *(bf16x4 *)(mid(s) + (9 * (W + 2) + x + 1) * SM + (16 * hc + 4 * cb) * 2) = pack4(silu1(acc[0]), silu1(acc[1]), silu1(acc[2]), silu1(acc[3]));
        }
    };

    pf_load(ra[0], rb[0], item_of(0, 0));
    pf_load(ra[1], rb[1], item_of(1, 0));
    stage_in(ra[0], rb[0], 0, item_of(0, 0));
    stage_in(ra[1], rb[1], 1, item_of(1, 0));
    pf_load(ra[0], rb[0], item_of(0, 1));
    pf_load(ra[1], rb[1], item_of(1, 1));
    __syncthreads();                           // constants, halo columns, the first bands' tiles
    for (int n = 0; n < niter; ++n) {
#ifdef IDM_TIMELINE
        const int tls_n = n;
#endif
        const Item X = item_of(0, n), Y = item_of(1, n);
        const Item Xp = item_of(0, n - 1), Yp = item_of(1, n - 1);
        top_row(0, X);
        top_row(1, Y);
        carry(0, Xp);
        carry(1, Yp);
        x_fetch(0, Xp, n - 1);
        x_fetch(1, Yp, n - 1);
        BAR(1);
        update(0, Xp, n - 1);
        update(1, Yp, n - 1);
        clear_pad_rows(0, X);                  // (after the carry copy, which reads the rows this clears)
        clear_pad_rows(1, Y);
        BAR(2);
        stage_in(ra[0], rb[0], 0, item_of(0, n + 1));
        stage_in(ra[1], rb[1], 1, item_of(1, n + 1));
        pf_load(ra[0], rb[0], item_of(0, n + 2));
        pf_load(ra[1], rb[1], item_of(1, n + 2));
        BAR(3);
    }
    __syncthreads();                           // the heads of the last bands
    x_fetch(0, item_of(0, niter - 1), niter - 1);
    x_fetch(1, item_of(1, niter - 1), niter - 1);
    update(0, item_of(0, niter - 1), niter - 1);
    update(1, item_of(1, niter - 1), niter - 1);
}

// fused_b_copy_kernel: NT1, WD1, NT2, WD2
#ifndef IDM_CFG_BSPLIT
#define IDM_CFG_BSPLIT 3, 2, 2, 3
#endif
// fused_b_pair_kernel: WD1, WD2
#ifndef IDM_CFG_BPAIR
#define IDM_CFG_BPAIR 1, 2
#endif
hipError_t launch_b_split(const FusedArgs &a, int64_t B, hipStream_t s)
{
    static thread_local struct { bool configured = false; int cus = 0; } all_st[IDM_MAX_DEVICES][8];   // one slot per kernel function
    const int slot = idm_device_slot();
    if (slot < 0) return hipErrorInvalidDevice;
    // ISINGDIFF_B_DEFER=0: epilogues right after their MFMA loops (comparison runs)
    static const bool defer = [] { const char *e = getenv("ISINGDIFF_B_DEFER"); return !(e && e[0] == '0'); }();
    // the merged-stream kernel (same bits as the alternating one; a little faster itself, and it draws less power, which the
    // other kernels of the step get back as clock: whole job +0.8 %) unless ISINGDIFF_B_PAIR=0
    static const bool pair = [] { const char *e = getenv("ISINGDIFF_B_PAIR"); return !(e && e[0] == '0'); }();
    // the top band's ninth conv8 row on the copy wavefronts (merged and deferred schedules; needs w1_alt) unless
    // ISINGDIFF_B_TOP9=0 (then every row is summed in the order of the four-wavefront kernels)
    static const bool top9_env = [] { const char *e = getenv("ISINGDIFF_B_TOP9"); return !(e && e[0] == '0'); }();
    const bool top9 = top9_env && a.w1_alt != nullptr;
    auto kernel = pair ? (top9 ? fused_b_pair_kernel<IDM_CFG_BPAIR, true> : fused_b_pair_kernel<IDM_CFG_BPAIR>)
                  : defer ? (top9 ? fused_b_copy_kernel<IDM_CFG_BSPLIT, true, true> : fused_b_copy_kernel<IDM_CFG_BSPLIT, true>)
                          : fused_b_copy_kernel<IDM_CFG_BSPLIT, false>;
    auto &st = all_st[slot][(pair ? 4 : 0) + (defer ? 2 : 0) + (top9 ? 1 : 0)];
    constexpr int tile_a = ((8 / 2 + 2) * (32 / 2 + 2) * 144 + 15) & ~15, tile_b = 12 * 34 * 80, mid = 10 * 34 * 80, score = 8 * 32 * 4;
    constexpr int lds = 2 * (tile_a + tile_b + mid + score) + 512 + 3072;   // constants (400 B) and the scratch area of the deferred epilogues
    static_assert(lds <= 160 * 1024, "the tiles must fit a CU");
    if (!st.configured) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t prop;
        if ((e = hipGetDevice(&dev)) != hipSuccess || (e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
        st.cus = prop.multiProcessorCount;
        st.configured = true;
    }
    if (B >= ((int64_t)1 << 29)) return hipErrorInvalidValue;
    const int64_t grid = std::min<int64_t>((B + 1) / 2, (int64_t)st.cus);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(512), lds, s, a);
    return hipGetLastError();
}
