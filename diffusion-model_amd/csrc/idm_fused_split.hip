// fused_B (conv8, conv9, head, reverse update) for the L = 32 lattice with the work split by kind over the eight
// wavefronts of a workgroup.  This translation unit is compiled with -fno-slp-vectorize: its epilogues must stay on the
// plain (unpacked) float32 ALU, because beside a wavefront that streams MFMAs packed float32 instructions wait for the
// matrix pipe while plain ones do not (inline assembly is no alternative: the compiler does not see that it consumes a
// transcendental's result and leaves out the wait state that needs).
#include "idm_conv_stage.h"

#include <algorithm>

#ifdef IDM_TIMELINE
// fused_b_split_kernel: [role][iteration][2 k] = arrival at barrier k + 1, [2 k + 1] = departure from it (wavefronts 0 and 4
// of one workgroup).
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
#define TLS(i) do { } while (0)
#define BAR(k) __syncthreads()
#endif

// ---------------------------------------------------------------------------------------------
// fused_B for the L = 32 lattice with the work split by KIND over eight wavefronts (one workgroup per CU).
//
// A wavefront issues one instruction per ~8 cycles whatever its kind, and an MFMA holds the pipe for four such slots,
// so the ~1 250 non-MFMA instructions a band costs a wavefront of the kernel above (epilogues, tile copies, noise
// draw, update) cannot hide behind its own 144 MFMAs, and two independent workgroups per CU overlap them only by
// chance.  Here wavefronts 0-3 (one per SIMD) run nothing but the two MFMA loops and hand their raw float32
// accumulators over through LDS; wavefronts 4-7 (the second wavefront of each SIMD) do everything else.  Two lattices
// ("streams" X and Y, each with its own input tiles and mid tile) are in flight so that both kinds always have work;
// eight workgroup barriers per pair of bands keep them in step:
//
//     MFMA wavefronts                       other wavefronts
//     conv8 loop (X)                        conv9 epilogue + head + state update of Y's previous band, carry rows
//  B1 accumulators -> hand-over (X)         stage-in of Y's band (registers -> input tiles)
//  B2 conv8 loop (Y)                        conv8 epilogue (X): hand-over -> bias/silu/pack -> mid tile; prefetch X's next band
//  B3 accumulators -> hand-over (Y)
//  B4 conv9 loop (X)                        conv8 epilogue (Y)
//  B5 accumulators -> hand-over (X)
//  B6 conv9 loop (Y)                        conv9 epilogue + head + update (X), carry rows
//  B7 accumulators -> hand-over (Y)         stage-in of X's next band; prefetch Y's next band
//  B8
//
// The hand-over buffer of a stream lies over its own input tiles (dead between a band's conv8 loop and the next
// stage-in), which is why the accumulators are written only after the barrier that ends the loop (B1, B3, B5, B7: issued
// from inside the convolution stage, between its last MFMA and its epilogue), and why stage-in rewrites the tiles' halo
// columns every band.  Pixel stride of the hand-over is 144 bytes (32 float32 + 16): as for the activation tiles, the
// 16-lane groups of a 16-byte access then fall on distinct banks.  The arithmetic of every epilogue, the Philox counters
// and the update are those of fused2_kernel, so the results are bit-identical to it.
// ---------------------------------------------------------------------------------------------
template <int NT1, int WD1, int NT2, int WD2>
__global__ __launch_bounds__(512, 1) void fused_b_split_kernel(FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int W = 32, H = 32, R = 8, NBANDS = 4, CA = 64, CB = 32, C1 = 32, C2 = 32;
    constexpr int SA = CA * 2 + 16, SB = CB * 2 + 16, SM = C1 * 2 + 16, SH = 32 * 4 + 16;
    constexpr int TILE_A = ((R / 2 + 2) * (W / 2 + 2) * SA + 15) & ~15, TILE_B = (R + 4) * (W + 2) * SB;
    constexpr int IN = TILE_A + TILE_B, MID = (R + 2) * (W + 2) * SM, STREAM = IN + MID;
    static_assert((R + 2) * W * SH <= IN, "the hand-over buffer must fit the input tiles it lies over");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool helper = wave >= 4;          // wave-uniform
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
    auto handover = [&](int s) { return smem + s * STREAM; };
    clear_halo<C1, 512>(mid(0), W, R + 2, 0, 0, tid);   // halo columns of the mid tiles: never written afterwards
    clear_halo<C1, 512>(mid(1), W, R + 2, 0, 0, tid);

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
    // lattices of stream X: blockIdx.x, blockIdx.x + 2 grid, ...; the loop runs until X has none left (Y ends no later)
    const int per = 2 * (int)gridDim.x;
    const int nx = (int)blockIdx.x < nlat ? (nlat - 1 - (int)blockIdx.x) / per + 1 : 0;
    const int niter = nx * NBANDS;
    const LaneGeo geo = make_geo(W, li);
#ifdef IDM_BS_NOARM
    constexpr bool ARM = false;
#else
    constexpr bool ARM = true;
#endif

    if (!helper) {
        // ------------------------------ MFMA wavefronts ------------------------------
#ifdef IDM_BS_PRIO
        __builtin_amdgcn_s_setprio(IDM_BS_PRIO);
#endif
        using Conv1 = ConvStage<CA, CB, true, C1, NT1, 1, WD1, false>;
        using Conv2 = ConvStage<C1, 0, false, C2, NT2, 1, WD2, false>;
        Conv1 cs1;
        Conv2 cs2;
        auto begin8 = [&](int s, Item it) {
            if (!it.act) return;
            int rlo, rhi;
            band_geo(it, rlo, rhi);
            cs1.begin(tile_a(s), tile_b(s), c_b1, nullptr, geo, rlo, rhi - rlo, (const bf16x8 *)a.w1, rw, lane, 0, 4, ARM);
        };
        auto begin9 = [&](int s, Item it) {
            if (!it.act) return;
            cs2.begin(mid(s), nullptr, c_b2, nullptr, geo, 0, R, (const bf16x8 *)a.w2, rw, lane, 0, 4, ARM);
        };
        // runs the loop of one stage, meets the other wavefronts at the barrier that ends it, then writes the accumulators
        int tls_n = 0;
        (void)tls_n;
        // (the first weights of a stage's next run are requested as soon as its loop ends: arm())
        auto run_stage = [&](auto &cs, int s, bool act, int k) {
            (void)k;
            bool synced = false;
            if (act) {
                char *ho = handover(s);
                cs.run(
                    [&](const f32x16 &acc, const f32x4 *, const f32x4 *, int, int r, int x, bool valid) {
                        if (!valid) return;
                        char *dst = ho + (r * W + x) * SH + 16 * h;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *(f32x4 *)(dst + 32 * q) = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                    },
                    [&](int i) {
                        if (i == 1 && !synced) {
                            if (ARM) cs.arm();
                            BAR(k);
                            synced = true;
                        }
                    });
            }
            if (!synced) {
                if (act && ARM) cs.arm();   // a wavefront without a piece in this band
                BAR(k);
            }
        };
        __syncthreads();                       // constants, halo columns, first stage-in
        if (ARM) {
            cs1.set_weights((const bf16x8 *)a.w1, lane);
            cs2.set_weights((const bf16x8 *)a.w2, lane);
            cs1.arm();
            cs2.arm();
        }
        begin8(0, item_of(0, 0));
        for (int n = 0; n < niter; ++n) {
            const Item X = item_of(0, n), Y = item_of(1, n);
            run_stage(cs1, 0, X.act, 1);       // B1 inside
            begin8(1, Y);
            BAR(2);
            run_stage(cs1, 1, Y.act, 3);       // B3 inside
            begin9(0, X);
            BAR(4);
            run_stage(cs2, 0, X.act, 5);       // B5 inside
            begin9(1, Y);
            BAR(6);
            run_stage(cs2, 1, Y.act, 7);       // B7 inside
            begin8(0, item_of(0, n + 1));
            BAR(8);
#ifdef IDM_TIMELINE
            ++tls_n;
#endif
        }
        return;
    }

    // ------------------------------ the other wavefronts ------------------------------
    TileRegs<CA, 3, 256> ra;
    TileRegs<CB, 6, 256> rb;
    auto pf_load = [&](Item it) {
        if (!it.act) return;
        int t = rtid;
        asm volatile("" : "+v"(t));
        __builtin_assume(t >= 0 && t < 256);
        ra.load((const __bf16 *)a.in_a + (size_t)it.b * (H / 2) * (W / 2) * CA, H / 2, W / 2, (it.y0 >> 1) - 1, (R >> 1) + 2, t);
        rb.load((const __bf16 *)a.in_b + (size_t)it.b * H * W * CB, H, W, it.y0 - 2, R + 4, t);
    };
    auto stage_in = [&](int s, Item it) {
        if (!it.act) return;
        int t = rtid;
        asm volatile("" : "+v"(t));
        __builtin_assume(t >= 0 && t < 256);
        ra.template store<true>(tile_a(s), W >> 1, (R >> 1) + 2, t);
        rb.template store<true>(tile_b(s), W, R + 4, t);
    };
    // conv8's epilogue for the rows MFMA wavefront rw computed (the same even split of the band's rows)
    auto epilogue8 = [&](int s, Item it) {
        if (!it.act) return;
        int rlo, rhi;
        band_geo(it, rlo, rhi);
        char *m = mid(s);
        if (it.y0 == 0) clear_halo<C1, 256>(m, W, 0, 0, 1, rtid);              // padding row above the map
        if (rhi < R + 2) clear_halo<C1, 256>(m, W, 0, rhi, R + 2, rtid);       // padding rows below it
        const int tot = rhi - rlo, f = (rw * tot) >> 2, f1 = ((rw + 1) * tot) >> 2;
        const char *ho = handover(s);
        for (int r = rlo + f; r < rlo + f1; ++r) {
            const char *src = ho + (r * W + li) * SH + 16 * h;
            char *dst = m + (r * (W + 2) + li + 1) * SM + (4 * h) * 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 z = *(const f32x4 *)(src + 32 * q);
                *(bf16x4 *)(dst + 16 * q) = pack4(silu1(z[0]), silu1(z[1]), silu1(z[2]), silu1(z[3]));
            }
        }
    };
    // conv9's epilogue, the 1x1 head and the state update for rows 2 rw and 2 rw + 1 of the band: after the reduction
    // over the two half-waves every lane holds both rows' scores of its column, and lane half h updates row 2 rw + h.
    auto epilogue9_update = [&](int s, Item it) {
        if (!it.act) return;
        const int row = 2 * rw + h;
        const int e = (it.y0 + row) * W + li;
        const int64_t off = (int64_t)it.b * (H * W) + e;
        float xv = 0.f, zin = 0.f;
        if (a.do_update) {
            xv = a.x[off];
            if (a.add_noise && a.noise) zin = a.noise[off];
        }
        f32x4 kw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) kw[q] = *(const f32x4 *)(c_wh + 8 * q + 4 * h);
        const char *ho = handover(s);
        float sc2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const char *src = ho + ((2 * rw + t) * W + li) * SH + 16 * h;
            float p4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 z = *(const f32x4 *)(src + 32 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) p4[i] += silu1(z[i]) * kw[q][i];
            }
            float part = (p4[0] + p4[1]) + (p4[2] + p4[3]);
            part += __shfl_xor(part, 32, 64);
            sc2[t] = part + c_bh[0];
        }
        const float sc = h ? sc2[1] : sc2[0];
        if (a.score_out) a.score_out[off] = sc;
        if (a.do_update) {
            float z = zin;
            if (a.add_noise && !a.noise) {
                float z4[4];
                uint32_t k0 = (uint32_t)a.ph.seed, k1 = (uint32_t)(a.ph.seed >> 32);
                normal4(((uint64_t)k1 << 32) | k0, a.ph.first_global_index + it.b, (uint32_t)(e >> 2), a.ph.t, z4);
                z = (e & 2) ? ((e & 1) ? z4[3] : z4[2]) : ((e & 1) ? z4[1] : z4[0]);
            }
            a.x[off] = reverse_update(a.c1, xv, a.beta, sc, z, a.sd, a.add_noise != 0);
        }
        // rows R, R+1 of the mid tile are rows 0, 1 of this lattice's next band
        if (it.y0 + R < H) {
            char *m = mid(s);
            const int bytes = 2 * (W + 2) * SM;
            const char *src = m + R * (W + 2) * SM;
            for (int i = rtid * 16; i < bytes; i += 256 * 16) *(f32x4 *)(m + i) = *(const f32x4 *)(src + i);
        }
    };

    int tls_n = 0;
    (void)tls_n;
    pf_load(item_of(0, 0));
    stage_in(0, item_of(0, 0));
    pf_load(item_of(1, 0));
    __syncthreads();                           // constants, halo columns, first stage-in
    for (int n = 0; n < niter; ++n) {
        const Item X = item_of(0, n), Y = item_of(1, n);
        epilogue9_update(1, item_of(1, n - 1));
        BAR(1);
        stage_in(1, Y);
        BAR(2);
        epilogue8(0, X);
        pf_load(item_of(0, n + 1));
        BAR(3);
        BAR(4);
        epilogue8(1, Y);
        BAR(5);
        BAR(6);
        epilogue9_update(0, X);
        BAR(7);
        stage_in(0, item_of(0, n + 1));
        pf_load(item_of(1, n + 1));
        BAR(8);
#ifdef IDM_TIMELINE
        ++tls_n;
#endif
    }
    epilogue9_update(1, item_of(1, niter - 1));
}

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
template <int NT1, int WD1, int NT2, int WD2>
__global__ __launch_bounds__(512, 1) void fused_m3_split_kernel(FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int W = 16, H = 16, R = 16, CA = 128, CB = 64, C1 = 64, C2 = 64;
    constexpr int SA = CA * 2 + 16, SB = CB * 2 + 16, SM = C1 * 2 + 16, SO = C2 * 2 + 16;
    constexpr int TILE_A = ((R / 2 + 2) * (W / 2 + 2) * SA + 15) & ~15, TILE_B = (R + 4) * (W + 2) * SB;
    constexpr int IN = TILE_A + TILE_B, MID = (R + 2) * (W + 2) * SM, STAGE = R * W * SO;
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
    // the mid tile's halo columns and its two padding rows (above and below the map) are never written afterwards
    clear_halo<C1, 512>(mid, W, R + 2, 0, 1, tid);
    clear_halo<C1, 512>(mid, W, 0, R + 1, R + 2, tid);
    const int first = (int)blockIdx.x, step = (int)gridDim.x;
    const LaneGeo geo = make_geo(W, lane & 31);

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

    TileRegs<CA, 5, 256> ra;
    TileRegs<CB, 10, 256> rb;
    auto pf_load = [&](int b) {
        if (b >= nlat) return;
        int t = rtid;
        asm volatile("" : "+v"(t));
        __builtin_assume(t >= 0 && t < 256);
        ra.load((const __bf16 *)a.in_a + (size_t)b * (H / 2) * (W / 2) * CA, H / 2, W / 2, -1, (R >> 1) + 2, t);
        rb.load((const __bf16 *)a.in_b + (size_t)b * H * W * CB, H, W, -2, R + 4, t);
    };
    auto stage_in = [&](int b) {
        if (b >= nlat) return;
        int t = rtid;
        asm volatile("" : "+v"(t));
        __builtin_assume(t >= 0 && t < 256);
        ra.template store<true>(tile_a, W >> 1, (R >> 1) + 2, t);
        rb.template store<true>(tile_b, W, R + 4, t);
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
    auto kernel = fused_m3_split_kernel<4, 2, 4, 3>;
    constexpr int tile_a = ((16 / 2 + 2) * (16 / 2 + 2) * 272 + 15) & ~15, tile_b = 20 * 18 * 144, mid = 18 * 18 * 144,
                  stage = 16 * 16 * 144;
    constexpr int lds = tile_a + tile_b + mid + stage + (64 + 64) * 4;
    static_assert(lds <= 160 * 1024, "the tiles must fit a CU");
    static thread_local struct { bool configured = false; int cus = 0; } all_st[IDM_MAX_DEVICES];
    const int slot = idm_device_slot();
    if (slot < 0) return hipErrorInvalidDevice;
    auto &st = all_st[slot];
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

// fused_b_split_kernel: one workgroup of eight wavefronts per CU, two lattices each.
#ifndef IDM_CFG_BSPLIT
#define IDM_CFG_BSPLIT 3, 2, 2, 3
#endif
hipError_t launch_b_split(const FusedArgs &a, int64_t B, hipStream_t s)
{
    static thread_local struct { bool configured = false; int cus = 0; } all_st[IDM_MAX_DEVICES];
    const int slot = idm_device_slot();
    if (slot < 0) return hipErrorInvalidDevice;
    auto &st = all_st[slot];
    auto kernel = fused_b_split_kernel<IDM_CFG_BSPLIT>;
    constexpr int tile_a = ((8 / 2 + 2) * (32 / 2 + 2) * 144 + 15) & ~15, tile_b = 12 * 34 * 80, mid = 10 * 34 * 80;
    constexpr int lds = 2 * (tile_a + tile_b + mid) + (32 + 3 * 32 + 4) * 4;
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

