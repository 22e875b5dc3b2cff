// Fused two-convolution band kernels (bf16 MFMA, float32 accumulate).
//
// A persistent workgroup owns whole lattices and walks each one's bands of R output rows top to bottom:
//     copy-in -> conv1 -> mid tile (LDS) -> conv2 -> epilogue / copy-out
// so the intermediate map never leaves the CU.  conv2 on rows y0 .. y0+R-1 needs conv1 on rows y0-1 .. y0+R
// (mid tile rows 0 .. R+1).  Its two bottom rows are the next band's rows 0 and 1: they are CARRIED OVER in
// LDS (copied to the top of the mid tile once conv2 has read them), so conv1 computes every row of the map
// exactly once (rows y0+1 .. y0+R per band, one more in a lattice's top band) and nothing is recomputed.
//
// Barriers per band (B1..B4; HEAD kernels have no B4) and what each phase touches:
//     [prefetched input registers -> input tiles]                      writes: input tiles
//  B1 the input tiles are complete; the previous band's carry copy (all wavefronts) is done
//     clear the mid rows that lie outside the map (top band: row 0; last band: rows below the map) - AFTER B1,
//     because in a lattice's last band those are the very rows the previous band's carry copy wrote
//     conv1                                                            reads: input tiles; writes: mid rows inside the map
//  B2 the mid tile is complete
//     conv2                                                            reads: mid tile; writes: staging tile (= input region) / score tile
//  B3 conv2 has read every mid row; staging / score tile complete
//     carry rows R, R+1 -> rows 0, 1; copy-out / head + state update    reads: mid rows R, R+1, staging / score tile
//  B4 (store kernels) the staging tile, which aliases the input region, has been read before the next copy-in.
//     HEAD keeps its score tile apart from the input tiles, so its next copy-in needs no barrier: the carry copy
//     and the next band's clears are ordered by the next B1.
// Moving a clear ahead of B1, or the carry copy ahead of B3, is a race that shows only as rare wrong bits; the
// batch-composition property test (a lattice's result does not depend on which workgroup ran it) guards it.
// The conv2 + conv3 instance on the whole 16 x 16 map (ALIAS in the kernel) keeps its staging tile OVER the mid tile:
//  B2' between conv2's MFMA loop and its epilogue - every wavefront has issued its last MFMA on the mid tile before any
//     writes the staging tile (each wavefront runs exactly one piece, so all reach it once);
//  B4 then also orders the copy-out's reads of the staging tile before the next lattice's conv1 epilogue writes the mid
//     tile, whose halo columns and padding rows are cleared again after B1 (the staged output lay over them).
//
// The UNet's data movement is folded into the edges:
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
#include "idm_conv_stage.h"

#include <algorithm>
#include <cstdlib>

namespace {

// First layer (Cin = 1): K = 9 taps in one MFMA step; x = hi + lo in bf16, two MFMAs per tile.
// xs: tile [(rows)][W+2] of (hi, lo) bf16 pairs, split once per pixel when the band is staged (split_state) instead of
// once per pixel and tap here.  wfirst: one A fragment (lane l: cout l&31, k = 8*(l>>5)+j -> tap k).
DEVI uint32_t split_state(float v)
{
    const __bf16 hi = (__bf16)v;
    const __bf16 lo = (__bf16)(v - (float)hi);
    return (uint32_t)__builtin_bit_cast(uint16_t, hi) | ((uint32_t)__builtin_bit_cast(uint16_t, lo) << 16);
}

template <class Epi>
DEVI void conv_first_stage(const uint32_t *xs, const float *c0, int W, int r0, int npix, const bf16x8 *__restrict__ wfirst,
                           int wave, int lane, Epi &&epi)
{
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    f32x4 k0[4], k1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        k0[q] = *(const f32x4 *)(c0 + 8 * q + 4 * (lane >> 5));
        k1[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int ntiles = (npix + 31) >> 5;
    const int h = lane >> 5, li = lane & 31, Wp2 = W + 2;
    const bf16x8 a = wfirst[lane];
    for (int t = wave; t < ntiles; t += 4) {
        const int p = t * 32 + li;
        const bool valid = p < npix;
        const int pc = valid ? p : 0;
        const int r = pc / W + r0, x = pc - (r - r0) * W;
        const uint32_t *q = xs + r * Wp2 + x;
        uint32_t v[8];
        if (h == 0) {
            v[0] = q[0]; v[1] = q[1]; v[2] = q[2];
            v[3] = q[Wp2]; v[4] = q[Wp2 + 1]; v[5] = q[Wp2 + 2];
            v[6] = q[2 * Wp2]; v[7] = q[2 * Wp2 + 1];
        } else {
            v[0] = q[2 * Wp2 + 2];
#pragma unroll
            for (int j = 1; j < 8; ++j) v[j] = 0u;
        }
        u32x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            hi[j] = __builtin_amdgcn_perm(v[2 * j + 1], v[2 * j], 0x05040100u);   // the low halves of the two words
            lo[j] = __builtin_amdgcn_perm(v[2 * j + 1], v[2 * j], 0x07060302u);   // the high halves
        }
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = BIAS_IN_ACC ? k0[i >> 2][i & 3] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, hi), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, lo), acc, 0, 0, 0);
        epi(acc, k0, k1, 0, r, x, valid);
    }
}

}  // namespace

#ifdef IDM_TIMELINE
// Debug build only: phase timestamps of one workgroup (wavefronts 0 and 3), [kind][wave sel][band][stamp], and, for
// fused_B, of wavefront 0 of the first 1024 workgroups during bands 8-23 together with the hardware id of the CU they
// run on (to line up the workgroups that share a CU).
__device__ long long g_tl[5][2][64][32];
__device__ long long g_tl2[1024][16][16];
__device__ unsigned g_hw[1024][2];
extern "C" int idm_debug_timeline(long long *host)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tl), sizeof(g_tl));
}
extern "C" int idm_debug_timeline2(long long *host, unsigned *hw)
{
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tl2), sizeof(g_tl2));
    if (e == hipSuccess) e = hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_hw), sizeof(g_hw));
    return (int)e;
}
#define TL(i)                                                                                          \
    do {                                                                                               \
        if (blockIdx.x == 7 && lane == 0 && (wave == 0 || wave == 3) && tl_it < 64)                    \
            g_tl[TLK][wave == 3][tl_it][i] = (i) == 9 ? (long long)wall_clock64() : (long long)clock64(); \
        if (TLK == 4 && blockIdx.x < 1024 && lane == 0 && wave == 0 && tl_it >= 8 && tl_it < 24 && (i) < 16) \
            g_tl2[blockIdx.x][tl_it - 8][i] = (long long)clock64();                                    \
    } while (0)
#define TLHOOK(base) [&](int i) { if ((base) + i < 24 && ((base) == 10 || i < 6)) TL((base) + i); }
#else
#define TL(i) do { } while (0)
#define TLHOOK(base) [](int) {}
#endif

// Persistent: each workgroup walks the bands of lattices blockIdx.x, blockIdx.x + gridDim.x, ...
//
// Vector-memory results return in issue order, so a wait for a weight fragment also waits for
// every older load or store of the wavefront.  The loop is therefore arranged so that long-latency
// traffic is always YOUNGER than the weights the next convolution needs first:
//     conv2 -> [weights of the next band's conv1] -> [next band's input into registers] -> copy-out
// and, where conv2's weights fit in registers as a whole (WD2 >= its supersteps), the next band's
// input is requested even before conv2.
// OCC: wavefronts per SIMD the register budget is sized for (= workgroups per CU when LDS allows).
// WT, RT: map width and band rows as compile-time constants (0 = taken from the arguments); the
// specialised instances turn the index arithmetic of the tile copies into shifts and immediates.
// NBC: the map is WT x WT, so the number of bands is the constant WT / RT (false: rectangular maps, a.nbands).
template <int MODE1, int CA, int CB, int C1, int C2, int EPI2, bool TADD, int NT1, int MS1, int WD1, int NT2, int MS2, int WD2, int NPA, int NPB, int OCC, int WT, int RT, int NW = 4, bool NBC = true>
__global__ __launch_bounds__(NW * 64, OCC) void fused2_kernel(FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WGS = NW * 64;   // threads per workgroup
    static_assert(MODE1 != M1_FIRST || NW == 4, "the first-layer stage deals its tiles to four wavefronts");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = tid >> 6;
    const int h = lane >> 5;
    const int W = WT ? WT : a.W, H = a.H, R = RT ? RT : a.R;
    // specialised instances: the L = 32 pyramid, whose state is 32 x 32 - or, for the first pair alone (WT = 64 / 128), the
    // square lattice of the map's own size
    constexpr int XT = (MODE1 == M1_FIRST && WT > 32) ? WT : 32;
    const int Hx = (WT && EPI2 != E2_HEAD2) ? XT : a.Hx, Wx = (WT && EPI2 != E2_HEAD2) ? XT : a.Wx;   // (the spectrum: its own size)
    (void)Hx; (void)Wx;
    constexpr int SA = CA * 2 + 16, SB = CB * 2 + 16, SM = C1 * 2 + 16, SO = C2 * 2 + 16;
    constexpr int CAT = MODE1 == M1_FIRST ? 8 : CA, CBT = MODE1 == M1_UPCAT ? CB : 8;
    (void)SA; (void)SB; (void)SO;
    // The conv2 + conv3 instance on the whole 16 x 16 map (one piece of four tiles per wavefront and convolution: every
    // weight fragment feeds four MFMAs, half the weight stream of 8-row bands).  Two workgroups fit a CU only if the
    // output staging tile lies OVER the mid tile, so conv3's epilogue waits (one more barrier) until every wavefront has
    // issued its last MFMA on the mid tile, and the mid tile's zero halo is restored for every lattice.
    constexpr bool ALIAS = MODE1 == M1_PLAIN && EPI2 == E2_STORE_POOL && CA == 32 && WT == 16 && RT == 16;
    // A workgroup owns whole lattices (blockIdx.x, blockIdx.x + gridDim.x, ...) and walks each one's bands top
    // to bottom: item n of a workgroup is band n % nbands of its (n / nbands)-th lattice.  The two bottom rows of
    // the mid tile (conv1 at global rows y0+R-1, y0+R) are what the next band needs as its rows 0 and 1, so they
    // are carried over in LDS and conv1 never computes a row twice.  Specialised instances know nbands.
    const int nbands = (WT && RT && NBC) ? WT / RT : a.nbands;
    const int nlat = (int)a.B;
    struct Item { int b, y0; };
    auto item_of = [&](int n) {
        Item it;
        const int q = n / nbands;
        it.b = (int)blockIdx.x + q * (int)gridDim.x;
        it.y0 = (n - q * nbands) * R;
        return it;
    };
#ifdef IDM_TIMELINE
    constexpr int TLK = MODE1 == M1_FIRST ? 0 : MODE1 == M1_PLAIN ? (CA == 32 ? 1 : 2) : (CA == 128 ? 3 : 4);
    int tl_it = 0;
    if (TLK == 4 && blockIdx.x < 1024 && tid == 0) {
        g_hw[blockIdx.x][0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
        g_hw[blockIdx.x][1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    }
#endif

    // LDS map: [input region | mid tile | epilogue constants | score tile (HEAD)]; the output staging
    // buffer reuses the input region.  HEAD keeps its score tile apart, so nothing but the tile copies
    // ever writes the input tiles: their halo columns are cleared once, and the band loop needs no
    // barrier between the state update and the next band's copy-in.
    char *lds_in = smem;
    char *lds_mid = smem + a.lds_in_bytes;
    char *tile_a = lds_in;
    char *tile_b = MODE1 == M1_UPCAT ? lds_in + ((((R >> 1) + 2) * ((W >> 1) + 2) * SA + 15) & ~15) : nullptr;
    // HEADX: a head follows the second convolution (E2_HEAD: one channel + the state update; E2_HEAD2: two channels, score
    // only); its score tile, [R][W] floats or (Re, Im) pairs, lies behind the constants
    constexpr bool HEADX = EPI2 == E2_HEAD || EPI2 == E2_HEAD2;
    constexpr int NCONST = 2 * C1 + 2 * C2 + 4 + (EPI2 == E2_HEAD2 ? C2 : 0);
    float *score_s = (float *)(smem + a.lds_in_bytes + a.lds_mid_bytes -
                               (EPI2 == E2_HEAD ? (R * W * 4 + 15) & ~15 : EPI2 == E2_HEAD2 ? (R * W * 8 + 15) & ~15 : 0));
    float *cst = score_s - NCONST;
    const float *c_b1 = cst, *c_t1 = cst + C1, *c_b2 = cst + 2 * C1, *c_wh = c_b2 + C2, *c_bh = c_b2 + 2 * C2, *c_wh2 = c_bh + 4;
    (void)c_t1; (void)c_wh; (void)c_bh; (void)c_wh2;
    for (int i = tid; i < C1; i += WGS) {
        cst[i] = a.b1[i];
        cst[C1 + i] = TADD ? a.tadd1[i] : 0.0f;
    }
    for (int i = tid; i < C2; i += WGS) {
        cst[2 * C1 + i] = a.b2[i];
        cst[2 * C1 + C2 + i] = EPI2 == E2_HEAD ? a.w_head[i] : EPI2 == E2_HEAD2 ? a.w_head[2 * i] : 0.0f;
        if constexpr (EPI2 == E2_HEAD2) cst[2 * C1 + 2 * C2 + 4 + i] = a.w_head[2 * i + 1];
    }
    if (tid == 0) {
        cst[2 * C1 + 2 * C2] = HEADX ? a.b_head[0] : 0.0f;
        cst[2 * C1 + 2 * C2 + 1] = EPI2 == E2_HEAD2 ? a.b_head[1] : 0.0f;
    }

    const LaneGeo geo = make_geo(W, lane & 31);
    using Conv1 = ConvStage<MODE1 == M1_FIRST ? 16 : CA, MODE1 == M1_UPCAT ? CB : 0, MODE1 == M1_UPCAT, C1, NT1, MS1, WD1, TADD>;
    using Conv2 = ConvStage<C1, 0, false, C2, NT2, MS2, WD2, HEADX>;
    constexpr bool EARLY_PF = Conv2::FULL;  // conv2 never waits on vector memory: request the next band before it
    Conv1 cs1;
    Conv2 cs2;
    (void)cs1;

    TileRegs<CAT, MODE1 == M1_FIRST ? 1 : NPA, WGS> ra;
    TileRegs<CBT, MODE1 == M1_UPCAT ? NPB : 1, WGS> rb;
    float rx[MODE1 == M1_FIRST ? NPA : 1];
    (void)ra; (void)rb; (void)rx;

    // (the thread index is made opaque per call: otherwise the per-item index arithmetic of the
    // tile copies is hoisted out of the band loop and held in registers across the convolutions)
    auto pf_load = [&](Item it) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        __builtin_assume(tid >= 0 && tid < WGS);
        const int b = it.b, y0 = it.y0;
        if constexpr (MODE1 == M1_FIRST) {
            const float *x = (const float *)a.in_a + (size_t)b * Hx * Wx;
#pragma unroll
            for (int k = 0; k < NPA; ++k) {
                const int i = tid + k * WGS, t = i / W, c = i - t * W, g = y0 - 2 + t;
                rx[k] = (t < R + 4 && g >= 0 && g < Hx && c < Wx) ? x[(size_t)g * Wx + c] : 0.0f;
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
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        __builtin_assume(tid >= 0 && tid < WGS);
        if constexpr (MODE1 == M1_FIRST) {
            uint32_t *xs = (uint32_t *)lds_in;
#pragma unroll
            for (int k = 0; k < NPA; ++k) {
                const int i = tid + k * WGS, t = i / W, c = i - t * W;
                if (t < R + 4) xs[t * (W + 2) + c + 1] = split_state(rx[k]);
            }
            for (int i = tid; i < (R + 4) * 2; i += WGS) xs[(i >> 1) * (W + 2) + (i & 1) * (W + 1)] = 0u;
        } else if constexpr (MODE1 == M1_PLAIN) {
            ra.store(tile_a, W, R + 4, tid);
        } else {
            ra.template store<!HEADX>(tile_a, W >> 1, (R >> 1) + 2, tid);
            rb.template store<!HEADX>(tile_b, W, R + 4, tid);
        }
    };

    // HEAD: the state update works on float4 quads (one Philox counter each).  The quads of a band
    // are dealt to the wavefronts with the fewest conv1 rows (all four when the split is even):
    // those draw the band's noise right after their conv1 share, while the others still compute.
    struct Quad { int64_t off; int qs; bool act, path; };
    auto quad_of = [&](Item it) {
        Quad q;
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));
        const int b = it.b, y0 = it.y0;
        const int rlo = y0 == 0 ? 1 : 2, rhi = min(R + 2, H - y0 + 1);   // as band_geo
        const int tot = Conv1::NSL * geo.ncb * ((rhi - rlo + geo.TR - 1) / geo.TR);
        int len[NW], mn = 1 << 30;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            len[w] = ((w + 1) * tot) / NW - (w * tot) / NW;
            mn = min(mn, len[w]);
        }
        int nl = 0, lr = 0, mine = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int light = len[w] == mn;
            lr += light && w < wave;
            mine |= light && w == wave;
            nl += light;
        }
        const int rowsx = min(min(R, H - y0), Hx - y0);
        const int qrow = Wx >> 2, nq = max(rowsx, 0) * qrow;
        const int qpw = nl == 4 ? (nq + 3) >> 2 : nl == 2 ? (nq + 1) >> 1 : nl == 1 ? nq : nl == 3 ? (nq + 2) / 3
                      : nl == 8 ? (nq + 7) >> 3 : nl == 6 ? (nq + 5) / 6 : (nq + nl - 1) / nl;
        q.path = (Wx & 3) == 0 && qpw <= 64;   // one quad per lane of the light wavefronts, else the generic loop
        const int qi = lr * qpw + lane;
        q.act = q.path && mine && lane < qpw && qi < nq;
        const int qr = qi / max(qrow, 1), qx = qi - qr * qrow;
        q.qs = qr * W + qx * 4;                                        // offset in the LDS score tile
        q.off = (int64_t)b * Hx * Wx + (y0 + qr) * Wx + qx * 4;  // offset in x / score / noise
        return q;
    };
    f32x4 x_next = {0.f, 0.f, 0.f, 0.f};
    auto x_fetch = [&](Item it) {
        if constexpr (EPI2 == E2_HEAD) {
            const Quad q = quad_of(it);
            if (q.act && a.do_update) x_next = *(const f32x4 *)(a.x + q.off);
        }
    };
    auto band_geo = [&](Item it, int &rlo, int &rhi) {
        rlo = it.y0 == 0 ? 1 : 2;   // mid row 0 of the top band is padding; rows 0, 1 of the others are carried over
        rhi = min(R + 2, H - it.y0 + 1);
    };
    auto conv1_begin = [&](Item it) {
        if constexpr (MODE1 != M1_FIRST) {
            int rlo, rhi;
            band_geo(it, rlo, rhi);
            cs1.begin(tile_a, tile_b, c_b1, c_t1, geo, rlo, rhi - rlo, (const bf16x8 *)a.w1, wave, lane, 0, NW);
        }
    };

    clear_halo<C1, WGS>(lds_mid, W, R + 2, 0, 0, tid);  // halo columns of the mid tile: never written afterwards
    if constexpr (HEADX && MODE1 == M1_UPCAT) {
        clear_halo<CA, WGS>(tile_a, W >> 1, (R >> 1) + 2, 0, 0, tid);
        clear_halo<CB, WGS>(tile_b, W, R + 4, 0, 0, tid);
    }
    Item cur = item_of(0);
    if (cur.b < nlat) {
        conv1_begin(cur);
        x_fetch(cur);
        pf_load(cur);
    }

    for (int n = 0; cur.b < nlat; ++n) {
        const int b = cur.b, y0 = cur.y0;
        int rlo, rhi;
        band_geo(cur, rlo, rhi);
        const f32x4 x_cur = x_next;
        (void)x_cur;

        TL(0);
        TL(9);
        pf_store();
        TL(1);
        __syncthreads();
        TL(2);
        // Rows of the mid tile outside the map are conv2's zero padding (an earlier band may have left data there).
        // Cleared after the barrier: before it, a slower wavefront may still be copying the previous band's bottom
        // rows (the carry-over), which are the rows cleared here in a lattice's last band; conv1's epilogue writes
        // only rows inside the map, and conv2 reads after the next barrier.
        if (y0 == 0) clear_halo<C1, WGS>(lds_mid, W, 0, 0, 1, tid);
        if (rhi < R + 2) clear_halo<C1, WGS>(lds_mid, W, 0, rhi, R + 2, tid);
        if constexpr (ALIAS) clear_halo<C1, WGS>(lds_mid, W, R + 2, 0, 0, tid);   // the previous lattice's staged output lay here

        // ---- conv1 on rows y0-1 .. y0+R (clipped to the map) -> mid tile row (g - (y0-1)) ----
        {
#ifdef IDM_TIMELINE
            int tl_epi = 0;
#endif
            auto epi1 = [&](const f32x16 &acc, const f32x4 *kb, const f32x4 *kt, int mt, int r, int x, bool valid) {
                if (!valid) return;
                char *dst = lds_mid + __mul24(__mul24(r, W + 2) + x + 1, SM) + (mt * 32 + 4 * h) * 2;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // Epilogues run on the plain float32 ALU (this file is built with -fno-slp-vectorize so that the
                    // compiler does not pair them up again): a v_pk_*_f32 instruction waits while another wavefront of
                    // the SIMD streams MFMAs, and with three workgroups per CU one nearly always does.  Same bits as the
                    // packed form (-DIDM_EPI_PACKED), which costs fused_A 4 %, fused_M1 5 % and fused_M2 2 %.
#ifndef IDM_EPI_PACKED
                    float z[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        z[i] = silu1(BIAS_IN_ACC ? acc[4 * g + i] : acc[4 * g + i] + kb[g][i]);
                        if constexpr (TADD) z[i] += kt[g][i];
                    }
#else
                    f32x4 z = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    z = silu4(BIAS_IN_ACC ? z : z + kb[g]);
                    if constexpr (TADD) z = z + kt[g];
#endif
                    *(bf16x4 *)(dst + 16 * g) = pack4(z[0], z[1], z[2], z[3]);
#ifdef IDM_TIMELINE
                    if (tl_epi == 0) TL(24 + g);
#endif
                }
#ifdef IDM_TIMELINE
                ++tl_epi;
#endif
            };
            if constexpr (MODE1 == M1_FIRST)
                conv_first_stage((const uint32_t *)lds_in, c_b1, W, rlo, (rhi - rlo) * W, (const bf16x8 *)a.w1, wave, lane, epi1);
            else
                cs1.run(epi1, TLHOOK(10));
        }
        float zq[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (EPI2 == E2_HEAD) {
#ifndef IDM_EXP_NOPHILOX   // timing experiment (wrong results): what the in-kernel noise draw costs
            if (a.do_update && a.add_noise && !a.noise) {
                const Quad q = quad_of(cur);
                if (q.act) {
                    const int64_t lat = (int64_t)b * Hx * Wx;
                    // (the key halves are made opaque per band: otherwise the ten round keys are kept in
                    // scalar registers across the whole loop, which spills other scalars into vector lanes)
                    uint32_t k0 = (uint32_t)a.ph.seed, k1 = (uint32_t)(a.ph.seed >> 32);
                    asm volatile("" : "+s"(k0), "+s"(k1));
                    normal4(((uint64_t)k1 << 32) | k0, a.ph.first_global_index + b, (uint32_t)((q.off - lat) >> 2), a.ph.t, zq);
                }
            }
#endif
        }
        TL(3);

        // ---- conv2 on rows y0 .. y0+R-1 (clipped) ----
        // (specialised instances: the map height is a multiple of the band height, so every band is full and conv2's
        // operand addressing is the same in every band; the HEAD kernel measures 2 % slower with that knowledge - fewer
        // registers and instructions, a worse interleaving with its co-resident workgroup - and keeps the general form)
        const int rows2 = (WT && RT && !HEADX) ? R : min(R, H - y0);
        const int npix2 = rows2 * W;
        // the last iteration reloads its own band (discarded): an unconditional load keeps the prefetch
        // registers dead across the convolutions instead of loop-carried
        const Item following = item_of(n + 1);
        const Item nxt = following.b < nlat ? following : cur;
        // rows R, R+1 of the mid tile become rows 0, 1 of the next band of this lattice (copied once conv2 is done)
        auto carry_rows = [&] {
            if (y0 + R < H) {
                const int bytes = 2 * (W + 2) * SM;   // a multiple of 16
                const char *src = lds_mid + R * (W + 2) * SM;
                for (int i = tid * 16; i < bytes; i += WGS * 16) *(f32x4 *)(lds_mid + i) = *(const f32x4 *)(src + i);
            }
        };
        cs2.begin(lds_mid, nullptr, c_b2, c_wh, geo, 0, rows2, (const bf16x8 *)a.w2, wave, lane, 0, NW);
        if constexpr (EARLY_PF) {
            x_fetch(nxt);
            pf_load(nxt);
        }
        __syncthreads();
        TL(4);
        if constexpr (EPI2 == E2_HEAD) {
            auto epi2 = [&](const f32x16 &acc, const f32x4 *kb, const f32x4 *kw, int mt, int r, int x, bool valid) {
                (void)mt;
#ifndef IDM_EPI_PACKED
                float p4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i) p4[i] += silu1(BIAS_IN_ACC ? acc[4 * g + i] : acc[4 * g + i] + kb[g][i]) * kw[g][i];
#else
                f32x4 p4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 z = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    p4 += silu4(BIAS_IN_ACC ? z : z + kb[g]) * kw[g];
                }
#endif
                float part = (p4[0] + p4[1]) + (p4[2] + p4[3]);
                part += __shfl_xor(part, 32, 64);
                if (valid && h == 0) score_s[r * W + x] = part + c_bh[0];
            };
            cs2.run(epi2, TLHOOK(18));
            TL(5);
            conv1_begin(nxt);
            if constexpr (!EARLY_PF) {
                x_fetch(nxt);
                pf_load(nxt);
            }
            __syncthreads();
            TL(6);
            carry_rows();

            const Quad q = quad_of(cur);
            if (q.path) {
                if (q.act) {
                    f32x4 sc = *(const f32x4 *)(score_s + q.qs);
                    if (a.score_out) *(f32x4 *)(a.score_out + q.off) = sc;
                    if (a.do_update) {
                        float z[4] = {zq[0], zq[1], zq[2], zq[3]};
                        if (a.add_noise && a.noise) {
                            const f32x4 zv = *(const f32x4 *)(a.noise + q.off);
#pragma unroll
                            for (int j = 0; j < 4; ++j) z[j] = zv[j];
                        }
                        f32x4 xv = x_cur;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            xv[j] = reverse_update(a.c1, xv[j], a.beta, sc[j], z[j], a.sd, a.add_noise != 0);
                        }
                        *(f32x4 *)(a.x + q.off) = xv;
                    }
                }
            } else {
                const int rowsx = min(rows2, Hx - y0);
                float *xg = a.x + (size_t)b * Hx * Wx;
                const size_t lat_off = (size_t)b * Hx * Wx;
                for (int i = tid; i < rowsx * Wx; i += WGS) {
                    const int r = i / Wx, xc = i - r * Wx;
                    const int e = (y0 + r) * Wx + xc;
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
                    xg[e] = reverse_update(a.c1, xg[e], a.beta, sc, zz, a.sd, a.add_noise != 0);
                }
            }
        } else if constexpr (EPI2 == E2_HEAD2) {
            // the two-channel 1x1 head of the spectral model: the score leaves as (Re, Im) pairs, cropped to the unpadded
            // spectrum; the state update stays a kernel of its own
            static_assert(C2 == 32 && MS2 == 1, "one cout slice");
            auto epi2 = [&](const f32x16 &acc, const f32x4 *kb, const f32x4 *kw, int mt, int r, int x, bool valid) {
                (void)mt;
                float pa[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 k2 = *(const f32x4 *)(c_wh2 + 8 * g + 4 * h);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float z = silu1(BIAS_IN_ACC ? acc[4 * g + i] : acc[4 * g + i] + kb[g][i]);
                        pa[i] += z * kw[g][i];
                        pb[i] += z * k2[i];
                    }
                }
                float sa = (pa[0] + pa[1]) + (pa[2] + pa[3]), sb = (pb[0] + pb[1]) + (pb[2] + pb[3]);
                sa += __shfl_xor(sa, 32, 64);
                sb += __shfl_xor(sb, 32, 64);
                if (valid && h == 0) *(f32x2 *)(score_s + (r * W + x) * 2) = f32x2{sa + c_bh[0], sb + c_bh[1]};
            };
            cs2.run(epi2, TLHOOK(18));
            conv1_begin(nxt);
            if constexpr (!EARLY_PF) pf_load(nxt);
            __syncthreads();
            carry_rows();
            const int rowsx = min(rows2, Hx - y0), colsx = min(W, Wx);
            if (rowsx > 0 && colsx > 0 && !a.transpose_out) {
                float *sg = a.score_out + (((int64_t)b * Hx + y0) * Wx) * 2;
                for (int i = tid; i < rowsx * colsx; i += WGS) {
                    const int r = i / colsx, xc = i - r * colsx;
                    *(f32x2 *)(sg + ((int64_t)r * Wx + xc) * 2) = *(const f32x2 *)(score_s + (r * W + xc) * 2);
                }
            } else if (rowsx > 0 && colsx > 0) {
                // the caller's tensor is (B, Wx, Hx, 2): element (row y0 + r, column xc) of this map is its (xc, y0 + r);
                // consecutive threads walk r, which is contiguous there
                float *sg = a.score_out + ((int64_t)b * Wx * Hx + y0) * 2;
                for (int i = tid; i < rowsx * colsx; i += WGS) {
                    const int xc = i / rowsx, r = i - xc * rowsx;
                    *(f32x2 *)(sg + ((int64_t)xc * Hx + r) * 2) = *(const f32x2 *)(score_s + (r * W + xc) * 2);
                }
            }
        } else {
            char *stage = ALIAS ? lds_mid : lds_in;  // [rows2][W] positions of SO bytes
            auto epi2 = [&](const f32x16 &acc, const f32x4 *kb, const f32x4 *, int mt, int r, int x, bool valid) {
                if (!valid) return;
                char *dst = stage + __mul24(__mul24(r, W) + x, SO) + (mt * 32 + 4 * h) * 2;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#ifndef IDM_EPI_PACKED
                    float z[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) z[i] = silu1(BIAS_IN_ACC ? acc[4 * g + i] : acc[4 * g + i] + kb[g][i]);
#else
                    f32x4 z = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    z = silu4(BIAS_IN_ACC ? z : z + kb[g]);
#endif
                    *(bf16x4 *)(dst + 16 * g) = pack4(z[0], z[1], z[2], z[3]);
                }
            };
            // (ALIAS: every wavefront runs exactly one piece here, so the barrier between its MFMA loop and its epilogue
            // is reached once by all: no wavefront writes the staging tile while another still reads the mid tile under it)
            if constexpr (ALIAS) cs2.run(epi2, [&](int i) { if (i == 1) __syncthreads(); });
            else cs2.run(epi2, TLHOOK(18));
            TL(5);
            conv1_begin(nxt);
            if constexpr (!EARLY_PF) pf_load(nxt);
            __syncthreads();
            TL(6);
            carry_rows();
            constexpr int CH = C2 / 8;
            __bf16 *og = (__bf16 *)a.out + ((size_t)b * H + y0) * W * C2;
            for (int i = tid; i < npix2 * CH; i += WGS) {
                const int ch = i % CH, pos = i / CH;
                *(bf16x8 *)(og + (size_t)pos * C2 + ch * 8) = *(const bf16x8 *)(stage + pos * SO + ch * 16);
            }
            if constexpr (EPI2 == E2_STORE_POOL) {
                const int Wh = W >> 1, rh = rows2 >> 1;
                __bf16 *pg = (__bf16 *)a.out_pool + ((size_t)b * (H >> 1) + (y0 >> 1)) * Wh * C2;
                for (int i = tid; i < rh * Wh * CH; i += WGS) {
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
        TL(7);
        if constexpr (!HEADX) __syncthreads();  // staging consumed before the next band's input overwrites the region
        TL(8);
        cur = following;
#ifdef IDM_TIMELINE
        ++tl_it;
#endif
    }
}

// ---------------------------------------------------------------------------------------------
// Host side: LDS sizing and launch.
// ---------------------------------------------------------------------------------------------
namespace {

// npa / npb: 16-byte items (FK_A: floats) per thread the kernel instance can hold in registers
// for the next band's input tiles; nw: wavefronts per workgroup.
// IDM_B_WAVES = 8 builds fused_B with eight wavefronts and 16-row bands (one workgroup per CU): correct, but
// 10 % slower than two four-wavefront workgroups per CU, whose barriers do not idle the whole CU.
#ifndef IDM_B_WAVES
#define IDM_B_WAVES 4
#endif
#ifndef IDM_M3_WAVES
#define IDM_M3_WAVES 4
#endif
struct KindInfo { int mode1, ca, cb, c1, c2, epi2, npa, npb, nw; };
constexpr KindInfo kKinds[FK_COUNT] = {
    {M1_FIRST, 1, 0, 32, 32, E2_STORE_POOL, 3, 0, 4},    // FK_A : x -> conv0 -> conv1 -> skip1, pooled
    {M1_PLAIN, 32, 0, 64, 64, E2_STORE_POOL, 5, 0, 4},   // FK_M1: pooled -> conv2 -> conv3 -> skip2, pooled
    {M1_PLAIN, 64, 0, 128, 128, E2_STORE, 3, 0, 4},      // FK_M2: pooled -> conv4 (+time) -> conv5
    {M1_UPCAT, 128, 64, 64, 64, E2_STORE, IDM_M3_WAVES == 8 ? 3 : 5, IDM_M3_WAVES == 8 ? 5 : 10, IDM_M3_WAVES},     // FK_M3: [up(conv5), skip2] -> conv6 -> conv7 (whole 16x16 map)
#if IDM_B_WAVES == 8
    {M1_UPCAT, 64, 32, 32, 32, E2_HEAD, 3, 5, 8},     // FK_B : [up(conv7), skip1] -> conv8 -> conv9 -> head -> update (16-row bands)
#elif IDM_B_WAVES == 2
    {M1_UPCAT, 64, 32, 32, 32, E2_HEAD, 6, 12, 2},    // FK_B with two wavefronts (four rows each) and 8-row bands
#else
    {M1_UPCAT, 64, 32, 32, 32, E2_HEAD, 3, 6, 4},     // FK_B with four wavefronts and 8-row bands
#endif
    {M1_UPCAT, 64, 32, 32, 32, E2_HEAD2, 3, 6, 4},    // FK_BH2: [up(conv7), skip1] -> conv8 -> conv9 -> two-channel head
};

inline int align16(int v) { return (v + 15) & ~15; }

}  // namespace

// True when the next band's input tiles fit the kernel instance's prefetch registers.
bool fused_band_fits(int kind, int W, int R)
{
    const KindInfo &k = kKinds[kind];
    const int wg = k.nw * 64;
    if (k.mode1 == M1_FIRST) return (R + 4) * W <= k.npa * wg;
    if (k.mode1 == M1_PLAIN) return (R + 4) * W * (k.ca / 8) <= k.npa * wg;
    return (R / 2 + 2) * (W / 2) * (k.ca / 8) <= k.npa * wg && (R + 4) * W * (k.cb / 8) <= k.npb * wg;
}

// Band height of the kinds that run one workgroup per CU, when it fits (0: use the two-per-CU rule).
// FK_M3 (conv6 + conv7: widest K, 25 % halo recompute at 8 rows) takes the whole map; FK_B with eight
// wavefronts takes 16 rows (halo 2 of 18 instead of 2 of 10, half the per-band bookkeeping).
int fused_tall_rows(int kind, int H, int W)
{
    int R = 0;
    if (kind == FK_M3) R = H;
    else if (kind == FK_B && kKinds[kind].nw == 8) R = H < 16 ? H : 16;
    if (R <= 0 || (R & 1)) return 0;
    int in = 0, mid = 0;
    fused_lds_bytes(kind, W, R, &in, &mid);
    return in + mid <= 159 * 1024 && fused_band_fits(kind, W, R) ? R : 0;
}

void fused_lds_bytes(int kind, int W, int R, int *in_bytes, int *mid_bytes)
{
    const KindInfo &k = kKinds[kind];
    const int sa = k.ca * 2 + 16, sb = k.cb * 2 + 16, sm = k.c1 * 2 + 16, so = k.c2 * 2 + 16;
    int in;
    if (k.mode1 == M1_FIRST) in = (R + 4) * (W + 2) * 4;
    else if (k.mode1 == M1_PLAIN) in = (R + 4) * (W + 2) * sa;
    else in = align16((R / 2 + 2) * (W / 2 + 2) * sa) + (R + 4) * (W + 2) * sb;
    const bool headx = k.epi2 == E2_HEAD || k.epi2 == E2_HEAD2;
    const int stage = headx ? 0 : R * W * so;
    *in_bytes = align16(in > stage ? in : stage);
    // mid tile, then the epilogue constants (2*C1 + 2*C2 + 4 floats, C2 more for the second head channel), then the heads'
    // score tile
    *mid_bytes = align16((R + 2) * (W + 2) * sm) + (2 * k.c1 + 2 * k.c2 + 4 + (k.epi2 == E2_HEAD2 ? k.c2 : 0)) * 4 +
                 (k.epi2 == E2_HEAD ? align16(R * W * 4) : k.epi2 == E2_HEAD2 ? align16(R * W * 8) : 0);
}

// conv2 + conv3 of the 32 x 32 lattice run on the whole 16 x 16 map (IDM_M1_WHOLE=0 at build time: 8-row bands).
#ifndef IDM_M1_WHOLE
#define IDM_M1_WHOLE 1
#endif
int fused_m1_whole_rows(int Hx, int Wx)
{
    return IDM_M1_WHOLE && Hx == 32 && Wx == 32 ? 16 : 0;
}

// Launch state per kernel instance (all instances share one function-pointer type, so the
// state cannot live in a function-local static of the launcher template).
struct LaunchState {
    bool configured = false;
    int cus = 0, cached_lds = -1, cached_blocks = 0;
};

template <class K>
static hipError_t launch_one(LaunchState &st, K kernel, const FusedArgs &a, int64_t B, hipStream_t s, int threads = WG)
{
#ifdef IDM_EXP_ONE_WG   // timing experiment: LDS request padded so that one workgroup fits a CU
    const int lds = std::max(a.lds_in_bytes + a.lds_mid_bytes, 84 * 1024);
#else
    const int lds = a.lds_in_bytes + a.lds_mid_bytes;
#endif
    if (!st.configured) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t prop;
        if ((e = hipGetDevice(&dev)) != hipSuccess || (e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
        st.cus = prop.multiProcessorCount;
        st.configured = true;
    }
    if (st.cached_lds != lds) {
        int n = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)kernel, threads, (size_t)lds);
        if (e != hipSuccess) return e;
        st.cached_blocks = n > 0 ? n : 1;
        st.cached_lds = lds;
    }
    const int64_t total = B * a.nbands;
    if (total >= ((int64_t)1 << 31)) return hipErrorInvalidValue;   // the kernels index work items in 32 bits
    const int64_t grid = std::min<int64_t>(B, (int64_t)st.cus * st.cached_blocks);   // a workgroup owns whole lattices
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(threads), lds, s, a);
    return hipGetLastError();
}

// Tunables per kernel: NT1, MS1, WD1, NT2, MS2, WD2, NPA, NPB, OCC (overridable at build time for tuning runs).
#ifndef IDM_CFG_A
#define IDM_CFG_A 1, 1, 3, 2, 1, 6, 3, 0, 3
#endif
#ifndef IDM_CFG_M1
#define IDM_CFG_M1 3, 1, 1, 2, 1, 1, 5, 0, 2
#endif
#ifndef IDM_CFG_M1W   // the whole-map instance: one four-tile piece per wavefront and convolution
#define IDM_CFG_M1W 4, 1, 3, 4, 1, 3, 5, 0, 2
#endif
#ifndef IDM_CFG_M2
#define IDM_CFG_M2 2, 1, 2, 2, 1, 2, 3, 0, 2
#endif
#ifndef IDM_CFG_M3
#define IDM_CFG_M3 4, 1, 2, 4, 1, 3, 5, 10, 1
#endif
#ifndef IDM_CFG_B
#if IDM_B_WAVES == 8
#define IDM_CFG_B 3, 1, 2, 2, 1, 3, 3, 5, 1
#else
#define IDM_CFG_B 3, 1, 2, 2, 1, 3, 3, 6, 2
#endif
#endif

hipError_t launch_fused(int kind, const FusedArgs &a, int64_t B, hipStream_t s)
{
    static thread_local LaunchState all_st[IDM_MAX_DEVICES][18];   // one slot per kernel FUNCTION (attributes and occupancy are per function)
    const int slot = idm_device_slot();
    if (slot < 0) return hipErrorInvalidDevice;
    LaunchState *st = all_st[slot];
    // (ISINGDIFF_FUSED_SHAPED=0, read once per process, keeps the run-time-shaped instances: identical results)
    static const bool shaped = [] { const char *e = getenv("ISINGDIFF_FUSED_SHAPED"); return !(e && e[0] == '0'); }();
    if (kind == FK_BH2) {
        // (the transposed spectral map at L = 32: 32 columns, 8-row bands)
        if (shaped && a.W == 32 && a.R == 8)
            return launch_one(st[15], fused2_kernel<M1_UPCAT, 64, 32, 32, 32, E2_HEAD2, false, IDM_CFG_B, 32, 8, IDM_B_WAVES, false>, a, B, s, IDM_B_WAVES * 64);
        return launch_one(st[11], fused2_kernel<M1_UPCAT, 64, 32, 32, 32, E2_HEAD2, false, IDM_CFG_B, 0, 0, IDM_B_WAVES>, a, B, s, IDM_B_WAVES * 64);
    }
    // the second level of that map (10 rows x 16 columns, one band): conv2 + conv3 and conv6 + conv7
    if (shaped && a.W == 16 && a.H == 10 && a.R == 10) {
        if (kind == FK_M1)
            return launch_one(st[16], fused2_kernel<M1_PLAIN, 32, 0, 64, 64, E2_STORE_POOL, false, IDM_CFG_M1, 16, 10, 4, false>, a, B, s);
        if (kind == FK_M3)
            return launch_one(st[17], fused2_kernel<M1_UPCAT, 128, 64, 64, 64, E2_STORE, false, IDM_CFG_M3, 16, 10, IDM_M3_WAVES, false>, a, B, s, IDM_M3_WAVES * 64);
    }
    // conv0 + conv1 of the L = 64 / 128 lattices in the bands the plan picks for them (4 / 2 rows): the same kernel with the
    // map size known at compile time (the run-time-shaped instance spends 20 vector instructions per MFMA, 40 % more than
    // the L = 32 one, on index arithmetic: it is bound by them)
    if (shaped && kind == FK_A && a.H == a.W && a.Hx == a.W && a.Wx == a.W) {
        if (a.W == 64 && a.R == 4)
            return launch_one(st[12], fused2_kernel<M1_FIRST, 1, 0, 32, 32, E2_STORE_POOL, false, IDM_CFG_A, 64, 4>, a, B, s);
        if (a.W == 128 && a.R == 2)
            return launch_one(st[13], fused2_kernel<M1_FIRST, 1, 0, 32, 32, E2_STORE_POOL, false, IDM_CFG_A, 128, 2>, a, B, s);
    }
    // ... and conv2 + conv3 of the L = 64 lattice (32 x 32 maps in 4-row bands) likewise
    if (shaped && kind == FK_M1 && a.W == 32 && a.H == 32 && a.R == 4)
        return launch_one(st[14], fused2_kernel<M1_PLAIN, 32, 0, 64, 64, E2_STORE_POOL, false, IDM_CFG_M1, 32, 4>, a, B, s);
    // instances specialised for the L = 32 pyramid (widths 32 / 16 / 8; 8-row bands, M3 the whole 16-row map; no padding)
    const int w32[5] = {32, 16, 8, 16, 32}, r32[5] = {8, 8, 8, 16, IDM_B_WAVES == 8 ? 16 : 8};
    const bool spec = a.W == w32[kind] && a.R == r32[kind] && a.Hx == 32 && a.Wx == 32;
    if (kind == FK_M1 && a.W == 16 && a.H == 16 && a.R == 16 && fused_m1_whole_rows(a.Hx, a.Wx) == 16) {
        // the 16x16x32-MFMA kernel of idm_fused_split.hip (same partition, the other MFMA shape's summation order; 9 %
        // faster, whole job +1.3 %) when the model carries that packing, unless ISINGDIFF_M1_MFMA16=0 asks for the
        // 32x32x16 instance below (read once per process)
        static const bool mf16 = [] { const char *e = getenv("ISINGDIFF_M1_MFMA16"); return !(e && e[0] == '0'); }();
        if (mf16 && a.w1_alt && a.w2_alt) return launch_m1_mf16(a, B, s);
        // whole-map instance: the staging tile lies over the mid tile, so the input region holds the input tile only
        FusedArgs b = a;
        b.lds_in_bytes = align16((16 + 4) * (16 + 2) * (32 * 2 + 16));
        return launch_one(st[10], fused2_kernel<M1_PLAIN, 32, 0, 64, 64, E2_STORE_POOL, false, IDM_CFG_M1W, 16, 16>, b, B, s);
    }
    if (spec) {
        switch (kind) {
        case FK_A: return launch_one(st[5], fused2_kernel<M1_FIRST, 1, 0, 32, 32, E2_STORE_POOL, false, IDM_CFG_A, 32, 8>, a, B, s);
        case FK_M1: return launch_one(st[6], fused2_kernel<M1_PLAIN, 32, 0, 64, 64, E2_STORE_POOL, false, IDM_CFG_M1, 16, 8>, a, B, s);
        case FK_M2: {
            // the lattice-pair kernel on the 16x16x32 MFMA (idm_fused_split.hip: half the weight stream, 18 % faster, whole
            // job +2.1 %) when the model carries that packing, unless ISINGDIFF_M2_MFMA16=0 asks for the one-lattice
            // 32x32x16 instance below (read once per process)
            static const bool mf16 = [] { const char *e = getenv("ISINGDIFF_M2_MFMA16"); return !(e && e[0] == '0'); }();
            if (mf16 && a.w1_alt && a.w2_alt && B < ((int64_t)1 << 19)) return launch_m2_pair(a, B, s);
            return launch_one(st[7], fused2_kernel<M1_PLAIN, 64, 0, 128, 128, E2_STORE, true, IDM_CFG_M2, 8, 8>, a, B, s);
        }
        case FK_M3: {
            // the schedule of idm_fused_split.hip (tile copies on wavefronts of their own; the same results bit for bit, 7 %
            // faster) unless ISINGDIFF_M3_SPLIT=0 asks for the four-wavefront kernel
            static const bool split = [] { const char *e = getenv("ISINGDIFF_M3_SPLIT"); return !(e && e[0] == '0'); }();
            if (split) return launch_m3_split(a, B, s);
            return launch_one(st[8], fused2_kernel<M1_UPCAT, 128, 64, 64, 64, E2_STORE, false, IDM_CFG_M3, 16, 16, IDM_M3_WAVES>, a, B, s, IDM_M3_WAVES * 64);
        }
        default: {
            // the schedule of idm_fused_split.hip (one workgroup per CU, two lattices in flight, tile copies / noise / update on
            // wavefronts of their own; the same results bit for bit, 3 % faster) unless ISINGDIFF_B_SPLIT=0 asks for the
            // two independent four-wavefront workgroups per CU
            static const bool split = [] { const char *e = getenv("ISINGDIFF_B_SPLIT"); return !(e && e[0] == '0'); }();
            if (split) return launch_b_split(a, B, s);
            return launch_one(st[9], fused2_kernel<M1_UPCAT, 64, 32, 32, 32, E2_HEAD, false, IDM_CFG_B, 32, IDM_B_WAVES == 8 ? 16 : 8, IDM_B_WAVES>, a, B, s, IDM_B_WAVES * 64);
        }
        }
    }
    if (kind == FK_M2 && a.W == 8 && a.H == 5 && a.w1_alt && a.w2_alt && B < ((int64_t)1 << 19)) {
        // the third level of the transposed spectral model (5 x 8 maps): the lattice-pair kernel as well
        static const bool mf16 = [] { const char *e = getenv("ISINGDIFF_M2_MFMA16"); return !(e && e[0] == '0'); }();
        if (mf16) return launch_m2_pair(a, B, s);
    }
    switch (kind) {
    case FK_A: return launch_one(st[0], fused2_kernel<M1_FIRST, 1, 0, 32, 32, E2_STORE_POOL, false, IDM_CFG_A, 0, 0>, a, B, s);
    case FK_M1: return launch_one(st[1], fused2_kernel<M1_PLAIN, 32, 0, 64, 64, E2_STORE_POOL, false, IDM_CFG_M1, 0, 0>, a, B, s);
    case FK_M2: return launch_one(st[2], fused2_kernel<M1_PLAIN, 64, 0, 128, 128, E2_STORE, true, IDM_CFG_M2, 0, 0>, a, B, s);
    case FK_M3: return launch_one(st[3], fused2_kernel<M1_UPCAT, 128, 64, 64, 64, E2_STORE, false, IDM_CFG_M3, 0, 0, IDM_M3_WAVES>, a, B, s, IDM_M3_WAVES * 64);
    default: return launch_one(st[4], fused2_kernel<M1_UPCAT, 64, 32, 32, 32, E2_HEAD, false, IDM_CFG_B, 0, 0, IDM_B_WAVES>, a, B, s, IDM_B_WAVES * 64);
    }
}
