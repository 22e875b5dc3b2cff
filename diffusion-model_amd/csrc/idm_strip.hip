// Single-convolution band/strip kernels (bf16 MFMA, float32 accumulate) for maps too large for the
// fused two-convolution bands: one workgroup owns R rows x WS columns of one lattice's output,
// stages the (R+2) x (WS+2) input window (zero outside the map) in LDS and runs one sliding-window
// convolution stage on it.  No halo is recomputed (only re-read), every MFMA lane is a real
// pixel, and the LDS footprint does not grow with the lattice size.  The UNet's data movement is
// folded in exactly as in the fused kernels: UP reads Concatenate([UpSampling2D(a), skip]) by an
// index shift on the half-resolution window, POOL also writes the 2x2 average, HEAD applies the
// 1x1 head conv and the reverse-diffusion update (Philox noise drawn in-kernel).
#include "idm_conv_stage.h"

#include <algorithm>

namespace {

enum { SE_STORE = 0, SE_STORE_POOL = 1, SE_HEAD = 2, SE_HEAD2 = 3 };   // SE_HEAD2: 1x1 head with two output channels, score only

// Register-held payload of one LDS input window: NP 16-byte items per thread (item i = tid + k*WG:
// row i / (ncols*C/8), then column, then 8-channel group).  load() reads rows [g0, g0+nrows) x
// columns [c0, c0+ncols) of a (Hmap, Wmap, C) bf16 map, zeros outside it (the buffer range check
// returns zero for the out-of-range offset given to such items); store() writes the window.
template <int C, int NP>
struct WinRegs {
    static constexpr int S = C * 2 + 16, CH = C / 8;
    bf16x8 v[NP];

    DEVI void load(const __bf16 *__restrict__ src, int Hmap, int Wmap, int g0, int nrows, int c0, int ncols, int tid)
    {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(src, (unsigned)(Hmap * Wmap * C * 2));
        const int ipr = ncols * CH;
        int row = tid / ipr, rem = tid - row * ipr;
        const int drow = WG / ipr, drem = WG - drow * ipr;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int col = rem / CH, ch = rem - col * CH;
            const int g = g0 + row, c = c0 + col;
            const bool ok = row < nrows && g >= 0 && g < Hmap && c >= 0 && c < Wmap;
            v[k] = buf_load16(rs, ok ? (unsigned)(((g * Wmap + c) * CH + ch) * 16) : 0x80000000u, 0);
            rem += drem;
            row += drow;
            if (rem >= ipr) { rem -= ipr; ++row; }
        }
    }

    DEVI void store(char *tile, int nrows, int ncols, int tid) const
    {
        const int ipr = ncols * CH;
        int row = tid / ipr, rem = tid - row * ipr;
        const int drow = WG / ipr, drem = WG - drow * ipr;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int col = rem / CH, ch = rem - col * CH;
            if (row < nrows) *(bf16x8 *)(tile + (row * ncols + col) * S + ch * 16) = v[k];
            rem += drem;
            row += drow;
            if (rem >= ipr) { rem -= ipr; ++row; }
        }
    }
};

}  // namespace

// Persistent: each workgroup walks work items wg = blockIdx.x, blockIdx.x + gridDim.x, ...;
// item -> (lattice, band, strip) with the strip index fastest, so neighbouring workgroups share
// their halo columns in L2.  Ordering of vector-memory traffic as in the fused kernels: the next
// item's first weights are requested before its input, and both before the copy-out.
// WST, RT: strip width and band rows as compile-time constants (0 = taken from the arguments).
template <int CA, int CB, bool UP, int COUT, int EPI, bool TADD, int NT, int MS, int WD, int NPA, int NPB, int OCC, int WST, int RT,
          bool ALIAS = false>
__global__ __launch_bounds__(WG, OCC) void strip_kernel(StripArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5;
    const int W = a.W, H = a.H, R = RT ? RT : a.R, WS = WST ? WST : a.WS;
    constexpr int SA = CA * 2 + 16, SB = CB * 2 + 16, SO = COUT * 2 + 16;
    (void)SB; (void)SO;
    // work items are indexed in 32 bits (checked on the host)
    const int per_lat = a.nbands * a.nstrips, total = (int)a.B * per_lat;

    // ALIAS: instances whose band is twice as tall as its LDS would otherwise allow (the host's table of tall instances):
    // every wavefront runs ONE four-tile piece per work item, so a weight fragment feeds four MFMAs instead of two and
    // fewer halo rows are re-read.  The output staging tile lies OVER the input windows; the epilogue therefore waits at a
    // barrier until every wavefront has issued its last MFMA (each runs exactly one piece, so all reach that barrier
    // once per item).
    static_assert(!ALIAS || ((WST == 32 || WST == 16) && RT > 0 && NT == 4 && (EPI == SE_STORE || EPI == SE_STORE_POOL)), "aliased staging: store kernels with one piece per wavefront");
    // LDS map: [window a | window b | output staging | epilogue constants]
    char *tile_a = smem;
    char *tile_b = UP ? smem + ((((R >> 1) + 2) * ((WS >> 1) + 2) * SA + 15) & ~15) : nullptr;
    char *stage = ALIAS ? smem : smem + a.lds_in_bytes;
    // constants: bias | second per-channel array (head weights / time add) | head bias (4 floats) | second head vector
    float *cst = (float *)(smem + a.lds_in_bytes + a.lds_stage_bytes);
    const float *c_b = cst, *c_2 = cst + COUT, *c_bh = cst + 2 * COUT, *c_3 = cst + 2 * COUT + 4;
    (void)c_bh; (void)c_3;
    for (int i = tid; i < COUT; i += WG) {
        cst[i] = a.bias[i];
        cst[COUT + i] = EPI == SE_HEAD ? a.w_head[i] : EPI == SE_HEAD2 ? a.w_head[2 * i] : (TADD ? a.tadd[i] : 0.0f);
        cst[2 * COUT + 4 + i] = EPI == SE_HEAD2 ? a.w_head[2 * i + 1] : 0.0f;
    }
    if (tid == 0) {
        cst[2 * COUT] = EPI == SE_HEAD || EPI == SE_HEAD2 ? a.b_head[0] : 0.0f;
        cst[2 * COUT + 1] = EPI == SE_HEAD2 ? a.b_head[1] : 0.0f;
    }

    // Layers with few input channels per output value spend most of their time in the epilogue: those run it on the
    // plain float32 ALU (measured per layer at L = 64 / 128: conv2, conv3, conv7, conv9 4-6 % faster; conv4, conv5,
    // conv6, conv8 3-4 % slower, so they keep the packed form).
    constexpr bool PLAIN_EPI = CA + CB <= 64 && COUT <= 64;
    const LaneGeo geo = make_geo(WS, lane & 31);
    using Conv = ConvStage<CA, CB, UP, COUT, NT, MS, WD, TADD || EPI == SE_HEAD || EPI == SE_HEAD2>;
    Conv cs;

    WinRegs<CA, NPA> ra;
    WinRegs<UP ? CB : 8, UP ? NPB : 1> rb;
    (void)rb;

    auto item = [&](int wg, int &b, int &y0, int &x0) {
        b = wg / per_lat;
        const int r = wg - b * per_lat;
        const int band = r / a.nstrips;
        y0 = band * R;
        x0 = (r - band * a.nstrips) * WS;
    };
    // (the thread index is made opaque per call so that the per-item index arithmetic of the window
    // copies is recomputed per work item instead of being held in registers across the convolution)
    auto pf_load = [&](int wg) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        int b, y0, x0;
        item(wg, b, y0, x0);
        if constexpr (!UP) {
            ra.load((const __bf16 *)a.in_a + (size_t)b * H * W * CA, H, W, y0 - 1, R + 2, x0 - 1, WS + 2, tid);
        } else {
            const int Hh = H >> 1, Wh = W >> 1;
            ra.load((const __bf16 *)a.in_a + (size_t)b * Hh * Wh * CA, Hh, Wh, (y0 >> 1) - 1, (R >> 1) + 2, (x0 >> 1) - 1,
                    (WS >> 1) + 2, tid);
            rb.load((const __bf16 *)a.in_b + (size_t)b * H * W * CB, H, W, y0 - 1, R + 2, x0 - 1, WS + 2, tid);
        }
    };
    auto pf_store = [&]() {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if constexpr (!UP) {
            ra.store(tile_a, R + 2, WS + 2, tid);
        } else {
            ra.store(tile_a, (R >> 1) + 2, (WS >> 1) + 2, tid);
            rb.store(tile_b, R + 2, WS + 2, tid);
        }
    };

    // HEAD: float4 quads of the state (one Philox counter each), dealt evenly to the four wavefronts
    struct Quad { int64_t off; int qs; bool act; };
    auto quad_of = [&](int wg) {
        Quad q;
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));
        int b, y0, x0;
        item(wg, b, y0, x0);
        const int rowsx = min(min(R, H - y0), a.Hx - y0);
        const int colsx = min(WS, a.Wx - x0);
        const int qrow = max(colsx, 0) >> 2, nq = max(rowsx, 0) * qrow, qpw = (nq + 3) >> 2;
        const int qi = wave * qpw + lane;
        q.act = lane < qpw && qi < nq;
        const int qr = qi / max(qrow, 1), qx = qi - qr * qrow;
        q.qs = qr * WS + qx * 4;
        q.off = (int64_t)b * a.Hx * a.Wx + (int64_t)(y0 + qr) * a.Wx + x0 + qx * 4;
        return q;
    };
    f32x4 x_next = {0.f, 0.f, 0.f, 0.f};
    auto x_fetch = [&](int wg) {
        if constexpr (EPI == SE_HEAD) {
            const Quad q = quad_of(wg);
            if (q.act && a.do_update) x_next = *(const f32x4 *)(a.x + q.off);
        }
    };
    auto conv_begin = [&](int wg) {
        int b, y0, x0;
        item(wg, b, y0, x0);
        cs.begin(tile_a, tile_b, c_b, c_2, geo, 0, min(R, H - y0), (const bf16x8 *)a.w, wave, lane, 1);
    };

    int wg = blockIdx.x;
    if (wg < total) {
        conv_begin(wg);
        x_fetch(wg);
        pf_load(wg);
    }
    for (; wg < total; wg += gridDim.x) {
        int b, y0, x0;
        item(wg, b, y0, x0);
        const int rows = min(R, H - y0);
        const f32x4 x_cur = x_next;
        (void)x_cur;
        const int wg_next = wg + (int)gridDim.x < total ? wg + (int)gridDim.x : wg;

        pf_store();
        __syncthreads();

        if constexpr (EPI == SE_HEAD) {
            float *score_s = (float *)stage;
            auto epi = [&](const f32x16 &acc, const f32x4 *kb, const f32x4 *kw, int mt, int r, int x, bool valid) {
                (void)mt;
                f32x4 p4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 z = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    p4 += silu4t<PLAIN_EPI>(BIAS_IN_ACC ? z : z + kb[g]) * kw[g];
                }
                float part = (p4[0] + p4[1]) + (p4[2] + p4[3]);
                part += __shfl_xor(part, 32, 64);
                if (valid && h == 0) score_s[r * WS + x] = part + c_bh[0];
            };
            cs.run(epi);
            float zq[4] = {0.f, 0.f, 0.f, 0.f};
            const Quad q = quad_of(wg);
            if (a.do_update && a.add_noise && !a.noise && q.act)
                normal4(a.ph.seed, a.ph.first_global_index + b, (uint32_t)((q.off - (int64_t)b * a.Hx * a.Wx) >> 2), a.ph.t, zq);
            conv_begin(wg_next);
            x_fetch(wg_next);
            pf_load(wg_next);
            __syncthreads();
            if (q.act) {
                const f32x4 sc = *(const f32x4 *)(score_s + q.qs);
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
        } else if constexpr (EPI == SE_HEAD2) {
            // conv9 + the two-channel 1x1 head of the spectral model: the score leaves as (Re, Im) pairs, cropped to the
            // unpadded spectrum; the state update stays a kernel of its own
            float *score_s = (float *)stage;   // [rows][WS][2]
            auto epi = [&](const f32x16 &acc, const f32x4 *kb, const f32x4 *kw, int mt, int r, int x, bool valid) {
                (void)mt;
                f32x4 pa = {0.f, 0.f, 0.f, 0.f}, pb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 z = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    z = silu4t<PLAIN_EPI>(BIAS_IN_ACC ? z : z + kb[g]);
                    pa += z * kw[g];
                    pb += z * *(const f32x4 *)(c_3 + 8 * g + 4 * h);
                }
                float sa = (pa[0] + pa[1]) + (pa[2] + pa[3]), sb = (pb[0] + pb[1]) + (pb[2] + pb[3]);
                sa += __shfl_xor(sa, 32, 64);
                sb += __shfl_xor(sb, 32, 64);
                if (valid && h == 0) *(f32x2 *)(score_s + (r * WS + x) * 2) = f32x2{sa + c_bh[0], sb + c_bh[1]};
            };
            cs.run(epi);
            conv_begin(wg_next);
            pf_load(wg_next);
            __syncthreads();
            const int rowsx = min(rows, a.Hx - y0), colsx = min(WS, a.Wx - x0);
            if (rowsx > 0 && colsx > 0 && !a.transpose_out) {
                float *sg = a.score_out + (((int64_t)b * a.Hx + y0) * a.Wx + x0) * 2;
                for (int i = tid; i < rowsx * colsx; i += WG) {
                    const int r = i / colsx, xc = i - r * colsx;
                    *(f32x2 *)(sg + ((int64_t)r * a.Wx + xc) * 2) = *(const f32x2 *)(score_s + (r * WS + xc) * 2);
                }
            } else if (rowsx > 0 && colsx > 0) {
                // the caller's tensor is (B, Wx, Hx, 2): element (row y0 + r, column x0 + xc) of this map is its
                // (x0 + xc, y0 + r); consecutive threads walk r, which is contiguous there
                float *sg = a.score_out + (((int64_t)b * a.Wx + x0) * a.Hx + y0) * 2;
                for (int i = tid; i < rowsx * colsx; i += WG) {
                    const int xc = i / rowsx, r = i - xc * rowsx;
                    *(f32x2 *)(sg + ((int64_t)xc * a.Hx + r) * 2) = *(const f32x2 *)(score_s + (r * WS + xc) * 2);
                }
            }
        } else {
            auto epi = [&](const f32x16 &acc, const f32x4 *kb, const f32x4 *kt, int mt, int r, int x, bool valid) {
                if (!valid) return;
                char *dst = stage + (r * WS + x) * SO + (mt * 32 + 4 * h) * 2;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 z = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    z = silu4t<PLAIN_EPI>(BIAS_IN_ACC ? z : z + kb[g]);
                    if constexpr (TADD) z = z + kt[g];
                    *(bf16x4 *)(dst + 16 * g) = pack4(z[0], z[1], z[2], z[3]);
                }
            };
            if constexpr (ALIAS) cs.run(epi, [&](int i) { if (i == 1) __syncthreads(); });
            else cs.run(epi);
            conv_begin(wg_next);
            pf_load(wg_next);
            __syncthreads();
            constexpr int CH = COUT / 8;
            __bf16 *og = (__bf16 *)a.out + (((size_t)b * H + y0) * W + x0) * COUT;
            for (int i = tid; i < rows * WS * CH; i += WG) {
                const int ch = i % CH, pos = i / CH;
                const int r = pos / WS, xx = pos - r * WS;
                *(bf16x8 *)(og + ((size_t)r * W + xx) * COUT + ch * 8) = *(const bf16x8 *)(stage + pos * SO + ch * 16);
            }
            if constexpr (EPI == SE_STORE_POOL) {
                const int Wh = W >> 1, wsh = WS >> 1, rh = rows >> 1;
                __bf16 *pg = (__bf16 *)a.out_pool + (((size_t)b * (H >> 1) + (y0 >> 1)) * Wh + (x0 >> 1)) * COUT;
                for (int i = tid; i < rh * wsh * CH; i += WG) {
                    const int ch = i % CH, pos = i / CH;
                    const int xh = pos % wsh, yh = pos / wsh;
                    const char *q = stage + ((2 * yh) * WS + 2 * xh) * SO + ch * 16;
                    const bf16x8 t0 = *(const bf16x8 *)q, t1 = *(const bf16x8 *)(q + SO);
                    const bf16x8 t2 = *(const bf16x8 *)(q + WS * SO), t3 = *(const bf16x8 *)(q + WS * SO + SO);
                    bf16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        o[j] = (__bf16)((((float)t0[j] + (float)t1[j]) + (float)t2[j] + (float)t3[j]) * 0.25f);
                    *(bf16x8 *)(pg + ((size_t)yh * Wh + xh) * COUT + ch * 8) = o;
                }
            }
        }
        __syncthreads();  // windows and staging consumed before the next item overwrites them
    }
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
namespace {

struct StripKind { int ca, cb, up, cout, epi, npa, npb; };
// npa / npb: 16-byte items per thread the instance can hold for the next item's windows
constexpr StripKind kStrip[SK_COUNT] = {
    {32, 0, 0, 32, SE_STORE_POOL, 6, 0},   // SK_CONV1
    {32, 0, 0, 64, SE_STORE, 6, 0},        // SK_CONV2
    {64, 0, 0, 64, SE_STORE_POOL, 11, 0},  // SK_CONV3
    {64, 0, 0, 128, SE_STORE, 11, 0},      // SK_CONV4 (+ time add)
    {128, 0, 0, 128, SE_STORE, 11, 0},     // SK_CONV5
    {128, 64, 1, 64, SE_STORE, 7, 11},     // SK_CONV6
    {64, 0, 0, 64, SE_STORE, 11, 0},       // SK_CONV7
    {64, 32, 1, 32, SE_STORE, 4, 6},       // SK_CONV8
    {32, 0, 0, 32, SE_HEAD, 6, 0},         // SK_CONV9 (+ head, update)
    {32, 0, 0, 32, SE_STORE, 6, 0},        // SK_CONV9S
    {32, 0, 0, 32, SE_HEAD2, 6, 0},        // SK_CONV9H2 (+ two-channel head)
};

inline int a16(int v) { return (v + 15) & ~15; }

// Tall instances for 32-column strips: a band of `rows` rows in which every wavefront runs one four-tile piece (16 work
// units), with more prefetch registers (npa / npb) and, where the taller windows leave no room for it, the staging tile
// over the input windows (alias).
struct TallKind { int rows; bool alias; int npa, npb; };
constexpr TallKind kTall[SK_COUNT] = {
    {0, false, 0, 0},      // SK_CONV1
    {0, false, 0, 0},      // SK_CONV2 (8 rows are 16 units already)
    {8, true, 11, 0},      // SK_CONV3
    {0, false, 0, 0},      // SK_CONV4 (4 rows are 16 units already)
    {4, true, 13, 0},      // SK_CONV5
    {8, true, 7, 11},      // SK_CONV6 (77 KB of windows: two workgroups per CU only without a staging tile of its own)
    {8, true, 11, 0},      // SK_CONV7
    {16, true, 6, 10},     // SK_CONV8
    {16, false, 10, 0},    // SK_CONV9 (its staging tile is the small score tile)
    {0, false, 0, 0},      // SK_CONV9S
    {0, false, 0, 0},      // SK_CONV9H2
};
// Strips narrower than 32 columns (the spectral model's 20-column top level): 16-row bands fit without aliasing; the
// instances take the strip width at run time and hold more prefetch registers.
constexpr TallKind kTallNarrow[SK_COUNT] = {
    {16, false, 9, 0},     // SK_CONV1
    {0, false, 0, 0}, {0, false, 0, 0}, {0, false, 0, 0}, {0, false, 0, 0}, {0, false, 0, 0}, {0, false, 0, 0},
    {16, false, 6, 9},     // SK_CONV8
    {0, false, 0, 0},      // SK_CONV9
    {0, false, 0, 0},      // SK_CONV9S
    {16, false, 9, 0},     // SK_CONV9H2
};
// 16-column strips (two row slots per tile) of conv5: 8-row bands are 16 work units; staging over the window.
constexpr TallKind kTall16Conv5 = {8, true, 12, 0};
inline const TallKind &tall_of(int kind, int WS)
{
    return WS == 32 ? kTall[kind] : WS == 16 && kind == SK_CONV5 ? kTall16Conv5 : kTallNarrow[kind];
}
// 10-row bands of 32-column strips (the spectral model's top level processed transposed: 20 rows x 32 columns) run on
// the instances of the narrow strips, which take band height and strip width at run time and hold enough prefetch registers.
inline bool is_ten_row(int kind, int R, int WS)
{
    return WS == 32 && R == 10 && (kind == SK_CONV1 || kind == SK_CONV8 || kind == SK_CONV9H2);
}
inline bool is_tall(int kind, int R, int WS)
{
    if (is_ten_row(kind, R, WS)) return true;
    return (WS >= 17 || (WS == 16 && kind == SK_CONV5)) && R > 0 && R == tall_of(kind, WS).rows;
}
inline bool is_alias(int kind, int R, int WS) { return !is_ten_row(kind, R, WS) && is_tall(kind, R, WS) && tall_of(kind, WS).alias; }

struct StripLaunch {
    bool configured = false;
    int cus = 0, cached_lds = -1, cached_blocks = 0;
};

template <class K>
hipError_t launch_strip_one(StripLaunch &st, K kernel, const StripArgs &a, hipStream_t s)
{
    const int lds = a.lds_in_bytes + a.lds_stage_bytes + a.lds_const_bytes;
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
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)kernel, WG, (size_t)lds);
        if (e != hipSuccess) return e;
        st.cached_blocks = n > 0 ? n : 1;
        st.cached_lds = lds;
    }
    const int64_t total = a.B * a.nbands * a.nstrips;
    if (total >= ((int64_t)1 << 31)) return hipErrorInvalidValue;
    const int64_t grid = std::min<int64_t>(total, (int64_t)st.cus * st.cached_blocks);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(WG), lds, s, a);
    return hipGetLastError();
}

}  // namespace

// LDS bytes of one instance for band height R and strip width WS.
void strip_lds_bytes(int kind, int R, int WS, int *in_bytes, int *stage_bytes, int *const_bytes)
{
    const StripKind &k = kStrip[kind];
    const int sa = k.ca * 2 + 16, sb = k.cb * 2 + 16, so = k.cout * 2 + 16;
    int in;
    if (!k.up) in = (R + 2) * (WS + 2) * sa;
    else in = a16((R / 2 + 2) * (WS / 2 + 2) * sa) + (R + 2) * (WS + 2) * sb;
    *in_bytes = a16(in);
    *stage_bytes = a16(k.epi == SE_HEAD ? R * WS * 4 : k.epi == SE_HEAD2 ? R * WS * 8 : R * WS * so);
    *const_bytes = (3 * k.cout + 4) * 4;
    if (is_alias(kind, R, WS)) {   // the staging tile shares the windows' bytes
        *in_bytes = *in_bytes > *stage_bytes ? *in_bytes : *stage_bytes;
        *stage_bytes = 0;
    }
}

// True when the next item's windows fit the instance's prefetch registers.
bool strip_fits(int kind, int R, int WS)
{
    const StripKind &k = kStrip[kind];
    int npa = k.npa, npb = k.npb;
    if (is_ten_row(kind, R, WS)) {
        npa = kTallNarrow[kind].npa;
        npb = kTallNarrow[kind].npb;
    } else if (is_tall(kind, R, WS)) {
        npa = tall_of(kind, WS).npa;
        npb = tall_of(kind, WS).npb;
    }
    if (!k.up) return (R + 2) * (WS + 2) * (k.ca / 8) <= npa * WG;
    return (R / 2 + 2) * (WS / 2 + 2) * (k.ca / 8) <= npa * WG && (R + 2) * (WS + 2) * (k.cb / 8) <= npb * WG;
}

#define STRIP_INST(idx, R8, ...)                                                                          \
    (a.WS == 32 && a.R == R8 ? launch_strip_one(st[SK_COUNT + idx], strip_kernel<__VA_ARGS__, 32, R8>, a, s)   \
                             : launch_strip_one(st[idx], strip_kernel<__VA_ARGS__, 0, 0>, a, s))

// Weight ring depth of the strip kernels (supersteps fetched ahead).
#ifndef IDM_STRIP_WD
#define IDM_STRIP_WD 2
#endif

hipError_t launch_strip(int kind, const StripArgs &a, hipStream_t s)
{
    static thread_local StripLaunch all_st[IDM_MAX_DEVICES][2 * SK_COUNT + 10];
    const int slot = idm_device_slot();
    if (slot < 0) return hipErrorInvalidDevice;
    StripLaunch *st = all_st[slot];
    if (is_tall(kind, a.R, a.WS) && (a.WS < 32 || is_ten_row(kind, a.R, a.WS))) {
        constexpr int T = 2 * SK_COUNT + 6;
        switch (kind) {
        case SK_CONV5: return launch_strip_one(st[T + 3], strip_kernel<128, 0, false, 128, SE_STORE, false, 4, 1, IDM_STRIP_WD, kTall16Conv5.npa, 0, 2, 16, 8, true>, a, s);
        case SK_CONV1: return launch_strip_one(st[T + 0], strip_kernel<32, 0, false, 32, SE_STORE_POOL, false, 4, 1, IDM_STRIP_WD, kTallNarrow[SK_CONV1].npa, 0, 2, 0, 0>, a, s);
        case SK_CONV8: return launch_strip_one(st[T + 1], strip_kernel<64, 32, true, 32, SE_STORE, false, 4, 1, IDM_STRIP_WD, kTallNarrow[SK_CONV8].npa, kTallNarrow[SK_CONV8].npb, 2, 0, 0>, a, s);
        default: return launch_strip_one(st[T + 2], strip_kernel<32, 0, false, 32, SE_HEAD2, false, 4, 1, IDM_STRIP_WD, kTallNarrow[SK_CONV9H2].npa, 0, 2, 0, 0>, a, s);
        }
    }
    if (is_tall(kind, a.R, a.WS)) {
        constexpr int T = 2 * SK_COUNT;
        switch (kind) {
        case SK_CONV3: return launch_strip_one(st[T + 0], strip_kernel<64, 0, false, 64, SE_STORE_POOL, false, 4, 1, IDM_STRIP_WD, kTall[SK_CONV3].npa, 0, 2, 32, 8, true>, a, s);
        case SK_CONV5: return launch_strip_one(st[T + 1], strip_kernel<128, 0, false, 128, SE_STORE, false, 4, 1, IDM_STRIP_WD, kTall[SK_CONV5].npa, 0, 2, 32, 4, true>, a, s);
        case SK_CONV6: return launch_strip_one(st[T + 5], strip_kernel<128, 64, true, 64, SE_STORE, false, 4, 1, IDM_STRIP_WD, kTall[SK_CONV6].npa, kTall[SK_CONV6].npb, 2, 32, 8, true>, a, s);
        case SK_CONV7: return launch_strip_one(st[T + 2], strip_kernel<64, 0, false, 64, SE_STORE, false, 4, 1, IDM_STRIP_WD, kTall[SK_CONV7].npa, 0, 2, 32, 8, true>, a, s);
        case SK_CONV8: return launch_strip_one(st[T + 3], strip_kernel<64, 32, true, 32, SE_STORE, false, 4, 1, IDM_STRIP_WD, kTall[SK_CONV8].npa, kTall[SK_CONV8].npb, 2, 32, 16, true>, a, s);
        default: return launch_strip_one(st[T + 4], strip_kernel<32, 0, false, 32, SE_HEAD, false, 4, 1, IDM_STRIP_WD, kTall[SK_CONV9].npa, 0, 2, 32, 16>, a, s);
        }
    }
    // second argument: the band height make_plan picks for 32-column strips (specialised instance)
    switch (kind) {
    case SK_CONV1: return STRIP_INST(0, 8, 32, 0, false, 32, SE_STORE_POOL, false, 3, 1, IDM_STRIP_WD, 6, 0, 2);
    case SK_CONV2: return STRIP_INST(1, 8, 32, 0, false, 64, SE_STORE, false, 4, 1, IDM_STRIP_WD, 6, 0, 2);   // (8 rows x 2 cout slices: one four-tile piece per wavefront)
    case SK_CONV3: return STRIP_INST(2, 4, 64, 0, false, 64, SE_STORE_POOL, false, 3, 1, IDM_STRIP_WD, 11, 0, 2);
    case SK_CONV4: return STRIP_INST(3, 4, 64, 0, false, 128, SE_STORE, true, 4, 1, IDM_STRIP_WD, 11, 0, 2);   // (4 rows x 4 cout slices: likewise)
    case SK_CONV5: return STRIP_INST(4, 2, 128, 0, false, 128, SE_STORE, false, 3, 1, IDM_STRIP_WD, 11, 0, 2);
    case SK_CONV6: return STRIP_INST(5, 4, 128, 64, true, 64, SE_STORE, false, 3, 1, IDM_STRIP_WD, 7, 11, 2);
    case SK_CONV7: return STRIP_INST(6, 4, 64, 0, false, 64, SE_STORE, false, 3, 1, IDM_STRIP_WD, 11, 0, 2);
    case SK_CONV8: return STRIP_INST(7, 8, 64, 32, true, 32, SE_STORE, false, 3, 1, IDM_STRIP_WD, 4, 6, 2);
    case SK_CONV9: return STRIP_INST(8, 8, 32, 0, false, 32, SE_HEAD, false, 3, 1, IDM_STRIP_WD, 6, 0, 2);
    case SK_CONV9H2: return STRIP_INST(10, 8, 32, 0, false, 32, SE_HEAD2, false, 3, 1, IDM_STRIP_WD, 6, 0, 2);
    default: return STRIP_INST(9, 8, 32, 0, false, 32, SE_STORE, false, 3, 1, IDM_STRIP_WD, 6, 0, 2);
    }
}
