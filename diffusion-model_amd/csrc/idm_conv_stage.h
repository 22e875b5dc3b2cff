// Device building blocks shared by the bf16 MFMA convolution kernels: packed silu, buffer
// addressing, MFMA tile geometry and the sliding-window convolution stage.
#pragma once
#include "idm_internal.h"
#include "idm_device.h"

#include <type_traits>

// Timing-experiment switches (IDM_EXP_*: builds that skip or fake part of the work to price it; most compute WRONG
// results) compile only together with -DIDM_EXPERIMENT, which idm_api.hip must be built with too: idm_model_create
// then refuses to create handles unless ISINGDIFF_ALLOW_EXPERIMENT=1 is set, so such a library cannot pass for the product.
#if (defined(IDM_EXP_NOSILU) || defined(IDM_EXP_FOLDSILU) || defined(IDM_SILU_SCALAR) || defined(IDM_EXP_HALFW) || \
     defined(IDM_EXP_NOMFMA) || defined(IDM_EXP_MFMA16) || defined(IDM_EXP_NOPHILOX) || defined(IDM_EXP_ONE_WG) ||    \
     defined(IDM_EXP_PRIO_MFMA) || defined(IDM_EXP_PRIO_EPI) || defined(IDM_EXP_C7_RAW) || defined(IDM_EXP_RIDE8_RAW)) && \
    !defined(IDM_EXPERIMENT)
#error "IDM_EXP_* / IDM_SILU_SCALAR are timing experiments: build the whole library with -DIDM_EXPERIMENT as well"
#endif

namespace {

constexpr int WG = 256;

// The convolution bias is the initial value of the accumulators (the first MFMA of a tile takes it as its C operand),
// so the epilogues do not add it.  IDM_NO_BIAS_IN_ACC restores the add in the epilogue (comparison builds).
#ifdef IDM_NO_BIAS_IN_ACC
constexpr bool BIAS_IN_ACC = false;
#else
constexpr bool BIAS_IN_ACC = true;
#endif

#ifndef IDM_FRAG_DEPTH
#define IDM_FRAG_DEPTH 1
#endif

enum { M1_PLAIN = 0, M1_UPCAT = 1, M1_FIRST = 2 };
enum { E2_STORE = 0, E2_STORE_POOL = 1, E2_HEAD = 2, E2_HEAD2 = 3 };   // E2_HEAD2: 1x1 head with two output channels, score only

typedef __attribute__((ext_vector_type(2))) float f32x2;

// silu of four values, written on float pairs so that everything but the two transcendentals
// (v_exp_f32, v_rcp_f32: quarter rate) goes through the packed float32 ALU.
DEVI f32x4 silu4(f32x4 z)
{
#ifdef IDM_EXP_NOSILU   // timing experiment (wrong results): what the activation costs
    return z;
#endif
#ifdef IDM_SILU_SCALAR  // one float at a time (build with -fno-slp-vectorize, or the compiler packs it again)
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float e = __builtin_amdgcn_exp2f(z[i] * -1.4426950408889634f);
        o[i] = z[i] * __builtin_amdgcn_rcpf(e + 1.0f);
    }
    return o;
#endif
    const f32x2 nl = {-1.4426950408889634f, -1.4426950408889634f}, one = {1.0f, 1.0f};
    f32x2 z0 = {z[0], z[1]}, z1 = {z[2], z[3]};
    f32x2 t0 = z0 * nl, t1 = z1 * nl;
    f32x2 e0 = {__builtin_amdgcn_exp2f(t0[0]), __builtin_amdgcn_exp2f(t0[1])};
    f32x2 e1 = {__builtin_amdgcn_exp2f(t1[0]), __builtin_amdgcn_exp2f(t1[1])};
    e0 = e0 + one;
    e1 = e1 + one;
    f32x2 r0 = {__builtin_amdgcn_rcpf(e0[0]), __builtin_amdgcn_rcpf(e0[1])};
    f32x2 r1 = {__builtin_amdgcn_rcpf(e1[0]), __builtin_amdgcn_rcpf(e1[1])};
    z0 = z0 * r0;
    z1 = z1 * r1;
    return f32x4{z0[0], z0[1], z1[0], z1[1]};
}

// silu of one value on the plain (unpacked) float32 ALU.
DEVI float silu1(float z)
{
#ifdef IDM_EXP_NOSILU
    return z;
#endif
#ifdef IDM_EXP_FOLDSILU   // timing experiment (wrong results): the activation without its input scaling
    const float e = __builtin_amdgcn_exp2f(z);
#else
    const float e = __builtin_amdgcn_exp2f(z * -1.4426950408889634f);
#endif
    return z * __builtin_amdgcn_rcpf(e + 1.0f);
}

// silu of four values, one float at a time when PLAIN (translation units built with -fno-slp-vectorize, or the compiler
// pairs the operations up again): a v_pk_*_f32 instruction waits while another wavefront of the SIMD streams MFMAs, which
// costs epilogue-heavy layers more than the packed form's fewer instructions save.  Both forms give the same bits.
template <bool PLAIN>
DEVI f32x4 silu4t(f32x4 z)
{
    if constexpr (PLAIN) return f32x4{silu1(z[0]), silu1(z[1]), silu1(z[2]), silu1(z[3])};
    else return silu4(z);
}

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

// Buffer addressing: one scalar resource per tensor, a 32-bit per-lane byte offset and a scalar
// constant per access (no 64-bit per-lane address arithmetic); reads beyond `bytes` return zero.
DEVI __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}

DEVI bf16x8 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return __builtin_bit_cast(bf16x8, v);
}

// Per-lane geometry of the 32-pixel MFMA tiles of a map of width W, fixed for a kernel launch.
// W <= 32: a tile is TR = 32 / W row slots of W pixels, and slot s of tile j is row j + s * Rh of the
// stage's output region (Rh = ceil(rows / TR)), so that vertically adjacent tiles j, j+1, ... touch
// input rows that differ by exactly one.  Which lane holds which pixel of the tile is free (the MFMA
// only sees 32 columns), and it decides the LDS bank conflicts of the operand reads: a ds_read_b128 is
// served in the 16-lane groups {0-3,12-15,20-27} and {4-11,16-19,28-31} (per half wave), i.e. by the
// parity of li >> 2, and a group is conflict-free when it reads 16 different columns of ONE row (the
// pixel stride is an odd number of 16-byte units).  W = 16 therefore puts one slot in each group.
// W = 8 puts slots {0,2} in one group and {1,3} in the other: two 8-column rows share a group without
// conflicts when they lie 128 bytes (mod 256) apart, which 2 * Rh row pitches are for Rh = 2, the
// whole-map 8x8 case.  Other widths: slot li / W, column li % W.
// W > 32: a tile is 32 consecutive columns of one row (ncb column blocks).
struct LaneGeo {
    int W, ncb, TR, slot, col;
    bool ok;
};

DEVI LaneGeo make_geo(int W, int li)
{
    LaneGeo g;
    g.W = W;
    g.ncb = (W + 31) >> 5;
    if (g.ncb == 1) {
        g.TR = 32 / W;
        const int par = ((li >> 2) ^ (li >> 3) ^ (li >> 4)) & 1;
        if (W == 16) {
            g.slot = par;
            g.col = ((li >> 1) & 12) | (li & 3);
        } else if (W == 8) {
            g.slot = par + 2 * (li >> 4);
            g.col = ((li >> 1) & 4) | (li & 3);
        } else {
            g.slot = li / W;
            g.col = li - g.slot * W;
        }
        g.ok = g.slot < g.TR;
        if (!g.ok) g.slot = 0;
    } else {
        g.TR = 1;
        g.slot = 0;
        g.col = li;
        g.ok = true;
    }
    return g;
}

// One 3x3 convolution stage over `nrows` output rows of an LDS-resident input tile.
//
// Work is cut into (cout slice, column block, row j) items and split evenly over the workgroup's
// wavefronts (four unless begin() is told otherwise); a wavefront processes up to NT vertically adjacent tiles at once.  Because tile n's
// operand for tap row ky is input row j + n + ky, the NT + 2 activation fragments of one
// (kx, channel group) "superstep" feed all 3 * NT MFMAs of that superstep: LDS traffic drops from
// one 1 KB read per MFMA to (NT + 2) / (3 * NT * MS).  Supersteps are fully unrolled (static
// register indices); activation fragments are read one superstep ahead, weights WD supersteps
// ahead (L2 latency) into a register ring.  begin() fixes the partition and issues the first WD
// weight fetches, so the caller can place it ahead of a barrier or a long-latency phase; run()
// does the rest.
//   PLAIN part: operand at  tile + ((row + ky)*(W+2) + col + kx)*S + 32*cg
//   UP part:    the same on the half-resolution tile at row (row+ky)>>1, column (col+kx+1)>>1
// Epilogue constants: c0 (bias) and, when HAS_C1, a second per-cout array c1 live in LDS as float32.  The 16 bias values a
// lane needs per cout tile are read before the first superstep and start the accumulators; the 16 of c1 are read during the
// last superstep.  Both are handed to epi(), whose accumulator already includes the bias.
// EXACT: every tile of a piece lies inside the stage's rows (32-pixel-wide maps: one row per tile, pieces never longer than
// the rows left), so the fragment rows need no clamping and are one base address plus compile-time steps.
template <int CA, int CB, bool UP, int COUT, int NT, int MS, int WD, bool HAS_C1, bool EXACT = false>
struct ConvStage {
    static constexpr int MTOT = COUT / 32, NSL = MTOT / MS;
    static constexpr int SA = CA * 2 + 16, SB = CB * 2 + 16;
    static constexpr int NCG_A = CA / 16, NCG_B = CB / 16, NCG = NCG_A + NCG_B;
    static constexpr int NS = 3 * NCG;
    static constexpr int NPRE = WD < NS ? WD : NS;       // supersteps of weights fetched by begin()
    static constexpr int WR = WD < NS ? WD + 1 : NS;     // ring slots
    static constexpr bool FULL = WD >= NS;               // every weight of the stage is fetched by begin()

    const char *lds_a, *lds_b;
    const float *c0, *c1;
    __amdgpu_buffer_rsrc_t wrsrc;   // weights: every fetch is buffer(base) + scalar constant + one 32-bit lane offset
    unsigned wlane, woff;
    LaneGeo g;
    int h, r0, nrows, Rh, f, f1, up_rb;
    int dual_off = 0;   // piece<.., DUAL = true>: bytes from the tiles this stage reads to the second lattice's tiles
    bf16x8 wreg[WR][3][MS];
    // A stage whose weights fit the ring as a whole (FULL) and that has a single cout slice needs them fetched only
    // once per kernel: they stay in registers from band to band (IDM_STATIONARY_WEIGHTS, on by default).
    static constexpr bool STATIONARY = FULL && NSL == 1
#ifdef IDM_NO_STATIONARY_WEIGHTS
                                       && false
#endif
        ;
    bool loaded = false;

    DEVI void set_slice(int sl)
    {
        // The slice offset is made opaque to the optimiser: otherwise the weight loads are
        // invariant across work items and get hoisted into (far too many) registers.  Laundering
        // the offset, not the pointer, keeps the loads global_load: a laundered pointer degrades
        // to flat_load, which also counts on lgkmcnt and turns every LDS wait into a wait for
        // the weight prefetch.
        unsigned o = (wlane + (unsigned)(sl * MS * 64)) * 16u;  // byte offset
        asm volatile("" : "+v"(o));
        woff = o;
    }

    template <int S>
    DEVI void issue_w()
    {
        constexpr int kx = S / NCG, cg = S % NCG;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int m = 0; m < MS; ++m) {
                // the low 12 bits of the constant ride in the instruction's immediate offset, so the
                // fetches of a stage share a few scalar 4 KB bases instead of one scalar constant each
                const unsigned k = (((ky * 3 + kx) * NCG + cg) * MTOT + m) * 1024;
#ifdef IDM_EXP_HALFW   // timing experiment (wrong results): odd supersteps reuse the previous superstep's weight fragments
                if constexpr (S & 1) { wreg[S % WR][ky][m] = wreg[(S - 1) % WR][ky][m]; continue; }
#endif
                wreg[S % WR][ky][m] = buf_load16(wrsrc, woff + (k & 0xFFFu), k & ~0xFFFu);
            }
    }

    DEVI void prologue()
    {
        static_for<0, NPRE>([&](auto S) { issue_w<decltype(S)::value>(); });
    }

    // up_row_bias: parity of the global row of window row 0 (UP part): the half-resolution window row of
    // full-resolution window row r is (r + up_row_bias) >> 1.
    // Single-slice stages fetch the same first weights in every run: arm() requests them ahead of time (the ring is free
    // as soon as a run's last MFMA has been issued), and the begin() that follows is told so.  set_weights() fixes the
    // weight tensor once, in uniform control flow: were the buffer resource reassigned inside the band loop's branches,
    // the compiler would keep it in vector registers and wrap every fetch in a readfirstlane loop.
    DEVI void set_weights(const bf16x8 *w, int lane)
    {
        wrsrc = make_rsrc(w, 9u * (CA + CB) * COUT * 2u);
        wlane = (unsigned)lane;
    }
    DEVI void arm()
    {
        static_assert(NSL == 1, "arm() is for stages with a single cout slice");
        set_slice(0);
        prologue();
    }

    DEVI void begin(const char *la, const char *lb, const float *c0_, const float *c1_, const LaneGeo &geo, int r0_,
                    int nrows_, const bf16x8 *w, int wave, int lane, int up_row_bias = 0, int nwaves = 4, bool armed = false)
    {
        up_rb = up_row_bias;
        lds_a = la; lds_b = lb; c0 = c0_; c1 = c1_; g = geo; h = lane >> 5; r0 = r0_; nrows = nrows_;
        if (!armed) set_weights(w, lane);   // (armed: set_weights() was called once, see there)
        Rh = (nrows + g.TR - 1) / g.TR;
        const int tot = NSL * g.ncb * Rh;
        f = (wave * tot) / nwaves;
        f1 = ((wave + 1) * tot) / nwaves;
        if constexpr (STATIONARY) {
            if (!loaded) {   // every wavefront, whether or not it has work in this band
                set_slice(0);
                prologue();
                loaded = true;
            }
        } else if (f < f1 && !armed) {
            set_slice(unit_of(f) % NSL);
            prologue();
        }
    }

    // f / Rh.  With one column block there are only NSL units, and a few compares replace the
    // division (a float-reciprocal sequence of ~20 vector instructions on this target).
    DEVI int unit_of(int ff) const
    {
        if (g.ncb != 1) return ff / Rh;
        int u = 0;
#pragma unroll
        for (int k = 1; k < NSL; ++k) u += ff >= k * Rh;
        return u;
    }

    struct NoHook { DEVI void operator()(int) const {} };
    struct NoInter { template <class S> DEVI void operator()(S) const {} };

    // hook(0): operands addressed, before the first fragment read; hook(1): after the last MFMA; hook(2): after the
    // epilogue; hook(3 + i): after the epilogue of the piece's i-th tile, i < 3 (timing builds; a no-op otherwise).
    // inter(integral_constant<int, st>) is called once in every superstep, in the same scheduling region as that
    // superstep's MFMAs: work that does not depend on this piece (another piece's epilogue) issues in their shadow.
    // An epi that takes a leading integral_constant<int, tile> is told which tile of the piece it is handed.
    // UPPAR: parity of the piece's first window row in the half-resolution tile's row numbering (row + up_rb), when the
    // caller knows it (0 / 1; -1 = unknown).  Two full-resolution rows share one half-resolution row, so of the N + 2
    // fragments of an upsampled channel group only every other one differs: with the parity known those are read once
    // and reused (N = 2: two or three LDS reads per superstep instead of four).
    // DUAL: the piece is computed for TWO lattices at once - the second one's tiles lie dual_off bytes behind the first's
    // and hold the same rows - with the same weight fragments (each then feeds 2 N MFMAs); epi is handed the second
    // lattice's tiles as tiles N .. 2 N - 1 (an epi that takes the tile index tells them apart).
    // Index of the fragment that holds fragment i of an upsampled channel group when the parity UPPAR of the piece's first
    // window row is known: two full-resolution rows that halve to the same row share one fragment (fragment i repeats
    // fragment i - 1 when i + UPPAR is odd).
    template <int UPPAR>
    static constexpr int up_src(int i)
    {
#ifdef IDM_NO_UP_DEDUPE
        return i;
#else
        return (UPPAR >= 0 && i > 0 && ((i + UPPAR) & 1)) ? i - 1 : i;
#endif
    }

    template <int N, int UPPAR = -1, bool DUAL = false, class Epi, class Hook, class Inter = NoInter>
    DEVI void piece(int j, int sl, int cb, Epi &&epi, Hook &&hook, Inter &&inter = Inter{})
    {
        const int W = g.W, Wp2 = W + 2, Wh2 = (W >> 1) + 2;
        const int cc = cb * 32 + g.col;
        const int ca = min(cc, W - 1);
        const int rbase = r0 + j + __mul24(g.slot, Rh), rmax = r0 + nrows + 1;
        // Fragment i of the piece reads input row min(rbase + i, rmax).  One 24-bit multiply per tile
        // gives the first row's address; the others follow by constant steps, clamped to the last row's.
        int a_pl[N + 2], a_up[N + 2], a_b[N + 2], c_up[3];
        if constexpr (UP) {
            const int PU = Wh2 * SA, t = rbase + up_rb;
            const int first = __mul24(t >> 1, PU), last = __mul24((rmax + up_rb) >> 1, PU);
            const int odd = (t & 1) ? PU : 0;   // half-resolution row of fragment i: (t >> 1) + ((t & 1) + i >> 1)
#pragma unroll
            for (int i = 0; i < N + 2; ++i) {
                const int v = first + (i >> 1) * PU + ((i & 1) ? odd : 0);
                a_up[i] = EXACT ? v : min(v, last);
            }
        } else {
            const int PA = Wp2 * SA, col = ca * SA + h * 16;
            const int first = __mul24(rbase, PA) + col, last = __mul24(rmax, PA) + col;
#pragma unroll
            for (int i = 0; i < N + 2; ++i) a_pl[i] = EXACT ? first + i * PA : min(first + i * PA, last);
        }
        if constexpr (CB > 0) {
            const int PB = Wp2 * SB, col = ca * SB + h * 16;
            const int first = __mul24(rbase, PB) + col, last = __mul24(rmax, PB) + col;
#pragma unroll
            for (int i = 0; i < N + 2; ++i) a_b[i] = EXACT ? first + i * PB : min(first + i * PB, last);
        }
        if constexpr (UP) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) c_up[kx] = ((ca + kx + 1) >> 1) * SA + h * 16;
        }
        (void)a_pl; (void)a_up; (void)a_b; (void)c_up;

        f32x4 k0[MS][4], k1[MS][4];
        if constexpr (BIAS_IN_ACC) {
#pragma unroll
            for (int m = 0; m < MS; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) k0[m][q] = *(const f32x4 *)(c0 + (sl * MS + m) * 32 + 8 * q + 4 * h);
        }
        constexpr int NY = DUAL ? N : 1;
        f32x16 acc[MS][N], accy[MS][NY];
#pragma unroll
        for (int m = 0; m < MS; ++m)
#pragma unroll
            for (int n = 0; n < N; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[m][n][r] = BIAS_IN_ACC ? k0[m][r >> 2][r & 3] : 0.0f;
                    if (DUAL) accy[m][n][r] = acc[m][n][r];
                }
        (void)accy;

        // activation fragments are read FD supersteps ahead of their MFMAs (LDS latency under load exceeds the MFMA time
        // of a superstep with few tiles)
        constexpr int FD = IDM_FRAG_DEPTH < NS ? IDM_FRAG_DEPTH : 1, FR = FD + 1;
        bf16x8 fr[FR][N + 2], fy[DUAL ? FR : 1][N + 2];
        (void)fy;
        auto read_f = [&](auto S) {
            constexpr int st = decltype(S)::value;
            constexpr int kx = st / NCG, cg = st % NCG;
#pragma unroll
            for (int i = 0; i < N + 2; ++i) {
                if constexpr (cg < NCG_A) {
#ifndef IDM_NO_UP_DEDUPE   // (comparison builds: read every fragment)
                    if constexpr (UP && UPPAR >= 0) {
                        if (up_src<UPPAR>(i) != i) continue;   // fragment i repeats fragment i - 1: the MFMAs read that one
                    }
#endif
                    int addr;
                    if constexpr (UP) addr = a_up[i] + c_up[kx];
                    else addr = a_pl[i] + kx * SA;
                    fr[st % FR][i] = *(const bf16x8 *)(lds_a + addr + cg * 32);
                    if constexpr (DUAL) fy[st % FR][i] = *(const bf16x8 *)(lds_a + dual_off + addr + cg * 32);
                } else {
                    fr[st % FR][i] = *(const bf16x8 *)(lds_b + a_b[i] + kx * SB + (cg - NCG_A) * 32);
                    if constexpr (DUAL) fy[st % FR][i] = *(const bf16x8 *)(lds_b + dual_off + a_b[i] + kx * SB + (cg - NCG_A) * 32);
                }
            }
        };
        if constexpr (!HAS_C1) {
#pragma unroll
            for (int m = 0; m < MS; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) k1[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        hook(0);
#ifdef IDM_EXP_PRIO_MFMA   // timing experiment: the MFMA loop at raised issue priority
        __builtin_amdgcn_s_setprio(IDM_EXP_PRIO_MFMA);
#endif
        static_for<0, FD>([&](auto S) { read_f(S); });
        static_for<0, NS>([&](auto S) {
            constexpr int st = decltype(S)::value;
            // pin the software pipeline: without this the scheduler hoists every load of the
            // unrolled loop to the top and spills
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (st + NPRE < NS) issue_w<st + NPRE>();
            if constexpr (st + FD < NS) read_f(std::integral_constant<int, st + FD>{});
#ifdef IDM_PIN_LOADS
            // ... and keep the scheduler from sinking the fragment reads to just before their first use (it does, to
            // reuse the registers of the fragments that die during this superstep, which exposes the LDS latency
            // once per superstep)
            __builtin_amdgcn_sched_barrier(0);
#endif
            if constexpr (st == NS - 1) {
#pragma unroll
                for (int m = 0; m < MS; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int co = (sl * MS + m) * 32 + 8 * q + 4 * h;
                        if constexpr (!BIAS_IN_ACC) k0[m][q] = *(const f32x4 *)(c0 + co);
                        if constexpr (HAS_C1) k1[m][q] = *(const f32x4 *)(c1 + co);
                    }
            }
            inter(S);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int m = 0; m < MS; ++m)
#pragma unroll
                    for (int n = 0; n < N; ++n)
#ifdef IDM_EXP_NOMFMA   // timing experiment (wrong results): everything but the MFMAs (operands are still fetched)
                    {
                        const bf16x8 ta = wreg[st % WR][ky][m], tb = fr[st % FR][n + ky];
                        asm volatile("" ::"v"(ta), "v"(tb));
                    }
#elif defined(IDM_EXP_MFMA16)   // timing experiment (wrong results): the same operands and FLOPs issued as two 16x16x32 MFMAs
                    {
                        f32x4 lo = {acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]};
                        f32x4 hi = {acc[m][n][4], acc[m][n][5], acc[m][n][6], acc[m][n][7]};
                        lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[st % WR][ky][m], fr[st % FR][n + ky], lo, 0, 0, 0);
                        hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[st % FR][n + ky], wreg[st % WR][ky][m], hi, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { acc[m][n][r] = lo[r]; acc[m][n][4 + r] = hi[r]; acc[m][n][8 + r] = lo[r] * 0.5f; acc[m][n][12 + r] = hi[r] * 0.5f; }
                    }
#else
                    {
                        // (an upsampled channel group's fragment may be the copy of its predecessor that was not read)
                        constexpr bool upg = UP && (st % NCG) < NCG_A;
                        const int fi = upg ? up_src<UPPAR>(n + ky) : n + ky;
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[st % WR][ky][m], fr[st % FR][fi],
                                                                            acc[m][n], 0, 0, 0);
                        if constexpr (DUAL)
                            accy[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[st % WR][ky][m], fy[st % FR][fi],
                                                                                 accy[m][n], 0, 0, 0);
                    }
#endif
        });
#ifdef IDM_EXP_PRIO_MFMA
        __builtin_amdgcn_s_setprio(0);
#endif
#ifdef IDM_EXP_PRIO_EPI   // timing experiment: the epilogue at raised issue priority
        __builtin_amdgcn_s_setprio(IDM_EXP_PRIO_EPI);
#endif
        hook(1);
#ifdef IDM_TIMELINE   // what the epilogue's first instructions wait for: the constants' LDS reads, then the last MFMAs
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        hook(6);
        {
            float t0 = acc[0][0][0], t1 = acc[0][N - 1][0];
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(t0) : "v"(t1));
            asm volatile("s_nop 0" ::"v"(t0));
        }
        hook(7);
#endif
        constexpr bool INDEXED = std::is_invocable_v<Epi &, std::integral_constant<int, 0>, const f32x16 &, const f32x4 *,
                                                     const f32x4 *, int, int, int, bool>;
        static_for<0, MS * N>([&](auto T) {
            constexpr int m = decltype(T)::value / N, n = decltype(T)::value % N;
            const bool valid = g.ok && cc < W && j + n + g.slot * Rh < nrows;
            if constexpr (INDEXED)
                epi(T, acc[m][n], k0[m], k1[m], sl * MS + m, rbase + n, cc, valid);
            else
                epi(acc[m][n], k0[m], k1[m], sl * MS + m, rbase + n, cc, valid);
            if (m * N + n < 3) hook(3 + m * N + n);
        });
        if constexpr (DUAL) {
            static_assert(MS == 1 && INDEXED, "the second lattice's tiles are told apart by the tile index");
            static_for<0, N>([&](auto T) {
                constexpr int n = decltype(T)::value;
                const bool valid = g.ok && cc < W && j + n + g.slot * Rh < nrows;
                epi(std::integral_constant<int, N + n>{}, accy[0][n], k0[0], k1[0], sl, rbase + n, cc, valid);
            });
        }
#ifdef IDM_EXP_PRIO_EPI
        __builtin_amdgcn_s_setprio(0);
#endif
        hook(2);
    }

    template <class Epi, class Hook = NoHook>
    DEVI void run(Epi &&epi, Hook &&hook = Hook{})
    {
        bool fetched = true;
#pragma unroll 1
        while (f < f1) {
            const int u = unit_of(f), j = f - u * Rh;
            const int len = min(f1 - f, Rh - j);
            // np pieces of at most NT tiles, as even as possible: cnt = ceil(len / np) <= NT
            const int np = (len + NT - 1) / NT;
            int cnt = 1;
#pragma unroll
            for (int k = 1; k < NT; ++k) cnt += k * np < len;
            const int sl = u % NSL, cb = u / NSL;
            if (!fetched && !STATIONARY) {
                set_slice(sl);
                prologue();
            }
            fetched = false;
            static_for<1, NT + 1>([&](auto N) {
                // (dispatching on the row parity here, so that piece can reuse the half-resolution tile's repeated fragments,
                // was measured on the 32-column strips of L = 64 / 128: conv6 -0.5 %, conv8 +2.4 %: not done; the eight-wavefront
                // fused_B does it for its own conv8, where it gains 1 %)
                if (cnt != decltype(N)::value) return;
                piece<decltype(N)::value>(j, sl, cb, epi, hook);
            });
            f += cnt;
        }
    }
};

// Register-held payload of one LDS input tile: NP 16-byte items per thread (item i = tid + k*WGS
// covers row i / (W*C/8), 8 channels).  load() reads rows [g0, g0+nrows) of a global NHWC bf16 map
// (zeros outside [0,Hmap)); store() writes them into the tile (row t = global row g0+t) and clears
// (unless HALO is false) the tile's two halo columns.  Splitting the two lets the loads of the next band fly while the
// current band is still being computed.
// PAD: bytes between the channels of neighbouring positions.  16 (position stride an odd number of 16-byte units) suits
// the operand reads of the 32x32x16 MFMA; 32 (stride = 2 mod 4 units) those of the 16x16x32 MFMA, whose ds_read_b128
// lane groups mix two 8-channel blocks of 16 pixels each.
template <int C, int NP, int WGS = WG, int PAD = 16>
struct TileRegs {
    static constexpr int S = C * 2 + PAD, CH = C / 8;
    bf16x8 v[NP];

    // src: this lattice's map (Hmap rows); rows outside it, and items beyond the tile, read as zero
    // through the buffer range check (their offsets wrap to >= 2^31).
    DEVI void load(const __bf16 *__restrict__ src, int Hmap, int W, int g0, int nrows, int tid)
    {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(src, (unsigned)(Hmap * W * C * 2));
        const int ipr = W * CH;
        int row = tid / ipr, rem = tid - row * ipr;
        const int drow = WGS / ipr, drem = WGS - drow * ipr;
        if (drem == 0) {  // a thread's items share one tile column: rows step by drow, offsets by a constant
            const int off0 = ((g0 + row) * ipr + rem) * 16;
#pragma unroll
            for (int k = 0; k < NP; ++k)
                v[k] = buf_load16(rs, row + k * drow < nrows ? (unsigned)(off0 + k * drow * ipr * 16) : 0x80000000u, 0);
            return;
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int off = ((g0 + row) * ipr + rem) * 16;
            v[k] = buf_load16(rs, row < nrows ? (unsigned)off : 0x80000000u, 0);
            rem += drem;
            row += drow;
            if (rem >= ipr) { rem -= ipr; ++row; }
        }
    }

    template <bool HALO = true>
    DEVI void store(char *tile, int W, int nrows, int tid) const
    {
        const int ipr = W * CH;
        int row = tid / ipr, rem = tid - row * ipr;
        const int drow = WGS / ipr, drem = WGS - drow * ipr;
        if (drem == 0) {
            char *p0 = tile + (row * (W + 2) + rem / CH + 1) * S + (rem % CH) * 16;
#pragma unroll
            for (int k = 0; k < NP; ++k)
                if (row + k * drow < nrows) *(bf16x8 *)(p0 + k * drow * (W + 2) * S) = v[k];
        } else {
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                if (row < nrows) *(bf16x8 *)(tile + (row * (W + 2) + rem / CH + 1) * S + (rem % CH) * 16) = v[k];
                rem += drem;
                row += drow;
                if (rem >= ipr) { rem -= ipr; ++row; }
            }
        }
        if constexpr (HALO) {
            bf16x8 z;
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.0f;
            for (int i = tid; i < nrows * 2 * CH; i += WGS) {
                const int ch = i % CH, side = (i / CH) & 1, r = i / (2 * CH);
                *(bf16x8 *)(tile + (r * (W + 2) + side * (W + 1)) * S + ch * 16) = z;
            }
        }
    }
};

// Zeroes the halo columns of rows [0,nrows) and the whole of rows [z0,z1) of an LDS tile.
template <int C, int WGS = WG, int PAD = 16>
DEVI void clear_halo(char *tile, int W, int nrows, int z0, int z1, int tid)
{
    constexpr int S = C * 2 + PAD, CH = C / 8;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < nrows * 2 * CH; i += WGS) {
        const int ch = i % CH, side = (i / CH) & 1, r = i / (2 * CH);
        *(f32x4 *)(tile + (r * (W + 2) + side * (W + 1)) * S + ch * 16) = z;
    }
    if (z1 > z0) {
        char *p = tile + z0 * (W + 2) * S;
        const int bytes = (z1 - z0) * (W + 2) * S;  // S is a multiple of 16
        for (int i = tid * 16; i < bytes; i += WGS * 16) *(f32x4 *)(p + i) = z;
    }
}


}  // namespace
