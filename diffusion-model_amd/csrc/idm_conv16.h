// 3x3 convolution stage on v_mfma_f32_16x16x32_bf16 for the 16 x 16 maps of the L = 32 pyramid (conv2, conv3, conv6, conv7):
//     D[16 couts][16 pixels] += W[16 couts][32 channels] * X[32 channels][16 pixels],  one map row per MFMA.
// The chip holds a higher clock on this MFMA shape than on 32x32x16 at the same FLOPs per cycle, so a stage that
// spends most of its time in the matrix pipe costs less energy per lattice - which the neighbouring kernels get back as
// clock.
//
// One wavefront computes what the 32x32x16 stage (ConvStage, four-tile pieces) gives it: a 32-cout slice (two 16-cout
// halves) of eight rows, in two groups of four vertically adjacent rows (j0 .. j0+3 and j0+8 .. j0+11).  A superstep is
// one (kx, 32-channel group): its six weight fragments (3 ky x 2 cout halves, 1 KB each, from L2 through a register
// ring) and the six input rows of each group feed 3 x 4 x 2 = 24 MFMAs per group.  The rows of a half-resolution
// (upsampled) tile repeat - full-resolution rows 2m-1 and 2m read the same one - so a group reads four fragments of it
// per superstep, not six.
//
// Operand lane maps (q = lane & 15, cb = lane >> 4):
//     B fragment: pixel pix(q), channels 8 cb .. +7 of the group: 16 bytes of the tile;
//     A fragment: cout q, the same channels (host-packed [tap][channel / 32][cout / 16][lane][8]);
//     D:          pixel pix(q), couts 4 cb .. +3 of the half.
// LDS tiles: [row][col][channel] with a zero column each side and positions 2 C + 32 bytes apart.  A ds_read_b128 is
// served in the 16-lane groups {0-3,12-15,20-27} and {4-11,16-19,28-31} (per half wave), which here mix two 8-channel
// blocks; with a stride of 2 (mod 4) 16-byte units and pix = (0-3, 8-15, 4-7) each group's eight pixels per block are
// consecutive columns, the two blocks take the even and the odd 16-byte slots, and every bank is touched once - also on
// the half-resolution tile, whose eight pixels span five columns.  (With the 2 C + 16 stride of the 32x32x16 stages
// every such group is 2-way conflicted, whatever the pixel map.)
// Rows: tile row = global row + 1; the map's top and bottom padding rows (global -1 and 16; half resolution -1 and 8)
// share the all-zero tile row 0, so a full-resolution tile has 17 rows and a half-resolution one 9.
// Every address is fixed per wavefront for the whole launch.
//
// PAIR geometry (8 x 8 maps, conv4 / conv5): the two "row groups" are two LATTICES whose tiles lie LAT bytes apart, and a
// 16-pixel MFMA covers rows n and n + 4 of the map (pixels 0-7 and 8-15), so that the four MFMAs of a tap row still read
// consecutive fragments (fragment e = tile rows e and e + 4, e = 0 .. 5) and a weight fragment feeds eight MFMAs.  Tile
// rows are 9 positions apart: the right halo column of a row IS the left halo column of the next, and the bottom padding
// row is the shared zero row 0 again (82 positions per tile).  The conflict argument above holds unchanged: the two
// 8-pixel halves of a ds_read_b128 lane group are 4 rows apart, an even number of 16-byte units.
#pragma once
#include "idm_conv_stage.h"

namespace {

// CA: channels of the upsampled (half-resolution) part, 0 = none; CB: channels of the full-resolution part.
// The output goes, activated, to a tile with positions SOUT bytes apart: OUT_HALO = a tile of the layout above (the next
// stage's input: row + 1, column + 1), else a plain [16][16] staging tile.
// PAIR: the lattice-pair geometry for 8 x 8 maps (CA must be 0); TADD: a per-cout constant is added after the activation.
// HROWS (PAIR): rows of the map, 5 .. 8; outputs of rows beyond it are not stored (an OUT_HALO tile keeps them zero).
template <int CA, int CB, int COUT, int WD, int SOUT, bool OUT_HALO, bool PAIR = false, bool TADD = false, int HROWS = 8>
struct ConvMfma16 {
    static_assert(HROWS >= 5 && HROWS <= 8 && (PAIR || HROWS == 8), "map rows");
    static constexpr int W = PAIR ? 8 : 16, SA = CA * 2 + 32, SB = CB * 2 + 32;
    static constexpr int NGA = CA / 32, NGB = CB / 32, NCG = NGA + NGB, NS = 3 * NCG, MT16 = COUT / 16;
    static constexpr int PITCH = PAIR ? W + 1 : W + 2;             // positions from one tile row to the next
    static constexpr int PU = (W / 2 + 2) * SA, PB = PITCH * SB, PO = (OUT_HALO ? PITCH : W) * SOUT;
    static constexpr int WR = WD + 1;
    static constexpr int ROWS_UP = W / 2 + 1, ROWS_FULL = W + 1;   // tile rows (the shared zero row included)
    static constexpr int PAIR_POS = (W + 1) * (W + 1) + 1;         // positions of a PAIR tile
    static_assert(WD >= 1 && WD < NS && CB % 32 == 0 && CA % 32 == 0 && COUT % 32 == 0 && !(PAIR && CA), "shape");

    const char *ta, *tb;
    char *dst0[2];                  // output tile: first row of each row group, this lane's pixel and first cout
    const float *bias;              // this lane's first cout of the slice
    const float *tadd = nullptr;    // TADD: likewise
    int upper = 0;                  // PAIR: this lane's pixel is in rows 4 .. 7 (n + 4) of the map
    __amdgpu_buffer_rsrc_t wrsrc;
    unsigned woff;
    // byte offsets of a group's input rows: consecutive rows follow the first by constant steps; only a group's LAST row
    // can be the bottom padding row, which is tile row 0 (kept apart).  cu: half-resolution column per kx.
    int ra0[2], ra3[2], rb0[2], rb5[2], cu[3];
    bf16x8 wq[WR][3][2];

    static DEVI int pix(int q) { return q < 4 ? q : q < 12 ? q + 4 : q - 8; }

    DEVI void setup(const char *tile_up, const char *tile_full, const float *c_bias, const void *w16, char *out, int lane,
                    int sl, int j0)
    {
        const int x = pix(lane & 15), cb = lane >> 4;
        ta = tile_up; tb = tile_full;
        bias = c_bias + 32 * sl + 4 * cb;
        wrsrc = make_rsrc(w16, 9u * (CA + CB) * COUT * 2u);
        unsigned o = (unsigned)(lane * 16 + 2 * sl * 1024);
        asm volatile("" : "+v"(o));     // (opaque: keeps the weight fetches from being hoisted out of the lattice loop)
        woff = o;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            // output rows j0 + 8 g + n, n < 4; their taps read global rows g0 .. g0 + 5, g0 = j0 + 8 g - 1 (odd)
            const int g0 = j0 + 8 * g - 1;
            const int t0 = g0 + 1, t5 = g0 + 6;                    // full-resolution tile rows of input rows 0 and 5
            rb0[g] = t0 * PB + x * SB + cb * 16;
            rb5[g] = (t5 == W + 1 ? 0 : t5) * PB + x * SB + cb * 16;
            const int h0 = (g0 + 2) >> 1;                           // half-resolution tile row of input row 0
            ra0[g] = h0 * PU;
            ra3[g] = (h0 + 3 == W / 2 + 1 ? 0 : h0 + 3) * PU;
            const int row = j0 + 8 * g;
            dst0[g] = out + (OUT_HALO ? ((row + 1) * (W + 2) + x + 1) : (row * W + x)) * SOUT + (32 * sl + 4 * cb) * 2;
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) cu[kx] = ((x + kx + 1) >> 1) * SA + cb * 16;
    }

    // PAIR: tile / out = lattice 0's, lattice 1's lie lat_in / lat_out bytes further; sl = the wavefront's 32-cout slice
    DEVI void setup_pair(const char *tile, int lat_in, const float *c_bias, const float *c_tadd, const void *w16, char *out,
                         int lat_out, int lane, int sl)
    {
        static_assert(!PAIR || CA == 0, "no upsampled part");
        const int p = pix(lane & 15), hi = p >> 3, col = p & 7, cb = lane >> 4;
        ta = nullptr; tb = tile;
        upper = hi;
        bias = c_bias + 32 * sl + 4 * cb;
        tadd = c_tadd ? c_tadd + 32 * sl + 4 * cb : nullptr;
        wrsrc = make_rsrc(w16, 9u * (CA + CB) * COUT * 2u);
        unsigned o = (unsigned)(lane * 16 + 2 * sl * 1024);
        asm volatile("" : "+v"(o));
        woff = o;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            // fragment e reads tile rows e (pixels 0-7) and e + 4 (pixels 8-15); row 9 of the upper half is the zero row 0
            rb0[g] = g * lat_in + (4 * hi * PITCH + col) * SB + cb * 16;
            rb5[g] = g * lat_in + ((hi ? 0 : 5) * PITCH + col) * SB + cb * 16;
            ra0[g] = ra3[g] = 0;
            const int row = 4 * hi;     // output rows n + 4 hi
            dst0[g] = out + g * lat_out + (OUT_HALO ? ((row + 1) * PITCH + col + 1) : (row * W + col)) * SOUT + (32 * sl + 4 * cb) * 2;
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) cu[kx] = 0;
    }

    template <int S>
    DEVI void issue_w()
    {
        constexpr int kx = S / NCG, cg = S % NCG;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int hc = 0; hc < 2; ++hc) {
                const unsigned k = (((ky * 3 + kx) * NCG + cg) * MT16 + hc) * 1024;
                wq[S % WR][ky][hc] = buf_load16(wrsrc, woff + (k & 0xFFFu), k & ~0xFFFu);
            }
    }

    // the first WD supersteps' weights: callable as soon as the previous run() has issued its last MFMA
    DEVI void prologue()
    {
        static_for<0, WD>([&](auto S) { issue_w<decltype(S)::value>(); });
    }

    // fragment of input row i of a group in superstep S: index into the fragments read for that superstep
    template <int S>
    static constexpr int fidx(int i) { return (S % NCG) < NGA ? (i + 1) >> 1 : i; }

    template <int S>
    DEVI void read_group(int g, bf16x8 (&f)[6]) const
    {
        constexpr int kx = S / NCG, cg = S % NCG;
        if constexpr (cg < NGA) {
#pragma unroll
            for (int k = 0; k < 3; ++k) f[k] = *(const bf16x8 *)(ta + ra0[g] + cu[kx] + k * PU + cg * 64);
            f[3] = *(const bf16x8 *)(ta + ra3[g] + cu[kx] + cg * 64);
        } else {
#pragma unroll
            for (int i = 0; i < 5; ++i) f[i] = *(const bf16x8 *)(tb + rb0[g] + i * PB + kx * SB + (cg - NGA) * 64);
            f[5] = *(const bf16x8 *)(tb + rb5[g] + kx * SB + (cg - NGA) * 64);
        }
    }

    // One map: prologue() has been called since the last run(), the input tiles are complete.  mid(): called between the
    // last MFMA and the epilogue's first store (a barrier, where the output tile lies over an input tile).
    struct NoMid { DEVI void operator()() const {} };
    template <class Mid = NoMid>
    DEVI void run(Mid &&mid = Mid{})
    {
        f32x4 acc[2][4][2];
        {
            const f32x4 k0 = *(const f32x4 *)bias, k1 = *(const f32x4 *)(bias + 16);
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int n = 0; n < 4; ++n) { acc[g][n][0] = k0; acc[g][n][1] = k1; }
        }
        bf16x8 f0[6], f1[6];
        read_group<0>(0, f0);
        static_for<0, NS>([&](auto SS) {
            constexpr int S = decltype(SS)::value;
            // Two scheduling regions per superstep, each 24 MFMAs with the NEXT region's operand requests dealt one per
            // MFMA at its head (weights WD supersteps ahead with the first group, the other group's rows one region
            // ahead): a 16x16x32 MFMA leaves the wavefront one or two issue slots, so requests issued in a clump - or
            // sunk to just before their first use, which is what the scheduler does when left alone - idle the pipe.
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (S + WD < NS) issue_w<S + WD>();
            read_group<S>(1, f1);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int hc = 0; hc < 2; ++hc)
                        acc[0][n][hc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[S % WR][ky][hc], f0[fidx<S>(n + ky)], acc[0][n][hc], 0, 0, 0);
            {
                constexpr int nds = (S % NCG) < NGA ? 4 : 6, nvm = S + WD < NS ? 6 : 0;
#pragma unroll
                for (int k = 0; k < nds; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read
                }
#pragma unroll
                for (int k = 0; k < nvm; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // one vector-memory read
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 24 - nds - nvm, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (S + 1 < NS) read_group<S + 1>(0, f0);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int hc = 0; hc < 2; ++hc)
                        acc[1][n][hc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[S % WR][ky][hc], f1[fidx<S>(n + ky)], acc[1][n][hc], 0, 0, 0);
            if constexpr (S + 1 < NS) {
                constexpr int nds = ((S + 1) % NCG) < NGA ? 4 : 6;
#pragma unroll
                for (int k = 0; k < nds; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 24 - nds, 0);
            }
        });
        mid();
        f32x4 kt[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if constexpr (TADD) { kt[0] = *(const f32x4 *)tadd; kt[1] = *(const f32x4 *)(tadd + 16); }
        // activation and store: this lane's four couts of one pixel per accumulator
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int hc = 0; hc < 2; ++hc) {
                    if constexpr (PAIR && HROWS < 8) {
                        if (upper && n >= HROWS - 4) continue;   // a row below the map
                    }
                    const f32x4 v = acc[g][n][hc];
#ifdef IDM_EXP_C7_RAW   // timing experiment (wrong results): conv7 stores its pre-activations (what moving the activation to the copy wavefronts would leave here)
                    if constexpr (!OUT_HALO) { *(bf16x4 *)(dst0[g] + n * PO + hc * 32) = pack4(v[0], v[1], v[2], v[3]); continue; }
#endif
                    if constexpr (TADD)
                        *(bf16x4 *)(dst0[g] + n * PO + hc * 32) = pack4(silu1(v[0]) + kt[hc][0], silu1(v[1]) + kt[hc][1], silu1(v[2]) + kt[hc][2], silu1(v[3]) + kt[hc][3]);
                    else
                        *(bf16x4 *)(dst0[g] + n * PO + hc * 32) = pack4(silu1(v[0]), silu1(v[1]), silu1(v[2]), silu1(v[3]));
                }
    }
};

}  // namespace
