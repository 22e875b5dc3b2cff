// LDS-resident 2-D inverse real FFT + sign projection for the spectral model:
//     spins = np.where(np.fft.irfft2(re + 1j*im, s=(L, L)) > 0, 1, -1)
// (ising_ifft, b_fft.py:12-17, followed by b_diffusion_model_fft_unet_2d_ising.py:214-217), float64 arithmetic,
// numpy's axis order: a complex inverse transform along axis 0 for each of the L/2+1 stored columns, then a
// complex-to-real transform along axis 1 that ignores the imaginary parts of the k = 0 and k = L/2 columns.
//
// Every 1-D transform is a mixed-radix (4, 2, 3, 5) Cooley-Tukey FFT held entirely in the registers of one
// thread (or of two / four neighbouring lanes that exchange parts through DPP for L = 64 / 128); loops are unrolled at
// compile time, so every twiddle index is a constant and the factors are scalar loads from the kernel argument.
//     step 1  thread (lattice, column k): 1-D inverse FFT over the L rows, read straight from global memory
//             (consecutive lanes read consecutive 8-byte (Re, Im) pairs), written to LDS as float64
//     step 2  thread (lattice, row pair p): rows 2p and 2p+1 are both real, so their two complex-to-real
//             transforms are ONE complex FFT of Z = Ya + i*Yb (Hermitian-extended), whose real part is row 2p and
//             whose imaginary part is row 2p+1; the signs leave as 2L contiguous bytes per thread
// A lattice crosses LDS once (L/2+1 columns of L float64 complex); HBM traffic is the algorithmic
// 8*L*(L/2+1) bytes in and L*L bytes out.
#include "idm_internal.h"
#include "idm_device.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace {

struct cplx { double x, y; };

struct FftTw { cplx w[128]; };   // w[q] = exp(+2 pi i q / L), q < L

DEVI cplx cadd(cplx a, cplx b) { return {a.x + b.x, a.y + b.y}; }
DEVI cplx csub(cplx a, cplx b) { return {a.x - b.x, a.y - b.y}; }
DEVI cplx cmul(cplx a, cplx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
DEVI cplx muli(cplx a) { return {-a.y, a.x}; }       // i * a
DEVI cplx cneg(cplx a) { return {-a.x, -a.y}; }

// a * w_L^idx with the trivial factors (1, i, -1, -i) resolved at compile time.
template <int L, int IDX>
DEVI cplx twiddle(cplx a, const cplx *__restrict__ tw)
{
    constexpr int q = ((IDX % L) + L) % L;
    if constexpr (q == 0) return a;
    else if constexpr (4 * q == L) return muli(a);
    else if constexpr (2 * q == L) return cneg(a);
    else if constexpr (4 * q == 3 * L) return cneg(muli(a));
    else return cmul(a, tw[q]);
}

// In-place inverse DFTs (kernel e^{+2 pi i s q / R}) of the radices the factorisation uses.
DEVI void dft_small(cplx (&v)[2])
{
    const cplx a = v[0], b = v[1];
    v[0] = cadd(a, b); v[1] = csub(a, b);
}

DEVI void dft_small(cplx (&v)[4])
{
    const cplx t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]), t2 = cadd(v[1], v[3]), t3 = muli(csub(v[1], v[3]));
    v[0] = cadd(t0, t2); v[1] = cadd(t1, t3); v[2] = csub(t0, t2); v[3] = csub(t1, t3);
}

DEVI void dft_small(cplx (&v)[3])
{
    const double h = 0.86602540378443864676;   // sin(2 pi / 3)
    const cplx s = cadd(v[1], v[2]), d = csub(v[1], v[2]);
    const cplx m = {v[0].x - 0.5 * s.x, v[0].y - 0.5 * s.y};
    const cplx r = {-h * d.y, h * d.x};         // i * h * d
    v[0] = cadd(v[0], s); v[1] = cadd(m, r); v[2] = csub(m, r);
}

DEVI void dft_small(cplx (&v)[5])
{
    const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;   // cos(2 pi / 5), cos(4 pi / 5)
    const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;    // sin(2 pi / 5), sin(4 pi / 5)
    const cplx a1 = cadd(v[1], v[4]), a2 = cadd(v[2], v[3]), b1 = csub(v[1], v[4]), b2 = csub(v[2], v[3]);
    const cplx m1 = {v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y};
    const cplx m2 = {v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y};
    const cplx r1 = {-(s1 * b1.y + s2 * b2.y), s1 * b1.x + s2 * b2.x};          // i * (s1 b1 + s2 b2)
    const cplx r2 = {-(s2 * b1.y - s1 * b2.y), s2 * b1.x - s1 * b2.x};          // i * (s2 b1 - s1 b2)
    v[0] = cadd(v[0], cadd(a1, a2));
    v[1] = cadd(m1, r1); v[4] = csub(m1, r1);
    v[2] = cadd(m2, r2); v[3] = csub(m2, r2);
}

constexpr int radix_of(int n) { return n % 4 == 0 ? 4 : n % 2 == 0 ? 2 : n % 3 == 0 ? 3 : 5; }

// Where the in-place transform below leaves X[k] of an n-point transform.
constexpr int fft_pos(int n, int k)
{
    if (n == 1) return 0;
    const int r = radix_of(n), m = n / r;
    return (k % r) * m + fft_pos(m, k / r);
}

template <int I, int N, class F>
DEVI void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// In-place N-point inverse DFT of a[BASE .. BASE+N), decimation in frequency: with N = r*M, n = n1*M + n2 and
// k = k1 + r*k2,
//     X[k1 + r k2] = sum_n2 [ (sum_n1 a[n1 M + n2] w_r^{n1 k1}) w_N^{n2 k1} ] w_M^{n2 k2},
// i.e. M radix-r butterflies whose outputs (twiddled) go back to the slots they came from, then the same on
// each of the r blocks of M.  X[k] ends at slot fft_pos(N, k).  Only one butterfly's temporaries are live besides
// the array itself.  Twiddles w_N^j = w_L^{j L / N} come from the table of the L-point transform.
template <int N, int L, int BASE, int NTOT>
DEVI void fft_inplace(cplx (&a)[NTOT], const cplx *__restrict__ tw)
{
    constexpr int R = radix_of(N), M = N / R;
    static_assert(M * R == N && L % N == 0, "unsupported length");
    static_for<0, M>([&](auto n2c) {
        constexpr int n2 = decltype(n2c)::value;
        cplx v[R];
#pragma unroll
        for (int n1 = 0; n1 < R; ++n1) v[n1] = a[BASE + n1 * M + n2];
        dft_small(v);
        static_for<0, R>([&](auto k1c) {
            constexpr int k1 = decltype(k1c)::value;
            a[BASE + k1 * M + n2] = twiddle<L, n2 * k1 * (L / N)>(v[k1], tw);
        });
    });
    if constexpr (M > 1)
        static_for<0, R>([&](auto k1c) { fft_inplace<M, L, BASE + decltype(k1c)::value * M, NTOT>(a, tw); });
}

// value of lane ^ 1 / lane ^ 2 (DPP quad permutes)
DEVI double swap1(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
DEVI double swap2(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// L-point inverse DFT whose input is dealt to SPLIT lanes (lane h holds x[SPLIT*i + h], i < N = L / SPLIT) and whose
// output comes back as the block X[N*hb + k], k < N, at slot fft_pos(N, k), hb = split_block<SPLIT>(h).
//   SPLIT = 2: each lane transforms its half, the halves are swapped through DPP and X[k] = E[k] + w^k O[k],
//              X[k + N] = E[k] - w^k O[k]; lane h keeps block h.
//   SPLIT = 4: two such levels.  Lanes h and h ^ 2 (residues r and r + 2 mod 4) combine into the 2N-point transform G_r of
//              the residue class r mod 2 (twiddle w_L^{2k}); lane bit 1 selects its lower or upper half.  Then lanes h and
//              h ^ 1 combine G_0 and G_1 with twiddle w_L^{k + N a} = i^a w_L^k (a = h >> 1), lane bit 0 selecting X[m] or
//              X[m + 2N]: lane h ends with block (h >> 1) + 2 (h & 1).
template <int SPLIT>
DEVI int split_block(int h)
{
    return SPLIT == 4 ? (h >> 1) + 2 * (h & 1) : h;
}

template <int L, int SPLIT, int K0 = 0>
DEVI void combine_split(cplx *y, int h, const cplx *__restrict__ tw)
{
    constexpr int N = L / SPLIT;
    if constexpr (SPLIT == 2 && K0 < N) {
        constexpr int pos = fft_pos(N, K0);
        cplx &slot = y[pos];
        const cplx mine = slot;
        const cplx other = {swap1(mine.x), swap1(mine.y)};
        const cplx e = h ? other : mine, o = h ? mine : other;
        const cplx t = twiddle<L, K0>(o, tw);
        slot = h ? csub(e, t) : cadd(e, t);
        combine_split<L, SPLIT, K0 + 1>(y, h, tw);
    } else if constexpr (SPLIT == 4 && K0 < N) {
        constexpr int pos = fft_pos(N, K0);
        cplx &slot = y[pos];
        const bool up1 = (h & 2) != 0, up2 = (h & 1) != 0;
        cplx mine = slot;
        cplx other = {swap2(mine.x), swap2(mine.y)};
        cplx e = up1 ? other : mine, o = up1 ? mine : other;
        cplx t = twiddle<L, 2 * K0>(o, tw);
        mine = up1 ? csub(e, t) : cadd(e, t);
        other = {swap1(mine.x), swap1(mine.y)};
        e = up2 ? other : mine;
        o = up2 ? mine : other;
        t = twiddle<L, K0>(o, tw);
        if (up1) t = muli(t);
        slot = up2 ? csub(e, t) : cadd(e, t);
        combine_split<L, SPLIT, K0 + 1>(y, h, tw);
    }
}

template <int L, int SPLIT, int NB, int WGS, int OCC>
__global__ __launch_bounds__(WGS, OCC) void irfft2_fft_kernel(const float *__restrict__ spec, int64_t B, int8_t *__restrict__ spins,
                                                         double *__restrict__ real_out, const FftTw tw)
{
    constexpr int K = L / 2 + 1, N = L / SPLIT, CS = L + 1, HP = L / 2;
    static_assert(NB * K * SPLIT <= WGS, "a group's column transforms must fit the workgroup");
    extern __shared__ __attribute__((aligned(16))) char fft_smem[];
    cplx *ylds = (cplx *)fft_smem;   // [NB][K][CS]: column k of a lattice, rows split by parity: even rows first
    const int t = threadIdx.x, h = SPLIT > 1 ? (t & (SPLIT - 1)) : 0, task = t / SPLIT;
    const int hb = split_block<SPLIT>(h);   // the output block this lane holds after combine_split
    const int64_t ngroups = (B + NB - 1) / NB;
    // The (Re, Im) pairs of the next group are requested while the current one is transformed (float32, so the
    // prefetch costs 2N registers): with one or two wavefronts per SIMD nothing else would cover the HBM latency.
    constexpr bool PF = N <= 16 || (N <= 32 && SPLIT == 1);   // (N = 32 on four lanes: the registers go to the transform)
    float2 pre[PF ? N : 1];
    const int lb1 = task / K, k1 = task - lb1 * K;
    auto fetch = [&](int64_t g) {
        const int64_t b = g * NB + lb1;
        if (lb1 < NB && b < B) {
            const float2 *src = (const float2 *)spec + (b * L) * K + k1;
#pragma unroll
            for (int i = 0; i < N; ++i) pre[i] = src[(int64_t)(i * SPLIT + h) * K];
        }
    };
    if constexpr (PF) {
        if ((int64_t)blockIdx.x < ngroups) fetch(blockIdx.x);
    }
    for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        {   // ---- step 1: inverse FFT along axis 0, one column per thread (pair) ----
            const int lb = lb1, k = k1;
            const int64_t b = g * NB + lb;
            const bool act = lb < NB && b < B;
            cplx y[N];
            if constexpr (PF) {
#pragma unroll
                for (int i = 0; i < N; ++i) y[i] = {(double)pre[i].x, (double)pre[i].y};
                if (g + gridDim.x < ngroups) fetch(g + gridDim.x);
            } else if (act) {
                const float2 *src = (const float2 *)spec + (b * L) * K + k;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const float2 v = src[(int64_t)(i * SPLIT + h) * K];
                    y[i] = {(double)v.x, (double)v.y};
                }
            }
            if (act) {
                fft_inplace<N, L, 0, N>(y, tw.w);
                combine_split<L, SPLIT>(y, h, tw.w);
                cplx *col = ylds + (lb * K + k) * CS;
                static_for<0, N>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, slot = fft_pos(N, i);
                    const int n = i + N * hb;   // N is even (or SPLIT == 1): the parity of n is that of i
                    col[(i & 1) * HP + (n >> 1)] = y[slot];
                });
            }
        }
        __syncthreads();
        {   // ---- step 2: rows 2p and 2p+1 as one complex FFT along axis 1 ----
            const int lb = task / HP, p = task - lb * HP;
            const int64_t b = g * NB + lb;
            if (lb < NB && b < B) {
                const cplx *lat = ylds + lb * K * CS;
                cplx y[N];
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const int j = i * SPLIT + h;
                    const bool ext = j > HP;             // Hermitian extension: Z[j] from column L - j, conjugated
                    const int kk = ext ? L - j : j;
                    const cplx ya = lat[kk * CS + p], yb = lat[kk * CS + HP + p];
                    if (j == 0 || j == HP) y[i] = {ya.x, yb.x};   // numpy's C2R drops these imaginary parts
                    else if (ext) y[i] = {ya.x + yb.y, yb.x - ya.y};
                    else y[i] = {ya.x - yb.y, ya.y + yb.x};
                }
                fft_inplace<N, L, 0, N>(y, tw.w);
                combine_split<L, SPLIT>(y, h, tw.w);
                const double norm = 1.0 / ((double)L * (double)L);
                // this thread holds columns [N*hb, N*hb + N) of rows 2p (real parts) and 2p+1 (imaginary parts)
                const int64_t row0 = (b * L + 2 * p) * L + N * hb;
                if (real_out) {
                    static_for<0, N>([&](auto ic) {
                        constexpr int i = decltype(ic)::value, slot = fft_pos(N, i);
                        real_out[row0 + i] = y[slot].x * norm;
                        real_out[row0 + L + i] = y[slot].y * norm;
                    });
                }
                uint32_t wa[N / 4 + 1], wb[N / 4 + 1];
#pragma unroll
                for (int i = 0; i < N / 4 + 1; ++i) wa[i] = wb[i] = 0;
                static_for<0, N>([&](auto ic) {
                    // (the sign does not need the 1 / L^2 normalisation)
                    constexpr int i = decltype(ic)::value, slot = fft_pos(N, i);
                    wa[i >> 2] |= (y[slot].x > 0.0 ? 0x01u : 0xFFu) << (8 * (i & 3));
                    wb[i >> 2] |= (y[slot].y > 0.0 ? 0x01u : 0xFFu) << (8 * (i & 3));
                });
                if constexpr (N % 16 == 0) {
                    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#pragma unroll
                    for (int i = 0; i < N / 16; ++i) {
                        *(u32x4 *)(spins + row0 + 16 * i) = u32x4{wa[4 * i], wa[4 * i + 1], wa[4 * i + 2], wa[4 * i + 3]};
                        *(u32x4 *)(spins + row0 + L + 16 * i) = u32x4{wb[4 * i], wb[4 * i + 1], wb[4 * i + 2], wb[4 * i + 3]};
                    }
                } else if constexpr (N % 4 == 0) {
#pragma unroll
                    for (int i = 0; i < N / 4; ++i) {
                        *(uint32_t *)(spins + row0 + 4 * i) = wa[i];
                        *(uint32_t *)(spins + row0 + L + 4 * i) = wb[i];
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        spins[row0 + i] = (int8_t)(wa[i >> 2] >> (8 * (i & 3)));
                        spins[row0 + L + i] = (int8_t)(wb[i >> 2] >> (8 * (i & 3)));
                    }
                }
            }
        }
        __syncthreads();   // the next group's columns overwrite the tile
    }
}

template <int L, int SPLIT, int WGS, int OCC = 1>
hipError_t launch_fft(const float *spec, int64_t B, int8_t *spins, double *real_out, hipStream_t s)
{
    constexpr int K = L / 2 + 1, NB = WGS / (K * SPLIT);
    static_assert(NB >= 1, "workgroup too small");
    constexpr size_t lds = (size_t)NB * K * (L + 1) * sizeof(cplx);
    static_assert(lds <= 160 * 1024, "tile exceeds LDS");
    auto kernel = irfft2_fft_kernel<L, SPLIT, NB, WGS, OCC>;
    static thread_local bool configured[IDM_MAX_DEVICES] = {};
    static thread_local int blocks_per_cu[IDM_MAX_DEVICES] = {}, cus[IDM_MAX_DEVICES] = {};
    const int slot = idm_device_slot();
    if (slot < 0) return hipErrorInvalidDevice;
    if (!configured[slot]) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        int n = 0;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)kernel, WGS, lds)) != hipSuccess) return e;
        hipDeviceProp_t prop;
        if ((e = hipGetDeviceProperties(&prop, slot)) != hipSuccess) return e;
        blocks_per_cu[slot] = n > 0 ? n : 1;
        cus[slot] = prop.multiProcessorCount;
        configured[slot] = true;
    }
    // long double sin / cos rounded once to double (the quarter turns, where that would leave 1e-20 instead
    // of 0, are never read: twiddle() resolves them at compile time)
    static const FftTw tw = [] {
        FftTw t;
        for (int q = 0; q < 128; ++q) {
            const long double a = 2.0L * 3.14159265358979323846264338327950288L * (long double)(q % L) / (long double)L;
            t.w[q].x = (double)cosl(a);
            t.w[q].y = (double)sinl(a);
        }
        return t;
    }();
    const int64_t ngroups = (B + NB - 1) / NB;
    const int64_t grid = std::min<int64_t>(ngroups, (int64_t)cus[slot] * blocks_per_cu[slot]);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(WGS), lds, s, spec, B, spins, real_out, tw);
    return hipGetLastError();
}

}  // namespace

// Lattice sizes with an in-register FFT instance; other even sizes use the direct-sum kernel.
bool idm_fft_supported(int L)
{
    switch (L) {
    case 4: case 8: case 10: case 12: case 16: case 20: case 24: case 32: case 40: case 64: case 128: return true;
    default: return false;
    }
}

hipError_t launch_irfft2_fft(const float *spec, int64_t B, int L, int8_t *spins, double *real_out, hipStream_t s)
{
#ifdef IDM_FFT_TUNE   // tuning builds: ISINGDIFF_FFT_CFG picks a candidate shape (lanes per transform, workgroup size)
    static const int cfg = [] { const char *e = getenv("ISINGDIFF_FFT_CFG"); return e ? atoi(e) : 0; }();
    if (L == 32) {
        switch (cfg) {
        case 1: return launch_fft<32, 2, 256, 1>(spec, B, spins, real_out, s);
        case 2: return launch_fft<32, 2, 128, 1>(spec, B, spins, real_out, s);
        case 3: return launch_fft<32, 4, 256, 1>(spec, B, spins, real_out, s);
        case 4: return launch_fft<32, 4, 512, 1>(spec, B, spins, real_out, s);
        case 5: return launch_fft<32, 2, 192, 1>(spec, B, spins, real_out, s);
        default: break;
        }
    }
    if (L == 20 && cfg == 1) return launch_fft<20, 2, 128, 1>(spec, B, spins, real_out, s);
    if (L == 64) {
        switch (cfg) {
        case 1: return launch_fft<64, 4, 512, 1>(spec, B, spins, real_out, s);
        case 2: return launch_fft<64, 4, 256, 1>(spec, B, spins, real_out, s);
        default: break;
        }
    }
#endif
    switch (L) {
    case 4: return launch_fft<4, 1, 128>(spec, B, spins, real_out, s);
    case 8: return launch_fft<8, 1, 128>(spec, B, spins, real_out, s);
    case 10: return launch_fft<10, 1, 128>(spec, B, spins, real_out, s);
    case 12: return launch_fft<12, 1, 128>(spec, B, spins, real_out, s);
    case 16: return launch_fft<16, 1, 128>(spec, B, spins, real_out, s);
    case 20: return launch_fft<20, 1, 128>(spec, B, spins, real_out, s);
    case 24: return launch_fft<24, 1, 128>(spec, B, spins, real_out, s);
    case 32: return launch_fft<32, 1, 128>(spec, B, spins, real_out, s);
    case 40: return launch_fft<40, 1, 128>(spec, B, spins, real_out, s);
    // 64: two lanes per transform, four lattices per workgroup (193 us per 16 384 lattices; three lattices in 256 threads:
    // 218 us; four lanes per transform: 229 - 253 us, the second combine level costs more than the registers it frees); 128: four lanes (287 us against 631 us per 4 096 lattices: two lanes hold 64 float64
    // complex each and spill)
    case 64: return launch_fft<64, 2, 320, 1>(spec, B, spins, real_out, s);
    case 128: return launch_fft<128, 4, 320, 1>(spec, B, spins, real_out, s);
    default: return hipErrorInvalidValue;
    }
}
