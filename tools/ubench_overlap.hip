// Micro-benchmark: do matrix (MFMA) and vector-ALU work of the two wavefronts of a SIMD overlap on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_overlap tools/ubench_overlap.hip && /tmp/ubench_overlap
// One workgroup of 8 wavefronts per CU (wavefronts w and w + 4 share a SIMD).  Modes:
//   0  every wavefront: NM MFMAs (32x32x16 bf16, 4 independent accumulators)
//   1  every wavefront: NV v_exp_f32 (8 independent chains)
//   2  wavefronts 0-3 MFMA, 4-7 exp              (cross-wavefront overlap)
//   3  as 2, MFMA wavefronts at raised priority
//   4  every wavefront: NM/2 MFMAs interleaved with NV/2 exps in program order (same-wavefront overlap)
//   5  every wavefront: NM/2 MFMAs, then NV/2 exps (phases, free-running wavefronts)
//   6  as 1 with packed fma instead of exp (full-rate vector ALU), same instruction count x4
//   7  wavefronts 0-3 MFMA, 4-7 packed fma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int NM = 4096;        // MFMAs per MFMA wavefront: 4096 * 32 cycles
constexpr int NV = 8192;        // exps per exp wavefront:   8192 * 16 cycles

__device__ __forceinline__ void mfma_block(f32x16 (&acc)[4], bf16x8 a, bf16x8 b)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
}

__device__ __forceinline__ void exp_block(float (&v)[8])
{
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
}

__device__ __forceinline__ void fma_block(f32x2 (&v)[8], f32x2 c)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = v[i] * c + c;
}

// Vector-ALU instruction kinds for modes >= 10: mode 10 + 2 j = eight wavefronts of kind j, 11 + 2 j = four MFMA wavefronts +
// four of kind j.  Each kind: NV * 4 instructions per wavefront in 8 independent chains.
enum { OP_FMA = 0, OP_PKMUL, OP_PKADD, OP_CVTPK, OP_MUL, OP_RCP, OP_LDS, OP_PKFMA_BF16, OP_MOV, NOPS };
static const char *op_names[NOPS] = {"v_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_cvt_pk_bf16_f32", "v_mul_f32", "v_rcp_f32 (NV only)", "ds_read_b128 (NV only)", "v_pk_fma_bf16?/f16", "v_mov_b32"};

template <int OP>
__device__ __forceinline__ void op_loop(float (&v)[8], f32x2 (&p)[8], f32x2 c, const char *lds, int lane)
{
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    if constexpr (OP == OP_LDS) {
        typedef __attribute__((ext_vector_type(4))) float f4;
        f4 acc4 = {0, 0, 0, 0};
#pragma unroll 1
        for (int it = 0; it < NV / 8; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f4 t = *(const volatile f4 *)(lds + ((lane * 16 + i * 1040 + it * 16) & 0xFFF0));
                acc4 += t;
            }
        }
        v[0] += acc4[0] + acc4[1] + acc4[2] + acc4[3];
        return;
    }
    constexpr int N = (OP == OP_RCP) ? NV / 8 : NV * 4 / 8;
#pragma unroll 1
    for (int it = 0; it < N; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c[0]));
            if constexpr (OP == OP_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c[0]));
            if constexpr (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(c[0]));
            if constexpr (OP == OP_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
            if constexpr (OP == OP_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c));
            if constexpr (OP == OP_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c));
            if constexpr (OP == OP_CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "+v"(v[i]) : "v"(c[0]), "v"(c[1]));
            if constexpr (OP == OP_PKFMA_BF16) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c[0]));
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float *out, int dummy)
{
    const int wave = threadIdx.x >> 6;
    const long long t_start = clock64();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x & 3); b[j] = (__bf16)(float)(dummy + j); }
    float v[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { v[i] = -0.001f * (threadIdx.x + i + dummy); p[i] = f32x2{v[i], v[i] + 1.f}; }
    const f32x2 c = {0.999f + dummy, 0.001f};

    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((MODE >= 10 && (MODE - 10) / 2 == OP_LDS) || MODE == 66 || MODE == 67 || MODE == 76 || MODE == 77) {
        for (int i = threadIdx.x; i < 16384; i += 512) ((float *)smem)[i] = 0.001f * i;
        __syncthreads();
    }
    const bool is_m = MODE == 0 || ((MODE == 2 || MODE == 3 || MODE == 7) && wave < 4) || (MODE >= 10 && MODE < 28 && (MODE & 1) && wave < 4);
    const bool is_v = MODE == 1 || ((MODE == 2 || MODE == 3) && wave >= 4);
    const bool is_p = MODE == 6 || (MODE == 7 && wave >= 4);
    if (MODE == 3 && wave < 4) __builtin_amdgcn_s_setprio(3);
    if (is_m) {
#pragma unroll 1
        for (int it = 0; it < NM / 4; ++it) mfma_block(acc, a, b);
    } else if (is_v) {
#pragma unroll 1
        for (int it = 0; it < NV / 8; ++it) exp_block(v);
    } else if (is_p) {
#pragma unroll 1
        for (int it = 0; it < NV * 4 / 8; ++it) fma_block(p, c);
    } else if (MODE == 28 || MODE == 32 || MODE == 34) {   // same wavefront: per MFMA, vector-ALU work in its shadow
#pragma unroll 1
        for (int it = 0; it < NM / 8; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                if (MODE == 28) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j]) : "v"(c[0]));
                } else if (MODE == 32) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(c));
                } else {
#pragma unroll
                    for (int j = 0; j < 3; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j]) : "v"(c[0]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (MODE == 29 || MODE == 30 || MODE == 31 || MODE == 33) {
        if (wave < 4) {
            if (MODE == 29) {
#pragma unroll 1
                for (int it = 0; it < NM / 4; ++it) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                        asm volatile("s_nop 7\n s_nop 7\n s_nop 7");
                    }
                }
            } else if (MODE != 31) {
#pragma unroll 1
                for (int it = 0; it < NM / 4; ++it) mfma_block(acc, a, b);
            }
        } else if (MODE != 33) {
            if (MODE == 30) __builtin_amdgcn_s_setprio(3);
            op_loop<OP_FMA>(v, p, c, smem, threadIdx.x & 63);
        }
    } else if (MODE >= 10 && MODE < 28) {
        op_loop<(MODE - 10) / 2 < NOPS ? (MODE - 10) / 2 : 0>(v, p, c, smem, threadIdx.x & 63);
    } else if (MODE == 4) {
#pragma unroll 1
        for (int it = 0; it < NM / 8; ++it) {   // per MFMA: one exp in its shadow
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                v[2 * i] = __builtin_amdgcn_exp2f(v[2 * i]);
                v[2 * i + 1] = __builtin_amdgcn_exp2f(v[2 * i + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (MODE >= 50 && MODE <= 56) {
        // single-wavefront issue rates (wavefronts 4-7 only, one per SIMD): NV * 4 instructions each
        if (wave >= 4) {
            int s0 = dummy, s1 = dummy + 1, s2 = dummy + 2, s3 = dummy + 3;
#pragma unroll 1
            for (int it = 0; it < NV * 4 / 8; ++it) {
                if (MODE == 50) {   // 8 scalar adds
                    asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                                 "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1"
                                 : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
                } else if (MODE == 51) {   // 4 x (v_fma, s_add)
                    asm volatile("v_fma_f32 %4, %4, %8, %8\n s_add_u32 %0, %0, 1\n v_fma_f32 %5, %5, %8, %8\n s_add_u32 %1, %1, 1\n"
                                 "v_fma_f32 %6, %6, %8, %8\n s_add_u32 %2, %2, 1\n v_fma_f32 %7, %7, %8, %8\n s_add_u32 %3, %3, 1"
                                 : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "v"(c[0]) : "scc");
                } else if (MODE == 52) {   // 8 dependent v_fma (one chain)
                    asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n"
                                 "v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1"
                                 : "+v"(v[0]) : "v"(c[0]));
                } else if (MODE == 53) {   // 8 v_fma, 2 independent chains
                    asm volatile("v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2\n v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2\n"
                                 "v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2\n v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2"
                                 : "+v"(v[0]), "+v"(v[1]) : "v"(c[0]));
                } else if (MODE == 54) {   // 8 v_mov from an immediate (no register read)
                    asm volatile("v_mov_b32 %0, 1.0\n v_mov_b32 %1, 1.0\n v_mov_b32 %2, 1.0\n v_mov_b32 %3, 1.0\n"
                                 "v_mov_b32 %4, 1.0\n v_mov_b32 %5, 1.0\n v_mov_b32 %6, 1.0\n v_mov_b32 %7, 1.0"
                                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]));
                } else if (MODE == 55) {   // 8 s_nop 0
                    asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
                } else {                   // 56: 8 independent v_fma, e32-style 2-operand v_mul (shorter encoding)
                    asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                                 "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7"
                                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(c[0]));
                }
            }
            v[0] += (float)(s0 + s1 + s2 + s3);
        }
    } else if (MODE >= 46 && MODE <= 49) {
        // MFMA wavefronts whose MFMAs form NCH dependent chains (46: 2 chains, 47: 1 chain, 48: 3 chains, 49: 2 chains with
        // an LDS read and wait per 6 MFMAs) next to the packed silu mix
        if (wave < 4) {
            constexpr int NCH = MODE == 47 ? 1 : MODE == 48 ? 3 : 2;
#pragma unroll 1
            for (int it = 0; it < NM / 12; ++it) {
#pragma unroll
                for (int i = 0; i < 12; ++i) acc[i % NCH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i % NCH], 0, 0, 0);
                if (MODE == 49) {
                    typedef __attribute__((ext_vector_type(4))) float f4;
                    f4 t = *(const f4 *)(smem + ((threadIdx.x * 16 + it * 1024) & 0xFFF0));
                    a[0] = (__bf16)t[0];
                }
            }
        } else {
#pragma unroll 1
            for (int it = 0; it < NV / 8; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x2 z = p[i] + c, t = z * c;
                    f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                    e = e + f32x2{1.f, 1.f};
                    f32x2 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
                    p[i] = z * r;
                }
            }
        }
    } else if (MODE >= 40 && MODE <= 45) {
        // 40: four wavefronts (4-7) run the packed silu mix alone; 41: the same next to four MFMA wavefronts;
        // 42 / 43: the same with the plain (unpacked) mix; 44 / 45: packed mix in all eight / MFMA in all eight + ...
        const bool with_m = MODE & 1;
        if (wave < 4) {
            if (with_m) {
#pragma unroll 1
                for (int it = 0; it < NM / 4; ++it) mfma_block(acc, a, b);
            }
        } else {
            const bool packed = MODE < 42;
#pragma unroll 1
            for (int it = 0; it < NV / 8; ++it) {
                if (packed) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {   // 4 x (2 values): add, mul, 2 exp, add, 2 rcp, mul, cvt
                        f32x2 z = p[i] + c, t = z * c;
                        f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                        e = e + f32x2{1.f, 1.f};
                        f32x2 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
                        p[i] = z * r;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float z, t, e, r;
                        asm volatile("v_add_f32 %0, %1, %2" : "=v"(z) : "v"(v[i]), "v"(c[0]));
                        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(z), "v"(c[0]));
                        e = __builtin_amdgcn_exp2f(t);
                        asm volatile("v_add_f32 %0, 1.0, %1" : "=v"(e) : "v"(e));
                        r = __builtin_amdgcn_rcpf(e);
                        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(z), "v"(r));
                    }
                }
            }
        }
    } else if (MODE == 35 || MODE == 36) {
        if (MODE == 35 && wave >= 4) __builtin_amdgcn_s_setprio(2);
        if (MODE == 36 && wave >= 4) {
#pragma unroll 1
            for (int it = 0; it < NV / 16 / 8; ++it) exp_block(v);
        }
#pragma unroll 1
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll 1
            for (int it = 0; it < NM / 8 / 8; ++it) mfma_block(acc, a, b);
            if (MODE == 36 && wave >= 4 && rep == 7) break;
#pragma unroll 1
            for (int it = 0; it < NV / 16 / 8; ++it) exp_block(v);
        }
    } else if (MODE >= 60 && MODE <= 79) {
        // Weight-stream pressure: every KM MFMAs take a fresh A operand (16 bytes per lane = 1 KB per wavefront) from a
        // window of WIN bytes of global memory that all wavefronts of the CU walk in step (the first to ask misses the
        // vector L1 when WIN exceeds it, the others hit), requested 4 loads ahead (modes 70-79 repeat 60-69 with 8 ahead).  At KM = 2 and one MFMA wavefront
        // per SIMD a CU at full MFMA rate asks for 4 KB per 64 cycles = 64 bytes per cycle.
        constexpr int M = MODE >= 70 ? MODE - 10 : MODE, RD = MODE >= 70 ? 8 : 3;
        constexpr int KM = (M == 62 || M == 64 || M == 67) ? 4 : (M == 68 ? 1 : 2);
        constexpr unsigned WIN = M == 60 ? 16384u : 65536u;
        constexpr bool ALL8 = M == 63 || M == 64;
        constexpr bool LDSW = M == 66 || M == 67;   // the same stream from LDS instead
        if (ALL8 || wave < 4) {
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(out + 65536), 0, WIN, 0x00020000);
            // Lean issue stream (a wavefront issues one instruction per ~8 cycles, so everything beside the MFMAs must fit
            // 3 slots per MFMA): the lane offset is fixed, steps of a group of four ride in the immediate offset, the scalar
            // base advances once per eight steps, and the MFMAs read the ring slot in place (the refill is issued after them).
            const unsigned voff = (threadIdx.x & 63) * 16;
            u32x4 ring[8];
            unsigned sbase = 0;
            auto fetch = [&](unsigned sb, int r) {
                if constexpr (LDSW) return *(const u32x4 *)(smem + voff + ((sb + r * 1024) & (WIN - 1)));
                else return __builtin_amdgcn_raw_buffer_load_b128(rs, voff + (r & 3) * 1024, sb + (r >> 2) * 4096, 0);
            };
            static_assert(RD == 3 || RD == 8, "ring");
            constexpr int AHEAD = RD == 8 ? 8 : 4;   // loads in flight: a full ring, or half of it
#pragma unroll
            for (int r = 0; r < AHEAD; ++r) ring[r] = fetch(sbase, r);
#pragma unroll 1
            for (int it = 0; it < NM / KM / 8; ++it) {
                const unsigned snext = (sbase + 8192) & (WIN - 1);
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const bf16x8 w = __builtin_bit_cast(bf16x8, ring[r]);
#pragma unroll
                    for (int i = 0; i < KM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, b, acc[i], 0, 0, 0);
                    constexpr int AH = AHEAD;
                    if (AH == 8) ring[r] = fetch(snext, r);
                    else ring[(r + 4) & 7] = r < 4 ? fetch(sbase, r + 4) : fetch(snext, r - 4);
                    __builtin_amdgcn_sched_barrier(0);
                }
                sbase = snext;
            }
        }
    } else if (MODE == 5) {
#pragma unroll 1
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll 1
            for (int it = 0; it < NM / 8 / 8; ++it) mfma_block(acc, a, b);
#pragma unroll 1
            for (int it = 0; it < NV / 16 / 8; ++it) exp_block(v);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1];
    asm volatile("" ::"v"(s));
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) ((long long *)out)[blockIdx.x * 8 + wave] = clock64() - t_start;
    if (s == 123.456f) out[threadIdx.x + 8192] = s;
}

template <int MODE>
static void run(const char *what, float *d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * 8;
    const int lds = 100 * 1024;   // one workgroup per CU
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), lds, 0, d, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), lds, 0, d, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[256 * 8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double lo = 0, hi = 0;
    for (int i = 0; i < 256; ++i)
        for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) += (double)h[i * 8 + w] / (256 * 4);
    printf("mode %d  %-64s %8.1f us per workgroup-round   cycles/wavefront: 0-3 %9.0f  4-7 %9.0f\n", MODE, what, 1e3 * ms / 8, lo, hi);
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    float *d;
    hipMalloc(&d, 1 << 20);
    hipMemset(d, 0x3c, 1 << 20);
    printf("NM = %d MFMAs (x32 cycles = %d), NV = %d exps (x16 cycles = %d)\n", NM, NM * 32, NV, NV * 16);
    run<0>("8 wavefronts MFMA (2 per SIMD)", d);
    run<1>("8 wavefronts exp", d);
    run<2>("4 MFMA + 4 exp (one of each per SIMD)", d);
    run<3>("4 MFMA (priority 3) + 4 exp", d);
    run<4>("8 wavefronts, MFMA and exp interleaved (half of each)", d);
    run<5>("8 wavefronts, MFMA phase then exp phase x8 (half of each)", d);
    run<6>("8 wavefronts packed fma (4x the instructions of mode 1)", d);
    run<7>("4 MFMA + 4 packed fma", d);
    run<10>("8 x v_fma_f32", d); run<11>("4 MFMA + 4 v_fma_f32", d);
    run<12>("8 x v_pk_mul_f32", d); run<13>("4 MFMA + 4 v_pk_mul_f32", d);
    run<14>("8 x v_pk_add_f32", d); run<15>("4 MFMA + 4 v_pk_add_f32", d);
    run<16>("8 x v_cvt_pk_bf16_f32", d); run<17>("4 MFMA + 4 v_cvt_pk_bf16_f32", d);
    run<18>("8 x v_mul_f32", d); run<19>("4 MFMA + 4 v_mul_f32", d);
    run<20>("8 x v_rcp_f32 (NV)", d); run<21>("4 MFMA + 4 v_rcp_f32 (NV)", d);
    run<22>("8 x ds_read_b128 (NV)", d); run<23>("4 MFMA + 4 ds_read_b128 (NV)", d);
    run<24>("8 x v_pk_fma_f16", d); run<25>("4 MFMA + 4 v_pk_fma_f16", d);
    run<26>("8 x v_mov_b32", d); run<27>("4 MFMA + 4 v_mov_b32", d);
    run<28>("8 wavefronts: NM/2 x (MFMA + 6 v_fma_f32)  [ideal 64]", d);
    run<34>("8 wavefronts: NM/2 x (MFMA + 3 v_fma_f32)  [ideal 64]", d);
    run<32>("8 wavefronts: NM/2 x (MFMA + 3 v_pk_mul_f32)", d);
    run<29>("4 MFMA padded with 24 s_nop cycles + 4 v_fma_f32", d);
    run<30>("4 MFMA + 4 v_fma_f32 at priority 3", d);
    run<31>("4 v_fma_f32 alone (one wavefront per SIMD)", d);
    run<33>("4 MFMA alone (one wavefront per SIMD)", d);
    run<5>("8 wavefronts, MFMA phase then exp phase x8 (half of each)", d);
    run<35>("as mode 5, wavefronts 4-7 at priority 2", d);
    run<36>("as mode 5, wavefronts 4-7 start with the exp phase", d);
    run<40>("4 x packed silu mix alone (NV values x 8 instr)", d);
    run<41>("4 MFMA + 4 packed silu mix", d);
    run<42>("4 x plain silu mix alone", d);
    run<43>("4 MFMA + 4 plain silu mix", d);
    run<46>("4 MFMA (2 dependent chains) + 4 packed silu mix", d);
    run<47>("4 MFMA (1 dependent chain) + 4 packed silu mix", d);
    run<48>("4 MFMA (3 dependent chains) + 4 packed silu mix", d);
    run<49>("4 MFMA (2 chains, LDS read per 12) + 4 packed silu mix", d);
    run<50>("one wavefront per SIMD: s_add_u32 (cycles / NV*4 = per instruction)", d);
    run<51>("one wavefront per SIMD: v_fma / s_add alternating", d);
    run<52>("one wavefront per SIMD: v_fma, one dependent chain", d);
    run<53>("one wavefront per SIMD: v_fma, two chains", d);
    run<54>("one wavefront per SIMD: v_mov from immediate", d);
    run<55>("one wavefront per SIMD: s_nop 0", d);
    run<56>("one wavefront per SIMD: v_mul_f32 e32, 8 chains", d);
    run<33>("4 MFMA alone (one wavefront per SIMD)", d);
    run<60>("4 x (A operand from a 16 KB window per 2 MFMAs)  [ideal 59]", d);
    run<61>("4 x (A operand from a 64 KB window per 2 MFMAs)", d);
    run<62>("4 x (A operand from a 64 KB window per 4 MFMAs)", d);
    run<68>("4 x (A operand from a 64 KB window per MFMA)", d);
    run<63>("8 x (A operand from a 64 KB window per 2 MFMAs)  [ideal 118]", d);
    run<64>("8 x (A operand from a 64 KB window per 4 MFMAs)", d);
    run<66>("4 x (A operand from LDS per 2 MFMAs)", d);
    run<67>("4 x (A operand from LDS per 4 MFMAs)", d);
    run<70>("RD 8: 4 x (A operand from a 16 KB window per 2 MFMAs)", d);
    run<71>("RD 8: 4 x (A operand from a 64 KB window per 2 MFMAs)", d);
    run<72>("RD 8: 4 x (A operand from a 64 KB window per 4 MFMAs)", d);
    run<78>("RD 8: 4 x (A operand from a 64 KB window per MFMA)", d);
    run<73>("RD 8: 8 x (A operand from a 64 KB window per 2 MFMAs)", d);
    run<74>("RD 8: 8 x (A operand from a 64 KB window per 4 MFMAs)", d);
    run<76>("RD 8: 4 x (A operand from LDS per 2 MFMAs)", d);
    (void)op_names;
    hipFree(d);
    return 0;
}
