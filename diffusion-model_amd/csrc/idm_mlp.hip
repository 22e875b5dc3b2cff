// Dense spectral score model (b_diffusion_model_fft_2d_ising.py:105-114) and its reverse sampler
// (:19-38) in the canonical float32 arithmetic: one kernel launch per reverse step runs all Dense
// layers of S samples out of LDS and applies the update.
//
// Canonical order: every output is one fmaf chain from +0 over ascending input index (the
// appended t_norm is the last input of layer 0), bias after; tanh(z) = copysign((1-e)/(1+e), z)
// with e = cexp(-2|z|); update ((c1*x) + (beta*score)) + (z*sd) with constants rounded on the host.
// This file must be compiled with -ffp-contract=off.
#include "../../include/isingdiff.h"
#include "idm_internal.h"
#include "idm_device.h"
#include "idm_canon_math.h"

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr int MAXW = 448;   // widest layer the kernel holds in LDS (the saved model: 441 -> 440)
constexpr int S = 8;        // samples per workgroup

DEVI float ctanh(float z)
{
    const float a = fabsf(z);
    const float e = cexp(__fmul_rn(-2.0f, a));
    const float r = __fdiv_rn(__fsub_rn(1.0f, e), __fadd_rn(1.0f, e));
    return copysignf(r, z);
}

struct MlpArgs {
    int n_layers;
    int in_dim[IDM_MLP_MAXLAYERS], out_dim[IDM_MLP_MAXLAYERS], act[IDM_MLP_MAXLAYERS];
    const float *w[IDM_MLP_MAXLAYERS], *b[IDM_MLP_MAXLAYERS];
    float *x;               // (B, n) state (read; updated in place when do_update)
    float *score_out;       // optional (B, n)
    const float *noise;     // optional injected standard normals (B, n)
    int64_t B;
    int n;
    float t_norm, c1, beta, sd;
    int do_update, add_noise;
    PhiloxArgs ph;
};

__global__ __launch_bounds__(MAXW) void mlp_step_kernel(MlpArgs a)
{
    __shared__ float buf[2][S][MAXW];
    const int tid = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * S;
    const int ns = (int)min((int64_t)S, a.B - b0);
    for (int i = tid; i < S * MAXW; i += MAXW) {
        const int s = i / MAXW, k = i - s * MAXW;
        float v = 0.0f;
        if (s < ns) {
            if (k < a.n) v = a.x[(b0 + s) * a.n + k];
            else if (k == a.n) v = a.t_norm;
        }
        buf[0][s][k] = v;
    }
    __syncthreads();
    int cur = 0;
    for (int l = 0; l < a.n_layers; ++l) {
        const int nin = a.in_dim[l], nout = a.out_dim[l];
        if (tid < nout) {
            float acc[S];
#pragma unroll
            for (int s = 0; s < S; ++s) acc[s] = 0.0f;
            const float *w = a.w[l] + tid;
            for (int k = 0; k < nin; ++k) {
                const float wv = w[(size_t)k * nout];
#pragma unroll
                for (int s = 0; s < S; ++s) acc[s] = __fmaf_rn(buf[cur][s][k], wv, acc[s]);
            }
            const float bv = a.b[l][tid];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float z = __fadd_rn(acc[s], bv);
                buf[cur ^ 1][s][tid] = a.act[l] ? ctanh(z) : z;
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    // buf[cur][s][0..n) = score
    const int nq = a.n >> 2;   // n is a multiple of 4 (checked on the host)
    for (int i = tid; i < ns * nq; i += MAXW) {
        const int s = i / nq, q = i - s * nq;
        const int64_t e = (b0 + s) * a.n + q * 4;
        const f32x4 sc = *(const f32x4 *)&buf[cur][s][q * 4];
        if (a.score_out) *(f32x4 *)(a.score_out + e) = sc;
        if (!a.do_update) continue;
        f32x4 xv = *(const f32x4 *)(a.x + e);
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.add_noise) {
            if (a.noise) {
                const f32x4 zv = *(const f32x4 *)(a.noise + e);
#pragma unroll
                for (int j = 0; j < 4; ++j) z[j] = zv[j];
            } else {
                normal4(a.ph.seed, a.ph.first_global_index + b0 + s, (uint32_t)q, a.ph.t, z);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = __fadd_rn(__fmul_rn(a.c1, xv[j]), __fmul_rn(a.beta, sc[j]));
            if (a.add_noise) v = __fadd_rn(v, __fmul_rn(z[j], a.sd));
            xv[j] = v;
        }
        *(f32x4 *)(a.x + e) = xv;
    }
}

// ---------------------------------------------------------------------------------------------
// bf16 MFMA path.  A workgroup (eight wavefronts) owns 64 samples (two MFMA N tiles; every weight fragment
// fetched from L2 feeds both, which halves the L2 -> CU weight traffic that bounds this kernel): D[cout][sample] += W^T[cout][k] * act[k][sample]
// on v_mfma_f32_32x32x16_bf16, activations of the current layer in LDS as [sample][k] bf16 (row stride
// 2*KP + 16 bytes: conflict-free 16-byte fragment reads), weights streamed from L2 as 1 KB fragments packed on
// the host as [k/16][cout/32][lane][8].  A wavefront computes two cout tiles x two sample tiles at a time.  The t_norm input of layer 0 is applied in float32 in the epilogue (bias + w_t * t_norm), the
// state stays float32 in global memory and is updated by the lanes that hold the last layer's outputs;
// noise counters are those of the float32 kernel, so both paths draw the same normals.
// ---------------------------------------------------------------------------------------------
// Two instances.  <4, true>: 128 samples per workgroup (every weight fragment feeds four sample tiles: half the weight
// traffic per sample of the other instance), one activation buffer rewritten in place - every wavefront owns at most one
// pair of cout tiles, keeps the layer's outputs in its accumulators, and all of them are stored only after a barrier that
// follows the last read of the inputs; needs every layer to have at most 16 cout tiles (512 outputs).  <2, false>: 64
// samples, two buffers, any width.
constexpr int FWG = 512;

struct MlpFastArgs {
    int n_layers;
    int kt[IDM_MLP_MAXLAYERS], mt[IDM_MLP_MAXLAYERS], out_dim[IDM_MLP_MAXLAYERS], act[IDM_MLP_MAXLAYERS];
    const bf16x8 *w[IDM_MLP_MAXLAYERS];
    const float *b[IDM_MLP_MAXLAYERS];
    const float *wt0;       // layer 0's weights of the t_norm input (out_dim[0] floats)
    float *x;
    float *score_out;
    const float *noise;
    int64_t B;
    int n, kp;              // kp: padded width of the LDS rows (multiple of 64)
    float t_norm, c1, beta, sd;
    int do_update, add_noise;
    PhiloxArgs ph;
};

DEVI float fast_tanh(float z)
{
    // 1 - 2 / (1 + e^{2z}); saturates correctly at both ends
    const float e = __builtin_amdgcn_exp2f(z * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}

template <int FNT, bool INPLACE>
__global__ __launch_bounds__(FWG, 1) void mlp_step_bf16_kernel(MlpFastArgs a)
{
    constexpr int FS = 32 * FNT;        // samples per workgroup
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = lane & 31, h = lane >> 5;
    const int str = a.kp * 2 + 16;
    // the two activation buffers are addressed as smem + index * bufsz (plain arithmetic on the LDS base keeps
    // the accesses ds_read / ds_write; a pointer picked from an array degrades them to flat accesses)
    const int bufsz = FS * str;
    const int64_t b0 = (int64_t)blockIdx.x * FS;
    const int ns = (int)min((int64_t)FS, a.B - b0);

    // stage-in: bf16(x) as [sample][k], zero beyond n and beyond the batch
    const int qpr = a.kp >> 2;   // float4 quads per padded row
    for (int i = tid; i < FS * qpr; i += FWG) {
        const int s = i / qpr, q = i - s * qpr;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (s < ns && q * 4 < a.n) v = *(const f32x4 *)(a.x + (b0 + s) * a.n + q * 4);
        *(bf16x4 *)(smem + s * str + q * 8) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    }
    __syncthreads();

    int cur = 0;
    for (int l = 0; l < a.n_layers; ++l) {
        const int KT = a.kt[l], MT = a.mt[l];
        const bool last = l == a.n_layers - 1;
        const int nxt = INPLACE ? 0 : cur ^ 1;
        const char *src = smem + cur * bufsz + nl * str + h * 16;
        char *dst = smem + nxt * bufsz + nl * str;
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8 *>(a.w[l]), 0, (unsigned)(KT * MT * 1024), 0x00020000);
        f32x16 acc[2][FNT];
        auto compute = [&](int m0) {
            const bool two = m0 + 1 < MT;
            unsigned wo = (unsigned)((m0 * 64 + lane) * 16);
            asm volatile("" : "+v"(wo));   // keeps the unrolled fetches inside the loop
            const unsigned wo1 = two ? wo + 1024u : 0x80000000u;   // odd tile count: the second tile reads zeros
            const unsigned kstep = (unsigned)MT * 1024u;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < FNT; ++q)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[t][q][i] = 0.0f;
            auto wload = [&](unsigned off, int kt) {
                return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, off, (unsigned)kt * kstep, 0));
            };
            // RING-slot register ring: the fragments of k-step kt + RING are requested as soon as those of kt are
            // used.  KT is a multiple of RING and the loop body has no branches (requests past the end re-read the
            // last fragments), so the waits inside it stay counted instead of draining the queue.
            constexpr int RING = FNT > 2 ? 2 : 4;   // (measured: 2, 4 and 7 slots run alike; deeper rings spill beside 128 accumulators)
            bf16x8 w0[RING], w1[RING];
#pragma unroll
            for (int j = 0; j < RING; ++j) { w0[j] = wload(wo, j); w1[j] = wload(wo1, j); }
            for (int kt = 0; kt < KT; kt += RING) {
#pragma unroll
                for (int j = 0; j < RING; ++j) {
#pragma unroll
                    for (int q = 0; q < FNT; ++q) {
                        const bf16x8 f = *(const bf16x8 *)(src + q * 32 * str + (kt + j) * 32);
                        acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0[j], f, acc[0][q], 0, 0, 0);
                        acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[j], f, acc[1][q], 0, 0, 0);
                    }
                    const int nk = min(kt + RING + j, KT - 1);
                    w0[j] = wload(wo, nk);
                    w1[j] = wload(wo1, nk);
                    __builtin_amdgcn_sched_barrier(0);   // keeps each k-step's requests in place (the scheduler would sink them all)
                }
            }
        };
        auto epilogue = [&](int m0) {
            const bool two = m0 + 1 < MT;
            // lane (sample nl, half h) holds couts 32*m + 8g + 4h + i
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t == 1 && !two) break;
#pragma unroll
              for (int q = 0; q < FNT; ++q) {
                __builtin_amdgcn_sched_barrier(0);   // one tile at a time: hoisting every tile's loads ahead spills
                const f32x16 &ac = acc[t][q];
                const int sn = q * 32 + nl;        // sample within the workgroup
                char *dstq = dst + q * 32 * str;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = (m0 + t) * 32 + 8 * g + 4 * h;
                    f32x4 z = {ac[4 * g], ac[4 * g + 1], ac[4 * g + 2], ac[4 * g + 3]};
                    const bool inr = c < a.out_dim[l];   // out_dim is a multiple of 4
                    if (inr) {
                        z += *(const f32x4 *)(a.b[l] + c);
                        if (l == 0) z += *(const f32x4 *)(a.wt0 + c) * a.t_norm;
                    }
                    if (!last) {
                        if (a.act[l]) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) z[i] = fast_tanh(z[i]);
                        }
                        if (!inr) z = f32x4{0.f, 0.f, 0.f, 0.f};
                        *(bf16x4 *)(dstq + c * 2) = bf16x4{(__bf16)z[0], (__bf16)z[1], (__bf16)z[2], (__bf16)z[3]};
                    } else if (inr && sn < ns) {
                        const int64_t e = (b0 + sn) * a.n + c;
                        if (a.score_out) *(f32x4 *)(a.score_out + e) = z;
                        if (a.do_update) {
                            f32x4 xv = *(const f32x4 *)(a.x + e);
                            float zz[4] = {0.f, 0.f, 0.f, 0.f};
                            if (a.add_noise) {
                                if (a.noise) {
                                    const f32x4 zv = *(const f32x4 *)(a.noise + e);
#pragma unroll
                                    for (int i = 0; i < 4; ++i) zz[i] = zv[i];
                                } else {
                                    // (the counter is made opaque here: the draw depends on nothing the layers compute, and
                                    // the optimiser would otherwise hoist all of a wavefront's draws to the top of the
                                    // kernel and hold them in registers through every layer)
                                    uint32_t quad = (uint32_t)(c >> 2);
                                    asm volatile("" : "+v"(quad));
                                    normal4(a.ph.seed, a.ph.first_global_index + b0 + sn, quad, a.ph.t, zz);
                                }
                            }
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                float v = a.c1 * xv[i] + a.beta * z[i];
                                if (a.add_noise) v += zz[i] * a.sd;
                                xv[i] = v;
                            }
                            *(f32x4 *)(a.x + e) = xv;
                        }
                    }
                }
              }
            }
        };
        if constexpr (INPLACE) {
            const int m0 = wave * 2;             // (host-checked: MT <= 16, one pair of cout tiles per wavefront at most)
            if (m0 < MT) compute(m0);
            __syncthreads();                     // every wavefront has read the layer's inputs
            if (!last) {
                if (m0 < MT) epilogue(m0);
            } else {
                // Last layer: an accumulator lane holds four outputs of ONE sample, so updating the state from the
                // accumulators touches global memory in 32-byte pieces 4 n bytes apart.  The float32 scores cross LDS
                // instead (the activation buffer is free now), 64 samples per round as [sample][n + 4], and the
                // workgroup updates x a quad per thread, consecutive threads on consecutive quads of a row: coalesced
                // reads and writes, one Philox call per quad with the counters of the float32 kernel.  Same expressions,
                // hence the same bits, as the accumulator-side update of the 64-sample instance.
                float *stage = (float *)smem;
                const int sstr = a.n + 4, nq = a.n >> 2;
                const int dS = FWG / nq, dQ = FWG - dS * nq;
#pragma unroll
                for (int r = 0; r < FNT / 2; ++r) {
                    if (m0 < MT) {
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            if (t == 1 && m0 + 1 >= MT) break;
#pragma unroll
                            for (int qq = 0; qq < 2; ++qq) {
                                const f32x16 &ac = acc[t][2 * r + qq];
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    const int c = (m0 + t) * 32 + 8 * g + 4 * h;
                                    if (c < a.out_dim[l]) {
                                        f32x4 z = {ac[4 * g], ac[4 * g + 1], ac[4 * g + 2], ac[4 * g + 3]};
                                        z += *(const f32x4 *)(a.b[l] + c);
                                        if (l == 0) z += *(const f32x4 *)(a.wt0 + c) * a.t_norm;
                                        *(f32x4 *)(stage + (qq * 32 + nl) * sstr + c) = z;
                                    }
                                }
                            }
                        }
                    }
                    __syncthreads();
                    const int rows = min(64, ns - 64 * r);
                    int sr = tid / nq, qd = tid - sr * nq;
                    for (int idx = tid; idx < rows * nq; idx += FWG) {
                        const f32x4 z = *(const f32x4 *)(stage + sr * sstr + 4 * qd);
                        const int sn = 64 * r + sr;
                        const int64_t e = (b0 + sn) * a.n + 4 * qd;
                        if (a.score_out) *(f32x4 *)(a.score_out + e) = z;
                        if (a.do_update) {
                            f32x4 xv = *(const f32x4 *)(a.x + e);
                            float zz[4] = {0.f, 0.f, 0.f, 0.f};
                            if (a.add_noise) {
                                if (a.noise) {
                                    const f32x4 zv = *(const f32x4 *)(a.noise + e);
#pragma unroll
                                    for (int i = 0; i < 4; ++i) zz[i] = zv[i];
                                } else {
                                    normal4(a.ph.seed, a.ph.first_global_index + b0 + sn, (uint32_t)qd, a.ph.t, zz);
                                }
                            }
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                float v = a.c1 * xv[i] + a.beta * z[i];
                                if (a.add_noise) v += zz[i] * a.sd;
                                xv[i] = v;
                            }
                            *(f32x4 *)(a.x + e) = xv;
                        }
                        sr += dS;
                        qd += dQ;
                        if (qd >= nq) { qd -= nq; ++sr; }
                    }
                    if (r + 1 < FNT / 2) __syncthreads();   // the staged scores are read before the next round overwrites them
                }
            }
        } else {
            for (int m0 = wave * 2; m0 < MT; m0 += 2 * (FWG / 64)) {
                compute(m0);
                epilogue(m0);
            }
        }
        // layers narrower than the padded row: the k range [32*MT, kp) of the next input must read as zero
        if (!last) {
            const int c0 = MT * 32;
            for (int i = tid; i < FS * ((a.kp - c0) >> 2); i += FWG) {
                const int per = (a.kp - c0) >> 2, s = i / per, q = i - s * per;
                *(bf16x4 *)(smem + nxt * bufsz + s * str + (c0 + q * 4) * 2) = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            }
        }
        __syncthreads();
        cur = nxt;
    }
}

}  // namespace

struct idm_mlp {
    int device = 0;
    int n_layers = 0, n = 0;
    int in_dim[IDM_MLP_MAXLAYERS] = {}, out_dim[IDM_MLP_MAXLAYERS] = {}, act[IDM_MLP_MAXLAYERS] = {};
    float *w[IDM_MLP_MAXLAYERS] = {}, *b[IDM_MLP_MAXLAYERS] = {};
    // bf16 path: packed weights, per-layer k-steps / cout tiles, layer 0's t_norm weights, padded row width
    void *wpk[IDM_MLP_MAXLAYERS] = {};
    int kt[IDM_MLP_MAXLAYERS] = {}, mt[IDM_MLP_MAXLAYERS] = {};
    float *wt0 = nullptr;
    int kp = 0;
    std::vector<void *> owned;
};

namespace {

#define MLPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return idm_set_error(IDM_ERR_HIP, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
    } while (0)

void fill_args(const idm_mlp *m, MlpArgs &a)
{
    a.n_layers = m->n_layers;
    for (int i = 0; i < m->n_layers; ++i) {
        a.in_dim[i] = m->in_dim[i]; a.out_dim[i] = m->out_dim[i]; a.act[i] = m->act[i];
        a.w[i] = m->w[i]; a.b[i] = m->b[i];
    }
    a.n = m->n;
}

void fill_fast_args(const idm_mlp *m, MlpFastArgs &a)
{
    a.n_layers = m->n_layers;
    for (int i = 0; i < m->n_layers; ++i) {
        a.kt[i] = m->kt[i]; a.mt[i] = m->mt[i]; a.out_dim[i] = m->out_dim[i]; a.act[i] = m->act[i];
        a.w[i] = (const bf16x8 *)m->wpk[i]; a.b[i] = m->b[i];
    }
    a.wt0 = m->wt0; a.n = m->n; a.kp = m->kp;
}

// The bf16 path needs every layer width to be a multiple of 4 (float4 / 4 x bf16 epilogue stores).
bool fast_ok(const idm_mlp *m)
{
    for (int i = 0; i < m->n_layers; ++i)
        if (m->out_dim[i] & 3) return false;
    return true;
}

int launch_fast(const idm_mlp *m, const MlpFastArgs &a, hipStream_t s)
{
    const size_t row = (size_t)m->kp * 2 + 16;
    bool inplace = 128 * row <= 160 * 1024;   // 114 KB for the saved model either way: one workgroup per CU
    for (int i = 0; i < m->n_layers; ++i) inplace = inplace && m->mt[i] <= 16;
    static thread_local bool configured[IDM_MAX_DEVICES][2] = {};
    const int slot = idm_device_slot();
    if (slot < 0) return idm_set_error(IDM_ERR_HIP, "no current device");
    const void *kernel = inplace ? (const void *)mlp_step_bf16_kernel<4, true> : (const void *)mlp_step_bf16_kernel<2, false>;
    if (!configured[slot][inplace]) {
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return idm_set_error(IDM_ERR_HIP, "hipFuncSetAttribute failed");
        configured[slot][inplace] = true;
    }
    if (inplace)
        hipLaunchKernelGGL((mlp_step_bf16_kernel<4, true>), dim3((unsigned)((a.B + 127) / 128)), dim3(FWG), 128 * row, s, a);
    else
        hipLaunchKernelGGL((mlp_step_bf16_kernel<2, false>), dim3((unsigned)((a.B + 63) / 64)), dim3(FWG), 2 * 64 * row, s, a);
    return IDM_OK;
}

}  // namespace

extern "C" {

int idm_mlp_create(idm_mlp **out, int device, const idm_mlp_desc *d)
{
    if (!out || !d || d->n_layers < 1 || d->n_layers > IDM_MLP_MAXLAYERS) return idm_set_error(IDM_ERR_INVALID, "bad MLP descriptor");
    for (int i = 0; i < d->n_layers; ++i) {
        if (!d->w[i] || !d->b[i] || d->in_dim[i] < 1 || d->out_dim[i] < 1) return idm_set_error(IDM_ERR_INVALID, "bad MLP layer");
        if (d->in_dim[i] > MAXW || d->out_dim[i] > MAXW) return idm_set_error(IDM_ERR_UNSUPPORTED, "MLP layer wider than 448");
        if (i > 0 && d->in_dim[i] != d->out_dim[i - 1]) return idm_set_error(IDM_ERR_INVALID, "MLP layer sizes do not chain");
    }
    const int n = d->in_dim[0] - 1;
    if (d->out_dim[d->n_layers - 1] != n || n < 4 || (n & 3))
        return idm_set_error(IDM_ERR_INVALID, "MLP must map [x | t] (n+1 inputs, n a multiple of 4) to n outputs");
    MLPCHK(hipSetDevice(device));
    idm_mlp *m = new idm_mlp();
    m->device = device; m->n_layers = d->n_layers; m->n = n;
    for (int i = 0; i < d->n_layers; ++i) {
        m->in_dim[i] = d->in_dim[i]; m->out_dim[i] = d->out_dim[i]; m->act[i] = d->act[i];
        const size_t nw = (size_t)d->in_dim[i] * d->out_dim[i];
        if (hipMalloc((void **)&m->w[i], nw * 4) != hipSuccess || hipMalloc((void **)&m->b[i], (size_t)d->out_dim[i] * 4) != hipSuccess) {
            idm_mlp_destroy(m);
            return idm_set_error(IDM_ERR_NOMEM, "MLP weight allocation failed");
        }
        m->owned.push_back(m->w[i]); m->owned.push_back(m->b[i]);
        (void)hipMemcpy(m->w[i], d->w[i], nw * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(m->b[i], d->b[i], (size_t)d->out_dim[i] * 4, hipMemcpyHostToDevice);
    }
    // bf16 path: A fragments [k/16][cout/32][lane][8] (lane: cout 32*mt + lane%32, k 16*kt + 8*(lane/32) + j),
    // zero-padded; layer 0 without its t_norm row, which is kept apart in float32
    int widest = 0;
    for (int i = 0; i < d->n_layers; ++i) widest = std::max(widest, std::max(i == 0 ? n : d->in_dim[i], d->out_dim[i]));
    m->kp = (widest + 63) / 64 * 64;
    for (int i = 0; i < d->n_layers; ++i) {
        const int kin = i == 0 ? n : d->in_dim[i], nout = d->out_dim[i];
        const int KT = (kin + 63) / 64 * 4, MT = (nout + 31) / 32;
        m->kt[i] = KT; m->mt[i] = MT;
        std::vector<uint16_t> pk((size_t)KT * MT * 512, 0);
        for (int kt = 0; kt < KT; ++kt)
            for (int mt = 0; mt < MT; ++mt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int k = 16 * kt + 8 * (lane >> 5) + j, c = 32 * mt + (lane & 31);
                        if (k < kin && c < nout) {
                            const float v = d->w[i][(size_t)k * nout + c];
                            uint32_t u; memcpy(&u, &v, 4);
                            u += 0x7FFFu + ((u >> 16) & 1u);   // round to nearest even (weights are finite)
                            pk[(((size_t)kt * MT + mt) * 64 + lane) * 8 + j] = (uint16_t)(u >> 16);
                        }
                    }
        if (hipMalloc(&m->wpk[i], pk.size() * 2) != hipSuccess) {
            idm_mlp_destroy(m);
            return idm_set_error(IDM_ERR_NOMEM, "MLP weight allocation failed");
        }
        m->owned.push_back(m->wpk[i]);
        (void)hipMemcpy(m->wpk[i], pk.data(), pk.size() * 2, hipMemcpyHostToDevice);
    }
    if (hipMalloc((void **)&m->wt0, (size_t)d->out_dim[0] * 4) != hipSuccess) {
        idm_mlp_destroy(m);
        return idm_set_error(IDM_ERR_NOMEM, "MLP weight allocation failed");
    }
    m->owned.push_back(m->wt0);
    (void)hipMemcpy(m->wt0, d->w[0] + (size_t)n * d->out_dim[0], (size_t)d->out_dim[0] * 4, hipMemcpyHostToDevice);
    *out = m;
    return IDM_OK;
}

int idm_mlp_destroy(idm_mlp *m)
{
    if (!m) return IDM_OK;
    (void)hipSetDevice(m->device);
    for (void *p : m->owned) (void)hipFree(p);
    delete m;
    return IDM_OK;
}

int idm_mlp_forward_precision(idm_mlp *m, const float *x_dev, float t_norm, float *score_dev, int64_t B, int precision,
                              void *stream)
{
    if (precision == IDM_FP32) return idm_mlp_forward(m, x_dev, t_norm, score_dev, B, stream);
    if (!m || !x_dev || !score_dev || B <= 0) return idm_set_error(IDM_ERR_INVALID, "bad argument");
    if (precision != IDM_BF16) return idm_set_error(IDM_ERR_INVALID, "unknown precision");
    if (!fast_ok(m)) return idm_set_error(IDM_ERR_UNSUPPORTED, "the bf16 dense path needs layer widths that are multiples of 4");
    MLPCHK(hipSetDevice(m->device));
    MlpFastArgs a = {};
    fill_fast_args(m, a);
    a.x = const_cast<float *>(x_dev); a.score_out = score_dev; a.noise = nullptr; a.B = B; a.t_norm = t_norm;
    a.do_update = 0; a.add_noise = 0;
    if (int rc = launch_fast(m, a, (hipStream_t)stream)) return rc;
    MLPCHK(hipGetLastError());
    return IDM_OK;
}

int idm_mlp_forward(idm_mlp *m, const float *x_dev, float t_norm, float *score_dev, int64_t B, void *stream)
{
    if (!m || !x_dev || !score_dev || B <= 0) return idm_set_error(IDM_ERR_INVALID, "bad argument");
    MLPCHK(hipSetDevice(m->device));
    MlpArgs a = {};
    fill_args(m, a);
    a.x = const_cast<float *>(x_dev); a.score_out = score_dev; a.noise = nullptr; a.B = B; a.t_norm = t_norm;
    a.do_update = 0; a.add_noise = 0;
    hipLaunchKernelGGL(mlp_step_kernel, dim3((unsigned)((B + S - 1) / S)), dim3(MAXW), 0, (hipStream_t)stream, a);
    MLPCHK(hipGetLastError());
    return IDM_OK;
}

int idm_mlp_sample(idm_mlp *m, const idm_sample_cfg *cfg, float *x_dev, const float *noise_dev, float *traj_dev,
                   int64_t B, void *stream)
{
    if (!m || !cfg || !x_dev || B <= 0 || cfg->T <= 0 || !cfg->betas_host) return idm_set_error(IDM_ERR_INVALID, "bad argument");
    if (cfg->precision != IDM_FP32 && cfg->precision != IDM_BF16) return idm_set_error(IDM_ERR_INVALID, "unknown precision");
    if (cfg->update_rule != IDM_UPDATE_SCORE) return idm_set_error(IDM_ERR_UNSUPPORTED, "the dense model's sampler has one update rule");
    const bool fast = cfg->precision == IDM_BF16;
    if (fast && !fast_ok(m)) return idm_set_error(IDM_ERR_UNSUPPORTED, "the bf16 dense path needs layer widths that are multiples of 4");
    MLPCHK(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    const int T = cfg->T;
    const size_t total = (size_t)B * m->n;
    PhiloxArgs ph = {cfg->seed, cfg->first_global_index, T};
    if (cfg->init_state) launch_philox_normal(x_dev, B, m->n, ph, s);
    if (traj_dev) MLPCHK(hipMemcpyAsync(traj_dev + (size_t)T * total, x_dev, total * 4, hipMemcpyDeviceToDevice, s));
    MlpArgs a = {};
    fill_args(m, a);
    a.x = x_dev; a.score_out = nullptr; a.B = B; a.do_update = 1; a.add_noise = cfg->add_noise != 0;
    MlpFastArgs fa = {};
    fill_fast_args(m, fa);
    fa.x = x_dev; fa.score_out = nullptr; fa.B = B; fa.do_update = 1; fa.add_noise = a.add_noise;
    for (int t = T - 1; t >= 0; --t) {
        const float beta = cfg->betas_host[t];
        volatile float half_beta = 0.5f * beta;
        volatile float c1 = 1.0f + half_beta;
        volatile float sd = sqrtf(beta);
        a.t_norm = (float)t / (float)T; a.c1 = c1; a.beta = beta; a.sd = sd;
        a.noise = noise_dev ? noise_dev + (size_t)t * total : nullptr;
        a.ph = ph; a.ph.t = t;
        if (fast) {
            fa.t_norm = a.t_norm; fa.c1 = a.c1; fa.beta = a.beta; fa.sd = a.sd; fa.noise = a.noise; fa.ph = a.ph;
            if (int rc = launch_fast(m, fa, s)) return rc;
        } else
            hipLaunchKernelGGL(mlp_step_kernel, dim3((unsigned)((B + S - 1) / S)), dim3(MAXW), 0, s, a);
        if (traj_dev) MLPCHK(hipMemcpyAsync(traj_dev + (size_t)t * total, x_dev, total * 4, hipMemcpyDeviceToDevice, s));
    }
    MLPCHK(hipGetLastError());
    return IDM_OK;
}

}  // extern "C"
