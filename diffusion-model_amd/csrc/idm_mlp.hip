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

}  // namespace

struct idm_mlp {
    int device = 0;
    int n_layers = 0, n = 0;
    int in_dim[IDM_MLP_MAXLAYERS] = {}, out_dim[IDM_MLP_MAXLAYERS] = {}, act[IDM_MLP_MAXLAYERS] = {};
    float *w[IDM_MLP_MAXLAYERS] = {}, *b[IDM_MLP_MAXLAYERS] = {};
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
    if (cfg->precision != IDM_FP32) return idm_set_error(IDM_ERR_UNSUPPORTED, "the dense model runs in float32 only");
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
    for (int t = T - 1; t >= 0; --t) {
        const float beta = cfg->betas_host[t];
        volatile float half_beta = 0.5f * beta;
        volatile float c1 = 1.0f + half_beta;
        volatile float sd = sqrtf(beta);
        a.t_norm = (float)t / (float)T; a.c1 = c1; a.beta = beta; a.sd = sd;
        a.noise = noise_dev ? noise_dev + (size_t)t * total : nullptr;
        a.ph = ph; a.ph.t = t;
        hipLaunchKernelGGL(mlp_step_kernel, dim3((unsigned)((B + S - 1) / S)), dim3(MAXW), 0, s, a);
        if (traj_dev) MLPCHK(hipMemcpyAsync(traj_dev + (size_t)t * total, x_dev, total * 4, hipMemcpyDeviceToDevice, s));
    }
    MLPCHK(hipGetLastError());
    return IDM_OK;
}

}  // extern "C"
