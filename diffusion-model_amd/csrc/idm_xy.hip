// 2-D XY model on the GPU: a Metropolis generator and the observables of the reference's C program
// (XY_model/xymetoropolis.c).  Angles are stored as the reference stores them: in turns, theta / 2 pi in [0, 1).
//
//   xy_metropolis_kernel   the update of xy_sampling (:169-211): propose x + (u - 1/2) * delta, wrap into [0,1),
//                          dE = -(sum_nn cos 2pi(x' - nn) - sum_nn cos 2pi(x - nn)), accept when dE <= 0 or
//                          exp(-beta dE) > u'.  The reference visits one random site per step with libc rand();
//                          here the two colours of the checkerboard are swept in turn (every site of a colour sees
//                          fixed neighbours, so the per-site move - symmetric proposal, Metropolis acceptance - keeps
//                          detailed balance and the stationary distribution is the same) with Philox uniforms, one
//                          workgroup per independent chain, the lattice in LDS for all sweeps.  float32 state,
//                          hardware cosine in turns (v_cos_f32 takes revolutions) and exp.
//   xy_observe_kernel      xyenergy (:32-50), xymagnetization (:53-69), xyvortex (:71-121) per lattice in float64:
//                          one wavefront per lattice, a shuffle reduction for the sums.
#include "../../include/isingdiff.h"
#include "idm_internal.h"
#include "idm_device.h"

namespace {

__global__ __launch_bounds__(256) void xy_metropolis_kernel(float *__restrict__ theta, int L, float beta, float delta, int n_sweeps,
                                                            PhiloxArgs ph, int sweep0)
{
    extern __shared__ float xy_lat[];
    const int64_t b = blockIdx.x;
    float *x = theta + b * L * L;
    for (int i = threadIdx.x; i < L * L; i += 256) xy_lat[i] = x[i];
    __syncthreads();
    const int64_t g = ph.first_global_index + b;
    const int half = L >> 1, nsite = L * half, npair = (nsite + 1) >> 1;
    for (int w = sweep0; w < sweep0 + n_sweeps; ++w)
        for (int c = 0; c < 2; ++c) {
            // the k-th site of colour c (row-major) draws (proposal, acceptance) = outputs (2 (k & 1), 2 (k & 1) + 1)
            // of Philox with counter (k / 2, chain lo, chain hi, 2 w + c)
            for (int q = threadIdx.x; q < npair; q += 256) {
                uint32_t r4[4];
                philox4x32_10((uint32_t)q, (uint32_t)g, (uint32_t)((uint64_t)g >> 32), (uint32_t)(2 * w + c), (uint32_t)ph.seed,
                              (uint32_t)(ph.seed >> 32), r4);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int k = 2 * q + j;
                    if (k >= nsite) break;
                    const int r = k / half, col = 2 * (k - r * half) + ((r + c) & 1);
                    const int rp = r + 1 == L ? 0 : r + 1, rm = r == 0 ? L - 1 : r - 1;
                    const int cp = col + 1 == L ? 0 : col + 1, cm = col == 0 ? L - 1 : col - 1;
                    const float v = xy_lat[r * L + col];
                    float vn = v + (u01(r4[2 * j]) - 0.5f) * delta;
                    vn -= floorf(vn);                      // into [0, 1)
                    if (vn >= 1.0f) vn = 0.0f;             // (a tiny negative value rounds up to 1)
                    const float n0 = xy_lat[rp * L + col], n1 = xy_lat[rm * L + col], n2 = xy_lat[r * L + cp], n3 = xy_lat[r * L + cm];
                    const float e_old = __builtin_amdgcn_cosf(v - n0) + __builtin_amdgcn_cosf(v - n1) + __builtin_amdgcn_cosf(v - n2) +
                                        __builtin_amdgcn_cosf(v - n3);
                    const float e_new = __builtin_amdgcn_cosf(vn - n0) + __builtin_amdgcn_cosf(vn - n1) + __builtin_amdgcn_cosf(vn - n2) +
                                        __builtin_amdgcn_cosf(vn - n3);
                    const float dE = -(e_new - e_old);
                    if (dE <= 0.0f || __expf(-beta * dE) > u01(r4[2 * j + 1])) xy_lat[r * L + col] = vn;
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < L * L; i += 256) x[i] = xy_lat[i];
}

__device__ __forceinline__ double wrap_half(double d)
{
    while (d > 0.5) d -= 1.0;      // as the reference: values of exactly +-0.5 stay
    while (d < -0.5) d += 1.0;
    return d;
}

__global__ __launch_bounds__(256) void xy_observe_kernel(const float *__restrict__ theta, int64_t B, int L, double *__restrict__ energy,
                                                         double *__restrict__ mag, int32_t *__restrict__ vpos, int32_t *__restrict__ vneg)
{
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float *u = theta + b * L * L;
    double e = 0.0, mx = 0.0, my = 0.0;
    int np = 0, nn = 0;
    for (int s = lane; s < L * L; s += 64) {
        const int i = s / L, j = s - i * L;
        const int ip = i + 1 == L ? 0 : i + 1, jp = j + 1 == L ? 0 : j + 1;
        const double t0 = u[i * L + j], t1 = u[ip * L + j], t2 = u[ip * L + jp], t3 = u[i * L + jp];
        e += cospi(2.0 * (t0 - t1)) + cospi(2.0 * (t0 - t3));
        double sn, cs;
        sincospi(2.0 * t0, &sn, &cs);
        mx += cs; my += sn;
        // plaquette (i,j) -> (i+1,j) -> (i+1,j+1) -> (i,j+1) -> back
        const double total = wrap_half(t1 - t0) + wrap_half(t2 - t1) + wrap_half(t3 - t2) + wrap_half(t0 - t3);
        const int wind = (int)round(total);
        np += wind == 1; nn += wind == -1;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        e += __shfl_down(e, off, 64); mx += __shfl_down(mx, off, 64); my += __shfl_down(my, off, 64);
        np += __shfl_down(np, off, 64); nn += __shfl_down(nn, off, 64);
    }
    if (lane == 0) {
        const double n = (double)L * L;
        if (energy) energy[b] = -e / n;
        if (mag) mag[b] = sqrt(mx * mx + my * my) / n;
        if (vpos) vpos[b] = np;
        if (vneg) vneg[b] = nn;
    }
}

}  // namespace

extern "C" {

int idm_xy_metropolis(int device, float *theta_dev, int64_t B, int L, float beta, float delta, int n_sweeps, uint64_t seed,
                      int64_t first_global_index, int sweep0, void *stream)
{
    if (!theta_dev || B <= 0 || n_sweeps < 0) return idm_set_error(IDM_ERR_INVALID, "bad argument");
    if (L < 2 || (L & 1) || L > 192) return idm_set_error(IDM_ERR_UNSUPPORTED, "the checkerboard XY sampler needs an even L in [2, 192]");
    if (!(delta > 0.0f) || delta > 1.0f) return idm_set_error(IDM_ERR_INVALID, "delta must be in (0, 1] turns");
    if (hipSetDevice(device) != hipSuccess) return idm_set_error(IDM_ERR_HIP, "hipSetDevice failed");
    static thread_local bool configured[IDM_MAX_DEVICES] = {};
    const int slot = idm_device_slot();
    if (slot < 0) return idm_set_error(IDM_ERR_HIP, "device index beyond IDM_MAX_DEVICES");
    if (!configured[slot]) {
        if (hipFuncSetAttribute((const void *)xy_metropolis_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return idm_set_error(IDM_ERR_HIP, "hipFuncSetAttribute failed");
        configured[slot] = true;
    }
    PhiloxArgs ph = {seed, first_global_index, 0};
    hipLaunchKernelGGL(xy_metropolis_kernel, dim3((unsigned)B), dim3(256), (size_t)L * L * sizeof(float), (hipStream_t)stream, theta_dev, L,
                       beta, delta, n_sweeps, ph, sweep0);
    if (hipGetLastError() != hipSuccess) return idm_set_error(IDM_ERR_HIP, "xy_metropolis_kernel launch failed");
    return IDM_OK;
}

int idm_xy_observe(int device, const float *theta_dev, int64_t B, int L, double *energy_dev, double *magnetization_dev,
                   int32_t *vortex_pos_dev, int32_t *vortex_neg_dev, void *stream)
{
    if (!theta_dev || B <= 0 || L < 2 || L > 4096) return idm_set_error(IDM_ERR_INVALID, "bad argument");
    if (hipSetDevice(device) != hipSuccess) return idm_set_error(IDM_ERR_HIP, "hipSetDevice failed");
    hipLaunchKernelGGL(xy_observe_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, theta_dev, B, L, energy_dev,
                       magnetization_dev, vortex_pos_dev, vortex_neg_dev);
    if (hipGetLastError() != hipSuccess) return idm_set_error(IDM_ERR_HIP, "xy_observe_kernel launch failed");
    return IDM_OK;
}

}  // extern "C"
