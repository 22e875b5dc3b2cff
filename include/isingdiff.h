/*
 * isingdiff.h — C ABI of libisingdiff.so: reverse-diffusion sampling of 2-D Ising lattices with
 * the reference's UNet score model on AMD MI355X (gfx950).
 *
 * The reference (Mizuki-OHASHI/diffusion-model) has no FFI of its own; the seam this library
 * sits behind is two Python callables in b_diffusion_model_unet_2d_ising.py:
 *     score_model([x, t_norm])                               (:88, built by create_unet_model :22-78)
 *     diffusion_backward(x_asym, beta_lst, score_model, T)   (:81-98)
 * plus the post-processing that follows them (:209-213 and b_pure_metoropolis_2d_ising.py:42-60).
 * Each entry point below names the reference lines it replaces.
 *
 * Conventions
 *   - Every function returns IDM_OK (0) or a negative idm_status; idm_last_error() then holds a
 *     message (thread-local, valid until the next call on that thread).  No C++ types or
 *     exceptions cross this boundary.
 *   - Pointers named *_dev are device pointers valid on the model's GPU; pointers named *_host
 *     are ordinary host memory.  The caller owns every buffer it passes.
 *   - Tensors are dense, NHWC, little-endian: state and score are float32 (B,H,W,C).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls enqueue work
 *     on that stream and return without synchronising unless stated otherwise.
 *   - A model handle is bound to one device and is not thread-safe; distinct handles are.  The calls
 *     on one handle share its activation workspace, so work enqueued through one handle must be
 *     ordered (one stream, or streams the caller synchronises); a call that has to grow the
 *     workspace synchronises the device first.
 */
#ifndef ISINGDIFF_H
#define ISINGDIFF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IDM_ABI_VERSION 3

typedef enum {
    IDM_OK = 0,
    IDM_ERR_INVALID = -1,   /* bad argument (null pointer, unsupported shape, …) */
    IDM_ERR_HIP = -2,       /* a HIP runtime call failed */
    IDM_ERR_NOMEM = -3,     /* device or host allocation failed */
    IDM_ERR_UNSUPPORTED = -4
} idm_status;

typedef enum {
    IDM_FP32 = 0,  /* float32 activations, float32 MFMA; canonical summation order (deterministic parity mode) */
    IDM_BF16 = 1   /* bfloat16 activations and weights on bf16 MFMA, float32 accumulate / state */
} idm_precision;

/* The reverse step x <- c1*x + c2*score + sqrt(beta_t)*z of b_diffusion_model_unet_2d_ising.py:
 *   IDM_UPDATE_SCORE  c1 = 1 + beta_t/2,        c2 = beta_t                                         (:96, what the script runs)
 *   IDM_UPDATE_DDPM   c1 = 1/sqrt(1 - beta_t),  c2 = -c1 * beta_t / sqrt(1 - alpha_cumprod[t])      (:95, commented out there:
 *                     the ancestral DDPM step for a model that predicts the noise, which is what the saved weights do)
 * The three constants are rounded to float32 on the host, one IEEE operation at a time; the device evaluates
 * ((c1*x) + (c2*score)) + (sd*z) with every product and sum rounded separately. */
typedef enum { IDM_UPDATE_SCORE = 0, IDM_UPDATE_DDPM = 1 } idm_update_rule;

#define IDM_NCONV 11
#define IDM_NDENSE 3

/* Weights of the UNet of create_unet_model (b_diffusion_model_unet_2d_ising.py:22-78; BatchNorm
 * variant b_diffusion_model_fft_unet_2d_ising.py:25-92), as host float32 arrays in Keras order:
 * conv kernels (kh,kw,Cin,Cout), dense kernels (in,out).  bn_scale/bn_shift are the folded
 * inference BatchNorm (scale = gamma/sqrt(var+eps), shift = beta - mean*scale) or all NULL. */
typedef struct {
    int32_t channels;       /* C of the state tensor: 1 (real space) or 2 (Re,Im) */
    int32_t base_filters;   /* 32 in the saved models */
    int32_t batchnorm;      /* 0 / 1 */
    int32_t reserved;
    const float *conv_w[IDM_NCONV];
    const float *conv_b[IDM_NCONV];
    const float *bn_scale[IDM_NCONV];
    const float *bn_shift[IDM_NCONV];
    const float *dense_w[IDM_NDENSE];
    const float *dense_b[IDM_NDENSE];
} idm_model_desc;

typedef struct idm_model idm_model;

/* Reverse-schedule configuration (b_diffusion_model_unet_2d_ising.py:168-172 for betas). */
typedef struct {
    int32_t T;                    /* number of steps; t runs T-1 .. 0 */
    int32_t precision;            /* idm_precision */
    const float *betas_host;      /* T float32 values */
    uint64_t seed;                /* Philox key (used when noise_dev == NULL and/or x_dev is initialised here) */
    int64_t first_global_index;   /* global index of lattice 0 of this call: RNG counters use
                                     first_global_index + b, so results do not depend on sharding */
    int32_t init_state;           /* 1: draw x_T ~ N(0,1) from Philox into x_dev first; 0: x_dev holds x_T */
    int32_t add_noise;            /* 1: add noise every step (reference behaviour); 0: deterministic drift only */
    int64_t chunk;                /* lattices processed together (bounds the workspace); 0 = automatic */
    int32_t update_rule;          /* idm_update_rule; 0 = the rule the reference runs (:96) */
    int32_t reserved;
    const float *alpha_cumprod_host; /* IDM_UPDATE_DDPM: T float32 values alpha_cumprod[t] (:169-173), or NULL to have
                                        them computed as float32(cumprod(1 - double(betas_host[t]))) in double precision.
                                        The reference takes the product over its float64 schedule and casts last: a caller
                                        that holds the float64 betas passes float32(np.cumprod(1 - betas64)) here to match
                                        it exactly; NULL starts from the float32 betas and can differ from that by an ulp */
} idm_sample_cfg;

int idm_abi_version(void);
const char *idm_last_error(void);

/* keras.models.load_model(...) (:199-201): copies and repacks the weights onto `device`. */
int idm_model_create(idm_model **out, int device, const idm_model_desc *desc);
int idm_model_destroy(idm_model *m);

/* The time-embedding sub-graph (:36-38, :56) evaluated for t_norm = float32(t)/float32(T),
 * t = 0..T-1: a (T, 4*base_filters) float32 table.  Cached in the handle; copied to
 * table_host when it is not NULL (synchronises). */
int idm_time_table(idm_model *m, int T, float *table_host);

/* score = score_model([x, t_norm]) (:88).  x_dev, score_dev: (B,H,W,C) float32; every lattice
 * shares t_norm.  H and W are arbitrary (zero-padded to a multiple of 4 and cropped, :29-34,:76). */
int idm_forward(idm_model *m, const float *x_dev, float t_norm, float *score_dev,
                int64_t B, int H, int W, int precision, void *stream);

/* idm_forward for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments and results as
 * idm_forward, bit for bit, but the activation workspace is a static device array of
 * IDM_NOALLOC_SCRATCH_BYTES that each GPU holds from library load until the process ends: the call performs
 * no hipMalloc / hipFree.  A batch whose workspace exceeds the array is rejected with IDM_ERR_INVALID (the
 * message states the largest B for that shape and precision; e.g. 390 lattices at L = 32 in bf16, 173 at
 * L = 20 in float32).  All handles on one GPU share the array: calls that overlap on one GPU (different
 * streams or threads) race; order them on one stream.  Per-kernel profiling is not recorded for these calls. */
/* What else the call touches: like every call on a handle it makes the handle's device current (hipSetDevice) and
 * rewrites the handle's time-embedding row (shared with idm_forward, hence the ordering rule above); the first
 * launch of each kernel function in a thread sets that function's LDS attribute and reads its occupancy (host-side
 * queries, legal under stream capture).  Captured into a HIP graph the call replays with the captured t_norm and
 * pointers; tests/test_zz_no_alloc.py captures a first call on a fresh handle and shape and replays it. */
#define IDM_NOALLOC_SCRATCH_BYTES (64u << 20)
int idm_forward_noalloc(idm_model *m, const float *x_dev, float t_norm, float *score_dev,
                        int64_t B, int H, int W, int precision, void *stream);

/* How idm_forward / idm_sample will cut a batch of B lattices of this shape and precision (no counterpart in the
 * reference, whose batch is one tensor): *chunk_out = lattices processed together (the fewest equal chunks whose
 * activation workspace fits the library's budget; B itself when one chunk suffices), *bytes_per_lattice_out = the
 * workspace a lattice needs.  Either pointer may be NULL.  Launches nothing. */
int idm_plan_chunk(idm_model *m, int64_t B, int H, int W, int precision, int64_t *chunk_out,
                   int64_t *bytes_per_lattice_out);

/* Optional taps for tests: like idm_forward, and copies layer i's activation (padded size,
 * float32, (B,h_i,w_i,c_i)) to taps_dev[i] for every non-NULL entry, i = 0..10. */
int idm_forward_taps(idm_model *m, const float *x_dev, float t_norm, float *score_dev,
                     int64_t B, int H, int W, int precision, float *const *taps_dev, void *stream);

/* diffusion_backward (:81-98) followed by the sign projection (:209-213).
 *   x_dev       (B,H,W,C) float32: x_T on entry (or filled here when cfg->init_state), final state on return
 *   noise_dev   NULL, or (T, B,H,W,C) float32 standard normals indexed by t (deterministic mode:
 *               the step adds noise_dev[t] * sqrt(beta_t)); NULL draws them in-kernel from Philox
 *   traj_dev    NULL, or (T+1, B,H,W,C): traj[T] = x_T, traj[t] = state after step t (reference return layout)
 *   scores_dev  NULL, or (T, B,H,W,C): the score used at step t
 *   spins_dev   NULL, or (B,H,W,C) int8: x > 0 ? +1 : -1 of the final state */
int idm_sample(idm_model *m, const idm_sample_cfg *cfg, float *x_dev, const float *noise_dev,
               float *traj_dev, float *scores_dev, int8_t *spins_dev,
               int64_t B, int H, int W, void *stream);

/* One reverse step on caller-held tensors: x <- ((1 + 0.5*beta)*x + beta*score) + z*sqrt(beta) (:96);
 * z_dev may be NULL (no noise).  n = element count. */
int idm_update(int device, float *x_dev, const float *score_dev, const float *z_dev, float beta,
               int64_t n, void *stream);

/* Fills out_dev (n_lattices, elems_per_lattice) with the standard normals the sampler draws at
 * step t for lattices first_global_index .. +n_lattices (t = T gives the x_T draw). */
int idm_philox_normal(int device, float *out_dev, int64_t n_lattices, int64_t elems_per_lattice,
                      uint64_t seed, int64_t first_global_index, int32_t t, void *stream);

/* np.where(x > 0, 1, -1) (:211).  n = element count. */
int idm_sign_project(int device, const float *x_dev, int8_t *spins_dev, int64_t n, void *stream);

/* ising_energy / ising_magnetization (b_pure_metoropolis_2d_ising.py:42-60) for ±1 spins
 * (B,L,L) int8, periodic boundaries, each bond once:
 *   M_dev[b] = sum s,  S_dev[b] = sum s*s_down + s*s_right   (energy = -J*S - h*M, magnetisation = M/L^2)
 *   hist_M_dev[(M + L*L)/2]   += 1   (L*L + 1 bins, uint64)
 *   hist_S_dev[(S + 2*L*L)/2] += 1   (2*L*L + 1 bins, uint64)
 *   moments_dev[0..5] += {count, sum M, sum |M|, sum M^2, sum S, sum S^2}   (int64)
 * Histogram and moment buffers accumulate (zero them first); any output pointer may be NULL. */
int idm_observe(int device, const int8_t *spins_dev, int64_t B, int L, int32_t *M_dev, int32_t *S_dev,
                unsigned long long *hist_M_dev, unsigned long long *hist_S_dev, long long *moments_dev,
                void *stream);

/* Spectral model post-processing: ising_ifft = np.fft.irfft2(re + 1j*im, s=(L,L)) (b_fft.py:12-17)
 * followed by np.where(x > 0, 1, -1) (b_diffusion_model_fft_unet_2d_ising.py:214-217).
 * spec_dev (B, L, L/2+1, 2) float32 (Re, Im); spins_dev (B,L,L) int8; real_dev NULL or (B,L,L) float64
 * receiving the transform itself.  numpy's axis order (complex transform along axis 0, then
 * complex-to-real along axis 1), float64 arithmetic, even L. */
int idm_irfft2_sign(int device, const float *spec_dev, int64_t B, int L, int8_t *spins_dev, double *real_dev,
                    void *stream);

/* Ising configurations at inverse temperature beta: checkerboard Metropolis sweeps with the acceptance
 * rule of pure_metropolis_next (b_pure_metoropolis_2d_ising.py:35-38): flip when u < exp(-beta*dE),
 * dE = 2 s (J*sum_nn + h), periodic boundaries.  The reference flips one random site at a time from
 * numpy's global RNG; here the two colours are swept in turn with Philox uniforms (same stationary
 * distribution, one independent chain per lattice).  spins_dev (B,L,L) int8 in/out, L even;
 * accept_host: 10 float32 = exp(-beta*dE) for s in {-1,+1} x sum_nn in {-4,-2,0,2,4} (row-major);
 * sweep0: index of the first sweep (continue a chain with sweep0 += n_sweeps). */
int idm_metropolis(int device, int8_t *spins_dev, int64_t B, int L, const float *accept_host, int n_sweeps,
                   uint64_t seed, int64_t first_global_index, int sweep0, void *stream);

/* 2-D XY model (XY_model/xymetoropolis.c, the reference's C program).  theta_dev (B,L,L) float32: angles in turns,
 * theta / 2 pi in [0, 1), as the reference's field2d holds them (there in double).
 * idm_xy_metropolis: n_sweeps sweeps of the single-site move of xy_sampling (:169-211: proposal x + (u - 1/2)*delta with
 *   delta = 0.1 in the reference, wrapped into [0,1); accept when dE <= 0 or exp(-beta*dE) > u'), the two checkerboard
 *   colours in turn with Philox uniforms instead of one random site per libc rand() draw: same stationary distribution,
 *   one independent chain per lattice; sweep0 = index of the first sweep (continue a chain with sweep0 += n_sweeps).
 * idm_xy_observe: per lattice xyenergy (:32-50, energy per site), xymagnetization (:53-69), and the counts of +1 / -1
 *   plaquette windings of xyvortex (:71-121), evaluated in float64; any output may be NULL. */
int idm_xy_metropolis(int device, float *theta_dev, int64_t B, int L, float beta, float delta, int n_sweeps, uint64_t seed,
                      int64_t first_global_index, int sweep0, void *stream);
int idm_xy_observe(int device, const float *theta_dev, int64_t B, int L, double *energy_dev, double *magnetization_dev,
                   int32_t *vortex_pos_dev, int32_t *vortex_neg_dev, void *stream);

/* Histogram exchange for a C caller that drives several GPUs from one process (no counterpart in the reference,
 * which is single-device; the sharded sampler needs exactly one exchange: the sum of the observable histograms
 * and moments).  idm_comm_create builds an RCCL communicator over `devices` (ncclCommInitAll; librccl is bound at
 * run time and only needed for n > 1); idm_allreduce_i64 sums `count` int64 values element-wise over the n device
 * buffers, in place (per_device_bufs[i] lives on devices[i]; streams[i] or NULL = default stream; enqueues and
 * returns).  With n = 1 it is the identity.  The Python host path uses torch.distributed (backend "nccl", which
 * is RCCL) for the same exchange, one process per GPU. */
typedef struct idm_comm idm_comm;
int idm_comm_create(idm_comm **out, const int *devices, int n);
int idm_allreduce_i64(idm_comm *c, int64_t *const *per_device_bufs, size_t count, void *const *streams);
int idm_comm_destroy(idm_comm *c);

/* Tuning switches (no counterpart in the reference).  "fused" (default 1): bf16 forward uses
 * the fused two-convolution band kernels when the shape fits; 0 forces one kernel per layer.
 * "strip_below_rows" (default 8): fused bands shorter than this run as single-convolution strip kernels.
 * "transpose" (default 1): two-channel (spectral) models in bf16 process the map transposed when its padded height
 * is a multiple of 32 and its width is not (the 32 x 17 spectrum of a 32 x 32 lattice), so that every MFMA lane holds a
 * pixel; inputs and outputs keep their layout. */
int idm_set_option(idm_model *m, const char *name, int value);

/* Per-kernel timing for roofline accounting (no counterpart in the reference).  Between
 * idm_profile_begin and idm_profile_end every kernel launch of idm_forward / idm_sample on this
 * handle is bracketed by HIP events recorded on the call's own stream.  idm_profile_end
 * synchronises the device and fills up to `cap` entries (one per kernel class, in launch order of
 * first use); *n receives the number of classes.  flop / bytes are the ALGORITHMIC work of the
 * launches (2*MACs of the convolutions; bytes each launch must read and write once). */
typedef struct {
    char name[32];
    int64_t launches;
    double ms;        /* sum of event-timed durations */
    double flop;
    double bytes;
} idm_profile_entry;

int idm_profile_begin(idm_model *m);
int idm_profile_end(idm_model *m, idm_profile_entry *entries, int cap, int *n);

/* ---------------------------------------------------------------------------------------------
 * MNIST bit-flip diffusion (mnist_diffusion/mnist_diffusion.py): residual UNet with
 * GroupNormalization(8) and per-block time vectors (create_improved_diffusion_unet :162-240),
 * sampler X ^= uniform < model([X, t/time_step]) (diffusion_backward :43-52).  Float32 parity
 * mode (canonical arithmetic) only.
 *
 * Weights: host float32 arrays in Keras layouts (conv (kh,kw,Cin,Cout), dense (in,out)), in this
 * order: in_w,in_b (Conv2D 1x1 :207); d1_w,d1_b,d2_w,d2_b (time MLP :202-203); for each of the
 * five ResBlocks (widths 32,64,128,64,32; inputs 32,32,64,192,96): t_w,t_b (Dense :173),
 * g1_g,g1_b (GroupNorm gamma,beta :177), c1_w,c1_b (:179), g2_g,g2_b (:183), c2_w,c2_b (:185),
 * s_w,s_b (1x1 shortcut :190, NULL where input width == width); out_w,out_b (:236). */
#define IDM_BITFLIP_NTENSOR 68

typedef struct {
    int32_t n_tensors;   /* IDM_BITFLIP_NTENSOR */
    int32_t reserved;
    const float *tensors[IDM_BITFLIP_NTENSOR];
} idm_bitflip_desc;

typedef struct {
    int32_t T;                     /* time_step; t runs T-1 .. 0 */
    int32_t reserved;
    const float *emb_table_host;   /* (T, 32): SinusoidalPositionalEmbedding (:67-99), row t = embedding of the value the
                                      model sees at step t.  The sampler feeds float32(t)/float32(T) (:47) to an Input
                                      declared int64 (:197): pass the rows of that float (what this library's Python
                                      binding does by default) or, if Keras casts to the declared dtype first, T copies
                                      of the row of position 0.  Parity with Keras is unpinned on this point. */
    uint64_t seed;                 /* Philox key for the uniforms (used when uniforms_dev == NULL) */
    int64_t first_global_index;    /* global index of sample 0 (uniforms do not depend on sharding) */
    int64_t chunk;                 /* samples processed together; 0 = automatic */
} idm_bitflip_cfg;

typedef struct idm_bitflip idm_bitflip;

int idm_bitflip_create(idm_bitflip **out, int device, const idm_bitflip_desc *desc);
int idm_bitflip_destroy(idm_bitflip *m);

/* IDM_FP32 (default): canonical float32 arithmetic, bit-exact against the CPU restatement.  IDM_BF16: the
 * 3x3 convolutions of the ResBlocks take bf16 operands (GroupNorm + swish output and weights rounded to
 * bf16) on bf16 MFMA with float32 accumulation; all other arithmetic stays float32.  On shapes its fused
 * kernels cover (maps 16 .. 32 wide, 28 x 28 among them) the tensors between kernels are stored as IEEE half
 * (saturating at +-65504) and the GroupNorms are applied inside the convolutions. */
int idm_bitflip_set_precision(idm_bitflip *m, int precision);

/* The five per-block time vectors swish(Dense_i(temb)) (:173, :200-203) for one embedding row
 * (32 host floats): 320 device floats.  Synchronises.  For tests. */
int idm_bitflip_time(idm_bitflip *m, const float *emb_host, float *tvec_dev);

/* flip_proba = model([X, t]) (:48).  x_dev (B,H,W) bytes 0/1; emb_host: the embedding row of t;
 * proba_dev (B,H,W) float32; taps_dev NULL or 5 device buffers receiving the ResBlock outputs
 * (B,h_i,w_i,width_i) (entries may be NULL).  H, W multiples of 4. */
int idm_bitflip_forward(idm_bitflip *m, const uint8_t *x_dev, const float *emb_host, float *proba_dev,
                        int64_t B, int H, int W, float *const *taps_dev, void *stream);

/* diffusion_backward (:43-52).  x_dev (B,H,W) bytes: X_b on entry, final state on return;
 * uniforms_dev NULL, or (T,B,H,W) float32 indexed by t (deterministic mode: flip = u[t] < p);
 * traj_dev NULL, or (T+1,B,H,W) bytes in the reference's order (traj[0] = X_b, traj[k] = after k steps). */
int idm_bitflip_sample(idm_bitflip *m, const idm_bitflip_cfg *cfg, uint8_t *x_dev, const float *uniforms_dev,
                       uint8_t *traj_dev, int64_t B, int H, int W, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Dense spectral score model (b_diffusion_model_fft_2d_ising.py:105-114: Sequential of Dense
 * layers on [x | t_norm], saved as models/b_diffusion_model_2d_ising.keras) and its sampler
 * (:19-38).  IDM_FP32: parity mode (canonical arithmetic); IDM_BF16: the Dense layers on bf16 MFMA with
 * float32 accumulation, float32 state, the t_norm column applied in float32.  Layer widths up to 448
 * (IDM_BF16: multiples of 4). */
#define IDM_MLP_MAXLAYERS 8

typedef struct {
    int32_t n_layers;
    int32_t reserved;
    int32_t in_dim[IDM_MLP_MAXLAYERS];    /* in_dim[0] = n + 1 (state plus the t_norm column) */
    int32_t out_dim[IDM_MLP_MAXLAYERS];   /* out_dim[last] = n */
    int32_t act[IDM_MLP_MAXLAYERS];       /* 0 = linear, 1 = tanh */
    const float *w[IDM_MLP_MAXLAYERS];    /* host, Keras layout (in, out) */
    const float *b[IDM_MLP_MAXLAYERS];
} idm_mlp_desc;

typedef struct idm_mlp idm_mlp;

int idm_mlp_create(idm_mlp **out, int device, const idm_mlp_desc *desc);
int idm_mlp_destroy(idm_mlp *m);

/* score = score_model(concat([x, t_norm])) (:28-30).  x_dev, score_dev (B, n) float32. */
int idm_mlp_forward(idm_mlp *m, const float *x_dev, float t_norm, float *score_dev, int64_t B, void *stream);

/* The same in the given precision (IDM_FP32 is idm_mlp_forward). */
int idm_mlp_forward_precision(idm_mlp *m, const float *x_dev, float t_norm, float *score_dev, int64_t B, int precision,
                              void *stream);

/* diffusion_backward (:19-38): cfg as for idm_sample (chunk unused, update_rule must be IDM_UPDATE_SCORE); both precisions draw the same noise.
 * x_dev (B, n): x_T on entry (or drawn here when cfg->init_state), final state on return;
 * noise_dev NULL or (T, B, n) standard normals indexed by t; traj_dev NULL or (T+1, B, n) with
 * traj[T] = x_T and traj[t] = state after step t (the reference's return layout). */
int idm_mlp_sample(idm_mlp *m, const idm_sample_cfg *cfg, float *x_dev, const float *noise_dev, float *traj_dev,
                   int64_t B, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ISINGDIFF_H */
