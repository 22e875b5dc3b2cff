/*
 * TEST INFRASTRUCTURE — CPU restatement (plain C) of the reference's Keras sampling path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product never links or calls it.
 *
 * PARITY UNPINNED by the reference's own tests: the reference ships no tests, golden
 * vectors or recorded outputs for this path, and its arithmetic lives in Keras 3.10.0 /
 * TensorFlow 2.19.0 (uv.lock:339-340, uv.lock:1001-1002), which are not installed here.
 * This file restates the published semantics of the Keras layers the reference calls
 * (Conv2D "same" = zero padding, cross-correlation, kernel (kh,kw,Cin,Cout); swish =
 * x*sigmoid(x); AveragePooling2D(2); UpSampling2D(2, nearest); Concatenate([up, skip]);
 * BatchNormalization inference; Dense x@W+b).  It is pinned by the saved weights, by the
 * trained-objective known answers in tests/test_oracle.py and by agreement to rounding with
 * the independent PyTorch restatement oracle/ref_torch.py.
 *
 * Follows:
 *   create_unet_model     b_diffusion_model_unet_2d_ising.py:22-78
 *                         b_diffusion_model_fft_unet_2d_ising.py:25-92 (Conv -> BN -> swish)
 *   diffusion_backward    b_diffusion_model_unet_2d_ising.py:81-98 (update rule :96)
 *   sign projection       b_diffusion_model_unet_2d_ising.py:209-213
 *   ising_energy          b_pure_metoropolis_2d_ising.py:42-53
 *   ising_magnetization   b_pure_metoropolis_2d_ising.py:56-60
 *
 * CANONICAL ARITHMETIC.  Float32 results depend on summation order, and the reference's
 * own order (Eigen/oneDNN inside TensorFlow) is not observable.  This file fixes one order
 * and one set of elementary functions so that a second implementation can reproduce it
 * bit for bit:
 *   - every contraction is a single float32 fused-multiply-add chain that starts at +0 and
 *     runs over k in ascending order; for a 3x3 conv k = (ky*3 + kx)*Cin + c (taps outer,
 *     channels inner, concatenated channels in [upsampled, skip] order); out-of-range taps
 *     contribute nothing; the bias is added after the chain (one rounding);
 *   - BatchNorm (inference) is z*scale + shift with two roundings, scale = gamma/sqrt(var+eps);
 *   - swish(y) = y / (1 + cexp(-y)) with cexp below (clamped argument, Cephes polynomial,
 *     every step an explicit fmaf / multiply / add, exact ldexp), IEEE division;
 *   - average pooling is (((a + b) + c) + d) * 0.25 over (0,0),(0,1),(1,0),(1,1);
 *   - the sampler update is ((c1*x) + (beta*score)) + (z*sqrt(beta)), c1 = 1 + 0.5*beta, every
 *     operation rounded separately (no contraction).
 * Compile with -ffp-contract=off so the compiler fuses nothing that is not written as fmaf.
 *
 * With quant_bf16 != 0 the conv inputs (stored activations) and conv weights are rounded to
 * bfloat16 (round-to-nearest-even) exactly where a bf16 tensor-core pipeline rounds them,
 * while accumulation, bias, activation and the sampler state stay float32.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REF_NCONV 11
#define REF_NDENSE 3

typedef struct {
    int channels;      /* C of the state tensor */
    int filters;       /* base_filters */
    int batchnorm;
    int quant_bf16;
    int cin[REF_NCONV], cout[REF_NCONV], ksize[REF_NCONV];
    float *w[REF_NCONV];      /* (k*k, Cin, Cout) */
    float *b[REF_NCONV];
    float *scale[REF_NCONV];  /* NULL without BatchNorm */
    float *shift[REF_NCONV];
    float *dw[REF_NDENSE];    /* (in, out) */
    float *db[REF_NDENSE];
    int din[REF_NDENSE], dout[REF_NDENSE];
} ref_model;

static float bf16_round(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return x; /* NaN stays */
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&x, &u, 4);
    return x;
}

float ref_cexp(float x)
{
    x = fminf(fmaxf(x, -87.0f), 87.0f);
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    p = fmaf(p, r2, r);
    p = p + 1.0f;
    return ldexpf(p, (int)n);
}

float ref_silu(float y) { return y / (1.0f + ref_cexp(-y)); }

static float *dup_f32(const float *src, size_t n, int quant)
{
    float *d = (float *)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) d[i] = quant ? bf16_round(src[i]) : src[i];
    return d;
}

ref_model *ref_model_create(int channels, int filters, int batchnorm, int quant_bf16,
                            const float *const *conv_w, const float *const *conv_b,
                            const float *const *bn_scale, const float *const *bn_shift,
                            const float *const *dense_w, const float *const *dense_b)
{
    ref_model *m = (ref_model *)calloc(1, sizeof(ref_model));
    const int f = filters, c = channels;
    const int cin[REF_NCONV] = {c, f, f, 2 * f, 2 * f, 4 * f, 6 * f, 2 * f, 3 * f, f, f};
    const int cout[REF_NCONV] = {f, f, 2 * f, 2 * f, 4 * f, 4 * f, 2 * f, 2 * f, f, f, c};
    m->channels = c; m->filters = f; m->batchnorm = batchnorm; m->quant_bf16 = quant_bf16;
    for (int i = 0; i < REF_NCONV; ++i) {
        m->cin[i] = cin[i]; m->cout[i] = cout[i]; m->ksize[i] = (i == 10) ? 1 : 3;
        size_t nw = (size_t)m->ksize[i] * m->ksize[i] * cin[i] * cout[i];
        /* layers 0 and 10 stay float32 in the bf16 pipeline (K too thin for matrix cores) */
        int q = quant_bf16 && i != 0 && i != 10;
        m->w[i] = dup_f32(conv_w[i], nw, q);
        m->b[i] = dup_f32(conv_b[i], cout[i], 0);
        if (batchnorm && i < 10) {
            m->scale[i] = dup_f32(bn_scale[i], cout[i], 0);
            m->shift[i] = dup_f32(bn_shift[i], cout[i], 0);
        }
    }
    const int din[REF_NDENSE] = {1, 4 * f, 4 * f};
    for (int i = 0; i < REF_NDENSE; ++i) {
        m->din[i] = din[i]; m->dout[i] = 4 * f;
        m->dw[i] = dup_f32(dense_w[i], (size_t)din[i] * 4 * f, 0);
        m->db[i] = dup_f32(dense_b[i], 4 * f, 0);
    }
    return m;
}

void ref_model_destroy(ref_model *m)
{
    if (!m) return;
    for (int i = 0; i < REF_NCONV; ++i) { free(m->w[i]); free(m->b[i]); free(m->scale[i]); free(m->shift[i]); }
    for (int i = 0; i < REF_NDENSE; ++i) { free(m->dw[i]); free(m->db[i]); }
    free(m);
}

/* time_add row for one t_norm: Dense -> swish -> Dense -> Dense (…unet_2d_ising.py:36-38,56) */
void ref_time_add(const ref_model *m, float t_norm, float *out /* 4F */)
{
    const int n = 4 * m->filters;
    float *e0 = (float *)malloc(3 * n * sizeof(float)), *e1 = e0 + n, *e2 = e1 + n;
    for (int j = 0; j < n; ++j) {
        float acc = fmaf(t_norm, m->dw[0][j], 0.0f);
        e0[j] = ref_silu(acc + m->db[0][j]);
    }
    for (int j = 0; j < n; ++j) {
        float acc = 0.0f;
        for (int i = 0; i < n; ++i) acc = fmaf(e0[i], m->dw[1][(size_t)i * n + j], acc);
        e1[j] = acc + m->db[1][j];
    }
    for (int j = 0; j < n; ++j) {
        float acc = 0.0f;
        for (int i = 0; i < n; ++i) acc = fmaf(e1[i], m->dw[2][(size_t)i * n + j], acc);
        e2[j] = acc + m->db[2][j];
    }
    memcpy(out, e2, n * sizeof(float));
    free(e0);
}

/* table[t] = time_add(float32(t) / float32(T)), t = 0..T-1 (…unet_2d_ising.py:87) */
void ref_time_table(const ref_model *m, int T, float *table)
{
    for (int t = 0; t < T; ++t) ref_time_add(m, (float)t / (float)T, table + (size_t)t * 4 * m->filters);
}

/* One conv layer on one image, NHWC.  in: (H,W,Cin); out: (H,W,Cout).
 * post: +bias -> (BN) -> (swish) -> (+time_add[c]). */
static void conv_image(const float *in, int H, int W, int Cin, int ks, const float *w, const float *bias,
                       const float *scale, const float *shift, int silu, const float *tadd, float *out, int Cout)
{
    const int r = ks / 2;
    float acc[256];
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            for (int co = 0; co < Cout; ++co) acc[co] = 0.0f;
            for (int ky = 0; ky < ks; ++ky) {
                const int yy = y + ky - r;
                if (yy < 0 || yy >= H) continue;
                for (int kx = 0; kx < ks; ++kx) {
                    const int xx = x + kx - r;
                    if (xx < 0 || xx >= W) continue;
                    const float *ip = in + ((size_t)yy * W + xx) * Cin;
                    const float *wp = w + (size_t)(ky * ks + kx) * Cin * Cout;
                    for (int c = 0; c < Cin; ++c) {
                        const float a = ip[c];
                        const float *wr = wp + (size_t)c * Cout;
                        for (int co = 0; co < Cout; ++co) acc[co] = fmaf(a, wr[co], acc[co]);
                    }
                }
            }
            float *op = out + ((size_t)y * W + x) * Cout;
            for (int co = 0; co < Cout; ++co) {
                float z = acc[co] + bias[co];
                if (scale) { z = z * scale[co]; z = z + shift[co]; }
                if (silu) z = ref_silu(z);
                if (tadd) z = z + tadd[co];
                op[co] = z;
            }
        }
}

static void quantize(float *p, size_t n) { for (size_t i = 0; i < n; ++i) p[i] = bf16_round(p[i]); }

static void avgpool2(const float *in, int H, int W, int C, float *out)
{
    const int Ho = H / 2, Wo = W / 2;
    for (int y = 0; y < Ho; ++y)
        for (int x = 0; x < Wo; ++x)
            for (int c = 0; c < C; ++c) {
                const float a = in[((size_t)(2 * y) * W + 2 * x) * C + c];
                const float b = in[((size_t)(2 * y) * W + 2 * x + 1) * C + c];
                const float d = in[((size_t)(2 * y + 1) * W + 2 * x) * C + c];
                const float e = in[((size_t)(2 * y + 1) * W + 2 * x + 1) * C + c];
                out[((size_t)y * Wo + x) * C + c] = (((a + b) + d) + e) * 0.25f;
            }
}

/* out (2H,2W,Cu+Cs) = concat(nearest_up(up (H,W,Cu)), skip (2H,2W,Cs)) */
static void up_concat(const float *up, int H, int W, int Cu, const float *skip, int Cs, float *out)
{
    const int Ho = 2 * H, Wo = 2 * W, C = Cu + Cs;
    for (int y = 0; y < Ho; ++y)
        for (int x = 0; x < Wo; ++x) {
            float *o = out + ((size_t)y * Wo + x) * C;
            memcpy(o, up + ((size_t)(y / 2) * W + x / 2) * Cu, Cu * sizeof(float));
            memcpy(o + Cu, skip + ((size_t)y * Wo + x) * Cs, Cs * sizeof(float));
        }
}

/* Single conv layer over a batch (exposed for per-kernel tests). */
void ref_conv_layer(const ref_model *m, int layer, const float *in, int B, int H, int W,
                    const float *tadd, float *out)
{
    const int Cin = m->cin[layer], Cout = m->cout[layer];
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b)
        conv_image(in + (size_t)b * H * W * Cin, H, W, Cin, m->ksize[layer], m->w[layer], m->b[layer],
                   m->scale[layer], m->shift[layer], layer < 10,
                   tadd, out + (size_t)b * H * W * Cout, Cout);
}

/* Optional per-layer taps: if taps != NULL, taps[i] (i = 0..10) receives layer i's output
 * for every image (caller-allocated, (B, h_i, w_i, cout_i)); entries may be NULL. */
void ref_forward_taps(const ref_model *m, const float *x, const float *tadd, float *score,
                      int B, int H, int W, float *const *taps)
{
    const int C = m->channels, F = m->filters;
    const int ph = (4 - H % 4) % 4, pw = (4 - W % 4) % 4;
    const int Hp = H + ph, Wp = W + pw;
    const int q = m->quant_bf16;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        const size_t np1 = (size_t)Hp * Wp, np2 = np1 / 4, np4 = np1 / 16;
        float *xin = (float *)calloc(np1 * C, sizeof(float));
        float *a0 = (float *)malloc(np1 * F * sizeof(float));
        float *s1 = (float *)malloc(np1 * F * sizeof(float));
        float *p1 = (float *)malloc(np2 * F * sizeof(float));
        float *a2 = (float *)malloc(np2 * 2 * F * sizeof(float));
        float *s2 = (float *)malloc(np2 * 2 * F * sizeof(float));
        float *p2 = (float *)malloc(np4 * 2 * F * sizeof(float));
        float *a4 = (float *)malloc(np4 * 4 * F * sizeof(float));
        float *a5 = (float *)malloc(np4 * 4 * F * sizeof(float));
        float *c6 = (float *)malloc(np2 * 6 * F * sizeof(float));
        float *a6 = (float *)malloc(np2 * 2 * F * sizeof(float));
        float *a7 = (float *)malloc(np2 * 2 * F * sizeof(float));
        float *c8 = (float *)malloc(np1 * 3 * F * sizeof(float));
        float *a8 = (float *)malloc(np1 * F * sizeof(float));
        float *a9 = (float *)malloc(np1 * F * sizeof(float));
        float *hd = (float *)malloc(np1 * C * sizeof(float));
        for (int y = 0; y < H; ++y)
            memcpy(xin + (size_t)y * Wp * C, x + (((size_t)b * H + y) * W) * C, (size_t)W * C * sizeof(float));
#define CONV(i, src, hh, ww, ta, dst) \
        conv_image(src, hh, ww, m->cin[i], m->ksize[i], m->w[i], m->b[i], m->scale[i], m->shift[i], (i) < 10, ta, dst, m->cout[i])
#define TAP(i, buf, n) \
        if (taps && taps[i]) memcpy(taps[i] + (size_t)b * (n), buf, (size_t)(n) * sizeof(float))
        CONV(0, xin, Hp, Wp, NULL, a0); if (q) quantize(a0, np1 * F);          TAP(0, a0, np1 * F);
        CONV(1, a0, Hp, Wp, NULL, s1);  if (q) quantize(s1, np1 * F);          TAP(1, s1, np1 * F);
        avgpool2(s1, Hp, Wp, F, p1);    if (q) quantize(p1, np2 * F);
        CONV(2, p1, Hp / 2, Wp / 2, NULL, a2); if (q) quantize(a2, np2 * 2 * F); TAP(2, a2, np2 * 2 * F);
        CONV(3, a2, Hp / 2, Wp / 2, NULL, s2); if (q) quantize(s2, np2 * 2 * F); TAP(3, s2, np2 * 2 * F);
        avgpool2(s2, Hp / 2, Wp / 2, 2 * F, p2); if (q) quantize(p2, np4 * 2 * F);
        CONV(4, p2, Hp / 4, Wp / 4, tadd, a4); if (q) quantize(a4, np4 * 4 * F); TAP(4, a4, np4 * 4 * F);
        CONV(5, a4, Hp / 4, Wp / 4, NULL, a5); if (q) quantize(a5, np4 * 4 * F); TAP(5, a5, np4 * 4 * F);
        up_concat(a5, Hp / 4, Wp / 4, 4 * F, s2, 2 * F, c6);
        CONV(6, c6, Hp / 2, Wp / 2, NULL, a6); if (q) quantize(a6, np2 * 2 * F); TAP(6, a6, np2 * 2 * F);
        CONV(7, a6, Hp / 2, Wp / 2, NULL, a7); if (q) quantize(a7, np2 * 2 * F); TAP(7, a7, np2 * 2 * F);
        up_concat(a7, Hp / 2, Wp / 2, 2 * F, s1, F, c8);
        CONV(8, c8, Hp, Wp, NULL, a8); if (q) quantize(a8, np1 * F);           TAP(8, a8, np1 * F);
        CONV(9, a8, Hp, Wp, NULL, a9);                                           TAP(9, a9, np1 * F);
        CONV(10, a9, Hp, Wp, NULL, hd);                                          TAP(10, hd, np1 * C);
#undef CONV
#undef TAP
        for (int y = 0; y < H; ++y)
            memcpy(score + (((size_t)b * H + y) * W) * C, hd + (size_t)y * Wp * C, (size_t)W * C * sizeof(float));
        free(xin); free(a0); free(s1); free(p1); free(a2); free(s2); free(p2); free(a4); free(a5);
        free(c6); free(a6); free(a7); free(c8); free(a8); free(a9); free(hd);
    }
}

/* score = model([x, t_norm]); tadd = time_add row for t_norm (4F floats). x, score: (B,H,W,C). */
void ref_forward(const ref_model *m, const float *x, const float *tadd, float *score, int B, int H, int W)
{
    ref_forward_taps(m, x, tadd, score, B, H, W, NULL);
}

/* x <- ((c1*x) + (beta*score)) + z*sqrt(beta); z may be NULL (no noise term). */
void ref_update(float *x, const float *score, const float *z, float beta, size_t n)
{
    const float half_beta = 0.5f * beta;
    const float c1 = 1.0f + half_beta;
    const float sd = sqrtf(beta);
    for (size_t i = 0; i < n; ++i) {
        float a = c1 * x[i];
        float b = beta * score[i];
        float s = a + b;
        if (z) { float nz = z[i] * sd; s = s + nz; }
        x[i] = s;
    }
}

/* The commented-out alternative update of b_diffusion_model_unet_2d_ising.py:95 (ancestral DDPM step for a noise-predicting
 * model):  x <- (1/sqrt(1-beta)) * (x - beta/sqrt(1-alpha_cumprod) * score) + noise, restated in the same canonical form as
 * ref_update: x <- ((c1*x) + (c2*score)) + (z*sd) with c1 = 1/sqrt(1-beta), c2 = -(c1 * (beta / sqrt(1-alpha_cumprod))),
 * every operation rounded to float32. */
void ref_update_ddpm(float *x, const float *score, const float *z, float beta, float alpha_cumprod, size_t n)
{
    const float om = 1.0f - beta;
    const float rs = sqrtf(om);
    const float c1 = 1.0f / rs;
    const float oma = 1.0f - alpha_cumprod;
    const float d = sqrtf(oma);
    const float e = beta / d;
    const float ae = c1 * e;
    const float c2 = -ae;
    const float sd = sqrtf(beta);
    for (size_t i = 0; i < n; ++i) {
        float a = c1 * x[i];
        float b = c2 * score[i];
        float s = a + b;
        if (z) { float nz = z[i] * sd; s = s + nz; }
        x[i] = s;
    }
}

/* Full reverse schedule, t = T-1 .. 0.  noise_z: (T, B*H*W*C) standard normals indexed by t,
 * or NULL.  scores_out: optional (T, n) per-step score record indexed by t.
 * alpha_cumprod: NULL = the update the reference runs (:96); else T values and the DDPM update (:95). */
void ref_sample_rule(const ref_model *m, float *x, const float *betas, const float *alpha_cumprod, int T, const float *noise_z,
                     int B, int H, int W, float *scores_out)
{
    const size_t n = (size_t)B * H * W * m->channels;
    const int nf = 4 * m->filters;
    float *score = (float *)malloc(n * sizeof(float));
    float *tadd = (float *)malloc(nf * sizeof(float));
    for (int t = T - 1; t >= 0; --t) {
        ref_time_add(m, (float)t / (float)T, tadd);
        ref_forward(m, x, tadd, score, B, H, W);
        if (scores_out) memcpy(scores_out + (size_t)t * n, score, n * sizeof(float));
        const float *z = noise_z ? noise_z + (size_t)t * n : NULL;
        if (alpha_cumprod) ref_update_ddpm(x, score, z, betas[t], alpha_cumprod[t], n);
        else ref_update(x, score, z, betas[t], n);
    }
    free(score); free(tadd);
}

void ref_sample(const ref_model *m, float *x, const float *betas, int T, const float *noise_z,
                int B, int H, int W, float *scores_out)
{
    const size_t n = (size_t)B * H * W * m->channels;
    const int nf = 4 * m->filters;
    float *score = (float *)malloc(n * sizeof(float));
    float *tadd = (float *)malloc(nf * sizeof(float));
    for (int t = T - 1; t >= 0; --t) {
        ref_time_add(m, (float)t / (float)T, tadd);
        ref_forward(m, x, tadd, score, B, H, W);
        if (scores_out) memcpy(scores_out + (size_t)t * n, score, n * sizeof(float));
        ref_update(x, score, noise_z ? noise_z + (size_t)t * n : NULL, betas[t], n);
    }
    free(score); free(tadd);
}

/* spin = x > 0 ? +1 : -1 */
void ref_sign_project(const float *x, int8_t *spins, size_t n)
{
    for (size_t i = 0; i < n; ++i) spins[i] = x[i] > 0.0f ? 1 : -1;
}

/* Per lattice: M = sum s; S = sum s*s_down + s*s_right (periodic, each bond once).
 * energy = -J*S - h*M. */
void ref_observe(const int8_t *spins, int B, int L, int64_t *M, int64_t *S)
{
    for (int b = 0; b < B; ++b) {
        const int8_t *s = spins + (size_t)b * L * L;
        int64_t mm = 0, ss = 0;
        for (int y = 0; y < L; ++y)
            for (int x = 0; x < L; ++x) {
                const int v = s[y * L + x];
                mm += v;
                ss += v * s[((y + 1) % L) * L + x] + v * s[y * L + (x + 1) % L];
            }
        M[b] = mm; S[b] = ss;
    }
}

/* Philox4x32-10 block (counter c[4], key k[2]) -> out[4]; the counter-based generator the
 * device sampler draws its noise from.  Constants: Salmon et al., SC'11. */
void ref_philox4x32(const uint32_t c[4], const uint32_t k[2], uint32_t out[4])
{
    uint32_t c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], k0 = k[0], k1 = k[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
