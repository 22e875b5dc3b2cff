/*
 * TEST INFRASTRUCTURE — CPU restatement (plain C) of the reference's MNIST bit-flip diffusion
 * sampler.  Only tests/ may load this library; the product never links or calls it.
 *
 * PARITY UNPINNED: the reference ships neither weights nor recorded outputs for this model
 * (mnist_diffusion/mnist_diffusion.py saves to a Colab drive), and Keras / TensorFlow are not
 * installed here.  The restatement is pinned only by agreement with the independent PyTorch
 * restatement in oracle/ref_torch.py on synthetic weights.
 *
 * Follows mnist_diffusion/mnist_diffusion.py:
 *   SinusoidalPositionalEmbedding        :67-99   (the embedding itself is computed by the host
 *                                                   in float32 and passed in: T rows of `dim`)
 *   create_improved_diffusion_unet       :162-240 (ResBlock :168-193)
 *   diffusion_backward                   :43-52   (X ^= uniform < model([X, t/time_step]))
 *
 * Keras semantics restated: GroupNormalization(groups=8) has epsilon 1e-3, statistics over
 * (H, W, C/groups) per sample and group, y = x*(gamma*rsqrt(var+eps)) + (beta - mean*gamma*rsqrt);
 * MaxPooling2D(2); UpSampling2D(2) nearest; Concatenate([up, skip]); Conv2D "same".
 *
 * CANONICAL ARITHMETIC (so that a second implementation can match bit for bit), on top of the
 * rules of ref_cpu.c (fma chains from +0 in ascending k, bias after, swish via ref_cexp):
 *   - GroupNorm sums: 64 partial sums, partial l takes elements e = l, l+64, ... in ascending
 *     order (e = pixel*Cg + channel-in-group); then a butterfly p[l] += p[l^off] for
 *     off = 32,16,8,4,2,1; mean = sum/N; variance = (sum of fmaf(d,d,.) with d = x-mean)/N;
 *   - rsqrt by the fixed iteration below (bit trick + 4 Newton steps), not the libm sqrt;
 *   - y = (x*a) + b with a = inv*gamma, b = (-mean*a) + beta, every operation rounded once;
 *   - sigmoid(z) = 1 / (1 + cexp(-z)), IEEE division; out = shortcut + h, one addition.
 * Compile with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

float ref_cexp(float x);
float ref_silu(float y);

#define BF_NBLK 5
#define BF_NTENSOR (6 + 12 * BF_NBLK + 2)

typedef struct {
    int emb_dim, temb;            /* 32, 128 */
    int cin[BF_NBLK], width[BF_NBLK];
    const float *in_w, *in_b, *d1_w, *d1_b, *d2_w, *d2_b, *out_w, *out_b;
    const float *t_w[BF_NBLK], *t_b[BF_NBLK];
    const float *g1_g[BF_NBLK], *g1_b[BF_NBLK], *c1_w[BF_NBLK], *c1_b[BF_NBLK];
    const float *g2_g[BF_NBLK], *g2_b[BF_NBLK], *c2_w[BF_NBLK], *c2_b[BF_NBLK];
    const float *s_w[BF_NBLK], *s_b[BF_NBLK];
} ref_bitflip;

/* tensors: in_w,in_b,d1_w,d1_b,d2_w,d2_b, then per block t_w,t_b,g1_g,g1_b,c1_w,c1_b,g2_g,g2_b,
 * c2_w,c2_b,s_w,s_b (s_* NULL for an identity shortcut), then out_w,out_b.  The caller keeps
 * the arrays alive. */
ref_bitflip *ref_bitflip_create(const float *const *t)
{
    ref_bitflip *m = (ref_bitflip *)calloc(1, sizeof(ref_bitflip));
    const int cin[BF_NBLK] = {32, 32, 64, 192, 96}, width[BF_NBLK] = {32, 64, 128, 64, 32};
    m->emb_dim = 32; m->temb = 128;
    int k = 0;
    m->in_w = t[k++]; m->in_b = t[k++]; m->d1_w = t[k++]; m->d1_b = t[k++]; m->d2_w = t[k++]; m->d2_b = t[k++];
    for (int i = 0; i < BF_NBLK; ++i) {
        m->cin[i] = cin[i]; m->width[i] = width[i];
        m->t_w[i] = t[k++]; m->t_b[i] = t[k++];
        m->g1_g[i] = t[k++]; m->g1_b[i] = t[k++]; m->c1_w[i] = t[k++]; m->c1_b[i] = t[k++];
        m->g2_g[i] = t[k++]; m->g2_b[i] = t[k++]; m->c2_w[i] = t[k++]; m->c2_b[i] = t[k++];
        m->s_w[i] = t[k++]; m->s_b[i] = t[k++];
    }
    m->out_w = t[k++]; m->out_b = t[k++];
    return m;
}

void ref_bitflip_destroy(ref_bitflip *m) { free(m); }

float ref_rsqrt_canon(float v)
{
    uint32_t i;
    memcpy(&i, &v, 4);
    i = 0x5f3759dfu - (i >> 1);
    float r;
    memcpy(&r, &i, 4);
    const float h = 0.5f * v;
    for (int k = 0; k < 4; ++k) {
        float t = r * r;
        t = fmaf(-h, t, 1.5f);
        r = r * t;
    }
    return r;
}

static void dense(const float *x, int n_in, const float *w, const float *b, int n_out, int silu, float *y)
{
    for (int j = 0; j < n_out; ++j) {
        float acc = 0.0f;
        for (int i = 0; i < n_in; ++i) acc = fmaf(x[i], w[(size_t)i * n_out + j], acc);
        acc = acc + b[j];
        y[j] = silu ? ref_silu(acc) : acc;
    }
}

/* emb: the sinusoidal embedding row of this step (emb_dim floats).  out: the five per-block
 * time vectors swish(Dense_i(temb)), concatenated (32+64+128+64+32 = 320 floats). (:173, :200-203) */
void ref_bitflip_time(const ref_bitflip *m, const float *emb, float *out)
{
    float a[128], b[128];
    dense(emb, m->emb_dim, m->d1_w, m->d1_b, m->temb, 1, a);
    dense(a, m->temb, m->d2_w, m->d2_b, m->temb, 1, b);
    int o = 0;
    for (int i = 0; i < BF_NBLK; ++i) {
        dense(b, m->temb, m->t_w[i], m->t_b[i], m->width[i], 1, out + o);
        o += m->width[i];
    }
}

/* GroupNorm(8) + swish on one image (H,W,C) */
static void gn_silu(const float *x, int H, int W, int C, const float *gamma, const float *beta, float *y)
{
    const int G = 8, Cg = C / G, N = H * W * Cg;
    for (int g = 0; g < G; ++g) {
        float p[64];
        for (int l = 0; l < 64; ++l) p[l] = 0.0f;
        for (int e = 0; e < N; ++e) p[e & 63] = p[e & 63] + x[(size_t)(e / Cg) * C + g * Cg + e % Cg];
        for (int off = 32; off >= 1; off >>= 1) {
            float q[64];
            for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ off];
            memcpy(p, q, sizeof(p));
        }
        const float mean = p[0] / (float)N;
        for (int l = 0; l < 64; ++l) p[l] = 0.0f;
        for (int e = 0; e < N; ++e) {
            const float d = x[(size_t)(e / Cg) * C + g * Cg + e % Cg] - mean;
            p[e & 63] = fmaf(d, d, p[e & 63]);
        }
        for (int off = 32; off >= 1; off >>= 1) {
            float q[64];
            for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ off];
            memcpy(p, q, sizeof(p));
        }
        const float var = p[0] / (float)N;
        const float inv = ref_rsqrt_canon(var + 1e-3f);
        for (int c = g * Cg; c < (g + 1) * Cg; ++c) {
            const float a = inv * gamma[c];
            const float b = (-mean * a) + beta[c];
            for (int px = 0; px < H * W; ++px) {
                const float z = (x[(size_t)px * C + c] * a) + b;
                y[(size_t)px * C + c] = ref_silu(z);
            }
        }
    }
}

static void conv3(const float *in, int H, int W, int Cin, const float *w, const float *bias, const float *tadd,
                  float *out, int Cout)
{
    float acc[128];
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            for (int co = 0; co < Cout; ++co) acc[co] = 0.0f;
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = y + ky - 1;
                if (yy < 0 || yy >= H) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    const int xx = x + kx - 1;
                    if (xx < 0 || xx >= W) continue;
                    const float *ip = in + ((size_t)yy * W + xx) * Cin;
                    const float *wp = w + (size_t)(ky * 3 + kx) * Cin * Cout;
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
                if (tadd) z = z + tadd[co];
                op[co] = z;
            }
        }
}

static void conv1(const float *in, int npix, int Cin, const float *w, const float *bias, float *out, int Cout)
{
    for (int p = 0; p < npix; ++p)
        for (int co = 0; co < Cout; ++co) {
            float acc = 0.0f;
            for (int c = 0; c < Cin; ++c) acc = fmaf(in[(size_t)p * Cin + c], w[(size_t)c * Cout + co], acc);
            out[(size_t)p * Cout + co] = acc + bias[co];
        }
}

/* ResBlock (:168-193): out = shortcut(x) + conv2(swish(GN(conv1(swish(GN(x))) + time))) */
static void resblock(const ref_bitflip *m, int i, const float *x, int H, int W, const float *tvec, float *out,
                     float *tmp_a, float *tmp_b)
{
    const int Cin = m->cin[i], Cw = m->width[i], npix = H * W;
    gn_silu(x, H, W, Cin, m->g1_g[i], m->g1_b[i], tmp_a);
    conv3(tmp_a, H, W, Cin, m->c1_w[i], m->c1_b[i], tvec, tmp_b, Cw);
    gn_silu(tmp_b, H, W, Cw, m->g2_g[i], m->g2_b[i], tmp_a);
    conv3(tmp_a, H, W, Cw, m->c2_w[i], m->c2_b[i], NULL, tmp_b, Cw);
    if (m->s_w[i]) {
        conv1(x, npix, Cin, m->s_w[i], m->s_b[i], tmp_a, Cw);
        for (size_t k = 0; k < (size_t)npix * Cw; ++k) out[k] = tmp_a[k] + tmp_b[k];
    } else {
        for (size_t k = 0; k < (size_t)npix * Cw; ++k) out[k] = x[k] + tmp_b[k];
    }
}

static void maxpool2(const float *in, int H, int W, int C, float *out)
{
    const int Ho = H / 2, Wo = W / 2;
    for (int y = 0; y < Ho; ++y)
        for (int x = 0; x < Wo; ++x)
            for (int c = 0; c < C; ++c) {
                const float a = in[((size_t)(2 * y) * W + 2 * x) * C + c], b = in[((size_t)(2 * y) * W + 2 * x + 1) * C + c];
                const float d = in[((size_t)(2 * y + 1) * W + 2 * x) * C + c], e = in[((size_t)(2 * y + 1) * W + 2 * x + 1) * C + c];
                out[((size_t)y * Wo + x) * C + c] = fmaxf(fmaxf(a, b), fmaxf(d, e));
            }
}

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

/* x: (B,H,W) bytes 0/1; tvec: 320 floats from ref_bitflip_time; proba: (B,H,W) float32.
 * taps (optional, may be NULL): 5 buffers receiving the ResBlock outputs (B,h_i,w_i,width_i). */
void ref_bitflip_forward(const ref_bitflip *m, const uint8_t *x, const float *tvec, float *proba, int B, int H, int W,
                         float *const *taps)
{
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
    const size_t big = (size_t)H * W * 192;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        float *buf = (float *)malloc(sizeof(float) * big * 8);
        float *x0 = buf, *r0 = buf + big, *r1 = r0 + big, *r2 = r1 + big, *cat = r2 + big, *ta = cat + big, *tb = ta + big,
              *pl = tb + big;
        const uint8_t *xi = x + (size_t)b * H * W;
        for (int p = 0; p < H * W; ++p)
            for (int c = 0; c < 32; ++c) x0[(size_t)p * 32 + c] = fmaf((float)xi[p], m->in_w[c], 0.0f) + m->in_b[c];
        const float *tv[BF_NBLK];
        int o = 0;
        for (int i = 0; i < BF_NBLK; ++i) { tv[i] = tvec + o; o += m->width[i]; }
        resblock(m, 0, x0, H, W, tv[0], r0, ta, tb);                 /* skip1 (H,W,32) */
        maxpool2(r0, H, W, 32, pl);
        resblock(m, 1, pl, H2, W2, tv[1], r1, ta, tb);               /* skip2 (H/2,W/2,64) */
        maxpool2(r1, H2, W2, 64, pl);
        resblock(m, 2, pl, H4, W4, tv[2], r2, ta, tb);               /* (H/4,W/4,128) */
        if (taps) {
            if (taps[0]) memcpy(taps[0] + (size_t)b * H * W * 32, r0, sizeof(float) * H * W * 32);
            if (taps[1]) memcpy(taps[1] + (size_t)b * H2 * W2 * 64, r1, sizeof(float) * H2 * W2 * 64);
            if (taps[2]) memcpy(taps[2] + (size_t)b * H4 * W4 * 128, r2, sizeof(float) * H4 * W4 * 128);
        }
        up_concat(r2, H4, W4, 128, r1, 64, cat);
        resblock(m, 3, cat, H2, W2, tv[3], r2, ta, tb);              /* (H/2,W/2,64) */
        if (taps && taps[3]) memcpy(taps[3] + (size_t)b * H2 * W2 * 64, r2, sizeof(float) * H2 * W2 * 64);
        up_concat(r2, H2, W2, 64, r0, 32, cat);
        resblock(m, 4, cat, H, W, tv[4], r1, ta, tb);                /* (H,W,32) */
        if (taps && taps[4]) memcpy(taps[4] + (size_t)b * H * W * 32, r1, sizeof(float) * H * W * 32);
        for (int p = 0; p < H * W; ++p) {
            float acc = 0.0f;
            for (int c = 0; c < 32; ++c) acc = fmaf(r1[(size_t)p * 32 + c], m->out_w[c], acc);
            acc = acc + m->out_b[0];
            proba[(size_t)b * H * W + p] = 1.0f / (1.0f + ref_cexp(-acc));
        }
        free(buf);
    }
}

/* diffusion_backward (:43-52) with injected uniforms u (T,B,H,W), indexed by t: X ^= (u[t] < p).
 * emb_table: T rows of emb_dim floats (row t = embedding of float32(t)/float32(T)).
 * traj (optional): (T+1,B,H,W) bytes in the reference's order: traj[0] = input, traj[k] = state
 * after k steps. */
void ref_bitflip_sample(const ref_bitflip *m, uint8_t *x, const float *emb_table, int T, const float *u, uint8_t *traj,
                        int B, int H, int W)
{
    const size_t n = (size_t)B * H * W;
    float *p = (float *)malloc(n * sizeof(float));
    float tvec[320];
    if (traj) memcpy(traj, x, n);
    for (int t = T - 1; t >= 0; --t) {
        ref_bitflip_time(m, emb_table + (size_t)t * m->emb_dim, tvec);
        ref_bitflip_forward(m, x, tvec, p, B, H, W, NULL);
        const float *ut = u + (size_t)t * n;
        for (size_t i = 0; i < n; ++i) x[i] ^= (uint8_t)(ut[i] < p[i]);
        if (traj) memcpy(traj + (size_t)(T - t) * n, x, n);
    }
    free(p);
}
