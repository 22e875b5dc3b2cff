// C-ABI of libisingdiff.so (declared in include/isingdiff.h): model handle, weight repacking,
// workspace management and the layer schedule of the UNet forward / reverse sampler.
#include "../../include/isingdiff.h"
#include "idm_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(IDM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));               \
    } while (0)

uint16_t bf16_bits(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// Keras kernel (3,3,Cin,Cout) -> bf16 A-fragments of v_mfma_f32_32x32x16_bf16, [tap][c/16][mt][lane][8]:
// lane l holds W[tap][16*cg + 8*(l>>5) + j][32*mt + (l&31)], j = 0..7.
std::vector<uint16_t> pack_weights_bf16(const float *w, int cin, int cout)
{
    const int ncg = cin / 16, mt = cout / 32;
    std::vector<uint16_t> p((size_t)9 * cin * cout);
    size_t o = 0;
    for (int tap = 0; tap < 9; ++tap)
        for (int cg = 0; cg < ncg; ++cg)
            for (int m = 0; m < mt; ++m)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int c = cg * 16 + 8 * (l >> 5) + j, co = m * 32 + (l & 31);
                        p[o++] = bf16_bits(w[((size_t)tap * cin + c) * cout + co]);
                    }
    return p;
}

// Keras kernel (3,3,Cin,Cout) -> bf16 A-fragments of v_mfma_f32_16x16x32_bf16, [tap][c/32][cout/16][lane][8]:
// lane l holds W[tap][32*cg + 8*(l>>4) + j][16*mt + (l&15)], j = 0..7.
std::vector<uint16_t> pack_weights_bf16_m16(const float *w, int cin, int cout)
{
    const int ncg = cin / 32, mt = cout / 16;
    std::vector<uint16_t> p((size_t)9 * cin * cout);
    size_t o = 0;
    for (int tap = 0; tap < 9; ++tap)
        for (int cg = 0; cg < ncg; ++cg)
            for (int m = 0; m < mt; ++m)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int c = cg * 32 + 8 * (l >> 4) + j, co = m * 16 + (l & 15);
                        p[o++] = bf16_bits(w[((size_t)tap * cin + c) * cout + co]);
                    }
    return p;
}

// Keras kernel -> float A operands of v_mfma_f32_32x32x2_f32, [tap][c/8][mt][lane][4]:
// element s of lane l is W[tap][8*cg + 2*s + (l>>5)][32*mt + (l&31)].
std::vector<float> pack_weights_f32(const float *w, int cin, int cout)
{
    const int ncg = cin / 8, mt = cout / 32;
    std::vector<float> p((size_t)9 * cin * cout);
    size_t o = 0;
    for (int tap = 0; tap < 9; ++tap)
        for (int cg = 0; cg < ncg; ++cg)
            for (int m = 0; m < mt; ++m)
                for (int l = 0; l < 64; ++l)
                    for (int s = 0; s < 4; ++s) {
                        const int c = cg * 8 + 2 * s + (l >> 5), co = m * 32 + (l & 31);
                        p[o++] = w[((size_t)tap * cin + c) * cout + co];
                    }
    return p;
}

// Layer 0 (Cin = 1) as a single A fragment: lane l holds W0[tap = 8*(l>>5) + j][cout = l&31], zero for tap > 8.
std::vector<uint16_t> pack_first_bf16(const float *w, int cout)
{
    std::vector<uint16_t> p(64 * 8, 0);
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * (l >> 5) + j, co = l & 31;
            if (k < 9 && co < cout) p[l * 8 + j] = bf16_bits(w[(size_t)k * cout + co]);
        }
    return p;
}

// Layer 0 of the two-channel model (w: (9, 2, 32) float32, per-cout scale folded in) as bf16 A fragments of two K steps, split
// into (hi, lo) pairs: [part][step][lane][8], lane l: cout l & 31, k = 16 step + 8 (l >> 5) + j = 2 tap + channel, zero for k >= 18.
std::vector<uint16_t> pack_first2_bf16(const float *w, const float *scale)
{
    std::vector<uint16_t> p(4 * 64 * 8, 0);
    for (int st = 0; st < 2; ++st)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * st + 8 * (l >> 5) + j, co = l & 31;
                if (k >= 18) continue;
                const float v = w[(size_t)k * 32 + co] * (scale ? scale[co] : 1.0f);
                const uint16_t hi = bf16_bits(v);
                uint32_t hb = (uint32_t)hi << 16;
                float hf;
                memcpy(&hf, &hb, 4);
                p[((0 * 2 + st) * 64 + l) * 8 + j] = hi;
                p[((1 * 2 + st) * 64 + l) * 8 + j] = bf16_bits(v - hf);
            }
    return p;
}

}  // namespace

// Static workspace of idm_forward_noalloc: lives in the code object, one copy per GPU the library runs on,
// held until the process ends.  In host code the array's name is only a handle for hipGetSymbolAddress; the
// device address is looked up once per device (idm_model_create, with that device current) and kept here.
static __device__ __align__(256) unsigned char scratch[IDM_NOALLOC_SCRATCH_BYTES];

namespace {
char *g_scratch_ptr[IDM_MAX_DEVICES] = {};
std::mutex g_scratch_mutex;   // handles may be created from several threads at once

// Device address of this GPU's copy of `scratch`, resolved on first use (the device must be current).
int scratch_address(int device, char **out)
{
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *ptr = nullptr;
        hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) {
            g_err = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e);
            return IDM_ERR_HIP;
        }
        g_scratch_ptr[device] = (char *)ptr;
    }
    *out = g_scratch_ptr[device];
    return IDM_OK;
}
}

int idm_set_error(int code, const char *msg) { return fail(code, msg ? msg : ""); }

std::vector<float> idm_pack_weights_f32(const float *w, int cin, int cout) { return pack_weights_f32(w, cin, cout); }

std::vector<uint16_t> idm_pack_weights_bf16(const float *w, int cin, int cout) { return pack_weights_bf16(w, cin, cout); }

std::vector<uint16_t> idm_pack_weights_bf16_m16(const float *w, int cin, int cout) { return pack_weights_bf16_m16(w, cin, cout); }

std::vector<float> idm_pack_weights_f32_1x1(const float *w, int cin, int cout)
{
    const int ncg = cin / 8, mt = cout / 32;
    std::vector<float> p((size_t)cin * cout);
    size_t o = 0;
    for (int cg = 0; cg < ncg; ++cg)
        for (int m = 0; m < mt; ++m)
            for (int l = 0; l < 64; ++l)
                for (int s = 0; s < 4; ++s) p[o++] = w[(size_t)(cg * 8 + 2 * s + (l >> 5)) * cout + m * 32 + (l & 31)];
    return p;
}

struct idm_model {
    int device = 0;
    int C = 1, F = 32;
    bool bn = false;
    int cin[IDM_NCONV], cout[IDM_NCONV];
    bool use_fused = true;                 // bf16: fused band kernels where they fit
    int strip_below_rows = 8;              // fused bands shorter than this run as strip kernels instead
    void *w_first_pack = nullptr;          // layer 0 as one bf16 MFMA A fragment (C == 1)
    float *w_first = nullptr;              // (9, C, F) float32
    float *w_head = nullptr;               // (F, C) float32
    void *wpack_f32[IDM_NCONV] = {};       // layers 1..9
    void *wpack_bf16[IDM_NCONV] = {};
    void *wpack_m16[IDM_NCONV] = {};       // layers whose fused kernel has a 16x16x32-MFMA instance
    void *wpack_fold[IDM_NCONV] = {};      // BatchNorm models: bf16 pack of W*scale (strip kernels)
    void *wpack_tr[IDM_NCONV] = {};        // two-channel models: the strip kernels' pack with the 3x3 taps transposed
    void *wpack_tr_m16[IDM_NCONV] = {};    // ... and its 16x16x32-MFMA packing for the layers on 5 x 8 maps (conv4, conv5)
    float *w_first_tr = nullptr;           // ... and layer 0's (9, C, F) kernel likewise
    void *w_first2_pack = nullptr, *w_first2_pack_tr = nullptr;   // C == 2, F == 32: layer 0 as (hi, lo) bf16 MFMA fragments, BatchNorm folded
    float *bias_first2 = nullptr;          // ... and its folded bias
    bool use_transpose = true;             // two-channel models: process the map transposed when that fills the MFMA lanes
    float *bias_fold[IDM_NCONV] = {};      // bias*scale + shift
    float *bias[IDM_NCONV] = {};
    float *scale[IDM_NCONV] = {};
    float *shift[IDM_NCONV] = {};
    float *dense_w[IDM_NDENSE] = {};
    float *dense_b[IDM_NDENSE] = {};
    float *ttable = nullptr;
    int ttable_T = 0;
    float *trow = nullptr;                 // time-embedding row for an arbitrary t_norm
    char *ws = nullptr;                    // activation workspace owned by the handle (grow-only)
    size_t ws_bytes = 0;
    char *ws_active = nullptr;             // workspace of the call in progress: ws, or the device's static scratch array
    std::vector<void *> owned;
    // profiling (idm_profile_begin/end)
    bool profiling = false;
    struct Span { int cls; hipEvent_t a, b; };
    std::vector<Span> spans;
    std::vector<idm_profile_entry> classes;
    std::vector<hipEvent_t> event_pool;
};

namespace {

template <typename T>
int upload(idm_model *m, const T *host, size_t n, T **dev)
{
    HIPCHK(hipMalloc((void **)dev, n * sizeof(T)));
    m->owned.push_back(*dev);
    HIPCHK(hipMemcpy(*dev, host, n * sizeof(T), hipMemcpyHostToDevice));
    return IDM_OK;
}

struct Plan {
    int H, W, Hp, Wp;
    size_t es;  // activation element size
    size_t off_a0, off_s1, off_a2, off_s2, off_a4, off_a5, off_a6, off_a7, off_a8, off_a9, off_score, per_lattice;
    bool fused;          // bf16 band kernels usable for this shape
    int R[5];            // band rows per fused kernel
    bool layers;         // bf16, 2-channel / BatchNorm model: layers 1..9 run as strip kernels (BatchNorm folded)
    bool tr;             // ... on the transposed map: H, W, Hp, Wp are then the transposed (processing) sizes
    bool lpair[5];       // ... except these pairs (FK_M1, FK_M2, FK_M3; FK_B = conv8 + conv9 + the two-channel head, the kernel of
                         // kind FK_BH2), which run as fused band kernels with R[] rows
    bool strip[5];       // this pair of layers runs as single-convolution strip kernels instead
    int sR[SK_COUNT], sWS[SK_COUNT];   // band rows / strip columns of the strip kernels
};

struct StepConst { float c1, c2, sd; };   // x <- ((c1*x) + (c2*score)) + (sd*z)

// Everything the last fused kernel needs to apply the reverse-diffusion update in place.
struct StepCtx {
    bool do_update = false;
    float *x = nullptr;
    const float *noise = nullptr;
    StepConst k = {1.0f, 0.0f, 0.0f};
    bool add_noise = false;
    PhiloxArgs ph = {0, 0, 0};
};

// Largest band height whose LDS footprint leaves room for two workgroups per CU.
int pick_band_rows(int kind, int H, int W)
{
    const int limit = 78 * 1024;
    if (const int tall = fused_tall_rows(kind, H, W)) return tall;   // kinds that run one workgroup per CU
    const int cand[5] = {H, 16, 8, 4, 2};
    for (int i = 0; i < 5; ++i) {
        const int R = cand[i];
        if (R > H || (i > 0 && R >= H)) continue;
        if ((R & 1) && kind != FK_M2) continue;
        int in = 0, mid = 0;
        fused_lds_bytes(kind, W, R, &in, &mid);
        if (in + mid <= limit && fused_band_fits(kind, W, R)) return R;
    }
    return 0;
}

// Band height for a single-convolution strip kernel (strip width = min(32, W)); 0 = not applicable.
int pick_strip_rows(int kind, int H, int W, int *WS)
{
    const int ws = W < 32 ? W : 32;
    *WS = ws;
    // strips tile the width exactly; pooled / upsampled layers need an even strip, the update float4 quads
    const bool even = kind == SK_CONV1 || kind == SK_CONV3 || kind == SK_CONV6 || kind == SK_CONV8;
    if (W % ws != 0 || (even && (ws & 1)) || (kind == SK_CONV9 && (ws & 3))) return 0;
    const int cand[5] = {16, 10, 8, 4, 2};   // (10: the 20-row top level of the transposed spectral map)
    for (int i = 0; i < 5; ++i) {
        const int R = cand[i];
        if (R > H || H % R != 0) continue;
        if (R == 10 && !(ws == 32 && (kind == SK_CONV1 || kind == SK_CONV8 || kind == SK_CONV9H2))) continue;
        int in = 0, st = 0, cs = 0;
        strip_lds_bytes(kind, R, ws, &in, &st, &cs);
        if (in + st + cs <= 78 * 1024 && strip_fits(kind, R, ws)) return R;
    }
    return 0;
}

// allow_tr: the caller does not ask for the per-layer maps (which would come out transposed).
Plan make_plan(const idm_model *m, int H, int W, int precision, bool allow_tr = true)
{
    Plan p;
    p.tr = false;
    {
        // The spectral state of an L x L lattice is L x (L/2 + 1): at L = 32 its padded top level is 32 rows x 20 columns, and
        // a 32-pixel MFMA tile holds one 20-pixel row.  A 3x3 "same" convolution, 2x2 pooling, nearest upsampling and the
        // concatenation all commute with transposition (the kernel's taps transposed), so the whole UNet can run on the
        // transposed map, 20 rows x 32 columns, where every MFMA lane is a pixel.
        const int hp = H + (4 - H % 4) % 4, wp = W + (4 - W % 4) % 4;
        if (allow_tr && precision == IDM_BF16 && m->use_fused && m->use_transpose && m->C == 2 && m->w_first_tr &&
            hp % 32 == 0 && wp % 32 != 0 && wp < hp) {
            p.tr = true;
            std::swap(H, W);
        }
    }
    p.H = H; p.W = W;
    p.Hp = H + (4 - H % 4) % 4;
    p.Wp = W + (4 - W % 4) % 4;
    p.es = precision == IDM_BF16 ? 2 : 4;
    for (int k = 0; k < 5; ++k) { p.R[k] = 0; p.strip[k] = false; p.lpair[k] = false; }
    // the fused band kernels cover the real-space model (one state channel, no BatchNorm)
    p.fused = precision == IDM_BF16 && m->C == 1 && !m->bn && m->use_fused && m->w_first_pack != nullptr;
    if (p.fused) {
        const int hh[5] = {p.Hp, p.Hp / 2, p.Hp / 4, p.Hp / 2, p.Hp}, ww[5] = {p.Wp, p.Wp / 2, p.Wp / 4, p.Wp / 2, p.Wp};
        // layers of each fused kernel as strip kernels: {conv1}, {conv2, conv3}, {conv4, conv5}, {conv6, conv7}, {conv8, conv9}
        const int first[5] = {SK_CONV1, SK_CONV2, SK_CONV4, SK_CONV6, SK_CONV8}, count[5] = {1, 2, 2, 2, 2};
        const bool quads = (p.W & 3) == 0 && p.H == p.Hp && p.W == p.Wp;
        for (int k = 0; k < 5; ++k) {
            p.R[k] = pick_band_rows(k, hh[k], ww[k]);
            bool ok = quads && k != FK_A;   // conv0 has no strip kernel: fused_A runs at any band height that fits
            for (int j = 0; j < count[k]; ++j) {
                const int sk = first[k] + j;
                p.sR[sk] = pick_strip_rows(sk, hh[k], ww[k], &p.sWS[sk]);
                ok = ok && p.sR[sk] > 0;
            }
            // full-height fused bands where they fit; strips where the band would have to shrink
            // (conv2 + conv3 stay fused down to 4-row bands: 290 us against 373 us as two strip launches at L = 64)
            const int below = k == FK_M1 ? std::min(m->strip_below_rows, 4) : m->strip_below_rows;
            p.strip[k] = ok && (p.R[k] < below && p.R[k] < hh[k]);
            if (p.R[k] == 0 && !p.strip[k]) p.fused = false;
        }
        // conv2 + conv3 of the 32 x 32 lattice: the instance that takes the whole 16 x 16 map
        if (p.fused && !p.strip[FK_M1] && p.H == p.Hp && p.W == p.Wp)
            if (const int r = fused_m1_whole_rows(p.H, p.W)) p.R[FK_M1] = r;
    }
    p.layers = false;
    if (precision == IDM_BF16 && !p.fused && m->use_fused && (m->C == 2 || m->bn)) {
        const int hh[3] = {p.Hp, p.Hp / 2, p.Hp / 4}, ww[3] = {p.Wp, p.Wp / 2, p.Wp / 4};
        const int lvl[SK_COUNT] = {0, 1, 1, 2, 2, 1, 1, 0, 0, 0, 0};
        p.layers = true;
        for (int sk = SK_CONV1; sk < SK_COUNT; ++sk)
            p.sR[sk] = sk == SK_CONV9 ? 0 : pick_strip_rows(sk, hh[lvl[sk]], ww[lvl[sk]], &p.sWS[sk]);
        // the three inner pairs have the real-space model's shapes: fused band kernels where a band of at least 8 rows
        // (or the whole map) fits
        const int lv[5] = {0, 1, 2, 1, 0};
        for (int k = FK_M1; k <= FK_M3; ++k) {
            p.R[k] = pick_band_rows(k, hh[lv[k]], ww[lv[k]]);
            p.lpair[k] = p.R[k] > 0 && (p.R[k] >= 8 || p.R[k] >= hh[lv[k]]);
        }
        // the last pair with the two-channel head (score only): 1 078 -> about 800 us per 16 384 spectra against the two strip
        // launches (ISINGDIFF_SPECTRAL_BPAIR=0 keeps those; read once per process)
        static const bool bpair = [] { const char *e = getenv("ISINGDIFF_SPECTRAL_BPAIR"); return !(e && e[0] == '0'); }();
        if (m->C == 2 && bpair) {
            p.R[FK_B] = pick_band_rows(FK_BH2, hh[0], ww[0]);
            p.lpair[FK_B] = p.R[FK_B] >= 8;
        }
        // Every layer needs its strip kernel (a caller that asks for the per-layer maps gets the strip launches) - except,
        // on the transposed map, where nobody can ask for them, the layers of a pair that runs fused and the stored conv9.
        const int pair_of[SK_COUNT] = {0, FK_M1, FK_M1, FK_M2, FK_M2, FK_M3, FK_M3, 0, 0, 0, 0};
        for (int sk = SK_CONV1; sk < SK_COUNT; ++sk) {
            if (sk == SK_CONV9) continue;
            const bool spared = p.tr && ((pair_of[sk] && p.lpair[pair_of[sk]]) || sk == SK_CONV9S);
            if (p.sR[sk] == 0 && !spared) p.layers = false;
        }
        if (!p.layers)
            for (int k = FK_M1; k <= FK_B; ++k) { p.R[k] = 0; p.lpair[k] = false; }
    }
    if (p.tr && !p.layers) return make_plan(m, W, H, precision, false);   // (only the strip schedule has transposed weights)
    // Workspace: one buffer per kind, sized for what the selected schedule stores in it.  The fused schedule
    // keeps only the maps that cross a kernel boundary (skip1 and its pool, skip2 and its pool, conv5, conv7:
    // 172 KB per lattice at L = 32 in bf16), plus one intermediate per pair of layers that runs as strips.
    const size_t n1 = (size_t)p.Hp * p.Wp, n2 = n1 / 4, n4 = n1 / 16, F = m->F;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    if (p.fused) {
        p.off_a0 = take(n2 * F * p.es);                                                   // pooled skip1
        p.off_s1 = take(n1 * F * p.es);
        p.off_a2 = take(n4 * 2 * F * p.es);                                               // pooled skip2
        p.off_s2 = take(n2 * 2 * F * p.es);
        p.off_a4 = take(p.strip[FK_M2] ? n4 * 4 * F * p.es : 0);                          // conv4 between strips
        p.off_a5 = take(n4 * 4 * F * p.es);
        p.off_a6 = take(p.strip[FK_M1] || p.strip[FK_M3] ? n2 * 2 * F * p.es : 0);        // conv2 / conv6 between strips
        p.off_a7 = take(n2 * 2 * F * p.es);
        p.off_a8 = take(p.strip[FK_B] ? n1 * F * p.es : 0);                               // conv8 between strips
        p.off_a9 = take(0);
    } else {
        p.off_a0 = take(n1 * F * p.es);
        p.off_s1 = take(n1 * F * p.es);
        p.off_a2 = take(n2 * 2 * F * p.es);
        p.off_s2 = take(n2 * 2 * F * p.es);
        p.off_a4 = take(n4 * 4 * F * p.es);
        p.off_a5 = take(n4 * 4 * F * p.es);
        p.off_a6 = take(n2 * 2 * F * p.es);
        p.off_a7 = take(n2 * 2 * F * p.es);
        p.off_a8 = take(n1 * F * p.es);
        p.off_a9 = take(n1 * F * 4);
    }
    p.off_score = take((size_t)H * W * m->C * 4);
    p.per_lattice = o;
    return p;
}

int ensure_ws(idm_model *m, size_t bytes)
{
    m->ws_active = m->ws;
    if (m->ws_bytes >= bytes) return IDM_OK;
    if (m->ws) {
        // work enqueued through this handle may still read the old workspace
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipFree(m->ws));
        m->ws = nullptr; m->ws_bytes = 0; m->ws_active = nullptr;
    }
    hipError_t e = hipMalloc((void **)&m->ws, bytes);
    if (e != hipSuccess) return fail(IDM_ERR_NOMEM, "workspace allocation of " + std::to_string(bytes) + " bytes failed");
    m->ws_bytes = bytes;
    m->ws_active = m->ws;
    return IDM_OK;
}

// Constants of the reverse step x <- ((c1*x) + (c2*score)) + (sd*z), each operation rounded to float32 on the host
// (volatile keeps the compiler from contracting or widening them).
//   IDM_UPDATE_SCORE (b_diffusion_model_unet_2d_ising.py:96): c1 = 1 + 0.5*beta, c2 = beta
//   IDM_UPDATE_DDPM  (:95): c1 = 1/sqrt(1 - beta), c2 = -(c1 * (beta / sqrt(1 - alpha_cumprod)))
StepConst step_constants(int rule, float beta, float alpha_cumprod)
{
    StepConst k;
    volatile float root = sqrtf(beta);
    k.sd = root;
    if (rule == IDM_UPDATE_DDPM) {
        volatile float om = 1.0f - beta;
        volatile float rs = sqrtf(om);
        volatile float a = 1.0f / rs;
        volatile float oma = 1.0f - alpha_cumprod;
        volatile float d = sqrtf(oma);
        volatile float e = beta / d;
        volatile float ae = a * e;
        k.c1 = a;
        k.c2 = -ae;
    } else {
        volatile float half_beta = 0.5f * beta;
        volatile float one_plus = 1.0f + half_beta;
        k.c1 = one_plus;
        k.c2 = beta;
    }
    return k;
}

int prof_class(idm_model *m, const char *name)
{
    for (size_t i = 0; i < m->classes.size(); ++i)
        if (!strcmp(m->classes[i].name, name)) return (int)i;
    idm_profile_entry e;
    memset(&e, 0, sizeof(e));
    snprintf(e.name, sizeof(e.name), "%s", name);
    m->classes.push_back(e);
    return (int)m->classes.size() - 1;
}

hipEvent_t prof_event(idm_model *m)
{
    if (!m->event_pool.empty()) { hipEvent_t e = m->event_pool.back(); m->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

// Runs `launch` and, when profiling, brackets it with events on stream s.
template <typename F>
void timed(idm_model *m, const char *name, double flop, double bytes, hipStream_t s, F &&launch)
{
    if (!m->profiling) { launch(); return; }
    const int cls = prof_class(m, name);
    m->classes[cls].launches += 1;
    m->classes[cls].flop += flop;
    m->classes[cls].bytes += bytes;
    idm_model::Span sp{cls, prof_event(m), prof_event(m)};
    (void)hipEventRecord(sp.a, s);
    launch();
    (void)hipEventRecord(sp.b, s);
    m->spans.push_back(sp);
}

// Lattices per chunk: the fewest equal chunks whose activation workspace stays within `budget` bytes
// (equal, so that every launch of a kernel class does the same work).
int64_t auto_chunk(const Plan &p, int64_t B, size_t budget = (size_t)6 << 30)
{
    const int64_t cmax = (int64_t)std::max<size_t>(1, budget / p.per_lattice);
    if (B <= cmax) return B;
    const int64_t nchunks = (B + cmax - 1) / cmax;
    return (B + nchunks - 1) / nchunks;
}

// Forward for `Bc` lattices whose workspace starts at m->ws_active.  Buffers of one kind are contiguous
// over the chunk: base + off_kind*Bc.
int forward_fused(idm_model *m, const Plan &p, const float *x, const float *tadd, float *score, int64_t Bc,
                  const StepCtx &ctx, float *const *taps, int64_t tap_b0, hipStream_t s)
{
    char *ws = m->ws_active;
    auto buf = [&](size_t off) { return (void *)(ws + off * (size_t)Bc); };
    void *s1 = buf(p.off_s1), *p1 = buf(p.off_a0), *s2 = buf(p.off_s2), *p2 = buf(p.off_a2), *a5 = buf(p.off_a5),
         *a7 = buf(p.off_a7);
    const int F = m->F, Hp = p.Hp, Wp = p.Wp;
    const double n1 = (double)Bc * Hp * Wp, n2 = n1 / 4, n4 = n1 / 16;
    auto run = [&](int kind, const char *name, FusedArgs &a, int H, int W, double flop, double bytes) -> int {
        a.H = H; a.W = W; a.R = p.R[kind]; a.nbands = (H + a.R - 1) / a.R;
        a.Hx = p.H; a.Wx = p.W; a.B = Bc;
        fused_lds_bytes(kind, W, a.R, &a.lds_in_bytes, &a.lds_mid_bytes);
        hipError_t e = hipSuccess;
        timed(m, name, flop, bytes, s, [&] { e = launch_fused(kind, a, Bc, s); });
        if (e != hipSuccess) return fail(IDM_ERR_HIP, std::string("launch ") + name + ": " + hipGetErrorString(e));
        return IDM_OK;
    };
    // one layer as a strip kernel: sk = SK_CONV<l>
    auto strip = [&](int sk, int l, const char *name, const void *in_a, const void *in_b, void *out, void *out_pool,
                     int H, int W, double flop, double bytes, bool head) -> int {
        StripArgs a = {};
        a.H = H; a.W = W; a.R = p.sR[sk]; a.WS = p.sWS[sk]; a.nbands = H / a.R; a.nstrips = W / a.WS;
        a.Hx = p.H; a.Wx = p.W; a.B = Bc;
        a.in_a = in_a; a.in_b = in_b; a.w = m->wpack_bf16[l]; a.bias = m->bias[l]; a.tadd = l == 4 ? tadd : nullptr;
        a.out = out; a.out_pool = out_pool;
        if (head) {
            a.w_head = m->w_head; a.b_head = m->bias[10]; a.score_out = score;
            a.do_update = ctx.do_update ? 1 : 0;
            a.x = ctx.x; a.noise = ctx.noise; a.add_noise = ctx.add_noise ? 1 : 0; a.ph = ctx.ph;
            a.c1 = ctx.k.c1; a.beta = ctx.k.c2; a.sd = ctx.k.sd;
        }
        strip_lds_bytes(sk, a.R, a.WS, &a.lds_in_bytes, &a.lds_stage_bytes, &a.lds_const_bytes);
        hipError_t e = hipSuccess;
        timed(m, name, flop, bytes, s, [&] { e = launch_strip(sk, a, s); });
        if (e != hipSuccess) return fail(IDM_ERR_HIP, std::string("launch ") + name + ": " + hipGetErrorString(e));
        return IDM_OK;
    };
    void *c2 = buf(p.off_a6), *c4 = buf(p.off_a4), *c6 = buf(p.off_a6), *c8 = buf(p.off_a8);
    auto layer = [&](FusedArgs &a, int l1, int l2) {
        a.w1 = l1 == 0 ? m->w_first_pack : m->wpack_bf16[l1];
        a.w1_alt = m->wpack_m16[l1];
        a.w2_alt = m->wpack_m16[l2];
        a.b1 = m->bias[l1];
        a.w2 = m->wpack_bf16[l2]; a.b2 = m->bias[l2];
    };
    int rc;
    {   // x -> conv0 -> conv1 -> skip1 (+ pooled)
        FusedArgs a = {};
        layer(a, 0, 1);
        a.in_a = x; a.out = s1; a.out_pool = p1;
        rc = run(FK_A, "fused_A(conv0,1)", a, Hp, Wp, 2.0 * 9 * (F + F * F) * n1, (double)Bc * p.H * p.W * 4 + n1 * F * 2 + n2 * F * 2);
        if (rc != IDM_OK) return rc;
    }
    if (p.strip[FK_M1]) {
        rc = strip(SK_CONV2, 2, "strip_conv2", p1, nullptr, c2, nullptr, Hp / 2, Wp / 2, 2.0 * 9 * F * 2 * F * n2, n2 * F * 2 + n2 * 2 * F * 2, false);
        if (rc == IDM_OK)
            rc = strip(SK_CONV3, 3, "strip_conv3", c2, nullptr, s2, p2, Hp / 2, Wp / 2, 2.0 * 9 * 4 * F * F * n2, n2 * 2 * F * 4 + n4 * 2 * F * 2, false);
        if (rc != IDM_OK) return rc;
    } else {   // pooled -> conv2 -> conv3 -> skip2 (+ pooled)
        FusedArgs a = {};
        layer(a, 2, 3);
        a.in_a = p1; a.out = s2; a.out_pool = p2;
        rc = run(FK_M1, "fused_M1(conv2,3)", a, Hp / 2, Wp / 2, 2.0 * 9 * (F * 2 * F + 4 * F * F) * n2, n2 * F * 2 + n2 * 2 * F * 2 + n4 * 2 * F * 2);
        if (rc != IDM_OK) return rc;
    }
    // conv4 + conv5 on 16 x 16 maps (L = 64): the whole-map 16x16x32 kernel of idm_fused_split.hip instead of two strip launches
    // (ISINGDIFF_M2_MFMA16=0: the strips; read once per process)
    static const bool m2_mf16 = [] { const char *e = getenv("ISINGDIFF_M2_MFMA16"); return !(e && e[0] == '0'); }();
    if (p.strip[FK_M2] && m2_mf16 && Hp / 4 == 16 && Wp / 4 == 16 && F == 32 && m->wpack_m16[4] && m->wpack_m16[5]) {
        FusedArgs a = {};
        layer(a, 4, 5);
        a.tadd1 = tadd;
        a.in_a = p2; a.out = a5;
        a.H = 16; a.W = 16; a.Hx = p.H; a.Wx = p.W; a.B = Bc;
        hipError_t e = hipSuccess;
        timed(m, "fused_M2(conv4,5)", 2.0 * 9 * (8 * F * F + 16 * F * F) * n4, n4 * 2 * F * 2 + n4 * 4 * F * 2, s,
              [&] { e = launch_m1_mf16(a, Bc, s, true); });
        if (e != hipSuccess) return fail(IDM_ERR_HIP, std::string("launch fused_M2(conv4,5): ") + hipGetErrorString(e));
    } else if (p.strip[FK_M2]) {
        rc = strip(SK_CONV4, 4, "strip_conv4", p2, nullptr, c4, nullptr, Hp / 4, Wp / 4, 2.0 * 9 * 8 * F * F * n4, n4 * 2 * F * 2 + n4 * 4 * F * 2, false);
        if (rc == IDM_OK)
            rc = strip(SK_CONV5, 5, "strip_conv5", c4, nullptr, a5, nullptr, Hp / 4, Wp / 4, 2.0 * 9 * 16 * F * F * n4, n4 * 4 * F * 4, false);
        if (rc != IDM_OK) return rc;
    } else {   // pooled -> conv4 (+time) -> conv5
        FusedArgs a = {};
        layer(a, 4, 5);
        a.tadd1 = tadd;
        a.in_a = p2; a.out = a5;
        rc = run(FK_M2, "fused_M2(conv4,5)", a, Hp / 4, Wp / 4, 2.0 * 9 * (8 * F * F + 16 * F * F) * n4, n4 * 2 * F * 2 + n4 * 4 * F * 2);
        if (rc != IDM_OK) return rc;
    }
    if (p.strip[FK_M3]) {
        rc = strip(SK_CONV6, 6, "strip_conv6", a5, s2, c6, nullptr, Hp / 2, Wp / 2, 2.0 * 9 * 12 * F * F * n2, n4 * 4 * F * 2 + n2 * 2 * F * 4, false);
        if (rc == IDM_OK)
            rc = strip(SK_CONV7, 7, "strip_conv7", c6, nullptr, a7, nullptr, Hp / 2, Wp / 2, 2.0 * 9 * 4 * F * F * n2, n2 * 2 * F * 4, false);
        if (rc != IDM_OK) return rc;
    } else {   // [up(conv5), skip2] -> conv6 -> conv7
        FusedArgs a = {};
        layer(a, 6, 7);
        a.in_a = a5; a.in_b = s2; a.out = a7;
        rc = run(FK_M3, "fused_M3(conv6,7)", a, Hp / 2, Wp / 2, 2.0 * 9 * (12 * F * F + 4 * F * F) * n2, n4 * 4 * F * 2 + n2 * 2 * F * 2 + n2 * 2 * F * 2);
        if (rc != IDM_OK) return rc;
    }
    if (p.strip[FK_B]) {
        rc = strip(SK_CONV8, 8, "strip_conv8", a7, s1, c8, nullptr, Hp, Wp, 2.0 * 9 * 3 * F * F * n1, n2 * 2 * F * 2 + n1 * F * 4, false);
        if (rc == IDM_OK)
            rc = strip(SK_CONV9, 9, "strip_conv9(head,step)", c8, nullptr, nullptr, nullptr, Hp, Wp, 2.0 * 9 * F * F * n1 + 2.0 * F * n1,
                       n1 * F * 2 + (double)Bc * p.H * p.W * 8, true);
        if (rc != IDM_OK) return rc;
    } else {   // [up(conv7), skip1] -> conv8 -> conv9 -> head -> (update)
        FusedArgs a = {};
        layer(a, 8, 9);
        a.in_a = a7; a.in_b = s1;
        a.w_head = m->w_head; a.b_head = m->bias[10];
        a.score_out = score;
        a.do_update = ctx.do_update ? 1 : 0;
        a.x = ctx.x; a.noise = ctx.noise; a.add_noise = ctx.add_noise ? 1 : 0; a.ph = ctx.ph;
        a.c1 = ctx.k.c1; a.beta = ctx.k.c2; a.sd = ctx.k.sd;
        rc = run(FK_B, "fused_B(conv8,9,head,step)", a, Hp, Wp, 2.0 * 9 * (3 * F * F + F * F) * n1 + 2.0 * F * n1,
                 n2 * 2 * F * 2 + n1 * F * 2 + (double)Bc * p.H * p.W * 8);
        if (rc != IDM_OK) return rc;
    }
    if (taps) {
        const size_t c1 = (size_t)Hp * Wp, c2 = c1 / 4, c4 = c1 / 16;
        const void *src[10] = {nullptr, s1, nullptr, s2, nullptr, a5, nullptr, a7, nullptr, nullptr};
        const size_t cnt[10] = {c1 * F, c1 * F, c2 * 2 * F, c2 * 2 * F, c4 * 4 * F, c4 * 4 * F, c2 * 2 * F, c2 * 2 * F, c1 * F, c1 * F};
        for (int i = 0; i < 10; ++i) {
            if (!taps[i]) continue;
            float *dst = taps[i] + (size_t)tap_b0 * cnt[i];
            if (src[i]) launch_cast_bf16_to_f32(src[i], dst, (int64_t)(cnt[i] * Bc), s);
            else HIPCHK(hipMemsetAsync(dst, 0xFF, cnt[i] * Bc * 4, s));  // NaN pattern: not materialised by the fused path
        }
        if (taps[10] && score)
            HIPCHK(hipMemcpyAsync(taps[10] + (size_t)tap_b0 * p.H * p.W * m->C, score, (size_t)Bc * p.H * p.W * m->C * 4,
                                  hipMemcpyDeviceToDevice, s));
    }
    HIPCHK(hipGetLastError());
    return IDM_OK;
}

int forward_chunk(idm_model *m, const Plan &p, const float *x, const float *tadd, float *score, int64_t Bc,
                  int precision, float *const *taps, int64_t tap_b0, hipStream_t s)
{
    char *ws = m->ws_active;
    auto buf = [&](size_t off) { return (void *)(ws + off * (size_t)Bc); };
    void *a0 = buf(p.off_a0), *s1 = buf(p.off_s1), *a2 = buf(p.off_a2), *s2 = buf(p.off_s2), *a4 = buf(p.off_a4),
         *a5 = buf(p.off_a5), *a6 = buf(p.off_a6), *a7 = buf(p.off_a7), *a8 = buf(p.off_a8);
    float *a9 = (float *)buf(p.off_a9);
    const bool bf = precision == IDM_BF16;
    const int F = m->F, Hp = p.Hp, Wp = p.Wp;

    const double npx = (double)Bc * Hp * Wp;
    timed(m, "conv_first", 2.0 * 9 * m->C * F * npx, (double)Bc * p.H * p.W * m->C * 4 + npx * F * p.es, s, [&] {
        // two-channel model: layer 0 on the matrix cores (ISINGDIFF_CONV0_MFMA=0: the vector-ALU kernel; read once per process)
        static const bool conv0_mfma = [] { const char *e = getenv("ISINGDIFF_CONV0_MFMA"); return !(e && e[0] == '0'); }();
        const void *w2p = p.tr ? m->w_first2_pack_tr : m->w_first2_pack;
        if (bf && conv0_mfma && m->C == 2 && w2p && launch_conv_first2_mfma(x, Bc, p.H, p.W, Hp, Wp, w2p, m->bias_first2, a0, s, p.tr)) {}
        else if (bf) launch_conv_first_bf16(x, Bc, p.H, p.W, Hp, Wp, m->C, p.tr ? m->w_first_tr : m->w_first, m->bias[0], m->scale[0], m->shift[0], a0, s, p.tr);
        else launch_conv_first(x, Bc, p.H, p.W, Hp, Wp, m->C, m->w_first, m->bias[0], m->scale[0], m->shift[0], a0, F, false, s);
    });

    auto conv = [&](int layer, const void *in0, const void *in1, int C0, int C1, int H, int W, int mode, void *out,
                    const float *ta, bool out_f32) {
        ConvArgs a;
        a.in0 = in0; a.in1 = in1; a.C0 = C0; a.C1 = C1; a.H = H; a.W = W; a.B = Bc;
        a.wpack = bf ? m->wpack_bf16[layer] : m->wpack_f32[layer];
        a.bias = m->bias[layer]; a.scale = m->scale[layer]; a.shift = m->shift[layer]; a.tadd = ta;
        a.out = out; a.Cout = m->cout[layer]; a.mode = mode; a.silu = 1; a.out_f32 = out_f32 ? 1 : 0;
        char name[32];
        snprintf(name, sizeof(name), "conv%d_%s", layer, bf ? "bf16" : "f32");
        const double px = (double)Bc * H * W;
        const double in_px = mode == CONV_POOL ? 4.0 * C0 : (mode == CONV_UPCAT ? 0.25 * C0 + C1 : (double)C0);
        timed(m, name, 2.0 * 9 * (C0 + C1) * a.Cout * px, px * (in_px * p.es + a.Cout * (out_f32 ? 4 : p.es)), s,
              [&] { if (bf) launch_conv_bf16(a, s); else launch_conv_f32(a, s); });
    };
    if (bf && p.layers) {
        // layers 1..9 as strip kernels (BatchNorm folded into weights and bias); pooled maps live in the
        // buffers of layers that run later (a4 before conv4, a5 before conv5)
        int rc = IDM_OK;
        // the two-channel head of the spectral model rides in conv9's epilogue unless the caller wants conv9's map itself
        const bool fuse_head = m->C == 2 && !taps;
        auto strip = [&](int sk, int l, const void *in_a, const void *in_b, void *out, void *out_pool, int H, int W) {
            if (rc != IDM_OK) return;
            StripArgs a = {};
            if (sk == SK_CONV9H2) { a.w_head = m->w_head; a.b_head = m->bias[10]; a.score_out = score; a.transpose_out = p.tr ? 1 : 0; }
            a.H = H; a.W = W; a.R = p.sR[sk]; a.WS = p.sWS[sk]; a.nbands = H / a.R; a.nstrips = W / a.WS;
            a.Hx = p.H; a.Wx = p.W; a.B = Bc;
            a.in_a = in_a; a.in_b = in_b;
            a.w = p.tr ? m->wpack_tr[l] : m->bn ? m->wpack_fold[l] : m->wpack_bf16[l];
            a.bias = m->bn ? m->bias_fold[l] : m->bias[l];
            a.tadd = l == 4 ? tadd : nullptr;
            a.out = out; a.out_pool = out_pool;
            strip_lds_bytes(sk, a.R, a.WS, &a.lds_in_bytes, &a.lds_stage_bytes, &a.lds_const_bytes);
            char name[32];
            snprintf(name, sizeof(name), "strip_conv%d", l);
            const double px = (double)Bc * H * W;
            hipError_t e = hipSuccess;
            timed(m, name, 2.0 * 9 * m->cin[l] * m->cout[l] * px, px * (m->cin[l] + m->cout[l]) * 2, s,
                  [&] { e = launch_strip(sk, a, s); });
            if (e != hipSuccess) rc = fail(IDM_ERR_HIP, std::string("launch ") + name + ": " + hipGetErrorString(e));
        };
        // a pair of layers as one fused band kernel (same weights, bias and operand order as the two strip launches)
        auto pair = [&](int kind, int l1, const void *in_a, const void *in_b, void *out, void *out_pool, int H, int W) {
            if (rc != IDM_OK) return;
            FusedArgs a = {};
            a.w1 = p.tr ? m->wpack_tr[l1] : m->bn ? m->wpack_fold[l1] : m->wpack_bf16[l1];
            a.b1 = m->bn ? m->bias_fold[l1] : m->bias[l1];
            a.w2 = p.tr ? m->wpack_tr[l1 + 1] : m->bn ? m->wpack_fold[l1 + 1] : m->wpack_bf16[l1 + 1];
            a.b2 = m->bn ? m->bias_fold[l1 + 1] : m->bias[l1 + 1];
            a.tadd1 = l1 == 4 ? tadd : nullptr;
            if (p.tr) { a.w1_alt = m->wpack_tr_m16[l1]; a.w2_alt = m->wpack_tr_m16[l1 + 1]; }
            a.in_a = in_a; a.in_b = in_b; a.out = out; a.out_pool = out_pool;
            if (kind == FK_BH2) { a.w_head = m->w_head; a.b_head = m->bias[10]; a.score_out = score; a.transpose_out = p.tr ? 1 : 0; }
            a.H = H; a.W = W; a.R = p.R[kind == FK_BH2 ? FK_B : kind]; a.nbands = (H + a.R - 1) / a.R;
            a.Hx = p.H; a.Wx = p.W; a.B = Bc;
            fused_lds_bytes(kind, W, a.R, &a.lds_in_bytes, &a.lds_mid_bytes);
            char name[32];
            snprintf(name, sizeof(name), "fused_conv%d,%d", l1, l1 + 1);
            const double px = (double)Bc * H * W;
            hipError_t e = hipSuccess;
            timed(m, name, 2.0 * 9 * (m->cin[l1] * m->cout[l1] + m->cin[l1 + 1] * m->cout[l1 + 1]) * px,
                  px * (m->cin[l1] + m->cout[l1 + 1]) * 2, s, [&] { e = launch_fused(kind, a, Bc, s); });
            if (e != hipSuccess) rc = fail(IDM_ERR_HIP, std::string("launch ") + name + ": " + hipGetErrorString(e));
        };
        strip(SK_CONV1, 1, a0, nullptr, s1, a4, Hp, Wp);
        // (a caller that wants the intermediate maps gets the strip launches, which store them)
        const bool pm1 = p.lpair[FK_M1] && !taps, pm2 = p.lpair[FK_M2] && !taps, pm3 = p.lpair[FK_M3] && !taps;
        if (pm1) pair(FK_M1, 2, a4, nullptr, s2, a5, Hp / 2, Wp / 2);
        else {
            strip(SK_CONV2, 2, a4, nullptr, a2, nullptr, Hp / 2, Wp / 2);
            strip(SK_CONV3, 3, a2, nullptr, s2, a5, Hp / 2, Wp / 2);
        }
        if (pm2) pair(FK_M2, 4, a5, nullptr, a4, nullptr, Hp / 4, Wp / 4);   // (conv5's map in a4 instead of a5)
        else {
            strip(SK_CONV4, 4, a5, nullptr, a4, nullptr, Hp / 4, Wp / 4);
            strip(SK_CONV5, 5, a4, nullptr, a5, nullptr, Hp / 4, Wp / 4);
        }
        const void *c5 = pm2 ? a4 : a5;
        if (pm3) pair(FK_M3, 6, c5, s2, a7, nullptr, Hp / 2, Wp / 2);
        else {
            strip(SK_CONV6, 6, c5, s2, a6, nullptr, Hp / 2, Wp / 2);
            strip(SK_CONV7, 7, a6, nullptr, a7, nullptr, Hp / 2, Wp / 2);
        }
        if (fuse_head && p.lpair[FK_B]) pair(FK_BH2, 8, a7, s1, nullptr, nullptr, Hp, Wp);
        else {
            strip(SK_CONV8, 8, a7, s1, a8, nullptr, Hp, Wp);
            if (fuse_head) strip(SK_CONV9H2, 9, a8, nullptr, nullptr, nullptr, Hp, Wp);
            else strip(SK_CONV9S, 9, a8, nullptr, a9, nullptr, Hp, Wp);
        }
        if (rc != IDM_OK) return rc;
        if (!fuse_head)
            timed(m, "head", 2.0 * F * m->C * Bc * p.H * p.W, (double)Bc * p.H * p.W * (F * 2 + m->C * 4), s,
                  [&] { launch_head_bf16(a9, Bc, p.H, p.W, Hp, Wp, F, m->C, m->w_head, m->bias[10], score, s); });
    } else {
    if (p.tr) return fail(IDM_ERR_HIP, "internal: transposed plan without the strip schedule");
    conv(1, a0, nullptr, F, 0, Hp, Wp, CONV_PLAIN, s1, nullptr, false);
    conv(2, s1, nullptr, F, 0, Hp / 2, Wp / 2, CONV_POOL, a2, nullptr, false);
    conv(3, a2, nullptr, 2 * F, 0, Hp / 2, Wp / 2, CONV_PLAIN, s2, nullptr, false);
    conv(4, s2, nullptr, 2 * F, 0, Hp / 4, Wp / 4, CONV_POOL, a4, tadd, false);
    conv(5, a4, nullptr, 4 * F, 0, Hp / 4, Wp / 4, CONV_PLAIN, a5, nullptr, false);
    conv(6, a5, s2, 4 * F, 2 * F, Hp / 2, Wp / 2, CONV_UPCAT, a6, nullptr, false);
    conv(7, a6, nullptr, 2 * F, 0, Hp / 2, Wp / 2, CONV_PLAIN, a7, nullptr, false);
    conv(8, a7, s1, 2 * F, F, Hp, Wp, CONV_UPCAT, a8, nullptr, false);
    conv(9, a8, nullptr, F, 0, Hp, Wp, CONV_PLAIN, a9, nullptr, true);
    timed(m, "head", 2.0 * F * m->C * Bc * p.H * p.W, (double)Bc * p.H * p.W * (F * 4 + m->C * 4), s,
          [&] { launch_head(a9, Bc, p.H, p.W, Hp, Wp, F, m->C, m->w_head, m->bias[10], score, s); });
    }

    if (taps) {
        const size_t n1 = (size_t)Hp * Wp, n2 = n1 / 4, n4 = n1 / 16;
        const void *src[10] = {a0, s1, a2, s2, a4, a5, a6, a7, a8, a9};
        const size_t cnt[10] = {n1 * F, n1 * F, n2 * 2 * F, n2 * 2 * F, n4 * 4 * F, n4 * 4 * F, n2 * 2 * F, n2 * 2 * F, n1 * F, n1 * F};
        for (int i = 0; i < 10; ++i) {
            if (!taps[i]) continue;
            float *dst = taps[i] + (size_t)tap_b0 * cnt[i];
            if (bf && (i != 9 || p.layers)) launch_cast_bf16_to_f32(src[i], dst, (int64_t)(cnt[i] * Bc), s);
            else HIPCHK(hipMemcpyAsync(dst, src[i], cnt[i] * Bc * 4, hipMemcpyDeviceToDevice, s));
        }
        if (taps[10]) {
            if (p.H != Hp || p.W != Wp) return fail(IDM_ERR_UNSUPPORTED, "tap 10 needs H and W to be multiples of 4");
            HIPCHK(hipMemcpyAsync(taps[10] + (size_t)tap_b0 * p.H * p.W * m->C, score, (size_t)Bc * p.H * p.W * m->C * 4,
                                  hipMemcpyDeviceToDevice, s));
        }
    }
    HIPCHK(hipGetLastError());
    return IDM_OK;
}

int check_shape(const idm_model *m, int64_t B, int H, int W, int precision)
{
    if (!m) return fail(IDM_ERR_INVALID, "null model");
    if (B <= 0 || H <= 0 || W <= 0) return fail(IDM_ERR_INVALID, "B, H and W must be positive");
    if (precision != IDM_FP32 && precision != IDM_BF16) return fail(IDM_ERR_INVALID, "unknown precision");
    if ((int64_t)H * W > (1 << 24)) return fail(IDM_ERR_UNSUPPORTED, "lattice too large");
    return IDM_OK;
}

}  // namespace

extern "C" {

int idm_abi_version(void) { return IDM_ABI_VERSION; }

const char *idm_last_error(void) { return g_err.c_str(); }

int idm_model_create(idm_model **out, int device, const idm_model_desc *d)
{
    if (!out || !d) return fail(IDM_ERR_INVALID, "null argument");
    *out = nullptr;
    if (d->base_filters != 32) return fail(IDM_ERR_UNSUPPORTED, "base_filters must be 32");
    if (d->channels != 1 && d->channels != 2) return fail(IDM_ERR_UNSUPPORTED, "channels must be 1 or 2");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(IDM_ERR_INVALID, "no such device");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(IDM_ERR_UNSUPPORTED, std::string("built for gfx950 (MI355X); device is ") + prop.gcnArchName);

    if (device >= IDM_MAX_DEVICES) return fail(IDM_ERR_UNSUPPORTED, "device index beyond IDM_MAX_DEVICES");
#ifdef IDM_EXPERIMENT
    // A library built with a timing-experiment switch (IDM_EXP_*: proxies that compute WRONG results by design) must not
    // pass for the product: handles are refused unless the measurement script says it knows what it loaded.
    {
        const char *allow = getenv("ISINGDIFF_ALLOW_EXPERIMENT");
        if (!(allow && allow[0] == '1'))
            return fail(IDM_ERR_UNSUPPORTED, "libisingdiff.so was built with -DIDM_EXPERIMENT (timing proxies, wrong results); "
                                             "ISINGDIFF_ALLOW_EXPERIMENT=1 lets a measurement script use it");
    }
#endif
    {
        char *unused = nullptr;
        int rc0 = scratch_address(device, &unused);
        if (rc0 != IDM_OK) return rc0;
    }

    idm_model *m = new idm_model();
    m->device = device; m->C = d->channels; m->F = d->base_filters; m->bn = d->batchnorm != 0;
    const int f = m->F, c = m->C;
    const int cin[IDM_NCONV] = {c, f, f, 2 * f, 2 * f, 4 * f, 6 * f, 2 * f, 3 * f, f, f};
    const int cout[IDM_NCONV] = {f, f, 2 * f, 2 * f, 4 * f, 4 * f, 2 * f, 2 * f, f, f, c};
    int rc = IDM_OK;
    for (int i = 0; i < IDM_NCONV && rc == IDM_OK; ++i) {
        m->cin[i] = cin[i]; m->cout[i] = cout[i];
        if (!d->conv_w[i] || !d->conv_b[i]) { rc = fail(IDM_ERR_INVALID, "missing conv weights"); break; }
        rc = upload(m, d->conv_b[i], (size_t)cout[i], &m->bias[i]);
        if (rc == IDM_OK && m->bn && i < 10) {
            if (!d->bn_scale[i] || !d->bn_shift[i]) { rc = fail(IDM_ERR_INVALID, "missing BatchNorm tensors"); break; }
            rc = upload(m, d->bn_scale[i], (size_t)cout[i], &m->scale[i]);
            if (rc == IDM_OK) rc = upload(m, d->bn_shift[i], (size_t)cout[i], &m->shift[i]);
        }
        if (rc != IDM_OK) break;
        if (i == 0) {
            rc = upload(m, d->conv_w[0], (size_t)9 * c * f, &m->w_first);
            if (rc == IDM_OK && c == 2) {
                std::vector<float> wt((size_t)9 * c * f);
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx)
                        std::copy(d->conv_w[0] + (size_t)(kx * 3 + ky) * c * f, d->conv_w[0] + (size_t)(kx * 3 + ky + 1) * c * f,
                                  wt.begin() + (size_t)(ky * 3 + kx) * c * f);
                rc = upload(m, wt.data(), wt.size(), &m->w_first_tr);
                if (rc == IDM_OK && f == 32) {
                    const float *sc = m->bn ? d->bn_scale[0] : nullptr, *sh = m->bn ? d->bn_shift[0] : nullptr;
                    std::vector<uint16_t> pk = pack_first2_bf16(d->conv_w[0], sc), pkt = pack_first2_bf16(wt.data(), sc);
                    std::vector<float> bf_(32);
                    for (int k = 0; k < 32; ++k) bf_[k] = d->conv_b[0][k] * (sc ? sc[k] : 1.0f) + (sh ? sh[k] : 0.0f);
                    rc = upload(m, pk.data(), pk.size(), (uint16_t **)&m->w_first2_pack);
                    if (rc == IDM_OK) rc = upload(m, pkt.data(), pkt.size(), (uint16_t **)&m->w_first2_pack_tr);
                    if (rc == IDM_OK) rc = upload(m, bf_.data(), bf_.size(), &m->bias_first2);
                }
            }
            if (rc == IDM_OK && c == 1) {
                std::vector<uint16_t> pk = pack_first_bf16(d->conv_w[0], f);
                rc = upload(m, pk.data(), pk.size(), (uint16_t **)&m->w_first_pack);
            }
        }
        else if (i == 10) rc = upload(m, d->conv_w[10], (size_t)f * c, &m->w_head);
        else {
            std::vector<uint16_t> pb = pack_weights_bf16(d->conv_w[i], cin[i], cout[i]);
            std::vector<float> pf = pack_weights_f32(d->conv_w[i], cin[i], cout[i]);
            rc = upload(m, pb.data(), pb.size(), (uint16_t **)&m->wpack_bf16[i]);
            if (rc == IDM_OK) rc = upload(m, pf.data(), pf.size(), (float **)&m->wpack_f32[i]);
            if (rc == IDM_OK && i >= 2 && i <= 8 && !m->bn) {   // the layers on the 16 x 16 and 8 x 8 maps at L = 32, and conv8 (ninth row of a top band)
                std::vector<uint16_t> p16 = pack_weights_bf16_m16(d->conv_w[i], cin[i], cout[i]);
                rc = upload(m, p16.data(), p16.size(), (uint16_t **)&m->wpack_m16[i]);
            }
            std::vector<float> wf(d->conv_w[i], d->conv_w[i] + (size_t)9 * cin[i] * cout[i]);   // what the strip kernels multiply by
            if (rc == IDM_OK && m->bn) {
                // inference BatchNorm folded into the convolution: (W x + b)*s + h = (W*s) x + (b*s + h)
                std::vector<float> bf_(cout[i]);
                for (size_t k = 0; k < wf.size(); ++k) wf[k] = d->conv_w[i][k] * d->bn_scale[i][k % cout[i]];
                for (int k = 0; k < cout[i]; ++k) bf_[k] = d->conv_b[i][k] * d->bn_scale[i][k] + d->bn_shift[i][k];
                std::vector<uint16_t> pfold = pack_weights_bf16(wf.data(), cin[i], cout[i]);
                rc = upload(m, pfold.data(), pfold.size(), (uint16_t **)&m->wpack_fold[i]);
                if (rc == IDM_OK) rc = upload(m, bf_.data(), bf_.size(), &m->bias_fold[i]);
            }
            if (rc == IDM_OK && c == 2) {
                // the same kernel with its taps transposed, (ky, kx) -> (kx, ky): the convolution of the transposed map
                const size_t per_tap = (size_t)cin[i] * cout[i];
                std::vector<float> wt(wf.size());
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx)
                        std::copy(wf.begin() + (kx * 3 + ky) * per_tap, wf.begin() + (kx * 3 + ky + 1) * per_tap,
                                  wt.begin() + (ky * 3 + kx) * per_tap);
                std::vector<uint16_t> ptr_ = pack_weights_bf16(wt.data(), cin[i], cout[i]);
                rc = upload(m, ptr_.data(), ptr_.size(), (uint16_t **)&m->wpack_tr[i]);
                if (rc == IDM_OK && (i == 4 || i == 5)) {
                    std::vector<uint16_t> p16 = pack_weights_bf16_m16(wt.data(), cin[i], cout[i]);
                    rc = upload(m, p16.data(), p16.size(), (uint16_t **)&m->wpack_tr_m16[i]);
                }
            }
        }
    }
    const int din[IDM_NDENSE] = {1, 4 * f, 4 * f};
    for (int i = 0; i < IDM_NDENSE && rc == IDM_OK; ++i) {
        if (!d->dense_w[i] || !d->dense_b[i]) { rc = fail(IDM_ERR_INVALID, "missing dense weights"); break; }
        rc = upload(m, d->dense_w[i], (size_t)din[i] * 4 * f, &m->dense_w[i]);
        if (rc == IDM_OK) rc = upload(m, d->dense_b[i], (size_t)4 * f, &m->dense_b[i]);
    }
    if (rc == IDM_OK) {
        hipError_t e = hipMalloc((void **)&m->trow, (size_t)4 * f * sizeof(float));
        if (e != hipSuccess) rc = fail(IDM_ERR_NOMEM, "hipMalloc failed");
        else m->owned.push_back(m->trow);
    }
    if (rc != IDM_OK) { std::string keep = g_err; idm_model_destroy(m); g_err = keep; return rc; }
    *out = m;
    return IDM_OK;
}

int idm_model_destroy(idm_model *m)
{
    if (!m) return IDM_OK;
    (void)hipSetDevice(m->device);
    for (void *p : m->owned) (void)hipFree(p);
    if (m->ttable) (void)hipFree(m->ttable);
    if (m->ws) (void)hipFree(m->ws);
    for (auto &sp : m->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (hipEvent_t e : m->event_pool) (void)hipEventDestroy(e);
    delete m;
    return IDM_OK;
}

int idm_time_table(idm_model *m, int T, float *table_host)
{
    if (!m || T <= 0) return fail(IDM_ERR_INVALID, "bad argument");
    HIPCHK(hipSetDevice(m->device));
    const int n = 4 * m->F;
    if (m->ttable_T != T) {
        if (m->ttable) { HIPCHK(hipFree(m->ttable)); m->ttable = nullptr; m->ttable_T = 0; }
        HIPCHK(hipMalloc((void **)&m->ttable, (size_t)T * n * sizeof(float)));
        launch_time_table(T, n, m->dense_w[0], m->dense_b[0], m->dense_w[1], m->dense_b[1], m->dense_w[2],
                          m->dense_b[2], m->ttable, nullptr);
        HIPCHK(hipGetLastError());
        HIPCHK(hipDeviceSynchronize());
        m->ttable_T = T;
    }
    if (table_host) HIPCHK(hipMemcpy(table_host, m->ttable, (size_t)T * n * sizeof(float), hipMemcpyDeviceToHost));
    return IDM_OK;
}

int idm_forward_taps(idm_model *m, const float *x_dev, float t_norm, float *score_dev, int64_t B, int H, int W,
                     int precision, float *const *taps_dev, void *stream)
{
    int rc = check_shape(m, B, H, W, precision);
    if (rc != IDM_OK) return rc;
    if (!x_dev || !score_dev) return fail(IDM_ERR_INVALID, "null tensor");
    HIPCHK(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    const Plan p = make_plan(m, H, W, precision, taps_dev == nullptr);
    const int64_t chunk = auto_chunk(p, B);
    rc = ensure_ws(m, p.per_lattice * (size_t)chunk);
    if (rc != IDM_OK) return rc;
    launch_time_row(t_norm, 4 * m->F, m->dense_w[0], m->dense_b[0], m->dense_w[1], m->dense_b[1], m->dense_w[2],
                    m->dense_b[2], m->trow, s);
    const size_t elems = (size_t)H * W * m->C;
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int64_t bc = std::min(chunk, B - b0);
        if (p.fused)
            rc = forward_fused(m, p, x_dev + b0 * elems, m->trow, score_dev + b0 * elems, bc, StepCtx(), taps_dev, b0, s);
        else
            rc = forward_chunk(m, p, x_dev + b0 * elems, m->trow, score_dev + b0 * elems, bc, precision, taps_dev, b0, s);
        if (rc != IDM_OK) return rc;
    }
    return IDM_OK;
}

int idm_plan_chunk(idm_model *m, int64_t B, int H, int W, int precision, int64_t *chunk_out, int64_t *bytes_per_lattice_out)
{
    int rc = check_shape(m, B, H, W, precision);
    if (rc != IDM_OK) return rc;
    const Plan p = make_plan(m, H, W, precision);
    if (chunk_out) *chunk_out = auto_chunk(p, B);
    if (bytes_per_lattice_out) *bytes_per_lattice_out = (int64_t)p.per_lattice;
    return IDM_OK;
}

int idm_forward(idm_model *m, const float *x_dev, float t_norm, float *score_dev, int64_t B, int H, int W,
                int precision, void *stream)
{
    return idm_forward_taps(m, x_dev, t_norm, score_dev, B, H, W, precision, nullptr, stream);
}

int idm_forward_noalloc(idm_model *m, const float *x_dev, float t_norm, float *score_dev, int64_t B, int H, int W,
                        int precision, void *stream)
{
    int rc = check_shape(m, B, H, W, precision);
    if (rc != IDM_OK) return rc;
    if (!x_dev || !score_dev) return fail(IDM_ERR_INVALID, "null tensor");
    HIPCHK(hipSetDevice(m->device));
    char *ws = nullptr;
    rc = scratch_address(m->device, &ws);   // resolved by idm_model_create; a lookup under the mutex here
    if (rc != IDM_OK) return rc;
    const Plan p = make_plan(m, H, W, precision);
    if (p.per_lattice * (size_t)B > (size_t)IDM_NOALLOC_SCRATCH_BYTES)
        return fail(IDM_ERR_INVALID, "idm_forward_noalloc: " + std::to_string(B) + " lattices of " + std::to_string(H) + "x" +
                                         std::to_string(W) + " need " + std::to_string(p.per_lattice * (size_t)B) +
                                         " bytes of workspace; the static scratch holds " +
                                         std::to_string((size_t)IDM_NOALLOC_SCRATCH_BYTES) + " (at most " +
                                         std::to_string((size_t)IDM_NOALLOC_SCRATCH_BYTES / p.per_lattice) +
                                         " lattices of this shape and precision per call)");
    hipStream_t s = (hipStream_t)stream;
    // No event bracketing here (events are created on demand), and every buffer of the workspace is written
    // before it is read, so nothing has to be cleared between calls.
    const bool was_profiling = m->profiling;
    m->profiling = false;
    m->ws_active = ws;
    launch_time_row(t_norm, 4 * m->F, m->dense_w[0], m->dense_b[0], m->dense_w[1], m->dense_b[1], m->dense_w[2],
                    m->dense_b[2], m->trow, s);
    if (p.fused) rc = forward_fused(m, p, x_dev, m->trow, score_dev, B, StepCtx(), nullptr, 0, s);
    else rc = forward_chunk(m, p, x_dev, m->trow, score_dev, B, precision, nullptr, 0, s);
    m->ws_active = m->ws;
    m->profiling = was_profiling;
    return rc;
}

int idm_sample(idm_model *m, const idm_sample_cfg *cfg, float *x_dev, const float *noise_dev, float *traj_dev,
               float *scores_dev, int8_t *spins_dev, int64_t B, int H, int W, void *stream)
{
    if (!cfg) return fail(IDM_ERR_INVALID, "null cfg");
    int rc = check_shape(m, B, H, W, cfg->precision);
    if (rc != IDM_OK) return rc;
    if (!x_dev) return fail(IDM_ERR_INVALID, "x_dev is null");
    if (cfg->T <= 0 || !cfg->betas_host) return fail(IDM_ERR_INVALID, "T and betas_host are required");
    if (cfg->chunk < 0) return fail(IDM_ERR_INVALID, "negative chunk");
    if (cfg->update_rule != IDM_UPDATE_SCORE && cfg->update_rule != IDM_UPDATE_DDPM) return fail(IDM_ERR_INVALID, "unknown update_rule");
    HIPCHK(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    const int T = cfg->T;
    std::vector<float> acp;   // alpha_cumprod[t] for the DDPM rule (:170-173)
    if (cfg->update_rule == IDM_UPDATE_DDPM) {
        acp.resize(T);
        double prod = 1.0;
        for (int t = 0; t < T; ++t) {
            prod *= 1.0 - (double)cfg->betas_host[t];
            acp[t] = cfg->alpha_cumprod_host ? cfg->alpha_cumprod_host[t] : (float)prod;
            if (!(acp[t] < 1.0f) || !(cfg->betas_host[t] < 1.0f)) return fail(IDM_ERR_INVALID, "the DDPM rule needs beta_t < 1 and alpha_cumprod[t] < 1");
        }
    }
    rc = idm_time_table(m, T, nullptr);
    if (rc != IDM_OK) return rc;
    const Plan p = make_plan(m, H, W, cfg->precision);
    int64_t chunk = cfg->chunk > 0 ? std::min<int64_t>(cfg->chunk, B) : auto_chunk(p, B);
    rc = ensure_ws(m, p.per_lattice * (size_t)chunk);
    if (rc != IDM_OK) return rc;
    const size_t elems = (size_t)H * W * m->C;
    const size_t total = (size_t)B * elems;
    const int nf = 4 * m->F;

    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int64_t bc = std::min(chunk, B - b0);
        float *x = x_dev + b0 * elems;
        float *score_ws = (float *)(m->ws_active + p.off_score * (size_t)bc);
        PhiloxArgs ph;
        ph.seed = cfg->seed;
        ph.first_global_index = cfg->first_global_index + b0;
        if (cfg->init_state) {
            ph.t = T;
            launch_philox_normal(x, bc, (int64_t)elems, ph, s);
        }
        if (traj_dev)
            HIPCHK(hipMemcpyAsync(traj_dev + (size_t)T * total + b0 * elems, x, bc * elems * 4, hipMemcpyDeviceToDevice, s));
        for (int t = T - 1; t >= 0; --t) {
            float *score = scores_dev ? scores_dev + (size_t)t * total + b0 * elems : score_ws;
            ph.t = t;
            const StepConst kt = step_constants(cfg->update_rule, cfg->betas_host[t], acp.empty() ? 0.0f : acp[t]);
            const float *z = noise_dev ? noise_dev + (size_t)t * total + b0 * elems : nullptr;
            if (p.fused) {
                StepCtx ctx;
                ctx.do_update = true; ctx.x = x; ctx.noise = z; ctx.k = kt;
                ctx.add_noise = cfg->add_noise != 0; ctx.ph = ph;
                rc = forward_fused(m, p, x, m->ttable + (size_t)t * nf, scores_dev ? score : nullptr, bc, ctx, nullptr, 0, s);
                if (rc != IDM_OK) return rc;
            } else {
                rc = forward_chunk(m, p, x, m->ttable + (size_t)t * nf, score, bc, cfg->precision, nullptr, 0, s);
                if (rc != IDM_OK) return rc;
                timed(m, "update", 0.0, (double)bc * elems * (z ? 16 : 12), s, [&] {
                    launch_update(x, score, z, kt.c1, kt.c2, kt.sd, bc, (int64_t)elems, cfg->add_noise != 0, ph, s);
                });
            }
            if (traj_dev)
                HIPCHK(hipMemcpyAsync(traj_dev + (size_t)t * total + b0 * elems, x, bc * elems * 4, hipMemcpyDeviceToDevice, s));
        }
        if (spins_dev) launch_sign_project(x, spins_dev + b0 * elems, (int64_t)(bc * elems), s);
        HIPCHK(hipGetLastError());
    }
    return IDM_OK;
}

int idm_update(int device, float *x_dev, const float *score_dev, const float *z_dev, float beta, int64_t n, void *stream)
{
    if (!x_dev || !score_dev || n <= 0) return fail(IDM_ERR_INVALID, "bad argument");
    HIPCHK(hipSetDevice(device));
    PhiloxArgs ph = {0, 0, 0};
    const StepConst k = step_constants(IDM_UPDATE_SCORE, beta, 0.0f);
    launch_update(x_dev, score_dev, z_dev, k.c1, k.c2, k.sd, 1, n, z_dev != nullptr, ph, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return IDM_OK;
}

int idm_philox_normal(int device, float *out_dev, int64_t n_lattices, int64_t elems_per_lattice, uint64_t seed,
                      int64_t first_global_index, int32_t t, void *stream)
{
    if (!out_dev || n_lattices <= 0 || elems_per_lattice <= 0) return fail(IDM_ERR_INVALID, "bad argument");
    HIPCHK(hipSetDevice(device));
    PhiloxArgs ph = {seed, first_global_index, t};
    launch_philox_normal(out_dev, n_lattices, elems_per_lattice, ph, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return IDM_OK;
}

int idm_sign_project(int device, const float *x_dev, int8_t *spins_dev, int64_t n, void *stream)
{
    if (!x_dev || !spins_dev || n <= 0) return fail(IDM_ERR_INVALID, "bad argument");
    HIPCHK(hipSetDevice(device));
    launch_sign_project(x_dev, spins_dev, n, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return IDM_OK;
}

int idm_observe(int device, const int8_t *spins_dev, int64_t B, int L, int32_t *M_dev, int32_t *S_dev,
                unsigned long long *hist_M_dev, unsigned long long *hist_S_dev, long long *moments_dev, void *stream)
{
    if (!spins_dev || B <= 0 || L <= 0) return fail(IDM_ERR_INVALID, "bad argument");
    if (L > 4096) return fail(IDM_ERR_UNSUPPORTED, "L too large");
    HIPCHK(hipSetDevice(device));
    launch_observe(spins_dev, B, L, M_dev, S_dev, hist_M_dev, hist_S_dev, moments_dev, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return IDM_OK;
}

int idm_irfft2_sign(int device, const float *spec_dev, int64_t B, int L, int8_t *spins_dev, double *real_dev, void *stream)
{
    if (!spec_dev || !spins_dev || B <= 0) return fail(IDM_ERR_INVALID, "bad argument");
    if (L < 2 || (L & 1)) return fail(IDM_ERR_UNSUPPORTED, "irfft2 needs an even lattice size");
    HIPCHK(hipSetDevice(device));
    if (idm_fft_supported(L)) {
        HIPCHK(launch_irfft2_fft(spec_dev, B, L, spins_dev, real_dev, (hipStream_t)stream));
        return IDM_OK;
    }
    // other even sizes: direct sums over a twiddle table (O(L^3) per lattice)
    if ((size_t)(2 * L + 4 * L * (L / 2 + 1)) * sizeof(double) > 160 * 1024)
        return fail(IDM_ERR_UNSUPPORTED, "no FFT instance for this lattice size, and it is too large for the direct-sum transform (L <= 98)");
    HIPCHK(launch_irfft2_sign(spec_dev, B, L, spins_dev, real_dev, (hipStream_t)stream));
    return IDM_OK;
}

int idm_metropolis(int device, int8_t *spins_dev, int64_t B, int L, const float *accept_host, int n_sweeps, uint64_t seed,
                   int64_t first_global_index, int sweep0, void *stream)
{
    if (!spins_dev || !accept_host || B <= 0 || n_sweeps < 0) return fail(IDM_ERR_INVALID, "bad argument");
    if (L < 2 || (L & 1) || L > 224) return fail(IDM_ERR_UNSUPPORTED, "checkerboard Metropolis needs an even L in [2, 224]");
    HIPCHK(hipSetDevice(device));
    PhiloxArgs ph = {seed, first_global_index, 0};
    launch_metropolis(spins_dev, B, L, accept_host, n_sweeps, ph, sweep0, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return IDM_OK;
}

int idm_set_option(idm_model *m, const char *name, int value)
{
    if (!m || !name) return fail(IDM_ERR_INVALID, "null argument");
    if (!strcmp(name, "fused")) { m->use_fused = value != 0; return IDM_OK; }
    if (!strcmp(name, "strip_below_rows")) { m->strip_below_rows = value; return IDM_OK; }
    if (!strcmp(name, "transpose")) { m->use_transpose = value != 0; return IDM_OK; }
    return fail(IDM_ERR_INVALID, std::string("unknown option ") + name);
}

int idm_profile_begin(idm_model *m)
{
    if (!m) return fail(IDM_ERR_INVALID, "null model");
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipDeviceSynchronize());
    for (auto &sp : m->spans) { m->event_pool.push_back(sp.a); m->event_pool.push_back(sp.b); }
    m->spans.clear();
    m->classes.clear();
    m->profiling = true;
    return IDM_OK;
}

int idm_profile_end(idm_model *m, idm_profile_entry *entries, int cap, int *n)
{
    if (!m || !n) return fail(IDM_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipDeviceSynchronize());
    m->profiling = false;
    for (auto &sp : m->spans) {
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, sp.a, sp.b));
        m->classes[sp.cls].ms += ms;
        m->event_pool.push_back(sp.a);
        m->event_pool.push_back(sp.b);
    }
    m->spans.clear();
    *n = (int)m->classes.size();
    for (int i = 0; i < *n && i < cap && entries; ++i) entries[i] = m->classes[i];
    return IDM_OK;
}

}  // extern "C"
