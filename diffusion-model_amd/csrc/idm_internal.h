// Internal interface between the C-ABI translation unit and the kernel translation units.
// Nothing here is exported from libisingdiff.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

// Records the message idm_last_error() returns on this thread; returns `code`.
int idm_set_error(int code, const char *msg);
// Keras kernel (3,3,Cin,Cout) -> float A operands of v_mfma_f32_32x32x2_f32, [tap][c/8][mt][lane][4].
std::vector<float> idm_pack_weights_f32(const float *w, int cin, int cout);
std::vector<uint16_t> idm_pack_weights_bf16(const float *w, int cin, int cout);   // (3,3,Cin,Cout) -> bf16 MFMA A fragments
// the same as A fragments of v_mfma_f32_16x16x32_bf16: [tap][c/32][cout/16][lane][8], lane l = cout l & 15, channels 8 (l >> 4) .. +7
std::vector<uint16_t> idm_pack_weights_bf16_m16(const float *w, int cin, int cout);
// the same for a (Cin,Cout) 1x1 kernel: one tap
std::vector<float> idm_pack_weights_f32_1x1(const float *w, int cin, int cout);

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

enum ConvMode { CONV_PLAIN = 0, CONV_POOL = 1, CONV_UPCAT = 2 };

// One 3x3 "same" convolution layer as an implicit GEMM  D[cout][pixel] = sum_k W[cout][k] X[k][pixel],
// k = (ky*3 + kx)*Cin + c.  NHWC tensors.  The loader modes fold the UNet's data movement into
// the operand fetch:
//   CONV_PLAIN  in0 is (B,H,W,C0)
//   CONV_POOL   in0 is (B,2H,2W,C0); the operand is its 2x2 average
//   CONV_UPCAT  in0 is (B,H/2,W/2,C0) read with nearest upsampling, channels [C0, C0+C1) come
//               from in1 (B,H,W,C1): Concatenate([UpSampling2D(x), skip])
struct ConvArgs {
    const void *in0;
    const void *in1;
    int C0, C1;
    int H, W;            // output (= conv input) spatial size
    int64_t B;
    const void *wpack;   // fragment-ordered weights (see pack_weights_* in idm_api.cpp)
    const float *bias;
    const float *scale;  // folded BatchNorm or nullptr
    const float *shift;
    const float *tadd;   // per-channel add applied after the activation, or nullptr
    void *out;           // (B,H,W,Cout)
    int Cout;
    int mode;
    int silu;
    int out_f32;         // bf16 pipeline only: store float32 instead of bf16
    // float32 kernel only: taps [tap_lo, tap_hi) of the 3x3 stencil are evaluated and wpack holds exactly those
    // (tap_lo = 4, tap_hi = 5 with a one-tap pack is a 1x1 convolution); resid (same shape as out) is added last
    int tap_lo = 0, tap_hi = 9;
    const float *resid = nullptr;
    // LDS-tiled kernel with CONV_UPCAT only (io_f16 and gn_ab go together): in0 / in1, out and resid hold IEEE half; the input
    // is the raw tensor and y = swish(x a + b), (a, b) pairs per (sample, channel) as (B, C0 + C1, 2) floats, is applied
    // while the tile is staged (GroupNorm folded into the convolution); the raw convolution leaves rounded once to half,
    // saturated at +-65504.  C0 = 0: a plain tensor in in1.
    int io_f16 = 0;
    const float *gn_ab = nullptr;
};

struct PhiloxArgs {
    uint64_t seed;
    int64_t first_global_index;
    int32_t t;
};

// ---- float32 canonical-order kernels (idm_canon.hip) ----
void launch_conv_f32(const ConvArgs &a, hipStream_t s);
// bf16 output, one thread per pixel and 8 channels (idm_fast.hip)
void launch_conv_first_bf16(const float *x, int64_t B, int H, int W, int Hp, int Wp, int C, const float *w, const float *bias,
                            const float *scale, const float *shift, void *out, hipStream_t s, bool transposed = false);
// first layer of the two-channel model on the matrix cores, 32 couts (idm_fast.hip); wpack: idm_api.hip's pack_first2_bf16, BatchNorm
// folded; false: shape beyond 32-bit indexing, nothing launched
bool launch_conv_first2_mfma(const float *x, int64_t B, int H, int W, int Hp, int Wp, const void *wpack, const float *biasf, void *out,
                             hipStream_t s, bool transposed);
void launch_conv_first(const float *x, int64_t B, int H, int W, int Hp, int Wp, int C, const float *w,
                       const float *bias, const float *scale, const float *shift, void *out, int F,
                       bool out_bf16, hipStream_t s);
void launch_head(const float *a9, int64_t B, int H, int W, int Hp, int Wp, int F, int C, const float *w,
                 const float *bias, float *score, hipStream_t s);
void launch_head_bf16(const void *a9, int64_t B, int H, int W, int Hp, int Wp, int F, int C, const float *w,
                      const float *bias, float *score, hipStream_t s);
// x <- ((c1*x) + (c2*score)) + (z*sd)
void launch_update(float *x, const float *score, const float *z, float c1, float c2, float sd, int64_t n_lattices,
                   int64_t elems, bool add_noise, const PhiloxArgs &ph, hipStream_t s);
void launch_philox_normal(float *out, int64_t n_lattices, int64_t elems, const PhiloxArgs &ph, hipStream_t s);
void launch_time_table(int T, int n, const float *w0, const float *b0, const float *w1, const float *b1,
                       const float *w2, const float *b2, float *table, hipStream_t s);
void launch_time_row(float t_norm, int n, const float *w0, const float *b0, const float *w1, const float *b1,
                     const float *w2, const float *b2, float *row, hipStream_t s);
void launch_sign_project(const float *x, int8_t *spins, int64_t n, hipStream_t s);
void launch_observe(const int8_t *spins, int64_t B, int L, int32_t *M, int32_t *S, unsigned long long *histM,
                    unsigned long long *histS, long long *moments, hipStream_t s);
hipError_t launch_irfft2_sign(const float *spec, int64_t B, int L, int8_t *spins, double *real_out, hipStream_t s);
// in-register FFT instances of the same transform (idm_fft.hip) for the lattice sizes idm_fft_supported() names
bool idm_fft_supported(int L);
hipError_t launch_irfft2_fft(const float *spec, int64_t B, int L, int8_t *spins, double *real_out, hipStream_t s);
void launch_metropolis(int8_t *spins, int64_t B, int L, const float *table10, int n_sweeps, const PhiloxArgs &ph, int sweep0,
                       hipStream_t s);
void launch_cast_bf16_to_f32(const void *in, float *out, int64_t n, hipStream_t s);

// ---- bf16 kernels (idm_fast.hip) ----
void launch_conv_bf16(const ConvArgs &a, hipStream_t s);

// LDS-tiled kernel for plain convolutions on maps at most 32 wide with raw float32 output (idm_tileconv.hip); false: shape
// not covered, nothing launched
bool launch_conv_tile_bf16(const ConvArgs &a, hipStream_t s);

// ---- fused two-convolution band kernels (idm_fused.hip) ----
// FK_BH2: FK_B's two convolutions followed by the two-channel head of the spectral model (score only, no update)
enum FusedKind { FK_A = 0, FK_M1 = 1, FK_M2 = 2, FK_M3 = 3, FK_B = 4, FK_BH2 = 5, FK_COUNT = 6 };

struct FusedArgs {
    int H, W;          // map size of this level (conv outputs)
    int R;             // band rows (even)
    int nbands;
    int Hx, Wx;        // FIRST / HEAD: size of the unpadded state x
    const void *in_a;  // PLAIN: (B,H,W,CA); UPCAT: (B,H/2,W/2,CA); FIRST: x float32 (B,Hx,Wx)
    const void *in_b;  // UPCAT: (B,H,W,CB)
    const void *w1;
    const void *w1_alt;        // the first convolution's weights packed for v_mfma_f32_16x16x32_bf16 ([tap][ch/32][cout/16][lane][8]), or null
    const float *b1, *tadd1;   // bias; per-channel add after the activation (FK_M2 only)
    const void *w2;
    const void *w2_alt;        // the same for the pair's second convolution
    const float *b2;
    void *out;         // STORE*: (B,H,W,C2) bf16
    void *out_pool;    // STORE_POOL: (B,H/2,W/2,C2) bf16
    // HEAD
    const float *w_head;
    const float *b_head;
    float *x;                // state, updated in place
    float *score_out;        // optional (B,Hx,Wx); FK_BH2: (B,Hx,Wx,2), or (B,Wx,Hx,2) when transpose_out
    int transpose_out;       // FK_BH2: the map is processed transposed, the score leaves in the caller's layout
    const float *noise;      // optional injected standard normals (B,Hx,Wx)
    float c1, beta, sd;
    int add_noise, do_update;
    PhiloxArgs ph;
    int lds_in_bytes, lds_mid_bytes;  // region sizes (host-computed, multiples of 16)
    int64_t B;                        // lattices in this launch (bands = B * nbands, walked persistently)
};


void fused_lds_bytes(int kind, int W, int R, int *in_bytes, int *mid_bytes);
bool fused_band_fits(int kind, int W, int R);
int fused_tall_rows(int kind, int H, int W);
int fused_m1_whole_rows(int Hx, int Wx);   // band rows of conv2 + conv3 when its whole-map instance applies, else 0
hipError_t launch_fused(int kind, const FusedArgs &a, int64_t B, hipStream_t s);
// fused_B for the L = 32 lattice, MFMA and non-MFMA work on separate wavefronts (idm_fused_split.hip)
hipError_t launch_b_split(const FusedArgs &a, int64_t B, hipStream_t s);
// fused_M3 for the L = 32 pyramid, tile copies on wavefronts of their own (idm_fused_split.hip)
hipError_t launch_m3_split(const FusedArgs &a, int64_t B, hipStream_t s);
// fused_M1 on the whole 16 x 16 map of the L = 32 pyramid on v_mfma_f32_16x16x32_bf16 (idm_fused_split.hip); needs w1_alt / w2_alt
// wide: its conv4 (+ time add) + conv5 instance for 16 x 16 maps of 64 -> 128 -> 128 channels (the third level at L = 64), no pooled copy
hipError_t launch_m1_mf16(const FusedArgs &a, int64_t B, hipStream_t s, bool wide = false);
// fused_M2 on the 8 x 8 maps of the L = 32 pyramid, two lattices per pass, on v_mfma_f32_16x16x32_bf16 (idm_fused_split.hip);
// needs w1_alt / w2_alt; B < 2^19
hipError_t launch_m2_pair(const FusedArgs &a, int64_t B, hipStream_t s);

// ---- single-convolution band/strip kernels (idm_strip.hip) ----
enum StripKindId { SK_CONV1 = 0, SK_CONV2, SK_CONV3, SK_CONV4, SK_CONV5, SK_CONV6, SK_CONV7, SK_CONV8, SK_CONV9, SK_CONV9S, SK_CONV9H2, SK_COUNT };   // SK_CONV9: conv9 + head + update; SK_CONV9S: conv9 stored; SK_CONV9H2: conv9 + two-channel head

struct StripArgs {
    int H, W;          // map size of this layer's output (= its full-resolution input)
    int R, WS;         // band rows (even) and strip columns (W % WS == 0, WS % 4 == 0)
    int nbands, nstrips;
    int Hx, Wx;        // HEAD: size of the unpadded state x
    const void *in_a;  // plain: (B,H,W,CA); UP: (B,H/2,W/2,CA)
    const void *in_b;  // UP: (B,H,W,CB)
    const void *w;
    const float *bias, *tadd;
    void *out;         // (B,H,W,COUT) bf16
    void *out_pool;    // POOL: (B,H/2,W/2,COUT) bf16
    const float *w_head, *b_head;
    float *x;
    float *score_out;  // HEAD: (B,Hx,Wx); two-channel head: (B,Hx,Wx,2), or (B,Wx,Hx,2) when transpose_out
    int transpose_out; // two-channel head: the map is processed transposed, the score leaves in the caller's layout
    const float *noise;
    float c1, beta, sd;
    int add_noise, do_update;
    PhiloxArgs ph;
    int lds_in_bytes, lds_stage_bytes, lds_const_bytes;
    int64_t B;
};

void strip_lds_bytes(int kind, int R, int WS, int *in_bytes, int *stage_bytes, int *const_bytes);
bool strip_fits(int kind, int R, int WS);
hipError_t launch_strip(int kind, const StripArgs &a, hipStream_t s);

// Per-kernel launch state (function attributes, occupancy) is set per device: the caches are indexed by the
// current device of the calling thread.  Returns -1 (and the launchers fail) beyond IDM_MAX_DEVICES.
constexpr int IDM_MAX_DEVICES = 64;
inline int idm_device_slot()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= IDM_MAX_DEVICES) return -1;
    return d;
}
