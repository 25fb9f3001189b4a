// Shared pieces of the implicit-GEMM convolution kernels (conv.hip: generic geometry; conv_fast.hip: the stride-1 hot path).
#pragma once
#include "gm_common.h"

// include/gm_amd.h: GmGnTables (the GroupNorm of a token GEMM's input as the statistic tables of its producers, gm_linear_rows_gn)
struct GmGnTables {
  const double* stats[2]; int S[2]; int C[2];
  const float* gamma; const float* beta; float eps; int groups;
};

struct GmConvDesc {
  const void* x; long long x_ld;
  const void* w;                  // packed by gm_pack_conv_weight: [chunk][tap][Cout_pad][BK]
  const float* bias;              // [Cout] or null
  const float* pre_scale;         // [N][Cin] or null
  const float* pre_shift;         // [N][Cin] or null
  const float* rowvec;            // [B][Cout] fp32 or null, added per (n, cout)
  long long rowvec_bstride;       // 0 -> broadcast one row over the batch
  const void* res; long long res_ld;  // residual in output geometry or null
  void* y; long long y_ld;
  int N, Cin, Cout;
  int Ds, Hs, Ws;                 // stored input dims
  int Do, Ho, Wo;
  int kd, kh, kw;
  int sd, sh, sw;
  int pd, ph, pw;                 // low-side padding (high side is implied by the output size)
  int dd, dh, dw;                 // dilation
  int in_mode;                    // 0 direct, 1 nearest up-sample by (fd,fh,fw), 2 zero-insertion by (fd,fh,fw)
  int fd, fh, fw;
  int pre_act;                    // 0 none, 1 SiLU, 2 ReLU (applied after the optional affine)
  int post_act;                   // 0 none, 1 ReLU, 2 tanh, 3 sigmoid, 4 SiLU, 5 LeakyReLU(0.01), 6 GELU (erf)
  int dtype;
  int ltd, lth, ltw;              // log2 of the output tile dims
  int cfg;                        // tile configuration id (see dispatch)
  int debug_flags;                // 0 in production; bench-only ablation switches of conv_fast.hip
  double* stats;                  // optional [S][N][Cout][2] (sum, sum of squares) of the OUTPUT, S = gm_conv_stats_slots(desc) = tiles per
                                  // sample: every work-group STORES its partial exactly once (no atomics, no zero fill); consumers add the S
                                  // partials in a fixed order, so the statistics -- and everything downstream -- are bit-reproducible run to run
  // optional fused 1x1 "skip" convolution (ResnetBlock shortcut): y += W_skip * cat(skip_x[0], skip_x[1]) + skip_bias, sources in
  // the OUTPUT geometry; only the LDS-DMA kernel (cfg 11) implements it
  const void* skip_x[2]; long long skip_ld[2]; int skip_cin[2];
  const void* skip_w;             // gm_pack_conv_weight of the [Cout][skip_cin[0] + skip_cin[1]] 1x1 kernel
  const float* skip_bias;         // [Cout] or null
  // optional second input source (LDS-DMA kernels only): the input is the channel concatenation cat(x[..., :cin_split], x2) of two
  // tensors in the same geometry -- torch.cat([h, skip], dim=1) of the decoder blocks, never materialised.  cin_split % BK == 0.
  const void* x2; long long x2_ld; int cin_split;
  // optional split-K (LDS-DMA 3x3x3 stride-1 kernels, small grids): the K chunks are dealt to `ksplit` work-groups per tile, each
  // writing its fp32 partial accumulators to kpartial[ks][n * V + voxel][Cout]; gm_conv_forward then runs the combine kernel (sum of the
  // slices + bias / timestep row / residual / activation / output statistics).  ksplit <= 1: off.
  int ksplit; float* kpartial;
  // optional GroupNorm prologue given as STATISTICS instead of (pre_scale, pre_shift) -- tile configurations 24 / 25 (conv_sn.hip) only: the consumer folds the
  // per-tile partials of its input (up to two channel-concatenated sources, S_i <= 64 rows of [N][C_i][2] fp64 each, as gm_gn_finalize_channels takes them) and
  // forms scale = rstd * gamma, shift = beta - mean * rstd * gamma itself, in its prologue: bit-identical to gm_gn_finalize_channels, one launch less per norm.
  // pre_stats[0] == NULL: off (pre_scale / pre_shift as before).  pre_act applies as usual.
  const double* pre_stats[2]; int pre_S[2]; int pre_C[2];
  const float* pre_gamma; const float* pre_beta;   // [Cin] fp32 or NULL (1 / 0)
  float pre_eps; int pre_groups;
};

// slot count of the zero-initialised, atomically accumulated statistic tables the BACKWARD kernels still use ([GM_STAT_SLOTS][N][C][2];
// gm_gn_bwd_stats, gm_layernorm_bwd).  The forward tables (convolution epilogues, gm_gn_channel_stats) hold one partial per tile instead.
#define GM_STAT_SLOTS 64

#define CONV_ROWB 80  // LDS row pitch in bytes: 64 B of operands + 16 B pad

template <typename T> struct ConvTraits;
template <> struct ConvTraits<bf16_raw> { static constexpr int BK = 32; static constexpr int VECW = 8; };
template <> struct ConvTraits<float> { static constexpr int BK = 16; static constexpr int VECW = 4; };

__device__ __forceinline__ float conv_act(float v, int act, bool precise) {
  switch (act) {
    case 1: return precise ? gm_silu_precise(v) : gm_silu(v);
    case 2: return fmaxf(v, 0.f);
    default: return v;
  }
}
// The activation of a whole register vector with the (wave-uniform, run-time) kind tested ONCE.  `for i: v[i] = conv_act(v[i], act, ..)` made
// hipcc emit the switch PER ELEMENT: every element became its own basic block -- s_cmp / s_cbranch, then the fully dependent chain v_mul ->
// v_exp -> v_add -> v_rcp -> v_mul with the transcendental latencies exposed and nothing to interleave (round-3 ISA read: the out head's fused
// GroupNorm + SiLU prologue cost 0.15 of its 0.26 ms, the in-LDS prologue of the LDS-DMA convolution as much as a separate pass over HBM).  Here
// the N chains of a vector sit in one straight-line block and the scheduler interleaves them.
template <int N>
__device__ __forceinline__ void conv_act_vec(float (&v)[N], int act, bool precise) {
  if (act == 1) {
    if (precise) {
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = gm_silu_precise(v[i]);
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = gm_silu(v[i]);
    }
  } else if (act == 2) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = fmaxf(v[i], 0.f);
  }
}
__device__ __forceinline__ float conv_post_act(float v, int act) {
  switch (act) {
    case 1: return fmaxf(v, 0.f);
    case 2: return tanhf(v);
    case 3: return 1.0f / (1.0f + expf(-v));
    case 4: return gm_silu_precise(v);
    case 5: return v > 0.f ? v : 0.01f * v;
    case 6: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));  // nn.GELU() (exact), MONAI MLPBlock
    default: return v;
  }
}

// 16-byte operand vector <-> 4/8 floats
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static __device__ __forceinline__ void unpack(const uint4& v, float* o) {
    o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
  }
  static __device__ __forceinline__ uint4 pack(const float* o) {
    return make_uint4(__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3]));
  }
};
template <> struct Vec16<bf16_raw> {
  static __device__ __forceinline__ void unpack(const uint4& v, float* o) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
  static __device__ __forceinline__ uint4 pack(const float* o) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(o[2 * i], o[2 * i + 1]);
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
};

template <typename T> struct Mma;
template <> struct Mma<bf16_raw> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

