// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of generativemodels_amd.
// Everything here is wave64 / CDNA4 specific by design: no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GM_F32 0
#define GM_BF16 1

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef uint16_t bf16_raw;  // storage type of a bf16 element in HBM / LDS

// ---- error plumbing shared by every extern "C" entry point ---------------------------------------------------------
extern "C" void gm_set_error(const char* where, int code, const char* msg);
#define GM_FAIL(code, msg)                 \
  do {                                     \
    gm_set_error(__func__, (code), (msg)); \
    return (code);                         \
  } while (0)
#define GM_REQUIRE(cond, msg) \
  do {                        \
    if (!(cond)) GM_FAIL(-1, msg); \
  } while (0)
#define GM_LAUNCH_CHECK()                                          \
  do {                                                             \
    hipError_t e__ = hipGetLastError();                            \
    if (e__ != hipSuccess) GM_FAIL((int)e__, hipGetErrorString(e__)); \
    return 0;                                                      \
  } while (0)

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) ------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_raw v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_raw f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_raw)((u >> 16) | 0x40);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_raw)(u >> 16);
}

template <typename T> struct ElemIO;
template <> struct ElemIO<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct ElemIO<bf16_raw> {
  static __device__ __forceinline__ float ld(const bf16_raw* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_raw* p, float v) { *p = f32_to_bf16(v); }
};

// two fp32 -> packed bf16x2 (lo in bits 0..15) with the gfx950 hardware converter (round-to-nearest-even)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// fast SiLU for the bf16 path: v_exp_f32 + v_rcp_f32 (about 1 ulp each; far below bf16 resolution)
__device__ __forceinline__ float gm_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float gm_silu_precise(float x) { return x / (1.0f + expf(-x)); }

// wave64 butterfly reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- GroupNorm finalisation over SHORT statistic tables (every source S <= GN_SHORT_MAX_ROWS rows of [N][C_i][2] fp64 partials) -- shared by gm_gn_finalize_channels
// (groupnorm.hip) and the consumer-side prologue of conv_sn.hip, which must agree bit for bit: a channel's partials are added in ROW order, a group is the sum
// of its channels in CHANNEL order, everything in fp64.  Channel c of the concatenation (C0 + C1 channels) lives in source 0 when c < C0.
#define GN_SHORT_MAX_ROWS 128  // (64 until the second half of round 6: the 32^3 level of a latent UNet leaves 128-row tables -- one partial per 256-voxel tile)
__device__ __forceinline__ double2 gn_short_channel_sum(const double* s0, int S0, int C0, const double* s1, int S1, int C1, int N, int n, int c) {
  const bool first = c < C0;
  const double* src = first ? s0 + ((long long)n * C0 + c) * 2 : s1 + ((long long)n * C1 + (c - C0)) * 2;
  const long long pitch = (long long)N * (first ? C0 : C1) * 2;
  const int S = first ? S0 : S1;
  double a = 0.0, b = 0.0;
  int sl = 0;
  for (; sl + 8 <= S; sl += 8) {  // eight rows in flight, added in row order
    double2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const double2*>(src + (long long)(sl + k) * pitch);
#pragma unroll
    for (int k = 0; k < 8; ++k) { a += v[k].x; b += v[k].y; }
  }
  for (; sl < S; ++sl) {
    const double2 v = *reinterpret_cast<const double2*>(src + (long long)sl * pitch);
    a += v.x; b += v.y;
  }
  return make_double2(a, b);
}
// (a, b) = the group's (sum, sum of squares) over cpg channels x V voxels -> this channel's scale = rstd * gamma, shift = beta - mean * rstd * gamma
__device__ __forceinline__ void gn_short_scale_shift(double a, double b, int cpg, long long V, float eps, float gamma, float beta, float& scale, float& shift) {
  const double cnt = (double)cpg * (double)V;
  const double mean = a / cnt;
  double var = b / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  scale = (float)(rstd * (double)gamma);
  shift = (float)((double)beta - mean * rstd * (double)gamma);
}

static inline int gm_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
