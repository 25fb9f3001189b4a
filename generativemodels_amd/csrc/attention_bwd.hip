// Flash-attention backward: dQ, dK, dV of O = softmax(scale Q K^T) V per (batch, head) without ever holding the
// L x L score matrix in HBM (reference: torch autograd through diffusion_model_unet.py:143-153 / 407-415, autoencoderkl.py:261-269).
//
// Three kernels, all in the forward kernel's orientation (attention.hip): the work-group's OWN rows are MFMA columns, one per lane, so
// every softmax quantity is lane-local and P / dS are consumed straight from the accumulators as the next MFMA's B operand.
//   attn_bwd_pre   own rows = queries: LSE[q] = log sum_k exp(scale q.k) (one QK^T sweep) and Dsum[q] = dO[q] . O[q]
//   attn_bwd_dq    own rows = queries: per key tile  S^T = K Q^T, dP^T = V dO^T, dS^T = P^T (dP^T - Dsum) scale, dQ^T += K^T dS^T
//   attn_bwd_dkv   own rows = keys:    per query tile S = Q K^T, dP = dO V^T, P, dS, dV^T += dO^T P, dK^T += Q^T dS
// (S is recomputed by both gradient kernels -- 7 GEMM units instead of 5 -- which keeps dQ free of atomics and every result
// deterministic.)  All products run on v_mfma_f32_16x16x4_f32 over fp32 LDS tiles, whatever the storage dtype: one k value per lane
// means every operand -- K^T, Q^T, dO^T included -- is read from a natural [row][channel] tile with no transposed staging.  That is
// 1/8 of the bf16 MFMA rate: right for the latent-resolution attention of a training step (C4: 512-4096 tokens), and what makes a
// 32768-token backward possible at all; a bf16 version needs transposed K / Q / dO images (ds_read_b64_tr_b16) and is the follow-up.
#include "gm_common.h"

struct GmAttnBwdDesc {
  const void* q; long long q_ld;
  const void* k; long long k_ld;
  const void* v; long long v_ld;
  const void* o; long long o_ld;       // forward output WITHOUT the residual
  const void* go; long long go_ld;     // gradient of the forward output
  void* dq; long long dq_ld;
  void* dk; long long dk_ld;
  void* dv; long long dv_ld;
  int B, H, Lq, Lk, dh;
  float scale;
  int dtype;
  void* workspace; long long workspace_bytes;  // gm_attention_backward_workspace_bytes
};

template <typename T> __device__ __forceinline__ float4 ab_load4(const T* p);
template <> __device__ __forceinline__ float4 ab_load4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ab_load4<bf16_raw>(const bf16_raw* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}
template <typename T> __device__ __forceinline__ void ab_store4(T* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void ab_store4<float>(float* p, float a, float b, float c, float d) { *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d); }
template <> __device__ __forceinline__ void ab_store4<bf16_raw>(bf16_raw* p, float a, float b, float c, float d) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
}

#define AB_ROWS 32  // rows of a streamed tile (two 16-row fragments)

// stage `AB_ROWS` rows [row0, row0 + AB_ROWS) of a [L][ld] operand (channels [0, DH)) as fp32 into lds[row][DH + 4]; rows >= L are zero
template <typename T, int DH>
__device__ __forceinline__ void ab_stage(const T* base, long long ld, int row0, int L, float* lds, int tid) {
  constexpr int PITCH = DH + 4;
  for (int it = tid; it < AB_ROWS * (DH / 4); it += (int)blockDim.x) {
    const int row = it / (DH / 4), c4 = it % (DH / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + row < L) v = ab_load4<T>(base + (long long)(row0 + row) * ld + c4 * 4);
    *reinterpret_cast<float4*>(lds + row * PITCH + c4 * 4) = v;
  }
}

// acc[f][r] (f = 16-row fragment of the tile, r) += sum_c tile[f*16 + l15][c] * own[c]: A = tile rows, B = own-row fragments
template <int DH>
__device__ __forceinline__ void ab_rows_dot(const float* lds, const float4 (&own)[DH / 16], f32x4_t (&acc)[2], int l15, int qg) {
  constexpr int PITCH = DH + 4;
#pragma unroll
  for (int s = 0; s < DH / 16; ++s)
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const float4 a = *reinterpret_cast<const float4*>(lds + (f * 16 + l15) * PITCH + s * 16 + qg * 4);
      acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, own[s].x, acc[f], 0, 0, 0);
      acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, own[s].y, acc[f], 0, 0, 0);
      acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, own[s].z, acc[f], 0, 0, 0);
      acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, own[s].w, acc[f], 0, 0, 0);
    }
}

// out[d][r] (channel c0 + d*16 + 4qg + r of this lane's own row) += sum_rows tile[row][c0 + d*16 + l15] * w[row]: A = tile^T read element-wise
// from the natural tile, B = the per-row weights held in the accumulator layout (w[f][i] belongs to tile row f*16 + 4qg + i)
template <int DH, int DF>
__device__ __forceinline__ void ab_cols_acc(const float* lds, const f32x4_t (&w)[2], f32x4_t (&out)[DF], int c0, int l15, int qg) {
  constexpr int PITCH = DH + 4;
#pragma unroll
  for (int d = 0; d < DF; ++d)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a = lds[(f * 16 + qg * 4 + i) * PITCH + c0 + d * 16 + l15];
        out[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w[f][i], out[d], 0, 0, 0);
      }
}

template <typename T, int DH>
__device__ __forceinline__ void ab_load_own(const T* row, bool ok, float4 (&own)[DH / 16], int qg) {
#pragma unroll
  for (int s = 0; s < DH / 16; ++s) own[s] = ok ? ab_load4<T>(row + s * 16 + qg * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- LSE and Dsum ---------------------------------------------------------------------------------------------------------------
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_bwd_pre_kernel(const GmAttnBwdDesc p, float* __restrict__ lse, float* __restrict__ dsum) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ldsK = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, qg = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int my_q = blockIdx.x * ((int)blockDim.x >> 2) + wave * 16 + l15;
  const bool q_ok = my_q < p.Lq;
  const T* Qb = reinterpret_cast<const T*>(p.q) + (long long)b * p.Lq * p.q_ld + h * p.dh;
  const T* Kb = reinterpret_cast<const T*>(p.k) + (long long)b * p.Lk * p.k_ld + h * p.dh;
  float4 qf[DH / 16];
  ab_load_own<T, DH>(Qb + (long long)(q_ok ? my_q : 0) * p.q_ld, q_ok, qf, qg);
  float m_run = -INFINITY, l_run = 0.f;
  for (int key0 = 0; key0 < p.Lk; key0 += AB_ROWS) {
    __syncthreads();
    ab_stage<T, DH>(Kb, p.k_ld, key0, p.Lk, ldsK, tid);
    __syncthreads();
    f32x4_t s[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
    ab_rows_dot<DH>(ldsK, qf, s, l15, qg);
    float tmax = -INFINITY;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sv = key0 + f * 16 + qg * 4 + r < p.Lk ? s[f][r] * p.scale : -INFINITY;
        s[f][r] = sv;
        tmax = fmaxf(tmax, sv);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    float psum = 0.f;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) psum += expf(s[f][r] - m_new);
    l_run = l_run * expf(m_run - m_new) + psum;
    m_run = m_new;
  }
  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  // Dsum: this lane's channels of dO . O, then over the four lanes of the query
  const T* Ob = reinterpret_cast<const T*>(p.o) + ((long long)b * p.Lq + (q_ok ? my_q : 0)) * p.o_ld + h * p.dh;
  const T* Gb = reinterpret_cast<const T*>(p.go) + ((long long)b * p.Lq + (q_ok ? my_q : 0)) * p.go_ld + h * p.dh;
  float dot = 0.f;
  if (q_ok) {
#pragma unroll
    for (int s = 0; s < DH / 16; ++s) {
      const float4 a = ab_load4<T>(Ob + s * 16 + qg * 4), g = ab_load4<T>(Gb + s * 16 + qg * 4);
      dot += a.x * g.x + a.y * g.y + a.z * g.z + a.w * g.w;
    }
  }
  dot += __shfl_xor(dot, 16, 64);
  dot += __shfl_xor(dot, 32, 64);
  if (q_ok && qg == 0) {
    lse[(long long)bh * p.Lq + my_q] = m_run + logf(l_tot);
    dsum[(long long)bh * p.Lq + my_q] = dot;
  }
}

// ---- dQ ---------------------------------------------------------------------------------------------------------------------------
// CS: the head dim is covered in CS channel slices (blockIdx.z) of DH / CS accumulated channels each (register budget at d = 256)
template <typename T, int DH, int CS>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const GmAttnBwdDesc p, const float* __restrict__ lse, const float* __restrict__ dsum) {
  constexpr int DF = DH / CS / 16, PITCH = DH + 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ldsK = reinterpret_cast<float*>(smem);
  float* ldsV = ldsK + AB_ROWS * PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, qg = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int c0 = blockIdx.z * (DH / CS);
  const int my_q = blockIdx.x * ((int)blockDim.x >> 2) + wave * 16 + l15;
  const bool q_ok = my_q < p.Lq;
  const long long qrow = (long long)b * p.Lq + (q_ok ? my_q : 0);
  const T* Kb = reinterpret_cast<const T*>(p.k) + (long long)b * p.Lk * p.k_ld + h * p.dh;
  const T* Vb = reinterpret_cast<const T*>(p.v) + (long long)b * p.Lk * p.v_ld + h * p.dh;
  float4 qf[DH / 16], gf[DH / 16];
  ab_load_own<T, DH>(reinterpret_cast<const T*>(p.q) + qrow * p.q_ld + h * p.dh, q_ok, qf, qg);
  ab_load_own<T, DH>(reinterpret_cast<const T*>(p.go) + qrow * p.go_ld + h * p.dh, q_ok, gf, qg);
  const float my_lse = q_ok ? lse[(long long)bh * p.Lq + my_q] : 0.f;
  const float my_d = q_ok ? dsum[(long long)bh * p.Lq + my_q] : 0.f;
  f32x4_t acc[DF];
#pragma unroll
  for (int d = 0; d < DF; ++d) acc[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int key0 = 0; key0 < p.Lk; key0 += AB_ROWS) {
    __syncthreads();
    ab_stage<T, DH>(Kb, p.k_ld, key0, p.Lk, ldsK, tid);
    ab_stage<T, DH>(Vb, p.v_ld, key0, p.Lk, ldsV, tid);
    __syncthreads();
    f32x4_t s[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}}, dp[2] = {s[0], s[0]};
    ab_rows_dot<DH>(ldsK, qf, s, l15, qg);   // S^T  = K Q^T
    ab_rows_dot<DH>(ldsV, gf, dp, l15, qg);  // dP^T = V dO^T
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = key0 + f * 16 + qg * 4 + r < p.Lk;
        const float pr = ok ? expf(s[f][r] * p.scale - my_lse) : 0.f;
        s[f][r] = pr * (dp[f][r] - my_d) * p.scale;  // dS^T
      }
    ab_cols_acc<DH, DF>(ldsK, s, acc, c0, l15, qg);  // dQ^T += K^T dS^T
  }
  if (!q_ok) return;
  T* out = reinterpret_cast<T*>(p.dq) + qrow * p.dq_ld + h * p.dh + c0;
#pragma unroll
  for (int d = 0; d < DF; ++d) ab_store4<T>(out + d * 16 + qg * 4, acc[d][0], acc[d][1], acc[d][2], acc[d][3]);
}

// ---- dK, dV ---------------------------------------------------------------------------------------------------------------------
template <typename T, int DH, int CS>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const GmAttnBwdDesc p, const float* __restrict__ lse, const float* __restrict__ dsum) {
  constexpr int DF = DH / CS / 16, PITCH = DH + 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ldsQ = reinterpret_cast<float*>(smem);
  float* ldsG = ldsQ + AB_ROWS * PITCH;
  float* ldsL = ldsG + AB_ROWS * PITCH;  // [AB_ROWS] lse, then [AB_ROWS] dsum
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, qg = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int c0 = blockIdx.z * (DH / CS);
  const int my_k = blockIdx.x * ((int)blockDim.x >> 2) + wave * 16 + l15;
  const bool k_ok = my_k < p.Lk;
  const long long krow = (long long)b * p.Lk + (k_ok ? my_k : 0);
  const T* Qb = reinterpret_cast<const T*>(p.q) + (long long)b * p.Lq * p.q_ld + h * p.dh;
  const T* Gb = reinterpret_cast<const T*>(p.go) + (long long)b * p.Lq * p.go_ld + h * p.dh;
  float4 kf[DH / 16], vf[DH / 16];
  ab_load_own<T, DH>(reinterpret_cast<const T*>(p.k) + krow * p.k_ld + h * p.dh, k_ok, kf, qg);
  ab_load_own<T, DH>(reinterpret_cast<const T*>(p.v) + krow * p.v_ld + h * p.dh, k_ok, vf, qg);
  f32x4_t dka[DF], dva[DF];
#pragma unroll
  for (int d = 0; d < DF; ++d) { dka[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dva[d] = dka[d]; }
  for (int q0 = 0; q0 < p.Lq; q0 += AB_ROWS) {
    __syncthreads();
    ab_stage<T, DH>(Qb, p.q_ld, q0, p.Lq, ldsQ, tid);
    ab_stage<T, DH>(Gb, p.go_ld, q0, p.Lq, ldsG, tid);
    if (tid < AB_ROWS) {
      const bool ok = q0 + tid < p.Lq;
      ldsL[tid] = ok ? lse[(long long)bh * p.Lq + q0 + tid] : INFINITY;  // exp(s - inf) = 0: rows past the sequence contribute nothing
      ldsL[AB_ROWS + tid] = ok ? dsum[(long long)bh * p.Lq + q0 + tid] : 0.f;
    }
    __syncthreads();
    f32x4_t s[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}}, dp[2] = {s[0], s[0]};
    ab_rows_dot<DH>(ldsQ, kf, s, l15, qg);   // S  = Q K^T   (rows: queries of the tile, column: this lane's key)
    ab_rows_dot<DH>(ldsG, vf, dp, l15, qg);  // dP = dO V^T
    f32x4_t pm[2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = f * 16 + qg * 4 + r;
        const float pr = k_ok ? expf(s[f][r] * p.scale - ldsL[row]) : 0.f;
        pm[f][r] = pr;
        s[f][r] = pr * (dp[f][r] - ldsL[AB_ROWS + row]) * p.scale;  // dS
      }
    ab_cols_acc<DH, DF>(ldsG, pm, dva, c0, l15, qg);  // dV^T += dO^T P
    ab_cols_acc<DH, DF>(ldsQ, s, dka, c0, l15, qg);   // dK^T += Q^T dS
  }
  if (!k_ok) return;
  T* ok_ = reinterpret_cast<T*>(p.dk) + krow * p.dk_ld + h * p.dh + c0;
  T* ov_ = reinterpret_cast<T*>(p.dv) + krow * p.dv_ld + h * p.dh + c0;
#pragma unroll
  for (int d = 0; d < DF; ++d) {
    ab_store4<T>(ok_ + d * 16 + qg * 4, dka[d][0], dka[d][1], dka[d][2], dka[d][3]);
    ab_store4<T>(ov_ + d * 16 + qg * 4, dva[d][0], dva[d][1], dva[d][2], dva[d][3]);
  }
}

// ---- host -------------------------------------------------------------------------------------------------------------------------
extern "C" long long gm_attention_backward_workspace_bytes(const GmAttnBwdDesc* d) {
  if (!d) return -1;
  return 2LL * d->B * d->H * d->Lq * (long long)sizeof(float);
}

template <typename KernT>
static void ab_set_lds(KernT kern) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) (void)hipGetLastError();
}

template <typename T, int DH, int CS>
static void launch_attn_bwd(const GmAttnBwdDesc& d, hipStream_t st) {
  float* lse = reinterpret_cast<float*>(d.workspace);
  float* dsum = lse + (long long)d.B * d.H * d.Lq;
  constexpr size_t tile = (size_t)AB_ROWS * (DH + 4) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    ab_set_lds(attn_bwd_pre_kernel<T, DH>);
    ab_set_lds(attn_bwd_dq_kernel<T, DH, CS>);
    ab_set_lds(attn_bwd_dkv_kernel<T, DH, CS>);
    attr_set = true;
  }
  // own rows per work-group: 64 (4 waves share every streamed tile).  Measured at L = 4096, d = 128, one head (64 work-groups on 256
  // CUs): one-wave work-groups (256 of them) were 2x SLOWER -- a lone wave staging whole tiles costs more than the idle CUs; short
  // single-head problems are better served by the composed backward (autograd.py), long or multi-head ones fill the chip anyway.
  auto waves = [&](int, int) { return 4; };
  const int wq0 = waves(d.Lq, 1), wq = waves(d.Lq, CS), wk = waves(d.Lk, CS);
  attn_bwd_pre_kernel<T, DH><<<dim3((d.Lq + 16 * wq0 - 1) / (16 * wq0), d.B * d.H), 64 * wq0, tile, st>>>(d, lse, dsum);
  attn_bwd_dq_kernel<T, DH, CS><<<dim3((d.Lq + 16 * wq - 1) / (16 * wq), d.B * d.H, CS), 64 * wq, 2 * tile, st>>>(d, lse, dsum);
  attn_bwd_dkv_kernel<T, DH, CS><<<dim3((d.Lk + 16 * wk - 1) / (16 * wk), d.B * d.H, CS), 64 * wk, 2 * tile + 2 * AB_ROWS * sizeof(float), st>>>(
      d, lse, dsum);
}

template <typename T>
static int dispatch_attn_bwd(const GmAttnBwdDesc& d, hipStream_t st) {
  switch (d.dh) {
    case 16: launch_attn_bwd<T, 16, 1>(d, st); return 0;
    case 32: launch_attn_bwd<T, 32, 1>(d, st); return 0;
    case 64: launch_attn_bwd<T, 64, 1>(d, st); return 0;
    case 128: launch_attn_bwd<T, 128, 1>(d, st); return 0;
    case 256: launch_attn_bwd<T, 256, 2>(d, st); return 0;
    default: return -1;
  }
}

extern "C" int gm_attention_backward(const GmAttnBwdDesc* dp, void* stream) {
  GM_REQUIRE(dp, "null descriptor");
  const GmAttnBwdDesc& d = *dp;
  GM_REQUIRE(d.q && d.k && d.v && d.o && d.go && d.dq && d.dk && d.dv, "null tensor pointer");
  GM_REQUIRE(d.B >= 0 && d.H > 0 && d.Lk > 0, "bad batch / head geometry");
  GM_REQUIRE((long long)d.B * d.H <= 65535, "too many (batch, head) pairs for one launch");
  GM_REQUIRE(d.workspace && d.workspace_bytes >= gm_attention_backward_workspace_bytes(dp), "workspace too small");
  const int vec = d.dtype == GM_F32 ? 4 : 4;  // rows are read and written as 4-element vectors (16 / 8 bytes)
  const int al = d.dtype == GM_F32 ? 16 : 8;
  auto ok = [&](const void* p, long long ld) { return ld % vec == 0 && (reinterpret_cast<uintptr_t>(p) & (al - 1)) == 0; };
  GM_REQUIRE(ok(d.q, d.q_ld) && ok(d.k, d.k_ld) && ok(d.v, d.v_ld) && ok(d.o, d.o_ld) && ok(d.go, d.go_ld) && ok(d.dq, d.dq_ld) &&
                 ok(d.dk, d.dk_ld) && ok(d.dv, d.dv_ld), "operands must be 4-element aligned rows");
  if (d.B == 0 || d.Lq == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (d.dtype == GM_F32) rc = dispatch_attn_bwd<float>(d, st);
  else if (d.dtype == GM_BF16) rc = dispatch_attn_bwd<bf16_raw>(d, st);
  else GM_FAIL(-2, "unsupported dtype");
  GM_REQUIRE(rc == 0, "head dim must be 16, 32, 64, 128 or 256");
  GM_LAUNCH_CHECK();
}
