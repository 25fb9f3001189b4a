// Latency-bound small-problem kernels: the shapes of an autoregressive decode step and of the timestep-embedding MLP, where the
// tiled implicit-GEMM convolution (K-chunk loop over LDS with two barriers per chunk) spends 10-30 us on a few rows.
//   gm_linear_rows      y[rows][cout] = post(pre(x)[rows][cin] W^T + b) (+ res) for a handful of rows: one wave per 16 output
//                       channels streams its weight rows global -> registers -> MFMA (no LDS, no barrier, loads unrolled 4 deep)
//                       (reference: nn.Linear in transformer blocks, time_embed / time_emb_proj, diffusion_model_unet.py:1758-1760)
//   gm_attention_decode softmax(scale q K^T) V for ONE query per (batch, head) over a KV cache: keys spread over the 256 threads,
//                       scores through LDS, the PV sum parallel over (channel, key slice)
//                       (reference: blocks/selfattention.py:117-147 evaluated for the last position only)
#include "attn_common.h"
#include "conv_common.h"

template <typename T>
__global__ __launch_bounds__(256) void linear_rows_kernel(const T* __restrict__ x, long long x_ld, const T* __restrict__ w,
                                                         const float* __restrict__ bias, const T* __restrict__ res, long long res_ld,
                                                         T* __restrict__ y, long long y_ld, int rows, int cin, int cout, int pre_act,
                                                         int post_act) {
  constexpr int BK = ConvTraits<T>::BK, VECW = ConvTraits<T>::VECW;
  constexpr bool PRECISE = sizeof(T) == 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int cout_pad = (cout + 15) & ~15;
  const int co0 = (blockIdx.x * 4 + wave) * 16;
  if (co0 >= cout_pad) return;  // wave-uniform
  const int r0 = blockIdx.y * 16;
  const int row = r0 + l15;
  const bool row_ok = row < rows;
  const int nchunks = (cin + BK - 1) / BK;
  const T* wrow = w + ((long long)(co0 + l15)) * BK + q * VECW;      // + chunk * cout_pad * BK
  const T* xrow = x + (long long)(row_ok ? row : 0) * x_ld + q * VECW;  // + chunk * BK
  f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  constexpr int U = 4;
  for (int c0 = 0; c0 < nchunks; c0 += U) {
    uint4 wf[U], xf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u < nchunks ? c0 + u : nchunks - 1;  // clamped: the duplicate is discarded below
      wf[u] = *reinterpret_cast<const uint4*>(wrow + (long long)c * cout_pad * BK);
      const bool ok = row_ok & (c * BK + q * VECW + VECW <= cin);  // host: cin % VECW == 0
      const uint4 v = *reinterpret_cast<const uint4*>(xrow + (ok ? c * BK : 0));
      xf[u] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (c0 + u >= nchunks) break;
      uint4 b = xf[u];
      if (pre_act) {
        float v[VECW];
        Vec16<T>::unpack(b, v);
#pragma unroll
        for (int i = 0; i < VECW; ++i) v[i] = conv_act(v[i], pre_act, PRECISE);
        b = Vec16<T>::pack(v);
      }
      Mma<T>::run(wf[u], b, acc);
    }
  }
  // D layout: column = row l15, rows = output channels co0 + 4q + i
  if (!row_ok) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int co = co0 + 4 * q + i;
    if (co < cout) {
      float v = acc[i] + (bias ? bias[co] : 0.f);
      v = conv_post_act(v, post_act);
      if (res) v += ElemIO<T>::ld(res + (long long)row * res_ld + co);
      ElemIO<T>::st(y + (long long)row * y_ld + co, v);
    }
  }
}

extern "C" int gm_linear_rows(const void* x, long long x_ld, const void* w, const float* bias, const void* res, long long res_ld, void* y,
                              long long y_ld, int rows, int cin, int cout, int pre_act, int post_act, int dtype, void* stream) {
  GM_REQUIRE(x && w && y, "null pointer");
  GM_REQUIRE(rows >= 0 && cin > 0 && cout > 0, "bad geometry");
  if (rows == 0) return 0;
  const int vecw = dtype == GM_F32 ? 4 : 8;
  GM_REQUIRE(cin % vecw == 0 && x_ld % vecw == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "x rows must be 16-byte vectors");
  hipStream_t st = (hipStream_t)stream;
  const int cout_pad = (cout + 15) & ~15;
  dim3 grid((cout_pad / 16 + 3) / 4, (rows + 15) / 16);
  if (dtype == GM_F32)
    linear_rows_kernel<float><<<grid, 256, 0, st>>>((const float*)x, x_ld, (const float*)w, bias, (const float*)res, res_ld, (float*)y, y_ld,
                                                    rows, cin, cout, pre_act, post_act);
  else if (dtype == GM_BF16)
    linear_rows_kernel<bf16_raw><<<grid, 256, 0, st>>>((const bf16_raw*)x, x_ld, (const bf16_raw*)w, bias, (const bf16_raw*)res, res_ld,
                                                       (bf16_raw*)y, y_ld, rows, cin, cout, pre_act, post_act);
  else
    GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// one query per (batch, head)
// ---------------------------------------------------------------------------------------------------------------------------------
#define DEC_MAX_KEYS 15360  // scores live in LDS (fp32): 60 KiB + query + scratch stay under the default 64 KiB dynamic limit

template <typename T>
__global__ __launch_bounds__(256) void attn_decode_kernel(const GmAttnDesc p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sc = reinterpret_cast<float*>(smem);            // [Lk] scores, then probabilities
  float* qs = sc + ((p.Lk + 3) & ~3);                     // [dh] query (fp32)
  float* red = qs + p.dh;                                 // [256] reduction scratch / [slices][dh] partial outputs
  const int tid = threadIdx.x;
  const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
  const int dh = p.dh;
  const T* Q = reinterpret_cast<const T*>(p.q) + (long long)b * p.q_ld + h * dh;  // Lq == 1
  const T* K = reinterpret_cast<const T*>(p.k) + (long long)b * (p.k_bs ? p.k_bs : (long long)p.Lk * p.k_ld) + h * dh;
  const T* V = reinterpret_cast<const T*>(p.v) + (long long)b * (p.v_bs ? p.v_bs : (long long)p.Lk * p.v_ld) + h * dh;
  for (int c = tid; c < dh; c += 256) qs[c] = ElemIO<T>::ld(Q + c) * p.scale;
  __syncthreads();
  // ---- scores: thread <-> key -------------------------------------------------------------------------------------------------
  float mx = -INFINITY;
  constexpr int VECW = 16 / (int)sizeof(T);
  const bool kvec = (dh % VECW == 0) && (p.k_ld % VECW == 0) && ((reinterpret_cast<uintptr_t>(K) & 15) == 0);
  for (int j = tid; j < p.Lk; j += 256) {
    const T* kr = K + (long long)j * p.k_ld;
    float s = 0.f;
    if (kvec) {
      for (int c = 0; c < dh; c += VECW) {
        float kv[VECW];
        Vec16<T>::unpack(*reinterpret_cast<const uint4*>(kr + c), kv);
#pragma unroll
        for (int i = 0; i < VECW; ++i) s += qs[c + i] * kv[i];
      }
    } else {
      for (int c = 0; c < dh; ++c) s += qs[c] * ElemIO<T>::ld(kr + c);
    }
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j < p.Lk; j += 256) {
    const float e = expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  __syncthreads();
  // ---- output: thread <-> (channel c, key slice) ------------------------------------------------------------------------------------
  const int cpt = dh <= 256 ? dh : 256;                  // channels handled per pass (dh <= 256)
  const int slices = 256 / cpt;
  const int c = tid % cpt, sl = tid / cpt;
  float o = 0.f;
  if (sl < slices)
    for (int j = sl; j < p.Lk; j += slices) o += sc[j] * ElemIO<T>::ld(V + (long long)j * p.v_ld + c);
  red[tid] = sl < slices ? o : 0.f;
  __syncthreads();
  if (tid < cpt) {
    float tot = 0.f;
    for (int s2 = 0; s2 < slices; ++s2) tot += red[s2 * cpt + tid];
    float out = tot * inv;
    if (p.res) out += ElemIO<T>::ld(reinterpret_cast<const T*>(p.res) + (long long)b * p.res_ld + h * dh + tid);
    ElemIO<T>::st(reinterpret_cast<T*>(p.o) + (long long)b * p.o_ld + h * dh + tid, out);
  }
}

// returns 1 if launched, 0 if the geometry is not a single-query decode this kernel covers
extern "C" int gm_attention_decode_try(const GmAttnDesc* dp, void* stream) {
  const GmAttnDesc& d = *dp;
  if (d.Lq != 1 || d.Lk > DEC_MAX_KEYS || d.dh > 256 || (256 % (d.dh <= 256 ? d.dh : 256)) != 0 || d.Lk < 1) return 0;
  if (d.dtype != GM_F32 && d.dtype != GM_BF16) return 0;
  hipStream_t st = (hipStream_t)stream;
  const size_t smem = (size_t)(((d.Lk + 3) & ~3) + d.dh + 256) * sizeof(float);
  if (d.dtype == GM_F32) attn_decode_kernel<float><<<d.B * d.H, 256, smem, st>>>(d);
  else attn_decode_kernel<bf16_raw><<<d.B * d.H, 256, smem, st>>>(d);
  return 1;
}
