// Latency-bound small-problem kernels: the shapes of an autoregressive decode step and of the timestep-embedding MLP, where the
// tiled implicit-GEMM convolution (K-chunk loop over LDS with two barriers per chunk) spends 10-30 us on a few rows.
//   gm_linear_rows      y[rows][cout] = post(pre(x)[rows][cin] W^T + b) (+ res) for a handful of rows: one wave per 16 output
//                       channels streams its weight rows global -> registers -> MFMA (no LDS, no barrier, loads unrolled 4 deep)
//                       (reference: nn.Linear in transformer blocks, time_embed / time_emb_proj, diffusion_model_unet.py:1758-1760)
//   gm_attention_decode softmax(scale q K^T) V for ONE query per (batch, head) over a KV cache: keys spread over the 256 threads,
//                       scores through LDS, the PV sum parallel over (channel, key slice)
//                       (reference: blocks/selfattention.py:117-147 evaluated for the last position only)
#include <cstdlib>
#include "attn_common.h"
#include "conv_common.h"

// optional extras of the small-row GEMM: a LayerNorm over the input row as its prologue (transformer pre-norm blocks) and a split of
// the output channels over three destinations (the stacked q | k | v projection writing k and v straight into the KV caches)
struct LinearRowsExtra {
  const float* ln_g; const float* ln_b; float ln_eps;  // ln_g != null: x <- LayerNorm(x) * g + b before the GEMM
  void* y1; void* y2; long long y12_ld; int split;     // split > 0: channels [split, 2*split) -> y1, [2*split, 3*split) -> y2 (row pitch y12_ld)
  const int* off_dev; long long off_mul;               // y1 / y2 are advanced by (*off_dev) * off_mul elements at run time (KV-cache row = position)
  const float* kv_ws; int kv_dh;                       // K-split kernel STAGE 1: x = merge of the split-KV attention partials (head size kv_dh)
  const float* mlp_p; int mlp_nj; const void* mlp_x1; const float* mlp_b2; void* mlp_x0;  // K-split kernel STAGE 2: x = x1 + b2 + sum_j P[j]
  const float* pre_scale; const float* pre_shift; long long ss_ld; int rows_per_sample;   // linear_rows_kernel: x <- x * scale[n][c] + shift[n][c], n = row / rows_per_sample
  // linear_rows_kernel, stacked q | k | v projection of an attention block: output channels >= vt_c0 (the V columns) are ALSO stored into the
  // transposed, key-permuted image VT[sample * H + head][channel][position] the LDS-DMA attention kernel consumes (attention_dma.hip:
  // vt_pack_kernel's layout; rows_per_sample = tokens per sample = the image's row pitch, a multiple of 64): no separate pack launch
  void* vt; int vt_c0; int vt_dh;
  // token_gemm_wide_kernel<.., GN = true>: the per-sample GroupNorm of x given as the statistic tables of its producer(s) -- finalised in the kernel's prologue
  // (gm_linear_rows_gn; the shared short-table order of gm_common.h: bit-identical to gm_gn_finalize_channels + the (scale, shift) form)
  const double* gn_stats[2]; int gn_S[2]; int gn_C[2]; const float* gn_gamma; const float* gn_beta; float gn_eps; int gn_groups; int gn_N;
};

// One channel of the split-KV single-query attention, merged from its GM_DECODE_KV_SPLITS partials in range order.  Workspace =
// o[BH][NS][dh], then m[BH][NS], l[BH][NS].  The split count is a compile-time constant so that all 3 NS loads are in flight at once (a
// run-time loop waits for every load in turn: 16 dependent L2 round trips, measured +22 us per call).
__device__ __forceinline__ float kv_merge_one(const float* __restrict__ ws, int BH, int dh, int bh, int c) {
  constexpr int NS = GM_DECODE_KV_SPLITS;
  const float* o = ws + (long long)bh * NS * dh + c;
  const float* m = ws + (long long)BH * NS * dh + (long long)bh * NS;
  const float* l = m + (long long)BH * NS;
  float ms[NS], ls[NS], os[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) { ms[s] = m[s]; ls[s] = l[s]; os[s] = o[(long long)s * dh]; }
  float M = -INFINITY;
#pragma unroll
  for (int s = 0; s < NS; ++s) M = fmaxf(M, ms[s]);
  float L = 0.f, acc = 0.f;
#pragma unroll
  for (int s = 0; s < NS; ++s) {  // range order
    const float e = ms[s] == -INFINITY ? 0.f : expf(ms[s] - M);
    L += ls[s] * e;
    acc += os[s] * e;
  }
  return acc * (1.0f / L);
}

template <typename T>
__global__ __launch_bounds__(256) void linear_rows_kernel(const T* __restrict__ x, long long x_ld, const T* __restrict__ w,
                                                         const float* __restrict__ bias, const T* __restrict__ res, long long res_ld,
                                                         T* __restrict__ y, long long y_ld, int rows, int cin, int cout, int pre_act,
                                                         int post_act, LinearRowsExtra ex) {
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
  const T* xrow = x + (long long)(row_ok ? row : 0) * x_ld + q * VECW;  // + chunk * BK; only dereferenced under its `ok` guard
  // ---- LayerNorm statistics of this lane's row: the 4 lanes sharing l15 cover the row between them (two passes: mean, then the
  //      centred second moment, like the reference's fp32 computation) ------------------------------------------------------------
  float mean = 0.f, rstd = 1.f;
  if (ex.ln_g) {
    float sum = 0.f;
    for (int c = 0; c < nchunks; ++c) {
      if (c * BK + q * VECW + VECW <= cin) {
        float v[VECW];
        Vec16<T>::unpack(*reinterpret_cast<const uint4*>(xrow + c * BK), v);
#pragma unroll
        for (int i = 0; i < VECW; ++i) sum += v[i];
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    mean = sum / (float)cin;
    float sq = 0.f;
    for (int c = 0; c < nchunks; ++c) {
      if (c * BK + q * VECW + VECW <= cin) {
        float v[VECW];
        Vec16<T>::unpack(*reinterpret_cast<const uint4*>(xrow + c * BK), v);
#pragma unroll
        for (int i = 0; i < VECW; ++i) sq += (v[i] - mean) * (v[i] - mean);
      }
    }
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    rstd = 1.0f / sqrtf(sq / (float)cin + ex.ln_eps);
  }
  // bias and residual of this lane's outputs: requested up front through substitute addresses (round 3: a conditional load is a branch + a wait)
  float bia[4], rsd[4];
  {
    const float* bsrc = bias ? bias : reinterpret_cast<const float*>(w);
    const T* rsrc = res ? res + (long long)(row_ok ? row : 0) * res_ld : w;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = co0 + 4 * q + i;
      bia[i] = bsrc[bias ? (co < cout ? co : cout - 1) : 0];
      rsd[i] = ElemIO<T>::ld(rsrc + (res ? (co < cout ? co : cout - 1) : 0));
    }
  }
  f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  constexpr int U = 4;
  for (int c0 = 0; c0 < nchunks; c0 += U) {
    uint4 wf[U], xf[U];
    float asc[U][VECW], ash[U][VECW];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u < nchunks ? c0 + u : nchunks - 1;  // clamped: the duplicate is discarded below
      wf[u] = *reinterpret_cast<const uint4*>(wrow + (long long)c * cout_pad * BK);
      const bool ok = row_ok & (c * BK + q * VECW + VECW <= cin);  // host: cin % VECW == 0
      // masked lanes read the first vector of the tensor: xrow + 0 is q * VECW elements into the row, which lies beyond a row (and, in
      // the last row, beyond the allocation) whenever cin < 4 * VECW -- a faulting read even though its value is discarded
      const uint4 v = *reinterpret_cast<const uint4*>(ok ? xrow + c * BK : x);
      xf[u] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
      if (ex.pre_scale) {  // (uniform) per-sample GroupNorm affine of this lane's channels: vector loads at a clamped offset, values selected below
        const int cb = c * BK + q * VECW + VECW <= cin ? c * BK + q * VECW : 0;
        const long long off = (long long)((row_ok ? row : 0) / ex.rows_per_sample) * ex.ss_ld + cb;
#pragma unroll
        for (int i = 0; i < VECW; i += 4) {
          const float4 a = *reinterpret_cast<const float4*>(ex.pre_scale + off + i), b2 = *reinterpret_cast<const float4*>(ex.pre_shift + off + i);
          asc[u][i] = a.x; asc[u][i + 1] = a.y; asc[u][i + 2] = a.z; asc[u][i + 3] = a.w;
          ash[u][i] = b2.x; ash[u][i + 1] = b2.y; ash[u][i + 2] = b2.z; ash[u][i + 3] = b2.w;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (c0 + u >= nchunks) break;
      uint4 b = xf[u];
      if (pre_act || ex.ln_g || ex.pre_scale) {
        float v[VECW];
        Vec16<T>::unpack(b, v);
        if (ex.pre_scale) {
          const bool okc = row_ok & ((c0 + u) * BK + q * VECW + VECW <= cin);
#pragma unroll
          for (int i = 0; i < VECW; ++i) v[i] = okc ? v[i] * asc[u][i] + ash[u][i] : 0.f;
        }
        if (ex.ln_g) {
          const int cbase = (c0 + u) * BK + q * VECW;
          const bool ok = row_ok & (cbase + VECW <= cin);
#pragma unroll
          for (int i = 0; i < VECW; ++i) v[i] = ok ? (v[i] - mean) * rstd * ex.ln_g[cbase + i] + (ex.ln_b ? ex.ln_b[cbase + i] : 0.f) : 0.f;
        }
        if (pre_act) {
          conv_act_vec(v, pre_act, PRECISE);
        }
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
      float v = acc[i] + (bias ? bia[i] : 0.f);
      v = conv_post_act(v, post_act);
      if (res) v += rsd[i];
      if (ex.split > 0 && co >= ex.split) {
        T* dst = reinterpret_cast<T*>(co < 2 * ex.split ? ex.y1 : ex.y2) + (ex.off_dev ? (long long)(*ex.off_dev) * ex.off_mul : 0);
        ElemIO<T>::st(dst + (long long)row * ex.y12_ld + (co - (co < 2 * ex.split ? ex.split : 2 * ex.split)), v);
      } else {
        ElemIO<T>::st(y + (long long)row * y_ld + co, v);
        if (ex.vt && co >= ex.vt_c0) {
          const int cch = co - ex.vt_c0, L = ex.rows_per_sample;
          const int smp = row / L, key = row - smp * L;
          const int pos = (key & ~31) + ((key & 15) >> 2) * 8 + ((key >> 4) & 1) * 4 + (key & 3);  // key = blk*32 + half*16 + qq*4 + r -> blk*32 + qq*8 + half*4 + r
          const int heads = (cout - ex.vt_c0) / ex.vt_dh;
          ElemIO<T>::st(reinterpret_cast<T*>(ex.vt) + ((long long)(smp * heads + cch / ex.vt_dh) * ex.vt_dh + cch % ex.vt_dh) * L + pos, v);
        }
      }
    }
  }
}

// The same GEMM for MANY token rows (round 6): y = post_act(act(x * scale[n] + shift[n]) W^T + b) (+ res) with a wave owning 16 rows x NB * 16 output channels.
// linear_rows_kernel gives every 16-channel block of a row group its own wave: each re-reads the group's x rows (and re-applies the GroupNorm affine) and
// stores its results two bytes at a time.  At 16 384 rows (the q | k | v projections of BASELINE configs[0]'s attention blocks: 16 x 32 x 32 tokens, 64 -> 192)
// that is 12 passes over x and 3 072 work-groups for 0.4 GFLOP.  Here the x fragment of a K chunk is loaded and transformed ONCE per NB blocks, the four waves
// of a work-group take four consecutive row groups (the weight fragments they share meet in the CU's L1), and a lane stores its four consecutive channels as
// one vector.  Same accumulation order per output (K chunks in order, one MFMA each) as linear_rows_kernel: bit-identical results.  C_in a multiple of the MFMA
// K step; affine prologue, pre-/post-activation, residual and the V^T image as there.
template <typename T, int NB, bool GN>
__global__ __launch_bounds__(256) void token_gemm_wide_kernel(const T* __restrict__ x, long long x_ld, const T* __restrict__ w, const float* __restrict__ bias,
                                                             const T* __restrict__ res, long long res_ld, T* __restrict__ y, long long y_ld, int rows, int cin,
                                                             int cout, int pre_act, int post_act, LinearRowsExtra ex) {
  constexpr int BK = ConvTraits<T>::BK, VECW = ConvTraits<T>::VECW;
  constexpr bool PRECISE = sizeof(T) == 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int cout_pad = (cout + 15) & ~15;
  const int co0 = blockIdx.x * (16 * NB);
  const int r0 = (blockIdx.y * 4 + wave) * 16;
  // ---- GN: (scale | shift) of the work-group's sample over all input channels, from the statistic tables (every thread takes part: before any wave leaves) -----
  extern __shared__ __attribute__((aligned(16))) char gn_smem[];  // [scale[cin] | shift[cin]] fp32, [cin] fp64 (sum, sum of squares)
  if constexpr (GN) {
    float* tab = reinterpret_cast<float*>(gn_smem);
    double* csum = reinterpret_cast<double*>(gn_smem + 2 * (size_t)cin * 4);
    const int n = (blockIdx.y * 64) / ex.rows_per_sample;  // host: rows_per_sample % 64 == 0 -- one sample per work-group
    const int cpg = cin / ex.gn_groups;
    for (int c = threadIdx.x; c < cin; c += 256) {
      const double2 v = gn_short_channel_sum(ex.gn_stats[0], ex.gn_S[0], ex.gn_C[0], ex.gn_stats[1], ex.gn_S[1], ex.gn_C[1], ex.gn_N, n, c);
      csum[2 * c] = v.x; csum[2 * c + 1] = v.y;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cin; c += 256) {
      const int g = c / cpg;
      double a = 0.0, b2 = 0.0;
      for (int j = 0; j < cpg; ++j) { a += csum[2 * (g * cpg + j)]; b2 += csum[2 * (g * cpg + j) + 1]; }
      float sc1, sh1;
      gn_short_scale_shift(a, b2, cpg, (long long)ex.rows_per_sample, ex.gn_eps, ex.gn_gamma ? ex.gn_gamma[c] : 1.f, ex.gn_beta ? ex.gn_beta[c] : 0.f, sc1, sh1);
      tab[c] = sc1;
      tab[cin + c] = sh1;
    }
    __syncthreads();
  }
  if (r0 >= rows) return;  // wave-uniform
  const int row = r0 + l15;
  const bool row_ok = row < rows;
  const int rowc = row_ok ? row : rows - 1;
  const int nchunks = cin / BK;  // host: cin % BK == 0
  const T* xrow = x + (long long)rowc * x_ld + q * VECW;
  // block nb of this wave: channels co0 + 16 nb .. + 15; blocks beyond the padded panel re-read the last one and are dropped at the store
  int wblk[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) wblk[nb] = co0 + 16 * nb < cout_pad ? co0 + 16 * nb : cout_pad - 16;
  const T* wrow = w + (long long)l15 * BK + q * VECW;  // + (chunk * cout_pad + block channel) * BK
  const long long aoff = ex.pre_scale ? (long long)(rowc / ex.rows_per_sample) * ex.ss_ld + q * VECW : 0;
  // bias / residual of this lane's outputs, requested up front at clamped addresses
  float bia[NB][4];
  uint2 rsd[NB];
  const bool vec4 = (cout & 3) == 0 && (y_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & (4 * sizeof(T) - 1)) == 0 &&
                    (!res || ((res_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(res) & (4 * sizeof(T) - 1)) == 0));
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = wblk[nb] + 4 * q + i;
      bia[nb][i] = bias ? bias[co < cout ? co : cout - 1] : 0.f;
    }
    rsd[nb] = make_uint2(0u, 0u);
    if (sizeof(T) == 2 && res && vec4) {
      const int co = wblk[nb] + 4 * q;
      rsd[nb] = *reinterpret_cast<const uint2*>(res + (long long)rowc * res_ld + (co + 3 < cout ? co : 0));
    }
  }
  f32x4_t acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  constexpr int U = 2;
  for (int c0 = 0; c0 < nchunks; c0 += U) {
    uint4 wf[U][NB], xf[U];
    float asc[U][VECW], ash[U][VECW];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u < nchunks ? c0 + u : nchunks - 1;  // clamped: the duplicate is discarded below
      xf[u] = *reinterpret_cast<const uint4*>(xrow + c * BK);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) wf[u][nb] = *reinterpret_cast<const uint4*>(wrow + ((long long)c * cout_pad + wblk[nb]) * BK);
      if (GN) {
        const float* tab = reinterpret_cast<const float*>(gn_smem) + c * BK + q * VECW;
#pragma unroll
        for (int i = 0; i < VECW; i += 4) {
          const float4 a = *reinterpret_cast<const float4*>(tab + i), b2 = *reinterpret_cast<const float4*>(tab + cin + i);
          asc[u][i] = a.x; asc[u][i + 1] = a.y; asc[u][i + 2] = a.z; asc[u][i + 3] = a.w;
          ash[u][i] = b2.x; ash[u][i + 1] = b2.y; ash[u][i + 2] = b2.z; ash[u][i + 3] = b2.w;
        }
      } else if (ex.pre_scale) {  // (uniform)
#pragma unroll
        for (int i = 0; i < VECW; i += 4) {
          const float4 a = *reinterpret_cast<const float4*>(ex.pre_scale + aoff + c * BK + i), b2 = *reinterpret_cast<const float4*>(ex.pre_shift + aoff + c * BK + i);
          asc[u][i] = a.x; asc[u][i + 1] = a.y; asc[u][i + 2] = a.z; asc[u][i + 3] = a.w;
          ash[u][i] = b2.x; ash[u][i + 1] = b2.y; ash[u][i + 2] = b2.z; ash[u][i + 3] = b2.w;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (c0 + u >= nchunks) break;
      uint4 b = row_ok ? xf[u] : make_uint4(0u, 0u, 0u, 0u);
      if (pre_act || ex.pre_scale || GN) {
        float v[VECW];
        Vec16<T>::unpack(b, v);
        if (ex.pre_scale || GN) {
#pragma unroll
          for (int i = 0; i < VECW; ++i) v[i] = row_ok ? v[i] * asc[u][i] + ash[u][i] : 0.f;
        }
        if (pre_act) conv_act_vec(v, pre_act, PRECISE);
        b = Vec16<T>::pack(v);
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) Mma<T>::run(wf[u][nb], b, acc[nb]);
    }
  }
  if (!row_ok) return;
  // D layout: column = row l15, rows = output channels block + 4q + i
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    if (co0 + 16 * nb >= cout_pad) break;  // (uniform)
    const int cob = co0 + 16 * nb + 4 * q;
    float o[4];
    float rs[4] = {0.f, 0.f, 0.f, 0.f};
    if (res) {
      if (sizeof(T) == 2 && vec4) {
        rs[0] = __uint_as_float(rsd[nb].x << 16); rs[1] = __uint_as_float(rsd[nb].x & 0xffff0000u);
        rs[2] = __uint_as_float(rsd[nb].y << 16); rs[3] = __uint_as_float(rsd[nb].y & 0xffff0000u);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) rs[i] = cob + i < cout ? ElemIO<T>::ld(res + (long long)row * res_ld + cob + i) : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = acc[nb][i] + (bias ? bia[nb][i] : 0.f);
      v = conv_post_act(v, post_act);
      if (res) v += rs[i];
      o[i] = v;
    }
    if (vec4) {
      if (cob < cout) {
        if (sizeof(T) == 2) *reinterpret_cast<uint2*>(y + (long long)row * y_ld + cob) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
        else *reinterpret_cast<float4*>(y + (long long)row * y_ld + cob) = make_float4(o[0], o[1], o[2], o[3]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (cob + i < cout) ElemIO<T>::st(y + (long long)row * y_ld + cob + i, o[i]);
    }
    if (ex.vt && cob + 3 >= ex.vt_c0) {
      const int L = ex.rows_per_sample, smp = row / L, key = row - smp * L;
      const int pos = (key & ~31) + ((key & 15) >> 2) * 8 + ((key >> 4) & 1) * 4 + (key & 3);  // key = blk*32 + half*16 + qq*4 + r -> blk*32 + qq*8 + half*4 + r
      const int heads = (cout - ex.vt_c0) / ex.vt_dh;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cch = cob + i - ex.vt_c0;
        if (cch >= 0 && cob + i < cout)
          ElemIO<T>::st(reinterpret_cast<T*>(ex.vt) + ((long long)(smp * heads + cch / ex.vt_dh) * ex.vt_dh + cch % ex.vt_dh) * L + pos, o[i]);
      }
    }
  }
}

// rows from which the wide form takes the affine token GEMMs (0 = never), and its blocks per wave: tools / tests pin them, the defaults are the measured ones
// (nb 0 = by row count: 4 blocks per wave from 8 192 rows, 2 below -- MI355X, profiles/r06_token_gemm_wide_ab.txt: 16 384 x 64 -> 192 runs best at 3-4, 4 096 x 128 -> 384 at 2)
static int gm_token_gemm_wide_rows = 2048, gm_token_gemm_wide_nb = 0;
extern "C" void gm_token_gemm_set_wide(int min_rows, int nb) {
  gm_token_gemm_wide_rows = min_rows < 0 ? 2048 : min_rows;
  gm_token_gemm_wide_nb = (nb == 2 || nb == 3 || nb == 4) ? nb : 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 3: the decode step's GEMMs re-built around ONE rule -- every global load of a launch is issued before the first wait.  A launch of
// these kernels is a handful of work-groups on an idle chip; its duration is the ~4.3 us every launch costs on this stack plus one L2 round
// trip (0.2-0.5 us at the clocks such a load runs at) per DEPENDENT load.  The round-2 kernel above spends 17-25 round trips per launch:
// hipcc branches around every conditional load (`ok ? p[i] : 0`, `bias ? bias[co] : 0`, the per-element LayerNorm gamma / beta) and waits
// at each join, and the LayerNorm statistics loops wait for each chunk in turn (rocprofv3: 8.7 us with the LayerNorm prologue, 5.3 without).
// Here: loads go to clamped / substitute addresses unconditionally and the VALUE is selected; the x rows are staged in LDS once per
// work-group (LayerNorm / activation applied there by one wave per row, not re-done by every wave); the first batch of weight fragments, the
// bias and the residual are requested at kernel entry, ahead of the staging.
//
// linear_rows_ksplit_kernel: the K chunks of one 16-channel output group are dealt to the FOUR waves of a work-group (the M -> C projection
// of the MLP, K = 1024, was eight dependent batches on 16 waves of the whole chip); the four partial accumulators meet in LDS in a fixed order.
// STAGE selects where the x rows come from.  0: global memory.  1 / 2: assembled in LDS from the partial results of the producing launch,
// which takes that producer's merge launch off the token's dependent chain:
//   1  x = merge of the split-KV single-query attention partials (kv_merge_one; out-projection of the decode step);
//   2  x = x1 + b2 + sum_j P[j]  -- the residual stream after the MLP whose down-projection left NJ K-slice partials (mlp_rows_kernel);
//      work-group 0 also stores the assembled rows (the out-projection two launches later reads them as its residual).
// The assembled rows are rounded to T, exactly what the un-fused chain stores and re-reads.  XF: 0 = rows used as they are (STAGE 0: straight
// from global memory, no LDS), 1 = LayerNorm and / or pre-activation applied to the staged rows.
// ---------------------------------------------------------------------------------------------------------------------------------
#define GM_ROWS_KCH 4  // 16-byte vectors per lane and row in the staging pass: cin <= 64 * VECW * GM_ROWS_KCH (2048 bf16 / 1024 fp32)

// rows [rows][cin] -> xs (LDS, as T), one wave per row: optional LayerNorm (two-pass statistics from registers, like the reference's fp32
// computation) and optional pre-activation.  SRC_LDS: the rows already sit in xs (assembled there), otherwise they come from global memory.
template <typename T, bool SRC_LDS>
__device__ __forceinline__ void stage_rows(const T* __restrict__ x, long long x_ld, T* xs, int rows, int cin, const float* __restrict__ ln_g,
                                           const float* __restrict__ ln_b, float ln_eps, int pre_act) {
  constexpr int VECW = ConvTraits<T>::VECW;
  constexpr bool PRECISE = sizeof(T) == 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* gsrc = ln_g ? ln_g : reinterpret_cast<const float*>(x ? (const void*)x : (const void*)xs);  // never dereferenced without ln_g
  const float* bsrc = ln_b ? ln_b : gsrc;
  const float bmul = ln_b ? 1.f : 0.f;
  for (int r = wave; r < rows; r += 4) {
    float v[GM_ROWS_KCH][VECW], g[GM_ROWS_KCH][VECW], bb[GM_ROWS_KCH][VECW];
    bool ok[GM_ROWS_KCH];
#pragma unroll
    for (int k = 0; k < GM_ROWS_KCH; ++k) {
      const int ch = (k * 64 + lane) * VECW;
      ok[k] = ch + VECW <= cin;
      const int c = ok[k] ? ch : 0;
      const uint4 raw = SRC_LDS ? *reinterpret_cast<const uint4*>(xs + (long long)r * cin + c) : *reinterpret_cast<const uint4*>(x + (long long)r * x_ld + c);
      Vec16<T>::unpack(raw, v[k]);
      if (ln_g) {  // (uniform; the loads inside are unconditional per lane)
#pragma unroll
        for (int i = 0; i < VECW; i += 4) {
          const float4 tg = *reinterpret_cast<const float4*>(gsrc + c + i), tb = *reinterpret_cast<const float4*>(bsrc + c + i);
          g[k][i] = tg.x; g[k][i + 1] = tg.y; g[k][i + 2] = tg.z; g[k][i + 3] = tg.w;
          bb[k][i] = tb.x; bb[k][i + 1] = tb.y; bb[k][i + 2] = tb.z; bb[k][i + 3] = tb.w;
        }
      }
    }
    if (ln_g) {
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < GM_ROWS_KCH; ++k)
#pragma unroll
        for (int i = 0; i < VECW; ++i) sum += ok[k] ? v[k][i] : 0.f;
      const float mean = wave_sum(sum) / (float)cin;
      float sq = 0.f;
#pragma unroll
      for (int k = 0; k < GM_ROWS_KCH; ++k)
#pragma unroll
        for (int i = 0; i < VECW; ++i) sq += ok[k] ? (v[k][i] - mean) * (v[k][i] - mean) : 0.f;
      const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)cin + ln_eps);
#pragma unroll
      for (int k = 0; k < GM_ROWS_KCH; ++k)
#pragma unroll
        for (int i = 0; i < VECW; ++i) v[k][i] = (v[k][i] - mean) * rstd * g[k][i] + bb[k][i] * bmul;
    }
#pragma unroll
    for (int k = 0; k < GM_ROWS_KCH; ++k) {
      if (pre_act) conv_act_vec(v[k], pre_act, PRECISE);
      if (ok[k]) *reinterpret_cast<uint4*>(xs + (long long)r * cin + (k * 64 + lane) * VECW) = Vec16<T>::pack(v[k]);
    }
  }
}

template <typename T, int STAGE, int XF>
__global__ __launch_bounds__(256) void linear_rows_ksplit_kernel(const T* __restrict__ x, long long x_ld, const T* __restrict__ w,
                                                                const float* __restrict__ bias, const T* __restrict__ res, long long res_ld,
                                                                T* __restrict__ y, long long y_ld, int rows, int cin, int cout, int pre_act,
                                                                int post_act, LinearRowsExtra ex) {
  constexpr int BK = ConvTraits<T>::BK, VECW = ConvTraits<T>::VECW;
  constexpr bool STAGED = STAGE != 0 || XF != 0;
  __shared__ float part[3][64][4];
  extern __shared__ __attribute__((aligned(16))) char staged_raw[];
  T* xs = reinterpret_cast<T*>(staged_raw);  // STAGED: [rows][cin]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int cout_pad = (cout + 15) & ~15;
  const int co0 = blockIdx.x * 16;
  const int row = l15;  // one row block (host: rows <= 16)
  const bool row_ok = row < rows;
  const int nchunks = (cin + BK - 1) / BK;
  const T* wrow = w + ((long long)(co0 + l15)) * BK + q * VECW;
  constexpr int U = 4;
  uint4 wf[U], xf[U];
  // ---- requested at entry: this wave's first batch of weight fragments (chunks wave, wave + 4, ...), bias and residual of its outputs ----------
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int cc = wave + 4 * u;
    wf[u] = *reinterpret_cast<const uint4*>(wrow + (long long)(cc < nchunks ? cc : nchunks - 1) * cout_pad * BK);
  }
  float bia[4], rsd[4];
  {
    const float* bsrc = bias ? bias : reinterpret_cast<const float*>(w);
    const T* rsrc = res ? res + (long long)(row_ok ? row : 0) * res_ld : w;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = co0 + 4 * q + i;
      bia[i] = bsrc[bias ? (co < cout ? co : cout - 1) : 0];
      rsd[i] = ElemIO<T>::ld(rsrc + (res ? (co < cout ? co : cout - 1) : 0));
    }
  }
  // this lane's 16-byte vector of chunk c (zero beyond the row / the last channel)
  auto xvec = [&](int c) __attribute__((always_inline)) -> uint4 {
    const bool ok = row_ok & (c * BK + q * VECW + VECW <= cin);
    const uint4 v = STAGED ? *reinterpret_cast<const uint4*>(xs + (ok ? (long long)row * cin + c * BK + q * VECW : 0))
                           : *reinterpret_cast<const uint4*>(ok ? x + (long long)row * x_ld + q * VECW + c * BK : x);
    return make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
  };
  if (!STAGED) {
#pragma unroll
    for (int u = 0; u < U; ++u) xf[u] = xvec(wave + 4 * u < nchunks ? wave + 4 * u : nchunks - 1);
  }
  // ---- the x rows into LDS ---------------------------------------------------------------------------------------------------------------
  if (STAGE == 1) {
    const int H = cin / ex.kv_dh;
    for (int idx = threadIdx.x; idx < rows * cin; idx += 256) {
      const int r = idx / cin, ch = idx - r * cin;
      ElemIO<T>::st(xs + idx, kv_merge_one(ex.kv_ws, rows * H, ex.kv_dh, r * H + ch / ex.kv_dh, ch % ex.kv_dh));
    }
  } else if (STAGE == 2) {
    const float* b2 = ex.mlp_b2 ? ex.mlp_b2 : ex.mlp_p;
    const float b2mul = ex.mlp_b2 ? 1.f : 0.f;
    for (int idx = threadIdx.x; idx < rows * cin; idx += 256) {
      const int ch = idx % cin;
      constexpr int PB = 8;  // partials in flight per batch
      float v = ElemIO<T>::ld(reinterpret_cast<const T*>(ex.mlp_x1) + idx) + b2[ex.mlp_b2 ? ch : 0] * b2mul;
      const float* pp = ex.mlp_p + idx;
      const long long pstride = (long long)rows * cin;
      for (int j0 = 0; j0 < ex.mlp_nj; j0 += PB) {
        float t[PB];
#pragma unroll
        for (int j = 0; j < PB; ++j) t[j] = pp[(j0 + j < ex.mlp_nj ? j0 + j : j0) * pstride];
#pragma unroll
        for (int j = 0; j < PB; ++j) v += j0 + j < ex.mlp_nj ? t[j] : 0.f;  // slice order
      }
      ElemIO<T>::st(xs + idx, v);
      if (blockIdx.x == 0 && ex.mlp_x0) reinterpret_cast<T*>(ex.mlp_x0)[idx] = xs[idx];
    }
  }
  if (STAGE != 0 && XF != 0) __syncthreads();
  if (XF != 0) {
    if (STAGE == 0) stage_rows<T, false>(x, x_ld, xs, rows, cin, ex.ln_g, ex.ln_b, ex.ln_eps, pre_act);
    else stage_rows<T, true>(nullptr, 0, xs, rows, cin, ex.ln_g, ex.ln_b, ex.ln_eps, pre_act);
  }
  if (STAGED) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) xf[u] = xvec(wave + 4 * u < nchunks ? wave + 4 * u : nchunks - 1);
  }
  // ---- GEMM over this wave's chunks ---------------------------------------------------------------------------------------------------------
  f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int c0 = wave; c0 < nchunks; c0 += 4 * U) {
    if (c0 != wave) {  // (the first batch was requested at entry)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + 4 * u < nchunks ? c0 + 4 * u : nchunks - 1;
        wf[u] = *reinterpret_cast<const uint4*>(wrow + (long long)c * cout_pad * BK);
        xf[u] = xvec(c);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (c0 + 4 * u < nchunks) Mma<T>::run(wf[u], xf[u], acc);
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) part[wave - 1][lane][i] = acc[i];
  }
  __syncthreads();
  if (wave > 0 || !row_ok) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = ((acc[i] + part[0][lane][i]) + part[1][lane][i]) + part[2][lane][i];  // fixed order: waves 0, 1, 2, 3
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int co = co0 + 4 * q + i;
    if (co < cout) {
      float v = acc[i] + (bias ? bia[i] : 0.f);
      v = conv_post_act(v, post_act);
      if (res) v += rsd[i];
      if (ex.split > 0 && co >= ex.split) {
        T* dst = reinterpret_cast<T*>(co < 2 * ex.split ? ex.y1 : ex.y2) + (ex.off_dev ? (long long)(*ex.off_dev) * ex.off_mul : 0);
        ElemIO<T>::st(dst + (long long)row * ex.y12_ld + (co - (co < 2 * ex.split ? ex.split : 2 * ex.split)), v);
      } else {
        ElemIO<T>::st(y + (long long)row * y_ld + co, v);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The MLP of a decode step in ONE launch (round 3): LayerNorm -> up-projection [C -> M] -> GELU -> down-projection [M -> C].  Work-group j
// owns the hidden slice [64 j, 64 j + 64): the LayerNorm'ed rows are staged in LDS once (stage_rows), its four waves each compute 16 hidden
// channels over all of K = C, the activated slice goes to LDS as T, and the work-group multiplies it with ITS 64-row K-slice of the
// down-projection: a [rows][C] fp32 partial P[j] -- no bias, no residual.  The M / 64 partials are summed (slice order) with the residual row
// and the bias by the consumer's prologue (linear_rows_ksplit_kernel STAGE 2).  Both projections' weight fragments are requested at kernel
// entry (C <= 256 bf16: all of them), so the launch has one exposed round trip.  One launch instead of two on the token's dependent chain,
// and the M -> C product runs on M / 64 work-groups.
// ---------------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mlp_rows_kernel(const T* __restrict__ x, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                      float ln_eps, const T* __restrict__ w1, const float* __restrict__ b1,
                                                      const T* __restrict__ w2, float* __restrict__ P, int rows, int C, int M, int act) {
  constexpr int BK = ConvTraits<T>::BK, VECW = ConvTraits<T>::VECW;
  constexpr int HC = 64 / BK;      // K chunks of the down-projection per hidden slice
  constexpr int GMAX = 8;          // output groups of the down-projection per wave held in registers: C <= 4 * 16 * GMAX = 512
  constexpr int U = 8;             // up-projection chunks in flight
  __shared__ __attribute__((aligned(16))) T hs[16][64 + VECW];  // activated hidden slice, [row][hidden]; + VECW: rows on different banks
  extern __shared__ __attribute__((aligned(16))) char staged_raw[];
  T* xs = reinterpret_cast<T*>(staged_raw);  // [rows][C] LayerNorm'ed rows
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int j = blockIdx.x;
  const int row = l15;
  const bool row_ok = row < rows;
  const int nchunks = (C + BK - 1) / BK;
  const int cpad = (C + 15) & ~15, mpad = (M + 15) & ~15;
  const int ngroups = cpad / 16;
  const int hc0 = j * 64 + wave * 16;  // this wave's hidden channels
  const T* w1row = w1 + ((long long)(hc0 + l15)) * BK + q * VECW;
  // ---- requested at entry: the up-projection's first U chunks, the bias of this lane's hidden channels, the down-projection fragments of this
  //      wave's output groups (wave, wave + 4, ...) for the slice's HC chunks -------------------------------------------------------------------
  uint4 wf[U];
#pragma unroll
  for (int u = 0; u < U; ++u) wf[u] = *reinterpret_cast<const uint4*>(w1row + (long long)(u < nchunks ? u : nchunks - 1) * mpad * BK);
  float b1v[4];
  {
    const float* bsrc = b1 ? b1 : reinterpret_cast<const float*>(w1);
#pragma unroll
    for (int i = 0; i < 4; ++i) b1v[i] = bsrc[b1 ? hc0 + 4 * q + i : 0];
  }
  uint4 w2f[GMAX][HC];
#pragma unroll
  for (int g = 0; g < GMAX; ++g) {
    const int grp = wave + 4 * g < ngroups ? wave + 4 * g : ngroups - 1;
#pragma unroll
    for (int c = 0; c < HC; ++c) w2f[g][c] = *reinterpret_cast<const uint4*>(w2 + ((long long)(j * HC + c) * cpad + grp * 16 + l15) * BK + q * VECW);
  }
  // ---- LayerNorm'ed rows into LDS ------------------------------------------------------------------------------------------------------------
  stage_rows<T, false>(x, C, xs, rows, C, ln_g, ln_b, ln_eps, 0);
  __syncthreads();
  auto xvec = [&](int c) __attribute__((always_inline)) -> uint4 {
    const bool ok = row_ok & (c * BK + q * VECW + VECW <= C);
    const uint4 v = *reinterpret_cast<const uint4*>(xs + (ok ? (long long)row * C + c * BK + q * VECW : 0));
    return make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
  };
  // ---- phase 1: up-projection of hidden channels 64 j + 16 wave .. + 16, GELU ---------------------------------------------------------------
  f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < nchunks; c0 += U) {
    if (c0 != 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) wf[u] = *reinterpret_cast<const uint4*>(w1row + (long long)(c0 + u < nchunks ? c0 + u : nchunks - 1) * mpad * BK);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (c0 + u < nchunks) Mma<T>::run(wf[u], xvec(c0 + u), acc);
  }
  // D layout: column = row l15, rows = hidden channels hc0 + 4 q + i  ->  hs[row][16 wave + 4 q + i] (rows beyond `rows` hold zeros)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float v = acc[i] + (b1 ? b1v[i] : 0.f);
    v = conv_post_act(v, act);
    ElemIO<T>::st(&hs[l15][wave * 16 + 4 * q + i], row_ok ? v : 0.f);
  }
  __syncthreads();
  // ---- phase 2: P[j][row][co] = sum over the slice's 64 hidden channels -------------------------------------------------------------------
  uint4 hf[HC];
#pragma unroll
  for (int c = 0; c < HC; ++c) hf[c] = *reinterpret_cast<const uint4*>(&hs[l15][c * BK + q * VECW]);
  float* Pj = P + (long long)j * rows * C;
#pragma unroll
  for (int g = 0; g < GMAX; ++g) {
    const int grp = wave + 4 * g;
    if (grp < ngroups) {
      f32x4_t a2 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < HC; ++c) Mma<T>::run(w2f[g][c], hf[c], a2);
      if (row_ok) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int co = grp * 16 + 4 * q + i;
          if (co < C) Pj[(long long)row * C + co] = a2[i];
        }
      }
    }
  }
}

// one row block and a long K: the K chunks of an output group are dealt to the four waves of a work-group
static bool linear_rows_takes_ksplit(int rows, int cin, int dtype) {
  static const bool ksplit = !(getenv("GM_LINEAR_KSPLIT") && getenv("GM_LINEAR_KSPLIT")[0] == '0');  // bench switch (tools/diag_c5.py)
  const int bk = dtype == GM_F32 ? 16 : 32, vecw = dtype == GM_F32 ? 4 : 8;
  // (its staging pass holds GM_ROWS_KCH vectors per lane and row, and the staged rows live in <= 48 KiB of LDS)
  return ksplit && (cin + bk - 1) / bk >= 8 && rows <= 16 && cin <= 64 * vecw * GM_ROWS_KCH && (long long)rows * cin * (dtype == GM_F32 ? 4 : 2) <= 48 * 1024;
}

static int linear_rows_launch(const void* x, long long x_ld, const void* w, const float* bias, const void* res, long long res_ld, void* y,
                              long long y_ld, int rows, int cin, int cout, int pre_act, int post_act, int dtype, const LinearRowsExtra& ex,
                              void* stream) {
  GM_REQUIRE((x || ex.kv_ws || ex.mlp_p) && w && y, "null pointer");
  GM_REQUIRE(rows >= 0 && cin > 0 && cout > 0, "bad geometry");
  if (rows == 0) return 0;
  const int vecw = dtype == GM_F32 ? 4 : 8;
  GM_REQUIRE(cin % vecw == 0 && x_ld % vecw == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "x rows must be 16-byte vectors");
  GM_REQUIRE(!(ex.kv_ws || ex.mlp_p) || linear_rows_takes_ksplit(rows, cin, dtype), "the partial-merging prologues live in the K-split kernel");
  GM_REQUIRE(ex.split == 0 || (ex.y1 && ex.y2 && cout == 3 * ex.split), "split output needs two extra destinations and cout = 3 * split");
  hipStream_t st = (hipStream_t)stream;
  const int cout_pad = (cout + 15) & ~15;
  if (linear_rows_takes_ksplit(rows, cin, dtype) && !ex.pre_scale) {
    dim3 gk(cout_pad / 16, 1);
    const int stage = ex.kv_ws ? 1 : ex.mlp_p ? 2 : 0;
    const int xf = (ex.ln_g || pre_act) ? 1 : 0;
    const size_t smem = (stage || xf) ? (size_t)rows * cin * (dtype == GM_F32 ? 4 : 2) : 0;  // the staged rows
#define GM_KSPLIT_LAUNCH(T, STAGE, XF)                                                                                                         \
  linear_rows_ksplit_kernel<T, STAGE, XF><<<gk, 256, smem, st>>>((const T*)x, x_ld, (const T*)w, bias, (const T*)res, res_ld, (T*)y, y_ld, rows, \
                                                                 cin, cout, pre_act, post_act, ex)
#define GM_KSPLIT_DISPATCH(T)                                                         \
  do {                                                                                \
    if (stage == 0) { if (xf) GM_KSPLIT_LAUNCH(T, 0, 1); else GM_KSPLIT_LAUNCH(T, 0, 0); } \
    else if (stage == 1) { if (xf) GM_KSPLIT_LAUNCH(T, 1, 1); else GM_KSPLIT_LAUNCH(T, 1, 0); } \
    else { if (xf) GM_KSPLIT_LAUNCH(T, 2, 1); else GM_KSPLIT_LAUNCH(T, 2, 0); }       \
  } while (0)
    if (dtype == GM_F32) GM_KSPLIT_DISPATCH(float);
    else if (dtype == GM_BF16) GM_KSPLIT_DISPATCH(bf16_raw);
    else GM_FAIL(-2, "unsupported dtype");
#undef GM_KSPLIT_DISPATCH
#undef GM_KSPLIT_LAUNCH
    GM_LAUNCH_CHECK();
  }
  const int bk = dtype == GM_F32 ? 16 : 32;
  const bool gn = ex.gn_stats[0] != nullptr;  // (gm_linear_rows_gn: the wide form whatever the row count)
  if ((gn || (gm_token_gemm_wide_rows > 0 && rows >= gm_token_gemm_wide_rows)) && cin % bk == 0 && cout_pad >= 32 && !ex.ln_g && !ex.kv_ws && !ex.mlp_p && ex.split == 0 &&
      !ex.off_dev && (!ex.pre_scale || ex.rows_per_sample > 0) && (dtype == GM_F32 || dtype == GM_BF16)) {
    const int nb = gm_token_gemm_wide_nb ? gm_token_gemm_wide_nb : (rows >= 8192 ? 4 : 2);
    dim3 gw((cout_pad + 16 * nb - 1) / (16 * nb), (rows + 63) / 64);
    const size_t gsm = gn ? (size_t)cin * (2 * 4 + 16) : 0;
#define GM_WIDE_LAUNCH(T, NBV)                                                                                                                                    \
  do {                                                                                                                                                            \
    if (gn) token_gemm_wide_kernel<T, NBV, true><<<gw, 256, gsm, st>>>((const T*)x, x_ld, (const T*)w, bias, (const T*)res, res_ld, (T*)y, y_ld, rows, cin, cout, \
                                                                       pre_act, post_act, ex);                                                                    \
    else token_gemm_wide_kernel<T, NBV, false><<<gw, 256, 0, st>>>((const T*)x, x_ld, (const T*)w, bias, (const T*)res, res_ld, (T*)y, y_ld, rows, cin, cout,     \
                                                                   pre_act, post_act, ex);                                                                        \
  } while (0)
    if (dtype == GM_F32) { if (nb == 2) GM_WIDE_LAUNCH(float, 2); else if (nb == 3) GM_WIDE_LAUNCH(float, 3); else GM_WIDE_LAUNCH(float, 4); }
    else { if (nb == 2) GM_WIDE_LAUNCH(bf16_raw, 2); else if (nb == 3) GM_WIDE_LAUNCH(bf16_raw, 3); else GM_WIDE_LAUNCH(bf16_raw, 4); }
#undef GM_WIDE_LAUNCH
    GM_LAUNCH_CHECK();
  }
  GM_REQUIRE(!gn, "the statistics form of the token GEMM's GroupNorm needs the wide kernel's geometry (cin a multiple of the MFMA K step, cout >= 32)");
  dim3 grid((cout_pad / 16 + 3) / 4, (rows + 15) / 16);
  if (dtype == GM_F32)
    linear_rows_kernel<float><<<grid, 256, 0, st>>>((const float*)x, x_ld, (const float*)w, bias, (const float*)res, res_ld, (float*)y, y_ld,
                                                    rows, cin, cout, pre_act, post_act, ex);
  else if (dtype == GM_BF16)
    linear_rows_kernel<bf16_raw><<<grid, 256, 0, st>>>((const bf16_raw*)x, x_ld, (const bf16_raw*)w, bias, (const bf16_raw*)res, res_ld,
                                                       (bf16_raw*)y, y_ld, rows, cin, cout, pre_act, post_act, ex);
  else
    GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

extern "C" int gm_linear_rows(const void* x, long long x_ld, const void* w, const float* bias, const void* res, long long res_ld, void* y,
                              long long y_ld, int rows, int cin, int cout, int pre_act, int post_act, int dtype, void* stream) {
  LinearRowsExtra ex = {};
  return linear_rows_launch(x, x_ld, w, bias, res, res_ld, y, y_ld, rows, cin, cout, pre_act, post_act, dtype, ex, stream);
}

// y = post_act(act(x * scale[n] + shift[n]) W^T + b) (+ res) over token rows: the 1x1 convolutions of the latent-resolution attention blocks
// (GroupNorm prologue + stacked q | k | v projection on a few thousand tokens), for which the tiled convolution kernels are a serial chain of
// stage -> barrier -> tap -> barrier steps per K chunk (20 us per launch whatever the size); here a wave owns 16 rows x 16 output channels and
// requests all of its K chunks (4 per wait) straight from L2.  scale / shift: fp32 [N][ss_ld] (nullable), n = row / rows_per_sample.
extern "C" int gm_linear_rows_affine(const void* x, long long x_ld, const float* pre_scale, const float* pre_shift, long long ss_ld, int rows_per_sample,
                                     const void* w, const float* bias, const void* res, long long res_ld, void* y, long long y_ld, int rows, int cin,
                                     int cout, int pre_act, int post_act, int dtype, void* stream) {
  GM_REQUIRE((pre_scale == nullptr) == (pre_shift == nullptr), "scale and shift come together");
  GM_REQUIRE(!pre_scale || (rows_per_sample > 0 && ss_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(pre_scale) & 15) == 0 &&
                            (reinterpret_cast<uintptr_t>(pre_shift) & 15) == 0), "the affine tables are 16-byte aligned fp32 rows");
  LinearRowsExtra ex = {};
  ex.pre_scale = pre_scale; ex.pre_shift = pre_shift; ex.ss_ld = ss_ld; ex.rows_per_sample = rows_per_sample;
  return linear_rows_launch(x, x_ld, w, bias, res, res_ld, y, y_ld, rows, cin, cout, pre_act, post_act, dtype, ex, stream);
}

// the same GEMM as the stacked q | k | v projection of an attention block whose V columns (output channels >= vt_c0, heads of vt_dh channels) are
// also written as the transposed key-permuted image of the LDS-DMA attention kernel (GmAttnDesc.vt_packed = 1 then skips its pack launch).
// rows_per_sample = tokens per sample, a multiple of 64 (the image has no padding keys); bf16.
extern "C" int gm_linear_rows_affine_vt(const void* x, long long x_ld, const float* pre_scale, const float* pre_shift, long long ss_ld, int rows_per_sample,
                                        const void* w, const float* bias, void* y, long long y_ld, int rows, int cin, int cout, int pre_act, void* vt,
                                        int vt_c0, int vt_dh, int dtype, void* stream) {
  GM_REQUIRE((pre_scale == nullptr) == (pre_shift == nullptr), "scale and shift come together");
  GM_REQUIRE(!pre_scale || (ss_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(pre_scale) & 15) == 0 && (reinterpret_cast<uintptr_t>(pre_shift) & 15) == 0),
             "the affine tables are 16-byte aligned fp32 rows");
  GM_REQUIRE(vt && dtype == GM_BF16 && rows_per_sample > 0 && rows_per_sample % 64 == 0 && rows % rows_per_sample == 0, "the V^T image needs bf16 and whole 64-key blocks per sample");
  GM_REQUIRE(vt_dh > 0 && vt_c0 >= 0 && vt_c0 < cout && (cout - vt_c0) % vt_dh == 0, "the V columns are whole heads");
  LinearRowsExtra ex = {};
  ex.pre_scale = pre_scale; ex.pre_shift = pre_shift; ex.ss_ld = ss_ld; ex.rows_per_sample = rows_per_sample;
  ex.vt = vt; ex.vt_c0 = vt_c0; ex.vt_dh = vt_dh;
  return linear_rows_launch(x, x_ld, w, bias, nullptr, 0, y, y_ld, rows, cin, cout, pre_act, 0, dtype, ex, stream);
}

// gm_linear_rows_affine(_vt) with the GroupNorm given as the statistic tables of x's producer(s): finalised in the GEMM's prologue, one launch less per attention block
extern "C" int gm_linear_rows_gn(const void* x, long long x_ld, const GmGnTables* gn, int n_samples, int rows_per_sample, const void* w, const float* bias,
                                 const void* res, long long res_ld, void* y, long long y_ld, int rows, int cin, int cout, int pre_act, int post_act, void* vt,
                                 int vt_c0, int vt_dh, int dtype, void* stream) {
  GM_REQUIRE(gn && gn->stats[0] && gn->groups > 0 && cin % gn->groups == 0 && cin <= 384, "GroupNorm tables: whole groups, at most 384 channels");
  GM_REQUIRE(gn->S[0] >= 1 && gn->S[0] <= GN_SHORT_MAX_ROWS && gn->C[0] > 0 &&
             ((gn->stats[1] == nullptr && gn->C[0] == cin) || (gn->stats[1] != nullptr && gn->S[1] >= 1 && gn->S[1] <= GN_SHORT_MAX_ROWS && gn->C[1] > 0 && gn->C[0] + gn->C[1] == cin)),
             "short statistic tables covering the input channels");
  GM_REQUIRE((reinterpret_cast<uintptr_t>(gn->stats[0]) & 15) == 0 && (reinterpret_cast<uintptr_t>(gn->stats[1]) & 15) == 0, "16-byte aligned tables");
  GM_REQUIRE(n_samples > 0 && rows_per_sample > 0 && rows_per_sample % 64 == 0 && rows == n_samples * rows_per_sample, "whole 64-row groups per sample");
  GM_REQUIRE(!vt || (dtype == GM_BF16 && vt_dh > 0 && vt_c0 >= 0 && vt_c0 < cout && (cout - vt_c0) % vt_dh == 0 && !res && post_act == 0), "the V^T image: bf16, whole heads, no residual");
  LinearRowsExtra ex = {};
  ex.rows_per_sample = rows_per_sample;
  ex.gn_stats[0] = gn->stats[0]; ex.gn_stats[1] = gn->stats[1]; ex.gn_S[0] = gn->S[0]; ex.gn_S[1] = gn->stats[1] ? gn->S[1] : 0;
  ex.gn_C[0] = gn->C[0]; ex.gn_C[1] = gn->stats[1] ? gn->C[1] : 0;
  ex.gn_gamma = gn->gamma; ex.gn_beta = gn->beta; ex.gn_eps = gn->eps; ex.gn_groups = gn->groups; ex.gn_N = n_samples;
  ex.vt = vt; ex.vt_c0 = vt_c0; ex.vt_dh = vt_dh;
  return linear_rows_launch(x, x_ld, w, bias, res, res_ld, y, y_ld, rows, cin, cout, pre_act, post_act, dtype, ex, stream);
}

// the decode step's fused forms: LayerNorm prologue; q | k | v projection writing k, v rows into the caches (internal to the library)
extern "C" int gm_linear_rows_ln(const void* x, long long x_ld, const float* ln_g, const float* ln_b, float ln_eps, const void* w,
                                 const float* bias, void* y, long long y_ld, void* y1, void* y2, long long y12_ld, int split, int rows, int cin,
                                 int cout, int post_act, int dtype, const int* off_dev, long long off_mul, void* stream) {
  LinearRowsExtra ex = {};
  ex.ln_g = ln_g; ex.ln_b = ln_b; ex.ln_eps = ln_eps;
  ex.y1 = y1; ex.y2 = y2; ex.y12_ld = y12_ld; ex.split = split;
  ex.off_dev = off_dev; ex.off_mul = off_mul;
  return linear_rows_launch(x, x_ld, w, bias, nullptr, 0, y, y_ld, rows, cin, cout, 0, post_act, dtype, ex, stream);
}

// out_proj of the decode step reading the split-KV attention partials directly (the merge is this GEMM's prologue: one launch fewer per
// block).  1 = launched, 0 = not this kernel's case (the caller merges with attn_decode_combine_kernel and calls gm_linear_rows), < 0 = error.
extern "C" int gm_linear_rows_kvmerge(const float* kv_ws, int kv_ns, int kv_dh, const void* w, const float* bias, const void* res, long long res_ld,
                                      void* y, long long y_ld, int rows, int cin, int cout, int dtype, void* stream) {
  if (!kv_ws || kv_ns != GM_DECODE_KV_SPLITS || kv_dh <= 0 || cin % kv_dh || !linear_rows_takes_ksplit(rows, cin, dtype)) return 0;
  LinearRowsExtra ex = {};
  ex.kv_ws = kv_ws; ex.kv_dh = kv_dh;
  const int rc = linear_rows_launch(nullptr, 0, w, bias, res, res_ld, y, y_ld, rows, cin, cout, 0, 0, dtype, ex, stream);
  return rc ? (rc > 0 ? -rc : rc) : 1;
}

// The decode step's MLP as one launch leaving K-slice partials, and the consumer form of the small-row GEMM that sums them (see mlp_rows_kernel).
// gm_mlp_rows_fusable: 1 when both kernels take this geometry.  P holds (M / 64) * rows * C floats.
extern "C" int gm_mlp_rows_fusable(int rows, int C, int M, int dtype) {
  static const bool on = !(getenv("GM_DECODE_MLP_FUSE") && getenv("GM_DECODE_MLP_FUSE")[0] == '0');  // bench switch (tools/diag_c5.py)
  const int vecw = dtype == GM_F32 ? 4 : 8;
  return on && (dtype == GM_F32 || dtype == GM_BF16) && rows >= 1 && rows <= 16 && C % vecw == 0 && C <= 512 && M % 64 == 0 &&
         linear_rows_takes_ksplit(rows, C, dtype);
}
extern "C" int gm_mlp_rows(const void* x, const float* ln_g, const float* ln_b, float ln_eps, const void* w1, const float* b1, const void* w2,
                           float* P, int rows, int C, int M, int act, int dtype, void* stream) {
  GM_REQUIRE(x && w1 && w2 && P && ln_g, "null pointer");
  GM_REQUIRE(gm_mlp_rows_fusable(rows, C, M, dtype), "geometry outside the fused MLP kernel");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GM_F32)
    mlp_rows_kernel<float><<<M / 64, 256, (size_t)rows * C * 4, st>>>((const float*)x, ln_g, ln_b, ln_eps, (const float*)w1, b1, (const float*)w2, P, rows, C, M, act);
  else
    mlp_rows_kernel<bf16_raw><<<M / 64, 256, (size_t)rows * C * 2, st>>>((const bf16_raw*)x, ln_g, ln_b, ln_eps, (const bf16_raw*)w1, b1, (const bf16_raw*)w2, P, rows, C, M, act);
  GM_LAUNCH_CHECK();
}
// y = act(LN(x0) W^T + b) with x0 = x1 + b2 + sum_j P[j] assembled in the prologue (and stored to `x0_out` when given); split outputs as gm_linear_rows_ln
extern "C" int gm_linear_rows_mlpmerge(const float* P, int nj, const void* x1, const float* b2, void* x0_out, const float* ln_g, const float* ln_b,
                                       float ln_eps, const void* w, const float* bias, void* y, long long y_ld, void* y1, void* y2, long long y12_ld,
                                       int split, int rows, int cin, int cout, int post_act, int dtype, const int* off_dev, long long off_mul,
                                       void* stream) {
  GM_REQUIRE(P && x1 && nj > 0, "null pointer");
  LinearRowsExtra ex = {};
  ex.ln_g = ln_g; ex.ln_b = ln_b; ex.ln_eps = ln_eps;
  ex.y1 = y1; ex.y2 = y2; ex.y12_ld = y12_ld; ex.split = split;
  ex.off_dev = off_dev; ex.off_mul = off_mul;
  ex.mlp_p = P; ex.mlp_nj = nj; ex.mlp_x1 = x1; ex.mlp_b2 = b2; ex.mlp_x0 = x0_out;
  return linear_rows_launch(nullptr, 0, w, bias, nullptr, 0, y, y_ld, rows, cin, cout, 0, post_act, dtype, ex, stream);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// one query per (batch, head)
// ---------------------------------------------------------------------------------------------------------------------------------
#define DEC_MAX_KEYS 15360  // scores live in LDS (fp32): 60 KiB + query + scratch stay under the default 64 KiB dynamic limit

template <typename T>
__global__ __launch_bounds__(256) void attn_decode_kernel(GmAttnDesc p, const int* __restrict__ lk_dev) {
  const int lds_keys = p.Lk;       // the launch sized the score buffer for this many keys
  if (lk_dev) p.Lk = *lk_dev + 1;  // decode-graph replay: keys 0 .. position (LDS was sized for the descriptor's Lk = the cache length)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sc = reinterpret_cast<float*>(smem);            // [Lk] scores, then probabilities
  float* qs = sc + ((((lk_dev ? lds_keys : p.Lk) > 2048 ? (lk_dev ? lds_keys : p.Lk) : 2048) + 3) & ~3);  // [dh] query (fp32); the score region doubles as the [KL][dh] partial table
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
  // ---- output: thread <-> (key lane, 16-byte channel vector): KL keys per iteration, 4 iterations of independent loads in flight
  //      (one 2-byte load per thread per key, accumulated serially, cost ~1 us of latency per key: 125 us per call at 3500 keys) -----
  const bool vvec = (dh % VECW == 0) && (p.v_ld % VECW == 0) && ((reinterpret_cast<uintptr_t>(V) & 15) == 0);
  float o[VECW];
#pragma unroll
  for (int i = 0; i < VECW; ++i) o[i] = 0.f;
  const int nv = vvec ? dh / VECW : dh;             // channel vectors (or single channels) per key
  const int KL = 256 / nv > 0 ? 256 / nv : 1;       // key lanes
  const int cv = tid % nv, kl = tid / nv;
  if (kl < KL) {
    if (vvec) {
      int j = kl;
      for (; j + 3 * KL < p.Lk; j += 4 * KL) {
        uint4 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const uint4*>(V + (long long)(j + u * KL) * p.v_ld + cv * VECW);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float vv[VECW];
          Vec16<T>::unpack(r[u], vv);
          const float pj = sc[j + u * KL];
#pragma unroll
          for (int i = 0; i < VECW; ++i) o[i] += pj * vv[i];
        }
      }
      for (; j < p.Lk; j += KL) {
        float vv[VECW];
        Vec16<T>::unpack(*reinterpret_cast<const uint4*>(V + (long long)j * p.v_ld + cv * VECW), vv);
        const float pj = sc[j];
#pragma unroll
        for (int i = 0; i < VECW; ++i) o[i] += pj * vv[i];
      }
    } else {
      for (int j = kl; j < p.Lk; j += KL) o[0] += sc[j] * ElemIO<T>::ld(V + (long long)j * p.v_ld + cv);
    }
  }
  __syncthreads();  // everyone is done reading the probabilities: the score buffer becomes the [KL][dh] partial-sum table
  const int per = vvec ? VECW : 1;
  if (kl < KL)
    for (int i = 0; i < per; ++i) sc[kl * dh + cv * per + i] = o[i];
  __syncthreads();
  if (tid < dh) {
    float tot = 0.f;
    for (int s2 = 0; s2 < KL; ++s2) tot += sc[s2 * dh + tid];
    float out = tot * inv;
    if (p.res) out += ElemIO<T>::ld(reinterpret_cast<const T*>(p.res) + (long long)b * p.res_ld + h * dh + tid);
    ElemIO<T>::st(reinterpret_cast<T*>(p.o) + (long long)b * p.o_ld + h * dh + tid, out);
  }
}

// returns 1 if launched, 0 if the geometry is not a single-query decode this kernel covers.  lk_dev != null: attend keys
// 0 .. *lk_dev (read on the device at run time); d.Lk is then the upper bound the LDS buffer is sized for.
extern "C" int gm_attention_decode_dev(const GmAttnDesc* dp, const int* lk_dev, void* stream) {
  const GmAttnDesc& d = *dp;
  if (d.Lq != 1 || d.Lk > DEC_MAX_KEYS || d.dh > 256 || d.Lk < 1) return 0;
  if (d.dtype != GM_F32 && d.dtype != GM_BF16) return 0;
  hipStream_t st = (hipStream_t)stream;
  const size_t smem = (size_t)((((d.Lk > 2048 ? d.Lk : 2048) + 3) & ~3) + d.dh + 256) * sizeof(float);
  if (d.dtype == GM_F32) attn_decode_kernel<float><<<d.B * d.H, 256, smem, st>>>(d, lk_dev);
  else attn_decode_kernel<bf16_raw><<<d.B * d.H, 256, smem, st>>>(d, lk_dev);
  return 1;
}

extern "C" int gm_attention_decode_try(const GmAttnDesc* dp, void* stream) { return gm_attention_decode_dev(dp, nullptr, stream); }

// ---------------------------------------------------------------------------------------------------------------------------------
// Split-KV form of the single-query attention (round 3).  `attn_decode_kernel` gives one (batch, head) to ONE work-group: at 4096 cached
// keys each thread walks 16 keys one dependent L2 round trip after the other and then 16 value rows -- 20-25 us per layer on 8 of the
// chip's 256 CUs.  Here the keys of a (batch, head) are cut into NS contiguous ranges (a multiple of 64 keys each, so a range is one key
// per thread at NS = 16 and 4096 keys); work-group (bh, s) writes its range's (running max m, sum l of exp(score - m), un-normalised
// output sum_j exp(score_j - m) v_j) to the workspace and the consumer -- `attn_decode_combine_kernel`, or the out-projection GEMM's
// prologue (`kv_merge_vec`) -- merges the NS partials in range order:
//     M = max_s m_s,   out = (sum_s e^{m_s - M} o_s) / (sum_s e^{m_s - M} l_s)   (+ residual).
// The partition depends on the key count only (host value, or `*lk_dev + 1` under graph replay), every sum has a fixed order: the eager
// and the replayed step give identical bits.  Workspace: B * H * NS * (dh + 2) floats.
// ---------------------------------------------------------------------------------------------------------------------------------
// One key range of the single-query attention: scores over keys K[0 .. n), softmax state, un-normalised output -> the range's partial
// (shared by attn_decode_split_kernel and the fused q|k|v + attention kernel below).  qs: the scaled query in LDS (published by the caller);
// sc: LDS scores / probabilities, later the [KL][dh] partial table; red: 4 floats.  K, V point at the range's first key.
template <typename T>
__device__ __forceinline__ void attn_range_body(const T* __restrict__ K, long long k_ld, const T* __restrict__ V, long long v_ld, int n, int dh,
                                                const float* qs, float* sc, float* red, float* wsp, float* wm, float* wl, int tid) {
  struct { long long k_ld, v_ld; } p = {k_ld, v_ld};
  float mx = -INFINITY;
  constexpr int VECW = 16 / (int)sizeof(T);
  const bool kvec = (dh % VECW == 0) && (p.k_ld % VECW == 0) && ((reinterpret_cast<uintptr_t>(K) & 15) == 0);
  for (int j = tid; j < n; j += 256) {
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
  for (int j = tid; j < n; j += 256) {
    const float e = expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  const float tot_l = ((red[0] + red[1]) + red[2]) + red[3];
  const bool vvec = (dh % VECW == 0) && (p.v_ld % VECW == 0) && ((reinterpret_cast<uintptr_t>(V) & 15) == 0);
  float o[VECW];
#pragma unroll
  for (int i = 0; i < VECW; ++i) o[i] = 0.f;
  const int nv = vvec ? dh / VECW : dh;
  const int KL = 256 / nv > 0 ? 256 / nv : 1;
  const int cv = tid % nv, kl = tid / nv;
  if (kl < KL) {
    if (vvec) {
      int j = kl;
      for (; j + 3 * KL < n; j += 4 * KL) {
        uint4 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const uint4*>(V + (long long)(j + u * KL) * p.v_ld + cv * VECW);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float vv[VECW];
          Vec16<T>::unpack(r[u], vv);
          const float pj = sc[j + u * KL];
#pragma unroll
          for (int i = 0; i < VECW; ++i) o[i] += pj * vv[i];
        }
      }
      for (; j < n; j += KL) {
        float vv[VECW];
        Vec16<T>::unpack(*reinterpret_cast<const uint4*>(V + (long long)j * p.v_ld + cv * VECW), vv);
        const float pj = sc[j];
#pragma unroll
        for (int i = 0; i < VECW; ++i) o[i] += pj * vv[i];
      }
    } else {
      for (int j = kl; j < n; j += KL) o[0] += sc[j] * ElemIO<T>::ld(V + (long long)j * p.v_ld + cv);
    }
  }
  __syncthreads();
  const int per = vvec ? VECW : 1;
  if (kl < KL)
    for (int i = 0; i < per; ++i) sc[kl * dh + cv * per + i] = o[i];
  __syncthreads();
  if (tid < dh) {
    float tot = 0.f;
    for (int s2 = 0; s2 < KL; ++s2) tot += sc[s2 * dh + tid];
    wsp[tid] = tot;
  }
  if (tid == 0) { *wm = mx; *wl = tot_l; }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_decode_split_kernel(GmAttnDesc p, const int* __restrict__ lk_dev, float* __restrict__ ws, int NS,
                                                               int sc_elems, int chunk, int cap) {
  if (!p.k_bs) p.k_bs = (long long)cap * p.k_ld;  // dense caches: `cap` rows per batch entry
  if (!p.v_bs) p.v_bs = (long long)cap * p.v_ld;
  if (lk_dev) p.Lk = *lk_dev + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sc = reinterpret_cast<float*>(smem);  // [chunk] scores, then probabilities; later the [KL][dh] partial table
  float* qs = sc + sc_elems;                   // [dh]
  float* red = qs + p.dh;                      // [4]
  const int tid = threadIdx.x;
  const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H, sp = blockIdx.y;
  const int dh = p.dh;
  const int k0 = sp * chunk, k1 = min(p.Lk, k0 + chunk);
  const long long BH = (long long)gridDim.x;
  float* wsp = ws + ((long long)bh * NS + sp) * dh;              // o[BH][NS][dh]
  float* wm = ws + BH * NS * dh + (long long)bh * NS + sp;       // m[BH][NS]
  float* wl = wm + BH * NS;                                      // l[BH][NS]
  if (k0 >= p.Lk) {  // an empty range: weight zero in the merge
    if (tid < dh) wsp[tid] = 0.f;
    if (tid == 0) { *wm = -INFINITY; *wl = 0.f; }
    return;
  }
  const int n = k1 - k0;
  const T* Q = reinterpret_cast<const T*>(p.q) + (long long)b * p.q_ld + h * dh;
  const T* K = reinterpret_cast<const T*>(p.k) + (long long)b * (p.k_bs ? p.k_bs : (long long)p.Lk * p.k_ld) + h * dh + (long long)k0 * p.k_ld;
  const T* V = reinterpret_cast<const T*>(p.v) + (long long)b * (p.v_bs ? p.v_bs : (long long)p.Lk * p.v_ld) + h * dh + (long long)k0 * p.v_ld;
  for (int c = tid; c < dh; c += 256) qs[c] = ElemIO<T>::ld(Q + c) * p.scale;
  __syncthreads();
  attn_range_body<T>(K, p.k_ld, V, p.v_ld, n, dh, qs, sc, red, wsp, wm, wl, tid);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// LayerNorm + q | k | v projection + one key range of the single-query attention in ONE launch (round 3): work-group (b * H + h, range) assembles
// the residual-stream row of sample b (plain, or x1 + b2 + sum_j P[j] behind a fused MLP -- work-group (h = 0, range 0) also stores it for the
// out-projection's residual), normalises it, and multiplies it with head h's 3 * dh rows of the stacked projection: the K chunks are dealt to the
// four waves, the partial sums meet in LDS in wave order.  q stays in LDS (scaled, rounded through T like the stored q of the two-launch form);
// the work-group whose range holds the position stores the new k / v rows into the caches and -- after a fence and a barrier -- reads them back
// with the rest of its range; the others never touch that row.  Every work-group recomputes q (3 * dh x C MACs: nothing) and requests ALL its
// weight fragments at entry; what the fusion buys is one ~4.5 us launch per block of the token's dependent chain.
// NG = 3 * dh / 16 output groups, UM = K chunks per wave (host: C % BK == 0, ceil(C / BK / 4) <= UM).
// ---------------------------------------------------------------------------------------------------------------------------------
struct QkvAttnArgs {
  const void* x0;                                                       // [B][C] rows, or null with the partial source below
  const float* mlp_p; int mlp_nj; const void* mlp_x1; const float* mlp_b2; void* x0_out;
  const float* ln_g; const float* ln_b; float ln_eps;
  const void* w; const float* bias;                                     // packed [chunk][3C pad 16][BK]; fp32 [3C] or null
  void* kcache; void* vcache;                                           // [B][cap][C]
  int B, H, C, dh, cap, pos; const int* pos_dev;
  float scale; float* ws; int chunk, sc_elems;
};

template <typename T, int NG, int UM>
__global__ __launch_bounds__(256) void qkv_attn_rows_kernel(QkvAttnArgs a) {
  constexpr int BK = ConvTraits<T>::BK, VECW = ConvTraits<T>::VECW, NS = GM_DECODE_KV_SPLITS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int C = a.C, dh = a.dh;
  float* sc = reinterpret_cast<float*>(smem);      // [sc_elems]
  float* qs = sc + a.sc_elems;                     // [dh]
  float* red = qs + dh;                            // [8]
  float* xrow = red + 8;                           // [C] the assembled row (values already rounded through T)
  float* part = xrow + C;                          // [4][3 dh]
  T* xn = reinterpret_cast<T*>(part + 4 * 3 * dh); // [C] LayerNorm'ed row (16-byte aligned: every count above is a multiple of 4 floats)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q4 = lane >> 4;
  const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H, sp = blockIdx.y;
  const int nchunks = C / BK, cout_pad = (3 * C + 15) & ~15;
  constexpr int GPH = NG / 3;  // output groups per head part (dh / 16)
  // ---- requested at entry: the position, every weight fragment of this wave, this thread's bias -------------------------------------------
  const int pos = a.pos_dev ? *a.pos_dev : a.pos;
  const T* W = reinterpret_cast<const T*>(a.w);
  uint4 wfr[NG][UM];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int co = (g / GPH) * C + h * dh + (g % GPH) * 16 + l15;
#pragma unroll
    for (int u = 0; u < UM; ++u) {
      const int c = wave + 4 * u < nchunks ? wave + 4 * u : nchunks - 1;
      wfr[g][u] = *reinterpret_cast<const uint4*>(W + ((long long)c * cout_pad + co) * BK + q4 * VECW);
    }
  }
  const int tt = tid < 3 * dh ? tid : 0;
  const float bval = (a.bias ? a.bias : reinterpret_cast<const float*>(a.w))[a.bias ? (tt / dh) * C + h * dh + tt % dh : 0];
  const int Lk = pos + 1;
  const int k0 = sp * a.chunk, k1 = min(Lk, k0 + a.chunk);
  const long long BH = (long long)gridDim.x;
  float* wsp = a.ws + ((long long)bh * NS + sp) * dh;
  float* wm = a.ws + BH * NS * dh + (long long)bh * NS + sp;
  float* wl = wm + BH * NS;
  if (k0 >= Lk) {  // an empty range (never range 0, never the position's range): weight zero in the merge
    if (tid < dh) wsp[tid] = 0.f;
    if (tid == 0) { *wm = -INFINITY; *wl = 0.f; }
    return;
  }
  const bool owner = pos >= k0 && pos < k0 + a.chunk;
  // ---- the residual-stream row of sample b ------------------------------------------------------------------------------------------------
  for (int idx = tid; idx < C; idx += 256) {
    float v;
    if (a.mlp_p) {
      const float* b2 = a.mlp_b2 ? a.mlp_b2 : a.mlp_p;
      v = ElemIO<T>::ld(reinterpret_cast<const T*>(a.mlp_x1) + (long long)b * C + idx) + b2[a.mlp_b2 ? idx : 0] * (a.mlp_b2 ? 1.f : 0.f);
      const float* pp = a.mlp_p + (long long)b * C + idx;
      const long long pstride = (long long)a.B * C;
      constexpr int PB = 8;
      for (int j0 = 0; j0 < a.mlp_nj; j0 += PB) {
        float t[PB];
#pragma unroll
        for (int j = 0; j < PB; ++j) t[j] = pp[(j0 + j < a.mlp_nj ? j0 + j : j0) * pstride];
#pragma unroll
        for (int j = 0; j < PB; ++j) v += j0 + j < a.mlp_nj ? t[j] : 0.f;  // slice order
      }
    } else {
      v = ElemIO<T>::ld(reinterpret_cast<const T*>(a.x0) + (long long)b * C + idx);
    }
    T r;
    ElemIO<T>::st(&r, v);
    xrow[idx] = ElemIO<T>::ld(&r);
    if (a.mlp_p && a.x0_out && h == 0 && sp == 0) reinterpret_cast<T*>(a.x0_out)[(long long)b * C + idx] = r;
  }
  __syncthreads();
  // ---- LayerNorm (two-pass statistics, like the reference's fp32 computation) -------------------------------------------------------------
  float part_s = 0.f;
  for (int idx = tid; idx < C; idx += 256) part_s += xrow[idx];
  part_s = wave_sum(part_s);
  if (lane == 0) red[wave] = part_s;
  __syncthreads();
  const float mean = (((red[0] + red[1]) + red[2]) + red[3]) / (float)C;
  float part_q = 0.f;
  for (int idx = tid; idx < C; idx += 256) part_q += (xrow[idx] - mean) * (xrow[idx] - mean);
  part_q = wave_sum(part_q);
  if (lane == 0) red[4 + wave] = part_q;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((((red[4] + red[5]) + red[6]) + red[7]) / (float)C + a.ln_eps);
  for (int idx = tid; idx < C; idx += 256)
    ElemIO<T>::st(xn + idx, (xrow[idx] - mean) * rstd * a.ln_g[idx] + (a.ln_b ? a.ln_b[idx] : 0.f));
  __syncthreads();
  // ---- q | k | v of head h: this wave's K chunks, then the four waves' partial sums in wave order ---------------------------------------------
  f32x4_t acc[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) acc[g] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < UM; ++u) {
    const int c = wave + 4 * u;
    if (c < nchunks) {  // (wave-uniform)
      const uint4 xv = *reinterpret_cast<const uint4*>(xn + c * BK + q4 * VECW);
      const uint4 xf = make_uint4(l15 == 0 ? xv.x : 0u, l15 == 0 ? xv.y : 0u, l15 == 0 ? xv.z : 0u, l15 == 0 ? xv.w : 0u);  // MFMA column 0 = the row
#pragma unroll
      for (int g = 0; g < NG; ++g) Mma<T>::run(wfr[g][u], xf, acc[g]);
    }
  }
  if (l15 == 0) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) part[wave * 3 * dh + g * 16 + 4 * q4 + i] = acc[g][i];
  }
  __syncthreads();
  if (tid < 3 * dh) {
    const float val = (((part[tid] + part[3 * dh + tid]) + part[6 * dh + tid]) + part[9 * dh + tid]) + (a.bias ? bval : 0.f);
    T r;
    ElemIO<T>::st(&r, val);
    const int which = tid / dh, c = tid - which * dh;
    if (which == 0) {
      qs[c] = ElemIO<T>::ld(&r) * a.scale;
    } else if (owner) {
      T* dst = reinterpret_cast<T*>(which == 1 ? a.kcache : a.vcache) + ((long long)b * a.cap + pos) * C + h * dh + c;
      *dst = r;
    }
  }
  if (owner) __threadfence();  // the new rows are in L2 before any wave of this work-group reads its range
  __syncthreads();
  const T* K = reinterpret_cast<const T*>(a.kcache) + ((long long)b * a.cap + k0) * C + h * dh;
  const T* V = reinterpret_cast<const T*>(a.vcache) + ((long long)b * a.cap + k0) * C + h * dh;
  attn_range_body<T>(K, C, V, C, k1 - k0, dh, qs, sc, red, wsp, wm, wl, tid);
}

// 1 = launched, 0 = not this kernel's case (the caller issues the LayerNorm + q|k|v GEMM and gm_attention_decode_split), < 0 = error.
extern "C" int gm_qkv_attn_rows(const QkvAttnArgs* ap, int dtype, void* stream) {
  const QkvAttnArgs& a = *ap;
  static const bool on = !(getenv("GM_DECODE_QKV_FUSE") && getenv("GM_DECODE_QKV_FUSE")[0] == '0');  // bench switch (tools/diag_c5.py)
  if (!on || (dtype != GM_F32 && dtype != GM_BF16)) return 0;
  const int bk = dtype == GM_F32 ? 16 : 32, vecw = dtype == GM_F32 ? 4 : 8;
  if (!a.ln_g || !a.w || !a.kcache || !a.vcache || !a.ws || (!a.x0 && !a.mlp_p)) return 0;
  if (a.dh % 16 || a.C % bk || a.C != a.H * a.dh || a.C % vecw) return 0;
  const int ng = 3 * a.dh / 16, nchunks = a.C / bk, um = (nchunks + 3) / 4;
  const int chunk = ((a.cap + GM_DECODE_KV_SPLITS - 1) / GM_DECODE_KV_SPLITS + 63) & ~63;
  const int sc_elems = chunk > 256 * vecw ? chunk : 256 * vecw;
  if (sc_elems > 32768 || (long long)a.B * a.H > 65535) return 0;
  QkvAttnArgs k = a;
  k.chunk = chunk; k.sc_elems = sc_elems;
  const size_t smem = (size_t)(sc_elems + a.dh + 8 + a.C + 12 * a.dh) * sizeof(float) + (size_t)a.C * (dtype == GM_F32 ? 4 : 2);
  if (smem > 160 * 1024) return 0;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(a.B * a.H, GM_DECODE_KV_SPLITS);
#define GM_QKV_LAUNCH(T, NG, UM)                                                                                          \
  do {                                                                                                                    \
    static bool attr = false;                                                                                             \
    if (!attr) {                                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(qkv_attn_rows_kernel<T, NG, UM>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr = true;                                                                                                        \
    }                                                                                                                     \
    qkv_attn_rows_kernel<T, NG, UM><<<grid, 256, smem, st>>>(k);                                                           \
    return hipGetLastError() == hipSuccess ? 1 : -1;                                                                      \
  } while (0)
  if (dtype == GM_BF16) {
    if (ng == 6 && um <= 2) GM_QKV_LAUNCH(bf16_raw, 6, 2);
    if (ng == 6 && um <= 4) GM_QKV_LAUNCH(bf16_raw, 6, 4);
    if (ng == 12 && um <= 2) GM_QKV_LAUNCH(bf16_raw, 12, 2);
  } else {
    if (ng == 6 && um <= 2) GM_QKV_LAUNCH(float, 6, 2);
    if (ng == 6 && um <= 4) GM_QKV_LAUNCH(float, 6, 4);
    if (ng == 12 && um <= 2) GM_QKV_LAUNCH(float, 12, 2);
  }
#undef GM_QKV_LAUNCH
  return 0;
}


template <typename T>
__global__ __launch_bounds__(64) void attn_decode_combine_kernel(GmAttnDesc p, const float* __restrict__ ws) {
  const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H, dh = p.dh;
  for (int c = threadIdx.x; c < dh; c += 64) {
    float out = kv_merge_one(ws, gridDim.x, dh, bh, c);
    if (p.res) out += ElemIO<T>::ld(reinterpret_cast<const T*>(p.res) + (long long)b * p.res_ld + h * dh + c);
    ElemIO<T>::st(reinterpret_cast<T*>(p.o) + (long long)b * p.o_ld + h * dh + c, out);
  }
}

// 1 = launched, 0 = not this kernel's case.  `ws` holds gm_attention_decode_split_ws_elems(B, H, dh, nsplit) floats.
extern "C" long long gm_attention_decode_split_ws_elems(int B, int H, int dh, int nsplit) { return (long long)B * H * nsplit * (dh + 2); }
// `merge`: 1 = partials + merge; 0 = partials only, for a consumer that merges them itself (gm_linear_rows_kvmerge); 2 = the merge launch only.
// `cap` = rows of the K / V caches per batch entry (>= the key count): the key ranges are cut from it, so they are the same for every position.
extern "C" int gm_attention_decode_split(const GmAttnDesc* dp, const int* lk_dev, float* ws, int nsplit, int merge, int cap, void* stream) {
  const GmAttnDesc& d = *dp;
  if (d.Lq != 1 || d.dh > 256 || d.Lk < 1 || cap < d.Lk || nsplit != GM_DECODE_KV_SPLITS || !ws) return 0;  // (the merge code unrolls over the split count)
  if (d.dtype != GM_F32 && d.dtype != GM_BF16) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int vecw = d.dtype == GM_F32 ? 4 : 8;
  const int chunk = ((cap + nsplit - 1) / nsplit + 63) & ~63;
  const int sc_elems = chunk > 256 * vecw ? chunk : 256 * vecw;  // the score buffer doubles as the [KL][dh] partial table
  if (sc_elems > 32768) return 0;
  const size_t smem = (size_t)(sc_elems + d.dh + 4) * sizeof(float);
  // (measured and removed: a variant issuing the key row, the value vectors, the query and the key count before its first wait -- one exposed
  //  round trip instead of four -- ran 6.4 vs 6.5 us: these launches are bound by their ~4.5 us floor and their instruction count, not by loads)
  dim3 grid(d.B * d.H, nsplit);
  if (d.dtype == GM_F32) {
    if (merge != 2) attn_decode_split_kernel<float><<<grid, 256, smem, st>>>(d, lk_dev, ws, nsplit, sc_elems, chunk, cap);
    if (merge) attn_decode_combine_kernel<float><<<d.B * d.H, 64, 0, st>>>(d, ws);
  } else {
    if (merge != 2) attn_decode_split_kernel<bf16_raw><<<grid, 256, smem, st>>>(d, lk_dev, ws, nsplit, sc_elems, chunk, cap);
    if (merge) attn_decode_combine_kernel<bf16_raw><<<d.B * d.H, 64, 0, st>>>(d, ws);
  }
  return 1;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// One categorical draw per row by inverse CDF: idx = min{ j : sum_{i <= j} p_i >= u * sum_i p_i }, u uniform in [0, 1) supplied by
// the caller's generator.  torch.multinomial validates its input with a device -> host read, i.e. one pipeline drain per sampled
// token (1.5 ms per token in tools/diag_c5.py); this kernel keeps the sampling loop asynchronous.  One wave per row.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sample_index_kernel(const float* __restrict__ probs, int V, const float* __restrict__ u,
                                                         long long* __restrict__ out) {
  const long long row = blockIdx.x;
  const int lane = threadIdx.x;
  const float* pr = probs + row * (long long)V;
  const int per = (V + 63) / 64, j0 = lane * per, j1 = min(V, j0 + per);
  float loc = 0.f;
  for (int j = j0; j < j1; ++j) loc += pr[j];
  float incl = loc;  // inclusive scan over lanes
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  const float total = __shfl(incl, 63, 64);
  const float target = u[row] * total;
  const float excl = incl - loc;
  // the first lane whose inclusive sum reaches the target owns the draw; rounding can leave no such lane: take the last non-empty
  const bool mine = (incl >= target) && (excl <= target) && loc > 0.f;
  unsigned long long ballot = __ballot(mine);
  int idx = -1;
  if (ballot) {
    const int owner = __ffsll((long long)ballot) - 1;
    if (lane == owner) {
      float c = excl;
      idx = j1 - 1;
      for (int j = j0; j < j1; ++j) {
        c += pr[j];
        if (c >= target && pr[j] > 0.f) { idx = j; break; }
      }
      out[row] = idx;
    }
  } else {
    // target beyond the accumulated total (u ~ 1 with rounding): the last entry with positive probability
    int last = -1;
    for (int j = j0; j < j1; ++j) if (pr[j] > 0.f) last = j;
    for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o, 64));
    if (lane == 0) out[row] = last < 0 ? 0 : last;
  }
}

extern "C" int gm_sample_index(const float* probs, long long rows, int V, const float* u, long long* out, void* stream) {
  GM_REQUIRE(probs && u && out, "null pointer");
  GM_REQUIRE(V > 0, "empty vocabulary");
  if (rows == 0) return 0;
  GM_REQUIRE(rows < (1LL << 31), "too many rows");
  sample_index_kernel<<<(unsigned)rows, 64, 0, (hipStream_t)stream>>>(probs, V, u, out);
  GM_LAUNCH_CHECK();
}
