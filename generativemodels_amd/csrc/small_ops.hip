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

// optional extras of the small-row GEMM: a LayerNorm over the input row as its prologue (transformer pre-norm blocks) and a split of
// the output channels over three destinations (the stacked q | k | v projection writing k and v straight into the KV caches)
struct LinearRowsExtra {
  const float* ln_g; const float* ln_b; float ln_eps;  // ln_g != null: x <- LayerNorm(x) * g + b before the GEMM
  void* y1; void* y2; long long y12_ld; int split;     // split > 0: channels [split, 2*split) -> y1, [2*split, 3*split) -> y2 (row pitch y12_ld)
  const int* off_dev; long long off_mul;               // y1 / y2 are advanced by (*off_dev) * off_mul elements at run time (KV-cache row = position)
};

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
  f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  constexpr int U = 4;
  for (int c0 = 0; c0 < nchunks; c0 += U) {
    uint4 wf[U], xf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u < nchunks ? c0 + u : nchunks - 1;  // clamped: the duplicate is discarded below
      wf[u] = *reinterpret_cast<const uint4*>(wrow + (long long)c * cout_pad * BK);
      const bool ok = row_ok & (c * BK + q * VECW + VECW <= cin);  // host: cin % VECW == 0
      // masked lanes read the first vector of the tensor: xrow + 0 is q * VECW elements into the row, which lies beyond a row (and, in
      // the last row, beyond the allocation) whenever cin < 4 * VECW -- a faulting read even though its value is discarded
      const uint4 v = *reinterpret_cast<const uint4*>(ok ? xrow + c * BK : x);
      xf[u] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (c0 + u >= nchunks) break;
      uint4 b = xf[u];
      if (pre_act || ex.ln_g) {
        float v[VECW];
        Vec16<T>::unpack(b, v);
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
      float v = acc[i] + (bias ? bias[co] : 0.f);
      v = conv_post_act(v, post_act);
      if (res) v += ElemIO<T>::ld(res + (long long)row * res_ld + co);
      if (ex.split > 0 && co >= ex.split) {
        T* dst = reinterpret_cast<T*>(co < 2 * ex.split ? ex.y1 : ex.y2) + (ex.off_dev ? (long long)(*ex.off_dev) * ex.off_mul : 0);
        ElemIO<T>::st(dst + (long long)row * ex.y12_ld + (co - (co < 2 * ex.split ? ex.split : 2 * ex.split)), v);
      } else {
        ElemIO<T>::st(y + (long long)row * y_ld + co, v);
      }
    }
  }
}

static int linear_rows_launch(const void* x, long long x_ld, const void* w, const float* bias, const void* res, long long res_ld, void* y,
                              long long y_ld, int rows, int cin, int cout, int pre_act, int post_act, int dtype, const LinearRowsExtra& ex,
                              void* stream) {
  GM_REQUIRE(x && w && y, "null pointer");
  GM_REQUIRE(rows >= 0 && cin > 0 && cout > 0, "bad geometry");
  if (rows == 0) return 0;
  const int vecw = dtype == GM_F32 ? 4 : 8;
  GM_REQUIRE(cin % vecw == 0 && x_ld % vecw == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "x rows must be 16-byte vectors");
  GM_REQUIRE(ex.split == 0 || (ex.y1 && ex.y2 && cout == 3 * ex.split), "split output needs two extra destinations and cout = 3 * split");
  hipStream_t st = (hipStream_t)stream;
  const int cout_pad = (cout + 15) & ~15;
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
