// Small kernels around the autoregressive transformer:
//   gm_embed_tokens   token + absolute position embedding            (reference: networks/nets/transformer.py:99-101)
//   gm_sample_probs   the sampling head: temperature, top-k crop, softmax, BOS probability zeroed
//                                                                    (reference: inferers/inferer.py:1221-1232)
//   gm_token_log_prob log(softmax(logits)[target]) per row           (reference: inferers/inferer.py:1290-1296, 1310-1316)
// One wave per row; the vocabulary (num_embeddings + 1, a few hundred to a few thousand entries) is walked by the 64 lanes.
#include "gm_common.h"

template <typename T>
__global__ __launch_bounds__(256) void embed_tokens_kernel(const long long* __restrict__ idx, const T* __restrict__ tok,
                                                          const T* __restrict__ pos, T* __restrict__ out, long long rows, int T_len,
                                                          int C, int pos0, int num_tokens, int max_pos, const int* __restrict__ pos_dev) {
  if (pos_dev) pos0 = *pos_dev;  // decode-graph replay: the position lives on the device
  const long long total = rows * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / C;
    const int c = (int)(i - r * C);
    long long t = idx[r];
    t = t < 0 ? 0 : (t >= num_tokens ? num_tokens - 1 : t);  // host validates; clamp keeps a bad index from faulting
    int ps = pos0 + (int)(r % T_len);
    ps = ps >= max_pos ? max_pos - 1 : ps;
    ElemIO<T>::st(out + i, ElemIO<T>::ld(tok + t * C + c) + ElemIO<T>::ld(pos + (long long)ps * C + c));
  }
}

extern "C" int gm_embed_tokens_dev(const long long* indices, const void* token_weight, const void* position_weight, void* out, long long batch,
                                   int seq_len, int C, int pos0, int num_tokens, int max_positions, int dtype, const int* pos_dev, void* stream);

extern "C" int gm_embed_tokens(const long long* indices, const void* token_weight, const void* position_weight, void* out,
                               long long batch, int seq_len, int C, int pos0, int num_tokens, int max_positions, int dtype, void* stream) {
  return gm_embed_tokens_dev(indices, token_weight, position_weight, out, batch, seq_len, C, pos0, num_tokens, max_positions, dtype, nullptr, stream);
}

// pos_dev != null: the first position is read from device memory at run time (pos0 is only validated)
extern "C" int gm_embed_tokens_dev(const long long* indices, const void* token_weight, const void* position_weight, void* out, long long batch,
                                   int seq_len, int C, int pos0, int num_tokens, int max_positions, int dtype, const int* pos_dev, void* stream) {
  GM_REQUIRE(indices && token_weight && position_weight && out, "null pointer");
  GM_REQUIRE(seq_len > 0 && C > 0 && pos0 >= 0 && pos0 + seq_len <= max_positions, "positions exceed the embedding table");
  const long long rows = batch * seq_len;
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  long long g = (rows * C + 255) / 256;
  if (g > 4096) g = 4096;
  if (dtype == GM_F32)
    embed_tokens_kernel<float><<<(int)g, 256, 0, st>>>(indices, (const float*)token_weight, (const float*)position_weight, (float*)out, rows,
                                                       seq_len, C, pos0, num_tokens, max_positions, pos_dev);
  else if (dtype == GM_BF16)
    embed_tokens_kernel<bf16_raw><<<(int)g, 256, 0, st>>>(indices, (const bf16_raw*)token_weight, (const bf16_raw*)position_weight,
                                                          (bf16_raw*)out, rows, seq_len, C, pos0, num_tokens, max_positions, pos_dev);
  else
    GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// order-preserving map float -> uint (larger float <-> larger uint)
__device__ __forceinline__ unsigned f2key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// probs[row] = softmax(crop_topk(logits[row] / temperature)); probs[row][bos] = 0.  One wave per row.
template <typename T>
__global__ __launch_bounds__(64) void sample_probs_kernel(const T* __restrict__ logits, long long ld, float* __restrict__ probs, int V,
                                                         float temperature, int top_k, int bos) {
  const long long row = blockIdx.x;
  const int lane = threadIdx.x;
  const T* lr = logits + row * ld;
  float* pr = probs + row * (long long)V;
  // scaled logits are kept in the output row (fp32) between the passes
  for (int j = lane; j < V; j += 64) pr[j] = ElemIO<T>::ld(lr + j) / temperature;
  float thr = -INFINITY;
  if (top_k > 0 && top_k < V) {
    // k-th largest value by a most-significant-bit-first radix walk over the order-preserving keys (ties are kept, like
    // `logits < v[:, [-1]]` in the reference)
    unsigned prefix = 0, mask = 0;
    int need = top_k;
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned cand = prefix | (1u << bit), m2 = mask | (1u << bit);
      int cnt = 0;
      for (int j = lane; j < V; j += 64) cnt += ((f2key(pr[j]) & m2) == cand) ? 1 : 0;
      cnt = wave_sum_i(cnt);
      if (cnt >= need) prefix = cand; else need -= cnt;
      mask = m2;
    }
    const unsigned u = (prefix & 0x80000000u) ? (prefix & 0x7fffffffu) : ~prefix;
    thr = __uint_as_float(u);
  }
  float mx = -INFINITY;
  for (int j = lane; j < V; j += 64) { const float v = pr[j]; if (v >= thr) mx = fmaxf(mx, v); }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < V; j += 64) { const float v = pr[j]; sum += v >= thr ? expf(v - mx) : 0.f; }
  sum = wave_sum(sum);
  for (int j = lane; j < V; j += 64) {
    const float v = pr[j];
    pr[j] = (v >= thr && j != bos) ? expf(v - mx) / sum : 0.f;
  }
}

extern "C" int gm_sample_probs(const void* logits, long long ld, float* probs, long long rows, int V, float temperature, int top_k,
                               int bos_index, int dtype, void* stream) {
  GM_REQUIRE(logits && probs, "null pointer");
  GM_REQUIRE(V > 0 && temperature > 0.f, "bad vocabulary size / temperature");
  if (rows == 0) return 0;
  GM_REQUIRE(rows < (1LL << 31), "too many rows");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GM_F32)
    sample_probs_kernel<float><<<(unsigned)rows, 64, 0, st>>>((const float*)logits, ld, probs, V, temperature, top_k, bos_index);
  else if (dtype == GM_BF16)
    sample_probs_kernel<bf16_raw><<<(unsigned)rows, 64, 0, st>>>((const bf16_raw*)logits, ld, probs, V, temperature, top_k, bos_index);
  else
    GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// out[row] = log(softmax(logits[row])[target[row]])
template <typename T>
__global__ __launch_bounds__(64) void token_log_prob_kernel(const T* __restrict__ logits, long long ld, const long long* __restrict__ target,
                                                           float* __restrict__ out, int V) {
  const long long row = blockIdx.x;
  const int lane = threadIdx.x;
  const T* lr = logits + row * ld;
  float mx = -INFINITY;
  for (int j = lane; j < V; j += 64) mx = fmaxf(mx, ElemIO<T>::ld(lr + j));
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < V; j += 64) sum += expf(ElemIO<T>::ld(lr + j) - mx);
  sum = wave_sum(sum);
  if (lane == 0) {
    long long t = target[row];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    out[row] = logf(expf(ElemIO<T>::ld(lr + t) - mx) / sum);
  }
}

extern "C" int gm_token_log_prob(const void* logits, long long ld, const long long* target, float* out, long long rows, int V, int dtype,
                                 void* stream) {
  GM_REQUIRE(logits && target && out, "null pointer");
  GM_REQUIRE(V > 0, "empty vocabulary");
  if (rows == 0) return 0;
  GM_REQUIRE(rows < (1LL << 31), "too many rows");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GM_F32)
    token_log_prob_kernel<float><<<(unsigned)rows, 64, 0, st>>>((const float*)logits, ld, target, out, V);
  else if (dtype == GM_BF16)
    token_log_prob_kernel<bf16_raw><<<(unsigned)rows, 64, 0, st>>>((const bf16_raw*)logits, ld, target, out, V);
  else
    GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// After a token has been drawn: append it to the sequence, make it the next step's input and advance the device-side position.
//   seq[b][pos + 1] = idx[b]; tokens[b] = idx[b]; pos += 1      (the bookkeeping of inferer.py:1237-1239 kept on the device so that a
// whole decode iteration -- step, sampling head, draw, this -- replays from one HIP graph)
__global__ void decode_advance_kernel(int* __restrict__ pos, long long* __restrict__ tokens, const long long* __restrict__ idx,
                                      long long* __restrict__ seq, int B, long long seq_ld) {
  const int p = *pos;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const long long t = idx[b];
    tokens[b] = t;
    if (p + 1 < seq_ld) seq[(long long)b * seq_ld + p + 1] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) *pos = p + 1;
}

extern "C" int gm_decode_advance(int* pos, long long* tokens, const long long* idx, long long* seq, int B, long long seq_ld, void* stream) {
  GM_REQUIRE(pos && tokens && idx && seq, "null pointer");
  if (B <= 0) return 0;
  decode_advance_kernel<<<1, 64, 0, (hipStream_t)stream>>>(pos, tokens, idx, seq, B, seq_ld);
  GM_LAUNCH_CHECK();
}
