// Vector-quantiser kernels (reference: networks/layers/vector_quantizer.py:86-138).
//   gm_vq_argmin : nearest code per token, d = |x|^2 + |e|^2 - 2 x.e in fp32 regardless of the storage dtype
//                  (vector_quantizer.py:102-116); ties -> lowest index (torch.max first-occurrence on CPU).
//   gm_vq_gather : codebook lookup into the N[D]HWC arena (vector_quantizer.py:124-138) and, optionally, the sum of squared
//                  differences to the encoder output for the eval-mode commitment loss (vector_quantizer.py:183).
// The NDHWC arena makes the reference's permute+flatten (vector_quantizer.py:106) free: a token row IS a voxel row.
#include "gm_common.h"

#define VQ_TILE_FLOATS 8192  // 32 KiB code tile in LDS

template <typename T>
__global__ __launch_bounds__(256) void vq_argmin_kernel(const T* __restrict__ x, long long x_ld, const float* __restrict__ emb,
                                                       long long* __restrict__ idx, long long tokens, int K, int D) {
  __shared__ float etile[VQ_TILE_FLOATS];
  __shared__ float enorm[VQ_TILE_FLOATS / 4];
  const long long tok = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool ok = tok < tokens;
  const T* xr = x + (ok ? tok : 0) * x_ld;
  float xx = 0.f;
  for (int c = 0; c < D; ++c) { const float v = ElemIO<T>::ld(xr + c); xx += v * v; }
  float best = INFINITY;
  int best_i = 0;
  int codes_per_tile = VQ_TILE_FLOATS / D;
  if (codes_per_tile > VQ_TILE_FLOATS / 4) codes_per_tile = VQ_TILE_FLOATS / 4;
  for (int k0 = 0; k0 < K; k0 += codes_per_tile) {
    const int kn = min(codes_per_tile, K - k0);
    __syncthreads();
    for (int i = threadIdx.x; i < kn * D; i += 256) etile[i] = emb[(long long)k0 * D + i];
    __syncthreads();
    for (int j = threadIdx.x; j < kn; j += 256) {
      float s = 0.f;
      for (int c = 0; c < D; ++c) s += etile[j * D + c] * etile[j * D + c];
      enorm[j] = s;
    }
    __syncthreads();
    if (ok) {
      for (int j = 0; j < kn; ++j) {
        float dot = 0.f;
        for (int c = 0; c < D; ++c) dot += ElemIO<T>::ld(xr + c) * etile[j * D + c];
        const float d = (xx + enorm[j]) - 2.0f * dot;
        if (d < best) { best = d; best_i = k0 + j; }
      }
    }
  }
  if (ok) idx[tok] = best_i;
}

extern "C" int gm_vq_argmin(const void* x, long long x_ld, const float* embedding, long long* indices, long long tokens,
                            int num_embeddings, int dim, int dtype, void* stream) {
  GM_REQUIRE(x && embedding && indices, "null pointer");
  GM_REQUIRE(dim > 0 && dim <= VQ_TILE_FLOATS && num_embeddings > 0, "bad codebook geometry");
  if (tokens == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int grid = gm_cdiv(tokens, 256);
  if (dtype == GM_F32) vq_argmin_kernel<float><<<grid, 256, 0, st>>>((const float*)x, x_ld, embedding, indices, tokens, num_embeddings, dim);
  else if (dtype == GM_BF16) vq_argmin_kernel<bf16_raw><<<grid, 256, 0, st>>>((const bf16_raw*)x, x_ld, embedding, indices, tokens, num_embeddings, dim);
  else GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// out[tok, c] = emb[idx[tok], c];  if x != null: partial[block] = sum over the block of (out - x)^2 (fp64)
template <typename T>
__global__ __launch_bounds__(256) void vq_gather_kernel(const long long* __restrict__ idx, const float* __restrict__ emb,
                                                       T* __restrict__ out, long long out_ld, const T* __restrict__ x,
                                                       long long x_ld, double* __restrict__ partial, long long tokens,
                                                       int K, int D) {
  __shared__ double red[4];
  double acc = 0.0;
  const long long total = tokens * D;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long tok = i / D;
    const int c = (int)(i - tok * D);
    long long k = idx[tok];
    if (k < 0) k = 0;
    if (k >= K) k = K - 1;
    const float e = emb[k * D + c];
    ElemIO<T>::st(out + tok * out_ld + c, e);
    if (x) { const double df = (double)e - (double)ElemIO<T>::ld(x + tok * x_ld + c); acc += df * df; }
  }
  if (partial) {
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  }
}

__global__ __launch_bounds__(64) void vq_sum_partials_kernel(const double* __restrict__ partial, int n, double scale,
                                                            float* __restrict__ out) {
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) a += partial[i];
  a = wave_sum(a);
  if (threadIdx.x == 0) *out = (float)(a * scale);
}

#define VQ_GATHER_MAX_BLOCKS 1024
extern "C" long long gm_vq_gather_workspace_bytes(void) { return VQ_GATHER_MAX_BLOCKS * (long long)sizeof(double); }

// sq_err_mean (optional, fp32 scalar on device) = mean((quantized - x)^2); needs x and workspace.
extern "C" int gm_vq_gather(const long long* indices, const float* embedding, void* out, long long out_ld, const void* x,
                            long long x_ld, float* sq_err_mean, void* workspace, long long tokens, int num_embeddings,
                            int dim, int dtype, void* stream) {
  GM_REQUIRE(indices && embedding && out, "null pointer");
  GM_REQUIRE(!sq_err_mean || (x && workspace), "sq_err_mean needs x and a workspace");
  if (tokens == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  long long g = (tokens * dim + 255) / 256;
  if (g > VQ_GATHER_MAX_BLOCKS) g = VQ_GATHER_MAX_BLOCKS;
  double* part = sq_err_mean ? (double*)workspace : nullptr;
  const void* xx = sq_err_mean ? x : nullptr;
  if (dtype == GM_F32)
    vq_gather_kernel<float><<<(int)g, 256, 0, st>>>(indices, embedding, (float*)out, out_ld, (const float*)xx, x_ld, part, tokens, num_embeddings, dim);
  else if (dtype == GM_BF16)
    vq_gather_kernel<bf16_raw><<<(int)g, 256, 0, st>>>(indices, embedding, (bf16_raw*)out, out_ld, (const bf16_raw*)xx, x_ld, part, tokens, num_embeddings, dim);
  else GM_FAIL(-2, "unsupported dtype");
  if (sq_err_mean) vq_sum_partials_kernel<<<1, 64, 0, st>>>(part, (int)g, 1.0 / ((double)tokens * dim), sq_err_mean);
  GM_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------------
// EMA codebook update of EMAQuantizer.forward in training mode (reference: networks/layers/vector_quantizer.py:166-180).
//   gm_vq_ema_stats : stats[e] = number of tokens assigned to code e, stats[K + e*D + d] = sum of their vectors ("encodings_sum" and
//                     "dw" of the reference, :168-169) -- ONE flat fp32 buffer, so that data-parallel ranks exchange it with ONE
//                     all-reduce instead of the reference's two (:155-157).  Two passes over a workspace of per-range partial tables (below):
//                     every index and every vector is read once; deterministic (no atomics: exclusive accumulator ownership, fixed orders).
//   gm_vq_ema_update: ema_cluster_size = decay * ema_cluster_size + (1 - decay) * counts; Laplace-smoothed weights; ema_w likewise;
//                     embedding = ema_w / weights (:173-180).  One work-group (the codebook is K x D <= a few hundred thousand floats).
// ---------------------------------------------------------------------------------------------------------------------
#define VQ_EMA_THREADS 256
#define VQ_EMA_RANGE 1024        // tokens per work-group of the partial pass (fewer, longer ranges above VQ_EMA_MAX_BLOCKS)
#define VQ_EMA_MAX_BLOCKS 1024
#define VQ_EMA_LDS_FLOATS 12288  // 48 KiB of per-code accumulators per work-group: codes are dealt to grid.y in chunks of this many / (D + 1)
// Pass 1 (round 4; rounds 1-3 ran one work-group per CODE that scanned every index 1 + D/32 times: O(K x tokens x D/32) index reads): a work-group
// owns a RANGE of tokens and a chunk of codes, walks its tokens ONCE in index order and adds each vector into the LDS accumulator row of its code.
// Ownership makes it deterministic without atomics: thread (g, d0) owns the accumulators (code e, dim d) with e % G == g and d % DT == d0
// (DT = 64-thread-aligned power of two <= D, G = 256 / DT), so every accumulator is written by exactly one thread, in token order.  The partial
// tables [block][K x (D + 1)] are summed in block order by pass 2.
template <typename T>
__global__ __launch_bounds__(VQ_EMA_THREADS) void vq_ema_partial_kernel(const T* __restrict__ x, long long x_ld, const long long* __restrict__ idx,
                                                                       long long tokens, long long range, int K, int D, int KC, int DT,
                                                                       float* __restrict__ part) {
  extern __shared__ float acc[];  // [KC][D + 1]: vector sums, then the count
  const int t = threadIdx.x, d0 = t % DT, g = t / DT, G = VQ_EMA_THREADS / DT;
  const int c0 = blockIdx.y * KC, kc = min(KC, K - c0);
  const long long r0 = (long long)blockIdx.x * range, r1 = min(tokens, r0 + range);
  for (int i = t; i < kc * (D + 1); i += VQ_EMA_THREADS) acc[i] = 0.f;
  __syncthreads();
  for (long long i = r0; i < r1; ++i) {
    const int e = (int)idx[i] - c0;  // the same address for every thread: one broadcast load
    if (e >= 0 && e < kc && (e % G) == g) {
      float* row = acc + e * (D + 1);
      const T* xr = x + i * x_ld;
      for (int d = d0; d < D; d += DT) row[d] += ElemIO<T>::ld(xr + d);
      if (d0 == 0) row[D] += 1.f;
    }
  }
  __syncthreads();
  float* out = part + ((long long)blockIdx.x * K + c0) * (D + 1);
  for (int i = t; i < kc * (D + 1); i += VQ_EMA_THREADS) out[i] = acc[i];
}

// Pass 2: stats[e] = sum over the blocks of the counts, stats[K + e*D + d] = ... of the vector sums, in block order (one thread per entry)
__global__ __launch_bounds__(VQ_EMA_THREADS) void vq_ema_fold_kernel(const float* __restrict__ part, int nblk, int K, int D, float* __restrict__ stats) {
  const long long n = (long long)K * (D + 1);
  for (long long i = (long long)blockIdx.x * VQ_EMA_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * VQ_EMA_THREADS) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // four loads in flight; combined in a fixed order
    int b = 0;
    for (; b + 3 < nblk; b += 4) {
      a0 += part[(long long)b * n + i]; a1 += part[(long long)(b + 1) * n + i]; a2 += part[(long long)(b + 2) * n + i]; a3 += part[(long long)(b + 3) * n + i];
    }
    for (; b < nblk; ++b) a0 += part[(long long)b * n + i];
    const float s = (a0 + a1) + (a2 + a3);
    const int e = (int)(i / (D + 1)), d = (int)(i % (D + 1));
    if (d == D) stats[e] = s; else stats[K + (long long)e * D + d] = s;
  }
}

static void vq_ema_plan(long long tokens, int K, int D, long long* range, int* nblk, int* KC, int* nchunk, int* DT) {
  long long r = VQ_EMA_RANGE;
  while ((tokens + r - 1) / r > VQ_EMA_MAX_BLOCKS) r *= 2;
  // byte budget for the partial tables (nblk x K x (D + 1) floats, written and re-read once per training forward): K = 2048, D = 64 over
  // >= 1 M tokens would otherwise be ~545 MB of scratch per call; 64 MiB keeps >= 123 blocks there, still a grid that fills the chip
  // once the K chunks multiply it.  The partition only changes which fixed-order partial sums are formed: results stay deterministic.
  const long long budget_elems = (64LL << 20) / 4, per_blk = (long long)K * (D + 1);
  while ((tokens + r - 1) / r > 1 && ((tokens + r - 1) / r) * per_blk > budget_elems) r *= 2;
  *range = r;
  *nblk = (int)((tokens + r - 1) / r);
  if (*nblk < 1) *nblk = 1;
  int kc = VQ_EMA_LDS_FLOATS / (D + 1);
  if (kc > K) kc = K;
  if (kc < 1) kc = 1;
  *KC = kc;
  *nchunk = (K + kc - 1) / kc;
  int dt = 1;
  while (dt * 2 <= D && dt * 2 <= 64) dt *= 2;
  *DT = dt;
}

// fp32 elements of the partial tables gm_vq_ema_stats needs as `workspace`
extern "C" long long gm_vq_ema_stats_workspace_elems(long long tokens, int num_embeddings, int dim) {
  if (num_embeddings <= 0 || dim <= 0) return 0;
  long long range; int nblk, KC, nchunk, DT;
  vq_ema_plan(tokens, num_embeddings, dim, &range, &nblk, &KC, &nchunk, &DT);
  return (long long)nblk * num_embeddings * (dim + 1);
}

extern "C" int gm_vq_ema_stats(const void* x, long long x_ld, const long long* indices, long long tokens, int num_embeddings, int dim,
                               float* stats, float* workspace, int dtype, void* stream) {
  GM_REQUIRE(stats && workspace && (tokens == 0 || (x && indices)), "null pointer");  // (no tokens: an empty tensor has no storage; zeros are written)
  GM_REQUIRE(num_embeddings > 0 && dim > 0 && tokens >= 0, "bad codebook shape");
  GM_REQUIRE((long long)(dim + 1) * 4 <= 48 * 1024, "embedding dim too large for the per-code LDS accumulators");
  hipStream_t st = (hipStream_t)stream;
  long long range; int nblk, KC, nchunk, DT;
  vq_ema_plan(tokens, num_embeddings, dim, &range, &nblk, &KC, &nchunk, &DT);
  const size_t smem = (size_t)KC * (dim + 1) * sizeof(float);
  dim3 grid((unsigned)nblk, (unsigned)nchunk);
  if (dtype == GM_F32)
    vq_ema_partial_kernel<float><<<grid, VQ_EMA_THREADS, smem, st>>>((const float*)x, x_ld, indices, tokens, range, num_embeddings, dim, KC, DT, workspace);
  else if (dtype == GM_BF16)
    vq_ema_partial_kernel<bf16_raw><<<grid, VQ_EMA_THREADS, smem, st>>>((const bf16_raw*)x, x_ld, indices, tokens, range, num_embeddings, dim, KC, DT, workspace);
  else GM_FAIL(-2, "unsupported dtype");
  const long long n = (long long)num_embeddings * (dim + 1);
  long long g = (n + VQ_EMA_THREADS - 1) / VQ_EMA_THREADS;
  if (g > 1024) g = 1024;
  vq_ema_fold_kernel<<<(unsigned)g, VQ_EMA_THREADS, 0, st>>>(workspace, nblk, num_embeddings, dim, stats);
  GM_LAUNCH_CHECK();
}

__global__ __launch_bounds__(VQ_EMA_THREADS) void vq_ema_update_kernel(const float* __restrict__ stats, float* __restrict__ cluster, float* __restrict__ ema_w,
                                                                      float* __restrict__ embedding, int K, int D, float decay, float epsilon) {
  __shared__ float red[VQ_EMA_THREADS];
  const int t = threadIdx.x;
  const float om = 1.0f - decay;
  float part = 0.f;
  for (int e = t; e < K; e += VQ_EMA_THREADS) {
    const float c = cluster[e] * decay + stats[e] * om;   // ema_cluster_size.mul_(decay).add_(encodings_sum * (1 - decay))
    cluster[e] = c;
    part += c;
  }
  red[t] = part;
  __syncthreads();
  for (int s = VQ_EMA_THREADS / 2; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  const float n = red[0];
  for (long long i = t; i < (long long)K * D; i += VQ_EMA_THREADS) {
    const int e = (int)(i / D);
    const float weight = (cluster[e] + epsilon) / (n + K * epsilon) * n;   // Laplace smoothing of the cluster size
    const float w = ema_w[i] * decay + stats[K + i] * om;
    ema_w[i] = w;
    embedding[i] = w / weight;
  }
}

// stats: the (all-reduced) buffer of gm_vq_ema_stats; cluster [K], ema_w / embedding [K][D]: fp32, updated in place
extern "C" int gm_vq_ema_update(const float* stats, float* cluster, float* ema_w, float* embedding, int num_embeddings, int dim, float decay,
                                float epsilon, void* stream) {
  GM_REQUIRE(stats && cluster && ema_w && embedding, "null pointer");
  vq_ema_update_kernel<<<1, VQ_EMA_THREADS, 0, (hipStream_t)stream>>>(stats, cluster, ema_w, embedding, num_embeddings, dim, decay, epsilon);
  GM_LAUNCH_CHECK();
}
