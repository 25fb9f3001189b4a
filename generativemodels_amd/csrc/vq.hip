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
//                     all-reduce instead of the reference's two (:155-157).  One work-group per code scans the indices; per-thread
//                     partial sums are merged in a fixed tree order: deterministic.
//   gm_vq_ema_update: ema_cluster_size = decay * ema_cluster_size + (1 - decay) * counts; Laplace-smoothed weights; ema_w likewise;
//                     embedding = ema_w / weights (:173-180).  One work-group (the codebook is K x D <= a few hundred thousand floats).
// ---------------------------------------------------------------------------------------------------------------------
#define VQ_EMA_THREADS 256
#define VQ_EMA_DCHUNK 32
template <typename T>
__global__ __launch_bounds__(VQ_EMA_THREADS) void vq_ema_stats_kernel(const T* __restrict__ x, long long x_ld, const long long* __restrict__ idx,
                                                                     long long tokens, int K, int D, float* __restrict__ stats) {
  __shared__ float red[VQ_EMA_THREADS];
  const int e = blockIdx.x, t = threadIdx.x;
  auto block_sum = [&](float v) {  // fixed-order tree over the 256 threads
    red[t] = v;
    __syncthreads();
    for (int s = VQ_EMA_THREADS / 2; s > 0; s >>= 1) {
      if (t < s) red[t] += red[t + s];
      __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
  };
  float cnt = 0.f;
  for (long long i = t; i < tokens; i += VQ_EMA_THREADS) cnt += idx[i] == e ? 1.f : 0.f;
  cnt = block_sum(cnt);
  if (t == 0) stats[e] = cnt;
  for (int d0 = 0; d0 < D; d0 += VQ_EMA_DCHUNK) {
    float acc[VQ_EMA_DCHUNK];
#pragma unroll
    for (int d = 0; d < VQ_EMA_DCHUNK; ++d) acc[d] = 0.f;
    if (cnt > 0.f) {  // block-uniform
      for (long long i = t; i < tokens; i += VQ_EMA_THREADS) {
        if (idx[i] == e) {
          const T* row = x + i * x_ld + d0;
#pragma unroll
          for (int d = 0; d < VQ_EMA_DCHUNK; ++d)
            if (d0 + d < D) acc[d] += ElemIO<T>::ld(row + d);
        }
      }
    }
#pragma unroll
    for (int d = 0; d < VQ_EMA_DCHUNK; ++d) {
      if (d0 + d < D) {  // block-uniform
        const float s = cnt > 0.f ? block_sum(acc[d]) : 0.f;
        if (t == 0) stats[K + (long long)e * D + d0 + d] = s;
      }
    }
  }
}

extern "C" int gm_vq_ema_stats(const void* x, long long x_ld, const long long* indices, long long tokens, int num_embeddings, int dim,
                               float* stats, int dtype, void* stream) {
  GM_REQUIRE(x && indices && stats, "null pointer");
  GM_REQUIRE(num_embeddings > 0 && dim > 0, "bad codebook shape");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GM_F32) vq_ema_stats_kernel<float><<<num_embeddings, VQ_EMA_THREADS, 0, st>>>((const float*)x, x_ld, indices, tokens, num_embeddings, dim, stats);
  else if (dtype == GM_BF16) vq_ema_stats_kernel<bf16_raw><<<num_embeddings, VQ_EMA_THREADS, 0, st>>>((const bf16_raw*)x, x_ld, indices, tokens, num_embeddings, dim, stats);
  else GM_FAIL(-2, "unsupported dtype");
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
