// GroupNorm statistics for N[D]HWC activations + LayerNorm over token rows.
//
// Reference semantics: nn.GroupNorm(G, C, eps, affine=True) as used at networks/nets/diffusion_model_unet.py:623,643,
// 275,377,1854 and networks/nets/autoencoderkl.py:146,156,227,433,579 -- biased variance over (C/G)*V elements of one
// sample.  The normalisation itself is never a separate pass here: gm_gn_stats + gm_gn_finalize produce per-(n, c)
// fp32 `scale = rstd*gamma`, `shift = beta - mean*rstd*gamma`, which the consumer (conv / linear prologue, conv.hip)
// applies while it stages its input tile.  gm_gn_apply exists for callers without a fusable consumer.
//
// HBM-bound: one read of the tensor (C*V*elt bytes per sample).  Reductions: per-thread fp32 partials over <= 64 rows,
// wave/LDS tree in fp64, block partials merged in fp64 in a fixed order -> deterministic and fp32-parity safe.
#include <cstdlib>
#include "gm_common.h"

#define GM_STAT_SLOTS 64  // see conv_common.h
#define GN_THREADS 256
#define GN_ROWS_PER_THREAD 64

template <typename T, int VEC> struct VecLd;
template <> struct VecLd<float, 4> {
  static __device__ __forceinline__ void ld(const float* p, float* o) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
};
template <> struct VecLd<float, 1> {
  static __device__ __forceinline__ void ld(const float* p, float* o) { o[0] = *p; }
};
template <> struct VecLd<bf16_raw, 8> {
  static __device__ __forceinline__ void ld(const bf16_raw* p, float* o) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = __uint_as_float(w[i] << 16);
      o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
};
template <> struct VecLd<bf16_raw, 1> {
  static __device__ __forceinline__ void ld(const bf16_raw* p, float* o) { o[0] = bf16_to_f32(*p); }
};

// grid (nblk, N).  partial[(n*nblk + blk)*G + g] = {sum, sumsq} (fp64) over rows [blk*rpb, (blk+1)*rpb) of sample n.
template <typename T, int VEC>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const T* __restrict__ x, long long ld, long long V, int C,
                                                             int G, int rows_per_block, double2* __restrict__ partial,
                                                             double* __restrict__ chan_out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int CV = C / VEC;              // channel vectors per row (<= GN_THREADS, checked on the host)
  const int R = GN_THREADS / CV;       // rows in flight
  float* part_s = reinterpret_cast<float*>(smem_raw);           // [R][C]
  float* part_q = part_s + (size_t)R * C;                       // [R][C]
  double* ch_s = reinterpret_cast<double*>(part_q + (size_t)R * C);  // [C]
  double* ch_q = ch_s + C;                                      // [C]

  const int n = blockIdx.y, blk = blockIdx.x;
  const int t = threadIdx.x;
  const int cv = t % CV, r0 = t / CV;
  const long long row_begin = (long long)blk * rows_per_block;
  long long row_end = row_begin + rows_per_block;
  if (row_end > V) row_end = V;

  float s[VEC], q[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s[i] = 0.f; q[i] = 0.f; }
  if (r0 < R) {
    const T* base = x + ((long long)n * V) * ld + (long long)cv * VEC;
    // eight rows requested per wait (clamped addresses, masked sums: same order of additions as a row-by-row walk).  The walk used to wait for
    // every row in turn -- 64 dependent round trips per block: 20 us for ANY tensor (rocprofv3, C3 latent UNet / C4 training step)
    constexpr int UB = 8;
    for (long long r = row_begin + r0; r < row_end; r += (long long)UB * R) {
      float v[UB][VEC];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const long long rr = r + (long long)u * R;
        VecLd<T, VEC>::ld(base + (rr < row_end ? rr : r) * ld, v[u]);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const bool ok = r + (long long)u * R < row_end;
#pragma unroll
        for (int i = 0; i < VEC; ++i) { s[i] += ok ? v[u][i] : 0.f; q[i] += ok ? v[u][i] * v[u][i] : 0.f; }
      }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      part_s[(size_t)r0 * C + cv * VEC + i] = s[i];
      part_q[(size_t)r0 * C + cv * VEC + i] = q[i];
    }
  }
  __syncthreads();
  for (int c = t; c < C; c += GN_THREADS) {
    double a = 0.0, b = 0.0;
    for (int r = 0; r < R; ++r) { a += (double)part_s[(size_t)r * C + c]; b += (double)part_q[(size_t)r * C + c]; }
    ch_s[c] = a; ch_q[c] = b;
    if (chan_out) {  // per-channel mode: one partial per block, [gridDim.x][N][C][2], plain stores (composable over channel concatenations;
      double* dst = chan_out + (((long long)blk * gridDim.y + n) * C + c) * 2;  // the consumers add the partials in a fixed order)
      *reinterpret_cast<double2*>(dst) = make_double2(a, b);
    }
  }
  if (!partial) return;
  __syncthreads();
  const int cpg = C / G;
  for (int g = t; g < G; g += GN_THREADS) {
    double a = 0.0, b = 0.0;
    for (int j = 0; j < cpg; ++j) { a += ch_s[g * cpg + j]; b += ch_q[g * cpg + j]; }
    partial[((long long)n * gridDim.x + blk) * G + g] = make_double2(a, b);
  }
}

// grid N*G blocks of one wave.  Merges the block partials, writes mean/rstd (optional) and per-channel scale/shift.
__global__ __launch_bounds__(64) void gn_finalize_kernel(const double2* __restrict__ partial, int nblk, int C, int G,
                                                        long long V, float eps, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ scale,
                                                        float* __restrict__ shift, float* __restrict__ mean_out,
                                                        float* __restrict__ rstd_out) {
  const int n = blockIdx.x / G, g = blockIdx.x % G;
  const int lane = threadIdx.x;
  double a = 0.0, b = 0.0;
  for (int i = lane; i < nblk; i += 64) {
    const double2 p = partial[((long long)n * nblk + i) * G + g];
    a += p.x; b += p.y;
  }
  a = wave_sum(a); b = wave_sum(b);
  const int cpg = C / G;
  const double cnt = (double)cpg * (double)V;
  const double mean = a / cnt;
  double var = b / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  if (lane == 0) {
    if (mean_out) mean_out[n * G + g] = (float)mean;
    if (rstd_out) rstd_out[n * G + g] = (float)rstd;
  }
  for (int j = lane; j < cpg; j += 64) {
    const int c = g * cpg + j;
    const double ga = gamma ? (double)gamma[c] : 1.0;
    const double be = beta ? (double)beta[c] : 0.0;
    scale[(long long)n * C + c] = (float)(rstd * ga);
    shift[(long long)n * C + c] = (float)(be - mean * rstd * ga);
  }
}

// rows per block of the statistics pass: 8 .. GN_ROWS_PER_THREAD rows per lane (a multiple of 8: the walk requests 8 rows per wait), as few as
// still give ~1024 blocks -- a 32^3 tensor got 16 blocks of 2048 rows with the fixed 64 rows per lane
static long long gn_rows_per_block(long long V, int R) {
  long long rpt = (V + (long long)R * 1024 - 1) / ((long long)R * 1024);
  rpt = (rpt + 7) / 8 * 8;
  if (rpt < 8) rpt = 8;
  if (rpt > GN_ROWS_PER_THREAD) rpt = GN_ROWS_PER_THREAD;
  return (long long)R * rpt;
}
static long long gn_nblk(long long V, int C, int vec) {
  const int CV = C / vec;
  const int R = GN_THREADS / CV;
  const long long rpb = gn_rows_per_block(V, R);
  return (V + rpb - 1) / rpb;
}

// Upper bound over both load widths the dispatcher may pick (vector when C, ld and the base are 16-byte friendly).
extern "C" long long gm_gn_workspace_bytes(int N, long long V, int C, int G, int dtype) {
  const int vecmax = dtype == GM_F32 ? 4 : 8;
  long long nblk = 0;
  if (C % vecmax == 0 && C / vecmax <= GN_THREADS) nblk = gn_nblk(V, C, vecmax);
  if (C <= GN_THREADS) {
    const long long nb1 = gn_nblk(V, C, 1);
    if (nb1 > nblk) nblk = nb1;
  }
  if (nblk == 0) return -1;
  return (long long)N * nblk * G * (long long)sizeof(double2);
}

template <typename T, int VEC>
static int launch_gn_stats(const void* x, long long ld, int N, long long V, int C, int G, double2* ws, int* nblk_out,
                           hipStream_t st, double* chan_out = nullptr) {
  const int CV = C / VEC;
  const int R = GN_THREADS / CV;
  const int rpb = (int)gn_rows_per_block(V, R);
  const int nblk = gm_cdiv(V, rpb);
  const size_t smem = (size_t)R * C * 2 * sizeof(float) + (size_t)C * 2 * sizeof(double);
  *nblk_out = nblk;
  dim3 grid(nblk, N);
  gn_stats_kernel<T, VEC><<<grid, GN_THREADS, smem, st>>>((const T*)x, ld, V, C, G, rpb, ws, chan_out);
  return 0;
}

// x: [N][V][ld] (C <= ld channels used), gamma/beta: fp32 [C] (nullable), scale/shift: fp32 [N][C] outputs,
// mean/rstd: optional fp32 [N][G] outputs, workspace: >= gm_gn_workspace_bytes(...) bytes.
extern "C" int gm_gn_scale_shift(const void* x, long long ld, int N, long long V, int C, int G, float eps,
                                 const float* gamma, const float* beta, float* scale, float* shift, float* mean,
                                 float* rstd, void* workspace, int dtype, void* stream) {
  GM_REQUIRE(x && scale && shift && workspace, "null pointer");
  GM_REQUIRE(G > 0 && C % G == 0, "channels must be divisible by groups");
  GM_REQUIRE(N <= 65535, "batch too large");
  if (N == 0 || V == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int vecmax = dtype == GM_F32 ? 4 : 8;
  const bool vec_ok = (C % vecmax == 0) && (ld % vecmax == 0) && (((uintptr_t)x & 15) == 0);
  const int CV = vec_ok ? C / vecmax : C;
  GM_REQUIRE(CV <= GN_THREADS, "too many channels for gm_gn_scale_shift");
  int nblk = 0;
  double2* ws = (double2*)workspace;
  if (dtype == GM_F32) {
    if (vec_ok) launch_gn_stats<float, 4>(x, ld, N, V, C, G, ws, &nblk, st);
    else launch_gn_stats<float, 1>(x, ld, N, V, C, G, ws, &nblk, st);
  } else if (dtype == GM_BF16) {
    if (vec_ok) launch_gn_stats<bf16_raw, 8>(x, ld, N, V, C, G, ws, &nblk, st);
    else launch_gn_stats<bf16_raw, 1>(x, ld, N, V, C, G, ws, &nblk, st);
  } else {
    GM_FAIL(-2, "unsupported dtype");
  }
  {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) GM_FAIL((int)e, hipGetErrorString(e));
  }
  gn_finalize_kernel<<<N * G, 64, 0, st>>>(ws, nblk, C, G, V, eps, gamma, beta, scale, shift, mean, rstd);
  GM_LAUNCH_CHECK();
}

// Slots S of the [S][N][C][2] table gm_gn_channel_stats writes for this tensor (one partial per block of rows; depends on the load width
// the dispatcher picks, hence the pointer / pitch arguments).  -1: channel count not covered.
extern "C" long long gm_gn_channel_stats_slots(const void* x, long long ld, long long V, int C, int dtype) {
  const int vecmax = dtype == GM_F32 ? 4 : 8;
  const bool vec_ok = (C % vecmax == 0) && (ld % vecmax == 0) && (((uintptr_t)x & 15) == 0);
  if ((vec_ok ? C / vecmax : C) > GN_THREADS) return -1;
  return V == 0 ? 1 : gn_nblk(V, C, vec_ok ? vecmax : 1);
}

// Per-channel statistics: chan_out[slot][n][c] = {sum, sum of squares} over the rows of block `slot` (every entry of the
// [gm_gn_channel_stats_slots][N][C][2] table is written exactly once: no zero fill, no atomics).
extern "C" int gm_gn_channel_stats(const void* x, long long ld, int N, long long V, int C, double* chan_out, int dtype, void* stream) {
  GM_REQUIRE(x && chan_out, "null pointer");
  GM_REQUIRE(N <= 65535, "batch too large");
  if (N == 0 || V == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int vecmax = dtype == GM_F32 ? 4 : 8;
  const bool vec_ok = (C % vecmax == 0) && (ld % vecmax == 0) && (((uintptr_t)x & 15) == 0);
  GM_REQUIRE((vec_ok ? C / vecmax : C) <= GN_THREADS, "too many channels for gm_gn_channel_stats");
  int nblk = 0;
  if (dtype == GM_F32) {
    if (vec_ok) launch_gn_stats<float, 4>(x, ld, N, V, C, 1, nullptr, &nblk, st, chan_out);
    else launch_gn_stats<float, 1>(x, ld, N, V, C, 1, nullptr, &nblk, st, chan_out);
  } else if (dtype == GM_BF16) {
    if (vec_ok) launch_gn_stats<bf16_raw, 8>(x, ld, N, V, C, 1, nullptr, &nblk, st, chan_out);
    else launch_gn_stats<bf16_raw, 1>(x, ld, N, V, C, 1, nullptr, &nblk, st, chan_out);
  } else {
    GM_FAIL(-2, "unsupported dtype");
  }
  GM_LAUNCH_CHECK();
}

// Fold a long table of partials [S][N][C][2] (one per 256-voxel tile: 8192 at 128^3) to GM_COMPACT_SLOTS = 256 rows in a fixed order:
// output row b = sum of input rows b, b + 256, b + 512, ... -- one block per output row, so every CU takes part (round 2 folded to 64 rows with
// 64 blocks: 11 us per table, 24 tables per C2 forward; the per-norm consumers read 256 rows as fast as 64: one row per thread).
#define GM_COMPACT_SLOTS 256
__global__ __launch_bounds__(256) void stats_compact_kernel(const double* __restrict__ in, int S, long long row_elems, double* __restrict__ out) {
  const int b = blockIdx.x;
  for (long long e = threadIdx.x; e < row_elems; e += 256) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;  // four loads in flight; combined in a fixed order below
    int sl = b;
    for (; sl + 3 * GM_COMPACT_SLOTS < S; sl += 4 * GM_COMPACT_SLOTS) {
      a0 += in[(long long)sl * row_elems + e];
      a1 += in[(long long)(sl + GM_COMPACT_SLOTS) * row_elems + e];
      a2 += in[(long long)(sl + 2 * GM_COMPACT_SLOTS) * row_elems + e];
      a3 += in[(long long)(sl + 3 * GM_COMPACT_SLOTS) * row_elems + e];
    }
    for (; sl < S; sl += GM_COMPACT_SLOTS) a0 += in[(long long)sl * row_elems + e];
    out[(long long)b * row_elems + e] = (a0 + a1) + (a2 + a3);
  }
}

extern "C" int gm_stats_compact_slots() { return GM_COMPACT_SLOTS; }

// stats_in: [S][N][C][2] with S >= gm_stats_compact_slots() -> stats_out: [gm_stats_compact_slots()][N][C][2] (every entry written)
extern "C" int gm_stats_compact(const double* stats_in, int S, int N, int C, double* stats_out, void* stream) {
  GM_REQUIRE(stats_in && stats_out && S >= GM_COMPACT_SLOTS, "gm_stats_compact folds tables of at least gm_stats_compact_slots() partials");
  if (N == 0 || C == 0) return 0;
  stats_compact_kernel<<<GM_COMPACT_SLOTS, 256, 0, (hipStream_t)stream>>>(stats_in, S, (long long)N * C * 2, stats_out);
  GM_LAUNCH_CHECK();
}

// GroupNorm scale/shift from per-channel statistic partials of up to two channel-concatenated sources (C = C0 + C1): source i holds
// S_i partials [S_i][N][C_i][2].  One 256-thread block per (n, group); every thread adds its strided share of the partials in index order,
// then lanes and waves are combined by a fixed butterfly / a fixed 4-term sum: the result depends on the table contents only.
__global__ __launch_bounds__(256) void gn_finalize_channels_kernel(const double* __restrict__ s0, int S0, int C0, const double* __restrict__ s1,
                                                                  int S1, int C1, int N, int G, long long V, float eps,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  float* __restrict__ scale, float* __restrict__ shift) {
  __shared__ double red[4][2];
  const int n = blockIdx.x / G, g = blockIdx.x % G;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int C = C0 + C1, cpg = C / G;
  // gamma / beta of this thread's first output channel: requested BEFORE the table (round 6: behind the reduction they were a second dependent
  // round trip of every one of the 36-46 finalisations of a forward)
  float ga0 = 1.f, be0 = 0.f;
  if (t < cpg) {
    if (gamma) ga0 = gamma[g * cpg + t];
    if (beta) be0 = beta[g * cpg + t];
  }
  double a = 0.0, b = 0.0;
  for (int j = 0; j < cpg; ++j) {  // a group may straddle the two sources: per channel, then per partial
    const int c = g * cpg + j;
    const bool first = c < C0;
    const double* src = first ? s0 + ((long long)n * C0 + c) * 2 : s1 + ((long long)n * C1 + (c - C0)) * 2;
    const long long pitch = (long long)N * (first ? C0 : C1) * 2;
    const int S = first ? S0 : S1;
    for (int sl = t; sl < S; sl += 256) {
      const double2 v = *reinterpret_cast<const double2*>(src + sl * pitch);
      a += v.x; b += v.y;
    }
  }
  a = wave_sum(a); b = wave_sum(b);
  if (lane == 0) { red[wave][0] = a; red[wave][1] = b; }
  __syncthreads();
  a = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
  b = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
  const double cnt = (double)cpg * (double)V;
  const double mean = a / cnt;
  double var = b / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  for (int j = t; j < cpg; j += 256) {
    const int c = g * cpg + j;
    const double ga = j == t ? (double)ga0 : (gamma ? (double)gamma[c] : 1.0);
    const double be = j == t ? (double)be0 : (beta ? (double)beta[c] : 0.0);
    scale[(long long)n * C + c] = (float)(rstd * ga);
    shift[(long long)n * C + c] = (float)(be - mean * rstd * ga);
  }
}

// SHORT tables (every source S <= GN_SHORT_MAX_ROWS): a thread owns a channel of the group and adds its rows in row order; thread 0 adds the channels in channel order -- the
// order conv_sn.hip's consumer-side prologue reproduces (gm_common.h: gn_short_*), so that a norm finalised there and one finalised here agree bit for bit.
__global__ __launch_bounds__(256) void gn_finalize_channels_short_kernel(const double* __restrict__ s0, int S0, int C0, const double* __restrict__ s1, int S1, int C1,
                                                                        int N, int G, long long V, float eps, const float* __restrict__ gamma,
                                                                        const float* __restrict__ beta, float* __restrict__ scale, float* __restrict__ shift,
                                                                        int staged_rows) {
  extern __shared__ double gn_short_sums[];  // [cpg][2], then the group's pair; staged form: behind them [rows][cpg][2]
  const int n = blockIdx.x / G, g = blockIdx.x % G, t = threadIdx.x;
  const int C = C0 + C1, cpg = C / G;
  if (staged_rows > 0) {
    // every (row, channel) partial of the group is requested at once by the whole block -- ONE round trip instead of S / 8 per channel thread -- and a thread then adds its
    // channel's rows from LDS in the same row order (fp64): the same sums, bit for bit, as gn_short_channel_sum (8-row batches walked by one thread: 9 us at 128 rows)
    double* rows = gn_short_sums + 2 * cpg + 2;
    for (int it0 = 0; it0 < staged_rows * cpg; it0 += 256 * 8) {  // eight requests per thread before the first wait
      double2 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int it = it0 + k * 256 + t;
        const int j = it % cpg, r = it / cpg, c = g * cpg + j;
        const bool first = c < C0;
        const bool ok = it < staged_rows * cpg && r < (first ? S0 : S1);
        const double* src = first ? s0 + ((long long)n * C0 + c) * 2 : s1 + ((long long)n * C1 + (c - C0)) * 2;
        v[k] = ok ? *reinterpret_cast<const double2*>(src + (long long)r * ((long long)N * (first ? C0 : C1) * 2)) : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int it = it0 + k * 256 + t;
        if (it < staged_rows * cpg) { rows[2 * it] = v[k].x; rows[2 * it + 1] = v[k].y; }
      }
    }
    __syncthreads();
    for (int j = t; j < cpg; j += 256) {
      const int S = (g * cpg + j) < C0 ? S0 : S1;
      double a = 0.0, b = 0.0;
      for (int r = 0; r < S; ++r) { a += rows[2 * (r * cpg + j)]; b += rows[2 * (r * cpg + j) + 1]; }
      gn_short_sums[2 * j] = a; gn_short_sums[2 * j + 1] = b;
    }
  } else {
    for (int j = t; j < cpg; j += 256) {
      const double2 v = gn_short_channel_sum(s0, S0, C0, s1, S1, C1, N, n, g * cpg + j);
      gn_short_sums[2 * j] = v.x; gn_short_sums[2 * j + 1] = v.y;
    }
  }
  __syncthreads();
  if (t == 0) {
    double a = 0.0, b = 0.0;
    for (int j = 0; j < cpg; ++j) { a += gn_short_sums[2 * j]; b += gn_short_sums[2 * j + 1]; }
    gn_short_sums[2 * cpg] = a; gn_short_sums[2 * cpg + 1] = b;
  }
  __syncthreads();
  const double a = gn_short_sums[2 * cpg], b = gn_short_sums[2 * cpg + 1];
  for (int j = t; j < cpg; j += 256) {
    const int c = g * cpg + j;
    float sc, sh;
    gn_short_scale_shift(a, b, cpg, V, eps, gamma ? gamma[c] : 1.f, beta ? beta[c] : 0.f, sc, sh);
    scale[(long long)n * C + c] = sc;
    shift[(long long)n * C + c] = sh;
  }
}

extern "C" int gm_gn_finalize_channels(const double* stats0, int S0, int C0, const double* stats1, int S1, int C1, int N, long long V, int G,
                                       float eps, const float* gamma, const float* beta, float* scale, float* shift, void* stream) {
  GM_REQUIRE(stats0 && scale && shift, "null pointer");
  GM_REQUIRE(C1 == 0 || stats1, "second source without statistics");
  GM_REQUIRE(S0 > 0 && (C1 == 0 || S1 > 0), "statistic tables need at least one partial");
  GM_REQUIRE(G > 0 && (C0 + C1) % G == 0, "channels must be divisible by groups");
  if (N == 0) return 0;
  const int cpg = (C0 + C1) / G;
  if (S0 <= GN_SHORT_MAX_ROWS && (C1 == 0 || S1 <= GN_SHORT_MAX_ROWS) && cpg <= 4096) {  // short tables: the order the consumer-side finalisation shares (ops.GnRecipe)
    const int smax = C1 && S1 > S0 ? S1 : S0;
    const size_t staged = (size_t)smax * cpg * 2 * sizeof(double);
    const int staged_rows = staged <= 48 * 1024 ? smax : 0;  // (else: a thread walks its channel's rows itself -- the same sums)
    gn_finalize_channels_short_kernel<<<N * G, 256, (size_t)(2 * cpg + 2) * sizeof(double) + (staged_rows ? staged : 0), (hipStream_t)stream>>>(
        stats0, S0, C0, stats1, S1, C1, N, G, V, eps, gamma, beta, scale, shift, staged_rows);
  }
  else
    gn_finalize_channels_kernel<<<N * G, 256, 0, (hipStream_t)stream>>>(stats0, S0, C0, stats1, S1, C1, N, G, V, eps, gamma, beta, scale, shift);
  GM_LAUNCH_CHECK();
}

// y[n, v, c] = act(x[n, v, c] * scale[n, c] + shift[n, c]);  act: 0 none, 1 SiLU, 2 ReLU
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, long long x_ld, T* __restrict__ y,
                                                      long long y_ld, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, long long ss_ld, long long V, int C,
                                                      long long total, int act) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / C;
    const int c = (int)(i - row * C);
    const long long n = row / V;
    float v = ElemIO<T>::ld(x + row * x_ld + c) * scale[n * ss_ld + c] + shift[n * ss_ld + c];
    if (act == 1) v = gm_silu_precise(v);
    else if (act == 2) v = fmaxf(v, 0.f);
    ElemIO<T>::st(y + row * y_ld + c, v);
  }
}

// 16-byte vector variant: one lane = 8 bf16 / 4 fp32 consecutive channels of one voxel (HBM-bound: read + write once).
// Measured 5.0 TB/s on 268 MB tensors; a 4x-unrolled variant with four loads in flight per lane measured the same, i.e. the mixed
// read + write stream is at what HBM sustains, not latency-bound.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void gn_apply_vec_kernel(const T* __restrict__ x, long long x_ld, T* __restrict__ y,
                                                          long long y_ld, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, long long ss_ld, long long V, int C,
                                                          long long total_vec, int act) {
  const int CV = C / VEC;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total_vec; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / CV;
    const int c = (int)(i - row * CV) * VEC;
    const long long n = row / V;
    float v[VEC], sc[VEC], sh[VEC];
    VecLd<T, VEC>::ld(x + row * x_ld + c, v);
    // scale / shift as 16-byte vectors (c, ss_ld multiples of 4 floats: checked by the launcher), the activation chosen OUTSIDE the
    // element loop: a per-element `if (act == ...)` chain made hipcc split the loop into eight blocks with scalar loads in each
    // (measured: 1.6 TB/s instead of 4.9)
#pragma unroll
    for (int k = 0; k < VEC; k += 4) {
      const float4 a = *reinterpret_cast<const float4*>(scale + n * ss_ld + c + k), b = *reinterpret_cast<const float4*>(shift + n * ss_ld + c + k);
      sc[k] = a.x; sc[k + 1] = a.y; sc[k + 2] = a.z; sc[k + 3] = a.w;
      sh[k] = b.x; sh[k + 1] = b.y; sh[k + 2] = b.z; sh[k + 3] = b.w;
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = v[k] * sc[k] + sh[k];
    if (act == 1) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) v[k] = sizeof(T) == 4 ? gm_silu_precise(v[k]) : gm_silu(v[k]);
    } else if (act == 2) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    if (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(y + row * y_ld + c) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      *reinterpret_cast<uint4*>(y + row * y_ld + c) =
          make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4 % VEC], v[5 % VEC]), pack_bf16x2(v[6 % VEC], v[7 % VEC]));
    }
  }
}

// The same pass with the lane <-> channel-vector assignment fixed (round 3): a block walks `rows_per_block` rows of one sample, thread t owns
// channel vector t % CV of rows r0 + k * R, so its scale / shift vectors are loaded ONCE and the addresses are 32-bit increments.  The
// grid-stride form above spends two 64-bit divisions and four 16-byte table loads per 16-byte vector -- as many VALU slots as the SiLU (1.25 T
// elements/s x ~26 lane-operations is most of the chip's 31 T lane-operations/s): 5.0 -> 5.3-5.5 TB/s with them gone.  Four rows are requested
// per wait.  grid (nblk, N); host: CV <= 256.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void gn_apply_rows_kernel(const T* __restrict__ x, long long x_ld, T* __restrict__ y, long long y_ld,
                                                           const float* __restrict__ scale, const float* __restrict__ shift, long long ss_ld,
                                                           long long V, int C, int rows_per_block, int act) {
  const int CV = C / VEC, R = 256 / CV;
  const int n = blockIdx.y, t = threadIdx.x;
  const int cv = t % CV, r0 = t / CV;
  if (r0 >= R) return;
  const long long row_begin = (long long)blockIdx.x * rows_per_block;
  long long row_end = row_begin + rows_per_block;
  if (row_end > V) row_end = V;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int k = 0; k < VEC; k += 4) {
    const float4 a = *reinterpret_cast<const float4*>(scale + n * ss_ld + cv * VEC + k), b = *reinterpret_cast<const float4*>(shift + n * ss_ld + cv * VEC + k);
    sc[k] = a.x; sc[k + 1] = a.y; sc[k + 2] = a.z; sc[k + 3] = a.w;
    sh[k] = b.x; sh[k + 1] = b.y; sh[k + 2] = b.z; sh[k + 3] = b.w;
  }
  const T* xb = x + ((long long)n * V) * x_ld + (long long)cv * VEC;
  T* yb = y + ((long long)n * V) * y_ld + (long long)cv * VEC;
  constexpr int UB = 4;
  for (long long r = row_begin + r0; r < row_end; r += (long long)UB * R) {
    float v[UB][VEC];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const long long rr = r + (long long)u * R < row_end ? r + (long long)u * R : r;
      VecLd<T, VEC>::ld(xb + rr * x_ld, v[u]);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) v[u][k] = v[u][k] * sc[k] + sh[k];
      if (act == 1) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[u][k] = sizeof(T) == 4 ? gm_silu_precise(v[u][k]) : gm_silu(v[u][k]);
      } else if (act == 2) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[u][k] = fmaxf(v[u][k], 0.f);
      }
      const long long rr = r + (long long)u * R;
      if (rr < row_end) {
        if (sizeof(T) == 4) {
          *reinterpret_cast<float4*>(yb + rr * y_ld) = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
        } else {
          *reinterpret_cast<uint4*>(yb + rr * y_ld) = make_uint4(pack_bf16x2(v[u][0], v[u][1]), pack_bf16x2(v[u][2], v[u][3]),
                                                                 pack_bf16x2(v[u][4 % VEC], v[u][5 % VEC]), pack_bf16x2(v[u][6 % VEC], v[u][7 % VEC]));
        }
      }
    }
  }
}

// scale/shift rows are ss_ld floats apart (ss_ld >= C: a channel slice of a wider [N][C_total] table is a valid operand)
extern "C" int gm_gn_apply(const void* x, long long x_ld, void* y, long long y_ld, const float* scale, const float* shift,
                           long long ss_ld, int N, long long V, int C, int act, int dtype, void* stream) {
  GM_REQUIRE(x && y && scale && shift, "null pointer");
  const long long total = (long long)N * V * C;
  if (total == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int vec = dtype == GM_F32 ? 4 : 8;
  const bool vec_ok = (C % vec == 0) && (x_ld % vec == 0) && (y_ld % vec == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)y & 15) == 0) &&
                      (ss_ld % 4 == 0) && (((uintptr_t)scale & 15) == 0) && (((uintptr_t)shift & 15) == 0);
  static const bool rows_form = !(getenv("GM_GN_APPLY_ROWS") && getenv("GM_GN_APPLY_ROWS")[0] == '0');  // bench switch (tools/layer_times.py)
  if (rows_form && vec_ok && (dtype == GM_F32 || dtype == GM_BF16) && C / vec <= 256 && N <= 65535) {
    const int R = 256 / (C / vec);
    long long iters = (V + (long long)R * 2048 - 1) / ((long long)R * 2048);  // rows per lane: 4 .. 16, about 2048+ blocks per sample
    iters = (iters + 3) / 4 * 4;
    if (iters < 4) iters = 4;
    if (iters > 16) iters = 16;
    const int rpb = (int)(R * iters);
    dim3 grid((unsigned)((V + rpb - 1) / rpb), (unsigned)N);
    if (dtype == GM_F32)
      gn_apply_rows_kernel<float, 4><<<grid, 256, 0, st>>>((const float*)x, x_ld, (float*)y, y_ld, scale, shift, ss_ld, V, C, rpb, act);
    else
      gn_apply_rows_kernel<bf16_raw, 8><<<grid, 256, 0, st>>>((const bf16_raw*)x, x_ld, (bf16_raw*)y, y_ld, scale, shift, ss_ld, V, C, rpb, act);
    GM_LAUNCH_CHECK();
  }
  if (vec_ok && (dtype == GM_F32 || dtype == GM_BF16)) {
    const long long tv = total / vec;
    long long g = (tv + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (dtype == GM_F32)
      gn_apply_vec_kernel<float, 4><<<(int)g, 256, 0, st>>>((const float*)x, x_ld, (float*)y, y_ld, scale, shift, ss_ld, V, C, tv, act);
    else
      gn_apply_vec_kernel<bf16_raw, 8><<<(int)g, 256, 0, st>>>((const bf16_raw*)x, x_ld, (bf16_raw*)y, y_ld, scale, shift, ss_ld, V, C, tv, act);
    GM_LAUNCH_CHECK();
  }
  long long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  if (dtype == GM_F32)
    gn_apply_kernel<float><<<(int)g, 256, 0, st>>>((const float*)x, x_ld, (float*)y, y_ld, scale, shift, ss_ld, V, C, total, act);
  else if (dtype == GM_BF16)
    gn_apply_kernel<bf16_raw><<<(int)g, 256, 0, st>>>((const bf16_raw*)x, x_ld, (bf16_raw*)y, y_ld, scale, shift, ss_ld, V, C, total, act);
  else GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// SPADE modulation (reference: generative/networks/blocks/spade_norm.py:79-96): y = act((x * scale[n, c] + shift[n, c]) * G[n, v, c] + Bm[n, v, c])
// with (scale, shift) the parameter-free GroupNorm / InstanceNorm of x and G = 1 + gamma(seg), Bm = beta(seg) per-voxel maps (constant
// over the timesteps of a sampling chain: computed once per layer and segmentation, read here every step).  One 16-byte vector per lane.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void spade_apply_kernel(const T* __restrict__ x, long long x_ld, T* __restrict__ y, long long y_ld,
                                                         const float* __restrict__ scale, const float* __restrict__ shift, long long ss_ld,
                                                         const T* __restrict__ g, const T* __restrict__ bm, long long gb_ld, long long V, int C,
                                                         long long total_vec, int act) {
  const int CV = C / VEC;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total_vec; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / CV;
    const int c = (int)(i - row * CV) * VEC;
    const long long n = row / V;
    float v[VEC], gv[VEC], bv[VEC];
    VecLd<T, VEC>::ld(x + row * x_ld + c, v);
    VecLd<T, VEC>::ld(g + row * gb_ld + c, gv);
    VecLd<T, VEC>::ld(bm + row * gb_ld + c, bv);
    const float* sc = scale + n * ss_ld + c;
    const float* sh = shift + n * ss_ld + c;
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = (v[k] * sc[k] + sh[k]) * gv[k] + bv[k];
    if (act == 1) {  // (tested once per vector: a per-element test compiles to one basic block per element, every SiLU's v_exp -> v_rcp chain exposed)
#pragma unroll
      for (int k = 0; k < VEC; ++k) v[k] = sizeof(T) == 4 ? gm_silu_precise(v[k]) : gm_silu(v[k]);
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) ElemIO<T>::st(y + row * y_ld + c + k, v[k]);
  }
}

extern "C" int gm_spade_apply(const void* x, long long x_ld, void* y, long long y_ld, const float* scale, const float* shift, long long ss_ld,
                              const void* g, const void* bm, long long gb_ld, int N, long long V, int C, int act, int dtype, void* stream) {
  GM_REQUIRE(x && y && scale && shift && g && bm, "null pointer");
  const long long total = (long long)N * V * C;
  if (total == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int vec = dtype == GM_F32 ? 4 : 8;
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool vec_ok = (C % vec == 0) && (x_ld % vec == 0) && (gb_ld % vec == 0) && al(x) && al(g) && al(bm);
  const long long items = vec_ok ? total / vec : total;
  long long grid = (items + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
#define GM_SPADE_LAUNCH(T, VEC)                                                                                                        \
  spade_apply_kernel<T, VEC><<<(int)grid, 256, 0, st>>>((const T*)x, x_ld, (T*)y, y_ld, scale, shift, ss_ld, (const T*)g, (const T*)bm, gb_ld, V, C, \
                                                     items, act)
  if (dtype == GM_F32) { if (vec_ok) GM_SPADE_LAUNCH(float, 4); else GM_SPADE_LAUNCH(float, 1); }
  else if (dtype == GM_BF16) { if (vec_ok) GM_SPADE_LAUNCH(bf16_raw, 8); else GM_SPADE_LAUNCH(bf16_raw, 1); }
  else GM_FAIL(-2, "unsupported dtype");
#undef GM_SPADE_LAUNCH
  GM_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm over the last dim of [rows][ld] (transformer blocks: diffusion_model_unet.py:219-223): one wave per row.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, long long x_ld, T* __restrict__ y,
                                                       long long y_ld, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, long long rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * x_ld;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += ElemIO<T>::ld(xr + c);
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = ElemIO<T>::ld(xr + c) - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
  T* yr = y + row * y_ld;
  for (int c = lane; c < C; c += 64) {
    const float v = (ElemIO<T>::ld(xr + c) - mean) * rstd;
    ElemIO<T>::st(yr + c, v * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f));
  }
}

extern "C" int gm_layernorm(const void* x, long long x_ld, void* y, long long y_ld, const float* gamma, const float* beta,
                            long long rows, int C, float eps, int dtype, void* stream) {
  GM_REQUIRE(x && y, "null pointer");
  if (rows == 0 || C == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int grid = gm_cdiv(rows, 4);
  if (dtype == GM_F32)
    layernorm_kernel<float><<<grid, 256, 0, st>>>((const float*)x, x_ld, (float*)y, y_ld, gamma, beta, rows, C, eps);
  else if (dtype == GM_BF16)
    layernorm_kernel<bf16_raw><<<grid, 256, 0, st>>>((const bf16_raw*)x, x_ld, (bf16_raw*)y, y_ld, gamma, beta, rows, C, eps);
  else GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}
