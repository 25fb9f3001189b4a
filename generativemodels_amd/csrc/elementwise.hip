// Element-wise / data-movement kernels of the sampling path (all HBM-bound; graded against the 8 TB/s roof).
//   gm_sched_step      fused DDIM / DDPM reverse step      (reference: networks/schedulers/ddim.py:156-237, ddpm.py:191-252)
//   gm_axpby_rows      add_noise / get_velocity            (reference: networks/schedulers/scheduler.py:169-200)
//   gm_lincomb         PNDM multi-step combinations        (reference: networks/schedulers/pndm.py:186-195,241-250)
//   gm_likelihood_term one term of get_likelihood's bound  (reference: inferers/inferer.py:203-256,281-321)
//   gm_cast, gm_copy_channels, gm_nchw_to_nhwc, gm_nhwc_to_nchw, gm_resample2x   layout plumbing of the NDHWC arena
//   gm_timestep_embedding                                   (reference: networks/nets/diffusion_model_unet.py:461-485)
//   gm_geglu                                                (MONAI MLPBlock act="GEGLU" as used at diffusion_model_unet.py:211)
//   gm_aekl_sample                                          (reference: networks/nets/autoencoderkl.py:731-753)
#include "gm_common.h"

// ---------------------------------------------------------------------------------------------------------------------
// Fused scheduler step.  All per-step scalars are computed on the host with the same fp32 torch-CPU expressions as the
// reference and passed by value; the kernel mirrors the reference's op order *without* fma contraction so that an fp32
// step is bit-identical to the reference CPU result.  bf16 tensors are computed in fp32 and rounded once.
// ---------------------------------------------------------------------------------------------------------------------
struct GmStepParams {
  int mode;       // 0 = DDIM, 1 = DDPM, 2 = PNDM transfer (formula (9): k0*x - (k1*e)/c_prev, e = model output or its v-prediction transform)
  int pred_type;  // 0 epsilon, 1 sample, 2 v_prediction
  float c_sa;     // alpha_prod_t ** 0.5
  float c_sb;     // beta_prod_t ** 0.5
  int clip;
  float clip_lo, clip_hi;
  float c_prev;   // DDIM: alpha_prod_t_prev ** 0.5
  float c_dir;    // DDIM: (1 - alpha_prod_t_prev - std_dev_t**2) ** 0.5
  float k0, k1;   // DDPM: pred_original_sample_coeff, current_sample_coeff
  int noise_mode; // 0 none, 1 c_noise * noise, 2 learned: sqrt(pv) * noise, 3 learned_range
  float c_noise;  // DDIM: variance**0.5 * eta ; DDPM fixed_*: variance ** 0.5
  float min_log, max_log;  // learned_range
};

// BITS: the Gaussian noise of a bf16 chain comes as the CPU generator's byte draws + the pair table of host_noise.py (gm_normal_bf16_from_bits' lookup, here in the
// step's own kernel: element i of block b = i / 16 is the cos branch (i % 16 < 8) or the sin branch of the byte pair (bits[16 b + i % 8], bits[16 b + 8 + i % 8]))
template <typename T, bool BITS>
__global__ __launch_bounds__(256) void sched_step_kernel(const T* __restrict__ sample, const T* __restrict__ mo,
                                                        const T* __restrict__ noise, T* __restrict__ prev,
                                                        T* __restrict__ x0out, long long inner, long long mo_bstride,
                                                        long long total, GmStepParams p, const unsigned char* __restrict__ bits,
                                                        const unsigned int* __restrict__ table) {
  // every operation is an explicitly rounded, never-contracted IEEE fp32 op (__f*_rn) in the reference's order
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / inner, r = i - n * inner;
    const float s = ElemIO<T>::ld(sample + i);
    const float m = ElemIO<T>::ld(mo + n * mo_bstride + r);
    if (p.mode == 2) {  // PNDMScheduler._get_prev_sample (pndm.py:276-316), same op order
      const float e = (p.pred_type == 2) ? __fadd_rn(__fmul_rn(p.c_sa, m), __fmul_rn(p.c_sb, s)) : m;
      ElemIO<T>::st(prev + i, __fsub_rn(__fmul_rn(p.k0, s), __fdiv_rn(__fmul_rn(p.k1, e), p.c_prev)));
      continue;
    }
    float x0, eps;
    if (p.pred_type == 0) {
      x0 = __fdiv_rn(__fsub_rn(s, __fmul_rn(p.c_sb, m)), p.c_sa);
      eps = m;
    } else if (p.pred_type == 1) {
      x0 = m;
      eps = __fdiv_rn(__fsub_rn(s, __fmul_rn(p.c_sa, x0)), p.c_sb);
    } else {
      x0 = __fsub_rn(__fmul_rn(p.c_sa, s), __fmul_rn(p.c_sb, m));
      eps = __fadd_rn(__fmul_rn(p.c_sa, m), __fmul_rn(p.c_sb, s));
    }
    if (p.clip) x0 = fminf(fmaxf(x0, p.clip_lo), p.clip_hi);
    float out;
    if (p.mode == 0) {
      const float dir = __fmul_rn(p.c_dir, eps);
      out = __fadd_rn(__fmul_rn(p.c_prev, x0), dir);
    } else {
      out = __fadd_rn(__fmul_rn(p.k0, x0), __fmul_rn(p.k1, s));
    }
    if (p.noise_mode != 0) {
      float z;
      if constexpr (BITS) {
        const long long base = (i >> 4) * 16 + (i & 7);
        const unsigned e = table[(unsigned)bits[base] * 256u + bits[base + 8]];
        z = __uint_as_float((i & 8) ? (e & 0xffff0000u) : (e << 16));  // (bf16 bits -> fp32)
      } else {
        z = ElemIO<T>::ld(noise + i);
      }
      float v;
      if (p.noise_mode == 1) {
        v = __fmul_rn(p.c_noise, z);
      } else {
        const float pv = ElemIO<T>::ld(mo + n * mo_bstride + inner + r);
        float var;
        if (p.noise_mode == 2) {
          var = pv;
        } else {
          const float frac = __fdiv_rn(__fadd_rn(pv, 1.0f), 2.0f);
          var = __fadd_rn(__fmul_rn(frac, p.max_log), __fmul_rn(__fsub_rn(1.0f, frac), p.min_log));
        }
        v = __fmul_rn(__fsqrt_rn(var), z);
      }
      out = __fadd_rn(out, v);
    }
    ElemIO<T>::st(prev + i, out);
    if (x0out) ElemIO<T>::st(x0out + i, x0);
  }
}

static int ew_grid(long long total, int per_block = 256) {
  long long g = (total + per_block - 1) / per_block;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int gm_sched_step(const void* sample, const void* model_output, const void* noise, void* prev, void* x0,
                             long long batch, long long inner, long long mo_bstride, int dtype, const GmStepParams* p,
                             void* stream) {
  GM_REQUIRE(sample && model_output && prev && p, "null pointer");
  GM_REQUIRE(p->noise_mode == 0 || noise, "noise_mode != 0 needs a noise tensor");
  const long long total = batch * inner;
  if (total == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GM_F32)
    sched_step_kernel<float, false><<<ew_grid(total), 256, 0, st>>>((const float*)sample, (const float*)model_output,
                                                                    (const float*)noise, (float*)prev, (float*)x0, inner,
                                                                    mo_bstride, total, *p, nullptr, nullptr);
  else if (dtype == GM_BF16)
    sched_step_kernel<bf16_raw, false><<<ew_grid(total), 256, 0, st>>>((const bf16_raw*)sample, (const bf16_raw*)model_output,
                                                                       (const bf16_raw*)noise, (bf16_raw*)prev, (bf16_raw*)x0,
                                                                       inner, mo_bstride, total, *p, nullptr, nullptr);
  else
    GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// gm_sched_step for a bf16 chain whose noise arrives as the CPU generator's byte draws (see gm_normal_bf16_from_bits): the lookup runs inside the step's kernel
extern "C" int gm_sched_step_noise_bits(const void* sample, const void* model_output, const unsigned char* bits, const unsigned int* table, void* prev, void* x0,
                                        long long batch, long long inner, long long mo_bstride, const GmStepParams* p, void* stream) {
  GM_REQUIRE(sample && model_output && prev && p && bits && table, "null pointer");
  GM_REQUIRE(p->noise_mode != 0, "a step without noise takes gm_sched_step");
  const long long total = batch * inner;
  GM_REQUIRE(total % 16 == 0, "whole blocks of 16 values");
  if (total == 0) return 0;
  sched_step_kernel<bf16_raw, true><<<ew_grid(total), 256, 0, (hipStream_t)stream>>>((const bf16_raw*)sample, (const bf16_raw*)model_output, nullptr, (bf16_raw*)prev,
                                                                                    (bf16_raw*)x0, inner, mo_bstride, total, *p, bits, table);
  GM_LAUNCH_CHECK();
}

// ---- Gaussian noise of a bf16 ancestral sampling chain from the host generator's BYTES (host_noise.py) ------------------------------------------------------
// torch.randn(n, dtype=bfloat16) on the CPU generator (what the reference's DDPMScheduler.step draws: ddpm.py:244-248) is, block of 16 by block of 16, a pure
// function of byte pairs: out[16 b + j] = cos branch, out[16 b + 8 + j] = sin branch of (bits[16 b + j], bits[16 b + 8 + j]), j < 8.  table[b1 * 256 + b2] holds
// the two bf16 results (low half: cos branch).  One thread per pair; the 256 KiB table lives in L2.
__global__ __launch_bounds__(256) void normal_bf16_from_bits_kernel(const unsigned char* __restrict__ bits, const unsigned int* __restrict__ table,
                                                                   unsigned short* __restrict__ out, long long npairs) {
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npairs; p += (long long)gridDim.x * blockDim.x) {
    const long long base = (p >> 3) * 16 + (p & 7);
    const unsigned e = table[(unsigned)bits[base] * 256u + bits[base + 8]];
    out[base] = (unsigned short)(e & 0xFFFFu);
    out[base + 8] = (unsigned short)(e >> 16);
  }
}

extern "C" int gm_normal_bf16_from_bits(const unsigned char* bits, const unsigned int* table, void* out, long long n, void* stream) {
  GM_REQUIRE(bits && table && out, "null pointer");
  GM_REQUIRE(n >= 0 && n % 16 == 0, "whole blocks of 16 values");
  if (n == 0) return 0;
  normal_bf16_from_bits_kernel<<<ew_grid(n / 2), 256, 0, (hipStream_t)stream>>>(bits, table, (unsigned short*)out, n / 2);
  GM_LAUNCH_CHECK();
}

// out = post_mul * (((c0*x0 + c1*x1) + c2*x2) + c3*x3) / post_div, left to right, every op rounded (no contraction): the
// linear multi-step / Runge-Kutta combinations of PNDMScheduler (pndm.py:186-195,241-250).  post_mul / post_div of 1 are skipped
// (x*1 and x/1 are exact anyway).  x0 may be null (the reference's integer-0 accumulator: 0 + t == t exactly).
struct GmLincomb {
  const void* x[4];
  float c[4];
  int k;
  float post_mul, post_div;
};

template <typename T>
__global__ __launch_bounds__(256) void lincomb_kernel(GmLincomb a, T* __restrict__ out, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    float acc = 0.0f;
    bool first = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < a.k && a.x[j]) {
        const float t = __fmul_rn(a.c[j], ElemIO<T>::ld((const T*)a.x[j] + i));
        acc = first ? t : __fadd_rn(acc, t);
        first = false;
      }
    }
    if (a.post_mul != 1.0f) acc = __fmul_rn(a.post_mul, acc);
    if (a.post_div != 1.0f) acc = __fdiv_rn(acc, a.post_div);
    ElemIO<T>::st(out + i, acc);
  }
}

extern "C" int gm_lincomb(const void* const* x, const float* c, int k, float post_mul, float post_div, void* out, long long n,
                          int dtype, void* stream) {
  GM_REQUIRE(x && c && out, "null pointer");
  GM_REQUIRE(k >= 1 && k <= 4, "1..4 terms");
  if (n == 0) return 0;
  GmLincomb a;
  bool any = false;
  for (int j = 0; j < 4; ++j) {
    a.x[j] = j < k ? x[j] : nullptr;
    a.c[j] = j < k ? c[j] : 0.0f;
    any |= a.x[j] != nullptr;
  }
  GM_REQUIRE(any, "all terms are null");
  a.k = k;
  a.post_mul = post_mul;
  a.post_div = post_div;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GM_F32)
    lincomb_kernel<float><<<ew_grid(n), 256, 0, st>>>(a, (float*)out, n);
  else if (dtype == GM_BF16)
    lincomb_kernel<bf16_raw><<<ew_grid(n), 256, 0, st>>>(a, (bf16_raw*)out, n);
  else
    GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------------
// One term of the variational bound (DiffusionInferer.get_likelihood, inferer.py:203-256): per element the predicted mean
// (DDPM formula (7) from the model output), the true posterior mean q(x_{t-1}|x_t,x_0), and either KL(q || p) between the two
// Gaussians (t > 0) or the discretised-Gaussian decoder NLL (t == 0, inferer.py:281-321); per sample the mean over elements is
// accumulated into total[n].  All scalar sub-expressions are evaluated on the host as the reference evaluates them (0-dim
// fp32 torch-CPU tensors) and passed by value.
// ---------------------------------------------------------------------------------------------------------------------
struct GmKlParams {
  int pred_type;      // 0 epsilon, 1 sample, 2 v_prediction
  float c_sa, c_sb;   // alpha_prod_t ** 0.5, beta_prod_t ** 0.5
  int clip;           // clamp predicted x0 to [-1, 1] (inferer.py:222-223)
  float k0, k1;       // predicted mean = k0 * pred_x0 + k1 * x_t
  float m0, m1;       // posterior mean = m0 * x_0 + m1 * x_t        (ddpm.py:151-154)
  int t0;             // 1: decoder NLL, 0: KL between normals
  float s;            // KL: (-1 + log_pred_var - log_post_var) + exp(log_post_var - log_pred_var)
  float e;            // KL: exp(-log_pred_var);   NLL: inv_stdv = exp(-log_scales)
  float half_bin;     // NLL: bin_width / 2
};

__device__ __forceinline__ float approx_std_normal_cdf(float x, float c) {
  const float x3 = __fmul_rn(__fmul_rn(x, x), x);  // torch.pow(x, 3) is x*x*x
  return __fmul_rn(0.5f, __fadd_rn(1.0f, tanhf(__fmul_rn(c, __fadd_rn(x, __fmul_rn(0.044715f, x3))))));
}

template <typename T>
__global__ __launch_bounds__(256) void likelihood_kl_kernel(const T* __restrict__ x0, const T* __restrict__ xt,
                                                           const T* __restrict__ mo, T* __restrict__ kl_out,
                                                           double* __restrict__ rowsum, long long inner,
                                                           long long mo_bstride, GmKlParams p, float cdf_c) {
  const long long n = blockIdx.y;
  __shared__ double part[4];
  double acc = 0.0;
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < inner; r += (long long)gridDim.x * blockDim.x) {
    const long long i = n * inner + r;
    const float x = ElemIO<T>::ld(x0 + i), s = ElemIO<T>::ld(xt + i), m = ElemIO<T>::ld(mo + n * mo_bstride + r);
    float px0;
    if (p.pred_type == 0)
      px0 = __fdiv_rn(__fsub_rn(s, __fmul_rn(p.c_sb, m)), p.c_sa);
    else if (p.pred_type == 1)
      px0 = m;
    else
      px0 = __fsub_rn(__fmul_rn(p.c_sa, s), __fmul_rn(p.c_sb, m));
    if (p.clip) px0 = fminf(fmaxf(px0, -1.0f), 1.0f);
    const float pmean = __fadd_rn(__fmul_rn(p.k0, px0), __fmul_rn(p.k1, s));
    float kl;
    if (!p.t0) {
      const float qmean = __fadd_rn(__fmul_rn(p.m0, x), __fmul_rn(p.m1, s));
      const float d = __fsub_rn(qmean, pmean);
      kl = __fmul_rn(0.5f, __fadd_rn(p.s, __fmul_rn(__fmul_rn(d, d), p.e)));
    } else {
      const float cx = __fsub_rn(x, pmean);
      const float cdf_plus = approx_std_normal_cdf(__fmul_rn(p.e, __fadd_rn(cx, p.half_bin)), cdf_c);
      const float cdf_min = approx_std_normal_cdf(__fmul_rn(p.e, __fsub_rn(cx, p.half_bin)), cdf_c);
      float lp;
      if (x < -0.999f)
        lp = logf(fmaxf(cdf_plus, 1e-12f));
      else if (x > 0.999f)
        lp = logf(fmaxf(__fsub_rn(1.0f, cdf_min), 1e-12f));
      else
        lp = logf(fmaxf(__fsub_rn(cdf_plus, cdf_min), 1e-12f));
      kl = -lp;
    }
    if (kl_out) {
      ElemIO<T>::st(kl_out + i, kl);
      kl = ElemIO<T>::ld(kl_out + i);  // the mean is taken over the stored (dtype-rounded) map, as the reference does
    }
    acc += (double)kl;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  // one partial per (sample, block), stored -- the fold below adds them in block order: the bound is bit-reproducible (round 2: fp64 atomics)
  if (threadIdx.x == 0) rowsum[n * gridDim.x + blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ void likelihood_fold_kernel(const double* __restrict__ rowsum, float* __restrict__ total, long long batch, int parts, double inv_inner) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n < batch) {
    double s = 0.0;
    for (int j = 0; j < parts; ++j) s += rowsum[n * parts + j];
    total[n] += (float)(s * inv_inner);
  }
}

static long long likelihood_blocks(long long inner) {
  long long gx = (inner + 255) / 256;
  return gx > 2048 ? 2048 : (gx < 1 ? 1 : gx);
}

// doubles of gm_likelihood_term's workspace: one partial per (sample, block); needs no initialisation
extern "C" long long gm_likelihood_workspace_elems(long long batch, long long inner) { return batch * likelihood_blocks(inner); }

extern "C" int gm_likelihood_term(const void* x0, const void* xt, const void* model_output, void* kl, float* total,
                                  double* workspace, long long batch, long long inner, long long mo_bstride, int dtype,
                                  const GmKlParams* p, void* stream) {
  GM_REQUIRE(x0 && xt && model_output && total && workspace && p, "null pointer");
  if (batch * inner == 0) return 0;
  GM_REQUIRE(batch <= 65535, "batch too large");
  hipStream_t st = (hipStream_t)stream;
  const float cdf_c = sqrtf((float)(2.0 / 3.14159265358979323846));  // torch.sqrt(torch.Tensor([2.0 / math.pi]))
  const long long gx = likelihood_blocks(inner);
  dim3 grid((unsigned)gx, (unsigned)batch);
  if (dtype == GM_F32)
    likelihood_kl_kernel<float><<<grid, 256, 0, st>>>((const float*)x0, (const float*)xt, (const float*)model_output,
                                                      (float*)kl, workspace, inner, mo_bstride, *p, cdf_c);
  else if (dtype == GM_BF16)
    likelihood_kl_kernel<bf16_raw><<<grid, 256, 0, st>>>((const bf16_raw*)x0, (const bf16_raw*)xt, (const bf16_raw*)model_output,
                                                         (bf16_raw*)kl, workspace, inner, mo_bstride, *p, cdf_c);
  else
    GM_FAIL(-2, "unsupported dtype");
  likelihood_fold_kernel<<<(unsigned)((batch + 63) / 64), 64, 0, st>>>(workspace, total, batch, (int)gx, 1.0 / (double)inner);
  GM_LAUNCH_CHECK();
}

// out[n, i] = a[n] * x[n, i] + b[n] * y[n, i]   (a, b: fp32 device vectors, one entry per batch row)
template <typename T>
__global__ __launch_bounds__(256) void axpby_rows_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                        const float* __restrict__ a, const float* __restrict__ b,
                                                        T* __restrict__ out, long long inner, long long total) {
#pragma clang fp contract(off)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / inner;
    const float u = __fmul_rn(a[n], ElemIO<T>::ld(x + i));
    const float v = __fmul_rn(b[n], ElemIO<T>::ld(y + i));
    ElemIO<T>::st(out + i, __fadd_rn(u, v));
  }
}

extern "C" int gm_axpby_rows(const void* x, const void* y, const float* a, const float* b, void* out, long long batch,
                             long long inner, int dtype, void* stream) {
  GM_REQUIRE(x && y && a && b && out, "null pointer");
  const long long total = batch * inner;
  if (total == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GM_F32)
    axpby_rows_kernel<float><<<ew_grid(total), 256, 0, st>>>((const float*)x, (const float*)y, a, b, (float*)out, inner, total);
  else if (dtype == GM_BF16)
    axpby_rows_kernel<bf16_raw><<<ew_grid(total), 256, 0, st>>>((const bf16_raw*)x, (const bf16_raw*)y, a, b,
                                                                (bf16_raw*)out, inner, total);
  else
    GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------------
// Strided 2-D copy with optional dtype conversion: dst[r, dst_off + c] = src[r, src_off + c], r < rows, c < C.
// Used to build channel concatenations in the NDHWC arena and for dtype casts (rows = 1).
// ---------------------------------------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void copy_channels_kernel(const TS* __restrict__ src, long long src_ld,
                                                           TD* __restrict__ dst, long long dst_ld, long long rows,
                                                           int C) {
  const long long total = rows * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / C;
    const int c = (int)(i - r * C);
    ElemIO<TD>::st(dst + r * dst_ld + c, ElemIO<TS>::ld(src + r * src_ld + c));
  }
}

// 16-byte vector variant (same dtype, C and both leading dims multiples of the vector width, 16-B aligned bases)
__global__ __launch_bounds__(256) void copy_channels_vec_kernel(const uint4* __restrict__ src, long long src_ld4,
                                                               uint4* __restrict__ dst, long long dst_ld4,
                                                               long long rows, int C4) {
  const long long total = rows * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / C4;
    const int c = (int)(i - r * C4);
    dst[r * dst_ld4 + c] = src[r * src_ld4 + c];
  }
}

extern "C" int gm_copy_channels(const void* src, long long src_ld, int src_dtype, void* dst, long long dst_ld,
                                int dst_dtype, long long rows, int C, void* stream) {
  GM_REQUIRE(src && dst, "null pointer");
  const long long total = rows * C;
  if (total == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int es = src_dtype == GM_F32 ? 4 : 2;
  if (src_dtype == dst_dtype) {
    const int vec = 16 / es;
    if (C % vec == 0 && src_ld % vec == 0 && dst_ld % vec == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0) {
      copy_channels_vec_kernel<<<ew_grid(total / vec), 256, 0, st>>>((const uint4*)src, src_ld / vec, (uint4*)dst,
                                                                     dst_ld / vec, rows, C / vec);
      GM_LAUNCH_CHECK();
    }
  }
  if (src_dtype == GM_F32 && dst_dtype == GM_F32)
    copy_channels_kernel<float, float><<<ew_grid(total), 256, 0, st>>>((const float*)src, src_ld, (float*)dst, dst_ld, rows, C);
  else if (src_dtype == GM_F32 && dst_dtype == GM_BF16)
    copy_channels_kernel<float, bf16_raw><<<ew_grid(total), 256, 0, st>>>((const float*)src, src_ld, (bf16_raw*)dst, dst_ld, rows, C);
  else if (src_dtype == GM_BF16 && dst_dtype == GM_F32)
    copy_channels_kernel<bf16_raw, float><<<ew_grid(total), 256, 0, st>>>((const bf16_raw*)src, src_ld, (float*)dst, dst_ld, rows, C);
  else if (src_dtype == GM_BF16 && dst_dtype == GM_BF16)
    copy_channels_kernel<bf16_raw, bf16_raw><<<ew_grid(total), 256, 0, st>>>((const bf16_raw*)src, src_ld, (bf16_raw*)dst, dst_ld, rows, C);
  else
    GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------------
// Boundary layout transforms: the public API is logically NC[D]HW (reference convention), the arena is N[D]HWC.
// [N][C][V] <-> [N][V][ld >= C] through a 32x33 LDS tile so that both sides are coalesced.
// ---------------------------------------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int C,
                                                          long long V, long long dst_ld) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const long long v0 = (long long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j;
    const long long v = v0 + tx;
    tile[j][tx] = (c < C && v < V) ? ElemIO<TS>::ld(src + ((long long)n * C + c) * V + v) : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const long long v = v0 + j;
    const int c = c0 + tx;
    if (c < C && v < V) ElemIO<TD>::st(dst + ((long long)n * V + v) * dst_ld + c, tile[tx][j]);
  }
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const TS* __restrict__ src, long long src_ld,
                                                          TD* __restrict__ dst, int C, long long V) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const long long v0 = (long long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const long long v = v0 + j;
    const int c = c0 + tx;
    tile[j][tx] = (c < C && v < V) ? ElemIO<TS>::ld(src + ((long long)n * V + v) * src_ld + c) : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j;
    const long long v = v0 + tx;
    if (c < C && v < V) ElemIO<TD>::st(dst + ((long long)n * C + c) * V + v, tile[tx][j]);
  }
}


template <typename TS, typename TD>
static void launch_nchw_to_nhwc(const void* src, void* dst, int N, int C, long long V, long long dst_ld, hipStream_t st) {
  dim3 grid(gm_cdiv(V, 32), gm_cdiv(C, 32), N);
  nchw_to_nhwc_kernel<TS, TD><<<grid, 256, 0, st>>>((const TS*)src, (TD*)dst, C, V, dst_ld);
}
template <typename TS, typename TD>
static void launch_nhwc_to_nchw(const void* src, long long src_ld, void* dst, int N, int C, long long V, hipStream_t st) {
  dim3 grid(gm_cdiv(V, 32), gm_cdiv(C, 32), N);
  nhwc_to_nchw_kernel<TS, TD><<<grid, 256, 0, st>>>((const TS*)src, src_ld, (TD*)dst, C, V);
}

extern "C" int gm_nchw_to_nhwc(const void* src, int src_dtype, void* dst, int dst_dtype, int N, int C, long long V,
                               long long dst_ld, void* stream) {
  GM_REQUIRE(src && dst, "null pointer");
  if ((long long)N * C * V == 0) return 0;
  GM_REQUIRE(N <= 65535 && gm_cdiv(C, 32) <= 65535, "batch / channel count too large for the grid");
  hipStream_t st = (hipStream_t)stream;
  if (src_dtype == GM_F32 && dst_dtype == GM_F32) launch_nchw_to_nhwc<float, float>(src, dst, N, C, V, dst_ld, st);
  else if (src_dtype == GM_F32 && dst_dtype == GM_BF16) launch_nchw_to_nhwc<float, bf16_raw>(src, dst, N, C, V, dst_ld, st);
  else if (src_dtype == GM_BF16 && dst_dtype == GM_F32) launch_nchw_to_nhwc<bf16_raw, float>(src, dst, N, C, V, dst_ld, st);
  else if (src_dtype == GM_BF16 && dst_dtype == GM_BF16) launch_nchw_to_nhwc<bf16_raw, bf16_raw>(src, dst, N, C, V, dst_ld, st);
  else GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

extern "C" int gm_nhwc_to_nchw(const void* src, long long src_ld, int src_dtype, void* dst, int dst_dtype, int N, int C,
                               long long V, void* stream) {
  GM_REQUIRE(src && dst, "null pointer");
  if ((long long)N * C * V == 0) return 0;
  GM_REQUIRE(N <= 65535 && gm_cdiv(C, 32) <= 65535, "batch / channel count too large for the grid");
  hipStream_t st = (hipStream_t)stream;
  if (src_dtype == GM_F32 && dst_dtype == GM_F32) launch_nhwc_to_nchw<float, float>(src, src_ld, dst, N, C, V, st);
  else if (src_dtype == GM_F32 && dst_dtype == GM_BF16) launch_nhwc_to_nchw<float, bf16_raw>(src, src_ld, dst, N, C, V, st);
  else if (src_dtype == GM_BF16 && dst_dtype == GM_F32) launch_nhwc_to_nchw<bf16_raw, float>(src, src_ld, dst, N, C, V, st);
  else if (src_dtype == GM_BF16 && dst_dtype == GM_BF16) launch_nhwc_to_nchw<bf16_raw, bf16_raw>(src, src_ld, dst, N, C, V, st);
  else GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------------
// 2x nearest up-sampling / 2x average pooling of an N[D]HWC tensor (only the `resblock_updown` ResnetBlock variant
// materialises these: reference diffusion_model_unet.py:635-639,674-682; plain Upsample is folded into the conv).
// mode 0: up (out dims = 2 * in dims on every active axis), mode 1: avg-pool down (out dims = in dims / 2).
// dims are given as D,H,W of the *input*; a 2-D tensor passes D = 1 and act_d = 0.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void resample2x_kernel(const T* __restrict__ src, long long src_ld, T* __restrict__ dst,
                                                        long long dst_ld, int N, int C, int Di, int Hi, int Wi, int act_d,
                                                        int mode) {
  const int fd = act_d ? 2 : 1;
  int Do, Ho, Wo;
  if (mode == 0) { Do = Di * fd; Ho = Hi * 2; Wo = Wi * 2; } else { Do = Di / fd; Ho = Hi / 2; Wo = Wi / 2; }
  const long long total = (long long)N * Do * Ho * Wo * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long r = i / C;
    const int w = (int)(r % Wo); r /= Wo;
    const int h = (int)(r % Ho); r /= Ho;
    const int d = (int)(r % Do);
    const long long n = r / Do;
    float v;
    if (mode == 0) {
      const long long sv = (((long long)n * Di + d / fd) * Hi + h / 2) * Wi + w / 2;
      v = ElemIO<T>::ld(src + sv * src_ld + c);
    } else {
      float acc = 0.f;
      for (int a = 0; a < fd; ++a)
        for (int b = 0; b < 2; ++b)
          for (int e = 0; e < 2; ++e) {
            const long long sv = (((long long)n * Di + d * fd + a) * Hi + h * 2 + b) * Wi + w * 2 + e;
            acc += ElemIO<T>::ld(src + sv * src_ld + c);
          }
      v = acc / (float)(fd * 4);
    }
    const long long ov = (((long long)n * Do + d) * Ho + h) * Wo + w;
    ElemIO<T>::st(dst + ov * dst_ld + c, v);
  }
}

extern "C" int gm_resample2x(const void* src, long long src_ld, void* dst, long long dst_ld, int N, int C, int Di, int Hi,
                             int Wi, int act_d, int mode, int dtype, void* stream) {
  GM_REQUIRE(src && dst, "null pointer");
  GM_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (up) or 1 (avg-pool)");
  const int fd = act_d ? 2 : 1;
  long long total = (long long)N * C * (mode == 0 ? (long long)Di * fd * Hi * 2 * Wi * 2 : (long long)(Di / fd) * (Hi / 2) * (Wi / 2));
  if (total == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GM_F32)
    resample2x_kernel<float><<<ew_grid(total), 256, 0, st>>>((const float*)src, src_ld, (float*)dst, dst_ld, N, C, Di, Hi, Wi, act_d, mode);
  else if (dtype == GM_BF16)
    resample2x_kernel<bf16_raw><<<ew_grid(total), 256, 0, st>>>((const bf16_raw*)src, src_ld, (bf16_raw*)dst, dst_ld, N, C, Di, Hi, Wi, act_d, mode);
  else GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// Phase image of a 2x sub-lattice: dst[n][d][h][w][c] = src[n][2 d + rd][2 h + rh][2 w + rw][c], phase = (rd << 2) | (rh << 1) | rw, output
// extents (X - r + 1) / 2 per active axis.  The weight gradient of a stride-2 convolution with an EVEN kernel (the k = 4 / s = 2 / p = 1
// down- and up-sampling convolutions of the VQ-VAE, vqvae.py:127-150,244-261) is assembled from stride-1 weight gradients over the 2^d
// phase images of its input (ops.conv_wgrad).  dims are D, H, W of the input; a 2-D tensor passes D = 1 and act_d = 0.
template <typename T>
__global__ __launch_bounds__(256) void phase2x_kernel(const T* __restrict__ src, long long src_ld, T* __restrict__ dst, long long dst_ld, int N, int C,
                                                     int Di, int Hi, int Wi, int act_d, int phase) {
  const int rd = act_d ? (phase >> 2) & 1 : 0, rh = (phase >> 1) & 1, rw = phase & 1;
  const int Do = act_d ? (Di - rd + 1) / 2 : Di, Ho = (Hi - rh + 1) / 2, Wo = (Wi - rw + 1) / 2;
  const long long total = (long long)N * Do * Ho * Wo * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long r = i / C;
    const int w = (int)(r % Wo); r /= Wo;
    const int h = (int)(r % Ho); r /= Ho;
    const int d = (int)(r % Do);
    const long long n = r / Do;
    const long long sv = ((n * Di + (act_d ? 2 * d + rd : d)) * Hi + 2 * h + rh) * Wi + 2 * w + rw;
    const long long ov = ((n * Do + d) * Ho + h) * Wo + w;
    dst[ov * dst_ld + c] = src[sv * src_ld + c];
  }
}

extern "C" int gm_phase2x(const void* src, long long src_ld, void* dst, long long dst_ld, int N, int C, int Di, int Hi, int Wi, int act_d,
                          int phase, int dtype, void* stream) {
  GM_REQUIRE(src && dst, "null pointer");
  GM_REQUIRE(phase >= 0 && phase < 8, "phase is a 3-bit (d, h, w) parity");
  const int rd = act_d ? (phase >> 2) & 1 : 0, rh = (phase >> 1) & 1, rw = phase & 1;
  const long long total = (long long)N * C * (act_d ? (Di - rd + 1) / 2 : Di) * ((Hi - rh + 1) / 2) * ((Wi - rw + 1) / 2);
  if (total <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GM_F32)
    phase2x_kernel<float><<<ew_grid(total), 256, 0, st>>>((const float*)src, src_ld, (float*)dst, dst_ld, N, C, Di, Hi, Wi, act_d, phase);
  else if (dtype == GM_BF16)
    phase2x_kernel<bf16_raw><<<ew_grid(total), 256, 0, st>>>((const bf16_raw*)src, src_ld, (bf16_raw*)dst, dst_ld, N, C, Di, Hi, Wi, act_d, phase);
  else GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------------
// Sinusoidal timestep embedding: out[b, :half] = cos(t_b * f_i), out[b, half:2*half] = sin(t_b * f_i),
// f_i = exp(-ln(max_period) * i / half); zero pad when dim is odd.  (reference diffusion_model_unet.py:461-485)
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void timestep_embedding_kernel(const float* __restrict__ t, T* __restrict__ out, int B, int dim,
                                          float max_period) {
#pragma clang fp contract(off)
  const int half = dim / 2;
  const int total = B * dim;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / dim, j = i - b * dim;
    float v = 0.f;
    if (j < 2 * half) {
      const int k = j < half ? j : j - half;
      const float e = (-logf(max_period) * (float)k) / (float)half;  // same op order as the reference (fp32)
      const float arg = t[b] * expf(e);
      v = j < half ? cosf(arg) : sinf(arg);
    }
    ElemIO<T>::st(out + i, v);
  }
}

extern "C" int gm_timestep_embedding(const float* timesteps, void* out, int B, int dim, float max_period, int dtype,
                                     void* stream) {
  GM_REQUIRE(timesteps && out, "null pointer");
  if (B * dim == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int grid = gm_cdiv((long long)B * dim, 256);
  if (dtype == GM_F32) timestep_embedding_kernel<float><<<grid, 256, 0, st>>>(timesteps, (float*)out, B, dim, max_period);
  else if (dtype == GM_BF16) timestep_embedding_kernel<bf16_raw><<<grid, 256, 0, st>>>(timesteps, (bf16_raw*)out, B, dim, max_period);
  else GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// GEGLU gate of the transformer feed-forward: out[r, j] = x[r, j] * gelu_erf(x[r, inner + j])
template <typename T>
__global__ __launch_bounds__(256) void geglu_kernel(const T* __restrict__ x, long long x_ld, T* __restrict__ out,
                                                   long long out_ld, long long rows, int inner) {
  const long long total = rows * inner;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / inner;
    const int j = (int)(i - r * inner);
    const float a = ElemIO<T>::ld(x + r * x_ld + j);
    const float g = ElemIO<T>::ld(x + r * x_ld + inner + j);
    const float gelu = 0.5f * g * (1.0f + erff(g * 0.70710678118654752440f));
    ElemIO<T>::st(out + r * out_ld + j, a * gelu);
  }
}

extern "C" int gm_geglu(const void* x, long long x_ld, void* out, long long out_ld, long long rows, int inner, int dtype,
                        void* stream) {
  GM_REQUIRE(x && out, "null pointer");
  const long long total = rows * inner;
  if (total == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GM_F32) geglu_kernel<float><<<ew_grid(total), 256, 0, st>>>((const float*)x, x_ld, (float*)out, out_ld, rows, inner);
  else if (dtype == GM_BF16) geglu_kernel<bf16_raw><<<ew_grid(total), 256, 0, st>>>((const bf16_raw*)x, x_ld, (bf16_raw*)out, out_ld, rows, inner);
  else GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// AutoencoderKL posterior: sigma = exp(clamp(log_var, -30, 20) / 2); z = mu + eps * sigma (eps may be null: sigma only)
template <typename T>
__global__ __launch_bounds__(256) void aekl_sample_kernel(const T* __restrict__ mu, const T* __restrict__ logvar,
                                                         const T* __restrict__ eps, T* __restrict__ sigma,
                                                         T* __restrict__ z, long long total) {
#pragma clang fp contract(off)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    float lv = ElemIO<T>::ld(logvar + i);
    lv = fminf(fmaxf(lv, -30.0f), 20.0f);
    const float s = expf(lv / 2.0f);
    if (sigma) ElemIO<T>::st(sigma + i, s);
    if (z) {
      const float e = ElemIO<T>::ld(eps + i) * s;
      ElemIO<T>::st(z + i, ElemIO<T>::ld(mu + i) + e);
    }
  }
}

extern "C" int gm_aekl_sample(const void* mu, const void* logvar, const void* eps, void* sigma, void* z, long long total,
                              int dtype, void* stream) {
  GM_REQUIRE(logvar, "null pointer");
  GM_REQUIRE(!z || (mu && eps), "z needs mu and eps");
  if (total == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GM_F32)
    aekl_sample_kernel<float><<<ew_grid(total), 256, 0, st>>>((const float*)mu, (const float*)logvar, (const float*)eps, (float*)sigma, (float*)z, total);
  else if (dtype == GM_BF16)
    aekl_sample_kernel<bf16_raw><<<ew_grid(total), 256, 0, st>>>((const bf16_raw*)mu, (const bf16_raw*)logvar, (const bf16_raw*)eps, (bf16_raw*)sigma, (bf16_raw*)z, total);
  else GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// out = a + b * c (AutoencoderKL.sampling with an explicit sigma: autoencoderkl.py:751-752)
template <typename T>
__global__ __launch_bounds__(256) void addcmul_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c,
                                                     T* __restrict__ out, long long total) {
#pragma clang fp contract(off)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const float m = ElemIO<T>::ld(b + i) * ElemIO<T>::ld(c + i);
    ElemIO<T>::st(out + i, ElemIO<T>::ld(a + i) + m);
  }
}

extern "C" int gm_addcmul(const void* a, const void* b, const void* c, void* out, long long total, int dtype, void* stream) {
  GM_REQUIRE(a && b && c && out, "null pointer");
  if (total == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GM_F32) addcmul_kernel<float><<<ew_grid(total), 256, 0, st>>>((const float*)a, (const float*)b, (const float*)c, (float*)out, total);
  else if (dtype == GM_BF16) addcmul_kernel<bf16_raw><<<ew_grid(total), 256, 0, st>>>((const bf16_raw*)a, (const bf16_raw*)b, (const bf16_raw*)c, (bf16_raw*)out, total);
  else GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// out = x * s (mode 0) or x / s (mode 1): the latent scale factor of LatentDiffusionInferer (inferers/inferer.py:386,472)
template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(const T* __restrict__ x, T* __restrict__ out, float s, int mode, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const float v = ElemIO<T>::ld(x + i);
    ElemIO<T>::st(out + i, mode == 0 ? v * s : v / s);
  }
}

extern "C" int gm_scale(const void* x, void* out, float s, int mode, long long total, int dtype, void* stream) {
  GM_REQUIRE(x && out, "null pointer");
  GM_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (multiply) or 1 (divide)");
  if (total == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GM_F32) scale_kernel<float><<<ew_grid(total), 256, 0, st>>>((const float*)x, (float*)out, s, mode, total);
  else if (dtype == GM_BF16) scale_kernel<bf16_raw><<<ew_grid(total), 256, 0, st>>>((const bf16_raw*)x, (bf16_raw*)out, s, mode, total);
  else GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------------
// A stand-alone activation over a contiguous tensor, and its backward from the PRE-activation: the training forward of layers whose
// activation cannot ride in a convolution epilogue -- MONAI's Convolution(adn_ordering="DA") puts a dropout BETWEEN the convolution and the
// activation (VQVAE with dropout > 0: vqvae.py:61-80,127-150), and tanh / sigmoid / SiLU / GELU derivatives need z, which the fused
// epilogue does not keep.  Codes = the epilogue's (GmConvDesc.post_act): 1 ReLU, 2 tanh, 3 sigmoid, 4 SiLU, 5 LeakyReLU(0.01), 6 GELU (erf).
//   mode 0: out = act(x);  mode 1: out = g * act'(x)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_fwd_code(float z, int act) {
  switch (act) {
    case 1: return fmaxf(z, 0.f);
    case 2: return tanhf(z);
    case 3: return 1.0f / (1.0f + expf(-z));
    case 4: return z / (1.0f + expf(-z));
    case 5: return z > 0.f ? z : 0.01f * z;
    case 6: return 0.5f * z * (1.0f + erff(z * 0.70710678118654752f));
    default: return z;
  }
}
__device__ __forceinline__ float act_grad_code(float z, int act) {
  switch (act) {
    case 1: return z > 0.f ? 1.f : 0.f;
    case 2: { const float t = tanhf(z); return 1.0f - t * t; }
    case 3: { const float s = 1.0f / (1.0f + expf(-z)); return s * (1.0f - s); }
    case 4: { const float s = 1.0f / (1.0f + expf(-z)); return s * (1.0f + z * (1.0f - s)); }
    case 5: return z > 0.f ? 1.f : 0.01f;
    case 6: return 0.5f * (1.0f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * expf(-0.5f * z * z);
    default: return 1.f;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void activation_kernel(const T* __restrict__ x, const T* __restrict__ g, T* __restrict__ out, int act, int mode, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const float z = ElemIO<T>::ld(x + i);
    ElemIO<T>::st(out + i, mode == 0 ? act_fwd_code(z, act) : ElemIO<T>::ld(g + i) * act_grad_code(z, act));
  }
}

extern "C" int gm_activation(const void* x, const void* g, void* out, int act, int mode, long long total, int dtype, void* stream) {
  GM_REQUIRE(x && out, "null pointer");
  GM_REQUIRE(mode == 0 || (mode == 1 && g), "mode must be 0 (forward) or 1 (backward: needs the upstream gradient)");
  GM_REQUIRE(act >= 0 && act <= 6, "unknown activation code");
  if (total == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GM_F32) activation_kernel<float><<<ew_grid(total), 256, 0, st>>>((const float*)x, (const float*)g, (float*)out, act, mode, total);
  else if (dtype == GM_BF16) activation_kernel<bf16_raw><<<ew_grid(total), 256, 0, st>>>((const bf16_raw*)x, (const bf16_raw*)g, (bf16_raw*)out, act, mode, total);
  else GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------------
// F.interpolate(x, size, mode="nearest") on an arena tensor (N, D, H, W, C): the ControlNet latent inferers resize the conditioning
// image to the latent grid (reference: inferers/inferer.py:926-927, 989-990, 1096-1097).  Source index = min(floor(dst * (in/out)),
// in - 1) with the scale evaluated in fp32, exactly torch's nearest_neighbor_compute_source_index.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nearest_resize_kernel(const T* __restrict__ x, long long x_ld, T* __restrict__ y, long long y_ld,
                                                            int N, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C,
                                                            float sd, float sh, float sw) {
  const long long total = (long long)N * Do * Ho * Wo * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long r = i / C;
    const int ow = (int)(r % Wo); r /= Wo;
    const int oh = (int)(r % Ho); r /= Ho;
    const int od = (int)(r % Do);
    const int n = (int)(r / Do);
    const int id = min((int)floorf(od * sd), Di - 1), ih = min((int)floorf(oh * sh), Hi - 1), iw = min((int)floorf(ow * sw), Wi - 1);
    const long long src = (((long long)n * Di + id) * Hi + ih) * Wi + iw;
    const long long dst = (((long long)n * Do + od) * Ho + oh) * Wo + ow;
    y[dst * y_ld + c] = x[src * x_ld + c];
  }
}

extern "C" int gm_nearest_resize(const void* x, long long x_ld, void* y, long long y_ld, int N, int Di, int Hi, int Wi, int Do, int Ho,
                                 int Wo, int C, int dtype, void* stream) {
  GM_REQUIRE(x && y, "null pointer");
  GM_REQUIRE(Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0, "empty spatial extent");
  const long long total = (long long)N * Do * Ho * Wo * C;
  if (total == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const float sd = (float)Di / (float)Do, sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  if (dtype == GM_F32)
    nearest_resize_kernel<float><<<ew_grid(total), 256, 0, st>>>((const float*)x, x_ld, (float*)y, y_ld, N, Di, Hi, Wi, Do, Ho, Wo, C, sd, sh, sw);
  else if (dtype == GM_BF16)
    nearest_resize_kernel<bf16_raw><<<ew_grid(total), 256, 0, st>>>((const bf16_raw*)x, x_ld, (bf16_raw*)y, y_ld, N, Di, Hi, Wi, Do, Ho, Wo, C, sd, sh, sw);
  else
    GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}
