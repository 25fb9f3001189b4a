// N[D]HWC convolution / linear layer as an LDS-staged implicit GEMM on the CDNA4 matrix cores.
//
// Replaces every MONAI `Convolution(conv_only=True)` (= nn.Conv{2,3}d / nn.ConvTranspose{2,3}d) and nn.Linear on the hot
// path, *fused with their surroundings in the reference graph*:
//   prologue : GroupNorm-apply (per-(n,c) scale/shift from groupnorm.hip) + SiLU / ReLU, nearest-2x up-sampling folded
//              into the input indexing (Upsample, diffusion_model_unet.py:572-585), zero-insertion for transposed convs
//   epilogue : bias, + timestep-embedding row (ResnetBlock, diffusion_model_unet.py:686-690), + residual / skip tensor
//              (diffusion_model_unet.py:692-696), activation.
//
// Tiling (per 256-thread workgroup = 4 wave64):
//   * an output tile of BM = 2^(ltd+lth+ltw) voxels x BN output channels;
//   * the K loop runs over input-channel chunks of 64 bytes (32 bf16 / 16 fp32 channels).  For each chunk the input
//     *halo patch* of the tile is staged ONCE into LDS (after the fused prologue) and then re-used by all k^3 taps --
//     this is what makes the 3x3x3 case LDS- rather than L2-fed, and applies SiLU once per element instead of 27x;
//   * per tap the [BN][64 B] weight panel is double-buffered through LDS (one 16-byte load per thread per tap);
//   * MFMA operands are read with ds_read_b128 from 80-byte padded rows (bank-conflict free for 16 consecutive voxels);
//   * orientation: A = weights (rows = output channels), B = activations (cols = voxels), so a lane ends up with 4
//     consecutive output channels of one voxel -> 8/16-byte NDHWC stores.
// bf16 uses v_mfma_f32_16x16x32_bf16, fp32 uses v_mfma_f32_16x16x4_f32 (exact fp32 products; the parity path).
#include "gm_common.h"

#include "conv_common.h"

// Post-activation of the generic kernels with the transcendental forms OUT OF LINE: conv_post_act inlines tanh / exp / erf expansions at every
// call site -- 32 copies in the epilogue below, a 12 000-instruction kernel (~70 KB: more than the 64 KB instruction cache two CUs share), and
// every launch of a small problem paid for streaming that code (rocprofv3: 20-24 us per launch whatever the problem size).
static __device__ __attribute__((noinline)) float conv_post_act_rare(float v, int act) { return conv_post_act(v, act); }
__device__ __forceinline__ float conv_post_act_lean(float v, int act) {
  if (act == 0) return v;
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 5) return v > 0.f ? v : 0.01f * v;
  return conv_post_act_rare(v, act);
}

template <typename T, int WM, int WN, int MF, int NFR>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const GmConvDesc p) {
  constexpr int BK = ConvTraits<T>::BK;
  constexpr int VECW = ConvTraits<T>::VECW;
  constexpr int BM = WM * MF * 16;
  constexpr int BN = WN * NFR * 16;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr bool PRECISE = sizeof(T) == 4;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int wm = wave % WM, wn = wave / WM;

  // ---- tile geometry ------------------------------------------------------------------------------------------------
  const int td = 1 << p.ltd, th = 1 << p.lth, tw = 1 << p.ltw;
  const int ntd = (p.Do + td - 1) >> p.ltd, nth = (p.Ho + th - 1) >> p.lth, ntw = (p.Wo + tw - 1) >> p.ltw;
  const int ncb = (p.Cout + BN - 1) / BN;
  int b = blockIdx.x;
  const int cb = b % ncb; b /= ncb;
  const int tw_i = b % ntw; b /= ntw;
  const int th_i = b % nth; b /= nth;
  const int td_i = b % ntd; b /= ntd;
  const int n = b;
  const int od0 = td_i << p.ltd, oh0 = th_i << p.lth, ow0 = tw_i << p.ltw;

  const int pD = (td - 1) * p.sd + (p.kd - 1) * p.dd + 1;
  const int pH = (th - 1) * p.sh + (p.kh - 1) * p.dh + 1;
  const int pW = (tw - 1) * p.sw + (p.kw - 1) * p.dw + 1;
  const int P = pD * pH * pW;
  // virtual (post up-sample / zero-insert) input extents and the patch origin in that space
  int Dv = p.Ds, Hv = p.Hs, Wv = p.Ws;
  if (p.in_mode == 1) { Dv *= p.fd; Hv *= p.fh; Wv *= p.fw; }
  else if (p.in_mode == 2) { Dv = (p.Ds - 1) * p.fd + 1; Hv = (p.Hs - 1) * p.fh + 1; Wv = (p.Ws - 1) * p.fw + 1; }
  const int ud0 = od0 * p.sd - p.pd, uh0 = oh0 * p.sh - p.ph, uw0 = ow0 * p.sw - p.pw;

  char* ldsA = smem;                                   // [P][CONV_ROWB]
  char* ldsB = smem + (size_t)P * CONV_ROWB;           // [2][BN][CONV_ROWB]

  const int T_taps = p.kd * p.kh * p.kw;
  const int nchunks = (p.Cin + BK - 1) / BK;
  const int cout_pad = (p.Cout + 15) & ~15;
  const long long total_steps = (long long)nchunks * T_taps;

  // ---- per-lane LDS read offsets -------------------------------------------------------------------------------------
  int aoff[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int m = (wm * MF + mf) * 16 + l15;
    const int a = m >> (p.lth + p.ltw), bb = (m >> p.ltw) & (th - 1), c = m & (tw - 1);
    aoff[mf] = ((a * p.sd * pH + bb * p.sh) * pW + c * p.sw) * CONV_ROWB + q * 16;
  }
  int boff[NFR];
#pragma unroll
  for (int nf = 0; nf < NFR; ++nf) boff[nf] = ((wn * NFR + nf) * 16 + l15) * CONV_ROWB + q * 16;

  f32x4_t acc[NFR][MF];
#pragma unroll
  for (int nf = 0; nf < NFR; ++nf)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- weight-panel staging helpers (one or two 16-byte items per thread) ---------------------------------------------
  constexpr int B_ITEMS = BN * 4;
  constexpr int B_PER_THREAD = (B_ITEMS + 255) / 256;
  const char* wbase = reinterpret_cast<const char*>(p.w);
  uint4 breg[B_PER_THREAD];
  auto load_b = [&](long long step) {
#pragma unroll
    for (int i = 0; i < B_PER_THREAD; ++i) {
      const int item = tid + i * 256;
      const int row = item >> 2, qq = item & 3;
      const int co = cb * BN + row;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (item < B_ITEMS && co < cout_pad)
        v = *reinterpret_cast<const uint4*>(wbase + ((step * cout_pad + co) * BK) * (long long)sizeof(T) + qq * 16);
      breg[i] = v;
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < B_PER_THREAD; ++i) {
      const int item = tid + i * 256;
      const int row = item >> 2, qq = item & 3;
      if (item < B_ITEMS)
        *reinterpret_cast<uint4*>(ldsB + ((size_t)buf * BN + row) * CONV_ROWB + qq * 16) = breg[i];
    }
  };

  // ---- input patch staging (fused prologue) --------------------------------------------------------------------------
  const bool vec_ok = (p.Cin % VECW == 0) && (p.x_ld % VECW == 0) && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0);
  const T* xin = reinterpret_cast<const T*>(p.x);
  const int sq = tid & 3;  // this thread's 16-byte slot inside a row (constant over its items)
  auto stage_a = [&](int chunk) {
    const int c0 = chunk * BK + sq * VECW;
    // (round 3: every load goes to a clamped / substitute address unconditionally and the VALUE is selected -- hipcc branches around a
    //  conditional load and waits at each join: the 2 * VECW table loads and the patch rows were one dependent round trip each, ~20 us per
    //  launch of a small 1x1 convolution whatever its size (rocprofv3: C3 latent UNet, C4 training step))
    float sc[VECW], sh[VECW];
    {
      const float* ps = p.pre_scale ? p.pre_scale + (long long)n * p.Cin : reinterpret_cast<const float*>(p.w);
      const float* ph = p.pre_scale ? p.pre_shift + (long long)n * p.Cin : reinterpret_cast<const float*>(p.w);
#pragma unroll
      for (int i = 0; i < VECW; ++i) {
        const int c = c0 + i;
        const int cc = p.pre_scale ? (c < p.Cin ? c : p.Cin - 1) : 0;
        const float a = ps[cc], b2 = ph[cc];
        sc[i] = c < p.Cin ? a : 0.f;
        sh[i] = c < p.Cin ? b2 : 0.f;
      }
    }
    constexpr int UA = MF * NFR > 4 ? 1 : 2;  // patch rows requested per wait (the 256-accumulator tiles keep their three waves per SIMD)
    // geometry of one patch row: source pointer (clamped in range) and validity
    auto place = [&](int pv, bool& ok) __attribute__((always_inline)) -> const T* {
      const int pc = pv % pW;
      const int t1 = pv / pW;
      const int pb = t1 % pH, pa = t1 / pH;
      int ud = ud0 + pa, uh = uh0 + pb, uw = uw0 + pc;
      ok = (ud >= 0) & (ud < Dv) & (uh >= 0) & (uh < Hv) & (uw >= 0) & (uw < Wv) & (c0 < p.Cin);
      if (p.in_mode == 1) { ud /= p.fd; uh /= p.fh; uw /= p.fw; }
      else if (p.in_mode == 2) {
        ok = ok && (ud % p.fd == 0) && (uh % p.fh == 0) && (uw % p.fw == 0);
        ud /= p.fd; uh /= p.fh; uw /= p.fw;
      }
      return ok ? xin + ((((long long)n * p.Ds + ud) * p.Hs + uh) * p.Ws + uw) * p.x_ld + c0 : xin;
    };
    auto finish = [&](float (&v)[VECW], bool ok, int pv) __attribute__((always_inline)) {
      if (p.pre_scale) {
#pragma unroll
        for (int i = 0; i < VECW; ++i) v[i] = v[i] * sc[i] + sh[i];
      }
      if (p.pre_act) {
        conv_act_vec(v, p.pre_act, PRECISE);
      }
#pragma unroll
      for (int i = 0; i < VECW; ++i) if (!ok || c0 + i >= p.Cin) v[i] = 0.f;  // padding and padded channels contribute nothing
      *reinterpret_cast<uint4*>(ldsA + (size_t)pv * CONV_ROWB + sq * 16) = Vec16<T>::pack(v);
    };
    if (vec_ok) {  // (uniform)
      for (int pv0 = tid >> 2; pv0 < P; pv0 += 64 * UA) {
        float v[UA][VECW];
        bool okk[UA];
#pragma unroll
        for (int u = 0; u < UA; ++u) {
          const T* src = place(pv0 + 64 * u < P ? pv0 + 64 * u : pv0, okk[u]);
          Vec16<T>::unpack(*reinterpret_cast<const uint4*>(src), v[u]);
        }
#pragma unroll
        for (int u = 0; u < UA; ++u)
          if (pv0 + 64 * u < P) finish(v[u], okk[u], pv0 + 64 * u);
      }
    } else {
      for (int pv = tid >> 2; pv < P; pv += 64) {
        bool ok;
        const T* src = place(pv, ok);
        float v[VECW];
#pragma unroll
        for (int i = 0; i < VECW; ++i) v[i] = ElemIO<T>::ld(src + ((ok && c0 + i < p.Cin) ? i : 0));
        finish(v, ok, pv);
      }
    }
  };

  // ---- main loop ------------------------------------------------------------------------------------------------------
  load_b(0);
  long long step = 0;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    stage_a(chunk);  // (one call site: the staging code is instantiated once.)  Every wave passed the barrier that ended the previous chunk's last tap
    if (chunk == 0) store_b(0);
    __syncthreads();
    for (int kd_i = 0; kd_i < p.kd; ++kd_i)
      for (int kh_i = 0; kh_i < p.kh; ++kh_i)
        for (int kw_i = 0; kw_i < p.kw; ++kw_i, ++step) {
          const bool more = step + 1 < total_steps;
          if (more) load_b(step + 1);
          const int tap_off = ((kd_i * p.dd * pH + kh_i * p.dh) * pW + kw_i * p.dw) * CONV_ROWB;
          const char* bsrc = ldsB + (size_t)(step & 1) * BN * CONV_ROWB;
          uint4 xf[MF], wf[NFR];
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) xf[mf] = *reinterpret_cast<const uint4*>(ldsA + aoff[mf] + tap_off);
#pragma unroll
          for (int nf = 0; nf < NFR; ++nf) wf[nf] = *reinterpret_cast<const uint4*>(bsrc + boff[nf]);
#pragma unroll
          for (int nf = 0; nf < NFR; ++nf)
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) Mma<T>::run(wf[nf], xf[mf], acc[nf][mf]);
          if (more) store_b((int)((step + 1) & 1));
          __syncthreads();
        }
  }

  // ---- epilogue -------------------------------------------------------------------------------------------------------
  T* yout = reinterpret_cast<T*>(p.y);
  const T* res = reinterpret_cast<const T*>(p.res);
  const bool st_vec = (p.Cout % 4 == 0) && (p.y_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.y) & (4 * sizeof(T) - 1)) == 0) &&
                      (!res || ((p.res_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.res) & (4 * sizeof(T) - 1)) == 0)));
  // per-channel addends of this lane's output channels and (vector form) its residual rows: requested up front through substitute pointers
  constexpr bool RES_ALL = MF * NFR <= 4;
  constexpr int ANF = RES_ALL ? NFR : 1;
  float bt[ANF][4], rt[ANF][4];
  const float* bsrc = p.bias ? p.bias : reinterpret_cast<const float*>(p.w);
  const float* rsrc = p.rowvec ? p.rowvec + (long long)n * p.rowvec_bstride : reinterpret_cast<const float*>(p.w);
  auto load_addends = [&](int nf, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = cb * BN + (wn * NFR + nf) * 16 + q * 4 + r;
      const int cc = co < p.Cout ? co : p.Cout - 1;
      bt[slot][r] = bsrc[p.bias ? cc : 0];
      rt[slot][r] = rsrc[p.rowvec ? cc : 0];
    }
  };
  if (RES_ALL) {
#pragma unroll
    for (int nf = 0; nf < NFR; ++nf) load_addends(nf, nf);
  }
  // residual rows (vector form): the small tiles request all of theirs up front; the 256-accumulator tiles keep the element-wise form below
  // (holding their MF x NFR = 16 vectors cost 110 registers and two of their three waves per SIMD)
  constexpr int RMF = RES_ALL ? MF : 1;
  float resv[RMF][NFR][4];
  auto load_res = [&](int mf, int slot) __attribute__((always_inline)) {
    const int m = (wm * MF + mf) * 16 + l15;
    const int a = m >> (p.lth + p.ltw), bb = (m >> p.ltw) & (th - 1), c = m & (tw - 1);
    const int od = od0 + a, oh = oh0 + bb, ow = ow0 + c;
    const bool vok = (od < p.Do) & (oh < p.Ho) & (ow < p.Wo);
    const long long vox = vok ? (((long long)n * p.Do + od) * p.Ho + oh) * p.Wo + ow : 0;
#pragma unroll
    for (int nf = 0; nf < NFR; ++nf) {
      const int co = cb * BN + (wn * NFR + nf) * 16 + q * 4;
      const T* rp = res + vox * p.res_ld + (co < p.Cout ? co : 0);
      if (sizeof(T) == 4) {
        const float4 rv = *reinterpret_cast<const float4*>(rp);
        resv[slot][nf][0] = rv.x; resv[slot][nf][1] = rv.y; resv[slot][nf][2] = rv.z; resv[slot][nf][3] = rv.w;
      } else {
        const uint2 rv = *reinterpret_cast<const uint2*>(rp);
        resv[slot][nf][0] = __uint_as_float(rv.x << 16); resv[slot][nf][1] = __uint_as_float(rv.x & 0xffff0000u);
        resv[slot][nf][2] = __uint_as_float(rv.y << 16); resv[slot][nf][3] = __uint_as_float(rv.y & 0xffff0000u);
      }
    }
  };
  if (RES_ALL && st_vec && res) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) load_res(mf, mf);
  }
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int m = (wm * MF + mf) * 16 + l15;
    const int a = m >> (p.lth + p.ltw), bb = (m >> p.ltw) & (th - 1), c = m & (tw - 1);
    const int od = od0 + a, oh = oh0 + bb, ow = ow0 + c;
    if (od >= p.Do || oh >= p.Ho || ow >= p.Wo) continue;
    const long long vox = (((long long)n * p.Do + od) * p.Ho + oh) * p.Wo + ow;
#pragma unroll
    for (int nf = 0; nf < NFR; ++nf) {
      const int co = cb * BN + (wn * NFR + nf) * 16 + q * 4;
      if (co >= p.Cout) continue;
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = acc[nf][mf][r];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (co + r < p.Cout) {
          if (RES_ALL) {
            if (p.bias) o[r] += bt[RES_ALL ? nf : 0][r];
            if (p.rowvec) o[r] += rt[RES_ALL ? nf : 0][r];
          } else {  // (the 256-accumulator tiles keep the register-lean element-wise form: large problems, throughput-bound)
            if (p.bias) o[r] += p.bias[co + r];
            if (p.rowvec) o[r] += p.rowvec[(long long)n * p.rowvec_bstride + co + r];
          }
        }
      }
      if (st_vec) {
        if (res) {
          if (RES_ALL) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] += resv[RES_ALL ? mf : 0][nf][r];
          } else if (sizeof(T) == 4) {
            const float4 rv = *reinterpret_cast<const float4*>(res + vox * p.res_ld + co);
            o[0] += rv.x; o[1] += rv.y; o[2] += rv.z; o[3] += rv.w;
          } else {
            const uint2 rv = *reinterpret_cast<const uint2*>(res + vox * p.res_ld + co);
            o[0] += __uint_as_float(rv.x << 16); o[1] += __uint_as_float(rv.x & 0xffff0000u);
            o[2] += __uint_as_float(rv.y << 16); o[3] += __uint_as_float(rv.y & 0xffff0000u);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = conv_post_act_lean(o[r], p.post_act);
        if (sizeof(T) == 4) {
          *reinterpret_cast<float4*>(yout + vox * p.y_ld + co) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
          uint2 pk;
          pk.x = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
          pk.y = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
          *reinterpret_cast<uint2*>(yout + vox * p.y_ld + co) = pk;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (co + r < p.Cout) {
            float v = o[r];
            if (res) v += ElemIO<T>::ld(res + vox * p.res_ld + co + r);
            ElemIO<T>::st(yout + vox * p.y_ld + co + r, conv_post_act_lean(v, p.post_act));
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
struct ConvCfg { int WM, WN, MF, NFR; };
static const ConvCfg kCfgs[] = {
    {4, 1, 4, 4},  // 0: 256 voxels x  64 channels
    {2, 2, 4, 4},  // 1: 128 voxels x 128 channels
    {1, 4, 4, 1},  // 2:  64 voxels x  64 channels (strided / big-halo / tiny problems)
    {4, 1, 4, 1},  // 3: 256 voxels x  16 channels (few output channels)
    {2, 2, 2, 2},  // 4:  64 voxels x  64 channels, 2x2 waves
};
static const int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

// fast stride-1 path (conv_fast.hip): cfg 5 = 256 voxels x 64 channels, cfg 6 = 256 voxels x 128 channels
extern "C" long long gm_conv_fast_lds_bytes(const GmConvDesc* d, int bn);
extern "C" long long gm_conv_fast_max_patch(int wn);
extern "C" int gm_conv_fast_launch(const GmConvDesc* dp, int wn, unsigned nblocks, void* stream);
#define CONV_CFG_FAST64 5
#define CONV_CFG_FAST128 6
#define CONV_CFG_FAST_LAST 10
extern "C" int gm_conv_fast_variant_geometry(int variant, int* voxels, int* channels, int* threads);
static inline bool conv_is_fast(int cfg) { return cfg >= CONV_CFG_FAST64 && cfg <= CONV_CFG_FAST_LAST; }
static inline int conv_fast_variant(int cfg) { return cfg - CONV_CFG_FAST64 + 1; }
static inline int conv_fast_bn(int cfg) { int v = 0, c = 0, t = 0; gm_conv_fast_variant_geometry(conv_fast_variant(cfg), &v, &c, &t); return c; }

// LDS-DMA 3x3x3 kernel (conv_dma.hip): cfg 11 = 256 voxels (4x4x16) x 64 channels, no fused prologue
#define CONV_CFG_DMA 11
extern "C" long long gm_conv_dma_lds_bytes(int stride);
extern "C" int gm_conv_dma_eligible(const GmConvDesc* d);
extern "C" int gm_conv_dma_launch(const GmConvDesc* dp, unsigned nblocks, void* stream);
extern "C" int gm_conv_dma_variant(int cfg);
// 14: 4 waves x 64 voxels; 15: stride 2; 16: 512 voxels, 16 waves; 17: sub-pixel up-sampling; 18: 512 voxels, 8 waves x 64 voxels; 19: 512 voxels x 128 channels
// (21 / 22 / 23: the three tile structures on v_mfma_f32_32x32x16_bf16 of rounds 4-5 -- each verified and measured equal or slower in time, and in round 6
//  costlier in joules per launch on every C2 shape (profiles/r06_taploop_energy.txt) -- live under experiments/conv_mw, conv_w8, conv_w4; the ids stay retired)
// 24: conv_sn.hip -- small volumes, K-complete on 16-channel output blocks (256 voxels x 16 channels per work-group, no split-K, epilogue + statistics in the kernel)
// 25: the same kernel over images (3x3 convolutions carried as depth-1 volumes): 16 x 16 pixels x 16 channels per work-group
#define CONV_CFG_SN 24
#define CONV_CFG_SN2D 25
extern "C" int gm_conv_sn_eligible(const GmConvDesc* d);
extern "C" long long gm_conv_sn_lds_bytes(int cfg);
static inline bool conv_is_sn(int cfg) { return cfg == CONV_CFG_SN || cfg == CONV_CFG_SN2D; }
// cfg 25 with stride 2 walks the stride-1 grid (2 n - 1 positions per axis) and stores its even positions (conv_sn.hip)
static inline long long conv_sn_walk(const GmConvDesc* d, long long n) { return (d->cfg == CONV_CFG_SN2D && d->sh == 2) ? 2 * n - 1 : n; }
static inline bool conv_is_dma(int cfg) { return cfg == CONV_CFG_DMA || (cfg >= 14 && cfg <= 19) || conv_is_sn(cfg); }
// HBM-bound end convolutions (conv_edge.hip): cfg 12 = C_in <= 4, cfg 13 = C_out == 1; 4x4x16 tiles
#define CONV_CFG_CIN 12
#define CONV_CFG_COUT1 13
extern "C" int gm_conv_cin_eligible(const GmConvDesc* d);
extern "C" long long gm_conv_cin_lds_bytes(const GmConvDesc* d);
extern "C" int gm_conv_cin_launch(const GmConvDesc* dp, unsigned nblocks, void* stream);
extern "C" int gm_conv_cout1_eligible(const GmConvDesc* d);
extern "C" long long gm_conv_cout1_lds_bytes();
extern "C" int gm_conv_cout1_launch(const GmConvDesc* dp, unsigned nblocks, void* stream);
// cfg 20 = C_out == 1, marching along depth: a work-group walks 2^ltd planes of an 8 x 2^ltw output column (conv_edge.hip)
#define CONV_CFG_COUT1M 20
extern "C" int gm_conv_cout1m_eligible(const GmConvDesc* d);
extern "C" long long gm_conv_cout1m_lds_bytes(const GmConvDesc* d);
extern "C" int gm_conv_cout1m_launch(const GmConvDesc* dp, unsigned nblocks, void* stream);
static inline bool conv_is_edge(int cfg) { return cfg == CONV_CFG_CIN || cfg == CONV_CFG_COUT1 || cfg == CONV_CFG_COUT1M; }

static bool conv_fast_eligible(const GmConvDesc& d) {
  const int vecw = d.dtype == GM_F32 ? 4 : 8;
  const long long td = 1 << d.ltd, th = 1 << d.lth, tw = 1 << d.ltw;
  const long long P = (td + d.kd - 1) * (th + d.kh - 1) * (tw + d.kw - 1);
  return d.sd == 1 && d.sh == 1 && d.sw == 1 && d.dd == 1 && d.dh == 1 && d.dw == 1 && (d.in_mode == 0 || d.in_mode == 1) &&
         d.Cin % vecw == 0 && d.x_ld % vecw == 0 && (reinterpret_cast<uintptr_t>(d.x) & 15) == 0 &&
         P <= gm_conv_fast_max_patch(conv_fast_variant(d.cfg)) &&
         (long long)d.N * d.Ds * d.Hs * d.Ws < (1LL << 31);
}

extern "C" int gm_conv_cfg_tile(int cfg, int* bm, int* bn) {
  if (cfg == 15) { *bm = 128; *bn = 64; return 0; }
  if (cfg == 16 || cfg == 18) { *bm = 512; *bn = 64; return 0; }
  if (cfg == 19) { *bm = 512; *bn = 128; return 0; }
  if (conv_is_sn(cfg)) { *bm = 256; *bn = 16; return 0; }
  if (conv_is_dma(cfg) || cfg == CONV_CFG_CIN) { *bm = 256; *bn = 64; return 0; }
  if (cfg == CONV_CFG_COUT1) { *bm = 256; *bn = 16; return 0; }
  if (cfg == CONV_CFG_COUT1M) { *bm = 256; *bn = 16; return 0; }  // (per plane; the depth extent of a work-group is GmConvDesc.ltd)
  if (conv_is_fast(cfg)) { int t = 0; return gm_conv_fast_variant_geometry(conv_fast_variant(cfg), bm, bn, &t); }
  if (cfg < 0 || cfg >= kNumCfgs) return -1;
  *bm = kCfgs[cfg].WM * kCfgs[cfg].MF * 16;
  *bn = kCfgs[cfg].WN * kCfgs[cfg].NFR * 16;
  return 0;
}

template <typename T, int WM, int WN, int MF, int NFR>
static int launch_conv(const GmConvDesc& d, size_t smem, long long nblocks, hipStream_t st) {
  static bool attr_set = false;  // one per instantiation
  auto kern = conv_igemm_kernel<T, WM, WN, MF, NFR>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { (void)hipGetLastError(); }
    attr_set = true;
  }
  kern<<<dim3((unsigned)nblocks), 256, smem, st>>>(d);
  return 0;
}

template <typename T>
static int dispatch_conv(const GmConvDesc& d, size_t smem, long long nblocks, hipStream_t st) {
  switch (d.cfg) {
    case 0: return launch_conv<T, 4, 1, 4, 4>(d, smem, nblocks, st);
    case 1: return launch_conv<T, 2, 2, 4, 4>(d, smem, nblocks, st);
    case 2: return launch_conv<T, 1, 4, 4, 1>(d, smem, nblocks, st);
    case 3: return launch_conv<T, 4, 1, 4, 1>(d, smem, nblocks, st);
    case 4: return launch_conv<T, 2, 2, 2, 2>(d, smem, nblocks, st);
    default: return -1;
  }
}

// LDS bytes a launch with this descriptor needs (-1: invalid descriptor)
extern "C" long long gm_conv_lds_bytes(const GmConvDesc* d) {
  if (d && (d->skip_x[0] || d->x2) && !conv_is_dma(d->cfg)) return -1;  // the fused 1x1 shortcut / second source exist in the LDS-DMA kernels only
  if (d && conv_is_sn(d->cfg)) return gm_conv_sn_eligible(d) ? gm_conv_sn_lds_bytes(d->cfg) : -1;
  if (d && conv_is_dma(d->cfg)) return gm_conv_dma_eligible(d) ? gm_conv_dma_lds_bytes(gm_conv_dma_variant(d->cfg)) : -1;
  if (d && d->cfg == CONV_CFG_CIN) return gm_conv_cin_eligible(d) ? gm_conv_cin_lds_bytes(d) : -1;
  if (d && d->cfg == CONV_CFG_COUT1) return gm_conv_cout1_eligible(d) ? gm_conv_cout1_lds_bytes() : -1;
  if (d && d->cfg == CONV_CFG_COUT1M) return gm_conv_cout1m_eligible(d) ? gm_conv_cout1m_lds_bytes(d) : -1;
  if (d && conv_is_fast(d->cfg)) {
    if (!conv_fast_eligible(*d)) return -1;
    return gm_conv_fast_lds_bytes(d, conv_fast_bn(d->cfg));
  }
  if (!d || d->cfg < 0 || d->cfg >= kNumCfgs) return -1;
  const int td = 1 << d->ltd, th = 1 << d->lth, tw = 1 << d->ltw;
  const long long pD = (long long)(td - 1) * d->sd + (long long)(d->kd - 1) * d->dd + 1;
  const long long pH = (long long)(th - 1) * d->sh + (long long)(d->kh - 1) * d->dh + 1;
  const long long pW = (long long)(tw - 1) * d->sw + (long long)(d->kw - 1) * d->dw + 1;
  const int bn = kCfgs[d->cfg].WN * kCfgs[d->cfg].NFR * 16;
  return pD * pH * pW * CONV_ROWB + 2LL * bn * CONV_ROWB;
}

// ---- split-K combine: y = act(sum_s partial[s] + bias + skip_bias + timestep row + residual), per-channel output statistics ----------------
// grid (blocks per sample, N); a block owns SK_ROWS consecutive voxels of one sample; thread = (row in flight r0, 16-byte channel vector cv)
#define SK_THREADS 256
#define SK_ITERS 8
template <typename T>
__global__ __launch_bounds__(SK_THREADS) void conv_splitk_combine_kernel(const GmConvDesc p, long long V, int iters) {
  constexpr int VECW = 16 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char sk_smem[];
  const int C = p.Cout, CV = C / VECW, R = SK_THREADS / CV;
  float* part_s = reinterpret_cast<float*>(sk_smem);  // [R][C]
  float* part_q = part_s + (size_t)R * C;
  const int n = blockIdx.y, t = threadIdx.x, cv = t % CV, r0 = t / CV, c = cv * VECW;
  const long long row_begin = (long long)blockIdx.x * R * iters;
  long long row_end = row_begin + (long long)R * iters;
  if (row_end > V) row_end = V;
  const long long nv = (long long)p.N * V;
  // Every load of a row is requested before the first wait (round 3): the per-channel addends come through substitute pointers (a branch per
  // `if (p.bias)` element made 24 dependent round trips), the slices of a row are read with a compile-time bound (a run-time slice loop
  // waited for each slice in turn) and the residual row with them.  Same order of additions as before: bit-identical results.
  constexpr int SK_MAX = 8;  // host: ksplit <= 8 (ops.SPLITK_MAX)
  float add[VECW], ss[VECW], sq[VECW];
  {
    const float* dummy = p.kpartial;  // any readable address
    const float* b0 = p.bias ? p.bias + c : dummy;
    const float* b1 = p.skip_bias ? p.skip_bias + c : dummy;
    const float* b2 = p.rowvec ? p.rowvec + (long long)n * p.rowvec_bstride + c : dummy;
    const bool lanes = r0 < R;
    float t0[VECW], t1[VECW], t2[VECW];
#pragma unroll
    for (int i = 0; i < VECW; ++i) { t0[i] = b0[p.bias && lanes ? i : 0]; t1[i] = b1[p.skip_bias && lanes ? i : 0]; t2[i] = b2[p.rowvec && lanes ? i : 0]; }
#pragma unroll
    for (int i = 0; i < VECW; ++i) {
      float a = 0.f;
      if (lanes) {
        if (p.bias) a += t0[i];
        if (p.skip_bias) a += t1[i];
        if (p.rowvec) a += t2[i];
      }
      add[i] = a; ss[i] = 0.f; sq[i] = 0.f;
    }
  }
  if (r0 < R) {
    const T* resp = p.res ? reinterpret_cast<const T*>(p.res) : reinterpret_cast<const T*>(p.y);
    const long long res_ld = p.res ? p.res_ld : p.y_ld;
    for (long long r = row_begin + r0; r < row_end; r += R) {
      const long long vox = (long long)n * V + r;
      float4 part[SK_MAX][VECW / 4];
#pragma unroll
      for (int s = 0; s < SK_MAX; ++s) {
        const float* src = p.kpartial + ((long long)(s < p.ksplit ? s : 0) * nv + vox) * C + c;
#pragma unroll
        for (int i = 0; i < VECW / 4; ++i) part[s][i] = *reinterpret_cast<const float4*>(src + 4 * i);
      }
      float rv[VECW];
      Vec16<T>::unpack(*reinterpret_cast<const uint4*>(resp + vox * res_ld + c), rv);  // (without a residual: the output row, not used)
      float o[VECW];
#pragma unroll
      for (int i = 0; i < VECW; ++i) o[i] = add[i];
#pragma unroll
      for (int s = 0; s < SK_MAX; ++s) {
        if (s < p.ksplit) {
#pragma unroll
          for (int i = 0; i < VECW / 4; ++i) { o[4 * i] += part[s][i].x; o[4 * i + 1] += part[s][i].y; o[4 * i + 2] += part[s][i].z; o[4 * i + 3] += part[s][i].w; }
        }
      }
      if (p.res) {
#pragma unroll
        for (int i = 0; i < VECW; ++i) o[i] += rv[i];
      }
      if (p.post_act) {
#pragma unroll
        for (int i = 0; i < VECW; ++i) o[i] = conv_post_act_lean(o[i], p.post_act);
      }
      const uint4 raw = Vec16<T>::pack(o);
      *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.y) + vox * p.y_ld + c) = raw;
      if (p.stats) {  // statistics of the values as stored (rounded to T), like the convolution epilogues
        Vec16<T>::unpack(raw, o);
#pragma unroll
        for (int i = 0; i < VECW; ++i) { ss[i] += o[i]; sq[i] += o[i] * o[i]; }
      }
    }
  }
  if (!p.stats) return;
  if (r0 < R) {
#pragma unroll
    for (int i = 0; i < VECW; ++i) { part_s[(size_t)r0 * C + c + i] = ss[i]; part_q[(size_t)r0 * C + c + i] = sq[i]; }
  }
  __syncthreads();
  for (int ch = t; ch < C; ch += SK_THREADS) {
    double a = 0.0, b2 = 0.0;
    for (int r = 0; r < R; ++r) { a += (double)part_s[(size_t)r * C + ch]; b2 += (double)part_q[(size_t)r * C + ch]; }
    *reinterpret_cast<double2*>(p.stats + (((long long)blockIdx.x * p.N + n) * C + ch) * 2) = make_double2(a, b2);
  }
}

static bool conv_splitk_no_empty_slice(const GmConvDesc& d) {  // the kernel deals ceil(nchunks / ksplit) chunks to each slice
  const int nchunks = d.Cin / (d.dtype == GM_F32 ? 16 : 32);
  if (d.ksplit < 2 || d.ksplit > nchunks) return false;
  const int cps = (nchunks + d.ksplit - 1) / d.ksplit;
  return (d.ksplit - 1) * cps < nchunks;
}
static bool conv_splitk_ok(const GmConvDesc& d) {  // configuration 11 (3x3x3, stride 1), vector epilogue, <= 256 channel vectors per row
  const int vecw = d.dtype == GM_F32 ? 4 : 8;
  return d.cfg == CONV_CFG_DMA && d.ksplit > 1 && d.Cout % vecw == 0 && d.Cout / vecw <= SK_THREADS && d.y_ld % vecw == 0 &&
         (reinterpret_cast<uintptr_t>(d.y) & 15) == 0 && (!d.res || (d.res_ld % vecw == 0 && (reinterpret_cast<uintptr_t>(d.res) & 15) == 0)) &&
         conv_splitk_no_empty_slice(d) && d.in_mode != 3;
}
// Rows a combine block owns = (rows in flight R) x (iterations).  Round 3 used SK_ITERS = 8 always: an 8^3 x 256-channel output was 8 blocks walking 8
// dependent round trips each (12 us per launch, profiles/r03_c3_latent_unet_kernel_stats_v27.csv).  The iteration count now follows the volume: as
// few as keep the launch at <= gm_stats_compact_slots() = 256 blocks per sample (the statistic table stays small enough to skip the compaction
// launch), at most SK_ITERS.  The block count is also the S of the statistic partials (gm_conv_stats_slots).
static int conv_splitk_iters(const GmConvDesc& d) {
  const int vecw = d.dtype == GM_F32 ? 4 : 8;
  const long long R = SK_THREADS / (d.Cout / vecw), V = (long long)d.Do * d.Ho * d.Wo;
  long long it = (V + R * 256 - 1) / (R * 256);
  return (int)(it < 1 ? 1 : (it > SK_ITERS ? SK_ITERS : it));
}
static long long conv_splitk_rows_per_block(const GmConvDesc& d) {
  const int vecw = d.dtype == GM_F32 ? 4 : 8;
  return (long long)(SK_THREADS / (d.Cout / vecw)) * conv_splitk_iters(d);
}
extern "C" long long gm_conv_splitk_workspace_bytes(const GmConvDesc* d) {
  if (!d || !conv_splitk_ok(*d)) return 0;
  return (long long)d->ksplit * d->N * d->Do * d->Ho * d->Wo * d->Cout * (long long)sizeof(float);
}

// Number S of per-tile partials a launch with this descriptor (tile configuration chosen) writes into GmConvDesc.stats, laid out
// [S][N][Cout][2] fp64 -- or 0 when this configuration does not fuse the output statistics (generic kernels, the C_out = 1 head, a
// ragged channel count that takes the scalar epilogue): the caller then runs gm_gn_channel_stats over the stored tensor instead.
extern "C" long long gm_conv_stats_slots(const GmConvDesc* d) {
  if (!d) return 0;
  const bool fused = conv_is_fast(d->cfg) || conv_is_dma(d->cfg) || d->cfg == CONV_CFG_CIN;
  if (!fused) return 0;
  if (conv_is_sn(d->cfg))  // four channels per lane straight from the accumulators: whatever the kernel takes, it also counts
    return gm_conv_sn_eligible(d) ? (long long)((d->Do + (1 << d->ltd) - 1) >> d->ltd) * ((conv_sn_walk(d, d->Ho) + (1 << d->lth) - 1) >> d->lth) * ((conv_sn_walk(d, d->Wo) + 15) >> 4) : 0;
  const int vecw = d->dtype == GM_F32 ? 4 : 8;
  const bool lds_epilogue = (d->Cout % vecw == 0) && (d->y_ld % vecw == 0) && ((reinterpret_cast<uintptr_t>(d->y) & 15) == 0) &&
                            (!d->res || ((d->res_ld % vecw == 0) && ((reinterpret_cast<uintptr_t>(d->res) & 15) == 0)));
  if (!lds_epilogue) return 0;
  if (d->ksplit > 1 && d->kpartial && conv_splitk_ok(*d)) {  // split-K: the combine kernel writes one partial per block of rows
    const long long rpb = conv_splitk_rows_per_block(*d), V = (long long)d->Do * d->Ho * d->Wo;
    return (V + rpb - 1) / rpb;
  }
  const bool subpixel = d->cfg == 17;
  const long long De = subpixel ? d->Ds : d->Do, He = subpixel ? d->Hs : d->Ho, We = subpixel ? d->Ws : d->Wo;
  const long long ntd = (De + (1 << d->ltd) - 1) >> d->ltd, nth = (He + (1 << d->lth) - 1) >> d->lth, ntw = (We + (1 << d->ltw) - 1) >> d->ltw;
  return ntd * nth * ntw * (subpixel ? 8 : 1);
}

extern "C" int gm_conv_forward(const GmConvDesc* dp, void* stream) {
  GM_REQUIRE(dp, "null descriptor");
  const GmConvDesc& d = *dp;
  GM_REQUIRE(d.x && d.w && d.y, "null tensor pointer");
  const bool fast = conv_is_fast(d.cfg);
  const bool dma = conv_is_dma(d.cfg);
  const bool edge = conv_is_edge(d.cfg);
  GM_REQUIRE(fast || dma || edge || (d.cfg >= 0 && d.cfg < kNumCfgs), "bad tile configuration");
  GM_REQUIRE((d.pre_scale == nullptr) == (d.pre_shift == nullptr), "pre_scale and pre_shift go together");
  GM_REQUIRE(d.N >= 0 && d.Cin > 0 && d.Cout > 0, "bad channel / batch count");
  GM_REQUIRE(d.kd > 0 && d.kh > 0 && d.kw > 0 && d.sd > 0 && d.sh > 0 && d.sw > 0 && d.dd > 0 && d.dh > 0 && d.dw > 0, "bad kernel geometry");
  GM_REQUIRE(d.in_mode == 0 || d.in_mode == 3 || (d.fd > 0 && d.fh > 0 && d.fw > 0), "bad input-mode factors");
  GM_REQUIRE(d.in_mode != 3 || d.cfg == 17, "in_mode 3 (sub-pixel up-sampling) is implemented by configuration 17 only");
  if (d.N == 0 || d.Do == 0 || d.Ho == 0 || d.Wo == 0) return 0;
  int bm = 0, bn = 0;
  gm_conv_cfg_tile(d.cfg, &bm, &bn);
  GM_REQUIRE(d.cfg == CONV_CFG_COUT1M || (1 << (d.ltd + d.lth + d.ltw)) == bm, "tile dims do not match the configuration");
  GM_REQUIRE(!fast || conv_fast_eligible(d), "geometry is not eligible for the fast stride-1 kernel");
  GM_REQUIRE(!dma || gm_conv_dma_eligible(dp), "geometry is not eligible for the LDS-DMA 3x3x3 kernel");
  GM_REQUIRE(dma || d.skip_x[0] == nullptr, "the fused 1x1 shortcut needs the LDS-DMA kernel (cfg 11)");
  GM_REQUIRE(dma || d.x2 == nullptr, "a second input source (virtual channel concatenation) needs an LDS-DMA configuration");
  GM_REQUIRE(!edge || (d.cfg == CONV_CFG_CIN ? gm_conv_cin_eligible(dp) : (d.cfg == CONV_CFG_COUT1M ? gm_conv_cout1m_eligible(dp) : gm_conv_cout1_eligible(dp))),
             "geometry is not eligible for the C_in<=4 / C_out==1 kernels");
  const long long smem = gm_conv_lds_bytes(dp);
  GM_REQUIRE(smem > 0 && smem <= 160 * 1024, "tile needs more than 160 KiB of LDS");
  GM_REQUIRE(d.stats == nullptr || gm_conv_stats_slots(dp) > 0, "this tile configuration does not fuse the output statistics");
  // cfg 17 (sub-pixel up-sampling): the tiles walk the low-resolution grid, once per output parity
  const bool subpixel = d.cfg == 17;
  const long long De = subpixel ? d.Ds : d.Do, He = subpixel ? d.Hs : conv_sn_walk(dp, d.Ho), We = subpixel ? d.Ws : conv_sn_walk(dp, d.Wo);
  const long long ntd = (De + (1 << d.ltd) - 1) >> d.ltd, nth = (He + (1 << d.lth) - 1) >> d.lth, ntw = (We + (1 << d.ltw) - 1) >> d.ltw;
  const long long ncb = (d.Cout + bn - 1) / bn;
  const bool splitk = d.ksplit > 1 && d.kpartial != nullptr;
  GM_REQUIRE(!splitk || conv_splitk_ok(d), "split-K needs configuration 11, a vector epilogue and ksplit <= the number of K chunks");
  GM_REQUIRE(!splitk || d.ksplit <= 8, "split-K: at most 8 slices (the combine kernel reads them with a compile-time bound)");
  const long long nblocks = (long long)d.N * ntd * nth * ntw * ncb * (subpixel ? 4 : 1) * (splitk ? d.ksplit : 1);  // sub-pixel: per (d, h) parity; a work item covers both W parities
  GM_REQUIRE(nblocks < (1LL << 31), "grid too large");
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (edge) {
    rc = d.cfg == CONV_CFG_CIN ? gm_conv_cin_launch(dp, (unsigned)nblocks, stream)
                               : (d.cfg == CONV_CFG_COUT1M ? gm_conv_cout1m_launch(dp, (unsigned)nblocks, stream) : gm_conv_cout1_launch(dp, (unsigned)nblocks, stream));
    GM_REQUIRE(rc == 0, "unsupported dtype");
    GM_LAUNCH_CHECK();
  }
  if (dma) {
    rc = gm_conv_dma_launch(dp, (unsigned)nblocks, stream);
    GM_REQUIRE(rc == 0, "unsupported dtype");
    if (splitk) {  // the slices' partial sums -> output (+ bias / row / residual / activation / statistics)
      const long long V = (long long)d.Do * d.Ho * d.Wo, rpb = conv_splitk_rows_per_block(d);
      const int vecw = d.dtype == GM_F32 ? 4 : 8;
      const size_t smem2 = (size_t)(SK_THREADS / (d.Cout / vecw)) * d.Cout * 2 * sizeof(float);
      dim3 grid((unsigned)((V + rpb - 1) / rpb), (unsigned)d.N);
      if (d.dtype == GM_F32) conv_splitk_combine_kernel<float><<<grid, SK_THREADS, smem2, st>>>(d, V, conv_splitk_iters(d));
      else conv_splitk_combine_kernel<bf16_raw><<<grid, SK_THREADS, smem2, st>>>(d, V, conv_splitk_iters(d));
    }
    GM_LAUNCH_CHECK();
  }
  if (fast) {
    rc = gm_conv_fast_launch(dp, conv_fast_variant(d.cfg), (unsigned)nblocks, stream);
    GM_REQUIRE(rc == 0, "unsupported dtype");
    GM_LAUNCH_CHECK();
  }
  if (d.dtype == GM_F32) rc = dispatch_conv<float>(d, (size_t)smem, nblocks, st);
  else if (d.dtype == GM_BF16) rc = dispatch_conv<bf16_raw>(d, (size_t)smem, nblocks, st);
  else GM_FAIL(-2, "unsupported dtype");
  GM_REQUIRE(rc == 0, "dispatch failed");
  GM_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight packing: reference layouts -> [chunk][tap][Cout_pad][BK] in the compute dtype (zero padded).
//   transposed == 0: src[Cout][Cin][kd][kh][kw]            (nn.ConvNd, nn.Linear with k = 1)
//   transposed == 1: src[Cin][Cout][kd][kh][kw], taps flipped (nn.ConvTransposeNd as a gather over zero-inserted input)
// ---------------------------------------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void pack_conv_weight_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int Cout,
                                                              int Cin, int kd, int kh, int kw, int transposed, int BK,
                                                              long long total) {
  const int T_taps = kd * kh * kw;
  const int cout_pad = (Cout + 15) & ~15;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int kk = (int)(i % BK);
    long long r = i / BK;
    const int co = (int)(r % cout_pad); r /= cout_pad;
    const int tap = (int)(r % T_taps);
    const int chunk = (int)(r / T_taps);
    const int ci = chunk * BK + kk;
    float v = 0.f;
    if (co < Cout && ci < Cin) {
      const int st = transposed ? (T_taps - 1 - tap) : tap;
      const long long sidx = transposed ? (((long long)ci * Cout + co) * T_taps + st) : (((long long)co * Cin + ci) * T_taps + st);
      v = ElemIO<TS>::ld(src + sidx);
    }
    ElemIO<TD>::st(dst + i, v);
  }
}

extern "C" long long gm_packed_conv_weight_elems(int Cout, int Cin, int kd, int kh, int kw, int dtype) {
  const int BK = dtype == GM_F32 ? 16 : 32;
  const long long nchunks = (Cin + BK - 1) / BK;
  const long long cout_pad = (Cout + 15) & ~15;
  return nchunks * kd * kh * kw * cout_pad * BK;
}

extern "C" int gm_pack_conv_weight(const void* src, int src_dtype, void* dst, int dst_dtype, int Cout, int Cin, int kd,
                                   int kh, int kw, int transposed, void* stream) {
  GM_REQUIRE(src && dst, "null pointer");
  const int BK = dst_dtype == GM_F32 ? 16 : 32;
  const long long total = gm_packed_conv_weight_elems(Cout, Cin, kd, kh, kw, dst_dtype);
  long long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (src_dtype == GM_F32 && dst_dtype == GM_F32)
    pack_conv_weight_kernel<float, float><<<(int)g, 256, 0, st>>>((const float*)src, (float*)dst, Cout, Cin, kd, kh, kw, transposed, BK, total);
  else if (src_dtype == GM_F32 && dst_dtype == GM_BF16)
    pack_conv_weight_kernel<float, bf16_raw><<<(int)g, 256, 0, st>>>((const float*)src, (bf16_raw*)dst, Cout, Cin, kd, kh, kw, transposed, BK, total);
  else if (src_dtype == GM_BF16 && dst_dtype == GM_F32)
    pack_conv_weight_kernel<bf16_raw, float><<<(int)g, 256, 0, st>>>((const bf16_raw*)src, (float*)dst, Cout, Cin, kd, kh, kw, transposed, BK, total);
  else if (src_dtype == GM_BF16 && dst_dtype == GM_BF16)
    pack_conv_weight_kernel<bf16_raw, bf16_raw><<<(int)g, 256, 0, st>>>((const bf16_raw*)src, (bf16_raw*)dst, Cout, Cin, kd, kh, kw, transposed, BK, total);
  else GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------------
// The 8 parity images of 2x2x2 kernels of a sub-pixel convolution (conv_dma.hip, in_mode 3), packed back to back, in ONE launch straight
// from the parameter: image par = (pd << 2) | (ph << 1) | pw, sub-tap (a, b, c) of an image = the SUM of the source taps selected by the
// per-axis bit masks m[parity][sub-tap] (bit k = source tap k of the K-tap kernel; fp32 sum, rounded once to the compute dtype):
//   nearest-2x + 3x3x3 (Upsample, diffusion_model_unet.py:572-585): parity 0 -> ({0}, {1, 2}), parity 1 -> ({0, 1}, {2});
//   stride-2 transposed convolution / stride-2 data gradient (vqvae.py:244-261, autoencoderkl.py:54-63, torch autograd through a stride-2
//   nn.Conv3d): one tap or none per sub-tap (ops.stride2_subpixel_taps), swap_io = 1: the source is [Cin][Cout][K][K][K].
// Round 2 built every image with torch slicing + one pack launch per parity (~80 tiny launches per strided convolution and training step).
// ---------------------------------------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void pack_subpixel_weight_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int Cout, int Cin, int K, int swap_io,
                                                                  int m00, int m01, int m10, int m11, int BK, long long per_image, long long total) {
  const int cout_pad = (Cout + 15) & ~15;
  const int K3 = K * K * K;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int par = (int)(i / per_image);
    long long r = i - (long long)par * per_image;
    const int kk = (int)(r % BK); r /= BK;
    const int co = (int)(r % cout_pad); r /= cout_pad;
    const int tap = (int)(r % 8);
    const int chunk = (int)(r / 8);
    const int ci = chunk * BK + kk;
    float v = 0.f;
    if (co < Cout && ci < Cin) {
      const int pd = (par >> 2) & 1, ph = (par >> 1) & 1, pw = par & 1;
      const int a = (tap >> 2) & 1, b = (tap >> 1) & 1, c = tap & 1;
      const int md = pd ? (a ? m11 : m10) : (a ? m01 : m00), mh = ph ? (b ? m11 : m10) : (b ? m01 : m00), mw = pw ? (c ? m11 : m10) : (c ? m01 : m00);
      const long long base = (swap_io ? ((long long)ci * Cout + co) : ((long long)co * Cin + ci)) * K3;
      for (int kd = 0; kd < K; ++kd)
        if ((md >> kd) & 1)
          for (int kh = 0; kh < K; ++kh)
            if ((mh >> kh) & 1)
              for (int kw = 0; kw < K; ++kw)
                if ((mw >> kw) & 1) v += ElemIO<TS>::ld(src + base + (kd * K + kh) * K + kw);
    }
    ElemIO<TD>::st(dst + i, v);
  }
}

extern "C" int gm_pack_subpixel_weight(const void* src, int src_dtype, void* dst, int dst_dtype, int Cout, int Cin, int K, int swap_io, int m00,
                                       int m01, int m10, int m11, void* stream) {
  GM_REQUIRE(src && dst, "null pointer");
  GM_REQUIRE(K >= 1 && K <= 4 && ((m00 | m01 | m10 | m11) >> K) == 0, "kernel extent 1..4, masks over its taps");
  const int BK = dst_dtype == GM_F32 ? 16 : 32;
  const long long per_image = gm_packed_conv_weight_elems(Cout, Cin, 2, 2, 2, dst_dtype), total = 8 * per_image;
  long long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipStream_t st = (hipStream_t)stream;
#define GM_PSW(TS, TD) pack_subpixel_weight_kernel<TS, TD><<<(int)g, 256, 0, st>>>((const TS*)src, (TD*)dst, Cout, Cin, K, swap_io, m00, m01, m10, m11, BK, per_image, total)
  if (src_dtype == GM_F32 && dst_dtype == GM_F32) GM_PSW(float, float);
  else if (src_dtype == GM_F32 && dst_dtype == GM_BF16) GM_PSW(float, bf16_raw);
  else if (src_dtype == GM_BF16 && dst_dtype == GM_F32) GM_PSW(bf16_raw, float);
  else if (src_dtype == GM_BF16 && dst_dtype == GM_BF16) GM_PSW(bf16_raw, bf16_raw);
  else GM_FAIL(-2, "unsupported dtype");
#undef GM_PSW
  GM_LAUNCH_CHECK();
}

