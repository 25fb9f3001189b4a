// Backward kernels of the fused forward ops (what a training step -- reference
// tutorials/generative/distributed_training/ddpm_training_ddp.py:249-270, engines/trainer.py:258-270 -- needs beyond the forward path):
//   gm_conv_wgrad          dW[co][ci][tap] = sum_v gy[v][co] * x[v * s - p + tap][ci]          (nn.ConvNd weight gradient)
//   gm_gn_bwd_stats / gm_gn_bwd_finalize / gm_gn_bwd_apply      GroupNorm (+ SiLU) backward     (nn.GroupNorm + nn.SiLU)
//   gm_stats_colsum        bias gradient from the per-channel statistics of gy
// The data gradient of a convolution needs no kernel of its own: it is the transposed convolution of gy with the same weight
// (ops.conv(transposed=True)), i.e. the forward LDS-DMA kernel at stride 1.
//
// Weight gradient on MFMA.  The contraction runs over VOXELS, which is the row index of both N[D]HWC operands, while an MFMA
// lane wants 8 consecutive k values (4 for fp32) in one 16-byte register group.  Both tiles are therefore transposed once, on
// their way into LDS: gyT[co][tile voxel] and xT[ci][halo-patch voxel], W innermost.  A k-step is then one 32-voxel W run:
//   A fragment = gyT[co = l15][row][8q .. 8q+7]                        one aligned ds_read_b128
//   B fragment = xT[ci = l15][row + kh][8q + kw .. 8q + kw + 7]        an UNALIGNED 8-element run: read the aligned 16 bytes plus
//                the next dword and funnel-shift in registers (kw = 1: four v_alignbyte_b32, kw = 2: register renaming), so
//                the three kw taps of a row share one read and every read stays naturally aligned (a misaligned b128 replays
//                at 64 cycles, cdna_hip_programming.md 6).  Stride 2 stores even and odd W columns as separate runs.
// A work-group owns one depth tap kd, a 64-channel block of C_out and of C_in (32 for fp32) and all 9 (kh, kw) taps -- 72
// accumulator VGPRs per lane --, walks a share of the voxel tiles (split K) and writes its partial sums; a second kernel
// reduces the splits in a fixed order (deterministic, no atomics) into the nn.ConvNd weight layout.
#include "conv_common.h"

struct GmWgradDesc {
  const void* x; long long x_ld;      // [N][Ds][Hs][Ws][Cin]
  const void* gy; long long gy_ld;    // [N][Do][Ho][Wo][Cout]
  float* dw;                          // fp32 [Cout][Cin][kd][kh][kw]
  void* workspace; long long workspace_bytes;
  int N, Cin, Cout, Ds, Hs, Ws, Do, Ho, Wo;
  int kd, kh, kw;                     // 1 or 3 per axis; kh == kw
  int stride;                         // 1 or 2, every axis (depth too when kd == 3)
  int pd, ph, pw;                     // low-side padding
  int dtype;                          // of x and gy
  int accumulate;                     // 0: dw is overwritten, 1: added to
};

static constexpr int wg_pad_pitch(int bytes) { return ((bytes - 16 + 255) / 256) * 256 + 16; }  // smallest >= bytes that is 16 mod 256

template <typename T> struct WgTraits;
template <> struct WgTraits<bf16_raw> { static constexpr int CIB = 64; };
template <> struct WgTraits<float> { static constexpr int CIB = 32; };

// the B fragments of the KHW taps of one patch row, from aligned reads
template <typename T, int S, int KHW> struct WgShift;
template <int S, int KHW> struct WgShift<bf16_raw, S, KHW> {
  static __device__ __forceinline__ void run(const char* rowp, uint4 (&b)[KHW]) {  // rowp: this lane's aligned 16-byte group
    if constexpr (KHW == 1) { b[0] = *reinterpret_cast<const uint4*>(rowp); } else {
    const uint4 v = *reinterpret_cast<const uint4*>(rowp);
    const uint32_t d4 = *reinterpret_cast<const uint32_t*>(rowp + 16);
    const uint4 sh1 = make_uint4(__builtin_amdgcn_alignbyte(v.y, v.x, 2), __builtin_amdgcn_alignbyte(v.z, v.y, 2),
                                 __builtin_amdgcn_alignbyte(v.w, v.z, 2), __builtin_amdgcn_alignbyte(d4, v.w, 2));
    if constexpr (S == 1) {
      b[0] = v;
      b[1] = sh1;
      b[KHW - 1] = make_uint4(v.y, v.z, v.w, d4);
    } else {  // even plane: kw = 0 -> index w, kw = 2 -> index w + 1; odd plane (40 elements further): kw = 1 -> index w
      b[0] = v;
      b[KHW - 1] = sh1;
      b[1] = *reinterpret_cast<const uint4*>(rowp + 40 * 2);
    }
    }
  }
};
template <int S, int KHW> struct WgShift<float, S, KHW> {
  static __device__ __forceinline__ void run(const char* rowp, uint4 (&b)[KHW]) {
    if constexpr (KHW == 1) { b[0] = *reinterpret_cast<const uint4*>(rowp); } else {
    const uint4 v = *reinterpret_cast<const uint4*>(rowp);
    const uint2 e = *reinterpret_cast<const uint2*>(rowp + 16);
    if constexpr (S == 1) {
      b[0] = v;
      b[1] = make_uint4(v.y, v.z, v.w, e.x);
      b[KHW - 1] = make_uint4(v.z, v.w, e.x, e.y);
    } else {
      b[0] = v;
      b[KHW - 1] = make_uint4(v.y, v.z, v.w, e.x);
      b[1] = *reinterpret_cast<const uint4*>(rowp + 40 * 4);
    }
    }
  }
};

template <typename T> __device__ __forceinline__ void wg_store_elem(char* p, uint32_t word, int half);
template <> __device__ __forceinline__ void wg_store_elem<bf16_raw>(char* p, uint32_t word, int half) {
  *reinterpret_cast<bf16_raw*>(p) = (bf16_raw)(half ? (word >> 16) : (word & 0xffffu));
}
template <> __device__ __forceinline__ void wg_store_elem<float>(char* p, uint32_t word, int) { *reinterpret_cast<uint32_t*>(p) = word; }

// S: stride; KHW: kernel extent along H and W (1 or 3); TD x TH x 32: output-voxel tile.  KHW == 1 is "flat": the rows of the two
// operands are walked as one long W axis (1x1 convolutions and nn.Linear layers: the spatial structure does not matter).
template <typename T, int S, int KHW, int TD, int TH>
__global__ __launch_bounds__(512, 2) void conv_wgrad_kernel(const GmWgradDesc p, float* __restrict__ partial, int nsplit, long long tiles_total) {
  constexpr int VECW = ConvTraits<T>::VECW;
  constexpr int ES = (int)sizeof(T);
  constexpr int KSTEP = 4 * VECW, TW = 32, KSR = TW / KSTEP;
  constexpr int CIB = WgTraits<T>::CIB, COB = 64;
  constexpr int NCIF = CIB / 16, CGR = 8 / NCIF, COFW = (COB / 16) / CGR;  // bf16: 4 ci fragments x 2 co groups of 2; fp32: 2 x 4 of 1
  constexpr int PH = S * (TH - 1) + KHW, NR = TD * PH;                    // patch rows held for this work-group's depth tap
  constexpr int PLW = 40, RW = S * PLW;                                   // elements per plane / per patch row
  constexpr int PCOLS = S * (TW - 1) + KHW;
  constexpr int XPITCH = wg_pad_pitch(NR * RW * ES), GROWS = TD * TH, GPITCH = wg_pad_pitch(GROWS * TW * ES);
  constexpr int NT = KHW * KHW;
  constexpr int CVX = CIB / VECW, CVG = COB / VECW;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xT = smem;                       // [CIB][XPITCH]
  char* gT = smem + CIB * XPITCH;        // [COB][GPITCH]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int ncib = (p.Cin + CIB - 1) / CIB, ncob = (p.Cout + COB - 1) / COB;
  unsigned b = blockIdx.x;
  const int cib = b % ncib; b /= ncib;
  const int cob = b % ncob; b /= ncob;
  const int kdi = b % p.kd; b /= p.kd;
  const int split = b;
  const int ci0 = cib * CIB, co0 = cob * COB;
  const int cif = wave % NCIF, cog = wave / NCIF;

  const int ntd = (p.Do + TD - 1) / TD, nth = (p.Ho + TH - 1) / TH, ntw = (p.Wo + TW - 1) / TW;
  const long long rows_flat = (long long)p.N * p.Do * p.Ho * p.Wo;  // KHW == 1

  f32x4_t acc[NT][COFW];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int cf = 0; cf < COFW; ++cf) acc[t][cf] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const T* xin = reinterpret_cast<const T*>(p.x);
  const T* gin = reinterpret_cast<const T*>(p.gy);

  for (long long tile = split; tile < tiles_total; tile += nsplit) {
    int n = 0, od0 = 0, oh0 = 0, ow0 = 0;
    if (KHW != 1) {
      long long t = tile;
      ow0 = (int)(t % ntw) * TW; t /= ntw;
      oh0 = (int)(t % nth) * TH; t /= nth;
      od0 = (int)(t % ntd) * TD; t /= ntd;
      n = (int)t;
    }
    // ---- stage xT: (patch voxel, 16-byte channel vector) items, the channel vector fastest over the lanes (coalesced rows) ----------
    for (int it = tid; it < NR * PCOLS * CVX; it += 512) {
      const int cv = it % CVX, pv = it / CVX;
      const int pc = pv % PCOLS, pr = pv / PCOLS;
      const int dd = pr / PH, hh = pr - dd * PH;
      bool ok;
      long long vox;
      if (KHW == 1) {
        vox = (tile * GROWS + pr) * TW + pc;
        ok = vox < rows_flat;
      } else {
        const int ud = S * (od0 + dd) - p.pd + kdi, uh = S * oh0 - p.ph + hh, uw = S * ow0 - p.pw + pc;
        ok = (ud >= 0) & (ud < p.Ds) & (uh >= 0) & (uh < p.Hs) & (uw >= 0) & (uw < p.Ws);
        vox = (((long long)n * p.Ds + ud) * p.Hs + uh) * p.Ws + uw;
      }
      const int c = ci0 + cv * VECW;
      ok = ok & (c < p.Cin);  // host: Cin % VECW == 0
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (ok) v = *reinterpret_cast<const uint4*>(xin + vox * p.x_ld + c);
      const int col = S == 2 ? (pc & 1) * PLW + (pc >> 1) : pc;
      char* dst = xT + (size_t)(cv * VECW) * XPITCH + (pr * RW + col) * ES;
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < VECW; ++i) wg_store_elem<T>(dst + (size_t)i * XPITCH, w[ES == 2 ? i >> 1 : i], i & 1);
    }
    // ---- stage gT -------------------------------------------------------------------------------------------------------------
    for (int it = tid; it < GROWS * TW * CVG; it += 512) {
      const int cv = it % CVG, pv = it / CVG;
      const int w_ = pv % TW, gr = pv / TW;
      bool ok;
      long long vox;
      if (KHW == 1) {
        vox = (tile * GROWS + gr) * TW + w_;
        ok = vox < rows_flat;
      } else {
        const int od = od0 + gr / TH, oh = oh0 + gr % TH, ow = ow0 + w_;
        ok = (od < p.Do) & (oh < p.Ho) & (ow < p.Wo);
        vox = (((long long)n * p.Do + od) * p.Ho + oh) * p.Wo + ow;
      }
      const int c = co0 + cv * VECW;
      ok = ok & (c < p.Cout);  // host: Cout % VECW == 0
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (ok) v = *reinterpret_cast<const uint4*>(gin + vox * p.gy_ld + c);
      char* dst = gT + (size_t)(cv * VECW) * GPITCH + (gr * TW + w_) * ES;
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < VECW; ++i) wg_store_elem<T>(dst + (size_t)i * GPITCH, w[ES == 2 ? i >> 1 : i], i & 1);
    }
    __syncthreads();
    // ---- multiply: one k-step per (tile row, 32-voxel run) ----------------------------------------------------------------------
    const char* arow = gT + (size_t)(cog * COFW * 16 + l15) * GPITCH + q * VECW * ES;
    const char* brow = xT + (size_t)(cif * 16 + l15) * XPITCH + q * VECW * ES;
#pragma unroll
    for (int gr = 0; gr < GROWS; ++gr) {
      const int d = gr / TH, h = gr % TH;
#pragma unroll
      for (int ks = 0; ks < KSR; ++ks) {
        uint4 a[COFW];
#pragma unroll
        for (int cf = 0; cf < COFW; ++cf)
          a[cf] = *reinterpret_cast<const uint4*>(arow + (size_t)cf * 16 * GPITCH + (gr * TW + ks * KSTEP) * ES);
#pragma unroll
        for (int kh = 0; kh < KHW; ++kh) {
          uint4 bf[KHW];
          WgShift<T, S, KHW>::run(brow + ((d * PH + S * h + kh) * RW + ks * KSTEP) * ES, bf);
#pragma unroll
          for (int kw = 0; kw < KHW; ++kw)
#pragma unroll
            for (int cf = 0; cf < COFW; ++cf) Mma<T>::run(a[cf], bf[kw], acc[kh * KHW + kw][cf]);
        }
      }
    }
    __syncthreads();
  }

  // ---- partial sums: D[co = 4q + r][ci = l15] -> partial[split][kd][tap][co_pad][ci_pad] ------------------------------------------
  const int cop = ncob * COB, cip = ncib * CIB;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int cf = 0; cf < COFW; ++cf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + (cog * COFW + cf) * 16 + 4 * q + r, ci = ci0 + cif * 16 + l15;
        partial[((((long long)split * p.kd + kdi) * NT + t) * cop + co) * cip + ci] = acc[t][cf][r];
      }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int nsplit, int KD, int NT,
                                                          int Cout, int Cin, int cop, int cip, int accumulate) {
  const long long total = (long long)Cout * Cin * KD * NT;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % NT);
    long long r = i / NT;
    const int a = (int)(r % KD); r /= KD;
    const int ci = (int)(r % Cin);
    const int co = (int)(r / Cin);
    float s = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) s += partial[((((long long)sp * KD + a) * NT + t) * cop + co) * cip + ci];
    dw[i] = accumulate ? dw[i] + s : s;
  }
}

struct WgPlan { int variant; int td, th; int cib; long long tiles; int nsplit; long long partial_elems; int ncob, ncib, nt; };

static bool wgrad_plan(const GmWgradDesc& d, WgPlan& pl) {
  if (d.dtype != GM_F32 && d.dtype != GM_BF16) return false;
  const int vecw = d.dtype == GM_F32 ? 4 : 8;
  if (d.Cin % vecw || d.Cout % vecw || d.x_ld % vecw || d.gy_ld % vecw) return false;
  if ((reinterpret_cast<uintptr_t>(d.x) & 15) || (reinterpret_cast<uintptr_t>(d.gy) & 15)) return false;
  if (d.kh != d.kw || (d.kh != 1 && d.kh != 3) || (d.kd != 1 && d.kd != 3) || (d.kd == 3 && d.kh != 3)) return false;
  if (d.stride != 1 && d.stride != 2) return false;
  pl.cib = d.dtype == GM_F32 ? 32 : 64;
  pl.ncob = (d.Cout + 63) / 64;
  pl.ncib = (d.Cin + pl.cib - 1) / pl.cib;
  if (d.kh == 1) {
    if (d.stride != 1 || d.kd != 1 || d.pd || d.ph || d.pw) return false;
    pl.variant = 3; pl.td = 2; pl.th = 4; pl.nt = 1;
    const long long rows = (long long)d.N * d.Do * d.Ho * d.Wo;
    pl.tiles = (rows + 255) / 256;
  } else {
    pl.nt = 9;
    if (d.stride == 2) { pl.variant = 2; pl.td = 1; pl.th = 4; }
    else if (d.kd == 1 && d.Do == 1) { pl.variant = 1; pl.td = 1; pl.th = 8; }
    else { pl.variant = 0; pl.td = 2; pl.th = 4; }
    pl.tiles = (long long)d.N * ((d.Do + pl.td - 1) / pl.td) * ((d.Ho + pl.th - 1) / pl.th) * ((d.Wo + 31) / 32);
  }
  const long long base = (long long)d.kd * pl.ncob * pl.ncib;
  long long ns = (768 + base - 1) / base;  // ~3 work-groups per CU in total (one is resident per CU: ~100 KiB of LDS)
  if (ns > pl.tiles) ns = pl.tiles;
  if (ns < 1) ns = 1;
  pl.nsplit = (int)ns;
  pl.partial_elems = ns * d.kd * pl.nt * (pl.ncob * 64LL) * ((long long)pl.ncib * pl.cib);
  return true;
}

extern "C" long long gm_conv_wgrad_workspace_bytes(const GmWgradDesc* d) {
  WgPlan pl;
  if (!d || !wgrad_plan(*d, pl)) return -1;
  return pl.partial_elems * 4;
}

template <typename T, int S, int KHW, int TD, int TH>
static void launch_wgrad(const GmWgradDesc& d, const WgPlan& pl, hipStream_t st) {
  constexpr int ES = (int)sizeof(T), CIB = WgTraits<T>::CIB;
  constexpr int PH = S * (TH - 1) + KHW, NR = TD * PH, RW = S * 40;
  constexpr size_t smem = (size_t)CIB * wg_pad_pitch(NR * RW * ES) + (size_t)64 * wg_pad_pitch(TD * TH * 32 * ES);
  static_assert(smem <= 160 * 1024, "tile does not fit the LDS");
  static bool attr_set = false;
  auto kern = conv_wgrad_kernel<T, S, KHW, TD, TH>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) (void)hipGetLastError();
    attr_set = true;
  }
  const unsigned grid = (unsigned)((long long)pl.nsplit * d.kd * pl.ncob * pl.ncib);
  kern<<<grid, 512, smem, st>>>(d, reinterpret_cast<float*>(d.workspace), pl.nsplit, pl.tiles);
}

template <typename T>
static void dispatch_wgrad(const GmWgradDesc& d, const WgPlan& pl, hipStream_t st) {
  switch (pl.variant) {
    case 0: launch_wgrad<T, 1, 3, 2, 4>(d, pl, st); break;
    case 1: launch_wgrad<T, 1, 3, 1, 8>(d, pl, st); break;
    case 2: launch_wgrad<T, 2, 3, 1, 4>(d, pl, st); break;
    default: launch_wgrad<T, 1, 1, 2, 4>(d, pl, st); break;
  }
}

extern "C" int gm_conv_wgrad(const GmWgradDesc* dp, void* stream) {
  GM_REQUIRE(dp && dp->x && dp->gy && dp->dw, "null pointer");
  const GmWgradDesc& d = *dp;
  WgPlan pl;
  GM_REQUIRE(wgrad_plan(d, pl), "geometry not covered: kernel 1 or 3 per axis, stride 1 or 2, channel counts multiples of one 16-byte vector");
  GM_REQUIRE(d.workspace && d.workspace_bytes >= pl.partial_elems * 4, "workspace too small (gm_conv_wgrad_workspace_bytes)");
  GM_REQUIRE((long long)pl.nsplit * d.kd * pl.ncob * pl.ncib < (1LL << 31), "grid too large");
  hipStream_t st = (hipStream_t)stream;
  if (d.N == 0 || d.Do * d.Ho * d.Wo == 0) {
    if (!d.accumulate) (void)hipMemsetAsync(d.dw, 0, sizeof(float) * d.Cout * d.Cin * d.kd * d.kh * d.kw, st);
    GM_LAUNCH_CHECK();
  }
  if (d.dtype == GM_F32) dispatch_wgrad<float>(d, pl, st);
  else dispatch_wgrad<bf16_raw>(d, pl, st);
  {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) GM_FAIL((int)e, hipGetErrorString(e));
  }
  const long long total = (long long)d.Cout * d.Cin * d.kd * pl.nt;
  long long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  wgrad_reduce_kernel<<<(int)g, 256, 0, st>>>(reinterpret_cast<const float*>(d.workspace), d.dw, pl.nsplit, d.kd, pl.nt, d.Cout, d.Cin,
                                              pl.ncob * 64, pl.ncib * pl.cib, d.accumulate);
  GM_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// GroupNorm (+ SiLU) backward.  Forward: y = act(x * scale[n, c] + shift[n, c]) with scale = rstd * gamma, shift = beta - mean * scale.
// With g = gy * act'(x * scale + shift):
//   dx = A[n, c] * g + B[n, group] * x + C[n, group],  A = rstd * gamma,  B = -rstd^2 * M2,  C = -rstd * M1 + rstd^2 * M2 * mean,
//   M1 = mean_group(gamma * g),  M2 = rstd * (mean_group(gamma * g * x) - mean * M1),
//   dgamma[c] = sum_n rstd * (sum_v g x - mean * sum_v g),  dbeta[c] = sum_n sum_v g.
// (reference: torch.nn.functional.group_norm / SiLU autograd as used by diffusion_model_unet.py:623-690)
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float silu_grad(float z) {
  const float s = 1.0f / (1.0f + expf(-z));
  return s * (1.0f + z * (1.0f - s));
}

// out[slot][n][c] += {sum_v g, sum_v g * x} (fp64 atomics, table zeroed by the caller); grid (nblk, N)
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const T* __restrict__ x, long long x_ld, const T* __restrict__ gy, long long gy_ld,
                                                          const float* __restrict__ scale, const float* __restrict__ shift, long long ss_ld,
                                                          long long V, int C, int act, int rows_per_block, double* __restrict__ out) {
  const int n = blockIdx.y, blk = blockIdx.x;
  const long long r0 = (long long)blk * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > V) r1 = V;
  // thread <-> channel (strided), rows walked serially: consecutive threads read consecutive channels of a row
  for (int c = threadIdx.x; c < C; c += 256) {
    const float sc = scale[n * ss_ld + c], sh = shift[n * ss_ld + c];
    float a = 0.f, b2 = 0.f;
    double da = 0.0, db = 0.0;
    int cnt = 0;
    for (long long r = r0; r < r1; ++r) {
      const long long row = (long long)n * V + r;
      const float xv = ElemIO<T>::ld(x + row * x_ld + c);
      float g = ElemIO<T>::ld(gy + row * gy_ld + c);
      if (act == 1) g *= silu_grad(xv * sc + sh);
      a += g; b2 += g * xv;
      if (++cnt == 256) { da += (double)a; db += (double)b2; a = 0.f; b2 = 0.f; cnt = 0; }
    }
    da += (double)a; db += (double)b2;
    double* dst = out + (((long long)(blk % GM_STAT_SLOTS) * gridDim.y + n) * C + c) * 2;
    atomicAdd(dst, da);
    atomicAdd(dst + 1, db);
  }
}

extern "C" int gm_gn_bwd_stats(const void* x, long long x_ld, const void* gy, long long gy_ld, const float* scale, const float* shift,
                               long long ss_ld, int N, long long V, int C, int act, double* out, int dtype, void* stream) {
  GM_REQUIRE(x && gy && scale && shift && out, "null pointer");
  GM_REQUIRE(N <= 65535, "batch too large");
  if (N == 0 || V == 0) return 0;
  long long nblk = (V + 511) / 512;
  if (nblk > 2048) nblk = 2048;
  const int rpb = (int)((V + nblk - 1) / nblk);
  nblk = (V + rpb - 1) / rpb;
  dim3 grid((unsigned)nblk, N);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GM_F32)
    gn_bwd_stats_kernel<float><<<grid, 256, 0, st>>>((const float*)x, x_ld, (const float*)gy, gy_ld, scale, shift, ss_ld, V, C, act, rpb, out);
  else if (dtype == GM_BF16)
    gn_bwd_stats_kernel<bf16_raw><<<grid, 256, 0, st>>>((const bf16_raw*)x, x_ld, (const bf16_raw*)gy, gy_ld, scale, shift, ss_ld, V, C, act, rpb, out);
  else
    GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// one wave per group; fwd = forward per-channel statistics {sum x, sum x^2}, bwd = {sum g, sum g x}, both [slots][N][C][2] fp64
__global__ __launch_bounds__(64) void gn_bwd_finalize_kernel(const double* __restrict__ fwd, const double* __restrict__ bwd, int N, int C, int G,
                                                            long long V, float eps, const float* __restrict__ gamma, float* __restrict__ A,
                                                            float* __restrict__ B, float* __restrict__ Cc, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const int cpg = C / G;
  const double m = (double)cpg * (double)V;
  auto slot_sum = [&](const double* t, int n, int c, int which) {
    double s = 0.0;
    for (int sl = 0; sl < GM_STAT_SLOTS; ++sl) s += t[(((long long)sl * N + n) * C + c) * 2 + which];
    return s;
  };
  for (int n = 0; n < N; ++n) {
    double sx = 0.0, sxx = 0.0;
    for (int j = lane; j < cpg; j += 64) { sx += slot_sum(fwd, n, g * cpg + j, 0); sxx += slot_sum(fwd, n, g * cpg + j, 1); }
    for (int o = 32; o > 0; o >>= 1) { sx += __shfl_xor(sx, o, 64); sxx += __shfl_xor(sxx, o, 64); }
    const double mean = sx / m;
    double var = sxx / m - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    double s1 = 0.0, s2 = 0.0;
    for (int j = lane; j < cpg; j += 64) {
      const int c = g * cpg + j;
      const double ga = gamma ? (double)gamma[c] : 1.0;
      const double sg = slot_sum(bwd, n, c, 0), sgx = slot_sum(bwd, n, c, 1);
      s1 += ga * sg;
      s2 += ga * (sgx - mean * sg);
      const double dg = rstd * (sgx - mean * sg);
      if (dgamma) dgamma[c] = (float)((n == 0 ? 0.0 : (double)dgamma[c]) + dg);   // the same lane owns channel c for every n
      if (dbeta) dbeta[c] = (float)((n == 0 ? 0.0 : (double)dbeta[c]) + sg);
    }
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    const double M1 = s1 / m, M2 = rstd * s2 / m;
    for (int j = lane; j < cpg; j += 64) {
      const int c = g * cpg + j;
      const double ga = gamma ? (double)gamma[c] : 1.0;
      A[(long long)n * C + c] = (float)(rstd * ga);
      B[(long long)n * C + c] = (float)(-rstd * rstd * M2);
      Cc[(long long)n * C + c] = (float)(-rstd * M1 + rstd * rstd * M2 * mean);
    }
  }
}

extern "C" int gm_gn_bwd_finalize(const double* fwd_stats, const double* bwd_stats, int N, int C, int G, long long V, float eps,
                                  const float* gamma, float* A, float* B, float* Cc, float* dgamma, float* dbeta, void* stream) {
  GM_REQUIRE(fwd_stats && bwd_stats && A && B && Cc, "null pointer");
  GM_REQUIRE(G > 0 && C % G == 0, "channels must be divisible by groups");
  if (N == 0) return 0;
  gn_bwd_finalize_kernel<<<G, 64, 0, (hipStream_t)stream>>>(fwd_stats, bwd_stats, N, C, G, V, eps, gamma, A, B, Cc, dgamma, dbeta);
  GM_LAUNCH_CHECK();
}

// dx = gy * act'(x * scale + shift) * A + x * B + Cc
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ x, long long x_ld, const T* __restrict__ gy, long long gy_ld,
                                                          T* __restrict__ dx, long long dx_ld, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, long long ss_ld, const float* __restrict__ A,
                                                          const float* __restrict__ B, const float* __restrict__ Cc, long long V, int C,
                                                          long long total, int act) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / C;
    const int c = (int)(i - row * C);
    const long long n = row / V;
    const float xv = ElemIO<T>::ld(x + row * x_ld + c);
    float g = ElemIO<T>::ld(gy + row * gy_ld + c);
    if (act == 1) g *= silu_grad(xv * scale[n * ss_ld + c] + shift[n * ss_ld + c]);
    ElemIO<T>::st(dx + row * dx_ld + c, g * A[n * C + c] + xv * B[n * C + c] + Cc[n * C + c]);
  }
}

extern "C" int gm_gn_bwd_apply(const void* x, long long x_ld, const void* gy, long long gy_ld, void* dx, long long dx_ld, const float* scale,
                               const float* shift, long long ss_ld, const float* A, const float* B, const float* Cc, int N, long long V, int C,
                               int act, int dtype, void* stream) {
  GM_REQUIRE(x && gy && dx && scale && shift && A && B && Cc, "null pointer");
  const long long total = (long long)N * V * C;
  if (total == 0) return 0;
  long long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GM_F32)
    gn_bwd_apply_kernel<float><<<(int)g, 256, 0, st>>>((const float*)x, x_ld, (const float*)gy, gy_ld, (float*)dx, dx_ld, scale, shift, ss_ld, A, B, Cc,
                                                       V, C, total, act);
  else if (dtype == GM_BF16)
    gn_bwd_apply_kernel<bf16_raw><<<(int)g, 256, 0, st>>>((const bf16_raw*)x, x_ld, (const bf16_raw*)gy, gy_ld, (bf16_raw*)dx, dx_ld, scale, shift,
                                                          ss_ld, A, B, Cc, V, C, total, act);
  else
    GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// out[c] = sum over slots and samples of stats[slot][n][c][0] (a bias gradient from the gm_gn_channel_stats table of gy), or, per sample,
// out[n][c] = sum over slots (the gradient of a per-sample row vector added by the convolution epilogue: the timestep embedding)
__global__ __launch_bounds__(256) void stats_colsum_kernel(const double* __restrict__ stats, int N, int C, float* __restrict__ out, int per_sample) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (per_sample ? N * C : C)) return;
  double s = 0.0;
  if (per_sample) {
    for (int sl = 0; sl < GM_STAT_SLOTS; ++sl) s += stats[((long long)sl * N * C + i) * 2];
  } else {
    for (long long j = 0; j < (long long)GM_STAT_SLOTS * N; ++j) s += stats[(j * C + i) * 2];
  }
  out[i] = (float)s;
}

extern "C" int gm_stats_colsum(const double* stats, int N, int C, float* out, int per_sample, void* stream) {
  GM_REQUIRE(stats && out, "null pointer");
  if (C == 0 || N == 0) return 0;
  const int total = per_sample ? N * C : C;
  stats_colsum_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(stats, N, C, out, per_sample);
  GM_LAUNCH_CHECK();
}
