// The two HBM-bound 3x3x3 convolutions at the ends of the 3-D networks (27 FLOP/B, graded against 8 TB/s):
//   conv_cin_kernel   C_in  <= 4 (DiffusionModelUNet.conv_in 1->64 at 128^3, AutoencoderKL encoder conv_in at 256^3):
//                     the 27 taps x C_in inputs ARE the GEMM K dimension (K = 27 -> one 32-deep MFMA step); the halo patch of a
//                     4x4x16 tile is 1.3 KB, the im2col operand is gathered from it in registers, the cost is the 64-channel
//                     output store (reference: networks/nets/diffusion_model_unet.py:1748-1756, autoencoderkl.py:343-352).
//   conv_cout1_kernel C_out == 1 (the `out` head GN -> SiLU -> conv 64->1, AutoencoderKL decoder's last conv):
//                     the 27 taps are the GEMM N dimension: Z[tap][v] = sum_c w[tap][c] * act(x[v][c]) for every voxel v of the
//                     halo patch (each voxel normalised, activated and multiplied exactly once, operands straight from global
//                     memory into MFMA layout), then out[v] = bias + sum_tap Z[tap][v + tap] is a 27-point gather over the fp32
//                     Z tile in LDS.  No 16x padding of the single output channel, no per-tap re-read of the activations
//                     (reference: diffusion_model_unet.py:1853-1867, autoencoderkl.py:590-597).
// The generic implicit-GEMM kernel ran these two at 0.37-0.5 TB/s (0.73 + 0.56 ms of a 26 ms forward).
#include "conv_epilogue.h"

// -----------------------------------------------------------------------------------------------------------------------
// C_in <= 4
// -----------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv_cin_kernel(const GmConvDesc p) {
  constexpr int VECW = ConvTraits<T>::VECW;
  constexpr int KB = 4 * VECW;                 // K values one Mma<T>::run consumes (32 bf16 / 16 fp32)
  constexpr int MF = 4, NFR = 4, NW = 4, BN = 64;
  constexpr int TD = 4, TH = 4, TW = 16, PH = TH + 2, PW = TW + 2, PROWS = (TD + 2) * PH * PW;  // 648
  constexpr int MAXK = 128;                    // 27 * C_in <= 108
  constexpr int MAXBLK = MAXK / KB;
  constexpr int WPITCH = MAXK * (int)sizeof(T) + 16;   // bytes per weight row [co][k], padded against bank conflicts

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* patch = reinterpret_cast<T*>(smem);                            // [PROWS][Cin]
  char* wlds = smem + ((PROWS * 4 * (int)sizeof(T) + 15) & ~15);    // [BN][WPITCH]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int ntd = (p.Do + TD - 1) / TD, nth = (p.Ho + TH - 1) / TH, ntw = (p.Wo + TW - 1) / TW;
  const int ncb = (p.Cout + BN - 1) / BN;
  unsigned b = xcd_remap(blockIdx.x, gridDim.x);
  const int cb = b % ncb; b /= ncb;
  const unsigned tile_id = b;  // (n, td, th, tw): statistics slot = tile_id modulo the tiles per sample
  const int tw_i = b % ntw; b /= ntw;
  const int th_i = b % nth; b /= nth;
  const int td_i = b % ntd; b /= ntd;
  const int n = b;
  const int od0 = td_i * TD, oh0 = th_i * TH, ow0 = tw_i * TW;
  const int Cin = p.Cin, K = 27 * Cin, nblk = (K + KB - 1) / KB;
  const int cout_pad = (p.Cout + 15) & ~15;
  constexpr int BK = ConvTraits<T>::BK;

  // ---- stage the patch (zero padded) and the weight block [co][k = tap * Cin + ci] -------------------------------------------
  const T* xin = reinterpret_cast<const T*>(p.x);
  for (int e = tid; e < PROWS * Cin; e += 256) {
    const int row = e / Cin, ci = e - row * Cin;
    const int pa = row / (PH * PW), rr = row - pa * (PH * PW), pb = rr / PW, pc = rr - pb * PW;
    const int ud = od0 - p.pd + pa, uh = oh0 - p.ph + pb, uw = ow0 - p.pw + pc;
    const bool ok = (ud >= 0) & (ud < p.Ds) & (uh >= 0) & (uh < p.Hs) & (uw >= 0) & (uw < p.Ws);
    const long long vox = ok ? (((long long)n * p.Ds + ud) * p.Hs + uh) * p.Ws + uw : 0;
    const T v = xin[vox * p.x_ld + ci];
    patch[e] = ok ? v : (T)0;
  }
  const T* wsrc = reinterpret_cast<const T*>(p.w);  // packed [chunk 0][tap][cout_pad][BK]
  for (int e = tid; e < BN * nblk * KB; e += 256) {
    const int col = e / (nblk * KB), k = e - col * (nblk * KB);
    const int co = cb * BN + col;
    const bool ok = (k < K) & (co < cout_pad);
    const int tap = ok ? k / Cin : 0, ci = ok ? k - tap * Cin : 0;
    const T v = wsrc[((long long)tap * cout_pad + (ok ? co : 0)) * BK + ci];
    *reinterpret_cast<T*>(wlds + (size_t)col * WPITCH + k * (int)sizeof(T)) = ok ? v : (T)0;
  }
  __syncthreads();

  // ---- this lane's K slots: element offset into the patch of k = blk*KB + q*VECW + i at tap (0,0,0) voxel, -1 beyond K ------------
  f32x4_t acc[NFR][MF];
#pragma unroll
  for (int nf = 0; nf < NFR; ++nf)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  int vrow[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int m = (wave * MF + mf) * 16 + l15;
    vrow[mf] = (((m >> 6) * PH + ((m >> 4) & 3)) * PW + (m & 15)) * Cin;
  }
  for (int blk = 0; blk < nblk; ++blk) {
    int koff[VECW];
#pragma unroll
    for (int i = 0; i < VECW; ++i) {
      const int k = blk * KB + q * VECW + i;
      const int tap = k / Cin, ci = k - tap * Cin;
      const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      koff[i] = k < K ? ((kd * PH + kh) * PW + kw) * Cin + ci : -1;
    }
    uint4 wf[NFR];
#pragma unroll
    for (int nf = 0; nf < NFR; ++nf)
      wf[nf] = *reinterpret_cast<const uint4*>(wlds + (size_t)(nf * 16 + l15) * WPITCH + (blk * KB + q * VECW) * (int)sizeof(T));
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      alignas(16) T g[VECW];
#pragma unroll
      for (int i = 0; i < VECW; ++i) {
        const T v = patch[vrow[mf] + (koff[i] >= 0 ? koff[i] : 0)];
        g[i] = koff[i] >= 0 ? v : (T)0;
      }
      const uint4 xf = *reinterpret_cast<const uint4*>(g);
#pragma unroll
      for (int nf = 0; nf < NFR; ++nf) Mma<T>::run(wf[nf], xf, acc[nf][mf]);
    }
  }

  // ---- epilogue: bias / timestep row / residual, 16-byte row stores, fused GroupNorm statistics (as conv_fast) -------------------
  __syncthreads();
  constexpr int EPASSES = (NFR * 16 * (int)sizeof(T) + 127) / 128;
  constexpr int CH_PER_PASS = 128 / (int)sizeof(T);
  float st_s[EPASSES][VECW], st_q[EPASSES][VECW];
#pragma unroll
  for (int e = 0; e < EPASSES; ++e)
#pragma unroll
    for (int i = 0; i < VECW; ++i) { st_s[e][i] = 0.f; st_q[e][i] = 0.f; }
  conv_epilogue_lds<T, MF, NFR>(p, acc, smem + (size_t)wave * MF * 16 * 144, n, wave * MF * 16, cb * BN, od0, oh0, ow0, lane, st_s, st_q);
  if (p.stats) {
    float* sst = reinterpret_cast<float*>(smem);  // [NW][64 channels][2]
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPASSES; ++e)
#pragma unroll
      for (int i = 0; i < VECW; ++i) {
        float a = st_s[e][i], b2 = st_q[e][i];
        a += __shfl_xor(a, 8, 64); b2 += __shfl_xor(b2, 8, 64);
        a += __shfl_xor(a, 16, 64); b2 += __shfl_xor(b2, 16, 64);
        a += __shfl_xor(a, 32, 64); b2 += __shfl_xor(b2, 32, 64);
        if (lane < 8) {
          const int ch = e * CH_PER_PASS + lane * VECW + i;
          sst[(wave * 64 + ch) * 2] = a;
          sst[(wave * 64 + ch) * 2 + 1] = b2;
        }
      }
    __syncthreads();
    if (tid < BN) {
      double a = 0.0, b2 = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        a += (double)sst[(w * 64 + tid) * 2];
        b2 += (double)sst[(w * 64 + tid) * 2 + 1];
      }
      const int co = cb * BN + tid;
      if (co < p.Cout) {
        const long long slot = tile_id % (unsigned)(ntd * nth * ntw);  // one plain store per (tile, channel): no atomics, fixed-order reduction later
        double* dst = p.stats + ((slot * p.N + n) * p.Cout + co) * 2;
        *reinterpret_cast<double2*>(dst) = make_double2(a, b2);
      }
    }
  }
}

// -----------------------------------------------------------------------------------------------------------------------
// C_out == 1
// -----------------------------------------------------------------------------------------------------------------------
template <typename T, int KS>  // KS = C_in / BK 64-byte channel steps (2 or 4 for bf16 64/128, 4 or 8 for fp32)
__global__ __launch_bounds__(256, 2) void conv_cout1_kernel(const GmConvDesc p) {
  constexpr int VECW = ConvTraits<T>::VECW;
  constexpr int BK = ConvTraits<T>::BK;
  constexpr int TD = 4, TH = 4, TW = 16, PH = TH + 2, PW = TW + 2, PROWS = (TD + 2) * PH * PW;  // 648 patch voxels
  constexpr int NFRAG = (PROWS + 15) / 16;     // 41 voxel fragments
  constexpr int ZP = 660;                      // floats per Z row: 4*ZP mod 64 == 16 -> the four row groups of a store hit distinct banks
  constexpr int NTAP = 27;
  constexpr bool PRECISE = sizeof(T) == 4;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Z = reinterpret_cast<float*>(smem);   // [27][ZP]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int ntd = (p.Do + TD - 1) / TD, nth = (p.Ho + TH - 1) / TH, ntw = (p.Wo + TW - 1) / TW;
  unsigned b = xcd_remap(blockIdx.x, gridDim.x);
  const int tw_i = b % ntw; b /= ntw;
  const int th_i = b % nth; b /= nth;
  const int td_i = b % ntd; b /= ntd;
  const int n = b;
  const int od0 = td_i * TD, oh0 = th_i * TH, ow0 = tw_i * TW;

  // ---- weights: A fragments [tap fragment 0/1][channel step], rows = taps (27 of 32) ------------------------------------------------
  const T* wsrc = reinterpret_cast<const T*>(p.w);  // packed [chunk][tap][cout_pad = 16][BK], output channel 0
  uint4 wf[2][KS];
#pragma unroll
  for (int tf = 0; tf < 2; ++tf)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int tap = tf * 16 + l15;
      const bool ok = tap < NTAP;
      const uint4 v = *reinterpret_cast<const uint4*>(wsrc + ((long long)(s * NTAP + (ok ? tap : 0)) * 16) * BK + q * VECW);
      wf[tf][s] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
    }
  // ---- fused GroupNorm affine (+ SiLU) of this lane's channels -----------------------------------------------------------------
  float sc[KS][VECW], sh[KS][VECW];
  if (p.pre_scale) {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int i = 0; i < VECW; ++i) {
        const int c = s * BK + q * VECW + i;
        sc[s][i] = p.pre_scale[(long long)n * p.Cin + c];
        sh[s][i] = p.pre_shift[(long long)n * p.Cin + c];
      }
  }

  // ---- phase 1: Z[tap][v] for every patch voxel; 16-voxel fragments, operands straight from global memory.  A wave owns fragments
  // wave, wave + 4, ...; their loads are issued in batches of FB fragments, batch b + 1 in flight while batch b is multiplied (one
  // fragment per round trip left 8 waves per CU waiting on HBM latency: 0.33 ms for the 64 -> 1 head at 128^3) -------------------
  const T* xin = reinterpret_cast<const T*>(p.x);
  constexpr int FB = KS <= 2 ? 4 : 2;                       // fragments per batch (register budget: 2 x FB x KS x 4 VGPRs)
  constexpr int FPW = (NFRAG + 3) / 4;                      // fragments per wave (11)
  constexpr int NBATCH = (FPW + FB - 1) / FB;
  auto frag_vox = [&](int f, bool& ok) __attribute__((always_inline)) -> long long {
    const int row = f * 16 + l15;
    const int pa = row / (PH * PW), rr = row - pa * (PH * PW), pb = rr / PW, pc = rr - pb * PW;
    const int ud = od0 - p.pd + pa, uh = oh0 - p.ph + pb, uw = ow0 - p.pw + pc;
    ok = (f < NFRAG) & (row < PROWS) & (ud >= 0) & (ud < p.Ds) & (uh >= 0) & (uh < p.Hs) & (uw >= 0) & (uw < p.Ws);
    return ok ? (((long long)n * p.Ds + ud) * p.Hs + uh) * p.Ws + uw : 0;
  };
  uint4 raw[2][FB][KS];
  auto load_batch = [&](int bi, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < FB; ++u) {
      bool ok;
      const long long vox = frag_vox(wave + 4 * (bi * FB + u), ok);
#pragma unroll
      for (int s = 0; s < KS; ++s) raw[slot][u][s] = *reinterpret_cast<const uint4*>(xin + vox * p.x_ld + s * BK + q * VECW);  // branch-free
    }
  };
  load_batch(0, 0);
#pragma unroll
  for (int bi = 0; bi < NBATCH; ++bi) {
    if (bi + 1 < NBATCH) load_batch(bi + 1, (bi + 1) & 1);
#pragma unroll
    for (int u = 0; u < FB; ++u) {
      const int f = wave + 4 * (bi * FB + u);
      if (f >= NFRAG) continue;  // wave-uniform
      bool ok;
      (void)frag_vox(f, ok);
      const int row = f * 16 + l15;
      f32x4_t z0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, z1 = z0;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        uint4 xf = raw[bi & 1][u][s];
        if (p.pre_scale || p.pre_act) {
          float v[VECW];
          Vec16<T>::unpack(xf, v);
          if (p.pre_scale) {
#pragma unroll
            for (int i = 0; i < VECW; ++i) v[i] = v[i] * sc[s][i] + sh[s][i];
          }
          if (p.pre_act) {
#pragma unroll
            for (int i = 0; i < VECW; ++i) v[i] = conv_act(v[i], p.pre_act, PRECISE);
          }
          xf = Vec16<T>::pack(v);
        }
        xf = make_uint4(ok ? xf.x : 0u, ok ? xf.y : 0u, ok ? xf.z : 0u, ok ? xf.w : 0u);  // zero padding of the ACTIVATED tensor
        Mma<T>::run(wf[0][s], xf, z0);
        Mma<T>::run(wf[1][s], xf, z1);
      }
      // D layout: column = voxel l15, rows = taps 4q + r (fragment 0) and 16 + 4q + r (fragment 1)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Z[(4 * q + r) * ZP + row] = z0[r];
        if (16 + 4 * q + r < NTAP) Z[(16 + 4 * q + r) * ZP + row] = z1[r];
      }
    }
  }
  __syncthreads();

  // ---- phase 2: 27-point gather, one output voxel per thread, fixed summation order -------------------------------------------
  const int m = tid;
  const int a = m >> 6, bb = (m >> 4) & 3, c = m & 15;
  const int od = od0 + a, oh = oh0 + bb, ow = ow0 + c;
  float sum = p.bias ? p.bias[0] : 0.f;
  if (p.rowvec) sum += p.rowvec[(long long)n * p.rowvec_bstride];
#pragma unroll
  for (int tap = 0; tap < NTAP; ++tap) {
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    sum += Z[tap * ZP + ((a + kd) * PH + bb + kh) * PW + c + kw];
  }
  if (od < p.Do && oh < p.Ho && ow < p.Wo) {
    const long long vox = (((long long)n * p.Do + od) * p.Ho + oh) * p.Wo + ow;
    if (p.res) sum += ElemIO<T>::ld(reinterpret_cast<const T*>(p.res) + vox * p.res_ld);
    ElemIO<T>::st(reinterpret_cast<T*>(p.y) + vox * p.y_ld, conv_post_act(sum, p.post_act));
  }
}

// -----------------------------------------------------------------------------------------------------------------------
static bool edge_common(const GmConvDesc* d) {
  return d->kd == 3 && d->kh == 3 && d->kw == 3 && d->sd == 1 && d->sh == 1 && d->sw == 1 && d->dd == 1 && d->dh == 1 && d->dw == 1 &&
         d->in_mode == 0 && d->ltd == 2 && d->lth == 2 && d->ltw == 4 && (long long)d->N * d->Ds * d->Hs * d->Ws < (1LL << 40);
}

extern "C" int gm_conv_cin_eligible(const GmConvDesc* d) {
  const int vecw = d->dtype == GM_F32 ? 4 : 8;
  return edge_common(d) && d->Cin >= 1 && d->Cin <= 4 && d->pre_scale == nullptr && d->pre_act == 0 && d->Cout % vecw == 0 &&
         d->y_ld % vecw == 0 && (reinterpret_cast<uintptr_t>(d->y) & 15) == 0 &&
         (!d->res || (d->res_ld % vecw == 0 && (reinterpret_cast<uintptr_t>(d->res) & 15) == 0));
}
extern "C" long long gm_conv_cin_lds_bytes(const GmConvDesc* d) {
  const long long es = d->dtype == GM_F32 ? 4 : 2;
  const long long operands = ((648 * 4 * es + 15) & ~15LL) + 64 * (128 * es + 16);
  const long long scratch = 4LL * 64 * 144;
  return operands > scratch ? operands : scratch;
}
extern "C" int gm_conv_cout1_eligible(const GmConvDesc* d) {
  const int bk = d->dtype == GM_F32 ? 16 : 32, vecw = d->dtype == GM_F32 ? 4 : 8;
  const int ks = d->Cin / bk;
  return edge_common(d) && d->Cout == 1 && d->Cin % bk == 0 && (ks == 2 || ks == 4 || (d->dtype == GM_F32 && ks == 8)) &&
         d->x_ld % vecw == 0 && (reinterpret_cast<uintptr_t>(d->x) & 15) == 0 && d->stats == nullptr;
}
extern "C" long long gm_conv_cout1_lds_bytes() { return 27LL * 660 * 4; }

template <typename KernT>
static void edge_launch(KernT kern, const GmConvDesc& d, unsigned nblocks, size_t smem, hipStream_t st) {
  static const void* seen[16];  // raise the dynamic-LDS limit once per kernel instantiation
  static int nseen = 0;
  const void* key = reinterpret_cast<const void*>(kern);
  bool found = false;
  for (int i = 0; i < nseen; ++i) found |= seen[i] == key;
  if (!found) {
    hipError_t e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) (void)hipGetLastError();
    if (nseen < 16) seen[nseen++] = key;
  }
  kern<<<dim3(nblocks), 256, smem, st>>>(d);
}

extern "C" int gm_conv_cin_launch(const GmConvDesc* dp, unsigned nblocks, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const size_t smem = (size_t)gm_conv_cin_lds_bytes(dp);
  if (dp->dtype == GM_F32) { edge_launch(conv_cin_kernel<float>, *dp, nblocks, smem, st); return 0; }
  if (dp->dtype == GM_BF16) { edge_launch(conv_cin_kernel<bf16_raw>, *dp, nblocks, smem, st); return 0; }
  return -2;
}

extern "C" int gm_conv_cout1_launch(const GmConvDesc* dp, unsigned nblocks, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const size_t smem = (size_t)gm_conv_cout1_lds_bytes();
  const int ks = dp->Cin / (dp->dtype == GM_F32 ? 16 : 32);
  if (dp->dtype == GM_F32) {
    if (ks == 2) edge_launch(conv_cout1_kernel<float, 2>, *dp, nblocks, smem, st);
    else if (ks == 4) edge_launch(conv_cout1_kernel<float, 4>, *dp, nblocks, smem, st);
    else edge_launch(conv_cout1_kernel<float, 8>, *dp, nblocks, smem, st);
    return 0;
  }
  if (dp->dtype == GM_BF16) {
    if (ks == 2) edge_launch(conv_cout1_kernel<bf16_raw, 2>, *dp, nblocks, smem, st);
    else edge_launch(conv_cout1_kernel<bf16_raw, 4>, *dp, nblocks, smem, st);
    return 0;
  }
  return -2;
}
