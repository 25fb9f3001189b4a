// The stride-1 hot path of the implicit-GEMM convolution (3^d / 1^d kernels, optional fused nearest-2x up-sampling): the
// ResnetBlock / Upsample / attention-projection convolutions that carry >90 % of a DiffusionModelUNet forward.
//
// Same data flow as conv.hip (halo patch of the output tile staged once per 64-byte input-channel chunk with the fused
// GroupNorm+SiLU prologue; weights streamed through LDS; A = weights, B = activations, 16x16 MFMA), restructured around what
// rocprof showed on MI355X -- the generic kernel spent its time waiting for one L2 round trip per tap:
//   * taps are processed in groups of 3 per barrier (48 MFMAs per wave between barriers instead of 16) and the next group's
//     weight panel is prefetched global->registers at the top of the group, written to the alternate LDS buffer after the MFMAs;
//   * the patch addresses (division-heavy, chunk-independent) are computed once per work-group and kept in registers; the
//     patch loads of a chunk are issued in batches of 4 before any of them is consumed;
//   * work-group ids are remapped so each XCD (private L2) owns a contiguous range of tiles;
//   * tile = 256 voxels (4x8x8 in 3-D) x 64 channels with 4 waves (2 work-groups / CU), or x 128 channels with 8 waves.
#include "conv_epilogue.h"

template <typename T, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_fast_kernel(const GmConvDesc p) {
  constexpr int BK = ConvTraits<T>::BK;
  constexpr int VECW = ConvTraits<T>::VECW;
  constexpr int NT = 64 * WM * WN;
  constexpr int MF = 4, NFR = 4, G = 3;
  constexpr int BN = WN * 64;
  constexpr int ROWS_PER_PASS = NT / 4;
  constexpr int MAX_ITEMS = NT == 256 ? 12 : 8;  // patch rows per thread (host guarantees P <= MAX_ITEMS * ROWS_PER_PASS)
  constexpr bool PRECISE = sizeof(T) == 4;
  static_assert(BN * 4 == NT, "one 16-byte weight item per thread per tap");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int wm = wave % WM, wn = wave / WM;

  // ---- tile geometry (stride 1, dilation 1) ----------------------------------------------------------------------------
  const int td = 1 << p.ltd, th = 1 << p.lth, tw = 1 << p.ltw;
  const int ntd = (p.Do + td - 1) >> p.ltd, nth = (p.Ho + th - 1) >> p.lth, ntw = (p.Wo + tw - 1) >> p.ltw;
  const int ncb = (p.Cout + BN - 1) / BN;
  unsigned b = xcd_remap(blockIdx.x, gridDim.x);
  const int cb = b % ncb; b /= ncb;
  const int tw_i = b % ntw; b /= ntw;
  const int th_i = b % nth; b /= nth;
  const int td_i = b % ntd; b /= ntd;
  const int n = b;
  const int od0 = td_i << p.ltd, oh0 = th_i << p.lth, ow0 = tw_i << p.ltw;
  const int pD = td + p.kd - 1, pH = th + p.kh - 1, pW = tw + p.kw - 1;
  const int P = pD * pH * pW;
  int Dv = p.Ds, Hv = p.Hs, Wv = p.Ws;
  if (p.in_mode == 1) { Dv *= p.fd; Hv *= p.fh; Wv *= p.fw; }
  const int ud0 = od0 - p.pd, uh0 = oh0 - p.ph, uw0 = ow0 - p.pw;

  char* ldsA = smem;                              // [P][CONV_ROWB]
  char* ldsB = smem + (size_t)P * CONV_ROWB;      // [2][G][BN][CONV_ROWB]

  const int T_taps = p.kd * p.kh * p.kw;
  const int ngroups = (T_taps + G - 1) / G;
  const int nchunks = (p.Cin + BK - 1) / BK;
  const int cout_pad = (p.Cout + 15) & ~15;
  const int total_gsteps = nchunks * ngroups;

  // ---- per-thread patch rows: source voxel index (chunk independent), -1 = zero padding, -2 = no such row ---------------
  const int sq = tid & 3;
  int vox[MAX_ITEMS];
#pragma unroll
  for (int j = 0; j < MAX_ITEMS; ++j) {
    const int pv = (tid >> 2) + j * ROWS_PER_PASS;
    int v = -2;
    if (pv < P) {
      const int pc = pv % pW;
      const int t1 = pv / pW;
      const int pb = t1 % pH, pa = t1 / pH;
      int ud = ud0 + pa, uh = uh0 + pb, uw = uw0 + pc;
      const bool ok = (ud >= 0) & (ud < Dv) & (uh >= 0) & (uh < Hv) & (uw >= 0) & (uw < Wv);
      if (p.in_mode == 1) { ud /= p.fd; uh /= p.fh; uw /= p.fw; }
      v = ok ? ((n * p.Ds + ud) * p.Hs + uh) * p.Ws + uw : -1;
    }
    vox[j] = v;
  }

  // ---- per-lane LDS read offsets ---------------------------------------------------------------------------------------
  int aoff[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int m = (wm * MF + mf) * 16 + l15;
    const int a = m >> (p.lth + p.ltw), bb = (m >> p.ltw) & (th - 1), c = m & (tw - 1);
    aoff[mf] = ((a * pH + bb) * pW + c) * CONV_ROWB + q * 16;
  }
  int boff[NFR];
#pragma unroll
  for (int nf = 0; nf < NFR; ++nf) boff[nf] = ((wn * NFR + nf) * 16 + l15) * CONV_ROWB + q * 16;

  f32x4_t acc[NFR][MF];
#pragma unroll
  for (int nf = 0; nf < NFR; ++nf)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- weight panels: group of G taps, one 16-byte item per thread per tap ----------------------------------------------
  const char* wbase = reinterpret_cast<const char*>(p.w);
  const int brow = tid >> 2;
  const int bco = cb * BN + brow;
  uint4 breg[G];
  auto load_b = [&](int gstep) {
    const int chunk = gstep / ngroups, grp = gstep - chunk * ngroups;
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int t = grp * G + u;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (t < T_taps && bco < cout_pad)
        v = *reinterpret_cast<const uint4*>(wbase + (((long long)(chunk * T_taps + t) * cout_pad + bco) * BK) * (long long)sizeof(T) + sq * 16);
      breg[u] = v;
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int u = 0; u < G; ++u)
      *reinterpret_cast<uint4*>(ldsB + ((size_t)(buf * G + u) * BN + brow) * CONV_ROWB + sq * 16) = breg[u];
  };

  // ---- patch staging with the fused prologue ---------------------------------------------------------------------------
  const T* xin = reinterpret_cast<const T*>(p.x);
  auto stage_a = [&](int chunk) {
    const int c0 = chunk * BK + sq * VECW;
    const bool cok = c0 < p.Cin;  // Cin % VECW == 0 (host-checked): a vector is entirely in or out of range
    float sc[VECW], sh[VECW];
    if (p.pre_scale) {
#pragma unroll
      for (int i = 0; i < VECW; ++i) {
        sc[i] = cok ? p.pre_scale[(long long)n * p.Cin + c0 + i] : 0.f;
        sh[i] = cok ? p.pre_shift[(long long)n * p.Cin + c0 + i] : 0.f;
      }
    }
#pragma unroll
    for (int jb = 0; jb < MAX_ITEMS; jb += 4) {
      uint4 raw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        raw[u] = make_uint4(0, 0, 0, 0);
        if (vox[jb + u] >= 0 && cok) raw[u] = *reinterpret_cast<const uint4*>(xin + (long long)vox[jb + u] * p.x_ld + c0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = jb + u;
        if (vox[j] == -2) continue;
        uint4 outv = raw[u];
        if (vox[j] >= 0 && cok && (p.pre_scale || p.pre_act)) {
          float v[VECW];
          Vec16<T>::unpack(raw[u], v);
          if (p.pre_scale) {
#pragma unroll
            for (int i = 0; i < VECW; ++i) v[i] = v[i] * sc[i] + sh[i];
          }
          if (p.pre_act) {
#pragma unroll
            for (int i = 0; i < VECW; ++i) v[i] = conv_act(v[i], p.pre_act, PRECISE);
          }
          outv = Vec16<T>::pack(v);
        }
        const int pv = (tid >> 2) + j * ROWS_PER_PASS;
        *reinterpret_cast<uint4*>(ldsA + (size_t)pv * CONV_ROWB + sq * 16) = outv;
      }
    }
  };

  // ---- main loop --------------------------------------------------------------------------------------------------------
  load_b(0);
  stage_a(0);
  store_b(0);
  __syncthreads();
  int gstep = 0;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    if (chunk > 0) {
      stage_a(chunk);
      __syncthreads();
    }
    int kd_i = 0, kh_i = 0, kw_i = 0;
    for (int grp = 0; grp < ngroups; ++grp, ++gstep) {
      const bool more = gstep + 1 < total_gsteps;
      if (more) load_b(gstep + 1);
      const char* bsrc = ldsB + (size_t)((gstep & 1) * G) * BN * CONV_ROWB;
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (grp * G + u < T_taps) {
          const int tap_off = ((kd_i * pH + kh_i) * pW + kw_i) * CONV_ROWB;
          uint4 xf[MF], wf[NFR];
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) xf[mf] = *reinterpret_cast<const uint4*>(ldsA + aoff[mf] + tap_off);
#pragma unroll
          for (int nf = 0; nf < NFR; ++nf) wf[nf] = *reinterpret_cast<const uint4*>(bsrc + (size_t)u * BN * CONV_ROWB + boff[nf]);
#pragma unroll
          for (int nf = 0; nf < NFR; ++nf)
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) Mma<T>::run(wf[nf], xf[mf], acc[nf][mf]);
          if (++kw_i == p.kw) { kw_i = 0; if (++kh_i == p.kh) { kh_i = 0; ++kd_i; } }
        }
      }
      if (more) store_b((gstep + 1) & 1);
      __syncthreads();
    }
  }

  conv_epilogue<T, MF, NFR>(p, acc, n, wm * MF * 16, cb * BN + wn * NFR * 16, od0, oh0, ow0, l15, q);
}

extern "C" long long gm_conv_fast_lds_bytes(const GmConvDesc* d, int bn) {
  const long long td = 1 << d->ltd, th = 1 << d->lth, tw = 1 << d->ltw;
  const long long P = (td + d->kd - 1) * (th + d->kh - 1) * (tw + d->kw - 1);
  return P * CONV_ROWB + 2LL * 3 * bn * CONV_ROWB;
}

extern "C" long long gm_conv_fast_max_patch(int wn) { return wn == 1 ? 12 * 64 : 8 * 128; }

template <typename T, int WM, int WN>
static void launch_fast(const GmConvDesc& d, size_t smem, unsigned nblocks, hipStream_t st) {
  static bool attr_set = false;
  auto kern = conv_fast_kernel<T, WM, WN>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) (void)hipGetLastError();
    attr_set = true;
  }
  kern<<<dim3(nblocks), 64 * WM * WN, smem, st>>>(d);
}

// wn = 1: 256 voxels x 64 channels (4 waves); wn = 2: 256 voxels x 128 channels (8 waves).  Returns 0 or a negative code.
extern "C" int gm_conv_fast_launch(const GmConvDesc* dp, int wn, unsigned nblocks, void* stream) {
  const GmConvDesc& d = *dp;
  const size_t smem = (size_t)gm_conv_fast_lds_bytes(dp, wn * 64);
  hipStream_t st = (hipStream_t)stream;
  if (d.dtype == GM_F32) {
    if (wn == 1) launch_fast<float, 4, 1>(d, smem, nblocks, st); else launch_fast<float, 4, 2>(d, smem, nblocks, st);
  } else if (d.dtype == GM_BF16) {
    if (wn == 1) launch_fast<bf16_raw, 4, 1>(d, smem, nblocks, st); else launch_fast<bf16_raw, 4, 2>(d, smem, nblocks, st);
  } else {
    return -2;
  }
  return 0;
}
