// 3x3x3 stride-1 convolution over a SMALL volume, K-COMPLETE on NARROW output-channel blocks (tile configuration 24, round 6): the ResnetBlock convolutions
// of the 32^3 / 16^3 / 8^3 levels of a latent UNet (reference: generative/networks/nets/diffusion_model_unet.py:589-696).
//
// Until round 5 these launches were split-K: too few 256-voxel x 64-channel tiles for 256 CUs, so the K chunks of a tile were dealt to `ksplit` work-groups
// (conv_sk.hip) that wrote fp32 partial sums, and a second launch (conv_splitk_combine_kernel) summed the slices and applied the epilogue: 17.3 + 7.7 us of
// kernel time per convolution, 34 convolutions per latent step = 0.85 of the step's 1.67 ms (rocprofv3, profiles/r06_c3_unet_kernel_stats_before.csv).
// The other way to make more work-groups out of a small volume is the OUTPUT-CHANNEL axis: a work-group owns a 256-voxel tile x 16 output channels and the
// WHOLE contraction.  What that buys: no partial sums in HBM, no second launch, the epilogue (bias + timestep row + residual + activation + GroupNorm
// statistics) in the kernel that produced the accumulators.  What it costs: the halo patch of a tile is staged once per 16-channel block instead of once per
// 64 (L2 hits: the blocks of a tile are neighbours in the XCD's work range) and one MFMA column per operand read.  Per K chunk a work-group stages 42 KiB of
// patch + 27 KiB of weights (ALL 27 taps of its 16 channels: one request burst, one wait, one barrier per chunk -- conv_sk.hip's latency recipe) = 73 KiB of
// LDS with the scale / shift copies: TWO work-groups per CU, one staging while the other multiplies.
// Same 4x4x16 tile, patch layout, source-side bank swizzle and packed weight image ([chunk][tap][Cout_pad][BK]) as conv_dma.hip / conv_sk.hip; also their
// fused GroupNorm-apply + activation prologue in LDS (bit-identical arithmetic to gm_gn_apply), the second input source of a virtual concatenation and the
// fused 1x1 shortcut.  bf16 and fp32.  Deterministic: fixed summation orders, statistics stored once per (tile, channel).
#include "conv_dma_shared.h"

__device__ __attribute__((aligned(64))) unsigned int gm_sn_zero_row[16] = {0};  // the source of every padding row

namespace sn {
constexpr int KS = 3, BM = 256, BN = 16, TW = 16, PW = 18;
constexpr int MAXW = 8;  // the LDS layout is sized for the 8-wave form
// ND = 3: 4 x 4 x 16 voxel tile, 6 planes of 6 x 18 patch rows (plane pitch padded to 112: depth offsets keep row mod 16), 27 taps.
// ND = 2 (tile configuration 25): the same kernel over IMAGES -- 16 x 16 pixel tile, one "plane" of 18 x 18 patch rows, 9 taps; the descriptor carries a
// 2-D convolution as depth 1 / kd 1 (ops.py), so every depth index below is 0.
template <int ND> struct Geom {
  static constexpr int NTAP = ND == 3 ? 27 : 9;
  static constexpr int TD = ND == 3 ? 4 : 1, TH = ND == 3 ? 4 : 16;
  static constexpr int PH = ND == 3 ? 6 : 18, PLANE = ND == 3 ? 112 : 336, PROWS = (ND == 3 ? 6 : 1) * PLANE, PPIECES = PROWS / 16;
  static constexpr int PATCH_BYTES = PROWS * DMA_ROWB, W_BYTES = NTAP * BN * DMA_ROWB;  // 43 008 + 27 648 (3-D), 21 504 + 9 216 (2-D)
  static constexpr int AFF_OFF = PATCH_BYTES + W_BYTES, AFF_WAVE = 256;  // per wave: [scale | shift] of the chunk being staged (PRE, scale / shift arrays given)
  static constexpr int MAX_CIN_TAB = 384;                                // ... or ONE table [scale[Cin] | shift[Cin]] built from the input's statistics (PRE, pre_stats),
  static constexpr int AFF_BYTES = 2 * MAX_CIN_TAB * 4 + MAX_CIN_TAB * 16;  //     behind it the per-channel fp64 (sum, sum of squares) the table is built from: 9 KiB
  static constexpr int STAT_OFF = AFF_OFF + AFF_BYTES;                   // [wave][16 channels][sum, sum of squares] fp32
  static constexpr int LDS_BYTES = STAT_OFF + MAXW * BN * 8;             // 80 896 (3-D): two work-groups per CU; 40 960 (2-D)
  static_assert(AFF_BYTES >= MAXW * AFF_WAVE, "the per-wave scale / shift copies fit too");
  static constexpr int SKIP_ROUND = 2 * BM * DMA_ROWB <= PATCH_BYTES ? 2 : 1;  // shortcut chunks staged per round (their voxel rows go into the patch buffer)
  static_assert(2 * LDS_BYTES <= 160 * 1024, "two work-groups per CU");
  static_assert(SKIP_ROUND * BM * DMA_ROWB <= PATCH_BYTES && SKIP_ROUND * BN * DMA_ROWB <= W_BYTES, "the shortcut's chunks of a round fit into the operand buffers");
  static_assert(PLANE % 16 == 0 && PH * PW <= PLANE, "whole DMA pieces per plane");
};
}  // namespace sn

__device__ __forceinline__ float sn_row16_sum(float v) {  // sum over the 16 lanes of a DPP row, every lane ends with it; fixed order
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xF, 0xF, true));  // row_ror:8
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x124, 0xF, 0xF, true));  // row_ror:4
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x122, 0xF, 0xF, true));  // row_ror:2
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x121, 0xF, 0xF, true));  // row_ror:1
  return v;
}

// NW waves x MF 16-voxel fragments cover the 256-voxel tile.  <8, 2> is what the library instantiates (a tap = 3 LDS reads for 2 MFMAs per wave); <4, 4> (5
// reads for 4 MFMAs: fewer LDS operand bytes) was measured and is SLOWER on the C3 latent UNet (1.562 vs 1.518 ms per forward, profiles/r06_c3_cfg24_policy_sweep.txt):
// sixteen waves per CU hide the request / barrier chain of a chunk better than eight.
template <typename T, bool PRE, int NW, int MF, int ND>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 2) void conv_sn_kernel(const GmConvDesc p) {
  using namespace sn;
  typedef Geom<ND> GE;
  constexpr int NTAP = GE::NTAP, TD = GE::TD, TH = GE::TH, PH = GE::PH, PLANE = GE::PLANE, PROWS = GE::PROWS, PPIECES = GE::PPIECES;
  constexpr int PATCH_BYTES = GE::PATCH_BYTES, AFF_OFF = GE::AFF_OFF, AFF_WAVE = GE::AFF_WAVE, STAT_OFF = GE::STAT_OFF, SKIP_ROUND = GE::SKIP_ROUND;
  static_assert(NW * MF * 16 == BM && NW <= MAXW && MF <= 4 && (ND == 2 || 4 % MF == 0), "waves x fragments cover the tile; a wave's fragments are lines of one plane");
  constexpr int PPW = (PPIECES + NW - 1) / NW, WPW = (NTAP + NW - 1) / NW;  // patch / weight pieces per wave and chunk (tap = wave + NW h: one 1 KiB piece per tap)
  // voxel m of the tile -> (plane, line, column): 3-D (m >> 6, (m >> 4) & 3, m & 15); 2-D (0, m >> 4, m & 15)
  auto m_plane = [](int m) { return ND == 3 ? m >> 6 : 0; };
  auto m_line = [](int m) { return ND == 3 ? (m >> 4) & 3 : m >> 4; };
  constexpr int BK = ConvTraits<T>::BK;
  constexpr int VECW = ConvTraits<T>::VECW;
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // [patch][27 taps x 16 weight rows][8 x (scale | shift)][8 x 16 statistic partials]
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;

  // ---- the work item: (tile, 16-channel block); XCD x owns a contiguous range, the blocks of a tile are neighbours in it (their patches meet in one L2) ----
  // images, stride 2 (the Downsample convolution of a 2-D UNet, diffusion_model_unet.py:510-518): out(i, j) = the stride-1 result at (2 i, 2 j) -- the tiles walk
  // the STRIDE-1 grid of (2 Ho - 1) x (2 Wo - 1) positions and the epilogue stores the even ones (4x the MFMAs of a launch whose cost is its latency chain; the
  // generic kernel it replaces took 13 us and left the statistics to a stand-alone pass)
  const bool s2 = ND == 2 && p.sh == 2;  // (work-group uniform; host: sh == sw)
  const int H1 = s2 ? 2 * p.Ho - 1 : p.Ho, W1 = s2 ? 2 * p.Wo - 1 : p.Wo;
  const int ntd = (p.Do + TD - 1) / TD, nth = (H1 + TH - 1) / TH, ntw = (W1 + TW - 1) / TW;
  const int ncb = (p.Cout + BN - 1) / BN;
  const int nchunks = p.Cin / BK, cout_pad = (p.Cout + 15) & ~15;
  const unsigned nwork = (unsigned)p.N * ntd * nth * ntw * ncb;
  const unsigned xcd = blockIdx.x & 7;
  const unsigned q8 = nwork >> 3, r8 = nwork & 7, cx = q8 + (xcd < r8 ? 1u : 0u), sx = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  if ((blockIdx.x >> 3) >= cx) return;  // (never: the grid is nwork work-groups)
  unsigned b = sx + (blockIdx.x >> 3);
  const int cb = __builtin_amdgcn_readfirstlane((int)(b % ncb)); b /= ncb;
  const int tw_i = __builtin_amdgcn_readfirstlane((int)(b % ntw)); b /= ntw;
  const int th_i = __builtin_amdgcn_readfirstlane((int)(b % nth)); b /= nth;
  const int td_i = __builtin_amdgcn_readfirstlane((int)(b % ntd)); b /= ntd;
  const int n = __builtin_amdgcn_readfirstlane((int)b);
  const int od0 = td_i * TD, oh0 = th_i * TH, ow0 = tw_i * TW;

  // ---- weights: tap t of this block = rows cb*16 .. +15 of the (chunk, tap) panel = one contiguous 1 KiB piece; LDS row t*16 + r, slot s <- source slot
  // s ^ swz(r) (rows 16 apart share the key: period 8) -----------------------------------------------------------------------------------------------
  const char* zero = reinterpret_cast<const char*>(gm_sn_zero_row);
  const char* wbase = reinterpret_cast<const char*>(p.w) + ((long long)cb * BN + (lane >> 2)) * DMA_ROWB + (((lane & 3) ^ dma_swz(lane >> 2)) << 4);
  auto issue_weights = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < WPW; ++h) {
      const int tap = wave + NW * h;  // wave-uniform
      if (tap < NTAP) dma16(wbase + (long long)(chunk * NTAP + tap) * cout_pad * DMA_ROWB, lds0 + PATCH_BYTES + (unsigned)tap * (BN * DMA_ROWB));
    }
  };

  issue_weights(0);  // (they need the channel block only: the patch placement arithmetic below runs under their flight)

  // ---- patch rows of this lane (conv_dma.hip's layout: 64-byte rows, swizzle keyed on the row's column within its W line, applied on the source side) ----
  int psw = 0, pvox[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int row = 16 * (wave + NW * j) + (lane >> 2);
    const int pa = row / PLANE, rr = row - pa * PLANE;
    const int pb = rr / PW, lc = rr - pb * PW;
    psw |= dma_swz(lc) << (2 * j);
    // in_mode 1: the input is the nearest-neighbour up-sampling by (fd, fh, fw) of the stored tensor, never materialised (reference Upsample,
    // diffusion_model_unet.py:572-585): bounds on the virtual grid, source voxel = virtual / factor
    int ud = od0 - p.pd + pa, uh = oh0 - p.ph + pb, uw = ow0 - p.pw + lc;
    const int Dv = p.in_mode == 1 ? p.Ds * p.fd : p.Ds, Hv = p.in_mode == 1 ? p.Hs * p.fh : p.Hs, Wv = p.in_mode == 1 ? p.Ws * p.fw : p.Ws;
    const bool ok = (row < PROWS) & (rr < PH * PW) & (ud >= 0) & (ud < Dv) & (uh >= 0) & (uh < Hv) & (uw >= 0) & (uw < Wv);
    if (p.in_mode == 1) { ud /= p.fd; uh /= p.fh; uw /= p.fw; }
    pvox[j] = ok ? ((n * p.Ds + ud) * p.Hs + uh) * p.Ws + uw : -1;
  }
  const char* xbase = reinterpret_cast<const char*>(p.x);
  const char* x2base = reinterpret_cast<const char*>(p.x2);
  const long long xrowb = p.x_ld * (long long)sizeof(T), x2rowb = p.x2_ld * (long long)sizeof(T);
  const int nchunks0 = p.x2 ? p.cin_split / BK : nchunks;
  const bool from_stats = PRE && p.pre_stats[0] != nullptr;  // (work-group uniform)
  auto issue_patch = [&](int chunk) __attribute__((always_inline)) {
    if (PRE && !from_stats) {
      const int nl = BK / 4;  // lanes 0 .. nl - 1 fetch the chunk's scale, nl .. 2 nl - 1 its shift (16 bytes each) into this wave's own copy
      if (lane < 2 * nl) {
        const float* src = (lane < nl ? p.pre_scale : p.pre_shift) + (long long)n * p.Cin + chunk * BK + 4 * (lane < nl ? lane : lane - nl);
        dma16(src, lds0 + AFF_OFF + (unsigned)wave * AFF_WAVE);
      }
    }
    const bool second = chunk >= nchunks0;  // wave-uniform: the chunk comes from x2 (virtual channel concatenation)
    const char* cbase = second ? x2base + (long long)(chunk - nchunks0) * (BK * (int)sizeof(T)) : xbase + (long long)chunk * (BK * (int)sizeof(T));
    const long long rowb = second ? x2rowb : xrowb;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      if (wave + NW * j < PPIECES) {  // wave-uniform
        const char* in_src = cbase + (long long)(pvox[j] & ~(pvox[j] >> 31)) * rowb + (((lane & 3) ^ ((psw >> (2 * j)) & 3)) << 4);
        const char* pad_src = zero + ((lane & 3) << 4);
        dma16(pvox[j] >= 0 ? in_src : pad_src, lds0 + (unsigned)(16 * (wave + NW * j)) * DMA_ROWB);
      }
    }
  };
  // GroupNorm-apply + activation IN LDS on this wave's own landed pieces (conv_dma.hip: transform_patch; same arithmetic and rounding as gm_gn_apply;
  // rows that came from the zero page stay zero: the reference pads the ACTIVATED tensor)
  auto transform_patch = [&](int chunk) __attribute__((always_inline)) {
    float sc[VECW], sh[VECW];
    // scale / shift of this lane's channel slot: the wave's own per-chunk copy, or the table over all input channels built from the statistics
    const float* aff = reinterpret_cast<const float*>(smem + AFF_OFF) + (from_stats ? chunk * BK : wave * (AFF_WAVE / 4));
    const int shoff = from_stats ? p.Cin : BK;
#pragma unroll
    for (int i = 0; i < VECW; i += 4) {
      const float4 a = *reinterpret_cast<const float4*>(aff + (lane & 3) * VECW + i), c = *reinterpret_cast<const float4*>(aff + shoff + (lane & 3) * VECW + i);
      sc[i] = a.x; sc[i + 1] = a.y; sc[i + 2] = a.z; sc[i + 3] = a.w;
      sh[i] = c.x; sh[i + 1] = c.y; sh[i + 2] = c.z; sh[i + 3] = c.w;
    }
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      if (wave + NW * j < PPIECES && pvox[j] >= 0) {
        char* a = smem + (16 * (wave + NW * j) + (lane >> 2)) * DMA_ROWB + (((lane & 3) ^ ((psw >> (2 * j)) & 3)) << 4);
        float v[VECW];
        Vec16<T>::unpack(*reinterpret_cast<const uint4*>(a), v);
#pragma unroll
        for (int i = 0; i < VECW; ++i) v[i] = v[i] * sc[i] + sh[i];
        conv_act_vec(v, p.pre_act, sizeof(T) == 4);
        *reinterpret_cast<uint4*>(a) = Vec16<T>::pack(v);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  issue_patch(0);

  // ---- PRE from statistics: scale[c] = rstd_g * gamma[c], shift[c] = beta[c] - mean_g * rstd_g * gamma[c] for every input channel, from the per-tile partials of
  // the input's producer(s), under the flight of the first requests.  The SHORT-TABLE order of gm_gn_finalize_channels (groupnorm.hip: gn_short_* -- shared
  // helpers, so the two are bit-identical): a thread owns a channel and adds its S <= GN_SHORT_MAX_ROWS rows in row order (fp64, eight loads in flight); a group is the sum of its
  // channels in channel order.  A group may straddle the two sources of a virtual concatenation.
  if (PRE && from_stats) {
    float* tab = reinterpret_cast<float*>(smem + AFF_OFF);                                   // [scale[Cin] | shift[Cin]]
    double* csum = reinterpret_cast<double*>(smem + AFF_OFF + 2 * GE::MAX_CIN_TAB * 4);      // [Cin][sum, sum of squares]
    const int G = p.pre_groups, cpg = p.Cin / G;
    const long long V = (long long)p.Ds * p.Hs * p.Ws;  // voxels of the NORMALISED tensor
    for (int c = tid; c < p.Cin; c += 64 * NW) {
      const double2 v = gn_short_channel_sum(p.pre_stats[0], p.pre_S[0], p.pre_C[0], p.pre_stats[1], p.pre_S[1], p.pre_C[1], p.N, n, c);
      csum[2 * c] = v.x; csum[2 * c + 1] = v.y;
    }
    __syncthreads();
    for (int c = tid; c < p.Cin; c += 64 * NW) {
      const int g = c / cpg;
      double a = 0.0, b2 = 0.0;
      for (int j = 0; j < cpg; ++j) { a += csum[2 * (g * cpg + j)]; b2 += csum[2 * (g * cpg + j) + 1]; }
      float sc1, sh1;
      gn_short_scale_shift(a, b2, cpg, V, p.pre_eps, p.pre_gamma ? p.pre_gamma[c] : 1.f, p.pre_beta ? p.pre_beta[c] : 0.f, sc1, sh1);
      tab[c] = sc1;
      tab[p.Cin + c] = sh1;
    }
    __syncthreads();  // (transform_patch of chunk 0 runs in front of the first tap-loop barrier: the table needs its own)
  }

  // ---- the epilogue's per-channel addend (bias + shortcut bias + timestep row, this order, fp32) and this lane's output rows: under the first flight ----
  const int co4 = cb * BN + q * 4;  // this lane's four output channels (accumulator rows 4q .. 4q + 3 of the 16x16 MFMA)
  // vec4: the lane's four channels as ONE vector access (C_out, the row pitches and the base addresses allow it: host-checked per operand below); else element-wise with a
  // per-channel bound -- output heads of 1-3 channels, ragged channel counts
  const int elt = (int)sizeof(T);
  const bool vec4 = (p.Cout & 3) == 0 && (p.y_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(p.y) & (4 * elt - 1)) == 0 &&
                    (!p.res || ((p.res_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(p.res) & (4 * elt - 1)) == 0)) &&
                    (!p.bias || (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0) && (!p.skip_bias || (reinterpret_cast<uintptr_t>(p.skip_bias) & 15) == 0) &&
                    (!p.rowvec || ((reinterpret_cast<uintptr_t>(p.rowvec) & 15) == 0 && (p.rowvec_bstride & 3) == 0));
  float add[4] = {0.f, 0.f, 0.f, 0.f};
  if (vec4) {
    if (co4 < p.Cout) {
      if (p.bias) { const float4 v = *reinterpret_cast<const float4*>(p.bias + co4); add[0] += v.x; add[1] += v.y; add[2] += v.z; add[3] += v.w; }
      if (p.skip_bias) { const float4 v = *reinterpret_cast<const float4*>(p.skip_bias + co4); add[0] += v.x; add[1] += v.y; add[2] += v.z; add[3] += v.w; }
      if (p.rowvec) { const float4 v = *reinterpret_cast<const float4*>(p.rowvec + (long long)n * p.rowvec_bstride + co4); add[0] += v.x; add[1] += v.y; add[2] += v.z; add[3] += v.w; }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (co4 + r < p.Cout) {
        if (p.bias) add[r] += p.bias[co4 + r];
        if (p.skip_bias) add[r] += p.skip_bias[co4 + r];
        if (p.rowvec) add[r] += p.rowvec[(long long)n * p.rowvec_bstride + co4 + r];
      }
    }
  }

  // ---- operand read addresses (conv_dma.hip: XADDR / WADDR) ---------------------------------------------------------------------------------------
  int xa[KS];
  {
    const int m0 = wave * MF * 16 + l15;
    const int a = m_plane(m0), bb0 = m_line(m0), c = m0 & 15;
#pragma unroll
    for (int kw = 0; kw < KS; ++kw) xa[kw] = (a * PLANE + bb0 * PW + c + kw) * DMA_ROWB + ((q ^ dma_swz(c + kw)) << 4);
  }
  const int wa0 = PATCH_BYTES + l15 * DMA_ROWB + ((q ^ dma_swz(l15)) << 4);
  f32x4_t acc[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) acc[mf] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  uint4 xf[2][MF], wf[2];
  auto read_tap = [&](int tap, int set) __attribute__((always_inline)) {
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;  // (2-D: nine taps, kd = 0)
    wf[set] = *reinterpret_cast<const uint4*>(smem + wa0 + tap * (BN * DMA_ROWB));
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) xf[set][mf] = *reinterpret_cast<const uint4*>(smem + xa[kw] + (mf + kh) * (PW * DMA_ROWB) + kd * (PLANE * DMA_ROWB));
  };

  for (int chunk = 0; chunk < nchunks; ++chunk) {
    dma_wait<0>();                    // this wave's pieces of the chunk (and, PRE, its scale / shift copy) have landed
    if (PRE) transform_patch(chunk);
    __builtin_amdgcn_s_barrier();     // ... everyone's
    read_tap(0, 0);
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      // two operand sets: the next tap's three reads are issued, THEN this tap's MFMAs (sched_barrier: hipcc would merge the sets otherwise)
      if (tap + 1 < NTAP) read_tap(tap + 1, (tap + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) Mma<T>::run(wf[tap & 1], xf[tap & 1][mf], acc[mf]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (chunk + 1 < nchunks) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // every wave is done with this chunk's patch and weights
      issue_weights(chunk + 1);
      issue_patch(chunk + 1);
    }
  }

  // ---- fused 1x1 shortcut convolution (conv_dma.hip: extra K chunks over the skip sources, centre tap only): two chunks per round -- every wave stages the
  // 64-byte chunk of ITS OWN 32 output voxels into the patch buffer, waves 0 / 1 the 16-row weight piece of chunk 0 / 1 ------------------------------------
  if (p.skip_x[0]) {
    const int nsc0 = p.skip_cin[0] / BK, nsc = nsc0 + (p.skip_x[1] ? p.skip_cin[1] / BK : 0);
    const int pswz = ((lane & 3) ^ dma_swz(lane >> 2)) << 4;
    int svox[MF];
#pragma unroll
    for (int h = 0; h < MF; ++h) {
      const int m = wave * (MF * 16) + h * 16 + (lane >> 2);
      const int od = od0 + m_plane(m), oh = oh0 + m_line(m), ow = ow0 + (m & 15);
      svox[h] = (od < p.Do && oh < p.Ho && ow < p.Wo) ? ((n * p.Do + od) * p.Ho + oh) * p.Wo + ow : -1;
    }
    int caddr[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int m = (wave * MF + mf) * 16 + l15;
      caddr[mf] = m * DMA_ROWB + ((q ^ dma_swz(m)) << 4);
    }
    const char* wsk = reinterpret_cast<const char*>(p.skip_w) + ((long long)cb * BN + (lane >> 2)) * DMA_ROWB + pswz;
    for (int sc0 = 0; sc0 < nsc; sc0 += SKIP_ROUND) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // the operand buffers are free
#pragma unroll
      for (int j = 0; j < SKIP_ROUND; ++j) {
        const int sc = sc0 + j;
        if (sc < nsc) {  // wave-uniform
          const int part = sc >= nsc0 ? 1 : 0, cip = sc - (part ? nsc0 : 0);
          const char* xb = reinterpret_cast<const char*>(p.skip_x[part]) + (long long)cip * (BK * (int)sizeof(T)) + pswz;
          const long long rowb = p.skip_ld[part] * (long long)sizeof(T);
#pragma unroll
          for (int h = 0; h < MF; ++h) {
            const char* src = svox[h] >= 0 ? xb + svox[h] * rowb : zero + ((lane & 3) << 4);
            dma16(src, lds0 + (unsigned)(j * BM + wave * (MF * 16) + h * 16) * DMA_ROWB);
          }
          if (wave == j) dma16(wsk + (long long)sc * cout_pad * DMA_ROWB, lds0 + PATCH_BYTES + (unsigned)(j * BN) * DMA_ROWB);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int j = 0; j < SKIP_ROUND; ++j) {
        if (sc0 + j < nsc) {
          const uint4 ws = *reinterpret_cast<const uint4*>(smem + wa0 + j * (BN * DMA_ROWB));
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) {
            const uint4 xs = *reinterpret_cast<const uint4*>(smem + caddr[mf] + j * (BM * DMA_ROWB));
            Mma<T>::run(ws, xs, acc[mf]);
          }
        }
      }
    }
  }

  // ---- epilogue from the accumulators: lane (l15, q) holds channels co4 .. co4 + 3 of voxels (wave * 2 + mf) * 16 + l15 -- y = act(acc + addend + residual),
  // rounded once; statistics of the values as stored ---------------------------------------------------------------------------------------------------
  float ss[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
  T* yout = reinterpret_cast<T*>(p.y);
  const T* res = reinterpret_cast<const T*>(p.res);
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int m = (wave * MF + mf) * 16 + l15;
    const int od = od0 + m_plane(m);
    int oh = oh0 + m_line(m), ow = ow0 + (m & 15);
    bool keep = true;
    if (s2) { keep = ((oh | ow) & 1) == 0; oh >>= 1; ow >>= 1; }  // the even positions of the stride-1 grid are the strided outputs
    if (keep && od < p.Do && oh < p.Ho && ow < p.Wo && co4 < p.Cout) {
      const long long vox = (((long long)n * p.Do + od) * p.Ho + oh) * p.Wo + ow;
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = acc[mf][r] + add[r];
      T* yp = yout + vox * p.y_ld + co4;
      if (vec4) {
        if (res) {
          const T* rp = res + vox * p.res_ld + co4;
          if (sizeof(T) == 4) {
            const float4 rv = *reinterpret_cast<const float4*>(rp);
            o[0] += rv.x; o[1] += rv.y; o[2] += rv.z; o[3] += rv.w;
          } else {
            const uint2 rv = *reinterpret_cast<const uint2*>(rp);
            o[0] += __uint_as_float(rv.x << 16); o[1] += __uint_as_float(rv.x & 0xffff0000u);
            o[2] += __uint_as_float(rv.y << 16); o[3] += __uint_as_float(rv.y & 0xffff0000u);
          }
        }
        if (p.post_act) {
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = conv_post_act(o[r], p.post_act);
        }
        if (sizeof(T) == 4) {
          *reinterpret_cast<float4*>(yp) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
          const uint2 raw = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
          *reinterpret_cast<uint2*>(yp) = raw;
          o[0] = __uint_as_float(raw.x << 16); o[1] = __uint_as_float(raw.x & 0xffff0000u);
          o[2] = __uint_as_float(raw.y << 16); o[3] = __uint_as_float(raw.y & 0xffff0000u);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (co4 + r < p.Cout) {
            if (res) o[r] += ElemIO<T>::ld(res + vox * p.res_ld + co4 + r);
            if (p.post_act) o[r] = conv_post_act(o[r], p.post_act);
            ElemIO<T>::st(yp + r, o[r]);
            if (sizeof(T) == 2) o[r] = __uint_as_float(pack_bf16x2(o[r], 0.f) << 16);  // (the value as stored: rounded to T)
          } else {
            o[r] = 0.f;
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { ss[r] += o[r]; sq[r] += o[r] * o[r]; }
    }
  }
  if (p.stats) {
    // lane sums over its two voxels -> sum over the 16 lanes of its DPP row (the 16 voxels of a W line) -> one partial per (wave, channel) in LDS ->
    // fixed-order fp64 sum over the 8 waves: one plain store per (tile, channel), no atomics
    float* part = reinterpret_cast<float*>(smem + STAT_OFF) + wave * (BN * 2);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a = sn_row16_sum(ss[r]), b2 = sn_row16_sum(sq[r]);
      if (l15 == 0) { part[(q * 4 + r) * 2] = a; part[(q * 4 + r) * 2 + 1] = b2; }
    }
    __syncthreads();
    if (tid < BN) {
      double a = 0.0, b2 = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float2 v = *reinterpret_cast<const float2*>(smem + STAT_OFF + (w * BN + tid) * 8);
        a += (double)v.x;
        b2 += (double)v.y;
      }
      const int co = cb * BN + tid;
      if (co < p.Cout) {
        const long long slot = ((long long)td_i * nth + th_i) * ntw + tw_i;  // the tile within its sample
        *reinterpret_cast<double2*>(p.stats + ((slot * p.N + n) * p.Cout + co) * 2) = make_double2(a, b2);
      }
    }
  }
}

// geometry this kernel takes: cfg 24 = 3x3x3 over volumes (tile 4 x 4 x 16), cfg 25 = 3x3 over images carried as depth-1 volumes (tile 1 x 16 x 16); stride 1,
// direct input, the vector-aligned operands of the other LDS-DMA configurations
extern "C" int gm_conv_sn_eligible(const GmConvDesc* d) {
  const int bk = d->dtype == GM_F32 ? 16 : 32;
  const int vecw = d->dtype == GM_F32 ? 4 : 8;
  const bool geom3 = d->cfg == 24 && d->kd == 3 && d->ltd == 2 && d->lth == 2 && d->ltw == 4;
  const bool geom2 = d->cfg == 25 && d->kd == 1 && d->Ds == 1 && d->Do == 1 && d->pd == 0 && d->ltd == 0 && d->lth == 4 && d->ltw == 4 &&
                     (d->in_mode == 0 || d->fd == 1);
  const bool stride1 = d->sd == 1 && d->sh == 1 && d->sw == 1;
  const bool stride2_image = geom2 && d->sd == 1 && d->sh == 2 && d->sw == 2 && d->in_mode == 0 && !d->skip_x[0];  // (stride-1 result, even positions stored)
  return (geom3 || geom2) && (d->dtype == GM_F32 || d->dtype == GM_BF16) && d->kh == 3 && d->kw == 3 && (stride1 || stride2_image) &&
         d->dd == 1 && d->dh == 1 && d->dw == 1 && (d->in_mode == 0 || (d->in_mode == 1 && d->fd >= 1 && d->fh >= 1 && d->fw >= 1)) && d->Cin % bk == 0 &&
         d->x_ld % vecw == 0 &&
         (reinterpret_cast<uintptr_t>(d->x) & 15) == 0 && !(d->ksplit > 1 && d->kpartial) &&
         ((d->pre_scale == nullptr && d->pre_shift == nullptr && d->pre_stats[0] == nullptr && d->pre_act == 0) ||
          (d->pre_stats[0] == nullptr && d->pre_scale != nullptr && d->pre_shift != nullptr && (reinterpret_cast<uintptr_t>(d->pre_scale) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(d->pre_shift) & 15) == 0 && (d->Cin % 4) == 0) ||
          // the statistics form: short tables (their rows all sit in one wave of the fold), whole groups, the table over all input channels in LDS
          (d->pre_stats[0] != nullptr && d->pre_scale == nullptr && d->pre_shift == nullptr && d->pre_groups > 0 && d->Cin % d->pre_groups == 0 &&
           d->Cin <= sn::Geom<3>::MAX_CIN_TAB && d->pre_S[0] >= 1 && d->pre_S[0] <= GN_SHORT_MAX_ROWS && d->pre_C[0] > 0 &&
           ((d->pre_stats[1] == nullptr && d->pre_C[1] == 0 && d->pre_C[0] == d->Cin) ||
            (d->pre_stats[1] != nullptr && d->pre_S[1] >= 1 && d->pre_S[1] <= GN_SHORT_MAX_ROWS && d->pre_C[1] > 0 && d->pre_C[0] + d->pre_C[1] == d->Cin)) &&
           (reinterpret_cast<uintptr_t>(d->pre_stats[0]) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->pre_stats[1]) & 15) == 0 && d->in_mode == 0)) &&
         (d->x2 == nullptr || (d->cin_split > 0 && d->cin_split < d->Cin && d->cin_split % bk == 0 && d->x2_ld % vecw == 0 &&
                               (reinterpret_cast<uintptr_t>(d->x2) & 15) == 0)) &&
         // (any C_out / row pitch / alignment of y, res, bias, rowvec: four channels per lane go out as one vector where all of them allow it, element-wise otherwise)
         (long long)d->N * d->Ds * d->Hs * d->Ws < (1LL << 31) && (long long)d->N * d->Do * d->Ho * d->Wo < (1LL << 31) &&
         (!d->skip_x[0] ||
          (d->skip_w && d->skip_cin[0] > 0 && d->skip_cin[0] % bk == 0 && d->skip_ld[0] % vecw == 0 && (reinterpret_cast<uintptr_t>(d->skip_x[0]) & 15) == 0 &&
           (!d->skip_x[1] || (d->skip_cin[1] > 0 && d->skip_cin[1] % bk == 0 && d->skip_ld[1] % vecw == 0 &&
                              (reinterpret_cast<uintptr_t>(d->skip_x[1]) & 15) == 0))));
}

extern "C" long long gm_conv_sn_lds_bytes(int cfg) { return cfg == 25 ? sn::Geom<2>::LDS_BYTES : sn::Geom<3>::LDS_BYTES; }

template <typename T, bool PRE, int NW, int MF, int ND>
static void launch_sn(const GmConvDesc& d, unsigned nblocks, hipStream_t st) {
  static bool attr_set = false;
  auto kern = conv_sn_kernel<T, PRE, NW, MF, ND>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) (void)hipGetLastError();
    attr_set = true;
  }
  kern<<<dim3(nblocks), 64 * NW, (size_t)sn::Geom<ND>::LDS_BYTES, st>>>(d);
}

template <typename T>
static void launch_sn_dt(const GmConvDesc& d, unsigned nblocks, hipStream_t st) {
  const bool pre = d.pre_scale != nullptr || d.pre_stats[0] != nullptr;
  if (d.cfg == 25) { if (pre) launch_sn<T, true, 8, 2, 2>(d, nblocks, st); else launch_sn<T, false, 8, 2, 2>(d, nblocks, st); }
  else { if (pre) launch_sn<T, true, 8, 2, 3>(d, nblocks, st); else launch_sn<T, false, 8, 2, 3>(d, nblocks, st); }
}

extern "C" int gm_conv_sn_launch(const GmConvDesc* dp, unsigned nblocks, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (dp->dtype == GM_F32) { launch_sn_dt<float>(*dp, nblocks, st); return 0; }
  if (dp->dtype == GM_BF16) { launch_sn_dt<bf16_raw>(*dp, nblocks, st); return 0; }
  return -2;
}
