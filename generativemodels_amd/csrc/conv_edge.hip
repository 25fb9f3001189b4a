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
#include "conv_dma_shared.h"  // the LDS-DMA kernels' epilogue: addend vector in LDS, residual rows requested ahead, wave-private transpose, lane-swap statistics

// -----------------------------------------------------------------------------------------------------------------------
// C_in <= 4
// -----------------------------------------------------------------------------------------------------------------------
// (round 3: C_in is a template constant -- with a run-time C_in the staging loops and the tap tables spent ~2 000 instructions per thread on
//  integer divisions (element -> row -> (plane, line, column), k -> (tap, channel)), a large part of the 28 k cycles a work-group lives)
// (round 5: a minimum-occupancy bound.  Without one hipcc kept the 64 accumulators in AGPRs next to 110 VGPRs = 174 registers = TWO waves per SIMD for
//  a kernel that is a chain of latencies -- patch load, barrier, gather, MFMA, transpose, store -- per tile; bounded, the same code needs 110-154
//  registers without a spill: four (three) work-groups per CU.  fp32 with 3-4 input channels would spill: left at two.)
template <typename T, int CIN>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? (CIN <= 3 ? 4 : 3) : (CIN == 1 ? 4 : (CIN == 2 ? 3 : 2))) void conv_cin_kernel(const GmConvDesc p) {
  constexpr int VECW = ConvTraits<T>::VECW;
  constexpr int KB = 4 * VECW;                 // K values one Mma<T>::run consumes (32 bf16 / 16 fp32)
  constexpr int MF = 4, NFR = 4, NW = 4, BN = 64;
  constexpr int TD = 4, TH = 4, TW = 16, PH = TH + 2, PW = TW + 2, PROWS = (TD + 2) * PH * PW;  // 648
  constexpr int MAXK = 128;                    // 27 * C_in <= 108
  constexpr int MAXBLK = MAXK / KB;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* patch = reinterpret_cast<T*>(smem);                            // [PROWS][Cin]

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
  constexpr int Cin = CIN, K = 27 * Cin, nblk = (K + KB - 1) / KB;
  const int cout_pad = (p.Cout + 15) & ~15;
  constexpr int BK = ConvTraits<T>::BK;

  // ---- weight fragments: straight from the K-MAJOR image [Cout padded to 64][nblk * KB] (k = tap * C_in + ci, zero padded; ops.packed_cin_weight)
  // into registers, requested before anything else (round 4: the first form gathered the block element by element from the tap-major panel
  // into LDS -- 8 scalar loads and ~450 index instructions per thread and tile, most of a work-group's 28 k cycles, for 4 KiB that every tile
  // of the launch reads identically; 16-byte loads of an L2-resident image need neither LDS nor a barrier)
  uint4 wfr[nblk][NFR];
  {
    const char* wimg = reinterpret_cast<const char*>(p.w) + ((size_t)(cb * BN + l15) * (nblk * KB) + q * VECW) * sizeof(T);
#pragma unroll
    for (int blk = 0; blk < nblk; ++blk)
#pragma unroll
      for (int nf = 0; nf < NFR; ++nf)
        wfr[blk][nf] = *reinterpret_cast<const uint4*>(wimg + ((size_t)nf * 16 * (nblk * KB) + blk * KB) * sizeof(T));
  }
  // ---- stage the patch (zero padded): all of a thread's elements are requested before the first wait ---------------------------------
  const T* xin = reinterpret_cast<const T*>(p.x);
  {
    constexpr int PMAX = (PROWS * 4 + 255) / 256;  // C_in <= 4
    T pv[PMAX];
    bool pok[PMAX];
#pragma unroll
    for (int it = 0; it < PMAX; ++it) {
      if (it * 256 >= PROWS * Cin) break;  // (uniform)
      const int e = tid + it * 256;
      const int ec = e < PROWS * Cin ? e : 0;
      const int row = ec / Cin, ci = ec - row * Cin;
      const int pa = row / (PH * PW), rr = row - pa * (PH * PW), pb = rr / PW, pc = rr - pb * PW;
      const int ud = od0 - p.pd + pa, uh = oh0 - p.ph + pb, uw = ow0 - p.pw + pc;
      pok[it] = (ud >= 0) & (ud < p.Ds) & (uh >= 0) & (uh < p.Hs) & (uw >= 0) & (uw < p.Ws);
      const long long vox = pok[it] ? (((long long)n * p.Ds + ud) * p.Hs + uh) * p.Ws + uw : 0;
      pv[it] = xin[vox * p.x_ld + ci];
    }
#pragma unroll
    for (int it = 0; it < PMAX; ++it) {
      if (it * 256 >= PROWS * Cin) break;
      const int e = tid + it * 256;
      if (e < PROWS * Cin) patch[e] = pok[it] ? pv[it] : (T)0;
    }
  }
  // per-channel epilogue addend (bias + shortcut bias + timestep row, this order) staged once per work-group behind the scratch region (round 5:
  // this kernel still ran the first shared epilogue -- 16 x 3 dependent branched loads per lane, a 64-bit voxel address per row group, the residual
  // of row group i + 1 behind the store of row group i, 48 ds_bpermute round trips for the statistics: most of a work-group's 28 k cycles)
  float* addv = reinterpret_cast<float*>(smem + 4 * 64 * 144);
  if (tid < BN) {
    const int co = cb * BN + tid;
    float addend = 0.f;
    if (co < p.Cout) {
      if (p.bias) addend += p.bias[co];
      if (p.skip_bias) addend += p.skip_bias[co];
      if (p.rowvec) addend += p.rowvec[(long long)n * p.rowvec_bstride + co];
    }
    addv[tid] = addend;
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
#pragma unroll
  for (int blk = 0; blk < nblk; ++blk) {
    int koff[VECW];
#pragma unroll
    for (int i = 0; i < VECW; ++i) {
      const int k = blk * KB + q * VECW + i;
      const int tap = k / Cin, ci = k - tap * Cin;
      const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      koff[i] = k < K ? ((kd * PH + kh) * PW + kw) * Cin + ci : -1;
    }
    const uint4 (&wf)[NFR] = wfr[blk];
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

  // ---- epilogue: the LDS-DMA kernels' (conv_dma_shared.h): residual rows requested first, accumulators + addend through the wave's own
  // transpose scratch, 16-byte row stores, statistics by lane swaps; fixed-order fp64 sum over the waves ------------------------------------------
  const EpTile et = {n, od0, oh0, ow0, cb * BN, 0};
  EpRows<MF * 2> rows0;
  dma_epilogue_rows<T, MF, 3, 0>(p, et, wave * MF, lane, rows0);
  __syncthreads();  // every wave has gathered its operands: the scratch overlays the patch
  constexpr int EPASSES = (NFR * 16 * (int)sizeof(T) + 127) / 128;
  constexpr int CH_PER_PASS = 128 / (int)sizeof(T);
  float st_s[EPASSES][VECW], st_q[EPASSES][VECW];
#pragma unroll
  for (int e = 0; e < EPASSES; ++e)
#pragma unroll
    for (int i = 0; i < VECW; ++i) { st_s[e][i] = 0.f; st_q[e][i] = 0.f; }
  char* scratch = smem + (size_t)wave * MF * 16 * 144;
  dma_epilogue_pass<T, MF, NFR, 3, 0>(p, acc, scratch, addv, et, wave * MF, lane, rows0, st_s, st_q);
  if constexpr (EPASSES > 1) {
    EpRows<MF * 2> rows;
    dma_epilogue_rows<T, MF, 3, 1>(p, et, wave * MF, lane, rows);
    dma_epilogue_pass<T, MF, NFR, 3, 1>(p, acc, scratch, addv, et, wave * MF, lane, rows, st_s, st_q);
  }
  if (p.stats) {
    float ra[EPASSES][VECW], rb[EPASSES][VECW];
#pragma unroll
    for (int e = 0; e < EPASSES; ++e)
#pragma unroll
      for (int i = 0; i < VECW; ++i) { ra[e][i] = wave_segment_sum(st_s[e][i]); rb[e][i] = wave_segment_sum(st_q[e][i]); }
    if (lane < 8) {  // lanes 0..7 hold the wave's sums of VECW consecutive channels each: (sum, sum of squares) pairs
      float* part = reinterpret_cast<float*>(scratch);
#pragma unroll
      for (int e = 0; e < EPASSES; ++e)
#pragma unroll
        for (int i = 0; i < VECW; i += 2)
          *reinterpret_cast<float4*>(part + 2 * (e * CH_PER_PASS + lane * VECW + i)) = make_float4(ra[e][i], rb[e][i], ra[e][i + 1], rb[e][i + 1]);
    }
    __syncthreads();
    if (tid < BN) {
      double a = 0.0, b2 = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float2 v = *reinterpret_cast<const float2*>(smem + (size_t)w * MF * 16 * 144 + tid * 8);
        a += (double)v.x;
        b2 += (double)v.y;
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
            conv_act_vec(v, p.pre_act, PRECISE);
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
// C_out == 1, marching along depth (round 3; configuration 20).  conv_cout1_kernel above reads a 6x6x18 halo per 4x4x16 tile -- every
// input row 2.5 times -- straight into registers with at most two fragment batches in flight and 8 waves per CU: 1.0 TB/s, 13 % of the HBM roof
// it is graded against.  Here a work-group owns an 8 x TWO column of the output and walks a segment of DS planes:
//   * input plane d (its (8 + 2) x (TWO + 2) halo rows, whole channel rows) is staged by the LDS-DMA engine into one of two raw buffers while
//     plane d - 1 is processed: every row is fetched once per work-group (1.33x halo in H / W, (DS + 2) / DS in depth), whole 128-byte rows
//     per DMA piece, nothing waits for HBM inside the plane loop;
//   * phase 1: GroupNorm-apply + SiLU on the raw rows (each halo voxel once per work-group) -> Z[tap][row] = sum_c w[tap][c] a[row][c] for all
//     27 taps by MFMA (taps = the GEMM M dimension, as above);
//   * phase 2: output plane o needs Z of input planes o - 1 (kd = 0), o (kd = 1), o + 1 (kd = 2): two running sums per output voxel live in
//     registers across the walk, each plane adds its nine-tap gathers in the ORDER of the 27-point sum of the tile kernel (bias, taps 0..26), so
//     the two kernels are bit-identical.
// LDS: 2 raw planes + Z = 128 KiB (TWO = 32, 128-byte rows) / 119 KiB (TWO = 16, 256-byte rows): one work-group of 4 waves per CU; the grid is
// sized by the host (depth segments of 2^ltd planes) to at least one work-group per CU.  (reference: diffusion_model_unet.py:1853-1867)
__device__ __attribute__((aligned(64))) unsigned int gm_edge_zero_row[16] = {0};  // the source of every padding row of the marching kernel

// Round 5: 128-byte rows may also walk 8 x 16 columns (LTW = 4): 2 x 24 KiB raw planes + 21 KiB of Z = 69 KiB, TWO work-groups per CU -- the
// kernel is bound by its 2.1 M SiLUs per plane column on ONE wave per SIMD with two barriers per plane (0.094 ms without the prologue, 0.197
// with it); a second work-group runs its transcendental chains under the first one's DMA wait, Z stores and 27-point gather.  Halo 1.41 instead of
// 1.33 rows per output row.
template <typename T, int KS, int LTW>  // KS = C_in / BK 64-byte channel steps (row = KS x 64 bytes); 8 x 2^LTW outputs per plane
__global__ __launch_bounds__(256, (KS == 2 && LTW == 4) ? 2 : 1) void conv_cout1_march_kernel(const GmConvDesc p) {
  constexpr int VECW = ConvTraits<T>::VECW;
  constexpr int BK = ConvTraits<T>::BK;
  constexpr int TH = 8, TWO = 1 << LTW, PH = TH + 2, PW = TWO + 2, PROWS = PH * PW;
  constexpr int NFRAG = (PROWS + 15) / 16, PROWS_PAD = NFRAG * 16;   // 22 fragments / 352 rows (TWO = 32), 12 / 192 (TWO = 16)
  constexpr int ROWB = KS * 64, SLOTS = ROWB / 16, RPP = 1024 / ROWB;  // bytes / 16-byte slots per row, rows per 1 KiB DMA piece
  static_assert(ROWB == 128 || ROWB == 256, "whole rows of 128 or 256 bytes");
  constexpr int NPIECE = PROWS_PAD / RPP, PPW = (NPIECE + 3) / 4;    // DMA pieces per plane / per wave
  constexpr int RAW_BYTES = PROWS_PAD * ROWB;
  constexpr int ZP = PROWS_PAD + 4;                                  // floats per Z row: 4 ZP = 16 (mod 32): the four tap rows of a store hit distinct banks
  constexpr int NTAP = 27, FPW = (NFRAG + 3) / 4;
  constexpr bool PRECISE = sizeof(T) == 4;

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  float* Z = reinterpret_cast<float*>(smem + 2 * RAW_BYTES);         // [27][ZP]
  const unsigned lds0 = (unsigned)(uintptr_t)smem;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int DS = 1 << p.ltd;
  const int ntd = (p.Do + DS - 1) / DS, nth = (p.Ho + TH - 1) / TH, ntw = (p.Wo + TWO - 1) / TWO;
  unsigned b = xcd_remap(blockIdx.x, gridDim.x);
  const int tw_i = b % ntw; b /= ntw;
  const int th_i = b % nth; b /= nth;
  const int td_i = b % ntd; b /= ntd;
  const int n = b;
  const int d_begin = td_i * DS, d_end = min(p.Do, d_begin + DS), oh0 = th_i * TH, ow0 = tw_i * TWO;
  const int D = p.Ds, H = p.Hs, W = p.Ws;  // stride 1, padding 1: output extents = input extents (host-checked)

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

  // ---- this lane's DMA sources: piece i = wave + 4 jj covers rows i RPP .. + RPP - 1, lane -> (row, LDS slot); the bank swizzle is applied to
  // the SOURCE (LDS slot s of row r receives channel slot s ^ swz(r)), padding rows read a zero page -----------------------------------
  auto swz = [](int r) { return ROWB == 128 ? (r >> 1) & 7 : r & 15; };
  int hw_off[PPW];   // voxel offset of this lane's row within an input plane, -1 for a padding row
  int csl[PPW];      // byte offset of the channel slot this lane fetches
#pragma unroll
  for (int jj = 0; jj < PPW; ++jj) {
    const int piece = wave + 4 * jj;
    const int r = piece * RPP + lane / SLOTS, slot = lane % SLOTS;
    const int ph = r / PW, pw = r - ph * PW;
    const int h = oh0 - 1 + ph, w = ow0 - 1 + pw;
    const bool ok = (piece < NPIECE) & (r < PROWS) & (h >= 0) & (h < H) & (w >= 0) & (w < W);
    hw_off[jj] = ok ? h * W + w : -1;
    csl[jj] = (slot ^ swz(r)) << 4;
  }
  const char* zero = reinterpret_cast<const char*>(gm_edge_zero_row);
  const char* xbase = reinterpret_cast<const char*>(p.x);
  const long long rowb = p.x_ld * (long long)sizeof(T);
  auto issue_plane = [&](int d, int buf) __attribute__((always_inline)) {
    const long long plane0 = ((long long)n * D + d) * H * W;
#pragma unroll
    for (int jj = 0; jj < PPW; ++jj) {
      if (wave + 4 * jj < NPIECE) {  // wave-uniform
        const char* src = hw_off[jj] >= 0 ? xbase + (plane0 + hw_off[jj]) * rowb + csl[jj] : zero + ((lane & 3) << 4);
        unsigned keep;
        const unsigned dst = lds0 + (unsigned)buf * RAW_BYTES + (unsigned)(wave + 4 * jj) * 1024u;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
      }
    }
  };

  // ---- this lane's fragment rows (phase 1): validity of the (h, w) position per fragment, LDS offsets ------------------------------------
  unsigned okm = 0;
#pragma unroll
  for (int j = 0; j < FPW; ++j) {
    const int r = (wave + 4 * j) * 16 + l15;
    const int ph = r / PW, pw = r - ph * PW;
    const int h = oh0 - 1 + ph, w = ow0 - 1 + pw;
    if ((wave + 4 * j < NFRAG) & (r < PROWS) & (h >= 0) & (h < H) & (w >= 0) & (w < W)) okm |= 1u << j;
  }
  const int sw_l = swz(l15);  // fragment bases are multiples of 16 rows: the swizzle of a fragment row depends on l15 only

  // ---- phase 2 bookkeeping: thread -> output (oh, ow) of the tile plane ---------------------------------------------------------------
  const bool gth = tid < TH * TWO;
  const int gh = tid >> LTW, gw = tid & (TWO - 1);
  const int oh = oh0 + gh, ow = ow0 + gw;
  const bool out_ok = gth && oh < p.Ho && ow < p.Wo;
  float add0 = p.bias ? p.bias[0] : 0.f;
  if (p.rowvec) add0 += p.rowvec[(long long)n * p.rowvec_bstride];
  float s_prev = add0, s_cur = add0;  // partial sums of output planes ip - 1 and ip (see the rotation below)

  const int ip0 = d_begin - 1, ip1 = d_end;  // input planes walked (inclusive)
  auto plane_valid = [&](int d) { return d >= 0 && d < D; };
  if (plane_valid(ip0)) issue_plane(ip0, 0);
  for (int ip = ip0; ip <= ip1; ++ip) {
    const int buf = (ip - ip0) & 1;
    const bool valid = plane_valid(ip);  // block-uniform
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of plane ip have landed (and its output stores have left)
    __syncthreads();                                  // ... everyone's; the previous plane's gather is over (Z is free), raw[buf ^ 1] has been read
    if (ip + 1 <= ip1 && plane_valid(ip + 1) && !(p.debug_flags & 4)) issue_plane(ip + 1, buf ^ 1);  // (debug_flags 1 / 2 / 4: bench-only ablations --
    if (valid && !(p.debug_flags & 1)) {                                                                  //  no phase 1 / no phase 2 / no DMA; results are garbage)
      const char* raw = smem + (size_t)buf * RAW_BYTES;
#pragma unroll
      for (int j = 0; j < FPW; ++j) {
        const int f = wave + 4 * j;
        if (f >= NFRAG) continue;  // wave-uniform
        const int row = f * 16 + l15;
        const bool ok = (okm >> j) & 1u;
        f32x4_t z0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, z1 = z0;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          uint4 xf = *reinterpret_cast<const uint4*>(raw + (size_t)row * ROWB + (((s * 4 + q) ^ sw_l) << 4));
          if (p.pre_scale || p.pre_act) {
            float v[VECW];
            Vec16<T>::unpack(xf, v);
            if (p.pre_scale) {
#pragma unroll
              for (int i = 0; i < VECW; ++i) v[i] = v[i] * sc[s][i] + sh[s][i];
            }
            if (p.pre_act) {
              conv_act_vec(v, p.pre_act, PRECISE);
            }
            xf = Vec16<T>::pack(v);
          }
          xf = make_uint4(ok ? xf.x : 0u, ok ? xf.y : 0u, ok ? xf.z : 0u, ok ? xf.w : 0u);  // zero padding of the ACTIVATED tensor
          Mma<T>::run(wf[0][s], xf, z0);
          Mma<T>::run(wf[1][s], xf, z1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          Z[(4 * q + r) * ZP + row] = z0[r];
          if (16 + 4 * q + r < NTAP) Z[(16 + 4 * q + r) * ZP + row] = z1[r];
        }
      }
    }
    __syncthreads();
    // ---- phase 2: this input plane's contribution to output planes ip + 1 (kd = 0), ip (kd = 1), ip - 1 (kd = 2, which completes it) ------
    if (gth && !(p.debug_flags & 2)) {
      float out_v = s_prev, mid = s_cur, nxt = add0;
      if (valid) {
        const float* zb = Z + (gh * PW + gw);
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) nxt += zb[t9 * ZP + (t9 / 3) * PW + (t9 % 3)];
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) mid += zb[(9 + t9) * ZP + (t9 / 3) * PW + (t9 % 3)];
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) out_v += zb[(18 + t9) * ZP + (t9 / 3) * PW + (t9 % 3)];
      }
      const int o = ip - 1;
      if (out_ok && o >= d_begin && o < d_end) {
        const long long vox = (((long long)n * p.Do + o) * p.Ho + oh) * p.Wo + ow;
        float sum = out_v;
        if (p.res) sum += ElemIO<T>::ld(reinterpret_cast<const T*>(p.res) + vox * p.res_ld);
        ElemIO<T>::st(reinterpret_cast<T*>(p.y) + vox * p.y_ld, conv_post_act(sum, p.post_act));
      }
      s_prev = mid;
      s_cur = nxt;
    }
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
  const long long operands = (648 * 4 * es + 15) & ~15LL;  // the halo patch (the weight fragments come from the K-major image in registers)
  const long long scratch = 4LL * 64 * 144;
  return (operands > scratch ? operands : scratch) + 256;  // + the epilogue addend vector (64 floats, at byte 4 * 64 * 144)
}
extern "C" int gm_conv_cout1_eligible(const GmConvDesc* d) {
  const int bk = d->dtype == GM_F32 ? 16 : 32, vecw = d->dtype == GM_F32 ? 4 : 8;
  const int ks = d->Cin / bk;
  return edge_common(d) && d->Cout == 1 && d->Cin % bk == 0 && (ks == 2 || ks == 4 || (d->dtype == GM_F32 && ks == 8)) &&
         d->x_ld % vecw == 0 && (reinterpret_cast<uintptr_t>(d->x) & 15) == 0 && d->stats == nullptr;
}
extern "C" long long gm_conv_cout1_lds_bytes() { return 27LL * 660 * 4; }

// configuration 20: the marching C_out == 1 kernel.  Rows of 128 bytes walk 8 x 32 output columns (ltw = 5), rows of 256 bytes 8 x 16 (ltw = 4);
// ltd = log2 of the planes per depth segment (2 .. 5), chosen by the host so that the grid has at least one work-group per CU.
extern "C" int gm_conv_cout1m_eligible(const GmConvDesc* d) {
  const int es = d->dtype == GM_F32 ? 4 : 2, vecw = 16 / es;
  const long long rowbytes = (long long)d->Cin * es;
  return d->kd == 3 && d->kh == 3 && d->kw == 3 && d->sd == 1 && d->sh == 1 && d->sw == 1 && d->dd == 1 && d->dh == 1 && d->dw == 1 &&
         d->pd == 1 && d->ph == 1 && d->pw == 1 && d->in_mode == 0 && d->Cout == 1 && d->Do == d->Ds && d->Ho == d->Hs && d->Wo == d->Ws && d->Ds >= 4 &&
         (rowbytes == 128 || rowbytes == 256) && d->lth == 3 && (d->ltw == 4 || (rowbytes == 128 && d->ltw == 5)) && d->ltd >= 2 && d->ltd <= 5 &&
         d->x_ld % vecw == 0 && (reinterpret_cast<uintptr_t>(d->x) & 15) == 0 && d->stats == nullptr && d->x2 == nullptr && !d->skip_x[0] &&
         ((d->pre_scale == nullptr) == (d->pre_shift == nullptr)) && (long long)d->N * d->Ds * d->Hs * d->Ws < (1LL << 40);
}
extern "C" long long gm_conv_cout1m_lds_bytes(const GmConvDesc* d) {
  const int es = d->dtype == GM_F32 ? 4 : 2;
  const long long rowb = (long long)d->Cin * es, two = 1LL << d->ltw;
  const long long prows = 10 * (two + 2), pad = (prows + 15) / 16 * 16;
  return 2 * pad * rowb + 27LL * (pad + 4) * 4;
}

template <typename KernT>
static void edge_launch(KernT kern, const GmConvDesc& d, unsigned nblocks, size_t smem, hipStream_t st) {
  static const void* seen[64];  // raise the dynamic-LDS limit once per kernel instantiation
  static int nseen = 0;
  const void* key = reinterpret_cast<const void*>(kern);
  bool found = false;
  for (int i = 0; i < nseen; ++i) found |= seen[i] == key;
  if (!found) {
    hipError_t e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) (void)hipGetLastError();
    if (nseen < 64) seen[nseen++] = key;
  }
  kern<<<dim3(nblocks), 256, smem, st>>>(d);
}

extern "C" int gm_conv_cin_launch(const GmConvDesc* dp, unsigned nblocks, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const size_t smem = (size_t)gm_conv_cin_lds_bytes(dp);
#define GM_CIN_LAUNCH(T)                                                        \
  switch (dp->Cin) {                                                            \
    case 1: edge_launch(conv_cin_kernel<T, 1>, *dp, nblocks, smem, st); return 0; \
    case 2: edge_launch(conv_cin_kernel<T, 2>, *dp, nblocks, smem, st); return 0; \
    case 3: edge_launch(conv_cin_kernel<T, 3>, *dp, nblocks, smem, st); return 0; \
    case 4: edge_launch(conv_cin_kernel<T, 4>, *dp, nblocks, smem, st); return 0; \
    default: return -3;                                                         \
  }
  if (dp->dtype == GM_F32) { GM_CIN_LAUNCH(float) }
  if (dp->dtype == GM_BF16) { GM_CIN_LAUNCH(bf16_raw) }
#undef GM_CIN_LAUNCH
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

extern "C" int gm_conv_cout1m_launch(const GmConvDesc* dp, unsigned nblocks, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const size_t smem = (size_t)gm_conv_cout1m_lds_bytes(dp);
  const long long rowb = (long long)dp->Cin * (dp->dtype == GM_F32 ? 4 : 2);
  if (dp->dtype == GM_F32) {
    if (rowb == 128 && dp->ltw == 5) edge_launch(conv_cout1_march_kernel<float, 2, 5>, *dp, nblocks, smem, st);
    else if (rowb == 128) edge_launch(conv_cout1_march_kernel<float, 2, 4>, *dp, nblocks, smem, st);
    else edge_launch(conv_cout1_march_kernel<float, 4, 4>, *dp, nblocks, smem, st);
    return 0;
  }
  if (dp->dtype == GM_BF16) {
    if (rowb == 128 && dp->ltw == 5) edge_launch(conv_cout1_march_kernel<bf16_raw, 2, 5>, *dp, nblocks, smem, st);
    else if (rowb == 128) edge_launch(conv_cout1_march_kernel<bf16_raw, 2, 4>, *dp, nblocks, smem, st);
    else edge_launch(conv_cout1_march_kernel<bf16_raw, 4, 4>, *dp, nblocks, smem, st);
    return 0;
  }
  return -2;
}

