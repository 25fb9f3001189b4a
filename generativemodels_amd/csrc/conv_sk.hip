// The K slices of a split-K 3x3x3 convolution over a SMALL volume (the 32^3 / 16^3 / 8^3 levels of a latent UNet: fewer 256-voxel x 64-channel
// tiles than the chip has CUs, so the K chunks of a tile are dealt to `ksplit` work-groups and a combine kernel sums their fp32 partials --
// GmConvDesc.ksplit / kpartial, conv.hip).  Round 3 ran those slices on the general tile kernel (conv_dma.hip, cfg 11): built for a CU that is
// shared by two work-groups and streams a long K loop, it keeps three 12 KiB weight panels in a ring and requests a panel two tap groups ahead.
// A slice is ONE chunk of 27 taps (rarely two or three): 216 MFMAs per wave against nine panel round trips of ~1 us each, plus the set-up of
// the multi-tile walk and an epilogue it does not need -- 24 us per launch for ~3 us of arithmetic (profiles/r04_c3_layers.txt).
// Here a work-group owns its CU (1 work-group of 8 waves per CU: the grid is at most one wave of work-groups) and spends the LDS on latency:
//   * the patch (42 KiB) and ALL NINE weight panels of a chunk (108 KiB) are requested back to back before anything waits: one exposed round
//     trip per chunk instead of nine.  Request order per wave: panels 0-2 (they need the K slice and the channel block of the work item only: the
//     ~3 k cycles of patch placement arithmetic run under their flight), the patch, panels 3-8; everything lands in request order, so the tap
//     groups start behind counted `s_waitcnt vmcnt(N)` + a barrier while the later panels are still in flight (barriers in front of groups
//     0, 3, 4 only), and the panel pieces are requested with a scalar base + 32-bit lane offset (no 64-bit VALU address arithmetic);
//   * the fused GroupNorm + SiLU prologue takes its scale / shift from a 256-byte LDS copy that every wave requests AHEAD of its patch pieces
//     (an ordinary global load would make hipcc wait for `vmcnt(0)` -- every panel -- at its first use);
//   * no tile walk, no epilogue: the partial sums go from the accumulators to kpartial exactly as conv_dma.hip's slices wrote them.
// Same tile (4x4x16 voxels x 64 channels, 8 waves x 32 voxels), same LDS layouts, same packed weights and the same order of MFMAs per
// accumulator as cfg 11: the partials are bit-identical to the round-3 path (tests/test_gpu_kernels.py pins that).
// (reference op: the ResnetBlock / Upsample convolutions of the latent UNet, generative/networks/nets/diffusion_model_unet.py:589-696)
#include "conv_dma_shared.h"

__device__ __attribute__((aligned(64))) unsigned int gm_sk_zero_row[16] = {0};  // the source of every padding row

// one LDS-DMA piece with a wave-uniform base (SGPR pair) and a 32-bit per-lane byte offset: no 64-bit address arithmetic on the VALU
// (8 waves x 18 panel pieces of a chunk: the requests, not their addresses, should bound the issue phase)
__device__ __forceinline__ void dma16_base(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  const unsigned long long a = (unsigned long long)sbase;  // (made uniform explicitly: an "s" operand the divergence analysis cannot prove uniform is
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));  // handed over in
  const unsigned long long sb = ((unsigned long long)hi << 32) | lo;  // VGPRs; the builtin returns int: widened unsigned, or bit 31 of the low half smears into the high one)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sb), "s"(lds_dst)
               : "memory");
}

namespace sk {
constexpr int NW = 8, MF = 2, NFR = 4, KS = 3, G = 3, NGROUPS = 9;
constexpr int TD = 4, TH = 4, TW = 16, BM = 256, BN = 64;
constexpr int PD = 6, PH = 6, PW = 18, PLANE = 112, PROWS = PD * PLANE, PPIECES = PROWS / 16, PPW = (PPIECES + NW - 1) / NW;
constexpr int WROWS = G * BN, WPW = 2;
constexpr int PATCH_BYTES = PROWS * DMA_ROWB, WBUF_BYTES = WROWS * DMA_ROWB;
constexpr int AFF_OFF = PATCH_BYTES + NGROUPS * WBUF_BYTES, AFF_WAVE = 256;
// PRE from the input's statistic tables (GmConvDesc.pre_stats: the GroupNorm finalised HERE, conv_sn.hip's recipe, round 6): behind the per-wave copies, the (scale |
// shift) of THIS SLICE's channels (<= TAB_CH) and the fp64 channel sums of the groups they belong to (<= COVER_CH channels: the slice + the rest of its first and last group)
constexpr int TAB_CH = 128, COVER_CH = 256;
constexpr int TAB_OFF = AFF_OFF + NW * AFF_WAVE, CSUM_OFF = TAB_OFF + 2 * TAB_CH * 4;
constexpr int LDS_BYTES = CSUM_OFF + COVER_CH * 16;  // 43 008 + 110 592 + 2 048 + 1 024 + 4 096 = 160 768
// request order of a chunk, per wave: panels 0 .. FIRST-1 (they need the channel block and the K slice of the work item only), [scale | shift],
// the patch pieces (they need the whole placement: ~3 k cycles of index arithmetic that now run under the first panels' flight), panels FIRST .. 8
constexpr int FIRST = 3;
constexpr int AFTER_PATCH = (NGROUPS - FIRST) * WPW;  // LDS-DMA instructions per wave behind the patch: 12
constexpr int LAST_BARRIER_GROUP = 4;                // group 0 waits for the patch, groups FIRST .. 3 for their own panel, group 4 for everything that is left
static_assert(LDS_BYTES <= 160 * 1024, "one work-group per CU");
}  // namespace sk

template <typename T, bool PRE>
__global__ __launch_bounds__(512, 2) void conv_sk_kernel(const GmConvDesc p) {
  using namespace sk;
  constexpr int BK = ConvTraits<T>::BK;
  constexpr int VECW = ConvTraits<T>::VECW;
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // [patch][9 weight panels][8 x (scale | shift) of the chunk]
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  // bench-only (tools/sk_timeline.py, debug_flags bit 12): thread 0 stamps the shader clock at the phase boundaries into 16 slots per work-group
  // BEHIND the partial sums (the tool allocates them); slots 14 / 15 = the 100 MHz wall clock at entry / after the last store was issued
  unsigned long long* stamps = nullptr;
  if ((p.debug_flags & 4096) && tid == 0) {
    stamps = reinterpret_cast<unsigned long long*>(p.kpartial + (long long)p.ksplit * p.N * p.Do * p.Ho * p.Wo * p.Cout) + (long long)blockIdx.x * 16;
    stamps[14] = __builtin_amdgcn_s_memrealtime();
    stamps[0] = __builtin_readcyclecounter();
  }
#define SK_STAMP(k) do { if (stamps) stamps[k] = __builtin_readcyclecounter(); } while (0)

  // ---- the work item: (K slice, tile, channel block), dealt to the XCDs in contiguous ranges like conv_dma.hip's work list ----------------
  const int ntd = (p.Do + TD - 1) / TD, nth = (p.Ho + TH - 1) / TH, ntw = (p.Wo + TW - 1) / TW;
  const int ncb = (p.Cout + BN - 1) / BN;
  const int ksplit = p.ksplit;
  const int nchunks = p.Cin / BK, cout_pad = (p.Cout + 15) & ~15;
  const int cps = (nchunks + ksplit - 1) / ksplit;
  const unsigned nwork = (unsigned)p.N * ntd * nth * ntw * ncb * (unsigned)ksplit;
  const unsigned xcd = blockIdx.x & 7;
  const unsigned q8 = nwork >> 3, r8 = nwork & 7, cx = q8 + (xcd < r8 ? 1u : 0u), sx = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  if ((blockIdx.x >> 3) >= cx) return;  // (never: the grid is nwork work-groups)
  unsigned b = sx + (blockIdx.x >> 3);
  const unsigned tiles_all = nwork / (unsigned)ksplit;
  // (the digits of a work item are wave-uniform; hipcc does the divisions on the VALU and then treats everything derived from them as divergent --
  // readfirstlane puts them, and the address arithmetic that follows, back on the scalar unit)
  const int ks = __builtin_amdgcn_readfirstlane((int)(b / tiles_all));
  b -= (unsigned)ks * tiles_all;
  const int cb = __builtin_amdgcn_readfirstlane((int)(b % ncb)); b /= ncb;
  const int c_begin = min(nchunks, ks * cps), c_end = min(nchunks, c_begin + cps);

  // ---- weight panels: per-lane byte offset within a (chunk, group) panel image; a row beyond cout_pad reads the last row instead (its products
  // land in accumulator rows that are never stored) ------------------------------------------------------------------------------------------
  const char* zero = reinterpret_cast<const char*>(gm_sk_zero_row);
  unsigned wsrc[WPW];
#pragma unroll
  for (int h = 0; h < WPW; ++h) {
    const int row = h == 0 ? 16 * wave + (lane >> 2) : 128 + 8 * wave + ((lane & 31) >> 2);
    const int u = row / BN, col = row % BN;
    const int co = min(cb * BN + col, cout_pad - 1);
    wsrc[h] = (unsigned)((u * cout_pad + co) * DMA_ROWB + (((lane & 3) ^ dma_swz(row)) << 4));
  }
  const char* wbase = reinterpret_cast<const char*>(p.w);
  auto issue_panels = [&](int chunk, int g0, int g1) __attribute__((always_inline)) {
#pragma unroll
    for (int g = g0; g < g1; ++g) {
      const char* panel = wbase + (long long)(chunk * NGROUPS + g) * G * cout_pad * DMA_ROWB;  // wave-uniform
      const unsigned dst = lds0 + PATCH_BYTES + (unsigned)g * WBUF_BYTES;
      dma16_base(panel, wsrc[0], dst + (unsigned)(16 * wave) * DMA_ROWB);
      if (lane < 32) dma16_base(panel, wsrc[1], dst + (unsigned)(128 + 8 * wave) * DMA_ROWB);
    }
  };
  if (c_begin < c_end) issue_panels(c_begin, 0, FIRST);
  SK_STAMP(1);

  // ---- the tile, and the patch rows of this lane (layouts of conv_dma.hip: rows of 64 bytes, the bank swizzle applied on the source side) ------
  const int tw_i = __builtin_amdgcn_readfirstlane((int)(b % ntw)); b /= ntw;
  const int th_i = __builtin_amdgcn_readfirstlane((int)(b % nth)); b /= nth;
  const int td_i = __builtin_amdgcn_readfirstlane((int)(b % ntd)); b /= ntd;
  const int n = __builtin_amdgcn_readfirstlane((int)b);
  const int od0 = td_i * TD, oh0 = th_i * TH, ow0 = tw_i * TW;
  int psw = 0, pvox[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int row = 16 * (wave + NW * j) + (lane >> 2);
    const int pa = row / PLANE, rr = row - pa * PLANE;
    const int pb = rr / PW, lc = rr - pb * PW;
    psw |= dma_swz(lc) << (2 * j);
    const int ud = od0 - p.pd + pa, uh = oh0 - p.ph + pb, uw = ow0 - p.pw + lc;
    const bool ok = (row < PROWS) & (rr < PH * PW) & (ud >= 0) & (ud < p.Ds) & (uh >= 0) & (uh < p.Hs) & (uw >= 0) & (uw < p.Ws);
    pvox[j] = ok ? ((n * p.Ds + ud) * p.Hs + uh) * p.Ws + uw : -1;
  }
  const char* xbase = reinterpret_cast<const char*>(p.x);
  const char* x2base = reinterpret_cast<const char*>(p.x2);
  const long long xrowb = p.x_ld * (long long)sizeof(T), x2rowb = p.x2_ld * (long long)sizeof(T);
  const int nchunks0 = p.x2 ? p.cin_split / BK : nchunks;

  const bool from_stats = PRE && p.pre_stats[0] != nullptr;  // (work-group uniform)
  // [scale | shift] (PRE) and the patch pieces of a chunk
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
        // (both addresses computed, then selected: a conditional 64-bit multiply compiles to a branch per piece)
        const char* in_src = cbase + (long long)(pvox[j] & ~(pvox[j] >> 31)) * rowb + (((lane & 3) ^ ((psw >> (2 * j)) & 3)) << 4);
        const char* pad_src = zero + ((lane & 3) << 4);
        dma16(pvox[j] >= 0 ? in_src : pad_src, lds0 + (unsigned)(16 * (wave + NW * j)) * DMA_ROWB);
      }
    }
  };
  // GroupNorm-apply + activation IN LDS on this wave's own landed pieces (conv_dma.hip: transform_patch; same arithmetic and rounding as gm_gn_apply)
  auto transform_patch = [&](int chunk) __attribute__((always_inline)) {
    float sc[VECW], sh[VECW];
    // the wave's own copy of the chunk's (scale | shift), or the slice's table built from the statistics
    const float* aff = from_stats ? reinterpret_cast<const float*>(smem + TAB_OFF) + (chunk - c_begin) * BK : reinterpret_cast<const float*>(smem + AFF_OFF + wave * AFF_WAVE);
    const int shoff = from_stats ? TAB_CH : BK;
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

  if (c_begin < c_end) {
    issue_patch(c_begin);
    issue_panels(c_begin, FIRST, NGROUPS);
  }
  SK_STAMP(2);

  // ---- PRE from statistics (conv_sn.hip's recipe, the shared gn_short_* order: bit-identical to gm_gn_finalize_channels' short form): the channel sums of the groups
  // this slice's channels belong to, then (scale, shift) of the slice's channels -- under the flight of the chunk's requests ------------------------------------------
  if (PRE && from_stats && c_begin < c_end) {
    float* tab = reinterpret_cast<float*>(smem + TAB_OFF);
    double* csum = reinterpret_cast<double*>(smem + CSUM_OFF);
    const int cpg = p.Cin / p.pre_groups;
    const int ch0 = c_begin * BK, ch1 = c_end * BK;
    const int cov0 = (ch0 / cpg) * cpg, cov1 = min(p.Cin, ((ch1 - 1) / cpg + 1) * cpg);  // whole groups (host-checked: cov1 - cov0 <= COVER_CH)
    const long long V = (long long)p.Ds * p.Hs * p.Ws;
    for (int c = cov0 + tid; c < cov1; c += 64 * NW) {
      const double2 v = gn_short_channel_sum(p.pre_stats[0], p.pre_S[0], p.pre_C[0], p.pre_stats[1], p.pre_S[1], p.pre_C[1], p.N, n, c);
      csum[2 * (c - cov0)] = v.x; csum[2 * (c - cov0) + 1] = v.y;
    }
    __syncthreads();
    for (int c = ch0 + tid; c < ch1; c += 64 * NW) {
      const int g = c / cpg;
      double a = 0.0, b2 = 0.0;
      for (int j = 0; j < cpg; ++j) { a += csum[2 * (g * cpg + j - cov0)]; b2 += csum[2 * (g * cpg + j - cov0) + 1]; }
      float sc1, sh1;
      gn_short_scale_shift(a, b2, cpg, V, p.pre_eps, p.pre_gamma ? p.pre_gamma[c] : 1.f, p.pre_beta ? p.pre_beta[c] : 0.f, sc1, sh1);
      tab[c - ch0] = sc1;
      tab[TAB_CH + (c - ch0)] = sh1;
    }
    __syncthreads();
  }

  // ---- operand read addresses (conv_dma.hip: XADDR / WADDR) -----------------------------------------------------------------------------------
  int xa[KS];
  {
    const int m0 = wave * MF * 16 + l15;
    const int a = m0 >> 6, bb0 = (m0 >> 4) & 3, c = m0 & 15;
#pragma unroll
    for (int kw = 0; kw < KS; ++kw) xa[kw] = (a * PLANE + bb0 * PW + c + kw) * DMA_ROWB + ((q ^ dma_swz(c + kw)) << 4);
  }
  const int wa0 = PATCH_BYTES + l15 * DMA_ROWB + ((q ^ dma_swz(l15)) << 4);
#define SK_XADDR(hk, kw) (xa[kw] + (hk) * (PW * DMA_ROWB))
#define SK_WADDR(nf) (wa0 + (nf) * (16 * DMA_ROWB))
  f32x4_t acc[NFR][MF];
#pragma unroll
  for (int nf = 0; nf < NFR; ++nf)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  uint4 xf[2][MF], wf[2][4];
  auto read_tap = [&](int tap, int set) __attribute__((always_inline)) {
    const int g = tap / G, u = tap % G;
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) wf[set][nf] = *reinterpret_cast<const uint4*>(smem + SK_WADDR(nf) + g * WBUF_BYTES + u * (BN * DMA_ROWB));
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) xf[set][mf] = *reinterpret_cast<const uint4*>(smem + SK_XADDR(mf + kh, kw) + kd * (PLANE * DMA_ROWB));
  };
  auto mma_tap = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) Mma<T>::run(wf[set][nf], xf[set][mf], acc[nf][mf]);
  };

  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    // the patch, this wave's scale / shift copy and panels 0 .. FIRST-1 have landed when only the AFTER_PATCH later panel instructions are in flight
    dma_wait<AFTER_PATCH>();
    if (chunk == c_begin) SK_STAMP(3);
    if (PRE) transform_patch(chunk);
    if (chunk == c_begin) SK_STAMP(4);
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int g = tap / G, u = tap % G;
      // a barrier in front of group 0 (everyone's patch pieces and first panels), of groups FIRST .. 3 (their own panel: the later ones stay in
      // flight) and of group 4 (everything that is left); no other group waits
      const bool barrier_here = u == 0 && (g == 0 || (g >= FIRST && g <= LAST_BARRIER_GROUP));
      if (barrier_here) {
        if (g == 0) dma_wait<AFTER_PATCH>();
        else if (g == 3) dma_wait<(NGROUPS - 1 - 3) * WPW>();
        else dma_wait<0>();
        static_assert(FIRST == 3 && LAST_BARRIER_GROUP == 4, "the wait counts above are written for panels 0-2 ahead of the patch");
        __builtin_amdgcn_s_barrier();
        if (chunk == c_begin && g == 0) SK_STAMP(5);
        if (chunk == c_begin && g == LAST_BARRIER_GROUP) SK_STAMP(6);
        read_tap(tap, tap & 1);
      }
      // two operand sets: the next tap's six reads are issued, THEN this tap's MFMAs (whose operands were requested a whole tap earlier).  hipcc
      // merges the sets and reads right in front of each use unless the order is pinned (sched_barrier: nothing moves across)
      const int gn = (tap + 1) / G;
      const bool next_behind_barrier = u == G - 1 && gn >= FIRST && gn <= LAST_BARRIER_GROUP;
      if (tap + 1 < 27 && !next_behind_barrier) read_tap(tap + 1, (tap + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_tap(tap & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (chunk == c_begin) SK_STAMP(7);
    if (chunk + 1 < c_end) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave is done with this chunk's patch and panels
      issue_panels(chunk + 1, 0, FIRST);
      issue_patch(chunk + 1);
      issue_panels(chunk + 1, FIRST, NGROUPS);
    }
  }

  // ---- fused 1x1 shortcut convolution: its K chunks ride with the last slice (conv_dma.hip, same LDS staging, same order of MFMAs) -----------
  if (p.skip_x[0] && ks == ksplit - 1) {
    const int nsc0 = p.skip_cin[0] / BK, nsc = nsc0 + (p.skip_x[1] ? p.skip_cin[1] / BK : 0);
    const int pswz = ((lane & 3) ^ dma_swz(lane >> 2)) << 4;
    int svox[MF];
#pragma unroll
    for (int h = 0; h < MF; ++h) {
      const int m = wave * (MF * 16) + h * 16 + (lane >> 2);
      const int od = od0 + (m >> 6), oh = oh0 + ((m >> 4) & 3), ow = ow0 + (m & 15);
      svox[h] = (od < p.Do && oh < p.Ho && ow < p.Wo) ? ((n * p.Do + od) * p.Ho + oh) * p.Wo + ow : -1;
    }
    const int wpiece = wave & 3, wcol = wpiece * 16 + (lane >> 2), wco = cb * BN + wcol;
    const int wswz = ((lane & 3) ^ dma_swz(wcol)) << 4;
    int caddr[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int m = (wave * MF + mf) * 16 + l15;
      caddr[mf] = m * DMA_ROWB + ((q ^ dma_swz(m)) << 4);
    }
    const char* wsk = reinterpret_cast<const char*>(p.skip_w);
    for (int sc0 = 0; sc0 < nsc; sc0 += 2) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // patch buffer and panel 0 are free
#pragma unroll
      for (int j = 0; j < 2; ++j) {
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
          if ((wave >> 2) == j) {  // waves 0-3 move panel 0, waves 4-7 panel 1
            const char* src = wco < cout_pad ? wsk + ((long long)sc * cout_pad + wco) * DMA_ROWB + wswz : zero + ((lane & 3) << 4);
            dma16(src, lds0 + PATCH_BYTES + (unsigned)(j * BN + wpiece * 16) * DMA_ROWB);
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (sc0 + j < nsc) {
          uint4 xs[MF], ws[4];
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) xs[mf] = *reinterpret_cast<const uint4*>(smem + caddr[mf] + j * (BM * DMA_ROWB));
#pragma unroll
          for (int nf = 0; nf < 4; ++nf) ws[nf] = *reinterpret_cast<const uint4*>(smem + SK_WADDR(nf) + j * (BN * DMA_ROWB));
#pragma unroll
          for (int nf = 0; nf < 4; ++nf)
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) Mma<T>::run(ws[nf], xs[mf], acc[nf][mf]);
        }
      }
    }
  }

  SK_STAMP(8);
  // ---- this slice's fp32 partial sums -> kpartial[ks][n * V + voxel][Cout] (the combine kernel applies the epilogue) ---------------------------
  const long long nv = (long long)p.N * p.Do * p.Ho * p.Wo;
  float* part = p.kpartial + (long long)ks * nv * p.Cout;
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int m = (wave * MF + mf) * 16 + l15;
    const int od = od0 + (m >> 6), oh = oh0 + ((m >> 4) & 3), ow = ow0 + (m & 15);
    if (od < p.Do && oh < p.Ho && ow < p.Wo) {
      float* row = part + ((((long long)n * p.Do + od) * p.Ho + oh) * p.Wo + ow) * p.Cout;
#pragma unroll
      for (int nf = 0; nf < NFR; ++nf) {
        const int co = cb * BN + nf * 16 + q * 4;
        if (co < p.Cout)  // host-checked: Cout % 4 == 0
          *reinterpret_cast<float4*>(row + co) = make_float4(acc[nf][mf][0], acc[nf][mf][1], acc[nf][mf][2], acc[nf][mf][3]);
      }
    }
  }
  SK_STAMP(9);
  if (stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (thread 0's own stores have left)
    stamps[10] = __builtin_readcyclecounter();
    stamps[15] = __builtin_amdgcn_s_memrealtime();
  }
}
#undef SK_STAMP
#undef SK_XADDR
#undef SK_WADDR

static int sk_slice_channels(const GmConvDesc* d) {  // channels of the longest K slice
  const int bk = d->dtype == GM_F32 ? 16 : 32, nchunks = d->Cin / bk;
  return ((nchunks + d->ksplit - 1) / d->ksplit) * bk;
}

// process-wide switch (A/B measurements and the bitwise test against conv_dma.hip's slices); results do not depend on it
static int g_sk_enabled = 1;
extern "C" void gm_conv_sk_set_enabled(int on) { g_sk_enabled = on; }

// a split-K launch this kernel takes: cfg 11 geometry (gm_conv_dma_eligible has been checked by the caller), direct input mode
extern "C" int gm_conv_sk_eligible(const GmConvDesc* d) {
  return g_sk_enabled && d->cfg == 11 && d->ksplit > 1 && d->kpartial != nullptr && d->in_mode == 0 && d->kd == 3 && d->kh == 3 && d->kw == 3 &&
         d->sd == 1 && d->sh == 1 && d->sw == 1 && d->ltd == 2 && d->lth == 2 && d->ltw == 4 && (d->dtype == GM_F32 || d->dtype == GM_BF16) &&
         ((d->pre_stats[0] == nullptr && (d->pre_scale == nullptr || d->pre_shift != nullptr)) ||
          // the statistics form: short tables, whole groups, a slice's channels and their groups within the LDS tables
          (d->pre_stats[0] != nullptr && d->pre_scale == nullptr && d->pre_shift == nullptr && d->pre_groups > 0 && d->Cin % d->pre_groups == 0 &&
           d->pre_S[0] >= 1 && d->pre_S[0] <= GN_SHORT_MAX_ROWS && d->pre_C[0] > 0 &&
           ((d->pre_stats[1] == nullptr && d->pre_C[1] == 0 && d->pre_C[0] == d->Cin) ||
            (d->pre_stats[1] != nullptr && d->pre_S[1] >= 1 && d->pre_S[1] <= GN_SHORT_MAX_ROWS && d->pre_C[1] > 0 && d->pre_C[0] + d->pre_C[1] == d->Cin)) &&
           (reinterpret_cast<uintptr_t>(d->pre_stats[0]) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->pre_stats[1]) & 15) == 0 &&
           sk_slice_channels(d) <= sk::TAB_CH && sk_slice_channels(d) + 2 * (d->Cin / d->pre_groups) <= sk::COVER_CH));
}

template <typename T, bool PRE>
static void launch_sk(const GmConvDesc& d, unsigned nblocks, hipStream_t st) {
  static bool attr_set = false;
  auto kern = conv_sk_kernel<T, PRE>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) (void)hipGetLastError();
    attr_set = true;
  }
  kern<<<dim3(nblocks), 512, (size_t)sk::LDS_BYTES, st>>>(d);
}

extern "C" int gm_conv_sk_launch(const GmConvDesc* dp, unsigned nblocks, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const bool pre = dp->pre_scale != nullptr || dp->pre_stats[0] != nullptr;
  if (dp->dtype == GM_F32) { if (pre) launch_sk<float, true>(*dp, nblocks, st); else launch_sk<float, false>(*dp, nblocks, st); return 0; }
  if (dp->dtype == GM_BF16) { if (pre) launch_sk<bf16_raw, true>(*dp, nblocks, st); else launch_sk<bf16_raw, false>(*dp, nblocks, st); return 0; }
  return -2;
}
