// Fused bf16 flash-attention backward on the LDS-DMA structure of attention_dma.hip (round 5; VERDICT r4 "missing" #1): dQ, dK, dV of
// O = softmax(scale Q K^T) V per (sample, head) with NOTHING L x L in HBM -- the composed path (attention_bwd.hip's score pass + three
// weight-gradient launches) writes and re-reads P, dS and dS^T.  (reference: torch autograd through diffusion_model_unet.py:407-415, :139-153)
//
// One kernel template, three modes, all in the forward kernel's orientation: a lane OWNS one row (an MFMA column: its fragments stay in
// registers for the whole launch), the OTHER rows stream through LDS in tiles copied by global_load_lds_dwordx4 (bank swizzle applied on
// the source side, tile t+1 in flight while tile t is multiplied, one barrier per tile), and the per-tile scores sit in the accumulators in
// exactly the layout the next MFMA wants as its B operand:
//   LSE  own = queries, stream = K:               S^T = K Q^T, online (max, sum) -> LSE[q]            (only when the caller has no LSE)
//   DQ   own = queries, stream = K, V, K^T:       S^T = K Q^T, dP^T = V dO^T, dS^T = P^T (dP^T - D) scale, dQ^T += K^T dS^T
//   DKV  own = keys,    stream = Q, dO, Q^T, dO^T: S = Q K^T, dP = dO V^T, P, dS, dV^T += dO^T P, dK^T += Q^T dS
// The row contractions (the last product of DQ, the last two of DKV) need the streamed operand TRANSPOSED: K^T, Q^T and dO^T images
// [channel][position] are written once per call by the forward's vt_pack_kernel (position order inside each 32-block = the order the MFMA
// consumes accumulator fragments; 16 MB each at 32 768 tokens x 256 channels, ~10 us), so a transposed tile is a set of plain row segments.
// S is recomputed by DQ and DKV (7 GEMM units for the 5 of the mathematics): no atomics, every result deterministic, dQ / dK / dV are
// accumulated in fp32 registers over the whole sweep and stored once as bf16.
#include "attn_common.h"

static __device__ __attribute__((aligned(64))) unsigned int abd_zero_row[16] = {0};

enum { ABD_DQ = 0, ABD_DKV = 1, ABD_LSE = 2 };

// D[q] = dO[q] . O[q] and the (LSE log2 e, D) pairs the gradient kernels read (they evaluate P = 2^(s scale log2 e - LSE log2 e): one FMA and a
// bare v_exp_f32 per score): ld[bh][q] for q < Lq, (+inf, 0) for the padding (a padded query's
// probability is exp(x - inf) = 0: no masks in the DKV loop).  One thread per (query, 8 channels), 256 threads = 256 * 8 / DH queries.
// `lse` NULL: the LSE sweep's (max, sum) pairs ms[slice][bh][q] of `nslices` key slices are merged here, slice order.
template <int DH>
__global__ __launch_bounds__(256) void abd_prep_kernel(const GmAttnBwdDesc p, const float* __restrict__ lse, const float2* __restrict__ ms, int nslices,
                                                       float2* __restrict__ ld, int ld_stride) {
  constexpr int VPR = DH / 8, QPB = 256 / VPR;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int q = blockIdx.x * QPB + threadIdx.x / VPR, cv = threadIdx.x % VPR;
  float part = 0.f;
  if (q < p.Lq) {
    const bf16_raw* O = reinterpret_cast<const bf16_raw*>(p.o) + ((long long)b * p.Lq + q) * p.o_ld + h * DH + cv * 8;
    const bf16_raw* G = reinterpret_cast<const bf16_raw*>(p.go) + ((long long)b * p.Lq + q) * p.go_ld + h * DH + cv * 8;
    const uint4 a = *reinterpret_cast<const uint4*>(O), g = *reinterpret_cast<const uint4*>(G);
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      part += __uint_as_float(aw[i] << 16) * __uint_as_float(gw[i] << 16) + __uint_as_float(aw[i] & 0xffff0000u) * __uint_as_float(gw[i] & 0xffff0000u);
  }
#pragma unroll
  for (int o = 1; o < VPR; o <<= 1) part += __shfl_xor(part, o, 64);  // VPR = 8 / 16 / 32 lanes of one query: a fixed tree (deterministic)
  if (cv == 0 && q < ld_stride) {
    float l = INFINITY;
    if (q < p.Lq) {
      if (lse) l = lse[(long long)bh * p.Lq + q];
      else {
        float M = -INFINITY, tot = 0.f;
        for (int z = 0; z < nslices; ++z) M = fmaxf(M, ms[((long long)z * gridDim.y + bh) * p.Lq + q].x);
        for (int z = 0; z < nslices; ++z) {
          const float2 t = ms[((long long)z * gridDim.y + bh) * p.Lq + q];
          tot += t.x > -INFINITY ? t.y * __expf(t.x - M) : 0.f;
        }
        l = M + __logf(tot);
      }
    }
    ld[(long long)bh * ld_stride + q] = q < p.Lq ? make_float2(l * 1.4426950408889634f, part) : make_float2(INFINITY, 0.f);  // (LSE in base-2 units)
  }
}

// out[row][c] (bf16) = sum over the slices, slice order, of part[slice][o][bh][row][c] (fp32): one thread per (row, 4 channels)
template <int DH>
__global__ __launch_bounds__(256) void abd_combine_kernel(const float* __restrict__ part, int nslices, int nout, int H, int L, bf16_raw* __restrict__ dst0,
                                                          long long dst0_ld, bf16_raw* __restrict__ dst1, long long dst1_ld) {
  constexpr int VPR = DH / 4;
  const int o = blockIdx.z;  // output of the sweep (DKV: 0 = dk, 1 = dv)
  bf16_raw* dst = o == 0 ? dst0 : dst1;
  const long long dst_ld = o == 0 ? dst0_ld : dst1_ld;
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const long long it = (long long)blockIdx.x * 256 + threadIdx.x;
  const int row = (int)(it / VPR), cv = (int)(it % VPR);
  if (row >= L) return;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int z = 0; z < nslices; ++z) {
    const float4 t = *reinterpret_cast<const float4*>(part + ((((long long)z * nout + o) * gridDim.y + bh) * L + row) * DH + cv * 4);
    a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
  }
  *reinterpret_cast<uint2*>(dst + ((long long)b * L + row) * dst_ld + h * DH + cv * 4) = make_uint2(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w));
}

template <int DH, int MODE, int NW>
__global__ __launch_bounds__(64 * NW, 2) void abd_kernel(const GmAttnBwdDesc p, const bf16_raw* __restrict__ t1, const bf16_raw* __restrict__ t2, int t_ld,
                                                         const float2* __restrict__ ld, int ld_stride, float* __restrict__ part) {
  constexpr int TR = (DH == 256 && MODE != ABD_LSE) ? 32 : 64;  // streamed rows per tile (LDS: 4 (DKV) / 3 (DQ) / 1 (LSE) images of TR x DH x 2 bytes, twice)
  constexpr int KF = TR / 16;                     // 16-row fragments per tile
  constexpr int S2N = TR / 32;                    // 32-position blocks per tile (one MFMA k-step of the row contractions each)
  constexpr int STEPS = DH / 32, DF = DH / 16;
  constexpr int NROWB = DH * 2;                   // natural tile row bytes
  constexpr int SPR = NROWB / 16;                 // 16-byte slots per natural row (8 / 16 / 32)
  constexpr int KNB = SPR < 16 ? SPR : 16;        // swizzle span
  constexpr int KSH = KNB == 16 ? 0 : 1;
  constexpr int NRPP = 64 / SPR;                  // natural rows per 1 KB DMA piece
  constexpr int TROWB = TR * 2;                   // transposed tile row bytes (64 / 128)
  constexpr int SPRT = TROWB / 16;                // 4 / 8
  constexpr int TRPP = 64 / SPRT;                 // transposed rows (channels) per DMA piece (16 / 8)
  constexpr int TBYTES = TR * NROWB;              // bytes of one tile, natural or transposed
  constexpr int PIECES = TBYTES / 1024, PPW = PIECES / NW;
  constexpr int NNAT = MODE == ABD_LSE ? 1 : 2, NTR = MODE == ABD_DKV ? 2 : (MODE == ABD_DQ ? 1 : 0);
  constexpr int LD_OFF = (NNAT + NTR) * TBYTES;   // DKV: the tile's 128 (LSE, D) pairs (1 KB piece)
  constexpr int BUF_BYTES = LD_OFF + (MODE == ABD_DKV ? 1024 : 0);
  static_assert(PIECES % NW == 0 && DH * TROWB == TBYTES, "tile bytes split evenly over the waves");

  extern __shared__ __attribute__((aligned(1024))) char smem[];  // [2][natural tiles][transposed tiles][(LSE, D)]
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, qg = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  constexpr bool OWN_K = MODE == ABD_DKV;
  const int Lown = OWN_K ? p.Lk : p.Lq, Lst = OWN_K ? p.Lq : p.Lk;
  const int own = blockIdx.x * (NW * 16) + wave * 16 + l15;
  const bool own_ok = own < Lown;

  // own-row operands (B fragments: lane (row l15, slot qg) holds channels s*32 + qg*8 .. +7) and the streamed ones
  const bf16_raw* X1 = reinterpret_cast<const bf16_raw*>(OWN_K ? p.k : p.q) + (long long)b * Lown * (OWN_K ? p.k_ld : p.q_ld) + h * DH;
  const bf16_raw* X2 = reinterpret_cast<const bf16_raw*>(OWN_K ? p.v : p.go) + (long long)b * Lown * (OWN_K ? p.v_ld : p.go_ld) + h * DH;
  const long long x1_ld = OWN_K ? p.k_ld : p.q_ld, x2_ld = OWN_K ? p.v_ld : p.go_ld;
  const long long n1_rowb = (OWN_K ? p.q_ld : p.k_ld) * 2, n2_rowb = (OWN_K ? p.go_ld : p.v_ld) * 2;
  const char* N1 = reinterpret_cast<const char*>(reinterpret_cast<const bf16_raw*>(OWN_K ? p.q : p.k) + (long long)b * Lst * (OWN_K ? p.q_ld : p.k_ld) + h * DH);
  const char* N2 = reinterpret_cast<const char*>(reinterpret_cast<const bf16_raw*>(OWN_K ? p.go : p.v) + (long long)b * Lst * (OWN_K ? p.go_ld : p.v_ld) + h * DH);
  const char* T1 = reinterpret_cast<const char*>(t1 + (long long)bh * DH * t_ld);
  const char* T2 = reinterpret_cast<const char*>(t2 + (long long)bh * DH * t_ld);
  const char* LD = reinterpret_cast<const char*>(ld + (long long)bh * ld_stride);
  const char* zero = reinterpret_cast<const char*>(abd_zero_row);

  // ---- this lane's DMA sources (two lane constants per tile kind) ------------------------------------------------------------------
  const int nrow0 = wave * PPW * NRPP + lane / SPR, nslot = lane % SPR;      // piece j: natural row nrow0 + j * NRPP
  const int trow0 = wave * PPW * TRPP + lane / SPRT, tslot = lane % SPRT;    // piece j: channel row trow0 + j * TRPP
  const int t_rowb = t_ld * 2;                                                // host-checked: DH * t_ld * 2 < 2^31
  auto tkey = [](int row) __attribute__((always_inline)) { return TR == 64 ? ((row >> 1) & 7) : ((0 - (row >> 2)) & 3); };
  // full tiles: a wave-uniform base per image (advanced on the scalar unit) + launch-constant lane offsets (attention_dma.hip, round 5); the last,
  // partial tile takes the general path (rows >= Lst -> zeros)
  unsigned noff1[PPW], noff2[NNAT == 2 ? PPW : 1], toff[NTR >= 1 ? PPW : 1];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int row = nrow0 + j * NRPP, trow = trow0 + j * TRPP;
    const unsigned slotb = (unsigned)((nslot ^ ((row >> KSH) & (KNB - 1))) * 16);
    noff1[j] = (unsigned)row * (unsigned)n1_rowb + slotb;
    if (NNAT == 2) noff2[NNAT == 2 ? j : 0] = (unsigned)row * (unsigned)n2_rowb + slotb;
    if (NTR >= 1) toff[NTR >= 1 ? j : 0] = (unsigned)trow * (unsigned)t_rowb + (unsigned)((tslot ^ tkey(trow)) * 16);
  }
  auto issue_tile = [&](int tile, int buf) __attribute__((always_inline)) {
    const int r0 = tile * TR;
    const unsigned dst = lds0 + (unsigned)buf * BUF_BYTES;
    if (r0 + TR <= Lst) {
      const char* b1 = N1 + (long long)r0 * n1_rowb;
      const char* b2 = N2 + (long long)r0 * n2_rowb;
      const char* bt1 = T1 + (long long)r0 * 2;
      const char* bt2 = T2 + (long long)r0 * 2;
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        attn_dma16_s(b1, noff1[j], dst + (unsigned)(wave * PPW + j) * 1024);
        if (NNAT == 2) attn_dma16_s(b2, noff2[NNAT == 2 ? j : 0], dst + TBYTES + (unsigned)(wave * PPW + j) * 1024);
        if (NTR >= 1) attn_dma16_s(bt1, toff[NTR >= 1 ? j : 0], dst + NNAT * TBYTES + (unsigned)(wave * PPW + j) * 1024);
        if (NTR == 2) attn_dma16_s(bt2, toff[NTR >= 1 ? j : 0], dst + (NNAT + 1) * TBYTES + (unsigned)(wave * PPW + j) * 1024);
      }
      if (MODE == ABD_DKV && wave == 0) attn_dma16_s(LD + (long long)r0 * 8, (unsigned)lane * 16u, dst + LD_OFF);
      return;
    }
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int row = nrow0 + j * NRPP;
      const int slotb = (nslot ^ ((row >> KSH) & (KNB - 1))) * 16;
      const bool ok = r0 + row < Lst;
      attn_dma16(ok ? N1 + (long long)(r0 + row) * n1_rowb + slotb : zero + ((lane & 3) << 4), dst + (unsigned)(wave * PPW + j) * 1024);
      if (NNAT == 2)
        attn_dma16(ok ? N2 + (long long)(r0 + row) * n2_rowb + slotb : zero + ((lane & 3) << 4), dst + TBYTES + (unsigned)(wave * PPW + j) * 1024);
      if (NTR >= 1) {
        const int trow = trow0 + j * TRPP;
        const int off = trow * t_rowb + r0 * 2 + ((tslot ^ tkey(trow)) * 16);
        attn_dma16(T1 + off, dst + NNAT * TBYTES + (unsigned)(wave * PPW + j) * 1024);
        if (NTR == 2) attn_dma16(T2 + off, dst + (NNAT + 1) * TBYTES + (unsigned)(wave * PPW + j) * 1024);
      }
    }
    if (MODE == ABD_DKV && wave == 0) attn_dma16(LD + (long long)r0 * 8 + lane * 16, dst + LD_OFF);  // 128 pairs; the array is padded by 128
  };

  // ---- operand read offsets (see attention_dma.hip: the swizzle XORs slot bits 0..3 only) ---------------------------------------------
  constexpr int KA = STEPS < 4 ? STEPS : 4;
  int naddr[KA];
  const int fk = (l15 >> KSH) & (KNB - 1);
#pragma unroll
  for (int s = 0; s < KA; ++s) naddr[s] = l15 * NROWB + (((s * 4 + qg) ^ fk) << 4);
  int taddr[S2N];   // transposed fragment (channel row l15 of fragment 0, 32-position block s2); fragment d adds d * 16 * TROWB
  const int fv = tkey(l15);
#pragma unroll
  for (int s2 = 0; s2 < S2N; ++s2) taddr[s2] = NNAT * TBYTES + l15 * TROWB + (((s2 * 4 + qg) ^ fv) << 4);

  uint4 x1[STEPS], x2[NNAT == 2 ? STEPS : 1];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const uint4 v = *reinterpret_cast<const uint4*>(X1 + (long long)(own_ok ? own : 0) * x1_ld + s * 32 + qg * 8);
    x1[s] = make_uint4(own_ok ? v.x : 0u, own_ok ? v.y : 0u, own_ok ? v.z : 0u, own_ok ? v.w : 0u);
    if constexpr (NNAT == 2) {
      const uint4 w = *reinterpret_cast<const uint4*>(X2 + (long long)(own_ok ? own : 0) * x2_ld + s * 32 + qg * 8);
      x2[s] = make_uint4(own_ok ? w.x : 0u, own_ok ? w.y : 0u, own_ok ? w.z : 0u, own_ok ? w.w : 0u);
    }
  }
  constexpr int NOUT = MODE == ABD_DKV ? 2 : (MODE == ABD_DQ ? 1 : 0);
  f32x4_t out[NOUT > 0 ? NOUT : 1][NOUT > 0 ? DF : 1];
#pragma unroll
  for (int o = 0; o < NOUT; ++o)
#pragma unroll
    for (int d = 0; d < DF; ++d) out[o][d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float own_lse = 0.f, own_d = 0.f;
  if (MODE == ABD_DQ) {
    const float2 t = ld[(long long)bh * ld_stride + (own_ok ? own : 0)];
    own_lse = t.x; own_d = t.y;
  }
  float m_run = -INFINITY, l_run = 0.f;  // LSE mode
  const float scale2 = p.scale * 1.4426950408889634f;

  // gridDim.z > 1: work-group z sweeps the z-th slice of the streamed tiles and leaves fp32 partial results in `part` (one head of 4 096 tokens
  // is only 32 work-groups of 128 own rows); abd_combine_kernel / abd_prep_kernel add the slices in slice order (deterministic)
  const int ntiles_all = (Lst + TR - 1) / TR;
  const int tps = (ntiles_all + (int)gridDim.z - 1) / (int)gridDim.z;
  const int tile0 = (int)blockIdx.z * tps;
  const int ntiles = min(ntiles_all, tile0 + tps);  // this slice: tiles [tile0, ntiles) (possibly empty)
  if (tile0 < ntiles) issue_tile(tile0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int tile = tile0; tile < ntiles; ++tile) {
    const int r0 = tile * TR;
    const char* buf = smem + (size_t)((tile - tile0) & 1) * BUF_BYTES;
    if (tile + 1 < ntiles) issue_tile(tile + 1, (tile + 1 - tile0) & 1);  // its buffer was last read two barriers ago

    // ---- stage 1: acc1 = N1 X1^T (scores), acc2 = N2 X2^T (dP): D[streamed row 16 kf + 4 qg + r][own row l15] -------------------------
    f32x4_t acc1[KF], acc2[NNAT == 2 ? KF : 1];
    {
      constexpr int NKQ = NNAT * STEPS * KF, PDW = MODE == ABD_DKV ? 2 : 6, PD = NKQ < PDW ? NKQ : PDW;
      uint4 kq[PD];
      // item i: operand (i % NNAT), fragment (i / NNAT) % KF, k-step i / (NNAT * KF)
      auto nread = [&](int i) __attribute__((always_inline)) {
        const int which = i % NNAT, kf = (i / NNAT) % KF, s = i / (NNAT * KF);
        return *reinterpret_cast<const uint4*>(buf + which * TBYTES + naddr[s % KA] + (s / KA) * (KA * 64) + kf * 16 * NROWB);
      };
#pragma unroll
      for (int i = 0; i < PD; ++i) kq[i] = nread(i);
      __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
#pragma unroll
      for (int i = 0; i < NKQ; ++i) {
        const int which = i % NNAT, kf = (i / NNAT) % KF, s = i / (NNAT * KF);
        const f32x4_t zero4 = (f32x4_t){0.f, 0.f, 0.f, 0.f};  // (the first k-step takes a literal zero accumulator: no register clears per tile)
        if (NNAT == 1 || which == 0)
          acc1[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kq[i % PD]), __builtin_bit_cast(bf16x8_t, x1[s]), s == 0 ? zero4 : acc1[kf], 0, 0, 0);
        else
          acc2[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kq[i % PD]), __builtin_bit_cast(bf16x8_t, x2[NNAT == 2 ? s : 0]),
                                                             s == 0 ? zero4 : acc2[NNAT == 2 ? kf : 0], 0, 0, 0);
        if (i + PD < NKQ) kq[i % PD] = nread(i + PD);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i + PD < NKQ) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }

    if constexpr (MODE == ABD_LSE) {  // online (max, sum) over this lane's 4 KF keys of the tile
      float x[KF][4], tmax = -INFINITY;
#pragma unroll
      for (int kf = 0; kf < KF; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          x[kf][r] = r0 + kf * 16 + qg * 4 + r < Lst ? acc1[kf][r] * p.scale : -INFINITY;
          tmax = fmaxf(tmax, x[kf][r]);
        }
      const float mn = fmaxf(m_run, tmax);
      if (mn > -INFINITY) {
        float add = 0.f;
#pragma unroll
        for (int kf = 0; kf < KF; ++kf)
#pragma unroll
          for (int r = 0; r < 4; ++r) add += __expf(x[kf][r] - mn);
        l_run = l_run * __expf(m_run - mn) + add;
        m_run = mn;
      }
    } else {
      // ---- stage 2: P = exp(scale S - LSE), dS = P (dP - D) scale, packed as the B operand of the row contractions (32 streamed rows per
      //      k-step, position order of vt_pack_kernel: fragment 2 s2 -> elements 0..3, fragment 2 s2 + 1 -> 4..7) ------------------------------
      uint4 pf[MODE == ABD_DKV ? S2N : 1], df[S2N];
#pragma unroll
      for (int s2 = 0; s2 < S2N; ++s2) {
        float pv[2][4], dv[2][4];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int kf = 2 * s2 + hf;
          float lse_r[4], d_r[4];
          if constexpr (MODE == ABD_DKV) {  // per streamed query: rows 16 kf + 4 qg + r of the tile's (LSE, D) pairs
            const float4 a = *reinterpret_cast<const float4*>(buf + LD_OFF + (kf * 16 + qg * 4) * 8);
            const float4 c = *reinterpret_cast<const float4*>(buf + LD_OFF + (kf * 16 + qg * 4) * 8 + 16);
            lse_r[0] = a.x; d_r[0] = a.y; lse_r[1] = a.z; d_r[1] = a.w; lse_r[2] = c.x; d_r[2] = c.y; lse_r[3] = c.z; d_r[3] = c.w;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (MODE == ABD_DKV) {
              pv[hf][r] = __builtin_amdgcn_exp2f(fmaf(acc1[kf][r], scale2, -lse_r[r]));
              dv[hf][r] = pv[hf][r] * (acc2[kf][r] - d_r[r]) * p.scale;
            } else {
              const bool ok = r0 + kf * 16 + qg * 4 + r < Lst;
              pv[hf][r] = ok ? __builtin_amdgcn_exp2f(fmaf(acc1[kf][r], scale2, -own_lse)) : 0.f;
              dv[hf][r] = pv[hf][r] * (acc2[kf][r] - own_d) * p.scale;
            }
          }
        }
        if (MODE == ABD_DKV)
          pf[s2] = make_uint4(pack_bf16x2(pv[0][0], pv[0][1]), pack_bf16x2(pv[0][2], pv[0][3]), pack_bf16x2(pv[1][0], pv[1][1]), pack_bf16x2(pv[1][2], pv[1][3]));
        df[s2] = make_uint4(pack_bf16x2(dv[0][0], dv[0][1]), pack_bf16x2(dv[0][2], dv[0][3]), pack_bf16x2(dv[1][0], dv[1][1]), pack_bf16x2(dv[1][2], dv[1][3]));
      }

      // ---- stage 3: out^T += T (P | dS): D[channel 16 d + 4 qg + r][own row l15] ---------------------------------------------------------
      {
        constexpr int NVQ = NTR * DF * S2N, PDW = MODE == ABD_DKV ? 2 : 6, PD = NVQ < PDW ? NVQ : PDW;
        uint4 vq[PD];
        // item i: transposed operand (i % NTR) (DKV: 0 = Q^T -> dK with dS, 1 = dO^T -> dV with P), block (i / NTR) % S2N, channel fragment i / (NTR * S2N)
        auto tread = [&](int i) __attribute__((always_inline)) {
          const int which = i % NTR, s2 = (i / NTR) % S2N, d = i / (NTR * S2N);
          return *reinterpret_cast<const uint4*>(buf + which * TBYTES + taddr[s2] + d * 16 * TROWB);
        };
#pragma unroll
        for (int i = 0; i < PD; ++i) vq[i] = tread(i);
        __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
#pragma unroll
        for (int i = 0; i < NVQ; ++i) {
          const int which = i % NTR, s2 = (i / NTR) % S2N, d = i / (NTR * S2N);
          const uint4 bop = (MODE == ABD_DKV && which == 1) ? pf[MODE == ABD_DKV ? s2 : 0] : df[s2];
          out[which][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vq[i % PD]), __builtin_bit_cast(bf16x8_t, bop), out[which][d], 0, 0, 0);
          if (i + PD < NVQ) vq[i % PD] = tread(i + PD);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (i + PD < NVQ) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
    }
    // the next tile has landed (this wave's pieces) and this wave is done reading the current one
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  if constexpr (MODE == ABD_LSE) {  // the four lanes sharing a query hold disjoint key subsets
    const float M = attn_quad_max(m_run);
    const float st = attn_quad_sum(m_run > -INFINITY ? l_run * __expf(m_run - M) : 0.f);
    if (qg == 0 && own_ok)  // (max, sum) of this slice: [slice][bh][q]
      reinterpret_cast<float2*>(part)[((long long)blockIdx.z * gridDim.y + bh) * p.Lq + own] = make_float2(M, st);
    return;
  }
  if (!own_ok) return;
  if (part) {  // [slice][output][bh][own row][DH] fp32
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      float* prow = part + ((((long long)blockIdx.z * NOUT + o) * gridDim.y + bh) * Lown + own) * DH;
#pragma unroll
      for (int d = 0; d < DF; ++d) *reinterpret_cast<float4*>(prow + d * 16 + qg * 4) = make_float4(out[o][d][0], out[o][d][1], out[o][d][2], out[o][d][3]);
    }
    return;
  }
  // ---- store: out[o][d][r] = channel 16 d + 4 qg + r of this lane's own row (DQ: dq; DKV: 0 -> dk, 1 -> dv) ------------------------------
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    bf16_raw* dst = MODE == ABD_DQ ? reinterpret_cast<bf16_raw*>(p.dq) + ((long long)b * p.Lq + own) * p.dq_ld + h * DH
                  : o == 0 ? reinterpret_cast<bf16_raw*>(p.dk) + ((long long)b * p.Lk + own) * p.dk_ld + h * DH
                           : reinterpret_cast<bf16_raw*>(p.dv) + ((long long)b * p.Lk + own) * p.dv_ld + h * DH;
#pragma unroll
    for (int d = 0; d < DF; ++d)
      *reinterpret_cast<uint2*>(dst + d * 16 + qg * 4) = make_uint2(pack_bf16x2(out[o][d][0], out[o][d][1]), pack_bf16x2(out[o][d][2], out[o][d][3]));
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
static long long abd_pad64(long long x) { return (x + 63) / 64 * 64; }
static long long abd_al(long long x) { return (x + 255) & ~255LL; }

static bool abd_eligible(const GmAttnBwdDesc& d) {
  auto al = [](const void* p, int a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
  // (row strides below 2^24 elements: the DMA lane offsets row * ld * 2 are 32-bit, as attn_dma_eligible bounds k_ld for the forward; at most 65 535
  //  (sample, head) pairs: they are a grid dimension)
  auto ok = [&](const void* p, long long ld) { return p && ld % 8 == 0 && ld > 0 && ld < (1LL << 24) && al(p, 16); };
  const long long lmax = abd_pad64(d.Lq > d.Lk ? d.Lq : d.Lk);
  return d.dtype == GM_BF16 && (long long)d.B * d.H <= 65535 && (d.dh == 64 || d.dh == 128 || d.dh == 256) && d.Lq >= 1 && d.Lk >= 1 && (long long)d.dh * lmax * 2 < (1LL << 31) &&
         ok(d.q, d.q_ld) && ok(d.k, d.k_ld) && ok(d.v, d.v_ld) && ok(d.o, d.o_ld) && ok(d.go, d.go_ld) && ok(d.dq, d.dq_ld) && ok(d.dk, d.dk_ld) &&
         ok(d.dv, d.dv_ld);
}

// slices of the streamed tiles: until every CU has a work-group, at least 4 tiles per slice (as attention_dma.hip's key slices)
static int abd_force_split = 0;  // 0 = by problem size (tests / benchmarks force a count)
extern "C" void gm_attention_backward_fused_set_split(int nsplit) { abd_force_split = (nsplit >= 1 && nsplit <= 16) ? nsplit : 0; }
static int abd_split(const GmAttnBwdDesc& d, int Lown, int Lst) {
  if (abd_force_split) return abd_force_split;
  const long long wgs = (long long)d.B * d.H * ((Lown + 127) / 128), tiles = (Lst + (d.dh == 256 ? 31 : 63)) / (d.dh == 256 ? 32 : 64);  // (the LSE sweep's 64-row tiles: >= 2 per slice)
  int sp = 1;
  while (wgs * sp < 256 && sp < 16 && tiles / (2 * sp) >= 4) sp *= 2;
  return sp;
}

// workspace: Q^T | dO^T | K^T images, the (LSE, D) pairs (padded by 128 per (sample, head)), the LSE sweep's (max, sum) pairs for callers
// without an LSE, the fp32 partial results of a sliced sweep
static long long abd_part_bytes(const GmAttnBwdDesc& d) {
  const long long bh = (long long)d.B * d.H;
  const int skv = abd_split(d, d.Lk, d.Lq), sq = abd_split(d, d.Lq, d.Lk);
  const long long a = skv > 1 ? (long long)skv * 2 * bh * d.Lk * d.dh * 4 : 0, b = sq > 1 ? (long long)sq * bh * d.Lq * d.dh * 4 : 0;
  return a > b ? a : b;
}
extern "C" long long gm_attention_backward_fused_workspace_bytes(const GmAttnBwdDesc* d) {
  if (!d || !abd_eligible(*d)) return 0;
  const long long bh = (long long)d->B * d->H, lqp = abd_pad64(d->Lq), lkp = abd_pad64(d->Lk);
  return 2 * abd_al(bh * d->dh * lqp * 2) + abd_al(bh * d->dh * lkp * 2) + abd_al(bh * (lqp + 128) * 8) + abd_al(16 * bh * d->Lq * 8) + abd_al(abd_part_bytes(*d));
}

template <int DH>
static void abd_launch(const GmAttnBwdDesc& d, const float* lse, hipStream_t st) {
  constexpr int NW = 8;
  constexpr int TR = DH == 256 ? 32 : 64, TB = TR * DH * 2;
  constexpr size_t lds_lse = 2 * (size_t)(64 * DH * 2), lds_dq = 2 * (size_t)(3 * TB), lds_dkv = 2 * (size_t)(4 * TB + 1024);
  static bool attr_set = false;
  if (!attr_set) {
    const void* ks[3] = {reinterpret_cast<const void*>(abd_kernel<DH, ABD_DQ, NW>), reinterpret_cast<const void*>(abd_kernel<DH, ABD_DKV, NW>),
                         reinterpret_cast<const void*>(abd_kernel<DH, ABD_LSE, NW>)};
    for (const void* k : ks)
      if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError();
    attr_set = true;
  }
  const long long bh = (long long)d.B * d.H, lqp = abd_pad64(d.Lq), lkp = abd_pad64(d.Lk);
  char* w = reinterpret_cast<char*>(d.workspace);
  bf16_raw* qt = reinterpret_cast<bf16_raw*>(w);   w += abd_al(bh * DH * lqp * 2);
  bf16_raw* got = reinterpret_cast<bf16_raw*>(w);  w += abd_al(bh * DH * lqp * 2);
  bf16_raw* kt = reinterpret_cast<bf16_raw*>(w);   w += abd_al(bh * DH * lkp * 2);
  float2* ld = reinterpret_cast<float2*>(w);       w += abd_al(bh * (lqp + 128) * 8);
  float2* ms = reinterpret_cast<float2*>(w);       w += abd_al(16 * bh * d.Lq * 8);
  float* part = reinterpret_cast<float*>(w);
  const int ld_stride = (int)lqp + 128;
  const bf16_raw* Q = reinterpret_cast<const bf16_raw*>(d.q);
  const bf16_raw* K = reinterpret_cast<const bf16_raw*>(d.k);
  const bf16_raw* G = reinterpret_cast<const bf16_raw*>(d.go);
  if (3 * bh <= 65535) {
    const bf16_raw* const rows[3] = {Q, G, K};
    const long long lds_[3] = {d.q_ld, d.go_ld, d.k_ld};
    bf16_raw* const images[3] = {qt, got, kt};
    const int ls[3] = {d.Lq, d.Lq, d.Lk}, lps[3] = {(int)lqp, (int)lqp, (int)lkp};
    gm_attn_pack_transposed3(rows, lds_, images, ls, lps, d.B, d.H, DH, st);
  } else {
    gm_attn_pack_transposed(Q, d.q_ld, qt, d.B, d.H, d.Lq, (int)lqp, DH, st);
    gm_attn_pack_transposed(G, d.go_ld, got, d.B, d.H, d.Lq, (int)lqp, DH, st);
    gm_attn_pack_transposed(K, d.k_ld, kt, d.B, d.H, d.Lk, (int)lkp, DH, st);
  }
  const int skv = abd_split(d, d.Lk, d.Lq), sq = abd_split(d, d.Lq, d.Lk);
  const unsigned wq = (d.Lq + NW * 16 - 1) / (NW * 16), wk = (d.Lk + NW * 16 - 1) / (NW * 16);
  if (!lse) abd_kernel<DH, ABD_LSE, NW><<<dim3(wq, (unsigned)bh, sq), 64 * NW, lds_lse, st>>>(d, nullptr, nullptr, 0, nullptr, 0, reinterpret_cast<float*>(ms));
  constexpr int QPB = 256 * 8 / DH;
  abd_prep_kernel<DH><<<dim3((ld_stride + QPB - 1) / QPB, (unsigned)bh), 256, 0, st>>>(d, lse, ms, sq, ld, ld_stride);
  abd_kernel<DH, ABD_DKV, NW><<<dim3(wk, (unsigned)bh, skv), 64 * NW, lds_dkv, st>>>(d, qt, got, (int)lqp, ld, ld_stride, skv > 1 ? part : nullptr);
  if (skv > 1) {
    const dim3 cg((unsigned)(((long long)d.Lk * (DH / 4) + 255) / 256), (unsigned)bh, 2);
    abd_combine_kernel<DH><<<cg, 256, 0, st>>>(part, skv, 2, d.H, d.Lk, reinterpret_cast<bf16_raw*>(d.dk), d.dk_ld, reinterpret_cast<bf16_raw*>(d.dv), d.dv_ld);
  }
  abd_kernel<DH, ABD_DQ, NW><<<dim3(wq, (unsigned)bh, sq), 64 * NW, lds_dq, st>>>(d, kt, nullptr, (int)lkp, ld, ld_stride, sq > 1 ? part : nullptr);
  if (sq > 1) {
    const dim3 cg((unsigned)(((long long)d.Lq * (DH / 4) + 255) / 256), (unsigned)bh);
    abd_combine_kernel<DH><<<cg, 256, 0, st>>>(part, sq, 1, d.H, d.Lq, reinterpret_cast<bf16_raw*>(d.dq), d.dq_ld, nullptr, 0);
  }
}

// dq, dk, dv (bf16, written once; no accumulation into the destinations) of o = softmax(scale q k^T) v.  `lse` = log sum_k exp(scale q.k) per
// (sample, head, query) as fp32 [B*H][Lq] when the caller has it (the training forward), else NULL: one more sweep computes it.
extern "C" int gm_attention_backward_fused(const GmAttnBwdDesc* dp, const float* lse, void* stream) {
  GM_REQUIRE(dp, "null descriptor");
  const GmAttnBwdDesc& d = *dp;
  GM_REQUIRE(d.B >= 0 && d.H > 0, "bad batch / head geometry");
  if (d.B == 0) return 0;
  GM_REQUIRE((long long)d.B * d.H <= 65535, "too many (batch, head) pairs for one launch");
  GM_REQUIRE(abd_eligible(d), "the fused bf16 backward takes bf16 operands, head dim 64 / 128 / 256, 16-byte aligned rows");
  GM_REQUIRE(d.workspace && d.workspace_bytes >= gm_attention_backward_fused_workspace_bytes(dp), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (d.dh == 64) abd_launch<64>(d, lse, st);
  else if (d.dh == 128) abd_launch<128>(d, lse, st);
  else abd_launch<256>(d, lse, st);
  GM_LAUNCH_CHECK();
}
