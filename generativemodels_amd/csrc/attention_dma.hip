// bf16 flash attention with the K and V tiles streamed global -> LDS by the LDS-DMA engine, double buffered.
//
// Same mathematics and orientation as attention.hip (S^T = K Q^T, O^T = V^T P^T, a lane owns one query, fp32 scores and
// softmax state; reference: networks/nets/diffusion_model_unet.py:407-415, :143-153, autoencoderkl.py:261-269).  What
// changes is how the operands reach LDS.  attention.hip stages K and V through registers (V transposed in 8x8 register
// blocks) between two barriers per tile, so every tile exposes a global round trip plus the transpose VALU work: 356 TFLOP/s
// at L = 32768, d = 256.  Here
//   * V is transposed ONCE per call by vt_pack_kernel into a scratch image VT[head][channel][key position] whose key order
//     inside each 32-key block is the order the PV MFMA consumes (16 MB at L = 32768: ~10 us), so that a V^T tile is a set of
//     plain 128-byte row segments;
//   * K tiles ([64 keys][d]) and V^T tiles ([d][64 keys]) are copied by global_load_lds_dwordx4 with the bank swizzle applied on
//     the source side (LDS-DMA writes lane-linear), tile t+1 in flight while tile t is multiplied: one barrier per tile, no
//     staging registers, no transposes in the loop;
//   * 8 waves x 16 queries per work-group share each tile (halves the L2 -> LDS traffic of the 4-wave kernel).
#include "attn_common.h"

// bench-only build (-DGM_ATTN_ABLATE=mask, tools/attn_ablate.hip): parts of the tile loop removed to price them (results are then wrong):
// 1 no max / exp / shuffles, 2 no DMA after the first two tiles, 4 no per-tile wait + barrier (with 2), 8 no K fragment reads, 16 no V^T
// fragment reads, 32 no O rescale.  The product library never defines it.
#ifndef GM_ATTN_ABLATE
#define GM_ATTN_ABLATE 0
#endif
#ifndef GM_ATTN_NH2
#define GM_ATTN_NH2 0  // bench-only: 1 = two 32-key online-softmax steps per 64-key tile at 16 queries per wave too (measured: see profiles/r05_attn_fwd_ablation.txt)
#endif
#ifndef GM_ATTN_PD
#define GM_ATTN_PD 8  // operand fragments in flight ahead of the MFMAs (256-register kernels); bench-only builds override it
#endif

__device__ __attribute__((aligned(64))) unsigned int gm_attn_zero_row[16] = {0};

// up to three tensors per launch (the fused backward packs Q, dO and K at once): blockIdx.z = set * (B * H) + (b * H + h)
struct GmPackSet { const bf16_raw* rows; long long ld; bf16_raw* image; int L, L_pad; };
struct GmPackSets { GmPackSet s[3]; };

// ---- V -> VT[b*H + h][c][pos], pos = blk*32 + qq*8 + half*4 + r  <->  key = blk*32 + half*16 + qq*4 + r; keys >= Lk are zero ----
__device__ __forceinline__ void vt_pack_body(const bf16_raw* __restrict__ v, long long v_ld, bf16_raw* __restrict__ vt, int H, int Lk, int Lk_pad, int dh, int bh);
__global__ __launch_bounds__(256) void vt_pack_kernel(const bf16_raw* __restrict__ v, long long v_ld, bf16_raw* __restrict__ vt,
                                                      int H, int Lk, int Lk_pad, int dh) {
  vt_pack_body(v, v_ld, vt, H, Lk, Lk_pad, dh, blockIdx.z);
}
__global__ __launch_bounds__(256) void vt_pack_sets_kernel(const GmPackSets ps, int BH, int H, int dh) {
  const GmPackSet& t = ps.s[blockIdx.z / BH];
  if ((int)blockIdx.x * 64 >= t.L_pad) return;  // (uniform: the grid covers the longest set)
  vt_pack_body(t.rows, t.ld, t.image, H, t.L, t.L_pad, dh, blockIdx.z % BH);
}
__device__ __forceinline__ void vt_pack_body(const bf16_raw* __restrict__ v, long long v_ld, bf16_raw* __restrict__ vt, int H, int Lk, int Lk_pad, int dh, int bh) {
  __shared__ bf16_raw tile[64][64 + 8];
  const int b = bh / H, h = bh % H;
  const int key0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const bf16_raw* vb = v + (long long)b * Lk * v_ld + h * dh;
  for (int it = threadIdx.x; it < 64 * 8; it += 256) {
    const int kr = it >> 3, ch = (it & 7) * 8;
    const bool ok = key0 + kr < Lk;
    const uint4 val = *reinterpret_cast<const uint4*>(vb + (long long)(ok ? key0 + kr : 0) * v_ld + c0 + ch);
    *reinterpret_cast<uint4*>(&tile[kr][ch]) = make_uint4(ok ? val.x : 0u, ok ? val.y : 0u, ok ? val.z : 0u, ok ? val.w : 0u);
  }
  __syncthreads();
  for (int it = threadIdx.x; it < 64 * 8; it += 256) {
    const int c = it >> 3, j = it & 7;           // channel row, 8-position chunk
    const int blk = j >> 2, qq = j & 3;
    alignas(16) bf16_raw o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = tile[blk * 32 + (i >> 2) * 16 + qq * 4 + (i & 3)][c];
    *reinterpret_cast<uint4*>(vt + ((long long)bh * dh + c0 + c) * Lk_pad + key0 + j * 8) = *reinterpret_cast<const uint4*>(o);
  }
}

// QF = 16-query fragments per wave.  QF = 1: 8 waves x 16 queries, one LDS operand read per MFMA (the LDS port bounds the kernel at about
// half the MFMA rate).  QF = 2: 8 waves x 32 queries -- every K / V^T fragment read feeds two MFMAs (0.5 reads per MFMA), the per-tile
// barrier and DMA issue are amortised over twice the matrix work, and the O rescale is skipped (wave-uniformly) on tiles where no running
// maximum moved; the price is registers: Q (64) + O (128) + S (32) accumulators per lane at head dim 256.
// Split-KV (gridDim.z > 1): work-group z handles the z-th slice of the key tiles and writes its un-normalised O (fp32), running maximum
// and sum to `part`; attn_combine_kernel merges the slices (the flash-decoding reduction).  One head of 32768 tokens is only 128
// 256-query tiles -- half the chip -- so the long single-head case of the 3-D UNets runs as 2 slices, short sequences as up to 8.
template <int DH, int QF, int NW, int MINW>
__global__ __launch_bounds__(64 * NW, MINW) void attn_dma_kernel(const GmAttnDesc p, const bf16_raw* __restrict__ vt, int Lk_pad, float* __restrict__ part) {
  constexpr int KT = 64, KF = KT / 16, QPW = 16 * QF;
  constexpr int STEPS = DH / 32;                 // 64-byte k-steps over the head dim
  constexpr int DF = DH / 16;                    // output channel fragments
  constexpr int KROWB = DH * 2;                  // K tile row bytes
  constexpr int SPR = KROWB / 16;                // 16-byte slots per K row (32 / 16 / 8)
  constexpr int KNB = SPR < 16 ? SPR : 16;       // swizzle span
  constexpr int KSH = KNB == 16 ? 0 : (KNB == 8 ? 1 : 2);
  constexpr int KRPP = 64 / SPR;                 // K rows per DMA piece
  constexpr int KPIECES = KT * KROWB / 1024;     // = DH / 8
  constexpr int PPW = KPIECES / NW;              // pieces per wave per operand (4 / 2 / 1)
  constexpr int KBYTES = KT * KROWB, VBYTES = DH * 128;
  static_assert(KPIECES % NW == 0 && DH * 128 / 1024 == KPIECES, "tile bytes split evenly over the waves");

  extern __shared__ __attribute__((aligned(1024))) char smem[];  // [2][K tile][V^T tile]
  const unsigned lds0 = (unsigned)(uintptr_t)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, qg = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  int my_q[QF];
  bool q_ok[QF];
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    my_q[f] = blockIdx.x * (NW * QPW) + wave * QPW + f * 16 + l15;
    q_ok[f] = my_q[f] < p.Lq;
  }

  const bf16_raw* Qb = reinterpret_cast<const bf16_raw*>(p.q) + (long long)b * p.Lq * p.q_ld + h * DH;
  const char* Kb = reinterpret_cast<const char*>(reinterpret_cast<const bf16_raw*>(p.k) + (long long)b * p.Lk * p.k_ld + h * DH);
  const char* Vt = reinterpret_cast<const char*>(vt + (long long)bh * DH * Lk_pad);
  const char* zero = reinterpret_cast<const char*>(gm_attn_zero_row);

  // ---- this lane's DMA sources: recomputed per piece from two lane constants (a per-piece table of 64-bit offsets costs 16 registers
  // the 32-queries-per-wave kernel does not have) ----------------------------------------------------------------------------------
  const int krow0 = wave * PPW * KRPP + lane / SPR, kslot = lane % SPR;   // piece j: K row krow0 + j * KRPP
  const int vrow0 = wave * PPW * 8 + (lane >> 3), vslot = lane & 7;       // piece j: V^T channel row vrow0 + 8 j
  const long long k_rowb = p.k_ld * 2;
  const int v_rowb = Lk_pad * 2;                                          // host-checked: DH * Lk_pad * 2 < 2^31
  // (round 5) 16 queries per wave: this lane's byte offsets inside a tile are launch constants (the swizzle of a row does not depend on the tile:
  // key0 is a multiple of 64), so a whole tile is a wave-uniform base (SGPR pair, advanced on the scalar unit) + PPW + PPW lane offsets -- ~45
  // VALU instructions per tile less than re-deriving 64-bit addresses; only the last, partial tile takes the general path (rows >= Lk -> zeros)
  constexpr bool FASTDMA = QF == 1;
  unsigned koff[FASTDMA ? PPW : 1], voff[FASTDMA ? PPW : 1];
  if (FASTDMA) {
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int krow = krow0 + j * KRPP, vrow = vrow0 + 8 * j;
      koff[j] = (unsigned)krow * (unsigned)k_rowb + (unsigned)((kslot ^ ((krow >> KSH) & (KNB - 1))) * 16);
      voff[j] = (unsigned)vrow * (unsigned)v_rowb + (unsigned)((vslot ^ ((vrow >> 1) & 7)) * 16);
    }
  }
  auto issue_tile = [&](int tile, int buf) __attribute__((always_inline)) {
    const int key0 = tile * KT;
    const unsigned kdst = lds0 + (unsigned)buf * (KBYTES + VBYTES), vdst = kdst + KBYTES;
    const char* vt0 = Vt + (long long)key0 * 2;
    if (FASTDMA && key0 + KT <= p.Lk) {
      const char* kt0 = Kb + (long long)key0 * k_rowb;
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        attn_dma16_s(kt0, koff[j], kdst + (unsigned)(wave * PPW + j) * 1024);
        attn_dma16_s(vt0, voff[j], vdst + (unsigned)(wave * PPW + j) * 1024);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int krow = krow0 + j * KRPP;
      const int kslotb = (kslot ^ ((krow >> KSH) & (KNB - 1))) * 16;
      const bool ok = key0 + krow < p.Lk;
      const char* ks = ok ? Kb + (long long)(key0 + krow) * k_rowb + kslotb : zero + ((lane & 3) << 4);
      attn_dma16(ks, kdst + (unsigned)(wave * PPW + j) * 1024);
      const int vrow = vrow0 + 8 * j;
      attn_dma16(vt0 + vrow * v_rowb + ((vslot ^ ((vrow >> 1) & 7)) * 16), vdst + (unsigned)(wave * PPW + j) * 1024);
    }
  };

  // ---- operand read offsets ---------------------------------------------------------------------------------------------------
  // K fragment (key row l15 of fragment 0, k-step s); fragment kf adds kf*16*KROWB.  The swizzle XORs slot bits 0..3 only, so k-steps
  // that differ in bit 4 and up of their slot index (s >= 4: +256 bytes per 4 steps) share a register and differ by an immediate.
  constexpr int KA = STEPS < 4 ? STEPS : 4;
  int kaddr[KA];
  const int fk = (l15 >> KSH) & (KNB - 1);
#pragma unroll
  for (int s = 0; s < KA; ++s) kaddr[s] = l15 * KROWB + (((s * 4 + qg) ^ fk) << 4);
  int vaddr[2];      // V^T fragment (channel row l15 of fragment 0, key step s2); fragment d adds d*16*128
  const int fv = (l15 >> 1) & 7;
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) vaddr[s2] = KBYTES + l15 * 128 + (((s2 * 4 + qg) ^ fv) << 4);

  // Q fragments (B operand of S^T = K Q^T): lane (query l15, slot qg) holds channels s*32 + qg*8 .. +7
  uint4 qf[QF][STEPS];
#pragma unroll
  for (int f = 0; f < QF; ++f)
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const uint4 v = *reinterpret_cast<const uint4*>(Qb + (long long)(q_ok[f] ? my_q[f] : 0) * p.q_ld + s * 32 + qg * 8);
      qf[f][s] = make_uint4(q_ok[f] ? v.x : 0u, q_ok[f] ? v.y : 0u, q_ok[f] ? v.z : 0u, q_ok[f] ? v.w : 0u);
    }
  f32x4_t oacc[QF][DF];
#pragma unroll
  for (int f = 0; f < QF; ++f)
#pragma unroll
    for (int d = 0; d < DF; ++d) oacc[f][d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float m_run[QF], l_run[QF];
#pragma unroll
  for (int f = 0; f < QF; ++f) { m_run[f] = -INFINITY; l_run[f] = 0.f; }
  const float scale2 = p.scale * 1.4426950408889634f;

  const int ntiles_all = (p.Lk + KT - 1) / KT;
  const int tps = (ntiles_all + (int)gridDim.z - 1) / (int)gridDim.z;  // key tiles per slice
  const int tile0 = (int)blockIdx.z * tps;
  const int ntiles = min(ntiles_all, tile0 + tps);                    // this slice: tiles [tile0, ntiles) (possibly empty)
  if (tile0 < ntiles) issue_tile(tile0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int tile = tile0; tile < ntiles; ++tile) {
    const int key0 = tile * KT;
    const char* buf = smem + (size_t)((tile - tile0) & 1) * (KBYTES + VBYTES);
    if (tile + 1 < ntiles && !((GM_ATTN_ABLATE & 2) && tile > tile0)) issue_tile(tile + 1, (tile + 1 - tile0) & 1);  // its buffer was last read two barriers ago

    // The 64-key tile is consumed in NH slices of KFH key fragments (QF = 2: two 32-key slices, so that the score / probability
    // registers of 32 queries stay at 24 per lane -- the kernel sits at the 256-register limit of two waves per SIMD); each slice is
    // one online-softmax step.
    constexpr int NH = (QF >= 2 || GM_ATTN_NH2) ? 2 : 1, KFH = KF / NH;
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
      // ---- S^T = K Q^T: one K fragment read feeds QF MFMAs ------------------------------------------------------------------
      f32x4_t sacc[QF][KFH];
#pragma unroll
      for (int f = 0; f < QF; ++f)
#pragma unroll
        for (int kf = 0; kf < KFH; ++kf) sacc[f][kf] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      // K fragments are read PD ahead of the MFMAs that consume them (ISA of the straightforward loop: wait - MFMA - read, one read in flight:
      // a wave issued one MFMA per LDS round trip and two waves per SIMD reached 31 % of the MFMA rate with 90 registers to spare)
      {
        constexpr int NKQ = STEPS * KFH, PDW = (QF == 1 && MINW <= 2) ? GM_ATTN_PD : (QF == 1 ? 4 : 2), PD = NKQ < PDW ? NKQ : PDW;  // as deep as the registers allow
        uint4 kq[PD];
        auto kread = [&](int i) __attribute__((always_inline)) {
          const int s = i / KFH, kf = i % KFH;
          return *reinterpret_cast<const uint4*>(buf + kaddr[s % KA] + (s / KA) * (KA * 64) + (hf * KFH + kf) * 16 * KROWB);
        };
#pragma unroll
        for (int i = 0; i < PD; ++i) kq[i] = kread(i);
        __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
#pragma unroll
        for (int i = 0; i < NKQ; ++i) {
          const int s = i / KFH, kf = i % KFH;
#pragma unroll
          for (int f = 0; f < QF; ++f)  // (the first k-step takes a literal zero accumulator: no register clears per tile)
            sacc[f][kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kq[i % PD]), __builtin_bit_cast(bf16x8_t, qf[f][s]),
                                                                  s == 0 ? (f32x4_t){0.f, 0.f, 0.f, 0.f} : sacc[f][kf], 0, 0, 0);
          if (i + PD < NKQ && !(GM_ATTN_ABLATE & 8)) kq[i % PD] = kread(i + PD);
          __builtin_amdgcn_sched_group_barrier(0x008, QF, 0);
          if (i + PD < NKQ && !(GM_ATTN_ABLATE & 8)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      // ---- online softmax: this lane's queries, keys key0 + (hf * KFH + kf) * 16 + qg * 4 + r ----------------------------------
      // (round 5) scores, running maximum and sums live in the base-2 domain: x = s * (scale log2 e), p = 2^(x - m) -- one multiply per score and a
      // bare v_exp_f32 (e^y costs the multiply by log2 e again); the key-bound select runs on the last tile only (wave-uniform branch)
      uint4 pf[QF][KFH / 2];
      float alpha[QF];
      bool moved = false;
      const bool full_tile = key0 + KT <= p.Lk;
#pragma unroll
      for (int f = 0; f < QF; ++f) {
        float tmax = -INFINITY;
        if (full_tile) {
#pragma unroll
          for (int kf = 0; kf < KFH; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float sv = sacc[f][kf][r] * scale2;
              sacc[f][kf][r] = sv;
              tmax = fmaxf(tmax, sv);
            }
        } else {
#pragma unroll
          for (int kf = 0; kf < KFH; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = key0 + (hf * KFH + kf) * 16 + qg * 4 + r;
              const float sv = key < p.Lk ? sacc[f][kf][r] * scale2 : -INFINITY;
              sacc[f][kf][r] = sv;
              tmax = fmaxf(tmax, sv);
            }
        }
        if (!(GM_ATTN_ABLATE & 1)) tmax = attn_quad_max(tmax);
        const float m_new = (GM_ATTN_ABLATE & 1) ? 0.f : fmaxf(m_run[f], tmax);
        alpha[f] = __builtin_amdgcn_exp2f(m_run[f] - m_new);
        moved |= m_new != m_run[f];
        float psum = 0.f;
#pragma unroll
        for (int kf = 0; kf < KFH; ++kf)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = (GM_ATTN_ABLATE & 1) ? sacc[f][kf][r] : __builtin_amdgcn_exp2f(sacc[f][kf][r] - m_new);
            sacc[f][kf][r] = pv;
            psum += pv;
          }
        l_run[f] = l_run[f] * alpha[f] + psum;
        m_run[f] = m_new;
        // P from the S^T accumulators, key positions as packed by vt_pack_kernel
#pragma unroll
        for (int s = 0; s < KFH / 2; ++s)
          pf[f][s] = make_uint4(pack_bf16x2(sacc[f][2 * s][0], sacc[f][2 * s][1]), pack_bf16x2(sacc[f][2 * s][2], sacc[f][2 * s][3]),
                                pack_bf16x2(sacc[f][2 * s + 1][0], sacc[f][2 * s + 1][1]), pack_bf16x2(sacc[f][2 * s + 1][2], sacc[f][2 * s + 1][3]));
      }
      // rescale O only when some lane's running maximum moved (alpha == 1 exactly otherwise: skipping the multiply changes nothing);
      // after the first tiles that is rare, and the 16 x QF x 4 multiplies per step are the largest VALU block of the loop
      if (!(GM_ATTN_ABLATE & 32) && __builtin_amdgcn_ballot_w64(moved) != 0) {
#pragma unroll
        for (int f = 0; f < QF; ++f)
#pragma unroll
          for (int d = 0; d < DF; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[f][d][r] *= alpha[f];
      }

      // ---- O^T += V^T P^T: one V^T fragment read feeds QF MFMAs --------------------------------------------------------------
      {
        constexpr int NVQ = DF * (KFH / 2), PDW = (QF == 1 && MINW <= 2) ? GM_ATTN_PD : (QF == 1 ? 4 : 2), PD = NVQ < PDW ? NVQ : PDW;
        uint4 vq[PD];
        auto vread = [&](int i) __attribute__((always_inline)) {
          const int d = i / (KFH / 2), s = i % (KFH / 2);
          return *reinterpret_cast<const uint4*>(buf + vaddr[hf * (KFH / 2) + s] + d * 16 * 128);
        };
#pragma unroll
        for (int i = 0; i < PD; ++i) vq[i] = vread(i);
        __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
#pragma unroll
        for (int i = 0; i < NVQ; ++i) {
          const int d = i / (KFH / 2), s = i % (KFH / 2);
#pragma unroll
          for (int f = 0; f < QF; ++f)
            oacc[f][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vq[i % PD]), __builtin_bit_cast(bf16x8_t, pf[f][s]),
                                                                 oacc[f][d], 0, 0, 0);
          if (i + PD < NVQ && !(GM_ATTN_ABLATE & 16)) vq[i % PD] = vread(i + PD);
          __builtin_amdgcn_sched_group_barrier(0x008, QF, 0);
          if (i + PD < NVQ && !(GM_ATTN_ABLATE & 16)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
    }
    // the next tile has landed (this wave's pieces) and this wave is done reading the current one
    if (!(GM_ATTN_ABLATE & 4)) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }

  // ---- finish: 1/l, residual, store (or, split-KV: the slice's un-normalised state for attn_combine_kernel) -----------------------
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    const float l_tot = attn_quad_sum(l_run[f]);
    const float inv = 1.0f / l_tot;
    if (!q_ok[f]) continue;
    const float m_nat = m_run[f] * 0.6931471805599453f;  // the running maximum in natural-log units (what the merge kernel and the LSE use)
    if (p.lse && !part && qg == 0) p.lse[(long long)bh * p.Lq + my_q[f]] = m_nat + __logf(l_tot);
    if (part) {  // [slice][b*H + h][query][DH + 2]: O (fp32, relative to the slice maximum), maximum, sum
      float* prow = part + (((long long)blockIdx.z * gridDim.y + bh) * p.Lq + my_q[f]) * (DH + 4);
#pragma unroll
      for (int d = 0; d < DF; ++d)
        *reinterpret_cast<float4*>(prow + d * 16 + qg * 4) = make_float4(oacc[f][d][0], oacc[f][d][1], oacc[f][d][2], oacc[f][d][3]);
      if (qg == 0) { prow[DH] = m_nat; prow[DH + 1] = l_tot; }
      continue;
    }
    bf16_raw* orow = reinterpret_cast<bf16_raw*>(p.o) + ((long long)b * p.Lq + my_q[f]) * p.o_ld + h * DH;
    const bf16_raw* rrow = p.res ? reinterpret_cast<const bf16_raw*>(p.res) + ((long long)b * p.Lq + my_q[f]) * p.res_ld + h * DH : nullptr;
#pragma unroll
    for (int d = 0; d < DF; ++d) {
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = oacc[f][d][r] * inv;
      const int c = d * 16 + qg * 4;
      if (rrow) {
        const uint2 rv = *reinterpret_cast<const uint2*>(rrow + c);
        o[0] += __uint_as_float(rv.x << 16); o[1] += __uint_as_float(rv.x & 0xffff0000u);
        o[2] += __uint_as_float(rv.y << 16); o[3] += __uint_as_float(rv.y & 0xffff0000u);
      }
      *reinterpret_cast<uint2*>(orow + c) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
    }
  }
}

// ---- split-KV merge: out[q] = sum_s e^{m_s - M} O_s / sum_s e^{m_s - M} l_s (+ residual); one thread per (query, 8 channels) ----------
// grid (blocks per sample-head, B * H), grid-stride over the (query, channel vector) items of one (sample, head).  Round 4: optional per-channel
// (sum, sum of squares) partials of the STORED output for the GroupNorm that follows the attention block (GmAttnDesc.stats, single head): the
// block's stride is a multiple of the channel vectors per row, so a thread keeps its 8 channels over the walk; threads of a channel vector are
// added in thread order in fp64 and every (block, channel) entry is stored once -- deterministic, and one gn_stats launch less per block.
#define ATTN_COMBINE_MAX_BLOCKS 256
// NS: the slice count as a compile-time constant (2 / 4 / 8; 0 = run-time): with a run-time bound hipcc walks the slices one dependent load at a time (two loops: the maxima,
// then the weighted sums); unrolled, every slice's loads are in flight before the first wait (the same finding as kv_merge_one, small_ops.hip).  Same arithmetic, same order.
template <int NS>
__global__ __launch_bounds__(256) void attn_combine_kernel(const GmAttnDesc p, const float* __restrict__ part, int nsplit_rt) {
  const int nsplit = NS ? NS : nsplit_rt;
  __shared__ float red[256][17];
  const int dv = p.dh / 8;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const long long items = (long long)p.Lq * dv;
  const long long stride = (long long)p.B * p.H * p.Lq * (p.dh + 4);
  float ss[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % dv) * 8;
    const int q = (int)(i / dv);
    const long long qi = (long long)bh * p.Lq + q;
    const float* row = part + qi * (p.dh + 4);
    float M = -INFINITY;
    float L = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (NS > 0) {
      float ms[NS], ls[NS];
      float4 av[NS], bv[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {  // every slice's state and output vectors requested before the first use
        const float* r = row + s * stride;
        ms[s] = r[p.dh]; ls[s] = r[p.dh + 1];
        av[s] = *reinterpret_cast<const float4*>(r + c); bv[s] = *reinterpret_cast<const float4*>(r + c + 4);
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) M = fmaxf(M, ms[s]);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const float w = __expf(ms[s] - M);             // an empty slice has maximum -inf: weight 0
        L += w * ls[s];
        o[0] += w * av[s].x; o[1] += w * av[s].y; o[2] += w * av[s].z; o[3] += w * av[s].w;
        o[4] += w * bv[s].x; o[5] += w * bv[s].y; o[6] += w * bv[s].z; o[7] += w * bv[s].w;
      }
    } else {
      for (int s = 0; s < nsplit; ++s) M = fmaxf(M, row[s * stride + p.dh]);
      for (int s = 0; s < nsplit; ++s) {
        const float* r = row + s * stride;
        const float w = __expf(r[p.dh] - M);             // an empty slice has maximum -inf: weight 0
        L += w * r[p.dh + 1];
        const float4 a = *reinterpret_cast<const float4*>(r + c), b2 = *reinterpret_cast<const float4*>(r + c + 4);
        o[0] += w * a.x; o[1] += w * a.y; o[2] += w * a.z; o[3] += w * a.w;
        o[4] += w * b2.x; o[5] += w * b2.y; o[6] += w * b2.z; o[7] += w * b2.w;
      }
    }
    const float inv = 1.0f / L;
    if (p.lse && c == 0) p.lse[qi] = M + __logf(L);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] *= inv;
    if (p.res) {
      const uint4 rv = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_raw*>(p.res) + ((long long)b * p.Lq + q) * p.res_ld + h * p.dh + c);
      const uint32_t w4[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) { o[2 * k] += __uint_as_float(w4[k] << 16); o[2 * k + 1] += __uint_as_float(w4[k] & 0xffff0000u); }
    }
    const uint4 raw = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_raw*>(p.o) + ((long long)b * p.Lq + q) * p.o_ld + h * p.dh + c) = raw;
    if (p.stats) {  // statistics of the values as stored (rounded to bf16), like a separate pass over the tensor would see them
      const uint32_t w4[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float lo = __uint_as_float(w4[k] << 16), hi = __uint_as_float(w4[k] & 0xffff0000u);
        ss[2 * k] += lo; sq[2 * k] += lo * lo; ss[2 * k + 1] += hi; sq[2 * k + 1] += hi * hi;
      }
    }
  }
  if (!p.stats) return;  // (uniform)
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[threadIdx.x][k] = ss[k]; red[threadIdx.x][8 + k] = sq[k]; }
  __syncthreads();
  const int ch = threadIdx.x;  // single head (host-checked): channel ch = channel vector ch / 8, element ch % 8
  if (ch < p.dh) {
    const int cv = ch >> 3, k = ch & 7;
    double a = 0.0, a2 = 0.0;
    for (int t = cv; t < 256; t += dv) { a += (double)red[t][k]; a2 += (double)red[t][8 + k]; }  // (thread t walks channel vector t % dv: 256 % dv == 0)
    *reinterpret_cast<double2*>(p.stats + (((long long)blockIdx.x * p.B + b) * p.dh + ch) * 2) = make_double2(a, a2);
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
void gm_attn_pack_transposed(const bf16_raw* rows, long long ld, bf16_raw* image, int B, int H, int L, int L_pad, int dh, hipStream_t st) {
  vt_pack_kernel<<<dim3(L_pad / 64, dh / 64, B * H), 256, 0, st>>>(rows, ld, image, H, L, L_pad, dh);
}
void gm_attn_pack_transposed3(const bf16_raw* const rows[3], const long long ld[3], bf16_raw* const image[3], const int L[3], const int L_pad[3], int B, int H, int dh,
                              hipStream_t st) {
  GmPackSets ps;
  int lmax = 0;
  for (int i = 0; i < 3; ++i) {
    ps.s[i] = GmPackSet{rows[i], ld[i], image[i], L[i], L_pad[i]};
    lmax = L_pad[i] > lmax ? L_pad[i] : lmax;
  }
  vt_pack_sets_kernel<<<dim3(lmax / 64, dh / 64, 3 * B * H), 256, 0, st>>>(ps, B * H, H, dh);
}

static int gm_attn_dma_force_qf = 0, gm_attn_dma_force_split = 0;  // 0 = choose by problem size (tests / benchmarks force a variant)
extern "C" void gm_attention_dma_set_variant(int qf, int nsplit) {
  gm_attn_dma_force_qf = (qf == 1 || qf == 2) ? qf : 0;
  gm_attn_dma_force_split = (nsplit >= 1 && nsplit <= 8) ? nsplit : 0;
}

static bool attn_dma_eligible(const GmAttnDesc& d) {
  auto al = [](const void* p, int a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
  return d.dtype == GM_BF16 && !d.causal && d.k_bs == 0 && d.v_bs == 0 && (d.dh == 64 || d.dh == 128 || d.dh == 256) && d.Lq >= 128 && d.Lk >= 128 &&
         (long long)d.dh * (((long long)d.Lk + 63) / 64 * 64) * 2 < (1LL << 31) &&  // 32-bit byte offsets inside one head's V^T image
         d.k_ld < (1LL << 24) &&                                                    // ... and inside one K tile (64 rows x k_ld x 2 bytes)
         d.q_ld % 8 == 0 && d.k_ld % 8 == 0 && d.v_ld % 8 == 0 && al(d.q, 16) && al(d.k, 16) && al(d.v, 16) &&
         d.o_ld % 8 == 0 && al(d.o, 16) && (!d.res || (d.res_ld % 8 == 0 && al(d.res, 16)));
}

// Queries per wave (16 * qf) and key slices.  Measured on MI355X (profiles/r02_attention_variants.txt): 32 queries per wave halve the LDS
// operand reads per MFMA but need 256 registers at head dim 256 (spills inside the tile loop) and run 10-15 % SLOWER than 16 per wave at
// every shape, so the automatic choice stays at qf = 1; key slices (powers of two, at least 4 key tiles each) are added until every CU has
// a work-group: 2.7x at 4096 tokens (32 -> 256 work-groups), nothing at 32768.  (Four waves x 64 queries with a SIMD's 512 registers
// each -- 0.25 reads per MFMA -- does not survive hipcc: 432 spilled registers.)
static unsigned attn_combine_blocks(const GmAttnDesc& d) {  // blocks per (sample, head) of the merge kernel = statistic slots per sample
  const long long items = (long long)d.Lq * (d.dh / 8);
  long long g = (items + 255) / 256;
  return (unsigned)(g > ATTN_COMBINE_MAX_BLOCKS ? ATTN_COMBINE_MAX_BLOCKS : (g < 1 ? 1 : g));
}
static void attn_dma_plan(const GmAttnDesc& d, int* qf, int* nsplit) {
  const long long tiles = ((long long)d.Lk + 63) / 64;
  const int f = gm_attn_dma_force_qf ? gm_attn_dma_force_qf : 1;
  const int qpb = 128 * f;
  const long long nq = (long long)d.B * d.H * ((d.Lq + qpb - 1) / qpb);
  int sp = 1;
  while (nq * sp < 256 && sp < 8 && tiles / (2 * sp) >= 4) sp *= 2;
  if (gm_attn_dma_force_split) sp = gm_attn_dma_force_split;
  *qf = f; *nsplit = sp;
}

extern "C" long long gm_attention_workspace_bytes(const GmAttnDesc* d) {
  if (!d || !attn_dma_eligible(*d)) return 0;
  const long long lk_pad = ((long long)d->Lk + 63) / 64 * 64;
  int qf, sp;
  attn_dma_plan(*d, &qf, &sp);
  const long long vt_bytes = (long long)d->B * d->H * d->dh * lk_pad * 2;
  const long long part_bytes = sp > 1 ? (long long)sp * d->B * d->H * d->Lq * (d->dh + 4) * 4 : 0;
  return ((vt_bytes + 255) & ~255LL) + part_bytes;
}

template <int DH, int QF, int NW = 8>
static void launch_attn_dma(const GmAttnDesc& d, bf16_raw* vt, int lk_pad, float* part, int nsplit, hipStream_t st) {
  static bool attr_set = false;
  // waves per SIMD the register allocation must leave room for: the LDS footprint (2 x 512 x DH bytes) admits 160 KiB / that many
  // 8-wave work-groups per CU
  constexpr int MINW = (DH == 256 || QF == 2) ? 2 : 4;
  auto kern = attn_dma_kernel<DH, QF, NW, MINW>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) (void)hipGetLastError();
    attr_set = true;
  }
  constexpr int QPB = NW * 16 * QF;  // queries per work-group
  dim3 grid((d.Lq + QPB - 1) / QPB, d.B * d.H, nsplit);
  kern<<<grid, 64 * NW, (size_t)2 * (64 * DH * 2 + DH * 128), st>>>(d, vt, lk_pad, nsplit > 1 ? part : nullptr);
}

// returns 1 if the LDS-DMA path was launched, 0 if the caller should use the register-staged kernel
extern "C" int gm_attention_dma_try(const GmAttnDesc* dp, void* stream) {
  const GmAttnDesc& d = *dp;
  if (!attn_dma_eligible(d) || !d.workspace || d.workspace_bytes < gm_attention_workspace_bytes(dp)) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int lk_pad = (d.Lk + 63) / 64 * 64;
  int qf, sp;
  attn_dma_plan(d, &qf, &sp);
  bf16_raw* vt = reinterpret_cast<bf16_raw*>(d.workspace);
  const long long vt_bytes = (long long)d.B * d.H * d.dh * lk_pad * 2;
  float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(d.workspace) + ((vt_bytes + 255) & ~255LL));
  dim3 pg(lk_pad / 64, d.dh / 64, d.B * d.H);
  if (!d.vt_packed)  // (else: the q | k | v projection stored the image, gm_linear_rows_affine_vt)
    vt_pack_kernel<<<pg, 256, 0, st>>>(reinterpret_cast<const bf16_raw*>(d.v), d.v_ld, vt, d.H, d.Lk, lk_pad, d.dh);
  if (d.dh == 64) { if (qf == 2) launch_attn_dma<64, 2>(d, vt, lk_pad, part, sp, st); else launch_attn_dma<64, 1>(d, vt, lk_pad, part, sp, st); }
  else if (d.dh == 128) { if (qf == 2) launch_attn_dma<128, 2>(d, vt, lk_pad, part, sp, st); else launch_attn_dma<128, 1>(d, vt, lk_pad, part, sp, st); }
  else { if (qf == 2) launch_attn_dma<256, 2>(d, vt, lk_pad, part, sp, st); else launch_attn_dma<256, 1>(d, vt, lk_pad, part, sp, st); }
  if (sp > 1) {
    const dim3 cg(attn_combine_blocks(d), (unsigned)(d.B * d.H));
    if (sp == 2) attn_combine_kernel<2><<<cg, 256, 0, st>>>(d, part, sp);
    else if (sp == 4) attn_combine_kernel<4><<<cg, 256, 0, st>>>(d, part, sp);
    else if (sp == 8) attn_combine_kernel<8><<<cg, 256, 0, st>>>(d, part, sp);
    else attn_combine_kernel<0><<<cg, 256, 0, st>>>(d, part, sp);
  }
  return 1;
}

// Per-channel statistic partials S the LDS-DMA path writes into GmAttnDesc.stats ([S][B][H * dh][2] fp64) for this geometry, 0 = none (the
// register-staged kernel, an unsplit launch, several heads): the caller leaves `stats` NULL then and runs gm_gn_channel_stats when a norm follows.
extern "C" long long gm_attention_stats_slots(const GmAttnDesc* d) {
  if (!d || !attn_dma_eligible(*d) || d->H != 1) return 0;
  int qf, sp;
  attn_dma_plan(*d, &qf, &sp);
  return sp > 1 ? (long long)attn_combine_blocks(*d) : 0;
}
