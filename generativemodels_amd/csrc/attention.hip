// Single-pass (flash-style) softmax attention on the CDNA4 matrix cores; the L x L score matrix never exists in HBM.
//
// Reference semantics (networks/nets/diffusion_model_unet.py:407-415 AttentionBlock, :143-153 CrossAttention,
// networks/nets/autoencoderkl.py:261-269):  O = softmax(scale * Q K^T) V  per (batch, head), optionally + residual.
// Scores and the softmax state are fp32 for every storage dtype (a superset of the reference's `upcast_attention`).
//
// Everything is kept in the *transposed* orientation so that a lane owns ONE query for the whole kernel:
//   S^T[key][query] = K Q^T   (A = K tile from LDS, B = Q fragments held in registers)
//   O^T[d][query]   = V^T P^T (A = V tile from LDS, B = P^T straight from the S^T accumulators -- no cross-lane moves)
// The C/D layout of the 16x16 MFMAs (col = lane&15, row = 4*(lane>>4)+reg) then gives every lane 4 keys of its own query
// after QK^T and 4 consecutive head channels of its own query after PV: running max / sum / rescale are lane-local, the
// row max needs two xor-shuffles, and the output is stored as 8/16-byte NDHWC vectors.
// bf16: v_mfma_f32_16x16x32_bf16, V is transposed (8x8 register blocks) while it is staged into LDS so the PV A-operand is
// one ds_read_b128 per MFMA.  fp32: v_mfma_f32_16x16x4_f32 (exact fp32 products), V stays row-major.
#include "gm_common.h"

#include "attn_common.h"

template <typename T> struct AttnTraits;
template <> struct AttnTraits<bf16_raw> { static constexpr int VECW = 8; static constexpr int KT = 64; };
template <> struct AttnTraits<float> { static constexpr int VECW = 4; static constexpr int KT = 32; };

__device__ __forceinline__ void unpack16(const uint4& v, float* o, bf16_raw) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}

// load 16 bytes of row `row` at element offset c0 (zero fill outside [0, climit) / invalid rows)
template <typename T>
__device__ __forceinline__ uint4 load_row16(const T* base, long long ld, long long row, bool row_ok, int c0, int climit, bool vec_ok) {
  constexpr int VECW = 16 / sizeof(T);
  uint4 r = make_uint4(0, 0, 0, 0);
  if (!row_ok || c0 >= climit) return r;
  const T* p = base + row * ld + c0;
  if (vec_ok) return *reinterpret_cast<const uint4*>(p);
  T tmp[VECW];
#pragma unroll
  for (int i = 0; i < VECW; ++i) tmp[i] = (c0 + i < climit) ? p[i] : (T)0;
  return *reinterpret_cast<uint4*>(tmp);
}

// G = 2 / 4 (round 6): G groups of four waves per work-group share the 64 queries and deal the key tiles between them (group g takes tiles g, g + G, ...), each with
// its own K / V buffers; their softmax states meet in LDS at the end (merged into group 0 in group order).  Eight waves per CU instead of four: one group's softmax and staging run under the other's
// MFMAs (a 256-work-group grid -- 16 x 1024 queries -- has one work-group per CU), without the second launch a split over work-groups needs.
template <typename T, int DH, int G>
__global__ __launch_bounds__(256 * G) void attn_kernel(const GmAttnDesc p) {
  constexpr int VECW = AttnTraits<T>::VECW;
  constexpr int KT = AttnTraits<T>::KT;
  constexpr int KF = KT / 16;                      // key fragments per tile
  constexpr int ROWB_K = DH * (int)sizeof(T) + 16; // K tile row pitch (bytes)
  constexpr int STEPS = DH * (int)sizeof(T) / 64;  // 64-byte k-steps over the head dim
  constexpr int DF = DH / 16;                      // output channel fragments
  constexpr bool IS_BF16 = sizeof(T) == 2;
  constexpr int ROWB_V = IS_BF16 ? (KT * 2 + 16) : (DH * 4 + 16);  // bf16: V^T rows of KT keys; fp32: V rows of DH

  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int GROUP_BYTES = KT * ROWB_K + (IS_BF16 ? DH * ROWB_V : KT * ROWB_V);
  const int grp = G == 1 ? 0 : (int)(threadIdx.x >> 8);  // wave group (wave-uniform)
  char* ldsK = smem + (size_t)grp * GROUP_BYTES;  // [KT][ROWB_K]
  char* ldsV = ldsK + (size_t)KT * ROWB_K;        // bf16: [DH][ROWB_V] (transposed), fp32: [KT][ROWB_V]

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;  // (within the group)
  const int l15 = lane & 15, qg = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const int my_q = q0 + l15;
  const bool q_ok = my_q < p.Lq;

  const T* Qb = reinterpret_cast<const T*>(p.q) + (long long)b * p.Lq * p.q_ld + h * p.dh;
  const T* Kb = reinterpret_cast<const T*>(p.k) + (long long)b * (p.k_bs ? p.k_bs : p.Lk * p.k_ld) + h * p.dh;
  const T* Vb = reinterpret_cast<const T*>(p.v) + (long long)b * (p.v_bs ? p.v_bs : p.Lk * p.v_ld) + h * p.dh;
  const bool qvec = (p.dh % VECW == 0) && (p.q_ld % VECW == 0) && ((reinterpret_cast<uintptr_t>(Qb) & 15) == 0);
  const bool kvec = (p.dh % VECW == 0) && (p.k_ld % VECW == 0) && ((reinterpret_cast<uintptr_t>(Kb) & 15) == 0);
  const bool vvec = (p.dh % VECW == 0) && (p.v_ld % VECW == 0) && ((reinterpret_cast<uintptr_t>(Vb) & 15) == 0);

  // Q fragments (B operand of S^T = K Q^T): lane (query l15, slot qg) holds channels step*BK + qg*VECW .. +VECW-1
  uint4 qf[STEPS];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) qf[s] = load_row16<T>(Qb, p.q_ld, my_q, q_ok, (s * 4 + qg) * VECW, p.dh, qvec);

  f32x4_t oacc[DF];
#pragma unroll
  for (int d = 0; d < DF; ++d) oacc[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  // causal: this lane's query sees keys <= my_q + (Lk - Lq); the work-group stops after the tile holding its last visible key
  const int kmax = p.causal ? my_q + (p.Lk - p.Lq) : p.Lk;
  int ntiles = (p.Lk + KT - 1) / KT;
  if (p.causal) {
    const int last = min(p.Lk - 1, (int)blockIdx.x * 64 + 63 + (p.Lk - p.Lq));
    ntiles = max(1, min(ntiles, last / KT + 1));
  }
  // ---- K / V staging, software-pipelined (round 6): tile t + 1 is REQUESTED into registers right after tile t has been committed to LDS, so its global-memory
  // latency runs under tile t's MFMAs and softmax instead of in front of them (the loop used to be load -> wait -> barrier -> compute per 32 / 64 keys: 3.4 us per
  // tile at 16 x 1024 x 1024 x 64 fp32, 110 us per attention block of BASELINE configs[0]).  Per thread: (KT * row chunks) / 256 16-byte vectors of K and as many
  // of V (bf16: one 8 x 8 block of V, transposed at the commit).
  constexpr int CHK = DH * (int)sizeof(T) / 16;            // 16-byte chunks per K row (fp32: and per V row)
  constexpr int NKI = (KT * CHK + 255) / 256;              // K items per thread
  constexpr int PB = KT / 8, DB = DH / 8;                  // bf16 V: 8-key x 8-channel blocks
  constexpr int NVI = IS_BF16 ? (PB * DB + 255) / 256 : NKI;
  uint4 kreg[NKI];
  uint4 vreg[NVI][IS_BF16 ? 8 : 1];
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    const int key0 = tile * KT;
#pragma unroll
    for (int it = 0; it < NKI; ++it) {
      const int item = tid + it * 256;
      const int row = item / CHK, ch = item % CHK;
      kreg[it] = (KT * CHK % 256 == 0 || item < KT * CHK) ? load_row16<T>(Kb, p.k_ld, key0 + row, key0 + row < p.Lk, ch * VECW, p.dh, kvec) : make_uint4(0, 0, 0, 0);
    }
    if constexpr (IS_BF16) {
      // 8 keys x 8 channels per item, transposed in registers at the commit.  Key order inside a V^T row: position
      // pos = s*32 + qq*8 + half*4 + r  <->  key = (2s+half)*16 + qq*4 + r, i.e. exactly the 8 keys lane-group qq
      // feeds into k-step s of the PV MFMA are contiguous (one ds_read_b128).
#pragma unroll
      for (int it = 0; it < NVI; ++it) {
        const int item = tid + it * 256;
        const int pb = item % PB, db = item / PB;
        const int sq = pb >> 2, qq = pb & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int key = key0 + (2 * sq + (j >> 2)) * 16 + qq * 4 + (j & 3);
          vreg[it][j] = item < PB * DB ? load_row16<T>(Vb, p.v_ld, key, key < p.Lk, db * 8, p.dh, vvec) : make_uint4(0, 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < NVI; ++it) {
        const int item = tid + it * 256;
        const int row = item / CHK, ch = item % CHK;
        vreg[it][0] = (KT * CHK % 256 == 0 || item < KT * CHK) ? load_row16<T>(Vb, p.v_ld, key0 + row, key0 + row < p.Lk, ch * VECW, p.dh, vvec) : make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NKI; ++it) {
      const int item = tid + it * 256;
      if (KT * CHK % 256 == 0 || item < KT * CHK) *reinterpret_cast<uint4*>(ldsK + (size_t)(item / CHK) * ROWB_K + (item % CHK) * 16) = kreg[it];
    }
    if constexpr (IS_BF16) {
#pragma unroll
      for (int it = 0; it < NVI; ++it) {
        const int item = tid + it * 256;
        if (item < PB * DB) {
          const int pb = item % PB, db = item / PB;
#pragma unroll
          for (int d = 0; d < 8; ++d) {
            uint32_t w[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const uint32_t a = reinterpret_cast<const uint32_t*>(&vreg[it][2 * c])[d >> 1];
              const uint32_t bq = reinterpret_cast<const uint32_t*>(&vreg[it][2 * c + 1])[d >> 1];
              w[c] = (d & 1) ? ((a >> 16) | (bq & 0xffff0000u)) : ((a & 0xffffu) | (bq << 16));
            }
            *reinterpret_cast<uint4*>(ldsV + (size_t)(db * 8 + d) * ROWB_V + pb * 16) = make_uint4(w[0], w[1], w[2], w[3]);
          }
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < NVI; ++it) {
        const int item = tid + it * 256;
        if (KT * CHK % 256 == 0 || item < KT * CHK) *reinterpret_cast<uint4*>(ldsV + (size_t)(item / CHK) * ROWB_V + (item % CHK) * 16) = vreg[it][0];
      }
    }
  };
  if (grp < ntiles) fetch(grp);
  for (int round = 0; round * G < ntiles; ++round) {
    const int tile = round * G + grp;  // this group's tile of the round; the last round may have none for group 1 (it still meets the barriers)
    const bool live = tile < ntiles;   // (group-uniform)
    const int key0 = tile * KT;
    __syncthreads();  // the previous tile's LDS reads are complete
    if (live) commit();
    __syncthreads();
    if (!live) continue;
    if (tile + G < ntiles) fetch(tile + G);  // (group-uniform) in flight under this tile's arithmetic

    // ---- S^T = K Q^T ------------------------------------------------------------------------------------------------
    f32x4_t sacc[KF];
#pragma unroll
    for (int kf = 0; kf < KF; ++kf) sacc[kf] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
#pragma unroll
      for (int kf = 0; kf < KF; ++kf) {
        const uint4 kfrag = *reinterpret_cast<const uint4*>(ldsK + (size_t)(kf * 16 + l15) * ROWB_K + s * 64 + qg * 16);
        if constexpr (IS_BF16) {
          sacc[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kfrag),
                                                             __builtin_bit_cast(bf16x8_t, qf[s]), sacc[kf], 0, 0, 0);
        } else {
          sacc[kf] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(kfrag.x), __uint_as_float(qf[s].x), sacc[kf], 0, 0, 0);
          sacc[kf] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(kfrag.y), __uint_as_float(qf[s].y), sacc[kf], 0, 0, 0);
          sacc[kf] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(kfrag.z), __uint_as_float(qf[s].z), sacc[kf], 0, 0, 0);
          sacc[kf] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(kfrag.w), __uint_as_float(qf[s].w), sacc[kf], 0, 0, 0);
        }
      }
    }
    // ---- online softmax: this lane's query, keys key0 + kf*16 + qg*4 + r ---------------------------------------------
    float tmax = -INFINITY;
#pragma unroll
    for (int kf = 0; kf < KF; ++kf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + kf * 16 + qg * 4 + r;
        const float sv = (key < p.Lk) & (key <= kmax) ? sacc[kf][r] * p.scale : -INFINITY;
        sacc[kf][r] = sv;
        tmax = fmaxf(tmax, sv);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    // (a group whose first tile is entirely masked for this query -- causal, G = 2 -- has no maximum yet: exponents against 0 then, every weight e^(-inf) = 0)
    const float m_ref = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = IS_BF16 ? __expf(m_run - m_ref) : expf(m_run - m_ref);
    float psum = 0.f;
#pragma unroll
    for (int kf = 0; kf < KF; ++kf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = IS_BF16 ? __expf(sacc[kf][r] - m_ref) : expf(sacc[kf][r] - m_ref);
        sacc[kf][r] = pv;
        psum += pv;
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < DF; ++d)
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[d][r] *= alpha;

    // ---- O^T += V^T P^T ---------------------------------------------------------------------------------------------
    if constexpr (IS_BF16) {
      uint4 pf[KF / 2];
#pragma unroll
      for (int s = 0; s < KF / 2; ++s) {
        uint32_t w[4];
        w[0] = (uint32_t)f32_to_bf16(sacc[2 * s][0]) | ((uint32_t)f32_to_bf16(sacc[2 * s][1]) << 16);
        w[1] = (uint32_t)f32_to_bf16(sacc[2 * s][2]) | ((uint32_t)f32_to_bf16(sacc[2 * s][3]) << 16);
        w[2] = (uint32_t)f32_to_bf16(sacc[2 * s + 1][0]) | ((uint32_t)f32_to_bf16(sacc[2 * s + 1][1]) << 16);
        w[3] = (uint32_t)f32_to_bf16(sacc[2 * s + 1][2]) | ((uint32_t)f32_to_bf16(sacc[2 * s + 1][3]) << 16);
        pf[s] = make_uint4(w[0], w[1], w[2], w[3]);
      }
#pragma unroll
      for (int d = 0; d < DF; ++d)
#pragma unroll
        for (int s = 0; s < KF / 2; ++s) {
          const uint4 vfrag = *reinterpret_cast<const uint4*>(ldsV + (size_t)(d * 16 + l15) * ROWB_V + s * 64 + qg * 16);
          oacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vfrag),
                                                            __builtin_bit_cast(bf16x8_t, pf[s]), oacc[d], 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int d = 0; d < DF; ++d)
#pragma unroll
        for (int kf = 0; kf < KF; ++kf)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float vv = *reinterpret_cast<const float*>(ldsV + (size_t)(kf * 16 + qg * 4 + i) * ROWB_V + (d * 16 + l15) * 4);
            oacc[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, sacc[kf][i], oacc[d], 0, 0, 0);
          }
    }
  }

  // ---- G = 2: group 1 hands its state (running maximum, this lane's partial sum, un-normalised output) to the same lane of group 0 through LDS ----------------
  if constexpr (G > 1) {
    __syncthreads();  // every tile has been read: the operand buffers are free
    constexpr int XW = 2 + DF * 4;  // floats per lane: running maximum, partial sum, un-normalised output
    float* xch = reinterpret_cast<float*>(smem) + ((size_t)(grp > 0 ? grp - 1 : 0) * 256 + tid) * XW;
    if (grp > 0) {
      xch[0] = m_run; xch[1] = l_run;
#pragma unroll
      for (int d = 0; d < DF; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) xch[2 + d * 4 + r] = oacc[d][r];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g = 1; g < G; ++g) {  // group order: a fixed summation order
      const float* xg = reinterpret_cast<const float*>(smem) + ((size_t)(g - 1) * 256 + tid) * XW;
      const float m1 = xg[0], l1 = xg[1];
      const float M = fmaxf(m_run, m1);  // (group 0 always has a tile: finite)
      const float a0 = IS_BF16 ? __expf(m_run - M) : expf(m_run - M), a1 = IS_BF16 ? __expf(m1 - M) : expf(m1 - M);  // a group without a tile: e^(-inf) = 0
      l_run = l_run * a0 + l1 * a1;
      m_run = M;
#pragma unroll
      for (int d = 0; d < DF; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) oacc[d][r] = oacc[d][r] * a0 + xg[2 + d * 4 + r] * a1;
    }
  }

  // ---- finish: 1/l, residual, store ------------------------------------------------------------------------------------
  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  const float inv = 1.0f / l_tot;
  if (!q_ok) return;
  T* orow = reinterpret_cast<T*>(p.o) + ((long long)b * p.Lq + my_q) * p.o_ld + h * p.dh;
  const T* rrow = p.res ? reinterpret_cast<const T*>(p.res) + ((long long)b * p.Lq + my_q) * p.res_ld + h * p.dh : nullptr;
#pragma unroll
  for (int d = 0; d < DF; ++d) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = d * 16 + qg * 4 + r;
      if (c < p.dh) {
        float v = oacc[d][r] * inv;
        if (rrow) v += ElemIO<T>::ld(rrow + c);
        ElemIO<T>::st(orow + c, v);
      }
    }
  }
}

template <typename T, int DH, int G>
static int launch_attn_g(const GmAttnDesc& d, hipStream_t st) {
  constexpr int KT = AttnTraits<T>::KT;
  constexpr bool IS_BF16 = sizeof(T) == 2;
  constexpr size_t group = (size_t)KT * (DH * sizeof(T) + 16) + (IS_BF16 ? (size_t)DH * (KT * 2 + 16) : (size_t)KT * (DH * 4 + 16));
  constexpr size_t xch = G > 1 ? (size_t)(G - 1) * 256 * (2 + DH / 16 * 4) * 4 : 0;  // the hand-over of the other groups' states overlays the operand buffers
  constexpr size_t smem = G * group > xch ? G * group : xch;
  static_assert(smem <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  auto kern = attn_kernel<T, DH, G>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) (void)hipGetLastError();
    attr_set = true;
  }
  dim3 grid((d.Lq + 63) / 64, d.B * d.H);
  kern<<<grid, 256 * G, smem, st>>>(d);
  return 0;
}

// two wave groups once a work-group has at least four key tiles to deal, four from eight tiles on where 128 registers per wave suffice (fp32, head dim <= 64: BASELINE
// configs[0]'s 16 x 1024-token blocks: 0.970 -> 0.919 ms per fp32 forward); process-wide override for measurements: gm_attention_set_wave_groups
static int gm_attn_wave_groups = 0;  // 0 = by key count
extern "C" void gm_attention_set_wave_groups(int g) { gm_attn_wave_groups = (g == 1 || g == 2 || g == 4) ? g : 0; }
template <typename T, int DH>
static int launch_attn(const GmAttnDesc& d, hipStream_t st) {
  constexpr int KT = AttnTraits<T>::KT;
  const int tiles = (d.Lk + KT - 1) / KT;
  const int g = gm_attn_wave_groups ? gm_attn_wave_groups : (tiles >= 8 ? 4 : (tiles >= 4 ? 2 : 1));  // (four: the fp32 kernel at head dim <= 64 only, see below)
  if constexpr (sizeof(T) == 4 && DH == 256) return launch_attn_g<T, DH, 1>(d, st);  // (two groups = 256 registers per wave: the fp32 kernel at head dim 256 would spill 81)
  else if constexpr (sizeof(T) == 4 && DH <= 64) {  // four groups = 128 registers per wave: the small fp32 head dims only
    if (g == 4) return launch_attn_g<T, DH, 4>(d, st);
    return g == 2 ? launch_attn_g<T, DH, 2>(d, st) : launch_attn_g<T, DH, 1>(d, st);
  } else return g >= 2 ? launch_attn_g<T, DH, 2>(d, st) : launch_attn_g<T, DH, 1>(d, st);
}

template <typename T>
static int dispatch_attn(const GmAttnDesc& d, hipStream_t st) {
  if (d.dh <= 32) return launch_attn<T, 32>(d, st);
  if (d.dh <= 64) return launch_attn<T, 64>(d, st);
  if (d.dh <= 128) return launch_attn<T, 128>(d, st);
  if (d.dh <= 256) return launch_attn<T, 256>(d, st);
  return -1;
}

extern "C" int gm_attention_max_head_dim(void) { return 256; }
extern "C" int gm_attention_dma_try(const GmAttnDesc* dp, void* stream);     // attention_dma.hip
extern "C" int gm_attention_decode_try(const GmAttnDesc* dp, void* stream);  // small_ops.hip

extern "C" int gm_attention_forward(const GmAttnDesc* dp, void* stream) {
  GM_REQUIRE(dp, "null descriptor");
  const GmAttnDesc& d = *dp;
  GM_REQUIRE(d.q && d.k && d.v && d.o, "null tensor pointer");
  GM_REQUIRE(d.B >= 0 && d.H > 0 && d.dh > 0, "bad batch / head geometry");
  GM_REQUIRE(d.Lk > 0, "attention needs at least one key");
  GM_REQUIRE(d.dh <= 256, "head dim > 256 is not supported by the gfx950 attention kernel");
  GM_REQUIRE((long long)d.B * d.H <= 65535, "too many (batch, head) pairs for one launch");
  if (d.B == 0 || d.Lq == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (!d.lse && gm_attention_decode_try(dp, stream)) GM_LAUNCH_CHECK();  // one query per (batch, head): the KV-cache decode kernel (small_ops.hip)
  if (gm_attention_dma_try(dp, stream)) GM_LAUNCH_CHECK();  // bf16, d in {64,128,256}, workspace given: LDS-DMA kernel
  GM_REQUIRE(d.stats == nullptr, "GmAttnDesc.stats is written by the split-KV LDS-DMA path only (see gm_attention_stats_slots)");
  GM_REQUIRE(d.lse == nullptr, "GmAttnDesc.lse is written by the LDS-DMA path only (a workspace of gm_attention_workspace_bytes() > 0 bytes)");
  GM_REQUIRE(!d.vt_packed, "GmAttnDesc.vt_packed needs the LDS-DMA path (a workspace of gm_attention_workspace_bytes() bytes)");
  int rc;
  if (d.dtype == GM_F32) rc = dispatch_attn<float>(d, st);
  else if (d.dtype == GM_BF16) rc = dispatch_attn<bf16_raw>(d, st);
  else GM_FAIL(-2, "unsupported dtype");
  GM_REQUIRE(rc == 0, "dispatch failed");
  GM_LAUNCH_CHECK();
}
