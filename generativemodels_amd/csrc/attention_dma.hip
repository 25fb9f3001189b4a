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

__device__ __attribute__((aligned(64))) unsigned int gm_attn_zero_row[16] = {0};

__device__ __forceinline__ void attn_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// ---- V -> VT[b*H + h][c][pos], pos = blk*32 + qq*8 + half*4 + r  <->  key = blk*32 + half*16 + qq*4 + r; keys >= Lk are zero ----
__global__ __launch_bounds__(256) void vt_pack_kernel(const bf16_raw* __restrict__ v, long long v_ld, bf16_raw* __restrict__ vt,
                                                      int H, int Lk, int Lk_pad, int dh) {
  __shared__ bf16_raw tile[64][64 + 8];
  const int bh = blockIdx.z, b = bh / H, h = bh % H;
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

template <int DH>
__global__ __launch_bounds__(512, 2) void attn_dma_kernel(const GmAttnDesc p, const bf16_raw* __restrict__ vt, int Lk_pad) {
  constexpr int KT = 64, KF = KT / 16, NW = 8;
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
  const int my_q = blockIdx.x * (NW * 16) + wave * 16 + l15;
  const bool q_ok = my_q < p.Lq;

  const bf16_raw* Qb = reinterpret_cast<const bf16_raw*>(p.q) + (long long)b * p.Lq * p.q_ld + h * DH;
  const char* Kb = reinterpret_cast<const char*>(reinterpret_cast<const bf16_raw*>(p.k) + (long long)b * p.Lk * p.k_ld + h * DH);
  const char* Vt = reinterpret_cast<const char*>(vt + (long long)bh * DH * Lk_pad);
  const char* zero = reinterpret_cast<const char*>(gm_attn_zero_row);

  // ---- this lane's DMA sources (tile-independent part) ------------------------------------------------------------------------
  int krow[PPW], kslotb[PPW];
  long long vsrc[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int pc = wave * PPW + j;
    const int row = pc * KRPP + lane / SPR, slot = lane % SPR;
    krow[j] = row;
    kslotb[j] = (slot ^ ((row >> KSH) & (KNB - 1))) * 16;
    const int vrow = pc * 8 + (lane >> 3), vslot = lane & 7;
    vsrc[j] = (long long)vrow * Lk_pad * 2 + ((vslot ^ ((vrow >> 1) & 7)) * 16);
  }
  const long long k_rowb = p.k_ld * 2;
  auto issue_tile = [&](int tile, int buf) __attribute__((always_inline)) {
    const int key0 = tile * KT;
    const unsigned kdst = lds0 + (unsigned)buf * (KBYTES + VBYTES), vdst = kdst + KBYTES;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const bool ok = key0 + krow[j] < p.Lk;
      const char* ks = ok ? Kb + (long long)(key0 + krow[j]) * k_rowb + kslotb[j] : zero + ((lane & 3) << 4);
      attn_dma16(ks, kdst + (unsigned)(wave * PPW + j) * 1024);
      attn_dma16(Vt + vsrc[j] + (long long)key0 * 2, vdst + (unsigned)(wave * PPW + j) * 1024);
    }
  };

  // ---- operand read offsets ---------------------------------------------------------------------------------------------------
  int kaddr[STEPS];  // K fragment (key row l15 of fragment 0, k-step s); fragment kf adds kf*16*KROWB
  const int fk = (l15 >> KSH) & (KNB - 1);
#pragma unroll
  for (int s = 0; s < STEPS; ++s) kaddr[s] = l15 * KROWB + (((s * 4 + qg) ^ fk) << 4);
  int vaddr[2];      // V^T fragment (channel row l15 of fragment 0, key step s2); fragment d adds d*16*128
  const int fv = (l15 >> 1) & 7;
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) vaddr[s2] = KBYTES + l15 * 128 + (((s2 * 4 + qg) ^ fv) << 4);

  // Q fragments (B operand of S^T = K Q^T): lane (query l15, slot qg) holds channels s*32 + qg*8 .. +7
  uint4 qf[STEPS];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const uint4 v = *reinterpret_cast<const uint4*>(Qb + (long long)(q_ok ? my_q : 0) * p.q_ld + s * 32 + qg * 8);
    qf[s] = make_uint4(q_ok ? v.x : 0u, q_ok ? v.y : 0u, q_ok ? v.z : 0u, q_ok ? v.w : 0u);
  }
  f32x4_t oacc[DF];
#pragma unroll
  for (int d = 0; d < DF; ++d) oacc[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const int ntiles = (p.Lk + KT - 1) / KT;
  issue_tile(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int tile = 0; tile < ntiles; ++tile) {
    const int key0 = tile * KT;
    const char* buf = smem + (size_t)(tile & 1) * (KBYTES + VBYTES);
    if (tile + 1 < ntiles) issue_tile(tile + 1, (tile + 1) & 1);  // its buffer was last read two barriers ago

    // ---- S^T = K Q^T ------------------------------------------------------------------------------------------------------
    f32x4_t sacc[KF];
#pragma unroll
    for (int kf = 0; kf < KF; ++kf) sacc[kf] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
      for (int kf = 0; kf < KF; ++kf) {
        const uint4 kfrag = *reinterpret_cast<const uint4*>(buf + kaddr[s] + kf * 16 * KROWB);
        sacc[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kfrag), __builtin_bit_cast(bf16x8_t, qf[s]),
                                                           sacc[kf], 0, 0, 0);
      }
    // ---- online softmax: this lane's query, keys key0 + kf*16 + qg*4 + r ----------------------------------------------------
    float tmax = -INFINITY;
#pragma unroll
    for (int kf = 0; kf < KF; ++kf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + kf * 16 + qg * 4 + r;
        const float sv = key < p.Lk ? sacc[kf][r] * p.scale : -INFINITY;
        sacc[kf][r] = sv;
        tmax = fmaxf(tmax, sv);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int kf = 0; kf < KF; ++kf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = __expf(sacc[kf][r] - m_new);
        sacc[kf][r] = pv;
        psum += pv;
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < DF; ++d)
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[d][r] *= alpha;

    // ---- O^T += V^T P^T: P from the S^T accumulators, key positions as packed by vt_pack_kernel --------------------------------
    uint4 pf[KF / 2];
#pragma unroll
    for (int s = 0; s < KF / 2; ++s)
      pf[s] = make_uint4(pack_bf16x2(sacc[2 * s][0], sacc[2 * s][1]), pack_bf16x2(sacc[2 * s][2], sacc[2 * s][3]),
                         pack_bf16x2(sacc[2 * s + 1][0], sacc[2 * s + 1][1]), pack_bf16x2(sacc[2 * s + 1][2], sacc[2 * s + 1][3]));
#pragma unroll
    for (int d = 0; d < DF; ++d)
#pragma unroll
      for (int s = 0; s < KF / 2; ++s) {
        const uint4 vfrag = *reinterpret_cast<const uint4*>(buf + vaddr[s] + d * 16 * 128);
        oacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vfrag), __builtin_bit_cast(bf16x8_t, pf[s]),
                                                          oacc[d], 0, 0, 0);
      }
    // the next tile has landed (this wave's pieces) and this wave is done reading the current one
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- finish: 1/l, residual, store ------------------------------------------------------------------------------------------
  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  const float inv = 1.0f / l_tot;
  if (!q_ok) return;
  bf16_raw* orow = reinterpret_cast<bf16_raw*>(p.o) + ((long long)b * p.Lq + my_q) * p.o_ld + h * DH;
  const bf16_raw* rrow = p.res ? reinterpret_cast<const bf16_raw*>(p.res) + ((long long)b * p.Lq + my_q) * p.res_ld + h * DH : nullptr;
#pragma unroll
  for (int d = 0; d < DF; ++d) {
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = oacc[d][r] * inv;
    const int c = d * 16 + qg * 4;
    if (rrow) {
      const uint2 rv = *reinterpret_cast<const uint2*>(rrow + c);
      o[0] += __uint_as_float(rv.x << 16); o[1] += __uint_as_float(rv.x & 0xffff0000u);
      o[2] += __uint_as_float(rv.y << 16); o[3] += __uint_as_float(rv.y & 0xffff0000u);
    }
    *reinterpret_cast<uint2*>(orow + c) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
static bool attn_dma_eligible(const GmAttnDesc& d) {
  auto al = [](const void* p, int a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
  return d.dtype == GM_BF16 && !d.causal && d.k_bs == 0 && d.v_bs == 0 && (d.dh == 64 || d.dh == 128 || d.dh == 256) && d.Lq >= 128 && d.Lk >= 128 &&
         d.q_ld % 8 == 0 && d.k_ld % 8 == 0 && d.v_ld % 8 == 0 && al(d.q, 16) && al(d.k, 16) && al(d.v, 16) &&
         d.o_ld % 4 == 0 && al(d.o, 8) && (!d.res || (d.res_ld % 4 == 0 && al(d.res, 8)));
}

extern "C" long long gm_attention_workspace_bytes(const GmAttnDesc* d) {
  if (!d || !attn_dma_eligible(*d)) return 0;
  const long long lk_pad = ((long long)d->Lk + 63) / 64 * 64;
  return (long long)d->B * d->H * d->dh * lk_pad * 2;
}

template <int DH>
static void launch_attn_dma(const GmAttnDesc& d, bf16_raw* vt, int lk_pad, hipStream_t st) {
  static bool attr_set = false;
  auto kern = attn_dma_kernel<DH>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) (void)hipGetLastError();
    attr_set = true;
  }
  dim3 grid((d.Lq + 127) / 128, d.B * d.H);
  kern<<<grid, 512, (size_t)2 * (64 * DH * 2 + DH * 128), st>>>(d, vt, lk_pad);
}

// returns 1 if the LDS-DMA path was launched, 0 if the caller should use the register-staged kernel
extern "C" int gm_attention_dma_try(const GmAttnDesc* dp, void* stream) {
  const GmAttnDesc& d = *dp;
  if (!attn_dma_eligible(d) || !d.workspace || d.workspace_bytes < gm_attention_workspace_bytes(dp)) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int lk_pad = (d.Lk + 63) / 64 * 64;
  bf16_raw* vt = reinterpret_cast<bf16_raw*>(d.workspace);
  dim3 pg(lk_pad / 64, d.dh / 64, d.B * d.H);
  vt_pack_kernel<<<pg, 256, 0, st>>>(reinterpret_cast<const bf16_raw*>(d.v), d.v_ld, vt, d.H, d.Lk, lk_pad, d.dh);
  if (d.dh == 64) launch_attn_dma<64>(d, vt, lk_pad, st);
  else if (d.dh == 128) launch_attn_dma<128>(d, vt, lk_pad, st);
  else launch_attn_dma<256>(d, vt, lk_pad, st);
  return 1;
}
