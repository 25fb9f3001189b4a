// Pieces shared by the LDS-DMA convolution kernels (conv_dma.hip, conv_sk.hip): the LDS-DMA issue / wait primitives, the opaque kernarg
// descriptor access and the LDS-transposed epilogue with fused GroupNorm statistics.
#pragma once
#include "conv_epilogue.h"

#define DMA_ROWB 64
// A per-lane value that is invariant over the tiles of a work-group (an address built from the lane id) is hoisted out of the tile loop by
// the optimiser and then LIVE across every phase of the tile -- with 128 registers per wave that pushed such values into scratch memory, and
// a scratch reload issued behind the next tile's patch request returns only after the patch (vector memory returns in order).  Each phase
// therefore derives its addresses from its own opaque copy of the lane id: one live register instead of a table.
#define OPAQUE_LANE(name) int name = lane; asm volatile("" : "+v"(name))
__device__ __forceinline__ int dma_swz(int row) { return (row ^ (row >> 1)) & 3; }  // period 8 rows


// one LDS-DMA piece: 64 lanes x 16 bytes -> LDS [lds_dst, lds_dst + 1 KiB), lane i from gsrc(i).  M0 is compiler-reserved: saved
// and restored inside the statement (cdna_hip_programming.md 5.7).  Not counted by hipcc: callers wait with dma_wait<N>().
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
// wait until at most N of this wave's DMA pieces are in flight AND all of its LDS reads have returned: the barrier that follows
// releases other waves to overwrite the ring slot / patch this wave has been reading
template <int N>
__device__ __forceinline__ void dma_wait() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

// ---- epilogue of the LDS-DMA kernels ---------------------------------------------------------------------------------------------
// What the cycle stamps of tools/conv_timeline.py showed for the shared conv_epilogue_lds() on a 64 -> 64 channel tile (63.6 k cycles of
// work-group life, 27 k of them in the tap loop): 15.5 k cycles in the epilogue and 6.9 k in the statistics.  The ISA had (1) 16 x 3
// dependent, individually branched global_load_dword for bias / shortcut bias / timestep row, (2) the residual load of row group it + 1
// ordered behind the output store of row group it (res and y may alias: hipcc cannot hoist it), i.e. a load round trip plus a store
// acknowledge per 8 rows, (3) 48 ds_bpermute round trips for the cross-lane statistic sums and a modulo by the tile count.  Here:
//   * the per-channel addend (bias + shortcut bias + timestep row, same order of additions) is staged ONCE per work-group into LDS
//     while the first patch is in flight (dma_addv) and read back with one ds_read_b128 per channel fragment;
//   * all residual rows of a pass are requested before the first LDS write, so their latency overlaps the transpose;
//   * the transpose scratch is wave-private: a wave-level fence replaces the work-group barrier between its write and read halves;
//   * statistics: lane sums over its rows -> one DPP row rotate (lane ^ 8) -> per-(wave, 16-lane row) partials in LDS -> one fixed-order
//     fp64 sum per channel.  Deterministic, no atomics.
// The tile geometry is the kernel's (TH = 4, TW = 16): a wave's rows m_base + v are MF W-lines, (depth, height) of a line are wave-uniform.
// The descriptor is a by-value kernel argument (~300 bytes = 75 SGPRs if every field is kept live).  In the tile loop the optimiser hoists every
// field read out of the loop; with ~100 SGPRs per wave that spilled ~170 of them into VGPR lanes (v_writelane / v_readlane around every phase).
// The phases outside the tap loop therefore read their fields through an opaque pointer to the kernarg segment: a scalar load next to the
// use (scalar cache hit), nothing live across the tap loop.
typedef const __attribute__((address_space(4))) GmConvDesc KDesc;
__device__ __forceinline__ KDesc& cold_desc() {
  KDesc* k = (KDesc*)__builtin_amdgcn_kernarg_segment_ptr();  // the descriptor is the kernel's only argument: offset 0
  asm volatile("" : "+s"(k));
  return *k;
}

__device__ __forceinline__ float dpp_row_ror8(float v) {  // value of lane ^ 8 (rotate by 8 inside each row of 16 lanes)
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xF, 0xF, true));
}

// sum over the four lanes {l, l ^ 8, l ^ 16, l ^ 32} ... of a wave: the rows (lane / 8) of one 16-byte output segment.  One DPP rotate and
// the two gfx950 lane-swap instructions, no LDS round trip; every lane ends with the same value (fixed order of additions).
__device__ __forceinline__ float wave_segment_sum(float v) {
  v += dpp_row_ror8(v);
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);  // rows 1 <-> 0, 3 <-> 2
  v = __int_as_float(a[0]) + __int_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);  // lanes 32.. <-> ..31
  return __int_as_float(b[0]) + __int_as_float(b[1]);
}

// output rows of one epilogue pass: inside-the-volume flag and the residual values of this lane's 16-byte segment
template <int NIT> struct EpRows { bool inside[NIT]; uint4 rv[NIT]; };

struct EpTile { int n, od0, oh0, ow0, co_base, par; };  // (wave-uniform) output tile of the work-group

template <typename T, int MF, int KS, typename D>
__device__ __forceinline__ void dma_epilogue_place(const D& p, const EpTile& t, int line0, int lane, int co, int it, bool& in, long long& vox) {
  const int Dl = KS == 2 ? p.Ds : p.Do, Hl = KS == 2 ? p.Hs : p.Ho, Wl = KS == 2 ? p.Ws : p.Wo;  // KS = 2: the tile walks the low-resolution grid
  const int line = line0 + (it >> 1);                         // wave-uniform: W-line of the tile, (depth, height) = (line / 4, line % 4)
  const int od = t.od0 + (line >> 2), oh = t.oh0 + (line & 3), ow = t.ow0 + (it & 1) * 8 + (lane >> 3);
  in = co < p.Cout && od < Dl && oh < Hl && ow < Wl;
  vox = KS == 2 ? (((long long)t.n * p.Do + 2 * od + ((t.par >> 2) & 1)) * p.Ho + 2 * oh + ((t.par >> 1) & 1)) * p.Wo + 2 * ow + (t.par & 1)
                : (((long long)t.n * p.Do + od) * p.Ho + oh) * p.Wo + ow;
}

// addresses + residual requests of all row groups of pass PASS (no output activation: the hot form).  Called ahead of the transpose -- for
// pass 0 right after the tap loop -- so that the residual's latency is covered by whatever runs in between.
template <typename T, int MF, int KS, int PASS, typename D>
__device__ __forceinline__ void dma_epilogue_rows(const D& p, const EpTile& t, int line0, int lane, EpRows<MF * 2>& R) {
  constexpr int VECW = 16 / (int)sizeof(T), NF_PER_PASS = 128 / (16 * (int)sizeof(T)), NIT = MF * 2;
  const T* res = reinterpret_cast<const T*>(p.res);
  const int co = t.co_base + PASS * NF_PER_PASS * 16 + (lane & 7) * VECW;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    R.inside[it] = false;
    R.rv[it] = make_uint4(0u, 0u, 0u, 0u);
  }
  if (p.post_act == 0) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      long long vox;
      dma_epilogue_place<T, MF, KS>(p, t, line0, lane, co, it, R.inside[it], vox);
      if (res && R.inside[it]) R.rv[it] = *reinterpret_cast<const uint4*>(res + vox * p.res_ld + co);
    }
  }
}

// first half of an epilogue pass: accumulators of the 16x16x32 / 16x16x4 MFMA layout (lane = voxel l15, 4 channels q * 4 ..) + addend -> the
// wave's transpose scratch, row = voxel, 144-byte pitch
// PERM: the accumulator rows stand for channels in the order of direct_chan() below (the kernels with the register-direct epilogue, whose rare
// output-activation form still comes through here)
template <typename T, int MF, int NFR, int PASS, bool PERM = false>
__device__ __forceinline__ void dma_epilogue_write(f32x4_t (&acc)[NFR][MF], char* lds, const float* addv, int lane) {
  constexpr int ROWB_E = 144;                                   // 128 B of channels + 16 B pad
  constexpr int NF_PER_PASS = 128 / (16 * (int)sizeof(T));      // 4 (bf16) or 2 (fp32) channel fragments per pass
  const int l15 = lane & 15, q = lane >> 4;
  // ---- accumulators + addend -> LDS, row = voxel, 4 channels per lane ---------------------------------------------------------------
#pragma unroll
  for (int nl = 0; nl < NF_PER_PASS; ++nl) {
    constexpr int NF0 = PASS * NF_PER_PASS;
    if (NF0 + nl < NFR) {
      const int nf = NF0 + nl < NFR ? NF0 + nl : NFR - 1;
      // channel of this lane's first row of fragment nf, relative to the work-group's block / to the pass's 128-byte run
      const int ch = PERM && sizeof(T) == 2 ? 32 * (nf >> 1) + 8 * q + 4 * (nf & 1) : nf * 16 + q * 4;
      const int chl = PERM && sizeof(T) == 2 ? ch : nl * 16 + q * 4;
      const float4 add = *reinterpret_cast<const float4*>(addv + ch);
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        char* dst = lds + (mf * 16 + l15) * ROWB_E + chl * (int)sizeof(T);
        const float o0 = acc[nf][mf][0] + add.x, o1 = acc[nf][mf][1] + add.y, o2 = acc[nf][mf][2] + add.z, o3 = acc[nf][mf][3] + add.w;
        if (sizeof(T) == 4) *reinterpret_cast<float4*>(dst) = make_float4(o0, o1, o2, o3);
        else *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
      }
    }
  }
}

// second half: transpose scratch -> global (lane = row it * 8 + lane / 8, 16-byte segment lane % 8), residual, activation, statistics.  The
// accumulator layout does not matter here.  EPASSES = epilogue passes of the tile (dims of st_s / st_q).
template <typename T, int MF, int KS, int PASS, int EPASSES, typename D>
__device__ __forceinline__ void dma_epilogue_store(const D& p, char* lds, const EpTile& t, int line0, int lane, const EpRows<MF * 2>& R,
                                                   float (&st_s)[EPASSES][16 / (int)sizeof(T)], float (&st_q)[EPASSES][16 / (int)sizeof(T)]) {
  constexpr int VECW = 16 / (int)sizeof(T);
  constexpr int ROWB_E = 144;
  constexpr int NF_PER_PASS = 128 / (16 * (int)sizeof(T));
  constexpr int NIT = MF * 2;                                   // 8 rows x 8 segments per wave instruction
  const int seg = lane & 7, lw = lane >> 3;
  T* yout = reinterpret_cast<T*>(p.y);
  const T* res = reinterpret_cast<const T*>(p.res);
  const int co = t.co_base + PASS * NF_PER_PASS * 16 + seg * VECW;
  const bool fast = p.post_act == 0;  // wave-uniform; the activation form below is compact, sequential code (VQ-VAE / discriminator convolutions)
  // the scratch is this wave's own and a wave's LDS instructions execute in order: order the two halves for the compiler, no s_barrier
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // ---- LDS -> global: lane = (row it*8 + lane/8, 16-byte segment lane%8) -------------------------------------------------------------
  if (fast) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      uint4 raw = *reinterpret_cast<const uint4*>(lds + (it * 8 + lw) * ROWB_E + seg * 16);
      if (R.inside[it]) {
        if (res) {
          float o[VECW], r[VECW];
          Vec16<T>::unpack(raw, o);
          Vec16<T>::unpack(R.rv[it], r);
#pragma unroll
          for (int i = 0; i < VECW; ++i) o[i] += r[i];
          raw = Vec16<T>::pack(o);
        }
        bool in;
        long long vox;
        dma_epilogue_place<T, MF, KS>(p, t, line0, lane, co, it, in, vox);  // (recomputed: cheaper than 2 live registers per row group)
        *reinterpret_cast<uint4*>(yout + vox * p.y_ld + co) = raw;
        if (p.stats) {  // statistics of the values as stored (rounded to T), like a separate pass over the tensor would see them
          float o[VECW];
          Vec16<T>::unpack(raw, o);
#pragma unroll
          for (int i = 0; i < VECW; ++i) { st_s[PASS][i] += o[i]; st_q[PASS][i] += o[i] * o[i]; }
        }
      }
    }
  } else {
#pragma unroll 1
    for (int it = 0; it < NIT; ++it) {
      bool in;
      long long vox;
      dma_epilogue_place<T, MF, KS>(p, t, line0, lane, co, it, in, vox);
      if (in) {
        float o[VECW];
        Vec16<T>::unpack(*reinterpret_cast<const uint4*>(lds + (it * 8 + lw) * ROWB_E + seg * 16), o);
        if (res) {
          float r[VECW];
          Vec16<T>::unpack(*reinterpret_cast<const uint4*>(res + vox * p.res_ld + co), r);
#pragma unroll
          for (int i = 0; i < VECW; ++i) o[i] += r[i];
        }
#pragma unroll
        for (int i = 0; i < VECW; ++i) o[i] = conv_post_act(o[i], p.post_act);
        const uint4 raw = Vec16<T>::pack(o);
        *reinterpret_cast<uint4*>(yout + vox * p.y_ld + co) = raw;
        if (p.stats) {
          Vec16<T>::unpack(raw, o);
#pragma unroll
          for (int i = 0; i < VECW; ++i) { st_s[PASS][i] += o[i]; st_q[PASS][i] += o[i] * o[i]; }
        }
      }
    }
  }
  // the next pass (or the statistic partials) overwrites the scratch this one read
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T, int MF, int NFR, int KS, int PASS, bool PERM = false, typename D>
__device__ __forceinline__ void dma_epilogue_pass(const D& p, f32x4_t (&acc)[NFR][MF], char* lds, const float* addv, const EpTile& t, int line0,
                                                  int lane, const EpRows<MF * 2>& R,
                                                  float (&st_s)[(NFR * 16 * (int)sizeof(T) + 127) / 128][16 / (int)sizeof(T)],
                                                  float (&st_q)[(NFR * 16 * (int)sizeof(T) + 127) / 128][16 / (int)sizeof(T)]) {
  dma_epilogue_write<T, MF, NFR, PASS, PERM>(acc, lds, addv, lane);
  dma_epilogue_store<T, MF, KS, PASS, (NFR * 16 * (int)sizeof(T) + 127) / 128>(p, lds, t, line0, lane, R, st_s, st_q);
}


// ---- register-direct epilogue (round 5) -------------------------------------------------------------------------------------------------------
// The 16x16 MFMA leaves lane (l15 = voxel column, q = lane / 16) of fragment nf with output rows 4q .. 4q + 3 of that fragment's 16: four
// channels of one voxel.  WHICH channel an MFMA row stands for is free -- it is decided by the order of the weight rows in the LDS panel, i.e. by
// the source addresses of the panel's DMA requests.  With the order below a lane's 16 values per voxel are 16-byte runs that the four lanes
// q = 0..3 of a voxel continue into 64 contiguous bytes: store st of a voxel row covers bytes [64 st + 16 q, + 16) of the work-group's
// channel block (bf16: 2 stores, fp32: 4), a wave instruction writes sixteen 64-byte segments, and the LDS transpose of the epilogue (16
// ds_write_b64 + 8 ds_read_b128 + two fences per wave and tile, plus a 64-bit voxel address per row group and lane) is gone: a row's address is
// a scalar base per W line + one per-lane offset.
//   bf16: channel(nf, q, j) = 32 (nf / 2) + 8 q + 4 (nf % 2) + j        fp32: channel(nf, q, j) = 16 nf + 4 q + j  (the natural order)
template <typename T>
__device__ __forceinline__ constexpr int direct_chan(int nf, int q, int j) {
  return sizeof(T) == 2 ? 32 * (nf >> 1) + 8 * q + 4 * (nf & 1) + j : 16 * nf + 4 * q + j;
}
// the output channel (within the work-group's 64) whose weights LDS panel column `col` holds: column nf * 16 + i is MFMA row i of fragment nf
template <typename T>
__device__ __forceinline__ int direct_col_chan(int col) {
  const int nf = col >> 4, i = col & 15;
  return direct_chan<T>(nf, i >> 2, i & 3);
}
__device__ __forceinline__ float dpp_row_ror4(float v) {  // value of lane (l + 4) mod 16 of the same row of 16 lanes
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x124, 0xF, 0xF, true));
}
