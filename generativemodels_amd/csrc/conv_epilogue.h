// Epilogue + work-group placement helpers shared by the MFMA convolution kernels.
#pragma once
#include "conv_common.h"

// acc[nf][mf][r] = out[cout = co_base + nf*16 + 4*(lane>>4) + r][voxel m = m_base + mf*16 + (lane&15)]
// -> + bias + timestep row + residual, activation, 8/16-byte NDHWC stores (scalar stores for ragged channel counts).
template <typename T, int MF, int NFR>
__device__ __forceinline__ void conv_epilogue(const GmConvDesc& p, f32x4_t (&acc)[NFR][MF], int n, int m_base, int co_base,
                                              int od0, int oh0, int ow0, int l15, int q) {
  T* yout = reinterpret_cast<T*>(p.y);
  const T* res = reinterpret_cast<const T*>(p.res);
  const int th = 1 << p.lth, tw = 1 << p.ltw;
  const bool st_vec = (p.Cout % 4 == 0) && (p.y_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.y) & (4 * sizeof(T) - 1)) == 0) &&
                      (!res || ((p.res_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.res) & (4 * sizeof(T) - 1)) == 0)));
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int m = m_base + mf * 16 + l15;
    const int a = m >> (p.lth + p.ltw), bb = (m >> p.ltw) & (th - 1), c = m & (tw - 1);
    const int od = od0 + a, oh = oh0 + bb, ow = ow0 + c;
    if (od >= p.Do || oh >= p.Ho || ow >= p.Wo) continue;
    const long long vox = (((long long)n * p.Do + od) * p.Ho + oh) * p.Wo + ow;
#pragma unroll
    for (int nf = 0; nf < NFR; ++nf) {
      const int co = co_base + nf * 16 + q * 4;
      if (co >= p.Cout) continue;
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = acc[nf][mf][r];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (co + r < p.Cout) {
          if (p.bias) o[r] += p.bias[co + r];
          if (p.rowvec) o[r] += p.rowvec[(long long)n * p.rowvec_bstride + co + r];
        }
      }
      if (st_vec) {
        if (res) {
          if (sizeof(T) == 4) {
            const float4 rv = *reinterpret_cast<const float4*>(res + vox * p.res_ld + co);
            o[0] += rv.x; o[1] += rv.y; o[2] += rv.z; o[3] += rv.w;
          } else {
            const uint2 rv = *reinterpret_cast<const uint2*>(res + vox * p.res_ld + co);
            o[0] += __uint_as_float(rv.x << 16); o[1] += __uint_as_float(rv.x & 0xffff0000u);
            o[2] += __uint_as_float(rv.y << 16); o[3] += __uint_as_float(rv.y & 0xffff0000u);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = conv_post_act(o[r], p.post_act);
        if (sizeof(T) == 4) {
          *reinterpret_cast<float4*>(yout + vox * p.y_ld + co) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
          uint2 pk;
          pk.x = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
          pk.y = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
          *reinterpret_cast<uint2*>(yout + vox * p.y_ld + co) = pk;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (co + r < p.Cout) {
            float v = o[r];
            if (res) v += ElemIO<T>::ld(res + vox * p.res_ld + co + r);
            ElemIO<T>::st(yout + vox * p.y_ld + co + r, conv_post_act(v, p.post_act));
          }
        }
      }
    }
  }
}

// XCD-aware work-group id: the dispatcher round-robins consecutive work-groups over the 8 XCDs (each with a private L2);
// remap so that every XCD walks a contiguous range of tiles and neighbouring halo patches hit the same L2.  Bijective for
// any grid size; a pure speed choice (results do not depend on placement).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nb) {
  const unsigned q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Coalesced epilogue: each wave transposes its 64-voxel x (NFR*16)-channel accumulator tile through LDS so that every lane
// ends up with 16 contiguous bytes of one voxel row and a wave-instruction writes eight complete 128-byte row segments
// (the per-lane-4-channel store of conv_epilogue() wrote 32-byte fragments and cost ~half the kernel time on MI355X).
// `lds` = this wave's private 64 x 144-byte scratch (the operand buffers are dead after the main loop's last barrier).
// Requirements (checked by the caller): Cout, y_ld (and res_ld) multiples of 16/sizeof(T), 16-byte aligned bases.
// Optionally accumulates, per lane, sum / sum of squares of the values it stores (st_s/st_q[pass][i] = channel
// co_base + pass*NF_PER_PASS*16 + (lane&7)*VECW + i over this lane's rows) for the fused GroupNorm statistics.
// Sub-pixel output placement (in_mode 3, conv_dma.hip): the tile walks the LOW-resolution grid (Dl x Hl x Wl) and every voxel lands at
// (2 od + pd, 2 oh + ph, 2 ow + pw) of the full-resolution output p.Do x p.Ho x p.Wo.
struct ConvOutMap { int Dl, Hl, Wl, pd, ph, pw; };

// One pass (128 bytes of channels per voxel row) of the LDS-transposed epilogue; PASS is a template constant so that the accumulator
// fragments it touches are compile-time register indices (a run-time pass loop that the unroller gives up on would force the whole
// accumulator array into scratch memory -- seen on the 128-channel tile: one scratch store behind every MFMA).
template <typename T, int MF, int NFR, int PASS>
__device__ __forceinline__ void conv_epilogue_lds_pass(const GmConvDesc& p, f32x4_t (&acc)[NFR][MF], char* lds, int n, int m_base,
                                                       int co_base, int od0, int oh0, int ow0, int lane,
                                                       float (&st_s)[(NFR * 16 * (int)sizeof(T) + 127) / 128][16 / (int)sizeof(T)],
                                                       float (&st_q)[(NFR * 16 * (int)sizeof(T) + 127) / 128][16 / (int)sizeof(T)],
                                                       const ConvOutMap* om) {
  constexpr int VECW = 16 / (int)sizeof(T);
  constexpr int ROWB_E = 144;                                   // 128 B of channels + 16 B pad
  constexpr int NF_PER_PASS = 128 / (16 * (int)sizeof(T));      // 4 (bf16) or 2 (fp32) channel fragments per pass
  const int l15 = lane & 15, q = lane >> 4;
  const int th = 1 << p.lth, tw = 1 << p.ltw;
  T* yout = reinterpret_cast<T*>(p.y);
  const T* res = reinterpret_cast<const T*>(p.res);
  // ---- accumulators (+ bias, + timestep row) -> LDS, row = voxel, 4 channels per lane ---------------------------------
#pragma unroll
  for (int nl = 0; nl < NF_PER_PASS; ++nl) {
    constexpr int NF0 = PASS * NF_PER_PASS;
    if (NF0 + nl < NFR) {
      const int nf = NF0 + nl < NFR ? NF0 + nl : NFR - 1;
      const int co = co_base + nf * 16 + q * 4;
      float add[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = 0.f;
        if (co + r < p.Cout) {
          if (p.bias) a += p.bias[co + r];
          if (p.skip_bias) a += p.skip_bias[co + r];
          if (p.rowvec) a += p.rowvec[(long long)n * p.rowvec_bstride + co + r];
        }
        add[r] = a;
      }
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        char* dst = lds + (mf * 16 + l15) * ROWB_E + (nl * 16 + q * 4) * (int)sizeof(T);
        const float o0 = acc[nf][mf][0] + add[0], o1 = acc[nf][mf][1] + add[1], o2 = acc[nf][mf][2] + add[2], o3 = acc[nf][mf][3] + add[3];
        if (sizeof(T) == 4) *reinterpret_cast<float4*>(dst) = make_float4(o0, o1, o2, o3);
        else *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
      }
    }
  }
  __syncthreads();
  // ---- LDS -> global: lane = (voxel it*8 + lane/8, 16-byte segment lane%8) ---------------------------------------------
  const int seg = lane & 7;
  const int co = co_base + PASS * NF_PER_PASS * 16 + seg * VECW;
#pragma unroll
  for (int it = 0; it < MF * 2; ++it) {
    const int v = it * 8 + (lane >> 3);
    const int m = m_base + v;
    const int a = m >> (p.lth + p.ltw), bb = (m >> p.ltw) & (th - 1), c = m & (tw - 1);
    const int od = od0 + a, oh = oh0 + bb, ow = ow0 + c;
    const bool inside = om ? (od < om->Dl && oh < om->Hl && ow < om->Wl) : (od < p.Do && oh < p.Ho && ow < p.Wo);
    if (inside && co < p.Cout) {
      const long long vox = om ? (((long long)n * p.Do + 2 * od + om->pd) * p.Ho + 2 * oh + om->ph) * p.Wo + 2 * ow + om->pw
                               : (((long long)n * p.Do + od) * p.Ho + oh) * p.Wo + ow;
      uint4 raw = *reinterpret_cast<const uint4*>(lds + v * ROWB_E + seg * 16);
      if (res || p.post_act) {
        float o[VECW];
        Vec16<T>::unpack(raw, o);
        if (res) {
          float rv[VECW];
          Vec16<T>::unpack(*reinterpret_cast<const uint4*>(res + vox * p.res_ld + co), rv);
#pragma unroll
          for (int i = 0; i < VECW; ++i) o[i] += rv[i];
        }
        if (p.post_act) {
#pragma unroll
          for (int i = 0; i < VECW; ++i) o[i] = conv_post_act(o[i], p.post_act);
        }
        raw = Vec16<T>::pack(o);
      }
      *reinterpret_cast<uint4*>(yout + vox * p.y_ld + co) = raw;
      if (p.stats) {  // statistics of the values as stored (rounded to T), like a separate pass over the tensor would see them
        float o[VECW];
        Vec16<T>::unpack(raw, o);
#pragma unroll
        for (int i = 0; i < VECW; ++i) { st_s[PASS][i] += o[i]; st_q[PASS][i] += o[i] * o[i]; }
      }
    }
  }
}

template <typename T, int MF, int NFR>
__device__ __forceinline__ void conv_epilogue_lds(const GmConvDesc& p, f32x4_t (&acc)[NFR][MF], char* lds, int n, int m_base,
                                                  int co_base, int od0, int oh0, int ow0, int lane,
                                                  float (&st_s)[(NFR * 16 * (int)sizeof(T) + 127) / 128][16 / (int)sizeof(T)],
                                                  float (&st_q)[(NFR * 16 * (int)sizeof(T) + 127) / 128][16 / (int)sizeof(T)],
                                                  const ConvOutMap* om = nullptr) {
  constexpr int NF_PER_PASS = 128 / (16 * (int)sizeof(T));
  constexpr int PASSES = (NFR + NF_PER_PASS - 1) / NF_PER_PASS;
  static_assert(PASSES <= 4, "at most 4 epilogue passes (128 output channels in fp32)");
  conv_epilogue_lds_pass<T, MF, NFR, 0>(p, acc, lds, n, m_base, co_base, od0, oh0, ow0, lane, st_s, st_q, om);
  if constexpr (PASSES > 1) { __syncthreads(); conv_epilogue_lds_pass<T, MF, NFR, 1>(p, acc, lds, n, m_base, co_base, od0, oh0, ow0, lane, st_s, st_q, om); }
  if constexpr (PASSES > 2) { __syncthreads(); conv_epilogue_lds_pass<T, MF, NFR, 2>(p, acc, lds, n, m_base, co_base, od0, oh0, ow0, lane, st_s, st_q, om); }
  if constexpr (PASSES > 3) { __syncthreads(); conv_epilogue_lds_pass<T, MF, NFR, 3>(p, acc, lds, n, m_base, co_base, od0, oh0, ow0, lane, st_s, st_q, om); }
}

template <typename T>
__device__ __forceinline__ bool conv_epilogue_lds_ok(const GmConvDesc& p) {
  constexpr int VECW = 16 / (int)sizeof(T);
  return (p.Cout % VECW == 0) && (p.y_ld % VECW == 0) && ((reinterpret_cast<uintptr_t>(p.y) & 15) == 0) &&
         (!p.res || ((p.res_ld % VECW == 0) && ((reinterpret_cast<uintptr_t>(p.res) & 15) == 0)));
}
