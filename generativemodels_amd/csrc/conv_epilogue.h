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
