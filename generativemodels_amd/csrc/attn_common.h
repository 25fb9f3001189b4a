// Attention descriptor shared by attention.hip (register-staged, any dtype / head dim) and attention_dma.hip (bf16, LDS-DMA).
#pragma once
#include "gm_common.h"

// key ranges of the split-KV single-query attention of the decode step (small_ops.hip; the merge code unrolls over it)
#define GM_DECODE_KV_SPLITS 16

struct GmAttnDesc {
  const void* q; long long q_ld;
  const void* k; long long k_ld;
  const void* v; long long v_ld;
  const void* res; long long res_ld;  // optional residual, same geometry as o
  void* o; long long o_ld;
  int B, H, Lq, Lk, dh;
  float scale;
  int dtype;
  void* workspace;             // optional scratch (gm_attention_workspace_bytes): enables the LDS-DMA kernel
  long long workspace_bytes;
  int causal;                  // 1: query i attends keys j <= i + (Lk - Lq) only (SABlock causal mask, blocks/selfattention.py:133-134)
  long long k_bs, v_bs;        // batch strides of k / v in elements; 0 = dense (Lk * ld).  A KV cache is [B][max_len][C] read up to Lk.
  double* stats;               // optional [gm_attention_stats_slots][B][H * dh][2] per-channel (sum, sum of squares) partials of the stored output
  int vt_packed;               // 1: the workspace already holds the transposed V image (written by gm_linear_rows_affine_vt): no pack launch
  float* lse;                  // optional [B*H][Lq] fp32: log sum_k exp(scale q.k), written by the LDS-DMA path only (the training forward keeps it for
                               // gm_attention_backward_fused)
};

// Attention backward descriptor (attention_bwd.hip: fp32-MFMA fused kernels and the bf16 score pass; attention_bwd_dma.hip: fused bf16 LDS-DMA kernels)
struct GmAttnBwdDesc {
  const void* q; long long q_ld;
  const void* k; long long k_ld;
  const void* v; long long v_ld;
  const void* o; long long o_ld;       // forward output WITHOUT the residual
  const void* go; long long go_ld;     // gradient of the forward output
  void* dq; long long dq_ld;
  void* dk; long long dk_ld;
  void* dv; long long dv_ld;
  int B, H, Lq, Lk, dh;
  float scale;
  int dtype;
  void* workspace; long long workspace_bytes;  // gm_attention_backward_workspace_bytes / gm_attention_backward_fused_workspace_bytes
};

#ifdef __HIPCC__
// The four lanes l15 + 16 * {0, 1, 2, 3} of a wave hold partial results of one MFMA column: max / sum over them by gfx950's lane-swap VALU
// instructions (v_permlane32_swap exchanges the wave's halves, v_permlane16_swap the odd and even 16-lane rows) instead of two ds_bpermute
// round trips through the LDS pipeline.  Every lane receives the same value, operands combined in a fixed order.
__device__ __forceinline__ float attn_quad_max(float x) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float attn_quad_sum(float x) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float m = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// one 1 KB LDS-DMA piece: every lane's 16 bytes at gsrc land at lds_dst + 16 * lane (M0 saved and restored around the request)
__device__ __forceinline__ void attn_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
// ... the same piece from a wave-uniform base (SGPR pair) + this lane's 32-bit byte offset: per-tile address arithmetic on the scalar unit
__device__ __forceinline__ void attn_dma16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
// rows [L][ld] (channels [h * dh, (h + 1) * dh) per head) -> image[b * H + h][channel][position], zero beyond L, position order inside each
// 32-block = the order a 16x16x32 MFMA consumes two 16-row accumulator fragments (attention_dma.hip: vt_pack_kernel); dh a multiple of 64
void gm_attn_pack_transposed(const bf16_raw* rows, long long ld, bf16_raw* image, int B, int H, int L, int L_pad, int dh, hipStream_t st);
// ... three tensors in one launch (3 * B * H <= 65535)
void gm_attn_pack_transposed3(const bf16_raw* const rows[3], const long long ld[3], bf16_raw* const image[3], const int L[3], const int L_pad[3], int B, int H, int dh,
                              hipStream_t st);
#endif
