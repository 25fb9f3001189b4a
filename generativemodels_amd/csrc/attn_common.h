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
};
