// One autoregressive decoding step of the decoder-only transformer, issued from native code.
//
// A token step was 62 tiny launches in round 2 (per block: LayerNorm + stacked q|k|v GEMM writing k, v into the KV caches, 1 x t attention, out_proj +
// residual, LayerNorm + MLP up + GELU, MLP down + residual).  Issued one by one through the Python binding it cost 2.7 ms per token
// (20 us of interpreter + descriptor work per launch) -- no faster than the reference's recompute-the-prefix loop on this model
// size.  This entry point walks the block table in C++ and enqueues the kernels back to back.  Round 3: 38 launches where the fused kernels of
// small_ops.hip take the geometry -- per block [LayerNorm + q|k|v + split-KV attention ranges, its prologue assembling the residual stream from
// the previous block's MLP partials] -> [out_proj + residual, its prologue merging the attention ranges] -> [LayerNorm + MLP up + GELU + MLP down
// as K-slice partials] -- and the round-2 sequence (or a partly fused one) otherwise
// (reference semantics: networks/nets/transformer.py:98-106, blocks/transformerblock.py:86-91, blocks/selfattention.py:98-147).
#include <cstdlib>
#include "attn_common.h"
#include "conv_common.h"

extern "C" int gm_embed_tokens_dev(const long long* indices, const void* token_weight, const void* position_weight, void* out, long long batch,
                                   int seq_len, int C, int pos0, int num_tokens, int max_positions, int dtype, const int* pos_dev, void* stream);
extern "C" int gm_attention_decode_dev(const GmAttnDesc* dp, const int* lk_dev, void* stream);
extern "C" int gm_attention_decode_split(const GmAttnDesc* dp, const int* lk_dev, float* ws, int nsplit, int merge, int cap, void* stream);
extern "C" int gm_mlp_rows_fusable(int rows, int C, int M, int dtype);
extern "C" int gm_mlp_rows(const void* x, const float* ln_g, const float* ln_b, float ln_eps, const void* w1, const float* b1, const void* w2,
                           float* P, int rows, int C, int M, int act, int dtype, void* stream);
extern "C" int gm_linear_rows_mlpmerge(const float* P, int nj, const void* x1, const float* b2, void* x0_out, const float* ln_g, const float* ln_b,
                                       float ln_eps, const void* w, const float* bias, void* y, long long y_ld, void* y1, void* y2, long long y12_ld,
                                       int split, int rows, int cin, int cout, int post_act, int dtype, const int* off_dev, long long off_mul,
                                       void* stream);
struct QkvAttnArgs {  // small_ops.hip
  const void* x0;
  const float* mlp_p; int mlp_nj; const void* mlp_x1; const float* mlp_b2; void* x0_out;
  const float* ln_g; const float* ln_b; float ln_eps;
  const void* w; const float* bias;
  void* kcache; void* vcache;
  int B, H, C, dh, cap, pos; const int* pos_dev;
  float scale; float* ws; int chunk, sc_elems;
};
extern "C" int gm_qkv_attn_rows(const QkvAttnArgs* ap, int dtype, void* stream);
extern "C" int gm_linear_rows_kvmerge(const float* kv_ws, int kv_ns, int kv_dh, const void* w, const float* bias, const void* res, long long res_ld,
                                      void* y, long long y_ld, int rows, int cin, int cout, int dtype, void* stream);
extern "C" int gm_layernorm(const void* x, long long x_ld, void* y, long long y_ld, const float* gamma, const float* beta, long long rows,
                            int C, float eps, int dtype, void* stream);
extern "C" int gm_copy_channels(const void* src, long long src_ld, int src_dtype, void* dst, long long dst_ld, int dst_dtype,
                                long long rows, int C, void* stream);
extern "C" int gm_attention_forward(const GmAttnDesc* dp, void* stream);

struct GmDecodeBlock {
  const float *ln1_g, *ln1_b;
  const void* w_qkv; const float* b_qkv;  // packed [3C][C] (gm_pack_conv_weight), bias or null
  const void* w_o; const float* b_o;
  const float *ln3_g, *ln3_b;
  const void* w_1; const float* b_1;      // C -> M
  const void* w_2; const float* b_2;      // M -> C
  void* k_cache; void* v_cache;           // [B][max_len][C]
};

struct GmDecodeDesc {
  int B, C, M, heads, depth, max_len, num_tokens, dtype;
  float ln_eps;
  int pos;                                // position of the token being fed (its K/V land in cache row `pos`)
  const long long* tokens;                // [B] token ids on the device
  const void* tok_emb; const void* pos_emb;
  const GmDecodeBlock* blocks;            // host array of `depth` entries
  const void* w_logits; const float* b_logits;
  void* logits;                           // [B][num_tokens] in dtype
  void* scratch; long long scratch_bytes; // gm_decode_scratch_bytes(B, C, M, dtype)
  const int* pos_dev;                     // optional: the position is read from device memory at run time (HIP-graph replay); `pos` is ignored
};

static long long elt(int dtype) { return dtype == GM_F32 ? 4 : 2; }
// Context windows longer than DECODE_SPLIT_MIN_LEN keys run the single-query attention split over DECODE_KV_SPLITS key ranges (small_ops.hip).
// The choice depends on the window (max_len), not on the position, so a step issued with a host position and its graph replay agree bit for bit.
#define DECODE_KV_SPLITS GM_DECODE_KV_SPLITS
#define DECODE_SPLIT_MIN_LEN 256

extern "C" long long gm_decode_scratch_bytes(int B, int C, int M, int dtype) {
  // x0, x1, h, y: B*C each; qkv: 3*B*C; a: B*M; split-KV attention partials: B * heads * DECODE_KV_SPLITS * (dh + 2) floats, bounded
  // without the head count by B * DECODE_KV_SPLITS * 3 * C (heads <= C); every buffer rounded up to 256 bytes
  auto r = [](long long v) { return (v + 255) / 256 * 256; };
  return 4 * r((long long)B * C * elt(dtype)) + r(3LL * B * C * elt(dtype)) + r((long long)B * M * elt(dtype)) +
         r((long long)B * DECODE_KV_SPLITS * 3 * C * 4) + r((long long)((M + 63) / 64) * B * C * 4);  // + the MLP's K-slice partials
}

extern "C" int gm_linear_rows(const void* x, long long x_ld, const void* w, const float* bias, const void* res, long long res_ld, void* y,
                              long long y_ld, int rows, int cin, int cout, int pre_act, int post_act, int dtype, void* stream);

extern "C" int gm_linear_rows_ln(const void* x, long long x_ld, const float* ln_g, const float* ln_b, float ln_eps, const void* w,
                                 const float* bias, void* y, long long y_ld, void* y1, void* y2, long long y12_ld, int split, int rows, int cin,
                                 int cout, int post_act, int dtype, const int* off_dev, long long off_mul, void* stream);

// y[rows][cout] = act(x[rows][cin] W^T + b) (+ res) through the small-row GEMM kernel (small_ops.hip)
static int linear_rows(const void* x, long long x_ld, const void* w, const float* b, const void* res, long long res_ld, void* y, long long y_ld,
                       int rows, int cin, int cout, int post_act, int dtype, void* stream) {
  return gm_linear_rows(x, x_ld, w, b, res, res_ld, y, y_ld, rows, cin, cout, 0, post_act, dtype, stream);
}

extern "C" int gm_transformer_decode_step(const GmDecodeDesc* dp, void* stream) {
  GM_REQUIRE(dp, "null descriptor");
  const GmDecodeDesc& d = *dp;
  GM_REQUIRE(d.tokens && d.tok_emb && d.pos_emb && d.blocks && d.w_logits && d.logits && d.scratch, "null pointer");
  GM_REQUIRE(d.B > 0 && d.C > 0 && d.M > 0 && d.heads > 0 && d.C % d.heads == 0 && d.depth > 0, "bad geometry");
  GM_REQUIRE(d.pos_dev || (d.pos >= 0 && d.pos < d.max_len), "position outside the context window");
  const int hpos = d.pos_dev ? 0 : d.pos;  // host-side position (0 when the device supplies it)
  GM_REQUIRE(d.scratch_bytes >= gm_decode_scratch_bytes(d.B, d.C, d.M, d.dtype), "scratch too small");
  const long long es = elt(d.dtype);
  auto r = [](long long v) { return (v + 255) / 256 * 256; };
  char* s = reinterpret_cast<char*>(d.scratch);
  char* x0 = s;  s += r((long long)d.B * d.C * es);
  char* x1 = s;  s += r((long long)d.B * d.C * es);
  char* h = s;   s += r((long long)d.B * d.C * es);
  char* y = s;   s += r((long long)d.B * d.C * es);
  char* qkv = s; s += r(3LL * d.B * d.C * es);
  char* a = s;   s += r((long long)d.B * d.M * es);
  float* kv_ws = reinterpret_cast<float*>(s); s += r((long long)d.B * DECODE_KV_SPLITS * 3 * d.C * 4);
  float* mlp_p = reinterpret_cast<float*>(s);
  // the MLP as one launch leaving M / 64 K-slice partials that the next launch's prologue sums with the residual row (small_ops.hip)
  bool mlp_fuse = gm_mlp_rows_fusable(d.B, d.C, d.M, d.dtype) == 1;
  for (int i = 0; i < d.depth; ++i) mlp_fuse = mlp_fuse && d.blocks[i].ln3_g;  // (its staging pass is the LayerNorm)
  const int mlp_nj = d.M / 64;
  const int C = d.C;
  int rc = gm_embed_tokens_dev(d.tokens, d.tok_emb, d.pos_emb, x0, d.B, 1, C, hpos, d.num_tokens, d.max_len, d.dtype, d.pos_dev, stream);
  if (rc) return rc;
  const float scale = 1.0f / sqrtf((float)(C / d.heads));
  for (int i = 0; i < d.depth; ++i) {
    const GmDecodeBlock& b = d.blocks[i];
    GM_REQUIRE(b.w_qkv && b.w_o && b.w_1 && b.w_2 && b.k_cache && b.v_cache, "null block parameter");
    static const bool kv_split = !(getenv("GM_DECODE_KV_SPLIT") && getenv("GM_DECODE_KV_SPLIT")[0] == '0');  // bench switches (tools/diag_c5.py)
    static const bool kv_fuse = !(getenv("GM_DECODE_KV_FUSE") && getenv("GM_DECODE_KV_FUSE")[0] == '0');
    const bool split = kv_split && d.max_len > DECODE_SPLIT_MIN_LEN;
    // LayerNorm + q | k | v + the attention ranges as ONE launch where that kernel takes the geometry (small_ops.hip: qkv_attn_rows_kernel) ...
    bool qkv_done = false;
    if (split) {
      QkvAttnArgs qa = {};
      if (mlp_fuse && i > 0) { qa.mlp_p = mlp_p; qa.mlp_nj = mlp_nj; qa.mlp_x1 = x1; qa.mlp_b2 = d.blocks[i - 1].b_2; qa.x0_out = x0; }
      else qa.x0 = x0;
      qa.ln_g = b.ln1_g; qa.ln_b = b.ln1_b; qa.ln_eps = d.ln_eps;
      qa.w = b.w_qkv; qa.bias = b.b_qkv;
      qa.kcache = b.k_cache; qa.vcache = b.v_cache;
      qa.B = d.B; qa.H = d.heads; qa.C = C; qa.dh = C / d.heads; qa.cap = d.max_len; qa.pos = hpos; qa.pos_dev = d.pos_dev;
      qa.scale = scale; qa.ws = kv_ws;
      rc = gm_qkv_attn_rows(&qa, d.dtype, stream);
      if (rc < 0) return rc;
      qkv_done = rc == 1;
    }
    // ... otherwise LayerNorm + stacked q | k | v projection in one launch; the key / value rows land directly in cache row `pos` of every sequence.
    // Behind a fused MLP the launch first assembles its input x0 = x1 + b2 + sum_j P[j] (and stores it: the out-projection below adds it as the residual)
    char* kdst = reinterpret_cast<char*>(b.k_cache) + (long long)hpos * C * es;
    char* vdst = reinterpret_cast<char*>(b.v_cache) + (long long)hpos * C * es;
    if (qkv_done)
      rc = 0;
    else if (mlp_fuse && i > 0)
      rc = gm_linear_rows_mlpmerge(mlp_p, mlp_nj, x1, d.blocks[i - 1].b_2, x0, b.ln1_g, b.ln1_b, d.ln_eps, b.w_qkv, b.b_qkv, qkv, 3 * C, kdst, vdst,
                                   (long long)d.max_len * C, C, d.B, C, 3 * C, 0, d.dtype, d.pos_dev, C, stream);
    else
      rc = gm_linear_rows_ln(x0, C, b.ln1_g, b.ln1_b, d.ln_eps, b.w_qkv, b.b_qkv, qkv, 3 * C, kdst, vdst, (long long)d.max_len * C, C, d.B, C, 3 * C,
                             0, d.dtype, d.pos_dev, C, stream);
    if (rc) return rc;
    GmAttnDesc at = {};
    at.q = qkv; at.q_ld = 3 * C;
    at.k = b.k_cache; at.k_ld = C; at.k_bs = (long long)d.max_len * C;
    at.v = b.v_cache; at.v_ld = C; at.v_bs = (long long)d.max_len * C;
    at.o = y; at.o_ld = C;
    at.B = d.B; at.H = d.heads; at.Lq = 1; at.Lk = d.pos_dev ? d.max_len : d.pos + 1; at.dh = C / d.heads;
    at.scale = scale; at.dtype = d.dtype;
    bool out_done = false;
    if (split) {
      // the partials are merged by the out-projection's prologue where its K-split kernel applies, by a combine launch otherwise: the two
      // forms give the same bits, and which one runs depends on the geometry only
      const bool fuse = kv_fuse && at.dh % (d.dtype == GM_F32 ? 4 : 8) == 0;
      if (!qkv_done)
        GM_REQUIRE(gm_attention_decode_split(&at, d.pos_dev, kv_ws, DECODE_KV_SPLITS, 0, d.max_len, stream) == 1, "head size beyond the single-query attention kernels");
      rc = fuse ? gm_linear_rows_kvmerge(kv_ws, DECODE_KV_SPLITS, at.dh, b.w_o, b.b_o, x0, C, x1, C, d.B, C, C, d.dtype, stream) : 0;
      if (rc < 0) return rc;
      out_done = rc == 1;
      if (!out_done) GM_REQUIRE(gm_attention_decode_split(&at, d.pos_dev, kv_ws, DECODE_KV_SPLITS, 2, d.max_len, stream) == 1, "merge of the attention partials");
    } else if (d.pos_dev) {
      GM_REQUIRE(gm_attention_decode_dev(&at, d.pos_dev, stream) == 1, "context window too long for the single-query attention kernel");
    } else if ((rc = gm_attention_forward(&at, stream))) {
      return rc;
    }
    if (!out_done && (rc = linear_rows(y, C, b.w_o, b.b_o, x0, C, x1, C, d.B, C, C, 0, d.dtype, stream))) return rc;
    if (mlp_fuse) {
      if ((rc = gm_mlp_rows(x1, b.ln3_g, b.ln3_b, d.ln_eps, b.w_1, b.b_1, b.w_2, mlp_p, d.B, C, d.M, 6, d.dtype, stream))) return rc;
      continue;
    }
    if ((rc = gm_linear_rows_ln(x1, C, b.ln3_g, b.ln3_b, d.ln_eps, b.w_1, b.b_1, a, d.M, nullptr, nullptr, 0, 0, d.B, C, d.M, 6, d.dtype,
                                nullptr, 0, stream))) return rc;
    if ((rc = linear_rows(a, d.M, b.w_2, b.b_2, x1, C, x0, C, d.B, d.M, C, 0, d.dtype, stream))) return rc;
  }
  if (mlp_fuse)
    return gm_linear_rows_mlpmerge(mlp_p, mlp_nj, x1, d.blocks[d.depth - 1].b_2, nullptr, nullptr, nullptr, 0.f, d.w_logits, d.b_logits, d.logits,
                                   d.num_tokens, nullptr, nullptr, 0, 0, d.B, C, d.num_tokens, 0, d.dtype, nullptr, 0, stream);
  return linear_rows(x0, C, d.w_logits, d.b_logits, nullptr, 0, d.logits, d.num_tokens, d.B, C, d.num_tokens, 0, d.dtype, stream);
}
