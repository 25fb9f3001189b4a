/* gm_amd.h -- C ABI of libgmamd.so, the MI355X (gfx950) kernel library behind generativemodels_amd.
 *
 * The reference (Project-MONAI/GenerativeModels) has no native code and no FFI: its "kernels" are torch ops called from
 * Python.  This header is therefore the boundary a maintainer would bind (ctypes / cffi) to replace
 * those op sequences; every entry point names the reference lines it replaces (paths relative to generative/).
 *
 * Conventions
 *   - plain pointers + sizes; device pointers unless stated; no torch / framework types.
 *   - activations live in N[D]HWC ("channels last") layout: a tensor is (ptr, leading dim `ld` in ELEMENTS between
 *     consecutive voxels, C used channels <= ld).  2-D data passes D = 1.
 *   - dtype: GM_F32 (parity path, exact-fp32 MFMA) or GM_BF16 (bf16 storage, fp32 accumulation).
 *   - every call enqueues on `stream` (a hipStream_t; NULL = default stream) and returns immediately.
 *   - return value 0 = success; otherwise a negative argument error or a positive hipError_t, message via gm_last_error().
 */
#ifndef GM_AMD_H
#define GM_AMD_H
#ifdef __cplusplus
extern "C" {
#endif

#define GM_F32 0
#define GM_BF16 1

int gm_abi_version(void);
const char* gm_last_error(void);

/* ---- scheduler (networks/schedulers/ddim.py:156-237, ddpm.py:191-252) ------------------------------------------------
 * One fused element-wise kernel per reverse step.  The per-step scalars are computed by the host with the reference's
 * own fp32 expressions (see generativemodels_amd/networks/schedulers). */
typedef struct GmStepParams {
  int mode;        /* 0 DDIM, 1 DDPM, 2 PNDM transfer x_(t-d) = k0*x - (k1*e)/c_prev (networks/schedulers/pndm.py:276-316; x0 is not written) */
  int pred_type;   /* 0 epsilon, 1 sample, 2 v_prediction */
  float c_sa;      /* alpha_prod_t ** 0.5 */
  float c_sb;      /* beta_prod_t ** 0.5 */
  int clip;        /* clamp x0 to [clip_lo, clip_hi] */
  float clip_lo, clip_hi;
  float c_prev;    /* DDIM: alpha_prod_t_prev ** 0.5 */
  float c_dir;     /* DDIM: (1 - alpha_prod_t_prev - std_dev_t**2) ** 0.5 */
  float k0, k1;    /* DDPM: pred_original_sample_coeff, current_sample_coeff; PNDM: sample_coeff, (abar_prev - abar_t) */
  int noise_mode;  /* 0 none, 1 c_noise*noise, 2 learned variance, 3 learned_range */
  float c_noise;
  float min_log, max_log;
} GmStepParams;
/* sample/prev/x0/noise: [batch][inner]; model_output: [batch][mo_bstride] (mo_bstride = 2*inner for learned variance).
 * x0 and noise may be NULL. */
int gm_sched_step(const void* sample, const void* model_output, const void* noise, void* prev, void* x0,
                  long long batch, long long inner, long long mo_bstride, int dtype, const GmStepParams* p, void* stream);
/* The register-staged attention kernel (fp32; bf16 outside the LDS-DMA kernel's geometries: causal, few keys, head dim 32) deals the key tiles of a work-group to
 * g = 1, 2 or 4 groups of four waves (4: fp32 at head dim <= 64 only, else 2); anything else = by key-tile count (the default: 2 from four tiles, 4 from eight).  Process-wide, for measurements and tests. */
void gm_attention_set_wave_groups(int g);
/* The affine token GEMMs (gm_linear_rows_affine / _vt, gm_linear_rows) over at least `min_rows` rows run with a wave owning 16 rows x nb * 16 output channels
 * (nb = 2, 3 or 4, anything else = chosen by the row count; small_ops.hip: token_gemm_wide_kernel) instead of one 16-channel block per wave.  min_rows 0 = never,
 * < 0 = the default (2048): a process-wide switch for measurements and tests; results are bit-identical either way. */
void gm_token_gemm_set_wide(int min_rows, int nb);
/* out[0..n) = torch.randn(n, dtype=torch.bfloat16) of the CPU generator -- the noise DDPMScheduler.step / DDIMScheduler.step (eta > 0) draw for a bf16 chain
 * (networks/schedulers/ddpm.py:244-248, ddim.py:231-234) -- from that generator's n byte draws: torch's bf16 fill is a function of byte pairs within blocks of 16
 * (generativemodels_amd/host_noise.py).  bits: n bytes (device); table: [256 * 256] (cos branch | sin branch << 16) bf16 pairs (device); n % 16 == 0. */
int gm_normal_bf16_from_bits(const unsigned char* bits, const unsigned int* table, void* out, long long n, void* stream);
/* gm_sched_step (noise_mode != 0) of a bf16 chain with the table lookup of gm_normal_bf16_from_bits inside the step's kernel: `bits` = the batch * inner byte draws
 * of the CPU generator (device), `table` as there; batch * inner a multiple of 16. */
int gm_sched_step_noise_bits(const void* sample, const void* model_output, const unsigned char* bits, const unsigned int* table, void* prev, void* x0,
                             long long batch, long long inner, long long mo_bstride, const GmStepParams* p, void* stream);
/* out = post_mul * (((c0*x0 + c1*x1) + c2*x2) + c3*x3) / post_div over n elements, k = 1..4 terms, left to right with every
 * operation rounded: the Runge-Kutta / linear multi-step combinations of PNDMScheduler.step_prk / step_plms
 * (networks/schedulers/pndm.py:186-195, 241-250).  A NULL x[j] is skipped (the reference's integer-0 accumulator). */
int gm_lincomb(const void* const* x, const float* c, int k, float post_mul, float post_div, void* out, long long n, int dtype,
               void* stream);
/* One term of DiffusionInferer.get_likelihood's variational bound (inferers/inferer.py:203-256; decoder NLL :281-321).
 * x0 = clean inputs, xt = inputs noised to t, model_output = the UNet's prediction for xt; all [batch][inner] (model_output
 * row stride mo_bstride).  Writes the per-element KL / NLL map to kl (may be NULL) and adds its per-sample mean to total[batch]
 * (fp32).  workspace: batch doubles, zero before the first call (left zero on return). */
typedef struct GmKlParams {
  int pred_type;     /* 0 epsilon, 1 sample, 2 v_prediction */
  float c_sa, c_sb;  /* alpha_prod_t ** 0.5, beta_prod_t ** 0.5 */
  int clip;          /* clamp predicted x0 to [-1, 1] */
  float k0, k1;      /* predicted mean = k0 * pred_x0 + k1 * xt */
  float m0, m1;      /* posterior mean = m0 * x0 + m1 * xt (networks/schedulers/ddpm.py:151-154) */
  int t0;            /* 1: t == 0, decoder negative log-likelihood; 0: KL between the two normals */
  float s;           /* KL: (-1 + log_pred_var - log_post_var) + exp(log_post_var - log_pred_var) */
  float e;           /* KL: exp(-log_pred_var); NLL: exp(-log_scales) */
  float half_bin;    /* NLL: bin_width / 2 */
} GmKlParams;
/* workspace: gm_likelihood_workspace_elems(batch, inner) doubles (one partial per sample and block, summed in block order: no atomics,
 * no initialisation needed) */
long long gm_likelihood_workspace_elems(long long batch, long long inner);
int gm_likelihood_term(const void* x0, const void* xt, const void* model_output, void* kl, float* total, double* workspace,
                       long long batch, long long inner, long long mo_bstride, int dtype, const GmKlParams* p, void* stream);
/* out[n,i] = a[n]*x[n,i] + b[n]*y[n,i]: Scheduler.add_noise / get_velocity (networks/schedulers/scheduler.py:169-200) */
int gm_axpby_rows(const void* x, const void* y, const float* a, const float* b, void* out, long long batch,
                  long long inner, int dtype, void* stream);

/* ---- layout plumbing of the NDHWC arena ------------------------------------------------------------------------------ */
int gm_copy_channels(const void* src, long long src_ld, int src_dtype, void* dst, long long dst_ld, int dst_dtype,
                     long long rows, int C, void* stream);            /* torch.cat / casts (diffusion_model_unet.py:1232) */
int gm_nchw_to_nhwc(const void* src, int src_dtype, void* dst, int dst_dtype, int N, int C, long long V, long long dst_ld,
                    void* stream);
int gm_nhwc_to_nchw(const void* src, long long src_ld, int src_dtype, void* dst, int dst_dtype, int N, int C, long long V,
                    void* stream);
/* F.interpolate(x, size, mode="nearest") on an arena tensor (N, Di, Hi, Wi, C) -> (N, Do, Ho, Wo, C): the ControlNet latent inferers
 * resize the conditioning image to the latent grid (inferers/inferer.py:926-927). */
int gm_nearest_resize(const void* x, long long x_ld, void* y, long long y_ld, int N, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                      int C, int dtype, void* stream);
/* mode 0: nearest 2x up, mode 1: 2x average pool (ResnetBlock up/down path, diffusion_model_unet.py:635-639,674-682) */
int gm_resample2x(const void* src, long long src_ld, void* dst, long long dst_ld, int N, int C, int Di, int Hi, int Wi,
                  int act_d, int mode, int dtype, void* stream);
/* dst[n][d][h][w][c] = src[n][2 d + rd][2 h + rh][2 w + rw][c], phase = (rd << 2) | (rh << 1) | rw, extents (X - r + 1) / 2: the 2^d phase
 * images of a stride-2 sub-lattice.  The weight gradient of an EVEN-kernel stride-2 convolution (VQ-VAE k = 4 / s = 2 / p = 1 down- and
 * up-sampling, vqvae.py:127-150,244-261, under torch autograd) is assembled from stride-1 weight gradients over them. */
int gm_phase2x(const void* src, long long src_ld, void* dst, long long dst_ld, int N, int C, int Di, int Hi, int Wi, int act_d, int phase,
               int dtype, void* stream);
/* get_timestep_embedding (diffusion_model_unet.py:461-485): [cos | sin], zero pad if dim is odd */
int gm_timestep_embedding(const float* timesteps, void* out, int B, int dim, float max_period, int dtype, void* stream);
/* MONAI MLPBlock(act="GEGLU") gate (diffusion_model_unet.py:211): out = x[:, :inner] * gelu(x[:, inner:]) */
int gm_geglu(const void* x, long long x_ld, void* out, long long out_ld, long long rows, int inner, int dtype, void* stream);
/* AutoencoderKL.encode/sampling tail (autoencoderkl.py:731-753); eps/z/mu may be NULL when only sigma is wanted */
int gm_aekl_sample(const void* mu, const void* logvar, const void* eps, void* sigma, void* z, long long total, int dtype,
                   void* stream);

/* out = a + b*c element-wise: AutoencoderKL.sampling (autoencoderkl.py:751-752) */
int gm_addcmul(const void* a, const void* b, const void* c, void* out, long long total, int dtype, void* stream);

/* out = x*s (mode 0) or x/s (mode 1): latent scale factor of LatentDiffusionInferer (inferers/inferer.py:386,472) */
int gm_scale(const void* x, void* out, float s, int mode, long long total, int dtype, void* stream);

/* Stand-alone activation over a contiguous tensor (codes = GmConvDesc.post_act) and its backward from the pre-activation: mode 0 out = act(x),
 * mode 1 out = g * act'(x).  The training forward of MONAI Convolution(adn_ordering="DA") layers with a dropout between convolution and
 * activation, and of activations whose derivative needs z (vqvae.py:61-80,127-150; monai Convolution / ADN). */
int gm_activation(const void* x, const void* g, void* out, int act, int mode, long long total, int dtype, void* stream);

/* ---- GroupNorm / LayerNorm (diffusion_model_unet.py:623,643,275,377,1854; autoencoderkl.py:146,156,227,433,579) ------
 * gm_gn_scale_shift reads x once and emits fp32 scale[n][c] = rstd*gamma, shift[n][c] = beta - mean*rstd*gamma that the
 * consumer convolution applies in its prologue (no normalised tensor is ever written). */
long long gm_gn_workspace_bytes(int N, long long V, int C, int G, int dtype);
int gm_gn_scale_shift(const void* x, long long ld, int N, long long V, int C, int G, float eps, const float* gamma,
                      const float* beta, float* scale, float* shift, float* mean, float* rstd, void* workspace, int dtype,
                      void* stream);
/* scale/shift rows are ss_ld floats apart (a channel slice of a wider [N][C_total] table is a valid operand) */
int gm_gn_apply(const void* x, long long x_ld, void* y, long long y_ld, const float* scale, const float* shift, long long ss_ld,
                int N, long long V, int C, int act, int dtype, void* stream);
/* SPADE modulation (generative/networks/blocks/spade_norm.py:79-96): y = act((x * scale[n][c] + shift[n][c]) * g[n][v][c] + bm[n][v][c]);
 * (scale, shift) = the parameter-free GroupNorm / InstanceNorm of x, g = 1 + gamma(seg), bm = beta(seg) in the dtype of x (row pitch gb_ld) */
int gm_spade_apply(const void* x, long long x_ld, void* y, long long y_ld, const float* scale, const float* shift, long long ss_ld,
                   const void* g, const void* bm, long long gb_ld, int N, long long V, int C, int act, int dtype, void* stream);
/* Composable form: per-channel {sum, sum of squares} partials in fp64, [S][N][C][2] with one partial per block of rows / per output tile
 * (S = gm_gn_channel_stats_slots for this entry point, gm_conv_stats_slots for a convolution epilogue writing GmConvDesc.stats).  Every
 * entry is STORED exactly once (no atomics, no zero fill) and the consumers add the S partials in a fixed order, so GroupNorm -- and with
 * it a whole sampling chain -- is bit-reproducible run to run.  gm_gn_finalize_channels is the GroupNorm finalisation over up to two
 * channel-concatenated sources (S0 / S1 partials each) -- torch.cat([h, skip]) followed by GroupNorm (diffusion_model_unet.py:1232 + 671)
 * without ever materialising the concatenation. */
long long gm_gn_channel_stats_slots(const void* x, long long ld, long long V, int C, int dtype);
/* fold a long table [S][N][C][2] (S >= gm_stats_compact_slots() = 256: one partial per tile of a large volume) to
 * [gm_stats_compact_slots()][N][C][2] in a fixed order (output row b = input rows b, b + 256, ...), so that the per-group finalisation
 * reads a small table */
int gm_stats_compact_slots(void);
int gm_stats_compact(const double* stats_in, int S, int N, int C, double* stats_out, void* stream);
int gm_gn_channel_stats(const void* x, long long ld, int N, long long V, int C, double* chan_out, int dtype, void* stream);
int gm_gn_finalize_channels(const double* stats0, int S0, int C0, const double* stats1, int S1, int C1, int N, long long V, int G, float eps,
                            const float* gamma, const float* beta, float* scale, float* shift, void* stream);
int gm_layernorm(const void* x, long long x_ld, void* y, long long y_ld, const float* gamma, const float* beta,
                 long long rows, int C, float eps, int dtype, void* stream);

/* ---- convolution / linear as LDS-staged implicit GEMM on MFMA ---------------------------------------------------------
 * Replaces MONAI Convolution(conv_only=True) (nn.ConvNd / nn.ConvTransposeNd) and nn.Linear, fused with
 * GroupNorm-apply + SiLU (prologue), nearest-2x Upsample (diffusion_model_unet.py:572-585), bias, timestep-embedding add
 * (:686-690), residual add (:692-696) and the output activation. */
typedef struct GmConvDesc {
  const void* x; long long x_ld;
  const void* w;             /* packed by gm_pack_conv_weight; two configurations read their own image instead: cfg 22 the HALVES image of that
                              * panel, [chunk32][half][tap][Cout_pad][16] (the 16-input-channel weight block of a tap contiguous); cfg 12
                              * (C_in <= 4) the K-MAJOR image [Cout padded to 64][27 * C_in padded to 32 bf16 / 16 fp32], k = tap * C_in + ci */
  const float* bias;         /* [Cout] or NULL */
  const float* pre_scale;    /* [N][Cin] or NULL */
  const float* pre_shift;    /* [N][Cin] or NULL */
  const float* rowvec;       /* [B][Cout] fp32 or NULL: added per (n, cout) */
  long long rowvec_bstride;  /* 0: one row broadcast over the batch */
  const void* res; long long res_ld;
  void* y; long long y_ld;
  int N, Cin, Cout;
  int Ds, Hs, Ws;
  int Do, Ho, Wo;
  int kd, kh, kw;
  int sd, sh, sw;
  int pd, ph, pw;            /* low-side padding */
  int dd, dh, dw;            /* dilation */
  int in_mode;               /* 0 direct, 1 nearest up-sample, 2 zero insertion (transposed conv), 3 nearest 2x up-sample + 3x3x3 kernel
                              * evaluated as 8 sub-pixel 2x2x2 kernels on the low-resolution input: kd = kh = kw = 2, w = the 8 parity
                              * images (d, h, w parity = bits 2, 1, 0) of pre-summed weights, Do/Ho/Wo = 2 x Ds/Hs/Ws, cfg = 17 */
  int fd, fh, fw;
  int pre_act;               /* 0 none, 1 SiLU, 2 ReLU */
  int post_act;              /* 0 none, 1 ReLU, 2 tanh, 3 sigmoid, 4 SiLU, 5 LeakyReLU(0.01), 6 GELU (erf) */
  int dtype;
  int ltd, lth, ltw;         /* log2 output tile dims; product must equal the configuration's voxel count */
  int cfg;                   /* tile configuration, see gm_conv_cfg_tile */
  int debug_flags;           /* must be 0 (bench-only ablation switches: results are wrong when set) */
  double* stats;             /* optional [gm_conv_stats_slots(desc)][N][Cout][2]: per-tile partials of the per-channel sum / sum of squares
                                of the stored output for the next GroupNorm (plain stores, one per tile and channel: no zero fill needed);
                                must be NULL when gm_conv_stats_slots() returns 0 for the chosen configuration */
  /* optional fused 1x1 shortcut convolution of a ResnetBlock (diffusion_model_unet.py:684-696, autoencoderkl.py:188-193):
   * y += W_skip * cat(skip_x[0], skip_x[1]) + skip_bias with the sources in the output geometry (skip_x[1] may be NULL).
   * Implemented by the LDS-DMA configurations (cfg 11, 14, 16, 18, 19): gm_conv_lds_bytes() returns -1 for any other when skip_x[0] is set. */
  const void* skip_x[2]; long long skip_ld[2]; int skip_cin[2];
  const void* skip_w;        /* gm_pack_conv_weight image of the [Cout][skip_cin[0]+skip_cin[1]] 1x1 kernel */
  const float* skip_bias;    /* [Cout] or NULL */
  /* optional second input source (LDS-DMA configurations only): the convolution input is cat(x[..., :cin_split], x2) along the channels
   * -- the decoder's torch.cat([h, skip], dim=1) (diffusion_model_unet.py:1232,1340,1461), never materialised; x2 in x's geometry,
   * cin_split a multiple of the 64-byte K chunk (32 bf16 / 16 fp32 channels); Cin counts both parts */
  const void* x2; long long x2_ld; int cin_split;
  /* optional split-K for small grids (LDS-DMA 3x3x3 stride-1 configurations): a convolution over a 32^3 .. 8^3 latent has fewer tiles
   * than the chip has CUs and every tile is a serial chain of K chunks (exposed LDS-DMA round trips), so the chunks are dealt to
   * `ksplit` work-groups per tile that write fp32 partial sums to `kpartial` (gm_conv_splitk_workspace_bytes) and a combine kernel
   * applies the epilogue (bias, timestep row, residual, activation, output statistics).  ksplit <= 1 or kpartial NULL: off. */
  int ksplit; float* kpartial;
  /* optional GroupNorm prologue given as STATISTICS instead of (pre_scale, pre_shift) -- tile configurations 24 / 25 only (conv_sn.hip): the consumer folds the
   * per-tile partials of its input (up to two channel-concatenated sources, S_i <= 64 rows of [N][C_i][2] fp64 each, exactly what gm_gn_finalize_channels takes)
   * and forms scale = rstd * gamma, shift = beta - mean * rstd * gamma in its prologue: bit-identical to gm_gn_finalize_channels (reference nn.GroupNorm,
   * diffusion_model_unet.py:671-684), one launch less per norm.  pre_stats[0] == NULL: off. */
  const double* pre_stats[2]; int pre_S[2]; int pre_C[2];
  const float* pre_gamma; const float* pre_beta;
  float pre_eps; int pre_groups;
} GmConvDesc;
int gm_conv_cfg_tile(int cfg, int* voxels, int* channels);
long long gm_conv_lds_bytes(const GmConvDesc* d);
/* partials S a launch writes into GmConvDesc.stats ([S][N][Cout][2]); 0 = this configuration does not fuse the output statistics */
long long gm_conv_stats_slots(const GmConvDesc* d);
/* bytes of GmConvDesc.kpartial for this descriptor (ksplit set); 0 when the configuration does not support split-K */
long long gm_conv_splitk_workspace_bytes(const GmConvDesc* d);
int gm_conv_forward(const GmConvDesc* d, void* stream);
/* Grid policy of the LDS-DMA configurations (process-wide; results do not depend on it).  0 (default): one work-group per tile.  -1: the launch
 * holds at most as many work-groups as the device runs at once and every work-group walks several tiles, requesting the next tile's first input
 * patch before the epilogue of the current one (measured 1 % slower on MI355X: DESIGN.md 4.1).  n > 0: at most n work-groups (tests: forces
 * the multi-tile walk on small inputs). */
void gm_conv_dma_set_persistent(int max_work_groups);
/* One-time phase offset between the work-groups that share a CU in the LDS-DMA configurations (process-wide; results do not depend on it): the
 * co-resident work-groups of a CU are dispatched together and run equal-length tiles, i.e. in lock step -- both in the tap loop, then both in
 * the epilogue with the MFMA pipe idle.  The work-groups of the first residency round that sit in an odd work-group slot of their CU sleep
 * `cycles` once; later work-groups inherit the phase of their slot.  0 = off (default: measured no gain on MI355X, profiles/r05_phase_skew_sweep.txt),
 * -1 = half the modelled tile life of the launch. */
void gm_conv_dma_set_phase_skew(int cycles);
/* Kernel of the K slices of a split-K launch (process-wide; results do not depend on it: the partial sums are bit-identical): 1 (default) =
 * conv_sk.hip (one work-group per CU, the patch and all nine weight panels of a K chunk requested up front); 0 = the general cfg 11 tile kernel
 * (the round-3 path; A/B measurements and the bitwise test). */
void gm_conv_sk_set_enabled(int on);
long long gm_packed_conv_weight_elems(int Cout, int Cin, int kd, int kh, int kw, int dtype);
/* src: [Cout][Cin][kd][kh][kw] (transposed = 0) or [Cin][Cout][kd][kh][kw] (transposed = 1, nn.ConvTransposeNd) */
int gm_pack_conv_weight(const void* src, int src_dtype, void* dst, int dst_dtype, int Cout, int Cin, int kd, int kh, int kw,
                        int transposed, void* stream);
/* The 8 parity images of 2x2x2 kernels of a sub-pixel convolution (GmConvDesc.in_mode 3), packed back to back (8 x gm_packed_conv_weight_elems(
 * Cout, Cin, 2, 2, 2)), in one launch from the parameter: sub-tap s of parity p along an axis = the sum of the source taps in bit mask
 * m<p><s> (bit k = tap k of the K-tap kernel).  swap_io = 0: src [Cout][Cin][K][K][K] (Upsample = nearest 2x + 3x3x3 convolution,
 * diffusion_model_unet.py:572-585: masks 1, 6, 3, 4); swap_io = 1: src [Cin][Cout][K][K][K] (stride-2 transposed convolutions and the data
 * gradient of stride-2 convolutions: one tap or none per sub-tap, vqvae.py:244-261, autoencoderkl.py:54-63). */
int gm_pack_subpixel_weight(const void* src, int src_dtype, void* dst, int dst_dtype, int Cout, int Cin, int K, int swap_io, int m00, int m01,
                            int m10, int m11, void* stream);

/* ---- attention (diffusion_model_unet.py:143-153,407-415; autoencoderkl.py:261-269) ------------------------------------
 * O = softmax(scale * Q K^T) V (+ residual) per (batch, head); q/k/v/o rows are tokens, head h at element offset h*dh. */
typedef struct GmAttnDesc {
  const void* q; long long q_ld;
  const void* k; long long k_ld;
  const void* v; long long v_ld;
  const void* res; long long res_ld;
  void* o; long long o_ld;
  int B, H, Lq, Lk, dh;
  float scale;
  int dtype;
  void* workspace;            /* optional scratch of gm_attention_workspace_bytes(): enables the LDS-DMA kernel (bf16, dh 64/128/256) */
  long long workspace_bytes;
  int causal;                 /* 1: query i sees keys j <= i + (Lk - Lq) (causal SABlock, blocks/selfattention.py:133-134) */
  long long k_bs, v_bs;       /* batch strides of k / v in elements, 0 = dense (Lk * ld): a KV cache [B][max_len][C] read up to Lk */
  double* stats;              /* optional [gm_attention_stats_slots(desc)][B][H * dh][2]: per-block partials of the per-channel sum / sum of squares of
                               * the stored output (+ residual) for the GroupNorm that follows an attention block (plain stores, one per block and
                               * channel); must be NULL when gm_attention_stats_slots() returns 0 for this geometry */
  int vt_packed;              /* 1: `workspace` already holds the transposed, key-permuted V image of the LDS-DMA kernel -- the stacked q | k | v projection
                               * wrote it (gm_linear_rows_affine_vt; Lk a multiple of 64): the pack launch is skipped.  Only with a workspace of
                               * gm_attention_workspace_bytes() > 0 bytes */
  float* lse;                 /* optional fp32 [B*H][Lq]: log sum_k exp(scale q.k) per query, written by the LDS-DMA path only (must be NULL otherwise);
                               * the training forward keeps it for gm_attention_backward_fused */
} GmAttnDesc;
int gm_attention_max_head_dim(void);
/* bytes of scratch the fastest kernel for this geometry wants (0: none; the descriptor's workspace fields are not read) */
long long gm_attention_workspace_bytes(const GmAttnDesc* d);
/* partials S the launch writes into GmAttnDesc.stats; 0 = this geometry does not fuse the output statistics (the split-KV form of the LDS-DMA
 * kernel with one head does: its merge kernel holds every output row anyway) */
long long gm_attention_stats_slots(const GmAttnDesc* d);
/* Tests / benchmarks only: pin the LDS-DMA kernel variant -- queries per wave = 16 * qf (qf 1 or 2) and the number of key slices of the
 * split-KV form (1..8) -- instead of the size-based choice; 0 restores the automatic choice for that knob.  Process-wide. */
void gm_attention_dma_set_variant(int qf, int nsplit);
int gm_attention_forward(const GmAttnDesc* d, void* stream);

/* ---- autoregressive transformer helpers (networks/nets/transformer.py, inferers/inferer.py:1126-1330) ------------------------ */
/* out[b][t][:] = token_weight[indices[b][t]] + position_weight[pos0 + t]  (transformer.py:99-101) */
int gm_embed_tokens(const long long* indices, const void* token_weight, const void* position_weight, void* out, long long batch,
                    int seq_len, int C, int pos0, int num_tokens, int max_positions, int dtype, void* stream);
/* sampling head (inferer.py:1221-1232): probs = softmax(crop_topk(logits / temperature)), probs[:, bos_index] = 0; top_k <= 0: no crop.
 * logits: [rows][ld >= V] in dtype, probs: fp32 [rows][V] */
int gm_sample_probs(const void* logits, long long ld, float* probs, long long rows, int V, float temperature, int top_k, int bos_index,
                    int dtype, void* stream);
/* one categorical draw per row by inverse CDF: out[row] = min{ j : sum_{i<=j} probs[row][i] >= u[row] * sum_i probs[row][i] } with u
 * uniform in [0, 1) from the caller's generator (the draw of inferer.py:1235 without torch.multinomial's device -> host validation) */
int gm_sample_index(const float* probs, long long rows, int V, const float* u, long long* out, void* stream);
/* out[row] = log(softmax(logits[row])[target[row]])  (inferer.py:1290-1296, 1316) */
int gm_token_log_prob(const void* logits, long long ld, const long long* target, float* out, long long rows, int V, int dtype,
                      void* stream);

/* y[rows][cout] = post_act(pre_act(x)[rows][cin] W^T + bias) (+ res) for a handful of rows (decode steps, the timestep MLP): one wave
 * per 16 output channels, weights global -> registers -> MFMA, no LDS.  w: gm_pack_conv_weight image of the [cout][cin] matrix;
 * pre_act / post_act as in GmConvDesc; cin and x_ld multiples of 16 bytes. */
int gm_linear_rows(const void* x, long long x_ld, const void* w, const float* bias, const void* res, long long res_ld, void* y,
                   long long y_ld, int rows, int cin, int cout, int pre_act, int post_act, int dtype, void* stream);
/* The same over token rows with a per-sample GroupNorm affine in front: y = post_act(pre_act(x * scale[n] + shift[n]) W^T + bias) (+ res),
 * n = row / rows_per_sample, scale / shift fp32 [N][ss_ld] (both null: no affine).  The GroupNorm -> q | k | v projection of the
 * latent-resolution AttentionBlocks (diffusion_model_unet.py:395-405) and the other 1x1 convolutions over a few thousand tokens. */
/* gm_linear_rows_affine as the stacked q | k | v projection of an attention block (diffusion_model_unet.py:407-415): the V columns (output channels
 * >= vt_c0, heads of vt_dh channels) are also stored as the V^T image of the LDS-DMA attention kernel at the head of its workspace; bf16,
 * rows_per_sample (tokens per sample) a multiple of 64 */
int gm_linear_rows_affine_vt(const void* x, long long x_ld, const float* pre_scale, const float* pre_shift, long long ss_ld, int rows_per_sample,
                             const void* w, const float* bias, void* y, long long y_ld, int rows, int cin, int cout, int pre_act, void* vt,
                             int vt_c0, int vt_dh, int dtype, void* stream);
int gm_linear_rows_affine(const void* x, long long x_ld, const float* pre_scale, const float* pre_shift, long long ss_ld, int rows_per_sample,
                          const void* w, const float* bias, const void* res, long long res_ld, void* y, long long y_ld, int rows, int cin,
                          int cout, int pre_act, int post_act, int dtype, void* stream);
/* The same GEMM with the GroupNorm given as the per-tile statistic tables of x's producer(s) instead of (scale, shift): finalised in the GEMM's prologue (the
 * short-table order of gm_gn_finalize_channels: bit-identical to that launch + gm_linear_rows_affine), so an attention block's norm costs no launch of its own
 * (diffusion_model_unet.py:424-441).  stats[i]: [S_i][N][C_i][2] fp64 (sum, sum of squares), S_i <= 128, stats[1] NULL for one source; gamma / beta fp32 over
 * the C0 + C1 = cin <= 384 channels (nullable); rows = n_samples * rows_per_sample, rows_per_sample a multiple of 64; cin a multiple of the MFMA K step (32 bf16 /
 * 16 fp32).  vt != NULL: the V^T image as gm_linear_rows_affine_vt (bf16, no residual). */
typedef struct GmGnTables {
  const double* stats[2]; int S[2]; int C[2];
  const float* gamma; const float* beta; float eps; int groups;
} GmGnTables;
int gm_linear_rows_gn(const void* x, long long x_ld, const GmGnTables* gn, int n_samples, int rows_per_sample, const void* w, const float* bias,
                      const void* res, long long res_ld, void* y, long long y_ld, int rows, int cin, int cout, int pre_act, int post_act, void* vt,
                      int vt_c0, int vt_dh, int dtype, void* stream);

/* One KV-cache decoding step of the decoder-only transformer issued natively (38-62 launches back to back, by how many of the fused kernels of small_ops.hip take the geometry): embed the fed token at
 * `pos`, per block LayerNorm -> q|k|v -> append k, v to the caches -> 1 x (pos+1) attention -> out_proj + x -> LayerNorm -> MLP(GELU) + x,
 * then to_logits (networks/nets/transformer.py:98-106, blocks/transformerblock.py:86-91, blocks/selfattention.py:98-147; no cross
 * attention).  Weights are gm_pack_conv_weight images of the nn.Linear matrices (q, k, v stacked along the output dim). */
typedef struct GmDecodeBlock {
  const float *ln1_g, *ln1_b;
  const void* w_qkv; const float* b_qkv;
  const void* w_o; const float* b_o;
  const float *ln3_g, *ln3_b;
  const void* w_1; const float* b_1;
  const void* w_2; const float* b_2;
  void* k_cache; void* v_cache;      /* [B][max_len][C] */
} GmDecodeBlock;
typedef struct GmDecodeDesc {
  int B, C, M, heads, depth, max_len, num_tokens, dtype;
  float ln_eps;
  int pos;
  const long long* tokens;           /* [B] device */
  const void* tok_emb; const void* pos_emb;
  const GmDecodeBlock* blocks;       /* host array, `depth` entries */
  const void* w_logits; const float* b_logits;
  void* logits;                      /* [B][num_tokens], dtype */
  void* scratch; long long scratch_bytes;
  const int* pos_dev;                /* optional: position read from device memory at run time (for HIP-graph replay); `pos` is then ignored */
} GmDecodeDesc;
long long gm_decode_scratch_bytes(int B, int C, int M, int dtype);
int gm_transformer_decode_step(const GmDecodeDesc* d, void* stream);
/* after a draw: seq[b][*pos + 1] = idx[b]; tokens[b] = idx[b]; *pos += 1 (inferer.py:1237-1239 kept on the device) */
int gm_decode_advance(int* pos, long long* tokens, const long long* idx, long long* seq, int B, long long seq_ld, void* stream);

/* ---- backward kernels (the reference trains through torch autograd:
 * tutorials/generative/distributed_training/ddpm_training_ddp.py:249-270, generative/engines/trainer.py:258-270) -------------------------
 * The data gradient of a convolution is gm_conv_forward with in_mode = 2 (transposed convolution of gy with the same weight). */
typedef struct GmWgradDesc {
  const void* x; long long x_ld;      /* forward input  [N][Ds][Hs][Ws][Cin] */
  const void* gy; long long gy_ld;    /* output gradient [N][Do][Ho][Wo][Cout] */
  float* dw;                          /* fp32 [Cout][Cin][kd][kh][kw]: the nn.ConvNd / nn.Linear weight layout */
  void* workspace; long long workspace_bytes;  /* gm_conv_wgrad_workspace_bytes(d) */
  int N, Cin, Cout, Ds, Hs, Ws, Do, Ho, Wo;
  int kd, kh, kw;                     /* 1 or 3 per axis, kh == kw */
  int stride;                         /* 1 or 2 on every axis */
  int pd, ph, pw;                     /* low-side padding */
  int dtype;                          /* of x and gy */
  int accumulate;                     /* 0: dw is overwritten, 1: added to */
} GmWgradDesc;
/* -1: geometry not covered */
long long gm_conv_wgrad_workspace_bytes(const GmWgradDesc* d);
/* dW[co][ci][tap] = sum_v gy[v][co] * x[v * stride - pad + tap][ci] (torch.nn.grad.conv*_weight); split-K partial sums reduced in
 * a fixed order: deterministic */
int gm_conv_wgrad(const GmWgradDesc* d, void* stream);
/* GroupNorm (+ SiLU when act = 1, ReLU when act = 2, LeakyReLU(0.01) when act = 3) backward for y = act(x * scale[n][c] + shift[n][c]) (nn.GroupNorm + nn.SiLU,
 * diffusion_model_unet.py:623-690).  With g = gy * act'(x * scale + shift):
 *   gm_gn_bwd_stats     out[block][n][c] = {sum_v g, sum_v g x} over the rows of the block: fp64 [gm_gn_bwd_stats_slots(N, V)][N][C][2],
 *                       one plain store each (no atomics, no zero fill; summed in a fixed order by gm_gn_bwd_finalize: bit-reproducible)
 *   gm_gn_bwd_finalize  per-(n, c) coefficients A, B, Cc of dx = A g + B x + Cc, and dgamma[c], dbeta[c] (nullable);
 *                       fwd_stats = the [fwd_slots][N][C][2] forward table of x (gm_gn_channel_stats or a convolution epilogue)
 *   gm_gn_bwd_apply     dx = g * A + x * B + Cc */
int gm_gn_bwd_stats(const void* x, long long x_ld, const void* gy, long long gy_ld, const float* scale, const float* shift, long long ss_ld,
                    int N, long long V, int C, int act, double* out, int dtype, void* stream);
long long gm_gn_bwd_stats_slots(int N, long long V);
int gm_gn_bwd_finalize(const double* fwd_stats, int fwd_slots, const double* bwd_stats, int bwd_slots, int N, int C, int G, long long V, float eps,
                       const float* gamma, float* A, float* B, float* Cc, float* dgamma, float* dbeta, void* stream);
int gm_gn_bwd_apply(const void* x, long long x_ld, const void* gy, long long gy_ld, void* dx, long long dx_ld, const float* scale,
                    const float* shift, long long ss_ld, const float* A, const float* B, const float* Cc, int N, long long V, int C, int act,
                    int dtype, void* stream);
/* SPADE modulation backward (blocks/spade_norm.py:79-96 under torch autograd): y = act(xn * g + bm); with gu = gy * act'(xn * g + bm):
 * dxn = gu * g, dg = gu * xn, dbm = gu (act: 0 none, 1 SiLU); g / bm share the row pitch gb_ld, dg / dbm share dgb_ld */
int gm_spade_bwd(const void* xn, long long x_ld, const void* g, const void* bm, long long gb_ld, const void* gy, long long gy_ld, void* dxn,
                 long long dx_ld, void* dg, void* dbm, long long dgb_ld, long long rows, int C, int act, int dtype, void* stream);
/* per_sample = 0: out[c] = sum over slots and samples of stats[slot][n][c][0] (a bias gradient from the gm_gn_channel_stats table of
 * gy); per_sample = 1: out[n][c] = sum over slots (gradient of the per-sample row vector a convolution epilogue adds: the timestep
 * embedding projection, diffusion_model_unet.py:684-686) */
int gm_stats_colsum(const double* stats, int slots, int N, int C, float* out, int per_sample, void* stream);

/* Flash-attention backward: dq, dk, dv of o = softmax(scale q k^T) v per (batch, head) (torch autograd through
 * diffusion_model_unet.py:143-153 / 407-415); the L x L scores are recomputed tile by tile, never stored.  o = the forward output
 * WITHOUT the residual, go = its gradient.  head dim in {16, 32, 64, 128, 256}; fp32 accumulation and fp32 MFMA products for every
 * storage dtype; deterministic (no atomics). */
typedef struct GmAttnBwdDesc {
  const void* q; long long q_ld;
  const void* k; long long k_ld;
  const void* v; long long v_ld;
  const void* o; long long o_ld;
  const void* go; long long go_ld;
  void* dq; long long dq_ld;
  void* dk; long long dk_ld;
  void* dv; long long dv_ld;
  int B, H, Lq, Lk, dh;
  float scale;
  int dtype;
  void* workspace; long long workspace_bytes;  /* gm_attention_backward_workspace_bytes(d): LSE and dO.O per query */
} GmAttnBwdDesc;
long long gm_attention_backward_workspace_bytes(const GmAttnBwdDesc* d);
int gm_attention_backward(const GmAttnBwdDesc* d, void* stream);
/* bf16-MFMA score pass of the attention backward (same autograd node as above; diffusion_model_unet.py:407-415): probs = softmax(scale Q K^T)
 * and dscores = probs (dO V^T - rowsum(dO O)) scale per (sample, head), bf16 [B*H][Lq][pd_ld] with pd_ld >= Lk rounded up to 64, plus
 * dscores_t = dscores^T as [B*H][Lk rounded up to 64][st_ld], st_ld >= Lq rounded up to 64 (padding written as zeros); bf16 operands, head
 * size 32 / 64 / 128 / 256; uses q, k, v, o, go, geometry, scale and workspace (gm_attention_bwd_scores_workspace_bytes) of the descriptor.  The row contractions dV = P^T dO,
 * dK = dS^T Q, dQ = (dS^T)^T K then run on gm_conv_wgrad (generativemodels_amd/ops.py: attention_backward_bf16). */
long long gm_attention_bwd_scores_workspace_bytes(const GmAttnBwdDesc* d);
int gm_attention_bwd_scores(const GmAttnBwdDesc* d, void* probs, void* dscores, long long pd_ld, void* dscores_t, long long st_ld, void* stream);
/* Fused bf16 flash backward of the same autograd node (generativemodels_amd/csrc/attention_bwd_dma.hip; diffusion_model_unet.py:407-415 under
 * torch autograd): dq / dk / dv written once as bf16, nothing L x L in HBM, deterministic.  bf16 operands, head size 64 / 128 / 256, 16-byte
 * aligned rows; `lse` = log sum_k exp(scale q.k) per (sample, head, query), fp32 [B*H][Lq], or NULL (one more sweep computes it).
 * gm_attention_backward_fused_workspace_bytes is 0 for operands this path does not take. */
long long gm_attention_backward_fused_workspace_bytes(const GmAttnBwdDesc* d);
int gm_attention_backward_fused(const GmAttnBwdDesc* d, const float* lse, void* stream);
/* slices of the streamed rows per sweep (fp32 partial results added in slice order): 0 = by problem size (until every CU has a work-group), 1..16 forced */
void gm_attention_backward_fused_set_split(int nsplit);
/* dscores = scale * probs * (dprobs - rowsum(dprobs * probs)): softmax backward of the attention scores scale * Q K^T, fp32 [rows][V]
 * (the softmax of diffusion_model_unet.py:143-153 / 407-415 under torch autograd) */
int gm_softmax_bwd(const float* probs, const float* dprobs, float* dscores, long long rows, int V, float scale, void* stream);

/* nn.LayerNorm backward (diffusion_model_unet.py:219-223 under autograd): dx per row; param_stats[block][c] = {sum_rows gy xhat, sum_rows gy}
 * over the rows a block walks (fp64 [gm_layernorm_bwd_slots(rows)][C][2], plain stores -- no atomics, nothing to zero; nullable):
 * dgamma / dbeta = gm_stats_colsum over the two components, a fixed-order sum (bit-reproducible) */
int gm_layernorm_bwd_slots(long long rows);
int gm_layernorm_bwd(const void* x, long long x_ld, const void* gy, long long gy_ld, void* dx, long long dx_ld, const float* gamma,
                     long long rows, int C, float eps, double* param_stats, int dtype, void* stream);
/* GEGLU backward: x = [a | gate] (2 * inner channels), y = a * gelu(gate) -> dx = [gy gelu(gate) | gy a gelu'(gate)] */
int gm_geglu_bwd(const void* x, long long x_ld, const void* gy, long long gy_ld, void* dx, long long dx_ld, long long rows, int inner, int dtype,
                 void* stream);

/* ---- vector quantiser (networks/layers/vector_quantizer.py:86-138,183) ------------------------------------------------ */
int gm_vq_argmin(const void* x, long long x_ld, const float* embedding, long long* indices, long long tokens,
                 int num_embeddings, int dim, int dtype, void* stream);
/* EMA codebook update of EMAQuantizer.forward in training mode (vector_quantizer.py:166-180).  gm_vq_ema_stats fills ONE flat fp32 buffer
 * stats[K + K*D] = (tokens per code | sum of the token vectors per code) -- the reference's `encodings_sum` and `dw`, exchanged between
 * data-parallel ranks by ONE all-reduce instead of two (:155-157); gm_vq_ema_update applies the decayed update, the Laplace smoothing and
 * rewrites the embedding (fp32 buffers, in place). */
/* workspace: gm_vq_ema_stats_workspace_elems(tokens, K, D) floats (per-token-range partial tables, summed in range order: every index and
 * every vector is read once, no atomics) */
long long gm_vq_ema_stats_workspace_elems(long long tokens, int num_embeddings, int dim);
int gm_vq_ema_stats(const void* x, long long x_ld, const long long* indices, long long tokens, int num_embeddings, int dim, float* stats,
                    float* workspace, int dtype, void* stream);
int gm_vq_ema_update(const float* stats, float* cluster, float* ema_w, float* embedding, int num_embeddings, int dim, float decay,
                     float epsilon, void* stream);
long long gm_vq_gather_workspace_bytes(void);
int gm_vq_gather(const long long* indices, const float* embedding, void* out, long long out_ld, const void* x,
                 long long x_ld, float* sq_err_mean, void* workspace, long long tokens, int num_embeddings, int dim,
                 int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GM_AMD_H */
