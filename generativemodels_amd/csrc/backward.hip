// Backward kernels of the fused forward ops (what a training step -- reference
// tutorials/generative/distributed_training/ddpm_training_ddp.py:249-270, engines/trainer.py:258-270 -- needs beyond the forward path):
//   gm_conv_wgrad          dW[co][ci][tap] = sum_v gy[v][co] * x[v * s - p + tap][ci]          (nn.ConvNd weight gradient)
//   gm_gn_bwd_stats / gm_gn_bwd_finalize / gm_gn_bwd_apply      GroupNorm (+ SiLU) backward     (nn.GroupNorm + nn.SiLU)
//   gm_stats_colsum        bias gradient from the per-channel statistics of gy
// The data gradient of a convolution needs no kernel of its own: it is the transposed convolution of gy with the same weight
// (ops.conv(transposed=True)), i.e. the forward LDS-DMA kernel at stride 1.
//
// Weight gradient on MFMA.  The contraction runs over VOXELS, which is the row index of both N[D]HWC operands, while an MFMA
// lane wants 8 consecutive k values (4 for fp32) in one 16-byte register group.  Both tiles are therefore transposed once, on
// their way into LDS: gyT[co][tile voxel] and xT[ci][halo-patch voxel], W innermost.  A k-step is then one 32-voxel W run:
//   A fragment = gyT[co = l15][row][8q .. 8q+7]                        one aligned ds_read_b128
//   B fragment = xT[ci = l15][row + kh][8q + kw .. 8q + kw + 7]        an UNALIGNED 8-element run: read the aligned 16 bytes plus
//                the next dword and funnel-shift in registers (kw = 1: four v_alignbyte_b32, kw = 2: register renaming), so
//                the three kw taps of a row share one read and every read stays naturally aligned (a misaligned b128 replays
//                at 64 cycles, cdna_hip_programming.md 6).  Stride 2 stores even and odd W columns as separate runs.
// A work-group owns one depth tap kd, a 64-channel block of C_out and of C_in (32 for fp32) and all 9 (kh, kw) taps -- 72
// accumulator VGPRs per lane --, walks a share of the voxel tiles (split K) and writes its partial sums; a second kernel
// reduces the splits in a fixed order (deterministic, no atomics) into the nn.ConvNd weight layout.
#include "conv_common.h"

struct GmWgradDesc {
  const void* x; long long x_ld;      // [N][Ds][Hs][Ws][Cin]
  const void* gy; long long gy_ld;    // [N][Do][Ho][Wo][Cout]
  float* dw;                          // fp32 [Cout][Cin][kd][kh][kw]
  void* workspace; long long workspace_bytes;
  int N, Cin, Cout, Ds, Hs, Ws, Do, Ho, Wo;
  int kd, kh, kw;                     // 1 or 3 per axis; kh == kw
  int stride;                         // 1 or 2, every axis (depth too when kd == 3)
  int pd, ph, pw;                     // low-side padding
  int dtype;                          // of x and gy
  int accumulate;                     // 0: dw is overwritten, 1: added to
};

static constexpr int wg_pad_pitch(int bytes) { return ((bytes - 16 + 255) / 256) * 256 + 16; }  // smallest >= bytes that is 16 mod 256

template <typename T> struct WgTraits;
template <> struct WgTraits<bf16_raw> { static constexpr int CIB = 64; };
template <> struct WgTraits<float> { static constexpr int CIB = 32; };

// the B fragments of the KHW taps of one patch row, from aligned reads
template <typename T, int S, int KHW> struct WgShift;
template <int S, int KHW> struct WgShift<bf16_raw, S, KHW> {
  static __device__ __forceinline__ void run(const char* rowp, uint4 (&b)[KHW]) {  // rowp: this lane's aligned 16-byte group
    if constexpr (KHW == 1) { b[0] = *reinterpret_cast<const uint4*>(rowp); } else {
    const uint4 v = *reinterpret_cast<const uint4*>(rowp);
    const uint32_t d4 = *reinterpret_cast<const uint32_t*>(rowp + 16);
    const uint4 sh1 = make_uint4(__builtin_amdgcn_alignbyte(v.y, v.x, 2), __builtin_amdgcn_alignbyte(v.z, v.y, 2),
                                 __builtin_amdgcn_alignbyte(v.w, v.z, 2), __builtin_amdgcn_alignbyte(d4, v.w, 2));
    if constexpr (S == 1) {
      b[0] = v;
      b[1] = sh1;
      b[KHW - 1] = make_uint4(v.y, v.z, v.w, d4);
    } else {  // even plane: kw = 0 -> index w, kw = 2 -> index w + 1; odd plane (40 elements further): kw = 1 -> index w
      b[0] = v;
      b[KHW - 1] = sh1;
      b[1] = *reinterpret_cast<const uint4*>(rowp + 40 * 2);
    }
    }
  }
};
template <int S, int KHW> struct WgShift<float, S, KHW> {
  static __device__ __forceinline__ void run(const char* rowp, uint4 (&b)[KHW]) {
    if constexpr (KHW == 1) { b[0] = *reinterpret_cast<const uint4*>(rowp); } else {
    const uint4 v = *reinterpret_cast<const uint4*>(rowp);
    const uint2 e = *reinterpret_cast<const uint2*>(rowp + 16);
    if constexpr (S == 1) {
      b[0] = v;
      b[1] = make_uint4(v.y, v.z, v.w, e.x);
      b[KHW - 1] = make_uint4(v.z, v.w, e.x, e.y);
    } else {
      b[0] = v;
      b[KHW - 1] = make_uint4(v.y, v.z, v.w, e.x);
      b[1] = *reinterpret_cast<const uint4*>(rowp + 40 * 4);
    }
    }
  }
};

// GS x GS transpose across GS consecutive lanes (GS = elements per 16-byte vector): in, lane v holds elements (channels) 0 .. GS-1 of
// voxel v; out, lane j holds voxels 0 .. GS-1 of channel j.  Butterfly: swap lane bit k with register-index bit k, one exchange each.
// lane ^ 1 and lane ^ 2 are DPP quad permutes (VALU, no LDS traffic), lane ^ 4 a ds_swizzle; __shfl_xor compiled to ds_bpermute_b32.
__device__ __forceinline__ uint32_t wg_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }  // quad_perm [1,0,3,2]
__device__ __forceinline__ uint32_t wg_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true); }  // quad_perm [2,3,0,1]
__device__ __forceinline__ uint32_t wg_xor4(uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x101F); }           // and 0x1f, xor 4
template <typename T> __device__ __forceinline__ uint4 wg_transpose(uint4 r, int lane);
template <> __device__ __forceinline__ uint4 wg_transpose<float>(uint4 r, int lane) {
  {  // lane bit 1 <-> dword bit 1
    const bool hi = lane & 2;
    const uint32_t a = wg_xor2(hi ? r.x : r.z), b = wg_xor2(hi ? r.y : r.w);
    if (hi) { r.x = a; r.y = b; } else { r.z = a; r.w = b; }
  }
  {  // lane bit 0 <-> dword bit 0
    const bool hi = lane & 1;
    const uint32_t a = wg_xor1(hi ? r.x : r.y), b = wg_xor1(hi ? r.z : r.w);
    if (hi) { r.x = a; r.z = b; } else { r.y = a; r.w = b; }
  }
  return r;
}
template <> __device__ __forceinline__ uint4 wg_transpose<bf16_raw>(uint4 r, int lane) {
  {  // lane bit 2 <-> dword bit 1
    const bool hi = lane & 4;
    const uint32_t a = wg_xor4(hi ? r.x : r.z), b = wg_xor4(hi ? r.y : r.w);
    if (hi) { r.x = a; r.y = b; } else { r.z = a; r.w = b; }
  }
  {  // lane bit 1 <-> dword bit 0
    const bool hi = lane & 2;
    const uint32_t a = wg_xor2(hi ? r.x : r.y), b = wg_xor2(hi ? r.z : r.w);
    if (hi) { r.x = a; r.z = b; } else { r.y = a; r.w = b; }
  }
  {  // lane bit 0 <-> half-word: each lane sends the half its partner needs, packed two to a dword
    const bool hi = lane & 1;
    // even lane keeps the low halves and needs the partner's low halves; odd lane keeps the high halves and needs the partner's high ones
    const uint32_t s0 = hi ? __builtin_amdgcn_perm(r.y, r.x, 0x05040100) : __builtin_amdgcn_perm(r.y, r.x, 0x07060302);  // what the PARTNER needs
    const uint32_t s1 = hi ? __builtin_amdgcn_perm(r.w, r.z, 0x05040100) : __builtin_amdgcn_perm(r.w, r.z, 0x07060302);
    const uint32_t p0 = wg_xor1(s0), p1 = wg_xor1(s1);  // even lane: partner's (x.lo, y.lo), (z.lo, w.lo); odd lane: partner's high halves
    if (hi) {
      r.x = (p0 & 0xffffu) | (r.x & 0xffff0000u); r.y = (p0 >> 16) | (r.y & 0xffff0000u);
      r.z = (p1 & 0xffffu) | (r.z & 0xffff0000u); r.w = (p1 >> 16) | (r.w & 0xffff0000u);
    } else {
      r.x = (r.x & 0xffffu) | (p0 << 16); r.y = (r.y & 0xffffu) | (p0 & 0xffff0000u);
      r.z = (r.z & 0xffffu) | (p1 << 16); r.w = (r.w & 0xffffu) | (p1 & 0xffff0000u);
    }
  }
  return r;
}

// S: stride; KHW: kernel extent along H and W (1 or 3); TD x TH x 32: output-voxel tile.  KHW == 1 is "flat": the rows of the two
// operands are walked as one long W axis (1x1 convolutions and nn.Linear layers: the spatial structure does not matter).
template <typename T, int S, int KHW, int TD, int TH>
__global__ __launch_bounds__(512, 2) void conv_wgrad_kernel(const GmWgradDesc p, float* __restrict__ partial, int nsplit, long long tiles_total) {
  constexpr int VECW = ConvTraits<T>::VECW;
  constexpr int ES = (int)sizeof(T);
  constexpr int KSTEP = 4 * VECW, TW = 32, KSR = TW / KSTEP;
  constexpr int CIB = WgTraits<T>::CIB, COB = 64;
  constexpr int NCIF = CIB / 16, CGR = 8 / NCIF, COFW = (COB / 16) / CGR;  // bf16: 4 ci fragments x 2 co groups of 2; fp32: 2 x 4 of 1
  constexpr int PH = S * (TH - 1) + KHW, NR = TD * PH;                    // patch rows held for this work-group's depth tap
  constexpr int PLW = 40, RW = S * PLW;                                   // elements per plane / per patch row
  constexpr int PCOLS = S * (TW - 1) + KHW;
  constexpr int XPITCH = wg_pad_pitch(NR * RW * ES), GROWS = TD * TH, GPITCH = wg_pad_pitch(GROWS * TW * ES);
  constexpr int NT = KHW * KHW;
  constexpr int CVX = CIB / VECW, CVG = COB / VECW;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xT = smem;                       // [CIB][XPITCH]
  char* gT = smem + CIB * XPITCH;        // [COB][GPITCH]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int ncib = (p.Cin + CIB - 1) / CIB, ncob = (p.Cout + COB - 1) / COB;
  unsigned b = blockIdx.x;
  const int cib = b % ncib; b /= ncib;
  const int cob = b % ncob; b /= ncob;
  const int kdi = b % p.kd; b /= p.kd;
  const int split = b;
  const int ci0 = cib * CIB, co0 = cob * COB;
  const int cif = wave % NCIF, cog = wave / NCIF;

  const int ntd = (p.Do + TD - 1) / TD, nth = (p.Ho + TH - 1) / TH, ntw = (p.Wo + TW - 1) / TW;
  const long long rows_flat = (long long)p.N * p.Do * p.Ho * p.Wo;  // KHW == 1

  f32x4_t acc[NT][COFW];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int cf = 0; cf < COFW; ++cf) acc[t][cf] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const T* xin = reinterpret_cast<const T*>(p.x);
  const T* gin = reinterpret_cast<const T*>(p.gy);

  // A tile's operands travel global -> registers -> (transposed) LDS.  GS = VECW consecutive lanes hold the 16-byte channel vectors of
  // GS consecutive voxels (same channel vector): a GS x GS transpose across those lanes (a butterfly of DPP / swizzle exchanges) leaves
  // lane j with GS consecutive voxels of channel j, written by ONE ds_write_b128.  The lane groups of a wave-load are the channel
  // vectors of the same voxels, so global reads stay full 128-byte lines.  All loads of a tile are issued back to back, and those of
  // tile t + 1 right after tile t's LDS image is complete: they fly during its multiply.
  // What the first versions taught (rocprofv3 + the ISA): eight 2-byte LDS writes per vector with 4-way bank conflicts, then ~1000
  // integer instructions per tile and wave re-deriving every item's patch coordinates and LDS address (div / mod chains) plus 96
  // ds_bpermute per tile -- 2850 non-MFMA instructions against 144 MFMAs, 440-520 TFLOP/s.  Everything tile independent (patch
  // coordinates, channel vector, validity, LDS destination) is therefore packed into ONE register per item before the tile loop.
  constexpr int GS = VECW, XGR = PLW / GS, GGR = TW / GS;
  constexpr int XITEMS = NR * S * PLW * CVX, GITEMS = GROWS * TW * CVG;
  constexpr int XI = (XITEMS + 511) / 512, GI = (GITEMS + 511) / 512;
  static_assert(XITEMS % 64 == 0 && GITEMS % 64 == 0, "whole waves of items");
  static_assert(S * (TD - 1) < 4 && PH <= 16 && PCOLS <= 128 && CVX <= 8 && CVG <= 16, "field widths of the packed item descriptor");
  static_assert(CIB * XPITCH < (1 << 18) && 64 * GPITCH < (1 << 18), "LDS offsets fit the packed item descriptor");
  // x item: [1:0] S*dd  [5:2] hh  [12:6] pc  [15:13] channel vector  [16] valid  [31:17] LDS offset / 16
  // gy item: [0] gd  [3:1] gh  [8:4] w  [12:9] channel vector  [16] valid  [31:17] LDS offset / 16 (from gT)
  unsigned xpk[XI], gpk[GI];
#pragma unroll
  for (int j = 0; j < XI; ++j) {
    const int it = tid + j * 512;
    const int v = it % GS, cv = (it / GS) % CVX, grp = it / (GS * CVX);
    const int g = grp % XGR, plane = (grp / XGR) % S, pr = grp / (XGR * S);
    const int idx = g * GS + v, pc = S == 1 ? idx : 2 * idx + plane;
    const int dd = pr / PH, hh = pr - dd * PH;
    const bool valid = (it < XITEMS) & (pc < PCOLS) & (ci0 + cv * VECW < p.Cin);  // host: Cin % VECW == 0
    const unsigned dst = (unsigned)((cv * VECW + v) * XPITCH + (pr * RW + plane * PLW + g * GS) * ES);
    xpk[j] = (unsigned)(KHW == 1 ? 0 : S * dd) | ((unsigned)(KHW == 1 ? pr : hh) << 2) | ((unsigned)(pc & 127) << 6) | ((unsigned)cv << 13) |
             ((unsigned)valid << 16) | ((dst >> 4) << 17);
  }
#pragma unroll
  for (int j = 0; j < GI; ++j) {
    const int it = tid + j * 512;
    const int v = it % GS, cv = (it / GS) % CVG, grp = it / (GS * CVG);
    const int g = grp % GGR, gr = grp / GGR;
    const int w_ = g * GS + v;
    const bool valid = (it < GITEMS) & (co0 + cv * VECW < p.Cout);  // host: Cout % VECW == 0
    const unsigned dst = (unsigned)((cv * VECW + v) * GPITCH + (gr * TW + g * GS) * ES);
    gpk[j] = (unsigned)(KHW == 1 ? 0 : gr / TH) | ((unsigned)(KHW == 1 ? gr : gr % TH) << 1) | ((unsigned)w_ << 4) | ((unsigned)cv << 9) |
             ((unsigned)valid << 16) | ((dst >> 4) << 17);
  }
  uint4 xreg[XI], greg[GI];
  // tile coordinates of the next tile to load, advanced by nsplit tiles with mixed-radix carries (decoding tile -> (n, td, th, tw) with
  // 64-bit divisions cost ~600 scalar instructions per tile)
  int c_tw = 0, c_th = 0, c_td = 0, c_n = 0, s_tw = 0, s_th = 0, s_td = 0, s_n = 0;
  if (KHW != 1) {
    long long t = split;
    c_tw = (int)(t % ntw); t /= ntw;
    c_th = (int)(t % nth); t /= nth;
    c_td = (int)(t % ntd); t /= ntd;
    c_n = (int)t;
    t = nsplit;
    s_tw = (int)(t % ntw); t /= ntw;
    s_th = (int)(t % nth); t /= nth;
    s_td = (int)(t % ntd); t /= ntd;
    s_n = (int)t;
  }
  auto load_tile = [&](long long tile) __attribute__((always_inline)) {
    const int n = c_n, od0 = c_td * TD, oh0 = c_th * TH, ow0 = c_tw * TW;
    if (KHW != 1) {
      c_tw += s_tw;
      int carry = c_tw >= ntw ? 1 : 0;
      c_tw -= carry * ntw;
      c_th += s_th + carry;
      carry = c_th >= nth ? 1 : 0;
      c_th -= carry * nth;
      c_td += s_td + carry;
      carry = c_td >= ntd ? 1 : 0;
      c_td -= carry * ntd;
      c_n += s_n + carry;
    }
    const int ud0 = S * od0 - p.pd + kdi, uh0 = S * oh0 - p.ph, uw0 = S * ow0 - p.pw;
    const long long flat0 = tile * (GROWS * TW);
#pragma unroll
    for (int j = 0; j < XI; ++j) {
      const unsigned pk = xpk[j];
      const int a0 = pk & 3, a1 = (pk >> 2) & 15, a2 = (pk >> 6) & 127, cv = (pk >> 13) & 7;
      bool ok = (pk >> 16) & 1;
      long long vox;
      if (KHW == 1) {
        vox = flat0 + a1 * TW + a2;
        ok = ok & (vox < rows_flat);
      } else {
        const int ud = ud0 + a0, uh = uh0 + a1, uw = uw0 + a2;
        ok = ok & ((unsigned)ud < (unsigned)p.Ds) & ((unsigned)uh < (unsigned)p.Hs) & ((unsigned)uw < (unsigned)p.Ws);
        vox = ((n * p.Ds + ud) * p.Hs + uh) * p.Ws + uw;  // 32-bit: host checks N * D * H * W < 2^31
      }
      xreg[j] = make_uint4(0u, 0u, 0u, 0u);
      if (ok) xreg[j] = *reinterpret_cast<const uint4*>(xin + vox * p.x_ld + ci0 + cv * VECW);
    }
#pragma unroll
    for (int j = 0; j < GI; ++j) {
      const unsigned pk = gpk[j];
      const int a0 = pk & 1, a1 = (pk >> 1) & 7, a2 = (pk >> 4) & 31, cv = (pk >> 9) & 15;
      bool ok = (pk >> 16) & 1;
      long long vox;
      if (KHW == 1) {
        vox = flat0 + a1 * TW + a2;
        ok = ok & (vox < rows_flat);
      } else {
        const int od = od0 + a0, oh = oh0 + a1, ow = ow0 + a2;
        ok = ok & (od < p.Do) & (oh < p.Ho) & (ow < p.Wo);
        vox = ((n * p.Do + od) * p.Ho + oh) * p.Wo + ow;
      }
      greg[j] = make_uint4(0u, 0u, 0u, 0u);
      if (ok) greg[j] = *reinterpret_cast<const uint4*>(gin + vox * p.gy_ld + co0 + cv * VECW);
    }
  };
  auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < XI; ++j) {
      const uint4 tr = wg_transpose<T>(xreg[j], lane);  // every lane of the wave takes part in the exchanges
      if (tid + j * 512 < XITEMS) *reinterpret_cast<uint4*>(xT + ((xpk[j] >> 17) << 4)) = tr;
    }
#pragma unroll
    for (int j = 0; j < GI; ++j) {
      const uint4 tr = wg_transpose<T>(greg[j], lane);
      if (tid + j * 512 < GITEMS) *reinterpret_cast<uint4*>(gT + ((gpk[j] >> 17) << 4)) = tr;
    }
  };

  if (split < tiles_total) load_tile(split);
  for (long long tile = split; tile < tiles_total; tile += nsplit) {
    store_tile();
    __syncthreads();
    if (tile + nsplit < tiles_total) load_tile(tile + nsplit);
    // ---- multiply: one k-step per (tile row, 32-voxel run) ----------------------------------------------------------------------
    const char* arow = gT + (size_t)(cog * COFW * 16 + l15) * GPITCH + q * VECW * ES;
    const char* brow = xT + (size_t)(cif * 16 + l15) * XPITCH + q * VECW * ES;
#pragma unroll
    for (int gr = 0; gr < GROWS; ++gr) {
      const int d = gr / TH, h = gr % TH;
#pragma unroll
      for (int ks = 0; ks < KSR; ++ks) {
        uint4 a[COFW];
#pragma unroll
        for (int cf = 0; cf < COFW; ++cf)
          a[cf] = *reinterpret_cast<const uint4*>(arow + (size_t)cf * 16 * GPITCH + (gr * TW + ks * KSTEP) * ES);
#pragma unroll
        for (int kh = 0; kh < KHW; ++kh) {
          uint4 bf[KHW];
          WgShift<T, S, KHW>::run(brow + ((d * PH + S * h + kh) * RW + ks * KSTEP) * ES, bf);
#pragma unroll
          for (int kw = 0; kw < KHW; ++kw)
#pragma unroll
            for (int cf = 0; cf < COFW; ++cf) Mma<T>::run(a[cf], bf[kw], acc[kh * KHW + kw][cf]);
        }
      }
    }
    __syncthreads();
  }

  // ---- partial sums: D[co = 4q + r][ci = l15] -> partial[split][kd][tap][co_pad][ci_pad] ------------------------------------------
  const int cop = ncob * COB, cip = ncib * CIB;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int cf = 0; cf < COFW; ++cf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + (cog * COFW + cf) * 16 + 4 * q + r, ci = ci0 + cif * 16 + l15;
        partial[((((long long)split * p.kd + kdi) * NT + t) * cop + co) * cip + ci] = acc[t][cf][r];
      }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int nsplit, int KD, int NT,
                                                          int Cout, int Cin, int cop, int cip, int accumulate) {
  // (round 3: 32-bit index arithmetic -- host: the element count fits -- and eight slices requested per wait; the additions keep the slice order.
  //  The run-time slice loop waited for every slice in turn: 11 us per launch on average, 136 launches per C4 step)
  const int total = Cout * Cin * KD * NT;
  const long long sstride = (long long)KD * NT * cop * cip;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int t = i % NT;
    int r = i / NT;
    const int a = r % KD; r /= KD;
    const int ci = r % Cin;
    const int co = r / Cin;
    const float* src = partial + (((long long)a * NT + t) * cop + co) * cip + ci;
    const float prev = accumulate ? dw[i] : 0.f;
    float s = 0.f;
    int sp = 0;
    for (; sp + 8 <= nsplit; sp += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(long long)(sp + j) * sstride];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    if (sp < nsplit) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(long long)(sp + j < nsplit ? sp + j : sp) * sstride];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += sp + j < nsplit ? v[j] : 0.f;
    }
    dw[i] = accumulate ? prev + s : s;
  }
}

struct WgPlan { int variant; int td, th; int cib; long long tiles; int nsplit; long long partial_elems; int ncob, ncib, nt; };

static bool wgrad_plan(const GmWgradDesc& d, WgPlan& pl) {
  if (d.dtype != GM_F32 && d.dtype != GM_BF16) return false;
  const int vecw = d.dtype == GM_F32 ? 4 : 8;
  if (d.Cin % vecw || d.Cout % vecw || d.x_ld % vecw || d.gy_ld % vecw) return false;
  if ((reinterpret_cast<uintptr_t>(d.x) & 15) || (reinterpret_cast<uintptr_t>(d.gy) & 15)) return false;
  if (d.kh != d.kw || (d.kh != 1 && d.kh != 3) || (d.kd != 1 && d.kd != 3) || (d.kd == 3 && d.kh != 3)) return false;
  if (d.stride != 1 && d.stride != 2) return false;
  if ((long long)d.N * d.Ds * d.Hs * d.Ws >= (1LL << 31) || (long long)d.N * d.Do * d.Ho * d.Wo >= (1LL << 31)) return false;
  pl.cib = d.dtype == GM_F32 ? 32 : 64;
  pl.ncob = (d.Cout + 63) / 64;
  pl.ncib = (d.Cin + pl.cib - 1) / pl.cib;
  if (d.kh == 1) {
    if (d.stride != 1 || d.kd != 1 || d.pd || d.ph || d.pw) return false;
    pl.variant = 3; pl.td = 2; pl.th = 4; pl.nt = 1;
    const long long rows = (long long)d.N * d.Do * d.Ho * d.Wo;
    pl.tiles = (rows + 255) / 256;
  } else {
    pl.nt = 9;
    if (d.stride == 2) { pl.variant = 2; pl.td = 1; pl.th = 4; }
    else if (d.kd == 1 && d.Do == 1) { pl.variant = 1; pl.td = 1; pl.th = 8; }
    else { pl.variant = 0; pl.td = 2; pl.th = 4; }
    pl.tiles = (long long)d.N * ((d.Do + pl.td - 1) / pl.td) * ((d.Ho + pl.th - 1) / pl.th) * ((d.Wo + 31) / 32);
  }
  const long long base = (long long)d.kd * pl.ncob * pl.ncib;
  long long ns = 256 / base;  // one work-group per CU (~100 KiB of LDS each), every one with the same number of tiles (+-1)
  if (ns > pl.tiles) ns = pl.tiles;
  if (ns < 1) ns = 1;
  pl.nsplit = (int)ns;
  pl.partial_elems = ns * d.kd * pl.nt * (pl.ncob * 64LL) * ((long long)pl.ncib * pl.cib);
  return true;
}

extern "C" long long gm_conv_wgrad_workspace_bytes(const GmWgradDesc* d) {
  WgPlan pl;
  if (!d || !wgrad_plan(*d, pl)) return -1;
  return pl.partial_elems * 4;
}

template <typename T, int S, int KHW, int TD, int TH>
static void launch_wgrad(const GmWgradDesc& d, const WgPlan& pl, hipStream_t st) {
  constexpr int ES = (int)sizeof(T), CIB = WgTraits<T>::CIB;
  constexpr int PH = S * (TH - 1) + KHW, NR = TD * PH, RW = S * 40;
  constexpr size_t smem = (size_t)CIB * wg_pad_pitch(NR * RW * ES) + (size_t)64 * wg_pad_pitch(TD * TH * 32 * ES);
  static_assert(smem <= 160 * 1024, "tile does not fit the LDS");
  static bool attr_set = false;
  auto kern = conv_wgrad_kernel<T, S, KHW, TD, TH>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) (void)hipGetLastError();
    attr_set = true;
  }
  const unsigned grid = (unsigned)((long long)pl.nsplit * d.kd * pl.ncob * pl.ncib);
  kern<<<grid, 512, smem, st>>>(d, reinterpret_cast<float*>(d.workspace), pl.nsplit, pl.tiles);
}

template <typename T>
static void dispatch_wgrad(const GmWgradDesc& d, const WgPlan& pl, hipStream_t st) {
  switch (pl.variant) {
    case 0: launch_wgrad<T, 1, 3, 2, 4>(d, pl, st); break;
    case 1: launch_wgrad<T, 1, 3, 1, 8>(d, pl, st); break;
    case 2: launch_wgrad<T, 2, 3, 1, 4>(d, pl, st); break;
    default: launch_wgrad<T, 1, 1, 2, 4>(d, pl, st); break;
  }
}

extern "C" int gm_conv_wgrad(const GmWgradDesc* dp, void* stream) {
  GM_REQUIRE(dp && dp->x && dp->gy && dp->dw, "null pointer");
  const GmWgradDesc& d = *dp;
  WgPlan pl;
  GM_REQUIRE(wgrad_plan(d, pl), "geometry not covered: kernel 1 or 3 per axis, stride 1 or 2, channel counts multiples of one 16-byte vector");
  GM_REQUIRE(d.workspace && d.workspace_bytes >= pl.partial_elems * 4, "workspace too small (gm_conv_wgrad_workspace_bytes)");
  GM_REQUIRE((long long)pl.nsplit * d.kd * pl.ncob * pl.ncib < (1LL << 31), "grid too large");
  hipStream_t st = (hipStream_t)stream;
  if (d.N == 0 || d.Do * d.Ho * d.Wo == 0) {
    if (!d.accumulate) (void)hipMemsetAsync(d.dw, 0, sizeof(float) * d.Cout * d.Cin * d.kd * d.kh * d.kw, st);
    GM_LAUNCH_CHECK();
  }
  if (d.dtype == GM_F32) dispatch_wgrad<float>(d, pl, st);
  else dispatch_wgrad<bf16_raw>(d, pl, st);
  {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) GM_FAIL((int)e, hipGetErrorString(e));
  }
  const long long total = (long long)d.Cout * d.Cin * d.kd * pl.nt;
  GM_REQUIRE(total < (1LL << 31), "weight gradient with 2^31 or more elements");
  long long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  wgrad_reduce_kernel<<<(int)g, 256, 0, st>>>(reinterpret_cast<const float*>(d.workspace), d.dw, pl.nsplit, d.kd, pl.nt, d.Cout, d.Cin,
                                              pl.ncob * 64, pl.ncib * pl.cib, d.accumulate);
  GM_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// GroupNorm (+ SiLU) backward.  Forward: y = act(x * scale[n, c] + shift[n, c]) with scale = rstd * gamma, shift = beta - mean * scale.
// With g = gy * act'(x * scale + shift):
//   dx = A[n, c] * g + B[n, group] * x + C[n, group],  A = rstd * gamma,  B = -rstd^2 * M2,  C = -rstd * M1 + rstd^2 * M2 * mean,
//   M1 = mean_group(gamma * g),  M2 = rstd * (mean_group(gamma * g * x) - mean * M1),
//   dgamma[c] = sum_n rstd * (sum_v g x - mean * sum_v g),  dbeta[c] = sum_n sum_v g.
// (reference: torch.nn.functional.group_norm / SiLU autograd as used by diffusion_model_unet.py:623-690)
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float silu_grad(float z) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-z));  // v_exp_f32 + v_rcp_f32: ~1 ulp each
  return s * (1.0f + z * (1.0f - s));
}
// act'(z) for the activation codes of the forward prologue (0 none, 1 SiLU, 2 ReLU: torch's relu backward passes the gradient where z > 0)
// and 3 = LeakyReLU(0.01) (MONAI's default slope: the SPADE map convolutions, spade_norm.py:56-66)
__device__ __forceinline__ float act_grad(float z, int act) {
  return act == 1 ? silu_grad(z) : (act == 2 ? (z > 0.f ? 1.f : 0.f) : (act == 3 ? (z > 0.f ? 1.f : 0.01f) : 1.f));
}
// g[i] *= act'(z[i]) for a register vector, the (wave-uniform) activation kind tested once: N independent chains in one basic block (a per-element
// test serialises them -- see conv_act_vec in conv_common.h)
template <int N>
__device__ __forceinline__ void act_grad_mul(float (&g)[N], const float (&z)[N], int act) {
  if (act == 1) {
#pragma unroll
    for (int i = 0; i < N; ++i) g[i] *= silu_grad(z[i]);
  } else if (act == 2) {
#pragma unroll
    for (int i = 0; i < N; ++i) g[i] = z[i] > 0.f ? g[i] : 0.f;
  } else if (act == 3) {
#pragma unroll
    for (int i = 0; i < N; ++i) g[i] = z[i] > 0.f ? g[i] : 0.01f * g[i];
  }
}

// out[blk][n][c] = {sum_v g, sum_v g * x} over the rows of block blk: ONE plain fp64 store per (block, sample, channel) -- no atomics, no zero
// fill; gm_gn_bwd_finalize adds the gm_gn_bwd_stats_slots(N, V) partials in a fixed order, so a training step is bit-reproducible (round 2
// accumulated them with fp64 atomics into 64 slots: run-to-run differences in the last bit).  grid (nblk, N).
// Lane <-> 16-byte channel vector, 256 / CV rows in flight per block, fp32 partial sums over <= rows_per_block / R rows, LDS reduction
// over the rows in flight (HBM-bound: x and gy read once).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const T* __restrict__ x, long long x_ld, const T* __restrict__ gy, long long gy_ld,
                                                          const float* __restrict__ scale, const float* __restrict__ shift, long long ss_ld,
                                                          long long V, int C, int act, int rows_per_block, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int CV = C / VEC, R = 256 / CV;                 // host: CV <= 256
  float* part_a = reinterpret_cast<float*>(smem_raw);   // [R][C]
  float* part_b = part_a + (size_t)R * C;
  const int n = blockIdx.y, blk = blockIdx.x, t = threadIdx.x;
  const int cv = t % CV, r0 = t / CV;
  const long long row_begin = (long long)blk * rows_per_block;
  long long row_end = row_begin + rows_per_block;
  if (row_end > V) row_end = V;
  if (r0 < R) {
    float sc[VEC], sh[VEC], a[VEC], b2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      sc[i] = scale[n * ss_ld + cv * VEC + i];
      sh[i] = shift[n * ss_ld + cv * VEC + i];
      a[i] = 0.f; b2[i] = 0.f;
    }
    const T* xb = x + ((long long)n * V) * x_ld + (long long)cv * VEC;
    const T* gb = gy + ((long long)n * V) * gy_ld + (long long)cv * VEC;
    constexpr int UB = 4;  // rows requested per wait (clamped addresses, masked sums: the additions keep the row order)
    for (long long r = row_begin + r0; r < row_end; r += (long long)UB * R) {
      float xv[UB][VEC], gv[UB][VEC];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const long long rr = r + (long long)u * R < row_end ? r + (long long)u * R : r;
        if constexpr (VEC == 1) { xv[u][0] = ElemIO<T>::ld(xb + rr * x_ld); gv[u][0] = ElemIO<T>::ld(gb + rr * gy_ld); }
        else {
          Vec16<T>::unpack(*reinterpret_cast<const uint4*>(xb + rr * x_ld), xv[u]);
          Vec16<T>::unpack(*reinterpret_cast<const uint4*>(gb + rr * gy_ld), gv[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const bool ok = r + (long long)u * R < row_end;
        if (act) {
          float z[VEC];
#pragma unroll
          for (int i = 0; i < VEC; ++i) z[i] = xv[u][i] * sc[i] + sh[i];
          act_grad_mul(gv[u], z, act);
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) { a[i] += ok ? gv[u][i] : 0.f; b2[i] += ok ? gv[u][i] * xv[u][i] : 0.f; }
      }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      part_a[(size_t)r0 * C + cv * VEC + i] = a[i];
      part_b[(size_t)r0 * C + cv * VEC + i] = b2[i];
    }
  }
  __syncthreads();
  for (int c = t; c < C; c += 256) {
    double da = 0.0, db = 0.0;
    for (int r = 0; r < R; ++r) { da += (double)part_a[(size_t)r * C + c]; db += (double)part_b[(size_t)r * C + c]; }
    double* dst = out + (((long long)blk * gridDim.y + n) * C + c) * 2;
    *reinterpret_cast<double2*>(dst) = make_double2(da, db);
  }
}

// the launch geometry of gm_gn_bwd_stats = the number of partial rows its table holds (>= 256 rows per block, at most ~4 blocks per CU and sample)
static void gn_bwd_stats_geometry(int N, long long V, long long& nblk, int& rpb) {
  nblk = (V + 255) / 256;
  const long long cap = 1024 / (N < 1 ? 1 : (N > 8 ? 8 : N)) + 1;
  if (nblk > cap) nblk = cap;
  if (nblk < 1) nblk = 1;
  rpb = (int)((V + nblk - 1) / nblk);
  if (rpb < 1) rpb = 1;
  nblk = (V + rpb - 1) / rpb;
  if (nblk < 1) nblk = 1;
}

extern "C" long long gm_gn_bwd_stats_slots(int N, long long V) {
  long long nblk; int rpb;
  gn_bwd_stats_geometry(N, V, nblk, rpb);
  return nblk;
}

template <typename T, int VEC>
static void launch_gn_bwd_stats(const void* x, long long x_ld, const void* gy, long long gy_ld, const float* scale, const float* shift,
                                long long ss_ld, int N, long long V, int C, int act, double* out, hipStream_t st) {
  const int CV = C / VEC, R = 256 / CV;
  long long nblk; int rpb;
  gn_bwd_stats_geometry(N, V, nblk, rpb);
  dim3 grid((unsigned)nblk, N);
  const size_t smem = (size_t)R * C * 2 * sizeof(float);
  gn_bwd_stats_kernel<T, VEC><<<grid, 256, smem, st>>>((const T*)x, x_ld, (const T*)gy, gy_ld, scale, shift, ss_ld, V, C, act, rpb, out);
}

extern "C" int gm_gn_bwd_stats(const void* x, long long x_ld, const void* gy, long long gy_ld, const float* scale, const float* shift,
                               long long ss_ld, int N, long long V, int C, int act, double* out, int dtype, void* stream) {
  GM_REQUIRE(x && gy && scale && shift && out, "null pointer");
  GM_REQUIRE(N <= 65535, "batch too large");
  if (N == 0 || V == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int vec = dtype == GM_F32 ? 4 : 8;
  const bool vec_ok = (C % vec == 0) && (x_ld % vec == 0) && (gy_ld % vec == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)gy & 15) == 0);
  GM_REQUIRE((vec_ok ? C / vec : C) <= 256, "too many channels for gm_gn_bwd_stats");
  if (dtype == GM_F32) {
    if (vec_ok) launch_gn_bwd_stats<float, 4>(x, x_ld, gy, gy_ld, scale, shift, ss_ld, N, V, C, act, out, st);
    else launch_gn_bwd_stats<float, 1>(x, x_ld, gy, gy_ld, scale, shift, ss_ld, N, V, C, act, out, st);
  } else if (dtype == GM_BF16) {
    if (vec_ok) launch_gn_bwd_stats<bf16_raw, 8>(x, x_ld, gy, gy_ld, scale, shift, ss_ld, N, V, C, act, out, st);
    else launch_gn_bwd_stats<bf16_raw, 1>(x, x_ld, gy, gy_ld, scale, shift, ss_ld, N, V, C, act, out, st);
  } else {
    GM_FAIL(-2, "unsupported dtype");
  }
  GM_LAUNCH_CHECK();
}

// one wave per group; fwd = forward per-channel statistics {sum x, sum x^2} [S_fwd][N][C][2] (gm_gn_channel_stats / a convolution epilogue),
// bwd = {sum g, sum g x} [S_bwd][N][C][2] (gm_gn_bwd_stats: one partial per block), both fp64.  Every table sum runs over the slots with all 64
// lanes (lane l takes slots l, l + 64, ...) followed by an xor-shuffle tree: a fixed order of additions whatever the hardware schedules.
__global__ __launch_bounds__(64) void gn_bwd_finalize_kernel(const double* __restrict__ fwd, int S_fwd, const double* __restrict__ bwd, int S_bwd, int N, int C,
                                                            int G, long long V, float eps, const float* __restrict__ gamma, float* __restrict__ A,
                                                            float* __restrict__ B, float* __restrict__ Cc, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const int cpg = C / G;
  const double m = (double)cpg * (double)V;
  auto slot_sum2 = [&](const double* t, int slots, int n, int c, double& s0, double& s1) {  // wave-uniform (n, c): both components at once
    double a = 0.0, b = 0.0;
    for (int sl = lane; sl < slots; sl += 64) {
      const double2 v = *reinterpret_cast<const double2*>(t + (((long long)sl * N + n) * C + c) * 2);
      a += v.x; b += v.y;
    }
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    s0 = a; s1 = b;
  };
  for (int n = 0; n < N; ++n) {
    double sx = 0.0, sxx = 0.0;
    for (int j = 0; j < cpg; ++j) {
      double a, b;
      slot_sum2(fwd, S_fwd, n, g * cpg + j, a, b);
      sx += a; sxx += b;
    }
    const double mean = sx / m;
    double var = sxx / m - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    double s1 = 0.0, s2 = 0.0;
    for (int j = 0; j < cpg; ++j) {
      const int c = g * cpg + j;
      const double ga = gamma ? (double)gamma[c] : 1.0;
      double sg, sgx;
      slot_sum2(bwd, S_bwd, n, c, sg, sgx);
      s1 += ga * sg;
      s2 += ga * (sgx - mean * sg);
      const double dg = rstd * (sgx - mean * sg);
      if (lane == 0) {
        if (dgamma) dgamma[c] = (float)((n == 0 ? 0.0 : (double)dgamma[c]) + dg);   // lane 0 owns every channel of the group for every n
        if (dbeta) dbeta[c] = (float)((n == 0 ? 0.0 : (double)dbeta[c]) + sg);
      }
    }
    const double M1 = s1 / m, M2 = rstd * s2 / m;
    for (int j = lane; j < cpg; j += 64) {
      const int c = g * cpg + j;
      const double ga = gamma ? (double)gamma[c] : 1.0;
      A[(long long)n * C + c] = (float)(rstd * ga);
      B[(long long)n * C + c] = (float)(-rstd * rstd * M2);
      Cc[(long long)n * C + c] = (float)(-rstd * M1 + rstd * rstd * M2 * mean);
    }
  }
}

extern "C" int gm_gn_bwd_finalize(const double* fwd_stats, int fwd_slots, const double* bwd_stats, int bwd_slots, int N, int C, int G, long long V,
                                  float eps, const float* gamma, float* A, float* B, float* Cc, float* dgamma, float* dbeta, void* stream) {
  GM_REQUIRE(fwd_stats && bwd_stats && A && B && Cc && fwd_slots > 0 && bwd_slots > 0, "null pointer");
  GM_REQUIRE(G > 0 && C % G == 0, "channels must be divisible by groups");
  if (N == 0) return 0;
  gn_bwd_finalize_kernel<<<G, 64, 0, (hipStream_t)stream>>>(fwd_stats, fwd_slots, bwd_stats, bwd_slots, N, C, G, V, eps, gamma, A, B, Cc, dgamma, dbeta);
  GM_LAUNCH_CHECK();
}

// dx = gy * act'(x * scale + shift) * A + x * B + Cc.  Lane <-> one 16-byte channel vector, its five coefficient vectors in registers,
// rows walked with 256 / CV rows in flight per block (a grid-stride loop over (voxel, vector) items re-read the five [N][C] tables per
// element: 0.6-2.3 ms on 268-537 MB tensors against 0.2-0.4 ms of HBM time).  grid (nblk, N)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ x, long long x_ld, const T* __restrict__ gy, long long gy_ld,
                                                          T* __restrict__ dx, long long dx_ld, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, long long ss_ld, const float* __restrict__ A,
                                                          const float* __restrict__ B, const float* __restrict__ Cc, long long V, int C,
                                                          int rows_per_block, int act) {
  const int CV = C / VEC, R = 256 / CV;
  const int n = blockIdx.y, t = threadIdx.x;
  const int cv = t % CV, r0 = t / CV;
  if (r0 >= R) return;
  const long long row_begin = (long long)blockIdx.x * rows_per_block;
  long long row_end = row_begin + rows_per_block;
  if (row_end > V) row_end = V;
  float sc[VEC], sh[VEC], ca[VEC], cb[VEC], cc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int c = cv * VEC + i;
    sc[i] = scale[n * ss_ld + c]; sh[i] = shift[n * ss_ld + c];
    ca[i] = A[(long long)n * C + c]; cb[i] = B[(long long)n * C + c]; cc[i] = Cc[(long long)n * C + c];
  }
  const T* xb = x + ((long long)n * V) * x_ld + (long long)cv * VEC;
  const T* gb = gy + ((long long)n * V) * gy_ld + (long long)cv * VEC;
  T* db = dx + ((long long)n * V) * dx_ld + (long long)cv * VEC;
  constexpr int UB = 4;  // rows requested per wait
  for (long long r = row_begin + r0; r < row_end; r += (long long)UB * R) {
    float xv[UB][VEC], gv[UB][VEC];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const long long rr = r + (long long)u * R < row_end ? r + (long long)u * R : r;
      if constexpr (VEC == 1) { xv[u][0] = ElemIO<T>::ld(xb + rr * x_ld); gv[u][0] = ElemIO<T>::ld(gb + rr * gy_ld); }
      else {
        Vec16<T>::unpack(*reinterpret_cast<const uint4*>(xb + rr * x_ld), xv[u]);
        Vec16<T>::unpack(*reinterpret_cast<const uint4*>(gb + rr * gy_ld), gv[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const long long rr = r + (long long)u * R;
      float o[VEC];
      if (act) {
        float z[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) z[k] = xv[u][k] * sc[k] + sh[k];
        act_grad_mul(gv[u], z, act);
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) o[k] = gv[u][k] * ca[k] + xv[u][k] * cb[k] + cc[k];
      if (rr < row_end) {
        if constexpr (VEC == 1) ElemIO<T>::st(db + rr * dx_ld, o[0]);
        else *reinterpret_cast<uint4*>(db + rr * dx_ld) = Vec16<T>::pack(o);
      }
    }
  }
}

extern "C" int gm_gn_bwd_apply(const void* x, long long x_ld, const void* gy, long long gy_ld, void* dx, long long dx_ld, const float* scale,
                               const float* shift, long long ss_ld, const float* A, const float* B, const float* Cc, int N, long long V, int C,
                               int act, int dtype, void* stream) {
  GM_REQUIRE(x && gy && dx && scale && shift && A && B && Cc, "null pointer");
  GM_REQUIRE(N <= 65535, "batch too large");
  if ((long long)N * V * C == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int vec = dtype == GM_F32 ? 4 : 8;
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool vec_ok = (C % vec == 0) && (x_ld % vec == 0) && (gy_ld % vec == 0) && (dx_ld % vec == 0) && al(x) && al(gy) && al(dx);
  const int CV = vec_ok ? C / vec : C;
  GM_REQUIRE(CV <= 256, "too many channels for gm_gn_bwd_apply");
  const int R = 256 / CV;
  long long nblk = (V + (long long)R * 8 - 1) / ((long long)R * 8);  // >= 8 rows per lane
  const long long cap = 4096 / (N > 16 ? 16 : N) + 1;
  if (nblk > cap) nblk = cap;
  const int rpb = (int)((V + nblk - 1) / nblk);
  nblk = (V + rpb - 1) / rpb;
  dim3 grid((unsigned)nblk, N);
#define GM_GNB_LAUNCH(T, VEC) \
  gn_bwd_apply_kernel<T, VEC><<<grid, 256, 0, st>>>((const T*)x, x_ld, (const T*)gy, gy_ld, (T*)dx, dx_ld, scale, shift, ss_ld, A, B, Cc, V, C, rpb, act)
  if (dtype == GM_F32) { if (vec_ok) GM_GNB_LAUNCH(float, 4); else GM_GNB_LAUNCH(float, 1); }
  else if (dtype == GM_BF16) { if (vec_ok) GM_GNB_LAUNCH(bf16_raw, 8); else GM_GNB_LAUNCH(bf16_raw, 1); }
  else GM_FAIL(-2, "unsupported dtype");
#undef GM_GNB_LAUNCH
  GM_LAUNCH_CHECK();
}

// SPADE modulation backward (generative/networks/blocks/spade_norm.py:79-96 under torch autograd): y = act(xn * g + bm) with xn the
// parameter-free-normalised tensor and g = 1 + gamma(seg), bm = beta(seg) per-voxel maps.  With gu = gy * act'(xn * g + bm):
//   dxn = gu * g,  dg = gu * xn,  dbm = gu       (dg / dbm share the row pitch gb_ld, like g / bm).  Element-wise, HBM-bound.
template <typename T>
__global__ __launch_bounds__(256) void spade_bwd_kernel(const T* __restrict__ xn, long long x_ld, const T* __restrict__ g, const T* __restrict__ bm,
                                                       long long gb_ld, const T* __restrict__ gy, long long gy_ld, T* __restrict__ dxn, long long dx_ld,
                                                       T* __restrict__ dg, T* __restrict__ dbm, long long dgb_ld, int C, long long total, int act) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / C;
    const int c = (int)(i - row * C);
    const float x = ElemIO<T>::ld(xn + row * x_ld + c), gv = ElemIO<T>::ld(g + row * gb_ld + c), bv = ElemIO<T>::ld(bm + row * gb_ld + c);
    float gu = ElemIO<T>::ld(gy + row * gy_ld + c);
    if (act) gu *= act_grad(x * gv + bv, act);
    ElemIO<T>::st(dxn + row * dx_ld + c, gu * gv);
    ElemIO<T>::st(dg + row * dgb_ld + c, gu * x);
    ElemIO<T>::st(dbm + row * dgb_ld + c, gu);
  }
}

extern "C" int gm_spade_bwd(const void* xn, long long x_ld, const void* g, const void* bm, long long gb_ld, const void* gy, long long gy_ld, void* dxn,
                            long long dx_ld, void* dg, void* dbm, long long dgb_ld, long long rows, int C, int act, int dtype, void* stream) {
  GM_REQUIRE(xn && g && bm && gy && dxn && dg && dbm, "null pointer");
  const long long total = rows * C;
  if (total == 0) return 0;
  long long grid = (total + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GM_F32)
    spade_bwd_kernel<float><<<(int)grid, 256, 0, st>>>((const float*)xn, x_ld, (const float*)g, (const float*)bm, gb_ld, (const float*)gy, gy_ld,
                                                       (float*)dxn, dx_ld, (float*)dg, (float*)dbm, dgb_ld, C, total, act);
  else if (dtype == GM_BF16)
    spade_bwd_kernel<bf16_raw><<<(int)grid, 256, 0, st>>>((const bf16_raw*)xn, x_ld, (const bf16_raw*)g, (const bf16_raw*)bm, gb_ld, (const bf16_raw*)gy,
                                                          gy_ld, (bf16_raw*)dxn, dx_ld, (bf16_raw*)dg, (bf16_raw*)dbm, dgb_ld, C, total, act);
  else GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// out[c] = sum over slots and samples of stats[slot][n][c][0] (a bias gradient from the gm_gn_channel_stats table of gy), or, per sample,
// out[n][c] = sum over slots (the gradient of a per-sample row vector added by the convolution epilogue: the timestep embedding)
__global__ __launch_bounds__(256) void stats_colsum_kernel(const double* __restrict__ stats, int slots, int N, int C, float* __restrict__ out, int per_sample) {
  // one wave per output element, lanes over the slot (x sample) copies, xor-shuffle reduction in a fixed order: a thread walking the 64
  // copies serially cost 22 us per call = 2.6 ms per C4 training step (profiles/r01_c4_train_kernel_stats.csv)
  const int lane = threadIdx.x & 63;
  const int w = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (w >= (per_sample ? N * C : C)) return;  // wave-uniform
  double s = 0.0;
  if (per_sample) {
    for (int sl = lane; sl < slots; sl += 64) s += stats[((long long)sl * N * C + w) * 2];
  } else {
    for (long long j = lane; j < (long long)slots * N; j += 64) s += stats[(j * C + w) * 2];
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) out[w] = (float)s;
}

extern "C" int gm_stats_colsum(const double* stats, int slots, int N, int C, float* out, int per_sample, void* stream) {
  GM_REQUIRE(stats && out && slots > 0, "null pointer");
  if (C == 0 || N == 0) return 0;
  const long long total = per_sample ? (long long)N * C : C;
  stats_colsum_kernel<<<(unsigned)((total * 64 + 255) / 256), 256, 0, (hipStream_t)stream>>>(stats, slots, N, C, out, per_sample);
  GM_LAUNCH_CHECK();
}

// dS[row][j] = scale * P[row][j] * (dP[row][j] - sum_k dP[row][k] P[row][k]): the softmax backward of attention scores S = scale Q K^T
// (torch softmax autograd as used by diffusion_model_unet.py:143-153, 407-415).  One wave per row, fp32.
__global__ __launch_bounds__(64) void softmax_bwd_kernel(const float* __restrict__ P, const float* __restrict__ dP, float* __restrict__ dS, int V,
                                                        float scale) {
  const long long row = blockIdx.x;
  const int lane = threadIdx.x;
  const float* pr = P + row * (long long)V;
  const float* dp = dP + row * (long long)V;
  float dot = 0.f;
  for (int j = lane; j < V; j += 64) dot += pr[j] * dp[j];
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
  float* out = dS + row * (long long)V;
  for (int j = lane; j < V; j += 64) out[j] = scale * pr[j] * (dp[j] - dot);
}

extern "C" int gm_softmax_bwd(const float* probs, const float* dprobs, float* dscores, long long rows, int V, float scale, void* stream) {
  GM_REQUIRE(probs && dprobs && dscores, "null pointer");
  GM_REQUIRE(V > 0 && rows < (1LL << 31), "bad geometry");
  if (rows == 0) return 0;
  softmax_bwd_kernel<<<(unsigned)rows, 64, 0, (hipStream_t)stream>>>(probs, dprobs, dscores, V, scale);
  GM_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// LayerNorm backward (nn.LayerNorm of the transformer blocks, diffusion_model_unet.py:219-223): one wave per row.
//   xhat = (x - mean) rstd,  g = gy gamma,  dx = rstd (g - mean_c(g) - xhat mean_c(g xhat));
//   param_stats[block][c] = {sum_rows gy xhat, sum_rows gy} over the rows the block walks  (fp64 [gm_layernorm_bwd_slots(rows)][C][2], one plain
//   store each -- no atomics, nothing to zero; nullable)
// ---------------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ x, long long x_ld, const T* __restrict__ gy, long long gy_ld,
                                                           T* __restrict__ dx, long long dx_ld, const float* __restrict__ gamma, long long rows,
                                                           int C, float eps, double* __restrict__ param_stats) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* acc = reinterpret_cast<float*>(smem_raw);  // [4 waves][C][2]: this wave's sums over the rows it walks (its own lanes' channels only)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = lane; c < C; c += 64) { acc[(wave * C + c) * 2] = 0.f; acc[(wave * C + c) * 2 + 1] = 0.f; }
  // block b walks rows 4 b + wave, + 4 gridDim.x, ...: a fixed assignment, so every partial -- and the fixed-order sum over the partials -- is
  // the same on every run (round 2: one block per 4 rows, fp64 atomics into 64 slots)
  for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
    const T* xr = x + row * x_ld;
    const T* gr = gy + row * gy_ld;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += ElemIO<T>::ld(xr + c);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = ElemIO<T>::ld(xr + c) - mean; q += d * d; }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    float m1 = 0.f, m2 = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float xh = (ElemIO<T>::ld(xr + c) - mean) * rstd, g = ElemIO<T>::ld(gr + c) * (gamma ? gamma[c] : 1.f);
      m1 += g; m2 += g * xh;
    }
    for (int o = 32; o > 0; o >>= 1) { m1 += __shfl_xor(m1, o, 64); m2 += __shfl_xor(m2, o, 64); }
    m1 /= (float)C; m2 /= (float)C;
    T* dr = dx + row * dx_ld;
    for (int c = lane; c < C; c += 64) {
      const float xh = (ElemIO<T>::ld(xr + c) - mean) * rstd, gv = ElemIO<T>::ld(gr + c);
      ElemIO<T>::st(dr + c, rstd * (gv * (gamma ? gamma[c] : 1.f) - m1 - xh * m2));
      acc[(wave * C + c) * 2] += gv * xh;
      acc[(wave * C + c) * 2 + 1] += gv;
    }
  }
  if (!param_stats) return;
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < 4; ++w) { a += (double)acc[(w * C + c) * 2]; b += (double)acc[(w * C + c) * 2 + 1]; }
    *reinterpret_cast<double2*>(param_stats + ((long long)blockIdx.x * C + c) * 2) = make_double2(a, b);  // one plain store per (block, channel)
  }
}

// blocks of gm_layernorm_bwd = rows of its [slots][C][2] partial table: one block per 4 rows up to 256 blocks, which then walk the rows
extern "C" int gm_layernorm_bwd_slots(long long rows) {
  const long long g = (rows + 3) / 4;
  return (int)(g < 1 ? 1 : (g > 256 ? 256 : g));
}

extern "C" int gm_layernorm_bwd(const void* x, long long x_ld, const void* gy, long long gy_ld, void* dx, long long dx_ld, const float* gamma,
                                long long rows, int C, float eps, double* param_stats, int dtype, void* stream) {
  GM_REQUIRE(x && gy && dx, "null pointer");
  GM_REQUIRE(C > 0 && C <= 4096, "LayerNorm width out of range");
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)gm_layernorm_bwd_slots(rows);
  const size_t smem = (size_t)4 * C * 2 * sizeof(float);
  if (dtype == GM_F32)
    layernorm_bwd_kernel<float><<<grid, 256, smem, st>>>((const float*)x, x_ld, (const float*)gy, gy_ld, (float*)dx, dx_ld, gamma, rows, C, eps, param_stats);
  else if (dtype == GM_BF16)
    layernorm_bwd_kernel<bf16_raw><<<grid, 256, smem, st>>>((const bf16_raw*)x, x_ld, (const bf16_raw*)gy, gy_ld, (bf16_raw*)dx, dx_ld, gamma, rows, C,
                                                            eps, param_stats);
  else
    GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}

// GEGLU backward (MONAI MLPBlock act = "GEGLU", diffusion_model_unet.py:211): y = a * gelu(gate), x = [a | gate]:
//   dx[:, :inner] = gy * gelu(gate),  dx[:, inner:] = gy * a * gelu'(gate),  gelu'(g) = Phi(g) + g phi(g)
template <typename T>
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const T* __restrict__ x, long long x_ld, const T* __restrict__ gy, long long gy_ld,
                                                       T* __restrict__ dx, long long dx_ld, long long rows, int inner) {
  const long long total = rows * inner;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / inner;
    const int j = (int)(i - r * inner);
    const float a = ElemIO<T>::ld(x + r * x_ld + j), g = ElemIO<T>::ld(x + r * x_ld + inner + j), go = ElemIO<T>::ld(gy + r * gy_ld + j);
    const float cdf = 0.5f * (1.0f + erff(g * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * g * g);
    ElemIO<T>::st(dx + r * dx_ld + j, go * g * cdf);
    ElemIO<T>::st(dx + r * dx_ld + inner + j, go * a * (cdf + g * pdf));
  }
}

extern "C" int gm_geglu_bwd(const void* x, long long x_ld, const void* gy, long long gy_ld, void* dx, long long dx_ld, long long rows, int inner,
                            int dtype, void* stream) {
  GM_REQUIRE(x && gy && dx, "null pointer");
  const long long total = rows * inner;
  if (total == 0) return 0;
  long long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GM_F32) geglu_bwd_kernel<float><<<(int)g, 256, 0, st>>>((const float*)x, x_ld, (const float*)gy, gy_ld, (float*)dx, dx_ld, rows, inner);
  else if (dtype == GM_BF16)
    geglu_bwd_kernel<bf16_raw><<<(int)g, 256, 0, st>>>((const bf16_raw*)x, x_ld, (const bf16_raw*)gy, gy_ld, (bf16_raw*)dx, dx_ld, rows, inner);
  else GM_FAIL(-2, "unsupported dtype");
  GM_LAUNCH_CHECK();
}
