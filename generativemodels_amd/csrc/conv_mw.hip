// 3x3x3 stride-1 bf16 convolution, THREE work-groups per CU (tile configuration 21): the round-4 form of the LDS-DMA kernel for the large
// prologue-free ResnetBlock convolutions (reference: diffusion_model_unet.py:669-696, autoencoderkl.py:96-122).
//
// What conv_dma.hip's cycle stamps said: a 64 -> 64 channel tile lives 44.5 k cycles, 24 k of them outside its tap loop (address
// set-up, first-patch round trip, epilogue + statistics, chunk boundaries), and with 78 KiB of LDS and 256 registers per wave only TWO tiles
// share a CU.  The barriers make a work-group ONE customer of its SIMDs' matrix pipes: with think time Z = 24 k and matrix demand S = 13.8 k
// per tile, two customers cannot exceed 2 / (Z + S) = one tile per 18.9 k cycles (measured: 22.3 k), three reach the pipe's own 13.8 k.
// This kernel is built for the third customer:
//   * K advances in HALF-chunks of 16 input channels: 32-byte LDS rows, halo patch 24 KiB, 3-tap weight panel 6 KiB x 3 ring slots ->
//     42.25 KiB per work-group;
//   * v_mfma_f32_32x32x16_bf16 (measured ceiling 2 382 vs 2 075 TFLOP/s for 16x16x32, half the matrix issue slots per FLOP): a wave's 64
//     voxels x 64 channels are 2 x 2 blocks; per tap 4 MFMAs and 4 ds_read_b128, two operand register sets software-pipelined over the taps
//     and the group barrier as in conv_dma.hip -- 64 accumulator + 32 operand registers, <= 168 per wave (three waves per SIMD);
//   * tile-invariant set-up: a wave moves the SAME four in-plane pieces of every patch plane (plane pitch = 4 LDS-DMA pieces, 4 waves), so a
//     lane's (line, column) inside the plane is a launch constant and placing a tile costs one bounds test in H / W plus a scalar test per
//     plane (conv_dma.hip: 11 pieces per wave, two constant divisions + six compares each, per tile);
//   * the weight image is gm_pack_conv_weight's [chunk32][tap][Cout_pad][32]: the half-chunk is a 32-byte column of its 64-byte rows (the
//     LDS-DMA source address is free per lane), no second packing.
// Index arithmetic lives in conv_mw_index.h and is replayed on the host by tests/emulate_conv_mw.cpp (DMA pieces, fragments, MFMA lane
// maps, bank conflicts).  bf16 only; stride 1, 3x3x3, direct input (in_mode 0), no fused prologue, no split-K: everything else stays on
// conv_dma.hip.  Epilogue (bias + timestep row + residual, LDS-transposed 16-byte stores, fused GroupNorm statistics, fused 1x1 shortcut,
// second input source of the virtual concatenation) as there.
#include "conv_dma_shared.h"
#include "conv_mw_index.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __attribute__((aligned(64))) unsigned int gm_mw_zero_row[16] = {0};  // the source of every padding row (this TU's own: no RDC)

__device__ __forceinline__ void mma32(const uint4& a, const uint4& b, f32x16_t& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

#ifdef GM_CONV_TIMELINE
#define MW_STAMP(k)                                                                                                            \
  do {                                                                                                                         \
    if ((p.debug_flags & 4096) && threadIdx.x == 0)                                                                            \
      reinterpret_cast<unsigned long long*>(p.kpartial)[(long long)blockIdx.x * 64 + (k)] = __builtin_readcyclecounter();      \
  } while (0)
#else
#define MW_STAMP(k)
#endif

__global__ __launch_bounds__(256, 3) void conv_mw_kernel(const GmConvDesc p) {
  using namespace mw;
  typedef bf16_raw T;
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // [patch 24 KiB][3 weight panels x 6 KiB][addend vector 256 B]
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- the tile of this work-group: XCD x owns a contiguous range of the tile list (neighbouring halo patches meet in one L2) -------------
  const int ntd = (p.Do + TD - 1) / TD, nth = (p.Ho + TH - 1) / TH, ntw = (p.Wo + TW - 1) / TW;
  const int ncb = (p.Cout + BN - 1) / BN;
  MW_STAMP(0);
  MW_STAMP(1);
  unsigned b = xcd_remap(blockIdx.x, gridDim.x);
  const int cb = b % ncb; b /= ncb;
  const int tw_i = b % ntw; b /= ntw;
  const int th_i = b % nth; b /= nth;
  const int td_i = b % ntd; b /= ntd;
  const int n = (int)b;
  const int od0 = td_i * TD, oh0 = th_i * TH, ow0 = tw_i * TW;
  const int nchunks = p.Cin / BKH;                          // host-checked: Cin % 16 == 0
  const int nchunks0 = p.x2 ? p.cin_split / BKH : nchunks;  // half-chunks of the first source (virtual concatenation: cin_split % 16 == 0)
  const int cout_pad = (p.Cout + 15) & ~15;
  const char* zero = reinterpret_cast<const char*>(gm_mw_zero_row) + ((lane & 1) << 4);

  // ---- patch placement: (line, column) of this lane's row inside a plane is a launch constant ------------------------------------------------
  const PatchLane pl = patch_lane(wave, lane);
  const int ud0 = od0 - p.pd;
  int pv0;        // source voxel of this lane's row in patch plane 0 (planes: + j * Hs * Ws)
  bool hw_ok;
  {
    const int uh = oh0 - p.ph + pl.line, uw = ow0 - p.pw + pl.col;
    hw_ok = pl.valid && uh >= 0 && uh < p.Hs && uw >= 0 && uw < p.Ws;
    pv0 = ((n * p.Ds + ud0) * p.Hs + uh) * p.Ws + uw;
  }
  const int plane_vox = p.Hs * p.Ws;
  const char* xbase = reinterpret_cast<const char*>(p.x) + (pl.slot << 4);
  const char* x2base = reinterpret_cast<const char*>(p.x2) + (pl.slot << 4);
  const long long xrowb = p.x_ld * 2, x2rowb = p.x2_ld * 2;
  auto issue_patch = [&](int chunk) __attribute__((always_inline)) {
#ifdef GM_CONV_ABLATE
    if (p.debug_flags & 1024) return;
#endif
    const bool second = chunk >= nchunks0;  // wave-uniform
    const char* cbase = second ? x2base + (long long)(chunk - nchunks0) * ROWB : xbase + (long long)chunk * ROWB;
    const long long rowb = second ? x2rowb : xrowb;
#pragma unroll
    for (int j = 0; j < PD; ++j) {
      const int ud = ud0 + j;                                  // wave-uniform
      const bool ok = hw_ok && ud >= 0 && ud < p.Ds;
      int pv = pv0 + j * plane_vox;
      asm volatile("" : "+v"(pv));                            // keeps the 64-bit row offsets out of the chunk loop's live set
      const char* src = ok ? cbase + pv * rowb : zero;
      dma16(src, lds0 + (unsigned)patch_piece_dst(j, wave));
    }
  };

  // ---- weight panels -------------------------------------------------------------------------------------------------------------------------
  int wsrc_f, wsrc_h;  // byte offset of this lane's 16 bytes within a (chunk32, group) panel image (half 0), or -1 beyond cout_pad
  {
    const WLane f = wpanel_lane(wave, lane, 0), h = wpanel_lane(wave, lane, 1);
    const int cof = cb * BN + f.co, coh = cb * BN + h.co;
    wsrc_f = cof < cout_pad ? (int)wsrc_offset(f.tap, cof, cout_pad, 0, f.slot) : -1;
    wsrc_h = coh < cout_pad ? (int)wsrc_offset(h.tap, coh, cout_pad, 0, h.slot) : -1;
  }
  const char* wbase = reinterpret_cast<const char*>(p.w);
  auto issue_w = [&](int chunk, int g, int slot) __attribute__((always_inline)) {  // panel (half-chunk, group g) -> ring slot; g < 9 (caller carries into the chunk)
#ifdef GM_CONV_ABLATE
    if (p.debug_flags & 512) return;
#endif
    const char* panel = wbase + ((long long)((chunk >> 1) * 27 + G * g) * cout_pad) * SRC_ROWB + (chunk & 1) * ROWB;
    dma16(wsrc_f >= 0 ? panel + wsrc_f : zero, lds0 + (unsigned)wpanel_piece_dst(slot, wave, 0));
    if (lane < 32) dma16(wsrc_h >= 0 ? panel + wsrc_h : zero, lds0 + (unsigned)wpanel_piece_dst(slot, wave, 1));
  };
  constexpr int WPW = 2;  // LDS-DMA instructions per wave per panel

  // ---- prologue: first patch, first two panels, the epilogue addend ---------------------------------------------------------------------------
  issue_patch(0);
  issue_w(0, 0, 0);
  issue_w(0, 1, 1);
  MW_STAMP(55);
  float* addv = reinterpret_cast<float*>(smem + ADDV_OFF);
  {
    KDesc& pa = cold_desc();
    float addend = 0.f;  // bias + shortcut bias + timestep row (this order), fp32
    if (tid < BN) {
      const int co = cb * BN + tid;
      if (co < pa.Cout) {
        if (pa.bias) addend += pa.bias[co];
        if (pa.skip_bias) addend += pa.skip_bias[co];
        if (pa.rowvec) addend += pa.rowvec[(long long)n * pa.rowvec_bstride + co];
      }
      addv[tid] = addend;
    }
  }
  // operand read addresses (bytes from smem): lane base per tap column; lines / planes / ring slots are immediates
  int xa[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) xa[kw] = b_lane_base(wave, lane, kw);
  const int wa0 = a_lane_base(lane);
  f32x16_t acc[2][2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][mb][r] = 0.f;
  MW_STAMP(56);
  dma_wait<0>();
  __builtin_amdgcn_s_barrier();
  MW_STAMP(2);

  // ---- main loop: per half-chunk 9 groups of 3 taps; group g reads ring slot g % 3 (9 % 3 == 0: compile-time) -------------------------------
  // Two operand register sets, software-pipelined over the taps and over the group barrier (conv_dma.hip PIPE2): tap k+1's four reads are issued
  // under tap k's four MFMAs; the next group's first tap is read right after the barrier that publishes its panel.
  uint4 xf[2][2], wf[2][2];
  auto read_tap = [&](int g, int u, int set) __attribute__((always_inline)) {
    const int tap = g * G + u;
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) wf[set][nb] = *reinterpret_cast<const uint4*>(smem + wa0 + a_offset(g % RING, u, nb));
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) xf[set][mb] = *reinterpret_cast<const uint4*>(smem + xa[kw] + b_offset(mb, kd, kh));
  };
  auto mma_tap = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) mma32(wf[set][nb], xf[set][mb], acc[nb][mb]);
  };
  constexpr int NRD = 4, NMMA = 4;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const bool last_chunk = chunk + 1 == nchunks;
#pragma unroll
    for (int g = 0; g < NGROUPS; ++g) {
      const int X = (g * G) & 1, Y = X ^ 1;                                  // operand set of a tap = (tap index) & 1
      const int LASTSET = (g * G + G - 1) & 1, NEXTSET = ((g + 1) * G) & 1;  // ... of the group's last tap / the next group's first
      if (g == 0) read_tap(0, 0, X);  // (a chunk's first group reads its own tap 0: the patch has just been replaced)
      read_tap(g, 1, Y);
      mma_tap(X);
      if (g == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2 * NRD, 0);   // the operand reads first, then the tap's MFMAs
      else __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NMMA, 0);
      // panel t+2 goes into the ring slot group t-1 read from (every wave is past the barrier that ended it)
      if (g < NGROUPS - 2) issue_w(chunk, g + 2, (g + 2) % RING);
      else if (!last_chunk) issue_w(chunk + 1, g + 2 - NGROUPS, (g + 2) % RING);
      read_tap(g, 2, X);
      mma_tap(Y);
      __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NMMA, 0);
      if (g == NGROUPS - 1) {
        if (!last_chunk) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();  // every wave is done with this half-chunk's patch
          issue_patch(chunk + 1);
          dma_wait<0>();                 // patch + the two panels in flight
          __builtin_amdgcn_s_barrier();
        }
      } else {
        // panel t+1 (issued a group ago) must have landed; panel t+2 (WPW instructions, just issued) may stay in flight.  The wait also retires
        // every LDS read of the group: the barrier releases other waves to DMA into the ring slot this group read.
        if (g < NGROUPS - 2 || !last_chunk) dma_wait<WPW>(); else dma_wait<0>();
        __builtin_amdgcn_s_barrier();
        read_tap(g + 1, 0, NEXTSET);
      }
      mma_tap(LASTSET);
      if (g < NGROUPS - 1) {
        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NMMA, 0);
      }
      if (chunk < 5) MW_STAMP(3 + chunk * 10 + g);
    }
  }
  MW_STAMP(60);

  const EpTile et = {n, od0, oh0, ow0, cb * BN, 0};
  EpRows<8> rows0;
  OPAQUE_LANE(lane_e);
  dma_epilogue_rows<T, 4, 3, 0>(cold_desc(), et, wave * 4, lane_e, rows0);  // residual rows: requested now, used after the transpose

  // ---- fused 1x1 shortcut convolution: extra half-chunks over the (virtually concatenated) skip sources, centre tap only -------------------------
  // Rounds of up to 4 half-chunks: each wave DMAs the 32-byte channel slices of ITS OWN 64 output voxels (2 pieces per half-chunk) and wave j
  // the 2-piece weight panel of the round's half-chunk j; one wait + barrier, then 4 MFMAs per half-chunk.
  KDesc& ps = cold_desc();
  if (ps.skip_x[0]) {
    const int nsc0 = ps.skip_cin[0] / BKH, nsc = nsc0 + (ps.skip_x[1] ? ps.skip_cin[1] / BKH : 0);
    OPAQUE_LANE(lane_k);
    int svox[2];  // output voxel of this lane's rows (piece h: row 64 wave + 32 h + lane / 2), -1 outside the volume
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int m = 32 * h + (lane_k >> 1);  // within the wave's plane: line m / 16, column m % 16
      const int od = od0 + wave, oh = oh0 + (m >> 4), ow = ow0 + (m & 15);
      svox[h] = (od < ps.Do && oh < ps.Ho && ow < ps.Wo) ? ((n * ps.Do + od) * ps.Ho + oh) * ps.Wo + ow : -1;
    }
    const int xslot = sc_x_lane_slot(lane_k) << 4;
    const int wrow = lane_k >> 1;                       // weight row of this lane within a piece; piece h = rows 32 h ..
    const int wslot = ((lane_k & 1) ^ row_swz(wrow)) << 4;
    const int cbl = sc_b_lane_base(wave, lane_k), cal = sc_a_lane_base(lane_k);
    const char* wsk = reinterpret_cast<const char*>(ps.skip_w);
    const char* zk = reinterpret_cast<const char*>(gm_mw_zero_row) + ((lane_k & 1) << 4);
    for (int sc0 = 0; sc0 < nsc; sc0 += SC_ROUND) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // patch buffer and ring are free
#pragma unroll
      for (int j = 0; j < SC_ROUND; ++j) {
        const int sc = sc0 + j;
        if (sc < nsc) {  // wave-uniform
          const int part = sc >= nsc0 ? 1 : 0, cip = sc - (part ? nsc0 : 0);
          const char* xb = reinterpret_cast<const char*>(ps.skip_x[part]) + (long long)cip * ROWB + xslot;
          const long long rowb = ps.skip_ld[part] * 2;
#pragma unroll
          for (int h = 0; h < 2; ++h) dma16(svox[h] >= 0 ? xb + svox[h] * rowb : zk, lds0 + (unsigned)sc_x_piece_dst(j, wave, h));
          if (wave == j) {
            const char* wpan = wsk + (long long)(sc >> 1) * cout_pad * SRC_ROWB + (sc & 1) * ROWB + wslot;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int wco = cb * BN + 32 * h + wrow;
              dma16(wco < cout_pad ? wpan + (long long)wco * SRC_ROWB : zk, lds0 + (unsigned)sc_w_piece_dst(j, h));
            }
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int j = 0; j < SC_ROUND; ++j) {
        if (sc0 + j < nsc) {
          uint4 sx[2], sw[2];
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) sx[mb] = *reinterpret_cast<const uint4*>(smem + cbl + sc_b_offset(j, mb));
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) sw[nb] = *reinterpret_cast<const uint4*>(smem + cal + sc_a_offset(j, nb));
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) mma32(sw[nb], sx[mb], acc[nb][mb]);
        }
      }
    }
  }
  MW_STAMP(61);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // every wave is done with the operand buffers: the transpose scratch overlays them
  MW_STAMP(57);

  // ---- epilogue: accumulators + addend -> wave-private LDS scratch (row = voxel) -> 16-byte row stores, fused GroupNorm statistics ------------
  KDesc& pe = cold_desc();
#ifdef GM_CONV_ABLATE
  if (pe.debug_flags & 256) return;
#endif
  float st_s[1][8], st_q[1][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { st_s[0][i] = 0.f; st_q[0][i] = 0.f; }
  char* scratch = smem + (size_t)wave * SCRATCH_WAVE;
  {
    const int hi = lane_e >> 5;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) {
        const int ch = nb * 32 + 8 * jq + 4 * hi;  // = nb * 32 + acc_channel(lane, 4 jq)
        const float4 add = *reinterpret_cast<const float4*>(addv + ch);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          char* dst = scratch + acc_row(mb, lane_e) * 144 + ch * 2;
          const float o0 = acc[nb][mb][4 * jq] + add.x, o1 = acc[nb][mb][4 * jq + 1] + add.y, o2 = acc[nb][mb][4 * jq + 2] + add.z,
                      o3 = acc[nb][mb][4 * jq + 3] + add.w;
          *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
        }
      }
    }
  }
  dma_epilogue_store<T, 4, 3, 0, 1>(pe, scratch, et, wave * 4, lane_e, rows0, st_s, st_q);
  MW_STAMP(62);
  if (pe.stats) {
    // lane sums over its rows -> sum over the 8 row lanes of a segment (registers) -> one partial per (wave, channel) in the wave's own scratch
    // block -> fixed-order fp64 sum over the waves: deterministic, one plain store per (tile, channel)
    OPAQUE_LANE(lane_s);
    float ra[8], rb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { ra[i] = wave_segment_sum(st_s[0][i]); rb[i] = wave_segment_sum(st_q[0][i]); }
    if (lane_s < 8) {
      float* part = reinterpret_cast<float*>(scratch);
#pragma unroll
      for (int i = 0; i < 8; i += 2) *reinterpret_cast<float4*>(part + 2 * (lane_s * 8 + i)) = make_float4(ra[i], rb[i], ra[i + 1], rb[i + 1]);
    }
    __syncthreads();
    const int ch = wave * 64 + lane_s;
    if (ch < BN) {
      double a = 0.0, b2 = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float2 v = *reinterpret_cast<const float2*>(smem + w * SCRATCH_WAVE + ch * 8);
        a += (double)v.x;
        b2 += (double)v.y;
      }
      const int co = cb * BN + ch;
      if (co < pe.Cout) {
        const long long slot = (long long)(td_i * nth + th_i) * ntw + tw_i;  // the tile within its sample
        double* dst = pe.stats + ((slot * pe.N + n) * pe.Cout + co) * 2;
        *reinterpret_cast<double2*>(dst) = make_double2(a, b2);
      }
    }
  }
  MW_STAMP(63);
}

extern "C" long long gm_conv_mw_lds_bytes() { return mw::LDS_BYTES; }

// geometry this kernel covers: bf16, 3x3x3, stride 1, direct input, no prologue / split-K / output activation beyond the epilogue's forms
extern "C" int gm_conv_mw_eligible(const GmConvDesc* d) {
  const int vecw = 8;
  return d->dtype == GM_BF16 && d->kd == 3 && d->kh == 3 && d->kw == 3 && d->sd == 1 && d->sh == 1 && d->sw == 1 && d->dd == 1 && d->dh == 1 &&
         d->dw == 1 && d->in_mode == 0 && d->Cin % mw::BKH == 0 && d->x_ld % vecw == 0 && (reinterpret_cast<uintptr_t>(d->x) & 15) == 0 &&
         d->pre_scale == nullptr && d->pre_shift == nullptr && d->pre_act == 0 &&
         (d->x2 == nullptr || (d->cin_split > 0 && d->cin_split < d->Cin && d->cin_split % mw::BKH == 0 && d->x2_ld % vecw == 0 &&
                               (reinterpret_cast<uintptr_t>(d->x2) & 15) == 0)) &&
         d->ltd == 2 && d->lth == 2 && d->ltw == 4 && d->Cout % vecw == 0 && d->y_ld % vecw == 0 && (reinterpret_cast<uintptr_t>(d->y) & 15) == 0 &&
         (!d->res || (d->res_ld % vecw == 0 && (reinterpret_cast<uintptr_t>(d->res) & 15) == 0)) && !(d->ksplit > 1 && d->kpartial) &&
         (long long)d->N * d->Ds * d->Hs * d->Ws < (1LL << 31) && (long long)d->N * d->Do * d->Ho * d->Wo < (1LL << 31) &&
         (!d->skip_x[0] ||
          (d->skip_w && d->skip_cin[0] > 0 && d->skip_cin[0] % mw::BKH == 0 && d->skip_ld[0] % vecw == 0 &&
           (reinterpret_cast<uintptr_t>(d->skip_x[0]) & 15) == 0 &&
           (!d->skip_x[1] || (d->skip_cin[1] > 0 && d->skip_cin[1] % mw::BKH == 0 && d->skip_ld[1] % vecw == 0 &&
                              (reinterpret_cast<uintptr_t>(d->skip_x[1]) & 15) == 0))));
}

extern "C" int gm_conv_mw_launch(const GmConvDesc* dp, unsigned nblocks, void* stream) {
  static bool attr_set = false;
  if (dp->dtype != GM_BF16) return -2;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) (void)hipGetLastError();
    attr_set = true;
  }
  conv_mw_kernel<<<dim3(nblocks), 256, (size_t)mw::LDS_BYTES, (hipStream_t)stream>>>(*dp);
  return 0;
}
