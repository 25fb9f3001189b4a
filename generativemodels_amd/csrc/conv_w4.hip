// 3x3x3 stride-1 bf16 convolution on 512-voxel tiles with v_mfma_f32_32x32x16_bf16, FOUR waves of 4 x 2 MFMA blocks (tile configuration 23, round 5).
//
// Why (profiles/r05_mfma_energy.txt, r05_clock_energy.json): the convolutions run at the package power cap, so time = joules / cap, and cfg 14's
// tap loop -- v_mfma_f32_16x16x32_bf16, 8 operand reads per 16 MFMAs -- costs 0.954 pJ/FLOP at the instruction level (measured in the kernel:
// 0.995).  The 32x32x16 shape reads half the operand registers per FLOP: 0.713 pJ/FLOP register-resident against 0.915-0.94, and an LDS-fed loop of
// 4 x 2 blocks per wave (6 reads per 8 MFMAs) 0.830 -- 13 % under cfg 14's floor; configuration 22 (2 x 2 blocks, 8 waves at 128 registers, one
// operand set: 0.911) was too close to cfg 14 to beat it past its larger fixed costs.  This kernel keeps configuration 22's LDS image
// (conv_w8_index.h: 512-voxel tiles, 64-byte patch rows under a per-column bank key, weights in 16-channel halves through a two-slot ring,
// 79.75 KiB = two work-groups per CU) and changes the ownership: 4 waves, each two depth planes = 128 voxels x 64 channels, 128 accumulator
// registers + two operand sets of 24 at 256 registers per wave (two waves per SIMD, as cfg 14).
// Index arithmetic: conv_w4_index.h, replayed on the host by tests/emulate_conv_w4.cpp.  bf16 only; stride 1, 3x3x3, direct input, second input
// source and fused 1x1 shortcut as configuration 22; no fused prologue, no split-K.  Reference: diffusion_model_unet.py:669-696, autoencoderkl.py:96-122.
#include "conv_dma_shared.h"
#include "conv_w4_index.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __attribute__((aligned(64))) unsigned int gm_w4_zero_row[16] = {0};  // the source of every padding row (this TU's own: no RDC)

__device__ __forceinline__ void w4_mma(const uint4& a, const uint4& b, f32x16_t& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

#ifdef GM_CONV_TIMELINE
#define W4_STAMP(k)                                                                                                            \
  do {                                                                                                                         \
    if ((p.debug_flags & 4096) && threadIdx.x == 0)                                                                            \
      reinterpret_cast<unsigned long long*>(p.kpartial)[(long long)blockIdx.x * 64 + (k)] = __builtin_readcyclecounter();      \
  } while (0)
#else
#define W4_STAMP(k)
#endif

// PIPE2: two operand register sets, software-pipelined over the taps and the group barrier (the default here: 256 registers hold them)
template <bool PIPE2>
__global__ __launch_bounds__(256, 2) void conv_w4_kernel(const GmConvDesc p) {
  using namespace w4;
  typedef bf16_raw T;
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // [patch 67.5 KiB][2 weight panels x 6 KiB][addend vector 256 B]
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- the tile of this work-group: XCD x owns a contiguous range of the tile list (neighbouring halo patches meet in one L2) -------------
  const int ntd = (p.Do + TD - 1) / TD, nth = (p.Ho + TH - 1) / TH, ntw = (p.Wo + TW - 1) / TW;
  const int ncb = (p.Cout + BN - 1) / BN;
  W4_STAMP(0);
  unsigned b = xcd_remap(blockIdx.x, gridDim.x);
  const int cb = b % ncb; b /= ncb;
  const int tw_i = b % ntw; b /= ntw;
  const int th_i = b % nth; b /= nth;
  const int td_i = b % ntd; b /= ntd;
  const int n = (int)b;
  const int od0 = td_i * TD, oh0 = th_i * TH, ow0 = tw_i * TW;
  const int nchunks = p.Cin / BK;                          // host-checked: Cin % 32 == 0
  const int nchunks0 = p.x2 ? p.cin_split / BK : nchunks;  // chunks of the first source (virtual concatenation: cin_split % 32 == 0)
  const int cout_pad = (p.Cout + 15) & ~15;
  const char* zero = reinterpret_cast<const char*>(gm_w4_zero_row) + ((lane & 3) << 4);

  // ---- patch placement: wave w moves pieces w, w + 4, ... (17 per wave); this lane's row advances by 64 = three lines + ten columns per piece ---------
  // Two registers per lane across the tap loop: pv0 = source voxel of this lane's row of piece 0 (plain arithmetic, also when that row is
  // padding) and its patch coordinates packed (pd | ph << 8 | pw << 16).  A chunk's request walks the 17 pieces incrementally: two carries, one
  // voxel add, the column's bank key -- ~14 instructions per piece; the bounds tests run only for tiles at the volume's surface (wave-uniform).
  constexpr int PPW = NPIECES / NW;  // 17
  static_assert(NPIECES % NW == 0, "every wave moves the same number of pieces (the last one of wave 3 is the half piece)");
  const int plane_vox = p.Hs * p.Ws;
  const int pstep = 3 * p.Ws + 10, pstep_w = p.Ws - LINE, pstep_h = plane_vox - PH * p.Ws;  // + 64 rows; column carry; line carry
  int pv0, pc0;
  bool interior;
  int ud0, uh0, uw0;
  {
    KDesc& pk = cold_desc();
    const PatchRow r0 = patch_row(PIECE_ROWS * wave + (lane >> 2));
    ud0 = od0 - pk.pd; uh0 = oh0 - pk.ph; uw0 = ow0 - pk.pw;  // source coordinates of patch row (0, 0, 0)
    pv0 = ((n * pk.Ds + (ud0 + r0.pd)) * pk.Hs + (uh0 + r0.ph)) * pk.Ws + (uw0 + r0.pw);
    pc0 = r0.pd | (r0.ph << 8) | (r0.pw << 16);
    interior = ud0 >= 0 && ud0 + PD <= pk.Ds && uh0 >= 0 && uh0 + PH <= pk.Hs && uw0 >= 0 && uw0 + PW <= pk.Ws;  // wave-uniform
  }
  auto issue_patch = [&](int chunk) __attribute__((always_inline)) {
#ifdef GM_CONV_ABLATE
    if (p.debug_flags & 1024) return;
#endif
    const bool second = chunk >= nchunks0;  // wave-uniform
    const char* cbase = second ? reinterpret_cast<const char*>(p.x2) + (long long)(chunk - nchunks0) * ROWB : reinterpret_cast<const char*>(p.x) + (long long)chunk * ROWB;
    const long long rowb = (second ? p.x2_ld : p.x_ld) * 2;
    int pv = pv0, pc = pc0;
    asm volatile("" : "+v"(pv), "+v"(pc));  // opaque: nothing derived from the placement stays live across the tap loop
    int pd = pc & 255, ph = (pc >> 8) & 255, pw = pc >> 16;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int piece = wave + NW * j;  // wave-uniform
      if (j > 0) {
        pw += 10;
        const bool cw = pw >= LINE;
        pw -= cw ? LINE : 0;
        ph += 3 + (cw ? 1 : 0);
        const bool ch = ph >= PH;
        ph -= ch ? PH : 0;
        pd += ch ? 1 : 0;
        pv += pstep + (cw ? pstep_w : 0) + (ch ? pstep_h : 0);
      }
      bool ok = true;
      if (!interior) ok = (unsigned)(ud0 + pd) < (unsigned)p.Ds && (unsigned)(uh0 + ph) < (unsigned)p.Hs && (unsigned)(uw0 + pw) < (unsigned)p.Ws;
      const char* src = ok ? cbase + pv * rowb + (patch_lane_quarter(lane, pw) << 4) : zero;
      if (piece < NPIECES - 1 || lane < 32) dma16(src, lds0 + (unsigned)patch_piece_dst(piece));  // (the last piece is 8 rows)
    }
  };
  // ---- weight panels: six 1 KiB pieces (piece q = tap q >> 1, output channels 32 (q & 1) ..): wave w moves pieces w and, for w < 2, w + 4 -----------
  int wsrc[2];  // byte offset of this lane's 16 bytes from the panel's first row, or -1 beyond cout_pad / no such piece
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int piece = wave + NW * h;
    const WLane f = wpanel_lane(piece < 6 ? piece : 0, lane);
    const int co = cb * BN + f.co;
    wsrc[h] = (piece < 6 && co < cout_pad) ? (int)whalves_offset(0, 0, f.tap, co, cout_pad, f.slot) : -1;
  }
  const char* wbase = reinterpret_cast<const char*>(p.w);
  auto issue_w = [&](int chunk, int half, int g, int slot) __attribute__((always_inline)) {  // panel (chunk, half, group g) -> ring slot
#ifdef GM_CONV_ABLATE
    if (p.debug_flags & 512) return;
#endif
    const char* panel = wbase + (((long long)chunk * 2 + half) * 27 + G * g) * cout_pad * WROWB;
    dma16(wsrc[0] >= 0 ? panel + wsrc[0] : zero, lds0 + (unsigned)wpanel_piece_dst(slot, wave));
    if (wave < 2) dma16(wsrc[1] >= 0 ? panel + wsrc[1] : zero, lds0 + (unsigned)wpanel_piece_dst(slot, wave + NW));
  };

  // ---- prologue: first patch, first panel, the epilogue addend ----------------------------------------------------------------------------------
  issue_patch(0);
  issue_w(0, 0, 0, 0);
  W4_STAMP(55);
  float* addv = reinterpret_cast<float*>(smem + ADDV_OFF);
  {
    KDesc& pa = cold_desc();
    if (tid < BN) {  // bias + shortcut bias + timestep row (this order), fp32: the three loads go out together through substitute addresses
      const int co = cb * BN + tid;
      const bool in = co < pa.Cout;
      const float* dummy = reinterpret_cast<const float*>(gm_w4_zero_row);
      const float b0 = (pa.bias && in ? pa.bias + co : dummy)[0];
      const float b1 = (pa.skip_bias && in ? pa.skip_bias + co : dummy)[0];
      const float b2 = (pa.rowvec && in ? pa.rowvec + (long long)n * pa.rowvec_bstride + co : dummy)[0];
      float addend = 0.f;
      if (pa.bias && in) addend += b0;
      if (pa.skip_bias && in) addend += b1;
      if (pa.rowvec && in) addend += b2;
      addv[tid] = addend;
    }
  }
  // operand read addresses (bytes from smem): lane base per tap column; lines / planes / ring slots are immediates
  int xa[3];  // (channel half 1 = the same row with bit 5 of the address flipped: the key is XORed onto the slot)
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) xa[kw] = b_lane_base(wave, lane, kw, 0);
  const int wa0 = a_lane_base(lane);
  constexpr int MB = 4;  // voxel blocks per wave (two planes x two line pairs)
  f32x16_t acc[2][MB];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][mb][r] = 0.f;
  W4_STAMP(56);
  dma_wait<0>();
  __builtin_amdgcn_s_barrier();
  W4_STAMP(2);

  // ---- main loop: per 32-channel chunk 2 halves x 9 groups of 3 taps; group gg = 9 half + g reads ring slot gg & 1 (18 groups: compile-time) ---
  // Every group: request panel gg + 1 into the slot group gg - 1 read (every wave is past the barrier that ended it), three taps of 6 reads +
  // 8 MFMAs, wait for the panel, barrier.  A chunk ends with: barrier (every wave is done with the patch), patch request, wait, [transform], barrier.
  constexpr int NRD = 2 + MB, NMMA = 2 * MB, NGG = 2 * NGROUPS;
  auto read_ops = [&](int gg, int u, uint4 (&wfr)[2], uint4 (&xfr)[MB]) __attribute__((always_inline)) {
    const int half = gg / NGROUPS, tap = (gg % NGROUPS) * G + u;
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) wfr[nb] = *reinterpret_cast<const uint4*>(smem + wa0 + a_offset(gg % RING, u, nb));
    const int xb = half ? (xa[kw] ^ 32) : xa[kw];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) xfr[mb] = *reinterpret_cast<const uint4*>(smem + xb + b_offset(mb, kd, kh));
  };
  auto mma_ops = [&](const uint4 (&wfr)[2], const uint4 (&xfr)[MB]) __attribute__((always_inline)) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) w4_mma(wfr[nb], xfr[mb], acc[nb][mb]);
  };
  auto next_panel = [&](int chunk, int gg, bool last_chunk) __attribute__((always_inline)) {
    if (gg < NGG - 1) issue_w(chunk, (gg + 1) / NGROUPS, (gg + 1) % NGROUPS, (gg + 1) % RING);
    else if (!last_chunk) issue_w(chunk + 1, 0, 0, 0);
  };
  auto chunk_boundary = [&](int chunk) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done with this chunk's patch
    issue_patch(chunk + 1);
    dma_wait<0>();                 // patch + the next chunk's first panel
    __builtin_amdgcn_s_barrier();
  };
  if constexpr (PIPE2) {
    // two operand sets: tap k+1's four reads are issued under tap k's four MFMAs; the next group's first tap is read right after the barrier
    // that publishes its panel
    uint4 xf[2][MB], wf[2][2];
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      const bool last_chunk = chunk + 1 == nchunks;
#pragma unroll
      for (int half_ = 0; half_ < 2; ++half_)
#pragma unroll
      for (int g_ = 0; g_ < NGROUPS; ++g_) {
        const int gg = half_ * NGROUPS + g_;
        const int X = (gg * G) & 1, Y = X ^ 1;                                    // operand set of a tap = (tap index within the chunk) & 1
        const int LASTSET = (gg * G + G - 1) & 1, NEXTSET = ((gg + 1) * G) & 1;   // ... of the group's last tap / the next group's first
        if (gg == 0) read_ops(0, 0, wf[X], xf[X]);  // (a chunk's first group reads its own tap 0: the patch has just been replaced)
        read_ops(gg, 1, wf[Y], xf[Y]);
        mma_ops(wf[X], xf[X]);
        if (gg == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2 * NRD, 0);     // the operand reads first, then the tap's MFMAs
        else __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NMMA, 0);
        next_panel(chunk, gg, last_chunk);
        read_ops(gg, 2, wf[X], xf[X]);
        mma_ops(wf[Y], xf[Y]);
        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NMMA, 0);
        if (gg == NGG - 1) {
          mma_ops(wf[LASTSET], xf[LASTSET]);
          if (!last_chunk) chunk_boundary(chunk);
        } else {
          // the panel issued at the top of this group must have landed; the wait also retires every LDS read of the group: the barrier
          // releases other waves to DMA into the ring slot this group read
          dma_wait<0>();
          __builtin_amdgcn_s_barrier();
          read_ops(gg + 1, 0, wf[NEXTSET], xf[NEXTSET]);
          mma_ops(wf[LASTSET], xf[LASTSET]);
          __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, NMMA, 0);
        }
        if (chunk < 2) W4_STAMP(3 + chunk * 18 + gg);
      }
    }
  } else {
    uint4 xf[MB], wf[2];
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      const bool last_chunk = chunk + 1 == nchunks;
#pragma unroll
      for (int half_ = 0; half_ < 2; ++half_)
#pragma unroll
      for (int g_ = 0; g_ < NGROUPS; ++g_) {
        const int gg = half_ * NGROUPS + g_;
        next_panel(chunk, gg, last_chunk);
#pragma unroll
        for (int u = 0; u < G; ++u) {
          read_ops(gg, u, wf, xf);
          mma_ops(wf, xf);
        }
        if (gg == NGG - 1) {
          if (!last_chunk) chunk_boundary(chunk);
        } else {
          dma_wait<0>();
          __builtin_amdgcn_s_barrier();
        }
        if (chunk < 2) W4_STAMP(3 + chunk * 18 + gg);
      }
    }
  }
  W4_STAMP(60);

  const EpTile et = {n, od0, oh0, ow0, cb * BN, 0};
  constexpr int WL = 4 * WAVE_PLANES;  // W lines per wave
  EpRows<2 * WL> rows0;
  OPAQUE_LANE(lane_e);
  dma_epilogue_rows<T, WL, 3, 0>(cold_desc(), et, wave * WL, lane_e, rows0);  // residual rows: requested now, used after the transpose

  // ---- fused 1x1 shortcut convolution: extra 32-channel chunks over the (virtually concatenated) skip sources, centre tap only --------------------
  // Rounds of up to 2 chunks: each wave DMAs the 64-byte channel slices of ITS OWN 128 output voxels (8 pieces per chunk) and piece `wave` of the
  // 4-piece weight panel of each chunk of the round (rows of the standard packed image); one wait + barrier, then 2 halves x 8 MFMAs per chunk.
  KDesc& ps = cold_desc();
  if (ps.skip_x[0]) {
    const int nsc0 = ps.skip_cin[0] / BK, nsc = nsc0 + (ps.skip_x[1] ? ps.skip_cin[1] / BK : 0);
    OPAQUE_LANE(lane_k);
    const char* wsk = reinterpret_cast<const char*>(ps.skip_w);
    const char* zk = reinterpret_cast<const char*>(gm_w4_zero_row) + ((lane_k & 3) << 4);
    for (int sc0 = 0; sc0 < nsc; sc0 += SC_ROUND) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // patch buffer and ring are free
#pragma unroll
      for (int j = 0; j < SC_ROUND; ++j) {
        const int sc = sc0 + j;
        if (sc < nsc) {  // wave-uniform
          const int part = sc >= nsc0 ? 1 : 0, cip = sc - (part ? nsc0 : 0);
          const char* xb = reinterpret_cast<const char*>(ps.skip_x[part]) + (long long)cip * ROWB;
          const long long rowb = ps.skip_ld[part] * 2;
#pragma unroll
          for (int h = 0; h < 2 * WL / 2; ++h) {
            const int m = 16 * h + (lane_k >> 2);  // within the wave's two planes: plane m / 64, line (m / 16) % 4, column m % 16
            const int od = od0 + WAVE_PLANES * wave + (m >> 6), oh = oh0 + ((m >> 4) & 3), ow = ow0 + (m & 15);
            const bool ok = od < ps.Do && oh < ps.Ho && ow < ps.Wo;
            const int vox = ((n * ps.Do + od) * ps.Ho + oh) * ps.Wo + ow;
            dma16(ok ? xb + vox * rowb + (sc_x_lane_quarter(lane_k, h) << 4) : zk, lds0 + (unsigned)sc_x_piece_dst(j, wave, h));
          }
          {
            const int h = wave;
            const int wco = cb * BN + 16 * h + (lane_k >> 2);
            const char* wpan = wsk + ((long long)sc * cout_pad + wco) * ROWB + (sc_w_lane_quarter(lane_k, h) << 4);
            dma16(wco < cout_pad ? wpan : zk, lds0 + (unsigned)sc_w_piece_dst(j, h));
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int j = 0; j < SC_ROUND; ++j) {
        if (sc0 + j < nsc) {
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            uint4 sx[MB], sw[2];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) sx[mb] = *reinterpret_cast<const uint4*>(smem + sc_b_lane_base(wave, lane_k, half) + sc_b_offset(j, mb));
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) sw[nb] = *reinterpret_cast<const uint4*>(smem + sc_a_lane_base(lane_k, half) + sc_a_offset(j, nb));
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
              for (int mb = 0; mb < MB; ++mb) w4_mma(sw[nb], sx[mb], acc[nb][mb]);
          }
        }
      }
    }
  }
  W4_STAMP(61);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // every wave is done with the operand buffers: the transpose scratch overlays them
  W4_STAMP(57);

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
        for (int mb = 0; mb < MB; ++mb) {
          char* dst = scratch + acc_row(mb, lane_e) * 144 + ch * 2;
          const float o0 = acc[nb][mb][4 * jq] + add.x, o1 = acc[nb][mb][4 * jq + 1] + add.y, o2 = acc[nb][mb][4 * jq + 2] + add.z,
                      o3 = acc[nb][mb][4 * jq + 3] + add.w;
          *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
        }
      }
    }
  }
  dma_epilogue_store<T, WL, 3, 0, 1>(pe, scratch, et, wave * WL, lane_e, rows0, st_s, st_q);
  W4_STAMP(62);
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
  W4_STAMP(63);
}

extern "C" long long gm_conv_w4_lds_bytes() { return w4::LDS_BYTES; }

// geometry this kernel covers: configuration 22's without the fused prologue.  GmConvDesc.w must point at the HALVES image of the weights
// ([chunk32][half][tap][Cout_pad][16]); the shortcut's skip_w at the standard one.
extern "C" int gm_conv_w4_eligible(const GmConvDesc* d) {
  const int vecw = 8;
  return d->dtype == GM_BF16 && d->kd == 3 && d->kh == 3 && d->kw == 3 && d->sd == 1 && d->sh == 1 && d->sw == 1 && d->dd == 1 && d->dh == 1 &&
         d->dw == 1 && d->in_mode == 0 && d->Cin % w4::BK == 0 && d->x_ld % vecw == 0 && (reinterpret_cast<uintptr_t>(d->x) & 15) == 0 &&
         d->pre_scale == nullptr && d->pre_shift == nullptr && d->pre_act == 0 &&
         (d->x2 == nullptr || (d->cin_split > 0 && d->cin_split < d->Cin && d->cin_split % w4::BK == 0 && d->x2_ld % vecw == 0 &&
                               (reinterpret_cast<uintptr_t>(d->x2) & 15) == 0)) &&
         d->ltd == 3 && d->lth == 2 && d->ltw == 4 && d->Cout % vecw == 0 && d->y_ld % vecw == 0 && (reinterpret_cast<uintptr_t>(d->y) & 15) == 0 &&
         (!d->res || (d->res_ld % vecw == 0 && (reinterpret_cast<uintptr_t>(d->res) & 15) == 0)) && !(d->ksplit > 1 && d->kpartial) &&
         (long long)d->N * d->Ds * d->Hs * d->Ws < (1LL << 31) && (long long)d->N * d->Do * d->Ho * d->Wo < (1LL << 31) &&
         (!d->skip_x[0] ||
          (d->skip_w && d->skip_cin[0] > 0 && d->skip_cin[0] % w4::BK == 0 && d->skip_ld[0] % vecw == 0 &&
           (reinterpret_cast<uintptr_t>(d->skip_x[0]) & 15) == 0 &&
           (!d->skip_x[1] || (d->skip_cin[1] > 0 && d->skip_cin[1] % w4::BK == 0 && d->skip_ld[1] % vecw == 0 &&
                              (reinterpret_cast<uintptr_t>(d->skip_x[1]) & 15) == 0))));
}

static int g_w4_pipe2 = 1;  // two operand sets (256 registers per wave hold them); 0 = one set (bench A/B)
extern "C" void gm_conv_w4_set_pipe2(int on) { g_w4_pipe2 = on; }

template <bool PIPE2>
static void launch_w4(const GmConvDesc& d, unsigned nblocks, hipStream_t st) {
  static bool attr_set = false;
  auto kern = conv_w4_kernel<PIPE2>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) (void)hipGetLastError();
    attr_set = true;
  }
  kern<<<dim3(nblocks), 64 * w4::NW, (size_t)w4::LDS_BYTES, st>>>(d);
}

extern "C" int gm_conv_w4_launch(const GmConvDesc* dp, unsigned nblocks, void* stream) {
  if (dp->dtype != GM_BF16) return -2;
  hipStream_t st = (hipStream_t)stream;
  if (g_w4_pipe2) launch_w4<true>(*dp, nblocks, st); else launch_w4<false>(*dp, nblocks, st);
  return 0;
}
