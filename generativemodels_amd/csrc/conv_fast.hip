// The stride-1 hot path of the implicit-GEMM convolution (3^d / 1^d kernels, optional fused nearest-2x up-sampling): the
// ResnetBlock / Upsample / attention-projection convolutions that carry >90 % of a DiffusionModelUNet forward.
//
// Same data flow as conv.hip (halo patch of the output tile staged once per 64-byte input-channel chunk with the fused
// GroupNorm+SiLU prologue; weights streamed through LDS; A = weights, B = activations, 16x16 MFMA), restructured around what
// rocprof showed on MI355X -- the generic kernel spent its time waiting for one L2 round trip per tap:
//   * taps are processed in groups of 3 per barrier (48 MFMAs per wave between barriers instead of 16) and the next group's
//     weight panel is prefetched global->registers at the top of the group, written to the alternate LDS buffer after the MFMAs;
//   * the patch addresses (division-heavy, chunk-independent) are computed once per work-group and kept in registers; the
//     patch loads of a chunk are issued in batches of 4 before any of them is consumed;
//   * work-group ids are remapped so each XCD (private L2) owns a contiguous range of tiles;
//   * tile = 256 voxels (4x8x8 in 3-D) x 64 channels with 4 waves (2 work-groups / CU), or x 128 channels with 8 waves.
#include "conv_epilogue.h"

// LDS operand rows are 64 bytes (one K chunk) with NO padding; the 16-byte slot a (row, slot) pair lives in is
// slot ^ fast_swz(row).  With 16 consecutive rows per MFMA fragment (tile width 16, or the weight rows) every
// ds_read_b128 lane group of gfx950 then touches 16 distinct 16-byte bank slots: conflict-free for any starting row
// (exhaustively checked against the lane-group table of MI355X_MICROARCH.md, tools/lds_swizzle_search.py).
#define FAST_ROWB 64
__device__ __forceinline__ int fast_swz(int row) { return (row ^ (row >> 1)) & 3; }

// bench-only ablation switches (tools/ablate_conv.py): compiled out of the production library
#ifdef GM_CONV_ABLATE
#define ABLATE(bit) (p.debug_flags & (bit))
#else
#define ABLATE(bit) false
#endif

// WM x WN waves, each owning MF voxel fragments (16 voxels) x 4 channel fragments; MINW = resident waves per SIMD the
// register allocation must allow (work-groups per CU x waves per work-group / 4)
template <typename T, int WM, int WN, int MF, int MINW>
__global__ __launch_bounds__(64 * WM * WN, MINW) void conv_fast_kernel(const GmConvDesc p) {
  constexpr int BK = ConvTraits<T>::BK;
  constexpr int VECW = ConvTraits<T>::VECW;
  constexpr int NT = 64 * WM * WN;
  constexpr int NFR = 4, G = 3;
  constexpr int BM = WM * MF * 16;
  constexpr int BN = WN * 64;
  constexpr int ROWS_PER_PASS = NT / 4;
  constexpr int MAX_ITEMS = NT == 256 ? 12 : (NT == 512 ? (BM == 512 ? 10 : 6) : (BM == 512 ? 5 : 3));  // patch rows per thread (host: P <= MAX_ITEMS * ROWS_PER_PASS)
  constexpr int A_BATCH = MINW >= 4 ? 2 : (MAX_ITEMS >= 10 ? (MAX_ITEMS + 1) / 2 : MAX_ITEMS);  // patch loads in flight per thread (128-VGPR variants: 3)
  constexpr bool PRECISE = sizeof(T) == 4;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int wm = wave % WM, wn = wave / WM;

  // ---- tile geometry (stride 1, dilation 1) ----------------------------------------------------------------------------
  const int td = 1 << p.ltd, th = 1 << p.lth, tw = 1 << p.ltw;
  const int ntd = (p.Do + td - 1) >> p.ltd, nth = (p.Ho + th - 1) >> p.lth, ntw = (p.Wo + tw - 1) >> p.ltw;
  const int ncb = (p.Cout + BN - 1) / BN;
  unsigned b = xcd_remap(blockIdx.x, gridDim.x);
  const int cb = b % ncb; b /= ncb;
  const unsigned tile_id = b;  // (n, td, th, tw): statistics slot = tile_id modulo the tiles per sample
  const int tw_i = b % ntw; b /= ntw;
  const int th_i = b % nth; b /= nth;
  const int td_i = b % ntd; b /= ntd;
  const int n = b;
  const int od0 = td_i << p.ltd, oh0 = th_i << p.lth, ow0 = tw_i << p.ltw;
  const int pD = td + p.kd - 1, pH = th + p.kh - 1, pW = tw + p.kw - 1;
  const int P = pD * pH * pW;
  int Dv = p.Ds, Hv = p.Hs, Wv = p.Ws;
  if (p.in_mode == 1) { Dv *= p.fd; Hv *= p.fh; Wv *= p.fw; }
  const int ud0 = od0 - p.pd, uh0 = oh0 - p.ph, uw0 = ow0 - p.pw;

  char* ldsA = smem;                              // [P][FAST_ROWB], slots swizzled
  char* ldsB = smem + (size_t)P * FAST_ROWB;      // [2][G][BN][FAST_ROWB], slots swizzled

  const int T_taps = p.kd * p.kh * p.kw;
  const int ngroups = (T_taps + G - 1) / G;
  const int nchunks = (p.Cin + BK - 1) / BK;
  const int cout_pad = (p.Cout + 15) & ~15;
  const int total_gsteps = nchunks * ngroups;

  // ---- per-thread patch rows: source voxel index (chunk independent), -1 = zero padding, -2 = no such row ---------------
  const int sq = tid & 3;
  int vox[MAX_ITEMS];
#pragma unroll
  for (int j = 0; j < MAX_ITEMS; ++j) {
    const int pv = (tid >> 2) + j * ROWS_PER_PASS;
    int v = -2;
    if (pv < P) {
      const int pc = pv % pW;
      const int t1 = pv / pW;
      const int pb = t1 % pH, pa = t1 / pH;
      int ud = ud0 + pa, uh = uh0 + pb, uw = uw0 + pc;
      const bool ok = (ud >= 0) & (ud < Dv) & (uh >= 0) & (uh < Hv) & (uw >= 0) & (uw < Wv);
      if (p.in_mode == 1) { ud /= p.fd; uh /= p.fh; uw /= p.fw; }
      v = ok ? ((n * p.Ds + ud) * p.Hs + uh) * p.Ws + uw : -1;
    }
    vox[j] = v;
  }

  // ---- per-lane LDS read offsets ---------------------------------------------------------------------------------------
  int arow64[MF];  // 64 x patch row of this lane's voxel at tap (0,0,0)
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int m = (wm * MF + mf) * 16 + l15;
    const int a = m >> (p.lth + p.ltw), bb = (m >> p.ltw) & (th - 1), c = m & (tw - 1);
    arow64[mf] = ((a * pH + bb) * pW + c) * FAST_ROWB;
  }
  int boff[NFR];  // weight rows: panel bases are multiples of 8 rows, so the swizzle term is a per-lane constant
#pragma unroll
  for (int nf = 0; nf < NFR; ++nf) {
    const int r = (wn * NFR + nf) * 16 + l15;
    boff[nf] = r * FAST_ROWB + ((q ^ fast_swz(r)) << 4);
  }

  f32x4_t acc[NFR][MF];
#pragma unroll
  for (int nf = 0; nf < NFR; ++nf)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- weight panels: group of G taps = G*BN*4 16-byte items, spread over the NT threads; prefetched TWO groups ahead into
  //      two register sets (a panel is consumed ~2 x 48 MFMAs after its loads were issued: covers an L2 round trip under load)
  constexpr int B_ITEMS = G * BN * 4;
  constexpr int B_PER_THREAD = (B_ITEMS + NT - 1) / NT;
  const char* wbase = reinterpret_cast<const char*>(p.w);
  uint4 bregA[B_PER_THREAD], bregB[B_PER_THREAD];
  auto load_b = [&](uint4 (&breg)[B_PER_THREAD], int gstep) __attribute__((always_inline)) {
    const int chunk = gstep / ngroups, grp = gstep - chunk * ngroups;
#pragma unroll
    for (int u = 0; u < B_PER_THREAD; ++u) {
      const int item = tid + u * NT;
      const int gi = item / (BN * 4), rem = item % (BN * 4);
      const int row = rem >> 2, qq = rem & 3;
      const int t = grp * G + gi, co = cb * BN + row;
      // branch-free: an `if (valid) load` makes hipcc branch around the load AND drain vmcnt(0) at the join, which killed the
      // prefetch distance (the panel was waited for in the very next tap).  Load from a clamped in-range address instead and
      // zero the result with a select at consumption time.
      const bool ok = (item < B_ITEMS) & (t < T_taps) & (co < cout_pad);
      const long long tt = ok ? (long long)(chunk * T_taps + t) : 0, cc = ok ? co : 0;
      const uint4 v = *reinterpret_cast<const uint4*>(wbase + ((tt * cout_pad + cc) * BK) * (long long)sizeof(T) + qq * 16);
      breg[u] = v;
    }
  };
  auto store_b = [&](const uint4 (&breg)[B_PER_THREAD], int gstep) __attribute__((always_inline)) {  // panel of group `gstep`
    const int buf = gstep & 1;
    const int grp = gstep % ngroups;
#pragma unroll
    for (int u = 0; u < B_PER_THREAD; ++u) {
      const int item = tid + u * NT;
      const int gi = item / (BN * 4), rem = item % (BN * 4);
      const int row = rem >> 2, qq = rem & 3;
      const bool ok = (grp * G + gi < T_taps) & (cb * BN + row < cout_pad);  // out-of-range rows / taps were loaded from a clamped address
      uint4 v = breg[u];
      v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
      if (item < B_ITEMS)
        *reinterpret_cast<uint4*>(ldsB + ((size_t)(buf * G + gi) * BN + row) * FAST_ROWB + ((qq ^ fast_swz(row)) << 4)) = v;
    }
  };

  // ---- patch staging with the fused prologue ---------------------------------------------------------------------------
  const T* xin = reinterpret_cast<const T*>(p.x);
  auto stage_a = [&](int chunk) __attribute__((always_inline)) {
    const int c0 = chunk * BK + sq * VECW;
    const bool cok = c0 < p.Cin;  // Cin % VECW == 0 (host-checked): a vector is entirely in or out of range
    float sc[VECW], sh[VECW];
    if (p.pre_scale) {
#pragma unroll
      for (int i = 0; i < VECW; ++i) {
        sc[i] = cok ? p.pre_scale[(long long)n * p.Cin + c0 + i] : 0.f;
        sh[i] = cok ? p.pre_shift[(long long)n * p.Cin + c0 + i] : 0.f;
      }
    }
#pragma unroll
    for (int jb = 0; jb < MAX_ITEMS; jb += A_BATCH) {
      uint4 raw[A_BATCH];
#pragma unroll
      for (int u = 0; u < A_BATCH; ++u) {
        raw[u] = make_uint4(0, 0, 0, 0);
        if (jb + u < MAX_ITEMS) {  // compile-time; the load itself is branch-free (clamped address + select, see load_b)
          const bool ok = (vox[jb + u] >= 0) & cok;
          const long long vv = ok ? vox[jb + u] : 0;
          const uint4 v = *reinterpret_cast<const uint4*>(xin + vv * p.x_ld + (cok ? c0 : 0));
          raw[u].x = ok ? v.x : 0u; raw[u].y = ok ? v.y : 0u; raw[u].z = ok ? v.z : 0u; raw[u].w = ok ? v.w : 0u;
        }
      }
#pragma unroll
      for (int u = 0; u < A_BATCH; ++u) {
        const int j = jb + u;
        if (j >= MAX_ITEMS) break;
        if (vox[j] == -2) continue;
        uint4 outv = raw[u];
        if (vox[j] >= 0 && cok && (p.pre_scale || p.pre_act)) {
          float v[VECW];
          Vec16<T>::unpack(raw[u], v);
          if (p.pre_scale) {
#pragma unroll
            for (int i = 0; i < VECW; ++i) v[i] = v[i] * sc[i] + sh[i];
          }
          if (p.pre_act) {
            conv_act_vec(v, p.pre_act, PRECISE);
          }
          outv = Vec16<T>::pack(v);
        }
        const int pv = (tid >> 2) + j * ROWS_PER_PASS;
        *reinterpret_cast<uint4*>(ldsA + (size_t)pv * FAST_ROWB + ((sq ^ fast_swz(pv)) << 4)) = outv;
      }
    }
  };

  // ---- main loop --------------------------------------------------------------------------------------------------------
  load_b(bregA, 0);
  stage_a(0);
  store_b(bregA, 0);
  if (total_gsteps > 1) load_b(bregB, 1);
  __syncthreads();
  int chunk = 0, grp = 0, kd_i = 0, kh_i = 0, kw_i = 0;
  // one tap group; `r_next` holds the panel of gstep+1 (stored to LDS after the MFMAs), `r_far` receives the panel of gstep+2
  auto group_body = [&](int gstep, uint4 (&r_next)[B_PER_THREAD], uint4 (&r_far)[B_PER_THREAD]) __attribute__((always_inline)) {
    if (grp == 0 && chunk > 0) {
      if (!(ABLATE(1))) stage_a(chunk);  // every wave passed the barrier that ended the previous chunk
      __syncthreads();
    }
    const char* bsrc = ldsB + (size_t)((gstep & 1) * G) * BN * FAST_ROWB;
#pragma unroll
    for (int u = 0; u < G; ++u) {
      if (grp * G + u < T_taps && !(ABLATE(8))) {
        const int tap_row = (kd_i * pH + kh_i) * pW + kw_i;
        uint4 xf[MF], wf[NFR];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const int rb = arow64[mf] + tap_row * FAST_ROWB;                 // row * 64
          const int sw = (((rb ^ (rb >> 1)) >> 2) & 0x30) ^ (q << 4);      // (q ^ fast_swz(row)) << 4, computed on the byte offset
          xf[mf] = *reinterpret_cast<const uint4*>(ldsA + rb + sw);
        }
#pragma unroll
        for (int nf = 0; nf < NFR; ++nf) wf[nf] = *reinterpret_cast<const uint4*>(bsrc + (size_t)u * BN * FAST_ROWB + boff[nf]);
        if (!(ABLATE(4))) {
#pragma unroll
          for (int nf = 0; nf < NFR; ++nf)
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) Mma<T>::run(wf[nf], xf[mf], acc[nf][mf]);
#ifndef GM_CONV_NO_SCHED
          // issue ALL operand reads of the tap before its first MFMA: under the 128-VGPR cap the scheduler otherwise funnels the
          // four weight fragments through one register quad (read -> wait -> 2 MFMAs, four times = four exposed LDS latencies)
          __builtin_amdgcn_sched_group_barrier(0x100, MF + NFR, 0);                               // DS reads
          __builtin_amdgcn_sched_group_barrier(0x008, MF * NFR * (sizeof(T) == 2 ? 1 : 4), 0);   // MFMAs
#endif
        } else {
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) asm volatile("" ::"v"(xf[mf].x), "v"(xf[mf].w), "v"(wf[mf].x), "v"(wf[mf].w));
        }
        if (++kw_i == p.kw) { kw_i = 0; if (++kh_i == p.kh) { kh_i = 0; ++kd_i; } }
      }
      // the weight-panel prefetch is issued AFTER the group's first tap: hipcc guards the first operand read of a group with a
      // conservative `s_waitcnt vmcnt(0)` (register re-use after the patch-staging loads); issued before it, the prefetch was
      // drained on the spot and its latency exposed
      if (u == 0 && gstep + 2 < total_gsteps && !(ABLATE(2))) load_b(r_far, gstep + 2);
    }
    if (gstep + 1 < total_gsteps) store_b(r_next, gstep + 1);
    __syncthreads();
    if (++grp == ngroups) { grp = 0; ++chunk; kd_i = kh_i = kw_i = 0; }
  };
  for (int gstep = 0; gstep < total_gsteps; gstep += 2) {
    group_body(gstep, bregB, bregA);
    if (gstep + 1 < total_gsteps) group_body(gstep + 1, bregA, bregB);
  }

  if (ABLATE(16)) return;
  if (conv_epilogue_lds_ok<T>(p) && !ABLATE(32)) {
    // every wave passed the main loop's final barrier: the operand buffers are free; one 64 x 144 B scratch per wave
    constexpr int EPASSES = (NFR * 16 * (int)sizeof(T) + 127) / 128;
    constexpr int CH_PER_PASS = 128 / (int)sizeof(T);
    float st_s[EPASSES][VECW], st_q[EPASSES][VECW];
#pragma unroll
    for (int e = 0; e < EPASSES; ++e)
#pragma unroll
      for (int i = 0; i < VECW; ++i) { st_s[e][i] = 0.f; st_q[e][i] = 0.f; }
    conv_epilogue_lds<T, MF, NFR>(p, acc, smem + (size_t)wave * MF * 16 * 144, n, wm * MF * 16, cb * BN + wn * NFR * 16, od0, oh0, ow0, lane,
                                  st_s, st_q);
    if (p.stats) {
      // fused GroupNorm statistics of the output: lanes sharing (lane & 7) hold the same channels for different rows
      float* sst = reinterpret_cast<float*>(smem);  // [NWAVES][64 channels][2], after everyone left the transpose scratch
      __syncthreads();
#pragma unroll
      for (int e = 0; e < EPASSES; ++e)
#pragma unroll
        for (int i = 0; i < VECW; ++i) {
          float a = st_s[e][i], b2 = st_q[e][i];
          a += __shfl_xor(a, 8, 64); b2 += __shfl_xor(b2, 8, 64);
          a += __shfl_xor(a, 16, 64); b2 += __shfl_xor(b2, 16, 64);
          a += __shfl_xor(a, 32, 64); b2 += __shfl_xor(b2, 32, 64);
          if (lane < 8) {
            const int ch = e * CH_PER_PASS + lane * VECW + i;  // within this wave's 64 channels
            sst[(wave * 64 + ch) * 2] = a;
            sst[(wave * 64 + ch) * 2 + 1] = b2;
          }
        }
      __syncthreads();
      if (tid < BN) {
        const int wn_c = tid >> 6, ch = tid & 63;
        double a = 0.0, b2 = 0.0;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          a += (double)sst[((wn_c * WM + w) * 64 + ch) * 2];
          b2 += (double)sst[((wn_c * WM + w) * 64 + ch) * 2 + 1];
        }
        const int co = cb * BN + tid;
        if (co < p.Cout) {
          const long long slot = tile_id % (unsigned)(ntd * nth * ntw);  // one plain store per (tile, channel): no atomics, fixed-order reduction later
          double* dst = p.stats + ((slot * p.N + n) * p.Cout + co) * 2;
          *reinterpret_cast<double2*>(dst) = make_double2(a, b2);
        }
      }
    }
  } else {
    conv_epilogue<T, MF, NFR>(p, acc, n, wm * MF * 16, cb * BN + wn * NFR * 16, od0, oh0, ow0, l15, q);
  }
}

extern "C" long long gm_conv_fast_lds_bytes(const GmConvDesc* d, int bn) {
  const long long td = 1 << d->ltd, th = 1 << d->lth, tw = 1 << d->ltw;
  const long long P = (td + d->kd - 1) * (th + d->kh - 1) * (tw + d->kw - 1);
  const long long need = P * FAST_ROWB + 2LL * 3 * bn * FAST_ROWB;
  const long long scratch = (1LL << (d->ltd + d->lth + d->ltw)) * (bn / 64) * 144;  // epilogue transpose scratch (64 x 144 B per 64x64 sub-tile) re-uses the operand buffers
  return need > scratch ? need : scratch;
}

// variant: 1 = 256 voxels x 64 ch (4 waves), 2 = 256 voxels x 128 ch (8 waves), 3 = 512 voxels x 64 ch (8 waves)
extern "C" long long gm_conv_fast_max_patch(int variant) {
  switch (variant) { case 1: return 12 * 64; case 2: return 6 * 128; case 3: return 10 * 128; case 4: return 6 * 128; case 5: return 3 * 256; case 6: return 5 * 256; default: return 0; }
}

template <typename T, int WM, int WN, int MF, int MINW>
static void launch_fast(const GmConvDesc& d, size_t smem, unsigned nblocks, hipStream_t st) {
  static bool attr_set = false;
  auto kern = conv_fast_kernel<T, WM, WN, MF, MINW>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) (void)hipGetLastError();
    attr_set = true;
  }
  kern<<<dim3(nblocks), 64 * WM * WN, smem, st>>>(d);
}

template <typename T>
static int dispatch_fast(const GmConvDesc& d, int variant, size_t smem, unsigned nblocks, hipStream_t st) {
  switch (variant) {
    case 1: launch_fast<T, 4, 1, 4, 2>(d, smem, nblocks, st); return 0;   // 256 vox x  64 ch,  4 waves, 2 WG/CU
    case 2: launch_fast<T, 4, 2, 4, 2>(d, smem, nblocks, st); return 0;   // 256 vox x 128 ch,  8 waves, 1 WG/CU
    case 3: launch_fast<T, 8, 1, 4, 2>(d, smem, nblocks, st); return 0;   // 512 vox x  64 ch,  8 waves, 1 WG/CU
    case 4: launch_fast<T, 8, 1, 2, 4>(d, smem, nblocks, st); return 0;   // 256 vox x  64 ch,  8 waves, 2 WG/CU (16 waves/CU)
    case 5: launch_fast<T, 8, 2, 2, 4>(d, smem, nblocks, st); return 0;   // 256 vox x 128 ch, 16 waves, 1 WG/CU
    case 6: launch_fast<T, 16, 1, 2, 4>(d, smem, nblocks, st); return 0;  // 512 vox x  64 ch, 16 waves, 1 WG/CU
    default: return -3;
  }
}

extern "C" int gm_conv_fast_variant_geometry(int variant, int* voxels, int* channels, int* threads) {
  static const int g[7][3] = {{0, 0, 0}, {256, 64, 256}, {256, 128, 512}, {512, 64, 512}, {256, 64, 512}, {256, 128, 1024}, {512, 64, 1024}};
  if (variant < 1 || variant > 6) return -1;
  *voxels = g[variant][0]; *channels = g[variant][1]; *threads = g[variant][2];
  return 0;
}

extern "C" int gm_conv_fast_launch(const GmConvDesc* dp, int variant, unsigned nblocks, void* stream) {
  const GmConvDesc& d = *dp;
  int vox = 0, ch = 0, thr = 0;
  if (gm_conv_fast_variant_geometry(variant, &vox, &ch, &thr) != 0) return -3;
  const size_t smem = (size_t)gm_conv_fast_lds_bytes(dp, ch);
  hipStream_t st = (hipStream_t)stream;
  if (d.dtype == GM_F32) return dispatch_fast<float>(d, variant, smem, nblocks, st);
  if (d.dtype == GM_BF16) return dispatch_fast<bf16_raw>(d, variant, smem, nblocks, st);
  return -2;
}
