// Host replay of conv_mw.hip's data movement (tests/test_host_logic.py compiles and runs it with g++; no GPU): every LDS-DMA piece, every
// ds_read_b128 operand fragment and the v_mfma_f32_32x32x16_bf16 lane maps are evaluated through the SAME index functions the kernel calls
// (generativemodels_amd/csrc/conv_mw_index.h), for whole tiles of a ragged problem, and the result is compared with a direct convolution.
// Also counts LDS bank conflicts per ds_read_b128 lane group (MI355X_MICROARCH.md: four groups of 16 lanes, 64 banks x 4 B).
// Test infrastructure: checks index arithmetic only (values are small integers, exact in every format).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../generativemodels_amd/csrc/conv_mw_index.h"

using namespace mw;
typedef short el;  // stands for a bf16 element (2 bytes)

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails < 20) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } ++fails; } } while (0)

// ds_read_b128 lane groups (one LDS cycle each when conflict-free)
static const int GROUPS[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                  {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                  {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                  {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
static long long conflicts = 0, reads = 0;
static void check_banks(const int (&addr)[64]) {
  for (int g = 0; g < 4; ++g) {
    int seen[16] = {0};
    for (int i = 0; i < 16; ++i) {
      const int a = addr[GROUPS[g][i]];
      CHECK(a % 16 == 0, "unaligned ds_read_b128 address %d", a);
      const int slot = (a / 16) % 16;  // 16-byte slot within the 256-byte bank row
      if (seen[slot]++) ++conflicts;
    }
  }
  ++reads;
}

struct Problem {
  int N, D, H, W, Cin, Cout, cin_split;  // cin_split > 0: input = cat(x[..., :cin_split], x2)
  int skip0, skip1;                      // fused 1x1 shortcut sources (channels), 0 = none
};

static int run(const Problem& P) {
  const int cout_pad = (P.Cout + 15) & ~15;
  const int c0 = P.cin_split > 0 ? P.cin_split : P.Cin, c1 = P.Cin - c0;
  const long long V = (long long)P.N * P.D * P.H * P.W;
  std::vector<el> x((size_t)V * c0), x2((size_t)V * (c1 > 0 ? c1 : 1)), s0((size_t)V * (P.skip0 ? P.skip0 : 1)), s1((size_t)V * (P.skip1 ? P.skip1 : 1));
  srand(1234);
  auto rnd = []() { return (el)(rand() % 7 - 3); };
  for (auto& v : x) v = rnd();
  for (auto& v : x2) v = rnd();
  for (auto& v : s0) v = rnd();
  for (auto& v : s1) v = rnd();
  std::vector<el> w((size_t)P.Cout * P.Cin * 27), wsk((size_t)P.Cout * (P.skip0 + P.skip1 ? P.skip0 + P.skip1 : 1));
  for (auto& v : w) v = rnd();
  for (auto& v : wsk) v = rnd();
  // gm_pack_conv_weight: [chunk32][tap][cout_pad][32], zero padded
  const int nc32 = (P.Cin + 31) / 32;
  std::vector<el> wp((size_t)nc32 * 27 * cout_pad * 32, 0);
  for (int co = 0; co < P.Cout; ++co)
    for (int ci = 0; ci < P.Cin; ++ci)
      for (int t = 0; t < 27; ++t) wp[(((size_t)(ci / 32) * 27 + t) * cout_pad + co) * 32 + ci % 32] = w[((size_t)co * P.Cin + ci) * 27 + t];
  const int sk = P.skip0 + P.skip1, nsk32 = (sk + 31) / 32;
  std::vector<el> wskp((size_t)(nsk32 ? nsk32 : 1) * cout_pad * 32, 0);
  for (int co = 0; co < P.Cout; ++co)
    for (int ci = 0; ci < sk; ++ci) wskp[((size_t)(ci / 32) * cout_pad + co) * 32 + ci % 32] = wsk[(size_t)co * sk + ci];
  const el zero_row[32] = {0};

  // direct convolution (pad 1) + shortcut
  std::vector<long long> want((size_t)V * P.Cout, 0);
  auto in_at = [&](int n, int d, int h, int ww, int ci) -> long long {
    if (d < 0 || d >= P.D || h < 0 || h >= P.H || ww < 0 || ww >= P.W) return 0;
    const long long vox = (((long long)n * P.D + d) * P.H + h) * P.W + ww;
    return ci < c0 ? x[vox * c0 + ci] : x2[vox * c1 + (ci - c0)];
  };
  for (int n = 0; n < P.N; ++n)
    for (int d = 0; d < P.D; ++d)
      for (int h = 0; h < P.H; ++h)
        for (int ww = 0; ww < P.W; ++ww) {
          const long long vox = (((long long)n * P.D + d) * P.H + h) * P.W + ww;
          for (int co = 0; co < P.Cout; ++co) {
            long long a = 0;
            for (int t = 0; t < 27; ++t)
              for (int ci = 0; ci < P.Cin; ++ci) a += (long long)w[((size_t)co * P.Cin + ci) * 27 + t] * in_at(n, d + t / 9 - 1, h + (t / 3) % 3 - 1, ww + t % 3 - 1, ci);
            for (int ci = 0; ci < sk; ++ci) a += (long long)wsk[(size_t)co * sk + ci] * (ci < P.skip0 ? s0[vox * P.skip0 + ci] : s1[vox * P.skip1 + (ci - P.skip0)]);
            want[vox * P.Cout + co] = a;
          }
        }

  std::vector<long long> got((size_t)V * P.Cout, -777777);
  const int ntd = (P.D + TD - 1) / TD, nth = (P.H + TH - 1) / TH, ntw = (P.W + TW - 1) / TW, ncb = (P.Cout + BN - 1) / BN;
  const int nchunks = P.Cin / BKH, nchunks0 = c0 / BKH;
  std::vector<char> lds(LDS_BYTES);
  // one LDS-DMA instruction: lane i copies 16 bytes from src[i] to lds[dst + 16 i]
  auto dma = [&](const char* const (&src)[64], int dst, int nlanes) {
    for (int l = 0; l < nlanes; ++l) {
      CHECK(dst + 16 * l + 16 <= ADDV_OFF, "DMA beyond the operand buffers: %d", dst + 16 * l);
      memcpy(&lds[dst + 16 * l], src[l], 16);
    }
  };
  for (int n = 0; n < P.N; ++n)
    for (int td_i = 0; td_i < ntd; ++td_i)
      for (int th_i = 0; th_i < nth; ++th_i)
        for (int tw_i = 0; tw_i < ntw; ++tw_i)
          for (int cb = 0; cb < ncb; ++cb) {
            const int od0 = td_i * TD, oh0 = th_i * TH, ow0 = tw_i * TW;
            static long long acc[NW][2][2][64][16];
            memset(acc, 0, sizeof(acc));
            auto mfma = [&](int wave, int nb, int mb, const el (&a)[64][8], const el (&b)[64][8]) {
              // D[i][j] += sum_k A[i][k] B[k][j]; lane l supplies A[l & 31][8 (l >> 5) + e] and B[8 (l >> 5) + e][l & 31]
              for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 16; ++r) {
                  const int i = acc_channel(l, r), j = acc_voxel(l);
                  long long s = 0;
                  for (int kh = 0; kh < 2; ++kh)
                    for (int e = 0; e < 8; ++e) s += (long long)a[i + 32 * kh][e] * b[j + 32 * kh][e];
                  acc[wave][nb][mb][l][r] += s;
                }
            };
            auto frag = [&](int wave_unused, const int (&addr)[64], el (&out)[64][8]) {
              (void)wave_unused;
              check_banks(addr);
              for (int l = 0; l < 64; ++l) {
                CHECK(addr[l] >= 0 && addr[l] + 16 <= ADDV_OFF, "fragment read out of range: %d", addr[l]);
                memcpy(out[l], &lds[addr[l]], 16);
              }
            };
            for (int chunk = 0; chunk < nchunks; ++chunk) {
              memset(lds.data(), 0x55, lds.size());  // stale garbage: every byte read must have been written for this half-chunk
              // ---- patch ----
              const bool second = chunk >= nchunks0;
              for (int wave = 0; wave < NW; ++wave)
                for (int j = 0; j < PD; ++j) {
                  const char* src[64];
                  for (int lane = 0; lane < 64; ++lane) {
                    const PatchLane pl = patch_lane(wave, lane);
                    const int ud = od0 - 1 + j, uh = oh0 - 1 + pl.line, uw = ow0 - 1 + pl.col;
                    const bool ok = pl.valid && uh >= 0 && uh < P.H && uw >= 0 && uw < P.W && ud >= 0 && ud < P.D;
                    const long long pv = (((long long)n * P.D + ud) * P.H + uh) * P.W + uw;
                    const char* base = second ? (const char*)x2.data() + (long long)(chunk - nchunks0) * ROWB : (const char*)x.data() + (long long)chunk * ROWB;
                    const long long rowb = (second ? c1 : c0) * 2;
                    src[lane] = ok ? base + pv * rowb + (pl.slot << 4) : (const char*)zero_row + ((lane & 1) << 4);
                  }
                  dma(src, patch_piece_dst(j, wave), 64);
                }
              for (int g = 0; g < NGROUPS; ++g) {
                const int slot = g % RING;
                // ---- weight panel (half-chunk, g) -> ring slot ----
                for (int wave = 0; wave < NW; ++wave)
                  for (int hp = 0; hp < 2; ++hp) {
                    const char* src[64];
                    for (int lane = 0; lane < 64; ++lane) {
                      const WLane wl = wpanel_lane(wave, lane, hp);
                      const int co = cb * BN + wl.co;
                      const char* panel = (const char*)wp.data() + ((long long)((chunk >> 1) * 27 + G * g) * cout_pad) * SRC_ROWB + (chunk & 1) * ROWB;
                      src[lane] = co < cout_pad ? panel + wsrc_offset(wl.tap, co, cout_pad, 0, wl.slot) : (const char*)zero_row + ((lane & 1) << 4);
                    }
                    dma(src, wpanel_piece_dst(slot, wave, hp), hp ? 32 : 64);
                  }
                for (int u = 0; u < G; ++u) {
                  const int tap = g * G + u, kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
                  for (int wave = 0; wave < NW; ++wave) {
                    el af[2][64][8], bf[2][64][8];
                    for (int nb = 0; nb < 2; ++nb) {
                      int addr[64];
                      for (int l = 0; l < 64; ++l) addr[l] = a_lane_base(l) + a_offset(slot, u, nb);
                      frag(wave, addr, af[nb]);
                    }
                    for (int mb = 0; mb < 2; ++mb) {
                      int addr[64];
                      for (int l = 0; l < 64; ++l) addr[l] = b_lane_base(wave, l, kw) + b_offset(mb, kd, kh);
                      frag(wave, addr, bf[mb]);
                    }
                    for (int nb = 0; nb < 2; ++nb)
                      for (int mb = 0; mb < 2; ++mb) mfma(wave, nb, mb, af[nb], bf[mb]);
                  }
                }
              }
            }
            // ---- fused 1x1 shortcut ----
            if (sk) {
              const int nsc0 = P.skip0 / BKH, nsc = nsc0 + P.skip1 / BKH;
              for (int sc0 = 0; sc0 < nsc; sc0 += SC_ROUND) {
                memset(lds.data(), 0x55, lds.size());
                for (int j = 0; j < SC_ROUND; ++j) {
                  const int sc = sc0 + j;
                  if (sc >= nsc) continue;
                  const int part = sc >= nsc0 ? 1 : 0, cip = sc - (part ? nsc0 : 0);
                  for (int wave = 0; wave < NW; ++wave) {
                    for (int h = 0; h < 2; ++h) {
                      const char* src[64];
                      for (int lane = 0; lane < 64; ++lane) {
                        const int m = 32 * h + (lane >> 1);
                        const int od = od0 + wave, oh = oh0 + (m >> 4), ow = ow0 + (m & 15);
                        const bool ok = od < P.D && oh < P.H && ow < P.W;
                        const long long vox = (((long long)n * P.D + od) * P.H + oh) * P.W + ow;
                        const char* xb = (part ? (const char*)s1.data() : (const char*)s0.data()) + (long long)cip * ROWB + (sc_x_lane_slot(lane) << 4);
                        src[lane] = ok ? xb + vox * (part ? P.skip1 : P.skip0) * 2 : (const char*)zero_row + ((lane & 1) << 4);
                      }
                      dma(src, sc_x_piece_dst(j, wave, h), 64);
                    }
                    if (wave == j)
                      for (int h = 0; h < 2; ++h) {
                        const char* src[64];
                        for (int lane = 0; lane < 64; ++lane) {
                          const int wrow = lane >> 1, wslot = ((lane & 1) ^ row_swz(wrow)) << 4, wco = cb * BN + 32 * h + wrow;
                          const char* wpan = (const char*)wskp.data() + (long long)(sc >> 1) * cout_pad * SRC_ROWB + (sc & 1) * ROWB + wslot;
                          src[lane] = wco < cout_pad ? wpan + (long long)wco * SRC_ROWB : (const char*)zero_row + ((lane & 1) << 4);
                        }
                        dma(src, sc_w_piece_dst(j, h), 64);
                      }
                  }
                }
                for (int j = 0; j < SC_ROUND; ++j) {
                  if (sc0 + j >= nsc) continue;
                  for (int wave = 0; wave < NW; ++wave) {
                    el af[2][64][8], bf[2][64][8];
                    for (int mb = 0; mb < 2; ++mb) {
                      int addr[64];
                      for (int l = 0; l < 64; ++l) addr[l] = sc_b_lane_base(wave, l) + sc_b_offset(j, mb);
                      frag(wave, addr, bf[mb]);
                    }
                    for (int nb = 0; nb < 2; ++nb) {
                      int addr[64];
                      for (int l = 0; l < 64; ++l) addr[l] = sc_a_lane_base(l) + sc_a_offset(j, nb);
                      frag(wave, addr, af[nb]);
                    }
                    for (int nb = 0; nb < 2; ++nb)
                      for (int mb = 0; mb < 2; ++mb) mfma(wave, nb, mb, af[nb], bf[mb]);
                  }
                }
              }
            }
            // ---- epilogue: accumulators -> transpose scratch (row = voxel, channel offset) -> rows it * 8 + lane / 8, segment lane % 8 ----
            for (int wave = 0; wave < NW; ++wave) {
              static long long scratch[64][64];
              for (int a = 0; a < 64; ++a)
                for (int c = 0; c < 64; ++c) scratch[a][c] = -999999;
              for (int lane = 0; lane < 64; ++lane)
                for (int nb = 0; nb < 2; ++nb)
                  for (int jq = 0; jq < 4; ++jq)
                    for (int mb = 0; mb < 2; ++mb) {
                      const int ch = nb * 32 + 8 * jq + 4 * (lane >> 5);
                      CHECK(ch == nb * 32 + acc_channel(lane, 4 * jq), "channel map");
                      for (int i = 0; i < 4; ++i) scratch[acc_row(mb, lane)][ch + i] = acc[wave][nb][mb][lane][4 * jq + i];
                    }
              for (int it = 0; it < 8; ++it)
                for (int lane = 0; lane < 64; ++lane) {
                  const int row = it * 8 + (lane >> 3), seg = lane & 7;
                  const int line = wave * 4 + (it >> 1);  // dma_epilogue_place with line0 = wave * 4
                  const int od = od0 + (line >> 2), oh = oh0 + (line & 3), ow = ow0 + (it & 1) * 8 + (lane >> 3);
                  const int co = cb * BN + seg * 8;
                  if (co < P.Cout && od < P.D && oh < P.H && ow < P.W) {
                    const long long vox = (((long long)n * P.D + od) * P.H + oh) * P.W + ow;
                    for (int i = 0; i < 8; ++i) got[vox * P.Cout + co + i] = scratch[row][seg * 8 + i];
                  }
                }
            }
          }
  long long bad = 0;
  for (size_t i = 0; i < want.size(); ++i)
    if (want[i] != got[i]) ++bad;
  CHECK(bad == 0, "%lld of %zu outputs differ (N %d D %d H %d W %d Cin %d Cout %d split %d skip %d+%d)", bad, want.size(), P.N, P.D, P.H, P.W, P.Cin, P.Cout,
        P.cin_split, P.skip0, P.skip1);
  return bad == 0;
}

int main() {
  static_assert(LDS_BYTES * 3 <= 160 * 1024, "three work-groups per CU");
  const Problem probs[] = {
      {1, 4, 4, 16, 16, 64, 0, 0, 0},      // one tile, one half-chunk
      {1, 5, 6, 20, 32, 80, 0, 0, 0},      // ragged extents, ragged channel block
      {2, 3, 9, 33, 48, 64, 16, 0, 0},     // two samples, second source from half-chunk 1 on
      {1, 6, 4, 18, 32, 72, 0, 48, 32},    // fused shortcut over two sources (5 half-chunks: two rounds)
      {1, 4, 5, 16, 64, 64, 32, 16, 0},    // both
  };
  int ok = 1;
  for (const Problem& p : probs) ok &= run(p);
  printf("ds_read_b128 fragments replayed: %lld, lane-group bank conflicts: %lld\n", reads, conflicts);
  CHECK(conflicts == 0, "bank conflicts");
  if (fails || !ok) { printf("FAILED (%d)\n", fails); return 1; }
  printf("conv_mw index replay OK\n");
  return 0;
}
