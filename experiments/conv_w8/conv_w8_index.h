// Index arithmetic of the 512-voxel LDS-DMA convolution with v_mfma_f32_32x32x16_bf16 (conv_w8.hip, tile configuration 22).  Plain functions of
// (wave, lane, tap, ...) with no HIP dependency: the kernel calls them on the device and tests/emulate_conv_w8.cpp replays them on the host --
// LDS-DMA pieces, ds_read_b128 fragments and the MFMA lane maps -- against a direct convolution, and counts LDS bank conflicts per
// ds_read_b128 lane group, before any GPU time is spent.
//
// Why this shape (profiles/r04_conv_dma_ablation.txt, r04_lds_dma_patterns.txt): with every LDS-DMA request removed the
// 256-voxel tiles run at 1 850 - 2 050 TFLOP/s and with them at 1 030 - 1 290 -- operand movement L2 -> LDS, not the matrix pipe, is what a
// tile waits for, and two thirds of those bytes are weight panels, re-fetched by every 256 voxels.  One CU moves at most ~64 B/clk or ~one
// request per clock, and a request of 32 bytes costs as much as one of 64.  So:
//   * 512 output voxels (8 x 4 x 16) x 64 output channels per work-group, 8 waves: a weight panel serves twice the voxels and the halo
//     shrinks from 2.53 to 2.11 rows per output voxel -- 0.58x the bytes per FLOP of the 256-voxel tiles;
//   * the halo patch keeps 64-byte rows (32 input channels: the efficient request size) while the WEIGHTS advance in halves of 16 input
//     channels -- v_mfma_f32_32x32x16_bf16 consumes K = 16 -- so a 3-tap panel is 6 KiB and a two-slot ring 12 KiB:
//     patch 67.5 KiB + ring 12 KiB + addend 256 B = 79.75 KiB, TWO work-groups per CU = 4 waves per SIMD at <= 128 registers;
//   * the 16-channel panels come from their own packed image [chunk32][half][tap][Cout_pad][16] (ops.packed_conv_weight_halves), so that
//     a panel is three contiguous 2 KiB runs.
// Geometry.  Wave w owns depth plane w of the tile (64 voxels = 4 W-lines of 16) x 64 channels = 2 x 2 blocks of the 32x32x16 MFMA
// (A = weights: 32 output channels x 16 k, B = activations: 16 k x 32 voxels; voxel block mb = lines 2 mb, 2 mb + 1).
#pragma once

#if defined(__HIPCC__)
#define W8_HD __host__ __device__ __forceinline__
#else
#define W8_HD inline
#endif

namespace w8 {
constexpr int NW = 8;              // waves per work-group
constexpr int TD = 8, TH = 4, TW = 16;
constexpr int BM = TD * TH * TW;   // 512 output voxels per work-group
constexpr int BN = 64;             // output channels per work-group
constexpr int BK = 32;             // input channels per patch chunk (64-byte rows)
constexpr int BKH = 16;            // input channels per weight panel / MFMA K step
constexpr int ROWB = 64;           // bytes per patch row
constexpr int PIECE_ROWS = 16;     // one LDS-DMA instruction: 64 lanes x 16 B = 16 patch rows
constexpr int PD = TD + 2, PH = TH + 2, PW = TW + 2;
constexpr int LINE = PW;           // 18: no padding anywhere -- the patch is exactly 1080 rows
constexpr int PLANE = PH * LINE;   // 108
constexpr int PROWS = PD * PLANE;  // 1080
constexpr int PATCH_BYTES = PROWS * ROWB;            // 69120
constexpr int NPIECES = (PROWS + PIECE_ROWS - 1) / PIECE_ROWS;  // 68: the last one is half a piece (lanes 0..31)
constexpr int G = 3;                           // taps per weight panel
constexpr int NGROUPS = 9;                     // panels per 16-channel half
constexpr int RING = 2;                        // panels in the LDS ring
constexpr int WROWB = 32;                      // bytes per weight-panel row (16 bf16)
constexpr int WROWS = G * BN;                  // 192 rows per panel = 6 pieces of 32 rows
constexpr int WBUF_BYTES = WROWS * WROWB;      // 6 KiB
constexpr int RING_OFF = PATCH_BYTES;
constexpr int RING_BYTES = RING * WBUF_BYTES;  // 12 KiB
constexpr int ADDV_OFF = RING_OFF + RING_BYTES;      // 81408: the per-channel epilogue addend (64 floats)
constexpr int LDS_BYTES = ADDV_OFF + BN * 4;         // 81664 -> two work-groups per CU (2 x 79.75 KiB of 160)
// fused 1x1 shortcut: rounds of up to SC_ROUND 32-channel chunks; chunk j of a round keeps its 512 voxel rows (64 bytes) at j * SC_XBYTES
// and its 64-row weight panel (64-byte rows of the standard packed image) at SC_WOFF + j * SC_WBYTES (patch and ring are dead by then)
constexpr int SC_ROUND = 2;
constexpr int SC_XBYTES = BM * ROWB;           // 32 KiB
constexpr int SC_WOFF = SC_ROUND * SC_XBYTES;  // 64 KiB
constexpr int SC_WBYTES = BN * ROWB;           // 4 KiB
static_assert(SC_WOFF + SC_ROUND * SC_WBYTES <= ADDV_OFF, "the shortcut's operands fit under the addend vector");
constexpr int SCRATCH_WAVE = 64 * 144;         // epilogue transpose scratch per wave (64 voxel rows x (128 B + 16 B pad))
static_assert(NW * SCRATCH_WAVE <= ADDV_OFF, "the transpose scratch fits under the addend vector");

// ---- bank swizzles: the 16-byte slot a channel quarter lands in is (quarter) ^ key ------------------------------------------------------
// A ds_read_b128 is served in four groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32); with 64-byte rows four
// consecutive rows span the 64 banks, so the four lanes of a group whose rows agree modulo 4 must read four different slots.
// Patch rows of one 32-voxel fragment: lanes 0..15 read columns c + kw of line l, lanes 16..31 of line l + 1 = 18 rows further (= 2 mod 4).
// The key is a function of the patch COLUMN alone (so a tap's address = lane base for (kw, half) + immediate); this table is a 4-colouring of
// the conflict graph over all taps and both lane groups (found by search, verified by the host replay): key(col) = T[col >> 1].
constexpr unsigned PATCH_KEYS = 3u | (1u << 2) | (0u << 4) | (1u << 6) | (2u << 8) | (2u << 10) | (0u << 12) | (3u << 14) | (3u << 16);
W8_HD int patch_key(int col) { return (int)((PATCH_KEYS >> (col & ~1)) & 3u); }
// 32 consecutive 64-byte rows (shortcut operands): the rows of a lane group that agree modulo 4 are {r, r + 12, r + 20, r + 24} or {r + 4, r + 8, r + 16, r + 28}
W8_HD int row64_key(int row) { return (row >> 2) & 3; }
// 32 consecutive 32-byte rows (weight panels): 8 rows span the banks; rows of a lane group that agree modulo 8 differ in bit 3
W8_HD int row32_key(int row) { return (row >> 3) & 1; }

// ---- LDS-DMA pieces (lane i writes LDS bytes [dst + 16 i, +16)) ------------------------------------------------------------------------
// patch: piece p covers rows 16 p .. 16 p + 15; lane -> row 16 p + (lane >> 2), slot lane & 3.  Wave w issues pieces w, w + 8, ..., so a
// lane's row advances by 128 = one plane + one line + two columns per piece.
struct PatchRow { int pd, ph, pw; };
W8_HD PatchRow patch_row(int row) {
  PatchRow r;
  r.pd = row / PLANE;
  const int q = row - r.pd * PLANE;
  r.ph = q / LINE;
  r.pw = q - r.ph * LINE;
  return r;
}
W8_HD PatchRow patch_row_next(PatchRow r) {  // the row 128 further
  r.pw += 2; r.ph += 1; r.pd += 1;
  if (r.pw >= LINE) { r.pw -= LINE; r.ph += 1; }
  if (r.ph >= PH) { r.ph -= PH; r.pd += 1; }
  return r;
}
W8_HD int patch_piece_dst(int piece) { return piece * PIECE_ROWS * ROWB; }
W8_HD int patch_lane_quarter(int lane, int pw) { return (lane & 3) ^ patch_key(pw); }  // the channel quarter (16 bytes of the chunk's 64) this lane fetches
W8_HD int pieces_of_wave(int wave) { return (NPIECES - wave + NW - 1) / NW; }          // 9 for waves 0..3, 8 for waves 4..7

// weight panel (3 taps x 64 output channels x 16 input channels): 6 pieces of 32 rows; wave w < 6 moves piece w = tap w >> 1, channels 32 (w & 1) ..
struct WLane { int tap, co, slot; };
W8_HD WLane wpanel_lane(int wave, int lane) {
  WLane r;
  r.tap = wave >> 1;
  r.co = 32 * (wave & 1) + (lane >> 1);
  r.slot = (lane & 1) ^ row32_key(lane >> 1);
  return r;
}
W8_HD int wpanel_piece_dst(int ring_slot, int wave) { return RING_OFF + ring_slot * WBUF_BYTES + wave * 1024; }
// byte offset of (chunk32 c, half h, tap t, output channel co, 16-byte slot s) in the halves image [chunk32][half][tap][cout_pad][16]
W8_HD long long whalves_offset(int c, int h, int t, int co, int cout_pad, int slot) {
  return ((((long long)c * 2 + h) * 27 + t) * cout_pad + co) * WROWB + slot * 16;
}

// ---- operand fragments (ds_read_b128): lane supplies 8 k-values of row (lane & 31), k-half lane >> 5 ------------------------------------
// A: weight rows of output-channel block nb (32 channels) of tap u in ring slot s
W8_HD int a_lane_base(int lane) { return RING_OFF + (lane & 31) * WROWB + (((lane >> 5) ^ row32_key(lane & 31)) << 4); }
W8_HD int a_offset(int ring_slot, int u, int nb) { return ring_slot * WBUF_BYTES + u * (BN * WROWB) + nb * (32 * WROWB); }
// B: patch rows of voxel block mb (lines 2 mb, 2 mb + 1 of plane `wave`) at tap (kd, kh, kw), channel half h
W8_HD int b_lane_base(int wave, int lane, int kw, int half) {
  const int col = (lane & 15) + kw;
  return ((wave * PH + ((lane >> 4) & 1)) * LINE + col) * ROWB + (((2 * half + (lane >> 5)) ^ patch_key(col)) << 4);
}
W8_HD int b_offset(int mb, int kd, int kh) { return (kd * PLANE + (2 * mb + kh) * LINE) * ROWB; }

// ---- accumulator layout of v_mfma_f32_32x32x16_bf16 (C/D): lane holds voxel column (lane & 31) and 16 output channels -------------------
W8_HD int acc_channel(int lane, int reg) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }  // within the 32-channel block
W8_HD int acc_voxel(int lane) { return lane & 31; }                                                  // within the 32-voxel block
W8_HD int acc_row(int mb, int lane) { return mb * 32 + (lane & 31); }                                // row of the epilogue's transpose scratch

// ---- fused 1x1 shortcut (32-channel chunks, 64-byte rows) ---------------------------------------------------------------------------------
// voxel rows of chunk j of a round: wave w DMAs its own 64 voxels (4 pieces: rows 64 w + 16 h + (lane >> 2)); the B fragment of block mb
// reads rows 64 w + 32 mb + (lane & 31)
W8_HD int sc_x_piece_dst(int j, int wave, int h) { return j * SC_XBYTES + (64 * wave + 16 * h) * ROWB; }
W8_HD int sc_x_lane_quarter(int lane, int h) { return (lane & 3) ^ row64_key((16 * h + (lane >> 2)) & 31); }
W8_HD int sc_b_lane_base(int wave, int lane, int half) {
  return (64 * wave + (lane & 31)) * ROWB + (((2 * half + (lane >> 5)) ^ row64_key(lane & 31)) << 4);
}
W8_HD int sc_b_offset(int j, int mb) { return j * SC_XBYTES + mb * (32 * ROWB); }
// weight panel of chunk j: 64 rows of 64 bytes = 4 pieces, moved by waves 4 j .. 4 j + 3 (piece h: rows 16 h ..); the A fragment of block nb
// reads rows 32 nb + (lane & 31)
W8_HD int sc_w_piece_dst(int j, int h) { return SC_WOFF + j * SC_WBYTES + h * (16 * ROWB); }
W8_HD int sc_w_lane_quarter(int lane, int h) { return (lane & 3) ^ row64_key((16 * h + (lane >> 2)) & 31); }
W8_HD int sc_a_lane_base(int lane, int half) { return SC_WOFF + (lane & 31) * ROWB + (((2 * half + (lane >> 5)) ^ row64_key(lane & 31)) << 4); }
W8_HD int sc_a_offset(int j, int nb) { return j * SC_WBYTES + nb * (32 * ROWB); }
}  // namespace w8
