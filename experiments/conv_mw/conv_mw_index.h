// Index arithmetic of the three-work-groups-per-CU LDS-DMA convolution (conv_mw.hip, tile configuration 21).  Plain functions of
// (wave, lane, tap, ...) with no HIP dependency: the kernel calls them on the device and tests/emulate_conv_mw.cpp replays them on the host
// -- LDS-DMA pieces, ds_read_b128 fragments and the 32x32x16 MFMA lane maps -- against a direct convolution, and counts LDS bank conflicts
// per ds_read_b128 lane group, before any GPU time is spent.
//
// Geometry (bf16 only).  Tile 4 x 4 x 16 output voxels x 64 output channels, 4 waves; wave w owns depth plane w of the tile (64 voxels =
// 4 W-lines of 16) x 64 channels = 2 x 2 blocks of v_mfma_f32_32x32x16_bf16 (A = weights: 32 output channels x 16 k, B = activations:
// 16 k x 32 voxels; voxel block mb = lines 2 mb, 2 mb + 1).  K advances in HALF-chunks of 16 input channels: LDS rows are 32 bytes, so the
// halo patch (6 planes x 6 lines x 18 columns) is 24 KiB instead of 42 and a 3-tap weight panel 6 KiB instead of 12 -- 42.25 KiB per
// work-group, three work-groups per CU (conv_dma.hip's 78 KiB tiles: two).
#pragma once

#if defined(__HIPCC__)
#define MW_HD __host__ __device__ __forceinline__
#else
#define MW_HD inline
#endif

namespace mw {
constexpr int NW = 4;              // waves per work-group
constexpr int TD = 4, TH = 4, TW = 16;
constexpr int BM = TD * TH * TW;   // 256 output voxels per work-group
constexpr int BN = 64;             // output channels per work-group
constexpr int BKH = 16;            // input channels per half-chunk
constexpr int ROWB = 32;           // bytes per LDS row (16 bf16)
constexpr int PIECE_ROWS = 32;     // one LDS-DMA instruction: 64 lanes x 16 B = 32 rows
constexpr int PD = TD + 2, PH = TH + 2, PW = TW + 2;
constexpr int LINE = 20;           // patch line pitch in rows (18 columns + 2 pad): = 4 mod 8, see patch_swz()
constexpr int PLANE = 128;         // patch plane pitch in rows (6 lines x 20 = 120 -> 4 DMA pieces): plane j = pieces 4j .. 4j+3, one per wave
constexpr int PROWS = PD * PLANE;  // 768
constexpr int PATCH_BYTES = PROWS * ROWB;      // 24 KiB
constexpr int G = 3;                           // taps per weight panel
constexpr int NGROUPS = 9;                     // panels per half-chunk
constexpr int RING = 3;                        // panels in the LDS ring
constexpr int WROWS = G * BN;                  // 192 rows per panel: 4 full pieces (taps 0, 1) + 4 half pieces (tap 2)
constexpr int WBUF_BYTES = WROWS * ROWB;       // 6 KiB
constexpr int RING_BYTES = RING * WBUF_BYTES;  // 18 KiB
constexpr int ADDV_OFF = PATCH_BYTES + RING_BYTES;   // 43008: the per-channel epilogue addend (64 floats)
constexpr int LDS_BYTES = ADDV_OFF + BN * 4;         // 43264 -> 3 work-groups per CU
constexpr int SRC_ROWB = 64;                   // bytes per (chunk32, tap, co) row of the packed weight image (gm_pack_conv_weight, BK = 32)
// fused 1x1 shortcut: rounds of up to SC_ROUND half-chunks; half-chunk j of a round keeps its 256 voxel rows at j * SC_XBYTES and its 64-row
// weight panel at SC_WOFF + j * SC_WBYTES (the patch and the ring are dead by then)
constexpr int SC_ROUND = 4;
constexpr int SC_XBYTES = BM * ROWB;           // 8 KiB
constexpr int SC_WOFF = SC_ROUND * SC_XBYTES;  // 32 KiB
constexpr int SC_WBYTES = BN * ROWB;           // 2 KiB
static_assert(SC_WOFF + SC_ROUND * SC_WBYTES <= ADDV_OFF, "the shortcut's operands fit under the addend vector");
constexpr int SCRATCH_WAVE = 64 * 144;         // epilogue transpose scratch per wave (64 voxel rows x (128 B + 16 B pad))
static_assert(NW * SCRATCH_WAVE <= ADDV_OFF, "the transpose scratch fits under the addend vector");

// ---- bank swizzles: the 16-byte slot a k-half lands in is (k-half) ^ swz --------------------------------------------------------------
// A ds_read_b128 is served in four groups of 16 lanes; with 32-byte rows 8 consecutive rows span the 64 banks, so the 16 lanes of a group
// must split into two sets of 8 rows that differ in the slot bit.
// Patch rows of one 32-voxel fragment: lanes 0..15 read columns c + kw of line l, lanes 16..31 of line l + 1 = 20 rows further (= 4 mod 8).
// A lane group holds columns {0-3, 12-15} of one line and {4-11} of the other: rows that share (row mod 8) are the pairs (lc, lc +- 4),
// so keying the bit on bit 2 of the patch COLUMN separates them -- and keeps a tap's address = (lane base for kw) + immediate.
MW_HD int patch_swz(int lc) { return (lc >> 2) & 1; }
// Weight-panel / shortcut rows of one fragment are 32 consecutive rows: a lane group holds rows {0-3, 12-15, 20-27} or {4-11, 16-19, 28-31};
// rows that share (row mod 8) differ in bit 3.
MW_HD int row_swz(int row) { return (row >> 3) & 1; }

// ---- LDS-DMA pieces (lane i writes LDS bytes [dst + 16 i, +16)) ------------------------------------------------------------------------
// patch: piece (plane j, wave w) covers rows j * 128 + 32 w .. + 31; lane -> row 32 w + (lane >> 1) of the plane, slot lane & 1
struct PatchLane { int line, col, slot; bool valid; };   // in-plane placement of this lane's row: tile-independent
MW_HD PatchLane patch_lane(int wave, int lane) {
  const int rr = PIECE_ROWS * wave + (lane >> 1);
  PatchLane p;
  p.line = rr / LINE;
  p.col = rr - p.line * LINE;
  p.valid = p.line < PH && p.col < PW;
  p.slot = (lane & 1) ^ patch_swz(p.col);   // the channel slot (16 bytes of the half-chunk's 32) this lane fetches
  return p;
}
MW_HD int patch_piece_dst(int plane, int wave) { return (plane * PLANE + PIECE_ROWS * wave) * ROWB; }

// weight panel (3 taps x 64 output channels): wave w moves one full piece (rows 32 w .. 32 w + 31 = tap w >> 1, channels 32 (w & 1) ..) and
// one half piece (lanes 0..31: rows 128 + 16 w .. + 15 = tap 2, channels 16 w ..)
struct WLane { int tap, co, slot; };
MW_HD WLane wpanel_lane(int wave, int lane, int half_piece) {
  WLane r;
  const int row = half_piece ? 128 + 16 * wave + ((lane & 31) >> 1) : PIECE_ROWS * wave + (lane >> 1);
  r.tap = row / BN;
  r.co = row % BN;
  r.slot = (lane & 1) ^ row_swz(row);
  return r;
}
MW_HD int wpanel_piece_dst(int ring_slot, int wave, int half_piece) {
  return PATCH_BYTES + ring_slot * WBUF_BYTES + (half_piece ? (128 + 16 * wave) * ROWB : PIECE_ROWS * wave * ROWB);
}
// byte offset of (tap-in-image t, output channel co, half h, channel slot s) inside one 32-channel chunk image of the packed weights
MW_HD long long wsrc_offset(int t, int co, int cout_pad, int half, int slot) {
  return ((long long)t * cout_pad + co) * SRC_ROWB + half * ROWB + slot * 16;
}

// ---- operand fragments (ds_read_b128): lane supplies 8 k-values of row (lane & 31), k-half lane >> 5 ------------------------------------
// A: weight rows of output-channel block nb (32 channels) of tap u in ring slot s
MW_HD int a_lane_base(int lane) { return PATCH_BYTES + (lane & 31) * ROWB + (((lane >> 5) ^ row_swz(lane & 31)) << 4); }
MW_HD int a_offset(int ring_slot, int u, int nb) { return ring_slot * WBUF_BYTES + u * (BN * ROWB) + nb * (32 * ROWB); }
// B: patch rows of voxel block mb (lines 2 mb, 2 mb + 1 of plane `wave`) at tap (kd, kh, kw)
MW_HD int b_lane_base(int wave, int lane, int kw) {
  const int lc = (lane & 15) + kw;
  return (wave * PLANE + ((lane >> 4) & 1) * LINE + lc) * ROWB + (((lane >> 5) ^ patch_swz(lc)) << 4);
}
MW_HD int b_offset(int mb, int kd, int kh) { return (kd * PLANE + (2 * mb + kh) * LINE) * ROWB; }

// ---- accumulator layout of v_mfma_f32_32x32x16_bf16 (C/D): lane holds voxel column (lane & 31) and 16 output channels -------------------
MW_HD int acc_channel(int lane, int reg) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }  // within the 32-channel block
MW_HD int acc_voxel(int lane) { return lane & 31; }                                                  // within the 32-voxel block
// wave-local voxel index m = line * 16 + column of (block mb, lane): the row of the epilogue's transpose scratch
MW_HD int acc_row(int mb, int lane) { return mb * 32 + (lane & 31); }

// ---- fused 1x1 shortcut ------------------------------------------------------------------------------------------------------------------
// voxel rows of half-chunk j of a round: wave w DMAs its own 64 voxels (2 pieces: rows 64 w + 32 h + (lane >> 1)); the B fragment of block mb
// reads rows 64 w + 32 mb + (lane & 31)
MW_HD int sc_x_piece_dst(int j, int wave, int h) { return j * SC_XBYTES + (64 * wave + 32 * h) * ROWB; }
MW_HD int sc_x_lane_slot(int lane) { return (lane & 1) ^ row_swz(lane >> 1); }  // (row = 64 w + 32 h + (lane >> 1): bit 3 of the row = bit 3 of lane >> 1)
MW_HD int sc_b_lane_base(int wave, int lane) { return (64 * wave + (lane & 31)) * ROWB + (((lane >> 5) ^ row_swz(lane & 31)) << 4); }
MW_HD int sc_b_offset(int j, int mb) { return j * SC_XBYTES + mb * (32 * ROWB); }
// weight panel of half-chunk j: 64 rows = 2 pieces, both moved by wave j; the A fragment of block nb reads rows 32 nb + (lane & 31)
MW_HD int sc_w_piece_dst(int j, int h) { return SC_WOFF + j * SC_WBYTES + h * (32 * ROWB); }
MW_HD int sc_a_lane_base(int lane) { return SC_WOFF + (lane & 31) * ROWB + (((lane >> 5) ^ row_swz(lane & 31)) << 4); }
MW_HD int sc_a_offset(int j, int nb) { return j * SC_WBYTES + nb * (32 * ROWB); }
}  // namespace mw
