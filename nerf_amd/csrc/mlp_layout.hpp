// Layout of the packed weight streams and of the saved-activation buffer of the fused MLP kernels.
// Pure index arithmetic, usable from host and device code.
//
// Register picture (one wavefront = 32 points): an activation "tile" is 16 VGPRs per lane holding
// 32 features x 32 points; lane l belongs to point l&31, and its register r of tile t holds
// feature 32t + (r&3) + 8(r>>2) + 4(l>>5) — exactly the C/D layout of v_mfma_f32_32x32x2_f32, so
// the accumulators of one layer are, register by register, the B operands (k = feature, lanes
// 0-31 supply k, lanes 32-63 supply k+4) of the next layer's MFMAs and never leave the register
// file.  The matching A operand (32 output rows x 2 k) is one float per lane read from the stream.
//
// A stream is a sequence of 1 KiB "blocks": block = [64 lanes][4 floats]; float e of lane l of the
// block (to, ti, q) of a section holds  W[row 32*to + (l&31)][column of input tile ti, register
// 4q+e, half l>>5].  Blocks are ordered to-major, then ti, then q, so a lane reads them with
// consecutive ds_read_b128 and feeds 4 MFMAs per read.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define NL_HD __host__ __device__
#else
#define NL_HD
#endif

namespace nerflayout {

constexpr int BLOCK_FLOATS = 256;        // 1 KiB
constexpr int SLAB_BLOCKS = 16;          // blocks staged per LDS buffer (every section starts on a slab boundary)
constexpr int SLAB_FLOATS = SLAB_BLOCKS * BLOCK_FLOATS;
constexpr int W = 256;                   // hidden width
constexpr int IN_PTS = 63, IN_DIRS = 27; // encoded widths (L = 10, L = 4)
constexpr int L_PTS = 10, L_DIRS = 4;
constexpr int TILE_PTS = 128;            // points per workgroup (4 waves x 32)

// feature held by register r (0..15) of activation tile t in lane half h
NL_HD inline int tile_feature(int t, int r, int h) { return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h; }

// encoded-position tiles E0,E1: 32 register slots per lane half; slot -> encoded feature (or -1 = padding).
// Each half evaluates 15 of the 30 (frequency, axis) pairs with one sincos and keeps both results.
NL_HD inline int enc_pts_feature(int slot, int h) {
    if (slot == 0) return h ? 2 : 0;
    if (slot == 1) return h ? -1 : 1;
    const int i = (slot - 2) >> 1, which = (slot - 2) & 1;
    const int c = 15 * h + i, k = c / 3, d = c % 3;
    return 3 + 6 * k + 3 * which + d;
}
// encoded-direction tile D: 16 slots per half, 6 of the 12 (frequency, axis) pairs each.
NL_HD inline int enc_dirs_feature(int slot, int h) {
    if (slot == 0) return h ? 2 : 0;
    if (slot == 1) return h ? -1 : 1;
    if (slot >= 14) return -1;
    const int i = (slot - 2) >> 1, which = (slot - 2) & 1;
    const int c = 6 * h + i, k = c / 3, d = c % 3;
    return 3 + 6 * k + 3 * which + d;
}

// ---- flat parameter vector (state_dict order of the module) -----------------------------------
struct ParamOffsets {
    int64_t pts_w[8], pts_b[8];
    int64_t views_w, views_b;        // views_linears.0
    int64_t feat_w, feat_b, alpha_w, alpha_b, rgb_w, rgb_b;  // view branch
    int64_t out_w, out_b;            // output_linear (no view branch)
    int64_t total;
    int views_in;                    // columns of views_linears.0.weight
};

NL_HD inline ParamOffsets param_offsets(int use_viewdirs, int output_ch) {
    ParamOffsets o{};
    int64_t p = 0;
    for (int i = 0; i < 8; ++i) {
        const int in = (i == 0) ? IN_PTS : (i == 5 ? W + IN_PTS : W);
        o.pts_w[i] = p; p += (int64_t)W * in;
        o.pts_b[i] = p; p += W;
    }
    o.views_in = W + (use_viewdirs ? IN_DIRS : 0);
    o.views_w = p; p += (int64_t)(W / 2) * o.views_in;
    o.views_b = p; p += W / 2;
    if (use_viewdirs) {
        o.feat_w = p; p += (int64_t)W * W;
        o.feat_b = p; p += W;
        o.alpha_w = p; p += W;
        o.alpha_b = p; p += 1;
        o.rgb_w = p; p += 3 * (W / 2);
        o.rgb_b = p; p += 3;
        o.out_w = o.out_b = -1;
    } else {
        o.feat_w = o.feat_b = o.alpha_w = o.alpha_b = o.rgb_w = o.rgb_b = -1;
        o.out_w = p; p += (int64_t)output_ch * W;
        o.out_b = p; p += output_ch;
    }
    o.total = p;
    return o;
}

// ---- forward stream ------------------------------------------------------------------------------
// sections, in consumption order (blocks): L0 [64] | L1..L4 [256 each] | L5 encoded-input part [64] |
// L5 hidden part [256] | L6, L7 [256 each] | (view branch) feature [256] | views [144]
constexpr int FWD_SEC_L0 = 0, FWD_SEC_L5E = 5, FWD_SEC_L5H = 6, FWD_SEC_FEAT = 9, FWD_SEC_VIEWS = 10;
NL_HD inline int fwd_section_blocks(int sec) {
    if (sec == FWD_SEC_L0 || sec == FWD_SEC_L5E) return 64;
    if (sec == FWD_SEC_VIEWS) return 144;
    return 256;
}
NL_HD inline int fwd_num_sections(int use_viewdirs) { return use_viewdirs ? 11 : 9; }
NL_HD inline int fwd_stream_blocks(int use_viewdirs) { return use_viewdirs ? 2320 : 1920; }

// aux tables appended after the stream blocks (floats), loaded into LDS once per workgroup:
//   bias[10][256] as [layer][tile][half][16]  (layers 0..7, feature, views(128 used))
//   head weights: view branch: alpha [8 tiles][2][16], rgb [3][4 tiles][2][16], then alpha_b, rgb_b[3]
//                 else: output_linear [out_ch][8][2][16], then out_b[out_ch]
constexpr int AUX_BIAS = 0;
constexpr int AUX_HEAD = 10 * 256;
constexpr int AUX_HEAD_MAX = 8 * 256 + 8;   // room for output_ch up to 8
constexpr int AUX_FLOATS = AUX_HEAD + AUX_HEAD_MAX;
NL_HD inline int64_t fwd_packed_floats(int use_viewdirs) {
    return (int64_t)fwd_stream_blocks(use_viewdirs) * BLOCK_FLOATS + AUX_FLOATS;
}

// Source of float `idx` of the forward packed buffer: index into the flat parameter vector, or -1 (zero).
NL_HD inline int64_t fwd_source(int use_viewdirs, int output_ch, int64_t idx) {
    const ParamOffsets po = param_offsets(use_viewdirs, output_ch);
    const int64_t stream = (int64_t)fwd_stream_blocks(use_viewdirs) * BLOCK_FLOATS;
    if (idx < stream) {
        int64_t blk = idx / BLOCK_FLOATS;
        const int within = (int)(idx % BLOCK_FLOATS);
        const int lane = within >> 2, e = within & 3, h = lane >> 5, row32 = lane & 31;
        int sec = 0;
        while (blk >= fwd_section_blocks(sec)) { blk -= fwd_section_blocks(sec); ++sec; }
        const int q = (int)(blk & 3);
        const int r = 4 * q + e;
        if (sec == FWD_SEC_L0 || sec == FWD_SEC_L5E) {           // inputs: E0, E1
            const int ti = (int)((blk >> 2) % 2), to = (int)(blk >> 3);
            const int f = enc_pts_feature(16 * ti + r, h);
            if (f < 0) return -1;
            const int row = 32 * to + row32;
            return sec == FWD_SEC_L0 ? po.pts_w[0] + (int64_t)row * IN_PTS + f
                                     : po.pts_w[5] + (int64_t)row * (W + IN_PTS) + f;
        }
        if (sec == FWD_SEC_VIEWS) {                              // inputs: 8 feature tiles, D
            const int ti = (int)((blk >> 2) % 9), to = (int)(blk / 36);
            const int row = 32 * to + row32;
            int col;
            if (ti < 8) col = tile_feature(ti, r, h);
            else {
                const int f = enc_dirs_feature(r, h);
                if (f < 0) return -1;
                col = W + f;
            }
            return po.views_w + (int64_t)row * po.views_in + col;
        }
        const int ti = (int)((blk >> 2) % 8), to = (int)(blk >> 5);
        const int row = 32 * to + row32, col = tile_feature(ti, r, h);
        if (sec == FWD_SEC_FEAT) return po.feat_w + (int64_t)row * W + col;
        if (sec == FWD_SEC_L5H) return po.pts_w[5] + (int64_t)row * (W + IN_PTS) + IN_PTS + col;
        const int layer = sec <= 4 ? sec : sec - 1;              // sections 7,8 are layers 6,7
        return po.pts_w[layer] + (int64_t)row * W + col;
    }
    int a = (int)(idx - stream);
    if (a < AUX_HEAD) {
        const int layer = a >> 8, w = a & 255, t = w >> 5, h = (w >> 4) & 1, r = w & 15;
        const int f = tile_feature(t, r, h);
        if (layer < 8) return po.pts_b[layer] + f;
        if (!use_viewdirs) return -1;
        if (layer == 8) return po.feat_b + f;
        return f < W / 2 ? po.views_b + f : -1;
    }
    a -= AUX_HEAD;
    if (use_viewdirs) {
        if (a < 256) return po.alpha_w + tile_feature(a >> 5, a & 15, (a >> 4) & 1);
        a -= 256;
        if (a < 384) {
            const int c = a >> 7, w = a & 127;
            return po.rgb_w + (int64_t)c * (W / 2) + tile_feature(w >> 5, w & 15, (w >> 4) & 1);
        }
        a -= 384;
        if (a == 0) return po.alpha_b;
        if (a < 4) return po.rgb_b + (a - 1);
        return -1;
    }
    if (a < output_ch * 256) {
        const int c = a >> 8, w = a & 255;
        return po.out_w + (int64_t)c * W + tile_feature(w >> 5, w & 15, (w >> 4) & 1);
    }
    a -= output_ch * 256;
    if (a < output_ch) return po.out_b + a;
    return -1;
}

// ---- saved activations (training) --------------------------------------------------------------
// Blocked by forward workgroup: tile T = points [128 T, 128 T + 128) holds all rows of those points, and
// within a tile rows are grouped by four:
//     float (row, point p)  at  ((T * rows/4 + row/4) * 128 + p % 128) * 4 + row % 4.
// Four consecutive accumulator registers of a lane are four consecutive feature rows of its point, so
// the MLP kernels save and reload activations as one 16-byte access per register quad, a wavefront's
// access covers 512 contiguous bytes per lane half, and a workgroup's whole tile is one contiguous
// region.  The weight-gradient GEMM stages [group][32 points][4 rows] chunks, 512 contiguous bytes per
// group, and reads a lane's four rows with one ds_read_b128.  n_pts is rounded up to whole tiles;
// padding points hold finite values (their gradients are exactly zero).  Rows:
//   rows   0.. 63  encoded positions (row 63 zero)        rows 64.. 95  encoded directions (27 used)
//   rows  96+256*l .. +255  output of pts layer l after ReLU (l = 0..7)
//   rows 2144..2399 feature_linear output                  rows 2400..2527 views layer output after ReLU
// Behind the rows of a tile: the ReLU masks, one bit per activation, as the backward-data kernel wants them:
// 9 masked layers (pts 0..7, views) x [wave 4][lane 64] x 128 bits; bit 16 (t%2) + r of dword t/2 = "register r of
// tile t of this lane was positive".  Two row groups (4 KiB) per layer, 72 rows in all.
constexpr int ACT_ENC_PTS = 0, ACT_ENC_DIRS = 64, ACT_H0 = 96, ACT_FEAT = 96 + 8 * 256, ACT_VIEWS = ACT_FEAT + 256;
constexpr int ACT_ROWS = ACT_VIEWS + 128;
constexpr int ACT_MASK_ROWS = 9 * 8;
constexpr int GROUP_FLOATS = TILE_PTS * 4;   // one group: 4 rows x 128 points
NL_HD constexpr int act_data_rows(int use_viewdirs) { return use_viewdirs ? ACT_ROWS : ACT_FEAT; }
NL_HD constexpr int act_rows(int use_viewdirs) { return act_data_rows(use_viewdirs) + ACT_MASK_ROWS; }
// float offset, inside a tile, of this lane's 16-byte mask word of masked layer m (0..7: pts layers, 8: views layer)
NL_HD constexpr int act_mask_offset(int use_viewdirs, int m, int wave, int lane) {
    return (act_data_rows(use_viewdirs) / 4 + 2 * m) * GROUP_FLOATS + (wave * 64 + lane) * 4;
}
NL_HD inline int64_t padded_points(int64_t n_pts) { return (n_pts + TILE_PTS - 1) / TILE_PTS * TILE_PTS; }
NL_HD inline int64_t act_floats(int use_viewdirs, int64_t n_pts) {
    return (int64_t)act_rows(use_viewdirs) * padded_points(n_pts);
}
NL_HD inline int64_t blocked_index(int rows_per_tile, int row, int64_t p) {
    return ((p / TILE_PTS) * (rows_per_tile / 4) + row / 4) * GROUP_FLOATS + (p % TILE_PTS) * 4 + row % 4;
}

// ---- backward workspace --------------------------------------------------------------------------
// dL/d(pre-activation) of every layer in the blocked layout of the activations, with the same number of
// rows per tile as the activations (one tile stride serves both operands of the weight-gradient GEMM):
//   rows 0..2047 pts layers 0..7 | (view branch) 2048..2303 feature_linear | 2304..2431 views layer |
//   head rows A (8 rows at scr_head): dL/d(rgb) and a zero, or dL/d(out channel c) zero padded to 8
//   head rows B (4 rows at scr_head + 32, view branch): dL/d(sigma) and three zeros.
// Then KSPLIT slabs of partial parameter gradients (one per slice of the point range), each
// `slab_floats` long, summed in slab order by the final reduction.
constexpr int SCR_H0 = 0, SCR_FEAT = 8 * 256, SCR_VIEWS = SCR_FEAT + 256;
NL_HD constexpr int scr_head(int use_viewdirs) { return use_viewdirs ? SCR_VIEWS + 128 : 8 * 256; }
NL_HD constexpr int scr_rows(int use_viewdirs) { return act_rows(use_viewdirs); }
static_assert(SCR_VIEWS + 128 + 64 <= ACT_ROWS + ACT_MASK_ROWS && 8 * 256 + 64 <= ACT_FEAT + ACT_MASK_ROWS,
              "head rows fit below the activations' row count");
constexpr int KSPLIT = 32;               // upper bound of point-range slices per weight tile
NL_HD inline int64_t slab_floats(int use_viewdirs, int output_ch) {
    return (param_offsets(use_viewdirs, output_ch).total + 63) / 64 * 64;
}
NL_HD inline int64_t scratch_floats(int use_viewdirs, int output_ch, int64_t n_pts) {
    return (int64_t)scr_rows(use_viewdirs) * padded_points(n_pts) + (int64_t)KSPLIT * slab_floats(use_viewdirs, output_ch);
}

}  // namespace nerflayout
