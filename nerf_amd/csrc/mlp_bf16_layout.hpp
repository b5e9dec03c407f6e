// Weight-stream layout of the MLP kernels built on v_mfma_f32_16x16x32_bf16 (mlp_fwd_bf16.hip, mlp_fwd_bf16x3.hip).
#pragma once
#include "mlp_tile16.hpp"

namespace nerfbf16 {
using namespace nerflayout;
using namespace nerftile16;

// ---- stream layout ------------------------------------------------------------------------------
// Block = one A fragment of v_mfma_f32_16x16x32_bf16: [64 lanes][8 bf16]; lane l, element j holds
// W[row 16*to + (l&15)][input feature of k-step ks, group l>>4, slot j].  Blocks are ordered row tile by
// row tile, a tile's k-steps consecutively.  Sections (blocks):
//   L0 32 (16 tiles x 2 k-steps of encoded position) | L1..L4 128 each | L5 160 (per tile: 8 hidden + 2 encoded) |
//   L6, L7 128 | feature 128 | views 72 (8 tiles x (8 feature + 1 encoded direction)); every section starts on a slab.
__host__ __device__ inline int bf16_stream_blocks(int use_viewdirs) { return use_viewdirs ? 1160 : 960; }

// feature (inside a 32-feature k-step) that k-slot j of lane group g stands for
__host__ __device__ inline int kslot_feature(int g, int j) { return j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4); }

// flat-parameter index of stream element (block, lane, j), or -1 for zero padding
__host__ __device__ inline int64_t bf16_stream_source(int use_viewdirs, int output_ch, int blk, int lane, int j) {
    const ParamOffsets po = param_offsets(use_viewdirs, output_ch);
    const int r16 = lane & 15, g = lane >> 4;
    if (blk < 32) {
        const int f = enc16_pts_feature(blk & 1, g, j);
        return f < 0 ? -1 : po.pts_w[0] + (int64_t)(16 * (blk >> 1) + r16) * IN_PTS + f;
    }
    blk -= 32;
    for (int l = 1; l < 8; ++l) {
        const int n = (l == 5) ? 160 : 128;
        if (blk < n) {
            const int per = (l == 5) ? 10 : 8, row = 16 * (blk / per) + r16, ks = blk % per;
            const int ld = (l == 5) ? W + IN_PTS : W;
            if (ks < 8) return po.pts_w[l] + (int64_t)row * ld + (l == 5 ? IN_PTS : 0) + 32 * ks + kslot_feature(g, j);
            const int f = enc16_pts_feature(ks - 8, g, j);
            return f < 0 ? -1 : po.pts_w[5] + (int64_t)row * ld + f;
        }
        blk -= n;
    }
    if (!use_viewdirs) return -1;
    if (blk < 128) return po.feat_w + (int64_t)(16 * (blk >> 3) + r16) * W + 32 * (blk & 7) + kslot_feature(g, j);
    blk -= 128;
    const int row = 16 * (blk / 9) + r16, ks = blk % 9;
    if (ks < 8) return po.views_w + (int64_t)row * po.views_in + 32 * ks + kslot_feature(g, j);
    const int f = enc16_dirs_feature(g, j);
    return f < 0 ? -1 : po.views_w + (int64_t)row * po.views_in + W + f;
}


}  // namespace nerfbf16
