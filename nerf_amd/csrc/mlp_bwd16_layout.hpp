// Transposed weight stream of the backward-data kernels built on the 16x16x32 matrix instructions
// (mlp_bwd_bf16x3.hip, mlp_bwd_f16x2.hip): which parameter each element of each A fragment of W^T is.
#pragma once
#include "mlp_bf16_layout.hpp"

namespace nerfbwd16 {
using namespace nerflayout;
using namespace nerfbf16;

// Sections (blocks = A fragments of M = W^T; row tile by row tile, a tile's k-steps consecutively; every section is a
// whole number of 8-block slabs):
//   view branch: Wviews[:, :256]^T 64 (16 row tiles x 4 k-steps) | Wfeature^T 128 (16 x 8) |
//   then W7^T, W6^T, W5[:, 63:]^T, W4^T, ... W1^T, 128 each.
__host__ __device__ inline int bwd16_stream_blocks(int use_viewdirs) { return use_viewdirs ? 64 + 8 * 128 : 7 * 128; }

// flat-parameter index of stream element (block t, lane, j): M[16 to + r][32 ks + kslot(g, j)]
__host__ __device__ inline int64_t bwd16_stream_source(int use_viewdirs, int output_ch, int t, int lane, int j) {
    const ParamOffsets po = param_offsets(use_viewdirs, output_ch);
    const int r16 = lane & 15, g = lane >> 4;
    if (use_viewdirs) {
        if (t < 64) return po.views_w + (int64_t)(32 * (t & 3) + kslot_feature(g, j)) * po.views_in + 16 * (t >> 2) + r16;
        t -= 64;
        if (t < 128) return po.feat_w + (int64_t)(32 * (t & 7) + kslot_feature(g, j)) * W + 16 * (t >> 3) + r16;
        t -= 128;
    }
    const int layer = 7 - t / 128;   // 7, 6, ..., 1: the section turns dY of this layer into dL/d(output of layer - 1)
    t %= 128;
    const int64_t row = 32 * (t & 7) + kslot_feature(g, j), col = 16 * (t >> 3) + r16;
    if (layer == 5) return po.pts_w[5] + row * (W + IN_PTS) + IN_PTS + col;
    return po.pts_w[layer] + row * W + col;
}

}  // namespace nerfbwd16
