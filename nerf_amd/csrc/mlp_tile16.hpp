// Pieces shared by the MLP kernels whose activation tiles are 16 points wide (lane l = point l & 15, lane group
// g = l >> 4): which encoded feature each lane group evaluates, and the fp32 tables (biases, head rows) that sit
// behind a packed weight stream.
#pragma once
#include "mlp_layout.hpp"

namespace nerftile16 {
using namespace nerflayout;

// Encoded position: 64 slots per point = 2 k-steps x 4 groups x 8; group g evaluates the (frequency, axis) pairs
// c0(g) .. c0(g)+np(g)-1 (8, 8, 7, 7 of the 30) and keeps sin and cos side by side; the raw coordinates fill
// the last slots of groups 2 and 3.  -> encoded feature index (run_nerf_helpers.py:13-66 order), or -1 (padding).
__host__ __device__ inline int enc16_first_pair(int g) { return g < 2 ? 8 * g : 16 + 7 * (g - 2); }
__host__ __device__ inline int enc16_pts_feature(int ks, int g, int j) {
    const int q = 8 * ks + j, i = q >> 1, np = g < 2 ? 8 : 7;
    if (i < np) {
        const int c = enc16_first_pair(g) + i, k = c / 3, d = c % 3;
        return 3 + 6 * k + 3 * (q & 1) + d;
    }
    if (g == 2) return q - 14;            // x, y
    return q == 14 ? 2 : -1;              // z, padding
}
// Encoded direction: 32 slots = 4 groups x 8; group g evaluates pairs 3g .. 3g+2, raw components in slots 6, 7.
__host__ __device__ inline int enc16_dirs_feature(int g, int j) {
    if (j < 6) {
        const int c = 3 * g + (j >> 1), k = c / 3, d = c % 3;
        return 3 + 6 * k + 3 * (j & 1) + d;
    }
    if (g == 0) return j - 6;             // x, y
    return (g == 1 && j == 6) ? 2 : -1;   // z
}

// aux tables behind the stream (floats, natural feature order), copied to LDS once per workgroup:
//   bias[10][256] (layers 0..7, feature, views (128 used)); view branch: alpha[256], rgb[3][128], alpha_b, rgb_b[3];
//   otherwise output_linear[8][256], out_b[8]
constexpr int AUX16_BIAS = 0, AUX16_HEAD = 10 * 256, AUX16_FLOATS = AUX16_HEAD + 8 * 256 + 8;
__host__ __device__ inline int64_t aux16_source(int use_viewdirs, int output_ch, int a) {
    const ParamOffsets po = param_offsets(use_viewdirs, output_ch);
    if (a < AUX16_HEAD) {
        const int layer = a >> 8, f = a & 255;
        if (layer < 8) return po.pts_b[layer] + f;
        if (!use_viewdirs) return -1;
        if (layer == 8) return po.feat_b + f;
        return f < W / 2 ? po.views_b + f : -1;
    }
    a -= AUX16_HEAD;
    if (use_viewdirs) {
        if (a < 256) return po.alpha_w + a;
        a -= 256;
        if (a < 384) return po.rgb_w + a;
        a -= 384;
        if (a == 0) return po.alpha_b;
        return a < 4 ? po.rgb_b + (a - 1) : -1;
    }
    if (a < 8 * 256) return (a >> 8) < output_ch ? po.out_w + a : -1;
    a -= 8 * 256;
    return a < output_ch ? po.out_b + a : -1;
}

}  // namespace nerftile16
