// Pieces shared by the weight-gradient stages of the parameter-gradient pass (mlp_bwd.hip: fp32 matrix instruction;
// mlp_dw_x3.hip: split operands on the bf16 matrix instruction).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "mlp_layout.hpp"

namespace nerfdw {

// Upper bound of the slices (= partial-gradient slabs) of one task for `ldp` padded points: a slice is at least 256
// points long, and there are at most KSPLIT of them.  Every caller that sizes a slab area uses this bound.
inline int max_slabs(int64_t ldp) {
    const int64_t by_len = ldp / 256;
    return (int)(by_len < 1 ? 1 : (by_len > nerflayout::KSPLIT ? nerflayout::KSPLIT : by_len));
}

// d_params[i] = sum over the first nslabs slabs, in slab order (mlp_bwd.hip)
int reduce_slabs(const float* slabs, int64_t slab_stride, int64_t n, int nslabs, float* d_params, hipStream_t stream);

// The same sum restricted to what the tasks wrote: region r is a rows x cols block (row stride ld) of the parameter
// vector at `out`, present in the first nsplit slabs only.  Slab entries outside every region are never read, so the
// slabs need no clearing; d_params entries outside every region are the caller's to zero.
struct SlabRegion {
    int64_t out;
    int rows, cols, ld, nsplit;
};
struct SlabRegions {
    SlabRegion r[44];
    int n;
};
int reduce_regions(const SlabRegions& regions, const float* slabs, int64_t slab_stride, float* d_params, hipStream_t stream);

// all weight and bias gradients from the saved activations `act` and the dY rows in `scratch` (mlp_dw_x3.hip)
int run_dw_x3(int64_t ldp, int use_viewdirs, int output_ch, const float* act, float* scratch, float* slabs, int64_t slab,
              float* d_params, hipStream_t stream);

// ---- the two-piece form of the same stage (mlp_dw_x3.hip, k_dw_x3<true>) ----------------------------------------
// The two-piece backward-data kernel (mlp_bwd_f16x2.hip) leaves, per 128-point tile, DY_STAGES words: the largest
// magnitude it stored for each group of dY rows as exponent bits + 1 (0 = nothing but zeros).  Stages 0..7: the head
// rows, one per channel of dL/d(raw) (view branch: 0..2 colour, 4 density); 8: view layer; 9: feature layer; 17 - l:
// pts layer l.  One small kernel turns them, and the tile
// flags of both two-piece stages, into DW_CTL_PARTS partial results of DW_CTL_STRIDE words each (DW_CTL_WORDS of
// workspace), which every weight-gradient workgroup reduces to: any tile marked (the two-piece kernel then leaves the
// whole batch to the three-piece one), and per stage the power of two that takes its largest magnitude below 2^14.
constexpr int DY_STAGES = 18;
constexpr int DW_CTL_PARTS = 128, DW_CTL_STRIDE = 32;
constexpr int DW_CTL_WORDS = DW_CTL_PARTS * DW_CTL_STRIDE;
constexpr int DY_STAGE_VIEWS = 8, DY_STAGE_FEAT = 9;
constexpr int dy_stage_layer(int l) { return 17 - l; }
int run_dw_x2(int64_t ldp, int use_viewdirs, int output_ch, const float* act, float* scratch, float* slabs, int64_t slab,
              float* d_params, const uint32_t* fwd_flags, const uint32_t* bwd_flags, const uint32_t* tile_max, uint32_t* ctl,
              hipStream_t stream);

}  // namespace nerfdw
