// Backward of the fused MLP w.r.t. the pre-activations of every layer (stage 1 of the parameter-gradient pass:
// see mlp_bwd.hip) on 16-point tiles: the mirror of mlp_fwd_x4.hip.  A wave carries 16 points; gradient tiles live in
// accumulators and feed W^T dY of the layer below straight from registers; the transposed weights stream through
// the slab ring; two waves per SIMD, eight per workgroup, one workgroup per 128-point tile of the saved activations.
// ReLU masks are the bit words the forward left behind each tile; every layer's dY goes to the workspace in the
// blocked layout the weight-gradient GEMM reads (mlp_layout.hpp), one 16-byte store per tile and lane.
#include "common.hpp"
#include "mlp_tile16.hpp"
#include "mlp_x4.hpp"

using namespace nerfhip;
using namespace nerflayout;
using namespace nerfmlp;
using namespace nerftile16;
using namespace nerfx4;

namespace {

// ---- transposed stream ---------------------------------------------------------------------------
// Sections (blocks of 1 KiB, layout of mlp_x4.hpp; every section is a whole number of slabs):
//   view branch: Wviews[:, :256]^T 128 (16 row tiles x 8 k-blocks) | Wfeature^T 256 |
//   then W7^T, W6^T, W5[:, 63:]^T, W4^T, ... W1^T, 256 each (16 x 16).
__host__ __device__ inline int x4b_stream_blocks(int use_viewdirs) { return use_viewdirs ? 128 + 8 * 256 : 7 * 256; }

// flat-parameter index of stream element (block, lane, j): M[16 to + r][16 kb + 4 g + j] with M = W^T
__host__ __device__ inline int64_t x4b_stream_source(int use_viewdirs, int output_ch, int blk, int lane, int j) {
    const ParamOffsets po = param_offsets(use_viewdirs, output_ch);
    const int r16 = lane & 15, g = lane >> 4;
    int to = 0, kb = 0;
    auto locate = [&](int b, int nkb) { const int w = b % (2 * nkb); kb = w >> 1; to = 2 * (b / (2 * nkb)) + (w & 1); };
    if (use_viewdirs) {
        if (blk < 128) {
            locate(blk, 8);
            return po.views_w + (int64_t)(16 * kb + 4 * g + j) * po.views_in + 16 * to + r16;
        }
        blk -= 128;
        if (blk < 256) {
            locate(blk, 16);
            return po.feat_w + (int64_t)(16 * kb + 4 * g + j) * W + 16 * to + r16;
        }
        blk -= 256;
    }
    const int layer = 7 - blk / 256;   // 7, 6, ..., 1: the section turns dY of this layer into dL/d(output of layer - 1)
    locate(blk % 256, 16);
    const int64_t row = 16 * kb + 4 * g + j, col = 16 * to + r16;
    if (layer == 5) return po.pts_w[5] + row * (W + IN_PTS) + IN_PTS + col;
    return po.pts_w[layer] + row * W + col;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// One backward-data layer: out = keep(start + M in), M = the section's W^T.  keep (MASK) zeroes the entries whose
// forward activation was not positive: bit 8 (t % 4) + 4 (g >> 1) + i of dword t / 4 of the mask word of lane
// (point, g & 1) stands for register i of tile t (mlp_fwd_x4.hip, save_mask).  ALPHA: start = d_sigma times the density
// head's weight column (the layer that also feeds that head), otherwise 0.  A finished tile pair is stored to `srows`
// (+ 4 t row groups), this lane's 16 bytes of the blocked workspace.
template <int NOUT, int NKB, bool MASK, bool ALPHA, typename In>
__device__ __forceinline__ void layer_x4b(StreamX4& st, int lane, int g, const float*& cur, In in, f32x4 (&out)[16],
                                          const float* alpha_g, float d_sigma, const u32x4& mw, float* srows) {
    auto start = [&]<int U>() {
        static_for<2>([&]<int TT>() {
            constexpr int t = 2 * U + TT;
            if constexpr (ALPHA) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(alpha_g + 16 * t);
#pragma unroll
                for (int i = 0; i < 4; ++i) out[t][i] = w[i] * d_sigma;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) out[t][i] = 0.0f;
            }
        });
    };
    const int shift = 4 * (g >> 1);
    auto finish = [&]<int U>() {
        static_for<2>([&]<int TT>() {
            constexpr int t = 2 * U + TT;
            if constexpr (MASK) {
                const uint32_t nib = mw[t >> 2] >> (8 * (t & 3) + shift);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t m = (uint32_t)((int32_t)(nib << (31 - i)) >> 31);
                    out[t][i] = __uint_as_float(__float_as_uint(out[t][i]) & m);
                }
            }
            save16(srows + 4 * t * GROUP_FLOATS, out[t]);
        });
    };
    x4_layer<NOUT, NKB>(st, lane, cur, in, out, start, finish);
}

template <bool VIEWDIRS>
__global__ void __launch_bounds__(X4_THREADS, 2)
k_mlp_bwd_data_x4(int64_t n_pts, int output_ch, const float* __restrict__ d_out, const float* __restrict__ act,
                  const float* __restrict__ packed_x4, const float* __restrict__ packed_bwd, float* __restrict__ scr) {
    extern __shared__ __attribute__((aligned(16))) float smem_x4b[];
    float* ring = smem_x4b;
    float* aux = smem_x4b + X4_RING_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    constexpr int UV = VIEWDIRS ? 1 : 0;
    const int pw = wave * 16 + (lane & 15);                       // point inside the 128-point tile
    const int64_t p = (int64_t)blockIdx.x * TILE_PTS + pw;
    const bool valid = p < n_pts;
    const float* const at = act + (int64_t)blockIdx.x * act_rows(UV) * TILE_PTS;
    float* const st_ = scr + (int64_t)blockIdx.x * scr_rows(UV) * TILE_PTS;
    float* const lane_rows = st_ + g * GROUP_FLOATS + pw * 4;
    auto rows = [&](int row) { return lane_rows + (row / 4) * GROUP_FLOATS; };
    constexpr int SCR_HEAD = scr_head(UV), SCR_HEAD2 = SCR_HEAD + 32;
    {   // the fp32 tables behind the forward stream: head rows in natural feature order
        const float* gaux = packed_x4 + (int64_t)(VIEWDIRS ? 2336 : 1920) * 256;
        for (int i = tid * 4; i < AUX16_FLOATS; i += X4_THREADS * 4)
            *reinterpret_cast<f32x4*>(aux + i) = *reinterpret_cast<const f32x4*>(gaux + i);
    }
    StreamX4 st;
    st.ring = ring;
    st.mine = ring + wave * X4_WAVE_BLOCKS * 256;
    st.g = packed_bwd + wave * X4_WAVE_BLOCKS * 256 + lane * 4;
    st.next = 0;
    st.nslabs = x4b_stream_blocks(UV) / X4_SLAB;
    __syncthreads();
    st.fetch(0);
    const float* cur = nullptr;

    // the mask word of masked layer m that holds this lane's bits (written by lane (point, g & 1) of the forward)
    auto mask_word = [&](int m) {
        return *reinterpret_cast<const u32x4*>(at + act_mask_offset(UV, m, pw >> 5, (g & 1) * 32 + (pw & 31)));
    };
    const int shift = 4 * (g >> 1);
    auto on = [&](const u32x4& mw, int t, int i) { return ((mw[t >> 2] >> (8 * (t & 3) + shift + i)) & 1u) != 0u; };

    f32x4 P[16], Q[16];
    auto inP = [&]<int kb, int j>() -> float { return P[kb][j]; };
    auto inQ = [&]<int kb, int j>() -> float { return Q[kb][j]; };
    const float* hw = aux + AUX16_HEAD;
    const u32x4 none = {0u, 0u, 0u, 0u};

    if constexpr (VIEWDIRS) {
        f32x4 gd = {0, 0, 0, 0};
        if (valid) gd = *reinterpret_cast<const f32x4*>(d_out + p * 4);
        const float d_sigma = gd[3];
        if (g == 0) {   // head operands of the weight-gradient GEMM: one row group each, zero padded
            const f32x4 ga = {gd[0], gd[1], gd[2], 0.0f}, gb = {gd[3], 0.0f, 0.0f, 0.0f};
            *reinterpret_cast<f32x4*>(st_ + (SCR_HEAD / 4) * GROUP_FLOATS + pw * 4) = ga;
            *reinterpret_cast<f32x4*>(st_ + (SCR_HEAD2 / 4) * GROUP_FLOATS + pw * 4) = gb;
        }
        // views layer: d(hidden) = Wrgb^T d_rgb, masked -> P[0..7], stored
        const u32x4 m8 = mask_word(8);
        const float* wrgb = hw + 256 + 4 * g;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wrgb + 16 * t);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wrgb + 128 + 16 * t);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(wrgb + 256 + 16 * t);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = fmaf(w2[i], gd[2], fmaf(w1[i], gd[1], w0[i] * gd[0]));
                P[t][i] = on(m8, t, i) ? v : 0.0f;
            }
            save16(rows(SCR_VIEWS) + 4 * t * GROUP_FLOATS, P[t]);
        }
        // d(feature) = Wviews[:, :256]^T dY_views -> Q (feature_linear has no activation)
        layer_x4b<16, 8, false, false>(st, lane, g, cur, inP, Q, nullptr, 0.0f, none, rows(SCR_FEAT));
        // d(h7) = Wfeature^T d(feature) + Walpha^T d_sigma, masked -> P
        const u32x4 m7 = mask_word(7);
        layer_x4b<16, 16, true, true>(st, lane, g, cur, inQ, P, hw + 4 * g, d_sigma, m7, rows(SCR_H0 + 7 * 256));
    } else {
        // d(h7) = Wout^T d_out, masked -> P, stored
        float gd[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) gd[c] = (valid && c < output_ch) ? d_out[p * output_ch + c] : 0.0f;
        if (g == 0) {
            const f32x4 ga = {gd[0], gd[1], gd[2], gd[3]}, gb = {gd[4], gd[5], gd[6], gd[7]};
            *reinterpret_cast<f32x4*>(st_ + (SCR_HEAD / 4) * GROUP_FLOATS + pw * 4) = ga;
            *reinterpret_cast<f32x4*>(st_ + (SCR_HEAD / 4 + 1) * GROUP_FLOATS + pw * 4) = gb;
        }
        const u32x4 m7 = mask_word(7);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            f32x4 v = {0, 0, 0, 0};
            for (int c = 0; c < output_ch; ++c) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(hw + c * 256 + 16 * t + 4 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaf(w[i], gd[c], v[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) P[t][i] = on(m7, t, i) ? v[i] : 0.0f;
            save16(rows(SCR_H0 + 7 * 256) + 4 * t * GROUP_FLOATS, P[t]);
        }
    }

    // dY of pts layers 6 .. 1, two per trip: P (layer l + 1) -> Q (layer l) -> P (layer l - 1); then layer 0
#pragma unroll 1
    for (int l = 6; l > 0; l -= 2) {
        const u32x4 ma = mask_word(l);
        layer_x4b<16, 16, true, false>(st, lane, g, cur, inP, Q, nullptr, 0.0f, ma, rows(SCR_H0 + 256 * l));
        const u32x4 mb = mask_word(l - 1);
        layer_x4b<16, 16, true, false>(st, lane, g, cur, inQ, P, nullptr, 0.0f, mb, rows(SCR_H0 + 256 * (l - 1)));
    }
    const u32x4 m0 = mask_word(0);
    layer_x4b<16, 16, true, false>(st, lane, g, cur, inP, Q, nullptr, 0.0f, m0, rows(SCR_H0));
}

// flat parameters -> transposed x4 stream
__global__ void __launch_bounds__(256) k_pack_x4b(int use_viewdirs, int output_ch, const float* __restrict__ params,
                                                  float* __restrict__ out) {
    const int64_t total = (int64_t)x4b_stream_blocks(use_viewdirs) * 256;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = params[x4b_stream_source(use_viewdirs, output_ch, (int)(i >> 8), (int)((i >> 2) & 63), (int)(i & 3))];
}

constexpr size_t X4B_LDS_BYTES = (X4_RING_FLOATS + AUX16_FLOATS) * sizeof(float);

}  // namespace

namespace nerfx4 {

int launch_bwd_data_x4(int64_t n_pts, int use_viewdirs, int output_ch, const float* d_out, const float* act,
                       const float* packed_x4, const float* packed_bwd_x4, float* scratch, hipStream_t stream) {
    const int64_t blocks = padded_points(n_pts) / TILE_PTS;
    hipError_t e;
    if (use_viewdirs) {
        e = lds_limit_once(reinterpret_cast<const void*>(k_mlp_bwd_data_x4<true>), X4B_LDS_BYTES);
        if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_bwd_x4: LDS attribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(k_mlp_bwd_data_x4<true>, dim3((unsigned)blocks), dim3(X4_THREADS), X4B_LDS_BYTES, stream, n_pts, 4, d_out,
                           act, packed_x4, packed_bwd_x4, scratch);
    } else {
        e = lds_limit_once(reinterpret_cast<const void*>(k_mlp_bwd_data_x4<false>), X4B_LDS_BYTES);
        if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_bwd_x4: LDS attribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(k_mlp_bwd_data_x4<false>, dim3((unsigned)blocks), dim3(X4_THREADS), X4B_LDS_BYTES, stream, n_pts,
                           output_ch, d_out, act, packed_x4, packed_bwd_x4, scratch);
    }
    return launched("nerf_mlp_bwd_x4 (data)");
}

}  // namespace nerfx4

extern "C" {

int64_t nerf_mlp_packed_bwd_x4_floats(int use_viewdirs) { return (int64_t)x4b_stream_blocks(use_viewdirs) * 256; }

int nerf_pack_weights_bwd_x4(int use_viewdirs, int output_ch, const float* params, float* packed_bwd_x4, void* stream) {
    NERF_REQUIRE(params && packed_bwd_x4, "nerf_pack_weights_bwd_x4: null pointer");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_pack_weights_bwd_x4: output_ch=%d not in 1..8", output_ch);
    hipLaunchKernelGGL(k_pack_x4b, dim3(1024), dim3(256), 0, as_stream(stream), use_viewdirs, output_ch, params, packed_bwd_x4);
    return launched("nerf_pack_weights_bwd_x4");
}

}  // extern "C"
