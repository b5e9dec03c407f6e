// Fused positional encoding + NeRF MLP forward in fp32 for inference, on v_mfma_f32_16x16x4_f32.
//
// Same arithmetic as mlp_fwd.hip (fp32 products, fp32 sums); different shape.  The 32x32x2 kernel keeps 32 points
// per wave, i.e. 2 x 128 activation registers: one wave per SIMD, so nothing runs under a wave's own weight
// fetch and epilogue.  Here a wave carries 16 points:
//   * an output tile is 16 features x 16 points = 4 registers per lane (lane l = point l & 15, lane group
//     g = l >> 4, register i = feature 16 t + 4 g + i), a layer's activations are 16 such tiles = 64 registers,
//     two arrays ping-pong: under 256 registers, TWO waves per SIMD, eight per workgroup (128 points);
//   * register i of input tile t is, as it stands, the B operand of one MFMA whose four k-slots (the lane groups)
//     are features 16 t + 4 g + i; the weight stream is packed in that k order: a 1 KiB block holds, per lane,
//     W[16 to + (l & 15)][16 t + 4 g + 0..3] -- four consecutive floats of a weight row;
//   * two row tiles are multiplied together, MFMA by MFMA (tile 2u, tile 2u+1, ...), so consecutive MFMAs never
//     depend on each other (40-cycle dependent latency against 32-cycle issue);
//   * bias = the accumulators' initial value, ReLU behind the first blocks of the next tile pair.
#include "common.hpp"
#include "mlp_tile16.hpp"
#include "mlp_x4.hpp"

using namespace nerfhip;
using namespace nerflayout;
using namespace nerfmlp;
using namespace nerftile16;
using namespace nerfx4;

namespace {

// ---- stream layout ------------------------------------------------------------------------------
// Sections (blocks; every section starts on a slab): L0 64 (16 row tiles x 4 k-blocks of encoded position) |
// L1..L4 256 each (16 x 16) | L5 320 (16 x (16 hidden + 4 encoded)) | L6, L7 256 | feature 256 |
// views 144 + 16 padding (8 x (16 feature + 2 encoded direction)).
// Inside a section the blocks of row tiles 2u and 2u+1 alternate: block b -> pair u = b / (2 NKB),
// k-block kb = (b % (2 NKB)) / 2, row tile 2u + (b & 1).
__host__ __device__ inline int x4_stream_blocks(int use_viewdirs) { return use_viewdirs ? 2336 : 1920; }
__host__ __device__ inline int64_t x4_packed_floats(int use_viewdirs) {
    return (int64_t)x4_stream_blocks(use_viewdirs) * 256 + AUX16_FLOATS;
}

// flat-parameter index of stream element (block, lane, j), or -1 for zero padding
__host__ __device__ inline int64_t x4_stream_source(int use_viewdirs, int output_ch, int blk, int lane, int j) {
    const ParamOffsets po = param_offsets(use_viewdirs, output_ch);
    const int r16 = lane & 15, g = lane >> 4;
    int to = 0, kb = 0;
    auto locate = [&](int b, int nkb) { const int w = b % (2 * nkb); kb = w >> 1; to = 2 * (b / (2 * nkb)) + (w & 1); };
    auto enc_pts = [&](int k4) { const int q = 4 * k4 + j; return enc16_pts_feature(q >> 3, g, q & 7); };
    if (blk < 64) {
        locate(blk, 4);
        const int f = enc_pts(kb);
        return f < 0 ? -1 : po.pts_w[0] + (int64_t)(16 * to + r16) * IN_PTS + f;
    }
    blk -= 64;
    for (int l = 1; l < 8; ++l) {
        const int n = (l == 5) ? 320 : 256;
        if (blk < n) {
            locate(blk, l == 5 ? 20 : 16);
            const int64_t row = 16 * to + r16;
            const int ld = (l == 5) ? W + IN_PTS : W;
            if (kb < 16) return po.pts_w[l] + row * ld + (l == 5 ? IN_PTS : 0) + 16 * kb + 4 * g + j;
            const int f = enc_pts(kb - 16);
            return f < 0 ? -1 : po.pts_w[5] + row * ld + f;
        }
        blk -= n;
    }
    if (!use_viewdirs) return -1;
    if (blk < 256) {
        locate(blk, 16);
        return po.feat_w + (int64_t)(16 * to + r16) * W + 16 * kb + 4 * g + j;
    }
    blk -= 256;
    if (blk < 144) {
        locate(blk, 18);
        const int64_t row = 16 * to + r16;
        if (kb < 16) return po.views_w + row * po.views_in + 16 * kb + 4 * g + j;
        const int f = enc16_dirs_feature(g, 4 * (kb - 16) + j);
        return f < 0 ? -1 : po.views_w + row * po.views_in + W + f;
    }
    return -1;
}

// One layer: NOUT row tiles (16 features each, an even number) x NKB k-blocks (16 input features each).
// in(kb, j) = this lane's value of input feature 16 kb + 4 g + j for its point (a register of the input array, or of
// the encodings).  A step = the two blocks (row tiles 2u, 2u+1) of one k-block = eight MFMAs that alternate between
// the two accumulators; the blocks of step s+DEPTH are read from LDS before the MFMAs of step s issue.
// SAVE (training): a finished tile pair is also stored for the backward pass, one 16-byte store per tile and lane
// into the blocked layout (`srows` = the layer's first row group + this lane's offset g * GROUP_FLOATS + 4 * (point in
// tile): a lane's four registers of tile t are the four rows of row group 4 t + g); with MASK its ReLU mask bits are
// collected in `mb`, dword t / 4, bit 8 (t % 4) + i for register i of tile t (the lane's share of the layout the
// backward-data kernel reads; see save_mask()).
template <int NOUT, int NKB, bool RELU, bool SAVE, bool MASK, typename In>
__device__ __forceinline__ void layer_x4(StreamX4& st, int lane, const float*& cur, In in, f32x4 (&out)[16],
                                         const float* bias_g, float* srows, uint32_t (&mb)[4]) {
    auto load_bias = [&]<int U>() {
        out[2 * U] = *reinterpret_cast<const f32x4*>(bias_g + 32 * U);
        out[2 * U + 1] = *reinterpret_cast<const f32x4*>(bias_g + 32 * U + 16);
    };
    auto relu_pair = [&]<int U>() {
        static_for<2>([&]<int TT>() {
            constexpr int t = 2 * U + TT;
            if constexpr (RELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float y = max_keep_nan(out[t][i], 0.0f);
                    out[t][i] = y;
                    if constexpr (MASK) {   // y >= 0, so "positive" = "bits != 0"
                        const uint32_t one = __float_as_uint(y) < 1u ? __float_as_uint(y) : 1u;
                        mb[t >> 2] |= one << (8 * (t & 3) + i);
                    }
                }
            }
            if constexpr (SAVE) save16(srows + 4 * t * GROUP_FLOATS, out[t]);
        });
    };
    if constexpr (MASK) mb[0] = mb[1] = mb[2] = mb[3] = 0u;
    x4_layer<NOUT, NKB>(st, lane, cur, in, out, load_bias, relu_pair);
}

// dot product of NT register tiles with a head row in LDS (natural feature order), summed over the four lane groups
template <int NT>
__device__ __forceinline__ float head_x4(const f32x4 (&A)[16], const float* w_g) {
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(w_g + 16 * t);
        s0 = fmaf(A[t][0], w[0], s0);
        s1 = fmaf(A[t][1], w[1], s1);
        s0 = fmaf(A[t][2], w[2], s0);
        s1 = fmaf(A[t][3], w[3], s1);
    }
    float s = s0 + s1;
    s += __shfl_xor(s, 16, 64);
    return s + __shfl_xor(s, 32, 64);
}

template <bool VIEWDIRS, bool SAVE>
__global__ void __launch_bounds__(X4_THREADS, 2)
k_mlp_fwd_x4(int64_t n_pts, int pts_per_ray, const float* __restrict__ pts, const float* __restrict__ viewdirs,
             int output_ch, const float* __restrict__ packed, float* __restrict__ out, float* __restrict__ act) {
    extern __shared__ __attribute__((aligned(16))) float smem_x4[];
    float* ring = smem_x4;
    float* aux = smem_x4 + X4_RING_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    constexpr int UV = VIEWDIRS ? 1 : 0;
    StreamX4 st;
    st.ring = ring;
    st.mine = ring + wave * X4_WAVE_BLOCKS * 256;
    st.g = packed + wave * X4_WAVE_BLOCKS * 256 + lane * 4;
    st.next = 0;
    st.nslabs = x4_stream_blocks(UV) / X4_SLAB;
    {
        const float* gaux = packed + (int64_t)x4_stream_blocks(UV) * 256;
        for (int i = tid * 4; i < AUX16_FLOATS; i += X4_THREADS * 4)
            *reinterpret_cast<f32x4*>(aux + i) = *reinterpret_cast<const f32x4*>(gaux + i);
    }

    // ---- encodings: this lane's group evaluates its share of the sin/cos pairs of its point (mlp_tile16.hpp)
    const int64_t p = (int64_t)blockIdx.x * X4_WG_PTS + wave * 16 + (lane & 15);
    const bool valid = p < n_pts;
    const int64_t pl = valid ? p : n_pts - 1;
    float E[16], D[8];
    {
        const float x0 = pts[pl * 3 + 0], x1 = pts[pl * 3 + 1], x2 = pts[pl * 3 + 2];
        const int c0 = enc16_first_pair(g);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = c0 + i, k = c / 3, d = c - 3 * k;   // (groups 2, 3: pair 7 is overwritten below)
            const float arg = (d == 0 ? x0 : (d == 1 ? x1 : x2)) * __int_as_float((127 + k) << 23);
            sincosf(arg, &E[2 * i], &E[2 * i + 1]);
        }
        if (g == 2) { E[14] = x0; E[15] = x1; }
        if (g == 3) { E[14] = x2; E[15] = 0.0f; }
        if (VIEWDIRS) {
            const int64_t ray = pl / pts_per_ray;
            const float v0 = viewdirs[ray * 3 + 0], v1 = viewdirs[ray * 3 + 1], v2 = viewdirs[ray * 3 + 2];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int c = 3 * g + i, k = c / 3, d = c - 3 * k;
                const float arg = (d == 0 ? v0 : (d == 1 ? v1 : v2)) * __int_as_float((127 + k) << 23);
                sincosf(arg, &D[2 * i], &D[2 * i + 1]);
            }
            D[6] = g == 0 ? v0 : (g == 1 ? v2 : 0.0f);
            D[7] = g == 0 ? v1 : 0.0f;
        }
    }

    __syncthreads();   // tables visible; nothing in flight into the ring yet
    st.fetch(0);       // (started before the prologue instead, the first slab measured 0.3 % slower)
    const float* cur = nullptr;

    f32x4 P[16], Q[16];
    const float* bias = aux + AUX16_BIAS + 4 * g;
    const float* hw = aux + AUX16_HEAD;
    auto inE = [&]<int kb, int j>() -> float { return E[4 * kb + j]; };
    auto inP = [&]<int kb, int j>() -> float { return P[kb][j]; };
    auto inQ = [&]<int kb, int j>() -> float { return Q[kb][j]; };
    auto inPE = [&]<int kb, int j>() -> float { if constexpr (kb < 16) return P[kb][j]; else return E[4 * (kb - 16) + j]; };
    auto inPD = [&]<int kb, int j>() -> float { if constexpr (kb < 16) return P[kb][j]; else return D[4 * (kb - 16) + j]; };

    // ---- training: this workgroup's tile of the saved activations (mlp_layout.hpp), this lane's point inside it
    const int pw = wave * 16 + (lane & 15);
    float* const tile = SAVE ? act + (int64_t)blockIdx.x * act_rows(UV) * TILE_PTS : nullptr;
    float* const lane_rows = SAVE ? tile + g * GROUP_FLOATS + pw * 4 : nullptr;
    auto rows = [&](int row) { return lane_rows + (row / 4) * GROUP_FLOATS; };
    uint32_t mb[4] = {0u, 0u, 0u, 0u};
    // The backward-data kernel works on 32-point tiles: its lane (point, half h) reads one 16-byte word per masked layer
    // whose dword d, bit 16 (T % 2) + r stands for register r of its tile T = 2 d + ..., i.e. for feature row
    // 32 T + (r & 3) + 8 (r >> 2) + 4 h.  Lanes (point, g = h) and (point, g = h + 2) of this kernel hold those rows
    // between them, in nibbles that alternate: bit 8 (t % 4) + 4 (g >> 1) + i of dword t / 4.
    auto save_mask = [&](int m) {
        if constexpr (SAVE) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            u32x4 w;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t mine = mb[d] << (4 * (g >> 1));
                w[d] = mine | (uint32_t)__shfl_xor((int)mine, 32, 64);
            }
            if (g < 2) *reinterpret_cast<u32x4*>(tile + act_mask_offset(UV, m, pw >> 5, g * 32 + (pw & 31))) = w;
        }
    };
    if constexpr (SAVE) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int f = enc16_pts_feature(q >> 3, g, q & 7), row = ACT_ENC_PTS + (f < 0 ? 63 : f);
            tile[(row >> 2) * GROUP_FLOATS + pw * 4 + (row & 3)] = (valid && f >= 0) ? E[q] : 0.0f;
        }
        if (VIEWDIRS) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int f = enc16_dirs_feature(g, q);
                // the five slots without a feature zero the padding rows 27..31
                const int row = ACT_ENC_DIRS + (f >= 0 ? f : 27 + (g == 1 ? 0 : 2 * (g - 2) + 1 + (q - 6)));
                tile[(row >> 2) * GROUP_FLOATS + pw * 4 + (row & 3)] = (valid && f >= 0) ? D[q] : 0.0f;
            }
        }
    }

    layer_x4<16, 4, true, SAVE, SAVE>(st, lane, cur, inE, P, bias, rows(ACT_H0), mb);                     // L0: encoded position -> P
    save_mask(0);
    layer_x4<16, 16, true, SAVE, SAVE>(st, lane, cur, inP, Q, bias + 1 * 256, rows(ACT_H0 + 1 * 256), mb); // L1
    save_mask(1);
    layer_x4<16, 16, true, SAVE, SAVE>(st, lane, cur, inQ, P, bias + 2 * 256, rows(ACT_H0 + 2 * 256), mb); // L2
    save_mask(2);
    layer_x4<16, 16, true, SAVE, SAVE>(st, lane, cur, inP, Q, bias + 3 * 256, rows(ACT_H0 + 3 * 256), mb); // L3
    save_mask(3);
    layer_x4<16, 16, true, SAVE, SAVE>(st, lane, cur, inQ, P, bias + 4 * 256, rows(ACT_H0 + 4 * 256), mb); // L4
    save_mask(4);
    layer_x4<16, 20, true, SAVE, SAVE>(st, lane, cur, inPE, Q, bias + 5 * 256, rows(ACT_H0 + 5 * 256), mb); // L5: hidden ++ encoded position (skip)
    save_mask(5);
    layer_x4<16, 16, true, SAVE, SAVE>(st, lane, cur, inQ, P, bias + 6 * 256, rows(ACT_H0 + 6 * 256), mb); // L6
    save_mask(6);
    layer_x4<16, 16, true, SAVE, SAVE>(st, lane, cur, inP, Q, bias + 7 * 256, rows(ACT_H0 + 7 * 256), mb); // L7 -> Q
    save_mask(7);
    if constexpr (VIEWDIRS) {
        const float* hb = hw + 256 + 384;
        const float sigma = head_x4<16>(Q, hw + 4 * g) + hb[0];
        layer_x4<16, 16, false, SAVE, false>(st, lane, cur, inQ, P, bias + 8 * 256, rows(ACT_FEAT), mb);  // feature_linear (no activation) -> P
        layer_x4<8, 18, true, SAVE, SAVE>(st, lane, cur, inPD, Q, bias + 9 * 256, rows(ACT_VIEWS), mb);   // views layer: feature ++ encoded direction -> Q[0..7]
        save_mask(8);
        const float r = head_x4<8>(Q, hw + 256 + 4 * g) + hb[1];
        const float gg = head_x4<8>(Q, hw + 256 + 128 + 4 * g) + hb[2];
        const float b = head_x4<8>(Q, hw + 256 + 256 + 4 * g) + hb[3];
        if (valid && g == 0) {
            const f32x4 o = {r, gg, b, sigma};
            *reinterpret_cast<f32x4*>(out + p * 4) = o;
        }
    } else {
        for (int ch = 0; ch < output_ch; ++ch) {
            const float v = head_x4<16>(Q, hw + ch * 256 + 4 * g) + hw[8 * 256 + ch];
            if (valid && g == 0) out[p * output_ch + ch] = v;
        }
    }
}

// flat parameters -> x4 stream + fp32 aux tables
__global__ void __launch_bounds__(256) k_pack_x4(int use_viewdirs, int output_ch, const float* __restrict__ params,
                                                 float* __restrict__ out) {
    const int64_t stream = (int64_t)x4_stream_blocks(use_viewdirs) * 256;
    const int64_t total = stream + AUX16_FLOATS;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t src = i < stream ? x4_stream_source(use_viewdirs, output_ch, (int)(i >> 8), (int)((i >> 2) & 63), (int)(i & 3))
                                       : aux16_source(use_viewdirs, output_ch, (int)(i - stream));
        out[i] = src < 0 ? 0.0f : params[src];
    }
}

constexpr size_t X4_LDS_BYTES = (X4_RING_FLOATS + AUX16_FLOATS) * sizeof(float);

}  // namespace

extern "C" {

int64_t nerf_mlp_packed_x4_floats(int use_viewdirs) { return x4_packed_floats(use_viewdirs); }

int nerf_pack_weights_x4(int use_viewdirs, int output_ch, const float* params, float* packed_x4, void* stream) {
    NERF_REQUIRE(params && packed_x4, "nerf_pack_weights_x4: null pointer");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_pack_weights_x4: output_ch=%d not in 1..8", output_ch);
    hipLaunchKernelGGL(k_pack_x4, dim3(1024), dim3(256), 0, as_stream(stream), use_viewdirs, output_ch, params, packed_x4);
    return launched("nerf_pack_weights_x4");
}

int nerf_mlp_fwd_x4(int64_t n_pts, int pts_per_ray, const float* pts, const float* viewdirs, int use_viewdirs,
                    int output_ch, const float* packed_x4, float* out, float* act, void* stream) {
    NERF_REQUIRE(n_pts >= 0, "nerf_mlp_fwd_x4: n_pts=%lld", (long long)n_pts);
    if (n_pts == 0) return 0;
    NERF_REQUIRE(pts && packed_x4 && out, "nerf_mlp_fwd_x4: null pointer");
    NERF_REQUIRE(!use_viewdirs || (viewdirs && pts_per_ray >= 1), "nerf_mlp_fwd_x4: view branch needs viewdirs and pts_per_ray >= 1");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_mlp_fwd_x4: output_ch=%d not in 1..8", output_ch);
    NERF_REQUIRE((reinterpret_cast<uintptr_t>(packed_x4) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                 "nerf_mlp_fwd_x4: packed_x4 and out must be 16-byte aligned");
    const int64_t blocks = (n_pts + X4_WG_PTS - 1) / X4_WG_PTS;
    NERF_REQUIRE(blocks <= 0x7fffffff, "nerf_mlp_fwd_x4: n_pts=%lld too large for one launch", (long long)n_pts);
    auto go = [&](auto kernel, int ppr, const float* vd, int och) -> int {
        hipError_t e = lds_limit_once(reinterpret_cast<const void*>(kernel), X4_LDS_BYTES);
        if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_fwd_x4: LDS attribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(X4_THREADS), X4_LDS_BYTES, as_stream(stream), n_pts, ppr, pts, vd,
                           och, packed_x4, out, act);
        return launched("nerf_mlp_fwd_x4");
    };
    NERF_REQUIRE(!act || (reinterpret_cast<uintptr_t>(act) & 15) == 0, "nerf_mlp_fwd_x4: act must be 16-byte aligned");
    if (use_viewdirs) return act ? go(k_mlp_fwd_x4<true, true>, pts_per_ray, viewdirs, 4) : go(k_mlp_fwd_x4<true, false>, pts_per_ray, viewdirs, 4);
    return act ? go(k_mlp_fwd_x4<false, true>, 1, nullptr, output_ch) : go(k_mlp_fwd_x4<false, false>, 1, nullptr, output_ch);
}

}  // extern "C"
