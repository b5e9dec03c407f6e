// Backward of the fused MLP w.r.t. the pre-activations of every layer (stage 1 of the parameter-gradient pass, see
// mlp_bwd.hip) with fp32 accuracy on the bf16 matrix instruction: the mirror of mlp_fwd_bf16x3.hip.  Gradient tiles
// (32 features x 32 points, tile picture in mlp_x3.hpp) live in accumulators; a finished tile is masked with the ReLU
// bits the forward left behind, stored unsplit (fp32) to the workspace the weight-gradient stage reads, and split exactly
// into three bf16 pieces that are, as they stand, the B fragments of the layer below; W^T streams through the three-slab
// ring as triples (hi | mid | lo) split by the packer; six piece products per product, fp32 sums.  One wave per SIMD,
// four waves = one 128-point tile of the saved activations per workgroup.
#include "common.hpp"
#include "mlp_core.hpp"
#include "mlp_tile16.hpp"
#include "mlp_x3.hpp"

using namespace nerfhip;
using namespace nerflayout;
using namespace nerfmlp;
using namespace nerftile16;
using namespace nerfx3;

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- transposed stream ---------------------------------------------------------------------------
// Sections (triples; row tile by row tile, a tile's k-steps consecutively; every section is a whole number of slabs):
//   view branch: Wviews[:, :256]^T 64 (8 row tiles x 8 k-steps) | Wfeature^T 128 (8 x 16) |
//   then W7^T, W6^T, W5[:, 63:]^T, W4^T, ... W1^T, 128 each.
__host__ __device__ inline int x3b_stream_triples(int use_viewdirs) { return use_viewdirs ? 64 + 8 * 128 : 7 * 128; }
constexpr int x3_fwd_stream_slabs(int use_viewdirs) { return (use_viewdirs ? 1160 : 960) / SLABT; }

// flat-parameter index of stream element (triple, lane, j): M[32 to + r][k] with M = W^T, k = the output feature that
// k-slot j of lane half h stands for in k-step ks (mlp_x3.hpp)
__host__ __device__ inline int64_t x3b_stream_source(int use_viewdirs, int output_ch, int t, int lane, int j) {
    const ParamOffsets po = param_offsets(use_viewdirs, output_ch);
    const int r = lane & 31, h = lane >> 5;
    auto kfeat = [&](int ks) { return 32 * (ks >> 1) + kslot32(ks & 1, h, j); };
    if (use_viewdirs) {
        if (t < 64) return po.views_w + (int64_t)kfeat(t & 7) * po.views_in + 32 * (t >> 3) + r;
        t -= 64;
        if (t < 128) return po.feat_w + (int64_t)kfeat(t & 15) * W + 32 * (t >> 4) + r;
        t -= 128;
    }
    const int layer = 7 - t / 128;   // 7, 6, ..., 1: the section turns dY of this layer into dL/d(output of layer - 1)
    t %= 128;
    const int64_t row = kfeat(t & 15), col = 32 * (t >> 4) + r;
    if (layer == 5) return po.pts_w[5] + row * (W + IN_PTS) + IN_PTS + col;
    return po.pts_w[layer] + row * W + col;
}

// One backward-data layer: out = keep(start + M in), M = the section's W^T: NOUT row tiles x NKS k-steps.  keep (MASK)
// zeroes the entries whose forward activation was not positive: bit 16 (t % 2) + R of dword t / 2 of `mw` stands for
// value R of tile t.  ALPHA: start = d_sigma times the density head's weight column (the layer that also feeds that
// head), otherwise 0; the start value is written into the accumulator while the tile before multiplies.  A finished
// tile is stored to srow (+ row groups 8 t + 2 q, one 16-byte store per register quad) and, with STORE, split into the
// pieces of k-steps 2 t, 2 t + 1 of OUT -- two values per k-step behind the next tile's matrix instructions.
template <int NOUT, int NKS, bool MASK, bool ALPHA, bool STORE, typename InFrag>
__device__ __forceinline__ void layer_x3b(Stream3& st, int lane, int h, const char*& cur, InFrag in_frag,
                                          bf16x8 (&OUT)[3][16], const float* alpha_h, float d_sigma, const u32x4& mw,
                                          float* srow) {
    constexpr int NB = NOUT * NKS, DEPTH = 3;
    constexpr int PPS = 2;
    static_assert(NKS >= 8, "a tile's epilogue (two values per k-step) fits behind the next tile");
    bf16x8 abuf[DEPTH + 1][3];
    f32x16 acc[2];
    f32x4 sv = {0, 0, 0, 0};
    auto load_start = [&]<int TO>() {
        static_for<4>([&]<int Q>() {
            if constexpr (ALPHA) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(alpha_h + 32 * TO + 8 * Q);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[TO & 1][4 * Q + e] = w4[e] * d_sigma;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[TO & 1][4 * Q + e] = 0.0f;
            }
        });
    };
    auto piece = [&]<int TP, int R>() {
        float v = acc[TP & 1][R];
        if constexpr (MASK) {
            const bool on = ((mw[TP >> 1] >> (16 * (TP & 1) + R)) & 1u) != 0u;
            v = on ? v : 0.0f;
        }
        sv[R & 3] = v;
        if constexpr ((R & 3) == 3) *reinterpret_cast<f32x4*>(srow + (8 * TP + 2 * (R >> 2)) * GROUP_FLOATS) = sv;
        if constexpr (STORE) {
            __bf16 hi, mid, low;
            split3(v, hi, mid, low);
            OUT[0][2 * TP + (R >> 3)][R & 7] = hi; OUT[1][2 * TP + (R >> 3)][R & 7] = mid; OUT[2][2 * TP + (R >> 3)][R & 7] = low;
        }
    };

    load_start.template operator()<0>();
    static_for<NB + DEPTH>([&]<int I>() {
        if constexpr (I < NB) {
            if constexpr (I % SLABT == 0) cur = st.advance(lane);
#pragma unroll
            for (int s = 0; s < 3; ++s)
                abuf[I % (DEPTH + 1)][s] = *reinterpret_cast<const bf16x8*>(cur + (I % SLABT) * TRIPLE_BYTES + s * 1024);
        }
        if constexpr (I >= DEPTH) {
            constexpr int b = I - DEPTH;
            constexpr int to = b / NKS, ks = b % NKS;
            static_for<6>([&]<int P>() {
                constexpr int pw = PROD_W[P], px = PROD_X[P];
                acc[to & 1] = mfma32(abuf[b % (DEPTH + 1)][pw], in_frag.template operator()<ks, px>(), acc[to & 1]);
            });
            if constexpr (to > 0) {   // the previous row tile's epilogue behind this tile's MFMAs
                static_for<PPS>([&]<int Q>() {
                    constexpr int R = ks * PPS + Q;
                    if constexpr (R < 16) piece.template operator()<to - 1, R>();
                });
            }
            if constexpr (to + 1 < NOUT && ks == NKS - 3) load_start.template operator()<to + 1>();
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    static_for<16>([&]<int R>() { piece.template operator()<NOUT - 1, R>(); });
}

template <bool VIEWDIRS>
__global__ void __launch_bounds__(X3_THREADS, 1)
k_mlp_bwd_data_bf16x3(int64_t n_pts, int output_ch, const float* __restrict__ d_out, const float* __restrict__ act,
                      const char* __restrict__ packed_x3, const char* __restrict__ packed_bwd, float* __restrict__ scr) {
    extern __shared__ __attribute__((aligned(16))) char smem3b[];
    char* ring = smem3b;
    float* aux = reinterpret_cast<float*>(smem3b + RING3_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5;
    constexpr int UV = VIEWDIRS ? 1 : 0;
    {   // the fp32 tables behind the forward stream: head rows in natural feature order
        const float* gaux = reinterpret_cast<const float*>(packed_x3 + (int64_t)x3_fwd_stream_slabs(UV) * SLAB3_BYTES);
        for (int i = tid * 4; i < AUX16_FLOATS; i += X3_THREADS * 4)
            *reinterpret_cast<f32x4*>(aux + i) = *reinterpret_cast<const f32x4*>(gaux + i);
    }
    const float* const at = act + (int64_t)blockIdx.x * act_rows(UV) * TILE_PTS;
    float* const st_ = scr + (int64_t)blockIdx.x * scr_rows(UV) * TILE_PTS;
    const int pw = wave * WAVE_PTS + (lane & 31);             // point inside the 128-point tile
    const int64_t p = (int64_t)blockIdx.x * TILE_PTS + pw;
    const bool valid = p < n_pts;
    float* const lane_rows = st_ + h * GROUP_FLOATS + pw * 4;
    constexpr int SCR_HEAD = scr_head(UV), SCR_HEAD2 = SCR_HEAD + 32;

    Stream3 st;
    st.ring = ring;
    st.mine = ring + wave * WAVE_BLOCKS3 * 1024;
    st.g = packed_bwd + wave * WAVE_BLOCKS3 * 1024 + lane * 16;
    st.next = 0;
    st.nslabs = x3b_stream_triples(UV) / SLABT;
    __syncthreads();
    st.fetch(0);
#ifdef X3_HYBRID
    st.fetch_hybrid(1, lane);
#else
    st.fetch(1);
#endif
    const char* cur = nullptr;

    // the mask word of masked layer m of this lane, as the forward wrote it
    u32x4 mw = {0u, 0u, 0u, 0u};
    auto load_mask = [&](int m) { mw = *reinterpret_cast<const u32x4*>(at + act_mask_offset(UV, m, wave, lane)); };
    auto on = [&](int t, int R) { return ((mw[t >> 1] >> (16 * (t & 1) + R)) & 1u) != 0u; };
    auto rows = [&](int row) { return lane_rows + (row / 4) * GROUP_FLOATS; };

    bf16x8 X[3][16], Y[3][16];
    auto fromX = [&]<int ks, int s>() -> bf16x8 { return X[s][ks]; };
    auto fromY = [&]<int ks, int s>() -> bf16x8 { return Y[s][ks]; };
    const float* hw = aux + AUX16_HEAD;
    float d_sigma = 0.0f;
    // a head gradient quad (VALU): registers 4 q .. 4 q + 3 of tile t, already masked: stored, split into X
    auto put = [&]<int t, int q>(const f32x4& v, float* srow) {
        *reinterpret_cast<f32x4*>(srow + (8 * t + 2 * q) * GROUP_FLOATS) = v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            constexpr int dummy = 0; (void)dummy;
            __bf16 hi, mid, low;
            split3(v[i], hi, mid, low);
            const int R = 4 * q + i;
            X[0][2 * t + (R >> 3)][R & 7] = hi; X[1][2 * t + (R >> 3)][R & 7] = mid; X[2][2 * t + (R >> 3)][R & 7] = low;
        }
    };

    if constexpr (VIEWDIRS) {
        f32x4 gd = {0, 0, 0, 0};
        if (valid) gd = *reinterpret_cast<const f32x4*>(d_out + p * 4);
        d_sigma = gd[3];
        if (h == 0) {   // head operands of the weight-gradient stage: one row group each, zero padded
            const f32x4 ga = {gd[0], gd[1], gd[2], 0.0f}, gb = {gd[3], 0.0f, 0.0f, 0.0f};
            *reinterpret_cast<f32x4*>(st_ + (SCR_HEAD / 4) * GROUP_FLOATS + pw * 4) = ga;
            *reinterpret_cast<f32x4*>(st_ + (SCR_HEAD2 / 4) * GROUP_FLOATS + pw * 4) = gb;
        }
        // views layer: d(hidden) = Wrgb^T d_rgb, masked -> X k-steps 0..7, stored
        load_mask(8);
        float* sr = rows(SCR_VIEWS);
        const float* wrgb = hw + 256 + 4 * h;
        static_for<4>([&]<int t>() {
            static_for<4>([&]<int q>() {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wrgb + 32 * t + 8 * q);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(wrgb + 128 + 32 * t + 8 * q);
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(wrgb + 256 + 32 * t + 8 * q);
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float y = fmaf(w2[i], gd[2], fmaf(w1[i], gd[1], w0[i] * gd[0]));
                    v[i] = on(t, 4 * q + i) ? y : 0.0f;
                }
                put.template operator()<t, q>(v, sr);
            });
        });
        // d(feature) = Wviews[:, :256]^T dY_views -> Y (feature_linear has no activation)
        layer_x3b<8, 8, false, false, true>(st, lane, h, cur, fromX, Y, nullptr, d_sigma, mw, rows(SCR_FEAT));
        // d(h7) = Wfeature^T d(feature) + Walpha^T d_sigma, masked -> X
        load_mask(7);
        layer_x3b<8, 16, true, true, true>(st, lane, h, cur, fromY, X, hw + 4 * h, d_sigma, mw, rows(SCR_H0 + 7 * 256));
    } else {
        // d(h7) = Wout^T d_out, masked -> X, stored
        float gd[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) gd[c] = (valid && c < output_ch) ? d_out[p * output_ch + c] : 0.0f;
        if (h == 0) {
            const f32x4 ga = {gd[0], gd[1], gd[2], gd[3]}, gb = {gd[4], gd[5], gd[6], gd[7]};
            *reinterpret_cast<f32x4*>(st_ + (SCR_HEAD / 4) * GROUP_FLOATS + pw * 4) = ga;
            *reinterpret_cast<f32x4*>(st_ + (SCR_HEAD / 4 + 1) * GROUP_FLOATS + pw * 4) = gb;
        }
        load_mask(7);
        float* sr = rows(SCR_H0 + 7 * 256);
        static_for<8>([&]<int t>() {
            static_for<4>([&]<int q>() {
                f32x4 v = {0, 0, 0, 0};
#pragma unroll
                for (int c = 0; c < 8; ++c) {   // (gradients of the channels past output_ch are zero)
                    const f32x4 w = *reinterpret_cast<const f32x4*>(hw + c * 256 + 32 * t + 8 * q + 4 * h);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaf(w[i], gd[c], v[i]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = on(t, 4 * q + i) ? v[i] : 0.0f;
                put.template operator()<t, q>(v, sr);
            });
        });
    }

    // dY of pts layers 6 .. 0: X (layer l + 1) -> Y (layer l) -> X (layer l - 1) ...
    load_mask(6);
    layer_x3b<8, 16, true, false, true>(st, lane, h, cur, fromX, Y, nullptr, d_sigma, mw, rows(SCR_H0 + 6 * 256));
    load_mask(5);
    layer_x3b<8, 16, true, false, true>(st, lane, h, cur, fromY, X, nullptr, d_sigma, mw, rows(SCR_H0 + 5 * 256));
    load_mask(4);
    layer_x3b<8, 16, true, false, true>(st, lane, h, cur, fromX, Y, nullptr, d_sigma, mw, rows(SCR_H0 + 4 * 256));
    load_mask(3);
    layer_x3b<8, 16, true, false, true>(st, lane, h, cur, fromY, X, nullptr, d_sigma, mw, rows(SCR_H0 + 3 * 256));
    load_mask(2);
    layer_x3b<8, 16, true, false, true>(st, lane, h, cur, fromX, Y, nullptr, d_sigma, mw, rows(SCR_H0 + 2 * 256));
    load_mask(1);
    layer_x3b<8, 16, true, false, true>(st, lane, h, cur, fromY, X, nullptr, d_sigma, mw, rows(SCR_H0 + 1 * 256));
    load_mask(0);
    layer_x3b<8, 16, true, false, false>(st, lane, h, cur, fromX, Y, nullptr, d_sigma, mw, rows(SCR_H0));
}

// flat parameters -> transposed stream of triples
__global__ void __launch_bounds__(256) k_pack_x3b(int use_viewdirs, int output_ch, const float* __restrict__ params,
                                                  char* __restrict__ out) {
    const int64_t elems = (int64_t)x3b_stream_triples(use_viewdirs) * 512;   // one per (triple, lane, slot)
    unsigned short* s16 = reinterpret_cast<unsigned short*>(out);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < elems; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i >> 9;
        const int w = (int)(i & 511);
        __bf16 h, m, l;
        split3(params[x3b_stream_source(use_viewdirs, output_ch, (int)t, w >> 3, w & 7)], h, m, l);
        unsigned short* dst = s16 + t * (TRIPLE_BYTES / 2) + w;
        dst[0] = __builtin_bit_cast(unsigned short, h);
        dst[512] = __builtin_bit_cast(unsigned short, m);
        dst[1024] = __builtin_bit_cast(unsigned short, l);
    }
}

constexpr size_t X3B_LDS_BYTES = RING3_BYTES + AUX16_FLOATS * sizeof(float);

}  // namespace

namespace nerfx3 {

int launch_bwd_data_x3(int64_t n_pts, int use_viewdirs, int output_ch, const float* d_out, const float* act,
                       const void* packed_x3, const void* packed_bwd_x3, float* scratch, hipStream_t stream) {
    const int64_t blocks = padded_points(n_pts) / TILE_PTS;
    auto go = [&](auto kernel, int och) -> int {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)X3B_LDS_BYTES);
        if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_bwd_bf16x3: LDS attribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(X3_THREADS), X3B_LDS_BYTES, stream, n_pts, och, d_out, act,
                           reinterpret_cast<const char*>(packed_x3), reinterpret_cast<const char*>(packed_bwd_x3), scratch);
        return launched("nerf_mlp_bwd_bf16x3 (data)");
    };
    return use_viewdirs ? go(k_mlp_bwd_data_bf16x3<true>, 4) : go(k_mlp_bwd_data_bf16x3<false>, output_ch);
}

}  // namespace nerfx3

extern "C" {

int64_t nerf_mlp_packed_bwd_bf16x3_bytes(int use_viewdirs) { return (int64_t)x3b_stream_triples(use_viewdirs) * TRIPLE_BYTES; }

int nerf_pack_weights_bwd_bf16x3(int use_viewdirs, int output_ch, const float* params, void* packed, void* stream) {
    NERF_REQUIRE(params && packed, "nerf_pack_weights_bwd_bf16x3: null pointer");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_pack_weights_bwd_bf16x3: output_ch=%d not in 1..8", output_ch);
    hipLaunchKernelGGL(k_pack_x3b, dim3(1024), dim3(256), 0, as_stream(stream), use_viewdirs, output_ch, params,
                       reinterpret_cast<char*>(packed));
    return launched("nerf_pack_weights_bwd_bf16x3");
}

}  // extern "C"
