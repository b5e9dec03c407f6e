// Backward of the fused MLP w.r.t. the pre-activations of every layer (stage 1 of the parameter-gradient pass, see
// mlp_bwd.hip) with fp32 accuracy on the bf16 matrix instruction: the mirror of mlp_fwd_bf16x3.hip.  Gradient tiles
// live in accumulators; a finished tile is masked with the ReLU bits the forward left behind, stored unsplit (fp32) to
// the workspace the weight-gradient GEMM reads, and split exactly into three bf16 pieces that are, as they stand, the
// B fragments of the layer below; W^T streams through the three-slab ring as triples (hi | mid | lo) split by the
// packer; six piece products per product, fp32 sums.  One wave per SIMD, four waves = one 128-point tile of the saved
// activations per workgroup.
#include "common.hpp"
#include "mlp_core.hpp"
#include "mlp_tile16.hpp"
#include "mlp_bf16_layout.hpp"
#include "mlp_x3.hpp"
#include "mlp_bwd16_layout.hpp"

using namespace nerfhip;
using namespace nerflayout;
using namespace nerfmlp;
using namespace nerftile16;
using namespace nerfbf16;
using namespace nerfx3;
using namespace nerfbwd16;

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// One backward-data layer: out = keep(start + M in), M = the section's W^T.  keep (MASK) zeroes the entries whose forward
// activation was not positive: bit 8 (t % 4) + 4 (g >> 1) + r of dword t / 4 of mw[ct] stands for value r of tile t
// (mlp_fwd_bf16x3.hip, save_mask).  ALPHA: start = d_sigma times the density head's weight column (the layer that also
// feeds that head), otherwise 0; the start value is the C operand of a tile's first MFMAs.  A finished tile is stored
// to srow[ct] (+ 4 t row groups) and, with STORE, split into the pieces of fragment t / 2 of OUT.
template <int NOUT, int NKS, bool MASK, bool ALPHA, bool STORE, typename InFrag>
__device__ __forceinline__ void layer_x3b(Stream3& st, int lane, int g, const char*& cur, InFrag in_frag,
                                          bf16x8 (&OUT)[3][NCT][8], const float* alpha_g, const float (&d_sigma)[NCT],
                                          const u32x4 (&mw)[NCT], float* (&srow)[NCT]) {
    constexpr int NB = NOUT * NKS, DEPTH = 3, NPIECE = 4 * NCT;
    constexpr int PPS = (NPIECE + (NKS < 8 ? NKS : 8) - 1) / (NKS < 8 ? NKS : 8);   // epilogue pieces per k-step
    bf16x8 abuf[DEPTH + 1][3];
    f32x4 acc[2][NCT];
    f32x4 wnext = {0, 0, 0, 0};                   // ALPHA: the density head's weights of the next tile's rows
    f32x4 sv = {0, 0, 0, 0};
    const int shift = 4 * (g >> 1);
    auto load_start = [&]<int TO>() { if constexpr (ALPHA) wnext = *reinterpret_cast<const f32x4*>(alpha_g + 16 * TO); };
    auto piece = [&]<int TP, int K>() {           // K = 4 * column tile + register: one value of the tile per piece
        constexpr int ct = K >> 2, r = K & 3;
        float v = acc[TP & 1][ct][r];
        if constexpr (MASK) {
            const bool on = ((mw[ct][TP >> 2] >> (8 * (TP & 3) + shift + r)) & 1u) != 0u;
            v = on ? v : 0.0f;
        }
        sv[r] = v;
        if constexpr (r == 3) *reinterpret_cast<f32x4*>(srow[ct] + 4 * TP * GROUP_FLOATS) = sv;
        if constexpr (STORE) {
            constexpr int f = TP >> 1, slot = 4 * (TP & 1) + r;
            __bf16 h, m, l;
            split3(v, h, m, l);
            OUT[0][ct][f][slot] = h; OUT[1][ct][f][slot] = m; OUT[2][ct][f][slot] = l;
        }
    };

    load_start.template operator()<0>();
    static_for<NB + DEPTH>([&]<int I>() {
        if constexpr (I < NB) {
            if constexpr (I % SLABT == 0) {
                cur = st.template advance<(I > 0) ? x3_stores_since_open(I, NKS) : 0>(lane);
            }
#pragma unroll
            for (int s = 0; s < 3; ++s)
                abuf[I % (DEPTH + 1)][s] = *reinterpret_cast<const bf16x8*>(cur + (I % SLABT) * TRIPLE_BYTES + s * 1024);
        }
        if constexpr (I >= DEPTH) {
            constexpr int b = I - DEPTH;
            constexpr int to = b / NKS, ks = b % NKS;
            static_for<6>([&]<int P>() {
                constexpr int pw = PROD_W[P], px = PROD_X[P];
                const bf16x8 a = abuf[b % (DEPTH + 1)][pw];
                static_for<NCT>([&]<int CT>() {
                    const bf16x8 x = in_frag.template operator()<CT, ks, px>();
                    if constexpr (ks == 0 && P == 0) {
                        f32x4 c0 = {0, 0, 0, 0};
                        if constexpr (ALPHA) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) c0[i] = wnext[i] * d_sigma[CT];
                        }
                        acc[to & 1][CT] = mfma16(a, x, c0);
                    } else {
                        acc[to & 1][CT] = mfma16(a, x, acc[to & 1][CT]);
                    }
                });
            });
            if constexpr (to > 0) {   // the previous row tile's epilogue, piece by piece, behind this tile's MFMAs
                static_for<PPS>([&]<int Q>() {
                    constexpr int K = ks * PPS + Q;
                    if constexpr (K < NPIECE) piece.template operator()<to - 1, K>();
                });
            }
            if constexpr (to + 1 < NOUT && ks == (NKS >= 4 ? NKS - 3 : 0)) load_start.template operator()<to + 1>();
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    static_for<NPIECE>([&]<int K>() { piece.template operator()<NOUT - 1, K>(); });
}

// one 128-point tile (tile_index) by one workgroup
template <bool VIEWDIRS>
__device__ __forceinline__ void bwd_tile_x3(int64_t tile_index, int64_t n_pts, int output_ch, const float* __restrict__ d_out,
                                            const float* __restrict__ act, const char* __restrict__ packed_x3,
                                            const char* __restrict__ packed_bwd, float* __restrict__ scr) {
    extern __shared__ __attribute__((aligned(16))) char smem3b[];
    char* ring = smem3b;
    float* aux = reinterpret_cast<float*>(smem3b + RING3_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4;
    constexpr int UV = VIEWDIRS ? 1 : 0;
    {   // the fp32 tables behind the forward stream: head rows in natural feature order
        const float* gaux = reinterpret_cast<const float*>(packed_x3 + (int64_t)(bf16_stream_blocks(UV) / SLABT) * SLAB3_BYTES);
        for (int i = tid * 4; i < AUX16_FLOATS; i += X3_THREADS * 4)
            *reinterpret_cast<f32x4*>(aux + i) = *reinterpret_cast<const f32x4*>(gaux + i);
    }
    const float* const at = act + tile_index * act_rows(UV) * TILE_PTS;
    float* const st_ = scr + tile_index * scr_rows(UV) * TILE_PTS;
    int pw[NCT];
    bool valid[NCT];
    int64_t p[NCT];
    float* lane_rows[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        pw[ct] = wave * WAVE_PTS + ct * 16 + (lane & 15);      // point inside the 128-point tile
        p[ct] = tile_index * TILE_PTS + pw[ct];
        valid[ct] = p[ct] < n_pts;
        lane_rows[ct] = st_ + g * GROUP_FLOATS + pw[ct] * 4;
    }
    constexpr int SCR_HEAD = scr_head(UV), SCR_HEAD2 = SCR_HEAD + 32;

    Stream3 st;
    st.ring = ring;
    st.mine = ring + wave * WAVE_BLOCKS3 * 1024;
    st.g = packed_bwd + wave * WAVE_BLOCKS3 * 1024 + lane * 16;
    st.next = 0;
    st.nslabs = bwd16_stream_blocks(UV) / SLABT;
    __syncthreads();
    st.fetch(0);
    st.fetch(1);
    const char* cur = nullptr;

    // the mask word of masked layer m that holds this lane's bits (written by lane (point, g & 1) of the forward)
    u32x4 mw[NCT];
    auto load_masks = [&](int m) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
            mw[ct] = *reinterpret_cast<const u32x4*>(at + act_mask_offset(UV, m, pw[ct] >> 5, (g & 1) * 32 + (pw[ct] & 31)));
    };
    const int shift = 4 * (g >> 1);
    auto on = [&](int ct, int t, int i) { return ((mw[ct][t >> 2] >> (8 * (t & 3) + shift + i)) & 1u) != 0u; };
    float* sr[NCT];
    auto rows = [&](int row) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) sr[ct] = lane_rows[ct] + (row / 4) * GROUP_FLOATS;
    };

    bf16x8 X[3][NCT][8], Y[3][NCT][8];
    auto fromX = [&]<int ct, int ks, int s>() -> bf16x8 { return X[s][ct][ks]; };
    auto fromY = [&]<int ct, int ks, int s>() -> bf16x8 { return Y[s][ct][ks]; };
    const float* hw = aux + AUX16_HEAD;
    float d_sigma[NCT] = {0.0f, 0.0f};
    // a head gradient tile (VALU): masked, stored, split into fragment t / 2 of X
    auto put = [&]<int ct, int t>(const f32x4& v) {
        *reinterpret_cast<f32x4*>(sr[ct] + 4 * t * GROUP_FLOATS) = v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __bf16 h, m, l;
            split3(v[i], h, m, l);
            X[0][ct][t >> 1][4 * (t & 1) + i] = h; X[1][ct][t >> 1][4 * (t & 1) + i] = m; X[2][ct][t >> 1][4 * (t & 1) + i] = l;
        }
    };

    if constexpr (VIEWDIRS) {
        f32x4 gd[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            gd[ct] = f32x4{0, 0, 0, 0};
            if (valid[ct]) gd[ct] = *reinterpret_cast<const f32x4*>(d_out + p[ct] * 4);
            d_sigma[ct] = gd[ct][3];
            if (g == 0) {   // head operands of the weight-gradient GEMM: one row group each, zero padded
                const f32x4 ga = {gd[ct][0], gd[ct][1], gd[ct][2], 0.0f}, gb = {gd[ct][3], 0.0f, 0.0f, 0.0f};
                *reinterpret_cast<f32x4*>(st_ + (SCR_HEAD / 4) * GROUP_FLOATS + pw[ct] * 4) = ga;
                *reinterpret_cast<f32x4*>(st_ + (SCR_HEAD2 / 4) * GROUP_FLOATS + pw[ct] * 4) = gb;
            }
        }
        // views layer: d(hidden) = Wrgb^T d_rgb, masked -> X fragments 0..3, stored
        load_masks(8);
        rows(SCR_VIEWS);
        const float* wrgb = hw + 256 + 4 * g;
        static_for<8>([&]<int t>() {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wrgb + 16 * t);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wrgb + 128 + 16 * t);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(wrgb + 256 + 16 * t);
            static_for<NCT>([&]<int ct>() {
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float y = fmaf(w2[i], gd[ct][2], fmaf(w1[i], gd[ct][1], w0[i] * gd[ct][0]));
                    v[i] = on(ct, t, i) ? y : 0.0f;
                }
                put.template operator()<ct, t>(v);
            });
        });
        // d(feature) = Wviews[:, :256]^T dY_views -> Y (feature_linear has no activation)
        rows(SCR_FEAT);
        layer_x3b<16, 4, false, false, true>(st, lane, g, cur, fromX, Y, nullptr, d_sigma, mw, sr);
        // d(h7) = Wfeature^T d(feature) + Walpha^T d_sigma, masked -> X
        load_masks(7);
        rows(SCR_H0 + 7 * 256);
        layer_x3b<16, 8, true, true, true>(st, lane, g, cur, fromY, X, hw + 4 * g, d_sigma, mw, sr);
    } else {
        // d(h7) = Wout^T d_out, masked -> X, stored
        float gd[NCT][8];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
            for (int c = 0; c < 8; ++c) gd[ct][c] = (valid[ct] && c < output_ch) ? d_out[p[ct] * output_ch + c] : 0.0f;
            if (g == 0) {
                const f32x4 ga = {gd[ct][0], gd[ct][1], gd[ct][2], gd[ct][3]}, gb = {gd[ct][4], gd[ct][5], gd[ct][6], gd[ct][7]};
                *reinterpret_cast<f32x4*>(st_ + (SCR_HEAD / 4) * GROUP_FLOATS + pw[ct] * 4) = ga;
                *reinterpret_cast<f32x4*>(st_ + (SCR_HEAD / 4 + 1) * GROUP_FLOATS + pw[ct] * 4) = gb;
            }
        }
        load_masks(7);
        rows(SCR_H0 + 7 * 256);
        static_for<16>([&]<int t>() {
            static_for<NCT>([&]<int ct>() {
                f32x4 v = {0, 0, 0, 0};
#pragma unroll
                for (int c = 0; c < 8; ++c) {   // (gradients of the channels past output_ch are zero)
                    const f32x4 w = *reinterpret_cast<const f32x4*>(hw + c * 256 + 16 * t + 4 * g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaf(w[i], gd[ct][c], v[i]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = on(ct, t, i) ? v[i] : 0.0f;
                put.template operator()<ct, t>(v);
            });
        });
    }

    // dY of pts layers 6 .. 0: X (layer l + 1) -> Y (layer l) -> X (layer l - 1) ...
    load_masks(6); rows(SCR_H0 + 6 * 256);
    layer_x3b<16, 8, true, false, true>(st, lane, g, cur, fromX, Y, nullptr, d_sigma, mw, sr);
    load_masks(5); rows(SCR_H0 + 5 * 256);
    layer_x3b<16, 8, true, false, true>(st, lane, g, cur, fromY, X, nullptr, d_sigma, mw, sr);
    load_masks(4); rows(SCR_H0 + 4 * 256);
    layer_x3b<16, 8, true, false, true>(st, lane, g, cur, fromX, Y, nullptr, d_sigma, mw, sr);
    load_masks(3); rows(SCR_H0 + 3 * 256);
    layer_x3b<16, 8, true, false, true>(st, lane, g, cur, fromY, X, nullptr, d_sigma, mw, sr);
    load_masks(2); rows(SCR_H0 + 2 * 256);
    layer_x3b<16, 8, true, false, true>(st, lane, g, cur, fromX, Y, nullptr, d_sigma, mw, sr);
    load_masks(1); rows(SCR_H0 + 1 * 256);
    layer_x3b<16, 8, true, false, true>(st, lane, g, cur, fromY, X, nullptr, d_sigma, mw, sr);
    load_masks(0); rows(SCR_H0);
    layer_x3b<16, 8, true, false, false>(st, lane, g, cur, fromX, Y, nullptr, d_sigma, mw, sr);
}

// `marked` null: one workgroup per tile.  Otherwise (the second launch behind the two-piece kernel) one word per tile, only
// tiles whose word is not zero are computed, and a small grid walks the words.
template <bool VIEWDIRS>
__global__ void __launch_bounds__(X3_THREADS, 1)
k_mlp_bwd_data_bf16x3(int64_t n_pts, int output_ch, const float* __restrict__ d_out, const float* __restrict__ act,
                      const char* __restrict__ packed_x3, const char* __restrict__ packed_bwd, float* __restrict__ scr,
                      const uint32_t* __restrict__ marked) {
    const int64_t n_tiles = padded_points(n_pts) / TILE_PTS;
    // workgroup b looks at the tiles b, b + G, b + 2 G, ... (G workgroups), X3_THREADS of them at a time, one word per
    // thread, and works through the marked ones: neighbouring tiles go to different workgroups.  (Without `marked` the
    // grid is the number of tiles and thread 0's tile, b, is the only one in range.)
    __shared__ unsigned long long found[X3_WAVES];
    const int tid = threadIdx.x;
    for (int64_t base = blockIdx.x; base < n_tiles; base += (int64_t)gridDim.x * X3_THREADS) {
        const int64_t mine = base + (int64_t)tid * gridDim.x;
        const unsigned long long b = __ballot(mine < n_tiles && (marked == nullptr || marked[mine] != 0u));
        if ((tid & 63) == 0) found[tid >> 6] = b;
        __syncthreads();
        for (int w = 0; w < X3_WAVES; ++w) {
            unsigned long long todo = found[w];
            while (todo != 0ull) {
                const int i = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                const int64_t t = base + (int64_t)(64 * w + i) * gridDim.x;
                bwd_tile_x3<VIEWDIRS>(t, n_pts, output_ch, d_out, act, packed_x3, packed_bwd, scr);
                __syncthreads();      // (the next tile of this workgroup refills the LDS tables)
            }
        }
        __syncthreads();
    }
}

// flat parameters -> transposed stream of triples
__global__ void __launch_bounds__(256) k_pack_x3b(int use_viewdirs, int output_ch, const float* __restrict__ params,
                                                  char* __restrict__ out) {
    const int64_t elems = (int64_t)bwd16_stream_blocks(use_viewdirs) * 512;   // one per (triple, lane, slot)
    unsigned short* s16 = reinterpret_cast<unsigned short*>(out);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < elems; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i >> 9;
        const int w = (int)(i & 511);
        __bf16 h, m, l;
        split3(params[bwd16_stream_source(use_viewdirs, output_ch, (int)t, w >> 3, w & 7)], h, m, l);
        unsigned short* dst = s16 + t * (TRIPLE_BYTES / 2) + w;
        dst[0] = __builtin_bit_cast(unsigned short, h);
        dst[512] = __builtin_bit_cast(unsigned short, m);
        dst[1024] = __builtin_bit_cast(unsigned short, l);
    }
}

constexpr size_t X3B_LDS_BYTES = RING3_BYTES + AUX16_FLOATS * sizeof(float);

}  // namespace

namespace nerfx3 {

// marked: nullptr = every 128-point tile; else one word per tile, and only the tiles whose word is not 0 are computed
int launch_bwd_data_x3(int64_t n_pts, int use_viewdirs, int output_ch, const float* d_out, const float* act,
                       const void* packed_x3, const void* packed_bwd_x3, float* scratch, const uint32_t* marked,
                       hipStream_t stream) {
    const int64_t blocks = padded_points(n_pts) / TILE_PTS;
    auto go = [&](auto kernel, int och) -> int {
        hipError_t e = lds_limit_once(reinterpret_cast<const void*>(kernel), X3B_LDS_BYTES);
        if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_bwd_bf16x3: LDS attribute: %s", hipGetErrorString(e));
        const int64_t grid = (marked != nullptr && blocks > compute_units()) ? compute_units() : blocks;
        hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(X3_THREADS), X3B_LDS_BYTES, stream, n_pts, och, d_out, act,
                           reinterpret_cast<const char*>(packed_x3), reinterpret_cast<const char*>(packed_bwd_x3), scratch, marked);
        return launched("nerf_mlp_bwd_bf16x3 (data)");
    };
    return use_viewdirs ? go(k_mlp_bwd_data_bf16x3<true>, 4) : go(k_mlp_bwd_data_bf16x3<false>, output_ch);
}

}  // namespace nerfx3

extern "C" {

int64_t nerf_mlp_packed_bwd_bf16x3_bytes(int use_viewdirs) { return (int64_t)bwd16_stream_blocks(use_viewdirs) * TRIPLE_BYTES; }

int nerf_pack_weights_bwd_bf16x3(int use_viewdirs, int output_ch, const float* params, void* packed, void* stream) {
    NERF_REQUIRE(params && packed, "nerf_pack_weights_bwd_bf16x3: null pointer");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_pack_weights_bwd_bf16x3: output_ch=%d not in 1..8", output_ch);
    hipLaunchKernelGGL(k_pack_x3b, dim3(1024), dim3(256), 0, as_stream(stream), use_viewdirs, output_ch, params,
                       reinterpret_cast<char*>(packed));
    return launched("nerf_pack_weights_bwd_bf16x3");
}

}  // extern "C"
