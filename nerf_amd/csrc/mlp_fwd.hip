// NeRF MLP forward on MFMA (fp32 in, fp32 accumulate) for inputs that are ALREADY encoded: NeRF.forward(x) as upstream
// calls it (run_nerf_helpers.py:109-133).  The render path does not come through here (its kernels fuse the encoding).
//
// One workgroup = 4 wavefronts = 128 points; each wavefront carries 32 points through all layers.
// The activations of a wavefront never leave its registers: the 32x32 accumulator tiles of
// v_mfma_f32_32x32x2_f32 (column = point on the lane, row = feature in the register) are used
// directly as the B operands of the next layer (see mlp_layout.hpp).  Only the weights move: the
// packed stream is pulled through a two-slab LDS ring (16 KiB per slab) with global_load_lds, each
// wave fetching a quarter of the next slab while all four consume the current one, so per
// workgroup the whole 2.3 MiB stream is read once from L2 and every A operand is one
// ds_read_b128 per four MFMAs.  The bias is the accumulators' initial value (loaded from LDS straight into
// their registers), ReLU runs in place behind the next tile's MFMAs; the 1- and 3-row heads (alpha, rgb)
// are VALU dot products on the register-resident activations.
//
// Bound: MFMA.  9280 MFMAs (4096 FLOP each) per 32 points = 1.188 MFLOP per point, of which
// 1.187 MFLOP is algorithmic (the rest is zero padding of the 63/27-wide inputs).
#include "common.hpp"
#include "mlp_core.hpp"

using namespace nerfhip;
using namespace nerflayout;
using namespace nerfmlp;

namespace {

// Save NT activation tiles into the blocked layout (mlp_layout.hpp).  `rows` points at the tiles' first row
// group inside this workgroup's tile (wave-uniform), `lane_off` = half * GROUP_FLOATS + 4 * (point in tile):
// registers 4i..4i+3 of a tile are rows 8i + 4*half + 0..3, i.e. one row group = one 16-byte store.
// Padding points of the last tile carry the activations of the last real point: finite values that the
// backward multiplies by exactly-zero gradients.
template <int T, int I>   // register quad I of tile T: one row group, one 16-byte store
__device__ __forceinline__ void save_quad(const f32x16& a, float* rows, uint32_t lane_off) {
    const f32x4 v = {a[4 * I], a[4 * I + 1], a[4 * I + 2], a[4 * I + 3]};
    *reinterpret_cast<f32x4*>(rows + (8 * T + 2 * I) * GROUP_FLOATS + lane_off) = v;
}
template <int NT>
__device__ __forceinline__ void save_tiles(const f32x16 (&A)[8], float* rows, uint32_t lane_off) {
    static_for<NT * 4>([&]<int K>() { save_quad<K / 4, K % 4>(A[K / 4], rows, lane_off); });
}

// dot product of the register-resident activations (NT tiles) with a head row stored as [tile][half][16]
template <int NT>
__device__ __forceinline__ float head_dot(const f32x16 (&A)[8], const float* w_lds, int h) {
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float* wp = w_lds + t * 32 + h * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wp + 4 * i);
            s0 = fmaf(A[t][4 * i + 0], w[0], s0);
            s1 = fmaf(A[t][4 * i + 1], w[1], s1);
            s0 = fmaf(A[t][4 * i + 2], w[2], s0);
            s1 = fmaf(A[t][4 * i + 3], w[3], s1);
        }
    }
    const float s = s0 + s1;
    return s + __shfl_xor(s, 32, 64);  // the two lane halves hold disjoint feature sets of one point
}

template <bool VIEWDIRS, bool SAVE>
__global__ void __launch_bounds__(THREADS, 1)
k_mlp_fwd(int64_t n_pts, int pts_per_ray, const float* __restrict__ pts, const float* __restrict__ viewdirs,
          int output_ch, const float* __restrict__ packed, float* __restrict__ out, float* __restrict__ act) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ring = smem;
    float* aux = smem + RING_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int64_t p = (int64_t)blockIdx.x * TILE_PTS + wave * 32 + (lane & 31);
    const bool valid = p < n_pts;
    const int64_t pl = valid ? p : n_pts - 1;
    constexpr bool save = SAVE;
    // this workgroup's tile of the saved activations, and this lane's point inside it
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    float* const tile = save ? act + (int64_t)blockIdx.x * act_rows(VIEWDIRS ? 1 : 0) * TILE_PTS : nullptr;
    const int pw = wave * 32 + (lane & 31);

    // aux tables -> LDS (plain loads, before the ring starts)
    const int stream_blocks = fwd_stream_blocks(VIEWDIRS ? 1 : 0);
    {
        const float* gaux = packed + (int64_t)stream_blocks * BLOCK_FLOATS;
        for (int i = tid * 4; i < AUX_FLOATS; i += THREADS * 4)
            *reinterpret_cast<f32x4*>(aux + i) = *reinterpret_cast<const f32x4*>(gaux + i);
    }

    // ---- positional encoding into register tiles (slot maps: mlp_layout.hpp)
    f32x16 E[2], D;
    {
        // pts is the reference's pre-encoded input [n_pts, 63 (+27)]; pts_per_ray carries its row length
        const float* row = pts + pl * pts_per_ray;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int f0 = enc_pts_feature(s, 0), f1 = enc_pts_feature(s, 1);
            const float v = (h && f1 < 0) ? 0.0f : row[h ? f1 : f0];
            if (s < 16) E[0][s] = v; else E[1][s - 16] = v;
        }
        if (VIEWDIRS) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int f0 = enc_dirs_feature(s, 0), f1 = enc_dirs_feature(s, 1);
                const int f = h ? f1 : f0;
                D[s] = (s >= 14 || (h && f1 < 0)) ? 0.0f : row[IN_PTS + (f < 0 ? 0 : f)];
            }
        }
    }
    {
        if constexpr (save) {
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                // row of slot s: the two halves' features differ by a constant per slot except for the padding slot
                const int f0 = enc_pts_feature(s, 0), f1 = enc_pts_feature(s, 1);
                const int f = h ? (f1 < 0 ? 63 : f1) : f0;
                const float v = (valid && !(h && f1 < 0)) ? (s < 16 ? E[0][s] : E[1][s - 16]) : 0.0f;
                tile[((ACT_ENC_PTS + f) >> 2) * GROUP_FLOATS + pw * 4 + (f & 3)] = v;
            }
            if (VIEWDIRS) {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    // slots 14,15 and slot 1 of the upper half carry no feature: they zero the padding rows 27..31
                    const int f0 = s < 14 ? enc_dirs_feature(s, 0) : 27 + (s - 14), f1 = enc_dirs_feature(s, 1);
                    const int g1 = s == 1 ? 29 : (s == 14 ? 30 : (s == 15 ? 31 : f1));
                    const int f = h ? g1 : f0;
                    const bool real = h ? (f1 >= 0) : (s < 14);
                    tile[((ACT_ENC_DIRS + f) >> 2) * GROUP_FLOATS + pw * 4 + (f & 3)] = (valid && real) ? D[s] : 0.0f;
                }
            }
        }
    }

    Stream st;
    st.ring = ring;
    st.mine = ring + wave * WAVE_BLOCKS * BLOCK_FLOATS;
    st.g = packed + wave * WAVE_BLOCKS * BLOCK_FLOATS + lane * 4;
    st.next = 0;
    st.nslabs = (stream_blocks + SLAB_BLOCKS - 1) / SLAB_BLOCKS;
    __syncthreads();  // aux visible; no glds in flight yet
    st.fetch(0);

    // Activations ping-pong between two tile arrays: a layer reads one and accumulates into the other.
    f32x16 P[8], Q[8];
    const uint32_t lane_off = (uint32_t)(h * GROUP_FLOATS + pw * 4);
    const float* bias_h = aux + AUX_BIAS + h * 16;
    auto inE = [&]<int ti>() -> const f32x16& { return E[ti]; };
    auto inP = [&]<int ti>() -> const f32x16& { return P[ti]; };
    auto inQ = [&]<int ti>() -> const f32x16& { return Q[ti]; };
    auto rows = [&](int row) { return tile + (row / 4) * GROUP_FLOATS; };

    // Training: a layer's output is stored for the backward pass while the NEXT layer, which only reads it,
    // is multiplying: one 16-byte store behind every other weight block of the second half of each output tile.
    float* srows = nullptr;   // first row group of the array being stored
    auto saveP = [&]<int TO, int BI>() {
        if constexpr (save && BI >= 16 && BI < 24 && (BI & 1) == 0) save_quad<TO, (BI - 16) / 2>(P[TO], srows, lane_off);
    };
    auto saveQ = [&]<int TO, int BI>() {
        if constexpr (save && BI >= 16 && BI < 24 && (BI & 1) == 0) save_quad<TO, (BI - 16) / 2>(Q[TO], srows, lane_off);
    };

    // ... and every ReLU layer leaves its mask bits (16 bytes per lane) behind the tile's rows.
    uint32_t mb[4] = {0u, 0u, 0u, 0u};
    auto put_mask = [&](int m) {
        if constexpr (save) {
            const u32x4 w = {mb[0], mb[1], mb[2], mb[3]};
            *reinterpret_cast<u32x4*>(tile + act_mask_offset(VIEWDIRS ? 1 : 0, m, wave, lane)) = w;
        }
    };

    run_layer<8, 2, true, save>(st, lane, P, inE, bias_h, true, NoSide{}, mb);    // layer 0: encoded position -> P
    put_mask(0);
#pragma unroll 1
    for (int l = 1; l < 7; l += 2) {                                             // layers (1,2) (3,4) (5,6): P -> Q -> P
        if (l == 5) run_layer<8, 2, false, false>(st, lane, Q, inE, bias_h + 5 * 256, true, NoSide{}, mb);   // skip: encoded inputs first
        if constexpr (save) srows = rows(ACT_H0 + 256 * (l - 1));
        run_layer<8, 8, true, save>(st, lane, Q, inP, bias_h + l * 256, l != 5, saveP, mb);
        put_mask(l);
        if constexpr (save) srows = rows(ACT_H0 + 256 * l);
        run_layer<8, 8, true, save>(st, lane, P, inQ, bias_h + (l + 1) * 256, true, saveQ, mb);
        put_mask(l + 1);
    }
    if constexpr (save) srows = rows(ACT_H0 + 256 * 6);
    run_layer<8, 8, true, save>(st, lane, Q, inP, bias_h + 7 * 256, true, saveP, mb);  // layer 7: P -> Q
    put_mask(7);

    if (VIEWDIRS) {
        const float sigma = head_dot<8>(Q, aux + AUX_HEAD, h) + aux[AUX_HEAD + 256 + 384];
        if constexpr (save) srows = rows(ACT_H0 + 256 * 7);
        run_layer<8, 8, false, false>(st, lane, P, inQ, bias_h + 8 * 256, true, saveQ, mb); // feature_linear (no activation): Q -> P
        if constexpr (save) srows = rows(ACT_FEAT);
        run_layer<4, 9, true, save>(st, lane, Q, [&]<int ti>() -> const f32x16& { if constexpr (ti < 8) return P[ti]; else return D; },
                              bias_h + 9 * 256, true,                            // views layer: P ++ D -> Q[0..3]
                              [&]<int TO, int BI>() {   // four output tiles read eight input tiles: eight stores per tile
                                  if constexpr (save && BI >= 16 && BI < 32 && (BI & 1) == 0)
                                      save_quad<2 * TO + (BI - 16) / 8, ((BI - 16) / 2) % 4>(P[2 * TO + (BI - 16) / 8], srows, lane_off);
                              }, mb);
        put_mask(8);
        if constexpr (save) save_tiles<4>(Q, rows(ACT_VIEWS), lane_off);
        const float* wrgb = aux + AUX_HEAD + 256;
        const float* hb = aux + AUX_HEAD + 256 + 384;
        const float r = head_dot<4>(Q, wrgb, h) + hb[1];
        const float g = head_dot<4>(Q, wrgb + 128, h) + hb[2];
        const float b = head_dot<4>(Q, wrgb + 256, h) + hb[3];
        if (valid && h == 0) {
            f32x4 o = {r, g, b, sigma};
            *reinterpret_cast<f32x4*>(out + p * 4) = o;
        }
    } else {
        if constexpr (save) save_tiles<8>(Q, rows(ACT_H0 + 256 * 7), lane_off);
        for (int c = 0; c < output_ch; ++c) {
            const float v = head_dot<8>(Q, aux + AUX_HEAD + c * 256, h) + aux[AUX_HEAD + output_ch * 256 + c];
            if (valid && h == 0) out[p * output_ch + c] = v;
        }
    }
}

__global__ void __launch_bounds__(256) k_pack(int use_viewdirs, int output_ch, const float* __restrict__ params,
                                              float* __restrict__ fwd, int64_t n_fwd) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_fwd; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = fwd_source(use_viewdirs, output_ch, i);
        fwd[i] = s < 0 ? 0.0f : params[s];
    }
}

__global__ void __launch_bounds__(256) k_gather(const int32_t* __restrict__ table, const float* __restrict__ params,
                                                float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t s = table[i];
        out[i] = s < 0 ? 0.0f : params[s];
    }
}

constexpr size_t FWD_LDS_BYTES = (RING_FLOATS + AUX_FLOATS) * sizeof(float);

}  // namespace

extern "C" {

int64_t nerf_mlp_n_params(int use_viewdirs, int output_ch) { return param_offsets(use_viewdirs, output_ch).total; }
int64_t nerf_mlp_packed_floats(int use_viewdirs) { return fwd_packed_floats(use_viewdirs); }
int64_t nerf_mlp_act_floats(int use_viewdirs, int64_t n_pts) { return act_floats(use_viewdirs, n_pts); }

int64_t nerf_pack_source(int use_viewdirs, int output_ch, int64_t idx) { return fwd_source(use_viewdirs, output_ch, idx); }

int nerf_pack_source_table(int use_viewdirs, int output_ch, int64_t* table /* HOST */, int64_t n) {
    NERF_REQUIRE(table, "nerf_pack_source_table: null pointer");
    const int64_t want = fwd_packed_floats(use_viewdirs);
    NERF_REQUIRE(n == want, "nerf_pack_source_table: table has %lld entries, stream has %lld", (long long)n, (long long)want);
    for (int64_t i = 0; i < n; ++i) table[i] = fwd_source(use_viewdirs, output_ch, i);
    return 0;
}

int nerf_pack_weights(int use_viewdirs, int output_ch, const float* params, float* packed_fwd, void* stream) {
    NERF_REQUIRE(params && packed_fwd, "nerf_pack_weights: null pointer");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_pack_weights: output_ch=%d not in 1..8",
                 output_ch);
    hipLaunchKernelGGL(k_pack, dim3(1024), dim3(256), 0, as_stream(stream), use_viewdirs, output_ch, params, packed_fwd,
                       fwd_packed_floats(use_viewdirs));
    return launched("nerf_pack_weights");
}

int nerf_gather(const int32_t* table, const float* params, float* out, int64_t n, void* stream) {
    NERF_REQUIRE(n >= 0, "nerf_gather: n=%lld", (long long)n);
    if (n == 0) return 0;
    NERF_REQUIRE(table && params && out, "nerf_gather: null pointer");
    hipLaunchKernelGGL(k_gather, dim3(1024), dim3(256), 0, as_stream(stream), table, params, out, n);
    return launched("nerf_gather");
}

int nerf_mlp_fwd_encoded(int64_t n_pts, const float* x, int use_viewdirs, int output_ch, const float* packed_fwd,
                         float* out, float* act, void* stream) {
    NERF_REQUIRE(n_pts >= 0, "nerf_mlp_fwd_encoded: n_pts=%lld", (long long)n_pts);
    if (n_pts == 0) return 0;
    NERF_REQUIRE(x && packed_fwd && out, "nerf_mlp_fwd_encoded: null pointer");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_mlp_fwd_encoded: output_ch=%d not in 1..8", output_ch);
    NERF_REQUIRE((reinterpret_cast<uintptr_t>(packed_fwd) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                 "nerf_mlp_fwd_encoded: packed_fwd and out must be 16-byte aligned");
    const int64_t blocks = (n_pts + TILE_PTS - 1) / TILE_PTS;
    NERF_REQUIRE(blocks <= 0x7fffffff, "nerf_mlp_fwd_encoded: n_pts=%lld too large for one launch", (long long)n_pts);
    NERF_REQUIRE(!act || (reinterpret_cast<uintptr_t>(act) & 15) == 0, "nerf_mlp_fwd_encoded: act must be 16-byte aligned");
    const int width = IN_PTS + (use_viewdirs ? IN_DIRS : 0);
    auto go = [&](auto kernel, int och) -> int {
        hipError_t e = lds_limit_once(reinterpret_cast<const void*>(kernel), FWD_LDS_BYTES);
        if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_fwd_encoded: LDS attribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(THREADS), FWD_LDS_BYTES, as_stream(stream), n_pts, width, x,
                           static_cast<const float*>(nullptr), och, packed_fwd, out, act);
        return launched("nerf_mlp_fwd_encoded");
    };
    if (use_viewdirs) return act ? go(k_mlp_fwd<true, true>, 4) : go(k_mlp_fwd<true, false>, 4);
    return act ? go(k_mlp_fwd<false, true>, output_ch) : go(k_mlp_fwd<false, false>, output_ch);
}

}  // extern "C"
