// Backward of the fused MLP w.r.t. its parameters.
//
// Three stages on one stream:
//  1. k_mlp_bwd_data: dL/d(pre-activation) of every layer.  Same structure as the forward: one
//     wavefront carries 32 points, gradients live in MFMA accumulator tiles and feed the next
//     (earlier) layer's W^T product straight from registers; the transposed weights stream through
//     the LDS slab ring.  ReLU masks come from the saved activations; each layer's dY is written
//     feature-major next to them.                                      Bound: MFMA (9088 per 32 points).
//  2. k_dw_gemm: dW_l = dY_l * X_l^T, a reduction over points.  Both operands are feature-major rows,
//     i.e. K(points)-contiguous, so a [rows][32 points] tile goes to LDS with whole-line loads and a
//     lane reads its MFMA operands as 16-byte vectors.  One workgroup owns a whole (up to 256x256)
//     weight tile for one of KSPLIT slices of the point range: every operand byte is read from HBM
//     once (64 FLOP per byte).  Bias gradients are row sums of the dY fragments already in registers.
//  3. k_reduce_slabs: sums the KSPLIT partial slabs in a fixed order (bitwise reproducible, unlike
//     float atomics).
#include "common.hpp"
#include "mlp_core.hpp"

using namespace nerfhip;
using namespace nerflayout;
using namespace nerfmlp;

namespace {

// ------------------------------------------------------------------------------------------------
// stage 1
// ------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void load_tiles(f32x16 (&M)[8], const float* rows, int64_t ldp, uint32_t lane_off) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) M[t][r] = (rows + (int64_t)(32 * t + (r & 3) + 8 * (r >> 2)) * ldp)[lane_off];
}

template <int NT>
__device__ __forceinline__ void store_tiles(const f32x16 (&A)[8], float* rows, int64_t ldp, uint32_t lane_off) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) (rows + (int64_t)(32 * t + (r & 3) + 8 * (r >> 2)) * ldp)[lane_off] = A[t][r];
}

template <bool VIEWDIRS>
__global__ void __launch_bounds__(THREADS, 1)
k_mlp_bwd_data(int64_t n_pts, int output_ch, const float* __restrict__ d_out, const float* __restrict__ act,
               const float* __restrict__ packed_fwd, const float* __restrict__ packed_bwd, float* __restrict__ scr) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ring = smem;
    float* aux = smem + RING_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int64_t p = (int64_t)blockIdx.x * TILE_PTS + wave * 32 + (lane & 31);
    const bool valid = p < n_pts;
    const int64_t ldp = padded_points(n_pts);
    const uint32_t lane_off = (uint32_t)(4 * h * ldp + p);

    {
        const float* gaux = packed_fwd + (int64_t)fwd_stream_blocks(VIEWDIRS ? 1 : 0) * BLOCK_FLOATS;
        for (int i = tid * 4; i < AUX_FLOATS; i += THREADS * 4)
            *reinterpret_cast<f32x4*>(aux + i) = *reinterpret_cast<const f32x4*>(gaux + i);
    }
    Stream st;
    st.ring = ring;
    st.mine = ring + wave * 4 * BLOCK_FLOATS;
    st.g = packed_bwd + wave * 4 * BLOCK_FLOATS + lane * 4;
    st.next = 0;
    st.nslabs = bwd_stream_blocks(VIEWDIRS ? 1 : 0) / SLAB_BLOCKS;
    __syncthreads();
    st.fetch(0);

    f32x16 A[8], B[8], M[8];
    float d_sigma = 0.0f;
    int first_layer;  // the pts layer whose output gradient the first 256x256 section produces
    if constexpr (VIEWDIRS) {
        f32x4 g = {0, 0, 0, 0};
        if (valid) g = *reinterpret_cast<const f32x4*>(d_out + p * 4);
        d_sigma = g[3];
        if (h == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) scr[(int64_t)(SCR_HEAD + c) * ldp + p] = g[c];
            scr[(int64_t)SCR_HEAD2 * ldp + p] = g[3];
        }
        // views layer: d(hidden) = Wrgb^T d_rgb, masked by the saved post-ReLU activations
        load_tiles<4>(M, act + (int64_t)ACT_VIEWS * ldp, ldp, lane_off);
        const float* wrgb = aux + AUX_HEAD + 256;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wrgb + t * 32 + h * 16 + 4 * i);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(wrgb + 128 + t * 32 + h * 16 + 4 * i);
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(wrgb + 256 + t * 32 + h * 16 + 4 * i);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = fmaf(w2[e], g[2], fmaf(w1[e], g[1], w0[e] * g[0]));
                    A[t][4 * i + e] = M[t][4 * i + e] > 0.0f ? v : 0.0f;
                }
            }
        store_tiles<4>(A, scr + (int64_t)SCR_VIEWS * ldp, ldp, lane_off);
        // d(feature) = Wviews[:, :256]^T dY_views; feature_linear has no activation
        run_section<8, 4, false>(st, lane, B, [&]<int ti>() -> const f32x16& { return A[ti]; });
#pragma unroll
        for (int t = 0; t < 8; ++t) A[t] = B[t];
        store_tiles<8>(A, scr + (int64_t)SCR_FEAT * ldp, ldp, lane_off);
        first_layer = 7;
    } else {
        // d(h7) = Wout^T d_out
        float g[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) g[c] = (valid && c < output_ch) ? d_out[p * output_ch + c] : 0.0f;
        if (h == 0) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < output_ch) scr[(int64_t)(SCR_HEAD + c) * ldp + p] = g[c];
        }
        load_tiles<8>(M, act + (int64_t)(ACT_H0 + 7 * 256) * ldp, ldp, lane_off);
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = 0.0f;
                for (int c = 0; c < output_ch; ++c) v = fmaf(aux[AUX_HEAD + c * 256 + t * 32 + h * 16 + r], g[c], v);
                A[t][r] = M[t][r] > 0.0f ? v : 0.0f;
            }
        store_tiles<8>(A, scr + (int64_t)(SCR_H0 + 7 * 256) * ldp, ldp, lane_off);
        first_layer = 6;
    }

#pragma unroll 1
    for (int l = first_layer; l >= 0; --l) {
        // saved output of pts layer l (ReLU mask), requested before the matrix work that hides its latency
        load_tiles<8>(M, act + (int64_t)(ACT_H0 + 256 * l) * ldp, ldp, lane_off);
        run_section<8, 8, false>(st, lane, B, [&]<int ti>() -> const f32x16& { return A[ti]; });
        if (VIEWDIRS && l == 7) {
            const float* wa = aux + AUX_HEAD;
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(wa + t * 32 + h * 16 + 4 * i);
#pragma unroll
                    for (int e = 0; e < 4; ++e) B[t][4 * i + e] = fmaf(w[e], d_sigma, B[t][4 * i + e]);
                }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) A[t][r] = M[t][r] > 0.0f ? B[t][r] : 0.0f;
        store_tiles<8>(A, scr + (int64_t)(SCR_H0 + 256 * l) * ldp, ldp, lane_off);
    }
}

// ------------------------------------------------------------------------------------------------
// stage 2
// ------------------------------------------------------------------------------------------------
struct DwTask {
    int cfg;        // tile shape, see k_dw_gemm
    int a_row;      // first dY row in the workspace
    int b_row;      // first input row in the saved activations
    int M, N;       // valid output rows / columns (<= tile)
    int ldc;        // row stride of the weight matrix in the flat parameter vector
    int64_t out;    // offset of W[0][first column] in the flat parameter vector
    int64_t bias;   // offset of the bias vector, or -1
};
struct DwTasks {
    DwTask t[16];
};

constexpr int KT = 32;           // points per K chunk
constexpr int LDT = KT + 4;      // padded LDS row (floats): 16-byte aligned, conflict-free b128 reads

constexpr int DW_THREADS = 512;

template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void dw_body(const DwTask& task, int split, int64_t ldp, const float* __restrict__ act,
                                        const float* __restrict__ scr_rows, float* __restrict__ slabs,
                                        int64_t slab_stride, float* smem) {
    static_assert(64 * WM * WN == DW_THREADS, "every tile shape uses all eight waves");
    constexpr int NTHREADS = DW_THREADS, TILE_M = WM * TM * 32, TILE_N = WN * TN * 32;
    constexpr int ROWS = TILE_M + TILE_N;
    constexpr int LOADS = (ROWS * (KT / 4) + NTHREADS - 1) / NTHREADS;  // float4 per thread per chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;

    const int64_t nchunks = ldp / KT;
    const int64_t per = (nchunks + KSPLIT - 1) / KSPLIT;
    const int64_t c0 = split * per, c1 = (c0 + per < nchunks) ? c0 + per : nchunks;

    const float* a_base = scr_rows + (int64_t)task.a_row * ldp;
    const float* b_base = act + (int64_t)task.b_row * ldp;

    // staging map: float4 i of the [ROWS][KT] chunk -> row i/8, column 4*(i%8); rows < TILE_M come from dY
    const float* src[LOADS];
    int dst[LOADS];
#pragma unroll
    for (int k = 0; k < LOADS; ++k) {
        const int i = tid + k * NTHREADS;
        const int row = i >> 3, c4 = (i & 7) * 4;
        const bool ok = row < ROWS;
        const int rr = ok ? row : 0;
        src[k] = (rr < TILE_M ? a_base + (int64_t)rr * ldp : b_base + (int64_t)(rr - TILE_M) * ldp) + c4;
        dst[k] = ok ? rr * LDT + c4 : -1;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float bsum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) bsum[i] = 0.0f;

    f32x4 stage[LOADS];
    auto issue = [&](int64_t chunk) {
#pragma unroll
        for (int k = 0; k < LOADS; ++k) stage[k] = *reinterpret_cast<const f32x4*>(src[k] + chunk * KT);
    };
    auto commit = [&](int buf) {
        float* base = smem + buf * (ROWS * LDT);
#pragma unroll
        for (int k = 0; k < LOADS; ++k)
            if (dst[k] >= 0) *reinterpret_cast<f32x4*>(base + dst[k]) = stage[k];
    };

    if (c0 < c1) {
        issue(c0);
        commit(0);
    }
    __syncthreads();
    for (int64_t c = c0; c < c1; ++c) {
        const int buf = (int)((c - c0) & 1);
        if (c + 1 < c1) issue(c + 1);  // in flight under this chunk's MFMAs
        const float* At = smem + buf * (ROWS * LDT) + (wm * TM * 32 + l32) * LDT + 4 * h;
        const float* Bt = smem + buf * (ROWS * LDT) + (TILE_M + wn * TN * 32 + l32) * LDT + 4 * h;
#pragma unroll
        for (int q = 0; q < KT / 8; ++q) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(At + i * 32 * LDT + 8 * q);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bt + j * 32 * LDT + 8 * q);
            if (wn == 0) {
#pragma unroll
                for (int i = 0; i < TM; ++i) bsum[i] += (a[i][0] + a[i][1]) + (a[i][2] + a[i][3]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        if (c + 1 < c1) commit(buf ^ 1);
        __syncthreads();
    }

    float* out = slabs + (int64_t)split * slab_stride;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = (wn * TN + j) * 32 + l32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < task.M && n < task.N) out[task.out + (int64_t)m * task.ldc + n] = acc[i][j][r];
            }
        }
    if (task.bias >= 0 && wn == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float s = bsum[i] + __shfl_xor(bsum[i], 32, 64);
            const int m = (wm * TM + i) * 32 + l32;
            if (h == 0 && m < task.M) out[task.bias + m] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// stage 3
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_reduce_slabs(const float* __restrict__ slabs, int64_t slab_stride, int64_t n,
                                                      float* __restrict__ d_params) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.0f;
#pragma unroll 8
        for (int k = 0; k < KSPLIT; ++k) s += slabs[k * slab_stride + i];
        d_params[i] = s;
    }
}

// Tile shapes (rows of dY x rows of X per workgroup, eight waves each):
//   0: 256x256  hidden layers, feature_linear      1: 256x64  encoded-position inputs of layers 0 and 5
//   2: 128x256  views layer, feature inputs        3: 128x64  views layer, encoded directions (27 used)
//   4:  64x128  rgb head (3 rows used)             5:  32x256 alpha / output_linear head
// All tasks of a backward run in ONE launch, largest first, so the short ones fill the tail of the long ones.
__global__ void __launch_bounds__(DW_THREADS)
k_dw_gemm(DwTasks tasks, int64_t ldp, const float* __restrict__ act, const float* __restrict__ scr_rows,
          float* __restrict__ slabs, int64_t slab_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DwTask task = tasks.t[blockIdx.x / KSPLIT];
    const int split = blockIdx.x % KSPLIT;
    switch (task.cfg) {
        case 0: dw_body<2, 4, 4, 2>(task, split, ldp, act, scr_rows, slabs, slab_stride, smem); break;
        case 1: dw_body<8, 1, 1, 2>(task, split, ldp, act, scr_rows, slabs, slab_stride, smem); break;
        case 2: dw_body<2, 4, 2, 2>(task, split, ldp, act, scr_rows, slabs, slab_stride, smem); break;
        case 3: dw_body<4, 2, 1, 1>(task, split, ldp, act, scr_rows, slabs, slab_stride, smem); break;
        case 4: dw_body<2, 4, 1, 1>(task, split, ldp, act, scr_rows, slabs, slab_stride, smem); break;
        default: dw_body<1, 8, 1, 1>(task, split, ldp, act, scr_rows, slabs, slab_stride, smem); break;
    }
}

constexpr size_t DW_LDS_BYTES = 2 * (size_t)512 * LDT * sizeof(float);  // the 256x256 shape: 512 rows, two buffers

}  // namespace

extern "C" {

int64_t nerf_mlp_scratch_floats(int use_viewdirs, int64_t n_pts) { return scratch_floats(use_viewdirs, 8, n_pts); }

int nerf_mlp_bwd(int64_t n_pts, int pts_per_ray, const float* d_out, const float* act, int use_viewdirs, int output_ch,
                 const float* packed_fwd, const float* packed_bwd, float* scratch, float* d_params, void* stream_) {
    (void)pts_per_ray;
    NERF_REQUIRE(n_pts >= 0, "nerf_mlp_bwd: n_pts=%lld", (long long)n_pts);
    NERF_REQUIRE(d_params, "nerf_mlp_bwd: null d_params");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_mlp_bwd: output_ch=%d not in 1..8", output_ch);
    hipStream_t stream = as_stream(stream_);
    const ParamOffsets po = param_offsets(use_viewdirs, output_ch);
    if (n_pts == 0) {
        hipError_t e = hipMemsetAsync(d_params, 0, po.total * sizeof(float), stream);
        return e == hipSuccess ? 0 : fail(NERF_E_LAUNCH, "nerf_mlp_bwd: memset: %s", hipGetErrorString(e));
    }
    NERF_REQUIRE(d_out && act && packed_fwd && packed_bwd && scratch, "nerf_mlp_bwd: null pointer");
    NERF_REQUIRE(((reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(act) | reinterpret_cast<uintptr_t>(scratch) |
                   reinterpret_cast<uintptr_t>(packed_fwd) | reinterpret_cast<uintptr_t>(packed_bwd)) & 15) == 0,
                 "nerf_mlp_bwd: buffers must be 16-byte aligned");
    const int64_t ldp = padded_points(n_pts);
    NERF_REQUIRE(ldp * 4 * 5 < (int64_t)0xffffffff, "nerf_mlp_bwd: n_pts=%lld too large for one call", (long long)n_pts);
    const int64_t blocks = ldp / TILE_PTS;
    const int64_t slab = slab_floats(use_viewdirs, output_ch);
    float* slabs = scratch + (int64_t)SCR_ROWS * ldp;

    // head rows beyond the output channels are zero operands of the head GEMMs; uncovered slab entries
    // (parameters that receive no gradient) must read as zero in the reduction
    hipError_t e = hipMemsetAsync(scratch + (int64_t)SCR_HEAD * ldp, 0, 64 * ldp * sizeof(float), stream);
    if (e == hipSuccess) e = hipMemsetAsync(slabs, 0, (size_t)KSPLIT * slab * sizeof(float), stream);
    if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_bwd: memset: %s", hipGetErrorString(e));

    const size_t lds = (RING_FLOATS + AUX_FLOATS) * sizeof(float);
    if (use_viewdirs) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_bwd_data<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_bwd: LDS attribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(k_mlp_bwd_data<true>, dim3((unsigned)blocks), dim3(THREADS), lds, stream, n_pts, 4, d_out, act,
                           packed_fwd, packed_bwd, scratch);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_bwd_data<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_bwd: LDS attribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(k_mlp_bwd_data<false>, dim3((unsigned)blocks), dim3(THREADS), lds, stream, n_pts, output_ch, d_out,
                           act, packed_fwd, packed_bwd, scratch);
    }
    int rc = launched("nerf_mlp_bwd (data)");
    if (rc) return rc;

    // ---- dW = dY X^T: one launch over all layers
    DwTasks tk{};
    int nt = 0;
    for (int l = 1; l < 8; ++l) {
        const int ldc = (l == 5) ? W + IN_PTS : W;
        tk.t[nt++] = DwTask{0, SCR_H0 + 256 * l, ACT_H0 + 256 * (l - 1), 256, 256, ldc, po.pts_w[l] + (l == 5 ? IN_PTS : 0), po.pts_b[l]};
    }
    if (use_viewdirs) {
        tk.t[nt++] = DwTask{0, SCR_FEAT, ACT_H0 + 256 * 7, 256, 256, W, po.feat_w, po.feat_b};
        tk.t[nt++] = DwTask{2, SCR_VIEWS, ACT_FEAT, 128, 256, po.views_in, po.views_w, po.views_b};
    }
    tk.t[nt++] = DwTask{1, SCR_H0, ACT_ENC_PTS, 256, IN_PTS, IN_PTS, po.pts_w[0], po.pts_b[0]};
    tk.t[nt++] = DwTask{1, SCR_H0 + 256 * 5, ACT_ENC_PTS, 256, IN_PTS, W + IN_PTS, po.pts_w[5], -1};
    if (use_viewdirs) {
        tk.t[nt++] = DwTask{5, SCR_HEAD2, ACT_H0 + 256 * 7, 1, 256, W, po.alpha_w, po.alpha_b};
        tk.t[nt++] = DwTask{3, SCR_VIEWS, ACT_ENC_DIRS, 128, IN_DIRS, po.views_in, po.views_w + W, -1};
        tk.t[nt++] = DwTask{4, SCR_HEAD, ACT_VIEWS, 3, 128, W / 2, po.rgb_w, po.rgb_b};
    } else {
        tk.t[nt++] = DwTask{5, SCR_HEAD, ACT_H0 + 256 * 7, output_ch, 256, W, po.out_w, po.out_b};
    }
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_dw_gemm), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DW_LDS_BYTES);
    if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_bwd: LDS attribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_dw_gemm, dim3(nt * KSPLIT), dim3(DW_THREADS), DW_LDS_BYTES, stream, tk, ldp, act, scratch, slabs, slab);
    rc = launched("nerf_mlp_bwd (dW)");
    if (rc) return rc;

    hipLaunchKernelGGL(k_reduce_slabs, dim3(1024), dim3(256), 0, stream, slabs, slab, po.total, d_params);
    return launched("nerf_mlp_bwd (reduce)");
}

}  // extern "C"
