// Backward of the fused MLP w.r.t. its parameters: the entry points, and the weight-gradient stage on the fp32 matrix
// instruction.
//
// Three stages on one stream:
//  1. backward-data: dL/d(pre-activation) of every layer, written in the blocked layout of mlp_layout.hpp
//     (mlp_bwd_bf16x3.hip: split operands on the bf16 matrix instruction, the default; mlp_bwd_x4.hip: fp32 matrix
//     instruction, for models with mlp_dtype 'f32_mfma').
//  2. dW_l = dY_l * X_l^T, a reduction over points (mlp_dw_x3.hip for the split-operand path; k_dw_gemm below for the
//     fp32 one).  k_dw_gemm: both operands live in the blocked layout: a chunk of 32 points is 512 contiguous bytes
//     per group of four rows, copied as-is into LDS, where a lane's ds_read_b128 yields four rows of one point = the A
//     (or B) operands of four different 32x32 output tiles.  One workgroup owns a whole (up to 256x256) weight tile for
//     one slice of the point range: every operand byte is read from HBM once (64 FLOP per byte).  The slices per task
//     are chosen so that all workgroups of the launch carry the same load and fit the chip in one round.  Bias
//     gradients are row sums of the dY operands.
//  3. k_reduce_slabs: sums the partial slabs in a fixed order (bitwise reproducible, unlike float atomics).
#include "common.hpp"
#include "mlp_core.hpp"
#include "mlp_x4.hpp"
#include "mlp_dw.hpp"
#include <mutex>

// stage 1 on the split-operand kernel (mlp_bwd_bf16x3.hip)
namespace nerfx3 {
int launch_bwd_data_x3(int64_t n_pts, int use_viewdirs, int output_ch, const float* d_out, const float* act,
                       const void* packed_x3, const void* packed_bwd_x3, float* scratch, const uint32_t* marked,
                       hipStream_t stream);
}

using namespace nerfhip;
using namespace nerflayout;
using namespace nerfmlp;

namespace {

// ------------------------------------------------------------------------------------------------
// stage 2
// ------------------------------------------------------------------------------------------------
// One task = one weight matrix (or two that share an operand).  Its operands are up to four runs of
// consecutive row groups, copied chunk by chunk into consecutive LDS "slots" (one slot = one group =
// 4 rows x 32 points):  input run 1 | dY run 1 | dY run 2 | input run 2.
struct DwTask {
    int kind;             // see wave_job()
    int a_grp, a_n;       // dY groups in the workspace: first, count
    int a2_grp, a2_n;
    int b_grp, b_n;       // input groups in the saved activations
    int b2_grp, b2_n;
    int M, N, N2;         // valid rows of the dY run(s); valid columns of input run 1 / run 2
    int ldc, ldc2;        // row stride of the weight matrices in the flat parameter vector
    int first, nsplit;    // workgroups [first, first + nsplit) slice the point range of this task
    int64_t out, out2;    // W[0][first column] for (dY 1 x input 1) and for the second product
    int64_t bias;         // bias vector of dY run 1, or -1
};
struct DwTasks {
    DwTask t[14];
    int n;
};

constexpr int KT = 32;              // points per chunk
constexpr int GS = KT * 4 + 4;      // floats per LDS slot: 32 points x 4 rows, padded so that lanes reading
                                    // consecutive slots start 4 banks apart (conflict-free b128 / b64 reads)
constexpr int MAX_SLOTS = 144;      // the encoded-position task: 2 x 64 dY groups + 16 input groups
constexpr int DW_THREADS = 512;
constexpr size_t DW_LDS_BYTES = 2 * (size_t)MAX_SLOTS * GS * sizeof(float);

// What one wave of a task's workgroup computes: a (32*TM) x (32*TN) block of one product.
struct WaveJob {
    int mode;         // 0 idle | 1: 128 rows x 64 cols | 2: 128 rows x 32 cols | 3: <=8 rows x 64 cols
    int a_slot, b_slot;
    int row0, col0;   // position of the block in the weight matrix
    int M, N, ldc;
    int64_t out, bias;
};

struct RunStarts { int a1, a2, b2; };   // first LDS slot of dY run 1, dY run 2, input run 2 (input run 1 starts at 0)
__host__ __device__ inline RunStarts run_starts(const DwTask& t) {
    RunStarts r;
    r.a1 = (t.b_n + 15) & ~15;
    r.a2 = r.a1 + ((t.a_n + 15) & ~15);
    r.b2 = r.a2 + ((t.a2_n + 15) & ~15);
    return r;
}

// kind 0  hidden layer, 256 x 256:  waves 2 x 4, each 128 x 64
// kind 1  encoded-position inputs of layers 0 and 5: dY0 ++ dY5 (512 rows) x 64 inputs: waves 0..3, 128 x 64 each
// kind 2  views layer: 128 rows x (256 feature ++ 32 direction inputs): waves 0..3 128 x 64, wave 4 128 x 32
// kind 3  head: <= 8 rows x (64 * b_n / 16) inputs: wave w covers columns 64 w
__device__ __forceinline__ WaveJob wave_job(const DwTask& t, int w) {
    const RunStarts rs = run_starts(t);
    WaveJob j{};
    j.M = t.M; j.N = t.N; j.ldc = t.ldc; j.out = t.out; j.bias = -1;
    switch (t.kind) {
        case 0:
            j.mode = 1; j.a_slot = rs.a1 + (w >> 2) * 32; j.b_slot = (w & 3) * 16; j.row0 = (w >> 2) * 128; j.col0 = (w & 3) * 64;
            if ((w & 3) == 0) j.bias = t.bias;
            break;
        case 1:
            if (w < 4) {
                j.mode = 1; j.a_slot = rs.a1 + w * 32; j.b_slot = 0; j.row0 = (w & 1) * 128; j.col0 = 0;
                if (w < 2) j.bias = t.bias; else { j.out = t.out2; j.ldc = t.ldc2; }
            }
            break;
        case 2:
            if (w < 4) {
                j.mode = 1; j.a_slot = rs.a1; j.b_slot = w * 16; j.row0 = 0; j.col0 = w * 64;
                if (w == 0) j.bias = t.bias;
            } else if (w == 4 && t.b2_n > 0) {
                j.mode = 2; j.a_slot = rs.a1; j.b_slot = rs.b2; j.row0 = 0; j.col0 = 0; j.out = t.out2; j.ldc = t.ldc2; j.N = t.N2;
            }
            break;
        default:
            if (w * 16 < t.b_n) {
                j.mode = 3; j.a_slot = rs.a1; j.b_slot = w * 16; j.row0 = 0; j.col0 = w * 64;
                if (w == 0) j.bias = t.bias;
            }
            break;
    }
    return j;
}

// The chunk loop of one wave.  All waves of the workgroup copy chunks (global -> registers -> LDS, two LDS
// buffers) and meet at one barrier per chunk; TM x TN MFMA tiles per wave depend on the job.
template <int MODE, int NLOADS>
__device__ __forceinline__ void dw_loop(const DwTask& task, const WaveJob& job, int split, int64_t nchunks,
                                        int64_t tile_floats, const float* __restrict__ act,
                                        const float* __restrict__ scr, float* __restrict__ slab, float* smem) {
    constexpr int TM = (MODE == 1 || MODE == 2) ? 4 : (MODE == 3 ? 1 : 0);
    constexpr int TN = (MODE == 1 || MODE == 3) ? 2 : (MODE == 2 ? 1 : 0);
    constexpr int NACC = TM * TN > 0 ? TM * TN : 1;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l32 = lane & 31;

    const int64_t per = (nchunks + task.nsplit - 1) / task.nsplit;
    const int64_t c0 = split * per, c1 = (c0 + per < nchunks) ? c0 + per : nchunks;

    // ---- staging map.  LDS slots: input run 1 | dY run 1 | dY run 2 | input run 2, each run starting at a
    // multiple of 16 slots, so that the 16 slots a load instruction k covers (slot 16 k + tid / 32, point tid % 32)
    // come from one buffer: scalar base per k, one per-lane offset for all k.
    const RunStarts rs = run_starts(task);
    const float* gbase[NLOADS];
    uint32_t lane_src[NLOADS];
    const int sl = tid >> 5;
#pragma unroll
    for (int k = 0; k < NLOADS; ++k) {
        const int s0 = 16 * k;
        int grp, n;
        const float* base;
        if (s0 < rs.a1)      { base = act; grp = task.b_grp + s0;            n = task.b_n - s0; }
        else if (s0 < rs.a2) { base = scr; grp = task.a_grp + (s0 - rs.a1);  n = task.a_n - (s0 - rs.a1); }
        else if (s0 < rs.b2) { base = scr; grp = task.a2_grp + (s0 - rs.a2); n = task.a2_n - (s0 - rs.a2); }
        else                 { base = act; grp = task.b2_grp + (s0 - rs.b2); n = task.b2_n - (s0 - rs.b2); }
        gbase[k] = base + (int64_t)grp * GROUP_FLOATS;
        // slots past the end of a run are filled with a copy of the run's first group: no lane is ever masked,
        // the loop body has no branches, and nothing reads those slots
        const int nvalid = n > 16 ? 16 : n;
        lane_src[k] = (uint32_t)((sl < nvalid ? sl : 0) * GROUP_FLOATS + l32 * 4);
    }
    const int dst0 = sl * GS + l32 * 4;

    f32x4 stage[NLOADS];
    auto issue = [&](int64_t chunk) {
        const int64_t off = (chunk >> 2) * tile_floats + (chunk & 3) * (KT * 4);   // both buffers share the tile stride
#pragma unroll
        for (int k = 0; k < NLOADS; ++k) stage[k] = *reinterpret_cast<const f32x4*>(gbase[k] + off + lane_src[k]);
    };
    auto commit = [&](int buf) {
        float* base = smem + buf * (MAX_SLOTS * GS) + dst0;
#pragma unroll
        for (int k = 0; k < NLOADS; ++k) *reinterpret_cast<f32x4*>(base + 16 * k * GS) = stage[k];
    };

    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    float bsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const bool want_bias = job.bias >= 0;

    // this lane's operand addresses inside a buffer (floats); step s adds 8 s (two points per step, lane half = parity)
    int a_off = 0, b_off = 0;
    bool a_live = true;
    if constexpr (MODE == 1 || MODE == 2) a_off = (job.a_slot + l32) * GS + 4 * h;
    if constexpr (MODE == 3) { a_off = (job.a_slot + (l32 >> 2)) * GS + (l32 & 3) + 4 * h; a_live = l32 < job.M; }
    if constexpr (MODE == 1 || MODE == 3) b_off = (job.b_slot + (l32 >> 1)) * GS + 2 * (l32 & 1) + 4 * h;
    if constexpr (MODE == 2) b_off = (job.b_slot + (l32 >> 2)) * GS + (l32 & 3) + 4 * h;
    if (!a_live) a_off = 0;

    // Operands of step s are read one step ahead of the MFMAs that use them (the scheduler, left alone, issues each
    // read right in front of its first use and parks the wave for the LDS latency sixteen times per chunk).
    struct Operands { float a[4]; float b[2]; };
    auto read_operands = [&](const float* buf, int s) {
        Operands o{};
        if constexpr (TM == 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(buf + a_off + 8 * s);
            o.a[0] = v[0]; o.a[1] = v[1]; o.a[2] = v[2]; o.a[3] = v[3];
        } else if constexpr (TM == 1) {
            const float v = buf[a_off + 8 * s];
            o.a[0] = a_live ? v : 0.0f;
        }
        if constexpr (TN == 2) {
            const float2 v = *reinterpret_cast<const float2*>(buf + b_off + 8 * s);
            o.b[0] = v.x; o.b[1] = v.y;
        } else if constexpr (TN == 1) {
            o.b[0] = buf[b_off + 8 * s];
        }
        return o;
    };
    // steps [s_begin, s_end) of the chunk in `buf`; `cur` holds the operands of s_begin on entry and those of s_end
    // on exit (when s_end is still inside the chunk)
    auto compute = [&](const float* buf, int s_begin, int s_end, Operands& cur, float (&csum)[4]) {
        if constexpr (MODE != 0) {
#pragma unroll
            for (int s = s_begin; s < s_end; ++s) {
                Operands nxt = cur;
                if (s + 1 < KT / 2) nxt = read_operands(buf, s + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < TM; ++e) csum[e] += cur.a[e];
#pragma unroll
                for (int e = 0; e < TM; ++e)
#pragma unroll
                    for (int f = 0; f < TN; ++f)
                        acc[e * TN + f] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[e], cur.b[f], acc[e * TN + f], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
        }
    };

    if (c0 < c1) {
        issue(c0);
        commit(0);
    }
    __syncthreads();
    for (int64_t c = c0; c < c1; ++c) {
        const int buf = (int)((c - c0) & 1);
        const float* cur = smem + buf * (MAX_SLOTS * GS);
        float csum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        Operands ops{};
        if constexpr (MODE != 0) ops = read_operands(cur, 0);
        issue(c + 1 < c1 ? c + 1 : c);          // in flight under the first half of this chunk's MFMAs
        __builtin_amdgcn_sched_barrier(0);      // (the scheduler would sink the loads to just above their stores)
        compute(cur, 0, KT / 4, ops, csum);     // (the last iteration re-stages its own chunk: no branch in the body)
        __builtin_amdgcn_sched_barrier(0);
        commit(buf ^ 1);                        // the other buffer was last read before the previous barrier
        __builtin_amdgcn_sched_barrier(0);
        compute(cur, KT / 4, KT / 2, ops, csum);
        if (want_bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bsum[e] += csum[e];
        }
        __syncthreads();
    }

    if constexpr (MODE != 0) {
        float* out = slab + job.out;
#pragma unroll
        for (int e = 0; e < TM; ++e)
#pragma unroll
            for (int f = 0; f < TN; ++f) {
                const int col = job.col0 + (TN == 2 ? 2 * l32 + f : l32);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int row = job.row0 + (TM == 4 ? 4 * m + e : m);
                    if (row < job.M && col < job.N) out[(int64_t)row * job.ldc + col] = acc[e * TN + f][r];
                }
            }
        if (want_bias) {
#pragma unroll
            for (int e = 0; e < TM; ++e) {
                const float s = bsum[e] + __shfl_xor(bsum[e], 32, 64);
                const int row = job.row0 + (TM == 4 ? 4 * l32 + e : l32);
                if (h == 0 && row < job.M) slab[job.bias + row] = s;
            }
        }
    }
}

__global__ void __launch_bounds__(DW_THREADS)
k_dw_gemm(DwTasks tasks, int64_t nchunks, int64_t tile_floats, const float* __restrict__ act,
          const float* __restrict__ scr, float* __restrict__ slabs, int64_t slab_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int ti = 0;
    for (int i = 1; i < tasks.n; ++i)
        if ((int)blockIdx.x >= tasks.t[i].first) ti = i;
    const DwTask task = tasks.t[ti];
    const int split = (int)blockIdx.x - task.first;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const WaveJob job = wave_job(task, wave);
    float* slab = slabs + (int64_t)split * slab_stride;
    // (wave job, 16-slot groups per chunk of the task's kind)
#define DW_GO(MODE, NLOADS) dw_loop<MODE, NLOADS>(task, job, split, nchunks, tile_floats, act, scr, slab, smem)
    switch (task.kind) {
        case 0: DW_GO(1, 8); break;
        case 1: if (job.mode == 1) DW_GO(1, 9); else DW_GO(0, 9); break;
        case 2: if (job.mode == 1) DW_GO(1, 7); else if (job.mode == 2) DW_GO(2, 7); else DW_GO(0, 7); break;
        default: if (job.mode == 3) DW_GO(3, 5); else DW_GO(0, 5); break;
    }
#undef DW_GO
}

// ------------------------------------------------------------------------------------------------
// stage 3
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_reduce_slabs(const float* __restrict__ slabs, int64_t slab_stride, int64_t n,
                                                      int nslabs, float* __restrict__ d_params) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.0f;
        for (int k = 0; k < nslabs; ++k) s += slabs[k * slab_stride + i];
        d_params[i] = s;
    }
}

__global__ void __launch_bounds__(256) k_reduce_regions(nerfdw::SlabRegions regions, const float* __restrict__ slabs,
                                                        int64_t slab_stride, float* __restrict__ d_params) {
    const nerfdw::SlabRegion& R = regions.r[blockIdx.y];
    const int64_t total = (int64_t)R.rows * R.cols;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / R.cols;
        const int64_t i = R.out + row * R.ld + (e - row * R.cols);
        // eight loads in flight, added in slab order (the order is what makes the gradients reproducible)
        const float* src = slabs + i;
        float s = 0.0f;
        int k = 0;
        for (; k + 8 <= R.nsplit; k += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[(int64_t)(k + j) * slab_stride];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; k < R.nsplit; ++k) s += src[(int64_t)k * slab_stride];
        d_params[i] = s;
    }
}

// Slices per task: repeatedly give one more slice to the task whose slices are longest, until the launch
// has one workgroup per compute unit.  cost = time of one chunk of the task relative to the others
// (MFMA issue per SIMD, or the operand copy where that is longer).
void assign_splits(DwTasks& tk, const int* cost, int64_t nchunks, int n_wg) {
    int used = 0;
    for (int i = 0; i < tk.n; ++i) { tk.t[i].nsplit = 1; ++used; }
    const int cap = nerfdw::max_slabs(nchunks * KT);
    while (used < n_wg) {
        int best = -1;
        double worst = 0.0;
        for (int i = 0; i < tk.n; ++i) {
            if (tk.t[i].nsplit >= cap) continue;
            const double per = (double)cost[i] / tk.t[i].nsplit;
            if (per > worst) { worst = per; best = i; }
        }
        if (best < 0) break;
        ++tk.t[best].nsplit;
        ++used;
    }
    int first = 0;
    for (int i = 0; i < tk.n; ++i) { tk.t[i].first = first; first += tk.t[i].nsplit; }
}

}  // namespace

// stages 2 and 3: dW = dY X^T over all layers in one launch, then the fixed-order sum of the partial slabs
static int run_dw(int64_t ldp, int use_viewdirs, int output_ch, const ParamOffsets& po, const float* act, float* scratch,
                  float* slabs, int64_t slab, float* d_params, hipStream_t stream) {
    const int SCR_HEAD = scr_head(use_viewdirs), SCR_HEAD2 = SCR_HEAD + 32;
    hipError_t e;
    int rc;
    // ---- dW = dY X^T: one launch over all layers
    DwTasks tk{};
    int cost[14];
    int nt = 0;
    auto big = [&](int a_row, int b_row, int ldc, int64_t out, int64_t bias) {
        DwTask t{};
        t.kind = 0; t.a_grp = a_row / 4; t.a_n = 64; t.b_grp = b_row / 4; t.b_n = 64; t.M = 256; t.N = 256; t.ldc = ldc;
        t.out = out; t.bias = bias;
        cost[nt] = 16; tk.t[nt++] = t;
    };
    for (int l = 1; l < 8; ++l)
        big(SCR_H0 + 256 * l, ACT_H0 + 256 * (l - 1), (l == 5) ? W + IN_PTS : W, po.pts_w[l] + (l == 5 ? IN_PTS : 0), po.pts_b[l]);
    if (use_viewdirs) big(SCR_FEAT, ACT_H0 + 256 * 7, W, po.feat_w, po.feat_b);
    {   // encoded positions feed layers 0 and 5
        DwTask t{};
        t.kind = 1; t.a_grp = SCR_H0 / 4; t.a_n = 64; t.a2_grp = (SCR_H0 + 256 * 5) / 4; t.a2_n = 64; t.b_grp = ACT_ENC_PTS / 4; t.b_n = 16;
        t.M = 256; t.N = IN_PTS; t.ldc = IN_PTS; t.ldc2 = W + IN_PTS; t.out = po.pts_w[0]; t.out2 = po.pts_w[5]; t.bias = po.pts_b[0];
        cost[nt] = 9; tk.t[nt++] = t;
    }
    if (use_viewdirs) {
        DwTask t{};
        t.kind = 2; t.a_grp = SCR_VIEWS / 4; t.a_n = 32; t.b_grp = ACT_FEAT / 4; t.b_n = 64; t.b2_grp = ACT_ENC_DIRS / 4; t.b2_n = 8;
        t.M = 128; t.N = 256; t.N2 = IN_DIRS; t.ldc = t.ldc2 = po.views_in; t.out = po.views_w; t.out2 = po.views_w + W; t.bias = po.views_b;
        cost[nt] = 12; tk.t[nt++] = t;
        DwTask a{};   // alpha: dL/d(sigma) x h7
        a.kind = 3; a.a_grp = SCR_HEAD2 / 4; a.a_n = 2; a.b_grp = (ACT_H0 + 256 * 7) / 4; a.b_n = 64; a.M = 1; a.N = 256; a.ldc = W;
        a.out = po.alpha_w; a.bias = po.alpha_b;
        cost[nt] = 5; tk.t[nt++] = a;
        DwTask c{};   // rgb: dL/d(rgb) x views hidden
        c.kind = 3; c.a_grp = SCR_HEAD / 4; c.a_n = 2; c.b_grp = ACT_VIEWS / 4; c.b_n = 32; c.M = 3; c.N = 128; c.ldc = W / 2;
        c.out = po.rgb_w; c.bias = po.rgb_b;
        cost[nt] = 3; tk.t[nt++] = c;
    } else {
        DwTask a{};   // output_linear: dL/d(out) x h7
        a.kind = 3; a.a_grp = SCR_HEAD / 4; a.a_n = 2; a.b_grp = (ACT_H0 + 256 * 7) / 4; a.b_n = 64; a.M = output_ch; a.N = 256; a.ldc = W;
        a.out = po.out_w; a.bias = po.out_b;
        cost[nt] = 5; tk.t[nt++] = a;
    }
    tk.n = nt;
    int n_cu = compute_units();
    if (n_cu < nt) n_cu = nt;
    const int64_t nchunks = ldp / KT;
    assign_splits(tk, cost, nchunks, n_cu);
    int n_wg = 0, nslabs = 0;
    for (int i = 0; i < nt; ++i) {
        n_wg += tk.t[i].nsplit;
        if (tk.t[i].nsplit > nslabs) nslabs = tk.t[i].nsplit;
    }

    // slab entries no task covers (parameters that receive no gradient) must read as zero in the reduction
    e = hipMemsetAsync(slabs, 0, (size_t)nslabs * slab * sizeof(float), stream);
    if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_bwd: memset: %s", hipGetErrorString(e));
    e = lds_limit_once(reinterpret_cast<const void*>(k_dw_gemm), DW_LDS_BYTES);
    if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_bwd: LDS attribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_dw_gemm, dim3(n_wg), dim3(DW_THREADS), DW_LDS_BYTES, stream, tk, nchunks,
                       (int64_t)act_rows(use_viewdirs) * TILE_PTS, act, scratch, slabs, slab);
    rc = launched("nerf_mlp_bwd (dW)");
    if (rc) return rc;

    return nerfdw::reduce_slabs(slabs, slab, po.total, nslabs, d_params, stream);
}

namespace nerfdw {
int reduce_regions(const SlabRegions& regions, const float* slabs, int64_t slab_stride, float* d_params, hipStream_t stream) {
    if (regions.n == 0) return 0;
    // one thread per element of the largest region (256 x 256 weights); the blocks of a smaller region that find nothing
    // to do leave at once
    int64_t most = 1;
    for (int r = 0; r < regions.n; ++r) {
        const int64_t elems = (int64_t)regions.r[r].rows * regions.r[r].cols;
        if (elems > most) most = elems;
    }
    const int64_t bx = (most + 255) / 256;
    hipLaunchKernelGGL(k_reduce_regions, dim3((unsigned)(bx > 512 ? 512 : bx), regions.n), dim3(256), 0, stream, regions, slabs,
                       slab_stride, d_params);
    return launched("nerf_mlp_bwd (reduce)");
}
int reduce_slabs(const float* slabs, int64_t slab_stride, int64_t n, int nslabs, float* d_params, hipStream_t stream) {
    hipLaunchKernelGGL(k_reduce_slabs, dim3(1024), dim3(256), 0, stream, slabs, slab_stride, n, nslabs, d_params);
    return launched("nerf_mlp_bwd (reduce)");
}
}  // namespace nerfdw

extern "C" {

int64_t nerf_mlp_scratch_floats(int use_viewdirs, int64_t n_pts) { return scratch_floats(use_viewdirs, 8, n_pts); }

/* The parameter-gradient pass on the fp32 matrix instruction: backward-data on 16-point tiles, then k_dw_gemm.
 * packed_x4: the stream of nerf_pack_weights_x4 (its fp32 tables are read); packed_bwd_x4: nerf_pack_weights_bwd_x4. */
int nerf_mlp_bwd_x4(int64_t n_pts, const float* d_out, const float* act, int use_viewdirs, int output_ch,
                    const float* packed_x4, const float* packed_bwd_x4, float* scratch, float* d_params, void* stream_) {
    NERF_REQUIRE(n_pts >= 0, "nerf_mlp_bwd_x4: n_pts=%lld", (long long)n_pts);
    NERF_REQUIRE(d_params, "nerf_mlp_bwd_x4: null d_params");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_mlp_bwd_x4: output_ch=%d not in 1..8", output_ch);
    hipStream_t stream = as_stream(stream_);
    const ParamOffsets po = param_offsets(use_viewdirs, output_ch);
    if (n_pts == 0) {
        hipError_t e0 = hipMemsetAsync(d_params, 0, po.total * sizeof(float), stream);
        return e0 == hipSuccess ? 0 : fail(NERF_E_LAUNCH, "nerf_mlp_bwd_x4: memset: %s", hipGetErrorString(e0));
    }
    NERF_REQUIRE(d_out && act && packed_x4 && packed_bwd_x4 && scratch, "nerf_mlp_bwd_x4: null pointer");
    NERF_REQUIRE(((reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(act) | reinterpret_cast<uintptr_t>(scratch) |
                   reinterpret_cast<uintptr_t>(packed_x4) | reinterpret_cast<uintptr_t>(packed_bwd_x4)) & 15) == 0,
                 "nerf_mlp_bwd_x4: buffers must be 16-byte aligned");
    const int64_t ldp = padded_points(n_pts);
    NERF_REQUIRE(ldp / TILE_PTS <= 0x7fffffff, "nerf_mlp_bwd_x4: n_pts=%lld too large for one call", (long long)n_pts);
    const int64_t slab = slab_floats(use_viewdirs, output_ch);
    float* slabs = scratch + (int64_t)scr_rows(use_viewdirs) * ldp;
    int rc = nerfx4::launch_bwd_data_x4(n_pts, use_viewdirs, output_ch, d_out, act, packed_x4, packed_bwd_x4, scratch, stream);
    if (rc) return rc;
    return run_dw(ldp, use_viewdirs, output_ch, po, act, scratch, slabs, slab, d_params, stream);
}

/* The parameter-gradient pass on the split-operand kernels, stage by stage (the stages only communicate through
 * `scratch`; nerf_mlp_bwd_bf16x3 runs both). */
static int check_bwd_x3(const char* who, int64_t n_pts, const float* d_out, const float* act, int use_viewdirs, int output_ch,
                        const void* packed_x3, const void* packed_bwd_x3, const float* scratch) {
    NERF_REQUIRE(n_pts >= 0, "%s: n_pts=%lld", who, (long long)n_pts);
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "%s: output_ch=%d not in 1..8", who, output_ch);
    if (n_pts == 0) return 0;
    NERF_REQUIRE(d_out && act && packed_x3 && packed_bwd_x3 && scratch, "%s: null pointer", who);
    NERF_REQUIRE(((reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(act) | reinterpret_cast<uintptr_t>(scratch) |
                   reinterpret_cast<uintptr_t>(packed_x3) | reinterpret_cast<uintptr_t>(packed_bwd_x3)) & 15) == 0,
                 "%s: buffers must be 16-byte aligned", who);
    NERF_REQUIRE(padded_points(n_pts) / TILE_PTS <= 0x7fffffff, "%s: n_pts=%lld too large for one call", who, (long long)n_pts);
    return 0;
}

int nerf_mlp_bwd_data_bf16x3(int64_t n_pts, const float* d_out, const float* act, int use_viewdirs, int output_ch,
                             const void* packed_x3, const void* packed_bwd_x3, float* scratch, void* stream_) {
    int rc = check_bwd_x3("nerf_mlp_bwd_data_bf16x3", n_pts, d_out, act, use_viewdirs, output_ch, packed_x3, packed_bwd_x3, scratch);
    if (rc || n_pts == 0) return rc;
    return nerfx3::launch_bwd_data_x3(n_pts, use_viewdirs, output_ch, d_out, act, packed_x3, packed_bwd_x3, scratch, nullptr, as_stream(stream_));
}

int nerf_mlp_bwd_weights_bf16x3(int64_t n_pts, const float* act, int use_viewdirs, int output_ch, float* scratch,
                                float* d_params, void* stream_) {
    NERF_REQUIRE(n_pts >= 0, "nerf_mlp_bwd_weights_bf16x3: n_pts=%lld", (long long)n_pts);
    NERF_REQUIRE(d_params, "nerf_mlp_bwd_weights_bf16x3: null d_params");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_mlp_bwd_weights_bf16x3: output_ch=%d not in 1..8", output_ch);
    hipStream_t stream = as_stream(stream_);
    const ParamOffsets po = param_offsets(use_viewdirs, output_ch);
    if (n_pts == 0) {
        hipError_t e0 = hipMemsetAsync(d_params, 0, po.total * sizeof(float), stream);
        return e0 == hipSuccess ? 0 : fail(NERF_E_LAUNCH, "nerf_mlp_bwd_weights_bf16x3: memset: %s", hipGetErrorString(e0));
    }
    NERF_REQUIRE(act && scratch, "nerf_mlp_bwd_weights_bf16x3: null pointer");
    NERF_REQUIRE(((reinterpret_cast<uintptr_t>(act) | reinterpret_cast<uintptr_t>(scratch)) & 15) == 0,
                 "nerf_mlp_bwd_weights_bf16x3: buffers must be 16-byte aligned");
    const int64_t ldp = padded_points(n_pts);
    const int64_t slab = slab_floats(use_viewdirs, output_ch);
    float* slabs = scratch + (int64_t)scr_rows(use_viewdirs) * ldp;
    return nerfdw::run_dw_x3(ldp, use_viewdirs, output_ch, act, scratch, slabs, slab, d_params, stream);
}

static_assert(nerfdw::DY_STAGES == NERF_DY_STAGES && nerfdw::DW_CTL_WORDS == NERF_DW_CTL_WORDS, "include/nerf_hip.h and mlp_dw.hpp");

int nerf_mlp_bwd_weights_f16x2(int64_t n_pts, const float* act, int use_viewdirs, int output_ch, float* scratch,
                               float* d_params, const uint32_t* fwd_flags, const uint32_t* bwd_flags, const uint32_t* tile_max,
                               uint32_t* ctl, void* stream_) {
    NERF_REQUIRE(n_pts >= 0, "nerf_mlp_bwd_weights_f16x2: n_pts=%lld", (long long)n_pts);
    NERF_REQUIRE(d_params, "nerf_mlp_bwd_weights_f16x2: null d_params");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_mlp_bwd_weights_f16x2: output_ch=%d not in 1..8", output_ch);
    hipStream_t stream = as_stream(stream_);
    const ParamOffsets po = param_offsets(use_viewdirs, output_ch);
    if (n_pts == 0) {
        hipError_t e0 = hipMemsetAsync(d_params, 0, po.total * sizeof(float), stream);
        return e0 == hipSuccess ? 0 : fail(NERF_E_LAUNCH, "nerf_mlp_bwd_weights_f16x2: memset: %s", hipGetErrorString(e0));
    }
    NERF_REQUIRE(act && scratch && bwd_flags && tile_max && ctl, "nerf_mlp_bwd_weights_f16x2: null pointer");
    NERF_REQUIRE(((reinterpret_cast<uintptr_t>(act) | reinterpret_cast<uintptr_t>(scratch)) & 15) == 0,
                 "nerf_mlp_bwd_weights_f16x2: buffers must be 16-byte aligned");
    const int64_t ldp = padded_points(n_pts);
    const int64_t slab = slab_floats(use_viewdirs, output_ch);
    float* slabs = scratch + (int64_t)scr_rows(use_viewdirs) * ldp;
    return nerfdw::run_dw_x2(ldp, use_viewdirs, output_ch, act, scratch, slabs, slab, d_params, fwd_flags, bwd_flags, tile_max, ctl, stream);
}

int nerf_mlp_bwd_bf16x3(int64_t n_pts, const float* d_out, const float* act, int use_viewdirs, int output_ch,
                        const void* packed_x3, const void* packed_bwd_x3, float* scratch, float* d_params, void* stream_) {
    NERF_REQUIRE(d_params, "nerf_mlp_bwd_bf16x3: null d_params");
    int rc = nerf_mlp_bwd_data_bf16x3(n_pts, d_out, act, use_viewdirs, output_ch, packed_x3, packed_bwd_x3, scratch, stream_);
    if (rc) return rc;
    return nerf_mlp_bwd_weights_bf16x3(n_pts, act, use_viewdirs, output_ch, scratch, d_params, stream_);
}

/* ---- the same backward for callers that cannot allocate ----------------------------------------------------------
 * Workspace of nerf_mlp_bwd_bf16x3 for up to NERF_STATIC_BWD_MAX_PTS points, held by the code object on every device
 * that loads the library: the dY rows of those points + the partial-gradient slabs that many points can need. */
}  // extern "C"

namespace {
constexpr int64_t STATIC_MAX_PTS = NERF_STATIC_BWD_MAX_PTS;
constexpr int64_t STATIC_SLABS = STATIC_MAX_PTS / (8 * 32) < KSPLIT ? STATIC_MAX_PTS / (8 * 32) : KSPLIT;
constexpr int64_t STATIC_SLAB_FLOATS = 595904;   // slab_floats() of the largest supported network (view branch); checked per call
constexpr int64_t STATIC_FLOATS = (int64_t)scr_rows(1) * STATIC_MAX_PTS + STATIC_SLABS * STATIC_SLAB_FLOATS;
static __device__ __align__(256) float scratch[STATIC_FLOATS];

// Device address of `scratch` on each device: the array's name is a host-side handle, not a pointer a kernel can use.
constexpr int MAX_DEVICES = 64;
float* g_scratch_ptr[MAX_DEVICES] = {};
std::mutex g_scratch_mutex;

int static_scratch(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= MAX_DEVICES)
        return fail(NERF_E_LAUNCH, "nerf_mlp_bwd_bf16x3_static: hipGetDevice: %s (device %d)", hipGetErrorString(e), dev);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return fail(NERF_E_LAUNCH, "nerf_mlp_bwd_bf16x3_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
        g_scratch_ptr[dev] = static_cast<float*>(p);
    }
    *out = g_scratch_ptr[dev];
    return 0;
}
}  // namespace

extern "C" {

int64_t nerf_mlp_bwd_static_max_points(void) { return STATIC_MAX_PTS; }

int nerf_mlp_bwd_bf16x3_static(int64_t n_pts, const float* d_out, const float* act, int use_viewdirs, int output_ch,
                               const void* packed_x3, const void* packed_bwd_x3, float* d_params, void* stream_) {
    NERF_REQUIRE(n_pts >= 0 && n_pts <= STATIC_MAX_PTS,
                 "nerf_mlp_bwd_bf16x3_static: n_pts=%lld exceeds the built-in workspace (at most %lld points per call; "
                 "use nerf_mlp_bwd_bf16x3 with a caller workspace for more)", (long long)n_pts, (long long)STATIC_MAX_PTS);
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_mlp_bwd_bf16x3_static: output_ch=%d not in 1..8", output_ch);
    NERF_REQUIRE(slab_floats(use_viewdirs, output_ch) <= STATIC_SLAB_FLOATS && scr_rows(use_viewdirs) <= scr_rows(1),
                 "nerf_mlp_bwd_bf16x3_static: network shape larger than the built-in workspace");
    float* ws = nullptr;
    int rc = static_scratch(&ws);
    if (rc) return rc;
    return nerf_mlp_bwd_bf16x3(n_pts, d_out, act, use_viewdirs, output_ch, packed_x3, packed_bwd_x3, ws, d_params, stream_);
}

}  // extern "C"
