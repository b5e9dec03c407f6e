// nn.Linear for network shapes the fused MLP kernels do not cover (other --netdepth / --netwidth / --multires, no
// positional encoding): y = act(x W^T + b), its backward with respect to x, and its parameter gradients
// (run_nerf_helpers.py:109-133 layer by layer, and autograd of it), on v_mfma_f32_32x32x2_f32.
//
// One kernel, C[i][j] = sum_k A(i, k) B(j, k), with both operands addressed through element strides (i or j stride, k
// stride), so that the three products of a linear layer are the same code:
//   forward        y  = x W^T     A = x  (i: row, k contiguous)   B = W  (j: row, k contiguous)
//   backward-data  dx = dy W      A = dy (k contiguous)           B = W  (j contiguous, k: row)
//   weights        dW = dy^T x    A = dy (i contiguous, k: row)   B = x  (j contiguous, k: row)   K = points, cut in slices
//   bias           db = dy^T 1    B = one value with both strides 0
// A workgroup (four waves) owns a 64 x 64 tile of C, a wave a 32 x 32 quarter (16 accumulator registers); 16 k-values
// of both operands are staged in LDS per step by the thread mapping that is coalesced for the operand's contiguous
// direction, and the next step's values are loaded into registers while the current ones multiply.  fp32 operands and
// sums: results differ from the library GEMM by summation order only.  Products over the points are cut into slices of
// k whose partial tiles are summed in fixed order (bitwise reproducible, like the fused path's gradients).
// Not a tuned kernel: the shapes it serves are not part of any shipped configuration.
#include "common.hpp"

using namespace nerfhip;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TM = 64, TN = 64, TK = 16, LDT = 68;    // LDS rows of 64 values + 4 of padding

struct Operand {
    const float* p;
    int64_t s_outer, s_k;     // element strides of the i (or j) index and of k
    int n_outer;              // valid i (or j)
};

// 64 (outer) x 16 (k) values -> four registers per thread.  KFAST: consecutive threads walk k (contiguous k), else the
// outer index.
template <bool KFAST>
__device__ __forceinline__ void fetch(const Operand& op, int outer0, int64_t k0, int64_t k1, int tid, float (&r)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q;
        const int o = KFAST ? e >> 4 : e & 63, k = KFAST ? e & 15 : e >> 6;
        const int go = outer0 + o;
        const int64_t gk = k0 + k;
        r[q] = (go < op.n_outer && gk < k1) ? op.p[go * op.s_outer + gk * op.s_k] : 0.0f;
    }
}
template <bool KFAST>
__device__ __forceinline__ void park(float* tile, int tid, const float (&r)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q;
        const int o = KFAST ? e >> 4 : e & 63, k = KFAST ? e & 15 : e >> 6;
        tile[k * LDT + o] = r[q];
    }
}

// C (or slice blockIdx.z of `partial`) = A B^T over k in this slice; bias / relu only when the product is not sliced.
template <bool AK, bool BK>
__global__ void __launch_bounds__(256) k_gemm(Operand A, Operand B, int64_t K, int64_t k_per_slice, const float* __restrict__ bias,
                                              int relu, float* __restrict__ C, int64_t ldc, int64_t slice_stride) {
    __shared__ float As[2][TK * LDT], Bs[2][TK * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * TM, j0 = blockIdx.y * TN;     // (x is the grid dimension that may be long: the points)
    const int64_t k_begin = (int64_t)blockIdx.z * k_per_slice;
    const int64_t k_end = k_begin + k_per_slice < K ? k_begin + k_per_slice : K;
    const int wr = 32 * (wave >> 1), wc = 32 * (wave & 1);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float ra[4], rb[4];
    fetch<AK>(A, i0, k_begin, k_end, tid, ra);
    fetch<BK>(B, j0, k_begin, k_end, tid, rb);
    int buf = 0;
    for (int64_t k0 = k_begin; k0 < k_end; k0 += TK) {
        park<AK>(As[buf], tid, ra);
        park<BK>(Bs[buf], tid, rb);
        __syncthreads();
        if (k0 + TK < k_end) {
            fetch<AK>(A, i0, k0 + TK, k_end, tid, ra);
            fetch<BK>(B, j0, k0 + TK, k_end, tid, rb);
        }
        const float* a = As[buf] + (lane >> 5) * LDT + wr + (lane & 31);
        const float* b = Bs[buf] + (lane >> 5) * LDT + wc + (lane & 31);
#pragma unroll
        for (int k = 0; k < TK; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k * LDT], b[k * LDT], acc, 0, 0, 0);
        buf ^= 1;   // the other buffer was last read before the barrier above: no second barrier needed
    }
    float* out = C + (int64_t)blockIdx.z * slice_stride;
    const int col = j0 + wc + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = i0 + wr + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < A.n_outer && col < B.n_outer) {
            float v = acc[r];
            if (bias) v += bias[col];
            if (relu) v = v > 0.0f ? v : (v != v ? v : 0.0f);     // keeps a NaN, like torch.relu
            out[(int64_t)row * ldc + col] = v;
        }
    }
}

__global__ void __launch_bounds__(256) k_sum_slices(const float* __restrict__ partial, int64_t slice_stride, int slices, int64_t n,
                                                    float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.0f;
        for (int z = 0; z < slices; ++z) s += partial[z * slice_stride + i];
        out[i] = s;
    }
}

// The gradient entering a ReLU layer's products: zero where the saved output is <= 0, dy elsewhere -- a NaN output lets
// dy through, as torch's threshold_backward does (its test is `output <= 0`).
__global__ void __launch_bounds__(256) k_relu_grad(int64_t n, const float* __restrict__ dy, const float* __restrict__ y,
                                                   float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = y[i] <= 0.0f ? 0.0f : dy[i];
}

// Column sums of dy [n, n_out] (the bias gradient) accumulated in fp64: a block of 16 x 64 threads sums the rows of one
// slice for 64 columns, 16 rows at a time, then across its 16 row lanes; the slice's sum is left as a (high, low) pair of
// floats.  Accumulating in fp64 keeps the error at one fp32 rounding of the result whatever the row count (the matrix
// kernel used before summed each slice's ~1000 rows one after the other in fp32).
constexpr int CS_ROWS = 16;
__global__ void __launch_bounds__(64 * CS_ROWS) k_col_sum(const float* __restrict__ dy, int64_t ldy, int64_t n, int n_out,
                                                          int64_t rows_per_slice, float* __restrict__ part) {
    __shared__ double red[CS_ROWS][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6, col = blockIdx.x * 64 + c;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_slice, r1 = r0 + rows_per_slice < n ? r0 + rows_per_slice : n;
    double s = 0.0;
    if (col < n_out)
        for (int64_t r = r0 + rl; r < r1; r += CS_ROWS) s += (double)dy[r * ldy + col];
    red[rl][c] = s;
    __syncthreads();
    if (rl == 0 && col < n_out) {
#pragma unroll
        for (int k = 1; k < CS_ROWS; ++k) s += red[k][c];
        const float hi = (float)s;
        float* dst = part + 2 * ((int64_t)blockIdx.y * n_out + col);
        dst[0] = hi;
        dst[1] = (float)(s - (double)hi);
    }
}
__global__ void __launch_bounds__(256) k_col_finish(const float* __restrict__ part, int slices, int n_out, float* __restrict__ db) {
    for (int col = blockIdx.x * blockDim.x + threadIdx.x; col < n_out; col += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int z = 0; z < slices; ++z) s += (double)part[2 * ((int64_t)z * n_out + col)] + (double)part[2 * ((int64_t)z * n_out + col) + 1];
        db[col] = (float)s;
    }
}

int gemm(const Operand& A, const Operand& B, int64_t K, int slices, const float* bias, int relu, float* C, int64_t ldc,
         int64_t slice_stride, hipStream_t stream, const char* who) {
    const int64_t per = ((K + slices - 1) / slices + TK - 1) / TK * TK;
    const dim3 grid((A.n_outer + TM - 1) / TM, (B.n_outer + TN - 1) / TN, slices);
    if (grid.y > 65535u) return fail(NERF_E_ARG, "%s: %d output columns are more than one launch covers", who, B.n_outer);
    const bool ak = A.s_k == 1, bk = B.s_k == 1;
#define GO(X, Y) hipLaunchKernelGGL((k_gemm<X, Y>), grid, dim3(256), 0, stream, A, B, K, per, bias, relu, C, ldc, slice_stride)
    if (ak && bk) GO(true, true); else if (ak) GO(true, false); else if (bk) GO(false, true); else GO(false, false);
#undef GO
    return launched(who);
}

constexpr int WSLICES = 64;     // slices of the point range in the weight-gradient products

}  // namespace

extern "C" {

int nerf_linear_fwd(int64_t n, int n_in, int n_out, const float* x, int64_t ldx, const float* w, const float* b, int relu,
                    float* y, int64_t ldy, void* stream) {
    NERF_REQUIRE(n >= 0 && n_in >= 1 && n_out >= 1 && ldx >= n_in && ldy >= n_out, "nerf_linear_fwd: n=%lld in=%d out=%d ldx=%lld ldy=%lld",
                 (long long)n, n_in, n_out, (long long)ldx, (long long)ldy);
    NERF_REQUIRE(n <= 0x7fffffff - TM, "nerf_linear_fwd: n=%lld too large for one launch", (long long)n);
    if (n == 0) return 0;
    NERF_REQUIRE(x && w && y, "nerf_linear_fwd: null pointer");
    const Operand A{x, ldx, 1, (int)n}, B{w, n_in, 1, n_out};
    return gemm(A, B, n_in, 1, b, relu, y, ldy, 0, as_stream(stream), "nerf_linear_fwd");
}

int nerf_relu_grad(int64_t n, const float* dy, const float* y, float* out, void* stream) {
    NERF_REQUIRE(n >= 0, "nerf_relu_grad: n=%lld", (long long)n);
    if (n == 0) return 0;
    NERF_REQUIRE(dy && y && out, "nerf_relu_grad: null pointer");
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_relu_grad, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, as_stream(stream), n, dy, y, out);
    return launched("nerf_relu_grad");
}

int nerf_linear_bwd_data(int64_t n, int n_in, int n_out, const float* dy, int64_t ldy, const float* w, float* dx, int64_t ldx,
                         void* stream) {
    NERF_REQUIRE(n >= 0 && n_in >= 1 && n_out >= 1 && ldx >= n_in && ldy >= n_out, "nerf_linear_bwd_data: n=%lld in=%d out=%d",
                 (long long)n, n_in, n_out);
    NERF_REQUIRE(n <= 0x7fffffff - TM, "nerf_linear_bwd_data: n=%lld too large for one launch", (long long)n);
    if (n == 0) return 0;
    NERF_REQUIRE(dy && w && dx, "nerf_linear_bwd_data: null pointer");
    const Operand A{dy, ldy, 1, (int)n}, B{w, 1, n_in, n_in};     // B(j, k) = W[k][j]
    return gemm(A, B, n_out, 1, nullptr, 0, dx, ldx, 0, as_stream(stream), "nerf_linear_bwd_data");
}

int64_t nerf_linear_workspace_floats(int n_in, int n_out) { return (int64_t)WSLICES * ((int64_t)n_out * n_in + n_out) + 1; }

int nerf_linear_bwd_weights(int64_t n, int n_in, int n_out, const float* dy, int64_t ldy, const float* x, int64_t ldx, float* dw,
                            float* db, float* workspace, void* stream_) {
    NERF_REQUIRE(n >= 0 && n_in >= 1 && n_out >= 1 && ldx >= n_in && ldy >= n_out, "nerf_linear_bwd_weights: n=%lld in=%d out=%d",
                 (long long)n, n_in, n_out);
    NERF_REQUIRE(dw, "nerf_linear_bwd_weights: null dw");
    hipStream_t stream = as_stream(stream_);
    const int64_t wn = (int64_t)n_out * n_in;
    if (n == 0) {
        hipError_t e = hipMemsetAsync(dw, 0, wn * sizeof(float), stream);
        if (e == hipSuccess && db) e = hipMemsetAsync(db, 0, n_out * sizeof(float), stream);
        return e == hipSuccess ? 0 : fail(NERF_E_LAUNCH, "nerf_linear_bwd_weights: memset: %s", hipGetErrorString(e));
    }
    NERF_REQUIRE(dy && x && workspace, "nerf_linear_bwd_weights: null pointer");
    int slices = (int)((n + 1023) / 1024);
    if (slices > WSLICES) slices = WSLICES;
    float* part_w = workspace;
    float* part_b = workspace + (int64_t)WSLICES * wn;
    const Operand A{dy, 1, ldy, n_out};                   // A(i, k) = dy[k][i]
    const Operand Bx{x, 1, ldx, n_in};                    // B(j, k) = x[k][j]
    int rc = gemm(A, Bx, n, slices, nullptr, 0, part_w, n_in, wn, stream, "nerf_linear_bwd_weights");
    if (rc) return rc;
    hipLaunchKernelGGL(k_sum_slices, dim3((unsigned)((wn + 255) / 256 > 1024 ? 1024 : (wn + 255) / 256)), dim3(256), 0, stream,
                       part_w, wn, slices, wn, dw);
    rc = launched("nerf_linear_bwd_weights (sum)");
    if (rc || !db) return rc;
    // db = column sums of dy, in fp64, over at most WSLICES / 2 slices (a slice leaves two floats per column)
    int bslices = (int)((n + 4095) / 4096);
    if (bslices > WSLICES / 2) bslices = WSLICES / 2;
    const int64_t rows_per_slice = (n + bslices - 1) / bslices;
    hipLaunchKernelGGL(k_col_sum, dim3((unsigned)((n_out + 63) / 64), (unsigned)bslices), dim3(64 * CS_ROWS), 0, stream, dy, ldy, n,
                       n_out, rows_per_slice, part_b);
    rc = launched("nerf_linear_bwd_weights (bias)");
    if (rc) return rc;
    hipLaunchKernelGGL(k_col_finish, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, stream, part_b, bslices, n_out, db);
    return launched("nerf_linear_bwd_weights (bias sum)");
}

}  // extern "C"
