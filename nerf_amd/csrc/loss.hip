// The photometric loss of a training batch, mean((x - y)^2) (upstream: img2mse, run_nerf_helpers.py:8, called at
// run_nerf.py:863-869), and its gradient with respect to x: one launch each.  A batch is a few thousand rays, so the
// forward is ONE workgroup -- per-thread fp64 partial sums, shuffles inside a wave, LDS across the sixteen waves, always in
// the same order: the result is reproducible bit for bit and within one rounding of the exact mean -- instead of the
// stock element-wise launch followed by a reduction launch (and three more on the way back).
#include "common.hpp"

using namespace nerfhip;

namespace {

constexpr int MSE_THREADS = 1024;
constexpr int64_t MSE_MAX_ELEMS = (int64_t)1 << 17;     // one workgroup reads at most 1 MiB (43,690 rays)

// sum over one input of (x - y)^2 by the whole workgroup, in fp64; every thread returns the total
__device__ __forceinline__ double block_sum_sq(int64_t n, const float* __restrict__ x, const float* __restrict__ y, double* part) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    double s = 0.0;
    // (the difference and its square in fp32, as the stock kernel forms them; 16-byte loads, four of them in flight)
    const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    const int64_t n4 = vec ? n / 4 : 0;
#pragma unroll 4
    for (int64_t i = threadIdx.x; i < n4; i += MSE_THREADS) {
        const f4 a = reinterpret_cast<const f4*>(x)[i], b = reinterpret_cast<const f4*>(y)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = a[e] - b[e];
            s += (double)(d * d);
        }
    }
    for (int64_t i = 4 * n4 + threadIdx.x; i < n; i += MSE_THREADS) {
        const float d = x[i] - y[i];
        s += (double)(d * d);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __syncthreads();                       // (part may still be read from the previous input)
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < MSE_THREADS / 64; ++w) t += part[w];
    return t;
}

// out0 = mean((x0 - y)^2); with a second input: out1 = mean((x1 - y)^2), out_sum = out0 + out1 (added in fp32, as the
// caller's own addition of the two losses would be)
__global__ void __launch_bounds__(MSE_THREADS) k_mse_fwd(int64_t n, const float* __restrict__ x0, const float* __restrict__ x1,
                                                         const float* __restrict__ y, float* __restrict__ out0,
                                                         float* __restrict__ out1, float* __restrict__ out_sum) {
    __shared__ double part[MSE_THREADS / 64];
    const float m0 = (float)(block_sum_sq(n, x0, y, part) / (double)n);
    float m1 = 0.0f;
    if (x1) m1 = (float)(block_sum_sq(n, x1, y, part) / (double)n);
    if (threadIdx.x == 0) {
        out0[0] = m0;
        if (x1) {
            out1[0] = m1;
            out_sum[0] = m0 + m1;
        }
    }
}

// dx_k = g_k * (2 / n) * (x_k - y); g_k = the upstream gradient of loss k plus that of the sum (device scalars, each may
// be absent); blocks [0, nb) take input 0, blocks [nb, 2 nb) input 1
__global__ void __launch_bounds__(256) k_mse_bwd(int64_t n, int nb, const float* __restrict__ x0, const float* __restrict__ x1,
                                                 const float* __restrict__ y, const float* __restrict__ g0,
                                                 const float* __restrict__ g1, const float* __restrict__ g_sum,
                                                 float* __restrict__ d0, float* __restrict__ d1) {
    const int which = (int)blockIdx.x >= nb ? 1 : 0;
    const int64_t i = (int64_t)((int)blockIdx.x - which * nb) * 256 + threadIdx.x;
    if (i >= n) return;
    const float* g = which ? g1 : g0;
    const float up = (g ? g[0] : 0.0f) + (g_sum ? g_sum[0] : 0.0f);
    const float scale = up * (float)(2.0 / (double)n);
    const float* x = which ? x1 : x0;
    (which ? d1 : d0)[i] = scale * (x[i] - y[i]);
}

}  // namespace

extern "C" {

int64_t nerf_mse_max_elems(void) { return MSE_MAX_ELEMS; }

int nerf_mse_fwd(int64_t n, const float* x, const float* y, float* out, void* stream) {
    NERF_REQUIRE(n >= 1 && n <= MSE_MAX_ELEMS, "nerf_mse_fwd: n=%lld not in 1..%lld", (long long)n, (long long)MSE_MAX_ELEMS);
    NERF_REQUIRE(x && y && out, "nerf_mse_fwd: null pointer");
    hipLaunchKernelGGL(k_mse_fwd, dim3(1), dim3(MSE_THREADS), 0, as_stream(stream), n, x, (const float*)nullptr, y, out,
                       (float*)nullptr, (float*)nullptr);
    return launched("nerf_mse_fwd");
}

int nerf_mse_bwd(int64_t n, const float* x, const float* y, const float* grad_out, float* dx, void* stream) {
    NERF_REQUIRE(n >= 1 && n <= MSE_MAX_ELEMS, "nerf_mse_bwd: n=%lld not in 1..%lld", (long long)n, (long long)MSE_MAX_ELEMS);
    NERF_REQUIRE(x && y && grad_out && dx, "nerf_mse_bwd: null pointer");
    const int nb = (int)((n + 255) / 256);
    hipLaunchKernelGGL(k_mse_bwd, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), n, nb, x, (const float*)nullptr, y, grad_out,
                       (const float*)nullptr, (const float*)nullptr, dx, (float*)nullptr);
    return launched("nerf_mse_bwd");
}

int nerf_mse_pair_fwd(int64_t n, const float* x0, const float* x1, const float* y, float* out0, float* out1, float* out_sum,
                      void* stream) {
    NERF_REQUIRE(n >= 1 && n <= MSE_MAX_ELEMS, "nerf_mse_pair_fwd: n=%lld not in 1..%lld", (long long)n, (long long)MSE_MAX_ELEMS);
    NERF_REQUIRE(x0 && x1 && y && out0 && out1 && out_sum, "nerf_mse_pair_fwd: null pointer");
    hipLaunchKernelGGL(k_mse_fwd, dim3(1), dim3(MSE_THREADS), 0, as_stream(stream), n, x0, x1, y, out0, out1, out_sum);
    return launched("nerf_mse_pair_fwd");
}

int nerf_mse_pair_bwd(int64_t n, const float* x0, const float* x1, const float* y, const float* grad0, const float* grad1,
                      const float* grad_sum, float* d0, float* d1, void* stream) {
    NERF_REQUIRE(n >= 1 && n <= MSE_MAX_ELEMS, "nerf_mse_pair_bwd: n=%lld not in 1..%lld", (long long)n, (long long)MSE_MAX_ELEMS);
    NERF_REQUIRE(x0 && x1 && y && d0 && d1, "nerf_mse_pair_bwd: null pointer");
    const int nb = (int)((n + 255) / 256);
    hipLaunchKernelGGL(k_mse_bwd, dim3((unsigned)(2 * nb)), dim3(256), 0, as_stream(stream), n, nb, x0, x1, y, grad0, grad1, grad_sum,
                       d0, d1);
    return launched("nerf_mse_pair_bwd");
}

}  // extern "C"
