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

__global__ void __launch_bounds__(MSE_THREADS) k_mse_fwd(int64_t n, const float* __restrict__ x, const float* __restrict__ y,
                                                         float* __restrict__ out) {
    __shared__ double part[MSE_THREADS / 64];
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
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < MSE_THREADS / 64; ++w) t += part[w];
        out[0] = (float)(t / (double)n);
    }
}

// dx = g * (2 / n) * (x - y); g: the upstream gradient of the scalar loss, on the device
__global__ void __launch_bounds__(256) k_mse_bwd(int64_t n, const float* __restrict__ x, const float* __restrict__ y,
                                                 const float* __restrict__ g, float* __restrict__ dx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float scale = g[0] * (float)(2.0 / (double)n);
    dx[i] = scale * (x[i] - y[i]);
}

}  // namespace

extern "C" {

int64_t nerf_mse_max_elems(void) { return MSE_MAX_ELEMS; }

int nerf_mse_fwd(int64_t n, const float* x, const float* y, float* out, void* stream) {
    NERF_REQUIRE(n >= 1 && n <= MSE_MAX_ELEMS, "nerf_mse_fwd: n=%lld not in 1..%lld", (long long)n, (long long)MSE_MAX_ELEMS);
    NERF_REQUIRE(x && y && out, "nerf_mse_fwd: null pointer");
    hipLaunchKernelGGL(k_mse_fwd, dim3(1), dim3(MSE_THREADS), 0, as_stream(stream), n, x, y, out);
    return launched("nerf_mse_fwd");
}

int nerf_mse_bwd(int64_t n, const float* x, const float* y, const float* grad_out, float* dx, void* stream) {
    NERF_REQUIRE(n >= 1 && n <= MSE_MAX_ELEMS, "nerf_mse_bwd: n=%lld not in 1..%lld", (long long)n, (long long)MSE_MAX_ELEMS);
    NERF_REQUIRE(x && y && grad_out && dx, "nerf_mse_bwd: null pointer");
    hipLaunchKernelGGL(k_mse_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), n, x, y, grad_out, dx);
    return launched("nerf_mse_bwd");
}

}  // extern "C"
