// torch.optim.Adam's update (what upstream's optimizer.step() runs, run_nerf.py:217-218, 868-881) over runs of parameters
// that lie back to back in memory, with their gradients and both moments laid out the same way: ONE launch for every
// run of a step instead of a multi-tensor launch per 64 Ki elements of each tensor.  The arithmetic is the library
// optimiser's: products and sums in fp64 from the fp32 operands and the fp64 hyper-parameters, each assignment rounded
// to fp32; bias corrections from the step count in fp64.
#include "common.hpp"

using namespace nerfhip;

namespace {

constexpr int ADAM_MAX_RUNS = 8;
constexpr int ADAM_THREADS = 256;

struct AdamRuns {
    float* p[ADAM_MAX_RUNS];
    const float* g[ADAM_MAX_RUNS];
    float* m[ADAM_MAX_RUNS];
    float* v[ADAM_MAX_RUNS];
    int64_t first[ADAM_MAX_RUNS + 1];   // first 4-element chunk of every run among all chunks of the launch
    int64_t count[ADAM_MAX_RUNS];       // elements
    const float* step[ADAM_MAX_RUNS];   // the step count of this update (already incremented), on the device; or null: AdamConst::step
    int aligned[ADAM_MAX_RUNS];         // all four pointers 16-byte aligned
    int n;
};

struct AdamConst {
    double beta1, beta2, eps, weight_decay, lr;
    const float* lr_dev;     // if not null: the learning rate lives on the device (a graph replay reads the current value)
    float step;              // the step count of this update for runs without a device counter
};

__device__ __forceinline__ void adam_one(float& p, float g32, float& m, float& v, const AdamConst& c, double step_size, double bc2_sqrt) {
    float g = g32;
    if (c.weight_decay != 0.0) g = (float)(g + p * c.weight_decay);
    m = (float)(c.beta1 * m + (1.0 - c.beta1) * g);
    v = (float)(c.beta2 * v + (1.0 - c.beta2) * g * g);
    const float denom = (float)((double)sqrtf(v) / bc2_sqrt + c.eps);
    p = (float)(p - (float)step_size * m / denom);
}

__global__ void __launch_bounds__(ADAM_THREADS) k_adam(AdamRuns r, AdamConst c) {
    __shared__ double sh_step_size[ADAM_MAX_RUNS], sh_bc2_sqrt[ADAM_MAX_RUNS];
    if ((int)threadIdx.x < r.n) {   // per run: its bias corrections from its own step count
        const double step = r.step[threadIdx.x] ? (double)*r.step[threadIdx.x] : (double)c.step;
        const double lr = c.lr_dev ? (double)*c.lr_dev : c.lr;
        sh_step_size[threadIdx.x] = lr / (1.0 - pow(c.beta1, step));
        sh_bc2_sqrt[threadIdx.x] = sqrt(1.0 - pow(c.beta2, step));
    }
    __syncthreads();
    const int64_t chunk = (int64_t)blockIdx.x * ADAM_THREADS + threadIdx.x;
    if (chunk >= r.first[r.n]) return;
    int i = 0;
#pragma unroll
    for (int k = 1; k < ADAM_MAX_RUNS; ++k)
        if (k < r.n && chunk >= r.first[k]) i = k;
    const double step_size = sh_step_size[i], bc2_sqrt = sh_bc2_sqrt[i];
    const int64_t e0 = (chunk - r.first[i]) * 4, left = r.count[i] - e0;
    float* p = r.p[i] + e0;
    const float* g = r.g[i] + e0;
    float* m = r.m[i] + e0;
    float* v = r.v[i] + e0;
    typedef float f4 __attribute__((ext_vector_type(4)));
    if (r.aligned[i] && left >= 4) {
        f4 pp = *reinterpret_cast<f4*>(p), mm = *reinterpret_cast<f4*>(m), vv = *reinterpret_cast<f4*>(v);
        const f4 gg = *reinterpret_cast<const f4*>(g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = pp[e], me = mm[e], ve = vv[e];
            adam_one(pe, gg[e], me, ve, c, step_size, bc2_sqrt);
            pp[e] = pe; mm[e] = me; vv[e] = ve;
        }
        *reinterpret_cast<f4*>(p) = pp;
        *reinterpret_cast<f4*>(m) = mm;
        *reinterpret_cast<f4*>(v) = vv;
    } else {
        for (int e = 0; e < 4 && e < left; ++e) adam_one(p[e], g[e], m[e], v[e], c, step_size, bc2_sqrt);
    }
}

}  // namespace

extern "C" {

int nerf_adam_max_runs(void) { return ADAM_MAX_RUNS; }

int nerf_adam_step(int n_runs, const int64_t* counts, float* const* params, const float* const* grads, float* const* exp_avg,
                   float* const* exp_avg_sq, const float* const* step_dev, float step, const float* lr_dev, double lr, double beta1,
                   double beta2, double eps, double weight_decay, void* stream) {
    NERF_REQUIRE(n_runs >= 0 && n_runs <= ADAM_MAX_RUNS, "nerf_adam_step: n_runs=%d not in 0..%d", n_runs, ADAM_MAX_RUNS);
    if (n_runs == 0) return 0;
    NERF_REQUIRE(counts && params && grads && exp_avg && exp_avg_sq, "nerf_adam_step: null pointer");
    NERF_REQUIRE(step_dev || step >= 1.0f, "nerf_adam_step: step=%g (the count of this update, starting at 1)", (double)step);
    AdamRuns r{};
    int64_t chunks = 0;
    for (int i = 0; i < n_runs; ++i) {
        NERF_REQUIRE(counts[i] >= 0, "nerf_adam_step: counts[%d]=%lld", i, (long long)counts[i]);
        NERF_REQUIRE(counts[i] == 0 || (params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i]), "nerf_adam_step: null pointer in run %d", i);
        r.p[i] = params[i]; r.g[i] = grads[i]; r.m[i] = exp_avg[i]; r.v[i] = exp_avg_sq[i];
        r.count[i] = counts[i];
        r.step[i] = step_dev ? step_dev[i] : nullptr;
        NERF_REQUIRE(r.step[i] || step >= 1.0f, "nerf_adam_step: run %d has no device step counter and step=%g", i, (double)step);
        r.aligned[i] = ((reinterpret_cast<uintptr_t>(params[i]) | reinterpret_cast<uintptr_t>(grads[i]) |
                         reinterpret_cast<uintptr_t>(exp_avg[i]) | reinterpret_cast<uintptr_t>(exp_avg_sq[i])) & 15) == 0;
        r.first[i] = chunks;
        chunks += (counts[i] + 3) / 4;
    }
    for (int i = n_runs; i <= ADAM_MAX_RUNS; ++i) r.first[i] = chunks;
    r.n = n_runs;
    if (chunks == 0) return 0;
    const int64_t blocks = (chunks + ADAM_THREADS - 1) / ADAM_THREADS;
    NERF_REQUIRE(blocks <= 0x7fffffff, "nerf_adam_step: too many elements for one launch");
    AdamConst c{beta1, beta2, eps, weight_decay, lr, lr_dev, step};
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(ADAM_THREADS), 0, as_stream(stream), r, c);
    return launched("nerf_adam_step");
}

}  // extern "C"
