// Alpha compositing along rays (forward and backward w.r.t. the network output).
//
// One wavefront per ray, one lane per sample, 64 samples per tile: lane i of a tile loads
// raw[ray][s0+i] as one 16-byte vector, so a wave reads 1 KiB of contiguous HBM per tile.  The
// exclusive transmittance product is a 64-lane Hillis-Steele scan carried in fp64 (the PyTorch CPU
// cumprod accumulates in fp64 and rounds each element to fp32; an fp64 scan reproduces those
// roundings up to double rounding), with a scalar carry between tiles.  Ray sums are fp32, like the
// reference's: per-lane partial sums over the tiles, then a balanced DPP tree over the lanes.  HBM-bound: 24*S + 36 bytes per ray forward, 36*S + 24 backward (+4*S with noise).
#include "common.hpp"
#include "wave_ops.hpp"
#pragma clang fp contract(off)

using namespace nerfhip;

namespace {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int MAX_TILES = 64;  // backward keeps one carry per 64-sample tile: S <= 4096

struct Sample {
    float4 raw;
    float z, dist, sig, alpha, keep;  // keep = 1 - alpha + 1e-10 as fp32
};

// Lane i of a tile holds sample s = s0 + i.  What a lane reads from memory for one tile ...
struct Loaded {
    float4 raw;
    float z, z_tile, nz;   // z_tile: first depth of the next tile (one address for the whole wave)
};
__device__ __forceinline__ Loaded fetch_tile(const float* __restrict__ raw, const float* __restrict__ z,
                                             const float* __restrict__ noise, int64_t base, int s0, int lane, int S) {
    const int s = s0 + lane;
    const bool live = s < S;
    Loaded f;
    f.raw = live ? reinterpret_cast<const float4*>(raw)[base + s] : make_float4(0.f, 0.f, 0.f, 0.f);
    f.z = live ? z[base + s] : 0.0f;
    f.z_tile = (s0 + 64 < S) ? z[base + s0 + 64] : 0.0f;
    f.nz = live && noise ? noise[base + s] : 0.0f;
    return f;
}
// ... and what it derives from that.  The gap to the next depth comes from the neighbouring lane (wave_shl:1); the
// last lane takes the first depth of the next tile.
__device__ __forceinline__ Sample make_sample(const Loaded& f, bool with_noise, int s0, int lane, int S, float dnorm) {
    const int s = s0 + lane;
    Sample q;
    q.raw = f.raw;
    q.z = f.z;
    // wave_shl:1 -- lane i receives lane i + 1, lane 63 keeps `old`
    const float z_next = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(f.z_tile), __float_as_int(f.z), 0x130, 0xf, 0xf, false));
    const float gap = (s < S - 1) ? (z_next - q.z) : 1e10f;
    q.dist = gap * dnorm;
    q.sig = with_noise ? (q.raw.w + f.nz) : q.raw.w;
    const float act = q.sig > 0.0f ? q.sig : (q.sig != q.sig ? q.sig : 0.0f);  // relu, NaN kept
    q.alpha = 1.0f - expf(-act * q.dist);
    q.keep = s < S ? (1.0f - q.alpha) + 1e-10f : 1.0f;
    return q;
}
__device__ __forceinline__ Sample load_sample(const float* __restrict__ raw, const float* __restrict__ z,
                                              const float* __restrict__ noise, int64_t base, int s0, int lane, int S,
                                              float dnorm) {
    return make_sample(fetch_tile(raw, z, noise, base, s0, lane, S), noise != nullptr, s0, lane, S, dnorm);
}

// Colour sigmoid on the hardware exp2 / reciprocal: within 3 ulp of the correctly rounded value over the whole
// range (saturating to 0 and 1 like it), four instructions instead of two dozen.  The density term above keeps the
// accurate expf: its error is raised to the power of the sample count by the transmittance product.
__device__ __forceinline__ float sigmoidf(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}

__device__ __forceinline__ float ray_norm(const float* rr) {
    const float dx = rr[0], dy = rr[1], dz = rr[2];
    return sqrtf((dx * dx + dy * dy) + dz * dz);
}

__global__ void __launch_bounds__(WAVE * WAVES_PER_BLOCK)
k_composite_fwd(int64_t n, int S, const float* __restrict__ raw, const float* __restrict__ z,
                const float* __restrict__ rays, int ld, const float* __restrict__ noise, int white,
                float* __restrict__ rgb, float* __restrict__ disp, float* __restrict__ acc,
                float* __restrict__ weights, float* __restrict__ depth) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WAVES_PER_BLOCK;
    // The work of a wave is a sequence of (ray, tile) items; the loads of the next item are issued before the
    // arithmetic of the current one, so two tiles per wave are in flight.
    const int ntiles = (S + 63) / 64;
    int64_t ray = wave0;
    int t = 0;
    Loaded cur{};
    float3 dir = make_float3(0.f, 0.f, 0.f);
    if (ray < n) {
        cur = fetch_tile(raw, z, noise, ray * S, 0, lane, S);
        const float* rr = rays + ray * ld;
        dir = make_float3(rr[0], rr[1], rr[2]);
    }
    double carry = 1.0;  // product of keep over all earlier tiles of the ray
    float sr = 0, sg = 0, sb = 0, sd = 0, sa = 0;   // this lane's share of the ray sums (fp32, like the reference's)
    float dnorm = 0.0f;
    while (ray < n) {
        int64_t nray = ray;
        int nt = t + 1;
        if (nt == ntiles) { nt = 0; nray = ray + nwaves; }
        Loaded nxt{};
        float3 ndir = dir;
        if (nray < n) {
            nxt = fetch_tile(raw, z, noise, nray * S, nt * 64, lane, S);
            if (nt == 0) {
                const float* rr = rays + nray * ld;
                ndir = make_float3(rr[0], rr[1], rr[2]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const int64_t base = ray * S;
        if (t == 0) {
            dnorm = sqrtf((dir.x * dir.x + dir.y * dir.y) + dir.z * dir.z);
            carry = 1.0;
            sr = sg = sb = sd = sa = 0.0f;
        }
        {
            const int s = t * 64 + lane;
            const bool live = s < S;
            const Sample q = make_sample(cur, noise != nullptr, t * 64, lane, S, dnorm);
            const double keep = (double)q.keep;
            const double incl = wave_scan_f64<OpMul>(keep);
            const double excl = dpp_f64<0x138, 0xf, 0xf>(1.0, incl);   // wave_shr:1, lane 0 keeps the identity
            const float T = (float)(carry * excl);
            carry *= lane_f64(incl, 63);
            // the reference rounds the running product to fp32 per element but keeps the fp64 value running
            if (live) {
                const float w = q.alpha * T;
                weights[base + s] = w;
                sr = sr + w * sigmoidf(q.raw.x);
                sg = sg + w * sigmoidf(q.raw.y);
                sb = sb + w * sigmoidf(q.raw.z);
                sd = sd + w * q.z;
                sa = sa + w;
            }
        }
        if (t == ntiles - 1) {
            const float a = wave_sum_f32(sa), d = wave_sum_f32(sd);
            float r = wave_sum_f32(sr), g = wave_sum_f32(sg), b = wave_sum_f32(sb);
            if (lane == 0) {
                if (white) {
                    const float bg = 1.0f - a;
                    r = r + bg; g = g + bg; b = b + bg;
                }
                rgb[ray * 3 + 0] = r; rgb[ray * 3 + 1] = g; rgb[ray * 3 + 2] = b;
                acc[ray] = a;
                depth[ray] = d;
                const float q = d / a;
                disp[ray] = 1.0f / ((q != q) ? q : (q > 1e-10f ? q : 1e-10f));
            }
        }
        cur = nxt; dir = ndir; ray = nray; t = nt;
    }
}

// Backward.  With w_i = alpha_i T_i, T_i = prod_{j<i} keep_j and G_i = dL/dw_i:
//   dL/dalpha_i = G_i T_i - (sum_{j>i} G_j w_j) / keep_i,   dalpha_i/dsigma_i = dist_i (1 - alpha_i) [sigma_i > 0],
//   dL/draw_c  = w_i g_rgb[c] s_c (1 - s_c).
__global__ void __launch_bounds__(WAVE * WAVES_PER_BLOCK)
k_composite_bwd(int64_t n, int S, const float* __restrict__ raw, const float* __restrict__ z,
                const float* __restrict__ rays, int ld, const float* __restrict__ noise, int white,
                const float* __restrict__ weights, const float* __restrict__ g_rgb, const float* __restrict__ g_disp,
                const float* __restrict__ g_acc, const float* __restrict__ g_weights,
                const float* __restrict__ g_depth, float* __restrict__ d_raw) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WAVES_PER_BLOCK;
    const int ntiles = (S + 63) / 64;
    __shared__ double s_carry[WAVES_PER_BLOCK][MAX_TILES];
    for (int64_t ray = wave0; ray < n; ray += nwaves) {
        const float dnorm = ray_norm(rays + ray * ld);
        const int64_t base = ray * S;
        float gr = 0, gg = 0, gb = 0, ga = 0, gd = 0;
        if (g_rgb) { gr = g_rgb[ray * 3 + 0]; gg = g_rgb[ray * 3 + 1]; gb = g_rgb[ray * 3 + 2]; }
        if (g_acc) ga = g_acc[ray];
        if (g_depth) gd = g_depth[ray];
        if (white) ga = ga - ((gr + gg) + gb);
        if (g_disp) {
            // disp = 1 / max(1e-10, depth/acc): needs the forward sums again
            float sd = 0, sa = 0;
            for (int s = lane; s < S; s += 64) {
                const float w = weights[base + s];
                sd = sd + w * z[base + s];
                sa = sa + w;
            }
            const float d = wave_sum_f32(sd), a = wave_sum_f32(sa);
            const float q = d / a;
            if (q > 1e-10f) {
                const float gq = -g_disp[ray] / (q * q);
                gd = gd + gq / a;
                ga = ga - gq * q / a;
            }
        }
        // carry-in transmittance of every tile, rebuilt exactly as the forward did
        double* carry_in = s_carry[threadIdx.x >> 6];
        {
            double carry = 1.0;
            for (int t = 0; t < ntiles; ++t) {
                if (lane == 0) carry_in[t] = carry;
                if (t == ntiles - 1) break;
                const double keep = (double)load_sample(raw, z, noise, base, t * 64, lane, S, dnorm).keep;
                carry *= wave_reduce_f64<OpMul>(keep);
            }
        }
        __builtin_amdgcn_wave_barrier();
        double suffix = 0.0;  // sum of G_j w_j over all later tiles
        for (int t = ntiles - 1; t >= 0; --t) {
            const int s = t * 64 + lane;
            const bool live = s < S;
            const Sample q = load_sample(raw, z, noise, base, t * 64, lane, S, dnorm);
            float w = 0.0f, G = 0.0f, sr = 0, sg = 0, sb = 0;
            const double keep = (double)q.keep;
            if (live) {
                w = weights[base + s];
                sr = sigmoidf(q.raw.x); sg = sigmoidf(q.raw.y); sb = sigmoidf(q.raw.z);
                G = (((gr * sr + gg * sg) + gb * sb) + gd * q.z) + ga;
                if (g_weights) G = G + g_weights[base + s];
            }
            const double incl_p = wave_scan_f64<OpMul>(keep);
            const double excl_p = dpp_f64<0x138, 0xf, 0xf>(1.0, incl_p);   // wave_shr:1
            const float T = (float)(carry_in[t] * excl_p);
            // sum of G_j w_j over the later samples = tile total - inclusive prefix (fp64: the cancellation is far
            // below the fp32 result's resolution) + everything in later tiles
            const double gw = (double)G * (double)w;
            const double incl = wave_scan_f64<OpAdd>(gw);
            const double total = lane_f64(incl, 63);
            const double after = (total - incl) + suffix;
            suffix += total;
            if (live) {
                float4 o;
                o.x = w * gr * (sr * (1.0f - sr));
                o.y = w * gg * (sg * (1.0f - sg));
                o.z = w * gb * (sb * (1.0f - sb));
                float dsig = 0.0f;
                if (q.sig > 0.0f) {
                    const float dalpha = (float)((double)G * (double)T - after / (double)q.keep);
                    dsig = dalpha * (q.dist * (1.0f - q.alpha));
                }
                o.w = dsig;
                reinterpret_cast<float4*>(d_raw)[base + s] = o;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

inline int grid_rays(int64_t n) {
    int64_t g = (n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    const int64_t cap = 256 * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" {

int nerf_composite_fwd(int64_t n, int S, const float* raw, const float* z, const float* rays, int ray_stride,
                       const float* noise, int white_bkgd, float* rgb, float* disp, float* acc, float* weights,
                       float* depth, void* stream) {
    NERF_REQUIRE(n >= 0 && S >= 1, "nerf_composite_fwd: n=%lld S=%d", (long long)n, S);
    NERF_REQUIRE(ray_stride >= 3, "nerf_composite_fwd: dir_stride=%d < 3", ray_stride);
    if (n == 0) return 0;
    NERF_REQUIRE(raw && z && rays && rgb && disp && acc && weights && depth, "nerf_composite_fwd: null pointer");
    NERF_REQUIRE((reinterpret_cast<uintptr_t>(raw) & 15) == 0, "nerf_composite_fwd: raw must be 16-byte aligned");
    hipLaunchKernelGGL(k_composite_fwd, dim3(grid_rays(n)), dim3(WAVE * WAVES_PER_BLOCK), 0, as_stream(stream), n, S,
                       raw, z, rays, ray_stride, noise, white_bkgd, rgb, disp, acc, weights, depth);
    return launched("nerf_composite_fwd");
}

int nerf_composite_bwd(int64_t n, int S, const float* raw, const float* z, const float* rays, int ray_stride,
                       const float* noise, int white_bkgd, const float* weights, const float* g_rgb,
                       const float* g_disp, const float* g_acc, const float* g_weights, const float* g_depth,
                       float* d_raw, void* stream) {
    NERF_REQUIRE(n >= 0 && S >= 1 && S <= 64 * MAX_TILES, "nerf_composite_bwd: n=%lld S=%d (S <= %d)", (long long)n, S,
                 64 * MAX_TILES);
    NERF_REQUIRE(ray_stride >= 3, "nerf_composite_bwd: dir_stride=%d < 3", ray_stride);
    if (n == 0) return 0;
    NERF_REQUIRE(raw && z && rays && weights && d_raw, "nerf_composite_bwd: null pointer");
    NERF_REQUIRE(((reinterpret_cast<uintptr_t>(raw) | reinterpret_cast<uintptr_t>(d_raw)) & 15) == 0,
                 "nerf_composite_bwd: raw and d_raw must be 16-byte aligned");
    hipLaunchKernelGGL(k_composite_bwd, dim3(grid_rays(n)), dim3(WAVE * WAVES_PER_BLOCK), 0, as_stream(stream), n, S,
                       raw, z, rays, ray_stride, noise, white_bkgd, weights, g_rgb, g_disp, g_acc, g_weights, g_depth,
                       d_raw);
    return launched("nerf_composite_bwd");
}

}  // extern "C"
