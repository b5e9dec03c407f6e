// Ray generation, NDC warp, stratified depth sampling and the stand-alone positional encoding.
//
// All four are streaming kernels bounded by HBM: one thread per output element (or per ray where
// the per-ray inputs dominate), consecutive lanes on consecutive addresses.  Arithmetic follows the
// upstream elementwise op order with one fp32 rounding per op (no FMA contraction), so results are
// bit-identical to the PyTorch CPU evaluation wherever libm differences cannot enter.
#include "common.hpp"
#pragma clang fp contract(off)

using namespace nerfhip;

namespace {

// run_nerf_helpers.py:169-196.  dirs = [(i-cx)/fx, -(j-cy)/fy, -1]; d = sum_k dirs[k] * c2w[r][k].
// IDX = uint32_t whenever the element count allows it: the index divisions are a fifth of the 64-bit ones.
template <typename IDX>
__global__ void __launch_bounds__(256) k_get_rays(int H, int W, float fx, float fy, float cx, float cy,
                                                  const float* __restrict__ c2w, float* __restrict__ rays_o,
                                                  float* __restrict__ rays_d) {
    const IDX total = (IDX)H * (IDX)W * 3;
    // one thread per output float: stores of a wave are 256 contiguous bytes in each array
    for (IDX e = (IDX)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (IDX)gridDim.x * blockDim.x) {
        const IDX pix = e / 3;
        const int r = (int)(e - pix * 3);
        const int j = (int)(pix / (IDX)W), i = (int)(pix - (IDX)j * (IDX)W);
        const float dx = ((float)i - cx) / fx;
        const float dy = -(((float)j - cy) / fy);
        const float dz = -1.0f;
        const float m0 = c2w[r * 4 + 0], m1 = c2w[r * 4 + 1], m2 = c2w[r * 4 + 2];
        rays_d[e] = (dx * m0 + dy * m1) + dz * m2;
        rays_o[e] = c2w[r * 4 + 3];
    }
}

// The ray batch of one optimisation step taken straight from the camera (run_nerf.py:833-854): ray k goes
// through pixel (row0 + q / win_cols, col0 + q % win_cols), q = pick[k], of a win_rows x win_cols window of
// the image (the whole image, or the central crop of the first iterations).  Same arithmetic as k_get_rays,
// so the rays equal get_rays(...)[row, col] bit for bit; the target colour is gathered in the same pass.
__global__ void __launch_bounds__(256) k_rays_at(int64_t n, const int64_t* __restrict__ pick, int W, int row0, int col0,
                                                 int win_cols, float fx, float fy, float cx, float cy,
                                                 const float* __restrict__ c2w, const float* __restrict__ image,
                                                 float* __restrict__ rays_o, float* __restrict__ rays_d,
                                                 float* __restrict__ target) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n * 3; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = e / 3;
        const int r = (int)(e - k * 3);
        const int64_t q = pick[k];
        const int j = row0 + (int)(q / win_cols), i = col0 + (int)(q % win_cols);
        const float dx = ((float)i - cx) / fx;
        const float dy = -(((float)j - cy) / fy);
        const float dz = -1.0f;
        const float m0 = c2w[r * 4 + 0], m1 = c2w[r * 4 + 1], m2 = c2w[r * 4 + 2];
        rays_d[e] = (dx * m0 + dy * m1) + dz * m2;
        rays_o[e] = c2w[r * 4 + 3];
        if (image) target[e] = image[((int64_t)j * W + i) * 3 + r];
    }
}

// Distinct pseudo-random pixels without a sort: k -> P(k), k < n, where P is a keyed permutation of [0, n_pix) -- a
// four-round Feistel network on the 2h-bit numbers (4^h >= n_pix > 4^(h-1)), walked along its cycle until the value
// is inside the range again (a permutation of the larger set restricted this way is a permutation of the smaller one;
// under four tries on average).  Stands in for np.random.choice(n_pix, n, replace=False) of run_nerf.py:848.
struct PickKey {
    uint32_t k[4];
    int h;
};
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint64_t keyed_pixel(uint64_t k, uint64_t n_pix, const PickKey& key) {
    const uint32_t mask = (1u << key.h) - 1u;
    uint64_t x = k;
    do {
        uint32_t l = (uint32_t)(x >> key.h) & mask, r = (uint32_t)x & mask;
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            const uint32_t f = mix32(r * 0x9e3779b1u + key.k[round]) & mask;
            const uint32_t t = l ^ f;
            l = r;
            r = t;
        }
        x = ((uint64_t)l << key.h) | r;
    } while (x >= n_pix);
    return x;
}

// k_rays_at with the pixels drawn inside the kernel (keyed_pixel); pick_out (optional) receives them.
__global__ void __launch_bounds__(256) k_rays_random(int64_t n, PickKey key, int64_t n_pix, int W, int row0, int col0,
                                                     int win_cols, float fx, float fy, float cx, float cy,
                                                     const float* __restrict__ c2w, const float* __restrict__ image,
                                                     float* __restrict__ rays_o, float* __restrict__ rays_d,
                                                     float* __restrict__ target, int64_t* __restrict__ pick_out) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n * 3; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = e / 3;
        const int r = (int)(e - k * 3);
        const int64_t q = (int64_t)keyed_pixel((uint64_t)k, (uint64_t)n_pix, key);
        if (pick_out && r == 0) pick_out[k] = q;
        const int j = row0 + (int)(q / win_cols), i = col0 + (int)(q % win_cols);
        const float dx = ((float)i - cx) / fx;
        const float dy = -(((float)j - cy) / fy);
        const float dz = -1.0f;
        const float m0 = c2w[r * 4 + 0], m1 = c2w[r * 4 + 1], m2 = c2w[r * 4 + 2];
        rays_d[e] = (dx * m0 + dy * m1) + dz * m2;
        rays_o[e] = c2w[r * 4 + 3];
        if (image) target[e] = image[((int64_t)j * W + i) * 3 + r];
    }
}

// run_nerf_helpers.py:230-255.
__global__ void __launch_bounds__(256) k_ndc(int64_t n, float sx, float sy, float near, const float* o_in,
                                             const float* d_in, float* o_out, float* d_out) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        float ox = o_in[p * 3 + 0], oy = o_in[p * 3 + 1], oz = o_in[p * 3 + 2];
        const float dx = d_in[p * 3 + 0], dy = d_in[p * 3 + 1], dz = d_in[p * 3 + 2];
        const float t = -(near + oz) / dz;
        ox = ox + t * dx;
        oy = oy + t * dy;
        oz = oz + t * dz;
        const float o0 = sx * ox / oz;
        const float o1 = sy * oy / oz;
        const float o2 = 1.0f + (2.0f * near) / oz;
        const float d0 = sx * (dx / dz - ox / oz);
        const float d1 = sy * (dy / dz - oy / oz);
        const float d2 = (-2.0f * near) / oz;
        o_out[p * 3 + 0] = o0; o_out[p * 3 + 1] = o1; o_out[p * 3 + 2] = o2;
        d_out[p * 3 + 0] = d0; d_out[p * 3 + 1] = d1; d_out[p * 3 + 2] = d2;
    }
}

// run_nerf.py:419-453.  One thread per (ray, sample).
template <typename IDX>
__global__ void __launch_bounds__(256) k_sample_coarse(int64_t n, int S, const float* __restrict__ rays, int ld,
                                                       const float* __restrict__ t_vals,
                                                       const float* __restrict__ t_rand, int lindisp,
                                                       float* __restrict__ z_out, float* __restrict__ pts) {
    const IDX total = (IDX)n * (IDX)S;
    for (IDX e = (IDX)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (IDX)gridDim.x * blockDim.x) {
        const IDX ray = e / (IDX)S;
        const int s = (int)(e - ray * (IDX)S);
        const float* rr = rays + (int64_t)ray * ld;
        const float near = rr[6], far = rr[7];
        auto depth = [&](int k) -> float {
            const float t = t_vals[k];
            if (!lindisp) return near * (1.0f - t) + far * t;
            return 1.0f / ((1.0f / near) * (1.0f - t) + (1.0f / far) * t);
        };
        float z = depth(s);
        if (t_rand) {
            const float lo = (s == 0) ? z : 0.5f * (z + depth(s - 1));
            const float hi = (s == S - 1) ? z : 0.5f * (depth(s + 1) + z);
            z = lo + (hi - lo) * t_rand[e];
        }
        z_out[e] = z;
        if (pts) {
            float* pp = pts + (int64_t)e * 3;
            pp[0] = rr[0] + rr[3] * z;
            pp[1] = rr[1] + rr[4] * z;
            pp[2] = rr[2] + rr[5] * z;
        }
    }
}

// run_nerf_helpers.py:13-66.  One thread per output element; frequencies are exact powers of two.
__global__ void __launch_bounds__(256) k_encode(int64_t n, int L, const float* __restrict__ x, float* __restrict__ out) {
    const int width = 3 + 6 * L;
    const int64_t total = n * width;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = e / width;
        const int c = (int)(e - p * width);
        float v;
        if (c < 3) {
            v = x[p * 3 + c];
        } else {
            const int k = (c - 3) / 6, w = (c - 3) - k * 6;
            const float arg = x[p * 3 + (w % 3)] * (float)(1 << k);
            v = (w < 3) ? sinf(arg) : cosf(arg);
        }
        out[e] = v;
    }
}

// run_nerf_helpers.py:13-47 with any of its keyword settings whose periodic functions are sines and cosines:
// out = [x (include_input)] ++ [fn_j(x * freq_b) for b < n_freq for j < n_fn], each block d wide.  freqs holds the
// reference's own fp32 band values (it evaluates 2 ** linspace or linspace on the CPU); bit j of cos_mask says that
// function j is the cosine.
__global__ void __launch_bounds__(256) k_encode_general(int64_t n, int d, int include_input, int n_freq,
                                                        const float* __restrict__ freqs, int n_fn, unsigned cos_mask,
                                                        const float* __restrict__ x, float* __restrict__ out) {
    const int lead = include_input ? d : 0;
    const int width = lead + n_freq * n_fn * d;
    const int64_t total = n * width;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = e / width;
        const int c = (int)(e - p * width);
        float v;
        if (c < lead) {
            v = x[p * d + c];
        } else {
            const int q = c - lead, band = q / (n_fn * d), w = q - band * (n_fn * d), fn = w / d, dim = w - fn * d;
            const float arg = x[p * d + dim] * freqs[band];
            v = ((cos_mask >> fn) & 1u) ? cosf(arg) : sinf(arg);
        }
        out[e] = v;
    }
}

// run_nerf.py:120-149: unit view directions (from the un-warped directions), optional NDC warp, and
// the packed record [o d near far (viewdir)].  One thread per ray.
__global__ void __launch_bounds__(256) k_pack_rays(int64_t n, const float* __restrict__ o_in, const float* __restrict__ d_in,
                                                   const float* __restrict__ vd_src, int ndc, float sx, float sy,
                                                   float ndc_near, float near, float far,
                                                   const float* __restrict__ near_v, const float* __restrict__ far_v,
                                                   float* __restrict__ out, int ld) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        float ox = o_in[p * 3 + 0], oy = o_in[p * 3 + 1], oz = o_in[p * 3 + 2];
        float dx = d_in[p * 3 + 0], dy = d_in[p * 3 + 1], dz = d_in[p * 3 + 2];
        float* r = out + p * ld;
        if (vd_src) {
            const float vx = vd_src[p * 3 + 0], vy = vd_src[p * 3 + 1], vz = vd_src[p * 3 + 2];
            const float nrm = sqrtf((vx * vx + vy * vy) + vz * vz);
            r[8] = vx / nrm; r[9] = vy / nrm; r[10] = vz / nrm;
        }
        if (ndc) {
            const float t = -(ndc_near + oz) / dz;
            ox = ox + t * dx; oy = oy + t * dy; oz = oz + t * dz;
            const float o0 = sx * ox / oz, o1 = sy * oy / oz, o2 = 1.0f + (2.0f * ndc_near) / oz;
            const float d0 = sx * (dx / dz - ox / oz), d1 = sy * (dy / dz - oy / oz), d2 = (-2.0f * ndc_near) / oz;
            ox = o0; oy = o1; oz = o2; dx = d0; dy = d1; dz = d2;
        }
        r[0] = ox; r[1] = oy; r[2] = oz; r[3] = dx; r[4] = dy; r[5] = dz;
        r[6] = near_v ? near_v[p] : near;
        r[7] = far_v ? far_v[p] : far;
    }
}

inline int grid_for(int64_t total, int block = 256) {
    int64_t g = (total + block - 1) / block;
    const int64_t cap = 256 * 8;  // 8 blocks per CU, grid-stride beyond that
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" {

int nerf_get_rays(int H, int W, float fx, float fy, float cx, float cy, const float* c2w, float* rays_o,
                  float* rays_d, void* stream) {
    NERF_REQUIRE(H > 0 && W > 0, "nerf_get_rays: H=%d W=%d", H, W);
    NERF_REQUIRE(c2w && rays_o && rays_d, "nerf_get_rays: null pointer");
    NERF_REQUIRE(fx != 0.0f && fy != 0.0f, "nerf_get_rays: zero focal length");
    const int64_t total = (int64_t)H * W * 3;
    if (total < (int64_t)0x7fffffff)
        hipLaunchKernelGGL(k_get_rays<uint32_t>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), H, W, fx, fy, cx, cy,
                           c2w, rays_o, rays_d);
    else
        hipLaunchKernelGGL(k_get_rays<int64_t>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), H, W, fx, fy, cx, cy,
                           c2w, rays_o, rays_d);
    return launched("nerf_get_rays");
}

int nerf_rays_at(int64_t n, const int64_t* pick, int H, int W, int row0, int col0, int win_rows, int win_cols, float fx,
                 float fy, float cx, float cy, const float* c2w, const float* image, float* rays_o, float* rays_d,
                 float* target, void* stream) {
    NERF_REQUIRE(n >= 0, "nerf_rays_at: n=%lld", (long long)n);
    NERF_REQUIRE(H > 0 && W > 0 && win_rows > 0 && win_cols > 0 && row0 >= 0 && col0 >= 0 && row0 + win_rows <= H &&
                 col0 + win_cols <= W, "nerf_rays_at: window %dx%d at (%d,%d) does not fit a %dx%d image", win_rows,
                 win_cols, row0, col0, H, W);
    NERF_REQUIRE(fx != 0.0f && fy != 0.0f, "nerf_rays_at: zero focal length");
    if (n == 0) return 0;
    NERF_REQUIRE(pick && c2w && rays_o && rays_d, "nerf_rays_at: null pointer");
    NERF_REQUIRE(!image || target, "nerf_rays_at: image given without a target buffer");
    hipLaunchKernelGGL(k_rays_at, dim3(grid_for(n * 3)), dim3(256), 0, as_stream(stream), n, pick, W, row0, col0, win_cols,
                       fx, fy, cx, cy, c2w, image, rays_o, rays_d, target);
    return launched("nerf_rays_at");
}

int nerf_rays_random(int64_t n, uint64_t seed, int H, int W, int row0, int col0, int win_rows, int win_cols, float fx,
                     float fy, float cx, float cy, const float* c2w, const float* image, float* rays_o, float* rays_d,
                     float* target, int64_t* pick_out, void* stream) {
    NERF_REQUIRE(n >= 0, "nerf_rays_random: n=%lld", (long long)n);
    NERF_REQUIRE(H > 0 && W > 0 && win_rows > 0 && win_cols > 0 && row0 >= 0 && col0 >= 0 && row0 + win_rows <= H &&
                 col0 + win_cols <= W, "nerf_rays_random: window %dx%d at (%d,%d) does not fit a %dx%d image", win_rows,
                 win_cols, row0, col0, H, W);
    const int64_t n_pix = (int64_t)win_rows * win_cols;
    NERF_REQUIRE(n <= n_pix, "nerf_rays_random: %lld distinct pixels asked of a window of %lld", (long long)n, (long long)n_pix);
    NERF_REQUIRE(fx != 0.0f && fy != 0.0f, "nerf_rays_random: zero focal length");
    if (n == 0) return 0;
    NERF_REQUIRE(c2w && rays_o && rays_d, "nerf_rays_random: null pointer");
    NERF_REQUIRE(!image || target, "nerf_rays_random: image given without a target buffer");
    PickKey key;
    key.h = 1;
    while (((int64_t)1 << (2 * key.h)) < n_pix) ++key.h;        // 4^h >= n_pix, h <= 31
    NERF_REQUIRE(key.h <= 31, "nerf_rays_random: window too large");
    uint64_t sm = seed;
    for (int r = 0; r < 4; ++r) {                                // splitmix64: four round keys out of the seed
        sm += 0x9e3779b97f4a7c15ull;
        uint64_t z = sm;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        key.k[r] = (uint32_t)((z ^ (z >> 31)) >> 16);
    }
    hipLaunchKernelGGL(k_rays_random, dim3(grid_for(n * 3)), dim3(256), 0, as_stream(stream), n, key, n_pix, W, row0, col0,
                       win_cols, fx, fy, cx, cy, c2w, image, rays_o, rays_d, target, pick_out);
    return launched("nerf_rays_random");
}

int nerf_ndc_rays(int64_t n, float sx, float sy, float near, const float* o_in, const float* d_in, float* o_out,
                  float* d_out, void* stream) {
    NERF_REQUIRE(n >= 0, "nerf_ndc_rays: n=%lld", (long long)n);
    if (n == 0) return 0;
    NERF_REQUIRE(o_in && d_in && o_out && d_out, "nerf_ndc_rays: null pointer");
    hipLaunchKernelGGL(k_ndc, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), n, sx, sy, near, o_in, d_in, o_out,
                       d_out);
    return launched("nerf_ndc_rays");
}

int nerf_pack_rays(int64_t n, const float* rays_o, const float* rays_d, const float* viewdir_src, int ndc, float sx,
                   float sy, float ndc_near, float near, float far, const float* near_v, const float* far_v,
                   float* rays, int ray_stride, void* stream) {
    NERF_REQUIRE(n >= 0, "nerf_pack_rays: n=%lld", (long long)n);
    NERF_REQUIRE(ray_stride >= (viewdir_src ? 11 : 8), "nerf_pack_rays: ray_stride=%d too small", ray_stride);
    if (n == 0) return 0;
    NERF_REQUIRE(rays_o && rays_d && rays, "nerf_pack_rays: null pointer");
    hipLaunchKernelGGL(k_pack_rays, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), n, rays_o, rays_d, viewdir_src, ndc,
                       sx, sy, ndc_near, near, far, near_v, far_v, rays, ray_stride);
    return launched("nerf_pack_rays");
}

int nerf_sample_coarse(int64_t n, int S, const float* rays, int ray_stride, const float* t_vals, const float* t_rand,
                       int lindisp, float* z, float* pts, void* stream) {
    NERF_REQUIRE(n >= 0 && S >= 1, "nerf_sample_coarse: n=%lld S=%d", (long long)n, S);
    NERF_REQUIRE(ray_stride >= 8, "nerf_sample_coarse: ray_stride=%d < 8", ray_stride);
    if (n == 0) return 0;
    NERF_REQUIRE(rays && t_vals && z, "nerf_sample_coarse: null pointer");
    if (n * S < (int64_t)0x7fffffff)
        hipLaunchKernelGGL(k_sample_coarse<uint32_t>, dim3(grid_for(n * S)), dim3(256), 0, as_stream(stream), n, S, rays,
                           ray_stride, t_vals, t_rand, lindisp, z, pts);
    else
        hipLaunchKernelGGL(k_sample_coarse<int64_t>, dim3(grid_for(n * S)), dim3(256), 0, as_stream(stream), n, S, rays,
                           ray_stride, t_vals, t_rand, lindisp, z, pts);
    return launched("nerf_sample_coarse");
}

int nerf_encode(int64_t n, int L, const float* x, float* out, void* stream) {
    NERF_REQUIRE(n >= 0 && L >= 0 && L <= 24, "nerf_encode: n=%lld L=%d", (long long)n, L);
    if (n == 0) return 0;
    NERF_REQUIRE(x && out, "nerf_encode: null pointer");
    hipLaunchKernelGGL(k_encode, dim3(grid_for(n * (3 + 6 * L))), dim3(256), 0, as_stream(stream), n, L, x, out);
    return launched("nerf_encode");
}

int nerf_encode_general(int64_t n, int d, int include_input, int n_freq, const float* freqs, int n_fn, unsigned cos_mask,
                        const float* x, float* out, void* stream) {
    NERF_REQUIRE(n >= 0 && d >= 1 && n_freq >= 0 && n_fn >= 0 && n_fn <= 32, "nerf_encode_general: n=%lld d=%d n_freq=%d n_fn=%d",
                 (long long)n, d, n_freq, n_fn);
    const int64_t width = (include_input ? d : 0) + (int64_t)n_freq * n_fn * d;
    NERF_REQUIRE(width <= 0x7fffffff, "nerf_encode_general: encoding too wide");
    if (n == 0 || width == 0) return 0;
    NERF_REQUIRE(x && out && (freqs || n_freq * n_fn == 0), "nerf_encode_general: null pointer");
    hipLaunchKernelGGL(k_encode_general, dim3(grid_for(n * width)), dim3(256), 0, as_stream(stream), n, d, include_input, n_freq,
                       freqs, n_fn, cos_mask, x, out);
    return launched("nerf_encode_general");
}

}  // extern "C"
