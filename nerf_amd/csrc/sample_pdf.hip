// Hierarchical (inverse-CDF) depth sampling fused with the merge of coarse and fine depths.
//
// One wavefront per ray; the ray's coarse depths, bin mid-points, cdf and the merged depth list
// live in LDS (about 8*(Sc+Si) bytes per wave).  Stages, all inside the wave:
//   1. weights+1e-5 and their sum, in the exact association order of ATen's vectorised fp32 sum
//      (8 interleaved partial sums with a 4-way cascade), so pdf = w/sum has the reference's bits;
//   2. cdf = inclusive prefix sum carried in fp64 (64-lane scan + carry per tile).  Every partial
//      sum of fp32 pdf values >= 2^-27 that totals <= 1 is exactly representable in fp64, so the
//      scan order cannot change a bit relative to the sequential fp64 accumulation of ATen's cumsum;
//   3. per uniform: binary search (searchsorted, right=True), clamp, gather, lerp, in fp32 with
//      one rounding per op;
//   4. z_std in fp64;
//   5. sort(cat(coarse, fine)) as a merge: both lists are checked for order with one ballot each (the
//      coarse depths always are, the fine ones are whenever the uniforms are, i.e. in every
//      deterministic render), an unordered list goes through an in-LDS bitonic sort of (order key,
//      index) pairs, and every element then finds its output slot by one binary search in the other
//      list; coarse elements precede equal fine ones, NaN sorts last: the order of a stable sort of
//      the concatenation.  Coalesced write-out of z and of the sample positions o + d*z.
// HBM traffic per ray: read 8*Sc (+4*Si uniforms), write 4*(Sc+Si) + 4 (+12*(Sc+Si) positions).
#include "common.hpp"
#include "wave_ops.hpp"
#pragma clang fp contract(off)

using namespace nerfhip;

namespace {

constexpr int WAVES_PER_BLOCK = 4;

__device__ __forceinline__ double wave_sum_d(double v) { return wave_reduce_f64<OpAdd>(v); }

// ATen row_sum over `count` items x[(r)*stride + k], r = 0..count-1 (aten/src/ATen/native/cpu/SumKernel.cpp:
// 4 interleaved accumulators, each a 4-level cascade with level step >= 16).
__device__ float cascade_row_sum(const float* x, int stride, int k, int count) {
    const int size_ilp = count >> 2;
    int clog = 0;
    while ((1 << clog) < size_ilp) ++clog;
    const int level_power = (clog / 4) > 4 ? (clog / 4) : 4;
    const int level_step = 1 << level_power;
    const int level_mask = level_step - 1;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0f;
    int i = 0;
    while (i + level_step <= size_ilp) {
        for (int j = 0; j < level_step; ++j, ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[0][q] = acc[0][q] + x[(i * 4 + q) * stride + k];
        }
#pragma unroll
        for (int j = 1; j < 4; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc[j][q] = acc[j][q] + acc[j - 1][q]; acc[j - 1][q] = 0.0f; }
            if ((i & (level_mask << (j * level_power))) != 0) break;
        }
    }
    for (; i < size_ilp; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[0][q] = acc[0][q] + x[(i * 4 + q) * stride + k];
    }
#pragma unroll
    for (int j = 1; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[0][q] = acc[0][q] + acc[j][q];
    for (int r = size_ilp * 4; r < count; ++r) acc[0][0] = acc[0][0] + x[r * stride + k];
#pragma unroll
    for (int q = 1; q < 4; ++q) acc[0][0] = acc[0][0] + acc[0][q];
    return acc[0][0];
}

// Order-preserving map float -> uint32: ascending, -0 == +0, every NaN last (and equal to every other NaN).
__device__ __forceinline__ uint32_t order_key(float f) {
    if (f != f) return 0xffffffffu;
    if (f == 0.0f) return 0x80000000u;
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__host__ __device__ inline int pow2_at_least(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

__host__ __device__ inline int pow2_at_most(int n) {
    int p = 1;
    while (2 * p <= n) p <<= 1;
    return p;
}

// Length of the longest prefix of an ordered list of L >= 1 items whose items all satisfy `holds(index)`, for a
// condition that is true on a prefix and false after it.  A fixed number of halving steps (floor(log2 L) + 1) with no
// lane-dependent control flow: every lane probes in every step, a probe past the end re-reads the last item and is
// ignored.
template <typename Holds>
__device__ __forceinline__ int prefix_length(int L, Holds holds) {
    int pos = 0;
    for (int step = pow2_at_most(L); step >= 1; step >>= 1) {
        const int probe = pos + step;
        const bool inside = probe <= L;
        const bool ok = holds((inside ? probe : L) - 1);
        pos = (inside && ok) ? probe : pos;
    }
    return pos;
}

// keys[i] = (order key << 32) | index for i < count, all-ones padding up to P; returns whether the list is ordered
__device__ __forceinline__ bool build_keys(uint64_t* keys, const float* vals, int first_index, int count, int P, int lane) {
    for (int i = lane; i < P; i += 64)
        keys[i] = i < count ? (((uint64_t)order_key(vals[i]) << 32) | (uint32_t)(first_index + i)) : ~0ull;
    __builtin_amdgcn_wave_barrier();
    bool ok = true;
    for (int i = lane; i + 1 < count; i += 64) ok = ok && (keys[i] < keys[i + 1]);
    return __all(ok);
}

// in-LDS bitonic sort of P (a power of two) 64-bit keys by one wavefront
__device__ __forceinline__ void bitonic_sort(uint64_t* a, int P, int lane) {
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (P >> 1); t += 64) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const uint64_t x = a[i], y = a[l];
                if ((x > y) == ((i & k) == 0)) { a[i] = y; a[l] = x; }
            }
            __builtin_amdgcn_wave_barrier();
        }
}

// SC, SI > 0: the sample counts as compile-time constants (the shipped configurations: every loop unrolls and the
// trip counts, LDS offsets and search depths fold); SC = 0: taken from the arguments.
template <int SC, int SI>
__global__ void __launch_bounds__(WAVE * WAVES_PER_BLOCK)
k_sample_fine(int64_t n, int Sc_arg, int Si_arg, int bins_given, const float* __restrict__ z,
              const float* __restrict__ weights, const float* __restrict__ u, int u_shared,
              const float* __restrict__ rays, int ld,
              float* __restrict__ z_out, float* __restrict__ pts_out, float* __restrict__ z_std,
              float* __restrict__ samples, int64_t* __restrict__ inds_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int Sc = SC > 0 ? SC : Sc_arg, Si = SC > 0 ? SI : Si_arg;
    const int Sf = Sc + Si, nb = Sc - 1, nw = Sc - 2;
    const int Sf4 = (Sf + 3) & ~3, nb4 = (nb + 3) & ~3;
    float* zs = smem + (size_t)wv * (2 * Sf4 + 3 * nb4);  // merged (unsorted) depths
    float* zsorted = zs + Sf4;
    float* cdf = zsorted + Sf4;   // [nb]
    float* bins = cdf + nb4;      // [nb]
    float* wp = bins + nb4;       // [nw] weights + 1e-5, then pdf
    const int Pc = pow2_at_least(Sc), Ps = pow2_at_least(Si);
    uint64_t* kc = reinterpret_cast<uint64_t*>(smem + (size_t)WAVES_PER_BLOCK * (2 * Sf4 + 3 * nb4)) + (size_t)wv * (Pc + Ps);
    uint64_t* ks = kc + Pc;       // (order key, index) of the coarse / fine depths
    const int64_t wave0 = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wv;
    const int64_t nwaves = (int64_t)gridDim.x * WAVES_PER_BLOCK;
    const double inv_si = 1.0 / (double)Si;
    for (int64_t ray = wave0; ray < n; ray += nwaves) {
        // ---- stage inputs
        if (bins_given) {  // stand-alone sample_pdf: z holds the nb bin edges, weights the nw = nb-1 bin weights
            for (int s = lane; s < nb; s += 64) bins[s] = z[ray * nb + s];
            for (int s = lane; s < nw; s += 64) wp[s] = weights[ray * nw + s] + 1e-5f;
        } else {
            for (int s = lane; s < Sc; s += 64) zs[s] = z[ray * Sc + s];
            for (int s = lane; s < nw; s += 64) wp[s] = weights[ray * Sc + 1 + s] + 1e-5f;
            __builtin_amdgcn_wave_barrier();
            for (int s = lane; s < nb; s += 64) bins[s] = 0.5f * (zs[s + 1] + zs[s]);
        }
        // ---- sum of the weights in ATen's order
        float total;
        if (nw >= 8) {
            const int vec = nw >> 3;
            float part = 0.0f;
            if (lane < 8) part = cascade_row_sum(wp, 8, lane, vec);
            float fin = 0.0f;
            for (int e = vec * 8; e < nw; ++e) fin = fin + wp[e];
#pragma unroll
            for (int k = 0; k < 8; ++k) fin = fin + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(part), k));
            total = fin;
        } else {
            total = cascade_row_sum(wp, 1, 0, nw);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- pdf and cdf
        double carry = 0.0;
        if (lane == 0) cdf[0] = 0.0f;
        for (int s0 = 0; s0 < nw; s0 += 64) {
            const int s = s0 + lane;
            double p = 0.0;
            if (s < nw) p = (double)(wp[s] / total);
            const double incl = wave_scan_f64<OpAdd>(p);
            if (s < nw) cdf[s + 1] = (float)(carry + incl);
            carry += lane_f64(incl, 63);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- invert the cdf
        double su = 0.0;
        for (int j = lane; j < Si; j += 64) {
            const float uj = u_shared ? u[j] : u[ray * Si + j];
            const int lo = prefix_length(nb, [&](int k) { return !(cdf[k] > uj); });  // first index with cdf[idx] > uj
            const int below = lo - 1 > 0 ? lo - 1 : 0;
            const int above = lo < nb - 1 ? lo : nb - 1;
            const float cb = cdf[below], ca = cdf[above];
            const float bb = bins[below], ba = bins[above];
            float denom = ca - cb;
            if (denom < 1e-5f) denom = 1.0f;
            const float t = (uj - cb) / denom;
            const float smp = bb + t * (ba - bb);
            zs[Sc + j] = smp;
            su += (double)smp;
            if (samples) samples[ray * Si + j] = smp;
            if (inds_out) inds_out[ray * Si + j] = (int64_t)lo;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- population standard deviation of the new samples
        if (z_std) {
            const double mean = wave_sum_d(su) * inv_si;
            double m2 = 0.0;
            for (int j = lane; j < Si; j += 64) {
                const double dv = (double)zs[Sc + j] - mean;
                m2 += dv * dv;
            }
            m2 = wave_sum_d(m2);
            // (the fp64 mean square is rounded once and its root taken in fp32: within an ulp of the fp64 root, a
            // tenth of the instructions; the reference works in fp32 throughout)
            if (lane == 0) z_std[ray] = sqrtf((float)(m2 * inv_si));
        }
        if (bins_given) { __builtin_amdgcn_wave_barrier(); continue; }
        // ---- sort(cat(coarse, fine)): order each list if it is not ordered yet, then merge by rank
        // Both lists already in order and free of NaN (every deterministic render): the ranks come from float
        // comparisons on the depths themselves -- no keys, 4-byte reads.  -0 == +0 in both formulations.
        bool ordered = true;
        for (int i = lane; i + 1 < Sc; i += 64) ordered = ordered && (zs[i] <= zs[i + 1]);
        for (int i = lane; i + 1 < Si; i += 64) ordered = ordered && (zs[Sc + i] <= zs[Sc + i + 1]);
        if (lane == 0) ordered = ordered && (zs[Sc - 1] == zs[Sc - 1]) && (zs[Sf - 1] == zs[Sf - 1]);   // lone elements
        if (__all(ordered)) {
            for (int i = lane; i < Sc; i += 64) {   // a coarse depth follows the fine depths strictly below it
                const float me = zs[i];
                zsorted[i + prefix_length(Si, [&](int k) { return zs[Sc + k] < me; })] = me;
            }
            for (int j = lane; j < Si; j += 64) {   // a fine depth follows the coarse depths below or equal
                const float me = zs[Sc + j];
                zsorted[j + prefix_length(Sc, [&](int k) { return zs[k] <= me; })] = me;
            }
        } else {
        if (!build_keys(kc, zs, 0, Sc, Pc, lane)) bitonic_sort(kc, Pc, lane);
        if (!build_keys(ks, zs + Sc, Sc, Si, Ps, lane)) bitonic_sort(ks, Ps, lane);
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < Sc; i += 64) {
            const uint64_t me = kc[i];
            const uint32_t key = (uint32_t)(me >> 32);
            zsorted[i + prefix_length(Si, [&](int k) { return (uint32_t)(ks[k] >> 32) < key; })] = zs[(uint32_t)me];
        }
        for (int j = lane; j < Si; j += 64) {
            const uint64_t me = ks[j];
            const uint32_t key = (uint32_t)(me >> 32);
            zsorted[j + prefix_length(Sc, [&](int k) { return (uint32_t)(kc[k] >> 32) <= key; })] = zs[(uint32_t)me];
        }
        }
        __builtin_amdgcn_wave_barrier();
        for (int s = lane; s < Sf; s += 64) z_out[ray * Sf + s] = zsorted[s];
        if (pts_out) {
            const float* rr = rays + ray * ld;
            const float ox = rr[0], oy = rr[1], oz = rr[2], dx = rr[3], dy = rr[4], dz = rr[5];
            float* po = pts_out + ray * Sf * 3;
            for (int s = lane; s < Sf; s += 64) {   // three stores of stride 12 bytes; the cache merges them into full lines
                const float zz = zsorted[s];
                po[3 * s + 0] = ox + dx * zz;
                po[3 * s + 1] = oy + dy * zz;
                po[3 * s + 2] = oz + dz * zz;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

extern "C" {

int nerf_sample_fine(int64_t n, int Sc, int Si, const float* z, const float* weights, const float* u, int u_shared,
                     const float* rays, int ray_stride, float* z_out, float* pts_out, float* z_std, float* samples,
                     int64_t* inds, void* stream) {
    NERF_REQUIRE(n >= 0 && Sc >= 3 && Si >= 1, "nerf_sample_fine: n=%lld Sc=%d Si=%d (need Sc>=3, Si>=1)", (long long)n,
                 Sc, Si);
    if (n == 0) return 0;
    NERF_REQUIRE(z && weights && u && z_out, "nerf_sample_fine: null pointer");
    NERF_REQUIRE(!pts_out || (rays && ray_stride >= 6), "nerf_sample_fine: pts_out needs rays with stride >= 6");
    const int Sf4 = (Sc + Si + 3) & ~3, nb4 = (Sc - 1 + 3) & ~3;
    const size_t lds = (size_t)WAVES_PER_BLOCK * ((2 * Sf4 + 3 * nb4) * sizeof(float) +
                                                  (size_t)(pow2_at_least(Sc) + pow2_at_least(Si)) * sizeof(uint64_t));
    NERF_REQUIRE(lds <= 160 * 1024, "nerf_sample_fine: Sc+Si=%d needs %zu bytes of LDS (> 160 KiB)", Sc + Si, lds);
    if (lds > 64 * 1024) {
        hipError_t e = lds_limit_once(reinterpret_cast<const void*>(k_sample_fine<0, 0>), lds);
        if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_sample_fine: LDS attribute: %s", hipGetErrorString(e));
    }
    int64_t g = (n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    const int64_t cap = 256 * 8;
    if (g > cap) g = cap;
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((int)g), dim3(WAVE * WAVES_PER_BLOCK), lds, as_stream(stream), n, Sc, Si, 0, z,
                           weights, u, u_shared, rays, ray_stride, z_out, pts_out, z_std, samples, inds);
    };
    if (Sc == 64 && Si == 128) go(k_sample_fine<64, 128>);
    else if (Sc == 64 && Si == 64) go(k_sample_fine<64, 64>);
    else go(k_sample_fine<0, 0>);
    return launched("nerf_sample_fine");
}

int nerf_sample_pdf(int64_t n, int n_bins, int Si, const float* bins, const float* weights, const float* u,
                    int u_shared, float* samples, int64_t* inds, void* stream) {
    NERF_REQUIRE(n >= 0 && n_bins >= 2 && Si >= 1, "nerf_sample_pdf: n=%lld n_bins=%d Si=%d (need n_bins>=2, Si>=1)",
                 (long long)n, n_bins, Si);
    if (n == 0) return 0;
    NERF_REQUIRE(bins && weights && u && samples, "nerf_sample_pdf: null pointer");
    const int Sc = n_bins + 1;
    const int Sf4 = (Sc + Si + 3) & ~3, nb4 = (n_bins + 3) & ~3;
    const size_t lds = (size_t)WAVES_PER_BLOCK * ((2 * Sf4 + 3 * nb4) * sizeof(float) +
                                                  (size_t)(pow2_at_least(Sc) + pow2_at_least(Si)) * sizeof(uint64_t));
    NERF_REQUIRE(lds <= 160 * 1024, "nerf_sample_pdf: n_bins+Si=%d needs %zu bytes of LDS (> 160 KiB)", n_bins + Si, lds);
    if (lds > 64 * 1024) {
        hipError_t e = lds_limit_once(reinterpret_cast<const void*>(k_sample_fine<0, 0>), lds);
        if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_sample_pdf: LDS attribute: %s", hipGetErrorString(e));
    }
    int64_t g = (n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    const int64_t cap = 256 * 8;
    if (g > cap) g = cap;
    hipLaunchKernelGGL((k_sample_fine<0, 0>), dim3((int)g), dim3(WAVE * WAVES_PER_BLOCK), lds, as_stream(stream), n, Sc, Si, 1,
                       bins, weights, u, u_shared, nullptr, 0, nullptr, nullptr, nullptr, samples, inds);
    return launched("nerf_sample_pdf");
}

}  // extern "C"
