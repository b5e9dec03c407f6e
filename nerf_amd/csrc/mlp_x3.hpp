// Machinery shared by the MLP kernels that rebuild fp32 products from three-way bf16 splits on
// v_mfma_f32_16x16x32_bf16 (mlp_fwd_bf16x3.hip, mlp_bwd_bf16x3.hip): tile constants, the exact split, the three-slab
// LDS ring of weight triples, the order of the six piece products.
#pragma once
#include "mlp_core.hpp"

namespace nerfx3 {
using namespace nerfmlp;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NCT = 2;                         // 16-point column tiles per wave
constexpr int X3_THREADS = 256;                // four waves, one per SIMD
constexpr int X3_WAVES = X3_THREADS / 64;
constexpr int WAVE_PTS = 16 * NCT;
constexpr int X3_WG_PTS = X3_WAVES * WAVE_PTS; // 128 points per workgroup
constexpr int TRIPLE_BYTES = 3 * 1024;
constexpr int SLABT = 8;                       // triples per slab (every section of the stream is a multiple of 8)
constexpr int SLAB3_BYTES = SLABT * TRIPLE_BYTES;
constexpr int RING_DEPTH = 3;                  // slabs in the LDS ring: slab n+2 is requested when slab n is opened
constexpr int RING3_BYTES = RING_DEPTH * SLAB3_BYTES;
constexpr int WAVE_BLOCKS3 = SLABT * 3 / X3_WAVES;   // 1 KiB blocks each wave fetches per slab

// x = hi + mid + lo, exactly (each difference of an fp32 and its bf16 rounding is an fp32)
__host__ __device__ inline void split3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
    hi = (__bf16)x;
    const float r1 = x - (float)hi;
    mid = (__bf16)r1;
    lo = (__bf16)(r1 - (float)mid);
}

struct Stream3 {
    const char* g;   // this lane's source for block WAVE_BLOCKS3 * wave of slab 0
    char* ring;
    char* mine;
    int next, nslabs;
    // One address pair per slab: the instruction's immediate offset moves BOTH the global and the LDS address, so the six
    // 1 KiB blocks of a wave are offsets -2048 .. 3072 from the middle of its share (the field is 13 bits, signed).
    __device__ __forceinline__ void fetch(int slab) {
        const char* src = g + (int64_t)slab * SLAB3_BYTES + 2048;
        char* dst = mine + (slab % RING_DEPTH) * SLAB3_BYTES + 2048;
        static_for<WAVE_BLOCKS3>([&]<int b>() {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, b * 1024 - 2048, 0);
        });
    }
    // block b (of this wave's six) of one slab alone: for kernels that spread a slab's requests over its triples
    template <int b> __device__ __forceinline__ void fetch_block(int slab) {
        const char* src = g + (int64_t)slab * SLAB3_BYTES + 2048;
        char* dst = mine + (slab % RING_DEPTH) * SLAB3_BYTES + 2048;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, b * 1024 - 2048, 0);
    }
    // Opens slab `next`: its loads (issued two slabs ago) have landed -- those of slab next+1 may still be in flight,
    // loads return in order -- and every wave is done with slab next-1, whose buffer the request for slab next+2 reuses.
    // (One request per triple among the MFMAs, instead of six here, measured 3.6 % slower.)
    // STORES: vector-memory stores this wave has issued since the previous opening (the training kernels write a tile's
    // values behind the next tile's MFMAs).  They sit in the same in-order counter behind the next slab's six loads, so
    // the wait must allow for them: with vmcnt(6) alone every opening would also wait for that many of the NEXT slab's
    // loads, issued one slab ago instead of two.  A smaller STORES than the truth only waits longer; a larger one would
    // open a slab before its loads have landed.
    template <int STORES = 0> __device__ __forceinline__ const char* advance(int lane) {
        static_assert(WAVE_BLOCKS3 == 6, "the wait below leaves one slab's loads of this wave in flight");
        static_assert(STORES >= 0 && 6 + STORES <= 63, "vmcnt is a six-bit field");
        if (next + 1 < nslabs) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + STORES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // a bare barrier: __syncthreads() would first wait for ALL of this wave's loads, the next slab's included.  The
        // buffer requested below was last read a whole slab ago (reads run three triples ahead of their MFMAs), so no
        // LDS read of it is still pending in any wave.
        asm volatile("s_barrier" ::: "memory");
        const char* cur = ring + (next % RING_DEPTH) * SLAB3_BYTES + lane * 16;
        if (next + 2 < nslabs) fetch(next + 2);
        ++next;
        return cur;
    }
};

// Stores of finished tiles that a layer loop of NKS k-steps per tile (layer_x3 / layer_x3b: the epilogue of tile t - 1 runs
// piece by piece behind the k-steps of tile t, three triples behind the fragment reads, and writes 16 bytes per column
// tile when a tile's fourth value is done) issues during the SLABT triples before triple I, i.e. since the slab opening
// before the one at I.  For Stream3::advance<STORES>.
constexpr int x3_stores_since_open(int I, int NKS) {
    constexpr int DEPTH = 3, NPIECE = 4 * NCT;
    const int div = NKS < 8 ? NKS : 8;
    const int pps = (NPIECE + div - 1) / div;
    int n = 0;
    for (int J = I - SLABT; J < I; ++J) {
        if (J < DEPTH) continue;
        const int b = J - DEPTH, to = b / NKS, ks = b % NKS;
        if (to == 0) continue;
        for (int q = 0; q < pps; ++q) {
            const int k = ks * pps + q;
            if (k < NPIECE && (k & 3) == 3) ++n;
        }
    }
    return n;
}

// 256-input layers (one tile per slab): nothing before the first tile is done, then two stores per slab; the 63-input
// layer (four tiles per slab): one store per triple
static_assert(x3_stores_since_open(8, 8) == 0 && x3_stores_since_open(16, 8) == 1 && x3_stores_since_open(24, 8) == 2 &&
              x3_stores_since_open(120, 8) == 2 && x3_stores_since_open(8, 2) == 3 && x3_stores_since_open(16, 2) == 8,
              "store bookkeeping of the slab ring");

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// (weight piece, activation piece) of the six products, smallest first
constexpr int PROD_W[6] = {1, 2, 0, 1, 0, 0};
constexpr int PROD_X[6] = {1, 0, 2, 0, 1, 0};

// eight fp32 values -> the three bf16 fragments of their pieces
__device__ __forceinline__ void split8(const float* v, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        __bf16 h, m, l;
        split3(v[i], h, m, l);
        hi[i] = h; mid[i] = m; lo[i] = l;
    }
}

// The elements of nerf_gather_split3 this thread of the grid takes (a grid-stride loop): element pair (triple, w, w + 1)
// reads its two sources through the map, splits them and writes one 4-byte word per piece.  seg0 triples | aux floats |
// seg1 triples.
__device__ __forceinline__ void gather_split3_elements(const int32_t* __restrict__ table, const float* __restrict__ params,
                                                       char* __restrict__ out, int64_t triples0, int64_t n_aux, int64_t triples1) {
    const int64_t pairs0 = triples0 * 256, pairs1 = triples1 * 256;
    const int64_t total = pairs0 + n_aux + pairs1;
    char* seg1 = out + triples0 * TRIPLE_BYTES + n_aux * 4;
    const int32_t* tab1 = table + triples0 * 512 + n_aux;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (i >= pairs0 && i < pairs0 + n_aux) {
            const int64_t a = i - pairs0;
            const int32_t s = table[triples0 * 512 + a];
            reinterpret_cast<float*>(out + triples0 * TRIPLE_BYTES)[a] = s < 0 ? 0.0f : params[s];
            continue;
        }
        const bool second = i >= pairs0;
        const int64_t q = second ? i - pairs0 - n_aux : i;
        const int2 s = reinterpret_cast<const int2*>(second ? tab1 : table)[q];
        __bf16 h0, m0, l0, h1, m1, l1;
        split3(s.x < 0 ? 0.0f : params[s.x], h0, m0, l0);
        split3(s.y < 0 ? 0.0f : params[s.y], h1, m1, l1);
        auto word = [](__bf16 a, __bf16 b) {
            return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
        };
        unsigned* dst = reinterpret_cast<unsigned*>((second ? seg1 : out) + (q >> 8) * TRIPLE_BYTES) + (q & 255);
        dst[0] = word(h0, h1);
        dst[256] = word(m0, m1);
        dst[512] = word(l0, l1);
    }
}

}  // namespace nerfx3
