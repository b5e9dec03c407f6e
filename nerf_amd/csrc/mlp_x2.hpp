// Machinery of the MLP kernels that rebuild fp32 products from TWO fp16 pieces per operand on
// v_mfma_f32_16x16x32_f16 (mlp_fwd_f16x2.hip): tile constants, the split, the three-slab LDS ring of weight pairs, the
// range rules that decide when a tile has to be redone by the three-piece bf16 kernel.
//
// x ~ hi + lo * 2^-11 with hi = fp16(x), lo = fp16((x - hi) * 2^11): fp16 carries 11 significant bits, so the two pieces
// keep 22 of an fp32's 24.  A product a.b is rebuilt as hi.hi (one accumulator) + (hi.lo + lo.hi) * 2^-11 (a second
// accumulator, scaled once at the end): THREE matrix instructions where the bf16 scheme needs six, 4 bytes per weight in
// the stream instead of 6, 2 x 128 activation registers instead of 2 x 192.  The dropped lo.lo term and the two missing
// bits are below the rounding of the fp32 accumulation: a 256-term dot product measures rms 1.5e-8 of sum |a||b| against
// fp64 where v_mfma_f32_16x16x4_f32 measures 2.4e-8 (tools/probe_f16x2.hip).
//
// That holds while the leading piece is a NORMAL fp16: |x| in [2^-14, 65504].  Below, hi loses bits (absolute error
// 2^-36 instead of relative 2^-22); above, hi is infinite.  The kernels therefore watch the largest magnitude each lane
// sees per layer: a lane whose largest value is >= 2^15 (or infinite), or whose largest value is < 2^-11, marks its
// 128-point tile (a largest value of exactly 0 stands: zeros are exact), and marked tiles are recomputed by the bf16
// three-piece kernel, whose exactness does not depend on magnitude.  Values far below a lane's largest contribute below fp32 rounding of the sum whether or not their own
// relative precision is kept.  The same rule applied per layer to the weights at packing time marks every tile.
#pragma once
#include "mlp_core.hpp"
#include "mlp_bf16_layout.hpp"

namespace nerfx2 {
using namespace nerfmlp;
using namespace nerfbf16;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int NCT2 = 2;                         // 16-point column tiles per wave
constexpr int X2_THREADS = 256;                 // four waves, one per SIMD
constexpr int X2_WAVES = X2_THREADS / 64;
constexpr int WAVE_PTS2 = 16 * NCT2;
constexpr int X2_WG_PTS = X2_WAVES * WAVE_PTS2; // 128 points per workgroup
constexpr int PAIR_BYTES = 2 * 1024;            // hi | lo A fragments of one weight block
constexpr int SLABP = 8;                        // pairs per slab (every section of the stream is a multiple of 8 blocks)
constexpr int SLAB2_BYTES = SLABP * PAIR_BYTES;
constexpr int RING2_DEPTH = 3;                  // slab n+2 is requested when slab n is opened
constexpr int RING2_BYTES = RING2_DEPTH * SLAB2_BYTES;
constexpr int WAVE_BLOCKS2 = SLABP * 2 / X2_WAVES;   // 1 KiB blocks each wave fetches per slab

constexpr float LO_SCALE = 2048.0f, LO_UNSCALE = 1.0f / 2048.0f;
// a lane's largest magnitude of a layer must lie in [RANGE_LOW, RANGE_HIGH) for the two-piece result to stand
constexpr float RANGE_HIGH = 32768.0f, RANGE_LOW = 1.0f / 2048.0f;
constexpr int RANGE_WORDS = 16;                 // per-section maxima of |weight| behind the packed stream (uint32 bits)

// hi = fp16(x) (round to nearest even), lo = fp16((x - hi) * 2^11); x - hi is exact whenever hi is finite
__host__ __device__ inline void split2(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)((x - (float)hi) * LO_SCALE);
}

// two values -> one 32-bit word per piece (value 0 in the low half).  The residual is one fused multiply-add per
// value, fp16(v * 2^11 - hi * 2^11): both products are exact (powers of two) and so is their difference, so the result
// equals scaling v - hi, and v_fma_mixlo / mixhi_f16 reads hi as fp16 and rounds the result to fp16 by itself: five
// instructions per pair (one packed conversion, two multiplications, two mixed fused multiply-adds) instead of eight.
__device__ __forceinline__ void split2_pair(float v0, float v1, uint32_t& hi, uint32_t& lo) {
    const f16x2 h = {(_Float16)v0, (_Float16)v1};
    hi = __builtin_bit_cast(uint32_t, h);
    const float s0 = v0 * LO_SCALE, s1 = v1 * LO_SCALE, c = -LO_SCALE;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(lo) : "v"(hi), "v"(c), "v"(s0), "v"(s1));
}

struct Stream2 {
    const char* g;   // this lane's source for block WAVE_BLOCKS2 * wave of slab 0
    char* ring;
    char* mine;
    int next, nslabs;
    // One address pair per slab: the instruction's immediate offset moves BOTH the global and the LDS address.
    __device__ __forceinline__ void fetch(int slab) {
        const char* src = g + (int64_t)slab * SLAB2_BYTES + 2048;
        char* dst = mine + (slab % RING2_DEPTH) * SLAB2_BYTES + 2048;
        static_for<WAVE_BLOCKS2>([&]<int b>() {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, b * 1024 - 2048, 0);
        });
    }
    // Opens slab `next`: its loads (issued two slabs ago) have landed -- those of slab next+1 may still be in flight,
    // loads return in order -- and every wave is done with slab next-1, whose buffer the request for slab next+2 reuses.
    // STORES: vector-memory stores this wave has issued since the previous opening; they sit in the same in-order
    // counter behind the next slab's loads, so the wait must allow for them (a smaller number only waits longer).
    template <int STORES = 0> __device__ __forceinline__ const char* advance(int lane) {
        static_assert(STORES >= 0 && WAVE_BLOCKS2 + STORES <= 63, "vmcnt is a six-bit field");
        if (next + 1 < nslabs) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAVE_BLOCKS2 + STORES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // a bare barrier: __syncthreads() would first wait for ALL of this wave's loads, the next slab's included.  The
        // buffer requested below was last read a whole slab ago, so no LDS read of it is still pending in any wave.
        asm volatile("s_barrier" ::: "memory");
        const char* cur = ring + (next % RING2_DEPTH) * SLAB2_BYTES + lane * 16;
        if (next + 2 < nslabs) fetch(next + 2);
        ++next;
        return cur;
    }
};

__device__ __forceinline__ f32x4 mfma16h(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// |v| folded into a running maximum (fmaxf skips a NaN: a NaN value reaches the output as NaN on either kernel)
__device__ __forceinline__ float run_max(float m, float v0, float v1) {
    return __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v0), __builtin_fabsf(v1)));
}
// (an all-zero operand set is exact on any scheme: a maximum of 0 stands)
__device__ __forceinline__ bool out_of_range(float m) { return !(m < RANGE_HIGH) || (m < RANGE_LOW && m > 0.0f); }
// The same rule on the fp16 bits of leading pieces (non-negative fp16 values order like their bits; a NaN sorts above
// infinity): two running maxima per 32-bit word, then one value.
constexpr uint32_t RANGE_HIGH_H = 0x7800u, RANGE_LOW_H = 0x1000u;     // fp16 bits of 2^15 and 2^-11
__device__ __forceinline__ uint32_t run_max_h(uint32_t m, uint32_t pair) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, m), __builtin_bit_cast(u16x2, pair)));
}
__device__ __forceinline__ bool out_of_range_h(uint32_t m) { return m >= RANGE_HIGH_H || (m < RANGE_LOW_H && m != 0u); }

// ---- the packed forward stream: pair blocks in the order of mlp_bf16_layout.hpp, then the fp32 tables (AUX16_FLOATS),
// then RANGE_WORDS words: the largest |weight| of every section
__host__ __device__ inline int x2_stream_slabs(int use_viewdirs) { return bf16_stream_blocks(use_viewdirs) / SLABP; }
__host__ __device__ inline int64_t x2_packed_bytes(int use_viewdirs) {
    return (int64_t)x2_stream_slabs(use_viewdirs) * SLAB2_BYTES + (int64_t)AUX16_FLOATS * 4 + RANGE_WORDS * 4;
}
// section of the stream a block belongs to: 0..7 pts layers, 8 feature, 9 views
__host__ __device__ inline int x2_block_section(int blk) {
    if (blk < 32) return 0;
    blk -= 32;
    for (int l = 1; l < 8; ++l) {
        const int n = (l == 5) ? 160 : 128;
        if (blk < n) return l;
        blk -= n;
    }
    return blk < 128 ? 8 : 9;
}
__host__ __device__ constexpr int x2_sections(int use_viewdirs) { return use_viewdirs ? 10 : 8; }

// Stores of finished tiles that a layer loop of NKS k-steps per tile issues during the SLABP blocks before block I (the
// epilogue of tile t - 1 runs value by value behind the k-steps of tile t, three blocks behind the fragment reads, and
// writes 16 bytes per column tile when a tile's fourth value is done).  For Stream2::advance<STORES>.
constexpr int x2_stores_since_open(int I, int NKS) {
    constexpr int DEPTH = 3, NPIECE = 4 * NCT2;
    const int div = NKS < 8 ? NKS : 8;
    const int pps = (NPIECE + div - 1) / div;
    int n = 0;
    for (int J = I - SLABP; J < I; ++J) {
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

// 2^k as a float, -126 <= k <= 127
__device__ __forceinline__ float pow2i(int k) { return __uint_as_float((uint32_t)(127 + k) << 23); }

}  // namespace nerfx2
