// Machinery of the MLP kernels that rebuild fp32 products from TWO fp16 pieces per operand on
// v_mfma_f32_32x32x16_f16 (mlp_fwd_f16x2.hip): tile constants, the split, the three-slab LDS ring of weight pairs, the
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
// 128-point tile, and marked tiles are recomputed by the bf16 three-piece kernel, whose exactness does not depend on
// magnitude.  Values far below a lane's largest contribute below fp32 rounding of the sum whether or not their own
// relative precision is kept.  The same rule applied per layer to the weights at packing time marks every tile.
#pragma once
#include "mlp_core.hpp"

namespace nerfx2 {
using namespace nerfmlp;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int X2_THREADS = 256;                 // four waves, one per SIMD
constexpr int X2_WAVES = X2_THREADS / 64;
constexpr int WAVE_PTS2 = 32;                   // one 32-point column tile per wave
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

// two values -> one 32-bit word per piece (value 0 in the low half)
__device__ __forceinline__ void split2_pair(float v0, float v1, uint32_t& hi, uint32_t& lo) {
    const f16x2 h = {(_Float16)v0, (_Float16)v1};
    const f16x2 l = {(_Float16)((v0 - (float)h[0]) * LO_SCALE), (_Float16)((v1 - (float)h[1]) * LO_SCALE)};
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
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

// 16 bytes of LDS at p + OFFSET, issued now; the value may be used only behind lds_wait (the compiler is not told that
// the read is asynchronous, so no wait of its own guards it)
template <int OFFSET> __device__ __forceinline__ u32x4 lds_read16(const char* p) {
    static_assert(OFFSET >= 0 && OFFSET < 65536 && OFFSET % 16 == 0, "ds_read_b128 immediate offset");
    u32x4 v;
    const uint32_t addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
#ifdef X2_HALF_BYTES   // (timing experiment: the same number of reads, half the bytes)
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 w;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(w) : "v"(addr), "n"(OFFSET));
    v[0] = w[0]; v[1] = w[1]; v[2] = w[0]; v[3] = w[1];
#else
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFFSET));
#endif
    return v;
}
// Waits until at most YOUNGER LDS operations of this wave are outstanding (LDS operations complete in order), i.e. until
// the reads of a and b -- issued before the YOUNGER most recent ones -- have landed.  a and b pass through the
// statement so that no use of them can be scheduled in front of it.  Counting fewer younger operations than there are
// only waits longer.
template <int YOUNGER> __device__ __forceinline__ void lds_wait(u32x4& a, u32x4& b) {
    static_assert(YOUNGER >= 0 && YOUNGER <= 15, "lgkmcnt is a four-bit field");
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(YOUNGER));
}

__device__ __forceinline__ f32x16 mfma32h(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// |v| folded into a running maximum (fmaxf skips a NaN: a NaN value reaches the output as NaN on either kernel)
__device__ __forceinline__ float run_max(float m, float v0, float v1) {
    return __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v0), __builtin_fabsf(v1)));
}
__device__ __forceinline__ bool out_of_range(float m) { return !(m < RANGE_HIGH) || m < RANGE_LOW; }

}  // namespace nerfx2
