// Device-side machinery shared by the fused MLP kernels: the LDS slab ring that streams packed
// weight blocks, and the section loop that turns blocks into chains of v_mfma_f32_32x32x2_f32 on
// register-resident activation tiles (layout: mlp_layout.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include "mlp_layout.hpp"

namespace nerfmlp {
using namespace nerflayout;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 256;
constexpr int RING_FLOATS = 2 * SLAB_FLOATS;
constexpr int WAVE_BLOCKS = SLAB_BLOCKS / 4;   // blocks of a slab each of the four waves fetches

__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// The weight stream as seen by one wave: a double-buffered slab ring in LDS.
struct Stream {
    const float* g;   // this lane's source pointer for block WAVE_BLOCKS*wave of slab 0
    float* ring;      // LDS ring base
    float* mine;      // this wave's destination inside a slab (wave-uniform)
    int next;         // next slab to consume
    int nslabs;

    __device__ __forceinline__ void fetch(int slab) {
        const float* src = g + (int64_t)slab * SLAB_FLOATS;
        float* dst = mine + (slab & 1) * SLAB_FLOATS;
#pragma unroll
        for (int b = 0; b < WAVE_BLOCKS; ++b) glds16(src + b * BLOCK_FLOATS, dst + b * BLOCK_FLOATS);
    }
    // Make slab `next` readable, start fetching the one after it, return this lane's read pointer.
    __device__ __forceinline__ const float* advance(int lane) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // every wave's quarter of slab `next` has landed; all reads of the other buffer are done
        const float* cur = ring + (next & 1) * SLAB_FLOATS + lane * 4;
        if (next + 1 < nslabs) fetch(next + 1);
        ++next;
        return cur;
    }
};

// max(v, lo) that keeps a NaN a NaN, as torch.relu does (fmaxf would return the other operand and turn a NaN
// pre-activation into 0); one v_maximum3_f32, where fmaxf costs a canonicalising v_max_f32 more
__device__ __forceinline__ float max_keep_nan(float v, float lo) { return __builtin_elementwise_maximum(v, lo); }

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// compile-time loop: body.template operator()<i>() for i = 0..N-1.  Register tiles are indexed by
// these constants only; a run-time index would send the whole tile array to scratch memory.
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f.template operator()<I>(), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// acc[to] (+)= sum over NIN input tiles: one section of NOUT*NIN*4 blocks = NOUT*NIN/4 slabs.
// The LDS reads run DEPTH blocks ahead of the MFMAs that consume them, also across the slab barrier:
// the last MFMAs of a slab issue after the barrier and behind the first reads of the next slab, so
// neither the LDS latency nor the barrier leaves the matrix pipe idle.
template <int NOUT, int NIN, bool ACCUM, typename InTile>
__device__ __forceinline__ void run_section(Stream& st, int lane, f32x16 (&acc)[8], InTile in_tile) {
    constexpr int NB = NOUT * NIN * 4, DEPTH = 2;
    // a section starts on a slab boundary; the last slab of the stream may be partly used (the fetch then
    // reads on into whatever follows the stream in the packed buffer, which nothing consumes)
    f32x4 abuf[DEPTH + 1];
    const float* cur = nullptr;
    static_for<NB + DEPTH>([&]<int I>() {
        if constexpr (I < NB) {
            if constexpr (I % SLAB_BLOCKS == 0) cur = st.advance(lane);
            abuf[I % (DEPTH + 1)] = *reinterpret_cast<const f32x4*>(cur + (I % SLAB_BLOCKS) * BLOCK_FLOATS);
        }
        if constexpr (I >= DEPTH) {
            constexpr int b = I - DEPTH;
            constexpr int to = b / (NIN * 4), ti = (b >> 2) % NIN, q = b & 3;
            const f32x4 a = abuf[b % (DEPTH + 1)];
            if constexpr (!ACCUM && ti == 0 && q == 0) {
                const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                acc[to] = mfma(a[0], in_tile.template operator()<ti>()[0], zero);
            } else {
                acc[to] = mfma(a[0], in_tile.template operator()<ti>()[4 * q + 0], acc[to]);
            }
            acc[to] = mfma(a[1], in_tile.template operator()<ti>()[4 * q + 1], acc[to]);
            acc[to] = mfma(a[2], in_tile.template operator()<ti>()[4 * q + 2], acc[to]);
            acc[to] = mfma(a[3], in_tile.template operator()<ti>()[4 * q + 3], acc[to]);
        }
        // keep the read of block I ahead of the MFMAs of block I-DEPTH+1: the scheduler would otherwise sink
        // every read to just before its first use and expose the LDS latency once per block pair
        __builtin_amdgcn_sched_barrier(0);
    });
}

// One forward layer (or one of the two sections of the skip layer) on ping-pong tile arrays:
//     out[to] = act( start + sum_ti W(to, ti) in[ti] ),   start = bias column (bias_on) or out[to]'s current value.
// The bias is not added afterwards: a tile's accumulator registers are loaded with its bias column from LDS while
// the previous tile is still multiplying.  Everything else that has to happen to a finished tile rides in the
// shadow of the matrix pipe, behind weight blocks of the NEXT tile: ReLU in place behind its third block; in training the
// ReLU mask bits (bit 16 (t % 2) + r of dword t / 2 of mask_bits <-> register r of tile t positive: the
// backward-data kernel masks with these 16 bytes per lane instead of re-reading the activations); and `side`,
// called once per block as side.template operator()<to, block in tile>(), for work that only READS the layer's
// inputs (storing them for the backward pass).  Only the last tile's ReLU is exposed.
// `bias_h` = bias table + 16 * lane half.
struct NoSide {
    template <int TO, int BI> __device__ __forceinline__ void operator()() const {}
};
template <int NOUT, int NIN, bool RELU, bool MASK, typename InTile, typename Side>
__device__ __forceinline__ void run_layer(Stream& st, int lane, f32x16 (&out)[8], InTile in_tile, const float* bias_h,
                                          bool bias_on, Side side, uint32_t (&mask_bits)[4]) {
    constexpr int BPT = NIN * 4, NB = NOUT * BPT, DEPTH = 2;   // blocks per tile, per layer
    static_assert(BPT >= 8, "a tile spans at least eight blocks");
    f32x4 abuf[DEPTH + 1];
    const float* cur = nullptr;
    auto load_bias = [&]<int TO>() {
        if (bias_on) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_h + TO * 32 + 4 * i);
#pragma unroll
                for (int e = 0; e < 4; ++e) out[TO][4 * i + e] = bb[e];
            }
        }
    };
    auto relu = [&]<int TO, int R>() {
        const float y = max_keep_nan(out[TO][R], 0.0f);
        out[TO][R] = y;
        if constexpr (MASK) {   // y >= 0, so "positive" = "bits != 0"
            const uint32_t one = __float_as_uint(y) < 1u ? __float_as_uint(y) : 1u;
            mask_bits[TO >> 1] |= one << (16 * (TO & 1) + R);
        }
    };
    if constexpr (MASK) mask_bits[0] = mask_bits[1] = mask_bits[2] = mask_bits[3] = 0u;
    load_bias.template operator()<0>();
    static_for<NB + DEPTH>([&]<int I>() {
        if constexpr (I < NB) {
            if constexpr (I % SLAB_BLOCKS == 0) cur = st.advance(lane);
            abuf[I % (DEPTH + 1)] = *reinterpret_cast<const f32x4*>(cur + (I % SLAB_BLOCKS) * BLOCK_FLOATS);
        }
        if constexpr (I >= DEPTH) {
            constexpr int b = I - DEPTH;
            constexpr int to = b / BPT, bi = b % BPT, ti = bi >> 2, q = bi & 3;
            const f32x4 a = abuf[b % (DEPTH + 1)];
#pragma unroll
            for (int e = 0; e < 4; ++e) out[to] = mfma(a[e], in_tile.template operator()<ti>()[4 * q + e], out[to]);
            if constexpr (ti == NIN - 1 && q == 0 && to + 1 < NOUT) load_bias.template operator()<to + 1>();
            // (one burst behind the third block: spread thinner, one register per block, it measured 1 % slower --
            // every visit pays the accumulator-file round trip and its wait states)
            if constexpr (RELU && to > 0 && bi == 2) static_for<16>([&]<int R>() { relu.template operator()<to - 1, R>(); });
            side.template operator()<to, bi>();
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (RELU) static_for<16>([&]<int R>() { relu.template operator()<NOUT - 1, R>(); });
}

}  // namespace nerfmlp
