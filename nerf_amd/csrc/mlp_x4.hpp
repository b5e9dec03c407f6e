// Machinery shared by the MLP kernels on 16-point tiles (v_mfma_f32_16x16x4_f32, two waves per SIMD, eight per
// workgroup = 128 points): the LDS slab ring of 1 KiB weight blocks and the layer loop.
//
// Tile layout: lane l = point l & 15 of the wave, lane group g = l >> 4; an activation (or gradient) tile is 16 rows x 16
// points, register i of tile t = row 16 t + 4 g + i.  Register i of input tile t is, as it stands, the B operand of one
// MFMA whose four k-slots (the lane groups) are rows 16 t + 4 g + i, so a weight block holds, per lane (r = l & 15, g),
// the four consecutive floats M[16 to + r][16 kb + 4 g + 0..3] of the matrix M the layer multiplies by (the weight
// matrix in the forward pass, its transpose in the backward pass).  Inside a layer's section of the stream the blocks of
// row tiles 2u and 2u+1 alternate: block b -> pair u = b / (2 NKB), k-block kb = (b % (2 NKB)) / 2, row tile 2u + (b & 1).
#pragma once
#include "mlp_core.hpp"

namespace nerfx4 {
using namespace nerfmlp;

constexpr int X4_THREADS = 512;
constexpr int X4_WAVES = X4_THREADS / 64;
constexpr int X4_WG_PTS = 16 * X4_WAVES;          // 128 points per workgroup
constexpr int X4_SLAB = 32;                       // 1 KiB blocks per slab (16 and 64 measured 0.2 % and 1 % slower)
constexpr int X4_SLAB_FLOATS = X4_SLAB * 256;
constexpr int X4_RING_FLOATS = 2 * X4_SLAB_FLOATS;
constexpr int X4_WAVE_BLOCKS = X4_SLAB / X4_WAVES;

// ---- the slab ring as one wave sees it ----------------------------------------------------------
struct StreamX4 {
    const float* g;   // this lane's source for block X4_WAVE_BLOCKS * wave of slab 0
    float* ring;
    float* mine;
    int next, nslabs;
    __device__ __forceinline__ void fetch(int slab) {
        const float* src = g + (int64_t)slab * X4_SLAB_FLOATS;
        float* dst = mine + (slab & 1) * X4_SLAB_FLOATS;
#pragma unroll
        for (int b = 0; b < X4_WAVE_BLOCKS; ++b) glds16(src + b * 256, dst + b * 256);
    }
    __device__ __forceinline__ const float* advance(int lane) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // every wave's share of slab `next` has landed; all reads of the other buffer are done
        const float* cur = ring + (next & 1) * X4_SLAB_FLOATS + lane * 4;
        if (next + 1 < nslabs) fetch(next + 1);
        ++next;
        return cur;
    }
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// 16-byte store of a finished tile to the saved activations / gradient workspace (a non-temporal store measured the
// same: +0.1 % on the train step)
__device__ __forceinline__ void save16(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }

// An empty statement that consumes two register quads: the compiler has to place the wait for their loads here.
__device__ __forceinline__ void need_now(const f32x4& a, const f32x4& b) { asm volatile("" ::"v"(a), "v"(b)); }

// One layer: NOUT row tiles (an even number) x NKB k-blocks (16 input rows each).
// in(kb, j) = this lane's value of input row 16 kb + 4 g + j for its point.  A step = the two blocks (row tiles 2u,
// 2u+1) of one k-block = eight MFMAs that alternate between the two accumulators, so consecutive MFMAs never depend on
// each other (40-cycle dependent latency against 32-cycle issue).  start<U>() writes the initial values of out[2U] and
// out[2U+1] (while the previous pair is still multiplying), finish<U>() is the pair's epilogue (behind the second
// k-block of the next pair; the last pair's is exposed).
// The blocks of step s+2 are requested at the END of step s, after the wait for those of step s+1: that wait is the
// only one in the loop and finds reads that were issued a whole step (eight MFMAs) earlier.
template <int NOUT, int NKB, typename In, typename Start, typename Finish>
__device__ __forceinline__ void x4_layer(StreamX4& st, int lane, const float*& cur, In in, f32x4 (&out)[16], Start start,
                                         Finish finish) {
    constexpr int NS = (NOUT / 2) * NKB;
    static_assert(NOUT % 2 == 0 && NKB >= 3, "row tiles come in pairs; the epilogue needs three steps of cover");
    f32x4 abuf[3][2];
    auto read_step = [&]<int S>() {
        if constexpr ((2 * S) % X4_SLAB == 0) cur = st.advance(lane);
        abuf[S % 3][0] = *reinterpret_cast<const f32x4*>(cur + ((2 * S) % X4_SLAB) * 256);
        abuf[S % 3][1] = *reinterpret_cast<const f32x4*>(cur + ((2 * S + 1) % X4_SLAB) * 256);
    };
    auto await_step = [&]<int S>() { need_now(abuf[S % 3][0], abuf[S % 3][1]); };
    start.template operator()<0>();
    read_step.template operator()<0>();
    if constexpr (NS > 1) read_step.template operator()<1>();
    await_step.template operator()<0>();
    static_for<NS>([&]<int s>() {
        constexpr int u = s / NKB, kb = s % NKB;
        const f32x4 a0 = abuf[s % 3][0], a1 = abuf[s % 3][1];
        static_for<4>([&]<int J>() {
            const float x = in.template operator()<kb, J>();
            out[2 * u] = mfma4(a0[J], x, out[2 * u]);
            out[2 * u + 1] = mfma4(a1[J], x, out[2 * u + 1]);
        });
        if constexpr (kb == NKB - 2 && 2 * u + 2 < NOUT) start.template operator()<u + 1>();
        if constexpr (kb == 1 && u > 0) finish.template operator()<u - 1>();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (s + 1 < NS) await_step.template operator()<s + 1>();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (s + 2 < NS) read_step.template operator()<s + 2>();
        __builtin_amdgcn_sched_barrier(0);
    });
    finish.template operator()<NOUT / 2 - 1>();
}

// host side: launches the backward-data kernel on 16-point tiles (mlp_bwd_x4.hip)
int launch_bwd_data_x4(int64_t n_pts, int use_viewdirs, int output_ch, const float* d_out, const float* act,
                       const float* packed_x4, const float* packed_bwd_x4, float* scratch, hipStream_t stream);

}  // namespace nerfx4
