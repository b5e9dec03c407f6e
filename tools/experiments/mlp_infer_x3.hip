// Inference forward: fused positional encoding + NeRF MLP with fp32 accuracy on v_mfma_f32_32x32x16_bf16.
//
// Arithmetic: as the training forward (mlp_fwd_bf16x3.hip) -- every fp32 operand split EXACTLY into three bf16 pieces, a
// product rebuilt from six piece products in the fp32 accumulator -- and bit for bit the same split, but on tiles of
// 32 features x 32 points and with the weights streamed as fp32:
//
//   * one wavefront carries 32 points through all layers; lane l is point l & 31, lane half h = l >> 5.  A result tile is
//     16 registers, register R = feature 32 t + (R & 3) + 8 (R >> 2) + 4 h.  Registers 8 s .. 8 s + 7 of a tile, split, are
//     as they stand the B fragment of k-step 2 t + s of the next layer (16 features per k-step): k-slot j of lane half h
//     stands for feature 16 s + 8 (j >> 2) + 4 h + (j & 3), and the weight stream is packed in that k order.  Half as many
//     matrix instructions as on 16x16x32 tiles, each twice as long behind the same 8 issue cycles: the fragment reads,
//     the epilogue and the weight split below fit in the gaps.
//   * the weight stream is fp32, 2 KiB per block of 32 rows x 16 k (4 bytes per weight instead of the 6 of a stream of
//     pre-split pieces).  Every workgroup reads the whole stream (3.5 MB of pieces per 128 points); measured on the
//     pre-split stream, the kernel ran at its compute-only time (15.6 ms per 4.19 M points) as long as a workgroup
//     fetched at most 5/6 of it and 5 ms slower with all of it -- whether the bytes came by LDS-DMA or through registers,
//     with any cache policy, wherever the requests were issued, with or without waiting for them, with 99.6 % L2 hits --
//     and at 1.93 GHz instead of 2.17: above about 24 GB/s per CU of L2 reads beside the matrix pipe the clock drops.
//     The fp32 stream needs 20 GB/s per CU.  Each wave loads two of a slab's eight blocks into registers (four 16-byte
//     loads per lane), splits them one slab period later (9 VALU instructions per pair of weights, in the matrix
//     instructions' shadow) and writes the three fragments of each block into the LDS ring, where all four waves read
//     them as before.
#include "common.hpp"
#include "mlp_core.hpp"
#include "mlp_tile16.hpp"
#include "mlp_x3.hpp"

using namespace nerfhip;
using namespace nerflayout;
using namespace nerfmlp;
using namespace nerftile16;
using namespace nerfx3;

namespace {

#ifndef XF_VALU_PER_GAP
#define XF_VALU_PER_GAP 6
#endif

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// feature (inside a 32-feature tile) that k-slot j of lane half h stands for in the tile's k-step s (0, 1)
__host__ __device__ inline int kslot32(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

// ---- stream layout ------------------------------------------------------------------------------
// Block = the A operand of one 32-row x 16-k weight block in fp32: [2][64 lanes][4 floats]; lane (r = l & 31, h = l >> 5),
// float e of half q is slot j = 4 q + e: W[32 to + r][input feature of k-step ks, lane half h, slot j].  Blocks are
// ordered row tile by row tile, a tile's k-steps consecutively; eight blocks make a slab.  Sections (blocks, each a whole
// number of slabs):
//   L0 32 (8 tiles x 4 k-steps of encoded position) | L1..L4 128 each | L5 160 (per tile: 16 hidden + 4 encoded) |
//   L6, L7 128 | feature 128 | views 72 (4 tiles x (16 feature + 2 encoded direction)).
constexpr int BLOCK_BYTES = 2048, SLABF_BYTES = SLABT * BLOCK_BYTES;
__host__ __device__ inline int xf_stream_blocks(int use_viewdirs) { return use_viewdirs ? 1160 : 960; }
__host__ __device__ inline int xf_stream_slabs(int use_viewdirs) { return xf_stream_blocks(use_viewdirs) / SLABT; }
__host__ __device__ inline int64_t xf_packed_bytes(int use_viewdirs) {
    return (int64_t)xf_stream_slabs(use_viewdirs) * SLABF_BYTES + (int64_t)AUX16_FLOATS * 4;
}
// flat-parameter index of slot j of lane `lane` of block t, or -1 for zero padding.  Encoded inputs sit in the slots
// the kernel computes them in: slot 8 ks + j of lane half h = enc_pts_feature / enc_dirs_feature (mlp_layout.hpp).
__host__ __device__ inline int64_t xf_stream_source(int use_viewdirs, int output_ch, int t, int lane, int j) {
    const ParamOffsets po = param_offsets(use_viewdirs, output_ch);
    const int r = lane & 31, h = lane >> 5;
    auto hidden = [&](int ks) { return 32 * (ks >> 1) + kslot32(ks & 1, h, j); };
    if (t < 32) {
        const int f = enc_pts_feature(8 * (t & 3) + j, h);
        return f < 0 ? -1 : po.pts_w[0] + (int64_t)(32 * (t >> 2) + r) * IN_PTS + f;
    }
    t -= 32;
    for (int l = 1; l < 8; ++l) {
        const int per = (l == 5) ? 20 : 16, n = 8 * per;
        if (t < n) {
            const int64_t row = 32 * (t / per) + r;
            const int ks = t % per, ld = (l == 5) ? W + IN_PTS : W;
            if (ks < 16) return po.pts_w[l] + row * ld + (l == 5 ? IN_PTS : 0) + hidden(ks);
            const int f = enc_pts_feature(8 * (ks - 16) + j, h);
            return f < 0 ? -1 : po.pts_w[5] + row * ld + f;
        }
        t -= n;
    }
    if (!use_viewdirs) return -1;
    if (t < 128) return po.feat_w + (int64_t)(32 * (t >> 4) + r) * W + hidden(t & 15);
    t -= 128;
    const int64_t row = 32 * (t / 18) + r;
    const int ks = t % 18;
    if (ks < 16) return po.views_w + row * po.views_in + hidden(ks);
    const int f = enc_dirs_feature(8 * (ks - 16) + j, h);
    return f < 0 ? -1 : po.views_w + row * po.views_in + W + f;
}

// LDS behind the ring: the fp32 tables, then the split encodings (fragments per piece and wave: four k-steps of encoded
// position, two of encoded direction)
__host__ __device__ constexpr int aux_floats(int use_viewdirs) { return use_viewdirs ? AUX16_HEAD + 648 : AUX16_FLOATS; }
__host__ __device__ constexpr int enc_frags(int use_viewdirs) { return use_viewdirs ? 6 : 4; }

// The weight stream of one workgroup.  Wave w owns blocks 2 w and 2 w + 1 of every slab: held[B] carries block 2 w + B
// of the slab after the one being multiplied from its loads to its split.  While slab n is multiplied, every k-step
// splits one pair of weights of slab n + 1 (step<K>: block K / 4, pair K % 4; nine VALU instructions); a block's fourth
// pair completes its three fragments, which go into ring slot (n + 1) % 3 -- last read during slab n - 2 -- and held[B]
// is reloaded from slab n + 2: a load has one slab period (0.8 us) to land.  advance() waits for this wave's LDS
// operations before the barrier that opens a slab (the fragment reads among them were issued a k-step of matrix
// instructions earlier); it does not wait for the loads in flight, as __syncthreads() would.
struct StreamF {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const char* g;      // this lane's first 16 bytes of block 2 w of slab 0
    char* ring;
    char* mine;         // this lane's fragment of block 2 w in ring slot 0
    int next, last;     // the slab the next advance() opens; the last slab of the stream
    f32x4 held[2][2];
    u32x4 part[3];      // the pairs of the block being split: piece s, pair i = part[s][i]
    template <int B> __device__ __forceinline__ void load(int slab) {
#ifdef XF_SAME   // (timing experiment: every load hits the same lines)
        const char* src = g + B * BLOCK_BYTES;
#else
        const char* src = g + (int64_t)(slab < last ? slab : last) * SLABF_BYTES + B * BLOCK_BYTES;   // (past the end: a re-read nobody uses)
#endif
        held[B][0] = *reinterpret_cast<const f32x4*>(src);
        held[B][1] = *reinterpret_cast<const f32x4*>(src + 1024);
    }
    template <int B, int PAIR> __device__ __forceinline__ void split_pair() {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const float a = held[B][PAIR >> 1][2 * (PAIR & 1)], b = held[B][PAIR >> 1][2 * (PAIR & 1) + 1];
        __bf16 pa[3], pb[3];
        split3(a, pa[0], pa[1], pa[2]);
        split3(b, pb[0], pb[1], pb[2]);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const bf16x2 p = {pa[s], pb[s]};
            part[s][PAIR] = __builtin_bit_cast(uint32_t, p);
        }
    }
    template <int B> __device__ __forceinline__ void write(int slab) {
        char* dst = mine + (slab % RING_DEPTH) * SLAB3_BYTES + B * TRIPLE_BYTES;
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x4*>(dst + s * 1024) = part[s];
    }
    // k-step K of the slab being multiplied (between advance(n) and advance(n + 1): next == n + 1)
    template <int K> __device__ __forceinline__ void step() {
        constexpr int B = K >> 2, PAIR = K & 3;
#ifndef XF_NO_SPLIT
        split_pair<B, PAIR>();
#endif
        if constexpr (PAIR == 3) {
            write<B>(next);
            load<B>(next + 1);
        }
    }
    __device__ __forceinline__ const char* advance(int lane) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const char* cur = ring + (next % RING_DEPTH) * SLAB3_BYTES + lane * 16;
        ++next;
        return cur;
    }
};

// One layer: NOUT row tiles (32 features each) x NKS k-steps (16 input features each).
// in_frag<ks, S>() = piece S of the input fragment of k-step ks.  Y[S][..] receives the pieces of the output (STORE):
// value R of tile t is slot R & 7 of k-step 2 t + (R >> 3).  A tile starts from its bias column, loaded from LDS into
// the accumulator while the tile before multiplies (layers with at least 8 k-steps; otherwise the epilogue adds it),
// and is finished off -- max with `lo` (0 = ReLU, -inf = none), NHEAD head dot products on the unsplit value, split --
// two values per k-step behind the next tile's matrix instructions.
template <int NOUT, int NKS, int NHEAD, bool STORE, typename InFrag>
__device__ __forceinline__ void layer_x3(StreamF& st, int lane, int h, const char*& cur, InFrag in_frag,
                                         bf16x8 (&Y)[3][16], const float* bias_lds, float lo, const float* head_lds,
                                         int head_stride, float (&head)[3]) {
    constexpr int NB = NOUT * NKS, DEPTH = 3;
    constexpr bool BIAS_INIT = NKS >= 8;
    constexpr int PPS = NKS >= 8 ? 2 : 16 / NKS;        // epilogue values per k-step: done after 8 k-steps at most
    static_assert(PPS * NKS >= 16, "a tile's epilogue fits behind the next tile");
    static_assert(NB % SLABT == 0, "a layer is a whole number of slabs");
    bf16x8 abuf[DEPTH + 1][3];
    f32x16 acc[2];
    const float* bp = bias_lds + 4 * h;
    const float* hp = head_lds + 4 * h;
    auto load_bias = [&]<int TO>() {
        static_for<4>([&]<int Q>() {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + 32 * TO + 8 * Q);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[TO & 1][4 * Q + e] = b4[e];
        });
    };
    // feature of register R inside its tile, without the lane half's 4 h (which bp / hp carry)
    auto piece = [&]<int TP, int R>() {
        constexpr int f = 32 * TP + 8 * (R >> 2) + (R & 3);
        float v = acc[TP & 1][R];
        if constexpr (!BIAS_INIT) v += bp[f];
        v = max_keep_nan(v, lo);
#pragma unroll
        for (int k = 0; k < NHEAD; ++k) {
            head[k] = fmaf(v, hp[k * head_stride + f], head[k]);
            asm volatile("" : "+v"(head[k]));   // keep the head sums in this order, one chain per head
        }
        if constexpr (STORE) {
            __bf16 hi, mid, low;
            split3(v, hi, mid, low);
            Y[0][2 * TP + (R >> 3)][R & 7] = hi; Y[1][2 * TP + (R >> 3)][R & 7] = mid; Y[2][2 * TP + (R >> 3)][R & 7] = low;
        }
    };

    if constexpr (BIAS_INIT) load_bias.template operator()<0>();
    static_for<NB + DEPTH>([&]<int I>() {
        if constexpr (I < NB) {
            if constexpr (I % SLABT == 0) cur = st.advance(lane);
            st.template step<I % SLABT>();
#pragma unroll
            for (int s = 0; s < 3; ++s)
                abuf[I % (DEPTH + 1)][s] = *reinterpret_cast<const bf16x8*>(cur + (I % SLABT) * TRIPLE_BYTES + s * 1024);
        }
        if constexpr (I >= DEPTH) {
            constexpr int b = I - DEPTH;
            constexpr int to = b / NKS, ks = b % NKS;
            static_for<6>([&]<int P>() {
                constexpr int pw = PROD_W[P], px = PROD_X[P];
                const bf16x8 a = abuf[b % (DEPTH + 1)][pw];
                const bf16x8 x = in_frag.template operator()<ks, px>();
                if constexpr (ks == 0 && P == 0 && !BIAS_INIT) {
                    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                    acc[to & 1] = mfma32(a, x, zero);
                } else {
                    acc[to & 1] = mfma32(a, x, acc[to & 1]);
                }
            });
            if constexpr (to > 0) {   // the previous row tile's epilogue, two values per k-step, behind this tile's MFMAs
                static_for<PPS>([&]<int Q>() {
                    constexpr int R = ks * PPS + Q;
                    if constexpr (R < 16) piece.template operator()<to - 1, R>();
                });
            }
            // the next tile's bias column goes into the accumulator that epilogue has drained
            if constexpr (BIAS_INIT && to + 1 < NOUT && ks == NKS - 3) load_bias.template operator()<to + 1>();
        }
#ifndef XF_NO_GROUPS
        if constexpr (I >= DEPTH) {   // one matrix instruction, then a share of the k-step's VALU work, six times
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, XF_VALU_PER_GAP, 0);
            }
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
    });
    static_for<16>([&]<int R>() { piece.template operator()<NOUT - 1, R>(); });
}

template <bool VIEWDIRS>
__global__ void __launch_bounds__(X3_THREADS, 1)
k_mlp_infer_x3(int64_t n_pts, int pts_per_ray, const float* __restrict__ pts, const float* __restrict__ viewdirs,
               int output_ch, const char* __restrict__ packed, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smemf[];
    char* ring = smemf;
    float* aux = reinterpret_cast<float*>(smemf + RING3_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5;
    constexpr int UV = VIEWDIRS ? 1 : 0;

    StreamF st;
    st.ring = ring;
    st.mine = ring + 2 * wave * TRIPLE_BYTES + lane * 16;
    st.g = packed + 2 * wave * BLOCK_BYTES + lane * 16;
    st.next = 0;
    st.last = xf_stream_slabs(UV) - 1;
    st.template load<0>(0);
    st.template load<1>(0);
    {   // (with the view branch only the first 648 head floats are in use: alpha row, three rgb rows, their biases)
        const float* gaux = reinterpret_cast<const float*>(packed + (int64_t)xf_stream_slabs(UV) * SLABF_BYTES);
        for (int i = tid * 4; i < aux_floats(UV); i += X3_THREADS * 4)
            *reinterpret_cast<f32x4*>(aux + i) = *reinterpret_cast<const f32x4*>(gaux + i);
    }

    // ---- encodings, split, parked in LDS: fragment (piece s, slot f) of this lane at ed[(s * NF + f) * 64],
    // f = k-step 0..3 of the position, 4 + k-step of the direction
    constexpr int NF = enc_frags(UV);
    bf16x8* ed = reinterpret_cast<bf16x8*>(smemf + RING3_BYTES + aux_floats(UV) * 4) + wave * (3 * NF * 64) + lane;
    auto park = [&](int f, const float* v) {
        bf16x8 hi, mid, low;
        split8(v, hi, mid, low);
        ed[(0 * NF + f) * 64] = hi;
        ed[(1 * NF + f) * 64] = mid;
        ed[(2 * NF + f) * 64] = low;
    };
    const int pw = wave * WAVE_PTS + (lane & 31);               // point inside the workgroup
    const int64_t p = (int64_t)blockIdx.x * X3_WG_PTS + pw;
    const bool valid = p < n_pts;
    const int64_t pl = valid ? p : n_pts - 1;
    {
        // each lane half evaluates 15 of the 30 (frequency, axis) pairs with one sincos and keeps both results;
        // slots 0, 1 hold the raw coordinates (x, y | z, padding)
        const float x0 = pts[pl * 3 + 0], x1 = pts[pl * 3 + 1], x2 = pts[pl * 3 + 2];
        float e[32];
        e[0] = h ? x2 : x0;
        e[1] = h ? 0.0f : x1;
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            const int c = 15 * h + i, k = c / 3, d = c - 3 * k;
            const float arg = (d == 0 ? x0 : (d == 1 ? x1 : x2)) * __int_as_float((127 + k) << 23);
            sincosf(arg, &e[2 + 2 * i], &e[3 + 2 * i]);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) park(ks, e + 8 * ks);
        if (VIEWDIRS) {
            const int64_t ray = pl / pts_per_ray;
            const float v0 = viewdirs[ray * 3 + 0], v1 = viewdirs[ray * 3 + 1], v2 = viewdirs[ray * 3 + 2];
            float dd[16];
            dd[0] = h ? v2 : v0;
            dd[1] = h ? 0.0f : v1;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int c = 6 * h + i, k = c / 3, d = c - 3 * k;
                const float arg = (d == 0 ? v0 : (d == 1 ? v1 : v2)) * __int_as_float((127 + k) << 23);
                sincosf(arg, &dd[2 + 2 * i], &dd[3 + 2 * i]);
            }
            dd[14] = dd[15] = 0.0f;
            park(4, dd);
            park(5, dd + 8);
        }
    }
    // slab 0 goes into the ring before the first barrier; slab 1 waits in the registers for its split
    static_for<8>([&]<int K>() {
        constexpr int B = K >> 2;
        st.template split_pair<B, (K & 3)>();
        if constexpr ((K & 3) == 3) {
            st.template write<B>(0);
            st.template load<B>(1);
        }
    });
    const char* cur = nullptr;

    bf16x8 X[3][16], Y[3][16];
    float sig[3] = {0.0f, 0.0f, 0.0f}, rgb[3] = {0.0f, 0.0f, 0.0f};
    const float NOCLAMP = -__builtin_inff();
    const float* bias = aux + AUX16_BIAS;
    const float* hw = aux + AUX16_HEAD;
    auto fromE = [&]<int ks, int s>() -> bf16x8 { return ed[(s * NF + ks) * 64]; };
    auto fromX = [&]<int ks, int s>() -> bf16x8 { return X[s][ks]; };
    auto fromY = [&]<int ks, int s>() -> bf16x8 { return Y[s][ks]; };
    auto fromXE = [&]<int ks, int s>() -> bf16x8 {
        if constexpr (ks < 16) return X[s][ks]; else return ed[(s * NF + ks - 16) * 64];
    };
    auto fromXD = [&]<int ks, int s>() -> bf16x8 {
        if constexpr (ks < 16) return X[s][ks]; else return ed[(s * NF + 4 + ks - 16) * 64];
    };
#define XF_LAYER(NOUT, NKS, NHEAD, STORE, IN, OUT, BIAS, LO, HEADW, HSTRIDE, HEAD) \
    layer_x3<NOUT, NKS, NHEAD, STORE>(st, lane, h, cur, IN, OUT, BIAS, LO, HEADW, HSTRIDE, HEAD)

    XF_LAYER(8, 4, 0, true, fromE, X, bias, 0.0f, hw, 0, sig);                        // L0
    XF_LAYER(8, 16, 0, true, fromX, Y, bias + 1 * 256, 0.0f, hw, 0, sig);             // L1
    XF_LAYER(8, 16, 0, true, fromY, X, bias + 2 * 256, 0.0f, hw, 0, sig);             // L2
    XF_LAYER(8, 16, 0, true, fromX, Y, bias + 3 * 256, 0.0f, hw, 0, sig);             // L3
    XF_LAYER(8, 16, 0, true, fromY, X, bias + 4 * 256, 0.0f, hw, 0, sig);             // L4
    XF_LAYER(8, 20, 0, true, fromXE, Y, bias + 5 * 256, 0.0f, hw, 0, sig);            // L5 (skip)
    XF_LAYER(8, 16, 0, true, fromY, X, bias + 6 * 256, 0.0f, hw, 0, sig);             // L6
    auto over_halves = [](float v) { return v + __shfl_xor(v, 32, 64); };
    if constexpr (VIEWDIRS) {
        XF_LAYER(8, 16, 1, true, fromX, Y, bias + 7 * 256, 0.0f, hw, 0, sig);         // L7 + alpha head
        XF_LAYER(8, 16, 0, true, fromY, X, bias + 8 * 256, NOCLAMP, hw, 0, sig);      // feature
        XF_LAYER(4, 18, 3, false, fromXD, Y, bias + 9 * 256, 0.0f, hw + 256, 128, rgb);   // views + rgb heads
        const float* hb = hw + 256 + 384;
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = over_halves(rgb[k]) + hb[1 + k];
        o[3] = over_halves(sig[0]) + hb[0];
        if (valid && h == 0) *reinterpret_cast<f32x4*>(out + p * 4) = o;
    } else {
        // output_linear: rows 0..2 as fp32 head dot products inside the L7 epilogue, rows 3..7 from the pieces afterwards
        XF_LAYER(8, 16, 3, true, fromX, Y, bias + 7 * 256, 0.0f, hw, 256, sig);
        for (int ch = 0; ch < output_ch; ++ch) {
            float s;
            if (ch < 3) {
                s = ch == 0 ? sig[0] : (ch == 1 ? sig[1] : sig[2]);
            } else {
                const float* wp = hw + ch * 256;
                s = 0.0f;
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float y = ((float)Y[0][ks][j] + (float)Y[1][ks][j]) + (float)Y[2][ks][j];
                        s = fmaf(y, wp[32 * (ks >> 1) + kslot32(ks & 1, h, j)], s);
                    }
            }
            s = over_halves(s) + hw[8 * 256 + ch];
            if (valid && h == 0) out[p * output_ch + ch] = s;
        }
    }
#undef XF_LAYER
}

// flat parameters -> fp32 stream of blocks + fp32 aux tables
__global__ void __launch_bounds__(256) k_pack_infer_x3(int use_viewdirs, int output_ch, const float* __restrict__ params,
                                                       float* __restrict__ out) {
    const int64_t elems = (int64_t)xf_stream_blocks(use_viewdirs) * 512;     // one per (block, half, lane, float)
    const int64_t total = elems + AUX16_FLOATS;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t src;
        if (i < elems) {
            const int w = (int)(i & 511);
            src = xf_stream_source(use_viewdirs, output_ch, (int)(i >> 9), (w >> 2) & 63, 4 * (w >> 8) + (w & 3));
        } else {
            src = aux16_source(use_viewdirs, output_ch, (int)(i - elems));
        }
        out[i] = src < 0 ? 0.0f : params[src];
    }
}

constexpr size_t xf_lds_bytes(int uv) { return RING3_BYTES + aux_floats(uv) * sizeof(float) + X3_WAVES * 3 * enc_frags(uv) * 64 * 16; }
static_assert(xf_lds_bytes(1) <= 160 * 1024 && xf_lds_bytes(0) <= 160 * 1024, "LDS of one compute unit");

}  // namespace

extern "C" {

int64_t nerf_mlp_packed_infer_x3_bytes(int use_viewdirs) { return xf_packed_bytes(use_viewdirs); }

int nerf_pack_weights_infer_x3(int use_viewdirs, int output_ch, const float* params, void* packed, void* stream) {
    NERF_REQUIRE(params && packed, "nerf_pack_weights_infer_x3: null pointer");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_pack_weights_infer_x3: output_ch=%d not in 1..8", output_ch);
    hipLaunchKernelGGL(k_pack_infer_x3, dim3(1024), dim3(256), 0, as_stream(stream), use_viewdirs, output_ch, params,
                       reinterpret_cast<float*>(packed));
    return launched("nerf_pack_weights_infer_x3");
}

int nerf_mlp_infer_x3(int64_t n_pts, int pts_per_ray, const float* pts, const float* viewdirs, int use_viewdirs,
                      int output_ch, const void* packed, float* out, void* stream) {
    NERF_REQUIRE(n_pts >= 0, "nerf_mlp_infer_x3: n_pts=%lld", (long long)n_pts);
    if (n_pts == 0) return 0;
    NERF_REQUIRE(pts && packed && out, "nerf_mlp_infer_x3: null pointer");
    NERF_REQUIRE(!use_viewdirs || (viewdirs && pts_per_ray >= 1), "nerf_mlp_infer_x3: view branch needs viewdirs and pts_per_ray >= 1");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_mlp_infer_x3: output_ch=%d not in 1..8", output_ch);
    NERF_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                 "nerf_mlp_infer_x3: packed and out must be 16-byte aligned");
    const int64_t blocks = (n_pts + X3_WG_PTS - 1) / X3_WG_PTS;
    NERF_REQUIRE(blocks <= 0x7fffffff, "nerf_mlp_infer_x3: n_pts=%lld too large for one launch", (long long)n_pts);
    auto go = [&](auto kernel, int ppr, const float* vd, int och) -> int {
        const size_t lds = xf_lds_bytes(use_viewdirs);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_infer_x3: LDS attribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(X3_THREADS), lds, as_stream(stream), n_pts, ppr, pts,
                           vd, och, reinterpret_cast<const char*>(packed), out);
        return launched("nerf_mlp_infer_x3");
    };
    if (use_viewdirs) return go(k_mlp_infer_x3<true>, pts_per_ray, viewdirs, 4);
    return go(k_mlp_infer_x3<false>, 1, nullptr, output_ch);
}

}  // extern "C"
