// Fused positional encoding + NeRF MLP forward with fp32 accuracy on the fp16 matrix instruction, two pieces per operand.
//
// Every fp32 operand is cut into two fp16 pieces, x ~ hi + lo * 2^-11 (mlp_x2.hpp), and a product is rebuilt from three
// of the four piece products on v_mfma_f32_16x16x32_f16 with fp32 accumulation: hi.hi into one accumulator, hi.lo and
// lo.hi into a second one that is scaled by 2^-11 when the tile is finished.  Half the matrix instructions of the
// three-piece bf16 kernel (mlp_fwd_bf16x3.hip), two thirds of its weight stream, two thirds of its activation registers;
// measured error against fp64 below that of the fp32 matrix instruction while the operands stay inside fp16's normal
// range.  Each workgroup watches that range (per layer and point, and the packer per layer for the weights) and marks
// its 128-point tile in `tile_flags` when it is left; nerf_mlp_fwd_f16x2 then has the three-piece kernel redo exactly
// the marked tiles, so the result never depends on magnitudes.
//
// Dataflow: one wavefront carries 32 points through all layers on v_mfma_f32_32x32x16_f16; lane l is point l & 31, lane
// half h = l >> 5.  A result tile is 32 features x 32 points = 16 registers per lane, register R = feature
// 32 t + (R & 3) + 8 (R >> 2) + 4 h.  Registers 8 s .. 8 s + 7 of a tile, split, are as they stand the B fragment of
// k-step 2 t + s of the next layer (16 features per k-step): k-slot j of lane half h stands for feature
// 16 s + 8 (j >> 2) + 4 h + (j & 3), and the weight stream is packed in that k order -- activations never leave the
// register file (2 x 128 registers, one wave per SIMD, four waves = 128 points per workgroup).  Weights stream through a
// three-slab LDS ring by LDS-DMA behind a bare s_barrier; a tile's bias column is loaded into its accumulator while the
// tile before multiplies; a tile's epilogue (scale-and-add of the two accumulators, ReLU, heads, the split of its values)
// rides two values per k-step behind the next tile's matrix instructions.
// Why 32 x 32 tiles: a matrix instruction holds the SIMD's vector issue for 8 cycles whatever its shape.  On 16x16x32
// tiles (16 cycles each) that leaves 8 cycles per instruction for everything else, and the two 16-byte fragment reads
// per weight block alone then cost 37 % of the kernel (12.5 ms per 4.19 M points; 7.8 ms with the reads taken out, 6.1 ms
// with reads and epilogue taken out = the matrix pipe's own time); 32x32x16 (32 cycles each) leaves 24.
#include "common.hpp"
#include "mlp_core.hpp"
#include "mlp_tile16.hpp"
#include "mlp_x2.hpp"

using namespace nerfhip;
using namespace nerflayout;
using namespace nerfmlp;
using namespace nerftile16;
using namespace nerfx2;

namespace {

// ---- stream layout ------------------------------------------------------------------------------
// Block = the A operand of one 32-row x 16-k weight block as a pair of 1 KiB fragments hi | lo: [64 lanes][8 fp16]; lane
// (r = l & 31, h = l >> 5), slot j holds W[32 to + r][input feature of k-step ks, lane half h, slot j].  Blocks are
// ordered row tile by row tile, a tile's k-steps consecutively; eight blocks make a slab.  Sections (blocks, each a whole
// number of slabs):
//   L0 32 (8 tiles x 4 k-steps of encoded position) | L1..L4 128 each | L5 160 (per tile: 16 hidden + 4 encoded) |
//   L6, L7 128 | feature 128 | views 72 (4 tiles x (16 feature + 2 encoded direction)).
__host__ __device__ inline int x2_stream_blocks(int use_viewdirs) { return use_viewdirs ? 1160 : 960; }
__host__ __device__ inline int x2_stream_slabs(int use_viewdirs) { return x2_stream_blocks(use_viewdirs) / SLABP; }
__host__ __device__ inline int64_t x2_packed_bytes(int use_viewdirs) {
    return (int64_t)x2_stream_slabs(use_viewdirs) * SLAB2_BYTES + (int64_t)AUX16_FLOATS * 4 + RANGE_WORDS * 4;
}
// feature (inside a 32-feature tile) that k-slot j of lane half h stands for in the tile's k-step s (0, 1)
__host__ __device__ inline int kslot32(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }
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
// flat-parameter index of slot j of lane `lane` of block t, or -1 for zero padding.  Encoded inputs sit in the slots
// the kernel computes them in: slot 8 ks + j of lane half h = enc_pts_feature / enc_dirs_feature (mlp_layout.hpp).
__host__ __device__ inline int64_t x2_stream_source(int use_viewdirs, int output_ch, int t, int lane, int j) {
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
// position, two of encoded direction), then one flag per wave
__host__ __device__ constexpr int aux2_floats(int use_viewdirs) { return use_viewdirs ? AUX16_HEAD + 648 : AUX16_FLOATS; }
__host__ __device__ constexpr int enc2_frags(int use_viewdirs) { return use_viewdirs ? 6 : 4; }

// One layer: NOUT row tiles (32 features each) x NKS k-steps (16 input features each).
// in_frag<ks, S>() = piece S (0 hi, 1 lo) of the input fragment of k-step ks.  Y[S][..] receives the pieces of the output
// (STORE): value R of tile t is slot R & 7 of k-step 2 t + (R >> 3); mx = the largest magnitude this lane has produced.
// A tile starts from its bias column, loaded from LDS into the hi.hi accumulator while the tile before multiplies
// (layers with at least 8 k-steps; otherwise the epilogue adds it), and is finished off -- the two accumulators joined,
// max with `lo` (0 = ReLU, -inf = none), NHEAD head dot products on the unsplit value, split -- two values per k-step
// behind the next tile's matrix instructions.
template <int NOUT, int NKS, int NHEAD, bool STORE, typename InFrag>
__device__ __forceinline__ void layer_x2(Stream2& st, int lane, int h, const char*& cur, InFrag in_frag,
                                         u32x4 (&Y)[2][16], const float* bias_lds, float lo, const float* head_lds,
                                         int head_stride, float (&head)[3], float& mx) {
    constexpr int NB = NOUT * NKS, DEPTH = 3;
    constexpr bool BIAS_INIT = NKS >= 8;
#ifdef X2_PPS1
    constexpr int PPS = NKS >= 16 ? 1 : (NKS >= 8 ? 2 : 16 / NKS);
#else
    constexpr int PPS = NKS >= 8 ? 2 : 16 / NKS;        // epilogue values per k-step: done after 8 k-steps at most
#endif
    static_assert(PPS * NKS >= 16, "a tile's epilogue fits behind the next tile");
    static_assert(NB % SLABP == 0, "a layer is a whole number of slabs");
    u32x4 abuf[DEPTH + 1][2];
    f32x16 amain[2], acorr[2];                    // [alternating set]: hi.hi | hi.lo + lo.hi
    const float* bp = bias_lds + 4 * h;
    const float* hp = head_lds + 4 * h;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float pv = 0.0f;                              // STORE: the even value of the pair being collected
#ifdef X2_PPS1
    f32x16 bnext = zero;
    auto load_bias = [&]<int TO>() {
        static_for<4>([&]<int Q>() {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + 32 * TO + 8 * Q);
#pragma unroll
            for (int e = 0; e < 4; ++e) bnext[4 * Q + e] = b4[e];
        });
    };
#else
    auto load_bias = [&]<int TO>() {
        static_for<4>([&]<int Q>() {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + 32 * TO + 8 * Q);
#pragma unroll
            for (int e = 0; e < 4; ++e) amain[TO & 1][4 * Q + e] = b4[e];
        });
    };
#endif
    auto piece = [&]<int TP, int R>() {
        constexpr int f = 32 * TP + 8 * (R >> 2) + (R & 3);   // feature of register R, without the lane half's 4 h (bp / hp carry it)
#ifdef X2_CHEAP_EPI   // (timing experiment: keeps every matrix instruction alive with one VALU per value)
        if constexpr (NHEAD == 0 && STORE) {
            const uint32_t bits = __float_as_uint(acorr[TP & 1][R]) ^ __float_as_uint(amain[TP & 1][R]);
            if constexpr ((R & 1) == 0) { Y[0][2 * TP + (R >> 3)][(R & 7) >> 1] = bits; } else { Y[1][2 * TP + (R >> 3)][(R & 7) >> 1] = bits; }
            return;
        }
#endif
        float v = fmaf(acorr[TP & 1][R], LO_UNSCALE, amain[TP & 1][R]);
        if constexpr (!BIAS_INIT) v += bp[f];
        v = max_keep_nan(v, lo);
#pragma unroll
        for (int k = 0; k < NHEAD; ++k) {
            head[k] = fmaf(v, hp[k * head_stride + f], head[k]);
            asm volatile("" : "+v"(head[k]));   // keep the head sums in this order, one chain per head
        }
        if constexpr (STORE) {
            if constexpr ((R & 1) == 0) {
                pv = v;
            } else {
                uint32_t hw, lw;
                split2_pair(pv, v, hw, lw);
                Y[0][2 * TP + (R >> 3)][(R & 7) >> 1] = hw;
                Y[1][2 * TP + (R >> 3)][(R & 7) >> 1] = lw;
                mx = run_max(mx, pv, v);
            }
        }
    };

    if constexpr (BIAS_INIT) load_bias.template operator()<0>();
    static_for<NB + DEPTH>([&]<int I>() {
        if constexpr (I < NB) {
            if constexpr (I % SLABP == 0) cur = st.advance(lane);
            // The fragment reads are issued here and waited for by COUNT (younger_reads below): written out, because the
            // compiler waits for every outstanding LDS read (lgkmcnt(0)) before the first use of each fragment, the
            // reads issued one block ago included.
#ifdef X2_NO_AREAD    // (timing experiment: fragments of the first blocks of a layer reused)
            if constexpr (I < DEPTH + 1) {
                abuf[I % (DEPTH + 1)][0] = lds_read16<(I % SLABP) * PAIR_BYTES>(cur);
                abuf[I % (DEPTH + 1)][1] = lds_read16<(I % SLABP) * PAIR_BYTES>(cur + 1024);
            }
#elif defined(X2_ONE_READ)   // (timing experiment: one fragment read per block instead of two)
            abuf[I % (DEPTH + 1)][0] = lds_read16<(I % SLABP) * PAIR_BYTES>(cur);
            abuf[I % (DEPTH + 1)][1] = abuf[I % (DEPTH + 1)][0];
#else
#pragma unroll
            for (int s = 0; s < 2; ++s)
                abuf[I % (DEPTH + 1)][s] = lds_read16<(I % SLABP) * PAIR_BYTES>(cur + s * 1024);
#endif
        }
        if constexpr (I >= DEPTH) {
            constexpr int b = I - DEPTH;
            constexpr int to = b / NKS, ks = b % NKS;
            // fragment reads issued after those of block b: two per block, for min(DEPTH, blocks left) blocks
#ifdef X2_NO_AREAD
            constexpr int younger_reads = 0;
#elif defined(X2_ONE_READ)
            constexpr int younger_reads = 1 * ((NB - 1 - b) < DEPTH ? (NB - 1 - b) : DEPTH);
#else
            constexpr int younger_reads = 2 * ((NB - 1 - b) < DEPTH ? (NB - 1 - b) : DEPTH);
#endif
            lds_wait<younger_reads>(abuf[b % (DEPTH + 1)][0], abuf[b % (DEPTH + 1)][1]);
            const u32x4 ah = abuf[b % (DEPTH + 1)][0], al = abuf[b % (DEPTH + 1)][1];
            const u32x4 xh = in_frag.template operator()<ks, 0>(), xl = in_frag.template operator()<ks, 1>();
#ifdef X2_PPS1
            if constexpr (ks == 0) amain[to & 1] = mfma32h(ah, xh, bnext);
            else amain[to & 1] = mfma32h(ah, xh, amain[to & 1]);
#else
            if constexpr (ks == 0 && !BIAS_INIT) amain[to & 1] = mfma32h(ah, xh, zero);
            else amain[to & 1] = mfma32h(ah, xh, amain[to & 1]);                       // hi.hi
#endif
            if constexpr (ks == 0) acorr[to & 1] = mfma32h(ah, xl, zero);
            else acorr[to & 1] = mfma32h(ah, xl, acorr[to & 1]);                       // hi.lo
            acorr[to & 1] = mfma32h(al, xh, acorr[to & 1]);                            // lo.hi
            if constexpr (to > 0) {   // the previous row tile's epilogue, behind this tile's matrix instructions
                static_for<PPS>([&]<int Q>() {
                    constexpr int R = ks * PPS + Q;
                    if constexpr (R < 16) piece.template operator()<to - 1, R>();
                });
            }
            // the other accumulator set is free once the previous tile's epilogue is through (8 k-steps at most)
            if constexpr (BIAS_INIT && to + 1 < NOUT && ks == NKS - 3) load_bias.template operator()<to + 1>();
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    static_for<16>([&]<int R>() { piece.template operator()<NOUT - 1, R>(); });
}

template <bool VIEWDIRS>
__global__ void __launch_bounds__(X2_THREADS, 1)
k_mlp_fwd_f16x2(int64_t n_pts, int pts_per_ray, const float* __restrict__ pts, const float* __restrict__ viewdirs,
                int output_ch, const char* __restrict__ packed, float* __restrict__ out, uint32_t* __restrict__ tile_flags) {
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    char* ring = smem2;
    float* aux = reinterpret_cast<float*>(smem2 + RING2_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5;
    constexpr int UV = VIEWDIRS ? 1 : 0;
    const char* tail = packed + (int64_t)x2_stream_slabs(UV) * SLAB2_BYTES;
    {   // a weight section outside the range the two-piece split is exact for: every tile goes to the three-piece kernel
        const uint32_t* rw = reinterpret_cast<const uint32_t*>(tail + (int64_t)AUX16_FLOATS * 4);
        bool wbad = false;
#pragma unroll
        for (int s = 0; s < x2_sections(UV); ++s) wbad |= out_of_range(__uint_as_float(rw[s]));
        if (wbad) {
            if (tid == 0) tile_flags[blockIdx.x] = 1u;
            return;
        }
    }
    {   // (with the view branch only the first 648 head floats are in use: alpha row, three rgb rows, their biases)
        const float* gaux = reinterpret_cast<const float*>(tail);
        for (int i = tid * 4; i < aux2_floats(UV); i += X2_THREADS * 4)
            *reinterpret_cast<f32x4*>(aux + i) = *reinterpret_cast<const f32x4*>(gaux + i);
    }

    // ---- encodings, split, parked in LDS: fragment (piece s, k-step f) of this lane at ed[(s * NF + f) * 64],
    // f = 0..3 for the position, 4, 5 for the direction.  Each lane half evaluates 15 of the 30 (frequency, axis) pairs of
    // its point's position and 6 of the 12 of its direction with one accurate sincosf each (mlp_layout.hpp).
    constexpr int NF = enc2_frags(UV);
    u32x4* ed = reinterpret_cast<u32x4*>(smem2 + RING2_BYTES + aux2_floats(UV) * 4) + wave * (2 * NF * 64) + lane;
    uint32_t* wave_flags = reinterpret_cast<uint32_t*>(smem2 + RING2_BYTES + aux2_floats(UV) * 4 + X2_WAVES * 2 * NF * 64 * 16);
    bool bad = false;
    auto park = [&](int f, const float* v, float& m) {
        u32x4 hv, lv;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t hw, lw;
            split2_pair(v[2 * w], v[2 * w + 1], hw, lw);
            hv[w] = hw; lv[w] = lw;
            m = run_max(m, v[2 * w], v[2 * w + 1]);
        }
        ed[(0 * NF + f) * 64] = hv;
        ed[(1 * NF + f) * 64] = lv;
    };
    // a point's operands of one layer are spread over its two lanes: the range rule looks at their maximum
    auto both_halves_max = [](float m) { return __builtin_fmaxf(m, __shfl_xor(m, 32, 64)); };
    static_assert(X2_WG_PTS == 128, "one flag per 128 points");
    const int64_t p = (int64_t)blockIdx.x * X2_WG_PTS + wave * WAVE_PTS2 + (lane & 31);
    const bool valid = p < n_pts;
    const int64_t pl = valid ? p : n_pts - 1;
    {
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
        float m = 0.0f;
#pragma unroll
        for (int f = 0; f < 4; ++f) park(f, e + 8 * f, m);
        bad |= out_of_range(both_halves_max(m));
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
            float md = 0.0f;
            park(4, dd, md);
            park(5, dd + 8, md);
            bad |= out_of_range(both_halves_max(md));
        }
    }

    Stream2 st;
    st.ring = ring;
    st.mine = ring + wave * WAVE_BLOCKS2 * 1024;
    st.g = packed + wave * WAVE_BLOCKS2 * 1024 + lane * 16;
    st.next = 0;
    st.nslabs = x2_stream_slabs(UV);
    __syncthreads();
    st.fetch(0);
    st.fetch(1);
    const char* cur = nullptr;

    u32x4 X[2][16], Y[2][16];
    float sig[3] = {0.0f, 0.0f, 0.0f}, rgb[3] = {0.0f, 0.0f, 0.0f};
    float mx = 0.0f;
    const float NOCLAMP = -__builtin_inff();
    const float* bias = aux + AUX16_BIAS;
    const float* hw = aux + AUX16_HEAD;
    auto fromE = [&]<int ks, int s>() -> u32x4 { return ed[(s * NF + ks) * 64]; };
    auto fromX = [&]<int ks, int s>() -> u32x4 { return X[s][ks]; };
    auto fromY = [&]<int ks, int s>() -> u32x4 { return Y[s][ks]; };
    auto fromXE = [&]<int ks, int s>() -> u32x4 {
        if constexpr (ks < 16) return X[s][ks]; else return ed[(s * NF + ks - 16) * 64];
    };
    auto fromXD = [&]<int ks, int s>() -> u32x4 {
        if constexpr (ks < 16) return X[s][ks]; else return ed[(s * NF + 4 + ks - 16) * 64];
    };
    auto check = [&]() {
        bad |= out_of_range(both_halves_max(mx));
        mx = 0.0f;
    };
#define X2_LAYER(NOUT, NKS, NHEAD, STORE, IN, OUT, BIAS, LO, HEADW, HSTRIDE, HEAD) \
    layer_x2<NOUT, NKS, NHEAD, STORE>(st, lane, h, cur, IN, OUT, BIAS, LO, HEADW, HSTRIDE, HEAD, mx)

    X2_LAYER(8, 4, 0, true, fromE, X, bias, 0.0f, hw, 0, sig);  check();                                 // L0
    X2_LAYER(8, 16, 0, true, fromX, Y, bias + 1 * 256, 0.0f, hw, 0, sig);  check();                      // L1
    X2_LAYER(8, 16, 0, true, fromY, X, bias + 2 * 256, 0.0f, hw, 0, sig);  check();                      // L2
    X2_LAYER(8, 16, 0, true, fromX, Y, bias + 3 * 256, 0.0f, hw, 0, sig);  check();                      // L3
    X2_LAYER(8, 16, 0, true, fromY, X, bias + 4 * 256, 0.0f, hw, 0, sig);  check();                      // L4
    X2_LAYER(8, 20, 0, true, fromXE, Y, bias + 5 * 256, 0.0f, hw, 0, sig);  check();                     // L5 (skip)
    X2_LAYER(8, 16, 0, true, fromY, X, bias + 6 * 256, 0.0f, hw, 0, sig);  check();                      // L6
    auto both_halves = [](float v) { return v + __shfl_xor(v, 32, 64); };
    if constexpr (VIEWDIRS) {
        X2_LAYER(8, 16, 1, true, fromX, Y, bias + 7 * 256, 0.0f, hw, 0, sig);  check();                  // L7 + alpha head
        X2_LAYER(8, 16, 0, true, fromY, X, bias + 8 * 256, NOCLAMP, hw, 0, sig);  check();               // feature
        X2_LAYER(4, 18, 3, false, fromXD, Y, bias + 9 * 256, 0.0f, hw + 256, 128, rgb);                  // views + rgb heads
        const float* hb = hw + 256 + 384;
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = both_halves(rgb[k]) + hb[1 + k];
        o[3] = both_halves(sig[0]) + hb[0];
        if (valid && h == 0) *reinterpret_cast<f32x4*>(out + p * 4) = o;
    } else {
        // output_linear: rows 0..2 as fp32 head dot products inside the L7 epilogue, rows 3..7 from the pieces afterwards
        X2_LAYER(8, 16, 3, true, fromX, Y, bias + 7 * 256, 0.0f, hw, 256, sig);  check();
        for (int ch = 0; ch < output_ch; ++ch) {
            float s;
            if (ch < 3) {
                s = ch == 0 ? sig[0] : (ch == 1 ? sig[1] : sig[2]);
            } else {
                const float* wp = hw + ch * 256;
                s = 0.0f;
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    const f16x8 yh = __builtin_bit_cast(f16x8, Y[0][ks]), yl = __builtin_bit_cast(f16x8, Y[1][ks]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float y = fmaf((float)yl[j], LO_UNSCALE, (float)yh[j]);
                        s = fmaf(y, wp[32 * (ks >> 1) + kslot32(ks & 1, h, j)], s);
                    }
                }
            }
            s = both_halves(s) + hw[8 * 256 + ch];
            if (valid && h == 0) out[p * output_ch + ch] = s;
        }
    }
#undef X2_LAYER
    // one flag per 128-point tile: any lane of any wave that left the range
    const bool wave_bad = __any(bad);
    if (lane == 0) wave_flags[wave] = wave_bad ? 1u : 0u;
    __syncthreads();
    if (tid == 0) tile_flags[blockIdx.x] = wave_flags[0] | wave_flags[1] | wave_flags[2] | wave_flags[3];
}

// flat parameters -> stream of pairs + fp32 aux tables; `range` (RANGE_WORDS words behind them, zeroed beforehand)
// receives the largest |weight| of every section as uint32 bits (non-negative floats order like their bits; a NaN
// weight sorts above infinity and so counts as out of range)
__global__ void __launch_bounds__(256) k_pack_f16x2(int use_viewdirs, int output_ch, const float* __restrict__ params,
                                                    char* __restrict__ out) {
    const int64_t npairs = x2_stream_blocks(use_viewdirs);
    const int64_t elems = npairs * 512;           // one per (pair, lane, slot)
    const int64_t total = elems + AUX16_FLOATS;
    unsigned short* s16 = reinterpret_cast<unsigned short*>(out);
    float* aux = reinterpret_cast<float*>(out + npairs * PAIR_BYTES);
    uint32_t* range = reinterpret_cast<uint32_t*>(out + npairs * PAIR_BYTES + (int64_t)AUX16_FLOATS * 4);
    // a wave's 64 consecutive elements lie in one block, hence in one section
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total + 63) / 64 * 64; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t mag = 0u;
        if (i < elems) {
            const int64_t t = i >> 9;
            const int w = (int)(i & 511);
            const int64_t src = x2_stream_source(use_viewdirs, output_ch, (int)t, w >> 3, w & 7);
            const float x = src < 0 ? 0.0f : params[src];
            _Float16 h, l;
            split2(x, h, l);
            unsigned short* dst = s16 + t * (PAIR_BYTES / 2) + w;
            dst[0] = __builtin_bit_cast(unsigned short, h);
            dst[512] = __builtin_bit_cast(unsigned short, l);
            mag = __float_as_uint(x) & 0x7fffffffu;
        } else if (i < total) {
            const int64_t src = aux16_source(use_viewdirs, output_ch, (int)(i - elems));
            aux[i - elems] = src < 0 ? 0.0f : params[src];
        }
        // wave maximum, then one atomic per wave
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)mag, off, 64);
            mag = o > mag ? o : mag;
        }
        if ((threadIdx.x & 63) == 0 && i < elems && mag != 0u) atomicMax(range + x2_block_section((int)(i >> 9)), mag);
    }
}

constexpr size_t x2_lds_bytes(int uv) {
    return RING2_BYTES + aux2_floats(uv) * sizeof(float) + X2_WAVES * 2 * enc2_frags(uv) * 64 * 16 + X2_WAVES * sizeof(uint32_t);
}
static_assert(x2_lds_bytes(1) <= 160 * 1024 && x2_lds_bytes(0) <= 160 * 1024, "LDS of one compute unit");

}  // namespace

int nerf_mlp_fwd_bf16x3_marked(int64_t n_pts, int pts_per_ray, const float* pts, const float* viewdirs, int use_viewdirs,
                               int output_ch, const void* packed, float* out, float* act, const uint32_t* tile_flags,
                               void* stream);   // mlp_fwd_bf16x3.hip

extern "C" {

int64_t nerf_mlp_packed_f16x2_bytes(int use_viewdirs) { return x2_packed_bytes(use_viewdirs); }
int64_t nerf_mlp_f16x2_tile_points(void) { return X2_WG_PTS; }

int nerf_pack_weights_f16x2(int use_viewdirs, int output_ch, const float* params, void* packed, void* stream) {
    NERF_REQUIRE(params && packed, "nerf_pack_weights_f16x2: null pointer");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_pack_weights_f16x2: output_ch=%d not in 1..8", output_ch);
    char* range = reinterpret_cast<char*>(packed) + x2_packed_bytes(use_viewdirs) - RANGE_WORDS * 4;
    hipError_t e = hipMemsetAsync(range, 0, RANGE_WORDS * 4, as_stream(stream));
    if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_pack_weights_f16x2: clearing the range words: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_pack_f16x2, dim3(1024), dim3(256), 0, as_stream(stream), use_viewdirs, output_ch, params,
                       reinterpret_cast<char*>(packed));
    return launched("nerf_pack_weights_f16x2");
}

int nerf_mlp_fwd_f16x2(int64_t n_pts, int pts_per_ray, const float* pts, const float* viewdirs, int use_viewdirs,
                       int output_ch, const void* packed, const void* packed_bf16x3, float* out, uint32_t* tile_flags,
                       void* stream) {
    NERF_REQUIRE(n_pts >= 0, "nerf_mlp_fwd_f16x2: n_pts=%lld", (long long)n_pts);
    if (n_pts == 0) return 0;
    NERF_REQUIRE(pts && packed && out && tile_flags, "nerf_mlp_fwd_f16x2: null pointer");
    NERF_REQUIRE(!use_viewdirs || (viewdirs && pts_per_ray >= 1), "nerf_mlp_fwd_f16x2: view branch needs viewdirs and pts_per_ray >= 1");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_mlp_fwd_f16x2: output_ch=%d not in 1..8", output_ch);
    NERF_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                 "nerf_mlp_fwd_f16x2: packed and out must be 16-byte aligned");
    const int64_t blocks = (n_pts + X2_WG_PTS - 1) / X2_WG_PTS;
    NERF_REQUIRE(blocks <= 0x7fffffff, "nerf_mlp_fwd_f16x2: n_pts=%lld too large for one launch", (long long)n_pts);
    auto go = [&](auto kernel, int ppr, const float* vd, int och) -> int {
        const size_t lds = x2_lds_bytes(use_viewdirs);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_fwd_f16x2: LDS attribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(X2_THREADS), lds, as_stream(stream), n_pts, ppr, pts,
                           vd, och, reinterpret_cast<const char*>(packed), out, tile_flags);
        return launched("nerf_mlp_fwd_f16x2");
    };
    const int rc = use_viewdirs ? go(k_mlp_fwd_f16x2<true>, pts_per_ray, viewdirs, 4) : go(k_mlp_fwd_f16x2<false>, 1, nullptr, output_ch);
    if (rc != 0 || packed_bf16x3 == nullptr) return rc;
    // the marked tiles again, on the three-piece kernel
    return nerf_mlp_fwd_bf16x3_marked(n_pts, pts_per_ray, pts, viewdirs, use_viewdirs, output_ch, packed_bf16x3, out, nullptr,
                                      tile_flags, stream);
}

}  // extern "C"
