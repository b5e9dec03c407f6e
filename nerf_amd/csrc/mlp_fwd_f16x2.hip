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
// Dataflow as in mlp_fwd_bf16x3.hip: activations never leave the register file (2 x 128 registers, one wave per SIMD,
// four waves = 128 points per workgroup); weights stream through a three-slab LDS ring by LDS-DMA behind a bare
// s_barrier; bias = accumulator start value; a tile's epilogue (scale-and-add of the two accumulators, ReLU, heads, the
// split of its values) rides behind the next tile's matrix instructions.
#include "common.hpp"
#include "mlp_core.hpp"
#include "mlp_tile16.hpp"
#include "mlp_bf16_layout.hpp"
#include "mlp_x2.hpp"
#include "mlp_x3.hpp"

using namespace nerfhip;
using namespace nerflayout;
using namespace nerfmlp;
using namespace nerftile16;
using namespace nerfbf16;
using namespace nerfx2;

namespace {

// LDS behind the ring: the fp32 tables, then the split encodings (fragments per piece and wave), then one flag per wave
__host__ __device__ constexpr int aux2_floats(int use_viewdirs) { return use_viewdirs ? AUX16_HEAD + 648 : AUX16_FLOATS; }
__host__ __device__ constexpr int enc2_frags(int use_viewdirs) { return (use_viewdirs ? 3 : 2) * NCT2; }

// One layer: NOUT row tiles (16 features each) x NKS k-steps (32 input features each) x NCT2 column tiles.
// in_frag<CT, ks, S>() = piece S (0 hi, 1 lo) of the input fragment of column tile CT, k-step ks; Y[S][CT][..] receives
// the pieces of the output (STORE); mx[CT] the largest magnitude this lane has produced for column tile CT, kept as the
// fp16 bits of the leading pieces (two running maxima per word; SIGNED = the values may be negative).
// SAVE / MASK (training): as layer_x3 of mlp_fwd_bf16x3.hip -- the unsplit fp32 values of a finished tile are stored in
// the blocked layout of mlp_layout.hpp, their ReLU mask bits collected in mb.
template <int NOUT, int NKS, int NHEAD, bool STORE, bool SAVE, bool MASK, bool SIGNED, typename InFrag>
__device__ __forceinline__ void layer_x2(Stream2& st, int lane, int g, const char*& cur, InFrag in_frag,
                                         u32x4 (&Y)[2][NCT2][8], const float* bias_lds, float lo, const float* head_lds,
                                         int head_stride, float (&head)[NCT2][3], float* (&srow)[NCT2],
                                         uint32_t (&mb)[NCT2][4], uint32_t (&mx)[NCT2]) {
    constexpr int NB = NOUT * NKS, DEPTH = 3, NPIECE = 4 * NCT2;
    constexpr bool BIAS_INIT = NKS >= 7;
    constexpr int PPS = (NPIECE + (NKS < 8 ? NKS : 8) - 1) / (NKS < 8 ? NKS : 8);   // epilogue values per k-step
    u32x4 abuf[DEPTH + 1][2];
    f32x4 amain[2][NCT2], acorr[2][NCT2];         // [alternating set][column tile]
    const float* bp = bias_lds + 4 * g;
    const float* hp = head_lds + 4 * g;
    const f32x4 zero = {0, 0, 0, 0};

    f32x4 bnext = {0, 0, 0, 0};                   // the bias column of a tile is the C operand of its first MFMAs
    f32x4 sv = {0, 0, 0, 0};                      // SAVE: the four values of (tile, column tile) being collected
    float pv = 0.0f;                              // STORE: the even value of the pair being collected
    if constexpr (MASK) {
#pragma unroll
        for (int ct = 0; ct < NCT2; ++ct) mb[ct][0] = mb[ct][1] = mb[ct][2] = mb[ct][3] = 0u;
    }
    auto load_bias = [&]<int TO>() { bnext = *reinterpret_cast<const f32x4*>(bp + 16 * TO); };
    auto piece = [&]<int TP, int K>() {           // K = 4 * column tile + register: one value of the tile per piece
        constexpr int ct = K >> 2, r = K & 3;
        float v = fmaf(acorr[TP & 1][ct][r], LO_UNSCALE, amain[TP & 1][ct][r]);
        if constexpr (!BIAS_INIT) v += bp[16 * TP + r];
        v = max_keep_nan(v, lo);
#pragma unroll
        for (int k = 0; k < NHEAD; ++k) {
            head[ct][k] = fmaf(v, hp[k * head_stride + 16 * TP + r], head[ct][k]);
            asm volatile("" : "+v"(head[ct][k]));   // keep the head sums here (see mlp_fwd_bf16.hip)
        }
        if constexpr (MASK) {   // v >= 0 after the ReLU, so "positive" = "bits != 0"
            const uint32_t one = __float_as_uint(v) < 1u ? __float_as_uint(v) : 1u;
            mb[ct][TP >> 2] |= one << (8 * (TP & 3) + r);
        }
        if constexpr (SAVE) {
            sv[r] = v;
            if constexpr (r == 3) *reinterpret_cast<f32x4*>(srow[ct] + 4 * TP * GROUP_FLOATS) = sv;
        }
        if constexpr (STORE) {
            if constexpr ((r & 1) == 0) {
                pv = v;
            } else {
                constexpr int f = TP >> 1, w = 2 * (TP & 1) + (r >> 1);
                uint32_t h, l;
                split2_pair(pv, v, h, l);
                Y[0][ct][f][w] = h;
                Y[1][ct][f][w] = l;
                mx[ct] = run_max_h(mx[ct], SIGNED ? (h & 0x7fff7fffu) : h);   // (behind a ReLU the sign bits are clear already)
            }
        }
    };

    if constexpr (BIAS_INIT) load_bias.template operator()<0>();
    static_for<NB + DEPTH>([&]<int I>() {
        if constexpr (I < NB) {
            if constexpr (I % SLABP == 0) cur = st.template advance<(SAVE && I > 0) ? x2_stores_since_open(I, NKS) : 0>(lane);
#pragma unroll
            for (int s = 0; s < 2; ++s)
                abuf[I % (DEPTH + 1)][s] = *reinterpret_cast<const u32x4*>(cur + (I % SLABP) * PAIR_BYTES + s * 1024);
        }
        if constexpr (I >= DEPTH) {
            constexpr int b = I - DEPTH;
            constexpr int to = b / NKS, ks = b % NKS;
            const u32x4 ah = abuf[b % (DEPTH + 1)][0], al = abuf[b % (DEPTH + 1)][1];
            static_for<NCT2>([&]<int CT>() {        // hi.hi
                const u32x4 xh = in_frag.template operator()<CT, ks, 0>();
                if constexpr (ks == 0) amain[to & 1][CT] = mfma16h(ah, xh, bnext);   // (bnext stays zero where the bias is added later)
                else amain[to & 1][CT] = mfma16h(ah, xh, amain[to & 1][CT]);
            });
            static_for<NCT2>([&]<int CT>() {        // hi.lo
                const u32x4 xl = in_frag.template operator()<CT, ks, 1>();
                if constexpr (ks == 0) acorr[to & 1][CT] = mfma16h(ah, xl, zero);
                else acorr[to & 1][CT] = mfma16h(ah, xl, acorr[to & 1][CT]);
            });
            static_for<NCT2>([&]<int CT>() {        // lo.hi
                const u32x4 xh = in_frag.template operator()<CT, ks, 0>();
                acorr[to & 1][CT] = mfma16h(al, xh, acorr[to & 1][CT]);
            });
            if constexpr (to > 0) {   // the previous row tile's epilogue, value by value, behind this tile's MFMAs
                static_for<PPS>([&]<int Q>() {
                    constexpr int K = ks * PPS + Q;
                    if constexpr (K < NPIECE) piece.template operator()<to - 1, K>();
                });
            }
            if constexpr (BIAS_INIT && to + 1 < NOUT && ks == NKS - 3) load_bias.template operator()<to + 1>();
        }
        // Order inside a block: its two fragment reads, then every matrix instruction followed by two of the epilogue's
        // vector instructions.  Left alone the scheduler issues the matrix instructions in runs of four and the vector
        // instructions in runs of eight, during which the matrix pipe drains (1 % of a render's launch).
        if constexpr (I >= DEPTH) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            static_for<3 * NCT2>([&]<int J>() {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            });
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    static_for<NPIECE>([&]<int K>() { piece.template operator()<NOUT - 1, K>(); });
}

template <bool VIEWDIRS, bool SAVE>
__global__ void __launch_bounds__(X2_THREADS, 1)
k_mlp_fwd_f16x2(int64_t n_pts, int pts_per_ray, const float* __restrict__ pts, const float* __restrict__ viewdirs,
                int output_ch, const char* __restrict__ packed, float* __restrict__ out, float* __restrict__ act,
                uint32_t* __restrict__ tile_flags) {
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    char* ring = smem2;
    float* aux = reinterpret_cast<float*>(smem2 + RING2_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4;
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

    // ---- encodings, split, parked in LDS: fragment (piece s, slot f) of this lane at ed[(s * NF + f) * 64],
    // f = 2 ct + k-step for the position, 2 NCT2 + ct for the direction
    constexpr int NF = enc2_frags(UV);
    u32x4* ed = reinterpret_cast<u32x4*>(smem2 + RING2_BYTES + aux2_floats(UV) * 4) + wave * (2 * NF * 64) + lane;
    uint32_t* wave_flags = reinterpret_cast<uint32_t*>(smem2 + RING2_BYTES + aux2_floats(UV) * 4 + X2_WAVES * 2 * NF * 64 * 16);
    uint32_t mx[NCT2];
    bool bad = false;
    auto park = [&](int f, const float* v, float& m) {
        u32x4 h, l;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t hw, lw;
            split2_pair(v[2 * w], v[2 * w + 1], hw, lw);
            h[w] = hw; l[w] = lw;
            m = run_max(m, v[2 * w], v[2 * w + 1]);
        }
        ed[(0 * NF + f) * 64] = h;
        ed[(1 * NF + f) * 64] = l;
    };
    // a point's operands of one layer are spread over its four lanes (groups g): the range rule looks at their maximum
    auto over_groups_max = [](float m) {
        m = __builtin_fmaxf(m, __shfl_xor(m, 16, 64));
        return __builtin_fmaxf(m, __shfl_xor(m, 32, 64));
    };
    auto check = [&]() {
#pragma unroll
        for (int ct = 0; ct < NCT2; ++ct) {
            uint32_t m = (mx[ct] & 0xffffu) > (mx[ct] >> 16) ? (mx[ct] & 0xffffu) : (mx[ct] >> 16);
            const uint32_t m1 = (uint32_t)__shfl_xor((int)m, 16, 64);
            m = m > m1 ? m : m1;
            const uint32_t m2 = (uint32_t)__shfl_xor((int)m, 32, 64);
            m = m > m2 ? m : m2;
            bad |= out_of_range_h(m);
            mx[ct] = 0u;
        }
    };
    int64_t p[NCT2];
    bool valid[NCT2];
    const int c0 = enc16_first_pair(g);
    static_assert(X2_WG_PTS == TILE_PTS, "one workgroup per tile of the saved activations");
    float* const tile = SAVE ? act + (int64_t)blockIdx.x * act_rows(UV) * TILE_PTS : nullptr;
#pragma unroll
    for (int ct = 0; ct < NCT2; ++ct) {
        p[ct] = (int64_t)blockIdx.x * X2_WG_PTS + wave * WAVE_PTS2 + ct * 16 + (lane & 15);
        valid[ct] = p[ct] < n_pts;
        const int64_t pl = valid[ct] ? p[ct] : n_pts - 1;
        const float x0 = pts[pl * 3 + 0], x1 = pts[pl * 3 + 1], x2 = pts[pl * 3 + 2];
        float e[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = c0 + i, k = c / 3, d = c - 3 * k;   // (groups 2, 3: pair 7 is overwritten below)
            const float arg = (d == 0 ? x0 : (d == 1 ? x1 : x2)) * __int_as_float((127 + k) << 23);
            sincosf(arg, &e[2 * i], &e[2 * i + 1]);
        }
        if (g == 2) { e[14] = x0; e[15] = x1; }
        if (g == 3) { e[14] = x2; e[15] = 0.0f; }
        float m = 0.0f;
        park(2 * ct + 0, e, m);
        park(2 * ct + 1, e + 8, m);
        bad |= out_of_range(over_groups_max(m));
        const int pw = wave * WAVE_PTS2 + ct * 16 + (lane & 15);   // point inside the tile
        if constexpr (SAVE) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int f = enc16_pts_feature(q >> 3, g, q & 7), row = ACT_ENC_PTS + (f < 0 ? 63 : f);
                tile[(row >> 2) * GROUP_FLOATS + pw * 4 + (row & 3)] = (valid[ct] && f >= 0) ? e[q] : 0.0f;
            }
        }
        if (VIEWDIRS) {
            const int64_t ray = pl / pts_per_ray;
            const float v0 = viewdirs[ray * 3 + 0], v1 = viewdirs[ray * 3 + 1], v2 = viewdirs[ray * 3 + 2];
            float dd[8];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int c = 3 * g + i, k = c / 3, d = c - 3 * k;
                const float arg = (d == 0 ? v0 : (d == 1 ? v1 : v2)) * __int_as_float((127 + k) << 23);
                sincosf(arg, &dd[2 * i], &dd[2 * i + 1]);
            }
            dd[6] = g == 0 ? v0 : (g == 1 ? v2 : 0.0f);
            dd[7] = g == 0 ? v1 : 0.0f;
            float md = 0.0f;
            park(2 * NCT2 + ct, dd, md);
            bad |= out_of_range(over_groups_max(md));
            if constexpr (SAVE) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int f = enc16_dirs_feature(g, q);
                    // the five slots without a feature zero the padding rows 27..31
                    const int row = ACT_ENC_DIRS + (f >= 0 ? f : 27 + (g == 1 ? 0 : 2 * (g - 2) + 1 + (q - 6)));
                    tile[(row >> 2) * GROUP_FLOATS + pw * 4 + (row & 3)] = (valid[ct] && f >= 0) ? dd[q] : 0.0f;
                }
            }
        }
        mx[ct] = 0u;
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

    u32x4 X[2][NCT2][8], Y[2][NCT2][8];
    float sig[NCT2][3], rgb[NCT2][3];
#pragma unroll
    for (int ct = 0; ct < NCT2; ++ct)
#pragma unroll
        for (int k = 0; k < 3; ++k) sig[ct][k] = rgb[ct][k] = 0.0f;
    const float NOCLAMP = -__builtin_inff();
    const float* bias = aux + AUX16_BIAS;
    const float* hw = aux + AUX16_HEAD;
    auto fromE = [&]<int ct, int ks, int s>() -> u32x4 { return ed[(s * NF + 2 * ct + ks) * 64]; };
    auto fromX = [&]<int ct, int ks, int s>() -> u32x4 { return X[s][ct][ks]; };
    auto fromY = [&]<int ct, int ks, int s>() -> u32x4 { return Y[s][ct][ks]; };
    auto fromXE = [&]<int ct, int ks, int s>() -> u32x4 {
        if constexpr (ks < 8) return X[s][ct][ks]; else return ed[(s * NF + 2 * ct + ks - 8) * 64];
    };
    auto fromXD = [&]<int ct, int ks, int s>() -> u32x4 {
        if constexpr (ks < 8) return X[s][ct][ks]; else return ed[(s * NF + 2 * NCT2 + ct) * 64];
    };

    // training: per column tile, this lane's slot in the tile (row group g, its point) and the mask-word writer
    float* lane_rows[NCT2];
    uint32_t mb[NCT2][4];
#pragma unroll
    for (int ct = 0; ct < NCT2; ++ct) {
        lane_rows[ct] = SAVE ? tile + g * GROUP_FLOATS + (wave * WAVE_PTS2 + ct * 16 + (lane & 15)) * 4 : nullptr;
        mb[ct][0] = mb[ct][1] = mb[ct][2] = mb[ct][3] = 0u;
    }
    // The backward-data kernels read one 16-byte word per masked layer and lane (point, half h): lanes (point, g = h) and
    // (point, g = h + 2) of this kernel hold its bits between them, in nibbles that alternate (mlp_fwd_x4.hip, save_mask).
    auto save_mask = [&](int m) {
        if constexpr (SAVE) {
#pragma unroll
            for (int ct = 0; ct < NCT2; ++ct) {
                u32x4 w;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const uint32_t mine = mb[ct][d] << (4 * (g >> 1));
                    w[d] = mine | (uint32_t)__shfl_xor((int)mine, 32, 64);
                }
                const int pw = wave * WAVE_PTS2 + ct * 16 + (lane & 15);
                if (g < 2) *reinterpret_cast<u32x4*>(tile + act_mask_offset(UV, m, pw >> 5, g * 32 + (pw & 31))) = w;
            }
        }
    };
    auto rows = [&](int row, float* (&dst)[NCT2]) {
#pragma unroll
        for (int ct = 0; ct < NCT2; ++ct) dst[ct] = lane_rows[ct] + (row / 4) * GROUP_FLOATS;
    };
    float* sr[NCT2];
#define X2_LAYER(NOUT, NKS, NHEAD, STORE, MASKED, IN, OUT, BIAS, LO, HEADW, HSTRIDE, HEAD, ROW)                        \
    rows(ROW, sr);                                                                                                    \
    layer_x2<NOUT, NKS, NHEAD, STORE, SAVE, SAVE && MASKED, !(MASKED)>(st, lane, g, cur, IN, OUT, BIAS, LO, HEADW, HSTRIDE, HEAD, sr, mb, mx)

    X2_LAYER(16, 2, 0, true, true, fromE, X, bias, 0.0f, hw, 0, sig, ACT_H0);                               // L0
    save_mask(0); check();
    X2_LAYER(16, 8, 0, true, true, fromX, Y, bias + 1 * 256, 0.0f, hw, 0, sig, ACT_H0 + 1 * 256);           // L1
    save_mask(1); check();
    X2_LAYER(16, 8, 0, true, true, fromY, X, bias + 2 * 256, 0.0f, hw, 0, sig, ACT_H0 + 2 * 256);           // L2
    save_mask(2); check();
    X2_LAYER(16, 8, 0, true, true, fromX, Y, bias + 3 * 256, 0.0f, hw, 0, sig, ACT_H0 + 3 * 256);           // L3
    save_mask(3); check();
    X2_LAYER(16, 8, 0, true, true, fromY, X, bias + 4 * 256, 0.0f, hw, 0, sig, ACT_H0 + 4 * 256);           // L4
    save_mask(4); check();
    X2_LAYER(16, 10, 0, true, true, fromXE, Y, bias + 5 * 256, 0.0f, hw, 0, sig, ACT_H0 + 5 * 256);         // L5 (skip)
    save_mask(5); check();
    X2_LAYER(16, 8, 0, true, true, fromY, X, bias + 6 * 256, 0.0f, hw, 0, sig, ACT_H0 + 6 * 256);           // L6
    save_mask(6); check();
    auto over_groups = [](float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); };
    if constexpr (VIEWDIRS) {
        X2_LAYER(16, 8, 1, true, true, fromX, Y, bias + 7 * 256, 0.0f, hw, 0, sig, ACT_H0 + 7 * 256);       // L7 + alpha head
        save_mask(7); check();
        X2_LAYER(16, 8, 0, true, false, fromY, X, bias + 8 * 256, NOCLAMP, hw, 0, sig, ACT_FEAT);          // feature
        check();
        X2_LAYER(8, 9, 3, false, true, fromXD, Y, bias + 9 * 256, 0.0f, hw + 256, 128, rgb, ACT_VIEWS);    // views + rgb heads
        save_mask(8);
        const float* hb = hw + 256 + 384;
#pragma unroll
        for (int ct = 0; ct < NCT2; ++ct) {
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 3; ++k) o[k] = over_groups(rgb[ct][k]) + hb[1 + k];
            o[3] = over_groups(sig[ct][0]) + hb[0];
            if (valid[ct] && g == (ct & 3)) *reinterpret_cast<f32x4*>(out + p[ct] * 4) = o;
        }
    } else {
        // output_linear: rows 0..2 as fp32 head dot products inside the L7 epilogue, rows 3..7 from the pieces afterwards
        X2_LAYER(16, 8, 3, true, true, fromX, Y, bias + 7 * 256, 0.0f, hw, 256, sig, ACT_H0 + 7 * 256);
        save_mask(7); check();
        for (int ch = 0; ch < output_ch; ++ch) {
#pragma unroll
            for (int ct = 0; ct < NCT2; ++ct) {
                float s;
                if (ch < 3) {
                    s = ch == 0 ? sig[ct][0] : (ch == 1 ? sig[ct][1] : sig[ct][2]);
                } else {
                    const float* wp = hw + ch * 256;
                    s = 0.0f;
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        const f16x8 yh = __builtin_bit_cast(f16x8, Y[0][ct][ks]), yl = __builtin_bit_cast(f16x8, Y[1][ct][ks]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float y = fmaf((float)yl[j], LO_UNSCALE, (float)yh[j]);
                            s = fmaf(y, wp[32 * ks + kslot_feature(g, j)], s);
                        }
                    }
                }
                s = over_groups(s) + hw[8 * 256 + ch];
                if (valid[ct] && g == (ct & 3)) out[p[ct] * output_ch + ch] = s;
            }
        }
    }
#undef X2_LAYER
    // one flag per 128-point tile: any lane of any wave that left the range
    const bool wave_bad = __any(bad);
    if (lane == 0) wave_flags[wave] = wave_bad ? 1u : 0u;
    __syncthreads();
    if (tid == 0) tile_flags[blockIdx.x] = wave_flags[0] | wave_flags[1] | wave_flags[2] | wave_flags[3];
}

// Largest |weight| per section, block by block: a wave's maximum goes to a per-block table in LDS, and a block sends
// each section's maximum to the range words once, at its end.  (One global atomicMax per wave, 4,640 of them on ten
// addresses, ran one after the other: 48 of the gather's 55 us.)
struct BlockRange {
    uint32_t* lds;   // RANGE_WORDS words, zeroed by begin()
    __device__ __forceinline__ void begin() {
        if (threadIdx.x < RANGE_WORDS) lds[threadIdx.x] = 0u;
        __syncthreads();
    }
    // all 64 lanes of a wave call this with their magnitudes of ONE section (a wave's elements lie in one block)
    __device__ __forceinline__ void add(uint32_t mag, int section) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)mag, off, 64);
            mag = o > mag ? o : mag;
        }
        if ((threadIdx.x & 63) == 0 && mag != 0u) atomicMax(lds + section, mag);
    }
    __device__ __forceinline__ void end(uint32_t* range) {
        __syncthreads();
        if (threadIdx.x < RANGE_WORDS && lds[threadIdx.x] != 0u) atomicMax(range + threadIdx.x, lds[threadIdx.x]);
    }
};

// flat parameters -> stream of pairs + fp32 aux tables; `range` (RANGE_WORDS words behind them, zeroed beforehand)
// receives the largest |weight| of every section as uint32 bits (non-negative floats order like their bits; a NaN
// weight sorts above infinity and so counts as out of range)
__global__ void __launch_bounds__(256) k_pack_f16x2(int use_viewdirs, int output_ch, const float* __restrict__ params,
                                                    char* __restrict__ out) {
    const int64_t npairs = bf16_stream_blocks(use_viewdirs);
    const int64_t elems = npairs * 512;           // one per (pair, lane, slot)
    const int64_t total = elems + AUX16_FLOATS;
    unsigned short* s16 = reinterpret_cast<unsigned short*>(out);
    float* aux = reinterpret_cast<float*>(out + npairs * PAIR_BYTES);
    uint32_t* range = reinterpret_cast<uint32_t*>(out + npairs * PAIR_BYTES + (int64_t)AUX16_FLOATS * 4);
    __shared__ uint32_t smax[RANGE_WORDS];
    BlockRange br{smax};
    br.begin();
    // a wave's 64 consecutive elements lie in one block, hence in one section; every thread of a block makes the same
    // number of trips (the bound is a multiple of the block size)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total + 255) / 256 * 256; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t mag = 0u;
        if (i < elems) {
            const int64_t t = i >> 9;
            const int w = (int)(i & 511);
            const int64_t src = bf16_stream_source(use_viewdirs, output_ch, (int)t, w >> 3, w & 7);
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
        br.add(mag, i < elems ? x2_block_section((int)(i >> 9)) : 0);
    }
    br.end(range);
}

// The same stream as ONE gather through a map (nerf_gather_split2): element pair (pair block, w, w + 1) reads its two
// sources, splits them and writes one 4-byte word per piece; then the fp32 tables; the range words as in k_pack_f16x2;
// behind them a second stream of pair blocks (the transposed one of mlp_bwd_f16x2.hip), whose elements are the same
// weights again and do not count for the range words.
__device__ __forceinline__ void gather_split2_elements(const int32_t* __restrict__ table, const float* __restrict__ params,
                                                       char* __restrict__ out, int64_t npairs, int64_t n_aux, int64_t npairs1) {
    const int64_t words = npairs * 256;            // two stream elements each
    const int64_t first = words + n_aux, total = first + npairs1 * 256;
    uint32_t* range = reinterpret_cast<uint32_t*>(out + npairs * PAIR_BYTES + n_aux * 4);
    char* out1 = out + npairs * PAIR_BYTES + n_aux * 4 + RANGE_WORDS * 4;
    __shared__ uint32_t smax[RANGE_WORDS];
    BlockRange br{smax};
    br.begin();
    auto pair = [&](int64_t at, uint32_t* dst) -> uint32_t {   // table entries 2 at, 2 at + 1 -> one word per piece
        const int2 src = reinterpret_cast<const int2*>(table)[at];
        const float x0 = src.x < 0 ? 0.0f : params[src.x], x1 = src.y < 0 ? 0.0f : params[src.y];
        uint32_t hw, lw;
        split2_pair(x0, x1, hw, lw);
        dst[0] = hw;
        dst[256] = lw;
        const uint32_t m0 = __float_as_uint(x0) & 0x7fffffffu, m1 = __float_as_uint(x1) & 0x7fffffffu;
        return m0 > m1 ? m0 : m1;
    };
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total + 255) / 256 * 256; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t mag = 0u;
        if (i < words) {
            mag = pair(i, reinterpret_cast<uint32_t*>(out + (i >> 8) * PAIR_BYTES) + (i & 255));
        } else if (i < first) {
            const int32_t src = table[2 * words + (i - words)];
            reinterpret_cast<float*>(out + npairs * PAIR_BYTES)[i - words] = src < 0 ? 0.0f : params[src];
        } else if (i < total) {
            const int64_t j = i - first;           // (the table's second part starts n_aux entries later: n_aux is even)
            pair(words + n_aux / 2 + j, reinterpret_cast<uint32_t*>(out1 + (j >> 8) * PAIR_BYTES) + (j & 255));
        }
        br.add(mag, i < words ? x2_block_section((int)(i >> 8)) : 0);   // a wave's 64 words lie in one block, hence in one section
    }
    br.end(range);
}

__global__ void __launch_bounds__(256) k_gather_split2(const int32_t* __restrict__ table, const float* __restrict__ params,
                                                       char* __restrict__ out, int64_t npairs, int64_t n_aux, int64_t npairs1) {
    gather_split2_elements(table, params, out, npairs, n_aux, npairs1);
}

// nerf_gather_split2 and nerf_gather_split3 of one network in one launch: every stream a training step on the default
// kernels reads (both fp16-pair streams, both bf16-triple streams) from one pass of the grid over both maps
__global__ void __launch_bounds__(256) k_gather_split23(const int32_t* __restrict__ table2, const int32_t* __restrict__ table3,
                                                        const float* __restrict__ params, char* __restrict__ out2,
                                                        char* __restrict__ out3, int64_t npairs, int64_t n_aux2, int64_t npairs1,
                                                        int64_t triples0, int64_t n_aux3, int64_t triples1) {
    gather_split2_elements(table2, params, out2, npairs, n_aux2, npairs1);
    nerfx3::gather_split3_elements(table3, params, out3, triples0, n_aux3, triples1);
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
    hipLaunchKernelGGL(k_pack_f16x2, dim3(256), dim3(256), 0, as_stream(stream), use_viewdirs, output_ch, params,
                       reinterpret_cast<char*>(packed));
    return launched("nerf_pack_weights_f16x2");
}

int nerf_gather_split2(const int32_t* table, const float* params, void* out, int64_t pairs, int64_t n_aux, int64_t pairs1,
                       void* stream) {
    NERF_REQUIRE(pairs >= 0 && n_aux >= 0 && pairs1 >= 0, "nerf_gather_split2: negative count");
    if (pairs + n_aux + pairs1 == 0) return 0;
    NERF_REQUIRE(table && params && out, "nerf_gather_split2: null pointer");
    NERF_REQUIRE(pairs % SLABP == 0 && pairs1 % SLABP == 0 && n_aux % 4 == 0,
                 "nerf_gather_split2: pairs=%lld and pairs1=%lld must be whole slabs and n_aux=%lld a multiple of 4",
                 (long long)pairs, (long long)pairs1, (long long)n_aux);
    NERF_REQUIRE((reinterpret_cast<uintptr_t>(table) & 7) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                 "nerf_gather_split2: table must be 8-byte and out 16-byte aligned");
    char* range = reinterpret_cast<char*>(out) + pairs * PAIR_BYTES + n_aux * 4;
    hipError_t e = hipMemsetAsync(range, 0, RANGE_WORDS * 4, as_stream(stream));
    if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_gather_split2: clearing the range words: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_gather_split2, dim3(256), dim3(256), 0, as_stream(stream), table, params, reinterpret_cast<char*>(out),
                       pairs, n_aux, pairs1);
    return launched("nerf_gather_split2");
}

int nerf_gather_split23(const int32_t* table2, const int32_t* table3, const float* params, void* out2, void* out3, int64_t pairs,
                        int64_t n_aux2, int64_t pairs1, int64_t triples0, int64_t n_aux3, int64_t triples1, void* stream) {
    NERF_REQUIRE(pairs > 0 && n_aux2 >= 0 && pairs1 >= 0 && triples0 > 0 && n_aux3 >= 0 && triples1 >= 0, "nerf_gather_split23: counts");
    NERF_REQUIRE(table2 && table3 && params && out2 && out3, "nerf_gather_split23: null pointer");
    NERF_REQUIRE(pairs % SLABP == 0 && pairs1 % SLABP == 0 && n_aux2 % 4 == 0 && n_aux3 % 4 == 0,
                 "nerf_gather_split23: pairs=%lld and pairs1=%lld must be whole slabs, n_aux2=%lld and n_aux3=%lld multiples of 4",
                 (long long)pairs, (long long)pairs1, (long long)n_aux2, (long long)n_aux3);
    NERF_REQUIRE(((reinterpret_cast<uintptr_t>(table2) | reinterpret_cast<uintptr_t>(table3)) & 7) == 0 &&
                     ((reinterpret_cast<uintptr_t>(out2) | reinterpret_cast<uintptr_t>(out3)) & 15) == 0,
                 "nerf_gather_split23: tables must be 8-byte and outputs 16-byte aligned");
    char* range = reinterpret_cast<char*>(out2) + pairs * PAIR_BYTES + n_aux2 * 4;
    hipError_t e = hipMemsetAsync(range, 0, RANGE_WORDS * 4, as_stream(stream));
    if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_gather_split23: clearing the range words: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_gather_split23, dim3(1024), dim3(256), 0, as_stream(stream), table2, table3, params,
                       reinterpret_cast<char*>(out2), reinterpret_cast<char*>(out3), pairs, n_aux2, pairs1, triples0, n_aux3, triples1);
    return launched("nerf_gather_split23");
}

int nerf_mlp_fwd_f16x2(int64_t n_pts, int pts_per_ray, const float* pts, const float* viewdirs, int use_viewdirs,
                       int output_ch, const void* packed, const void* packed_bf16x3, float* out, float* act,
                       uint32_t* tile_flags, void* stream) {
    NERF_REQUIRE(n_pts >= 0, "nerf_mlp_fwd_f16x2: n_pts=%lld", (long long)n_pts);
    if (n_pts == 0) return 0;
    NERF_REQUIRE(pts && packed && out && tile_flags, "nerf_mlp_fwd_f16x2: null pointer");
    NERF_REQUIRE(!use_viewdirs || (viewdirs && pts_per_ray >= 1), "nerf_mlp_fwd_f16x2: view branch needs viewdirs and pts_per_ray >= 1");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_mlp_fwd_f16x2: output_ch=%d not in 1..8", output_ch);
    NERF_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                 "nerf_mlp_fwd_f16x2: packed and out must be 16-byte aligned");
    NERF_REQUIRE(!act || (reinterpret_cast<uintptr_t>(act) & 15) == 0, "nerf_mlp_fwd_f16x2: act must be 16-byte aligned");
    const int64_t blocks = (n_pts + X2_WG_PTS - 1) / X2_WG_PTS;
    NERF_REQUIRE(blocks <= 0x7fffffff, "nerf_mlp_fwd_f16x2: n_pts=%lld too large for one launch", (long long)n_pts);
    auto go = [&](auto kernel, int ppr, const float* vd, int och) -> int {
        const size_t lds = x2_lds_bytes(use_viewdirs);
        hipError_t e = lds_limit_once(reinterpret_cast<const void*>(kernel), lds);
        if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_fwd_f16x2: LDS attribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(X2_THREADS), lds, as_stream(stream), n_pts, ppr, pts,
                           vd, och, reinterpret_cast<const char*>(packed), out, act, tile_flags);
        return launched("nerf_mlp_fwd_f16x2");
    };
    int rc;
    if (use_viewdirs) rc = act ? go(k_mlp_fwd_f16x2<true, true>, pts_per_ray, viewdirs, 4) : go(k_mlp_fwd_f16x2<true, false>, pts_per_ray, viewdirs, 4);
    else rc = act ? go(k_mlp_fwd_f16x2<false, true>, 1, nullptr, output_ch) : go(k_mlp_fwd_f16x2<false, false>, 1, nullptr, output_ch);
    if (rc != 0 || packed_bf16x3 == nullptr) return rc;
    // the marked tiles again, on the three-piece kernel
    return nerf_mlp_fwd_bf16x3_marked(n_pts, pts_per_ray, pts, viewdirs, use_viewdirs, output_ch, packed_bf16x3, out, act,
                                      tile_flags, stream);
}

}  // extern "C"
