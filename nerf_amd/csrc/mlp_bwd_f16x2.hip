// Backward of the fused MLP w.r.t. the pre-activations of every layer (stage 1 of the parameter-gradient pass, see
// mlp_bwd.hip) with fp32 accuracy on the fp16 matrix instruction, two pieces per operand: the mirror of
// mlp_fwd_f16x2.hip, and the two-piece form of mlp_bwd_bf16x3.hip (same dataflow, same workspace, same masks).
//
// Gradients lie tens of octaves below fp16's normal range, so every point carries a power-of-two scale: the stage is
// linear in each point's upstream gradient, and a multiplication by 2^k is exact.  A point starts at the scale that
// puts the largest component of its dL/d(raw) at 2^8, and after every layer the scale is moved so that the largest
// value the point has just produced sits at 2^8 again (one multiplication per value in the epilogue, where the value
// is also multiplied by 2^-k of its own scale for the fp32 store the weight-gradient stage reads).  The range rule of
// mlp_x2.hpp then applies per point and layer as in the forward: a point whose largest value of a layer leaves
// [2^-11, 2^15) -- a layer that grows a gradient by more than 2^7 or shrinks its largest entry by more than 2^19, an
// infinite or NaN gradient, weights out of range -- marks its 128-point tile in `tile_flags`, and
// nerf_mlp_bwd_data_f16x2 has the three-piece kernel redo exactly the marked tiles.
#include "common.hpp"
#include "mlp_core.hpp"
#include "mlp_tile16.hpp"
#include "mlp_bf16_layout.hpp"
#include "mlp_bwd16_layout.hpp"
#include "mlp_x2.hpp"
#include "mlp_dw.hpp"

using namespace nerfhip;
using namespace nerflayout;
using namespace nerfmlp;
using namespace nerftile16;
using namespace nerfbf16;
using namespace nerfbwd16;
using namespace nerfx2;

namespace {

constexpr int SCALE_TARGET = 8;        // a point's largest value is moved to [2^8, 2^9)
constexpr int SCALE_LIMIT = 120;       // |scale exponent|: 2^k and 2^-k stay normal floats
constexpr int BWD_STAGES = nerfdw::DY_STAGES;

// One backward-data layer: out = keep(start + M in), M = the section's W^T; `in` carries the point's scale 2^k.
// keep (MASK): as layer_x3b of mlp_bwd_bf16x3.hip.  ALPHA: start = d_sigma (at the input's scale) times the density
// head's weight column.  A finished value is stored times down[ct] = 2^-k to srow[ct] (+ 4 t row groups) and, with
// STORE, split times rs[ct] (the move to the next scale) into the pieces of fragment t / 2 of OUT; mx[ct] collects the
// largest leading piece as fp16 bits, stored[ct] the largest magnitude stored (unscaled).
template <int NOUT, int NKS, bool MASK, bool ALPHA, bool STORE, typename InFrag>
__device__ __forceinline__ void layer_x2b(Stream2& st, int lane, int g, const char*& cur, InFrag in_frag,
                                          u32x4 (&OUT)[2][NCT2][8], const float* alpha_g, const float (&d_sigma)[NCT2],
                                          const u32x4 (&mw)[NCT2], float* (&srow)[NCT2], const float (&down)[NCT2],
                                          const float (&rs)[NCT2], uint32_t (&mx)[NCT2], float (&stored)[NCT2]) {
    constexpr int NB = NOUT * NKS, DEPTH = 3, NPIECE = 4 * NCT2;
    constexpr int PPS = (NPIECE + (NKS < 8 ? NKS : 8) - 1) / (NKS < 8 ? NKS : 8);   // epilogue values per k-step
    u32x4 abuf[DEPTH + 1][2];
    f32x4 amain[2][NCT2], acorr[2][NCT2];         // [alternating set][column tile]
    const f32x4 zero = {0, 0, 0, 0};
    f32x4 wnext = {0, 0, 0, 0};                   // ALPHA: the density head's weights of the next tile's rows
    f32x4 sv = {0, 0, 0, 0};
    float pv = 0.0f;                              // STORE: the even value of the pair being collected
    const int shift = 4 * (g >> 1);
    auto load_start = [&]<int TO>() { if constexpr (ALPHA) wnext = *reinterpret_cast<const f32x4*>(alpha_g + 16 * TO); };
    auto piece = [&]<int TP, int K>() {           // K = 4 * column tile + register: one value of the tile per piece
        constexpr int ct = K >> 2, r = K & 3;
        float v = fmaf(acorr[TP & 1][ct][r], LO_UNSCALE, amain[TP & 1][ct][r]);
        if constexpr (MASK) {
            const bool on = ((mw[ct][TP >> 2] >> (8 * (TP & 3) + shift + r)) & 1u) != 0u;
            v = on ? v : 0.0f;
        }
        sv[r] = v * down[ct];
        stored[ct] = __builtin_fmaxf(stored[ct], __builtin_fabsf(sv[r]));
        if constexpr (r == 3) *reinterpret_cast<f32x4*>(srow[ct] + 4 * TP * GROUP_FLOATS) = sv;
        if constexpr (STORE) {
            const float vs = v * rs[ct];
            if constexpr ((r & 1) == 0) {
                pv = vs;
            } else {
                constexpr int f = TP >> 1, w = 2 * (TP & 1) + (r >> 1);
                uint32_t h, l;
                split2_pair(pv, vs, h, l);
                OUT[0][ct][f][w] = h;
                OUT[1][ct][f][w] = l;
                mx[ct] = run_max_h(mx[ct], h & 0x7fff7fffu);
            }
        }
    };

    load_start.template operator()<0>();
    static_for<NB + DEPTH>([&]<int I>() {
        if constexpr (I < NB) {
            if constexpr (I % SLABP == 0) cur = st.template advance<(I > 0) ? x2_stores_since_open(I, NKS) : 0>(lane);
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
                if constexpr (ks == 0) {
                    f32x4 c0 = {0, 0, 0, 0};
                    if constexpr (ALPHA) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) c0[i] = wnext[i] * d_sigma[CT];
                    }
                    amain[to & 1][CT] = mfma16h(ah, xh, c0);
                } else {
                    amain[to & 1][CT] = mfma16h(ah, xh, amain[to & 1][CT]);
                }
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
            if constexpr (to + 1 < NOUT && ks == (NKS >= 4 ? NKS - 3 : 0)) load_start.template operator()<to + 1>();
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    static_for<NPIECE>([&]<int K>() { piece.template operator()<NOUT - 1, K>(); });
}

template <bool VIEWDIRS>
__global__ void __launch_bounds__(X2_THREADS, 1)
k_mlp_bwd_data_f16x2(int64_t n_pts, int output_ch, const float* __restrict__ d_out, const float* __restrict__ act,
                     const char* __restrict__ packed_fwd, const char* __restrict__ packed_bwd, float* __restrict__ scr,
                     uint32_t* __restrict__ tile_flags, uint32_t* __restrict__ tile_max) {
    extern __shared__ __attribute__((aligned(16))) char smem2b[];
    char* ring = smem2b;
    float* aux = reinterpret_cast<float*>(smem2b + RING2_BYTES);
    uint32_t* wave_flags = reinterpret_cast<uint32_t*>(smem2b + RING2_BYTES + AUX16_FLOATS * 4);
    uint32_t* stage_max = wave_flags + X2_WAVES;      // per wave and stage: the largest stored magnitude, as exponent bits + 1
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4;
    constexpr int UV = VIEWDIRS ? 1 : 0;
    const char* tail = packed_fwd + (int64_t)x2_stream_slabs(UV) * SLAB2_BYTES;
    {   // a weight section this stage multiplies by (every one but layer 0) outside the range the two-piece split is
        // exact for: every tile goes to the three-piece kernel
        const uint32_t* rw = reinterpret_cast<const uint32_t*>(tail + (int64_t)AUX16_FLOATS * 4);
        bool wbad = false;
#pragma unroll
        for (int s = 1; s < x2_sections(UV); ++s) wbad |= out_of_range(__uint_as_float(rw[s]));
        if (wbad) {
            if (tid == 0) tile_flags[blockIdx.x] = 1u;
            return;
        }
    }
    {   // the fp32 tables behind the forward stream: head rows in natural feature order
        const float* gaux = reinterpret_cast<const float*>(tail);
        for (int i = tid * 4; i < AUX16_FLOATS; i += X2_THREADS * 4)
            *reinterpret_cast<f32x4*>(aux + i) = *reinterpret_cast<const f32x4*>(gaux + i);
    }
    static_assert(X2_WG_PTS == TILE_PTS, "one workgroup per tile of the saved activations");
    const float* const at = act + (int64_t)blockIdx.x * act_rows(UV) * TILE_PTS;
    float* const st_ = scr + (int64_t)blockIdx.x * scr_rows(UV) * TILE_PTS;
    int pw[NCT2];
    bool valid[NCT2];
    int64_t p[NCT2];
    float* lane_rows[NCT2];
#pragma unroll
    for (int ct = 0; ct < NCT2; ++ct) {
        pw[ct] = wave * WAVE_PTS2 + ct * 16 + (lane & 15);      // point inside the 128-point tile
        p[ct] = (int64_t)blockIdx.x * TILE_PTS + pw[ct];
        valid[ct] = p[ct] < n_pts;
        lane_rows[ct] = st_ + g * GROUP_FLOATS + pw[ct] * 4;
    }
    constexpr int SCR_HEAD = scr_head(UV), SCR_HEAD2 = SCR_HEAD + 32;

    Stream2 st;
    st.ring = ring;
    st.mine = ring + wave * WAVE_BLOCKS2 * 1024;
    st.g = packed_bwd + wave * WAVE_BLOCKS2 * 1024 + lane * 16;
    st.next = 0;
    st.nslabs = bwd16_stream_blocks(UV) / SLABP;
    for (int i = tid; i < X2_WAVES * BWD_STAGES; i += X2_THREADS) stage_max[i] = 0u;
    __syncthreads();
    st.fetch(0);
    st.fetch(1);
    const char* cur = nullptr;

    // the mask word of masked layer m that holds this lane's bits (written by lane (point, g & 1) of the forward)
    u32x4 mw[NCT2];
    auto load_masks = [&](int m) {
#pragma unroll
        for (int ct = 0; ct < NCT2; ++ct)
            mw[ct] = *reinterpret_cast<const u32x4*>(at + act_mask_offset(UV, m, pw[ct] >> 5, (g & 1) * 32 + (pw[ct] & 31)));
    };
    const int shift = 4 * (g >> 1);
    auto on = [&](int ct, int t, int i) { return ((mw[ct][t >> 2] >> (8 * (t & 3) + shift + i)) & 1u) != 0u; };
    float* sr[NCT2];
    auto rows = [&](int row) {
#pragma unroll
        for (int ct = 0; ct < NCT2; ++ct) sr[ct] = lane_rows[ct] + (row / 4) * GROUP_FLOATS;
    };

    // ---- scales: kexp = exponent of the newest values' scale; down / rs / up_in = what the next layer multiplies by
    int kexp[NCT2];
    float down[NCT2], rs[NCT2], up_in[NCT2];
    uint32_t mx[NCT2];
    bool bad = false;
    // starts a point at the scale that puts `m`, its largest |dL/d(raw)|, at 2^SCALE_TARGET (zero, subnormal, infinite
    // or NaN: scale 1 -- zeros are exact; a subnormal marks the tile here, the others when their pieces are seen)
    auto start_scale = [&](int ct, float m) {
        const int e = (int)((__float_as_uint(m) >> 23) & 255u);
        bad |= e == 0 && m != 0.0f;
        int k = (e == 0 || e == 255) ? 0 : SCALE_TARGET - (e - 127);
        k = k > SCALE_LIMIT ? SCALE_LIMIT : (k < -SCALE_LIMIT ? -SCALE_LIMIT : k);
        kexp[ct] = k;
        down[ct] = pow2i(-k);
        rs[ct] = 1.0f;
        up_in[ct] = pow2i(k);
        mx[ct] = 0u;
    };
    // after a layer: the range rule on what it produced, and the move of the point's scale for the layer after it.
    // also[ct]: the magnitude (unscaled) of a term the next layer adds to its sums besides its matrix product -- the
    // density head's -- so that the scale is set by whichever of the two is larger
    auto check = [&](const float (&also)[NCT2]) {
#pragma unroll
        for (int ct = 0; ct < NCT2; ++ct) {
            uint32_t m = (mx[ct] & 0xffffu) > (mx[ct] >> 16) ? (mx[ct] & 0xffffu) : (mx[ct] >> 16);
            const uint32_t m1 = (uint32_t)__shfl_xor((int)m, 16, 64);
            m = m > m1 ? m : m1;
            const uint32_t m2 = (uint32_t)__shfl_xor((int)m, 32, 64);
            m = m > m2 ? m : m2;
            bad |= out_of_range_h(m);
            const int e16 = (int)(m >> 10);                       // biased by 15; 0 = zero or subnormal, 31 = infinite / NaN
            const int ea = (int)((__float_as_uint(also[ct]) >> 23) & 255u);
            int e = e16;
            bool have = e16 != 0;
            if (ea != 0 && ea != 255) {                           // the same exponent for the other term, at the current scale
                const int et = ea - 127 + kexp[ct] + 15;
                if (!have || et > e) e = et;
                have = true;
            }
            int d = (have && e16 != 31) ? SCALE_TARGET + 15 - e : 0;
            d = d > 100 ? 100 : (d < -100 ? -100 : d);            // (one move stays a normal float)
            int kn = kexp[ct] + d;
            kn = kn > SCALE_LIMIT ? SCALE_LIMIT : (kn < -SCALE_LIMIT ? -SCALE_LIMIT : kn);
            down[ct] = pow2i(-kexp[ct]);
            up_in[ct] = pow2i(kexp[ct]);
            rs[ct] = pow2i(kn - kexp[ct]);
            kexp[ct] = kn;
            mx[ct] = 0u;
        }
    };

    // the largest magnitude this stage stored for the weight-gradient stage, per tile: an upper bound 2^(e - 127) with
    // e = exponent bits + 1 (0: nothing but zeros), collected over the workgroup in LDS
    float stored[NCT2] = {0.0f, 0.0f};
    // (the wave's largest word by bisection over ballots: scalar work, no LDS traffic; 64 lanes' atomics on one LDS word
    // took 1.6 us apiece)
    auto wave_max = [&](uint32_t e) -> uint32_t {            // e < 512
        uint32_t best = 0u;
#pragma unroll
        for (int bit = 8; bit >= 0; --bit) {
            const uint32_t cand = best | (1u << bit);
            if (__ballot(e >= cand) != 0ull) best = cand;
        }
        return best;
    };
    auto put_stage = [&](int stage, uint32_t e) {
        const uint32_t m = wave_max(e);
        if (lane == 0) stage_max[wave * BWD_STAGES + stage] = m;
    };
    auto word = [](float v) -> uint32_t {                    // exponent bits + 1, 0 for a zero
        const uint32_t b = __float_as_uint(v) & 0x7fffffffu;
        return b != 0u ? (b >> 23) + 1u : 0u;
    };
    auto record = [&](int stage) {
        put_stage(stage, word(__builtin_fmaxf(stored[0], stored[1])));
        stored[0] = stored[1] = 0.0f;
    };
    u32x4 X[2][NCT2][8], Y[2][NCT2][8];
    auto fromX = [&]<int ct, int ks, int s>() -> u32x4 { return X[s][ct][ks]; };
    auto fromY = [&]<int ct, int ks, int s>() -> u32x4 { return Y[s][ct][ks]; };
    const float* hw = aux + AUX16_HEAD;
    const float none[NCT2] = {0.0f, 0.0f};
    float d_sigma[NCT2] = {0.0f, 0.0f};     // unscaled
    float dsig_in[NCT2] = {0.0f, 0.0f};     // at the scale of the layer that adds the density head's term
    // a head gradient tile (VALU, at the point's first scale): masked, stored unscaled, split into fragment t / 2 of X
    auto put = [&]<int ct, int t>(const f32x4& v) {
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[i] = v[i] * down[ct];
            stored[ct] = __builtin_fmaxf(stored[ct], __builtin_fabsf(o[i]));
        }
        *reinterpret_cast<f32x4*>(sr[ct] + 4 * t * GROUP_FLOATS) = o;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            uint32_t h, l;
            split2_pair(v[2 * w], v[2 * w + 1], h, l);
            X[0][ct][t >> 1][2 * (t & 1) + w] = h;
            X[1][ct][t >> 1][2 * (t & 1) + w] = l;
            mx[ct] = run_max_h(mx[ct], h & 0x7fff7fffu);
        }
    };

    if constexpr (VIEWDIRS) {
        f32x4 gd[NCT2];
#pragma unroll
        for (int ct = 0; ct < NCT2; ++ct) {
            gd[ct] = f32x4{0, 0, 0, 0};
            if (valid[ct]) gd[ct] = *reinterpret_cast<const f32x4*>(d_out + p[ct] * 4);
            d_sigma[ct] = gd[ct][3];
            if (g == 0) {   // head operands of the weight-gradient stage: one row group each, zero padded
                const f32x4 ga = {gd[ct][0], gd[ct][1], gd[ct][2], 0.0f}, gb = {gd[ct][3], 0.0f, 0.0f, 0.0f};
                *reinterpret_cast<f32x4*>(st_ + (SCR_HEAD / 4) * GROUP_FLOATS + pw[ct] * 4) = ga;
                *reinterpret_cast<f32x4*>(st_ + (SCR_HEAD2 / 4) * GROUP_FLOATS + pw[ct] * 4) = gb;
            }
            stored[ct] = run_max(run_max(0.0f, gd[ct][0], gd[ct][1]), gd[ct][2], gd[ct][3]);
            // the first two layers see the colour terms only; the density term joins two layers down, where it takes part
            // in that layer's scale (check): a point starts at the scale of its colour terms if it has any
            const float m_rgb = run_max(run_max(0.0f, gd[ct][0], gd[ct][1]), gd[ct][2], 0.0f);
            start_scale(ct, m_rgb != 0.0f ? m_rgb : stored[ct]);
            stored[ct] = 0.0f;
        }
        // one word per channel of dL/d(raw): colour 0..2, density 4 (every lane group holds the same values)
        static_for<4>([&]<int c>() {
            put_stage(c == 3 ? 4 : c, word(run_max(0.0f, gd[0][c], gd[1][c])));
        });
#pragma unroll
        for (int ct = 0; ct < NCT2; ++ct)
#pragma unroll
            for (int i = 0; i < 3; ++i) gd[ct][i] *= up_in[ct];
        // views layer: d(hidden) = Wrgb^T d_rgb, masked -> X fragments 0..3, stored
        load_masks(8);
        rows(SCR_VIEWS);
        const float* wrgb = hw + 256 + 4 * g;
        static_for<8>([&]<int t>() {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wrgb + 16 * t);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wrgb + 128 + 16 * t);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(wrgb + 256 + 16 * t);
            static_for<NCT2>([&]<int ct>() {
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float y = fmaf(w2[i], gd[ct][2], fmaf(w1[i], gd[ct][1], w0[i] * gd[ct][0]));
                    v[i] = on(ct, t, i) ? y : 0.0f;
                }
                put.template operator()<ct, t>(v);
            });
        });
        check(none);
        record(nerfdw::DY_STAGE_VIEWS);
        // d(feature) = Wviews[:, :256]^T dY_views -> Y (feature_linear has no activation)
        rows(SCR_FEAT);
        layer_x2b<16, 4, false, false, true>(st, lane, g, cur, fromX, Y, nullptr, dsig_in, mw, sr, down, rs, mx, stored);
        record(nerfdw::DY_STAGE_FEAT);
        {   // the density head's term of the next layer: its largest entry per point
            float wa = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) wa = run_max(wa, hw[4 * lane + i], 0.0f);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) wa = __builtin_fmaxf(wa, __shfl_xor(wa, off, 64));
            float term[NCT2];
#pragma unroll
            for (int ct = 0; ct < NCT2; ++ct) term[ct] = wa * __builtin_fabsf(d_sigma[ct]);
            check(term);
        }
        // d(h7) = Wfeature^T d(feature) + Walpha^T d_sigma, masked -> X
#pragma unroll
        for (int ct = 0; ct < NCT2; ++ct) dsig_in[ct] = d_sigma[ct] * up_in[ct];
        load_masks(7);
        rows(SCR_H0 + 7 * 256);
        layer_x2b<16, 8, true, true, true>(st, lane, g, cur, fromY, X, hw + 4 * g, dsig_in, mw, sr, down, rs, mx, stored);
        check(none);
        record(nerfdw::dy_stage_layer(7));
    } else {
        // d(h7) = Wout^T d_out, masked -> X, stored
        float gd[NCT2][8];
#pragma unroll
        for (int ct = 0; ct < NCT2; ++ct) {
            float m = 0.0f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                gd[ct][c] = (valid[ct] && c < output_ch) ? d_out[p[ct] * output_ch + c] : 0.0f;
                m = run_max(m, gd[ct][c], 0.0f);
            }
            if (g == 0) {
                const f32x4 ga = {gd[ct][0], gd[ct][1], gd[ct][2], gd[ct][3]}, gb = {gd[ct][4], gd[ct][5], gd[ct][6], gd[ct][7]};
                *reinterpret_cast<f32x4*>(st_ + (SCR_HEAD / 4) * GROUP_FLOATS + pw[ct] * 4) = ga;
                *reinterpret_cast<f32x4*>(st_ + (SCR_HEAD / 4 + 1) * GROUP_FLOATS + pw[ct] * 4) = gb;
            }
            start_scale(ct, m);
        }
        static_for<8>([&]<int c>() { put_stage(c, word(run_max(0.0f, gd[0][c], gd[1][c]))); });
#pragma unroll
        for (int ct = 0; ct < NCT2; ++ct)
#pragma unroll
            for (int c = 0; c < 8; ++c) gd[ct][c] *= up_in[ct];
        load_masks(7);
        rows(SCR_H0 + 7 * 256);
        static_for<16>([&]<int t>() {
            static_for<NCT2>([&]<int ct>() {
                f32x4 v = {0, 0, 0, 0};
#pragma unroll
                for (int c = 0; c < 8; ++c) {   // (gradients of the channels past output_ch are zero)
                    const f32x4 w = *reinterpret_cast<const f32x4*>(hw + c * 256 + 16 * t + 4 * g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaf(w[i], gd[ct][c], v[i]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = on(ct, t, i) ? v[i] : 0.0f;
                put.template operator()<ct, t>(v);
            });
        });
        check(none);
        record(nerfdw::dy_stage_layer(7));
    }

    // dY of pts layers 6 .. 0: X (layer l + 1) -> Y (layer l) -> X (layer l - 1) ...
    load_masks(6); rows(SCR_H0 + 6 * 256);
    layer_x2b<16, 8, true, false, true>(st, lane, g, cur, fromX, Y, nullptr, dsig_in, mw, sr, down, rs, mx, stored);
    check(none); record(nerfdw::dy_stage_layer(6));
    load_masks(5); rows(SCR_H0 + 5 * 256);
    layer_x2b<16, 8, true, false, true>(st, lane, g, cur, fromY, X, nullptr, dsig_in, mw, sr, down, rs, mx, stored);
    check(none); record(nerfdw::dy_stage_layer(5));
    load_masks(4); rows(SCR_H0 + 4 * 256);
    layer_x2b<16, 8, true, false, true>(st, lane, g, cur, fromX, Y, nullptr, dsig_in, mw, sr, down, rs, mx, stored);
    check(none); record(nerfdw::dy_stage_layer(4));
    load_masks(3); rows(SCR_H0 + 3 * 256);
    layer_x2b<16, 8, true, false, true>(st, lane, g, cur, fromY, X, nullptr, dsig_in, mw, sr, down, rs, mx, stored);
    check(none); record(nerfdw::dy_stage_layer(3));
    load_masks(2); rows(SCR_H0 + 2 * 256);
    layer_x2b<16, 8, true, false, true>(st, lane, g, cur, fromX, Y, nullptr, dsig_in, mw, sr, down, rs, mx, stored);
    check(none); record(nerfdw::dy_stage_layer(2));
    load_masks(1); rows(SCR_H0 + 1 * 256);
    layer_x2b<16, 8, true, false, true>(st, lane, g, cur, fromY, X, nullptr, dsig_in, mw, sr, down, rs, mx, stored);
    check(none); record(nerfdw::dy_stage_layer(1));
    load_masks(0); rows(SCR_H0);
    layer_x2b<16, 8, true, false, false>(st, lane, g, cur, fromX, Y, nullptr, dsig_in, mw, sr, down, rs, mx, stored);
    record(nerfdw::dy_stage_layer(0));

    // one flag per 128-point tile: any lane of any wave that left the range
    const bool wave_bad = __any(bad);
    if (lane == 0) wave_flags[wave] = wave_bad ? 1u : 0u;
    __syncthreads();
    if (tid == 0) tile_flags[blockIdx.x] = wave_flags[0] | wave_flags[1] | wave_flags[2] | wave_flags[3];
    if (tile_max != nullptr && tid < BWD_STAGES) {
        uint32_t m = stage_max[tid];
#pragma unroll
        for (int w = 1; w < X2_WAVES; ++w) m = stage_max[w * BWD_STAGES + tid] > m ? stage_max[w * BWD_STAGES + tid] : m;
        tile_max[(int64_t)blockIdx.x * BWD_STAGES + tid] = m;
    }
}

// flat parameters -> transposed stream of pairs
__global__ void __launch_bounds__(256) k_pack_x2b(int use_viewdirs, int output_ch, const float* __restrict__ params,
                                                  char* __restrict__ out) {
    const int64_t elems = (int64_t)bwd16_stream_blocks(use_viewdirs) * 512;   // one per (pair, lane, slot)
    unsigned short* s16 = reinterpret_cast<unsigned short*>(out);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < elems; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i >> 9;
        const int w = (int)(i & 511);
        _Float16 h, l;
        split2(params[bwd16_stream_source(use_viewdirs, output_ch, (int)t, w >> 3, w & 7)], h, l);
        unsigned short* dst = s16 + t * (PAIR_BYTES / 2) + w;
        dst[0] = __builtin_bit_cast(unsigned short, h);
        dst[512] = __builtin_bit_cast(unsigned short, l);
    }
}

constexpr size_t X2B_LDS_BYTES = RING2_BYTES + AUX16_FLOATS * sizeof(float) + (X2_WAVES + X2_WAVES * BWD_STAGES) * sizeof(uint32_t);

}  // namespace

namespace nerfx3 {
int launch_bwd_data_x3(int64_t n_pts, int use_viewdirs, int output_ch, const float* d_out, const float* act,
                       const void* packed_x3, const void* packed_bwd_x3, float* scratch, const uint32_t* marked,
                       hipStream_t stream);   // mlp_bwd_bf16x3.hip
}

extern "C" {

int64_t nerf_mlp_packed_bwd_f16x2_bytes(int use_viewdirs) { return (int64_t)bwd16_stream_blocks(use_viewdirs) * PAIR_BYTES; }

int nerf_pack_weights_bwd_f16x2(int use_viewdirs, int output_ch, const float* params, void* packed, void* stream) {
    NERF_REQUIRE(params && packed, "nerf_pack_weights_bwd_f16x2: null pointer");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_pack_weights_bwd_f16x2: output_ch=%d not in 1..8", output_ch);
    hipLaunchKernelGGL(k_pack_x2b, dim3(1024), dim3(256), 0, as_stream(stream), use_viewdirs, output_ch, params,
                       reinterpret_cast<char*>(packed));
    return launched("nerf_pack_weights_bwd_f16x2");
}

int nerf_mlp_bwd_data_f16x2(int64_t n_pts, const float* d_out, const float* act, int use_viewdirs, int output_ch,
                            const void* packed_f16x2, const void* packed_bwd_f16x2, const void* packed_x3,
                            const void* packed_bwd_x3, float* scratch, uint32_t* tile_flags, uint32_t* tile_max, void* stream_) {
    NERF_REQUIRE(n_pts >= 0, "nerf_mlp_bwd_data_f16x2: n_pts=%lld", (long long)n_pts);
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_mlp_bwd_data_f16x2: output_ch=%d not in 1..8", output_ch);
    if (n_pts == 0) return 0;
    NERF_REQUIRE(d_out && act && packed_f16x2 && packed_bwd_f16x2 && scratch && tile_flags, "nerf_mlp_bwd_data_f16x2: null pointer");
    NERF_REQUIRE((packed_x3 == nullptr) == (packed_bwd_x3 == nullptr),
                 "nerf_mlp_bwd_data_f16x2: pass both three-piece streams (marked tiles are redone) or neither");
    NERF_REQUIRE(((reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(act) | reinterpret_cast<uintptr_t>(scratch) |
                   reinterpret_cast<uintptr_t>(packed_f16x2) | reinterpret_cast<uintptr_t>(packed_bwd_f16x2) |
                   reinterpret_cast<uintptr_t>(packed_x3) | reinterpret_cast<uintptr_t>(packed_bwd_x3)) & 15) == 0,
                 "nerf_mlp_bwd_data_f16x2: buffers must be 16-byte aligned");
    const int64_t blocks = padded_points(n_pts) / TILE_PTS;
    NERF_REQUIRE(blocks <= 0x7fffffff, "nerf_mlp_bwd_data_f16x2: n_pts=%lld too large for one call", (long long)n_pts);
    hipStream_t stream = as_stream(stream_);
    auto go = [&](auto kernel, int och) -> int {
        hipError_t e = lds_limit_once(reinterpret_cast<const void*>(kernel), X2B_LDS_BYTES);
        if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_bwd_data_f16x2: LDS attribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(X2_THREADS), X2B_LDS_BYTES, stream, n_pts, och, d_out, act,
                           reinterpret_cast<const char*>(packed_f16x2), reinterpret_cast<const char*>(packed_bwd_f16x2), scratch,
                           tile_flags, tile_max);
        return launched("nerf_mlp_bwd_data_f16x2");
    };
    int rc = use_viewdirs ? go(k_mlp_bwd_data_f16x2<true>, 4) : go(k_mlp_bwd_data_f16x2<false>, output_ch);
    if (rc != 0 || packed_x3 == nullptr) return rc;
    // the marked tiles again, on the three-piece kernel
    return nerfx3::launch_bwd_data_x3(n_pts, use_viewdirs, output_ch, d_out, act, packed_x3, packed_bwd_x3, scratch, tile_flags, stream);
}

}  // extern "C"
