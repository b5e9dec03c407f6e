// Fused positional encoding + NeRF MLP forward with fp32 accuracy on the bf16 matrix instruction.
//
// Every fp32 operand is split EXACTLY into three bf16 pieces, x = hi + mid + lo (hi = bf16(x), mid = bf16(x - hi),
// lo = bf16(x - hi - mid): 3 x 8 significant bits = the 24 of an fp32), and a product a.b is rebuilt from the six
// largest of the nine piece products -- hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid -- each exact in the fp32
// accumulator's input; the three dropped ones are below 2^-24 of the product, the size of one fp32 rounding.  A 256-term
// dot product built this way measures the same error against fp64 as v_mfma_f32_16x16x4_f32 does (rms 1.9e-8 against
// 2.4e-8 of sum |a||b|, tools/probe_bf16x3.hip), and six v_mfma_f32_16x16x32_bf16 cost 6/16 of the fp32 instructions
// they replace.
//
// Dataflow as in mlp_fwd_bf16.hip (activations never leave the register file; weights stream through an LDS slab
// ring; bias = accumulator start value; a tile's epilogue -- ReLU, heads, the three-way split of its values -- rides
// behind the next tile's MFMAs), with
//   * three A fragments per weight block (a "triple": hi | mid | lo, 3 KiB), split once per weight update by the packer;
//   * three B fragments per activation k-step and column tile: 2 x 192 registers for the two activation arrays, so ONE
//     wave per SIMD (up to 512 registers), four waves and 128 points per workgroup;
//   * twelve MFMAs per triple (six products x two column tiles) that alternate between the column tiles' accumulators.
#include "common.hpp"
#include "mlp_core.hpp"
#include "mlp_tile16.hpp"
#include "mlp_bf16_layout.hpp"
#include "mlp_x3.hpp"

using namespace nerfhip;
using namespace nerflayout;
using namespace nerfmlp;
using namespace nerftile16;
using namespace nerfbf16;
using namespace nerfx3;

namespace {

__host__ __device__ inline int x3_stream_slabs(int use_viewdirs) { return bf16_stream_blocks(use_viewdirs) / SLABT; }
__host__ __device__ inline int64_t x3_packed_bytes(int use_viewdirs) {
    return (int64_t)x3_stream_slabs(use_viewdirs) * SLAB3_BYTES + (int64_t)AUX16_FLOATS * 4;
}

// LDS behind the ring: the fp32 tables, then the split encodings (fragments per piece and wave)
__host__ __device__ constexpr int aux_floats(int use_viewdirs) { return use_viewdirs ? AUX16_HEAD + 648 : AUX16_FLOATS; }
__host__ __device__ constexpr int enc_frags(int use_viewdirs) { return (use_viewdirs ? 3 : 2) * NCT; }

// One layer: NOUT row tiles (16 features each) x NKS k-steps (32 input features each) x NCT column tiles.
// in_frag<CT, ks, S>() = piece S of the input fragment of column tile CT, k-step ks.  Y[S][CT][..] receives the pieces
// of the output (STORE).  Everything else as layer16 of mlp_fwd_bf16.hip.
// SAVE (training): the unsplit fp32 values of a finished tile are also stored for the backward pass, one 16-byte store
// per tile, column tile and lane into the blocked layout of mlp_layout.hpp (`srow[ct]` = the layer's first row group +
// this lane's offset g * GROUP_FLOATS + 4 * (point in the 128-point tile): a lane's four values of tile t are the
// four rows of row group 4 t + g); with MASK their ReLU mask bits are collected in mb[ct], dword t / 4, bit 8 (t % 4) + r
// (the lane's share of the words the backward-data kernels read; see save_mask in the kernel).
template <int NOUT, int NKS, int NHEAD, bool STORE, bool SAVE, bool MASK, typename InFrag>
__device__ __forceinline__ void layer_x3(Stream3& st, int lane, int g, const char*& cur, InFrag in_frag,
                                         bf16x8 (&Y)[3][NCT][8], const float* bias_lds, float lo, const float* head_lds,
                                         int head_stride, float (&head)[NCT][3], float* (&srow)[NCT],
                                         uint32_t (&mb)[NCT][4]) {
    constexpr int NB = NOUT * NKS, DEPTH = 3, NPIECE = 4 * NCT;
    constexpr bool BIAS_INIT = NKS >= 7;
    constexpr int PPS = (NPIECE + (NKS < 8 ? NKS : 8) - 1) / (NKS < 8 ? NKS : 8);   // epilogue pieces per k-step
    bf16x8 abuf[DEPTH + 1][3];
    f32x4 acc[2][NCT];                            // [alternating set][column tile]
    const float* bp = bias_lds + 4 * g;
    const float* hp = head_lds + 4 * g;

    // the bias column of a tile is the C operand of its first MFMAs; it is read while the tile before is multiplying
    f32x4 bnext = {0, 0, 0, 0};
    f32x4 sv = {0, 0, 0, 0};                      // SAVE: the four values of (tile, column tile) being collected
    if constexpr (MASK) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) mb[ct][0] = mb[ct][1] = mb[ct][2] = mb[ct][3] = 0u;
    }
    auto load_bias = [&]<int TO>() { bnext = *reinterpret_cast<const f32x4*>(bp + 16 * TO); };
    auto piece = [&]<int TP, int K>() {           // K = 4 * column tile + register: one value of the tile per piece
        constexpr int ct = K >> 2, r = K & 3;
        float v = acc[TP & 1][ct][r];
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
            constexpr int f = TP >> 1, slot = 4 * (TP & 1) + r;
            __bf16 h, m, l;
            split3(v, h, m, l);
            Y[0][ct][f][slot] = h; Y[1][ct][f][slot] = m; Y[2][ct][f][slot] = l;
        }
    };

    if constexpr (BIAS_INIT) load_bias.template operator()<0>();
    static_for<NB + DEPTH>([&]<int I>() {
        if constexpr (I < NB) {
            if constexpr (I % SLABT == 0) {
                cur = st.template advance<(SAVE && I > 0) ? x3_stores_since_open(I, NKS) : 0>(lane);
            }
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
                static_for<NCT>([&]<int CT>() {
                    const bf16x8 x = in_frag.template operator()<CT, ks, px>();
                    if constexpr (ks == 0 && P == 0) {   // (bnext stays zero in the layers that add the bias in the epilogue)
                        acc[to & 1][CT] = mfma16(a, x, bnext);
                    } else {
                        acc[to & 1][CT] = mfma16(a, x, acc[to & 1][CT]);
                    }
                });
            });
            if constexpr (to > 0) {   // the previous row tile's epilogue, piece by piece, behind this tile's MFMAs
                static_for<PPS>([&]<int Q>() {
                    constexpr int K = ks * PPS + Q;
                    if constexpr (K < NPIECE) piece.template operator()<to - 1, K>();
                });
            }
            if constexpr (BIAS_INIT && to + 1 < NOUT && ks == NKS - 3) load_bias.template operator()<to + 1>();
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    static_for<NPIECE>([&]<int K>() { piece.template operator()<NOUT - 1, K>(); });
}

// one 128-point tile (tile_index) by one workgroup
template <bool VIEWDIRS, bool SAVE>
__device__ __forceinline__ void fwd_tile_x3(int64_t tile_index, int64_t n_pts, int pts_per_ray, const float* __restrict__ pts,
                                            const float* __restrict__ viewdirs, int output_ch, const char* __restrict__ packed,
                                            float* __restrict__ out, float* __restrict__ act) {
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    char* ring = smem3;
    float* aux = reinterpret_cast<float*>(smem3 + RING3_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4;
    constexpr int UV = VIEWDIRS ? 1 : 0;
    {   // (with the view branch only the first 648 head floats are in use: alpha row, three rgb rows, their biases)
        const float* gaux = reinterpret_cast<const float*>(packed + (int64_t)x3_stream_slabs(UV) * SLAB3_BYTES);
        for (int i = tid * 4; i < aux_floats(UV); i += X3_THREADS * 4)
            *reinterpret_cast<f32x4*>(aux + i) = *reinterpret_cast<const f32x4*>(gaux + i);
    }

    // ---- encodings, split, parked in LDS: fragment (piece s, slot f) of this lane at ed[(s * NF + f) * 64],
    // f = 2 ct + k-step for the position, 2 NCT + ct for the direction
    constexpr int NF = enc_frags(UV);
    bf16x8* ed = reinterpret_cast<bf16x8*>(smem3 + RING3_BYTES + aux_floats(UV) * 4) + wave * (3 * NF * 64) + lane;
    auto park = [&](int f, const float* v) {
        bf16x8 h, m, l;
        split8(v, h, m, l);
        ed[(0 * NF + f) * 64] = h;
        ed[(1 * NF + f) * 64] = m;
        ed[(2 * NF + f) * 64] = l;
    };
    int64_t p[NCT];
    bool valid[NCT];
    const int c0 = enc16_first_pair(g);
    // training: this workgroup's 128-point tile of the saved activations (mlp_layout.hpp)
    static_assert(X3_WG_PTS == TILE_PTS, "one workgroup per tile of the saved activations");
    float* const tile = SAVE ? act + tile_index * act_rows(UV) * TILE_PTS : nullptr;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        p[ct] = tile_index * X3_WG_PTS + wave * WAVE_PTS + ct * 16 + (lane & 15);
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
        park(2 * ct + 0, e);
        park(2 * ct + 1, e + 8);
        const int pw = wave * WAVE_PTS + ct * 16 + (lane & 15);   // point inside the tile
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
            park(2 * NCT + ct, dd);
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
    }

    Stream3 st;
    st.ring = ring;
    st.mine = ring + wave * WAVE_BLOCKS3 * 1024;
    st.g = packed + wave * WAVE_BLOCKS3 * 1024 + lane * 16;
    st.next = 0;
    st.nslabs = x3_stream_slabs(UV);
    __syncthreads();
    st.fetch(0);
    st.fetch(1);
    const char* cur = nullptr;

    bf16x8 X[3][NCT][8], Y[3][NCT][8];
    float sig[NCT][3], rgb[NCT][3];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int k = 0; k < 3; ++k) sig[ct][k] = rgb[ct][k] = 0.0f;
    const float NOCLAMP = -__builtin_inff();
    const float* bias = aux + AUX16_BIAS;
    const float* hw = aux + AUX16_HEAD;
    auto fromE = [&]<int ct, int ks, int s>() -> bf16x8 { return ed[(s * NF + 2 * ct + ks) * 64]; };
    auto fromX = [&]<int ct, int ks, int s>() -> bf16x8 { return X[s][ct][ks]; };
    auto fromY = [&]<int ct, int ks, int s>() -> bf16x8 { return Y[s][ct][ks]; };
    auto fromXE = [&]<int ct, int ks, int s>() -> bf16x8 {
        if constexpr (ks < 8) return X[s][ct][ks]; else return ed[(s * NF + 2 * ct + ks - 8) * 64];
    };
    auto fromXD = [&]<int ct, int ks, int s>() -> bf16x8 {
        if constexpr (ks < 8) return X[s][ct][ks]; else return ed[(s * NF + 2 * NCT + ct) * 64];
    };

    // training: per column tile, this lane's slot in the tile (row group g, its point) and the mask-word writer
    float* lane_rows[NCT];
    uint32_t mb[NCT][4];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        lane_rows[ct] = SAVE ? tile + g * GROUP_FLOATS + (wave * WAVE_PTS + ct * 16 + (lane & 15)) * 4 : nullptr;
        mb[ct][0] = mb[ct][1] = mb[ct][2] = mb[ct][3] = 0u;
    }
    // The backward-data kernels read one 16-byte word per masked layer and lane (point, half h): lanes (point, g = h) and
    // (point, g = h + 2) of this kernel hold its bits between them, in nibbles that alternate (mlp_fwd_x4.hip, save_mask).
    auto save_mask = [&](int m) {
        if constexpr (SAVE) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                u32x4 w;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const uint32_t mine = mb[ct][d] << (4 * (g >> 1));
                    w[d] = mine | (uint32_t)__shfl_xor((int)mine, 32, 64);
                }
                const int pw = wave * WAVE_PTS + ct * 16 + (lane & 15);
                if (g < 2) *reinterpret_cast<u32x4*>(tile + act_mask_offset(UV, m, pw >> 5, g * 32 + (pw & 31))) = w;
            }
        }
    };
    auto rows = [&](int row, float* (&dst)[NCT]) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) dst[ct] = lane_rows[ct] + (row / 4) * GROUP_FLOATS;
    };
    float* sr[NCT];
#define X3_LAYER(NOUT, NKS, NHEAD, STORE, MASKED, IN, OUT, BIAS, LO, HEADW, HSTRIDE, HEAD, ROW)                        \
    rows(ROW, sr);                                                                                                    \
    layer_x3<NOUT, NKS, NHEAD, STORE, SAVE, SAVE && MASKED>(st, lane, g, cur, IN, OUT, BIAS, LO, HEADW, HSTRIDE, HEAD, sr, mb)

    X3_LAYER(16, 2, 0, true, true, fromE, X, bias, 0.0f, hw, 0, sig, ACT_H0);                               // L0
    save_mask(0);
    X3_LAYER(16, 8, 0, true, true, fromX, Y, bias + 1 * 256, 0.0f, hw, 0, sig, ACT_H0 + 1 * 256);           // L1
    save_mask(1);
    X3_LAYER(16, 8, 0, true, true, fromY, X, bias + 2 * 256, 0.0f, hw, 0, sig, ACT_H0 + 2 * 256);           // L2
    save_mask(2);
    X3_LAYER(16, 8, 0, true, true, fromX, Y, bias + 3 * 256, 0.0f, hw, 0, sig, ACT_H0 + 3 * 256);           // L3
    save_mask(3);
    X3_LAYER(16, 8, 0, true, true, fromY, X, bias + 4 * 256, 0.0f, hw, 0, sig, ACT_H0 + 4 * 256);           // L4
    save_mask(4);
    X3_LAYER(16, 10, 0, true, true, fromXE, Y, bias + 5 * 256, 0.0f, hw, 0, sig, ACT_H0 + 5 * 256);         // L5 (skip)
    save_mask(5);
    X3_LAYER(16, 8, 0, true, true, fromY, X, bias + 6 * 256, 0.0f, hw, 0, sig, ACT_H0 + 6 * 256);           // L6
    save_mask(6);
    auto over_groups = [](float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); };
    if constexpr (VIEWDIRS) {
        X3_LAYER(16, 8, 1, true, true, fromX, Y, bias + 7 * 256, 0.0f, hw, 0, sig, ACT_H0 + 7 * 256);       // L7 + alpha head
        save_mask(7);
        X3_LAYER(16, 8, 0, true, false, fromY, X, bias + 8 * 256, NOCLAMP, hw, 0, sig, ACT_FEAT);          // feature
        X3_LAYER(8, 9, 3, false, true, fromXD, Y, bias + 9 * 256, 0.0f, hw + 256, 128, rgb, ACT_VIEWS);    // views + rgb heads
        save_mask(8);
        const float* hb = hw + 256 + 384;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 3; ++k) o[k] = over_groups(rgb[ct][k]) + hb[1 + k];
            o[3] = over_groups(sig[ct][0]) + hb[0];
            if (valid[ct] && g == (ct & 3)) *reinterpret_cast<f32x4*>(out + p[ct] * 4) = o;
        }
    } else {
        // output_linear: rows 0..2 as fp32 head dot products inside the L7 epilogue, rows 3..7 from the pieces afterwards
        X3_LAYER(16, 8, 3, true, true, fromX, Y, bias + 7 * 256, 0.0f, hw, 256, sig, ACT_H0 + 7 * 256);
        save_mask(7);
        for (int ch = 0; ch < output_ch; ++ch) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                float s;
                if (ch < 3) {
                    s = ch == 0 ? sig[ct][0] : (ch == 1 ? sig[ct][1] : sig[ct][2]);
                } else {
                    const float* wp = hw + ch * 256;
                    s = 0.0f;
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float y = ((float)Y[0][ct][ks][j] + (float)Y[1][ct][ks][j]) + (float)Y[2][ct][ks][j];
                            s = fmaf(y, wp[32 * ks + kslot_feature(g, j)], s);
                        }
                }
                s = over_groups(s) + hw[8 * 256 + ch];
                if (valid[ct] && g == (ct & 3)) out[p[ct] * output_ch + ch] = s;
            }
        }
    }
#undef X3_LAYER
}

// `marked` null: one workgroup per tile (the grid is the number of tiles).  Otherwise one word per 128-point tile, only
// tiles whose word is not zero are computed (those the two-piece fp16 kernel handed back, mlp_fwd_f16x2.hip), and a small
// grid walks the words: with nothing marked the launch is over in a few microseconds whatever the number of tiles.
template <bool VIEWDIRS, bool SAVE>
__global__ void __launch_bounds__(X3_THREADS, 1)
k_mlp_fwd_bf16x3(int64_t n_pts, int pts_per_ray, const float* __restrict__ pts, const float* __restrict__ viewdirs,
                 int output_ch, const char* __restrict__ packed, float* __restrict__ out, float* __restrict__ act,
                 const uint32_t* __restrict__ marked) {
    const int64_t n_tiles = (n_pts + X3_WG_PTS - 1) / X3_WG_PTS;
    // workgroup b looks at the tiles b, b + G, b + 2 G, ... (G workgroups), X3_THREADS of them at a time, one word per
    // thread, and works through the marked ones: neighbouring tiles go to different workgroups.  (Without `marked` the
    // grid is the number of tiles and thread 0's tile, b, is the only one in range.)
    __shared__ unsigned long long found[X3_WAVES];
    const int tid = threadIdx.x;
    for (int64_t base = blockIdx.x; base < n_tiles; base += (int64_t)gridDim.x * X3_THREADS) {
        const int64_t mine = base + (int64_t)tid * gridDim.x;
        const unsigned long long b = __ballot(mine < n_tiles && (marked == nullptr || marked[mine] != 0u));
        if ((tid & 63) == 0) found[tid >> 6] = b;
        __syncthreads();
        for (int w = 0; w < X3_WAVES; ++w) {
            unsigned long long todo = found[w];
            while (todo != 0ull) {
                const int i = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                const int64_t t = base + (int64_t)(64 * w + i) * gridDim.x;
                fwd_tile_x3<VIEWDIRS, SAVE>(t, n_pts, pts_per_ray, pts, viewdirs, output_ch, packed, out, act);
                __syncthreads();      // (the next tile of this workgroup refills the LDS tables)
            }
        }
        __syncthreads();
    }
}

// flat parameters -> stream of triples + fp32 aux tables
__global__ void __launch_bounds__(256) k_pack_bf16x3(int use_viewdirs, int output_ch, const float* __restrict__ params,
                                                     char* __restrict__ out) {
    const int64_t ntriples = bf16_stream_blocks(use_viewdirs);
    const int64_t elems = ntriples * 512;           // one per (triple, lane, slot)
    const int64_t total = elems + AUX16_FLOATS;
    unsigned short* s16 = reinterpret_cast<unsigned short*>(out);
    float* aux = reinterpret_cast<float*>(out + ntriples * TRIPLE_BYTES);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < elems) {
            const int64_t t = i >> 9;
            const int w = (int)(i & 511);
            const int64_t src = bf16_stream_source(use_viewdirs, output_ch, (int)t, w >> 3, w & 7);
            __bf16 h, m, l;
            split3(src < 0 ? 0.0f : params[src], h, m, l);
            unsigned short* dst = s16 + t * (TRIPLE_BYTES / 2) + w;
            dst[0] = __builtin_bit_cast(unsigned short, h);
            dst[512] = __builtin_bit_cast(unsigned short, m);
            dst[1024] = __builtin_bit_cast(unsigned short, l);
        } else {
            const int64_t src = aux16_source(use_viewdirs, output_ch, (int)(i - elems));
            aux[i - elems] = src < 0 ? 0.0f : params[src];
        }
    }
}

// Both split streams (and the forward one's fp32 tables) as ONE gather through a map: element pair (triple, w, w + 1) reads
// its two sources, splits them and writes one 4-byte word per piece.  seg0 triples | aux floats | seg1 triples.
__global__ void __launch_bounds__(256) k_gather_split3(const int32_t* __restrict__ table, const float* __restrict__ params,
                                                       char* __restrict__ out, int64_t triples0, int64_t n_aux, int64_t triples1) {
    gather_split3_elements(table, params, out, triples0, n_aux, triples1);
}

constexpr size_t x3_lds_bytes(int uv) { return RING3_BYTES + aux_floats(uv) * sizeof(float) + X3_WAVES * 3 * enc_frags(uv) * 64 * 16; }
static_assert(x3_lds_bytes(1) <= 160 * 1024 && x3_lds_bytes(0) <= 160 * 1024, "LDS of one compute unit");

}  // namespace

int nerf_mlp_fwd_bf16x3_marked(int64_t n_pts, int pts_per_ray, const float* pts, const float* viewdirs, int use_viewdirs,
                               int output_ch, const void* packed, float* out, float* act, const uint32_t* tile_flags,
                               void* stream);

extern "C" {

int64_t nerf_mlp_packed_bf16x3_bytes(int use_viewdirs) { return x3_packed_bytes(use_viewdirs); }
int64_t nerf_mlp_bf16x3_triples(int use_viewdirs) { return (int64_t)x3_stream_slabs(use_viewdirs) * SLABT; }

int nerf_pack_weights_bf16x3(int use_viewdirs, int output_ch, const float* params, void* packed, void* stream) {
    NERF_REQUIRE(params && packed, "nerf_pack_weights_bf16x3: null pointer");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_pack_weights_bf16x3: output_ch=%d not in 1..8", output_ch);
    hipLaunchKernelGGL(k_pack_bf16x3, dim3(1024), dim3(256), 0, as_stream(stream), use_viewdirs, output_ch, params,
                       reinterpret_cast<char*>(packed));
    return launched("nerf_pack_weights_bf16x3");
}

int nerf_gather_split3(const int32_t* table, const float* params, void* out, int64_t triples0, int64_t n_aux, int64_t triples1,
                       void* stream) {
    NERF_REQUIRE(triples0 >= 0 && n_aux >= 0 && triples1 >= 0, "nerf_gather_split3: negative count");
    if (triples0 + n_aux + triples1 == 0) return 0;
    NERF_REQUIRE(table && params && out, "nerf_gather_split3: null pointer");
    NERF_REQUIRE(n_aux % 4 == 0, "nerf_gather_split3: n_aux=%lld must be a multiple of 4 (the second stream starts 16-byte aligned)",
                 (long long)n_aux);
    NERF_REQUIRE((reinterpret_cast<uintptr_t>(table) & 7) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                 "nerf_gather_split3: table must be 8-byte and out 16-byte aligned");
    hipLaunchKernelGGL(k_gather_split3, dim3(1024), dim3(256), 0, as_stream(stream), table, params, reinterpret_cast<char*>(out),
                       triples0, n_aux, triples1);
    return launched("nerf_gather_split3");
}

int nerf_mlp_fwd_bf16x3(int64_t n_pts, int pts_per_ray, const float* pts, const float* viewdirs, int use_viewdirs,
                        int output_ch, const void* packed, float* out, float* act, void* stream) {
    return nerf_mlp_fwd_bf16x3_marked(n_pts, pts_per_ray, pts, viewdirs, use_viewdirs, output_ch, packed, out, act, nullptr, stream);
}

}  // extern "C"

// the same launch restricted to the tiles marked in tile_flags (all tiles when it is null)
int nerf_mlp_fwd_bf16x3_marked(int64_t n_pts, int pts_per_ray, const float* pts, const float* viewdirs, int use_viewdirs,
                               int output_ch, const void* packed, float* out, float* act, const uint32_t* tile_flags,
                               void* stream) {
    NERF_REQUIRE(n_pts >= 0, "nerf_mlp_fwd_bf16x3: n_pts=%lld", (long long)n_pts);
    if (n_pts == 0) return 0;
    NERF_REQUIRE(pts && packed && out, "nerf_mlp_fwd_bf16x3: null pointer");
    NERF_REQUIRE(!use_viewdirs || (viewdirs && pts_per_ray >= 1), "nerf_mlp_fwd_bf16x3: view branch needs viewdirs and pts_per_ray >= 1");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_mlp_fwd_bf16x3: output_ch=%d not in 1..8", output_ch);
    NERF_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                 "nerf_mlp_fwd_bf16x3: packed and out must be 16-byte aligned");
    const int64_t blocks = (n_pts + X3_WG_PTS - 1) / X3_WG_PTS;
    NERF_REQUIRE(blocks <= 0x7fffffff, "nerf_mlp_fwd_bf16x3: n_pts=%lld too large for one launch", (long long)n_pts);
    auto go = [&](auto kernel, int ppr, const float* vd, int och) -> int {
        const size_t lds = x3_lds_bytes(use_viewdirs);
        hipError_t e = lds_limit_once(reinterpret_cast<const void*>(kernel), lds);
        if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_fwd_bf16x3: LDS attribute: %s", hipGetErrorString(e));
        const int64_t grid = (tile_flags != nullptr && blocks > compute_units()) ? compute_units() : blocks;
        hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(X3_THREADS), lds, as_stream(stream), n_pts, ppr, pts,
                           vd, och, reinterpret_cast<const char*>(packed), out, act, tile_flags);
        return launched("nerf_mlp_fwd_bf16x3");
    };
    NERF_REQUIRE(!act || (reinterpret_cast<uintptr_t>(act) & 15) == 0, "nerf_mlp_fwd_bf16x3: act must be 16-byte aligned");
    if (use_viewdirs) return act ? go(k_mlp_fwd_bf16x3<true, true>, pts_per_ray, viewdirs, 4) : go(k_mlp_fwd_bf16x3<true, false>, pts_per_ray, viewdirs, 4);
    return act ? go(k_mlp_fwd_bf16x3<false, true>, 1, nullptr, output_ch) : go(k_mlp_fwd_bf16x3<false, false>, 1, nullptr, output_ch);
}
