// Fused positional encoding + NeRF MLP forward, bf16 operands / fp32 accumulation (inference variant).
//
// Same dataflow as the fp32 kernel (mlp_fwd.hip): activations stay in registers from the encoding to the
// heads, weights stream through an LDS slab ring.  Built on v_mfma_f32_16x16x32_bf16, the bf16 shape that
// holds the highest clock under load on this chip:
//   * a wave carries 32 points as TWO 16-point column tiles that share every A (weight) fragment; a
//     workgroup is eight waves (256 points), two per SIMD, each under 256 registers;
//   * an output tile is 16 features x 16 points, 4 fp32 per lane: lane l = (point l&15, group g = l>>4)
//     holds features 16t + 4g + r.  Two consecutive row tiles (2s, 2s+1) therefore give a lane 8 values,
//     exactly one B fragment of k-step s of the next layer if k-slot j of group g stands for feature
//     32s + kslot_feature(g, j); the weight stream is packed in that k order, so rounding to bf16 and
//     packing pairwise is all that happens between layers.  No data leaves the register file;
//   * the bias is the accumulators' initial value (LDS -> accumulator registers while the previous tile is
//     multiplying) and the ReLU / rounding / packing of a tile is issued piecewise behind the first MFMAs
//     of the next tile; the 1- and 3-row heads are fp32 FMAs on the unrounded accumulators;
//   * the encodings are computed in fp32 and rounded once; each of the four lane groups of a point
//     evaluates a quarter of the sin/cos pairs.
// Numerics: inputs of every product are rounded to 8 significant bits, sums are fp32.  Against the
// fp32 path the raw outputs differ by about 1e-2 of their scale (tolerance stated in the tests).
#include "common.hpp"
#include "mlp_core.hpp"
#include "mlp_tile16.hpp"
#include "mlp_bf16_layout.hpp"

using namespace nerfhip;
using namespace nerflayout;
using namespace nerfmlp;
using namespace nerftile16;
using namespace nerfbf16;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// 16-point column tiles per wave.  Two (eight waves per workgroup, two per SIMD) is the measured optimum: four tiles
// at one wave per SIMD halve the LDS reads per FLOP but ran 14 % slower (no partner wave to cover barriers).
constexpr int NCT = 2;
constexpr int THREADS16 = 1024 / NCT;
constexpr int WAVE_PTS = 16 * NCT;
constexpr int WG_PTS = (THREADS16 / 64) * WAVE_PTS;   // points per workgroup (256)
constexpr int SLAB16 = 32;                    // 1 KiB blocks per slab
constexpr int SLAB16_BYTES = SLAB16 * 1024;
constexpr int RING16_BYTES = 2 * SLAB16_BYTES;

__host__ __device__ inline int bf16_stream_slabs(int use_viewdirs) { return (bf16_stream_blocks(use_viewdirs) + SLAB16 - 1) / SLAB16; }

__host__ __device__ inline int64_t bf16_packed_bytes(int use_viewdirs) {
    return (int64_t)bf16_stream_slabs(use_viewdirs) * SLAB16_BYTES + (int64_t)AUX16_FLOATS * 4;
}


// ---- the slab ring as one wave sees it ----------------------------------------------------------
struct Stream16 {
    const char* g;   // this lane's source for block 4*wave of slab 0
    char* ring;
    char* mine;
    int next, nslabs;
    __device__ __forceinline__ void fetch(int slab) {
        const char* src = g + (int64_t)slab * SLAB16_BYTES;
        char* dst = mine + (slab & 1) * SLAB16_BYTES;
#pragma unroll
        for (int b = 0; b < SLAB16 / (THREADS16 / 64); ++b)
            glds16(reinterpret_cast<const float*>(src + b * 1024), reinterpret_cast<float*>(dst + b * 1024));
    }
    __device__ __forceinline__ const char* advance(int lane) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* cur = ring + (next & 1) * SLAB16_BYTES + lane * 16;
        if (next + 1 < nslabs) fetch(next + 1);
        ++next;
        return cur;
    }
};

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 pack8(const float* v) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (__bf16)v[i];
    return r;
}

// One layer: NOUT row tiles (16 features each) x NKS k-steps (32 input features each) x 2 column tiles.
// Two accumulator sets alternate; the epilogue of row tile t -- y = max(acc, lo) (lo = 0: ReLU, -inf: none), head
// rows as fp32 dot products with y, y rounded to bf16 into slots 4(t%2)..+3 of fragment t/2 of both column tiles --
// is cut into four pieces (two registers -> one packed dword) issued one per k-step behind the first MFMAs of
// row tile t+1.  Layers with at least seven k-steps start each accumulator as the tile's bias column; the two-step
// first layer adds the bias in the epilogue.  Blocks are read DEPTH ahead, also across tiles and slab barriers.
template <int NOUT, int NKS, int NHEAD, bool STORE, typename InFrag>
__device__ __forceinline__ void layer16(Stream16& st, int lane, int g, const char*& cur, InFrag in_frag, bf16x8 (&Y)[NCT][8],
                                        const float* bias_lds, float lo, const float* head_lds, int head_stride,
                                        float (&head)[NCT][3]) {
    constexpr int NB = NOUT * NKS, DEPTH = 3, NPIECE = 2 * NCT;
    constexpr bool BIAS_INIT = NKS >= 7;
    constexpr int PPS = (NPIECE + (NKS < 8 ? NKS : 8) - 1) / (NKS < 8 ? NKS : 8);   // epilogue pieces per k-step
    bf16x8 abuf[DEPTH + 1];
    f32x4 acc[2][NCT];                            // [alternating set][column tile]
    const float* bp = bias_lds + 4 * g;
    const float* hp = head_lds + 4 * g;

    auto load_bias = [&]<int TO>() {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(bp + 16 * TO);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[TO & 1][ct] = bb;
    };
    auto piece = [&]<int TP, int K>() {           // K = 2 * column tile + register pair
        constexpr int ct = K >> 1, pr = K & 1;
        float v0 = acc[TP & 1][ct][2 * pr], v1 = acc[TP & 1][ct][2 * pr + 1];
        if constexpr (!BIAS_INIT) {
            const f32x2 bb = *reinterpret_cast<const f32x2*>(bp + 16 * TP + 2 * pr);
            v0 += bb[0];
            v1 += bb[1];
        }
        v0 = max_keep_nan(v0, lo);
        v1 = max_keep_nan(v1, lo);
#pragma unroll
        for (int k = 0; k < NHEAD; ++k) {
            const f32x2 w = *reinterpret_cast<const f32x2*>(hp + k * head_stride + 16 * TP + 2 * pr);
            head[ct][k] = fmaf(v1, w[1], fmaf(v0, w[0], head[ct][k]));
            // pin the head sums here: left alone the optimiser sinks their FMAs to the kernel's end and keeps every
            // fp32 activation alive until then
            asm volatile("" : "+v"(head[ct][k]));
        }
        if constexpr (STORE) {
            Y[ct][TP >> 1][4 * (TP & 1) + 2 * pr] = (__bf16)v0;
            Y[ct][TP >> 1][4 * (TP & 1) + 2 * pr + 1] = (__bf16)v1;
        }
    };

    if constexpr (BIAS_INIT) load_bias.template operator()<0>();
    static_for<NB + DEPTH>([&]<int I>() {
        if constexpr (I < NB) {
            if constexpr (I % SLAB16 == 0) cur = st.advance(lane);
            abuf[I % (DEPTH + 1)] = *reinterpret_cast<const bf16x8*>(cur + (I % SLAB16) * 1024);
        }
        if constexpr (I >= DEPTH) {
            constexpr int b = I - DEPTH;
            constexpr int to = b / NKS, ks = b % NKS;
            const bf16x8 a = abuf[b % (DEPTH + 1)];
            static_for<NCT>([&]<int CT>() {
                if constexpr (!BIAS_INIT && ks == 0) {
                    const f32x4 zero = {0, 0, 0, 0};
                    acc[to & 1][CT] = mfma16(a, in_frag.template operator()<CT, ks>(), zero);
                } else {
                    acc[to & 1][CT] = mfma16(a, in_frag.template operator()<CT, ks>(), acc[to & 1][CT]);
                }
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

template <bool VIEWDIRS>
__global__ void __launch_bounds__(THREADS16, NCT == 2 ? 2 : 1)
k_mlp_fwd_bf16(int64_t n_pts, int pts_per_ray, const float* __restrict__ pts, const float* __restrict__ viewdirs,
               int output_ch, const char* __restrict__ packed, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    char* ring = smem16;
    float* aux = reinterpret_cast<float*>(smem16 + RING16_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    constexpr int UV = VIEWDIRS ? 1 : 0;
    {
        const float* gaux = reinterpret_cast<const float*>(packed + (int64_t)bf16_stream_slabs(UV) * SLAB16_BYTES);
        for (int i = tid * 4; i < AUX16_FLOATS; i += THREADS16 * 4)
            *reinterpret_cast<f32x4*>(aux + i) = *reinterpret_cast<const f32x4*>(gaux + i);
    }

    // ---- encodings: this lane's group evaluates its share of the sin/cos pairs for its two points
    bf16x8* ed = reinterpret_cast<bf16x8*>(smem16 + RING16_BYTES + AUX16_FLOATS * 4) + wave * (3 * NCT * 64) + lane;
    int64_t p[NCT];
    bool valid[NCT];
    const int c0 = enc16_first_pair(g);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        p[ct] = (int64_t)blockIdx.x * WG_PTS + wave * WAVE_PTS + ct * 16 + (lane & 15);
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
        ed[(2 * ct + 0) * 64] = pack8(e);
        ed[(2 * ct + 1) * 64] = pack8(e + 8);
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
            ed[(2 * NCT + ct) * 64] = pack8(dd);
        }
    }

    Stream16 st;
    st.ring = ring;
    st.mine = ring + wave * (SLAB16 / (THREADS16 / 64)) * 1024;
    st.g = packed + wave * (SLAB16 / (THREADS16 / 64)) * 1024 + lane * 16;
    st.next = 0;
    st.nslabs = bf16_stream_slabs(UV);
    __syncthreads();
    st.fetch(0);
    const char* cur = nullptr;

    bf16x8 X[NCT][8], Y[NCT][8];
    float sig[NCT][3], rgb[NCT][3];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int k = 0; k < 3; ++k) sig[ct][k] = rgb[ct][k] = 0.0f;
    const float NOCLAMP = -__builtin_inff();
    const float* bias = aux + AUX16_BIAS;
    const float* hw = aux + AUX16_HEAD;
    // the encodings wait in LDS (6 fragments per lane) so that two waves per SIMD fit the register file
    auto fromE = [&]<int ct, int ks>() -> bf16x8 { return ed[(2 * ct + ks) * 64]; };
    auto fromX = [&]<int ct, int ks>() -> bf16x8 { return X[ct][ks]; };
    auto fromY = [&]<int ct, int ks>() -> bf16x8 { return Y[ct][ks]; };
    auto fromXE = [&]<int ct, int ks>() -> bf16x8 { if constexpr (ks < 8) return X[ct][ks]; else return ed[(2 * ct + ks - 8) * 64]; };
    auto fromXD = [&]<int ct, int ks>() -> bf16x8 { if constexpr (ks < 8) return X[ct][ks]; else return ed[(2 * NCT + ct) * 64]; };

    layer16<16, 2, 0, true>(st, lane, g, cur, fromE, X, bias, 0.0f, hw, 0, sig);                    // L0
    layer16<16, 8, 0, true>(st, lane, g, cur, fromX, Y, bias + 1 * 256, 0.0f, hw, 0, sig);          // L1
    layer16<16, 8, 0, true>(st, lane, g, cur, fromY, X, bias + 2 * 256, 0.0f, hw, 0, sig);          // L2
    layer16<16, 8, 0, true>(st, lane, g, cur, fromX, Y, bias + 3 * 256, 0.0f, hw, 0, sig);          // L3
    layer16<16, 8, 0, true>(st, lane, g, cur, fromY, X, bias + 4 * 256, 0.0f, hw, 0, sig);          // L4
    layer16<16, 10, 0, true>(st, lane, g, cur, fromXE, Y, bias + 5 * 256, 0.0f, hw, 0, sig);        // L5 (skip)
    layer16<16, 8, 0, true>(st, lane, g, cur, fromY, X, bias + 6 * 256, 0.0f, hw, 0, sig);          // L6
    auto over_groups = [](float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); };
    if constexpr (VIEWDIRS) {
        layer16<16, 8, 1, true>(st, lane, g, cur, fromX, Y, bias + 7 * 256, 0.0f, hw, 0, sig);      // L7 + alpha head
        layer16<16, 8, 0, true>(st, lane, g, cur, fromY, X, bias + 8 * 256, NOCLAMP, hw, 0, sig);   // feature
        layer16<8, 9, 3, false>(st, lane, g, cur, fromXD, Y, bias + 9 * 256, 0.0f, hw + 256, 128, rgb);   // views + rgb heads
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
        // output_linear: rows 0..2 as fp32 head dot products inside the L7 epilogue, rows 3..7 from the rounded
        // activations afterwards
        layer16<16, 8, 3, true>(st, lane, g, cur, fromX, Y, bias + 7 * 256, 0.0f, hw, 256, sig);
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
                        for (int j = 0; j < 8; ++j) s = fmaf((float)Y[ct][ks][j], wp[32 * ks + kslot_feature(g, j)], s);
                }
                s = over_groups(s) + hw[8 * 256 + ch];
                if (valid[ct] && g == (ct & 3)) out[p[ct] * output_ch + ch] = s;
            }
        }
    }
}

// flat parameters -> bf16 stream + fp32 aux tables
__global__ void __launch_bounds__(256) k_pack_bf16(int use_viewdirs, int output_ch, const float* __restrict__ params,
                                                   char* __restrict__ out) {
    const int64_t nblk = bf16_stream_blocks(use_viewdirs);
    const int64_t padded = (int64_t)bf16_stream_slabs(use_viewdirs) * SLAB16 * 512;
    const int64_t total = padded + AUX16_FLOATS;
    unsigned short* s16 = reinterpret_cast<unsigned short*>(out);
    float* aux = reinterpret_cast<float*>(out + padded * 2);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < padded) {
            const int64_t b = i >> 9;
            const int w = (int)(i & 511);
            const int64_t src = b < nblk ? bf16_stream_source(use_viewdirs, output_ch, (int)b, w >> 3, w & 7) : -1;
            const __bf16 r = (__bf16)(src < 0 ? 0.0f : params[src]);
            s16[i] = __builtin_bit_cast(unsigned short, r);
        } else {
            const int64_t src = aux16_source(use_viewdirs, output_ch, (int)(i - padded));
            aux[i - padded] = src < 0 ? 0.0f : params[src];
        }
    }
}

constexpr size_t FWD16_LDS_BYTES = RING16_BYTES + AUX16_FLOATS * sizeof(float) + (THREADS16 / 64) * 3 * NCT * 64 * 16;

}  // namespace

extern "C" {

int64_t nerf_mlp_packed_bf16_bytes(int use_viewdirs) { return bf16_packed_bytes(use_viewdirs); }

int nerf_pack_weights_bf16(int use_viewdirs, int output_ch, const float* params, void* packed_bf16, void* stream) {
    NERF_REQUIRE(params && packed_bf16, "nerf_pack_weights_bf16: null pointer");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_pack_weights_bf16: output_ch=%d not in 1..8", output_ch);
    hipLaunchKernelGGL(k_pack_bf16, dim3(1024), dim3(256), 0, as_stream(stream), use_viewdirs, output_ch, params,
                       reinterpret_cast<char*>(packed_bf16));
    return launched("nerf_pack_weights_bf16");
}

int nerf_mlp_fwd_bf16(int64_t n_pts, int pts_per_ray, const float* pts, const float* viewdirs, int use_viewdirs,
                      int output_ch, const void* packed_bf16, float* out, void* stream) {
    NERF_REQUIRE(n_pts >= 0, "nerf_mlp_fwd_bf16: n_pts=%lld", (long long)n_pts);
    if (n_pts == 0) return 0;
    NERF_REQUIRE(pts && packed_bf16 && out, "nerf_mlp_fwd_bf16: null pointer");
    NERF_REQUIRE(!use_viewdirs || (viewdirs && pts_per_ray >= 1), "nerf_mlp_fwd_bf16: view branch needs viewdirs and pts_per_ray >= 1");
    NERF_REQUIRE(use_viewdirs || (output_ch >= 1 && output_ch <= 8), "nerf_mlp_fwd_bf16: output_ch=%d not in 1..8", output_ch);
    NERF_REQUIRE((reinterpret_cast<uintptr_t>(packed_bf16) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                 "nerf_mlp_fwd_bf16: packed_bf16 and out must be 16-byte aligned");
    const int64_t blocks = (n_pts + WG_PTS - 1) / WG_PTS;
    NERF_REQUIRE(blocks <= 0x7fffffff, "nerf_mlp_fwd_bf16: n_pts=%lld too large for one launch", (long long)n_pts);
    auto go = [&](auto kernel, int ppr, const float* vd, int och) -> int {
        hipError_t e = lds_limit_once(reinterpret_cast<const void*>(kernel), FWD16_LDS_BYTES);
        if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_fwd_bf16: LDS attribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(THREADS16), FWD16_LDS_BYTES, as_stream(stream), n_pts, ppr, pts,
                           vd, och, reinterpret_cast<const char*>(packed_bf16), out);
        return launched("nerf_mlp_fwd_bf16");
    };
    if (use_viewdirs) return go(k_mlp_fwd_bf16<true>, pts_per_ray, viewdirs, 4);
    return go(k_mlp_fwd_bf16<false>, 1, nullptr, output_ch);
}

}  // extern "C"
