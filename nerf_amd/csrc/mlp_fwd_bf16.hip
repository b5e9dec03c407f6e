// Fused positional encoding + NeRF MLP forward, bf16 operands / fp32 accumulation (inference variant).
//
// Same dataflow as the fp32 kernel (mlp_fwd.hip): activations stay in registers from the encoding to
// the heads, weights stream through an LDS slab ring.  Differences that follow from the 16x faster
// v_mfma_f32_32x32x16_bf16:
//   * a workgroup is EIGHT waves (256 points), two per SIMD, each under 256 registers: while one wave of a
//     SIMD runs its VALU epilogue (bias, ReLU, bf16 rounding) the other keeps the matrix pipe busy, and
//     the L2 -> LDS weight traffic per point is half that of a four-wave group;
//   * after ReLU (fp32, VALU; the bias is the accumulator's initial value) the accumulators are rounded to bf16
//     and packed pairwise, in the shadow of the next tile's MFMAs; registers
//     8s..8s+7 of an accumulator tile are exactly the B fragment of k-step s of the next layer
//     (k = 16s + 8(j>>2) + 4*half + (j&3)), so again no data leaves the register file.  The stream is
//     the fp32 stream re-blocked (two fp32 blocks -> one 1 KiB block of 64 lanes x 8 bf16);
//   * the 1- and 3-row heads read the fp32 accumulators (before rounding), the encodings are computed
//     in fp32 and rounded once.
// Numerics: inputs of every product are rounded to 8 significant bits, sums are fp32.  Against the
// fp32 path the raw outputs differ by about 1e-2 of their scale (tolerance stated in the tests).
#include "common.hpp"
#include "mlp_core.hpp"

using namespace nerfhip;
using namespace nerflayout;
using namespace nerfmlp;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NCOL = 1;                       // 32-point column tiles per wave
constexpr int THREADS16 = 512;                // eight waves, two per SIMD
constexpr int WG_PTS = (THREADS16 / 64) * 32 * NCOL;  // points per workgroup
constexpr int SLAB16 = 32;                    // 1 KiB blocks per slab
constexpr int SLAB16_BYTES = SLAB16 * 1024;
constexpr int RING16_BYTES = 2 * SLAB16_BYTES;

__host__ __device__ inline int bf16_stream_blocks(int use_viewdirs) { return fwd_stream_blocks(use_viewdirs) / 2; }
__host__ __device__ inline int bf16_stream_slabs(int use_viewdirs) { return (bf16_stream_blocks(use_viewdirs) + SLAB16 - 1) / SLAB16; }
__host__ __device__ inline int64_t bf16_packed_bytes(int use_viewdirs) {
    return (int64_t)bf16_stream_slabs(use_viewdirs) * SLAB16_BYTES + (int64_t)AUX_FLOATS * 4;
}

struct Stream16 {
    const char* g;   // this lane's source for block 8*wave of slab 0
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

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 pack8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
    bf16x8 r;
    r[0] = (__bf16)v0; r[1] = (__bf16)v1; r[2] = (__bf16)v2; r[3] = (__bf16)v3;
    r[4] = (__bf16)v4; r[5] = (__bf16)v5; r[6] = (__bf16)v6; r[7] = (__bf16)v7;
    return r;
}

// One layer: for every 32-row output tile, NKS MFMAs (k-steps of 16) on the input fragments, then the tile's
// epilogue on the VALU: y = max(acc, lo) (lo = 0: ReLU, -inf: none), optional head rows as fp32 dot products
// with y, and y rounded to bf16 into the two B fragments the next layer reads (Y[2*to], Y[2*to+1]).
// Two accumulator tiles alternate, and the epilogue is cut into eight pieces (two registers -> one packed
// dword each) issued one per k-step behind the first eight MFMAs of the NEXT tile, so it runs in the shadow
// of the matrix pipe instead of stalling it; only a layer's last tile is finished in the open.  The bias is
// not added afterwards: an accumulator starts as the tile's bias column, read from LDS straight into its
// registers while the previous tile is still multiplying (layers with at least 12 k-steps; the 4-step first
// layer adds it in the epilogue).  Weight blocks are read DEPTH ahead, also across tiles and slab barriers.
// Every layer starts on a slab boundary.
template <int NOUT, int NKS, int NHEAD, bool STORE, typename InFrag>
__device__ __forceinline__ void layer16(Stream16& st, int lane, int h, const char*& cur, InFrag in_frag, bf16x8 (&Y)[16],
                                        const float* bias_lds, float lo, const float* head_lds, int head_stride,
                                        float (&head)[3]) {
    constexpr int NB = NOUT * NKS, DEPTH = 3;
    constexpr bool BIAS_INIT = NKS >= 12;
    constexpr int PPS = NKS >= 8 ? 1 : 8 / NKS;   // epilogue pieces per k-step
    bf16x8 abuf[DEPTH + 1];
    f32x16 acc[2];
    const float* bp = bias_lds + h * 16;
    const float* hp = head_lds + h * 16;

    auto load_bias = [&]<int TO, int Q>() {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(bp + TO * 32 + 4 * Q);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[TO & 1][4 * Q + e] = bb[e];
    };
    auto piece = [&]<int TP, int K>() {
        float v0 = acc[TP & 1][2 * K], v1 = acc[TP & 1][2 * K + 1];
        if constexpr (!BIAS_INIT) {
            const f32x2 bb = *reinterpret_cast<const f32x2*>(bp + TP * 32 + 2 * K);
            v0 += bb[0];
            v1 += bb[1];
        }
        v0 = fmaxf(v0, lo);
        v1 = fmaxf(v1, lo);
#pragma unroll
        for (int k = 0; k < NHEAD; ++k) {
            const f32x2 w = *reinterpret_cast<const f32x2*>(hp + k * head_stride + TP * 32 + 2 * K);
            head[k] = fmaf(v1, w[1], fmaf(v0, w[0], head[k]));
            // pin the head sums here: left alone the optimiser sinks all their FMAs to the kernel's end and keeps
            // (spills) every fp32 activation until then
            asm volatile("" : "+v"(head[k]));
        }
        if constexpr (STORE) {
            Y[2 * TP + (K >> 2)][2 * (K & 3)] = (__bf16)v0;
            Y[2 * TP + (K >> 2)][2 * (K & 3) + 1] = (__bf16)v1;
        }
    };

    if constexpr (BIAS_INIT) static_for<4>([&]<int Q>() { load_bias.template operator()<0, Q>(); });
    static_for<NB + DEPTH>([&]<int I>() {
        if constexpr (I < NB) {
            if constexpr (I % SLAB16 == 0) cur = st.advance(lane);
            abuf[I % (DEPTH + 1)] = *reinterpret_cast<const bf16x8*>(cur + (I % SLAB16) * 1024);
        }
        if constexpr (I >= DEPTH) {
            constexpr int b = I - DEPTH;
            constexpr int to = b / NKS, ks = b % NKS;
            const bf16x8 a = abuf[b % (DEPTH + 1)];
            if constexpr (!BIAS_INIT && ks == 0) {
                const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                acc[to & 1] = mfma16(a, in_frag.template operator()<ks>(), zero);
            } else {
                acc[to & 1] = mfma16(a, in_frag.template operator()<ks>(), acc[to & 1]);
            }
            if constexpr (to > 0) {   // the previous tile's epilogue, piece by piece, behind this tile's MFMAs
                static_for<PPS>([&]<int Q>() {
                    constexpr int K = ks * PPS + Q;
                    if constexpr (K < 8) piece.template operator()<to - 1, K>();
                });
            }
            if constexpr (BIAS_INIT && to + 1 < NOUT && ks >= 8 && ks < 12) load_bias.template operator()<to + 1, ks - 8>();
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    static_for<8>([&]<int K>() { piece.template operator()<NOUT - 1, K>(); });
}

template <bool VIEWDIRS>
__global__ void __launch_bounds__(THREADS16, 2)
k_mlp_fwd_bf16(int64_t n_pts, int pts_per_ray, const float* __restrict__ pts, const float* __restrict__ viewdirs,
               int output_ch, const char* __restrict__ packed, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    char* ring = smem16;
    float* aux = reinterpret_cast<float*>(smem16 + RING16_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    constexpr int UV = VIEWDIRS ? 1 : 0;
    {
        const float* gaux = reinterpret_cast<const float*>(packed + (int64_t)bf16_stream_slabs(UV) * SLAB16_BYTES);
        for (int i = tid * 4; i < AUX_FLOATS; i += THREADS16 * 4)
            *reinterpret_cast<f32x4*>(aux + i) = *reinterpret_cast<const f32x4*>(gaux + i);
    }

    int64_t p[NCOL];
    bool valid[NCOL];
    bf16x8 E[NCOL][4], D[NCOL][2];
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        p[c] = (int64_t)blockIdx.x * WG_PTS + wave * (32 * NCOL) + c * 32 + (lane & 31);
        valid[c] = p[c] < n_pts;
        const int64_t pl = valid[c] ? p[c] : n_pts - 1;
        float e[32];
        const float x0 = pts[pl * 3 + 0], x1 = pts[pl * 3 + 1], x2 = pts[pl * 3 + 2];
        e[0] = h ? x2 : x0;
        e[1] = h ? 0.0f : x1;
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            const int cc = 15 * h + i, k = cc / 3, d = cc - 3 * k;
            const float arg = (d == 0 ? x0 : (d == 1 ? x1 : x2)) * __int_as_float((127 + k) << 23);
            sincosf(arg, &e[2 + 2 * i], &e[3 + 2 * i]);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
            E[c][s] = pack8(e[8 * s], e[8 * s + 1], e[8 * s + 2], e[8 * s + 3], e[8 * s + 4], e[8 * s + 5], e[8 * s + 6], e[8 * s + 7]);
        if (VIEWDIRS) {
            const int64_t ray = pl / pts_per_ray;
            const float v0 = viewdirs[ray * 3 + 0], v1 = viewdirs[ray * 3 + 1], v2 = viewdirs[ray * 3 + 2];
            float dd[16];
            dd[0] = h ? v2 : v0;
            dd[1] = h ? 0.0f : v1;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int cc = 6 * h + i, k = cc / 3, d = cc - 3 * k;
                const float arg = (d == 0 ? v0 : (d == 1 ? v1 : v2)) * __int_as_float((127 + k) << 23);
                sincosf(arg, &dd[2 + 2 * i], &dd[3 + 2 * i]);
            }
            dd[14] = 0.0f;
            dd[15] = 0.0f;
            D[c][0] = pack8(dd[0], dd[1], dd[2], dd[3], dd[4], dd[5], dd[6], dd[7]);
            D[c][1] = pack8(dd[8], dd[9], dd[10], dd[11], dd[12], dd[13], dd[14], dd[15]);
        }
    }

    bf16x8* ed = reinterpret_cast<bf16x8*>(smem16 + RING16_BYTES + AUX_FLOATS * 4) + wave * (6 * 64) + lane;
#pragma unroll
    for (int k = 0; k < 4; ++k) ed[k * 64] = E[0][k];
    if (VIEWDIRS) {
        ed[4 * 64] = D[0][0];
        ed[5 * 64] = D[0][1];
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

    bf16x8 X[16], Y[16];
    float sig[3] = {0.0f, 0.0f, 0.0f}, rgb[3] = {0.0f, 0.0f, 0.0f};
    const float NOCLAMP = -__builtin_inff();
    const float* bias = aux + AUX_BIAS;
    // the encodings wait in LDS (6 fragments per lane) so that two waves per SIMD fit the register file
    auto fromE = [&]<int ks>() -> bf16x8 { return ed[ks * 64]; };
    auto fromX = [&]<int ks>() -> bf16x8 { return X[ks]; };
    auto fromY = [&]<int ks>() -> bf16x8 { return Y[ks]; };
    auto fromXE = [&]<int ks>() -> bf16x8 { if constexpr (ks < 16) return X[ks]; else return ed[(ks - 16) * 64]; };
    auto fromXD = [&]<int ks>() -> bf16x8 { if constexpr (ks < 16) return X[ks]; else return ed[(4 + ks - 16) * 64]; };

    // stream (1 KiB blocks): L0 32 | L1..L4 128 each | L5 160 (per tile: 16 hidden + 4 encoded k-steps) |
    //                        L6, L7 128 | feature 128 | views 72
    layer16<8, 4, 0, true>(st, lane, h, cur, fromE, X, bias, 0.0f, aux, 0, sig);                    // L0
    layer16<8, 16, 0, true>(st, lane, h, cur, fromX, Y, bias + 1 * 256, 0.0f, aux, 0, sig);         // L1
    layer16<8, 16, 0, true>(st, lane, h, cur, fromY, X, bias + 2 * 256, 0.0f, aux, 0, sig);         // L2
    layer16<8, 16, 0, true>(st, lane, h, cur, fromX, Y, bias + 3 * 256, 0.0f, aux, 0, sig);         // L3
    layer16<8, 16, 0, true>(st, lane, h, cur, fromY, X, bias + 4 * 256, 0.0f, aux, 0, sig);         // L4
    layer16<8, 20, 0, true>(st, lane, h, cur, fromXE, Y, bias + 5 * 256, 0.0f, aux, 0, sig);        // L5 (skip)
    layer16<8, 16, 0, true>(st, lane, h, cur, fromY, X, bias + 6 * 256, 0.0f, aux, 0, sig);         // L6
    if constexpr (VIEWDIRS) {
        layer16<8, 16, 1, true>(st, lane, h, cur, fromX, Y, bias + 7 * 256, 0.0f, aux + AUX_HEAD, 0, sig);   // L7 + alpha
        layer16<8, 16, 0, true>(st, lane, h, cur, fromY, X, bias + 8 * 256, NOCLAMP, aux, 0, sig);           // feature
        layer16<4, 18, 3, false>(st, lane, h, cur, fromXD, Y, bias + 9 * 256, 0.0f, aux + AUX_HEAD + 256, 128, rgb);  // views + rgb
        const float* hb = aux + AUX_HEAD + 256 + 384;
        float o4[4];
#pragma unroll
        for (int k = 0; k < 3; ++k) o4[k] = rgb[k] + __shfl_xor(rgb[k], 32, 64) + hb[1 + k];
        o4[3] = sig[0] + __shfl_xor(sig[0], 32, 64) + hb[0];
        if (valid[0] && h == 0) {
            f32x4 o = {o4[0], o4[1], o4[2], o4[3]};
            *reinterpret_cast<f32x4*>(out + p[0] * 4) = o;
        }
    } else {
        // output_linear: up to 8 rows, taken in two passes of fp32 head dot products inside the L7 epilogue
        float h0[3] = {0.0f, 0.0f, 0.0f}, h1[3] = {0.0f, 0.0f, 0.0f};
        (void)h1;
        layer16<8, 16, 3, true>(st, lane, h, cur, fromX, Y, bias + 7 * 256, 0.0f, aux + AUX_HEAD, 256, h0);
        for (int ch = 0; ch < output_ch; ++ch) {
            float s;
            if (ch < 3) {
                s = ch == 0 ? h0[0] : (ch == 1 ? h0[1] : h0[2]);
            } else {  // rows 3..7 from the rounded activations
                const float* wp = aux + AUX_HEAD + ch * 256;
                s = 0.0f;
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            s = fmaf((float)Y[2 * t + sidx][j], wp[t * 32 + h * 16 + 8 * sidx + j], s);
            }
            s = s + __shfl_xor(s, 32, 64) + aux[AUX_HEAD + output_ch * 256 + ch];
            if (valid[0] && h == 0) out[p[0] * output_ch + ch] = s;
        }
    }
}

// fp32 stream -> bf16 stream: block b, lane, element j  <-  fp32 block 2b + (j>>2), lane, element j&3
__global__ void __launch_bounds__(256) k_pack_bf16(int use_viewdirs, const float* __restrict__ packed_fwd,
                                                   char* __restrict__ out) {
    const int64_t nblk = bf16_stream_blocks(use_viewdirs);
    const int64_t padded = (int64_t)bf16_stream_slabs(use_viewdirs) * SLAB16 * 512;
    const int64_t total = padded + AUX_FLOATS;
    unsigned short* s16 = reinterpret_cast<unsigned short*>(out);
    float* aux = reinterpret_cast<float*>(out + padded * 2);
    const float* src_aux = packed_fwd + (int64_t)fwd_stream_blocks(use_viewdirs) * BLOCK_FLOATS;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < padded) {
            const int64_t b = i >> 9;
            const int w = (int)(i & 511), lane = w >> 3, j = w & 7;
            float v = 0.0f;
            if (b < nblk) {
                int64_t fb = 2 * b + (j >> 2);           // two fp32 blocks (4 registers each) make one bf16 block
                if (b >= 544 && b < 704) {               // layer 5: per output tile, hidden k-steps then encoded ones
                    const int lb = (int)b - 544, to = lb / 20, ks = lb % 20;
                    // fp32 stream: encoded-input part at blocks 1088..1151 (8 per tile), hidden part at 1152..1407 (32 per tile)
                    fb = ks < 16 ? 1152 + to * 32 + 2 * ks + (j >> 2) : 1088 + to * 8 + 2 * (ks - 16) + (j >> 2);
                }
                v = packed_fwd[fb * BLOCK_FLOATS + lane * 4 + (j & 3)];
            }
            const __bf16 r = (__bf16)v;
            s16[i] = __builtin_bit_cast(unsigned short, r);
        } else {
            aux[i - padded] = src_aux[i - padded];
        }
    }
}

constexpr size_t FWD16_LDS_BYTES = RING16_BYTES + AUX_FLOATS * sizeof(float) + (THREADS16 / 64) * 6 * 64 * 16;

}  // namespace

extern "C" {

int64_t nerf_mlp_packed_bf16_bytes(int use_viewdirs) { return bf16_packed_bytes(use_viewdirs); }

int nerf_pack_weights_bf16(int use_viewdirs, const float* packed_fwd, void* packed_bf16, void* stream) {
    NERF_REQUIRE(packed_fwd && packed_bf16, "nerf_pack_weights_bf16: null pointer");
    hipLaunchKernelGGL(k_pack_bf16, dim3(1024), dim3(256), 0, as_stream(stream), use_viewdirs, packed_fwd,
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
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)FWD16_LDS_BYTES);
        if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_fwd_bf16: LDS attribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(THREADS16), FWD16_LDS_BYTES, as_stream(stream), n_pts, ppr, pts,
                           vd, och, reinterpret_cast<const char*>(packed_bf16), out);
        return launched("nerf_mlp_fwd_bf16");
    };
    if (use_viewdirs) return go(k_mlp_fwd_bf16<true>, pts_per_ray, viewdirs, 4);
    return go(k_mlp_fwd_bf16<false>, 1, nullptr, output_ch);
}

}  // extern "C"
