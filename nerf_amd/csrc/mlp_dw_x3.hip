// Weight gradients of the fused MLP, dW_l = dY_l X_l^T (a reduction over sample points), with fp32 accuracy on the
// bf16 matrix instruction: the third stage of the parameter-gradient pass (autograd of run_nerf_helpers.py:109-133),
// on the split-operand scheme of mlp_fwd_bf16x3.hip.
//
// Both operands live in HBM as fp32 in the blocked layout of mlp_layout.hpp ([128-point tile][row / 4][point][row % 4]).
// A workgroup (8 waves, two per SIMD) owns one task -- a weight matrix, or matrices that share an operand -- for one
// slice of the point range and walks it in stages of 16 points (one k-step of v_mfma_f32_32x32x16_bf16):
//   * staging: a thread loads 16 bytes = four consecutive rows of one point (128-byte runs per row group and
//     instruction), splits the four values EXACTLY into three bf16 pieces (mlp_x3.hpp) and writes 8 bytes per piece
//     into an LDS image [piece][point][row]; the point stride is 64 bytes past a multiple of 256 and the 8-byte chunks
//     of a 64-byte group are permuted by the point index, which makes both these writes and the reads below free of
//     bank conflicts;
//   * fragments: ds_read_b64_tr_b16 reads a 4-point x 16-row block of the image transposed, i.e. as four k-slots
//     (points) of 16 operand rows; two reads = the 8 k-slots of a lane; six matrix instructions per tile pair rebuild
//     the product from the piece products hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid;
//   * two register sets hold the loads of the next two stages (64 KB per compute unit in flight: with one stage in
//     flight the kernel was bound by memory latency at 3 TB/s); a set is split behind the matrix work of the stage
//     before its own and re-requested at once; two LDS images alternate, one barrier per stage.
// Measured on 1.57 M points (tools/time_dw.py, tools/power_probe.sh): 12.3 ms at a 1.67 GHz shader clock; without the HBM
// loads 7.7 ms at 2.03 GHz (84 % of the matrix pipe's cycles busy); loads and split alone 8.0 ms at 2.19 GHz (3.9 TB/s):
// the board's power limit, not latency or issue slots, bounds the kernel -- the two parts add up instead of overlapping,
// whatever the prefetch depth (one or two stages in registers, or a seven-deep LDS-DMA ring: 12.3-13.5 ms).
// Variants that were measured and not kept (same 1.57 M points): a wave's load covering 256-byte instead of 128-byte runs
// (loads and split alone read 3.42 TB/s either way); three register sets = three stages in flight (loads and split alone
// go from 3.9 to 4.4 TB/s, but the 256 x 256 task already holds 248 of its 256 registers and spills 50-88 of them:
// 18.3 ms); a third stage in flight through an LDS slot filled by LDS-DMA (12.7 ms: a deeper queue of loads is not
// what the full kernel lacks); row-tile fragments not read ahead (12.4 ms); temporal instead of non-temporal loads (12.5 ms).
// Bias gradients are fp32 row sums of dY taken in the staging path.  Partial results go to per-slice slabs that a last
// kernel sums in fixed order (bitwise reproducible, no float atomics).
//
// The same kernel in a TWO-piece form (k_dw_x3<true>) on v_mfma_f32_32x32x16_f16, three matrix instructions per tile
// pair and stage into the SAME single accumulator (a second one, as the forward's two-piece scheme keeps, would be the
// whole register file for a 256 x 256 tile).  The gradient operand is scaled by ONE power of two per group of dY rows
// and batch -- taken from the maxima the two-piece backward-data kernel leaves per tile, so that the group's largest
// magnitude lies in [2^13, 2^14) -- and split into hi = fp16(x), the unscaled residual fp16(x - hi), and hi * 2^-11; the
// activation operand into hi = fp16(x) and lo = fp16((x - hi) * 2^11); the product is hi.hi + residual.hi +
// (hi * 2^-11).lo.  Every activation keeps 22 bits of ITS OWN magnitude (the scaled low piece; activations have no
// common scale: a unit may stay at 1e-5 for a whole batch).  The two extra gradient-side pieces are exact to 2^-25
// absolute = 2^-39 of the group's largest gradient in the batch: full relative precision over 17 octaves below that
// largest value, and below it an error that any sum with a term near the top does not see (a unit whose gradients
// stay a million times below its layer's largest for the whole batch keeps about five digits).  Any tile marked
// by the two-piece forward or backward-data kernels sends the whole batch to the three-piece form (decided on the
// device: both forms are launched, one of them returns at once).
#include "common.hpp"
#include "mlp_core.hpp"
#include "mlp_x3.hpp"
#include "mlp_x2.hpp"
#include "mlp_dw.hpp"

using namespace nerfhip;
using namespace nerflayout;
using namespace nerfmlp;
using namespace nerfx3;

namespace {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int DWX_THREADS = 512;
constexpr int SP = 16;                 // points per stage
constexpr int MAX_LOADS = 5;
constexpr int NSETS = 2;
constexpr bool RAW_SLOT = false;
static_assert(NSETS == 2 || NSETS == 3, "the stage loop is written for two or three sets");

// What a kind of task looks like to a workgroup.  The LDS image holds 128 * NL operand rows per point: the dY rows
// first (A_TILES tiles of 32), then the input rows; a wave multiplies TM row tiles by TN column tiles (32 x 32 each).
template <int KIND> struct DwKind;
// NGRAD: the first NGRAD loads are dY rows, the others saved activations.
template <> struct DwKind<0> { static constexpr int NL = 4, A_TILES = 8, TM = 4, TN = 2, NBIAS = 2, NGRAD = 2; };    // hidden layer 256 x 256
template <> struct DwKind<1> { static constexpr int NL = 5, A_TILES = 16, TM = 2, TN = 2, NBIAS = 2, NGRAD = 4; };   // dY0 ++ dY5 x encoded position
template <> struct DwKind<2> { static constexpr int NL = 4, A_TILES = 4, TM = 1, TN = 5, NBIAS = 1, NGRAD = 1; };    // views layer x (feature ++ direction)
template <> struct DwKind<3> { static constexpr int NL = 3, A_TILES = 4, TM = 1, TN = 1, NBIAS = 1, NGRAD = 1; };    // heads: <= 8 rows

template <int NL> struct Image {
    static constexpr int ROWS = 128 * NL;
    static constexpr int PS = 2 * ROWS + 64;      // bytes per point and piece
    static constexpr int PL = SP * PS;            // bytes per piece
    static constexpr int STAGE = 3 * PL;
};
constexpr size_t DWX_RAW_BYTES = 4 * 8 * 1024;        // the LDS-DMA slot: one KiB per load and wave
constexpr size_t DWX_LDS_BYTES = !RAW_SLOT || 2 * (size_t)Image<MAX_LOADS>::STAGE > 2 * (size_t)Image<4>::STAGE + DWX_RAW_BYTES
                                     ? 2 * (size_t)Image<MAX_LOADS>::STAGE : 2 * (size_t)Image<4>::STAGE + DWX_RAW_BYTES;
static_assert(DWX_LDS_BYTES <= 160 * 1024, "LDS of one compute unit");

struct DwxTask {
    int kind;
    int ld_src[MAX_LOADS];     // per load of 32 row groups: 0 = gradient workspace, 1 = saved activations
    int ld_grp[MAX_LOADS];     // first row group
    int ld_n[MAX_LOADS];       // valid row groups (the others repeat the first: nothing reads them)
    int ld_stage[MAX_LOADS];   // dY loads: which group of dY rows (mlp_dw.hpp), i.e. which scale of the two-piece form
    int M, N;                  // valid rows / columns of the product(s)
    int ldc, ldc2;             // row strides in the flat parameter vector
    int64_t out, out2;         // W[0][0] of the product, and of the second one (kind 1: layer 5's encoded-input columns)
    int64_t bias;              // bias of the first dY run, or -1
    int first, nsplit;         // workgroups [first, first + nsplit) slice the point range
};
struct DwxTasks {
    DwxTask t[14];
    int n;
};

__device__ __forceinline__ bf16x4 tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
}
__device__ __forceinline__ bf16x8 cat8(bf16x4 a, bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32h(bf16x8 a, bf16x8 b, f32x16 c) {     // the same 16 bytes read as eight fp16
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(nerfx2::f16x8, a), __builtin_bit_cast(nerfx2::f16x8, b), c, 0, 0, 0);
}
// two values -> hi = fp16 pair, lo = the pair of UNSCALED residuals fp16(v - hi) (one mixed fused multiply-add each)
__device__ __forceinline__ void split2u_pair(float v0, float v1, uint32_t& hi, uint32_t& lo) {
    const nerfx2::f16x2 h = {(_Float16)v0, (_Float16)v1};
    hi = __builtin_bit_cast(uint32_t, h);
    const float c = -1.0f;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(lo) : "v"(hi), "v"(c), "v"(v0), "v"(v1));
}
// position of 8-byte chunk `c` (four rows) inside its 64-byte group for point `pt`
__device__ __forceinline__ int swz(int c, int pt) { return c ^ (((pt >> 1) & 3) << 1); }

template <int KIND, bool TWO>
__device__ __forceinline__ void dwx_loop(const DwxTask& task, int split, int64_t nchunks, int64_t tile_floats,
                                         const float* __restrict__ act, const float* __restrict__ scr,
                                         float* __restrict__ slab, char* smem, const uint32_t* __restrict__ ctl) {
    using K = DwKind<KIND>;
    using Im = Image<K::NL>;
    constexpr int NL = K::NL, TM = K::TM, TN = K::TN, NBIAS = K::NBIAS, NGRAD = K::NGRAD;
    constexpr int NPB = TWO ? 2 : 3;                 // pieces of an activation fragment (a dY fragment has 3 either way)
    // two-piece form: the power of two each dY load is multiplied by, and the one this wave's block is divided by
    // (a head task, KIND 3: its rows are channels of dL/d(raw), each with a scale of its own, stages ld_stage[0] + row)
    float up[NGRAD];
    float up_row[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    float out_scale = 1.0f;
    if constexpr (TWO) {
#pragma unroll
        for (int k = 0; k < NGRAD; ++k) up[k] = __uint_as_float(ctl[1 + task.ld_stage[k]] << 23);
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;

    // a slice is an even number of stages (the stage loop is unrolled by two; nchunks is a multiple of eight)
    const int64_t per = ((nchunks + task.nsplit - 1) / task.nsplit + 1) & ~(int64_t)1;
    const int64_t c0 = split * per, c1 = (c0 + per < nchunks) ? c0 + per : nchunks;

    // ---- which 32 x 32 tiles this wave multiplies, and where its block of the product goes
    int a_tile0, b_tile0, row0, col0, ldc = task.ldc;
    int64_t out = task.out;
    bool idle = false;
    if constexpr (TWO) out_scale = __uint_as_float((254u - ctl[1 + task.ld_stage[(KIND == 1 && wave >= 4) ? 2 : 0]]) << 23);
    if constexpr (KIND == 0) {
        a_tile0 = 4 * (wave >> 2); b_tile0 = 2 * (wave & 3); row0 = 32 * a_tile0; col0 = 32 * b_tile0;
    } else if constexpr (KIND == 1) {
        a_tile0 = 2 * wave; b_tile0 = 0; row0 = 64 * (wave & 3); col0 = 0;
        if (wave >= 4) { out = task.out2; ldc = task.ldc2; }
    } else if constexpr (KIND == 2) {
        a_tile0 = wave >> 1; b_tile0 = 5 * (wave & 1); row0 = 32 * a_tile0; col0 = 32 * b_tile0;
    } else {
        a_tile0 = 0; b_tile0 = wave; row0 = 0; col0 = 32 * wave;
        idle = col0 >= task.N;
    }

    // ---- staging map: thread -> (row group 32 k + rgi, point pt) of every load k
    const int pt = 8 * (wave & 1) + (lane & 7), rgi = 8 * (wave >> 1) + (lane >> 3);
    const float* gsrc[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const float* base = task.ld_src[k] ? act : scr;
        gsrc[k] = base + (int64_t)(task.ld_grp[k] + (rgi < task.ld_n[k] ? rgi : 0)) * GROUP_FLOATS + pt * 4;
    }
    const int wr_off = pt * Im::PS + 8 * swz(rgi, pt);             // + piece * PL + 256 k
    if constexpr (TWO && KIND == 3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) up_row[e] = __uint_as_float(ctl[1 + task.ld_stage[0] + (rgi < 2 ? 4 * rgi + e : 0)] << 23);
    }
    // ---- fragment reads: in read r (0, 1) lane 4q+p of group g addresses point 8 (g >> 1) + 4 r + q, rows 4p..4p+3 of
    // the half (g & 1) of a 32-row tile; the lane ends up with rows (lane & 31), k-slots 8 (lane >> 5) + 0..7
    int rd[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int p_ = 8 * (g >> 1) + 4 * r + ((lane >> 2) & 3);
        rd[r] = p_ * Im::PS + 8 * swz(4 * (g & 1) + (lane & 3), p_);
    }
    const int off_a = 64 * a_tile0, off_b = 64 * (K::A_TILES + b_tile0);
    auto frag = [&](const char* img, int off) { return cat8(tr16(img + rd[0] + off), tr16(img + rd[1] + off)); };

    // ---- loads: two register sets, the stages after next; set (s - c0) & 1 holds stage s
    constexpr bool RAW = RAW_SLOT && NL <= 4;             // set 2 = the LDS slot behind the two images
    f32x4 stage[RAW ? 2 : NSETS][NL];
    char* const raw_mine = smem + 2 * Im::STAGE + wave * 1024;   // + 8 KiB per load; the lane's 16 bytes follow by themselves
    auto request = [&]<int SET, int k>(int64_t chunk) {
        const int64_t ch = chunk < c1 ? chunk : c1 - 1;          // past the slice: a repeat that nothing uses
        const int64_t off = (ch >> 3) * tile_floats + (ch & 7) * (SP * 4);
        if constexpr (RAW && SET == 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this lane's read of the slot (commit) has returned
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[k] + off),
                                             (__attribute__((address_space(3))) void*)(raw_mine + k * 8192), 16, 0, 0);
            return;
        }
        stage[SET][k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gsrc[k] + off));
    };
    float bsum[NBIAS][4];
#pragma unroll
    for (int k = 0; k < NBIAS; ++k) bsum[k][0] = bsum[k][1] = bsum[k][2] = bsum[k][3] = 0.0f;
    const bool want_bias = task.bias >= 0;
    // split the four rows of load k of stage `chunk` into image `img`; count them for the bias if the stage is inside
    // the slice
    auto commit = [&]<int SET, int k>(char* img, int64_t chunk) {
        f32x4 v;
        if constexpr (RAW && SET == 2) {
            // issued after this load: the rest of its stage, two stages of register loads, this stage's first k requests
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NL - 1) : "memory");
            v = *reinterpret_cast<const f32x4*>(raw_mine + k * 8192 + lane * 16);
        } else {
            v = stage[SET][k];
        }
        char* dst = img + wr_off + 256 * k;
        if constexpr (TWO) {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            u32x2 p0, p1;
            if constexpr (k < NGRAD) {       // dY rows, scaled: hi | unscaled residual (hi * 2^-11 is made from hi after the read)
                uint32_t h0, l0, h1, l1;
                if constexpr (KIND == 3) {
                    split2u_pair(v[0] * up_row[0], v[1] * up_row[1], h0, l0);
                    split2u_pair(v[2] * up_row[2], v[3] * up_row[3], h1, l1);
                } else {
                    split2u_pair(v[0] * up[k], v[1] * up[k], h0, l0);
                    split2u_pair(v[2] * up[k], v[3] * up[k], h1, l1);
                }
                p0 = u32x2{h0, h1}; p1 = u32x2{l0, l1};
            } else {                         // activations: hi | scaled residual
                uint32_t h0, l0, h1, l1;
                nerfx2::split2_pair(v[0], v[1], h0, l0);
                nerfx2::split2_pair(v[2], v[3], h1, l1);
                p0 = u32x2{h0, h1}; p1 = u32x2{l0, l1};
            }
            *reinterpret_cast<u32x2*>(dst) = p0;
            *reinterpret_cast<u32x2*>(dst + Im::PL) = p1;
        } else {
            bf16x4 h, m, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __bf16 a, b, c;
                split3(v[e], a, b, c);
                h[e] = a; m[e] = b; l[e] = c;
            }
            *reinterpret_cast<bf16x4*>(dst) = h;
            *reinterpret_cast<bf16x4*>(dst + Im::PL) = m;
            *reinterpret_cast<bf16x4*>(dst + 2 * Im::PL) = l;
        }
        if constexpr (k < NBIAS) {   // (no branch: the compiler's load bookkeeping stays exact in straight-line code)
            const bool live = chunk < c1;   // (a select, not a product with 0: an infinite dY of a repeated chunk must not turn into NaN)
#pragma unroll
            for (int e = 0; e < 4; ++e) bsum[k][e] += live ? v[e] : 0.0f;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    if (c0 < c1) {
        static_for<NSETS>([&]<int s_>() {
            static_for<NL>([&]<int k>() { request.template operator()<s_, k>(c0 + s_); });
        });
        static_for<NL>([&]<int k>() {
            commit.template operator()<0, k>(smem, c0);
            request.template operator()<0, k>(c0 + NSETS);
        });
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // one stage; P = (c - c0) & 1 = the image it multiplies from; S = (c - c0 + 1) % NSETS = the register set that holds
    // stage c + 1, which is split into the other image here and re-requested for stage c + 1 + NSETS
    auto run_stage = [&]<int P, int S>(int64_t c) {
        const char* cur = smem + P * Im::STAGE;
        char* nxt = smem + (P ^ 1) * Im::STAGE;
        bf16x8 B[TN][NPB];                               // the column fragments of this stage, piece by piece
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int s = 0; s < NPB; ++s) B[j][s] = frag(cur + s * Im::PL, off_b + 64 * j);
        constexpr int ABUF = 2;                          // a row tile's fragments, read one tile ahead
        bf16x8 A[ABUF][3];
        auto read_a = [&]<int i>() {
#pragma unroll
            for (int s = 0; s < (TWO ? 2 : 3); ++s) A[i % ABUF][s] = frag(cur + s * Im::PL, off_a + 64 * i);
        };
        // two pieces: the third gradient-side fragment is the leading one times 2^-11 (four packed multiplications
        // instead of an LDS plane written and read)
        auto third = [&]<int i>() {
            if constexpr (TWO) {
                const nerfx2::f16x8 h = __builtin_bit_cast(nerfx2::f16x8, A[i % ABUF][0]);
                A[i % ABUF][2] = __builtin_bit_cast(bf16x8, h * (_Float16)nerfx2::LO_UNSCALE);
            }
        };
        if constexpr (ABUF == 2) read_a.template operator()<0>();
        static_for<TM>([&]<int i>() {
            if constexpr (ABUF == 1) read_a.template operator()<i>();
            else if constexpr (i + 1 < TM) read_a.template operator()<i + 1>();
            __builtin_amdgcn_sched_barrier(0);
            third.template operator()<i>();
            // (idle waves of a head task multiply too: what they read is finite, and they store nothing)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (TWO) {                     // (hi * 2^-11).lo, residual.hi, hi.hi
                    acc[i][j] = mfma32h(A[i % ABUF][2], B[j][1], acc[i][j]);
                    acc[i][j] = mfma32h(A[i % ABUF][1], B[j][0], acc[i][j]);
                    acc[i][j] = mfma32h(A[i % ABUF][0], B[j][0], acc[i][j]);
                } else {
#pragma unroll
                    for (int q = 0; q < 6; ++q)          // (dY piece, input piece), smallest products first
                        acc[i][j] = mfma32(A[i % ABUF][PROD_W[q]], B[j][PROD_X[q]], acc[i][j]);
                }
            }
            // staging of stage c + 1 behind the matrix work: load k is split when row tile group(k) multiplies, and
            // its registers are re-requested for stage c + 3
            static_for<NL>([&]<int k>() {
                if constexpr (((k + 1) * TM - 1) / NL == i) {
                    commit.template operator()<S, k>(nxt, c + 1);
                    request.template operator()<S, k>(c + 1 + NSETS);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    if constexpr (NSETS == 2) {
        for (int64_t c = c0; c < c1; c += 2) {
            run_stage.template operator()<0, 1>(c);
            run_stage.template operator()<1, 0>(c + 1);
        }
    } else {
        int64_t c = c0;
        for (; c + 6 <= c1; c += 6) {
            run_stage.template operator()<0, 1>(c);
            run_stage.template operator()<1, 2>(c + 1);
            run_stage.template operator()<0, 0>(c + 2);
            run_stage.template operator()<1, 1>(c + 3);
            run_stage.template operator()<0, 2>(c + 4);
            run_stage.template operator()<1, 0>(c + 5);
        }
        if (c + 2 <= c1) {
            run_stage.template operator()<0, 1>(c);
            run_stage.template operator()<1, 2>(c + 1);
        }
        if (c + 4 <= c1) {
            run_stage.template operator()<0, 0>(c + 2);
            run_stage.template operator()<1, 1>(c + 3);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- this slice's block of the product -> its slab
    if (!idle) {
        float* o = slab + out;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = col0 + 32 * j + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < task.M && col < task.N) {
                        float val = acc[i][j][r];
                        if constexpr (TWO && KIND == 3) val *= __uint_as_float((254u - ctl[1 + task.ld_stage[0] + row]) << 23);
                        else if constexpr (TWO) val *= out_scale;
                        o[(int64_t)row * ldc + col] = val;
                    }
                }
            }
    }
    // ---- bias gradients: per-thread row sums -> LDS [row][point] -> one thread per row adds the 16 points in order
    if (want_bias) {
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int k = 0; k < NBIAS; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(4 * (32 * k + rgi) + e) * SP + pt] = bsum[k][e];
        __syncthreads();
        if (tid < 128 * NBIAS && tid < task.M) {
            float s = 0.0f;
#pragma unroll
            for (int q = 0; q < SP; ++q) s += red[tid * SP + q];
            slab[task.bias + tid] = s;
        }
    }
}

// Second half, at the start of a weight-gradient workgroup: out[0] = any tile marked; out[1 + s] = the power of two
// that takes stage s's largest magnitude below 2^14, as a biased exponent (the float 2^k has the bits word << 23).
__device__ __forceinline__ void control_words(const uint32_t* __restrict__ ctl, uint32_t* out) {
    const int t = threadIdx.x;
    if (t <= nerfdw::DY_STAGES) {
        uint32_t m = 0u;
        for (int b = 0; b < nerfdw::DW_CTL_PARTS; ++b) {
            const uint32_t v = ctl[b * nerfdw::DW_CTL_STRIDE + t];
            m = v > m ? v : m;
        }
        if (t > 0) {
            // the group's values are below 2^(e - 127): times 2^(14 + 127 - e) they are below 2^14
            int word = m == 0u ? 127 : 127 + 14 + 127 - (int)m;
            word = word < 2 ? 2 : (word > 252 ? 252 : word);     // (2^k and 2^-k both normal floats)
            m = (uint32_t)word;
        }
        out[t] = m;
    }
    __syncthreads();
}

// ctl (mlp_dw.hpp): the two-piece form runs only while no tile is marked and takes its scales from there; the
// three-piece form runs always (ctl == nullptr) or only when a tile is marked
template <bool TWO>
__global__ void __launch_bounds__(DWX_THREADS, 2)
k_dw_x3(DwxTasks tasks, int64_t nchunks, int64_t tile_floats, const float* __restrict__ act, const float* __restrict__ scr,
        float* __restrict__ slabs, int64_t slab_stride, const uint32_t* __restrict__ ctl) {
    __shared__ uint32_t ctl_s[nerfdw::DY_STAGES + 1];
    if (TWO || ctl != nullptr) {
        control_words(ctl, ctl_s);
        if (TWO ? ctl_s[0] != 0u : ctl_s[0] == 0u) return;
    }
    extern __shared__ __attribute__((aligned(16))) char smem_dw[];
    int ti = 0;
    for (int i = 1; i < tasks.n; ++i)
        if ((int)blockIdx.x >= tasks.t[i].first) ti = i;
    const DwxTask& task = tasks.t[ti];
    const int split = (int)blockIdx.x - task.first;
    float* slab = slabs + (int64_t)split * slab_stride;
    switch (task.kind) {
        case 0: dwx_loop<0, TWO>(task, split, nchunks, tile_floats, act, scr, slab, smem_dw, ctl_s); break;
        case 1: dwx_loop<1, TWO>(task, split, nchunks, tile_floats, act, scr, slab, smem_dw, ctl_s); break;
        case 2: dwx_loop<2, TWO>(task, split, nchunks, tile_floats, act, scr, slab, smem_dw, ctl_s); break;
        default: dwx_loop<3, TWO>(task, split, nchunks, tile_floats, act, scr, slab, smem_dw, ctl_s); break;
    }
}

// First half of the control words of the two-piece form (mlp_dw.hpp): DW_CTL_PARTS partial results, one per one-wave
// workgroup, over the per-tile words the two-piece stages left behind -- word 0: any tile marked; word 1 + s: the
// largest exponent word of stage s.  (The weight-gradient workgroups finish the reduction themselves, control_words()
// below: one workgroup scanning all tiles took 40-57 us.)
__global__ void __launch_bounds__(64) k_dw_scales(int64_t n_tiles, const uint32_t* __restrict__ fwd_flags,
                                                   const uint32_t* __restrict__ bwd_flags, const uint32_t* __restrict__ tile_max,
                                                   uint32_t* __restrict__ ctl) {
    uint32_t any = 0u, mx[nerfdw::DY_STAGES];
#pragma unroll
    for (int s = 0; s < nerfdw::DY_STAGES; ++s) mx[s] = 0u;
    for (int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x; t < n_tiles; t += (int64_t)gridDim.x * 64) {
        any |= bwd_flags[t] | (fwd_flags ? fwd_flags[t] : 0u);
#pragma unroll
        for (int s = 0; s < nerfdw::DY_STAGES; ++s) {
            const uint32_t e = tile_max[t * nerfdw::DY_STAGES + s];
            mx[s] = e > mx[s] ? e : mx[s];
        }
    }
    auto wave_max = [](uint32_t e) -> uint32_t {            // e < 512; bisection over ballots
        uint32_t best = 0u;
#pragma unroll
        for (int bit = 8; bit >= 0; --bit) {
            const uint32_t cand = best | (1u << bit);
            if (__ballot(e >= cand) != 0ull) best = cand;
        }
        return best;
    };
    uint32_t* mine = ctl + (int64_t)blockIdx.x * nerfdw::DW_CTL_STRIDE;
    const uint32_t flag = __ballot(any != 0u) != 0ull ? 1u : 0u;
    if (threadIdx.x == 0) mine[0] = flag;
#pragma unroll
    for (int s = 0; s < nerfdw::DY_STAGES; ++s) {
        const uint32_t m = wave_max(mx[s] > 511u ? 511u : mx[s]);
        if (threadIdx.x == 0) mine[1 + s] = m;
    }
}

}  // namespace

namespace nerfdw {

// dW = dY X^T over all layers in one launch on the split-operand kernel, then the fixed-order sum of the partial slabs.
// ctl == nullptr: three pieces.  Otherwise both forms are launched and the control words decide on the device which
// of them does the work (the other's workgroups return at once); they write the same slabs.
static int run_dw_split(int64_t ldp, int use_viewdirs, int output_ch, const float* act, float* scratch, float* slabs, int64_t slab,
                        float* d_params, const uint32_t* ctl, hipStream_t stream) {
    const ParamOffsets po = param_offsets(use_viewdirs, output_ch);
    const int SCR_HEAD = scr_head(use_viewdirs), SCR_HEAD2 = SCR_HEAD + 32;
    DwxTasks tk{};
    int cost[14];
    int nt = 0;
    // (stage: for dY loads, the group of dY rows as mlp_dw.hpp numbers them)
    auto load = [](DwxTask& t, int k, int src, int row, int rows, int stage = 0) {
        t.ld_src[k] = src; t.ld_grp[k] = row / 4; t.ld_n[k] = rows / 4 > 32 ? 32 : rows / 4; t.ld_stage[k] = stage;
    };
    auto big = [&](int a_row, int b_row, int ldc, int64_t out, int64_t bias, int stage) {
        DwxTask t{};
        t.kind = 0;
        load(t, 0, 0, a_row, 128, stage); load(t, 1, 0, a_row + 128, 128, stage); load(t, 2, 1, b_row, 128); load(t, 3, 1, b_row + 128, 128);
        t.M = 256; t.N = 256; t.ldc = ldc; t.out = out; t.bias = bias;
        cost[nt] = 16; tk.t[nt++] = t;
    };
    for (int l = 1; l < 8; ++l)
        big(SCR_H0 + 256 * l, ACT_H0 + 256 * (l - 1), (l == 5) ? W + IN_PTS : W, po.pts_w[l] + (l == 5 ? IN_PTS : 0), po.pts_b[l], dy_stage_layer(l));
    if (use_viewdirs) big(SCR_FEAT, ACT_H0 + 256 * 7, W, po.feat_w, po.feat_b, DY_STAGE_FEAT);
    {   // encoded positions feed layers 0 and 5
        DwxTask t{};
        t.kind = 1;
        load(t, 0, 0, SCR_H0, 128, dy_stage_layer(0)); load(t, 1, 0, SCR_H0 + 128, 128, dy_stage_layer(0));
        load(t, 2, 0, SCR_H0 + 256 * 5, 128, dy_stage_layer(5)); load(t, 3, 0, SCR_H0 + 256 * 5 + 128, 128, dy_stage_layer(5));
        load(t, 4, 1, ACT_ENC_PTS, 64);
        t.M = 256; t.N = IN_PTS; t.ldc = IN_PTS; t.ldc2 = W + IN_PTS; t.out = po.pts_w[0]; t.out2 = po.pts_w[5]; t.bias = po.pts_b[0];
        cost[nt] = 13; tk.t[nt++] = t;
    }
    if (use_viewdirs) {
        DwxTask t{};   // views layer: dY x (feature ++ encoded direction); the weight's 283 columns are contiguous
        t.kind = 2;
        load(t, 0, 0, SCR_VIEWS, 128, DY_STAGE_VIEWS); load(t, 1, 1, ACT_FEAT, 128); load(t, 2, 1, ACT_FEAT + 128, 128); load(t, 3, 1, ACT_ENC_DIRS, 32);
        t.M = 128; t.N = W + IN_DIRS; t.ldc = po.views_in; t.out = po.views_w; t.bias = po.views_b;
        cost[nt] = 12; tk.t[nt++] = t;
        DwxTask a{};   // alpha: dL/d(sigma) x h7
        a.kind = 3;
        load(a, 0, 0, SCR_HEAD2, 8, 4); load(a, 1, 1, ACT_H0 + 256 * 7, 128); load(a, 2, 1, ACT_H0 + 256 * 7 + 128, 128);   // (density: channel 4)
        a.M = 1; a.N = 256; a.ldc = W; a.out = po.alpha_w; a.bias = po.alpha_b;
        cost[nt] = 6; tk.t[nt++] = a;
        DwxTask c{};   // rgb: dL/d(rgb) x views hidden
        c.kind = 3;
        load(c, 0, 0, SCR_HEAD, 8); load(c, 1, 1, ACT_VIEWS, 128); load(c, 2, 1, ACT_VIEWS, 0);
        c.M = 3; c.N = 128; c.ldc = W / 2; c.out = po.rgb_w; c.bias = po.rgb_b;
        cost[nt] = 4; tk.t[nt++] = c;
    } else {
        DwxTask a{};   // output_linear: dL/d(out) x h7
        a.kind = 3;
        load(a, 0, 0, SCR_HEAD, 8); load(a, 1, 1, ACT_H0 + 256 * 7, 128); load(a, 2, 1, ACT_H0 + 256 * 7 + 128, 128);
        a.M = output_ch; a.N = 256; a.ldc = W; a.out = po.out_w; a.bias = po.out_b;
        cost[nt] = 6; tk.t[nt++] = a;
    }
    tk.n = nt;
    int n_cu = compute_units();
    if (n_cu < nt) n_cu = nt;
    const int64_t nchunks = ldp / SP;
    // slices per task: one more slice to the task whose slices are longest until the launch is one workgroup per CU
    const int cap = max_slabs(ldp);
    int used = 0;
    for (int i = 0; i < nt; ++i) { tk.t[i].nsplit = 1; ++used; }
    while (used < n_cu) {
        int best = -1;
        double worst = 0.0;
        for (int i = 0; i < nt; ++i) {
            if (tk.t[i].nsplit >= cap) continue;
            const double per = (double)cost[i] / tk.t[i].nsplit;
            if (per > worst) { worst = per; best = i; }
        }
        if (best < 0) break;
        ++tk.t[best].nsplit;
        ++used;
    }
    int n_wg = 0;
    for (int i = 0; i < nt; ++i) {
        tk.t[i].first = n_wg;
        n_wg += tk.t[i].nsplit;
    }
    // what the reduction reads: per task its product block(s) and bias run, in the slabs of its own slices only
    SlabRegions regions{};
    int64_t covered = 0;
    auto region = [&](int64_t out, int rows, int cols, int ld, int nsplit) {
        regions.r[regions.n++] = SlabRegion{out, rows, cols, ld, nsplit};
        covered += (int64_t)rows * cols;
    };
    for (int i = 0; i < nt; ++i) {
        const DwxTask& t = tk.t[i];
        region(t.out, t.M, t.N, t.ldc, t.nsplit);
        if (t.kind == 1) region(t.out2, t.M, t.N, t.ldc2, t.nsplit);
        if (t.bias >= 0) region(t.bias, 1, t.M, t.M, t.nsplit);
    }
    hipError_t e = hipSuccess;
    if (covered != po.total) {   // parameters no task covers (a module's unused layers) receive a zero gradient
        e = hipMemsetAsync(d_params, 0, (size_t)po.total * sizeof(float), stream);
        if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_bwd: memset: %s", hipGetErrorString(e));
    }
    const size_t lds = DWX_LDS_BYTES;
    e = lds_limit_once(reinterpret_cast<const void*>(k_dw_x3<false>), lds);
    if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_bwd: LDS attribute: %s", hipGetErrorString(e));
    if (ctl != nullptr) {
        e = lds_limit_once(reinterpret_cast<const void*>(k_dw_x3<true>), lds);
        if (e != hipSuccess) return fail(NERF_E_LAUNCH, "nerf_mlp_bwd: LDS attribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(k_dw_x3<true>, dim3(n_wg), dim3(DWX_THREADS), lds, stream, tk, nchunks,
                           (int64_t)act_rows(use_viewdirs) * TILE_PTS, act, scratch, slabs, slab, ctl);
        int rc2 = launched("nerf_mlp_bwd (dW, two pieces)");
        if (rc2) return rc2;
    }
    hipLaunchKernelGGL(k_dw_x3<false>, dim3(n_wg), dim3(DWX_THREADS), lds, stream, tk, nchunks,
                       (int64_t)act_rows(use_viewdirs) * TILE_PTS, act, scratch, slabs, slab, ctl);
    int rc = launched("nerf_mlp_bwd (dW, split operands)");
    if (rc) return rc;
    return reduce_regions(regions, slabs, slab, d_params, stream);
}

int run_dw_x3(int64_t ldp, int use_viewdirs, int output_ch, const float* act, float* scratch, float* slabs, int64_t slab,
              float* d_params, hipStream_t stream) {
    return run_dw_split(ldp, use_viewdirs, output_ch, act, scratch, slabs, slab, d_params, nullptr, stream);
}

int run_dw_x2(int64_t ldp, int use_viewdirs, int output_ch, const float* act, float* scratch, float* slabs, int64_t slab,
              float* d_params, const uint32_t* fwd_flags, const uint32_t* bwd_flags, const uint32_t* tile_max, uint32_t* ctl,
              hipStream_t stream) {
    hipLaunchKernelGGL(k_dw_scales, dim3(DW_CTL_PARTS), dim3(64), 0, stream, ldp / TILE_PTS, fwd_flags, bwd_flags, tile_max, ctl);
    int rc = launched("nerf_mlp_bwd (dW scales)");
    if (rc) return rc;
    return run_dw_split(ldp, use_viewdirs, output_ch, act, scratch, slabs, slab, d_params, ctl, stream);
}

}  // namespace nerfdw
