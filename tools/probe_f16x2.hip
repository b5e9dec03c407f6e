// Probe: a 256-long fp32 dot product rebuilt from TWO fp16 pieces per operand -- hi = fp16(x), lo = fp16((x - hi) * 2^11),
// products hi.hi into one accumulator and hi.lo + lo.hi into a second one that is scaled by 2^-11 at the end: three
// v_mfma_f32_16x16x32_f16 per 32 terms -- against the fp32 matrix instruction and the six-product bf16 split, each compared
// with an fp64 evaluation, over operand magnitudes 1e-6 .. 1e4.  fp16 holds 11 significant bits, so two pieces keep 22 of
// an fp32's 24 and a normal range of 2^-14 .. 65504 for the leading piece.
//   hipcc -O2 --offload-arch=gfx950 tools/probe_f16x2.hip -o gpurun_out/probe_f16x2 && gpurun_out/probe_f16x2
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 256;

__device__ inline void split3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
    hi = (__bf16)x;
    const float r1 = x - (float)hi;
    mid = (__bf16)r1;
    lo = (__bf16)(r1 - (float)mid);
}

__device__ inline void split2(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)((x - (float)hi) * 2048.0f);
}

// A: [16][K] row-major, B: [K][16] row-major, C*: [16][16]
__global__ void __launch_bounds__(64) k_probe(const float* A, const float* B, float* C32, float* C6, float* C2, float* C2b) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    f32x4 acc32 = {0, 0, 0, 0}, acc6 = {0, 0, 0, 0}, main2 = {0, 0, 0, 0}, corr2 = {0, 0, 0, 0}, quad = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 4)
        acc32 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k0 + g], B[(k0 + g) * 16 + r], acc32, 0, 0, 0);
    for (int k0 = 0; k0 < K; k0 += 32) {
        bf16x8 a[3], b[3];
        f16x8 ah, al, bh, bl;
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + 8 * g + j;
            __bf16 h, m, lo;
            split3(A[r * K + k], h, m, lo);
            a[0][j] = h; a[1][j] = m; a[2][j] = lo;
            split3(B[k * 16 + r], h, m, lo);
            b[0][j] = h; b[1][j] = m; b[2][j] = lo;
            _Float16 fh, fl;
            split2(A[r * K + k], fh, fl);
            ah[j] = fh; al[j] = fl;
            split2(B[k * 16 + r], fh, fl);
            bh[j] = fh; bl[j] = fl;
        }
        acc6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc6, 0, 0, 0);
        acc6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc6, 0, 0, 0);
        acc6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc6, 0, 0, 0);
        acc6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc6, 0, 0, 0);
        acc6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc6, 0, 0, 0);
        acc6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc6, 0, 0, 0);
        main2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, main2, 0, 0, 0);
        corr2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, corr2, 0, 0, 0);
        corr2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, corr2, 0, 0, 0);
        quad = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bl, quad, 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) {
        const int o = (4 * g + i) * 16 + r;
        C32[o] = acc32[i];
        C6[o] = acc6[i];
        C2[o] = fmaf(corr2[i], 1.0f / 2048.0f, main2[i]);
        C2b[o] = fmaf(fmaf(quad[i], 1.0f / 2048.0f, corr2[i]), 1.0f / 2048.0f, main2[i]);
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main() {
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, 16 * K * 4));
    CK(hipMalloc(&dB, 16 * K * 4));
    CK(hipMalloc(&dC, 4 * 256 * 4));
    struct Case { const char* name; int kind; double sa, sb; };
    const Case cases[] = {
        {"uniform [-1,1] x [-1,1]", 0, 1, 1},
        {"relu-like x N(0,1/16)", 1, 1, 0.0625},
        {"relu-like 1e-6 x 1/16", 1, 1e-6, 0.0625},
        {"relu-like 1e-5 x 1/16", 1, 1e-5, 0.0625},
        {"relu-like 1e-4 x 1/16", 1, 1e-4, 0.0625},
        {"relu-like 1e-3 x 1/16", 1, 1e-3, 0.0625},
        {"relu-like 1e-2 x 1/16", 1, 1e-2, 0.0625},
        {"relu-like 1e2 x 1/16", 1, 1e2, 0.0625},
        {"relu-like 1e3 x 1/16", 1, 1e3, 0.0625},
        {"relu-like 1e4 x 1/16", 1, 1e4, 0.0625},
        {"relu-like 3e4 x 1/16", 1, 3e4, 0.0625},
        {"relu-like 1 x 1e-4", 1, 1, 1e-4},
        {"relu-like 1 x 30", 1, 1, 30},
        {"wide range 2^[-12,0] both", 2, 1, 1},
        {"wide range 2^[-20,0] both", 4, 1, 1},
        {"cancelling (sum near 0)", 3, 1, 1},
    };
    for (const Case& c : cases) {
        double e[4] = {0, 0, 0, 0}, rr[4] = {0, 0, 0, 0}, bias[4] = {0, 0, 0, 0};
        srand(1234 + c.kind);
        auto u = []() { return rand() / (double)RAND_MAX; };
        const int trials = 50;
        for (int t = 0; t < trials; ++t) {
            std::vector<float> A(16 * K), B(16 * K), C(4 * 256);
            for (int i = 0; i < 16 * K; ++i) {
                double a = 2 * u() - 1, b = 2 * u() - 1;
                if (c.kind == 1) a = u() < 0.5 ? 0.0 : fabs(a);
                if (c.kind == 2) { a *= exp2(-12 * u()); b *= exp2(-12 * u()); }
                if (c.kind == 4) { a *= exp2(-20 * u()); b *= exp2(-20 * u()); }
                A[i] = (float)(a * c.sa); B[i] = (float)(b * c.sb);
            }
            if (c.kind == 3)
                for (int r = 0; r < 16; ++r)
                    for (int k = 0; k < K / 2; ++k) {
                        A[r * K + K / 2 + k] = -A[r * K + k] * (1.0f + 1e-3f * (float)(2 * u() - 1));
                        for (int n = 0; n < 16; ++n) B[(K / 2 + k) * 16 + n] = B[k * 16 + n];
                    }
            CK(hipMemcpy(dA, A.data(), 16 * K * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(dB, B.data(), 16 * K * 4, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dC + 256, dC + 512, dC + 768);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(C.data(), dC, 4 * 256 * 4, hipMemcpyDeviceToHost));
            for (int r = 0; r < 16; ++r)
                for (int n = 0; n < 16; ++n) {
                    double ref = 0, mag = 0;
                    for (int k = 0; k < K; ++k) {
                        ref += (double)A[r * K + k] * (double)B[k * 16 + n];
                        mag += fabs((double)A[r * K + k] * (double)B[k * 16 + n]);
                    }
                    for (int v = 0; v < 4; ++v) {
                        const double d = (C[v * 256 + r * 16 + n] - ref) / mag;
                        e[v] = fmax(e[v], fabs(d));
                        rr[v] += d * d;
                        bias[v] += d;
                    }
                }
        }
        const double cnt = trials * 256.0;
        printf("%-28s err/sum|a||b|: fp32 mfma max %.2e rms %.2e | bf16x3(6) max %.2e rms %.2e | f16x2(3) max %.2e rms %.2e mean %+.1e | f16x2(4) max %.2e rms %.2e\n",
               c.name, e[0], sqrt(rr[0] / cnt), e[1], sqrt(rr[1] / cnt), e[2], sqrt(rr[2] / cnt), bias[2] / cnt, e[3], sqrt(rr[3] / cnt));
    }
    return 0;
}
