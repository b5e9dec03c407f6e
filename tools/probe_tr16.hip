// Which LDS element does each lane of ds_read_b64_tr_b16 receive?  Image: [row][col] of 16-bit values holding
// 256 * row + col, row stride STRIDE elements; lane 4q+p of every 16-lane group addresses row R0 + q, columns 4p..4p+3
// (group g reads rows R0 + 4g ..).  Prints, per lane, the four (row, col) pairs it received.
// Build: hipcc -O2 --offload-arch=gfx950 tools/probe_tr16.hip -o tools/bin/probe_tr16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int STRIDE = 64;
__global__ void k(int* out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * STRIDE];
    for (int i = threadIdx.x; i < 64 * STRIDE; i += 64) lds[i] = (short)(256 * (i / STRIDE) + (i % STRIDE));
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    auto ptr = (__attribute__((address_space(3))) s16x4*)(lds + (4 * g + q) * STRIDE + 4 * p);
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = a[i];
}
int main() {
    int* d; int h[256];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int i = 0; i < 4; ++i) printf(" (r%d,c%d)", h[l * 4 + i] >> 8, h[l * 4 + i] & 255);
        printf("\n");
    }
    return 0;
}
