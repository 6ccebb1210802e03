// How fast is the Infinity Cache?  A 16-B-per-lane copy (and a read-only sum) over a working set of W MiB, repeated: below ~100 MiB the
// set stays in the 256-MiB cache between repetitions, far above it every byte comes from / goes to HBM.  If the cache-resident rate is
// well above the HBM rate, a two-pass transform whose passes hand over through the cache is not bound by the 6.3 TB/s of a streaming copy.
//   hipcc -O3 --offload-arch=gfx950 mall_bw.hip -o mall_bw && ./mall_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32;
typedef u32 __attribute__((ext_vector_type(4))) u32x4;
typedef __attribute__((address_space(1))) const u32x4* gptr4c;
typedef __attribute__((address_space(1))) u32x4* gptr4;

__global__ __launch_bounds__(256) void copy_k(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n4) {
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) if (i + 256 * k < n4) v[k] = *((gptr4c)s + i + 256 * k);
#pragma unroll
    for (int k = 0; k < 4; k++) if (i + 256 * k < n4) *((gptr4)d + i + 256 * k) = v[k];
}
__global__ __launch_bounds__(256) void read_k(const u32x4* __restrict__ s, u32* __restrict__ out, size_t n4) {
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    u32 acc = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) if (i + 256 * k < n4) { u32x4 v = *((gptr4c)s + i + 256 * k); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) out[0] = acc;
}
__global__ __launch_bounds__(256) void write_k(u32x4* __restrict__ d, size_t n4, u32 x) {
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; k++) if (i + 256 * k < n4) *((gptr4)d + i + 256 * k) = u32x4{x, x, x, x};
}

int main() {
    const size_t MAXB = (size_t)2 << 30;
    u32x4 *a, *b; u32* o;
    hipMalloc(&a, MAXB); hipMalloc(&b, MAXB); hipMalloc(&o, 64);
    hipMemset(a, 1, MAXB); hipMemset(b, 2, MAXB);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int sizes[] = {4, 8, 16, 32, 48, 64, 96, 128, 192, 256, 512, 2048};
    for (int mb : sizes) {
        const size_t bytes = (size_t)mb << 20, n4 = bytes / 16;
        const unsigned grid = (unsigned)((n4 + 1023) / 1024);
        const int reps = mb <= 64 ? 200 : mb <= 512 ? 50 : 10;
        float ms[3];
        for (int mode = 0; mode < 3; mode++) {
            for (int w = 0; w < 3; w++) {
                if (mode == 0) hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, 0, a, b, n4);
                else if (mode == 1) hipLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, 0, a, o, n4);
                else hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, b, n4, 7u);
            }
            hipEventRecord(e0, 0);
            for (int r = 0; r < reps; r++) {
                if (mode == 0) hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, 0, a, b, n4);
                else if (mode == 1) hipLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, 0, a, o, n4);
                else hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, b, n4, 7u);
            }
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[mode], e0, e1);
            ms[mode] /= reps;
        }
        printf("{\"working_set_MiB\": %d, \"copy_us\": %.2f, \"copy_TBs_rw\": %.2f, \"read_us\": %.2f, \"read_TBs\": %.2f, \"write_us\": %.2f, \"write_TBs\": %.2f}\n", mb, ms[0] * 1e3,
               2.0 * bytes / ms[0] / 1e9, ms[1] * 1e3, (double)bytes / ms[1] / 1e9, ms[2] * 1e3, (double)bytes / ms[2] / 1e9);
    }
    return 0;
}
