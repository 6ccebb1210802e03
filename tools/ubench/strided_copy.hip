// What do short contiguous runs cost?  A 512-lane block copies one 2^13-word tile of a 2^22-word column, the tile being 2^13/R runs of
// R words, consecutive runs n/(2^13/R) words apart — R = 16 is the access pattern of lde_mid_kernel (layers [13,22) + 4 low bits),
// R = 8192 that of the FIRST passes.  Columns: `cols` distinct ones (cols * 16 MiB in + out), repeated `reps` times.
//   hipcc -O3 --offload-arch=gfx950 strided_copy.hip -o strided_copy && ./strided_copy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32;
constexpr int LOGN = 22, TILE = 13;
typedef u32 __attribute__((ext_vector_type(4))) u32x4;
typedef __attribute__((address_space(1))) const u32x4* gptr4c;
typedef __attribute__((address_space(1))) u32x4* gptr4;

// XCD = 1: neighbouring tiles (the two 64-B halves of a 128-B line when R = 16) run on the same XCD, as lde_mid_kernel maps them
template <int LOGR, int XCD> __global__ __launch_bounds__(512) void tile_copy(const u32* __restrict__ src, u32* __restrict__ dst, int cols) {
    constexpr u32 R = 1u << LOGR, ROWS = (1u << TILE) / R, STRIDE = (1u << LOGN) / ROWS;
    const u32 tiles = 1u << (LOGN - TILE);
    u32 t, c;
    if (XCD) { const u32 x = blockIdx.x & 7, y = blockIdx.x >> 3; c = y % cols; t = x * (tiles >> 3) + y / cols; }
    else { t = blockIdx.x % tiles; c = blockIdx.x / tiles; }
    const u32* s = src + ((size_t)c << LOGN) + (size_t)t * R;
    u32* d = dst + ((size_t)c << LOGN) + (size_t)t * R;
    u32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u32 e = (i * 512 + threadIdx.x) * 4, row = e >> LOGR, w = e & (R - 1);
        v[i] = *(gptr4c)(s + (size_t)row * STRIDE + w);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u32 e = (i * 512 + threadIdx.x) * 4, row = e >> LOGR, w = e & (R - 1);
        *(gptr4)(d + (size_t)row * STRIDE + w) = v[i];
    }
}

template <int LOGR, int XCD> static double run(const u32* src, u32* dst, int cols, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const unsigned grid = (unsigned)cols << (LOGN - TILE);
    hipLaunchKernelGGL((tile_copy<LOGR, XCD>), dim3(grid), dim3(512), 0, 0, src, dst, cols);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((tile_copy<LOGR, XCD>), dim3(grid), dim3(512), 0, 0, src, dst, cols);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return 2.0 * 4.0 * cols * (double)(1u << LOGN) * reps / (ms * 1e-3) / 1e9;
}

int main() {
    const int max_cols = 64;
    u32 *src, *dst;
    hipMalloc(&src, (size_t)max_cols << (LOGN + 2)); hipMalloc(&dst, (size_t)max_cols << (LOGN + 2));
    hipMemset(src, 1, (size_t)max_cols << (LOGN + 2));
    printf("{\"unit\": \"GB/s read+write\", \"tile_words\": 8192, \"column_words\": %u", 1u << LOGN);
    for (int cols : {4, 64}) {       // 4 columns = 128 MiB of in + out (Infinity-Cache resident), 64 columns = 2 GiB (HBM)
        const int reps = cols == 4 ? 160 : 10;
        printf(", \"cols_%d\": {\"run_64B\": %.0f, \"run_128B\": %.0f, \"run_256B\": %.0f, \"run_1KiB\": %.0f, \"run_32KiB\": %.0f, "
               "\"xcd_run_64B\": %.0f, \"xcd_run_128B\": %.0f, \"xcd_run_256B\": %.0f, \"xcd_run_32KiB\": %.0f}", cols,
               run<4, 0>(src, dst, cols, reps), run<5, 0>(src, dst, cols, reps), run<6, 0>(src, dst, cols, reps), run<8, 0>(src, dst, cols, reps), run<13, 0>(src, dst, cols, reps),
               run<4, 1>(src, dst, cols, reps), run<5, 1>(src, dst, cols, reps), run<6, 1>(src, dst, cols, reps), run<13, 1>(src, dst, cols, reps));
    }
    printf("}\n");
    return 0;
}
