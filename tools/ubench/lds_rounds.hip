// Micro-benchmark: cost of one "LDS round trip" of the FFT tile (16 reads + 16 writes per lane + barrier)
// for different strides, widths, paddings and block sizes.  hipcc --offload-arch=gfx950 -O3 lds_rounds.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32;

__device__ __forceinline__ u32 pad(u32 t, int padsh) { return padsh ? t + (t >> padsh) : t; }

// MODE 0: uint2 elements (ds_read_b64 / ds_write_b64), 1: u32 elements, 2: uint4 elements
template <int MODE, int THREADS, bool BARRIER>
__global__ __launch_bounds__(THREADS) void k(u32* out, int iters, int bp, int padsh, int rows_log) {
    extern __shared__ __attribute__((aligned(16))) u32 lds_raw[];
    const u32 rows = 1u << rows_log;
    const u32 nblk = rows >> 4;
    const u32 estride = bp ? ((1u << bp) + (padsh ? ((1u << bp) >> padsh) : 0)) : 1u;
    u32 acc = 0;
    for (int it = 0; it < iters; it++) {
        for (u32 w = threadIdx.x; w < nblk; w += THREADS) {
            const u32 wl = w & ((1u << bp) - 1), wh = w >> bp;
            const u32 p0 = pad((wh << (bp + 4)) | wl, padsh);
            if (MODE == 0) {
                uint2* lds = (uint2*)lds_raw;
                uint2 v[16];
#pragma unroll
                for (int e = 0; e < 16; e++) v[e] = lds[p0 + e * estride];
#pragma unroll
                for (int e = 0; e < 16; e++) { v[e].x += v[e ^ 1].y + it; v[e].y ^= v[e ^ 2].x; }
#pragma unroll
                for (int e = 0; e < 16; e++) lds[p0 + e * estride] = v[e];
                acc += v[3].x;
            } else if (MODE == 1) {
                u32* lds = lds_raw;
                u32 v[16];
#pragma unroll
                for (int e = 0; e < 16; e++) v[e] = lds[p0 + e * estride];
#pragma unroll
                for (int e = 0; e < 16; e++) v[e] += v[e ^ 1] + it;
#pragma unroll
                for (int e = 0; e < 16; e++) lds[p0 + e * estride] = v[e];
                acc += v[3];
            } else {
                uint4* lds = (uint4*)lds_raw;
                uint4 v[16];
#pragma unroll
                for (int e = 0; e < 16; e++) v[e] = lds[p0 + e * estride];
#pragma unroll
                for (int e = 0; e < 16; e++) { v[e].x += v[e ^ 1].y + it; v[e].z ^= v[e ^ 2].w; }
#pragma unroll
                for (int e = 0; e < 16; e++) lds[p0 + e * estride] = v[e];
                acc += v[3].x;
            }
        }
        if (BARRIER) __syncthreads();
    }
    out[blockIdx.x * THREADS + threadIdx.x] = acc;
}

template <int MODE, int THREADS, bool BARRIER>
void run(const char* name, int bp, int padsh, int rows_log, int blocks_per_cu) {
    const int esz = MODE == 0 ? 8 : MODE == 1 ? 4 : 16;
    size_t rows = (size_t)1 << rows_log;
    size_t lds_bytes = (rows + (padsh ? (rows >> padsh) : 0) + 64) * esz;
    hipFuncSetAttribute((const void*)k<MODE, THREADS, BARRIER>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    u32* d; hipMalloc(&d, 256 * 8 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 256 * blocks_per_cu, iters = 200;
    hipLaunchKernelGGL((k<MODE, THREADS, BARRIER>), dim3(blocks), dim3(THREADS), lds_bytes, 0, d, 10, bp, padsh, rows_log);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, THREADS, BARRIER>), dim3(blocks), dim3(THREADS), lds_bytes, 0, d, iters, bp, padsh, rows_log);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    double us_per_round = ms * 1e3 / iters;   // per block-round, with blocks_per_cu blocks resident
    double bytes = (double)rows * esz * 2 * iters * blocks;   // read + write
    printf("%-34s bp=%d pad=%d rows=2^%d blk/CU=%d  %7.3f us/round  %7.2f TB/s LDS (r+w)  %s\n", name, bp, padsh, rows_log, blocks_per_cu,
           us_per_round, bytes / (ms * 1e-3) / 1e12, err == hipSuccess ? "" : hipGetErrorString(err));
    hipFree(d);
}

int main() {
    for (int bp : {0, 4, 8}) run<0, 512, true>("b64 512thr barrier", bp, 4, 13, 2);
    for (int bp : {0, 4, 8}) run<0, 512, false>("b64 512thr nobarrier", bp, 4, 13, 2);
    for (int bp : {0, 4, 8}) run<0, 512, true>("b64 512thr barrier nopad", bp, 0, 13, 2);
    for (int bp : {0, 4, 8}) run<0, 256, true>("b64 256thr barrier", bp, 4, 13, 2);
    for (int bp : {0, 4, 8}) run<0, 1024, true>("b64 1024thr barrier", bp, 4, 13, 2);
    for (int bp : {0, 4, 8}) run<0, 512, true>("b64 512thr barrier 1blk/CU", bp, 4, 13, 1);
    for (int bp : {0, 4, 8}) run<1, 512, true>("b32 512thr barrier (2^13 rows)", bp, 4, 13, 4);
    for (int bp : {0, 4, 8}) run<1, 256, true>("b32 256thr barrier (2^12 rows)", bp, 4, 12, 8);
    for (int bp : {0, 4, 8}) run<2, 512, true>("b128 512thr barrier (2^12 rows)", bp, 4, 12, 2);
    for (int bp : {0, 4, 8}) run<0, 256, true>("b64 256thr barrier (2^12 rows)", bp, 4, 12, 4);
    for (int bp : {0, 4, 8}) run<0, 512, true>("b64 512thr pad5", bp, 5, 13, 2);
    return 0;
}
