// Micro-benchmark for VERDICT r5 "next" #1: 12 butterfly layers of a 2^13-row FFT tile (the 13th is fused into the global staging in
// fft13.hip) as
//   MODE 0: three LDS round trips of 4 register layers each (what fft13_kernel does: tile bits [0,4), [4,8), [8,12));
//   MODE 1: two LDS round trips of 6 layers each: 4 register layers, then v_permlane16_swap_b32 / v_permlane32_swap_b32 exchange
//           register bits 3 / 2 with lane bits 4 / 5 (8 + 8 swaps per 16 registers), 2 more register layers;
//   MODE 2: as 1 with an XOR-swizzled tile (conflict-free reads and writes for both rounds by the guide's bank model) instead of
//           the 1/16 padding;
//   MODE 3: as 0 without butterfly arithmetic, MODE 4: as 1 without butterfly arithmetic (data movement only).
// Also the plain issue rate of the two swap instructions.  Real M31 butterflies (the doubled-twiddle carry-select form of
// field.cuh) on random canonical data; 512 lanes per block, 4 blocks per CU, every CU busy: the power state of the real kernel.
//   hipcc --offload-arch=gfx950 -O3 swap_rounds.hip -o swap_rounds && ./swap_rounds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32; typedef uint64_t u64;
constexpr u32 P = 0x7fffffffu;
#define FI __device__ __forceinline__
FI u32 csub(u32 s) { u32 d; bool br = __builtin_usub_overflow(s, P, &d); return br ? s : d; }
FI u32 madd(u32 a, u32 b) { return csub(a + b); }
FI u32 msub(u32 a, u32 b) { u32 d; bool br = __builtin_usub_overflow(a, b, &d); return br ? d + P : d; }
FI u32 mmul2(u32 a, u32 b2) { u64 p = (u64)a * b2; return csub((u32)(p >> 32) + ((u32)p >> 1)); }
template <bool ARITH> FI void bfly(u32& a, u32& b, u32 t2) {   // inverse butterfly
    if (ARITH) { u32 s = madd(a, b), d = msub(a, b); a = s; b = mmul2(d, t2); }
}

FI u32 pad13(u32 t) { return t + (t >> 4); }
// bank bits b0..3 = t0..3 ^ t6..9, b4 = t4 ^ t3 ^ t8 ^ t10 (a bijection on the 32 banks for every access of both 6-layer rounds)
FI u32 swz(u32 t) { return t ^ ((t >> 6) & 15u) ^ ((((t >> 3) ^ (t >> 8) ^ (t >> 10)) & 1u) << 4); }

template <int R, bool ARITH> FI void layers(u32* v, const u32* tw, int q0, int q1) {
#pragma unroll
    for (int q = q0; q < q1; q++)
#pragma unroll
        for (int e = 0; e < 16; e++) { if (e & (1 << q)) continue; bfly<ARITH>(v[e], v[e | (1 << q)], tw[(q * 5 + (e >> (q + 1))) & 15]); }
}

template <bool ARITH> FI void round4(u32* lds, int bp, const u32* tw) {
    const u32 w = threadIdx.x, wl = w & ((1u << bp) - 1), wh = w >> bp, t0 = (wh << (bp + 4)) | wl;
    const u32 es = bp ? ((1u << bp) + ((1u << bp) >> 4)) : 1u, p0 = pad13(t0);
    u32 v[16];
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = lds[p0 + e * es];
    layers<4, ARITH>(v, tw, 0, 4);
#pragma unroll
    for (int e = 0; e < 16; e++) lds[p0 + e * es] = v[ARITH ? e : (e ^ 1)];   // (without arithmetic a straight copy would be elided)
}

// tile bits [bp, bp+6), bp = 0 or 6; lane bits 4, 5 of the wave hold tile bits bp+4, bp+5 on the way in and bp+3, bp+2 on the way out
template <bool ARITH, bool SWZ> FI void round6(u32* lds, int bp, const u32* tw) {
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l4 = (lane >> 4) & 1, l5 = lane >> 5;
    const u32 o = (lane & 15) | (wv << 4);   // the 7 tile bits outside the round
    const u32 other = bp ? ((o & 63) | ((o >> 6) << 12)) : (o << 6);
    const u32 tin = other | (l4 << (bp + 4)) | (l5 << (bp + 5));
    const u32 tout = other | (l5 << (bp + 2)) | (l4 << (bp + 3));
    u32 v[16];
#pragma unroll
    for (int e = 0; e < 16; e++) { const u32 t = tin + ((u32)e << bp); v[e] = lds[SWZ ? swz(t) : pad13(t)]; }
    layers<4, ARITH>(v, tw, 0, 4);
#pragma unroll
    for (int e = 0; e < 8; e++) {
        auto r = __builtin_amdgcn_permlane16_swap(v[e], v[e | 8], false, false);
        v[e] = r[0]; v[e | 8] = r[1];
    }
#pragma unroll
    for (int e = 0; e < 16; e++) {
        if (e & 4) continue;
        auto r = __builtin_amdgcn_permlane32_swap(v[e], v[e | 4], false, false);
        v[e] = r[0]; v[e | 4] = r[1];
    }
    layers<4, ARITH>(v, tw, 3, 4);   // register bit 3 = tile bit bp+4
    layers<4, ARITH>(v, tw, 2, 3);   // register bit 2 = tile bit bp+5
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const u32 t = tout | ((u32)(e & 3) << bp) | ((u32)((e >> 3) & 1) << (bp + 4)) | ((u32)((e >> 2) & 1) << (bp + 5));
        lds[SWZ ? swz(t) : pad13(t)] = v[e];
    }
}

template <int MODE>
__global__ __launch_bounds__(512) void k(u32* out, int iters, u32 seed) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    for (u32 i = threadIdx.x; i < 8192 + 512; i += 512) lds[i] = (i * 2654435761u + seed + blockIdx.x) & 0x3fffffffu;
    u32 tw[16];
#pragma unroll
    for (int i = 0; i < 16; i++) tw[i] = 2 * (((threadIdx.x * 977 + i * 131 + seed) * 2246822519u) & 0x3fffffffu);
    __syncthreads();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0 || MODE == 3) {
            round4<MODE == 0>(lds, 0, tw); __syncthreads();
            round4<MODE == 0>(lds, 4, tw); __syncthreads();
            round4<MODE == 0>(lds, 8, tw); __syncthreads();
        } else {
            round6<MODE != 4, MODE == 2>(lds, 0, tw); __syncthreads();
            round6<MODE != 4, MODE == 2>(lds, 6, tw); __syncthreads();
        }
    }
    u32 acc = 0;
    for (u32 i = threadIdx.x; i < 8192 + 512; i += 512) acc ^= lds[i];
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(512) void swap_rate(u32* out, int iters) {
    u32 v[16];
    for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 31 + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int e = 0; e < 8; e++) { auto r = __builtin_amdgcn_permlane16_swap(v[e], v[e | 8], false, false); v[e] = r[0]; v[e | 8] = r[1]; }
#pragma unroll
        for (int e = 0; e < 16; e++) { if (e & 4) continue; auto r = __builtin_amdgcn_permlane32_swap(v[e], v[e | 4], false, false); v[e] = r[0]; v[e | 4] = r[1]; }
    }
    u32 a = 0; for (int i = 0; i < 16; i++) a ^= v[i];
    out[blockIdx.x * 512 + threadIdx.x] = a;
}

template <int MODE> void run(const char* name, int blocks_per_cu) {
    const size_t lds_bytes = (8192 + 512) * 4;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    u32* d; (void)hipMalloc(&d, 256 * 8 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * blocks_per_cu, iters = 400;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), lds_bytes, 0, d, 20, 1u);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), lds_bytes, 0, d, iters, 2u + rep);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipError_t err = hipGetLastError();
    const double tiles = (double)blocks * iters;
    const double bf = tiles * 4096.0 * 12.0;
    printf("{\"mode\": %d, \"name\": \"%s\", \"blocks_per_cu\": %d, \"ms\": %.3f, \"us_per_tile_slot\": %.3f, \"T_bfly_per_s\": %.3f%s}\n", MODE, name, blocks_per_cu,
           best, best * 1e3 / iters, bf / best / 1e9, err == hipSuccess ? "" : ", \"error\": true");
    (void)hipFree(d);
}

int main() {
    for (int bpc : {4, 3, 2}) {
        run<0>("3 LDS rounds x 4 layers", bpc);
        run<1>("2 LDS rounds x (4 + swap + 2) layers, padded", bpc);
        run<2>("2 LDS rounds x (4 + swap + 2) layers, xor swizzle", bpc);
        run<3>("3 LDS rounds, no arithmetic", bpc);
        run<4>("2 LDS rounds + swaps, no arithmetic", bpc);
    }
    {
        u32* d; (void)hipMalloc(&d, 256 * 8 * 512 * 4);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(swap_rate, dim3(2048), dim3(512), 0, 0, d, 100);
        (void)hipEventRecord(e0); hipLaunchKernelGGL(swap_rate, dim3(2048), dim3(512), 0, 0, d, 20000); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double winstr = 2048.0 * 8 * 20000 * 16;   // wave-instructions
        printf("{\"name\": \"v_permlane16/32_swap issue\", \"ms\": %.3f, \"cycles_per_wave_instr_per_simd_at_2.4GHz\": %.2f}\n", ms, ms * 1e-3 * 2.4e9 * 1024 / winstr);
    }
    return 0;
}
