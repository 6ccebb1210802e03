// Throughput of M31 butterfly formulations (8 independent butterflies per lane per iteration).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32; typedef uint64_t u64;
constexpr u32 P = 0x7fffffffu;
#define FI __device__ __forceinline__
FI u32 umin32(u32 a, u32 b) { return a < b ? a : b; }
// A: current (v_min based)
FI u32 addA(u32 a, u32 b) { u32 s = a + b; return umin32(s, s - P); }
FI u32 subA(u32 a, u32 b) { u32 d = a - b; return umin32(d, d + P); }
FI u32 mulA(u32 a, u32 b) { u64 p = (u64)a * b; u32 lo = (u32)p & P, hi = (u32)(p >> 31); u32 s = lo + hi; return umin32(s, s - P); }
// B: carry based, doubled twiddle
FI u32 csub(u32 s) { u32 d; bool br = __builtin_usub_overflow(s, P, &d); return br ? s : d; }
FI u32 addB(u32 a, u32 b) { return csub(a + b); }
FI u32 subB(u32 a, u32 b) { u32 d; bool br = __builtin_usub_overflow(a, b, &d); return br ? d + P : d; }
FI u32 mulB(u32 a, u32 b2) { u64 p = (u64)a * b2; return csub((u32)(p >> 32) + ((u32)p >> 1)); }
// C: doubled twiddle, min based reductions
FI u32 mulC(u32 a, u32 b2) { u64 p = (u64)a * b2; u32 s = (u32)(p >> 32) + ((u32)p >> 1); return umin32(s, s - P); }
// D: compare+select
FI u32 csubD(u32 s) { return s >= P ? s - P : s; }
FI u32 addD(u32 a, u32 b) { return csubD(a + b); }
FI u32 subD(u32 a, u32 b) { return a >= b ? a - b : a - b + P; }
FI u32 mulD(u32 a, u32 b2) { u64 p = (u64)a * b2; return csubD((u32)(p >> 32) + ((u32)p >> 1)); }
#define ITERS 1024
template <int V> __global__ __launch_bounds__(256) void k(u32* out, u32 seed) {
    u32 x[16];
    for (int i = 0; i < 16; i++) x[i] = (threadIdx.x * 977 + i * 131 + seed) & 0x3fffffff;
    u32 t = (seed * 2654435761u) & 0x3fffffff, t2 = 2 * t;
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                if (e & (1 << q)) continue;
                u32 &a = x[e], &b = x[e | (1 << q)];
                if (V == 0) { u32 m = mulA(b, t); u32 s = addA(a, m), d = subA(a, m); a = s; b = d; }
                if (V == 1) { u32 m = mulB(b, t2); u32 s = addB(a, m), d = subB(a, m); a = s; b = d; }
                if (V == 2) { u32 m = mulC(b, t2); u32 s = addA(a, m), d = subA(a, m); a = s; b = d; }
                if (V == 3) { u32 m = mulD(b, t2); u32 s = addD(a, m), d = subD(a, m); a = s; b = d; }
                if (V == 4) { u32 s = addB(a, b), d = subB(a, b); a = s; b = mulB(d, t2); }   // inverse butterfly
            }
        }
    }
    u32 r = 0; for (int i = 0; i < 16; i++) r ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int V> void run(const char* name) {
    u32* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int blocks = 256 * 8;
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, 2u); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double n = (double)blocks * 256 * ITERS * 32;
    printf("%-44s %8.3f ms  %6.2f T butterflies/s\n", name, ms, n / ms / 1e9);
    (void)hipFree(d);
}
int main() {
    run<0>("A: mad, and+alignbit, v_min reductions");
    run<1>("B: doubled twiddle, carry+cndmask");
    run<2>("C: doubled twiddle, v_min reductions");
    run<3>("D: doubled twiddle, compare+select");
    run<4>("E: inverse butterfly, B formulation");
    return 0;
}
