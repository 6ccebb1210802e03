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

// F: two forward butterflies in one hand-scheduled asm block: every VALU write of VCC is followed by >= 2 independent
// instructions before the v_cndmask that reads it (gfx940+: VALU-written SGPR/VCC read by a VALU needs 2 wait states), so
// no s_nop is needed.  y = a - m uses plain v_sub + v_cmp (the compare can be placed freely) instead of v_sub_co.
FI void bfly2_asm(u32& a0, u32& b0, u32& a1, u32& b1, u32 t0, u32 t1) {
    u64 p0, p1; u32 m0, m1, u0, u1, s0, s1, d0, d1, e0, e1;
    const u32 Pc = P;
    asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, 0\n v_mad_u64_u32 %1, vcc, %4, %5, 0" : "=&v"(p0), "=&v"(p1) : "v"(b0), "v"(t0), "v"(b1), "v"(t1) : "vcc");
    const u32 l0 = (u32)p0, h0 = (u32)(p0 >> 32), l1 = (u32)p1, h1 = (u32)(p1 >> 32);
    asm volatile(
        "v_lshrrev_b32 %[m0], 1, %[l0]\n"
        "v_lshrrev_b32 %[m1], 1, %[l1]\n"
        "v_add_u32 %[m0], %[m0], %[h0]\n"
        "v_add_u32 %[m1], %[m1], %[h1]\n"
        "v_subrev_co_u32 %[u0], vcc, %[P], %[m0]\n"      // V(m0)
        "s_nop 1\n"
        "v_cndmask_b32 %[m0], %[u0], %[m0], vcc\n"       // R
        "v_subrev_co_u32 %[u1], vcc, %[P], %[m1]\n"      // V(m1)
        "v_add_u32 %[s0], %[a0], %[m0]\n"                 // filler: x sum 0
        "v_sub_u32 %[d0], %[a0], %[m0]\n"                 // filler: y diff 0
        "v_cndmask_b32 %[m1], %[u1], %[m1], vcc\n"       // R
        "v_subrev_co_u32 %[u0], vcc, %[P], %[s0]\n"      // V(x0)
        "v_add_u32 %[s1], %[a1], %[m1]\n"                 // filler
        "v_sub_u32 %[d1], %[a1], %[m1]\n"                 // filler
        "v_cndmask_b32 %[e0], %[u0], %[s0], vcc\n"       // R: x0
        "v_subrev_co_u32 %[u1], vcc, %[P], %[s1]\n"      // V(x1)
        "v_add_u32 %[u0], %[P], %[d0]\n"                  // filler: d0 + P
        "v_add_u32 %[s0], %[P], %[d1]\n"                  // filler: d1 + P
        "v_cndmask_b32 %[e1], %[u1], %[s1], vcc\n"       // R: x1
        "v_cmp_lt_u32 vcc, %[a0], %[m0]\n"                // V(y0): borrow
        "s_nop 1\n"
        "v_cndmask_b32 %[b0], %[d0], %[u0], vcc\n"       // R: y0
        "v_cmp_lt_u32 vcc, %[a1], %[m1]\n"                // V(y1)
        "v_mov_b32 %[a0], %[e0]\n"                        // filler: x0 out
        "s_nop 0\n"
        "v_cndmask_b32 %[b1], %[d1], %[s0], vcc\n"       // R: y1
        "v_mov_b32 %[a1], %[e1]\n"                        // x1 out
        : [a0] "+v"(a0), [b0] "+v"(b0), [a1] "+v"(a1), [b1] "+v"(b1), [m0] "=&v"(m0), [m1] "=&v"(m1),
          [u0] "=&v"(u0), [u1] "=&v"(u1), [s0] "=&v"(s0), [s1] "=&v"(s1), [d0] "=&v"(d0), [d1] "=&v"(d1), [e0] "=&v"(e0), [e1] "=&v"(e1)
        : [l0] "v"(l0), [h0] "v"(h0), [l1] "v"(l1), [h1] "v"(h1), [P] "s"(Pc)
        : "vcc");
}

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
                if (V == 5) { if (!(e & (1 << q)) && (e & (1 << ((q + 1) & 3))) == 0) { int e2 = e | (1 << ((q + 1) & 3)); bfly2_asm(x[e], x[e | (1 << q)], x[e2], x[e2 | (1 << q)], t2, t2); } }
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
    run<5>("F: asm pair, hazard-free schedule");
    return 0;
}
