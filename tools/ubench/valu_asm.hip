// Issue rate of individual gfx950 VALU instructions (inline asm so the compiler cannot fuse or simplify them).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32; typedef uint64_t u64;
#define ITERS 2048
#define REP8(x) x x x x x x x x
template <int OP> __global__ __launch_bounds__(256) void k(u32* out, u32 seed) {
    u32 a = threadIdx.x + seed, b = blockIdx.x * 7 + 3, c = a ^ 0x1234567, d = b + 99, t = seed * 2654435761u;
    u64 q = a, r = b;
#pragma unroll 1
    for (int i = 0; i < ITERS; i++) {
        if (OP == 0) { REP8(asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(t));) }
        if (OP == 1) { REP8(asm volatile("v_and_b32 %0, %0, %2\n v_and_b32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(t));) }
        if (OP == 2) { REP8(asm volatile("v_min_u32 %0, %0, %2\n v_min_u32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(t));) }
        if (OP == 3) { REP8(asm volatile("v_alignbit_b32 %0, %0, %2, 31\n v_alignbit_b32 %1, %1, %2, 31" : "+v"(a), "+v"(b) : "v"(t));) }
        if (OP == 4) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %2\n v_mul_lo_u32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(t));) }
        if (OP == 5) { REP8(asm volatile("v_mul_hi_u32 %0, %0, %2\n v_mul_hi_u32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(t));) }
        if (OP == 6) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(q), "+v"(r) : "v"(t), "v"(c) : "vcc");) }
        if (OP == 7) { REP8(asm volatile("v_add3_u32 %0, %0, %2, %3\n v_add3_u32 %1, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(t), "v"(c));) }
        if (OP == 8) { REP8(asm volatile("v_mul_u32_u24 %0, %0, %2\n v_mul_u32_u24 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(t));) }
        if (OP == 9) { REP8(asm volatile("v_sub_u32 %0, %0, %2\n v_min_u32 %1, %1, %0" : "+v"(a), "+v"(b) : "v"(t));) }
        if (OP == 10) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %2, %3\n v_mad_u32_u24 %1, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(t), "v"(c));) }
        if (OP == 11) { REP8(asm volatile("v_lshl_add_u32 %0, %0, 1, %2\n v_lshl_add_u32 %1, %1, 1, %2" : "+v"(a), "+v"(b) : "v"(t));) }
        if (OP == 12) { REP8(asm volatile("v_pk_add_u16 %0, %0, %2\n v_pk_add_u16 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(t));) }
        if (OP == 14) { REP8(asm volatile("v_subrev_co_u32 %0, vcc, %2, %0\n s_nop 1\n v_cndmask_b32 %1, %1, %0, vcc" : "+v"(a), "+v"(b) : "v"(t) : "vcc");) }
        if (OP == 15) { REP8(asm volatile("v_subrev_co_u32 %0, vcc, %2, %0\n v_cndmask_b32 %1, %1, %0, vcc" : "+v"(a), "+v"(b) : "v"(t) : "vcc");) }
        if (OP == 16) { REP8(asm volatile("v_subrev_co_u32 %0, vcc, %2, %0\n v_add_u32 %3, %3, %2\n v_add_u32 %4, %4, %2\n v_cndmask_b32 %1, %1, %0, vcc" : "+v"(a), "+v"(b), "+v"(t), "+v"(c), "+v"(d) : : "vcc");) }
        if (OP == 13) { REP8(asm volatile("v_dot4_u32_u8 %0, %2, %3, %0\n v_dot4_u32_u8 %1, %2, %3, %1" : "+v"(a), "+v"(b) : "v"(t), "v"(c));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ (u32)q ^ (u32)(q >> 32) ^ (u32)r;
}
template <int OP> void run(const char* name, int waves_per_simd) {
    u32* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int blocks = 256 * waves_per_simd;   // 256-thread blocks = 4 waves = 1 per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * ITERS * 16;         // wave-instructions
    double per_simd_cycles = ms * 1e-3 * 2.4e9 / (winstr / 1024);   // at 2.4 GHz nominal
    printf("%-22s waves/SIMD=%d %8.3f ms  %7.2f T lane-ops/s  %5.2f cycles/wave-instr/SIMD @2.4GHz\n", name, waves_per_simd, ms, winstr * 64 / ms / 1e9, per_simd_cycles);
    (void)hipFree(d);
}
int main() {
    for (int w : {8, 4, 2, 1}) run<0>("v_add_u32", w);
    run<1>("v_and_b32", 8); run<2>("v_min_u32", 8); run<3>("v_alignbit_b32", 8); run<4>("v_mul_lo_u32", 8); run<5>("v_mul_hi_u32", 8);
    run<6>("v_mad_u64_u32", 8); run<7>("v_add3_u32", 8); run<8>("v_mul_u32_u24", 8); run<9>("v_sub+v_min (dep)", 8); run<10>("v_mad_u32_u24", 8);
    run<11>("v_lshl_add_u32", 8); run<12>("v_pk_add_u16", 8); run<13>("v_dot4_u32_u8", 8);
    run<14>("sub_co, s_nop 1, cndmask", 8); run<15>("sub_co, cndmask NO nop (timing only)", 8); run<16>("sub_co, add, add, cndmask (2 counted)", 8);
    run<14>("sub_co, s_nop 1, cndmask w=4", 4); run<15>("sub_co, cndmask NO nop w=4", 4); run<16>("sub_co, add, add, cndmask w=4", 4);
    return 0;
}
