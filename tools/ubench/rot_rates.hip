// Issue rates of the candidates for Blake2s' xor + rotate (G: d = rotr(d ^ a, 16 | 8), b = rotr(b ^ c, 12 | 7)) on gfx950: is any rotation cheaper than
// v_alignbit_b32 (4.4 cycles per wave-instruction against 2.8 of v_add / v_xor)?  Inline asm, 8 waves per SIMD, cycles at the nominal 2.4 GHz.
//   hipcc -O3 --offload-arch=gfx950 rot_rates.hip -o rot_rates && ./rot_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32;
#define ITERS 2048
#define REP8(x) x x x x x x x x
template <int OP> __global__ __launch_bounds__(256) void k(u32* out, u32 seed) {
    u32 a = threadIdx.x + seed, b = blockIdx.x * 7 + 3, c = a ^ 0x1234567, t = seed * 2654435761u, sel = 0x01000302u;
#pragma unroll 1
    for (int i = 0; i < ITERS; i++) {
        if (OP == 0) { REP8(asm volatile("v_xor_b32 %0, %0, %2\n v_xor_b32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(t));) }
        if (OP == 1) { REP8(asm volatile("v_alignbit_b32 %0, %0, %0, 16\n v_alignbit_b32 %1, %1, %1, 16" : "+v"(a), "+v"(b));) }
        if (OP == 2) { REP8(asm volatile("v_perm_b32 %0, %0, %0, %2\n v_perm_b32 %1, %1, %1, %2" : "+v"(a), "+v"(b) : "v"(sel));) }
        if (OP == 3) { REP8(asm volatile("v_alignbyte_b32 %0, %0, %0, 2\n v_alignbyte_b32 %1, %1, %1, 2" : "+v"(a), "+v"(b));) }
        if (OP == 4) { REP8(asm volatile("v_xor_b32_sdwa %0, %0, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n v_xor_b32_sdwa %1, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(a), "+v"(b) : "v"(t));) }
        if (OP == 5) { REP8(asm volatile("v_lshrrev_b32 %0, 12, %0\n v_lshrrev_b32 %1, 12, %1" : "+v"(a), "+v"(b));) }
        if (OP == 6) { REP8(asm volatile("v_lshl_or_b32 %0, %0, 20, %2\n v_lshl_or_b32 %1, %1, 20, %2" : "+v"(a), "+v"(b) : "v"(t));) }
        if (OP == 7) { REP8(asm volatile("v_bitop3_b32 %0, %0, %2, %3 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0x96" : "+v"(a), "+v"(b) : "v"(t), "v"(c));) }
        if (OP == 8) { REP8(asm volatile("v_bfi_b32 %0, %2, %0, %3\n v_bfi_b32 %1, %2, %1, %3" : "+v"(a), "+v"(b) : "v"(t), "v"(c));) }
        if (OP == 9) { REP8(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b));) }
        if (OP == 10) { REP8(asm volatile("v_xor_b32_dpp %0, %0, %2 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_xor_b32_dpp %1, %1, %2 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b) : "v"(t));) }
        if (OP == 11) { REP8(asm volatile("v_or_b32 %0, %0, %2\n v_lshlrev_b32 %1, 1, %1" : "+v"(a), "+v"(b) : "v"(t));) }
        if (OP == 12) { REP8(asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(t));) }
        if (OP == 13) { REP8(asm volatile("v_pk_lshrrev_b16 %0, 8, %0\n v_pk_lshrrev_b16 %1, 8, %1" : "+v"(a), "+v"(b));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c;
}
template <int OP> void run(const char* name, int waves_per_simd) {
    u32* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int blocks = 256 * waves_per_simd;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * ITERS * 16;
    printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"cycles_per_wave_instr_at_2.4GHz\": %.2f}\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / (winstr / 1024));
    (void)hipFree(d);
}
int main() {
    run<12>("v_add_u32", 8); run<0>("v_xor_b32", 8); run<1>("v_alignbit_b32 (rotr 16)", 8); run<2>("v_perm_b32 (rotr 16)", 8); run<3>("v_alignbyte_b32 (rotr 16)", 8);
    run<4>("v_xor_b32_sdwa (half of xor+rotr16)", 8); run<5>("v_lshrrev_b32", 8); run<6>("v_lshl_or_b32", 8); run<7>("v_bitop3_b32 (xor3)", 8); run<8>("v_bfi_b32", 8);
    run<9>("v_mov_b32_dpp quad_perm", 8); run<10>("v_xor_b32_dpp quad_perm", 8); run<11>("v_or_b32 / v_lshlrev_b32", 8); run<13>("v_pk_lshrrev_b16", 8);
    run<12>("v_add_u32", 1); run<1>("v_alignbit_b32 (rotr 16)", 1); run<4>("v_xor_b32_sdwa", 1);
    return 0;
}
