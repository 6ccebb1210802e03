// Micro-benchmark: issue rates of the integer instructions the M31 butterfly is made of (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../../nexus-zkvm_amd/csrc/field.cuh"
using namespace nx;
#define ITERS 4096
template <int OP> __global__ void k(u32* out, u32 seed) {
    u32 a = threadIdx.x + seed, b = blockIdx.x * 7 + 3, c = a ^ 0x1234567, d = b + 99;
    u32 t = (seed * 2654435761u) & P;
#pragma unroll 1
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (OP == 0) { a += b; c += d; b += a; d += c; }
            else if (OP == 1) { a = a * t + 1; c = c * t + 3; b = b * t + 5; d = d * t + 7; }           // v_mul_lo / mad
            else if (OP == 2) { a = __umulhi(a, t) + 1; c = __umulhi(c, t) + 3; b = __umulhi(b, t) + 5; d = __umulhi(d, t) + 7; }
            else if (OP == 3) { u64 p = (u64)a * t + b; a = (u32)p ^ (u32)(p >> 32); p = (u64)c * t + d; c = (u32)p ^ (u32)(p >> 32);
                                p = (u64)b * t + a; b = (u32)p ^ (u32)(p >> 32); p = (u64)d * t + c; d = (u32)p ^ (u32)(p >> 32); }
            else if (OP == 4) { a = m_mul(a & P, t); c = m_mul(c & P, t); b = m_mul(b & P, t); d = m_mul(d & P, t); }
            else if (OP == 5) { u32 m = m_mul(b, t); u32 x = m_add(a, m), y = m_sub(a, m); a = x; b = y; m = m_mul(d, t); x = m_add(c, m); y = m_sub(c, m); c = x; d = y;
                                m = m_mul(b, t); x = m_add(a, m); y = m_sub(a, m); a = x; b = y; m = m_mul(d, t); x = m_add(c, m); y = m_sub(c, m); c = x; d = y; }
            else if (OP == 6) { a = __builtin_amdgcn_alignbit(a, b, 7) ^ c; c = __builtin_amdgcn_alignbit(c, d, 9) ^ a; b = __builtin_amdgcn_alignbit(b, a, 3) ^ d; d = __builtin_amdgcn_alignbit(d, c, 5) ^ b; }
            else if (OP == 7) { a = __umul24(a, t) + b; c = __umul24(c, t) + d; b = __umul24(b, t) + a; d = __umul24(d, t) + c; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}
template <int OP> void run(const char* name, double ops_per_inner) {
    u32* d; hipMalloc(&d, 256 * 2048 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 256 * 8;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)blocks * 256 * ITERS * 8 * ops_per_inner;
    printf("%-28s %8.3f ms  %8.2f T lane-ops/s (counted units)\n", name, ms, n / ms / 1e9);
    hipFree(d);
}
int main() {
    run<0>("v_add_u32 (4/iter)", 4);
    run<1>("v_mul_lo_u32+add (4 mul)", 4);
    run<2>("v_mul_hi_u32+add (4 mul)", 4);
    run<3>("v_mad_u64_u32+xor (4 mad)", 4);
    run<4>("m_mul (4/iter)", 4);
    run<5>("butterfly fwd (4/iter)", 4);
    run<6>("v_alignbit+xor (4+4)", 8);
    run<7>("v_mul_u32_u24+add (4)", 4);
    return 0;
}
