// Blake2s compression throughput for different instruction selections of G (one hash chain per lane).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32;
#define FI __device__ __forceinline__
template <int V> FI u32 rotr(u32 x, int r) {
    if (V == 1 && r == 16) return __builtin_amdgcn_perm(x, x, 0x01000302u);   // byte permute
    if (V == 1 && r == 8) return __builtin_amdgcn_perm(x, x, 0x00030201u);
    if (V == 2) return (x >> r) | (x << (32 - r));                             // let the compiler choose
    return __builtin_amdgcn_alignbit(x, x, r);
}
// rotr(d ^ a, 16) as two SDWA xors (VOP2 encodings with word selects): the halves of the xor land swapped
FI u32 xor_rot16_sdwa(u32 d, u32 a) {
    u32 t;
    asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(t) : "v"(d), "v"(a));
    asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0" : "+v"(t) : "v"(d), "v"(a));
    return t;
}
template <int V> FI void G(u32& a, u32& b, u32& c, u32& d, u32 x, u32 y) {
    if (V == 3) { a = a + b; a = a + x; } else a = a + b + x;
    if (V == 4) d = xor_rot16_sdwa(d, a); else d = rotr<V>(d ^ a, 16);
    c = c + d; b = rotr<V>(b ^ c, 12);
    if (V == 3) { a = a + b; a = a + y; } else a = a + b + y;
    d = rotr<V>(d ^ a, 8); c = c + d; b = rotr<V>(b ^ c, 7);
}
__device__ __constant__ unsigned char SIG[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
constexpr unsigned char CSIG[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
template <int V> FI void compress(u32* h, const u32* m, u32 t0, u32 f0) {
    u32 v[16];
    for (int i = 0; i < 8; i++) v[i] = h[i];
    v[8] = 0x6A09E667u; v[9] = 0xBB67AE85u; v[10] = 0x3C6EF372u; v[11] = 0xA54FF53Au;
    v[12] = 0x510E527Fu ^ t0; v[13] = 0x9B05688Cu; v[14] = 0x1F83D9ABu ^ f0; v[15] = 0x5BE0CD19u;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        G<V>(v[0], v[4], v[8], v[12], m[CSIG[r][0]], m[CSIG[r][1]]);   G<V>(v[1], v[5], v[9], v[13], m[CSIG[r][2]], m[CSIG[r][3]]);
        G<V>(v[2], v[6], v[10], v[14], m[CSIG[r][4]], m[CSIG[r][5]]);  G<V>(v[3], v[7], v[11], v[15], m[CSIG[r][6]], m[CSIG[r][7]]);
        G<V>(v[0], v[5], v[10], v[15], m[CSIG[r][8]], m[CSIG[r][9]]);  G<V>(v[1], v[6], v[11], v[12], m[CSIG[r][10]], m[CSIG[r][11]]);
        G<V>(v[2], v[7], v[8], v[13], m[CSIG[r][12]], m[CSIG[r][13]]); G<V>(v[3], v[4], v[9], v[14], m[CSIG[r][14]], m[CSIG[r][15]]);
    }
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}
#define ITERS 1024   // ~15 ms per run: long enough for the clock to settle (the 64-iteration version measured the ramp)
template <int V, int NCH> __global__ __launch_bounds__(256) void k(u32* out, u32 seed) {
    u32 h[NCH][8], m[16];
    for (int c = 0; c < NCH; c++) for (int i = 0; i < 8; i++) h[c][i] = threadIdx.x * 31 + i + c * 7 + seed;
    for (int i = 0; i < 16; i++) m[i] = blockIdx.x * 17 + i * 3 + seed;
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int c = 0; c < NCH; c++) compress<V>(h[c], m, 64 * it, 0);
        m[it & 15] += h[0][0];
    }
    u32 r = 0; for (int c = 0; c < NCH; c++) for (int i = 0; i < 8; i++) r ^= h[c][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int V, int NCH> void run(const char* name, int blocks_per_cu) {
    u32* d = nullptr; hipError_t me = hipMalloc(&d, 256 * 16 * 256 * 4);
    if (me != hipSuccess) { printf("hipMalloc: %s\n", hipGetErrorString(me)); return; }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int blocks = 256 * blocks_per_cu;
    hipLaunchKernelGGL((k<V, NCH>), dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipLaunchKernelGGL((k<V, NCH>), dim3(blocks), dim3(256), 0, 0, d, 3u);
    (void)hipEventRecord(e0); hipLaunchKernelGGL((k<V, NCH>), dim3(blocks), dim3(256), 0, 0, d, 2u); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double n = (double)blocks * 256 * ITERS * NCH;
    printf("%-44s blk/CU=%d %8.3f ms  %6.2f G compressions/s  (369M per prove = %.2f ms)\n", name, blocks_per_cu, ms, n / ms / 1e6, 369e6 / (n / ms / 1e3) );
    (void)hipFree(d);
}
int main() {
    run<0, 1>("warm-up", 8);
    for (int o : {1, 2, 4, 8}) run<0, 1>("alignbit, add3 (current)", o);      // latency- or issue-bound?
    run<4, 1>("SDWA xor pair for rot16, alignbit, add3", 8);
    run<4, 1>("SDWA xor pair for rot16, alignbit, add3", 4);
    run<1, 1>("perm for 16/8, add3", 8);
    run<2, 1>("shift/or rotate (compiler's choice)", 8);
    run<3, 1>("alignbit, two adds", 8);
    run<0, 2>("alignbit, add3, 2 chains per lane", 4);
    return 0;
}
