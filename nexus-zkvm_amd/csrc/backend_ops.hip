// The remaining rows of the Stwo backend-trait family that stwo::prover::prove reaches through
// `impl Backend for B` (reference call sites prover/src/machine.rs:286-290, prover2/machine/src/prove.rs:124-128):
//   AccumulationOps::{accumulate, generate_secure_powers}      -> nx_secure_accumulate, nx_generate_secure_powers
//   FieldOps<BaseField / SecureField>::batch_inverse            -> nx_batch_inverse_m31, nx_batch_inverse_qm31
//   ColumnOps<SecureField>::bit_reverse_column                  -> nx_bit_reverse_secure
//   MerkleOps<Blake2sMerkleHasher>::commit_on_layer (one layer) -> nx_merkle_commit_on_layer
//   FriOps::decompose                                           -> nx_fri_decompose
// None of them is a hot spot of the commit-and-prove path (the composition accumulator and the logup kernels fuse what they
// need); they exist so that a per-op `HipBackend` type-checks against every supertrait of Stwo's `Backend`.
// Semantics follow Stwo's CpuBackend [upstream-recollection]; each has a parity test against the CPU checker in tests/.
#include "internal.h"
#include "air.h"
#include <algorithm>

namespace nx {

// one inverse per element: the inverse is unique, so Stwo's Montgomery batch trick and this give the same words
__global__ __launch_bounds__(256) void batch_inverse_m31_kernel(const u32* __restrict__ src, u32* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = m_inv(src[i]);
}
struct Sec4 { u32* c[4]; };
struct Sec4C { const u32* c[4]; };
__global__ __launch_bounds__(256) void batch_inverse_qm31_kernel(Sec4C s, Sec4 d, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const QM31 v = q_inv(qm(s.c[0][i], s.c[1][i], s.c[2][i], s.c[3][i]));
        d.c[0][i] = v.a.a; d.c[1][i] = v.a.b; d.c[2][i] = v.b.a; d.c[3][i] = v.b.b;
    }
}

// FriOps::decompose: lambda = (sum over the first half - sum over the second half) / N of a bit-reversed circle evaluation,
// g = f - lambda on the first half, f + lambda on the second.  Three launches: block partial sums (signed by the half), then the
// one block's reduction of the partials to lambda, then the subtraction.
constexpr int DEC_THREADS = 256, DEC_ITEMS = 16;
__global__ __launch_bounds__(DEC_THREADS) void decompose_sum_kernel(Sec4C s, u32 n, u32* __restrict__ partial /*[blocks][4]*/) {
    __shared__ u32 red[4][DEC_THREADS / 64];
    const u32 half = n >> 1;
    u32 acc[4] = {0, 0, 0, 0};
    const u32 base = blockIdx.x * DEC_THREADS * DEC_ITEMS;
    for (int k = 0; k < DEC_ITEMS; k++) {
        const u32 i = base + k * DEC_THREADS + threadIdx.x;
        if (i < n) {
            const bool neg = i >= half;
#pragma unroll
            for (int q = 0; q < 4; q++) { const u32 v = s.c[q][i]; acc[q] = neg ? m_sub(acc[q], v) : m_add(acc[q], v); }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        for (int off = 32; off > 0; off >>= 1) acc[q] = m_add(acc[q], __shfl_down(acc[q], off, 64));
        if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = acc[q];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        u32 t = 0;
        for (int w = 0; w < DEC_THREADS / 64; w++) t = m_add(t, red[threadIdx.x][w]);
        partial[4 * blockIdx.x + threadIdx.x] = t;
    }
}
// lambda = (sum of the block partials) / N, reduced ONCE by one block (the partial count grows with the column: 2^30 rows = 2^18 blocks)
__global__ __launch_bounds__(256) void decompose_lambda_kernel(const u32* __restrict__ partial, u32 n_partial, u32 n_inv, u32* __restrict__ lambda_out) {
    __shared__ u32 red[4][256];
    u32 acc[4] = {0, 0, 0, 0};
    for (u32 b = threadIdx.x; b < n_partial; b += 256)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = m_add(acc[q], partial[4 * b + q]);
#pragma unroll
    for (int q = 0; q < 4; q++) red[q][threadIdx.x] = acc[q];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
#pragma unroll
            for (int q = 0; q < 4; q++) red[q][threadIdx.x] = m_add(red[q][threadIdx.x], red[q][threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x < 4) lambda_out[threadIdx.x] = m_mul(red[threadIdx.x][0], n_inv);
}
__global__ __launch_bounds__(256) void decompose_apply_kernel(Sec4C s, Sec4 g, u32 n, const u32* __restrict__ lambda) {
    __shared__ u32 lam[4];
    if (threadIdx.x < 4) lam[threadIdx.x] = lambda[threadIdx.x];
    __syncthreads();
    const u32 half = n >> 1;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int q = 0; q < 4; q++) { const u32 v = s.c[q][i]; g.c[q][i] = i < half ? m_sub(v, lam[q]) : m_add(v, lam[q]); }
    }
}

}  // namespace nx

using namespace nx;

extern "C" {

int nx_secure_accumulate(nx_ctx* ctx, uint32_t* const* d_dst4, const uint32_t* const* d_src4, uint32_t log_size) {
    NX_GUARD(ctx);
    if (!ctx || !d_dst4 || !d_src4) return set_err(ctx, NX_ERR_ARG, "nx_secure_accumulate: NULL argument");
    if (log_size > 30) return set_err(ctx, NX_ERR_ARG, "nx_secure_accumulate: log_size too large");
    for (int q = 0; q < 4; q++) if (!d_dst4[q] || !d_src4[q]) return set_err(ctx, NX_ERR_ARG, "nx_secure_accumulate: NULL coordinate column");
    u32* d[4] = {d_dst4[0], d_dst4[1], d_dst4[2], d_dst4[3]};
    const u32* s[4] = {d_src4[0], d_src4[1], d_src4[2], d_src4[3]};
    return secure_accumulate(ctx, d, s, 1u << log_size);
}

// AccumulationOps::generate_secure_powers: [1, felt, felt^2, ...] (host arithmetic: the list is as long as an AIR has constraints)
int nx_generate_secure_powers(const uint32_t felt[4], uint32_t n_powers, uint32_t* out /* 4 words each */) {
    if (!felt || (n_powers && !out)) return set_err(nullptr, NX_ERR_ARG, "nx_generate_secure_powers: NULL argument");
    const QM31 f = q_load(felt);
    QM31 a = q_one();
    for (uint32_t i = 0; i < n_powers; i++) { q_store(out + 4 * (size_t)i, a); a = q_mul(a, f); }
    return NX_OK;
}

static unsigned grid_for(nx_ctx* ctx, size_t n) { return (unsigned)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, (size_t)ctx->n_cus * 16)); }

int nx_batch_inverse_m31(nx_ctx* ctx, const uint32_t* d_src, uint32_t* d_dst, size_t n) {
    NX_GUARD(ctx);
    if (!ctx || (n && (!d_src || !d_dst))) return set_err(ctx, NX_ERR_ARG, "nx_batch_inverse_m31: NULL argument");
    if (!n) return NX_OK;
    hipLaunchKernelGGL(batch_inverse_m31_kernel, dim3(grid_for(ctx, n)), dim3(256), 0, ctx->stream, d_src, d_dst, n);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

int nx_batch_inverse_qm31(nx_ctx* ctx, const uint32_t* const* d_src4, uint32_t* const* d_dst4, size_t n) {
    NX_GUARD(ctx);
    if (!ctx || !d_src4 || !d_dst4) return set_err(ctx, NX_ERR_ARG, "nx_batch_inverse_qm31: NULL argument");
    if (!n) return NX_OK;
    Sec4C s; Sec4 d;
    for (int q = 0; q < 4; q++) { if (!d_src4[q] || !d_dst4[q]) return set_err(ctx, NX_ERR_ARG, "nx_batch_inverse_qm31: NULL coordinate column"); s.c[q] = d_src4[q]; d.c[q] = d_dst4[q]; }
    hipLaunchKernelGGL(batch_inverse_qm31_kernel, dim3(grid_for(ctx, n)), dim3(256), 0, ctx->stream, s, d, n);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

int nx_bit_reverse_secure(nx_ctx* ctx, uint32_t* const* d_col4, uint32_t log_size) {
    if (!ctx || !d_col4) return set_err(ctx, NX_ERR_ARG, "nx_bit_reverse_secure: NULL argument");
    for (int q = 0; q < 4; q++) NX_TRY(nx_bit_reverse(ctx, d_col4[q], log_size));
    return NX_OK;
}

int nx_merkle_commit_on_layer(nx_ctx* ctx, uint32_t log_size, const uint32_t* d_prev_layer, const uint32_t* const* d_cols, uint32_t n_cols,
                              uint32_t* d_out) {
    NX_GUARD(ctx);
    if (!ctx || !d_out || (n_cols && !d_cols)) return set_err(ctx, NX_ERR_ARG, "nx_merkle_commit_on_layer: NULL argument");
    if (log_size > 30) return set_err(ctx, NX_ERR_ARG, "nx_merkle_commit_on_layer: layer too large");
    ColSet cs; NX_TRY(make_colset(ctx, d_cols, n_cols, &cs));
    return merkle_layer(ctx, cs, n_cols, d_prev_layer, d_out, log_size);
}

int nx_fri_decompose(nx_ctx* ctx, const uint32_t* const* d_src4, uint32_t log_size, uint32_t* const* d_g4, uint32_t lambda[4]) {
    NX_GUARD(ctx);
    if (!ctx || !d_src4 || !d_g4 || !lambda) return set_err(ctx, NX_ERR_ARG, "nx_fri_decompose: NULL argument");
    if (log_size < 1 || log_size > 30) return set_err(ctx, NX_ERR_ARG, "nx_fri_decompose: 1 <= log_size <= 30 required");
    Sec4C s; Sec4 g;
    for (int q = 0; q < 4; q++) { if (!d_src4[q] || !d_g4[q]) return set_err(ctx, NX_ERR_ARG, "nx_fri_decompose: NULL coordinate column"); s.c[q] = d_src4[q]; g.c[q] = d_g4[q]; }
    const u32 n = 1u << log_size;
    const u32 n_blocks = (n + DEC_THREADS * DEC_ITEMS - 1) / (DEC_THREADS * DEC_ITEMS);
    u32* d_part = nullptr;
    NX_TRY(dev_alloc(ctx, ((size_t)n_blocks * 4 + 4) * 4, (void**)&d_part));
    u32* d_lambda = d_part + (size_t)n_blocks * 4;
    hipLaunchKernelGGL(decompose_sum_kernel, dim3(n_blocks), dim3(DEC_THREADS), 0, ctx->stream, s, n, d_part);
    hipLaunchKernelGGL(decompose_lambda_kernel, dim3(1), dim3(256), 0, ctx->stream, (const u32*)d_part, n_blocks, m_inv(n % P), d_lambda);
    hipLaunchKernelGGL(decompose_apply_kernel, dim3(grid_for(ctx, n)), dim3(256), 0, ctx->stream, s, g, n, (const u32*)d_lambda);
    hipError_t e = hipGetLastError();
    int rc = e == hipSuccess ? nx_download(ctx, lambda, d_lambda, 4) : hip_fail(ctx, e, "nx_fri_decompose", __FILE__, __LINE__);
    dev_free(ctx, d_part);
    return rc;
}

}  // extern "C"
