// The synthetic wide-Fibonacci-style machine of SURVEY.md §8(d) on device: trace fill (written
// directly in bit-reversed circle-domain order, i.e. reference prover/src/trace/utils.rs:94-106
// fused away) and constraint-quotient evaluation on the evaluation domain (the device analogue of
// stwo-constraint-framework FrameworkComponent::evaluate_constraint_quotients_on_domain, which the
// reference reaches through MachineEval::evaluate, prover/src/components/mod.rs:39-57).
#include "internal.h"
#include "air.h"
#include <algorithm>

namespace nx {

__device__ __forceinline__ u64 splitmix64(u64 x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ u32 synth_rand(u64 seed, u32 tree, u32 comp, u32 col, u32 row) {
    u64 x = splitmix64(seed ^ ((u64)tree << 60) ^ ((u64)comp << 52) ^ ((u64)col << 32) ^ (u64)row);
    u32 v = (u32)(x >> 33);
    return v == P ? 0 : v;
}

// One lane per output position i (bit-reversed circle-domain order) -> natural row -> the requested columns of one tree of one
// component.  Writes are coalesced per column.
// `col_begin`: index (within the component's tree) of the first column of `cols`.  The two free columns of every SYNTH_GROUP restart
// the (a, b) recurrence, so a column range starting inside a group replays the group from its start (at most 15 columns) without
// storing — any contiguous column range (a GPU's shard) and any position range [pos_begin, pos_begin + n_pos) (a GPU's row block;
// cols then hold n_pos words each) fill independently.
__global__ __launch_bounds__(256) void synth_fill_kernel(ColSet cols, u32 log, u32 col_begin, u32 n_cols, u32 tree, u32 comp, u64 seed, u64 inter_seed, u32 pos_begin, u32 n_pos) {
    const u32 li = blockIdx.x * blockDim.x + threadIdx.x;
    u32 N = 1u << log;
    if (li >= n_pos) return;
    const u32 i = pos_begin + li;
    u32 d = bitrev(i, log);
    u32 row = d < N / 2 ? 2 * d : 2 * N - 1 - 2 * d;
    if (tree == 0) {
        for (u32 k = 0; k < n_cols; k++) {
            const u32 g = col_begin + k;
            u32 v;
            if (g == 0) v = row == 0;
            else if (g == 1) v = row == N - 1;
            else v = m_reduce64((u64)row * (u64)(g + 1) + 7ull * g);
            cols.col(k)[li] = v;
        }
        return;
    }
    const u32 g0 = col_begin - (col_begin % SYNTH_GROUP), g1 = col_begin + n_cols;
    u32 a = 0, b = 0;
    for (u32 g = g0; g < g1; g++) {
        u32 v;
        if (tree == 1 && g < 2) {
            const u32 s0 = synth_rand(seed, 1, comp, 0, 0xFFFFFFFFu);
            if (g == 0) v = m_add(s0, row);
            else {
                const u32 s1 = synth_rand(seed, 1, comp, 1, 0xFFFFFFFFu);
                const u32 tri = (u32)((((u64)row * (u64)(row ? row - 1 : 0)) / 2) % P);
                v = m_add(m_add(s1, m_mul(row, s0)), tri);
            }
        } else {
            v = (g % SYNTH_GROUP) < 2 ? synth_rand(tree == 1 ? seed : inter_seed, tree, comp, g, row) : m_add(m_sqr(b), m_sqr(a));
        }
        if (g >= col_begin) cols.col(g - col_begin)[li] = v;
        a = b; b = v;
    }
}

// Constraint quotients of one component on its evaluation domain (log size e = log_size + d).
// Constraints, in declaration order (same as oracle/air.h):
//   (main0' - main0 - 1)(1 - is_last), (main1' - main1 - main0)(1 - is_last),
//   main_k - main_{k-1}^2 - main_{k-2}^2  for k >= 2 with k % 16 >= 2,   likewise for the interaction tree.
// row_res = Σ_j pw[j] * C_j(row);  acc[row] += row_res * denom_inv[row >> log_size].
// A column shard evaluates the constraints of ITS columns: `main_begin`/`inter_begin` are the tree indices of the first local
// column (multiples of SYNTH_GROUP, so every constraint's two predecessor columns are local), `j_main`/`j_inter` the
// declaration index of the shard's first constraint (index into pw); `head` = this shard holds main columns 0, 1 and the
// preprocessed column 1 and therefore the two transition constraints.
struct SynthShard { u32 main_begin, n_main, inter_begin, n_inter, j_main, j_inter, head; };

// Rows [row_begin, row_end) of the evaluation domain (a GPU's row block; pointers biased by the caller so the GLOBAL row indexes them).
__global__ __launch_bounds__(256) void synth_constraints_kernel(ColSet pre, ColSet mainc, ColSet inter, SynthShard sh, int log_size, int e,
                                                                const u32* __restrict__ pw /*QM31 per constraint*/,
                                                                const u32* __restrict__ denom_inv, u32* a0, u32* a1, u32* a2, u32* a3, u32 row_begin, u32 row_end) {
    u32 r = row_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= row_end) return;
    u64 r0 = 0, r1 = 0, r2 = 0, r3 = 0;  // lazy 64-bit accumulation, folded at least every 4 constraints
    u32 j = 0;
#define ACC(val)                                                           \
    {                                                                      \
        u32 v__ = (val);                                                   \
        r0 = acc_mad(r0, pw[4 * j], v__); r1 = acc_mad(r1, pw[4 * j + 1], v__); \
        r2 = acc_mad(r2, pw[4 * j + 2], v__); r3 = acc_mad(r3, pw[4 * j + 3], v__); \
        j++;                                                               \
        if ((j & 3) == 0) { r0 = acc_fold(r0); r1 = acc_fold(r1); r2 = acc_fold(r2); r3 = acc_fold(r3); } \
    }
#define REFOLD() { r0 = acc_fold(r0); r1 = acc_fold(r1); r2 = acc_fold(r2); r3 = acc_fold(r3); }
    u32 a = 0, b = 0, k = 0;
    if (sh.head) {
        // row `offset` = +1 trace step away (stwo-constraint-framework offset_bit_reversed_circle_domain_index)
        u32 rn;
        {
            u32 idx = bitrev(r, e), half = 1u << (e - 1), step = 1u << (e - log_size - 1);
            if (idx < half) idx = (idx + step) & (half - 1);
            else idx = ((idx - half - step) & (half - 1)) + half;
            rn = bitrev(idx, e);
        }
        u32 not_last = m_sub(1, gld(pre.col(1) + r));
        u32 m0 = gld(mainc.col(0) + r), m1 = gld(mainc.col(1) + r);
        u32 m0n = gld(mainc.col(0) + rn), m1n = gld(mainc.col(1) + rn);
        ACC(m_mul(m_sub(m_sub(m0n, m0), 1), not_last));
        ACC(m_mul(m_sub(m_sub(m1n, m1), m0), not_last));
        a = m0; b = m1; k = 2;
    }
    REFOLD();
    j = sh.j_main;
    // columns are read 8 at a time: the loads of a chunk are independent of the running (a, b) pair, so 8 requests are in
    // flight per lane before the first constraint of the chunk is evaluated (one request at a time caps HBM at ~3.4 TB/s)
    {
        for (; k + 8 <= sh.n_main; k += 8) {
            u32 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = gld(mainc.col(k + u) + r);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (((sh.main_begin + k + u) % SYNTH_GROUP) >= 2) ACC(m_sub(m_sub(v[u], m_sqr(b)), m_sqr(a)));
                a = b; b = v[u];
            }
        }
        for (; k < sh.n_main; k++) {
            u32 v = gld(mainc.col(k) + r);
            if (((sh.main_begin + k) % SYNTH_GROUP) >= 2) ACC(m_sub(m_sub(v, m_sqr(b)), m_sqr(a)));
            a = b; b = v;
        }
    }
    REFOLD();
    j = sh.j_inter;
    a = 0; b = 0;
    {
        u32 k = 0;
        for (; k + 8 <= sh.n_inter; k += 8) {
            u32 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = gld(inter.col(k + u) + r);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (((sh.inter_begin + k + u) % SYNTH_GROUP) >= 2) ACC(m_sub(m_sub(v[u], m_sqr(b)), m_sqr(a)));
                a = b; b = v[u];
            }
        }
        for (; k < sh.n_inter; k++) {
            u32 v = gld(inter.col(k) + r);
            if (((sh.inter_begin + k) % SYNTH_GROUP) >= 2) ACC(m_sub(m_sub(v, m_sqr(b)), m_sqr(a)));
            a = b; b = v;
        }
    }
#undef ACC
#undef REFOLD
    u32 di = denom_inv[r >> log_size];
    a0[r] = m_add(a0[r], m_mul(acc_final(r0), di)); a1[r] = m_add(a1[r], m_mul(acc_final(r1), di));
    a2[r] = m_add(a2[r], m_mul(acc_final(r2), di)); a3[r] = m_add(a3[r], m_mul(acc_final(r3), di));
}

// dst[k][i] += src[k][i]  (AccumulationOps::accumulate)
__global__ void secure_accumulate_kernel(u32* d0, u32* d1, u32* d2, u32* d3, const u32* s0, const u32* s1, const u32* s2, const u32* s3, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    d0[i] = m_add(d0[i], s0[i]); d1[i] = m_add(d1[i], s1[i]); d2[i] = m_add(d2[i], s2[i]); d3[i] = m_add(d3[i], s3[i]);
}

int synth_constraints(nx_ctx* ctx, ColSet pre, ColSet mainc, ColSet inter, const SynthRange& rg, int log_size, int e, const u32* d_pw,
                      const u32* d_denom_inv, u32* const acc4[4], u32 row_begin, u32 n) {
    if (!n) return NX_OK;
    SynthShard sh;
    sh.main_begin = rg.main_begin; sh.n_main = rg.n_main; sh.inter_begin = rg.inter_begin; sh.n_inter = rg.n_inter;
    sh.head = rg.main_begin == 0 && rg.n_main >= 2 && rg.has_pre1;
    // constraint declaration order: 2 transition constraints, then main columns k >= 2 with k % 16 >= 2, then interaction columns
    auto count_free = [](u32 lo, u32 hi) { u32 c = 0; for (u32 k = lo; k < hi; k++) if ((k % SYNTH_GROUP) >= 2) c++; return c; };
    sh.j_main = 2 + count_free(2, std::max<u32>(2, rg.main_begin));
    sh.j_inter = 2 + count_free(2, rg.n_main_total) + count_free(0, rg.inter_begin);
    hipLaunchKernelGGL(synth_constraints_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, pre, mainc, inter, sh, log_size, e,
                       d_pw, d_denom_inv, acc4[0], acc4[1], acc4[2], acc4[3], row_begin, row_begin + n);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

int secure_accumulate(nx_ctx* ctx, u32* const dst4[4], const u32* const src4[4], u32 n) {
    hipLaunchKernelGGL(secure_accumulate_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, dst4[0], dst4[1], dst4[2], dst4[3], src4[0], src4[1],
                       src4[2], src4[3], n);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

int synth_fill_range(nx_ctx* ctx, const nx_component_spec& c, uint32_t ci, uint32_t tree, uint64_t seed, uint64_t inter_seed, uint32_t col_begin,
                     uint32_t n_cols, uint32_t* const* d_cols, uint32_t pos_begin, uint32_t n_pos) {
    if (c.n_pre < 2 || c.n_main < 2 || c.log_size < 1 || c.log_size > 28) return set_err(ctx, NX_ERR_ARG, "synthetic component needs n_pre >= 2, n_main >= 2, 1 <= log_size <= 28");
    if (n_cols == 0 || n_pos == 0) return NX_OK;
    if ((uint64_t)pos_begin + n_pos > ((uint64_t)1 << c.log_size)) return set_err(ctx, NX_ERR_ARG, "synthetic fill: position range outside the column");
    ColSet cs; NX_TRY(make_colset(ctx, d_cols, n_cols, &cs));
    hipLaunchKernelGGL(synth_fill_kernel, dim3((n_pos + 255) / 256), dim3(256), 0, ctx->stream, cs, c.log_size, col_begin, n_cols, tree, ci, seed, inter_seed, pos_begin, n_pos);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

}  // namespace nx

using namespace nx;

extern "C" int nx_synth_fill_tree(nx_ctx* ctx, const nx_component_spec* comps, uint32_t n_comps, uint32_t tree, uint64_t seed, uint64_t inter_seed,
                                  uint32_t* const* d_cols) {
    NX_GUARD(ctx);
    if (tree > 2) return set_err(ctx, NX_ERR_ARG, "nx_synth_fill_tree: tree must be 0, 1 or 2");
    size_t first = 0;
    for (uint32_t ci = 0; ci < n_comps; ci++) {
        const nx_component_spec& c = comps[ci];
        uint32_t n = tree == 0 ? c.n_pre : tree == 1 ? c.n_main : c.n_inter;
        if (c.n_pre < 2 || c.n_main < 2 || c.log_size < 1 || c.log_size > 28) return set_err(ctx, NX_ERR_ARG, "synthetic component needs n_pre >= 2, n_main >= 2, 1 <= log_size <= 28");
        NX_TRY(synth_fill_range(ctx, c, ci, tree, seed, inter_seed, 0, n, d_cols + first, 0, 1u << c.log_size));
        first += n;
    }
    return NX_OK;
}
