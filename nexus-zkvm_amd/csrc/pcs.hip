// OODS evaluation, DEEP quotients and FRI folds for gfx950 (K7, K8, K9 of SURVEY.md §8(a)).
//
// Replaces Stwo `PolyOps::eval_at_point`, `QuotientOps::accumulate_quotients` and
// `FriOps::{fold_line, fold_circle_into_line}` — all reached from `stwo::prover::prove`
// (reference prover/src/machine.rs:286-290, prover2/machine/src/prove.rs:124-128).
// All three are streaming kernels bounded by HBM bandwidth (one pass over the data they touch).
#include "internal.h"
#include <algorithm>
#include <map>
#include <string.h>

namespace nx {

// ------------------------------------------------------------------ K7: eval_at_point -------
// f(p) = Σ_j c_j · basis_j(p), basis_j = Π_k factor_k^{bit_k(j)} with factors [y, x, π(x), π²(x), ...].
// Split j = (j_hi, j_lo): basis_j = T_hi[j_hi] · T_lo[j_lo].  A lane owns a fixed set of j_lo and
// walks j_hi, so T_hi[j_hi] is wave-uniform (scalar registers) and every coefficient costs one
// M31 x QM31 multiply-add; T_lo is applied once per lane at the end.
constexpr int EVAL_LOG_LO = 10;
constexpr int EVAL_THREADS = 256;
constexpr int EVAL_KL = (1 << EVAL_LOG_LO) / EVAL_THREADS;  // j_lo values per lane

__global__ __launch_bounds__(EVAL_THREADS) void eval_at_point_kernel(ColSet polys, int log, const u32* __restrict__ t_lo /*4 x 2^L*/,
                                                                     const u32* __restrict__ t_hi /*4 x n_hi, AoS*/, u32 hi_per_block,
                                                                     u32* __restrict__ partial /*[poly][chunk][4]*/, u32 n_chunks) {
    const int L = log < EVAL_LOG_LO ? log : EVAL_LOG_LO;
    const u32 n_lo = 1u << L, n_hi = 1u << (log - L);
    const u32 poly = blockIdx.y, chunk = blockIdx.x;
    const u32* __restrict__ c = polys.col(poly);
    u64 acc[EVAL_KL][4];
#pragma unroll
    for (int k = 0; k < EVAL_KL; k++) for (int q = 0; q < 4; q++) acc[k][q] = 0;
    u32 h0 = chunk * hi_per_block, h1 = min(n_hi, h0 + hi_per_block);
    u32 jh = h0;
    // 4 rows of coefficients per step: 16 independent 1 KB reads in flight per wave before the first multiply-add, one fold per step
    for (; jh + 4 <= h1; jh += 4) {
        u32 cv[4][EVAL_KL];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const u32* row = c + ((size_t)(jh + u) << L);
#pragma unroll
            for (int k = 0; k < EVAL_KL; k++) { u32 jl = threadIdx.x + k * EVAL_THREADS; cv[u][k] = jl < n_lo ? gld(row + jl) : 0u; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const u32 th0 = t_hi[4 * (jh + u)], th1 = t_hi[4 * (jh + u) + 1], th2 = t_hi[4 * (jh + u) + 2], th3 = t_hi[4 * (jh + u) + 3];  // wave-uniform
#pragma unroll
            for (int k = 0; k < EVAL_KL; k++) {
                acc[k][0] = acc_mad(acc[k][0], th0, cv[u][k]); acc[k][1] = acc_mad(acc[k][1], th1, cv[u][k]);
                acc[k][2] = acc_mad(acc[k][2], th2, cv[u][k]); acc[k][3] = acc_mad(acc[k][3], th3, cv[u][k]);
            }
        }
#pragma unroll
        for (int k = 0; k < EVAL_KL; k++) for (int q = 0; q < 4; q++) acc[k][q] = acc_fold(acc[k][q]);
    }
    for (; jh < h1; jh++) {   // < 4 leftover rows: still within one fold period
        const u32 th0 = t_hi[4 * jh], th1 = t_hi[4 * jh + 1], th2 = t_hi[4 * jh + 2], th3 = t_hi[4 * jh + 3];
        const u32* row = c + ((size_t)jh << L);
#pragma unroll
        for (int k = 0; k < EVAL_KL; k++) {
            u32 jl = threadIdx.x + k * EVAL_THREADS;
            u32 cv = jl < n_lo ? row[jl] : 0u;
            acc[k][0] = acc_mad(acc[k][0], th0, cv); acc[k][1] = acc_mad(acc[k][1], th1, cv);
            acc[k][2] = acc_mad(acc[k][2], th2, cv); acc[k][3] = acc_mad(acc[k][3], th3, cv);
        }
    }
    QM31 tot = q_zero();
#pragma unroll
    for (int k = 0; k < EVAL_KL; k++) {
        u32 jl = threadIdx.x + k * EVAL_THREADS;
        QM31 a = qm(acc_final(acc[k][0]), acc_final(acc[k][1]), acc_final(acc[k][2]), acc_final(acc[k][3]));
        if (jl < n_lo) tot = q_add(tot, q_mul(a, qm(t_lo[jl], t_lo[n_lo + jl], t_lo[2 * n_lo + jl], t_lo[3 * n_lo + jl])));
    }
    // block reduction (wave shuffle then LDS)
    __shared__ u32 red[EVAL_THREADS / 64][4];
    u32 w[4] = {tot.a.a, tot.a.b, tot.b.a, tot.b.b};
#pragma unroll
    for (int q = 0; q < 4; q++)
        for (int off = 32; off > 0; off >>= 1) w[q] = m_add(w[q], __shfl_down(w[q], off, 64));
    if ((threadIdx.x & 63) == 0) for (int q = 0; q < 4; q++) red[threadIdx.x >> 6][q] = w[q];
    __syncthreads();
    if (threadIdx.x < 4) {
        u32 s = 0;
        for (int k = 0; k < EVAL_THREADS / 64; k++) s = m_add(s, red[k][threadIdx.x]);
        partial[((size_t)poly * n_chunks + chunk) * 4 + threadIdx.x] = s;
    }
}

// ------------------------------------------------------------------ K8: DEEP quotients ------
__device__ __forceinline__ u32 get_lane4(const uint4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
struct QBatchDev {       // one ColumnSampleBatch, device-resident description
    u32 first, count;    // range in the flattened (col_idx, c) arrays
    u32 prx[2], pry[2], pix[2], piy[2];  // CM31 parts of the sample point
    u32 sum_a[4], sum_b[4];              // Σ alpha^k a_k , Σ alpha^k b_k   (QM31)
    u32 coeff[4];                        // alpha^{count}
};

// 1 / den_b(d) for the 4 domain points of a lane, den (CM31) = (Re p.x - d.x) Im p.y - (Re p.y - d.y) Im p.x: the four M31 norms are
// inverted with ONE m_inv (Montgomery's trick; a zero denominator gives 0, as c_inv does, and leaves the other three alone)
__device__ __forceinline__ void quot_den_inv4(const QBatchDev& B, const Pt dp[4], CM31 di[4]) {
    const CM31 prx = cm(B.prx[0], B.prx[1]), pry = cm(B.pry[0], B.pry[1]), pix = cm(B.pix[0], B.pix[1]), piy = cm(B.piy[0], B.piy[1]);
    CM31 den[4]; u32 nrm[4], pre[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        den[i] = c_sub(c_mul(c_sub(prx, cm(dp[i].x, 0)), piy), c_mul(c_sub(pry, cm(dp[i].y, 0)), pix));
        nrm[i] = m_add(m_sqr(den[i].a), m_sqr(den[i].b));
        const u32 nz = nrm[i] ? nrm[i] : 1u;
        pre[i] = i ? m_mul(pre[i - 1], nz) : nz;
    }
    u32 inv = m_inv(pre[3]);
#pragma unroll
    for (int i = 3; i >= 0; i--) {
        const u32 nz = nrm[i] ? nrm[i] : 1u;
        const u32 ni = i ? m_mul(inv, pre[i - 1]) : inv;
        inv = m_mul(inv, nz);
        const u32 ninv = nrm[i] ? ni : 0u;
        di[i] = cm(m_mul(den[i].a, ninv), m_mul(m_neg(den[i].b), ninv));
    }
}

// Each lane owns 4 consecutive rows (one 16-byte read per column: 1 KB contiguous per wave and column instead of 256 B, which
// is what lets ~440 concurrent column streams run near HBM speed).  In bit-reversed order the 4 domain points of rows
// 4j..4j+3 are (x, y), (x, -y), (-x, -y), (-x, y): one scalar-multiple walk for all four.
// Rows [4 j0, 4 j1) of the domain: a GPU of a row-sharded prove owns one contiguous block of the (bit-reversed) rows; column and
// output pointers are biased by the caller so that indexing with the GLOBAL row works (single GPU: j0 = 0, j1 = 2^(log-2)).
__global__ __launch_bounds__(256) void quotient_kernel(ColSet cols, int log, u32 j0, u32 j1, const QBatchDev* __restrict__ batches, u32 n_batches,
                                                       const u32* __restrict__ col_idx, const u32* __restrict__ cks /*4 per entry*/,
                                                       u32* o0, u32* o1, u32* o2, u32* o3, GenTable gen) {
    const u32 j = j0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= j1) return;
    const u32 r = 4 * j;
    Pt dp[4];
    dp[0] = pt_from_index_tbl(gen, circle_domain_index(log, bitrev(r, log)));      // a sum over the index's set bits (~log / 2 additions), not 31 doublings
    // bitrev(4j + i) = bitrev2(i) * 2^(log-2) + bitrev(j): i = 1 lands in the conjugate half of the domain, i = 2 is
    // 2^(log-2) steps of 2^(32-log) = half a turn further, i = 3 both
    dp[1].x = dp[0].x; dp[1].y = m_neg(dp[0].y);
    dp[2].x = m_neg(dp[0].x); dp[2].y = dp[1].y;
    dp[3].x = dp[2].x; dp[3].y = dp[0].y;
    QM31 acc[4] = {q_zero(), q_zero(), q_zero(), q_zero()};
    for (u32 b = 0; b < n_batches; b++) {
        const QBatchDev& B = batches[b];
        // numerators: Σ_k c_k f_k(d) - (d.y Σ a_k + Σ b_k), 4 rows x 4 coordinates, lazily reduced
        u64 n[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++) for (int q = 0; q < 4; q++) n[i][q] = 0;
        const u32 end = B.first + B.count;
        u32 k = B.first;
        for (; k + 4 <= end; k += 4) {
            uint4 f[4];
#pragma unroll
            for (int u = 0; u < 4; u++) f[u] = gld4(cols.col(col_idx[k + u]) + r);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const u32 c0 = cks[4 * (k + u)], c1 = cks[4 * (k + u) + 1], c2 = cks[4 * (k + u) + 2], c3 = cks[4 * (k + u) + 3];
                const u32 fv[4] = {f[u].x, f[u].y, f[u].z, f[u].w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    n[i][0] = acc_mad(n[i][0], c0, fv[i]); n[i][1] = acc_mad(n[i][1], c1, fv[i]);
                    n[i][2] = acc_mad(n[i][2], c2, fv[i]); n[i][3] = acc_mad(n[i][3], c3, fv[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) for (int q = 0; q < 4; q++) n[i][q] = acc_fold(n[i][q]);
        }
        for (; k < end; k++) {   // < 4 products on top of a folded value: still below 2^64
            const uint4 f = gld4(cols.col(col_idx[k]) + r);
            const u32 c0 = cks[4 * k], c1 = cks[4 * k + 1], c2 = cks[4 * k + 2], c3 = cks[4 * k + 3];
            const u32 fv[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                n[i][0] = acc_mad(n[i][0], c0, fv[i]); n[i][1] = acc_mad(n[i][1], c1, fv[i]);
                n[i][2] = acc_mad(n[i][2], c2, fv[i]); n[i][3] = acc_mad(n[i][3], c3, fv[i]);
            }
        }
        const QM31 sa = q_load(B.sum_a), sb = q_load(B.sum_b), coeff = q_load(B.coeff);
        CM31 di[4];
        quot_den_inv4(B, dp, di);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            QM31 num = q_sub(qm(acc_final(n[i][0]), acc_final(n[i][1]), acc_final(n[i][2]), acc_final(n[i][3])), q_add(q_mul_m(sa, dp[i].y), sb));
            acc[i] = q_add(q_mul(acc[i], coeff), q_mul_c(num, di[i]));
        }
    }
    *reinterpret_cast<uint4*>(o0 + r) = make_uint4(acc[0].a.a, acc[1].a.a, acc[2].a.a, acc[3].a.a);
    *reinterpret_cast<uint4*>(o1 + r) = make_uint4(acc[0].a.b, acc[1].a.b, acc[2].a.b, acc[3].a.b);
    *reinterpret_cast<uint4*>(o2 + r) = make_uint4(acc[0].b.a, acc[1].b.a, acc[2].b.a, acc[3].b.a);
    *reinterpret_cast<uint4*>(o3 + r) = make_uint4(acc[0].b.b, acc[1].b.b, acc[2].b.b, acc[3].b.b);
}

// ---- the same quotients through the COEFFICIENTS (single GPU, wide size groups) -------------------------------------------------
// A batch's numerator Σ_k c_k f_k(d) is linear in the columns, so it is the evaluation at d of ONE secure polynomial whose
// coefficients are Σ_k c_k coef_k: combining the coefficient columns (half as long as their extensions at blowup 2), extending the
// 4 coordinate columns per batch and finishing row by row reads half the bytes of the row-wise sum over the extensions — and gives the
// same field elements, the arithmetic being exact.  Lane = 4 consecutive coefficients; out: [batch][coordinate] columns of n words.
// blockIdx.y = slice: a short column (few lanes) is combined by n_slices blocks per lane group, each over its share of every batch's entries, into
// its own copy of the output (slice_stride words apart); quotient_combine_reduce_kernel adds the copies.  n_slices = 1: the whole sum, as before.
__global__ __launch_bounds__(256) void quotient_combine_kernel(ColSet polys, u32 n4, const QBatchDev* __restrict__ batches, u32 n_batches,
                                                               const u32* __restrict__ col_idx, const u32* __restrict__ cks /*4 per entry*/,
                                                               u32* __restrict__ out, size_t col_stride, u32 n_slices, size_t slice_stride) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n4) return;
    const u32 r = 4 * j;
    const u32 sl = blockIdx.y;
    out += (size_t)sl * slice_stride;
    for (u32 b = 0; b < n_batches; b++) {
        const QBatchDev& B = batches[b];
        u64 n[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++) for (int q = 0; q < 4; q++) n[i][q] = 0;
        const u32 end = B.first + (u32)(((u64)B.count * (sl + 1)) / n_slices);
        u32 k = B.first + (u32)(((u64)B.count * sl) / n_slices);
#ifndef NX_QC_NO_PIPELINE
        // the next four columns are requested before the current four are consumed (a wave no longer sits without a load in flight while it
        // multiplies; quotient stage 2.50 -> 2.47 ms in a same-box A/B of 4 alternating rounds, profiles/r06_quotient_pipeline_ab.jsonl: the kernel is at ~0.8 of its HBM floor either way)
        uint4 g[4];
        if (k + 4 <= end) {
#pragma unroll
            for (int u = 0; u < 4; u++) g[u] = gld4(polys.col(col_idx[k + u]) + r);
        }
#endif
        for (; k + 4 <= end; k += 4) {
            uint4 f[4];
#ifndef NX_QC_NO_PIPELINE
#pragma unroll
            for (int u = 0; u < 4; u++) f[u] = g[u];
            if (k + 8 <= end) {
#pragma unroll
                for (int u = 0; u < 4; u++) g[u] = gld4(polys.col(col_idx[k + 4 + u]) + r);
            }
#else
#pragma unroll
            for (int u = 0; u < 4; u++) f[u] = gld4(polys.col(col_idx[k + u]) + r);
#endif
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const u32 c0 = cks[4 * (k + u)], c1 = cks[4 * (k + u) + 1], c2 = cks[4 * (k + u) + 2], c3 = cks[4 * (k + u) + 3];
                const u32 fv[4] = {f[u].x, f[u].y, f[u].z, f[u].w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    n[i][0] = acc_mad(n[i][0], c0, fv[i]); n[i][1] = acc_mad(n[i][1], c1, fv[i]);
                    n[i][2] = acc_mad(n[i][2], c2, fv[i]); n[i][3] = acc_mad(n[i][3], c3, fv[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) for (int q = 0; q < 4; q++) n[i][q] = acc_fold(n[i][q]);
        }
        for (; k < end; k++) {
            const uint4 f = gld4(polys.col(col_idx[k]) + r);
            const u32 c0 = cks[4 * k], c1 = cks[4 * k + 1], c2 = cks[4 * k + 2], c3 = cks[4 * k + 3];
            const u32 fv[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                n[i][0] = acc_mad(n[i][0], c0, fv[i]); n[i][1] = acc_mad(n[i][1], c1, fv[i]);
                n[i][2] = acc_mad(n[i][2], c2, fv[i]); n[i][3] = acc_mad(n[i][3], c3, fv[i]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            *reinterpret_cast<uint4*>(out + (size_t)(4 * b + q) * col_stride + r) = make_uint4(acc_final(n[0][q]), acc_final(n[1][q]), acc_final(n[2][q]), acc_final(n[3][q]));
    }
}

// out[0] += out[1] + ... + out[n_slices - 1] (canonical M31 words, 4 per lane)
__global__ __launch_bounds__(256) void quotient_combine_reduce_kernel(u32* __restrict__ out, size_t n4_total, u32 n_slices, size_t slice_stride) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n4_total) return;
    uint4 a = *reinterpret_cast<const uint4*>(out + 4 * j);
    for (u32 sl = 1; sl < n_slices; sl++) {
        const uint4 v = *reinterpret_cast<const uint4*>(out + (size_t)sl * slice_stride + 4 * j);
        a.x = m_add(a.x, v.x); a.y = m_add(a.y, v.y); a.z = m_add(a.z, v.z); a.w = m_add(a.w, v.w);
    }
    *reinterpret_cast<uint4*>(out + 4 * j) = a;
}

// rows 4j .. 4j+3 from the extended combinations: num_b = G_b(d) - (d.y Σ a + Σ b), acc = acc * alpha^count_b + num_b / den_b(d).  The
// CM31 denominators of the 4 rows of a batch are inverted together (one m_inv for their 4 norms).
__global__ __launch_bounds__(256) void quotient_finish_kernel(const u32* __restrict__ ext, size_t col_stride, int log, u32 n4, const QBatchDev* __restrict__ batches, u32 n_batches,
                                                              u32* o0, u32* o1, u32* o2, u32* o3, GenTable gen) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n4) return;
    const u32 r = 4 * j;
    Pt dp[4];
    dp[0] = pt_from_index_tbl(gen, circle_domain_index(log, bitrev(r, log)));
    dp[1].x = dp[0].x; dp[1].y = m_neg(dp[0].y);
    dp[2].x = m_neg(dp[0].x); dp[2].y = dp[1].y;
    dp[3].x = dp[2].x; dp[3].y = dp[0].y;
    QM31 acc[4] = {q_zero(), q_zero(), q_zero(), q_zero()};
    for (u32 b = 0; b < n_batches; b++) {
        const QBatchDev& B = batches[b];
        uint4 g[4];
#pragma unroll
        for (int q = 0; q < 4; q++) g[q] = gld4(ext + (size_t)(4 * b + q) * col_stride + r);
        const QM31 sa = q_load(B.sum_a), sb = q_load(B.sum_b), coeff = q_load(B.coeff);
        CM31 di[4];
        quot_den_inv4(B, dp, di);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const QM31 G = qm(get_lane4(g[0], i), get_lane4(g[1], i), get_lane4(g[2], i), get_lane4(g[3], i));
            const QM31 num = q_sub(G, q_add(q_mul_m(sa, dp[i].y), sb));
            acc[i] = q_add(q_mul(acc[i], coeff), q_mul_c(num, di[i]));
        }
    }
    *reinterpret_cast<uint4*>(o0 + r) = make_uint4(acc[0].a.a, acc[1].a.a, acc[2].a.a, acc[3].a.a);
    *reinterpret_cast<uint4*>(o1 + r) = make_uint4(acc[0].a.b, acc[1].a.b, acc[2].a.b, acc[3].a.b);
    *reinterpret_cast<uint4*>(o2 + r) = make_uint4(acc[0].b.a, acc[1].b.a, acc[2].b.a, acc[3].b.a);
    *reinterpret_cast<uint4*>(o3 + r) = make_uint4(acc[0].b.b, acc[1].b.b, acc[2].b.b, acc[3].b.b);
}

// log_size < 2: one lane per row
__global__ void quotient_small_kernel(ColSet cols, int log, u32 r0, u32 r1, const QBatchDev* __restrict__ batches, u32 n_batches, const u32* __restrict__ col_idx,
                                      const u32* __restrict__ cks, u32* o0, u32* o1, u32* o2, u32* o3) {
    u32 r = r0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= r1) return;
    Pt dp = pt_from_index(circle_domain_index(log, bitrev(r, log)));
    QM31 acc = q_zero();
    for (u32 b = 0; b < n_batches; b++) {
        const QBatchDev& B = batches[b];
        QM31 num = q_zero();
        for (u32 k = B.first; k < B.first + B.count; k++) num = q_add(num, q_mul_m(q_load(cks + 4 * k), cols.col(col_idx[k])[r]));
        num = q_sub(num, q_add(q_mul_m(q_load(B.sum_a), dp.y), q_load(B.sum_b)));
        CM31 den = c_sub(c_mul(c_sub(cm(B.prx[0], B.prx[1]), cm(dp.x, 0)), cm(B.piy[0], B.piy[1])),
                         c_mul(c_sub(cm(B.pry[0], B.pry[1]), cm(dp.y, 0)), cm(B.pix[0], B.pix[1])));
        acc = q_add(q_mul(acc, q_load(B.coeff)), q_mul_c(num, c_inv(den)));
    }
    o0[r] = acc.a.a; o1[r] = acc.a.b; o2[r] = acc.b.a; o3[r] = acc.b.b;
}

// ------------------------------------------------------------------ K9: FRI folds -----------
__device__ __forceinline__ u32 tw_at(const u32* tw, u32 tw_log, int n, int layer, u32 h) {
    return tw[(1u << tw_log) - (1u << (n - layer)) + h];
}
// 1/y of CanonicCoset(L).circle_domain().at(bitrev_L(2i)) == circle-layer inverse twiddle i of a size-2^L domain
__device__ __forceinline__ u32 circle_itw(const u32* itw, u32 tw_log, int L, u32 h) {
    if (L == 1) return m_inv(pt_from_index(half_odds_index(0, 0)).y);
    if (L == 2) { Pt p = pt_from_index(half_odds_index(1, 0)); u32 yi = m_inv(p.y); return h ? m_neg(yi) : yi; }
    u32 c = h >> 2;
    u32 x = tw_at(itw, tw_log, L, 1, 2 * c), y = tw_at(itw, tw_log, L, 1, 2 * c + 1);
    u32 sel = h & 3;
    u32 v = (sel & 2) ? x : y;
    return (sel == 1 || sel == 2) ? m_neg(v) : v;
}

struct Sec4 { u32* c[4]; };
struct Sec4C { const u32* c[4]; };

// The fold kernels work on outputs [i0, i1) (a row block of a row-sharded prove; pointers biased by the caller so that the GLOBAL
// index addresses them; single GPU: i0 = 0, i1 = 2^(L-1)).
__global__ void fold_circle_kernel(Sec4 dst, Sec4C src, int L, u32 i0, u32 i1, const u32* __restrict__ itw, u32 tw_log, QM31 alpha, QM31 alpha_sq) {
    u32 i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= i1) return;
    u32 yi = circle_itw(itw, tw_log, L, i);
    QM31 f0, f1;
    {
        uint2 a = *reinterpret_cast<const uint2*>(src.c[0] + 2 * i), b = *reinterpret_cast<const uint2*>(src.c[1] + 2 * i);
        uint2 c = *reinterpret_cast<const uint2*>(src.c[2] + 2 * i), d = *reinterpret_cast<const uint2*>(src.c[3] + 2 * i);
        f0 = qm(a.x, b.x, c.x, d.x); f1 = qm(a.y, b.y, c.y, d.y);
    }
    QM31 s = q_add(f0, f1), t = q_mul_m(q_sub(f0, f1), yi);   // ibutterfly
    QM31 fp = q_add(q_mul(alpha, t), s);
    QM31 d = qm(dst.c[0][i], dst.c[1][i], dst.c[2][i], dst.c[3][i]);
    d = q_add(q_mul(d, alpha_sq), fp);
    dst.c[0][i] = d.a.a; dst.c[1][i] = d.a.b; dst.c[2][i] = d.b.a; dst.c[3][i] = d.b.b;
}

// line domain of log size L is Coset::half_odds(L); 1/x at bitrev_L(2i) is itwiddle layer (H-L), entry i
__global__ void fold_line_kernel(Sec4 dst, Sec4C src, int L, u32 i0, u32 i1, const u32* __restrict__ itw, u32 tw_log, QM31 alpha) {
    u32 i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= i1) return;
    u32 xi = itw[(1u << tw_log) - (1u << L) + i];
    uint2 a = *reinterpret_cast<const uint2*>(src.c[0] + 2 * i), b = *reinterpret_cast<const uint2*>(src.c[1] + 2 * i);
    uint2 c = *reinterpret_cast<const uint2*>(src.c[2] + 2 * i), d = *reinterpret_cast<const uint2*>(src.c[3] + 2 * i);
    QM31 f0 = qm(a.x, b.x, c.x, d.x), f1 = qm(a.y, b.y, c.y, d.y);
    QM31 s = q_add(f0, f1), t = q_mul_m(q_sub(f0, f1), xi);
    QM31 o = q_add(s, q_mul(alpha, t));
    dst.c[0][i] = o.a.a; dst.c[1][i] = o.a.b; dst.c[2][i] = o.b.a; dst.c[3][i] = o.b.b;
}

// The evaluation tables of eval_at_point_kernel built on the device: entry j of a table is the product of the factors of the set
// bits of j (factors [y, x, pi(x), pi^2(x), ...] of the point; low table: bits [0, L), coordinate-major; high table: bits [L, n),
// 4 words per entry).  The host sends the <= 30 factors as a kernel argument instead of computing and copying 2^L + 2^(n-L) products.
struct EvalFactors { QM31 f[30]; };
__global__ __launch_bounds__(256) void eval_tables_kernel(EvalFactors F, int n, int L, u32* __restrict__ t_lo, u32* __restrict__ t_hi) {
    const u32 n_lo = 1u << L, n_hi = 1u << (n - L);
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_lo) {
        QM31 t = q_one();
        for (int k = 0; k < L; k++) if ((i >> k) & 1) t = q_mul(t, F.f[k]);
        t_lo[i] = t.a.a; t_lo[n_lo + i] = t.a.b; t_lo[2 * n_lo + i] = t.b.a; t_lo[3 * n_lo + i] = t.b.b;
    } else if (i - n_lo < n_hi) {
        const u32 j = i - n_lo;
        QM31 t = q_one();
        for (int k = L; k < n; k++) if ((j >> (k - L)) & 1) t = q_mul(t, F.f[k]);
        t_hi[4 * j] = t.a.a; t_hi[4 * j + 1] = t.a.b; t_hi[4 * j + 2] = t.b.a; t_hi[4 * j + 3] = t.b.b;
    }
}

// The same folds with the folding alpha read from device memory (4 words): the FRI commit phase keeps the channel on the device
// (merkle.hip: fri_channel_step / fri_tail), so the host never waits for a root to draw the next alpha.
__global__ void fold_circle_dev_kernel(Sec4 dst, Sec4C src, int L, const u32* __restrict__ itw, u32 tw_log, const u32* __restrict__ alpha_ptr) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;   // device-channel path: single GPU, whole layer
    if (i >= (1u << (L - 1))) return;
    const QM31 alpha = qm(alpha_ptr[0], alpha_ptr[1], alpha_ptr[2], alpha_ptr[3]), alpha_sq = q_sqr(alpha);
    u32 yi = circle_itw(itw, tw_log, L, i);
    uint2 a = *reinterpret_cast<const uint2*>(src.c[0] + 2 * i), b = *reinterpret_cast<const uint2*>(src.c[1] + 2 * i);
    uint2 c = *reinterpret_cast<const uint2*>(src.c[2] + 2 * i), d4 = *reinterpret_cast<const uint2*>(src.c[3] + 2 * i);
    QM31 f0 = qm(a.x, b.x, c.x, d4.x), f1 = qm(a.y, b.y, c.y, d4.y);
    QM31 s = q_add(f0, f1), t = q_mul_m(q_sub(f0, f1), yi);
    QM31 fp = q_add(q_mul(alpha, t), s);
    QM31 d = qm(dst.c[0][i], dst.c[1][i], dst.c[2][i], dst.c[3][i]);
    d = q_add(q_mul(d, alpha_sq), fp);
    dst.c[0][i] = d.a.a; dst.c[1][i] = d.a.b; dst.c[2][i] = d.b.a; dst.c[3][i] = d.b.b;
}
__global__ void fold_line_dev_kernel(Sec4 dst, Sec4C src, int L, const u32* __restrict__ itw, u32 tw_log, const u32* __restrict__ alpha_ptr) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << (L - 1))) return;
    const QM31 alpha = qm(alpha_ptr[0], alpha_ptr[1], alpha_ptr[2], alpha_ptr[3]);
    u32 xi = itw[(1u << tw_log) - (1u << L) + i];
    uint2 a = *reinterpret_cast<const uint2*>(src.c[0] + 2 * i), b = *reinterpret_cast<const uint2*>(src.c[1] + 2 * i);
    uint2 c = *reinterpret_cast<const uint2*>(src.c[2] + 2 * i), d = *reinterpret_cast<const uint2*>(src.c[3] + 2 * i);
    QM31 f0 = qm(a.x, b.x, c.x, d.x), f1 = qm(a.y, b.y, c.y, d.y);
    QM31 s = q_add(f0, f1), t = q_mul_m(q_sub(f0, f1), xi);
    QM31 o = q_add(s, q_mul(alpha, t));
    dst.c[0][i] = o.a.a; dst.c[1][i] = o.a.b; dst.c[2][i] = o.b.a; dst.c[3][i] = o.b.b;
}

int fold_circle_dev(nx_ctx* ctx, const nx_twiddles* tw, uint32_t* const* d_dst4, const uint32_t* const* d_src4, uint32_t src_log, const uint32_t* d_alpha) {
    if (src_log < 1 || (src_log >= 3 && src_log - 1 > tw->log_half)) return set_err(ctx, NX_ERR_ARG, "fold_circle_dev: bad log");
    Sec4 d; Sec4C s;
    for (int q = 0; q < 4; q++) { d.c[q] = d_dst4[q]; s.c[q] = d_src4[q]; }
    uint32_t n = 1u << (src_log - 1);
    hipLaunchKernelGGL(fold_circle_dev_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d, s, (int)src_log, tw->d_itw, tw->log_half, d_alpha);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}
int fold_line_dev(nx_ctx* ctx, const nx_twiddles* tw, const uint32_t* const* d_src4, uint32_t src_log, const uint32_t* d_alpha, uint32_t* const* d_dst4) {
    if (src_log < 1 || src_log > tw->log_half) return set_err(ctx, NX_ERR_ARG, "fold_line_dev: domain not covered by the twiddle tree");
    Sec4 d; Sec4C s;
    for (int q = 0; q < 4; q++) { d.c[q] = d_dst4[q]; s.c[q] = d_src4[q]; }
    uint32_t n = 1u << (src_log - 1);
    hipLaunchKernelGGL(fold_line_dev_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d, s, (int)src_log, tw->d_itw, tw->log_half, d_alpha);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}


// Row-block forms for a row-sharded prove: dst holds outputs [i0, i0 + n) of the folded layer, src the matching 2n inputs
// [2 i0, 2 i0 + 2n) (pairs are adjacent in bit-reversed order, so a fold never leaves its block).
int fold_circle_rows(nx_ctx* ctx, const nx_twiddles* tw, uint32_t* const* d_dst4, const uint32_t* const* d_src4, uint32_t src_log, const uint32_t alpha[4], uint32_t i0, uint32_t n) {
    if (src_log < 1 || (src_log >= 3 && src_log - 1 > tw->log_half) || (uint64_t)i0 + n > ((uint64_t)1 << (src_log - 1))) return set_err(ctx, NX_ERR_ARG, "fold_circle_rows: bad range");
    if (!n) return NX_OK;
    Sec4 d; Sec4C s;
    for (int q = 0; q < 4; q++) { d.c[q] = bias_rows(d_dst4[q], i0); s.c[q] = bias_rows(d_src4[q], 2 * (uint64_t)i0); }
    const QM31 a = q_load(alpha);
    hipLaunchKernelGGL(fold_circle_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d, s, (int)src_log, i0, i0 + n, tw->d_itw, tw->log_half, a, q_sqr(a));
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}
int fold_line_rows(nx_ctx* ctx, const nx_twiddles* tw, const uint32_t* const* d_src4, uint32_t src_log, const uint32_t alpha[4], uint32_t* const* d_dst4, uint32_t i0, uint32_t n) {
    if (src_log < 1 || src_log > tw->log_half || (uint64_t)i0 + n > ((uint64_t)1 << (src_log - 1))) return set_err(ctx, NX_ERR_ARG, "fold_line_rows: bad range");
    if (!n) return NX_OK;
    Sec4 d; Sec4C s;
    for (int q = 0; q < 4; q++) { d.c[q] = bias_rows(d_dst4[q], i0); s.c[q] = bias_rows(d_src4[q], 2 * (uint64_t)i0); }
    hipLaunchKernelGGL(fold_line_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d, s, (int)src_log, i0, i0 + n, tw->d_itw, tw->log_half, q_load(alpha));
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}
}  // namespace nx

using namespace nx;

extern "C" {

int nx_eval_at_points(nx_ctx* ctx, const uint32_t* const* d_polys, uint32_t log_size, const uint32_t* poly_idx,
                      const uint32_t* h_points, uint32_t n_evals, uint32_t* h_out) {
    NX_GUARD(ctx);
    std::vector<EvalJob> jobs;
    int rc = eval_at_points_enqueue(ctx, d_polys, log_size, poly_idx, h_points, n_evals, h_out, &jobs);
    int rc2 = eval_at_points_collect(ctx, &jobs);
    return rc != NX_OK ? rc : rc2;
}

}  // extern "C"

namespace nx {

int eval_at_points_enqueue(nx_ctx* ctx, const uint32_t* const* d_polys, uint32_t log_size, const uint32_t* poly_idx, const uint32_t* h_points, uint32_t n_evals,
                           uint32_t* h_out, std::vector<EvalJob>* jobs) {
    if (n_evals == 0) return NX_OK;
    if (log_size > 30) return set_err(ctx, NX_ERR_ARG, "nx_eval_at_points: log_size too large");
    const int n = (int)log_size;
    // group by distinct point (insertion order)
    std::vector<std::vector<uint32_t>> groups;
    std::vector<const uint32_t*> gpts;
    for (uint32_t i = 0; i < n_evals; i++) {
        size_t g = 0;
        for (; g < gpts.size(); g++) if (!memcmp(gpts[g], h_points + 8 * i, 32)) break;
        if (g == gpts.size()) { gpts.push_back(h_points + 8 * i); groups.emplace_back(); }
        groups[g].push_back(i);
    }
    const int L = std::min(n, EVAL_LOG_LO);
    const uint32_t n_lo = 1u << L, n_hi = 1u << (n - L);
    for (size_t g = 0; g < groups.size(); g++) {
        QPt p; p.x = q_load(gpts[g]); p.y = q_load(gpts[g] + 4);
        // factors for bit k of j: [y, x, pi(x), pi^2(x), ...]
        EvalFactors F;
        for (int k = 0; k < 30; k++) F.f[k] = q_one();
        if (n > 0) F.f[0] = p.y;
        { QM31 x = p.x; for (int k = 1; k < n; k++) { F.f[k] = x; x = q_double_x(x); } }
        const size_t tlo_words = 4 * (size_t)n_lo, thi_words = 4 * (size_t)n_hi;
        const uint32_t np = (uint32_t)groups[g].size();
        std::vector<const uint32_t*> sel(np);
        for (uint32_t i = 0; i < np; i++) sel[i] = d_polys[poly_idx[groups[g][i]]];
        uint32_t hi_per_block = std::max<uint32_t>(1, std::min<uint32_t>(n_hi, 128));
        uint32_t n_chunks = (n_hi + hi_per_block - 1) / hi_per_block;
        const size_t part_words = (size_t)np * n_chunks * 4;
        EvalJob job; job.blob = nullptr; job.np = np; job.n_chunks = n_chunks; job.evals = groups[g]; job.h_out = h_out;
        NX_TRY(dev_alloc(ctx, (tlo_words + thi_words + part_words) * 4, (void**)&job.blob));
        uint32_t* d_lo = (uint32_t*)job.blob; uint32_t* d_hi = d_lo + tlo_words; uint32_t* d_part = d_hi + thi_words;
        void* d_tab = nullptr; void* h_part = nullptr;
        int rc = stage(ctx, sel.data(), (size_t)np * 8, &d_tab);                       // pointer table through the pinned ring
        if (rc == NX_OK) rc = host_alloc(ctx, part_words * 4, &h_part);   // owned by the job until collect: later stage() calls may wrap the ring
        if (rc != NX_OK) { dev_free(ctx, job.blob); return rc; }
        job.h_part = (const uint32_t*)h_part;
        hipLaunchKernelGGL(eval_tables_kernel, dim3((n_lo + n_hi + 255) / 256), dim3(256), 0, ctx->stream, F, n, L, d_lo, d_hi);
        hipError_t e = hipGetLastError();
        for (uint32_t p0 = 0; p0 < np && e == hipSuccess; p0 += 32768) {
            uint32_t nb = std::min<uint32_t>(32768, np - p0);
            ColSet sub; sub.base = nullptr; sub.stride = 0; sub.table = (uint32_t* const*)d_tab + p0;
            hipLaunchKernelGGL(eval_at_point_kernel, dim3(n_chunks, nb), dim3(EVAL_THREADS), 0, ctx->stream, sub, n, d_lo, d_hi,
                               hi_per_block, d_part + (size_t)p0 * n_chunks * 4, n_chunks);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(h_part, d_part, part_words * 4, hipMemcpyDeviceToHost, ctx->stream);   // pinned target: asynchronous
        jobs->push_back(std::move(job));
        if (e != hipSuccess) return hip_fail(ctx, e, "nx_eval_at_points", __FILE__, __LINE__);
    }
    return NX_OK;
}

int eval_at_points_collect(nx_ctx* ctx, std::vector<EvalJob>* jobs) {
    if (jobs->empty()) return NX_OK;
    hipError_t e = hipStreamSynchronize(ctx->stream);
    for (auto& j : *jobs) {
        if (e == hipSuccess)
            for (uint32_t i = 0; i < j.np; i++) {
                u32 s4[4] = {0, 0, 0, 0};
                for (uint32_t c = 0; c < j.n_chunks; c++) for (int q = 0; q < 4; q++) s4[q] = m_add(s4[q], j.h_part[((size_t)i * j.n_chunks + c) * 4 + q]);
                memcpy(j.h_out + 4 * (size_t)j.evals[i], s4, 16);
            }
        dev_free(ctx, j.blob);
        host_free(ctx, (void*)j.h_part);
    }
    jobs->clear();
    if (e != hipSuccess) return hip_fail(ctx, e, "nx_eval_at_points(sync)", __FILE__, __LINE__);
    return NX_OK;
}

}  // namespace nx

using namespace nx;

extern "C" {

// `entry_local` (NULL = all): which flattened (column, value) entries this GPU holds the column of; the others only advance
// the alpha powers.  `include_line`: add the -(a·y + b) line terms of ALL entries (exactly one GPU of a column-sharded prove
// does, so that the partial accumulations of the GPUs sum to the full quotient).
// row_begin / n_rows: the block of (bit-reversed) rows this call computes — the whole domain on one GPU, one contiguous block per GPU
// in a row-sharded prove; d_cols and d_out4 then point at the BLOCK (n_rows words each).
// quotient_constants + ColumnSampleBatch descriptors of one size group (host): per batch the point, Σ alpha^k a_k, Σ alpha^k b_k and
// alpha^count; per entry the column index and c_k = alpha^k (conj(p.y) - p.y)
static int quotient_descriptors(nx_ctx* ctx, uint32_t n_cols, const uint32_t random_coeff[4], uint32_t n_batches, const uint32_t* points, const uint32_t* batch_counts,
                                const uint32_t* col_idx, const uint32_t* values, const uint8_t* entry_local, int include_line,
                                std::vector<QBatchDev>* hb_out, std::vector<uint32_t>* cks_out, std::vector<uint32_t>* lidx_out) {
    QM31 alpha = q_load(random_coeff);
    size_t total = 0;
    for (uint32_t b = 0; b < n_batches; b++) total += batch_counts[b];
    std::vector<QBatchDev>& hb = *hb_out; hb.assign(n_batches, QBatchDev());
    std::vector<uint32_t>& cks = *cks_out; cks.clear(); cks.reserve(4 * total);
    std::vector<uint32_t>& lidx = *lidx_out; lidx.clear(); lidx.reserve(total);
    size_t k = 0;
    for (uint32_t b = 0; b < n_batches; b++) {
        QBatchDev& B = hb[b];
        QPt p; p.x = q_load(points + 8 * b); p.y = q_load(points + 8 * b + 4);
        B.first = (uint32_t)lidx.size();
        B.prx[0] = p.x.a.a; B.prx[1] = p.x.a.b; B.pix[0] = p.x.b.a; B.pix[1] = p.x.b.b;
        B.pry[0] = p.y.a.a; B.pry[1] = p.y.a.b; B.piy[0] = p.y.b.a; B.piy[1] = p.y.b.b;
        QM31 a_pow = q_one(), sa = q_zero(), sb = q_zero();
        QM31 c0 = q_sub(q_conj(p.y), p.y);  // conj(p.y) - p.y
        for (uint32_t j = 0; j < batch_counts[b]; j++, k++) {
            a_pow = q_mul(a_pow, alpha);
            QM31 v = q_load(values + 4 * k);
            if (include_line) {
                QM31 a = q_sub(q_conj(v), v);
                QM31 bb = q_sub(q_mul(v, c0), q_mul(a, p.y));
                sa = q_add(sa, q_mul(a_pow, a));
                sb = q_add(sb, q_mul(a_pow, bb));
            }
            if (!entry_local || entry_local[k]) {
                if (col_idx[k] >= n_cols) return set_err(ctx, NX_ERR_ARG, "nx_accumulate_quotients: column index out of range");
                lidx.push_back(col_idx[k]);
                uint32_t w[4]; q_store(w, q_mul(a_pow, c0));
                cks.insert(cks.end(), w, w + 4);
            }
        }
        B.count = (uint32_t)lidx.size() - B.first;
        q_store(B.sum_a, sa); q_store(B.sum_b, sb); q_store(B.coeff, a_pow);
    }
    return NX_OK;
}

static int accumulate_quotients_impl(nx_ctx* ctx, uint32_t log_size, const uint32_t* const* d_cols, uint32_t n_cols, const uint32_t random_coeff[4],
                                     uint32_t n_batches, const uint32_t* points, const uint32_t* batch_counts, const uint32_t* col_idx,
                                     const uint32_t* values, const uint8_t* entry_local, int include_line, uint32_t* const* d_out4,
                                     uint64_t row_begin, uint64_t n_rows) {
    if (log_size < 1 || log_size > 30) return set_err(ctx, NX_ERR_ARG, "nx_accumulate_quotients: bad log_size");
    if (row_begin + n_rows > ((uint64_t)1 << log_size) || (log_size >= 2 && ((row_begin | n_rows) & 3))) return set_err(ctx, NX_ERR_ARG, "nx_accumulate_quotients: row block outside the domain or not a multiple of 4 rows");
    std::vector<QBatchDev> hb; std::vector<uint32_t> cks, lidx;
    NX_TRY(quotient_descriptors(ctx, n_cols, random_coeff, n_batches, points, batch_counts, col_idx, values, entry_local, include_line, &hb, &cks, &lidx));
    const size_t n_local = lidx.size();
    // descriptors, column indices, coefficients and the column pointer table travel in ONE stream-ordered copy through the pinned
    // staging ring (valid until the ring wraps, which synchronises): no allocation, no synchronisation per size group
    size_t bytes_b = hb.size() * sizeof(QBatchDev), bytes_i = n_local * 4, bytes_c = cks.size() * 4, bytes_t = (size_t)n_cols * 8;
    size_t off_i = (bytes_b + 15) & ~(size_t)15, off_c = off_i + ((bytes_i + 15) & ~(size_t)15), off_t = off_c + ((bytes_c + 15) & ~(size_t)15);
    std::vector<uint8_t> host(off_t + bytes_t + 16, 0);
    if (bytes_b) memcpy(host.data(), hb.data(), bytes_b);
    if (bytes_i) memcpy(host.data() + off_i, lidx.data(), bytes_i);
    if (bytes_c) memcpy(host.data() + off_c, cks.data(), bytes_c);
    for (uint32_t c = 0; c < n_cols; c++) { const uint32_t* biased = bias_rows(d_cols[c], row_begin); memcpy(host.data() + off_t + 8 * (size_t)c, &biased, 8); }   // kernels index with the global row
    void* staged = nullptr;
    NX_TRY(stage(ctx, host.data(), host.size(), &staged));
    const uint8_t* blob = (const uint8_t*)staged;
    ColSet cs; cs.base = nullptr; cs.stride = 0; cs.table = (uint32_t* const*)(blob + off_t);
    uint64_t alg = ((uint64_t)n_cols * 4 + 16) * n_rows;
    KTimer timer(ctx, NX_T_QUOT, alg);
    if (n_rows == 0) return NX_OK;
    u32* o[4]; for (int q = 0; q < 4; q++) o[q] = bias_rows(d_out4[q], row_begin);
    if (log_size >= 2)
        hipLaunchKernelGGL(quotient_kernel, dim3((unsigned)((n_rows / 4 + 255) / 256)), dim3(256), 0, ctx->stream, cs, (int)log_size, (u32)(row_begin / 4), (u32)((row_begin + n_rows) / 4),
                           (const QBatchDev*)blob, n_batches, (const u32*)(blob + off_i), (const u32*)(blob + off_c), o[0], o[1], o[2], o[3], gen_table());
    else
        hipLaunchKernelGGL(quotient_small_kernel, dim3(1), dim3(64), 0, ctx->stream, cs, (int)log_size, (u32)row_begin, (u32)(row_begin + n_rows), (const QBatchDev*)blob, n_batches,
                           (const u32*)(blob + off_i), (const u32*)(blob + off_c), o[0], o[1], o[2], o[3]);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

}  // extern "C"
namespace nx {
int accumulate_quotients_rows(nx_ctx* ctx, uint32_t log_size, const uint32_t* const* d_cols, uint32_t n_cols, const uint32_t random_coeff[4], uint32_t n_batches,
                              const uint32_t* points, const uint32_t* batch_counts, const uint32_t* col_idx, const uint32_t* values, uint32_t* const* d_out4,
                              uint64_t row_begin, uint64_t n_rows) {
    return accumulate_quotients_impl(ctx, log_size, d_cols, n_cols, random_coeff, n_batches, points, batch_counts, col_idx, values, nullptr, 1, d_out4, row_begin, n_rows);
}

// The quotients of one size group from the COEFFICIENT columns (d_polys: 2^log_coef words each, the polynomials whose extensions on
// 2^log_size points the row-wise path reads): combine per batch, extend the 4 n_batches combinations, finish row by row.  Same values
// as accumulate_quotients_rows on the extensions (exact arithmetic), about half the bytes at blowup 2.
int accumulate_quotients_coeffs(nx_ctx* ctx, const nx_twiddles* tw, uint32_t log_size, uint32_t log_coef, const uint32_t* const* d_polys, uint32_t n_cols,
                                const uint32_t random_coeff[4], uint32_t n_batches, const uint32_t* points, const uint32_t* batch_counts, const uint32_t* col_idx,
                                const uint32_t* values, uint32_t* const* d_out4) {
    if (log_coef < 2 || log_size <= log_coef || log_size > 30 || n_batches == 0) return set_err(ctx, NX_ERR_ARG, "accumulate_quotients_coeffs: bad shape");
    std::vector<QBatchDev> hb; std::vector<uint32_t> cks, lidx;
    NX_TRY(quotient_descriptors(ctx, n_cols, random_coeff, n_batches, points, batch_counts, col_idx, values, nullptr, 1, &hb, &cks, &lidx));
    size_t bytes_b = hb.size() * sizeof(QBatchDev), bytes_i = lidx.size() * 4, bytes_c = cks.size() * 4, bytes_t = (size_t)n_cols * 8;
    size_t off_i = (bytes_b + 15) & ~(size_t)15, off_c = off_i + ((bytes_i + 15) & ~(size_t)15), off_t = off_c + ((bytes_c + 15) & ~(size_t)15);
    std::vector<uint8_t> host(off_t + bytes_t + 16, 0);
    memcpy(host.data(), hb.data(), bytes_b);
    if (bytes_i) memcpy(host.data() + off_i, lidx.data(), bytes_i);
    if (bytes_c) memcpy(host.data() + off_c, cks.data(), bytes_c);
    memcpy(host.data() + off_t, d_polys, bytes_t);
    void* staged = nullptr;
    NX_TRY(stage(ctx, host.data(), host.size(), &staged));
    const uint8_t* blob = (const uint8_t*)staged;
    ColSet cs; cs.base = nullptr; cs.stride = 0; cs.table = (uint32_t* const*)(blob + off_t);
    const uint32_t nq = 4 * n_batches;
    const size_t nc = (size_t)1 << log_coef, ne = (size_t)1 << log_size;
    uint32_t* comb = nullptr; uint32_t* ext = nullptr;
    // short columns: nc / 4 lanes do not fill the chip and each walks every column — the entries are split over up to 16 slices (same sums: M31
    // addition does not care how they are grouped; profiles/r06_quotient_slices_ab.jsonl)
    const uint32_t n_slices = (uint32_t)std::min<size_t>(16, std::max<size_t>(1, ((size_t)1 << 18) / (nc / 4)));
    const size_t slice_stride = (size_t)nq * nc;
    NX_TRY(dev_alloc(ctx, slice_stride * n_slices * 4, (void**)&comb));
    { int rc = dev_alloc(ctx, (size_t)nq * ne * 4, (void**)&ext); if (rc != NX_OK) { dev_free(ctx, comb); return rc; } }
    int rc = NX_OK;
    {
        KTimer timer(ctx, NX_T_QUOT, ((uint64_t)n_cols * 4 + 16) * ne);     // the same algorithmic bytes as the row-wise path: what the stage computes
        hipLaunchKernelGGL(quotient_combine_kernel, dim3((unsigned)((nc / 4 + 255) / 256), n_slices), dim3(256), 0, ctx->stream, cs, (u32)(nc / 4), (const QBatchDev*)blob, n_batches,
                           (const u32*)(blob + off_i), (const u32*)(blob + off_c), comb, nc, n_slices, slice_stride);
        if (n_slices > 1)
            hipLaunchKernelGGL(quotient_combine_reduce_kernel, dim3((unsigned)((slice_stride / 4 + 255) / 256)), dim3(256), 0, ctx->stream, comb, slice_stride / 4, n_slices, slice_stride);
        if (hipGetLastError() != hipSuccess) rc = set_err(ctx, NX_ERR_HIP, "quotient_combine_kernel launch failed");
    }
    if (rc == NX_OK) {
        std::vector<const uint32_t*> src(nq); std::vector<uint32_t*> dst(nq);
        for (uint32_t q = 0; q < nq; q++) { src[q] = comb + (size_t)q * nc; dst[q] = ext + (size_t)q * ne; }
        rc = nx_evaluate_batch(ctx, tw, src.data(), nq, log_coef, log_size - log_coef, dst.data());
    }
    if (rc == NX_OK) {
        // the stage's second kernel, under the same kind (its bytes are in the span above: the stage's algorithmic bytes are counted once).
        // The extension in between is Circle-FFT work and is booked as such, bytes and time alike (fft_evaluate's own span).
        KTimer timer(ctx, NX_T_QUOT, 0);
        hipLaunchKernelGGL(quotient_finish_kernel, dim3((unsigned)((ne / 4 + 255) / 256)), dim3(256), 0, ctx->stream, (const u32*)ext, ne, (int)log_size, (u32)(ne / 4),
                           (const QBatchDev*)blob, n_batches, d_out4[0], d_out4[1], d_out4[2], d_out4[3], gen_table());
        if (hipGetLastError() != hipSuccess) rc = set_err(ctx, NX_ERR_HIP, "quotient_finish_kernel launch failed");
    }
    dev_free(ctx, comb); dev_free(ctx, ext);      // stream-ordered: reused only by later work on this stream
    return rc;
}
}  // namespace nx
extern "C" {

int nx_accumulate_quotients(nx_ctx* ctx, uint32_t log_size, const uint32_t* const* d_cols, uint32_t n_cols, const uint32_t random_coeff[4],
                            uint32_t n_batches, const uint32_t* points, const uint32_t* batch_counts, const uint32_t* col_idx,
                            const uint32_t* values, uint32_t* const* d_out4) {
    NX_GUARD(ctx);
    return accumulate_quotients_impl(ctx, log_size, d_cols, n_cols, random_coeff, n_batches, points, batch_counts, col_idx, values, nullptr, 1, d_out4, 0, (uint64_t)1 << log_size);
}

int nx_accumulate_quotients_partial(nx_ctx* ctx, uint32_t log_size, const uint32_t* const* d_cols, uint32_t n_cols, const uint32_t random_coeff[4],
                                    uint32_t n_batches, const uint32_t* points, const uint32_t* batch_counts, const uint32_t* col_idx,
                                    const uint32_t* values, const uint8_t* entry_local, int include_line_terms, uint32_t* const* d_out4) {
    NX_GUARD(ctx);
    return accumulate_quotients_impl(ctx, log_size, d_cols, n_cols, random_coeff, n_batches, points, batch_counts, col_idx, values, entry_local,
                                     include_line_terms, d_out4, 0, (uint64_t)1 << log_size);
}

int nx_fold_circle_into_line(nx_ctx* ctx, const nx_twiddles* tw, uint32_t* const* d_dst4, const uint32_t* const* d_src4, uint32_t src_log,
                             const uint32_t alpha[4]) {
    NX_GUARD(ctx);
    if (src_log < 1 || (src_log >= 3 && src_log - 1 > tw->log_half)) return set_err(ctx, NX_ERR_ARG, "nx_fold_circle_into_line: bad log");
    Sec4 d; Sec4C s;
    for (int q = 0; q < 4; q++) { d.c[q] = d_dst4[q]; s.c[q] = d_src4[q]; }
    QM31 a = q_load(alpha);
    uint32_t n = 1u << (src_log - 1);
    hipLaunchKernelGGL(fold_circle_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d, s, (int)src_log, 0u, n, tw->d_itw, tw->log_half, a, q_sqr(a));
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

int nx_fold_line(nx_ctx* ctx, const nx_twiddles* tw, const uint32_t* const* d_src4, uint32_t src_log, uint32_t n_doublings, const uint32_t alpha[4],
                 uint32_t* const* d_dst4) {
    NX_GUARD(ctx);
    (void)n_doublings;  // half_odds(k).double() == half_odds(k-1): the domain of log size L is always half_odds(L)
    if (src_log < 1 || src_log > tw->log_half) return set_err(ctx, NX_ERR_ARG, "nx_fold_line: domain not covered by the twiddle tree");
    Sec4 d; Sec4C s;
    for (int q = 0; q < 4; q++) { d.c[q] = d_dst4[q]; s.c[q] = d_src4[q]; }
    uint32_t n = 1u << (src_log - 1);
    hipLaunchKernelGGL(fold_line_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d, s, (int)src_log, 0u, n, tw->d_itw, tw->log_half, q_load(alpha));
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

}  // extern "C"
