// Logup interaction-trace generation on device: SURVEY.md §8(f) rank 2 ("next" row R8).
//
// Replaces, for columns that are already resident in HBM, the CPU loops of the reference's interaction-trace fill
// (prover/src/traits.rs:124-145 -> per chip e.g. prover/src/chips/range_check/range256.rs:271-288;
// prover2/machine/src/lookups/logup_trace_builder.rs:22-121) and the Stwo `LogupTraceGenerator` they drive:
//   relation.combine(tuple)            denom(row) = sum_i alpha^i * tuple_i(row) - z                      -> nx_logup_combine
//   LogupColGenerator::finalize_col    col_k(row) = num(row) / denom(row) + col_{k-1}(row)                 -> nx_logup_finalize_col
//       (prover2 merges two fractions per column first: (a d + b c) / (b d), logup_trace_builder.rs:93-97)
//   LogupTraceGenerator::finalize_last claimed_sum = sum over rows of the last column; the column becomes the inclusive
//       prefix sum, IN NATURAL COSET ORDER, of (value - claimed_sum / N)                                   -> nx_logup_finalize_last
// All columns are bit-reversed circle-domain order like every other column here; a secure (QM31) column is 4 coordinate
// columns.  One lane per row; the only cross-row step is the scan, done per 4096-row block in LDS with the block offsets
// scanned on the host (<= 4096 QM31 values).
#include "internal.h"
#include <algorithm>
#include <string.h>
#include <string>
#include <stdlib.h>

namespace nx {

struct Sec4 { u32* c[4]; };
struct Sec4C { const u32* c[4]; };

__global__ __launch_bounds__(256) void logup_combine_kernel(ColSet cols, u32 n_cols, const u32* __restrict__ coeffs /*4 per column*/, QM31 z, u32 n, Sec4 out) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    u64 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (u32 k = 0; k < n_cols; k++) {
        const u32 v = cols.col(k)[r];
        s0 = acc_mad(s0, coeffs[4 * k], v); s1 = acc_mad(s1, coeffs[4 * k + 1], v); s2 = acc_mad(s2, coeffs[4 * k + 2], v); s3 = acc_mad(s3, coeffs[4 * k + 3], v);
        if ((k & 3) == 3) { s0 = acc_fold(s0); s1 = acc_fold(s1); s2 = acc_fold(s2); s3 = acc_fold(s3); }
    }
    const QM31 d = q_sub(qm(acc_final(s0), acc_final(s1), acc_final(s2), acc_final(s3)), z);
    out.c[0][r] = d.a.a; out.c[1][r] = d.a.b; out.c[2][r] = d.b.a; out.c[3][r] = d.b.b;
}

// numerator = scale * mult[row] (mult == nullptr: scale): covers 1, -1, a multiplicity column and its negation
struct LogupFrac { const u32* mult; QM31 scale; Sec4C den; };

__device__ __forceinline__ QM31 frac_num(const LogupFrac& f, u32 r) { return f.mult ? q_mul_m(f.scale, f.mult[r]) : f.scale; }
__device__ __forceinline__ QM31 sec_at(const Sec4C& s, u32 r) { return qm(s.c[0][r], s.c[1][r], s.c[2][r], s.c[3][r]); }

__global__ __launch_bounds__(256) void logup_finalize_col_kernel(LogupFrac fa, LogupFrac fb, bool two, Sec4C prev, bool has_prev, u32 n, Sec4 out) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    QM31 num = frac_num(fa, r), den = sec_at(fa.den, r);
    if (two) {
        const QM31 c = frac_num(fb, r), d = sec_at(fb.den, r);
        num = q_add(q_mul(num, d), q_mul(den, c));   // a d + b c
        den = q_mul(den, d);                          // b d
    }
    QM31 v = q_mul(num, q_inv(den));
    if (has_prev) v = q_add(v, sec_at(prev, r));
    out.c[0][r] = v.a.a; out.c[1][r] = v.a.b; out.c[2][r] = v.b.a; out.c[3][r] = v.b.b;
}


// Fused form: the denominators never touch memory.  A fraction is (scale * mult(row)) / (sum_i alpha^i tuple_i(row) - z);
// per output column the kernel reads the tuple columns and multiplicities of one or two fractions plus the previous column and
// writes 4 words per row — for a one-limb range check 10 words per row instead of 26 through combine + finalize_col.
struct LogupTupleFrac { ColSet cols; u32 n_cols; const u32* coeffs; QM31 z; const u32* mult; QM31 scale; };

__device__ __forceinline__ QM31 tuple_den(const LogupTupleFrac& f, u32 r) {
    u64 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (u32 k = 0; k < f.n_cols; k++) {
        const u32 v = f.cols.col(k)[r];
        s0 = acc_mad(s0, f.coeffs[4 * k], v); s1 = acc_mad(s1, f.coeffs[4 * k + 1], v); s2 = acc_mad(s2, f.coeffs[4 * k + 2], v); s3 = acc_mad(s3, f.coeffs[4 * k + 3], v);
        if ((k & 3) == 3) { s0 = acc_fold(s0); s1 = acc_fold(s1); s2 = acc_fold(s2); s3 = acc_fold(s3); }
    }
    return q_sub(qm(acc_final(s0), acc_final(s1), acc_final(s2), acc_final(s3)), f.z);
}

__global__ __launch_bounds__(256) void logup_col_kernel(LogupTupleFrac fa, LogupTupleFrac fb, bool two, Sec4C prev, bool has_prev, u32 n, Sec4 out) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    QM31 num = fa.mult ? q_mul_m(fa.scale, fa.mult[r]) : fa.scale, den = tuple_den(fa, r);
    if (two) {
        const QM31 c = fb.mult ? q_mul_m(fb.scale, fb.mult[r]) : fb.scale, d = tuple_den(fb, r);
        num = q_add(q_mul(num, d), q_mul(den, c));
        den = q_mul(den, d);
    }
    QM31 v = q_mul(num, q_inv(den));
    if (has_prev) v = q_add(v, sec_at(prev, r));
    out.c[0][r] = v.a.a; out.c[1][r] = v.a.b; out.c[2][r] = v.b.a; out.c[3][r] = v.b.b;
}


// All the logup columns of a component in ONE launch (LogupTraceGenerator driven column after column): column j of a row is the
// running sum of the row's first j + 1 fractions, so a lane walks the fractions of its row once — every tuple column is read once,
// no previous column is re-read, 16 bytes are written per column.  Descriptors live in device memory (flat pointer / alpha-power
// tables), so the tuple width is not limited by the kernel argument size (the reference's widest relation has 200 elements).
// out_col: 1 + the logup column the running sum is stored to after this fraction (the last fraction of a batch), 0 = none
struct LogupBatchFrac { u32 first_col, n_cols, first_ap, out_col; const u32* mult; QM31 z, scale; };
// The QM31 inverse of a denominator is (conj-style) den^-1 = (a, -b) * D^-1 with D = a^2 - (2+i) b^2 in CM31 and D^-1 = conj(D) / N,
// N = D.a^2 + D.b^2 in M31: the one expensive step is the M31 inverse of N (37 multiplications of the 57 of q_inv).  A lane walks
// the fractions of its row in groups of LOGUP_GROUP and inverts the group's norms with ONE m_inv (Montgomery's trick on the M31 norms:
// 3 multiplications per element): 57 -> 20 + 3 + 37/8 multiplications per fraction.  Same values: an inverse is unique; a zero
// denominator (norm 0) still gives 0 like m_inv(0), and does not poison its group.
constexpr int LOGUP_GROUP = 8;
// STAGED: every memory read of a group — the tuple columns of its 8 fractions (up to LOGUP_TMAX of them) and their multiplicities — is
// requested before the first value is used.  The fractions' tuple widths are run-time values, so the values cannot sit in (statically
// indexed) registers: each lane parks its own row's values in LDS (stage[column][lane], read back by the same lane: no barrier) and the
// combine loop indexes that.  Without it every column read of every fraction is a dependent scalar load (the pointer) + vector load with
// a full wait in between: the kernel ran at the latency of ~500 serial round trips per lane (1.3 TB/s written at 250 fractions).
constexpr int LOGUP_TMAX = 32;            // 32 KB of LDS per 256-lane block: 5 blocks per CU, what the 88 VGPRs allow anyway
template <bool STAGED>
__global__ __launch_bounds__(256) void logup_cols_kernel(const LogupBatchFrac* __restrict__ fr, u32 n_fracs, const u32* const* __restrict__ tuple_cols,
                                                         const u32* __restrict__ ap /*4 words each*/, u32* const* __restrict__ out /*4 per logup column*/, u32 n) {
    __shared__ u32 stage[STAGED ? LOGUP_TMAX * 256 : 1];
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;                                               // no barrier below: a lane only reads back what it staged itself
    QM31 run = q_zero();
    for (u32 j0 = 0; j0 < n_fracs; j0 += LOGUP_GROUP) {
        QM31 den[LOGUP_GROUP]; CM31 dd[LOGUP_GROUP]; u32 nrm[LOGUP_GROUP], pre[LOGUP_GROUP], mv[LOGUP_GROUP];
        bool staged = false; u32 c0 = 0;
        if (STAGED) {
#pragma unroll
            for (int g = 0; g < LOGUP_GROUP; g++) {                   // the multiplicities first: in flight with everything below.  Branch-free
                const u32 jg = j0 + g < n_fracs ? j0 + g : n_fracs - 1;   // (a select, not a jump: a join would wait for the load): a fraction
                const u32* mp = fr[jg].mult;                          // without one reads a word of the first output column and drops it
                mv[g] = gld((mp ? mp : (const u32*)out[0]) + r);
            }
            const u32 jl = (j0 + LOGUP_GROUP < n_fracs ? j0 + LOGUP_GROUP : n_fracs) - 1;
            c0 = fr[j0].first_col;
            const u32 T = fr[jl].first_col + fr[jl].n_cols - c0;      // the group's tuple columns are consecutive in the table (logup_cols_launch)
            staged = T <= (u32)LOGUP_TMAX;                            // uniform; a wider group reads its columns where it uses them
            if (staged) {
                for (u32 t0 = 0; t0 < T; t0 += 8) {
                    u32 v[8];
#pragma unroll
                    for (u32 i = 0; i < 8; i++) { const u32 t = t0 + i < T ? t0 + i : T - 1; v[i] = gld(tuple_cols[c0 + t] + r); }
#pragma unroll
                    for (u32 i = 0; i < 8; i++) { const u32 t = t0 + i < T ? t0 + i : T - 1; stage[t * 256 + threadIdx.x] = v[i]; }
                }
            }
        }
        // phase 1, once per source of the values (the branch is taken per group, not per column read)
        auto denominators = [&](auto value_of) {
#pragma unroll
            for (int g = 0; g < LOGUP_GROUP; g++) {
                if (j0 + g < n_fracs) {                               // uniform
                    const LogupBatchFrac f = fr[j0 + g];
                    u64 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
                    for (u32 k = 0; k < f.n_cols; k++) {
                        const u32 v = value_of(f.first_col + k);
                        const u32* a = ap + 4 * (size_t)(f.first_ap + k);
                        s0 = acc_mad(s0, a[0], v); s1 = acc_mad(s1, a[1], v); s2 = acc_mad(s2, a[2], v); s3 = acc_mad(s3, a[3], v);
                        if ((k & 3) == 3) { s0 = acc_fold(s0); s1 = acc_fold(s1); s2 = acc_fold(s2); s3 = acc_fold(s3); }
                    }
                    den[g] = q_sub(qm(acc_final(s0), acc_final(s1), acc_final(s2), acc_final(s3)), f.z);
                    dd[g] = q_norm_cm(den[g]);
                    nrm[g] = m_add(m_sqr(dd[g].a), m_sqr(dd[g].b));
                } else { den[g] = q_zero(); dd[g] = cm(0, 0); nrm[g] = 1; }
                const u32 nz = nrm[g] ? nrm[g] : 1u;
                pre[g] = g ? m_mul(pre[g - 1], nz) : nz;
            }
        };
        if (STAGED && staged) denominators([&](u32 col) { return stage[(col - c0) * 256 + threadIdx.x]; });
        else denominators([&](u32 col) { return gld(tuple_cols[col] + r); });
        u32 inv = m_inv(pre[LOGUP_GROUP - 1]);
#pragma unroll
        for (int g = LOGUP_GROUP - 1; g >= 0; g--) {
            const u32 nz = nrm[g] ? nrm[g] : 1u;
            const u32 ni = g ? m_mul(inv, pre[g - 1]) : inv;          // 1 / nz
            inv = m_mul(inv, nz);
            pre[g] = nrm[g] ? ni : 0u;                                 // reused: 1 / norm (0 for a zero denominator, like m_inv)
        }
#pragma unroll
        for (int g = 0; g < LOGUP_GROUP; g++) {
            if (j0 + g < n_fracs) {
                const u32 j = j0 + g;
                const LogupBatchFrac f = fr[j];
                const u32 mval = f.mult ? (STAGED ? mv[g] : gld(f.mult + r)) : 0u;
                if ((f.scale.a.b | f.scale.b.a | f.scale.b.b) == 0) {          // uniform: a base-field numerator (+-1, a multiplicity): it rides on 1 / |D|^2,
                    const u32 nm = f.mult ? m_mul(f.scale.a.a, mval) : f.scale.a.a;     // and the fraction goes straight into the running sum's accumulators
                    const u32 sc = m_mul(pre[g], nm);
                    run = q_conj_times_add(run, den[g], cm(m_mul(dd[g].a, sc), m_mul(m_neg(dd[g].b), sc)));
                } else {
                    const CM31 di = cm(m_mul(dd[g].a, pre[g]), m_mul(m_neg(dd[g].b), pre[g]));
                    const QM31 qi = q_conj_times(den[g], di);
                    const QM31 num = f.mult ? q_mul_m(f.scale, mval) : f.scale;
                    run = q_add(run, q_mul(num, qi));
                }
                if (f.out_col) {                                               // uniform: the batch is complete (finalize_logup_batched: one column per batch)
                    const u32 oc = f.out_col - 1;
                    gst(out[4 * oc] + r, run.a.a); gst(out[4 * oc + 1] + r, run.a.b); gst(out[4 * oc + 2] + r, run.b.a); gst(out[4 * oc + 3] + r, run.b.b);
                }
            }
        }
    }
}

// position (bit-reversed circle-domain order) of natural coset row c
__device__ __forceinline__ u32 pos_of_coset_row(u32 c, int log) {
    const u32 N = 1u << log;
    const u32 d = (c & 1) ? N - 1 - (c >> 1) : (c >> 1);
    return bitrev(d, log);
}

constexpr int SCAN_ITEMS = 16, SCAN_THREADS = 256, SCAN_BLOCK = SCAN_ITEMS * SCAN_THREADS;

// phase 1: per block of 4096 consecutive coset rows, the block-local inclusive scan (written back in place) and the block total
__global__ __launch_bounds__(SCAN_THREADS) void logup_scan_local_kernel(const Sec4* __restrict__ cols /*one per blockIdx.y*/, int log,
                                                                        u32* __restrict__ all_sums /*[column][block][4]*/) {
    __shared__ u32 lds[4][SCAN_THREADS];
    const Sec4 col = cols[blockIdx.y];
    u32* __restrict__ block_sums = all_sums + (size_t)blockIdx.y * gridDim.x * 4;
    const u32 n = 1u << log;
    const u32 c0 = (blockIdx.x * SCAN_THREADS + threadIdx.x) * SCAN_ITEMS;
    u32 pos[SCAN_ITEMS];
    u32 v[4][SCAN_ITEMS];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        pos[i] = c0 + i < n ? pos_of_coset_row(c0 + i, log) : 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < 4; q++) v[q][i] = pos[i] != 0xFFFFFFFFu ? col.c[q][pos[i]] : 0u;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int i = 1; i < SCAN_ITEMS; i++) v[q][i] = m_add(v[q][i], v[q][i - 1]);
        lds[q][threadIdx.x] = v[q][SCAN_ITEMS - 1];
    }
    __syncthreads();
    // Hillis-Steele over the 256 lane totals (4 coordinates)
    for (int off = 1; off < SCAN_THREADS; off <<= 1) {
        u32 t[4];
#pragma unroll
        for (int q = 0; q < 4; q++) t[q] = threadIdx.x >= (u32)off ? lds[q][threadIdx.x - off] : 0u;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) lds[q][threadIdx.x] = m_add(lds[q][threadIdx.x], t[q]);
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const u32 before = threadIdx.x ? lds[q][threadIdx.x - 1] : 0u;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) if (pos[i] != 0xFFFFFFFFu) col.c[q][pos[i]] = m_add(v[q][i], before);
    }
    if (threadIdx.x == SCAN_THREADS - 1)
        for (int q = 0; q < 4; q++) block_sums[4 * blockIdx.x + q] = lds[q][SCAN_THREADS - 1];
}

// phase 2 (one block per column): exclusive scan of the block totals in place; meta[column] = claimed sum (4) ‖ claimed / N (4)
__global__ __launch_bounds__(SCAN_THREADS) void logup_scan_sums_kernel(u32* __restrict__ all_sums, u32 n_blocks, int log, u32* __restrict__ meta) {
    __shared__ u32 lds[4][SCAN_THREADS];
    u32* sums = all_sums + (size_t)blockIdx.x * n_blocks * 4;
    const u32 per = (n_blocks + SCAN_THREADS - 1) / SCAN_THREADS, b0 = threadIdx.x * per, b1 = min(n_blocks, b0 + per);
    QM31 tot = q_zero();
    for (u32 b = b0; b < b1; b++) tot = q_add(tot, qm(sums[4 * b], sums[4 * b + 1], sums[4 * b + 2], sums[4 * b + 3]));
    lds[0][threadIdx.x] = tot.a.a; lds[1][threadIdx.x] = tot.a.b; lds[2][threadIdx.x] = tot.b.a; lds[3][threadIdx.x] = tot.b.b;
    __syncthreads();
    for (int off = 1; off < SCAN_THREADS; off <<= 1) {
        u32 t[4];
#pragma unroll
        for (int q = 0; q < 4; q++) t[q] = threadIdx.x >= (u32)off ? lds[q][threadIdx.x - off] : 0u;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) lds[q][threadIdx.x] = m_add(lds[q][threadIdx.x], t[q]);
        __syncthreads();
    }
    QM31 run = threadIdx.x ? qm(lds[0][threadIdx.x - 1], lds[1][threadIdx.x - 1], lds[2][threadIdx.x - 1], lds[3][threadIdx.x - 1]) : q_zero();
    for (u32 b = b0; b < b1; b++) {
        const QM31 t = qm(sums[4 * b], sums[4 * b + 1], sums[4 * b + 2], sums[4 * b + 3]);
        sums[4 * b] = run.a.a; sums[4 * b + 1] = run.a.b; sums[4 * b + 2] = run.b.a; sums[4 * b + 3] = run.b.b;
        run = q_add(run, t);
    }
    if (threadIdx.x == SCAN_THREADS - 1) {
        const QM31 total = qm(lds[0][SCAN_THREADS - 1], lds[1][SCAN_THREADS - 1], lds[2][SCAN_THREADS - 1], lds[3][SCAN_THREADS - 1]);
        const QM31 shift = q_mul_m(total, m_inv((1u << log) % P));
        u32* o = meta + 8 * blockIdx.x;
        o[0] = total.a.a; o[1] = total.a.b; o[2] = total.b.a; o[3] = total.b.b;
        o[4] = shift.a.a; o[5] = shift.a.b; o[6] = shift.b.a; o[7] = shift.b.b;
    }
}

// phase 3: value = local + offset[block of its coset row] - (coset row + 1) * shift
__global__ __launch_bounds__(256) void logup_scan_fix_kernel(const Sec4* __restrict__ cols, int log, const u32* __restrict__ all_offsets /*exclusive, [column][block][4]*/,
                                                             u32 n_blocks, const u32* __restrict__ meta) {
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (1u << log)) return;
    const Sec4 col = cols[blockIdx.y];
    const u32* __restrict__ block_offsets = all_offsets + (size_t)blockIdx.y * n_blocks * 4;
    const u32* mm = meta + 8 * blockIdx.y;
    const QM31 shift = qm(mm[4], mm[5], mm[6], mm[7]);
    const u32 p = pos_of_coset_row(c, log), b = c / SCAN_BLOCK;
    const QM31 off = qm(block_offsets[4 * b], block_offsets[4 * b + 1], block_offsets[4 * b + 2], block_offsets[4 * b + 3]);
    const QM31 v = q_sub(q_add(qm(col.c[0][p], col.c[1][p], col.c[2][p], col.c[3][p]), off), q_mul_m(shift, m_reduce64((u64)c + 1)));
    col.c[0][p] = v.a.a; col.c[1][p] = v.a.b; col.c[2][p] = v.b.a; col.c[3][p] = v.b.b;
}

// ---- the same scan with coalesced accesses (log_size >= 13) --------------------------------------------------------------------
// Coset rows 2k and 2k+1 sit at positions 2m and N-1-2m with m = bitrev_{n-1}(k): the even row in the lower "E" word, the odd row in
// the mirrored "O" word.  S(2k) = X[k] + E[k], S(2k+1) = X[k] + E[k] + O[k] with X the exclusive scan of E + O over k — a scan over an
// array stored in bit-reversed order.  A block takes 2^T neighbouring values of the low part of m (runs of 2^(T+1) words: E's at the
// even words, the mirror block's O's between them) times all 2^H values of the top part: that is 2^T k-segments of 2^H consecutive
// pairs, transposed through LDS.  Levels: segment-local scan (this kernel), the segment totals (logup_scan_sums_kernel, as before),
// the fix-up with the same tiling.  One (column, coordinate) per blockIdx.y: a QM31 sum and a multiple of the shift are coordinate-wise.
constexpr int SC_H = 7, SC_T = 5, SC_MIN_LOG = SC_H + SC_T + 1;
struct ScanTile { u32 m, s, kl; };   // pair index in memory order; its segment (k >> H) and place in it
__device__ __forceinline__ ScanTile scan_tile_index(u32 mh, u32 ml_hi, u32 t, int n1) {
    const int L = n1 - SC_H;
    ScanTile x;
    x.m = (mh << L) | (ml_hi << SC_T) | t;
    x.s = (bitrev(t, SC_T) << (L - SC_T)) | (L > SC_T ? bitrev(ml_hi, L - SC_T) : 0u);
    x.kl = bitrev(mh, SC_H);
    return x;
}

__global__ __launch_bounds__(256) void logup_scan_tile_kernel(const Sec4* __restrict__ cols, int log, u32* __restrict__ all_sums /*[column][segment][4]*/) {
    constexpr u32 SEGS = 1u << SC_T, LEN = 1u << SC_H, PITCH = LEN + 1;
    __shared__ u32 E[SEGS * PITCH], O[SEGS * PITCH];
    const int n1 = log - 1;
    const u32 N = 1u << log, n_seg = 1u << (n1 - SC_H);
    const u32 q = blockIdx.y & 3;
    u32* __restrict__ col = cols[blockIdx.y >> 2].c[q];
    u32* __restrict__ sums = all_sums + (size_t)(blockIdx.y >> 2) * n_seg * 4;
    const u32 t = threadIdx.x & (SEGS - 1), mh0 = threadIdx.x >> SC_T;      // 8 values of mh per sweep
    for (u32 mh = mh0; mh < LEN; mh += 256 >> SC_T) {
        const ScanTile x = scan_tile_index(mh, blockIdx.x, t, n1);
        E[t * PITCH + x.kl] = gld(col + 2 * (size_t)x.m);
        O[t * PITCH + x.kl] = gld(col + (N - 1 - 2 * (size_t)x.m));
    }
    __syncthreads();
    {   // 8 lanes per segment, 16 consecutive pairs per lane
        const u32 seg = threadIdx.x >> 3, j = threadIdx.x & 7, base = seg * PITCH + j * (LEN / 8);
        u32 run = 0;
#pragma unroll
        for (u32 i = 0; i < LEN / 8; i++) run = m_add(run, m_add(E[base + i], O[base + i]));
        u32 incl = run;                                                     // inclusive scan of the 8 lane totals of the segment
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) { const u32 up = __shfl_up(incl, d, 8); if ((int)j >= d) incl = m_add(incl, up); }
        u32 x = m_sub(incl, run);                                           // exclusive offset of this lane inside its segment
#pragma unroll
        for (u32 i = 0; i < LEN / 8; i++) {
            const u32 e = E[base + i], o = O[base + i];
            const u32 se = m_add(x, e), so = m_add(se, o);
            E[base + i] = se; O[base + i] = so; x = so;
        }
        if (j == 7) {
            const u32 s = (bitrev(seg, SC_T) << (n1 - SC_H - SC_T)) | ((n1 - SC_H) > SC_T ? bitrev(blockIdx.x, n1 - SC_H - SC_T) : 0u);
            sums[4 * (size_t)s + q] = incl;
        }
    }
    __syncthreads();
    for (u32 mh = mh0; mh < LEN; mh += 256 >> SC_T) {
        const ScanTile x = scan_tile_index(mh, blockIdx.x, t, n1);
        gst(col + 2 * (size_t)x.m, E[t * PITCH + x.kl]);
        gst(col + (N - 1 - 2 * (size_t)x.m), O[t * PITCH + x.kl]);
    }
}

// fix-up with the same tiling: value += offset[segment] - (coset row + 1) * shift
__global__ __launch_bounds__(256) void logup_scan_tile_fix_kernel(const Sec4* __restrict__ cols, int log, const u32* __restrict__ all_offsets /*exclusive, [column][segment][4]*/,
                                                                  const u32* __restrict__ meta) {
    const int n1 = log - 1;
    const u32 N = 1u << log, n_seg = 1u << (n1 - SC_H);
    const u32 q = blockIdx.y & 3;
    u32* __restrict__ col = cols[blockIdx.y >> 2].c[q];
    const u32* __restrict__ offs = all_offsets + (size_t)(blockIdx.y >> 2) * n_seg * 4;
    const u32 shift = meta[8 * (blockIdx.y >> 2) + 4 + q];
    const u32 t = threadIdx.x & ((1u << SC_T) - 1);
    const u32 sweeps = (1u << SC_H) / (256 >> SC_T);                         // blockIdx.x = ml_hi * sweeps + sweep
    const u32 ml_hi = blockIdx.x / sweeps, mh = (blockIdx.x % sweeps) * (256 >> SC_T) + (threadIdx.x >> SC_T);
    const ScanTile x = scan_tile_index(mh, ml_hi, t, n1);
    const u32 off = offs[4 * (size_t)x.s + q];
    const u64 c = 2 * (((u64)x.s << SC_H) | x.kl);                           // the even coset row of the pair
    u32* pe = col + 2 * (size_t)x.m; u32* po = col + (N - 1 - 2 * (size_t)x.m);
    gst(pe, m_sub(m_add(gld(pe), off), m_mul(shift, m_reduce64(c + 1))));
    gst(po, m_sub(m_add(gld(po), off), m_mul(shift, m_reduce64(c + 2))));
}

}  // namespace nx

using namespace nx;

extern "C" {

int nx_logup_combine(nx_ctx* ctx, const uint32_t* const* d_tuple_cols, uint32_t n_cols, const uint32_t* alpha_powers, const uint32_t z[4], uint32_t log_size,
                     uint32_t* const* d_out4) {
    NX_GUARD(ctx);
    if (!ctx || !d_out4 || !z || (n_cols && (!d_tuple_cols || !alpha_powers))) return set_err(ctx, NX_ERR_ARG, "nx_logup_combine: NULL argument");
    if (log_size > 30) return set_err(ctx, NX_ERR_ARG, "nx_logup_combine: log_size too large");
    ColSet cs; NX_TRY(make_colset(ctx, d_tuple_cols, n_cols, &cs));
    void* d_co = nullptr;
    if (n_cols) NX_TRY(stage(ctx, alpha_powers, (size_t)n_cols * 16, &d_co));
    Sec4 o; for (int q = 0; q < 4; q++) o.c[q] = d_out4[q];
    const u32 n = 1u << log_size;
    hipLaunchKernelGGL(logup_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, cs, n_cols, (const u32*)d_co, q_load(z), n, o);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

static LogupFrac make_frac(const uint32_t* d_mult, const uint32_t scale[4], const uint32_t* const* d_den4) {
    LogupFrac f; f.mult = d_mult; f.scale = q_load(scale);
    for (int q = 0; q < 4; q++) f.den.c[q] = d_den4[q];
    return f;
}

int nx_logup_finalize_col(nx_ctx* ctx, uint32_t log_size, const uint32_t* d_mult_a, const uint32_t scale_a[4], const uint32_t* const* d_den_a4,
                          const uint32_t* d_mult_b, const uint32_t scale_b[4], const uint32_t* const* d_den_b4, const uint32_t* const* d_prev4,
                          uint32_t* const* d_out4) {
    NX_GUARD(ctx);
    if (!ctx || !scale_a || !d_den_a4 || !d_out4) return set_err(ctx, NX_ERR_ARG, "nx_logup_finalize_col: NULL argument");
    if (log_size > 30) return set_err(ctx, NX_ERR_ARG, "nx_logup_finalize_col: log_size too large");
    const bool two = d_den_b4 != nullptr;
    if (two && !scale_b) return set_err(ctx, NX_ERR_ARG, "nx_logup_finalize_col: second fraction without a numerator scale");
    LogupFrac fa = make_frac(d_mult_a, scale_a, d_den_a4), fb = two ? make_frac(d_mult_b, scale_b, d_den_b4) : fa;
    Sec4C prev; Sec4 o;
    for (int q = 0; q < 4; q++) { prev.c[q] = d_prev4 ? d_prev4[q] : nullptr; o.c[q] = d_out4[q]; }
    const u32 n = 1u << log_size;
    hipLaunchKernelGGL(logup_finalize_col_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, fa, fb, two, prev, d_prev4 != nullptr, n, o);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}


static int make_tuple_frac(nx_ctx* ctx, const nx_logup_frac* f, LogupTupleFrac* out) {
    if (!f->alpha_powers || !f->z || !f->scale || (f->n_tuple_cols && !f->d_tuple_cols)) return set_err(ctx, NX_ERR_ARG, "nx_logup_col: incomplete fraction");
    NX_TRY(make_colset(ctx, f->d_tuple_cols, f->n_tuple_cols, &out->cols));
    void* d_co = nullptr;
    if (f->n_tuple_cols) NX_TRY(stage(ctx, f->alpha_powers, (size_t)f->n_tuple_cols * 16, &d_co));
    out->n_cols = f->n_tuple_cols; out->coeffs = (const u32*)d_co; out->z = q_load(f->z); out->mult = f->d_mult; out->scale = q_load(f->scale);
    return NX_OK;
}

int nx_logup_col(nx_ctx* ctx, uint32_t log_size, const nx_logup_frac* frac_a, const nx_logup_frac* frac_b, const uint32_t* const* d_prev4, uint32_t* const* d_out4) {
    NX_GUARD(ctx);
    if (!ctx || !frac_a || !d_out4) return set_err(ctx, NX_ERR_ARG, "nx_logup_col: NULL argument");
    if (log_size > 30) return set_err(ctx, NX_ERR_ARG, "nx_logup_col: log_size too large");
    LogupTupleFrac fa, fb;
    NX_TRY(make_tuple_frac(ctx, frac_a, &fa));
    fb = fa;
    if (frac_b) NX_TRY(make_tuple_frac(ctx, frac_b, &fb));
    Sec4C prev; Sec4 o;
    for (int q = 0; q < 4; q++) { prev.c[q] = d_prev4 ? d_prev4[q] : nullptr; o.c[q] = d_out4[q]; }
    const u32 n = 1u << log_size;
    hipLaunchKernelGGL(logup_col_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, fa, fb, frac_b != nullptr, prev, d_prev4 != nullptr, n, o);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}


// The logup columns of one component in one launch: fraction i belongs to batch batching[i], column j = the sum of the fractions of
// batches <= j (finalize_logup_batched on the trace side).  The fractions are walked in batch order (a stable sort by batch: the sum
// does not depend on the order inside a batch); the last fraction of a batch stores the running sum.  d_out: 4 n_cols coordinate columns.
static int logup_cols_launch(nx_ctx* ctx, uint32_t log_size, const nx_logup_frac* fracs, uint32_t n_fracs, const uint32_t* batching, uint32_t n_cols, uint32_t* const* d_out,
                             const char* who) {
    if (!ctx || (n_fracs && !fracs) || (n_cols && !d_out)) return set_err(ctx, NX_ERR_ARG, std::string(who) + ": NULL argument");
    if (log_size > 30) return set_err(ctx, NX_ERR_ARG, std::string(who) + ": log_size too large");
    if (!n_cols && !n_fracs) return NX_OK;
    if (!n_cols || n_fracs < n_cols) return set_err(ctx, NX_ERR_ARG, std::string(who) + ": every logup column needs at least one fraction");
    std::vector<u32> order(n_fracs), batch(n_fracs);
    for (u32 i = 0; i < n_fracs; i++) {
        order[i] = i; batch[i] = batching ? batching[i] : i;
        if (batch[i] >= n_cols) return set_err(ctx, NX_ERR_ARG, std::string(who) + ": batch index outside the logup columns");
    }
    std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) { return batch[a] < batch[b]; });
    {
        std::vector<char> seen(n_cols, 0);
        for (u32 i = 0; i < n_fracs; i++) seen[batch[i]] = 1;
        for (u32 j = 0; j < n_cols; j++) if (!seen[j]) return set_err(ctx, NX_ERR_ARG, std::string(who) + ": a batch without fractions (finalize_logup_batched requires every batch 0 .. last)");
    }
    std::vector<LogupBatchFrac> h(n_fracs);
    std::vector<const u32*> cols; std::vector<u32> ap;
    for (u32 s = 0; s < n_fracs; s++) {
        const nx_logup_frac& f = fracs[order[s]];
        if (!f.alpha_powers || !f.z || !f.scale || (f.n_tuple_cols && !f.d_tuple_cols)) return set_err(ctx, NX_ERR_ARG, std::string(who) + ": incomplete fraction");
        h[s].first_col = (u32)cols.size(); h[s].n_cols = f.n_tuple_cols; h[s].first_ap = (u32)(ap.size() / 4);
        h[s].out_col = (s + 1 == n_fracs || batch[order[s + 1]] != batch[order[s]]) ? batch[order[s]] + 1 : 0;
        for (u32 k = 0; k < f.n_tuple_cols; k++) { if (!f.d_tuple_cols[k]) return set_err(ctx, NX_ERR_ARG, std::string(who) + ": NULL tuple column"); cols.push_back(f.d_tuple_cols[k]); }
        ap.insert(ap.end(), f.alpha_powers, f.alpha_powers + 4 * (size_t)f.n_tuple_cols);
        h[s].mult = f.d_mult; h[s].z = q_load(f.z); h[s].scale = q_load(f.scale);
    }
    for (size_t k = 0; k < 4 * (size_t)n_cols; k++) if (!d_out[k]) return set_err(ctx, NX_ERR_ARG, std::string(who) + ": NULL output column");
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t b_fr = h.size() * sizeof(LogupBatchFrac), b_cols = cols.size() * 8, b_ap = ap.size() * 4, b_out = (size_t)n_cols * 32;
    const size_t o_cols = al(b_fr), o_ap = o_cols + al(b_cols), o_out = o_ap + al(b_ap), total = o_out + al(b_out) + 16;
    std::vector<uint8_t> host(total, 0);
    memcpy(host.data(), h.data(), b_fr);
    if (b_cols) memcpy(host.data() + o_cols, cols.data(), b_cols);
    if (b_ap) memcpy(host.data() + o_ap, ap.data(), b_ap);
    memcpy(host.data() + o_out, d_out, b_out);
    uint8_t* blob = nullptr;
    NX_TRY(dev_alloc(ctx, total, (void**)&blob));      // owned (not the staging ring): hundreds of wide fractions exceed a ring slot's lifetime guarantees
    hipError_t e = hipSuccess;
    for (size_t off = 0; off < total && e == hipSuccess; off += (size_t)4 << 20) {
        const size_t nb = std::min(total - off, (size_t)4 << 20);
        void* st = nullptr;
        int rc = stage(ctx, host.data() + off, nb, &st);
        if (rc != NX_OK) { dev_free(ctx, blob); return rc; }
        e = hipMemcpyAsync(blob + off, st, nb, hipMemcpyDeviceToDevice, ctx->stream);
    }
    const u32 n = 1u << log_size;
    if (e == hipSuccess) {
        if (ctx->opt.logup_staged)
            hipLaunchKernelGGL(logup_cols_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const LogupBatchFrac*)blob, n_fracs, (const u32* const*)(blob + o_cols),
                               (const u32*)(blob + o_ap), (u32* const*)(blob + o_out), n);
        else
            hipLaunchKernelGGL(logup_cols_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const LogupBatchFrac*)blob, n_fracs, (const u32* const*)(blob + o_cols),
                               (const u32*)(blob + o_ap), (u32* const*)(blob + o_out), n);
        e = hipGetLastError();
    }
    dev_free(ctx, blob);   // stream-ordered: reused only by later work on this stream
    if (e != hipSuccess) return hip_fail(ctx, e, who, __FILE__, __LINE__);
    return NX_OK;
}

// n_cols logup columns of one component in one launch: column j = sum over i <= j of fraction i (what n_cols calls of nx_logup_col
// with d_prev4 = the previous column produce).  d_out: 4 n_cols coordinate columns.
int nx_logup_cols(nx_ctx* ctx, uint32_t log_size, const nx_logup_frac* fracs, uint32_t n_cols, uint32_t* const* d_out) {
    NX_GUARD(ctx);
    return logup_cols_launch(ctx, log_size, fracs, n_cols, nullptr, n_cols, d_out, "nx_logup_cols");
}

int nx_logup_cols_batched(nx_ctx* ctx, uint32_t log_size, const nx_logup_frac* fracs, uint32_t n_fracs, const uint32_t* batching, uint32_t n_cols, uint32_t* const* d_out) {
    NX_GUARD(ctx);
    if (batching) return logup_cols_launch(ctx, log_size, fracs, n_fracs, batching, n_cols, d_out, "nx_logup_cols_batched");
    std::vector<uint32_t> pairs(n_fracs);
    for (uint32_t i = 0; i < n_fracs; i++) pairs[i] = i / 2;
    return logup_cols_launch(ctx, log_size, fracs, n_fracs, pairs.data(), n_cols, d_out, "nx_logup_cols_batched");
}

// LogupTraceGenerator::finalize_last for n_cols secure columns of one size in three launches and ONE device-to-host copy (the
// claimed sums): per-block scans, a one-block scan of the block totals per column, the fix-up.  d_cols4: n_cols x 4 pointers.
int nx_logup_finalize_last_batch(nx_ctx* ctx, uint32_t log_size, uint32_t* const* d_cols4, uint32_t n_cols, uint32_t* claimed_sums) {
    NX_GUARD(ctx);
    if (!ctx || !d_cols4 || !claimed_sums) return set_err(ctx, NX_ERR_ARG, "nx_logup_finalize_last: NULL argument");
    if (log_size > 30) return set_err(ctx, NX_ERR_ARG, "nx_logup_finalize_last: log_size too large");
    if (n_cols == 0) return NX_OK;
    if (n_cols > 65535) return set_err(ctx, NX_ERR_ARG, "nx_logup_finalize_last: at most 65535 columns per call");
    // coalesced tiles from 2^13 rows (NX_LOGUP_SCAN_TILED=0: the per-row gather kernels at every size, kept for A/B and small columns)
    const bool tiled = ctx->opt.logup_scan_tiled && log_size >= (uint32_t)SC_MIN_LOG && n_cols * 4 <= 65535;
    const u32 n = 1u << log_size, n_blocks = tiled ? 1u << (log_size - 1 - SC_H) : (n + SCAN_BLOCK - 1) / SCAN_BLOCK;   // tiled: one total per 128-pair segment
    std::vector<Sec4> h(n_cols);
    for (u32 k = 0; k < n_cols; k++) for (int q = 0; q < 4; q++) { if (!d_cols4[4 * k + q]) return set_err(ctx, NX_ERR_ARG, "nx_logup_finalize_last: NULL column"); h[k].c[q] = d_cols4[4 * k + q]; }
    uint8_t* blob = nullptr;
    const size_t b_tab = ((size_t)n_cols * sizeof(Sec4) + 15) & ~(size_t)15, b_sums = (size_t)n_cols * n_blocks * 16, b_meta = (size_t)n_cols * 32;
    NX_TRY(dev_alloc(ctx, b_tab + b_sums + b_meta, (void**)&blob));
    const Sec4* d_tab = (const Sec4*)blob; u32* d_sums = (u32*)(blob + b_tab); u32* d_meta = (u32*)(blob + b_tab + b_sums);
    hipError_t e = hipMemcpyAsync(blob, h.data(), (size_t)n_cols * sizeof(Sec4), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && tiled) {
        const u32 tiles = 1u << (log_size - 1 - SC_H - SC_T), sweeps = (1u << SC_H) / (256 >> SC_T);
        hipLaunchKernelGGL(logup_scan_tile_kernel, dim3(tiles, n_cols * 4), dim3(256), 0, ctx->stream, d_tab, (int)log_size, d_sums);
        hipLaunchKernelGGL(logup_scan_sums_kernel, dim3(n_cols), dim3(SCAN_THREADS), 0, ctx->stream, d_sums, n_blocks, (int)log_size, d_meta);
        hipLaunchKernelGGL(logup_scan_tile_fix_kernel, dim3(tiles * sweeps, n_cols * 4), dim3(256), 0, ctx->stream, d_tab, (int)log_size, (const u32*)d_sums, (const u32*)d_meta);
        e = hipGetLastError();
    } else if (e == hipSuccess) {
        hipLaunchKernelGGL(logup_scan_local_kernel, dim3(n_blocks, n_cols), dim3(SCAN_THREADS), 0, ctx->stream, d_tab, (int)log_size, d_sums);
        hipLaunchKernelGGL(logup_scan_sums_kernel, dim3(n_cols), dim3(SCAN_THREADS), 0, ctx->stream, d_sums, n_blocks, (int)log_size, d_meta);
        hipLaunchKernelGGL(logup_scan_fix_kernel, dim3((n + 255) / 256, n_cols), dim3(256), 0, ctx->stream, d_tab, (int)log_size, (const u32*)d_sums, n_blocks, (const u32*)d_meta);
        e = hipGetLastError();
    }
    std::vector<u32> meta((size_t)n_cols * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(meta.data(), d_meta, meta.size() * 4, hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e2 = hipStreamSynchronize(ctx->stream);   // `h`, `meta` and the blob must outlive the copies / kernels
    dev_free(ctx, blob);
    if (e != hipSuccess) return hip_fail(ctx, e, "nx_logup_finalize_last", __FILE__, __LINE__);
    if (e2 != hipSuccess) return hip_fail(ctx, e2, "nx_logup_finalize_last(sync)", __FILE__, __LINE__);
    for (u32 k = 0; k < n_cols; k++) memcpy(claimed_sums + 4 * k, &meta[8 * k], 16);
    return NX_OK;
}

int nx_logup_finalize_last(nx_ctx* ctx, uint32_t log_size, uint32_t* const* d_col4, uint32_t claimed_sum[4]) {
    NX_GUARD(ctx);
    return nx_logup_finalize_last_batch(ctx, log_size, d_col4, 1, claimed_sum);
}

}  // extern "C"
