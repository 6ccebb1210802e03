// ORACLE — TEST INFRASTRUCTURE ONLY (see fields.h header).  PARITY UNPINNED.
//
// CPU restatement of stwo-constraint-framework's LogupTraceGenerator as the reference drives it
// (reference prover/src/traits.rs:124-145; prover/src/chips/range_check/range256.rs:271-288;
// prover2/machine/src/lookups/logup_trace_builder.rs:22-121) [upstream-recollection for the Stwo side]:
//   Relation::combine          denom = sum_i alpha^i * v_i - z
//   write_frac + finalize_col  col_k = num / denom + col_{k-1}        (prover2 first merges two fractions: (ad+bc)/(bd))
//   finalize_last              claimed_sum = sum of the last column; last column := inclusive prefix sum, in natural coset
//                              order, of (value - claimed_sum / N)  — written here the slow, literal way: undo the
//                              bit-reversed circle-domain storage order, scan, redo it.
#pragma once
#include <vector>
#include <thread>
#include "poly.h"

namespace orc {

// rows are independent in combine / finalize_col: the CPU baseline runs them on every host core (the result does not depend on it)
static inline int& logup_n_threads() { static int v = 1; return v; }
template <class F> static inline void logup_rows(u32 n, F f) {
    const int nt = n >= 4096 ? logup_n_threads() : 1;
    if (nt <= 1) { f(0u, n); return; }
    std::vector<std::thread> th;
    for (int k = 0; k < nt; k++) th.emplace_back(f, (u32)((u64)n * k / nt), (u32)((u64)n * (k + 1) / nt));
    for (auto& x : th) x.join();
}

static inline void logup_combine(const u32* const* cols, u32 n_cols, const u32* alpha_powers, const u32* z, int log, u32* const out4[4]) {
    logup_rows(1u << log, [&](u32 r0, u32 r1) {
    for (u32 r = r0; r < r1; r++) {
        QM31 s = qm31_zero();
        for (u32 k = 0; k < n_cols; k++) s = qm31_add(s, qm31_mul_m31(qm31_load(alpha_powers + 4 * k), cols[k][r]));
        s = qm31_sub(s, qm31_load(z));
        u32 w[4]; qm31_store(w, s);
        for (int q = 0; q < 4; q++) out4[q][r] = w[q];
    }
    });
}

struct LogupFrac { const u32* mult; QM31 scale; const u32* den[4]; };
static inline QM31 frac_num(const LogupFrac& f, u32 r) { return f.mult ? qm31_mul_m31(f.scale, f.mult[r]) : f.scale; }

static inline void logup_finalize_col(int log, const LogupFrac& fa, const LogupFrac* fb, const u32* const* prev4, u32* const out4[4]) {
    logup_rows(1u << log, [&](u32 r0, u32 r1) {
    for (u32 r = r0; r < r1; r++) {
        QM31 num = frac_num(fa, r), den = qm31(fa.den[0][r], fa.den[1][r], fa.den[2][r], fa.den[3][r]);
        if (fb) {
            QM31 c = frac_num(*fb, r), d = qm31(fb->den[0][r], fb->den[1][r], fb->den[2][r], fb->den[3][r]);
            num = qm31_add(qm31_mul(num, d), qm31_mul(den, c));
            den = qm31_mul(den, d);
        }
        QM31 v = qm31_mul(num, qm31_inv(den));
        if (prev4) v = qm31_add(v, qm31(prev4[0][r], prev4[1][r], prev4[2][r], prev4[3][r]));
        u32 w[4]; qm31_store(w, v);
        for (int q = 0; q < 4; q++) out4[q][r] = w[q];
    }
    });
}

static inline void logup_finalize_last(int log, u32* const col4[4], u32* claimed_sum) {
    const u32 N = 1u << log;
    // storage position of natural coset row c: bit_reverse(coset_index_to_circle_domain_index(c))
    std::vector<u32> pos(N);
    for (u32 c = 0; c < N; c++) pos[c] = bit_reverse_index(coset_index_to_circle_domain_index(c, log), log);
    QM31 total = qm31_zero();
    for (u32 c = 0; c < N; c++) total = qm31_add(total, qm31(col4[0][pos[c]], col4[1][pos[c]], col4[2][pos[c]], col4[3][pos[c]]));
    qm31_store(claimed_sum, total);
    const QM31 shift = qm31_mul_m31(total, m31_inv(N % P));
    QM31 run = qm31_zero();
    for (u32 c = 0; c < N; c++) {
        QM31 v = qm31_sub(qm31(col4[0][pos[c]], col4[1][pos[c]], col4[2][pos[c]], col4[3][pos[c]]), shift);
        run = qm31_add(run, v);
        u32 w[4]; qm31_store(w, run);
        for (int q = 0; q < 4; q++) col4[q][pos[c]] = w[q];
    }
}

}  // namespace orc
