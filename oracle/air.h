// ORACLE — TEST INFRASTRUCTURE ONLY (see fields.h header).  PARITY UNPINNED.
//
// The synthetic AIR used by BASELINE.json configs #2-#4 ("wide-Fibonacci-style").
// The reference's real AIR (347-column MachineEval, reference prover/src/components/mod.rs:39-57)
// is generic Rust and out of scope; this AIR mimics its *shape*:
//   * per component one log_size, columns in 3 trees (preprocessed / main / interaction),
//   * two main columns are read at mask [0, 1] (like Pc / IsPadding, reference
//     prover/src/column.rs:13-20), everything else at mask [0],
//   * degree-2 constraints, evaluated on CanonicCoset(log_size + log_constraint_degree)
//     (reference prover/src/components/mod.rs:12,44-46: LOG_CONSTRAINT_DEGREE = 2),
//   * is_first / is_last preprocessed columns (reference prover/src/trace/preprocessed.rs:28-41).
// Constraint folding follows stwo-constraint-framework FrameworkComponent
// (evaluate_constraint_quotients_on_domain / _at_point) and stwo prover/air/accumulation.rs.
#pragma once
#include <vector>
#include "fields.h"

namespace orc {

static const int AIR_GROUP = 16;  // every 16 columns two are "free" (unconstrained witness input)

// log_cd: the component's log constraint-degree bound (its constraints are evaluated on CanonicCoset(log_size + log_cd)); 0 = the
// configuration's log_constraint_degree.  Per component as in the reference: v1's main component +2 (prover/src/components/mod.rs:12,
// 44-45), every extension +1 (prover/src/extensions/multiplicity.rs:108-110); prover2: 1, the shifts 2 (framework/traits/builtin.rs:23).
struct ComponentSpec { int log_size, n_pre, n_main, n_inter, log_cd; };
template <class Cfg> static inline int comp_log_cd(const ComponentSpec& c, const Cfg& cfg) { return c.log_cd > 0 ? c.log_cd : cfg.log_constraint_degree; }

static inline bool col_is_free(int k) { return (k % AIR_GROUP) < 2; }
static inline int n_constraints(const ComponentSpec& c) {
    int n = 2;
    for (int k = 2; k < c.n_main; k++) if (!col_is_free(k)) n++;
    for (int k = 0; k < c.n_inter; k++) if (!col_is_free(k)) n++;
    return n;
}

// Stateless pseudo-random M31 (SplitMix64 finaliser) so CPU and device fill identical traces.
static inline u64 splitmix64(u64 x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static inline u32 synth_rand(u64 seed, u32 tree, u32 comp, u32 col, u32 row) {
    u64 x = splitmix64(seed ^ ((u64)tree << 60) ^ ((u64)comp << 52) ^ ((u64)col << 32) ^ (u64)row);
    u32 v = (u32)(x >> 33);
    return v == P ? 0 : v;
}

// Row `row` (natural coset order) of the synthetic trace of component `ci`.
// pre/main/inter must have room for n_pre / n_main / n_inter values.
static inline void synth_fill_row(const ComponentSpec& c, u32 ci, u64 seed, u64 inter_seed, u32 row,
                                  u32* pre, u32* main, u32* inter) {
    u32 N = 1u << c.log_size;
    for (int k = 0; k < c.n_pre; k++) {
        if (k == 0) pre[k] = row == 0;
        else if (k == 1) pre[k] = row == N - 1;
        else pre[k] = m31_reduce((u64)row * (u64)(k + 1) + 7u * (u64)k);
    }
    u32 s0 = synth_rand(seed, 1, ci, 0, 0xFFFFFFFFu), s1 = synth_rand(seed, 1, ci, 1, 0xFFFFFFFFu);
    // main[0] = s0 + row ; main[1] = s1 + row*s0 + row(row-1)/2   (prefix sum of main[0])
    main[0] = m31_add(s0, row % P);
    u32 tri = m31_reduce(((u64)row * (u64)(row ? row - 1 : 0)) / 2 % P);
    main[1] = m31_add(m31_add(s1, m31_mul(row % P, s0)), tri);
    for (int k = 2; k < c.n_main; k++)
        main[k] = col_is_free(k) ? synth_rand(seed, 1, ci, k, row) : m31_add(m31_sqr(main[k - 1]), m31_sqr(main[k - 2]));
    for (int k = 0; k < c.n_inter; k++)
        inter[k] = col_is_free(k) ? synth_rand(inter_seed, 2, ci, k, row) : m31_add(m31_sqr(inter[k - 1]), m31_sqr(inter[k - 2]));
}

// Generic constraint evaluation.  F is M31-like (domain) or QM31-like (OODS point).
//   V provides: pre(k), main(k), main_next(k) [k in {0,1}], inter(k), one().
//   acc(value) is called once per constraint in declaration order.
template <class V, class A>
static inline void eval_constraints(const ComponentSpec& c, const V& v, A& acc) {
    auto not_last = v.one() - v.pre(1);
    acc((v.main_next(0) - v.main(0) - v.one()) * not_last);
    acc((v.main_next(1) - v.main(1) - v.main(0)) * not_last);
    for (int k = 2; k < c.n_main; k++)
        if (!col_is_free(k)) acc(v.main(k) - v.main(k - 1) * v.main(k - 1) - v.main(k - 2) * v.main(k - 2));
    for (int k = 0; k < c.n_inter; k++)
        if (!col_is_free(k)) acc(v.inter(k) - v.inter(k - 1) * v.inter(k - 1) - v.inter(k - 2) * v.inter(k - 2));
}

struct FM { u32 v; };
static inline FM operator+(FM a, FM b) { return FM{m31_add(a.v, b.v)}; }
static inline FM operator-(FM a, FM b) { return FM{m31_sub(a.v, b.v)}; }
static inline FM operator*(FM a, FM b) { return FM{m31_mul(a.v, b.v)}; }
struct FQ { QM31 v; };
static inline FQ operator+(FQ a, FQ b) { return FQ{qm31_add(a.v, b.v)}; }
static inline FQ operator-(FQ a, FQ b) { return FQ{qm31_sub(a.v, b.v)}; }
static inline FQ operator*(FQ a, FQ b) { return FQ{qm31_mul(a.v, b.v)}; }

// stwo core/constraints.rs::coset_vanishing for a *canonic* coset of log size n: the rotation is
// the identity, so the value is double_x applied n-1 times to p.x.
static inline u32 canonic_coset_vanishing_m31(int n, Pt p) { u32 x = p.x; for (int i = 1; i < n; i++) x = double_x_m31(x); return x; }
static inline QM31 canonic_coset_vanishing_qm31(int n, QPt p) { QM31 x = p.x; for (int i = 1; i < n; i++) x = double_x_qm31(x); return x; }

// stwo-constraint-framework utils: index of the row `offset` trace-steps away, on a bit-reversed
// circle-domain evaluation of log size eval_log for a trace of log size domain_log.
static inline u32 offset_bit_reversed_circle_domain_index(u32 i, int domain_log, int eval_log, int offset) {
    int64_t prev = bit_reverse_index(i, eval_log);
    int64_t half = (int64_t)1 << (eval_log - 1);
    int64_t step = (int64_t)offset * ((int64_t)1 << (eval_log - domain_log - 1));
    if (prev < half) prev = ((prev + step) % half + half) % half;
    else prev = (((prev - step) % half + half) % half) + half;
    return bit_reverse_index((u32)prev, eval_log);
}

}  // namespace orc
