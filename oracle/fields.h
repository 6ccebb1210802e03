// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the product path
// (nexus-zkvm_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
//
// PARITY UNPINNED: the arithmetic of the hot path lives in the third-party crate
// `stwo @ 0790eba` (reference Cargo.toml:39-48), which is NOT vendored under /root/reference and
// cannot be built here (no Rust).  This file restates the *published* Stwo CpuBackend algorithms
// (core/fields/{m31,cm31,qm31}.rs, core/circle.rs) from their public definition, anchored on the
// reference's own restatement of the field tower (specification/zkvm-spec-3.0.pdf §3.1, p.14) and
// on the reference call sites (prover/src/machine.rs:184-290).
#pragma once
#include <cstdint>
#include <cstddef>

namespace orc {

typedef uint32_t u32;
typedef uint64_t u64;

static const u32 P = 0x7fffffffu;  // 2^31 - 1

// ---- M31 (stwo core/fields/m31.rs) ----
static inline u32 m31_reduce(u64 x) {  // valid for x < P^2 (M31::reduce)
    return (u32)((((((x >> 31) + x + 1) >> 31) + x)) & P);
}
static inline u32 m31_add(u32 a, u32 b) { u32 s = a + b; return s >= P ? s - P : s; }
static inline u32 m31_sub(u32 a, u32 b) { return a >= b ? a - b : a + P - b; }
static inline u32 m31_neg(u32 a) { return a ? P - a : 0; }
static inline u32 m31_mul(u32 a, u32 b) { return m31_reduce((u64)a * (u64)b); }
static inline u32 m31_sqr(u32 a) { return m31_mul(a, a); }
static inline u32 m31_pow(u32 a, u64 e) {
    u32 r = 1;
    while (e) { if (e & 1) r = m31_mul(r, a); a = m31_sqr(a); e >>= 1; }
    return r;
}
static inline u32 m31_inv(u32 a) { return m31_pow(a, P - 2); }  // a != 0

// ---- CM31 = M31[i]/(i^2+1) (stwo core/fields/cm31.rs) ----
struct CM31 { u32 a, b; };
static inline CM31 cm31(u32 a, u32 b) { CM31 r = {a, b}; return r; }
static inline CM31 cm31_add(CM31 x, CM31 y) { return cm31(m31_add(x.a, y.a), m31_add(x.b, y.b)); }
static inline CM31 cm31_sub(CM31 x, CM31 y) { return cm31(m31_sub(x.a, y.a), m31_sub(x.b, y.b)); }
static inline CM31 cm31_neg(CM31 x) { return cm31(m31_neg(x.a), m31_neg(x.b)); }
static inline CM31 cm31_mul(CM31 x, CM31 y) {
    return cm31(m31_sub(m31_mul(x.a, y.a), m31_mul(x.b, y.b)),
                m31_add(m31_mul(x.a, y.b), m31_mul(x.b, y.a)));
}
static inline CM31 cm31_mul_m31(CM31 x, u32 s) { return cm31(m31_mul(x.a, s), m31_mul(x.b, s)); }
static inline CM31 cm31_inv(CM31 x) {  // (a - bi) / (a^2 + b^2)
    u32 d = m31_inv(m31_add(m31_sqr(x.a), m31_sqr(x.b)));
    return cm31(m31_mul(x.a, d), m31_mul(m31_neg(x.b), d));
}
static inline bool cm31_eq(CM31 x, CM31 y) { return x.a == y.a && x.b == y.b; }

// ---- QM31 = CM31[u]/(u^2 - (2+i)) (stwo core/fields/qm31.rs; spec §3.1) ----
struct QM31 { CM31 a, b; };
static inline QM31 qm31(u32 a, u32 b, u32 c, u32 d) { QM31 r = {{a, b}, {c, d}}; return r; }
static inline QM31 qm31_from_m31(u32 a) { return qm31(a, 0, 0, 0); }
static inline QM31 qm31_zero() { return qm31(0, 0, 0, 0); }
static inline QM31 qm31_one() { return qm31(1, 0, 0, 0); }
static inline QM31 qm31_add(QM31 x, QM31 y) { QM31 r = {cm31_add(x.a, y.a), cm31_add(x.b, y.b)}; return r; }
static inline QM31 qm31_sub(QM31 x, QM31 y) { QM31 r = {cm31_sub(x.a, y.a), cm31_sub(x.b, y.b)}; return r; }
static inline QM31 qm31_neg(QM31 x) { QM31 r = {cm31_neg(x.a), cm31_neg(x.b)}; return r; }
static inline CM31 cm31_mul_R(CM31 x) {  // x * (2 + i)
    return cm31(m31_sub(m31_add(x.a, x.a), x.b), m31_add(m31_add(x.b, x.b), x.a));
}
static inline QM31 qm31_mul(QM31 x, QM31 y) {
    // (a + bu)(c + du) = (ac + R bd) + (ad + bc)u
    QM31 r = {cm31_add(cm31_mul(x.a, y.a), cm31_mul_R(cm31_mul(x.b, y.b))),
              cm31_add(cm31_mul(x.a, y.b), cm31_mul(x.b, y.a))};
    return r;
}
static inline QM31 qm31_sqr(QM31 x) { return qm31_mul(x, x); }
static inline QM31 qm31_mul_m31(QM31 x, u32 s) { QM31 r = {cm31_mul_m31(x.a, s), cm31_mul_m31(x.b, s)}; return r; }
static inline QM31 qm31_mul_cm31(QM31 x, CM31 s) { QM31 r = {cm31_mul(x.a, s), cm31_mul(x.b, s)}; return r; }
static inline QM31 qm31_add_m31(QM31 x, u32 s) { QM31 r = x; r.a.a = m31_add(r.a.a, s); return r; }
static inline QM31 qm31_conj(QM31 x) { QM31 r = {x.a, cm31_neg(x.b)}; return r; }  // ComplexConjugate
static inline QM31 qm31_inv(QM31 x) {
    // denom = a^2 - R b^2 ; inv = (a - bu) / denom
    CM31 b2 = cm31_mul(x.b, x.b);
    CM31 denom = cm31_sub(cm31_mul(x.a, x.a), cm31_mul_R(b2));
    CM31 di = cm31_inv(denom);
    QM31 r = {cm31_mul(x.a, di), cm31_mul(cm31_neg(x.b), di)};
    return r;
}
static inline bool qm31_eq(QM31 x, QM31 y) { return cm31_eq(x.a, y.a) && cm31_eq(x.b, y.b); }
static inline bool qm31_is_zero(QM31 x) { return !(x.a.a | x.a.b | x.b.a | x.b.b); }
static inline QM31 qm31_pow(QM31 a, u64 e) {
    QM31 r = qm31_one();
    while (e) { if (e & 1) r = qm31_mul(r, a); a = qm31_sqr(a); e >>= 1; }
    return r;
}
static inline void qm31_store(u32* out, QM31 x) { out[0] = x.a.a; out[1] = x.a.b; out[2] = x.b.a; out[3] = x.b.b; }
static inline QM31 qm31_load(const u32* in) { return qm31(in[0], in[1], in[2], in[3]); }

// ---- Circle group over M31 (stwo core/circle.rs) ----
// Generator (2, 1268011823), order 2^31.  Point indices are integers mod 2^31.
static const u32 CIRCLE_GEN_X = 2, CIRCLE_GEN_Y = 1268011823u;
static const int CIRCLE_LOG_ORDER = 31;
static const u32 CIRCLE_ORDER_MASK = 0x7fffffffu;

struct Pt { u32 x, y; };
static inline Pt pt_add(Pt p, Pt q) {
    Pt r = {m31_sub(m31_mul(p.x, q.x), m31_mul(p.y, q.y)), m31_add(m31_mul(p.x, q.y), m31_mul(p.y, q.x))};
    return r;
}
static inline Pt pt_conj(Pt p) { Pt r = {p.x, m31_neg(p.y)}; return r; }
static inline Pt pt_double(Pt p) { return pt_add(p, p); }
static inline Pt pt_from_index(u32 idx) {  // CirclePointIndex::to_point — double-and-add
    Pt res = {1, 0};
    Pt cur = {CIRCLE_GEN_X, CIRCLE_GEN_Y};
    idx &= CIRCLE_ORDER_MASK;
    while (idx) { if (idx & 1) res = pt_add(res, cur); cur = pt_double(cur); idx >>= 1; }
    return res;
}
static inline u32 double_x_m31(u32 x) { return m31_sub(m31_add(m31_sqr(x), m31_sqr(x)), 1); }

// Secure-field circle points.
struct QPt { QM31 x, y; };
static inline QPt qpt_add(QPt p, QPt q) {
    QPt r = {qm31_sub(qm31_mul(p.x, q.x), qm31_mul(p.y, q.y)), qm31_add(qm31_mul(p.x, q.y), qm31_mul(p.y, q.x))};
    return r;
}
static inline QPt qpt_from_pt(Pt p) { QPt r = {qm31_from_m31(p.x), qm31_from_m31(p.y)}; return r; }
static inline QM31 double_x_qm31(QM31 x) { QM31 s = qm31_sqr(x); return qm31_sub(qm31_add(s, s), qm31_one()); }

// ---- Coset / CircleDomain (stwo core/circle.rs, core/poly/circle/{canonic,domain}.rs) ----
struct Coset { u32 initial, step; int log; };  // indices mod 2^31
static inline u32 subgroup_gen(int log) { return 1u << (CIRCLE_LOG_ORDER - log); }
static inline Coset coset_odds(int log) { Coset c = {subgroup_gen(log + 1), subgroup_gen(log), log}; return c; }
static inline Coset coset_half_odds(int log) { Coset c = {subgroup_gen(log + 2), subgroup_gen(log), log}; return c; }
static inline Coset coset_double(Coset c) {
    Coset r = {(c.initial * 2) & CIRCLE_ORDER_MASK, (c.step * 2) & CIRCLE_ORDER_MASK, c.log - 1};
    return r;
}
static inline u32 coset_index_at(Coset c, u32 i) { return (c.initial + c.step * i) & CIRCLE_ORDER_MASK; }
static inline Pt coset_at(Coset c, u32 i) { return pt_from_index(coset_index_at(c, i)); }
// CanonicCoset(log).circle_domain() == CircleDomain(half_odds(log-1)).  at(i) for i in [0, 2^log).
static inline u32 circle_domain_index_at(int log, u32 i) {
    Coset h = coset_half_odds(log - 1);
    u32 half = 1u << (log - 1);
    if (i < half) return coset_index_at(h, i);
    return (0u - coset_index_at(h, i - half)) & CIRCLE_ORDER_MASK;
}
static inline Pt circle_domain_at(int log, u32 i) { return pt_from_index(circle_domain_index_at(log, i)); }

static inline u32 bit_reverse_index(u32 i, int log) {
    if (log == 0) return i;
    u32 r = 0;
    for (int k = 0; k < log; k++) r |= ((i >> k) & 1u) << (log - 1 - k);
    return r;
}

}  // namespace orc
