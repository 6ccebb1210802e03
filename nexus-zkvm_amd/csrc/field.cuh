// M31 / CM31 / QM31 arithmetic and the circle group for gfx950 kernels and the host driver.
// Field tower as restated by the reference's own spec (specification/zkvm-spec-3.0.pdf §3.1):
// M31 = F_p, p = 2^31-1; CM31 = M31[i]/(i^2+1); QM31 = CM31[u]/(u^2-2-i).
// All values are canonical ([0, p)) on every load and store: committed columns are hashed as raw
// u32 words, so `p` (== 0) must never be emitted (SURVEY.md §7 "Canonical representatives").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NX_HD __host__ __device__ __forceinline__

namespace nx {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr u32 P = 0x7fffffffu;

// In-register Mersenne reduction.  Instruction choice is measured (tools/ubench/valu_asm.hip, bfly_rates.hip):
// on gfx950 v_add/v_sub/v_and/v_lshr/v_cndmask issue in ~2 cycles per wave, v_min_u32, v_alignbit and every
// multiply in ~4, v_mad_u64_u32 in ~5.  So the conditional subtract is carry + select (v_sub_co, v_cndmask),
// not subtract + v_min, and a product is split at bit 31 by multiplying with the DOUBLED factor:
// a * (2b) = hi * 2^32 + 2 * lo  with  a*b = hi * 2^31 + lo  (1.5x the butterfly rate of and/alignbit/min).
NX_HD u32 umin32(u32 a, u32 b) { return a < b ? a : b; }
NX_HD u32 m_csub(u32 s) { u32 d; bool borrow = __builtin_usub_overflow(s, P, &d); return borrow ? s : d; }  // s < 2p
NX_HD u32 m_add(u32 a, u32 b) { return m_csub(a + b); }
NX_HD u32 m_sub(u32 a, u32 b) { u32 d; bool borrow = __builtin_usub_overflow(a, b, &d); return borrow ? d + P : d; }
NX_HD u32 m_neg(u32 a) { return a ? P - a : 0; }
// b2 = 2*b for a canonical b (e.g. a twiddle doubled once and reused)
NX_HD u32 m_mul_dbl(u32 a, u32 b2) {
    u64 p = (u64)a * (u64)b2;                 // v_mad_u64_u32
    return m_csub((u32)(p >> 32) + ((u32)p >> 1));
}
NX_HD u32 m_mul(u32 a, u32 b) { return m_mul_dbl(a, b << 1); }
NX_HD u32 m_sqr(u32 a) { return m_mul(a, a); }
NX_HD u32 m_reduce64(u64 x) {  // x < p^2
    return (u32)((((((x >> 31) + x + 1) >> 31) + x)) & P);
}
NX_HD u32 m_pow(u32 a, u32 e) {
    u32 r = 1;
    while (e) { if (e & 1) r = m_mul(r, a); a = m_sqr(a); e >>= 1; }
    return r;
}
// a^(p-2) = a^(2^31-3) by an addition chain: 30 squarings + 7 multiplications (the square-and-multiply walk of m_pow
// spends 30 + 29); same chain as Stwo's M31 inverse (pow2147483645) [upstream-recollection].
NX_HD u32 m_sqn(u32 a, int n) { for (int i = 0; i < n; i++) a = m_sqr(a); return a; }
NX_HD u32 m_inv(u32 a) {
    const u32 t2 = m_mul(m_sqr(a), a);              // a^(2^2-1)
    const u32 t4 = m_mul(m_sqn(t2, 2), t2);         // a^(2^4-1)
    const u32 t8 = m_mul(m_sqn(t4, 4), t4);         // a^(2^8-1)
    const u32 t16 = m_mul(m_sqn(t8, 8), t8);        // a^(2^16-1)
    const u32 t24 = m_mul(m_sqn(t16, 8), t8);       // a^(2^24-1)
    const u32 t28 = m_mul(m_sqn(t24, 4), t4);       // a^(2^28-1)
    const u32 t29 = m_mul(m_sqr(t28), a);           // a^(2^29-1)
    return m_mul(m_sqn(t29, 2), a);                 // a^(4(2^29-1)+1) = a^(2^31-3)
}
NX_HD u32 m_double_x(u32 x) { u32 s = m_sqr(x); return m_sub(m_add(s, s), 1); }

// Lazy dot-product accumulation: products of canonical values are added as raw 64-bit integers
// (one v_mad_u64_u32 each); acc_fold() brings the sum back below 2^34 and must run at least every
// 4 products (4*(p-1)^2 + 2^33 + 2^31 < 2^64).  acc_final() returns the canonical residue.
NX_HD u64 acc_mad(u64 acc, u32 a, u32 b) { return acc + (u64)a * (u64)b; }
NX_HD u64 acc_fold(u64 x) { return (x & (u64)P) + (x >> 31); }
NX_HD u32 acc_final(u64 x) {
    x = acc_fold(x);                       // < 2^34
    u32 s = ((u32)x & P) + (u32)(x >> 31); // < 2^31 + 8
    return m_csub(s);
}

struct CM31 { u32 a, b; };
NX_HD CM31 cm(u32 a, u32 b) { CM31 r; r.a = a; r.b = b; return r; }
NX_HD CM31 c_add(CM31 x, CM31 y) { return cm(m_add(x.a, y.a), m_add(x.b, y.b)); }
NX_HD CM31 c_sub(CM31 x, CM31 y) { return cm(m_sub(x.a, y.a), m_sub(x.b, y.b)); }
NX_HD CM31 c_neg(CM31 x) { return cm(m_neg(x.a), m_neg(x.b)); }
NX_HD CM31 c_mul(CM31 x, CM31 y) {
    return cm(m_sub(m_mul(x.a, y.a), m_mul(x.b, y.b)), m_add(m_mul(x.a, y.b), m_mul(x.b, y.a)));
}
NX_HD CM31 c_mul_m(CM31 x, u32 s) { return cm(m_mul(x.a, s), m_mul(x.b, s)); }
NX_HD CM31 c_mul_R(CM31 x) { return cm(m_sub(m_add(x.a, x.a), x.b), m_add(m_add(x.b, x.b), x.a)); }  // * (2+i)
NX_HD CM31 c_inv(CM31 x) {
    u32 d = m_inv(m_add(m_sqr(x.a), m_sqr(x.b)));
    return cm(m_mul(x.a, d), m_mul(m_neg(x.b), d));
}

struct QM31 { CM31 a, b; };
NX_HD QM31 qm(u32 a, u32 b, u32 c, u32 d) { QM31 r; r.a = cm(a, b); r.b = cm(c, d); return r; }
NX_HD QM31 q_from_m(u32 a) { return qm(a, 0, 0, 0); }
NX_HD QM31 q_zero() { return qm(0, 0, 0, 0); }
NX_HD QM31 q_one() { return qm(1, 0, 0, 0); }
NX_HD QM31 q_add(QM31 x, QM31 y) { QM31 r; r.a = c_add(x.a, y.a); r.b = c_add(x.b, y.b); return r; }
NX_HD QM31 q_sub(QM31 x, QM31 y) { QM31 r; r.a = c_sub(x.a, y.a); r.b = c_sub(x.b, y.b); return r; }
NX_HD QM31 q_neg(QM31 x) { QM31 r; r.a = c_neg(x.a); r.b = c_neg(x.b); return r; }
// (x.a + x.b u)(y.a + y.b u) with u^2 = 2 + i: every coordinate of the product is a sum of FOUR M31 products once y.b (2 + i) = (e, f) is formed —
//   r.a.a = xa ya - xb yb + xc e - xd f      r.a.b = xa yb + xb ya + xc f + xd e      r.b.a = xa yc - xb yd + xc ya - xd yb      r.b.b = xa yd + xb yc + xc yb + xd ya
// — so each is four raw 64-bit multiply-adds (a subtracted term enters as x (P - y): P - y <= P is a fine factor, the sum is reduced at the end)
// and ONE reduction, instead of four reduced products and three reduced additions: 16 v_mad_u64_u32 + 4 reductions against 16 + 16 + 20
// (round 6; the same values: tests/native/field_selftest.cpp — boundary values in every coordinate and random operands against the tower formula).
#ifndef NX_Q_MUL_NAIVE
NX_HD QM31 q_mul(QM31 x, QM31 y) {
    const CM31 r = c_mul_R(y.b);                                        // (e, f) = y.b (2 + i)
    const u32 nyb = P - y.a.b, nyd = P - y.b.b, nf = P - r.b;
    QM31 o;
    o.a.a = acc_final(acc_mad(acc_mad(acc_mad((u64)x.a.a * y.a.a, x.a.b, nyb), x.b.a, r.a), x.b.b, nf));
    o.a.b = acc_final(acc_mad(acc_mad(acc_mad((u64)x.a.a * y.a.b, x.a.b, y.a.a), x.b.a, r.b), x.b.b, r.a));
    o.b.a = acc_final(acc_mad(acc_mad(acc_mad((u64)x.a.a * y.b.a, x.a.b, nyd), x.b.a, y.a.a), x.b.b, nyb));
    o.b.b = acc_final(acc_mad(acc_mad(acc_mad((u64)x.a.a * y.b.b, x.a.b, y.b.a), x.b.a, y.a.b), x.b.b, y.a.a));
    return o;
}
#else
NX_HD QM31 q_mul(QM31 x, QM31 y) {
    QM31 r;
    r.a = c_add(c_mul(x.a, y.a), c_mul_R(c_mul(x.b, y.b)));
    r.b = c_add(c_mul(x.a, y.b), c_mul(x.b, y.a));
    return r;
}
#endif
NX_HD QM31 q_sqr(QM31 x) { return q_mul(x, x); }
// The two CM31 steps of a QM31 inverse, each coordinate one lazy sum of at most four products and one reduction (round 6, like q_mul):
//   q_norm_cm(x) = x.a^2 - (2 + i) x.b^2:   .a = a0^2 - a1^2 - 2 b0^2 + 2 b1^2 + 2 b0 b1 = a0 a0 + a1 (P - a1) + 2b0 (P - b0) + 2(b0 + b1) b1
//                                           .b = 2 a0 a1 - 4 b0 b1 - b0^2 + b1^2          = 2a0 a1 + 2b0 (P - 2b1) + b0 (P - b0) + b1 b1
//   q_conj_times(x, d) = (x.a d, -x.b d):   the inverse is q_conj_times(x, conj(D) / |D|^2) with D = q_norm_cm(x)
#ifndef NX_Q_MUL_NAIVE
NX_HD CM31 q_norm_cm(QM31 x) {
    const u32 a0 = x.a.a, a1 = x.a.b, b0 = x.b.a, b1 = x.b.b;
    const u32 a0d = m_add(a0, a0), b0d = m_add(b0, b0), b1d = m_add(b1, b1), sd = m_add(b0d, b1d), nb0 = P - b0;
    return cm(acc_final(acc_mad(acc_mad(acc_mad((u64)a0 * a0, a1, P - a1), b0d, nb0), sd, b1)),
              acc_final(acc_mad(acc_mad(acc_mad((u64)a0d * a1, b0d, P - b1d), b0, nb0), b1, b1)));
}
NX_HD QM31 q_conj_times(QM31 x, CM31 d) {
    const u32 nd0 = P - d.a, nd1 = P - d.b;
    QM31 r;
    r.a.a = acc_final(acc_mad((u64)x.a.a * d.a, x.a.b, nd1)); r.a.b = acc_final(acc_mad((u64)x.a.a * d.b, x.a.b, d.a));
    r.b.a = acc_final(acc_mad((u64)x.b.a * nd0, x.b.b, d.b)); r.b.b = acc_final(acc_mad((u64)x.b.a * nd1, x.b.b, nd0));
    return r;
}
// acc + (x.a d, -x.b d): a fraction num / x added to a running sum when d already carries num / |D|^2 — the sum's word rides in the accumulator
NX_HD QM31 q_conj_times_add(QM31 acc, QM31 x, CM31 d) {
    const u32 nd0 = P - d.a, nd1 = P - d.b;
    QM31 r;
    r.a.a = acc_final(acc_mad(acc_mad((u64)acc.a.a, x.a.a, d.a), x.a.b, nd1)); r.a.b = acc_final(acc_mad(acc_mad((u64)acc.a.b, x.a.a, d.b), x.a.b, d.a));
    r.b.a = acc_final(acc_mad(acc_mad((u64)acc.b.a, x.b.a, nd0), x.b.b, d.b)); r.b.b = acc_final(acc_mad(acc_mad((u64)acc.b.b, x.b.a, nd1), x.b.b, nd0));
    return r;
}
#else
NX_HD QM31 q_conj_times_add(QM31 acc, QM31 x, CM31 d) { QM31 r; r.a = c_mul(x.a, d); r.b = c_mul(c_neg(x.b), d); return q_add(acc, r); }
NX_HD CM31 q_norm_cm(QM31 x) { return c_sub(c_mul(x.a, x.a), c_mul_R(c_mul(x.b, x.b))); }
NX_HD QM31 q_conj_times(QM31 x, CM31 d) { QM31 r; r.a = c_mul(x.a, d); r.b = c_mul(c_neg(x.b), d); return r; }
#endif
NX_HD QM31 q_mul_m(QM31 x, u32 s) { QM31 r; r.a = c_mul_m(x.a, s); r.b = c_mul_m(x.b, s); return r; }
NX_HD QM31 q_mul_c(QM31 x, CM31 s) { QM31 r; r.a = c_mul(x.a, s); r.b = c_mul(x.b, s); return r; }
NX_HD QM31 q_conj(QM31 x) { QM31 r; r.a = x.a; r.b = c_neg(x.b); return r; }
NX_HD QM31 q_inv(QM31 x) { return q_conj_times(x, c_inv(q_norm_cm(x))); }
NX_HD bool q_eq(QM31 x, QM31 y) { return x.a.a == y.a.a && x.a.b == y.a.b && x.b.a == y.b.a && x.b.b == y.b.b; }
NX_HD bool q_is_zero(QM31 x) { return !(x.a.a | x.a.b | x.b.a | x.b.b); }
NX_HD QM31 q_double_x(QM31 x) { QM31 s = q_sqr(x); return q_sub(q_add(s, s), q_one()); }
NX_HD void q_store(u32* o, QM31 x) { o[0] = x.a.a; o[1] = x.a.b; o[2] = x.b.a; o[3] = x.b.b; }
NX_HD QM31 q_load(const u32* i) { return qm(i[0], i[1], i[2], i[3]); }
NX_HD QM31 q_pow(QM31 a, u64 e) {
    QM31 r = q_one();
    while (e) { if (e & 1) r = q_mul(r, a); a = q_sqr(a); e >>= 1; }
    return r;
}

// ---- circle group (generator (2, 1268011823), order 2^31) ----
struct Pt { u32 x, y; };
NX_HD Pt pt_add(Pt p, Pt q) {
    Pt r;
    r.x = m_sub(m_mul(p.x, q.x), m_mul(p.y, q.y));
    r.y = m_add(m_mul(p.x, q.y), m_mul(p.y, q.x));
    return r;
}
NX_HD Pt pt_from_index(u32 idx) {
    Pt res; res.x = 1; res.y = 0;
    Pt cur; cur.x = 2; cur.y = 1268011823u;
    idx &= 0x7fffffffu;
    while (idx) { if (idx & 1) res = pt_add(res, cur); cur = pt_add(cur, cur); idx >>= 1; }
    return res;
}
// G * 2^i for i = 0 .. 30: a point from its index as a sum over the SET bits only (pt_from_index doubles its way through all 31): what a kernel uses
// when every lane needs its own domain point (a by-value kernel argument of 248 bytes, built on the host by gen_table()).
struct GenTable { u32 x[31], y[31]; };
inline GenTable gen_table() { GenTable g; Pt cur; cur.x = 2; cur.y = 1268011823u; for (int i = 0; i < 31; i++) { g.x[i] = cur.x; g.y[i] = cur.y; cur = pt_add(cur, cur); } return g; }
NX_HD Pt pt_from_index_tbl(const GenTable& g, u32 idx) {
    Pt res; res.x = 1; res.y = 0;
    idx &= 0x7fffffffu;
    for (int i = 0; idx; i++, idx >>= 1)
        if (idx & 1) { Pt c; c.x = g.x[i]; c.y = g.y[i]; res = pt_add(res, c); }
    return res;
}
struct QPt { QM31 x, y; };
NX_HD QPt qpt_add(QPt p, QPt q) {
    QPt r;
    r.x = q_sub(q_mul(p.x, q.x), q_mul(p.y, q.y));
    r.y = q_add(q_mul(p.x, q.y), q_mul(p.y, q.x));
    return r;
}

// index of Coset::half_odds(log).at(i) and of CanonicCoset(log).circle_domain().at(i)
NX_HD u32 half_odds_index(int log, u32 i) { return ((1u << (31 - log - 2)) + (i << (31 - log))) & 0x7fffffffu; }
NX_HD u32 circle_domain_index(int log, u32 i) {
    u32 half = 1u << (log - 1);
    if (i < half) return half_odds_index(log - 1, i);
    return (0u - half_odds_index(log - 1, i - half)) & 0x7fffffffu;
}
NX_HD u32 bitrev(u32 i, int log) {
#if defined(__HIP_DEVICE_COMPILE__)
    return log ? (__brev(i) >> (32 - log)) : i;
#else
    u32 r = 0;
    for (int k = 0; k < log; k++) r |= ((i >> k) & 1u) << (log - 1 - k);
    return r;
#endif
}

}  // namespace nx
