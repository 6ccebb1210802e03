// ORACLE — TEST INFRASTRUCTURE ONLY (see fields.h header).  PARITY UNPINNED.
//
// Circle-FFT polynomial ops.  Restates Stwo CpuBackend `PolyOps`
// (prover/backend/cpu/circle.rs: precompute_twiddles / slow_precompute_twiddles, interpolate,
// evaluate, eval_at_point, extend; prover/backend/cpu/mod.rs: bit_reverse;
// core/poly/utils.rs: fold, domain_line_twiddles_from_tree; core/fft.rs: butterfly/ibutterfly)
// and the reference's own layout conversion
// (reference prover/src/trace/utils_external.rs:24-39, prover/src/trace/utils.rs:94-106).
// Call sites in the reference: prover/src/machine.rs:186 (twiddles), :209-263 (extend_evals/commit).
#pragma once
#include <vector>
#include <cassert>
#include "fields.h"

namespace orc {

// TwiddleTree for root coset `half_odds(root_log)` ( = CanonicCoset(root_log+1).half_coset() ).
struct Twiddles {
    int root_log;              // log size of the root half coset
    std::vector<u32> tw, itw;  // length 2^root_log each
};

static inline void bit_reverse_inplace(u32* v, int log) {
    u32 n = 1u << log;
    for (u32 i = 0; i < n; i++) { u32 j = bit_reverse_index(i, log); if (i < j) { u32 t = v[i]; v[i] = v[j]; v[j] = t; } }
}

// slow_precompute_twiddles: for each layer of the doubling tower, x-coords of the first half of the
// coset, bit-reversed; pad with 1.  itwiddles = element-wise inverses.
static inline Twiddles precompute_twiddles(int root_log) {
    Twiddles t; t.root_log = root_log;
    Coset c = coset_half_odds(root_log);
    t.tw.reserve((size_t)1 << root_log);
    for (int l = 0; l < root_log; l++) {
        size_t i0 = t.tw.size();
        u32 half = 1u << (c.log - 1);
        Pt p = pt_from_index(c.initial), s = pt_from_index(c.step);
        for (u32 i = 0; i < half; i++) { t.tw.push_back(p.x); p = pt_add(p, s); }
        bit_reverse_inplace(t.tw.data() + i0, c.log - 1);
        c = coset_double(c);
    }
    t.tw.push_back(1);
    t.itw.resize(t.tw.size());
    for (size_t i = 0; i < t.tw.size(); i++) t.itw[i] = m31_inv(t.tw[i]);
    return t;
}

// twiddle for line layer `layer` (1..n-1) of a size-2^n domain, butterfly group h.
static inline u32 line_twiddle(const std::vector<u32>& buf, int n, int layer, u32 h) {
    size_t TL = buf.size();
    return buf[TL - ((size_t)1 << (n - layer)) + h];
}
// circle-layer twiddle (layer 0), derived from the first line layer: chunks [x,y] -> [y,-y,-x,x].
static inline u32 circle_twiddle(const std::vector<u32>& buf, int n, u32 h) {
    u32 c = h >> 2;
    u32 x = line_twiddle(buf, n, 1, 2 * c), y = line_twiddle(buf, n, 1, 2 * c + 1);
    switch (h & 3) { case 0: return y; case 1: return m31_neg(y); case 2: return m31_neg(x); default: return x; }
}

static inline void butterfly(u32& v0, u32& v1, u32 t) { u32 tmp = m31_mul(v1, t); v1 = m31_sub(v0, tmp); v0 = m31_add(v0, tmp); }
static inline void ibutterfly(u32& v0, u32& v1, u32 t) { u32 tmp = v0; v0 = m31_add(tmp, v1); v1 = m31_mul(m31_sub(tmp, v1), t); }

// interpolate: bit-reversed evaluations on CanonicCoset(n).circle_domain() -> coefficients (in place).
static inline void interpolate(u32* v, int n, const Twiddles& T) {
    assert(n >= 1 && n - 1 <= T.root_log);
    u32 N = 1u << n;
    if (n == 1) {
        Pt p0 = pt_from_index(coset_half_odds(0).initial);
        u32 y_inv = m31_inv(p0.y);
        ibutterfly(v[0], v[1], y_inv);
        u32 ninv = m31_inv(2);
        v[0] = m31_mul(v[0], ninv); v[1] = m31_mul(v[1], ninv);
        return;
    }
    if (n == 2) {
        Pt p0 = pt_from_index(coset_half_odds(1).initial);
        u32 x_inv = m31_inv(p0.x), y_inv = m31_inv(p0.y);
        ibutterfly(v[0], v[1], y_inv);
        ibutterfly(v[2], v[3], m31_neg(y_inv));
        ibutterfly(v[0], v[2], x_inv);
        ibutterfly(v[1], v[3], x_inv);
        u32 ninv = m31_inv(4);
        for (int i = 0; i < 4; i++) v[i] = m31_mul(v[i], ninv);
        return;
    }
    for (u32 h = 0; h < N / 2; h++) ibutterfly(v[2 * h], v[2 * h + 1], circle_twiddle(T.itw, n, h));
    for (int layer = 1; layer < n; layer++) {
        u32 nh = 1u << (n - 1 - layer);
        for (u32 h = 0; h < nh; h++) {
            u32 t = line_twiddle(T.itw, n, layer, h);
            for (u32 l = 0; l < (1u << layer); l++) {
                u32 i0 = (h << (layer + 1)) + l, i1 = i0 + (1u << layer);
                ibutterfly(v[i0], v[i1], t);
            }
        }
    }
    u32 inv = m31_inv(N);  // N < P
    for (u32 i = 0; i < N; i++) v[i] = m31_mul(v[i], inv);
}

// evaluate: coefficients (2^n_coef, zero-extended to 2^n) -> bit-reversed evaluations on
// CanonicCoset(n).circle_domain().  `out` has 2^n entries.
static inline void evaluate(const u32* coeffs, int n_coef, u32* out, int n, const Twiddles& T) {
    assert(n >= n_coef && n >= 1 && n - 1 <= T.root_log);
    u32 N = 1u << n;
    for (u32 i = 0; i < N; i++) out[i] = i < (1u << n_coef) ? coeffs[i] : 0;
    u32* v = out;
    if (n == 1) {
        Pt p0 = pt_from_index(coset_half_odds(0).initial);
        butterfly(v[0], v[1], p0.y);
        return;
    }
    if (n == 2) {
        Pt p0 = pt_from_index(coset_half_odds(1).initial);
        butterfly(v[0], v[2], p0.x);
        butterfly(v[1], v[3], p0.x);
        butterfly(v[0], v[1], p0.y);
        butterfly(v[2], v[3], m31_neg(p0.y));
        return;
    }
    for (int layer = n - 1; layer >= 1; layer--) {
        u32 nh = 1u << (n - 1 - layer);
        for (u32 h = 0; h < nh; h++) {
            u32 t = line_twiddle(T.tw, n, layer, h);
            for (u32 l = 0; l < (1u << layer); l++) {
                u32 i0 = (h << (layer + 1)) + l, i1 = i0 + (1u << layer);
                butterfly(v[i0], v[i1], t);
            }
        }
    }
    for (u32 h = 0; h < N / 2; h++) butterfly(v[2 * h], v[2 * h + 1], circle_twiddle(T.tw, n, h));
}

// fold(values, factors): lhs + rhs * factor, recursively (core/poly/utils.rs::fold).
static inline QM31 fold_m31(const u32* values, size_t n, const QM31* factors) {
    if (n == 1) return qm31_from_m31(values[0]);
    QM31 l = fold_m31(values, n / 2, factors + 1), r = fold_m31(values + n / 2, n / 2, factors + 1);
    return qm31_add(l, qm31_mul(r, factors[0]));
}
static inline QM31 fold_qm31(const QM31* values, size_t n, const QM31* factors) {
    if (n == 1) return values[0];
    QM31 l = fold_qm31(values, n / 2, factors + 1), r = fold_qm31(values + n / 2, n / 2, factors + 1);
    return qm31_add(l, qm31_mul(r, factors[0]));
}

// eval_at_point: Σ_j c_j · y^{j0} x^{j1} π(x)^{j2} ...  (CpuBackend::eval_at_point)
static inline QM31 eval_at_point(const u32* coeffs, int n, QPt p) {
    if (n == 0) return qm31_from_m31(coeffs[0]);
    std::vector<QM31> m;
    m.push_back(p.y);
    QM31 x = p.x;
    for (int i = 1; i < n; i++) { m.push_back(x); x = double_x_qm31(x); }
    std::vector<QM31> r(m.rbegin(), m.rend());
    return fold_m31(coeffs, (size_t)1 << n, r.data());
}

// Direct (slow, obviously-right) evaluation of the FFT basis at an M31 point; used by self-tests.
static inline u32 eval_basis_at_m31_point(const u32* coeffs, int n, Pt p) {
    std::vector<u32> f;  // factor for bit k of j
    f.push_back(p.y);
    u32 x = p.x;
    for (int i = 1; i < n; i++) { f.push_back(x); x = double_x_m31(x); }
    u32 acc = 0;
    for (u32 j = 0; j < (1u << n); j++) {
        u32 b = coeffs[j];
        if (!b) continue;
        // index j: LSB (bit 0) <-> y ... but coefficient ordering in `fold` is MSB-first:
        // factors[0] (= π^{n-2}(x)) multiplies the top half => bit n-1 <-> π^{n-2}(x), bit 0 <-> y.
        for (int k = 0; k < n; k++) if ((j >> k) & 1) b = m31_mul(b, f[k]);
        acc = m31_add(acc, b);
    }
    return acc;
}

// reference prover/src/trace/utils_external.rs:24-39
static inline void coset_order_to_circle_domain_order(const u32* in, u32* out, int n) {
    u32 N = 1u << n, half = N / 2;
    for (u32 i = 0; i < half; i++) out[i] = in[i << 1];
    for (u32 i = 0; i < half; i++) out[half + i] = in[N - 1 - (i << 1)];
}
// stwo core/utils.rs::coset_index_to_circle_domain_index (used by the reference's test_order,
// prover/src/trace/utils.rs:117-128)
static inline u32 coset_index_to_circle_domain_index(u32 c, int n) {
    return (c & 1) ? (1u << n) - 1 - c / 2 : c / 2;
}
// reference prover/src/trace/utils.rs:94-106 finalize_columns (one column): natural coset order ->
// bit-reversed circle-domain order.
static inline void finalize_column(const u32* in, u32* out, int n) {
    coset_order_to_circle_domain_order(in, out, n);
    bit_reverse_inplace(out, n);
}

}  // namespace orc
