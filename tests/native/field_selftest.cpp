// Host-side self-test of nexus-zkvm_amd/csrc/field.cuh (the NX_HD functions run on the CPU too): the lazy four-product q_mul against the
// textbook tower formula — (a + b u)(c + d u) = (ac + (2 + i) bd) + (ad + bc) u over CM31 products built from reduced M31 operations —
// on every combination of boundary values in the eight coordinates and on random operands.  Built by tests/test_field_host_cpu.py with hipcc
// (no GPU needed).  Exit code 0 = all equal and canonical.
#include "../../nexus-zkvm_amd/csrc/field.cuh"
#include <stdio.h>
using namespace nx;
static CM31 ref_c_mul(CM31 x, CM31 y) { return cm(m_sub(m_mul(x.a, y.a), m_mul(x.b, y.b)), m_add(m_mul(x.a, y.b), m_mul(x.b, y.a))); }
static QM31 ref_q_mul(QM31 x, QM31 y) {
    QM31 r;
    r.a = c_add(ref_c_mul(x.a, y.a), c_mul_R(ref_c_mul(x.b, y.b)));
    r.b = c_add(ref_c_mul(x.a, y.b), ref_c_mul(x.b, y.a));
    return r;
}
static CM31 ref_norm_cm(QM31 x) { return c_sub(ref_c_mul(x.a, x.a), c_mul_R(ref_c_mul(x.b, x.b))); }
static QM31 ref_conj_times(QM31 x, CM31 d) { QM31 r; r.a = ref_c_mul(x.a, d); r.b = ref_c_mul(c_neg(x.b), d); return r; }
static bool canonical(QM31 x) { return x.a.a < P && x.a.b < P && x.b.a < P && x.b.b < P; }
int main() {
    const u32 edge[] = {0u, 1u, 2u, P - 1, P - 2, 1u << 30, 0x55555555u & P, 1268011823u};
    const int E = sizeof edge / sizeof edge[0];
    unsigned long long n = 0, bad = 0;
    for (int i0 = 0; i0 < E; i0++) for (int i1 = 0; i1 < E; i1++) for (int i2 = 0; i2 < E; i2++) for (int i3 = 0; i3 < E; i3++)
        for (int j0 = 0; j0 < E; j0++) for (int j1 = 0; j1 < E; j1++) for (int j2 = 0; j2 < E; j2++) for (int j3 = 0; j3 < E; j3++) {
            const QM31 x = qm(edge[i0], edge[i1], edge[i2], edge[i3]), y = qm(edge[j0], edge[j1], edge[j2], edge[j3]);
            const QM31 a = q_mul(x, y), b = ref_q_mul(x, y);
            n++; if (!q_eq(a, b) || !canonical(a)) bad++;
            // the lazy steps of the inverse: D = x.a^2 - (2 + i) x.b^2 and (x.a d, -x.b d), d taken from y
            const CM31 d1 = q_norm_cm(x), d2 = ref_norm_cm(x);
            if (d1.a != d2.a || d1.b != d2.b || d1.a >= P || d1.b >= P) bad++;
            const QM31 c1 = q_conj_times(x, y.a), c2 = ref_conj_times(x, y.a);
            if (!q_eq(c1, c2) || !canonical(c1)) bad++;
            const QM31 e1 = q_conj_times_add(y, x, y.b), e2 = q_add(y, ref_conj_times(x, y.b));
            if (!q_eq(e1, e2) || !canonical(e1)) bad++;
        }
    u64 s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (u32)((s >> 16) % P); };
    for (int k = 0; k < 2000000; k++) {
        const QM31 x = qm(rnd(), rnd(), rnd(), rnd()), y = qm(rnd(), rnd(), rnd(), rnd());
        const QM31 a = q_mul(x, y), b = ref_q_mul(x, y);
        n++; if (!q_eq(a, b) || !canonical(a)) bad++;
        { const CM31 d1 = q_norm_cm(x), d2 = ref_norm_cm(x); if (d1.a != d2.a || d1.b != d2.b) bad++; const QM31 c1 = q_conj_times(x, y.b), c2 = ref_conj_times(x, y.b); if (!q_eq(c1, c2) || !canonical(c1)) bad++; }
        // ring laws that the prover relies on: x * x^-1 = 1, (x y) z = x (y z)
        if (k < 2000 && !q_is_zero(x)) { if (!q_eq(q_mul(x, q_inv(x)), q_one())) bad++; const QM31 z = qm(rnd(), rnd(), rnd(), rnd()); if (!q_eq(q_mul(q_mul(x, y), z), q_mul(x, q_mul(y, z)))) bad++; }
    }
    printf("%llu products, %llu mismatches\n", n, bad);
    return bad ? 1 : 0;
}
