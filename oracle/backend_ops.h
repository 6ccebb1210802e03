// ORACLE — TEST INFRASTRUCTURE ONLY (see fields.h header).  PARITY UNPINNED.
//
// CPU restatement of the Stwo CpuBackend operations behind the remaining Backend supertraits (called by
// stwo::prover::prove at reference prover/src/machine.rs:286-290, prover2/machine/src/prove.rs:124-128) [upstream-recollection]:
//   core/fields/mod.rs::batch_inverse_in_place / FieldOps::batch_inverse   Montgomery's trick: prefix products, ONE inversion,
//                                                                           back-substitution (a different route than the device's
//                                                                           one-inverse-per-element: the results must agree)
//   prover/backend/cpu/accumulation.rs::{accumulate, generate_secure_powers}
//   prover/backend/cpu/fri.rs::{decompose, decomposition_coefficient}
//   prover/backend/cpu/blake2s.rs::commit_on_layer                          one Merkle layer from the layer below + the injected columns
#pragma once
#include <vector>
#include "fields.h"
#include "merkle.h"

namespace orc {

static inline void batch_inverse_m31(const u32* src, u32* dst, size_t n) {
    if (!n) return;
    std::vector<u32> prefix(n);
    u32 run = 1;
    for (size_t i = 0; i < n; i++) { prefix[i] = run; run = m31_mul(run, src[i]); }
    u32 inv = m31_inv(run);
    for (size_t i = n; i-- > 0;) { u32 s = src[i]; dst[i] = m31_mul(inv, prefix[i]); inv = m31_mul(inv, s); }
}
static inline void batch_inverse_qm31(const u32* const src4[4], u32* const dst4[4], size_t n) {
    if (!n) return;
    std::vector<QM31> prefix(n);
    QM31 run = qm31_one();
    for (size_t i = 0; i < n; i++) { prefix[i] = run; run = qm31_mul(run, qm31(src4[0][i], src4[1][i], src4[2][i], src4[3][i])); }
    QM31 inv = qm31_inv(run);
    for (size_t i = n; i-- > 0;) {
        QM31 s = qm31(src4[0][i], src4[1][i], src4[2][i], src4[3][i]);
        QM31 r = qm31_mul(inv, prefix[i]);
        inv = qm31_mul(inv, s);
        u32 w[4]; qm31_store(w, r);
        for (int q = 0; q < 4; q++) dst4[q][i] = w[q];
    }
}

static inline void secure_accumulate(u32* const dst4[4], const u32* const src4[4], size_t n) {
    for (int q = 0; q < 4; q++) for (size_t i = 0; i < n; i++) dst4[q][i] = m31_add(dst4[q][i], src4[q][i]);
}
static inline void generate_secure_powers(QM31 felt, size_t n, u32* out) {
    QM31 a = qm31_one();
    for (size_t i = 0; i < n; i++) { qm31_store(out + 4 * i, a); a = qm31_mul(a, felt); }
}

// FriOps::decompose: eval (bit-reversed circle evaluation, 2^log values) = g + lambda * v_n with v_n = +1 on the first half, -1 on
// the second; lambda = (sum first half - sum second half) / 2^log.
static inline QM31 fri_decompose(const u32* const src4[4], int log, u32* const g4[4]) {
    const size_t n = (size_t)1 << log, half = n / 2;
    QM31 a = qm31_zero(), b = qm31_zero();
    for (size_t i = 0; i < half; i++) a = qm31_add(a, qm31(src4[0][i], src4[1][i], src4[2][i], src4[3][i]));
    for (size_t i = half; i < n; i++) b = qm31_add(b, qm31(src4[0][i], src4[1][i], src4[2][i], src4[3][i]));
    const QM31 lambda = qm31_mul_m31(qm31_sub(a, b), m31_inv((u32)(n % P)));
    for (size_t i = 0; i < n; i++) {
        QM31 x = qm31(src4[0][i], src4[1][i], src4[2][i], src4[3][i]);
        QM31 g = i < half ? qm31_sub(x, lambda) : qm31_add(x, lambda);
        u32 w[4]; qm31_store(w, g);
        for (int q = 0; q < 4; q++) g4[q][i] = w[q];
    }
    return lambda;
}

// MerkleOps::commit_on_layer: prev = the 2^(log+1) nodes below (8 words each) or NULL; out = 2^log nodes
static inline void commit_on_layer(int log, const u32* prev, const u32* const* cols, size_t n_cols, int mode, u32* out) {
    std::vector<u32> vals(n_cols);
    for (size_t i = 0; i < ((size_t)1 << log); i++) {
        for (size_t c = 0; c < n_cols; c++) vals[c] = cols[c][i];
        Hash l, r;
        if (prev) { memcpy(l.w, prev + 16 * i, 32); memcpy(r.w, prev + 16 * i + 8, 32); }
        Hash h = hash_node(prev ? &l : nullptr, prev ? &r : nullptr, vals.data(), n_cols, mode);
        memcpy(out + 8 * i, h.w, 32);
    }
}

}  // namespace orc
