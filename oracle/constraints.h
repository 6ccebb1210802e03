// ORACLE — TEST INFRASTRUCTURE ONLY (see fields.h header).  PARITY UNPINNED.
//
// CPU restatement of the recorded-constraint evaluation: what stwo-constraint-framework's
// FrameworkComponent::evaluate_constraint_quotients_on_domain does when it drives the reference's AIR closure
// (reference prover/src/components/mod.rs:39-57, prover2/machine/src/framework/eval.rs:19-33) over the constraint domain —
// with the closure replaced by the straight-line program a recording EvalAtRow emits (include/nexus_hip.h, NX_C_*):
//   row_res = sum_j alpha_powers[j] * C_j(row);   acc[row] += row_res * denom_inv[row >> log_size]
// Masks follow offset_bit_reversed_circle_domain_index (air.h).  Plain loops, one row at a time.
#pragma once
#include <vector>
#include "air.h"

namespace orc {

struct CInstr { u32 op, dst, a, b; };
enum { C_LOAD = 0, C_CONST, C_ADD, C_SUB, C_MUL, C_NEG, C_CONSTE, C_ADDE, C_SUBE, C_MULE, C_MULEB, C_ADDEB, C_LOADE, C_CONSTRAINT_B, C_CONSTRAINT_E, C_FRAC, C_FRACB };

static inline void eval_constraint_program(const CInstr* prog, u32 n_instr, u32 n_regs, const u32* const* cols, const u32* econsts, const u32* pw,
                                           const u32* denom_inv, int log_size, int log_eval, u32* const acc4[4]) {
    std::vector<u32> R(n_regs);
    const u32 n = 1u << log_eval;
    for (u32 r = 0; r < n; r++) {
        QM31 sum = qm31_zero();
        u32 j = 0;
        for (u32 pc = 0; pc < n_instr; pc++) {
            const CInstr& in = prog[pc];
            auto E = [&](u32 i) { return qm31(R[i], R[i + 1], R[i + 2], R[i + 3]); };
            auto setE = [&](u32 i, QM31 v) { qm31_store(&R[i], v); };
            switch (in.op) {
            case C_LOAD: R[in.dst] = cols[in.a][offset_bit_reversed_circle_domain_index(r, log_size, log_eval, (int)in.b)]; break;
            case C_CONST: R[in.dst] = in.a; break;
            case C_ADD: R[in.dst] = m31_add(R[in.a], R[in.b]); break;
            case C_SUB: R[in.dst] = m31_sub(R[in.a], R[in.b]); break;
            case C_MUL: R[in.dst] = m31_mul(R[in.a], R[in.b]); break;
            case C_NEG: R[in.dst] = m31_neg(R[in.a]); break;
            case C_CONSTE: setE(in.dst, qm31_load(econsts + 4 * in.a)); break;
            case C_ADDE: setE(in.dst, qm31_add(E(in.a), E(in.b))); break;
            case C_SUBE: setE(in.dst, qm31_sub(E(in.a), E(in.b))); break;
            case C_MULE: setE(in.dst, qm31_mul(E(in.a), E(in.b))); break;
            case C_MULEB: setE(in.dst, qm31_mul_m31(E(in.a), R[in.b])); break;
            case C_ADDEB: setE(in.dst, qm31_add_m31(E(in.a), R[in.b])); break;
            case C_LOADE: { u32 rr = offset_bit_reversed_circle_domain_index(r, log_size, log_eval, (int)in.b);
                            setE(in.dst, qm31(cols[in.a][rr], cols[in.a + 1][rr], cols[in.a + 2][rr], cols[in.a + 3][rr])); break; }
            case C_CONSTRAINT_B: sum = qm31_add(sum, qm31_mul_m31(qm31_load(pw + 4 * j), R[in.a])); j++; break;
            case C_CONSTRAINT_E: sum = qm31_add(sum, qm31_mul(qm31_load(pw + 4 * j), E(in.a))); j++; break;
            default: break;
            }
        }
        QM31 res = qm31_mul_m31(sum, denom_inv[r >> log_size]);
        u32 w[4]; qm31_store(w, res);
        for (int k = 0; k < 4; k++) acc4[k][r] = m31_add(acc4[k][r], w[k]);
    }
}

// The interaction trace of a component from the relation entries its recorded AIR declares (include/nexus_hip.h nx_logup_program):
// the program's roots are fractions — C_FRAC E[a] / E[b], C_FRACB B[a] / E[b], of logup column (batch) dst, in batch order — and
// logup column j of a row is the sum of the fractions of batches <= j, each divided out by itself (LogupColGenerator::write_frac +
// finalize_col, one fraction at a time: the slow, literal way).  Rows are the trace domain's evaluations in bit-reversed
// circle-domain order; a row offset is +-k trace steps = +-k in natural coset order.
static inline void logup_program(const CInstr* prog, u32 n_instr, u32 n_regs, const u32* const* cols, const u32* econsts, int log_size, u32 n_logup_cols, u32* const* out) {
    std::vector<u32> R(n_regs);
    const u32 n = 1u << log_size;
    auto offset_row = [&](u32 r, int off) -> u32 {
        if (off == 0) return r;
        const u32 d = bit_reverse_index(r, log_size);
        const u32 c = d < n / 2 ? 2 * d : 2 * (n - 1 - d) + 1;                      // natural coset row of position r
        const u32 c2 = (u32)((c + (u32)off) & (n - 1));
        return bit_reverse_index(coset_index_to_circle_domain_index(c2, log_size), log_size);
    };
    for (u32 r = 0; r < n; r++) {
        QM31 run = qm31_zero();
        int cur = -1;
        auto flush = [&](int col) { u32 w[4]; qm31_store(w, run); for (int q = 0; q < 4; q++) out[4 * col + q][r] = w[q]; };
        for (u32 pc = 0; pc < n_instr; pc++) {
            const CInstr& in = prog[pc];
            auto E = [&](u32 i) { return qm31(R[i], R[i + 1], R[i + 2], R[i + 3]); };
            auto setE = [&](u32 i, QM31 v) { qm31_store(&R[i], v); };
            switch (in.op) {
            case C_LOAD: R[in.dst] = cols[in.a][offset_row(r, (int)in.b)]; break;
            case C_CONST: R[in.dst] = in.a; break;
            case C_ADD: R[in.dst] = m31_add(R[in.a], R[in.b]); break;
            case C_SUB: R[in.dst] = m31_sub(R[in.a], R[in.b]); break;
            case C_MUL: R[in.dst] = m31_mul(R[in.a], R[in.b]); break;
            case C_NEG: R[in.dst] = m31_neg(R[in.a]); break;
            case C_CONSTE: setE(in.dst, qm31_load(econsts + 4 * in.a)); break;
            case C_ADDE: setE(in.dst, qm31_add(E(in.a), E(in.b))); break;
            case C_SUBE: setE(in.dst, qm31_sub(E(in.a), E(in.b))); break;
            case C_MULE: setE(in.dst, qm31_mul(E(in.a), E(in.b))); break;
            case C_MULEB: setE(in.dst, qm31_mul_m31(E(in.a), R[in.b])); break;
            case C_ADDEB: setE(in.dst, qm31_add_m31(E(in.a), R[in.b])); break;
            case C_LOADE: { u32 rr = offset_row(r, (int)in.b); setE(in.dst, qm31(cols[in.a][rr], cols[in.a + 1][rr], cols[in.a + 2][rr], cols[in.a + 3][rr])); break; }
            case C_FRAC: case C_FRACB: {
                if (cur >= 0 && (int)in.dst != cur) flush(cur);
                cur = (int)in.dst;
                const QM31 num = in.op == C_FRAC ? E(in.a) : qm31(R[in.a], 0, 0, 0);
                run = qm31_add(run, qm31_mul(num, qm31_inv(E(in.b))));
                break;
            }
            default: break;
            }
        }
        if (cur >= 0) flush(cur);
        (void)n_logup_cols;
    }
}

}  // namespace orc
