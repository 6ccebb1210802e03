// ORACLE — TEST INFRASTRUCTURE ONLY (see fields.h header).  PARITY UNPINNED.
//
// stwo::prover::prove / core::verifier::verify for components given as RECORDED constraint programs (constraints.h) instead
// of the built-in synthetic machine: the CPU statement of what libnexus_hip's nx_prover_* session does.  A component is what
// FrameworkComponent<E> is to Stwo (reference prover/src/components/mod.rs:15-57, prover2/machine/src/framework/mod.rs):
// a trace log size, the columns it claims in the three trace trees (TraceLocationAllocator), the mask offsets each column is
// sampled at (InfoEvaluator, reference prover/src/components/mod.rs:59-67) and its constraints.  The session objects mirror
// CommitmentSchemeProver / CommitmentSchemeVerifier: the caller drives the transcript prefix (reference machine.rs:198-263 /
// :299-485: mix, commit tree, draw lookup elements, mix claimed sums, commit tree) and then calls prove / verify.
#pragma once
#include <memory>
#include <algorithm>
#include "pcs.h"
#include "constraints.h"

namespace orc {

struct GComponent {
    int log_size = 0;
    int log_cd = 0;                                 // log constraint-degree bound of this component; 0 = cfg.log_constraint_degree
    std::vector<CInstr> prog; u32 n_regs = 0;
    std::vector<u32> econsts;                       // 4 words per secure constant (lookup elements, claimed sums, ...)
    size_t n_constraints = 0;
    std::vector<std::pair<int, int>> cols;          // component column -> (tree, column in tree)
    std::vector<std::vector<int>> masks;            // component column -> row offsets sampled (units of the trace step)
};
struct GAir { std::vector<GComponent> comps; };

// flat u32 encoding used over the C boundary (tests/oracle_lib.py builds it):
//   n_comps, then per component: log_size, n_instr, n_regs, n_econsts, n_constraints, n_cols, n_mask_total, log_cd,
//   instrs[4*n_instr], econsts[4*n_econsts], col_tree[n_cols], col_index[n_cols], mask_count[n_cols], mask_offsets[n_mask_total] (int32)
static inline bool gair_decode(const u32* w, size_t n, GAir& air) {
    size_t i = 0;
    auto take = [&](size_t k) -> const u32* { if (i + k > n) return nullptr; const u32* p = w + i; i += k; return p; };
    const u32* h = take(1); if (!h) return false;
    u32 nc = h[0];
    for (u32 c = 0; c < nc; c++) {
        h = take(8); if (!h) return false;
        GComponent g; g.log_size = (int)h[0]; g.n_regs = h[2]; g.n_constraints = h[4]; g.log_cd = (int)h[7];
        u32 n_instr = h[1], n_ec = h[3], n_cols = h[5], n_mask = h[6];
        const u32* p = take(4 * (size_t)n_instr); if (!p) return false;
        for (u32 k = 0; k < n_instr; k++) g.prog.push_back({p[4 * k], p[4 * k + 1], p[4 * k + 2], p[4 * k + 3]});
        p = take(4 * (size_t)n_ec); if (!p) return false; g.econsts.assign(p, p + 4 * (size_t)n_ec);
        const u32 *ct = take(n_cols), *ci = take(n_cols), *mc = take(n_cols), *mo = take(n_mask);
        if (!ct || !ci || !mc || !mo) return false;
        size_t m = 0;
        for (u32 k = 0; k < n_cols; k++) {
            g.cols.push_back({(int)ct[k], (int)ci[k]});
            std::vector<int> offs; for (u32 j = 0; j < mc[k]; j++) { if (m >= n_mask) return false; offs.push_back((int)(int32_t)mo[m++]); }
            g.masks.push_back(offs);
        }
        if (m != n_mask) return false;
        air.comps.push_back(std::move(g));
    }
    return i == n;
}

// "" when the components are consistent with trees of the given column log sizes (any number of trace trees >= 1; the reference has 3).
static inline std::string gair_check(const GAir& air, const std::vector<std::vector<int>>& tree_logs) {
    if (tree_logs.empty()) return "at least one trace tree must be committed";
    const int NT = (int)tree_logs.size();
    if (air.comps.empty()) return "no components";
    std::vector<std::vector<char>> claimed(NT);
    for (int t = 0; t < NT; t++) claimed[t].assign(tree_logs[t].size(), 0);
    for (auto& c : air.comps) {
        if (c.cols.size() != c.masks.size()) return "cols / masks length mismatch";
        if (c.log_cd < 0 || c.log_cd > 2) return "component log constraint-degree bound outside {0 (default), 1, 2}";
        for (size_t k = 0; k < c.cols.size(); k++) {
            int t = c.cols[k].first, i = c.cols[k].second;
            if (t < 0 || t >= NT || i < 0 || (size_t)i >= tree_logs[t].size()) return "component column outside the committed trees";
            if (tree_logs[t][i] != c.log_size) return "component column of a different log size than the component";
            claimed[t][i] = 1;
        }
        size_t n_c = 0;
        for (auto& in : c.prog) {
            if (in.op == C_CONSTRAINT_B || in.op == C_CONSTRAINT_E) n_c++;
            if (in.op == C_LOAD || in.op == C_LOADE) {
                u32 w = in.op == C_LOADE ? 4 : 1;
                for (u32 j = 0; j < w; j++) {
                    if (in.a + j >= c.cols.size()) return "LOAD of a column the component does not claim";
                    const auto& m = c.masks[in.a + j];
                    if (std::find(m.begin(), m.end(), (int)in.b) == m.end()) return "LOAD at an offset missing from the column's mask";
                }
            }
        }
        if (n_c != c.n_constraints) return "constraint count mismatch";
    }
    for (int t = 0; t < NT; t++) for (char x : claimed[t]) if (!x) return "a committed column is claimed by no component";
    return "";
}

static inline QPt g_offset_point(QPt oods, int log_size, int offset) {
    if (offset == 0) return oods;
    int64_t idx = ((int64_t)offset * (int64_t)subgroup_gen(log_size)) & 0x7fffffffLL;
    return qpt_add(oods, qpt_from_pt(pt_from_index((u32)idx)));
}

// per tree, per column: the union (first-appearance order) of the offsets the components sample it at
static inline std::vector<std::vector<std::vector<int>>> g_mask_offsets(const GAir& air, const std::vector<size_t>& n_cols) {
    std::vector<std::vector<std::vector<int>>> r(n_cols.size());
    for (size_t t = 0; t < n_cols.size(); t++) r[t].resize(n_cols[t]);
    for (auto& c : air.comps)
        for (size_t k = 0; k < c.cols.size(); k++) {
            auto& dst = r[c.cols[k].first][c.cols[k].second];
            for (int o : c.masks[k]) if (std::find(dst.begin(), dst.end(), o) == dst.end()) dst.push_back(o);
        }
    return r;
}

static inline AirHooks g_hooks(const GAir& air, const PcsConfig& cfg, const std::vector<std::vector<int>>& tree_logs) {
    AirHooks h;
    h.composition_log = 0;
    for (auto& c : air.comps) h.composition_log = std::max(h.composition_log, c.log_size + (c.log_cd > 0 ? c.log_cd : cfg.log_constraint_degree));
    std::vector<size_t> n_cols; for (auto& t : tree_logs) n_cols.push_back(t.size());
    auto offs = std::make_shared<std::vector<std::vector<std::vector<int>>>>(g_mask_offsets(air, n_cols));
    auto logs = std::make_shared<std::vector<std::vector<int>>>(tree_logs);
    h.mask_points = [offs, logs](QPt oods) {
        MaskPoints r(offs->size());
        for (size_t t = 0; t < offs->size(); t++)
            for (size_t c = 0; c < (*offs)[t].size(); c++) {
                std::vector<QPt> pts; for (int o : (*offs)[t][c]) pts.push_back(g_offset_point(oods, (*logs)[t][c], o));
                r[t].push_back(pts);
            }
        return r;
    };
    h.eval_composition_at_point = [&air, offs](QPt point, const SampledValues& sv, QM31 rc) {
        QM31 acc = qm31_zero();
        for (auto& c : air.comps) {
            QM31 denom_inv = qm31_inv(canonic_coset_vanishing_qm31(c.log_size, point));
            std::vector<QM31> R(c.n_regs, qm31_zero());
            auto sampled = [&](u32 col, int off) {
                int t = c.cols[col].first, i = c.cols[col].second;
                const auto& o = (*offs)[t][i];
                size_t k = std::find(o.begin(), o.end(), off) - o.begin();
                return sv[t][i][k];
            };
            for (auto& in : c.prog) {
                switch (in.op) {
                case C_LOAD: R[in.dst] = sampled(in.a, (int)in.b); break;
                case C_CONST: R[in.dst] = qm31_from_m31(in.a); break;
                case C_ADD: case C_ADDE: case C_ADDEB: R[in.dst] = qm31_add(R[in.a], R[in.b]); break;
                case C_SUB: case C_SUBE: R[in.dst] = qm31_sub(R[in.a], R[in.b]); break;
                case C_MUL: case C_MULE: case C_MULEB: R[in.dst] = qm31_mul(R[in.a], R[in.b]); break;
                case C_NEG: R[in.dst] = qm31_sub(qm31_zero(), R[in.a]); break;
                case C_CONSTE: R[in.dst] = qm31_load(&c.econsts[4 * in.a]); break;
                case C_LOADE: { QM31 e[4]; for (int j = 0; j < 4; j++) e[j] = sampled(in.a + j, (int)in.b); R[in.dst] = from_partial_evals(e); break; }
                case C_CONSTRAINT_B: case C_CONSTRAINT_E: acc = qm31_add(qm31_mul(acc, rc), qm31_mul(denom_inv, R[in.a])); break;
                default: break;
                }
            }
        }
        return acc;
    };
    h.compute_composition = [&air, cfg](const Twiddles& tw, const std::vector<TreeData>& trees, QM31 rc, int n_threads) {
        size_t total = 0; for (auto& c : air.comps) total += c.n_constraints;
        std::vector<QM31> powers(total); { QM31 a = qm31_one(); for (size_t i = 0; i < total; i++) { powers[i] = a; a = qm31_mul(a, rc); } }
        std::map<int, SecureCols> sub;
        size_t remaining = total;
        for (auto& c : air.comps) {
            const int e = c.log_size + (c.log_cd > 0 ? c.log_cd : cfg.log_constraint_degree);
            const size_t nc = c.n_constraints;
            std::vector<u32> pw(4 * nc);          // the LAST nc remaining powers, reversed (accumulator.columns())
            for (size_t j = 0; j < nc; j++) qm31_store(&pw[4 * j], powers[remaining - 1 - j]);
            remaining -= nc;
            const bool extend = e != c.log_size + (int)cfg.log_blowup;
            std::vector<std::vector<u32>> ext(extend ? c.cols.size() : 0);
            std::vector<const u32*> ptrs(c.cols.size());
            parallel_for(c.cols.size(), n_threads, [&](size_t k) {
                const TreeData& t = trees[c.cols[k].first];
                if (!extend) { ptrs[k] = t.evals[c.cols[k].second].data(); return; }
                ext[k].resize((size_t)1 << e);
                evaluate(t.polys[c.cols[k].second].data(), c.log_size, ext[k].data(), e, tw);
                ptrs[k] = ext[k].data();
            });
            const int log_expand = e - c.log_size;
            std::vector<u32> denom_inv((size_t)1 << log_expand);
            for (u32 i = 0; i < denom_inv.size(); i++) denom_inv[i] = m31_inv(canonic_coset_vanishing_m31(c.log_size, circle_domain_at(e, i)));
            bit_reverse_inplace(denom_inv.data(), log_expand);
            if (!sub.count(e)) sub[e].init(e);
            u32* acc4[4] = {sub[e].c[0].data(), sub[e].c[1].data(), sub[e].c[2].data(), sub[e].c[3].data()};
            eval_constraint_program(c.prog.data(), (u32)c.prog.size(), c.n_regs, ptrs.data(), c.econsts.data(), pw.data(), denom_inv.data(), c.log_size, e, acc4);
        }
        std::vector<std::vector<u32>> cur; int cur_log = -1;     // DomainEvaluationAccumulator::finalize
        for (auto& kv : sub) {
            int log = kv.first; SecureCols& values = kv.second;
            if (cur_log >= 0)
                for (int k = 0; k < 4; k++) {
                    std::vector<u32> ev((size_t)1 << log);
                    evaluate(cur[k].data(), cur_log, ev.data(), log, tw);
                    for (size_t i = 0; i < ev.size(); i++) values.c[k][i] = m31_add(values.c[k][i], ev[i]);
                }
            cur.assign(4, std::vector<u32>());
            for (int k = 0; k < 4; k++) { cur[k] = values.c[k]; interpolate(cur[k].data(), log, tw); }
            cur_log = log;
        }
        return cur;
    };
    return h;
}

struct ProverSession {
    PcsConfig cfg; Twiddles tw; Channel ch; CommitmentSchemeProver cs;
    ProverSession(const PcsConfig& c, int max_log, int n_threads) : cfg(c), tw(precompute_twiddles(max_log + c.log_constraint_degree + (int)c.log_blowup - 1)) {
        cs.cfg = c; cs.tw = &tw; cs.n_threads = n_threads;
    }
    Proof prove(const GAir& air) {
        std::vector<std::vector<int>> tree_logs;
        for (auto& t : cs.trees) tree_logs.push_back(t.logs);
        std::string e = gair_check(air, tree_logs);
        if (!e.empty()) throw e;
        return prove_core(cs, ch, cfg, tw, g_hooks(air, cfg, tree_logs), cs.n_threads);
    }
};

struct VerifierSession {
    PcsConfig cfg; Channel ch;
    std::vector<std::vector<int>> tree_logs; std::vector<Hash> roots;
    void commit(const Hash& root, const std::vector<int>& logs) { ch.mix_root(root); roots.push_back(root); tree_logs.push_back(logs); }   // CommitmentSchemeVerifier::commit
    std::string verify(const GAir& air, const Proof& proof) {
        std::string e = gair_check(air, tree_logs);
        if (!e.empty()) return "InvalidStructure: " + e;
        if (proof.commitments.size() != tree_logs.size() + 1) return "InvalidStructure";
        for (size_t t = 0; t < tree_logs.size(); t++) if (memcmp(proof.commitments[t].w, roots[t].w, 32)) return "CommitmentMismatch";
        return verify_core(ch, cfg, proof, tree_logs, g_hooks(air, cfg, tree_logs));
    }
};

}  // namespace orc
