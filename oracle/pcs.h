// ORACLE — TEST INFRASTRUCTURE ONLY (see fields.h header).  PARITY UNPINNED.
//
// Polynomial commitment scheme, DEEP quotients, FRI and the prove / verify drivers.  Restates Stwo
//   prover/pcs/mod.rs            (CommitmentSchemeProver, TreeBuilder, prove_values)
//   core/pcs/quotients.rs        (ColumnSampleBatch, quotient_constants, denominator_inverses,
//   prover/backend/cpu/quotients.rs  accumulate_row_quotients, accumulate_quotients, fri_answers)
//   prover/fri.rs, prover/backend/cpu/fri.rs, core/fri.rs   (FriProver / FriVerifier, folds)
//   core/queries.rs              (Queries::generate / fold)
//   prover/mod.rs::prove, core/verifier.rs::verify, core/pcs/verifier.rs
//   prover/air/accumulation.rs   (DomainEvaluationAccumulator), core/air/accumulation.rs
// and the orchestration order of reference prover/src/machine.rs:184-296 (prove) / :299-485 (verify).
//
// UNVERIFIED UPSTREAM RULES kept switchable: hash_mode (merkle.h),
// fri_alpha_mode (FRI_ALPHA_PREV: circle columns folded with the previous layer's alpha, newer
// Stwo; FRI_ALPHA_FIRST: all circle columns folded with the first drawn alpha, older Stwo),
// pow_bits (PcsConfig::default is 5 or 10 depending on the revision).
#pragma once
#include <vector>
#include <map>
#include <set>
#include <string>
#include <thread>
#include <functional>
#include "fields.h"
#include "poly.h"
#include "merkle.h"
#include "air.h"

namespace orc {

enum FriAlphaMode { FRI_ALPHA_PREV = 0, FRI_ALPHA_FIRST = 1 };

struct PcsConfig {
    u32 pow_bits, log_blowup, n_queries, log_last_layer_degree_bound;
    int hash_mode, fri_alpha_mode;
    int log_constraint_degree;  // AIR parameter (1 or 2); kept here for convenience
};
static inline PcsConfig pcs_default() { PcsConfig c = {10, 1, 3, 0, HASH_STD, FRI_ALPHA_PREV, 1}; return c; }

static inline void parallel_for(size_t n, int n_threads, const std::function<void(size_t)>& f) {
    if (n_threads <= 1 || n <= 1) { for (size_t i = 0; i < n; i++) f(i); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++)
        th.emplace_back([&, t]() { for (size_t i = t; i < n; i += n_threads) f(i); });
    for (auto& x : th) x.join();
}

// ---------------- secure (QM31) columns by coordinates ----------------
struct SecureCols {
    int log;
    std::vector<u32> c[4];
    void init(int l) { log = l; for (int k = 0; k < 4; k++) c[k].assign((size_t)1 << l, 0); }
    QM31 at(size_t i) const { return qm31(c[0][i], c[1][i], c[2][i], c[3][i]); }
    void set(size_t i, QM31 v) { c[0][i] = v.a.a; c[1][i] = v.a.b; c[2][i] = v.b.a; c[3][i] = v.b.b; }
    size_t len() const { return c[0].size(); }
};

// ---------------- DEEP quotients ----------------
struct PointSample { QPt point; QM31 value; };
struct ColumnSampleBatch { QPt point; std::vector<std::pair<size_t, QM31>> cols; };

static inline bool qpt_eq(const QPt& a, const QPt& b) { return qm31_eq(a.x, b.x) && qm31_eq(a.y, b.y); }

// ColumnSampleBatch::new_vec — group by point, insertion-ordered.
static inline std::vector<ColumnSampleBatch> sample_batches_new(const std::vector<const std::vector<PointSample>*>& samples) {
    std::vector<ColumnSampleBatch> out;
    for (size_t ci = 0; ci < samples.size(); ci++)
        for (const PointSample& s : *samples[ci]) {
            size_t b = 0;
            for (; b < out.size(); b++) if (qpt_eq(out[b].point, s.point)) break;
            if (b == out.size()) { ColumnSampleBatch nb; nb.point = s.point; out.push_back(nb); }
            out[b].cols.push_back({ci, s.value});
        }
    return out;
}

struct LineCoeffs { QM31 a, b, c; };
struct QuotientConstants { std::vector<std::vector<LineCoeffs>> line_coeffs; std::vector<QM31> batch_random_coeffs; };

static inline QuotientConstants quotient_constants(const std::vector<ColumnSampleBatch>& batches, QM31 random_coeff) {
    QuotientConstants q;
    for (const auto& sb : batches) {
        std::vector<LineCoeffs> lc;
        QM31 alpha = qm31_one();
        for (const auto& cv : sb.cols) {
            alpha = qm31_mul(alpha, random_coeff);
            // complex_conjugate_line_coeffs(sample, alpha)
            QM31 a = qm31_sub(qm31_conj(cv.second), cv.second);
            QM31 c = qm31_sub(qm31_conj(sb.point.y), sb.point.y);
            QM31 b = qm31_sub(qm31_mul(cv.second, c), qm31_mul(a, sb.point.y));
            lc.push_back({qm31_mul(alpha, a), qm31_mul(alpha, b), qm31_mul(alpha, c)});
        }
        q.line_coeffs.push_back(lc);
        q.batch_random_coeffs.push_back(qm31_pow(random_coeff, sb.cols.size()));
    }
    return q;
}

static inline QM31 accumulate_row_quotients(const std::vector<ColumnSampleBatch>& batches, const u32* row_values,
                                            const QuotientConstants& qc, Pt dp) {
    QM31 acc = qm31_zero();
    for (size_t b = 0; b < batches.size(); b++) {
        const auto& sb = batches[b];
        // denominator: (Re(p.x) - d.x) * Im(p.y) - (Re(p.y) - d.y) * Im(p.x)   in CM31
        CM31 prx = sb.point.x.a, pry = sb.point.y.a, pix = sb.point.x.b, piy = sb.point.y.b;
        CM31 den = cm31_sub(cm31_mul(cm31_sub(prx, cm31(dp.x, 0)), piy), cm31_mul(cm31_sub(pry, cm31(dp.y, 0)), pix));
        CM31 den_inv = cm31_inv(den);
        QM31 num = qm31_zero();
        for (size_t k = 0; k < sb.cols.size(); k++) {
            const LineCoeffs& l = qc.line_coeffs[b][k];
            QM31 value = qm31_mul_m31(l.c, row_values[sb.cols[k].first]);
            QM31 linear = qm31_add(qm31_mul_m31(l.a, dp.y), l.b);
            num = qm31_add(num, qm31_sub(value, linear));
        }
        acc = qm31_add(qm31_mul(acc, qc.batch_random_coeffs[b]), qm31_mul_cm31(num, den_inv));
    }
    return acc;
}

// accumulate_quotients over the whole (bit-reversed) LDE domain of log size `log`.
static inline SecureCols accumulate_quotients(int log, const std::vector<const u32*>& cols, QM31 random_coeff,
                                              const std::vector<ColumnSampleBatch>& batches, int n_threads = 1) {
    SecureCols out; out.init(log);
    QuotientConstants qc = quotient_constants(batches, random_coeff);
    size_t N = (size_t)1 << log, chunk = 1024;
    parallel_for((N + chunk - 1) / chunk, n_threads, [&](size_t ch) {
        std::vector<u32> row(cols.size());
        for (size_t r = ch * chunk; r < std::min(N, (ch + 1) * chunk); r++) {
            Pt dp = circle_domain_at(log, bit_reverse_index((u32)r, log));
            for (size_t c = 0; c < cols.size(); c++) row[c] = cols[c][r];
            out.set(r, accumulate_row_quotients(batches, row.data(), qc, dp));
        }
    });
    return out;
}

// ---------------- FRI folds (prover/backend/cpu/fri.rs; core/fri.rs) ----------------
// Line domain of log size L used by FRI: LineDomain(Coset::half_odds(L)) and its doublings.
static inline QM31 ibutterfly_q(QM31& v0, QM31& v1, u32 itw) {
    QM31 t = v0; v0 = qm31_add(t, v1); v1 = qm31_mul_m31(qm31_sub(t, v1), itw); return v0;
}
// fold_line: eval on line domain `dom` (log size L) -> eval on dom.double() (log L-1)
static inline std::vector<QM31> fold_line(const std::vector<QM31>& eval, Coset dom, QM31 alpha) {
    int L = dom.log;
    std::vector<QM31> out(eval.size() / 2);
    for (size_t i = 0; i < out.size(); i++) {
        u32 x = coset_at(dom, bit_reverse_index((u32)(i << 1), L)).x;
        QM31 f0 = eval[2 * i], f1 = eval[2 * i + 1];
        ibutterfly_q(f0, f1, m31_inv(x));
        out[i] = qm31_add(f0, qm31_mul(alpha, f1));
    }
    return out;
}
// fold_circle_into_line: dst (line, log L-1) = dst*alpha^2 + fold(src on CanonicCoset(L).circle_domain())
static inline void fold_circle_into_line(std::vector<QM31>& dst, const SecureCols& src, QM31 alpha) {
    int L = src.log;
    QM31 alpha_sq = qm31_sqr(alpha);
    for (size_t i = 0; i < dst.size(); i++) {
        Pt p = circle_domain_at(L, bit_reverse_index((u32)(i << 1), L));
        QM31 f0 = src.at(2 * i), f1 = src.at(2 * i + 1);
        ibutterfly_q(f0, f1, m31_inv(p.y));
        QM31 fp = qm31_add(qm31_mul(alpha, f1), f0);
        dst[i] = qm31_add(qm31_mul(dst[i], alpha_sq), fp);
    }
}

// LineEvaluation::interpolate + into_ordered_coefficients (core/poly/line.rs), for the last layer.
static inline std::vector<QM31> line_interpolate_ordered(std::vector<QM31> v, Coset dom) {
    int L = dom.log;
    size_t n = v.size();
    // bit_reverse
    for (size_t i = 0; i < n; i++) { size_t j = bit_reverse_index((u32)i, L); if (i < j) std::swap(v[i], v[j]); }
    Coset d = dom;
    while (((size_t)1 << d.log) > 1) {
        size_t ds = (size_t)1 << d.log;
        for (size_t c0 = 0; c0 < n; c0 += ds)
            for (size_t i = 0; i < ds / 2; i++) {
                u32 x = coset_at(d, (u32)i).x;
                ibutterfly_q(v[c0 + i], v[c0 + ds / 2 + i], m31_inv(x));
            }
        d = coset_double(d);
    }
    u32 len_inv = m31_inv((u32)n);
    for (auto& x : v) x = qm31_mul_m31(x, len_inv);
    // LinePoly::new(coeffs) stores bit-reversed coefficients; into_ordered_coefficients bit-reverses.
    for (size_t i = 0; i < n; i++) { size_t j = bit_reverse_index((u32)i, L); if (i < j) std::swap(v[i], v[j]); }
    return v;
}
// LinePoly::eval_at_point for ordered coefficients.
static inline QM31 line_poly_eval_ordered(const std::vector<QM31>& ordered, QM31 x) {
    int L = 0; while (((size_t)1 << L) < ordered.size()) L++;
    std::vector<QM31> coeffs(ordered);
    for (size_t i = 0; i < coeffs.size(); i++) { size_t j = bit_reverse_index((u32)i, L); if (i < j) std::swap(coeffs[i], coeffs[j]); }
    std::vector<QM31> doublings;
    for (int i = 0; i < L; i++) { doublings.push_back(x); x = double_x_qm31(x); }
    return fold_qm31(coeffs.data(), coeffs.size(), doublings.data());
}

// ---------------- Queries ----------------
static inline std::vector<size_t> queries_generate(Channel& ch, int log_domain, u32 n_queries) {
    std::set<size_t> q; u32 cnt = 0;
    size_t mask = ((size_t)1 << log_domain) - 1;
    for (;;) {
        u32 w[8]; ch.draw_u32s(w);
        for (int i = 0; i < 8; i++) {
            q.insert((size_t)w[i] & mask);
            if (++cnt == n_queries) return std::vector<size_t>(q.begin(), q.end());
        }
    }
}
static inline std::vector<size_t> queries_fold(const std::vector<size_t>& q, int n_folds) {
    std::vector<size_t> r;
    for (size_t p : q) { size_t f = p >> n_folds; if (r.empty() || r.back() != f) r.push_back(f); }
    return r;
}

// ---------------- proof ----------------
struct FriLayerProof { std::vector<QM31> fri_witness; MerkleDecommitment decommitment; Hash commitment; };
struct Proof {
    PcsConfig config;
    std::vector<Hash> commitments;
    std::vector<std::vector<std::vector<QM31>>> sampled_values;  // tree -> column -> samples
    std::vector<MerkleDecommitment> decommitments;
    std::vector<std::vector<u32>> queried_values;
    u64 proof_of_work;
    FriLayerProof first_layer;
    std::vector<FriLayerProof> inner_layers;
    std::vector<QM31> last_layer_poly;  // ordered coefficients
};

static const u32 PROOF_MAGIC = 0x3150584Eu;  // "NXP1"

static inline void ser_hash(std::vector<u32>& o, const Hash& h) { for (int i = 0; i < 8; i++) o.push_back(h.w[i]); }
static inline void ser_q(std::vector<u32>& o, QM31 q) { u32 w[4]; qm31_store(w, q); o.insert(o.end(), w, w + 4); }
static inline void ser_decommit(std::vector<u32>& o, const MerkleDecommitment& d) {
    o.push_back((u32)d.hash_witness.size()); for (auto& h : d.hash_witness) ser_hash(o, h);
    o.push_back((u32)d.column_witness.size()); o.insert(o.end(), d.column_witness.begin(), d.column_witness.end());
}
static inline void ser_fri_layer(std::vector<u32>& o, const FriLayerProof& l) {
    o.push_back((u32)l.fri_witness.size()); for (auto& q : l.fri_witness) ser_q(o, q);
    ser_decommit(o, l.decommitment); ser_hash(o, l.commitment);
}
// Flat little-endian u32 wire format shared with the product (include/nexus_hip.h "NXP1").
static inline std::vector<u32> proof_serialize(const Proof& p) {
    std::vector<u32> o;
    o.push_back(PROOF_MAGIC);
    o.push_back(p.config.pow_bits); o.push_back(p.config.log_blowup); o.push_back(p.config.n_queries);
    o.push_back(p.config.log_last_layer_degree_bound);
    o.push_back((u32)p.commitments.size());
    for (auto& h : p.commitments) ser_hash(o, h);
    for (auto& t : p.sampled_values) {
        o.push_back((u32)t.size());
        for (auto& c : t) { o.push_back((u32)c.size()); for (auto& q : c) ser_q(o, q); }
    }
    for (auto& d : p.decommitments) ser_decommit(o, d);
    for (auto& v : p.queried_values) { o.push_back((u32)v.size()); o.insert(o.end(), v.begin(), v.end()); }
    o.push_back((u32)p.proof_of_work); o.push_back((u32)(p.proof_of_work >> 32));
    ser_fri_layer(o, p.first_layer);
    o.push_back((u32)p.inner_layers.size());
    for (auto& l : p.inner_layers) ser_fri_layer(o, l);
    o.push_back((u32)p.last_layer_poly.size());
    for (auto& q : p.last_layer_poly) ser_q(o, q);
    return o;
}

struct Reader {
    const u32* p; size_t n, i; bool ok;
    Reader(const u32* p_, size_t n_) : p(p_), n(n_), i(0), ok(true) {}
    u32 u() { if (i >= n) { ok = false; return 0; } return p[i++]; }
    size_t count(size_t unit) { u32 c = u(); if ((size_t)c * unit > n - i) { ok = false; return 0; } return c; }
    Hash h() { Hash x; for (int k = 0; k < 8; k++) x.w[k] = u(); return x; }
    QM31 q() { u32 a = u(), b = u(), c = u(), d = u(); return qm31(a, b, c, d); }
};
static inline void de_decommit(Reader& r, MerkleDecommitment& d) {
    size_t n = r.count(8); for (size_t i = 0; i < n; i++) d.hash_witness.push_back(r.h());
    n = r.count(1); for (size_t i = 0; i < n; i++) d.column_witness.push_back(r.u());
}
static inline void de_fri_layer(Reader& r, FriLayerProof& l) {
    size_t n = r.count(4); for (size_t i = 0; i < n; i++) l.fri_witness.push_back(r.q());
    de_decommit(r, l.decommitment); l.commitment = r.h();
}
static inline bool proof_deserialize(const u32* w, size_t n, Proof& p) {
    Reader r(w, n);
    if (r.u() != PROOF_MAGIC) return false;
    p.config = pcs_default();
    p.config.pow_bits = r.u(); p.config.log_blowup = r.u(); p.config.n_queries = r.u();
    p.config.log_last_layer_degree_bound = r.u();
    size_t nt = r.count(8);
    for (size_t t = 0; t < nt; t++) p.commitments.push_back(r.h());
    p.sampled_values.resize(nt);
    for (size_t t = 0; t < nt && r.ok; t++) {
        size_t nc = r.count(1); p.sampled_values[t].resize(nc);
        for (size_t c = 0; c < nc && r.ok; c++) { size_t ns = r.count(4); for (size_t s = 0; s < ns; s++) p.sampled_values[t][c].push_back(r.q()); }
    }
    p.decommitments.resize(nt);
    for (size_t t = 0; t < nt && r.ok; t++) de_decommit(r, p.decommitments[t]);
    p.queried_values.resize(nt);
    for (size_t t = 0; t < nt && r.ok; t++) { size_t nv = r.count(1); for (size_t i = 0; i < nv; i++) p.queried_values[t].push_back(r.u()); }
    u32 lo = r.u(), hi = r.u(); p.proof_of_work = (u64)lo | ((u64)hi << 32);
    de_fri_layer(r, p.first_layer);
    size_t nl = r.count(1);
    p.inner_layers.resize(nl);
    for (size_t l = 0; l < nl && r.ok; l++) de_fri_layer(r, p.inner_layers[l]);
    size_t nc = r.count(4);
    for (size_t i = 0; i < nc; i++) p.last_layer_poly.push_back(r.q());
    return r.ok && r.i == n;
}

// ---------------- commitment scheme (prover side) ----------------
struct TreeData {
    std::vector<int> logs;                 // polynomial log sizes, commit order
    std::vector<std::vector<u32>> polys;   // coefficients
    std::vector<std::vector<u32>> evals;   // LDE, bit-reversed, log = logs[i] + log_blowup
    MerkleTree merkle;
};

struct CommitmentSchemeProver {
    PcsConfig cfg;
    const Twiddles* tw;
    int n_threads;
    std::vector<TreeData> trees;
    // TreeBuilder::extend_evals + commit: columns are bit-reversed evaluations on
    // CanonicCoset(log).circle_domain().
    void commit_evals(std::vector<std::vector<u32>> cols, const std::vector<int>& logs, Channel& ch) {
        parallel_for(cols.size(), n_threads, [&](size_t i) { interpolate(cols[i].data(), logs[i], *tw); });
        commit_polys(std::move(cols), logs, ch);
    }
    // TreeBuilder::extend_polys + commit -> CommitmentTreeProver::new
    void commit_polys(std::vector<std::vector<u32>> polys, const std::vector<int>& logs, Channel& ch) {
        TreeData t; t.logs = logs; t.polys = std::move(polys);
        t.evals.resize(t.polys.size());
        parallel_for(t.polys.size(), n_threads, [&](size_t i) {
            int el = logs[i] + (int)cfg.log_blowup;
            t.evals[i].resize((size_t)1 << el);
            evaluate(t.polys[i].data(), logs[i], t.evals[i].data(), el, *tw);
        });
        std::vector<ColRef> refs;
        for (size_t i = 0; i < t.evals.size(); i++) refs.push_back({t.evals[i].data(), logs[i] + (int)cfg.log_blowup});
        merkle_n_threads() = n_threads;
        t.merkle = merkle_commit(refs, cfg.hash_mode);
        ch.mix_root(t.merkle.root());
        trees.push_back(std::move(t));
    }
};

// ---------------- FRI prover ----------------
struct FriLayer { SecureCols evals; MerkleTree merkle; Coset dom; };  // inner layer (line evaluation)
struct FriProverState {
    std::vector<SecureCols> columns;  // first layer (circle evaluations), decreasing size
    MerkleTree first_merkle;
    std::vector<FriLayer> inner;
    std::vector<QM31> last_layer_poly;
};

static inline std::vector<ColRef> coord_cols(const std::vector<SecureCols>& cols) {
    std::vector<ColRef> r;
    for (auto& s : cols) for (int k = 0; k < 4; k++) r.push_back({s.c[k].data(), s.log});
    return r;
}
static inline SecureCols to_secure_cols(const std::vector<QM31>& v, int log) {
    SecureCols s; s.init(log); for (size_t i = 0; i < v.size(); i++) s.set(i, v[i]); return s;
}

static inline FriProverState fri_commit(Channel& ch, const PcsConfig& cfg, std::vector<SecureCols> columns) {
    FriProverState st;
    st.columns = std::move(columns);
    // commit_first_layer
    st.first_merkle = merkle_commit(coord_cols(st.columns), cfg.hash_mode);
    ch.mix_root(st.first_merkle.root());
    // commit_inner_layers
    QM31 folding_alpha = ch.draw_secure_felt();
    const QM31 first_alpha = folding_alpha;
    int first_log = st.columns[0].log - 1;
    Coset dom = coset_half_odds(first_log);
    std::vector<QM31> layer((size_t)1 << first_log, qm31_zero());
    size_t ci = 0;
    size_t last_size = (size_t)1 << (cfg.log_last_layer_degree_bound + cfg.log_blowup);
    while (layer.size() > last_size) {
        while (ci < st.columns.size() && (st.columns[ci].len() >> 1) == layer.size()) {
            fold_circle_into_line(layer, st.columns[ci], cfg.fri_alpha_mode == FRI_ALPHA_PREV ? folding_alpha : first_alpha);
            ci++;
        }
        FriLayer L; L.dom = dom; L.evals = to_secure_cols(layer, dom.log);
        std::vector<ColRef> refs; for (int k = 0; k < 4; k++) refs.push_back({L.evals.c[k].data(), dom.log});
        L.merkle = merkle_commit(refs, cfg.hash_mode);
        ch.mix_root(L.merkle.root());
        folding_alpha = ch.draw_secure_felt();
        layer = fold_line(layer, dom, folding_alpha);
        dom = coset_double(dom);
        st.inner.push_back(std::move(L));
    }
    if (ci != st.columns.size()) throw std::string("fri: columns not consumed");
    // commit_last_layer
    std::vector<QM31> coeffs = line_interpolate_ordered(layer, dom);
    size_t bound = (size_t)1 << cfg.log_last_layer_degree_bound;
    for (size_t i = bound; i < coeffs.size(); i++) if (!qm31_is_zero(coeffs[i])) throw std::string("fri: invalid degree");
    coeffs.resize(bound);
    // channel.mix_felts(&last_layer_poly): LinePoly stores its coefficients bit-reversed [upstream-recollection] — that is what is mixed
    { std::vector<QM31> m(bound); for (size_t i = 0; i < bound; i++) m[i] = coeffs[bit_reverse_index((u32)i, (int)cfg.log_last_layer_degree_bound)]; ch.mix_felts(m.data(), m.size()); }
    st.last_layer_poly = coeffs;
    return st;
}

// compute_decommitment_positions_and_witness_evals
static inline void decommit_positions_and_witness(const SecureCols& col, const std::vector<size_t>& queries, int fold_step,
                                                  std::vector<size_t>& positions, std::vector<QM31>& witness) {
    size_t i = 0;
    while (i < queries.size()) {
        size_t j = i;
        while (j < queries.size() && (queries[j] >> fold_step) == (queries[i] >> fold_step)) j++;
        size_t start = (queries[i] >> fold_step) << fold_step;
        size_t qi = i;
        for (size_t pos = start; pos < start + ((size_t)1 << fold_step); pos++) {
            positions.push_back(pos);
            if (qi < j && queries[qi] == pos) { qi++; continue; }
            witness.push_back(col.at(pos));
        }
        i = j;
    }
}

// FriProver::decommit -> (FriProof pieces into `proof`, query positions per (LDE) log size)
static inline std::map<int, std::vector<size_t>> fri_decommit(Channel& ch, const PcsConfig& cfg, const FriProverState& st, Proof& proof) {
    int max_log = st.columns[0].log;
    std::vector<size_t> queries = queries_generate(ch, max_log, cfg.n_queries);
    std::map<int, std::vector<size_t>> by_log;
    for (auto& c : st.columns) by_log[c.log] = queries_fold(queries, max_log - c.log);
    // first layer
    {
        std::map<int, std::vector<size_t>> pos_by_log;
        for (auto& c : st.columns) {
            std::vector<size_t> cq = queries_fold(queries, max_log - c.log), pos;
            decommit_positions_and_witness(c, cq, 1, pos, proof.first_layer.fri_witness);
            pos_by_log[c.log] = pos;
        }
        std::vector<u32> unused;
        merkle_decommit(st.first_merkle, pos_by_log, coord_cols(st.columns), unused, proof.first_layer.decommitment);
        proof.first_layer.commitment = st.first_merkle.root();
    }
    // inner layers
    std::vector<size_t> lq = queries_fold(queries, 1);
    for (auto& L : st.inner) {
        FriLayerProof lp;
        std::vector<size_t> pos;
        decommit_positions_and_witness(L.evals, lq, 1, pos, lp.fri_witness);
        std::map<int, std::vector<size_t>> pos_by_log; pos_by_log[L.evals.log] = pos;
        std::vector<ColRef> refs; for (int k = 0; k < 4; k++) refs.push_back({L.evals.c[k].data(), L.evals.log});
        std::vector<u32> unused;
        merkle_decommit(L.merkle, pos_by_log, refs, unused, lp.decommitment);
        lp.commitment = L.merkle.root();
        proof.inner_layers.push_back(lp);
        lq = queries_fold(lq, 1);
    }
    proof.last_layer_poly = st.last_layer_poly;
    return by_log;
}

// ---------------- AIR glue ----------------
struct AirSpec { std::vector<ComponentSpec> comps; };

struct TraceLocation { size_t pre0, main0, inter0; };  // first column of the component in each tree
static inline std::vector<TraceLocation> trace_locations(const AirSpec& air) {
    std::vector<TraceLocation> r; size_t a = 0, b = 0, c = 0;
    for (auto& s : air.comps) { r.push_back({a, b, c}); a += s.n_pre; b += s.n_main; c += s.n_inter; }
    return r;
}

struct DomainRow {  // M31 view of one row of the constraint-evaluation domain
    const u32* const* p_; const u32* const* m_; const u32* const* i_; size_t r, rn;
    FM one() const { return FM{1}; }
    FM pre(int k) const { return FM{p_[k][r]}; }
    FM main(int k) const { return FM{m_[k][r]}; }
    FM main_next(int k) const { return FM{m_[k][rn]}; }
    FM inter(int k) const { return FM{i_[k][r]}; }
};
struct PointRow {  // QM31 view of the sampled mask values
    const std::vector<std::vector<QM31>>* p_; const std::vector<std::vector<QM31>>* m_; const std::vector<std::vector<QM31>>* i_;
    size_t p0, m0, i0;
    FQ one() const { return FQ{qm31_one()}; }
    FQ pre(int k) const { return FQ{(*p_)[p0 + k][0]}; }
    FQ main(int k) const { return FQ{(*m_)[m0 + k][0]}; }
    FQ main_next(int k) const { return FQ{(*m_)[m0 + k][1]}; }
    FQ inter(int k) const { return FQ{(*i_)[i0 + k][0]}; }
};

// components().eval_composition_polynomial_at_point
static inline QM31 eval_composition_at_point(const AirSpec& air, QPt point, const std::vector<std::vector<std::vector<QM31>>>& sampled, QM31 random_coeff) {
    QM31 acc = qm31_zero();
    auto locs = trace_locations(air);
    for (size_t ci = 0; ci < air.comps.size(); ci++) {
        const ComponentSpec& c = air.comps[ci];
        QM31 denom_inv = qm31_inv(canonic_coset_vanishing_qm31(c.log_size, point));
        PointRow row = {&sampled[0], &sampled[1], &sampled[2], locs[ci].pre0, locs[ci].main0, locs[ci].inter0};
        auto f = [&](FQ v) { acc = qm31_add(qm31_mul(acc, random_coeff), qm31_mul(denom_inv, v.v)); };
        eval_constraints(c, row, f);
    }
    return acc;
}

// mask points: tree -> column -> points
static inline std::vector<std::vector<std::vector<QPt>>> mask_points(const AirSpec& air, QPt oods) {
    std::vector<std::vector<std::vector<QPt>>> r(4);
    for (auto& c : air.comps) {
        QPt step = qpt_from_pt(pt_from_index(subgroup_gen(c.log_size)));
        for (int k = 0; k < c.n_pre; k++) r[0].push_back({oods});
        for (int k = 0; k < c.n_main; k++) { if (k < 2) r[1].push_back({oods, qpt_add(oods, step)}); else r[1].push_back({oods}); }
        for (int k = 0; k < c.n_inter; k++) r[2].push_back({oods});
    }
    for (int k = 0; k < 4; k++) r[3].push_back({oods});
    return r;
}

static inline QPt get_random_point(Channel& ch) {
    QM31 t = ch.draw_secure_felt();
    QM31 t2 = qm31_sqr(t);
    QM31 inv = qm31_inv(qm31_add(t2, qm31_one()));
    QPt p = {qm31_mul(qm31_sub(qm31_one(), t2), inv), qm31_mul(qm31_add(t, t), inv)};
    return p;
}

// What stwo::prover::prove / core::verifier::verify need from the components (Components / ComponentProvers): the
// synthetic machine below and the recorded AIRs of air_generic.h both provide it.
struct TreeData;
typedef std::vector<std::vector<std::vector<QM31>>> SampledValues;   // tree -> column -> mask
typedef std::vector<std::vector<std::vector<QPt>>> MaskPoints;
struct AirHooks {
    int composition_log;
    std::function<std::vector<std::vector<u32>>(const Twiddles&, const std::vector<TreeData>&, QM31, int)> compute_composition;
    std::function<MaskPoints(QPt)> mask_points;                                     // trees 0..2 (the composition tree is added by the driver)
    std::function<QM31(QPt, const SampledValues&, QM31)> eval_composition_at_point;
};

static inline int composition_log(const AirSpec& air, const PcsConfig& cfg) {
    int m = 0; for (auto& c : air.comps) m = std::max(m, c.log_size + comp_log_cd(c, cfg)); return m;   // max over components (prover2/machine/src/prove.rs:44-48)
}

// ComponentProvers::compute_composition_polynomial: 4 coordinate polynomials (coefficients) of log
// size composition_log().
static inline std::vector<std::vector<u32>> compute_composition(const AirSpec& air, const PcsConfig& cfg, const Twiddles& tw,
                                                                const std::vector<TreeData>& trees, QM31 random_coeff, int n_threads) {
    size_t total = 0; for (auto& c : air.comps) total += n_constraints(c);
    std::vector<QM31> powers(total); { QM31 a = qm31_one(); for (size_t i = 0; i < total; i++) { powers[i] = a; a = qm31_mul(a, random_coeff); } }
    std::map<int, SecureCols> sub;  // log size -> accumulation
    auto locs = trace_locations(air);
    size_t remaining = total;
    for (size_t ci = 0; ci < air.comps.size(); ci++) {
        const ComponentSpec& c = air.comps[ci];
        int e = c.log_size + comp_log_cd(c, cfg);
        size_t nc = n_constraints(c);
        // accumulator.columns(): this component takes the LAST nc remaining powers, then reverses them
        std::vector<QM31> pw(powers.begin() + (remaining - nc), powers.begin() + remaining);
        remaining -= nc;
        std::reverse(pw.begin(), pw.end());
        // trace on the evaluation domain (re-evaluate from coefficients if it differs from the committed LDE)
        bool extend = e != c.log_size + (int)cfg.log_blowup;
        std::vector<std::vector<u32>> ext;
        std::vector<const u32*> pre(c.n_pre), mainc(c.n_main), inter(c.n_inter);
        auto bind = [&](int tree, size_t first, int n, std::vector<const u32*>& dst) {
            for (int k = 0; k < n; k++) {
                if (!extend) { dst[k] = trees[tree].evals[first + k].data(); continue; }
                ext.emplace_back((size_t)1 << e);
                dst[k] = nullptr;
            }
        };
        bind(0, locs[ci].pre0, c.n_pre, pre); bind(1, locs[ci].main0, c.n_main, mainc); bind(2, locs[ci].inter0, c.n_inter, inter);
        if (extend) {
            std::vector<std::pair<int, size_t>> src;
            for (int k = 0; k < c.n_pre; k++) src.push_back({0, locs[ci].pre0 + k});
            for (int k = 0; k < c.n_main; k++) src.push_back({1, locs[ci].main0 + k});
            for (int k = 0; k < c.n_inter; k++) src.push_back({2, locs[ci].inter0 + k});
            parallel_for(src.size(), n_threads, [&](size_t i) {
                evaluate(trees[src[i].first].polys[src[i].second].data(), c.log_size, ext[i].data(), e, tw);
            });
            size_t i = 0;
            for (int k = 0; k < c.n_pre; k++) pre[k] = ext[i++].data();
            for (int k = 0; k < c.n_main; k++) mainc[k] = ext[i++].data();
            for (int k = 0; k < c.n_inter; k++) inter[k] = ext[i++].data();
        }
        // denominator inverses: 1/coset_vanishing(trace coset, eval_domain.at(i)), i < 2^log_expand, bit-reversed
        int log_expand = e - c.log_size;
        std::vector<u32> denom_inv((size_t)1 << log_expand);
        for (u32 i = 0; i < denom_inv.size(); i++) denom_inv[i] = m31_inv(canonic_coset_vanishing_m31(c.log_size, circle_domain_at(e, i)));
        bit_reverse_inplace(denom_inv.data(), log_expand);
        if (!sub.count(e)) sub[e].init(e);
        SecureCols& col = sub[e];
        size_t E = (size_t)1 << e, chunk = 1024;
        parallel_for((E + chunk - 1) / chunk, n_threads, [&](size_t ch) {
            for (size_t r = ch * chunk; r < std::min(E, (ch + 1) * chunk); r++) {
                DomainRow row = {pre.data(), mainc.data(), inter.data(), r, offset_bit_reversed_circle_domain_index((u32)r, c.log_size, e, 1)};
                QM31 row_res = qm31_zero(); size_t j = 0;
                auto f = [&](FM v) { row_res = qm31_add(row_res, qm31_mul_m31(pw[j++], v.v)); };
                eval_constraints(c, row, f);
                col.set(r, qm31_add(col.at(r), qm31_mul_m31(row_res, denom_inv[r >> c.log_size])));
            }
        });
    }
    // DomainEvaluationAccumulator::finalize
    std::vector<std::vector<u32>> cur;  // 4 coefficient vectors
    int cur_log = -1;
    for (auto& kv : sub) {  // ascending log size
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
}

// from_partial_evals: c0 + c1*i + c2*u + c3*iu
static inline QM31 from_partial_evals(const QM31 e[4]) {
    QM31 r = e[0];
    r = qm31_add(r, qm31_mul(e[1], qm31(0, 1, 0, 0)));
    r = qm31_add(r, qm31_mul(e[2], qm31(0, 0, 1, 0)));
    r = qm31_add(r, qm31_mul(e[3], qm31(0, 0, 0, 1)));
    return r;
}

// Natural-order synthetic trace of one tree, finalized (bit-reversed circle-domain order).
// tree: 0 pre, 1 main, 2 inter.
static inline void synth_tree_columns(const AirSpec& air, int tree, u64 seed, u64 inter_seed, int n_threads,
                                      std::vector<std::vector<u32>>& cols, std::vector<int>& logs) {
    cols.clear(); logs.clear();
    for (size_t ci = 0; ci < air.comps.size(); ci++) {
        const ComponentSpec& c = air.comps[ci];
        int n = tree == 0 ? c.n_pre : tree == 1 ? c.n_main : c.n_inter;
        size_t N = (size_t)1 << c.log_size;
        std::vector<std::vector<u32>> nat(n, std::vector<u32>(N));
        size_t chunk = 4096;
        parallel_for((N + chunk - 1) / chunk, n_threads, [&](size_t ch) {
            std::vector<u32> pre(c.n_pre), mainv(c.n_main), inter(c.n_inter);
            for (size_t r = ch * chunk; r < std::min(N, (ch + 1) * chunk); r++) {
                synth_fill_row(c, (u32)ci, seed, inter_seed, (u32)r, pre.data(), mainv.data(), inter.data());
                const u32* src = tree == 0 ? pre.data() : tree == 1 ? mainv.data() : inter.data();
                for (int k = 0; k < n; k++) nat[k][r] = src[k];
            }
        });
        size_t base = cols.size();
        cols.resize(base + n);
        parallel_for(n, n_threads, [&](size_t k) { cols[base + k].resize(N); finalize_column(nat[k].data(), cols[base + k].data(), c.log_size); });
        for (int k = 0; k < n; k++) logs.push_back(c.log_size);
    }
}

static inline u64 inter_seed_from(QM31 z) { return ((u64)z.a.a << 32) ^ (u64)z.a.b ^ ((u64)z.b.a << 16) ^ ((u64)z.b.b << 48); }

struct ProveStats { double t_commit, t_composition, t_oods, t_quotients, t_fri, t_decommit; };

static inline AirHooks synth_hooks(const AirSpec& air, const PcsConfig& cfg) {
    AirHooks h;
    h.composition_log = composition_log(air, cfg);
    h.compute_composition = [&air, cfg](const Twiddles& tw, const std::vector<TreeData>& trees, QM31 rc, int nt) { return compute_composition(air, cfg, tw, trees, rc, nt); };
    h.mask_points = [&air](QPt oods) { return mask_points(air, oods); };
    h.eval_composition_at_point = [&air](QPt p, const SampledValues& sv, QM31 rc) { return eval_composition_at_point(air, p, sv, rc); };
    return h;
}
static inline Proof prove_core(CommitmentSchemeProver& cs, Channel& ch, const PcsConfig& cfg, const Twiddles& tw, const AirHooks& air, int n_threads);

// The synthetic-machine prove: orchestration of reference prover/src/machine.rs:184-296 followed
// by stwo prover/mod.rs::prove and prover/pcs/mod.rs::prove_values.
static inline Proof prove_synth(const AirSpec& air, const PcsConfig& cfg, u64 seed, const uint8_t* ad, size_t ad_len, int n_threads) {
    int max_log = 0; for (auto& c : air.comps) max_log = std::max(max_log, c.log_size);
    // machine.rs:184-194: twiddles for CanonicCoset(max_log + LOG_CONSTRAINT_DEGREE + log_blowup).half_coset
    Twiddles tw = precompute_twiddles(max_log + cfg.log_constraint_degree + (int)cfg.log_blowup - 1);
    Channel ch;
    for (size_t i = 0; i < ad_len; i++) ch.mix_u64(ad[i]);                 // machine.rs:198-200
    CommitmentSchemeProver cs; cs.cfg = cfg; cs.tw = &tw; cs.n_threads = n_threads;
    for (auto& c : air.comps) ch.mix_u64((u64)c.log_size);                  // machine.rs:204-206
    std::vector<std::vector<u32>> cols; std::vector<int> logs;
    synth_tree_columns(air, 0, seed, 0, n_threads, cols, logs); cs.commit_evals(std::move(cols), logs, ch);   // tree 0
    synth_tree_columns(air, 1, seed, 0, n_threads, cols, logs); cs.commit_evals(std::move(cols), logs, ch);   // tree 1
    QM31 z = ch.draw_secure_felt();                                        // machine.rs:239-240 draw_lookup_elements
    synth_tree_columns(air, 2, seed, inter_seed_from(z), n_threads, cols, logs);
    std::vector<QM31> claimed(air.comps.size(), qm31_zero());
    ch.mix_felts(claimed.data(), claimed.size());                          // machine.rs:262
    cs.commit_evals(std::move(cols), logs, ch);                            // tree 2 (machine.rs:263)
    return prove_core(cs, ch, cfg, tw, synth_hooks(air, cfg), n_threads);
}

static inline Proof prove_core(CommitmentSchemeProver& cs, Channel& ch, const PcsConfig& cfg, const Twiddles& tw, const AirHooks& air, int n_threads) {
    // ---- stwo::prover::prove ----  (T trace trees are committed: the reference has 3 — preprocessed, main, interaction, machine.rs:208-263 —
    // Stwo itself takes any TreeVec; the composition polynomial's tree is number T)
    const int T = (int)cs.trees.size();
    QM31 random_coeff = ch.draw_secure_felt();
    std::vector<std::vector<u32>> comp = air.compute_composition(tw, cs.trees, random_coeff, n_threads);
    int clog = air.composition_log;
    cs.commit_polys(comp, std::vector<int>(4, clog), ch);                  // tree T
    QPt oods = get_random_point(ch);
    auto points = air.mask_points(oods);
    points.resize(T + 1); points[T].assign(4, std::vector<QPt>{oods});

    // ---- prove_values ----
    Proof proof; proof.config = cfg;
    std::vector<std::vector<std::vector<PointSample>>> samples(T + 1);
    proof.sampled_values.resize(T + 1);
    for (int t = 0; t <= T; t++) {
        size_t nc = cs.trees[t].polys.size();
        samples[t].resize(nc); proof.sampled_values[t].resize(nc);
        parallel_for(nc, n_threads, [&](size_t c) {
            for (auto& pt : points[t][c]) {
                QM31 v = eval_at_point(cs.trees[t].polys[c].data(), cs.trees[t].logs[c], pt);
                samples[t][c].push_back({pt, v}); proof.sampled_values[t][c].push_back(v);
            }
        });
    }
    { std::vector<QM31> flat; for (auto& t : proof.sampled_values) for (auto& c : t) for (auto& v : c) flat.push_back(v); ch.mix_felts(flat.data(), flat.size()); }
    QM31 q_coeff = ch.draw_secure_felt();
    // compute_fri_quotients: all columns, sorted by LDE log size descending (stable), grouped
    struct CRef { const u32* data; int log; const std::vector<PointSample>* s; };
    std::vector<CRef> all;
    for (int t = 0; t <= T; t++) for (size_t c = 0; c < cs.trees[t].evals.size(); c++)
        all.push_back({cs.trees[t].evals[c].data(), cs.trees[t].logs[c] + (int)cfg.log_blowup, &samples[t][c]});
    std::stable_sort(all.begin(), all.end(), [](const CRef& a, const CRef& b) { return a.log > b.log; });
    std::vector<SecureCols> quotients;
    for (size_t i = 0; i < all.size();) {
        size_t j = i; while (j < all.size() && all[j].log == all[i].log) j++;
        std::vector<const u32*> gcols; std::vector<const std::vector<PointSample>*> gs;
        for (size_t k = i; k < j; k++) { gcols.push_back(all[k].data); gs.push_back(all[k].s); }
        quotients.push_back(accumulate_quotients(all[i].log, gcols, q_coeff, sample_batches_new(gs), n_threads));
        i = j;
    }
    FriProverState fri = fri_commit(ch, cfg, std::move(quotients));
    proof.proof_of_work = ch.grind(cfg.pow_bits);
    ch.mix_u64(proof.proof_of_work);
    std::map<int, std::vector<size_t>> qpos = fri_decommit(ch, cfg, fri, proof);
    proof.decommitments.resize(T + 1); proof.queried_values.resize(T + 1);
    for (int t = 0; t <= T; t++) {
        std::vector<ColRef> refs;
        for (size_t c = 0; c < cs.trees[t].evals.size(); c++) refs.push_back({cs.trees[t].evals[c].data(), cs.trees[t].logs[c] + (int)cfg.log_blowup});
        merkle_decommit(cs.trees[t].merkle, qpos, refs, proof.queried_values[t], proof.decommitments[t]);
        proof.commitments.push_back(cs.trees[t].merkle.root());
    }
    // sanity check of stwo prover/mod.rs::prove
    QM31 ce[4]; for (int k = 0; k < 4; k++) ce[k] = proof.sampled_values[T][k][0];
    if (!qm31_eq(from_partial_evals(ce), air.eval_composition_at_point(oods, proof.sampled_values, random_coeff)))
        throw std::string("ConstraintsNotSatisfied");
    return proof;
}

// ---------------- verifier ----------------
static inline std::string verify_core(Channel& ch, const PcsConfig& cfg, const Proof& proof, std::vector<std::vector<int>> tree_logs, const AirHooks& air);
// core/verifier.rs::verify + core/pcs/verifier.rs::verify_values + core/fri.rs FriVerifier, with the
// transcript prefix of reference prover/src/machine.rs:299-485.
static inline std::string verify_synth(const AirSpec& air, const PcsConfig& cfg, const Proof& proof, const uint8_t* ad, size_t ad_len) {
    if (proof.commitments.size() != 4 || proof.sampled_values.size() != 4) return "InvalidStructure";
    if (proof.config.pow_bits != cfg.pow_bits || proof.config.log_blowup != cfg.log_blowup || proof.config.n_queries != cfg.n_queries ||
        proof.config.log_last_layer_degree_bound != cfg.log_last_layer_degree_bound) return "ConfigMismatch";
    Channel ch;
    for (size_t i = 0; i < ad_len; i++) ch.mix_u64(ad[i]);
    for (auto& c : air.comps) ch.mix_u64((u64)c.log_size);
    // column log sizes per tree (polynomial degree bounds)
    std::vector<std::vector<int>> tree_logs(3);
    for (auto& c : air.comps) {
        for (int k = 0; k < c.n_pre; k++) tree_logs[0].push_back(c.log_size);
        for (int k = 0; k < c.n_main; k++) tree_logs[1].push_back(c.log_size);
        for (int k = 0; k < c.n_inter; k++) tree_logs[2].push_back(c.log_size);
    }
    ch.mix_root(proof.commitments[0]);
    ch.mix_root(proof.commitments[1]);
    (void)ch.draw_secure_felt();  // lookup elements
    std::vector<QM31> claimed(air.comps.size(), qm31_zero());
    ch.mix_felts(claimed.data(), claimed.size());
    ch.mix_root(proof.commitments[2]);
    return verify_core(ch, cfg, proof, tree_logs, synth_hooks(air, cfg));
}

// core/verifier.rs::verify from the point where the trace trees are in the transcript.
static inline std::string verify_core(Channel& ch, const PcsConfig& cfg, const Proof& proof, std::vector<std::vector<int>> tree_logs, const AirHooks& air) {
    const int T = (int)tree_logs.size();                       // trace trees in the transcript so far; the composition tree is number T
    if ((int)proof.commitments.size() != T + 1 || (int)proof.sampled_values.size() != T + 1) return "InvalidStructure";
    if (proof.config.pow_bits != cfg.pow_bits || proof.config.log_blowup != cfg.log_blowup || proof.config.n_queries != cfg.n_queries ||
        proof.config.log_last_layer_degree_bound != cfg.log_last_layer_degree_bound) return "ConfigMismatch";
    tree_logs.resize(T + 1);
    tree_logs[T].assign(4, air.composition_log);
    QM31 random_coeff = ch.draw_secure_felt();
    ch.mix_root(proof.commitments[T]);
    QPt oods = get_random_point(ch);
    auto points = air.mask_points(oods);
    points.resize(T + 1); points[T].assign(4, std::vector<QPt>{oods});
    for (int t = 0; t <= T; t++) {
        if (proof.sampled_values[t].size() != points[t].size()) return "InvalidStructure";
        for (size_t c = 0; c < points[t].size(); c++) if (proof.sampled_values[t][c].size() != points[t][c].size()) return "InvalidStructure";
    }
    QM31 ce[4]; for (int k = 0; k < 4; k++) ce[k] = proof.sampled_values[T][k][0];
    if (!qm31_eq(from_partial_evals(ce), air.eval_composition_at_point(oods, proof.sampled_values, random_coeff))) return "OodsNotMatching";
    // verify_values
    { std::vector<QM31> flat; for (auto& t : proof.sampled_values) for (auto& c : t) for (auto& v : c) flat.push_back(v); ch.mix_felts(flat.data(), flat.size()); }
    QM31 q_coeff = ch.draw_secure_felt();
    // FRI commit phase (FriVerifier::commit): column bounds = distinct LDE log sizes, descending
    std::set<int, std::greater<int>> size_set;
    for (int t = 0; t <= T; t++) for (int l : tree_logs[t]) size_set.insert(l + (int)cfg.log_blowup);
    std::vector<int> col_logs(size_set.begin(), size_set.end());
    ch.mix_root(proof.first_layer.commitment);
    QM31 first_alpha = ch.draw_secure_felt();
    std::vector<QM31> inner_alphas;
    {
        int layer_log = col_logs[0] - 1;
        size_t expected_layers = 0;
        int last_log = (int)(cfg.log_last_layer_degree_bound + cfg.log_blowup);
        if (layer_log > last_log) expected_layers = layer_log - last_log;
        if (proof.inner_layers.size() != expected_layers) return "InvalidNumFriLayers";
        for (auto& l : proof.inner_layers) { ch.mix_root(l.commitment); inner_alphas.push_back(ch.draw_secure_felt()); }
        if (proof.last_layer_poly.size() > ((size_t)1 << cfg.log_last_layer_degree_bound)) return "LastLayerDegreeInvalid";
        {   // the prover mixed the bit-reversed storage order of the (full-length) coefficient vector
            const size_t bound = (size_t)1 << cfg.log_last_layer_degree_bound;
            if (proof.last_layer_poly.size() != bound) return "LastLayerDegreeInvalid";
            std::vector<QM31> m(bound); for (size_t i = 0; i < bound; i++) m[i] = proof.last_layer_poly[bit_reverse_index((u32)i, (int)cfg.log_last_layer_degree_bound)];
            ch.mix_felts(m.data(), m.size());
        }
    }
    if (!ch.verify_pow_nonce(cfg.pow_bits, proof.proof_of_work)) return "ProofOfWork";
    ch.mix_u64(proof.proof_of_work);
    int max_log = col_logs[0];
    std::vector<size_t> queries = queries_generate(ch, max_log, cfg.n_queries);
    std::map<int, std::vector<size_t>> qpos;
    for (int l : col_logs) qpos[l] = queries_fold(queries, max_log - l);
    // Merkle decommitments of the T + 1 trees
    for (int t = 0; t <= T; t++) {
        std::vector<int> lde_logs; for (int l : tree_logs[t]) lde_logs.push_back(l + (int)cfg.log_blowup);
        std::string e = merkle_verify(proof.commitments[t], lde_logs, qpos, proof.queried_values[t], proof.decommitments[t], cfg.hash_mode);
        if (!e.empty()) return "MerkleVerification(tree " + std::to_string(t) + "): " + e;
    }
    // fri_answers: quotient values at the queried positions, per size group (descending)
    // queried_values[t] order: by layer (descending size), by query position, by column.
    std::vector<size_t> qv_pos(T + 1, 0);
    std::vector<std::vector<QM31>> answers;  // per column-size group, per query
    for (int L : col_logs) {
        // columns of this size in flattened (tree, column) order
        std::vector<std::pair<int, size_t>> members;
        std::vector<std::vector<PointSample>> ms;
        for (int t = 0; t <= T; t++) for (size_t c = 0; c < tree_logs[t].size(); c++) if (tree_logs[t][c] + (int)cfg.log_blowup == L) {
            members.push_back({t, c});
            std::vector<PointSample> s; for (size_t k = 0; k < points[t][c].size(); k++) s.push_back({points[t][c][k], proof.sampled_values[t][c][k]});
            ms.push_back(s);
        }
        std::vector<const std::vector<PointSample>*> msp; for (auto& s : ms) msp.push_back(&s);
        auto batches = sample_batches_new(msp);
        QuotientConstants qc = quotient_constants(batches, q_coeff);
        std::vector<size_t> n_in_tree(T + 1, 0); for (auto& m : members) n_in_tree[m.first]++;
        std::vector<QM31> ans;
        for (size_t q : qpos[L]) {
            std::vector<u32> row;
            for (int t = 0; t <= T; t++) for (size_t k = 0; k < n_in_tree[t]; k++) {
                if (qv_pos[t] >= proof.queried_values[t].size()) return "QueriedValuesTooShort";
                row.push_back(proof.queried_values[t][qv_pos[t]++]);
            }
            Pt dp = circle_domain_at(L, bit_reverse_index((u32)q, L));
            ans.push_back(accumulate_row_quotients(batches, row.data(), qc, dp));
        }
        answers.push_back(ans);
    }
    // ---- FRI decommit ----
    // first layer: rebuild the pairs, verify the Merkle decommitment, fold circle -> line
    struct Sparse { std::vector<size_t> pair_index; std::vector<QM31> v0, v1; };
    std::vector<Sparse> first_sparse;
    {
        size_t wi = 0;
        std::map<int, std::vector<size_t>> pos_by_log;
        std::vector<u32> qvals;  // values in Merkle order are rebuilt below
        std::vector<std::vector<QM31>> full_by_col;  // per column: values at all decommitment positions
        for (size_t ci = 0; ci < col_logs.size(); ci++) {
            int L = col_logs[ci];
            const std::vector<size_t>& cq = qpos[L];
            Sparse sp; std::vector<size_t> pos; std::vector<QM31> full;
            size_t i = 0;
            while (i < cq.size()) {
                size_t j = i; while (j < cq.size() && (cq[j] >> 1) == (cq[i] >> 1)) j++;
                size_t start = (cq[i] >> 1) << 1, qi = i; QM31 pv[2];
                for (size_t p = start; p < start + 2; p++) {
                    pos.push_back(p);
                    if (qi < j && cq[qi] == p) { pv[p - start] = answers[ci][qi]; qi++; }
                    else { if (wi >= proof.first_layer.fri_witness.size()) return "FirstLayerEvaluationsInvalid"; pv[p - start] = proof.first_layer.fri_witness[wi++]; }
                    full.push_back(pv[p - start]);
                }
                sp.pair_index.push_back(start >> 1); sp.v0.push_back(pv[0]); sp.v1.push_back(pv[1]);
                i = j;
            }
            pos_by_log[L] = pos; first_sparse.push_back(sp); full_by_col.push_back(full);
        }
        if (wi != proof.first_layer.fri_witness.size()) return "FirstLayerEvaluationsInvalid";
        // Merkle: column log sizes = each column log ×4 coords; queried values ordered by layer desc, position, coord
        std::vector<int> mlogs; for (int L : col_logs) for (int k = 0; k < 4; k++) mlogs.push_back(L);
        for (size_t ci = 0; ci < col_logs.size(); ci++)
            for (QM31 v : full_by_col[ci]) { u32 w[4]; qm31_store(w, v); qvals.insert(qvals.end(), w, w + 4); }
        std::string e = merkle_verify(proof.first_layer.commitment, mlogs, pos_by_log, qvals, proof.first_layer.decommitment, cfg.hash_mode);
        if (!e.empty()) return "FirstLayerCommitmentInvalid: " + e;
    }
    // inner layers
    std::vector<size_t> lq = queries_fold(queries, 1);
    std::vector<QM31> lvals(lq.size(), qm31_zero());
    size_t fci = 0;
    QM31 prev_alpha = first_alpha;
    int layer_log = col_logs[0] - 1;
    Coset dom = coset_half_odds(layer_log);
    auto fold_in_circles = [&](int cur_layer_log) -> bool {
        while (fci < col_logs.size() && col_logs[fci] - 1 == cur_layer_log) {
            QM31 a = cfg.fri_alpha_mode == FRI_ALPHA_PREV ? prev_alpha : first_alpha;
            QM31 a2 = qm31_sqr(a);
            const Sparse& sp = first_sparse[fci];
            if (sp.pair_index.size() != lq.size()) return false;
            int L = col_logs[fci];
            for (size_t i = 0; i < lq.size(); i++) {
                if (sp.pair_index[i] != lq[i]) return false;
                Pt p = circle_domain_at(L, bit_reverse_index((u32)(sp.pair_index[i] << 1), L));
                QM31 f0 = sp.v0[i], f1 = sp.v1[i];
                ibutterfly_q(f0, f1, m31_inv(p.y));
                QM31 folded = qm31_add(f0, qm31_mul(a, f1));
                lvals[i] = qm31_add(qm31_mul(lvals[i], a2), folded);
            }
            fci++;
        }
        return true;
    };
    for (size_t li = 0; li < proof.inner_layers.size(); li++) {
        if (!fold_in_circles(layer_log)) return "FirstLayerFoldMismatch";
        const FriLayerProof& lp = proof.inner_layers[li];
        // rebuild pairs
        std::vector<size_t> pos, nq; std::vector<QM31> full, nv; size_t wi = 0, i = 0;
        while (i < lq.size()) {
            size_t j = i; while (j < lq.size() && (lq[j] >> 1) == (lq[i] >> 1)) j++;
            size_t start = (lq[i] >> 1) << 1, qi = i; QM31 pv[2];
            for (size_t p = start; p < start + 2; p++) {
                pos.push_back(p);
                if (qi < j && lq[qi] == p) { pv[p - start] = lvals[qi]; qi++; }
                else { if (wi >= lp.fri_witness.size()) return "InnerLayerEvaluationsInvalid"; pv[p - start] = lp.fri_witness[wi++]; }
                full.push_back(pv[p - start]);
            }
            u32 x = coset_at(dom, bit_reverse_index((u32)start, dom.log)).x;
            QM31 f0 = pv[0], f1 = pv[1];
            ibutterfly_q(f0, f1, m31_inv(x));
            nq.push_back(start >> 1); nv.push_back(qm31_add(f0, qm31_mul(inner_alphas[li], f1)));
            i = j;
        }
        if (wi != lp.fri_witness.size()) return "InnerLayerEvaluationsInvalid";
        std::map<int, std::vector<size_t>> pos_by_log; pos_by_log[layer_log] = pos;
        std::vector<u32> qvals; for (QM31 v : full) { u32 w[4]; qm31_store(w, v); qvals.insert(qvals.end(), w, w + 4); }
        std::string e = merkle_verify(lp.commitment, std::vector<int>(4, layer_log), pos_by_log, qvals, lp.decommitment, cfg.hash_mode);
        if (!e.empty()) return "InnerLayerCommitmentInvalid(" + std::to_string(li) + "): " + e;
        lq = nq; lvals = nv; prev_alpha = inner_alphas[li];
        layer_log--; dom = coset_double(dom);
    }
    if (fci != col_logs.size()) return "FirstLayerColumnsNotConsumed";
    // last layer
    for (size_t i = 0; i < lq.size(); i++) {
        u32 x = coset_at(dom, bit_reverse_index((u32)lq[i], dom.log)).x;
        if (!qm31_eq(lvals[i], line_poly_eval_ordered(proof.last_layer_poly, qm31_from_m31(x)))) return "LastLayerEvaluationsInvalid";
    }
    return "";
}

}  // namespace orc
