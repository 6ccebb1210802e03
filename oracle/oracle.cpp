// ORACLE — TEST INFRASTRUCTURE ONLY.  C ABI (for ctypes) over the CPU restatement in this directory.
// Never linked, imported or executed by the product path; see fields.h for the parity statement
// ("PARITY UNPINNED": Stwo @0790eba is not vendored under /root/reference and cannot be built here).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include "fields.h"
#include "blake2s.h"
#include "poly.h"
#include "merkle.h"
#include "air.h"
#include "pcs.h"
#include "constraints.h"
#include "air_generic.h"
#include "logup.h"
#include "backend_ops.h"

using namespace orc;

static PcsConfig cfg_from(const int* c) {
    // [pow_bits, log_blowup, n_queries, log_last_layer_degree_bound, hash_mode, fri_alpha_mode, log_constraint_degree]
    PcsConfig r; r.pow_bits = c[0]; r.log_blowup = c[1]; r.n_queries = c[2]; r.log_last_layer_degree_bound = c[3];
    r.hash_mode = c[4]; r.fri_alpha_mode = c[5]; r.log_constraint_degree = c[6];
    return r;
}
static AirSpec air_from(const int* comps, int n) {
    AirSpec a;
    for (int i = 0; i < n; i++) a.comps.push_back({comps[5 * i], comps[5 * i + 1], comps[5 * i + 2], comps[5 * i + 3], comps[5 * i + 4]});   // (log_size, n_pre, n_main, n_inter, log_cd)
    return a;
}

extern "C" {

// ---- fields / circle ----
uint32_t orc_m31_add(uint32_t a, uint32_t b) { return m31_add(a, b); }
uint32_t orc_m31_sub(uint32_t a, uint32_t b) { return m31_sub(a, b); }
uint32_t orc_m31_mul(uint32_t a, uint32_t b) { return m31_mul(a, b); }
uint32_t orc_m31_inv(uint32_t a) { return m31_inv(a); }
uint32_t orc_m31_reduce(uint64_t x) { return m31_reduce(x); }
void orc_cm31_mul(const uint32_t* a, const uint32_t* b, uint32_t* o) { CM31 r = cm31_mul(cm31(a[0], a[1]), cm31(b[0], b[1])); o[0] = r.a; o[1] = r.b; }
void orc_cm31_inv(const uint32_t* a, uint32_t* o) { CM31 r = cm31_inv(cm31(a[0], a[1])); o[0] = r.a; o[1] = r.b; }
void orc_qm31_mul(const uint32_t* a, const uint32_t* b, uint32_t* o) { qm31_store(o, qm31_mul(qm31_load(a), qm31_load(b))); }
void orc_qm31_add(const uint32_t* a, const uint32_t* b, uint32_t* o) { qm31_store(o, qm31_add(qm31_load(a), qm31_load(b))); }
void orc_qm31_inv(const uint32_t* a, uint32_t* o) { qm31_store(o, qm31_inv(qm31_load(a))); }
void orc_circle_point(uint32_t index, uint32_t* xy) { Pt p = pt_from_index(index); xy[0] = p.x; xy[1] = p.y; }
void orc_circle_domain_at(int log, uint32_t i, uint32_t* xy) { Pt p = circle_domain_at(log, i); xy[0] = p.x; xy[1] = p.y; }
void orc_canonic_coset_at(int log, uint32_t i, uint32_t* xy) { Pt p = coset_at(coset_odds(log), i); xy[0] = p.x; xy[1] = p.y; }
uint32_t orc_bit_reverse_index(uint32_t i, int log) { return bit_reverse_index(i, log); }
uint32_t orc_coset_index_to_circle_domain_index(uint32_t c, int log) { return coset_index_to_circle_domain_index(c, log); }

// ---- poly ----
void* orc_twiddles_new(int root_log) { return new Twiddles(precompute_twiddles(root_log)); }
void orc_twiddles_free(void* t) { delete (Twiddles*)t; }
void orc_twiddles_get(void* t, uint32_t* tw, uint32_t* itw) {
    Twiddles* T = (Twiddles*)t;
    memcpy(tw, T->tw.data(), T->tw.size() * 4); memcpy(itw, T->itw.data(), T->itw.size() * 4);
}
void orc_interpolate(void* t, uint32_t* values, int n) { interpolate(values, n, *(Twiddles*)t); }
void orc_evaluate(void* t, const uint32_t* coeffs, int n_coef, uint32_t* out, int n) { evaluate(coeffs, n_coef, out, n, *(Twiddles*)t); }
void orc_eval_at_point(const uint32_t* coeffs, int n, const uint32_t* pt, uint32_t* out) {
    QPt p = {qm31_load(pt), qm31_load(pt + 4)};
    qm31_store(out, eval_at_point(coeffs, n, p));
}
uint32_t orc_eval_basis_at_m31_point(const uint32_t* coeffs, int n, uint32_t x, uint32_t y) { Pt p = {x, y}; return eval_basis_at_m31_point(coeffs, n, p); }
void orc_coset_order_to_circle_domain_order(const uint32_t* in, uint32_t* out, int n) { coset_order_to_circle_domain_order(in, out, n); }
void orc_bit_reverse(uint32_t* v, int n) { bit_reverse_inplace(v, n); }
void orc_finalize_column(const uint32_t* in, uint32_t* out, int n) { finalize_column(in, out, n); }

// ---- hash / merkle ----
void orc_blake2s(const uint8_t* data, size_t len, uint8_t* out) { blake2s_hash(data, len, out); }
void orc_blake2s_compress(uint32_t* h, const uint32_t* m, uint32_t t0, uint32_t t1, uint32_t f0, uint32_t f1) { b2s_compress(h, m, t0, t1, f0, f1); }
void orc_hash_node(const uint32_t* children /*16 words or NULL*/, const uint32_t* vals, size_t nvals, int mode, uint32_t* out) {
    Hash l, r;
    if (children) { memcpy(l.w, children, 32); memcpy(r.w, children + 8, 32); }
    Hash h = hash_node(children ? &l : nullptr, children ? &r : nullptr, vals, nvals, mode);
    memcpy(out, h.w, 32);
}
// layers_out (optional): concatenation of layers from the largest (leaves) down to the root, 8 words per node.
int orc_merkle_commit(const uint32_t** cols, const int* logs, int ncols, int mode, uint32_t* root_out, uint32_t* layers_out) {
    std::vector<ColRef> refs; for (int i = 0; i < ncols; i++) refs.push_back({cols[i], logs[i]});
    MerkleTree t = merkle_commit(refs, mode);
    memcpy(root_out, t.root().w, 32);
    if (layers_out) { size_t o = 0; for (int l = (int)t.layers.size() - 1; l >= 0; l--) { memcpy(layers_out + o, t.layers[l].data(), t.layers[l].size() * 32); o += t.layers[l].size() * 8; } }
    return (int)t.layers.size();
}

// ---- channel ----
void* orc_channel_new() { return new Channel(); }
void orc_channel_free(void* c) { delete (Channel*)c; }
void orc_channel_digest(void* c, uint32_t* out) { memcpy(out, ((Channel*)c)->digest.w, 32); }
void orc_channel_set_digest(void* c, const uint32_t* in) { memcpy(((Channel*)c)->digest.w, in, 32); }
void orc_channel_mix_u64(void* c, uint64_t v) { ((Channel*)c)->mix_u64(v); }
void orc_channel_mix_u32s(void* c, const uint32_t* d, size_t n) { ((Channel*)c)->mix_u32s(d, n); }
void orc_channel_mix_root(void* c, const uint32_t* root) { Hash h; memcpy(h.w, root, 32); ((Channel*)c)->mix_root(h); }
void orc_channel_mix_felts(void* c, const uint32_t* f, size_t n) { std::vector<QM31> v(n); for (size_t i = 0; i < n; i++) v[i] = qm31_load(f + 4 * i); ((Channel*)c)->mix_felts(v.data(), n); }
void orc_channel_draw_secure_felt(void* c, uint32_t* out) { qm31_store(out, ((Channel*)c)->draw_secure_felt()); }
void orc_channel_draw_u32s(void* c, uint32_t* out) { ((Channel*)c)->draw_u32s(out); }
uint64_t orc_channel_grind(void* c, uint32_t pow_bits) { return ((Channel*)c)->grind(pow_bits); }
int orc_channel_verify_pow(void* c, uint32_t pow_bits, uint64_t nonce) { return ((Channel*)c)->verify_pow_nonce(pow_bits, nonce); }
void orc_get_random_point(void* c, uint32_t* out) { QPt p = get_random_point(*(Channel*)c); qm31_store(out, p.x); qm31_store(out + 4, p.y); }

// ---- quotients ----
// batches: n_batches; points[8*b]; counts[b]; col_idx[sum]; values[4*sum]
void orc_accumulate_quotients(int log, const uint32_t** cols, int ncols, const uint32_t* alpha, int n_batches, const uint32_t* points,
                              const int* counts, const int* col_idx, const uint32_t* values, int n_threads, uint32_t** out4) {
    std::vector<const u32*> c(cols, cols + ncols);
    std::vector<ColumnSampleBatch> b(n_batches); size_t k = 0;
    for (int i = 0; i < n_batches; i++) {
        b[i].point = {qm31_load(points + 8 * i), qm31_load(points + 8 * i + 4)};
        for (int j = 0; j < counts[i]; j++, k++) b[i].cols.push_back({(size_t)col_idx[k], qm31_load(values + 4 * k)});
    }
    SecureCols s = accumulate_quotients(log, c, qm31_load(alpha), b, n_threads);
    for (int q = 0; q < 4; q++) memcpy(out4[q], s.c[q].data(), s.c[q].size() * 4);
}

// ---- FRI folds (SoA: 4 coordinate columns) ----
void orc_fold_line(const uint32_t** src4, int log, const uint32_t* alpha, uint32_t** dst4) {
    // src on LineDomain(half_odds(log)) -- only correct for the first-generation line domain when
    // n_doublings == 0; use orc_fold_line_dom for later layers.
    size_t n = (size_t)1 << log; std::vector<QM31> e(n);
    for (size_t i = 0; i < n; i++) e[i] = qm31(src4[0][i], src4[1][i], src4[2][i], src4[3][i]);
    std::vector<QM31> o = fold_line(e, coset_half_odds(log), qm31_load(alpha));
    for (size_t i = 0; i < o.size(); i++) { dst4[0][i] = o[i].a.a; dst4[1][i] = o[i].a.b; dst4[2][i] = o[i].b.a; dst4[3][i] = o[i].b.b; }
}
// line domain = half_odds(log + n_doublings) doubled n_doublings times (log size `log`)
void orc_fold_line_dom(const uint32_t** src4, int log, int n_doublings, const uint32_t* alpha, uint32_t** dst4) {
    size_t n = (size_t)1 << log; std::vector<QM31> e(n);
    for (size_t i = 0; i < n; i++) e[i] = qm31(src4[0][i], src4[1][i], src4[2][i], src4[3][i]);
    Coset d = coset_half_odds(log + n_doublings); for (int i = 0; i < n_doublings; i++) d = coset_double(d);
    std::vector<QM31> o = fold_line(e, d, qm31_load(alpha));
    for (size_t i = 0; i < o.size(); i++) { dst4[0][i] = o[i].a.a; dst4[1][i] = o[i].a.b; dst4[2][i] = o[i].b.a; dst4[3][i] = o[i].b.b; }
}
void orc_fold_circle_into_line(uint32_t** dst4, const uint32_t** src4, int src_log, const uint32_t* alpha) {
    SecureCols s; s.init(src_log);
    for (int q = 0; q < 4; q++) memcpy(s.c[q].data(), src4[q], s.len() * 4);
    size_t n = s.len() / 2; std::vector<QM31> d(n);
    for (size_t i = 0; i < n; i++) d[i] = qm31(dst4[0][i], dst4[1][i], dst4[2][i], dst4[3][i]);
    fold_circle_into_line(d, s, qm31_load(alpha));
    for (size_t i = 0; i < n; i++) { dst4[0][i] = d[i].a.a; dst4[1][i] = d[i].a.b; dst4[2][i] = d[i].b.a; dst4[3][i] = d[i].b.b; }
}

// ---- synthetic AIR ----
int orc_n_constraints(const int* comp) { ComponentSpec c = {comp[0], comp[1], comp[2], comp[3], 0}; return n_constraints(c); }
// Finalized (bit-reversed circle-domain order) columns of one tree; out[i] must hold 2^log_i words.
void orc_synth_tree_columns(const int* comps, int ncomp, int tree, uint64_t seed, uint64_t inter_seed, int n_threads, uint32_t** out) {
    AirSpec air = air_from(comps, ncomp);
    std::vector<std::vector<u32>> cols; std::vector<int> logs;
    synth_tree_columns(air, tree, seed, inter_seed, n_threads, cols, logs);
    for (size_t i = 0; i < cols.size(); i++) memcpy(out[i], cols[i].data(), cols[i].size() * 4);
}
// Natural-order rows (row-major: n_pre+n_main+n_inter values per row) for one component.
void orc_synth_rows(const int* comp, uint32_t ci, uint64_t seed, uint64_t inter_seed, uint32_t row0, uint32_t nrows, uint32_t* out) {
    ComponentSpec c = {comp[0], comp[1], comp[2], comp[3], 0};
    size_t w = c.n_pre + c.n_main + c.n_inter;
    for (uint32_t r = 0; r < nrows; r++) synth_fill_row(c, ci, seed, inter_seed, row0 + r, out + r * w, out + r * w + c.n_pre, out + r * w + c.n_pre + c.n_main);
}
uint64_t orc_inter_seed_from(const uint32_t* z) { return inter_seed_from(qm31_load(z)); }

// LDE + commit of a batch of columns (bit-reversed evaluations in; coefficient and LDE out).
void orc_lde_commit(void* tw, uint32_t** cols /*in: evals, out: coeffs*/, const int* logs, int ncols, int log_blowup, int mode, int n_threads,
                    uint32_t** evals_out, uint32_t* root_out) {
    Twiddles* T = (Twiddles*)tw;
    parallel_for(ncols, n_threads, [&](size_t i) {
        interpolate(cols[i], logs[i], *T);
        evaluate(cols[i], logs[i], evals_out[i], logs[i] + log_blowup, *T);
    });
    std::vector<ColRef> refs; for (int i = 0; i < ncols; i++) refs.push_back({evals_out[i], logs[i] + log_blowup});
    merkle_n_threads() = n_threads;
    MerkleTree t = merkle_commit(refs, mode);
    memcpy(root_out, t.root().w, 32);
}

// ---- prove / verify ----
static thread_local std::string g_err;
const char* orc_last_error() { return g_err.c_str(); }

// Returns a malloc'd word buffer (free with orc_free) or NULL on error.
uint32_t* orc_prove_synth(const int* comps, int ncomp, const int* cfg, uint64_t seed, const uint8_t* ad, size_t ad_len, int n_threads, size_t* n_words) {
    try {
        Proof p = prove_synth(air_from(comps, ncomp), cfg_from(cfg), seed, ad, ad_len, n_threads);
        std::vector<u32> w = proof_serialize(p);
        uint32_t* out = (uint32_t*)malloc(w.size() * 4);
        memcpy(out, w.data(), w.size() * 4);
        *n_words = w.size();
        return out;
    } catch (const std::string& e) { g_err = e; return nullptr; }
}
void orc_free(void* p) { free(p); }

// 0 = accepted; otherwise error text in orc_last_error().
int orc_verify_synth(const int* comps, int ncomp, const int* cfg, const uint32_t* words, size_t n_words, const uint8_t* ad, size_t ad_len) {
    Proof p;
    if (!proof_deserialize(words, n_words, p)) { g_err = "Deserialize"; return 1; }
    std::string e;
    try { e = verify_synth(air_from(comps, ncomp), cfg_from(cfg), p, ad, ad_len); } catch (const std::string& x) { e = x; }
    if (e.empty()) return 0;
    g_err = e; return 1;
}

// ---- generic (recorded-AIR) prove / verify sessions (air_generic.h) ----
void* orc_prover_new(const int* cfg, int max_log, int n_threads) { return new ProverSession(cfg_from(cfg), max_log, n_threads); }
void orc_prover_free(void* p) { delete (ProverSession*)p; }
void* orc_prover_channel(void* p) { return &((ProverSession*)p)->ch; }          // use with orc_channel_* (do not free)
// columns: bit-reversed evaluations on CanonicCoset(log).circle_domain(); copied.
void orc_prover_commit(void* p, const uint32_t** cols, const int* logs, int n, uint32_t* root_out) {
    ProverSession* s = (ProverSession*)p;
    std::vector<std::vector<u32>> c(n); std::vector<int> l(logs, logs + n);
    for (int i = 0; i < n; i++) c[i].assign(cols[i], cols[i] + ((size_t)1 << logs[i]));
    s->cs.commit_evals(std::move(c), l, s->ch);
    memcpy(root_out, s->cs.trees.back().merkle.root().w, 32);
}
uint32_t* orc_prover_prove(void* p, const uint32_t* air_words, size_t n_air, size_t* n_words) {
    try {
        GAir air; if (!gair_decode(air_words, n_air, air)) throw std::string("malformed AIR description");
        Proof pr = ((ProverSession*)p)->prove(air);
        std::vector<u32> w = proof_serialize(pr);
        uint32_t* out = (uint32_t*)malloc(w.size() * 4);
        memcpy(out, w.data(), w.size() * 4);
        *n_words = w.size();
        return out;
    } catch (const std::string& e) { g_err = e; return nullptr; }
}
void* orc_verifier_new(const int* cfg) { VerifierSession* v = new VerifierSession(); v->cfg = cfg_from(cfg); return v; }
void orc_verifier_free(void* v) { delete (VerifierSession*)v; }
void* orc_verifier_channel(void* v) { return &((VerifierSession*)v)->ch; }
void orc_verifier_commit(void* v, const uint32_t* root, const int* logs, int n) { Hash h; memcpy(h.w, root, 32); ((VerifierSession*)v)->commit(h, std::vector<int>(logs, logs + n)); }
int orc_verifier_verify(void* v, const uint32_t* air_words, size_t n_air, const uint32_t* words, size_t n_words) {
    GAir air; if (!gair_decode(air_words, n_air, air)) { g_err = "malformed AIR description"; return 1; }
    Proof p;
    if (!proof_deserialize(words, n_words, p)) { g_err = "Deserialize"; return 1; }
    std::string e;
    try { e = ((VerifierSession*)v)->verify(air, p); } catch (const std::string& x) { e = x; }
    if (e.empty()) return 0;
    g_err = e; return 1;
}

void orc_eval_constraint_program(const uint32_t* prog, uint32_t n_instr, uint32_t n_regs, const uint32_t** cols, const uint32_t* econsts, const uint32_t* pw,
                                 const uint32_t* denom_inv, int log_size, int log_eval, uint32_t** acc4) {
    eval_constraint_program((const CInstr*)prog, n_instr, n_regs, cols, econsts, pw, denom_inv, log_size, log_eval, acc4);
}

void orc_logup_program(const uint32_t* prog, uint32_t n_instr, uint32_t n_regs, const uint32_t** cols, const uint32_t* econsts, int log_size, uint32_t n_logup_cols, uint32_t** out) {
    logup_program((const CInstr*)prog, n_instr, n_regs, cols, econsts, log_size, n_logup_cols, out);
}

void orc_logup_combine(const uint32_t** cols, uint32_t n_cols, const uint32_t* alpha_powers, const uint32_t* z, int log, uint32_t** out4) {
    logup_combine(cols, n_cols, alpha_powers, z, log, out4);
}
// mult_x may be NULL (constant numerator = scale_x); den_b4 NULL = a single fraction; prev4 NULL = first column
void orc_logup_finalize_col(int log, const uint32_t* mult_a, const uint32_t* scale_a, const uint32_t** den_a4, const uint32_t* mult_b, const uint32_t* scale_b,
                            const uint32_t** den_b4, const uint32_t** prev4, uint32_t** out4) {
    LogupFrac fa{mult_a, qm31_load(scale_a), {den_a4[0], den_a4[1], den_a4[2], den_a4[3]}};
    LogupFrac fb{};
    if (den_b4) fb = LogupFrac{mult_b, qm31_load(scale_b), {den_b4[0], den_b4[1], den_b4[2], den_b4[3]}};
    logup_finalize_col(log, fa, den_b4 ? &fb : nullptr, prev4, out4);
}
void orc_logup_finalize_last(int log, uint32_t** col4, uint32_t* claimed_sum) { logup_finalize_last(log, col4, claimed_sum); }
void orc_logup_set_threads(int n) { logup_n_threads() = n < 1 ? 1 : n; }

// ---- the remaining Backend supertraits (backend_ops.h) ----
void orc_batch_inverse_m31(const uint32_t* src, uint32_t* dst, size_t n) { batch_inverse_m31(src, dst, n); }
void orc_batch_inverse_qm31(const uint32_t** src4, uint32_t** dst4, size_t n) { batch_inverse_qm31(src4, dst4, n); }
void orc_secure_accumulate(uint32_t** dst4, const uint32_t** src4, size_t n) { secure_accumulate(dst4, src4, n); }
void orc_generate_secure_powers(const uint32_t* felt, size_t n, uint32_t* out) { generate_secure_powers(qm31_load(felt), n, out); }
void orc_fri_decompose(const uint32_t** src4, int log, uint32_t** g4, uint32_t* lambda) { qm31_store(lambda, fri_decompose(src4, log, g4)); }
void orc_commit_on_layer(int log, const uint32_t* prev, const uint32_t** cols, size_t n_cols, int mode, uint32_t* out) { commit_on_layer(log, prev, cols, n_cols, mode, out); }
void orc_channel_draw_secure_felts(void* c, size_t n, uint32_t* out) { std::vector<QM31> v = ((Channel*)c)->draw_secure_felts(n); for (size_t i = 0; i < n; i++) qm31_store(out + 4 * i, v[i]); }
// MerkleProver::decommit over columns given in commit order; one (log, count) per queried layer, positions back to back.
// Outputs: sizes in n_out[3] = {queried values, hashes, column witness words}; buffers must be large enough (call twice or over-allocate).
void orc_merkle_decommit(const uint32_t** cols, const int* logs, int ncols, int mode, const int* qlogs, const int* qcounts, int nq, const uint64_t* queries,
                         uint32_t* queried, uint32_t* hashes, uint32_t* colwit, size_t* n_out) {
    std::vector<ColRef> refs; for (int i = 0; i < ncols; i++) refs.push_back({cols[i], logs[i]});
    MerkleTree t = merkle_commit(refs, mode);
    std::map<int, std::vector<size_t>> qpl; size_t off = 0;
    for (int i = 0; i < nq; i++) { for (int k = 0; k < qcounts[i]; k++) qpl[qlogs[i]].push_back((size_t)queries[off + k]); off += qcounts[i]; }
    std::vector<u32> qv; MerkleDecommitment d;
    merkle_decommit(t, qpl, refs, qv, d);
    n_out[0] = qv.size(); n_out[1] = d.hash_witness.size(); n_out[2] = d.column_witness.size();
    if (queried) memcpy(queried, qv.data(), qv.size() * 4);
    if (hashes) for (size_t i = 0; i < d.hash_witness.size(); i++) memcpy(hashes + 8 * i, d.hash_witness[i].w, 32);
    if (colwit) memcpy(colwit, d.column_witness.data(), d.column_witness.size() * 4);
}

// Timed CPU baseline leg: one full prove, returns seconds (negative on error).
double orc_time_prove_synth(const int* comps, int ncomp, const int* cfg, uint64_t seed, int n_threads) {
    auto t0 = std::chrono::steady_clock::now();
    try { Proof p = prove_synth(air_from(comps, ncomp), cfg_from(cfg), seed, nullptr, 0, n_threads); (void)p; }
    catch (const std::string& e) { g_err = e; return -1.0; }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // extern "C"
