// The synthetic machines the bench and the parity suite prove: callers of the commitment scheme / prover of prover.h that play the
// role of `Machine::prove_with_extensions` (reference prover/src/machine.rs:130-297) for a synthetic wide-Fibonacci-style trace
// (SURVEY.md §8(d)): channel seeding (:197-206), preprocessed / main / interaction tree commits (:208-263), stwo::prover::prove
// (:286-290).
//   nx_prove_synth    the three trace trees are filled by a generator, the constraints are a hand-written kernel (air.hip)
//   nx_prove_machine  the reference's shape end to end: lookup elements drawn after the main commit (:239-240), the interaction trace
//                     is a REAL logup trace generated on the device from main-trace columns (LogupTraceGenerator: one fraction per
//                     column, finalize_col, finalize_last — reference traits.rs:124-145, chips/range_check/range256.rs:271-288),
//                     the claimed sums are mixed (:262), and the AIR — transition + degree-2 + logup constraints — is a RECORDED
//                     program compiled by nx_air_compile, i.e. the path a Rust shim takes for the reference's own AIR
// Both run on one GPU or as ONE proof on the GPUs of an nx_comm (row-sharded, prover.h).
#include "prover.h"
#include <mutex>
#include <array>

namespace nxhip {

struct Loc { size_t pre0, main0, inter0; };

static std::vector<Loc> locations(const nx_component_spec* comps, uint32_t n_comps, uint32_t* max_log) {
    std::vector<Loc> locs; size_t a = 0, b = 0, c = 0; *max_log = 0;
    for (uint32_t i = 0; i < n_comps; i++) { locs.push_back({a, b, c}); a += comps[i].n_pre; b += comps[i].n_main; c += comps[i].n_inter; *max_log = std::max(*max_log, comps[i].log_size); }
    return locs;
}

// ---------------------------------------------------------------- nx_prove_synth: hand-written constraint kernel ----------
// eval_composition_polynomial_at_point (the prover's OODS sanity check, stwo prover/mod.rs::prove)
static QM31 synth_eval_composition_at_point(const nx_component_spec* comps, uint32_t n_comps, const std::vector<Loc>& locs, QPt point,
                                            const SampledValues& sv, QM31 random_coeff) {
    QM31 acc = q_zero();
    for (uint32_t ci = 0; ci < n_comps; ci++) {
        const nx_component_spec& c = comps[ci];
        QM31 di = q_inv(coset_vanishing_q(c.log_size, point));
        auto add = [&](QM31 v) { acc = q_add(q_mul(acc, random_coeff), q_mul(di, v)); };
        auto M = [&](uint32_t k, int s = 0) { return sv[1][locs[ci].main0 + k][s]; };
        auto I = [&](uint32_t k) { return sv[2][locs[ci].inter0 + k][0]; };
        QM31 not_last = q_sub(q_one(), sv[0][locs[ci].pre0 + 1][0]);
        add(q_mul(q_sub(q_sub(M(0, 1), M(0)), q_one()), not_last));
        add(q_mul(q_sub(q_sub(M(1, 1), M(1)), M(0)), not_last));
        for (uint32_t k = 2; k < c.n_main; k++) if (!synth_col_is_free(k)) add(q_sub(q_sub(M(k), q_sqr(M(k - 1))), q_sqr(M(k - 2))));
        for (uint32_t k = 0; k < c.n_inter; k++) if (!synth_col_is_free(k)) add(q_sub(q_sub(I(k), q_sqr(I(k - 1))), q_sqr(I(k - 2))));
    }
    return acc;
}

struct SynthAir : AirProver {
    const nx_component_spec* comps; uint32_t n_comps; const std::vector<Loc>& locs;
    SynthAir(const nx_component_spec* c, uint32_t n, const std::vector<Loc>& l) : comps(c), n_comps(n), locs(l) {}

    // ComponentProvers::compute_composition_polynomial for the synthetic machine
    int compute_composition(CommitmentSchemeProver& cs, QM31 random_coeff, DevBuf* out_polys, uint32_t* out_log) override {
        nx_ctx* ctx = cs.ctx;
        size_t total = 0;
        for (uint32_t i = 0; i < n_comps; i++) total += synth_n_constraints(comps[i]);
        std::vector<QM31> powers(total);
        { QM31 a = q_one(); for (size_t i = 0; i < total; i++) { powers[i] = a; a = q_mul(a, random_coeff); } }
        std::map<uint32_t, SecureColumn> sub;  // evaluation-domain log size -> accumulation
        // alpha powers and vanishing denominators of ALL components in one owned upload (a prover2-style statement has dozens of components)
        std::vector<uint32_t> params; std::vector<size_t> off_pw(n_comps), off_den(n_comps);
        {
            size_t remaining = total;
            std::map<std::pair<uint32_t, uint32_t>, std::vector<uint32_t>> den_cache;
            for (uint32_t ci = 0; ci < n_comps; ci++) {
                const nx_component_spec& c = comps[ci];
                const uint32_t e = c.log_size + comp_log_cd(c.log_constraint_degree_bound, cs.cfg);
                const size_t nc = synth_n_constraints(c);
                off_pw[ci] = params.size();
                params.resize(params.size() + 4 * nc);   // this component takes the LAST nc remaining powers, reversed
                for (size_t j = 0; j < nc; j++) q_store(&params[off_pw[ci] + 4 * j], powers[remaining - 1 - j]);
                remaining -= nc;
                auto key = std::make_pair(c.log_size, e);
                if (!den_cache.count(key)) den_cache[key] = vanishing_denominators(c.log_size, e);
                const auto& den = den_cache[key];
                off_den[ci] = params.size();
                params.insert(params.end(), den.begin(), den.end());
                while (params.size() % 4) params.push_back(0);
            }
        }
        DevBuf d_params;   // owned: read by every component's kernel while later components stage their own descriptors
        H_TRY(upload_owned(ctx, params.data(), params.size(), &d_params));
        for (uint32_t ci = 0; ci < n_comps; ci++) {
            const nx_component_spec& c = comps[ci];
            const uint32_t e = c.log_size + comp_log_cd(c.log_constraint_degree_bound, cs.cfg);
            std::vector<std::pair<uint32_t, uint32_t>> cc;
            for (uint32_t k = 0; k < c.n_pre; k++) cc.push_back({0u, (uint32_t)locs[ci].pre0 + k});
            for (uint32_t k = 0; k < c.n_main; k++) cc.push_back({1u, (uint32_t)locs[ci].main0 + k});
            for (uint32_t k = 0; k < c.n_inter; k++) cc.push_back({2u, (uint32_t)locs[ci].inter0 + k});
            std::vector<char> masked(cc.size(), 0);
            masked[c.n_pre] = masked[c.n_pre + 1] = 1;                 // main columns 0 and 1 are read at the next row
            EvalDomainCols cols;
            H_TRY(columns_on_eval_domain(cs, cc, c.log_size, e, masked, &cols));
            ColSet pre, mainc, inter;
            H_TRY(make_colset(ctx, cols.ptrs.data(), c.n_pre, &pre));
            H_TRY(make_colset(ctx, cols.ptrs.data() + c.n_pre, c.n_main, &mainc));
            H_TRY(make_colset(ctx, cols.ptrs.data() + c.n_pre + c.n_main, c.n_inter, &inter));
            SecureColumn* acc = nullptr;
            H_TRY(composition_accumulator(cs, sub, e, &acc));
            const uint64_t rb = cs.dist.on() ? cs.dist.begin(e) : 0;
            u32* a4[4]; for (int k = 0; k < 4; k++) a4[k] = bias_rows(acc->c[k], rb);
            SynthRange rg{0, c.n_main, c.n_main, 0, c.n_inter, true};
            H_TRY(synth_constraints(ctx, pre, mainc, inter, rg, (int)c.log_size, (int)e, d_params.p + off_pw[ci], d_params.p + off_den[ci], a4, (u32)rb, (u32)acc->rows));
        }
        return finalize_accumulation(cs, sub, out_polys, out_log);
    }
    void mask_points(QPt oods, MaskPoints* points) override {
        points->assign(3, {});
        for (uint32_t i = 0; i < n_comps; i++) {
            QPt step; { Pt s = pt_from_index(1u << (31 - comps[i].log_size)); step.x = q_from_m(s.x); step.y = q_from_m(s.y); }
            for (uint32_t k = 0; k < comps[i].n_pre; k++) (*points)[0].push_back({oods});
            for (uint32_t k = 0; k < comps[i].n_main; k++) { if (k < 2) (*points)[1].push_back({oods, qpt_add(oods, step)}); else (*points)[1].push_back({oods}); }
            for (uint32_t k = 0; k < comps[i].n_inter; k++) (*points)[2].push_back({oods});
        }
    }
    QM31 eval_composition_at_point(QPt point, const SampledValues& sv, QM31 rc) override { return synth_eval_composition_at_point(comps, n_comps, locs, point, sv, rc); }
};

static int check_components(nx_ctx* ctx, const nx_component_spec* comps, uint32_t n_comps, const nx_pcs_config* ucfg) {
    if (n_comps == 0) return set_err(ctx, NX_ERR_ARG, "prove: no components");
    if (ucfg->log_blowup < 1 || ucfg->log_constraint_degree < 1 || ucfg->log_constraint_degree > 2) return set_err(ctx, NX_ERR_ARG, "prove: log_blowup >= 1 and log_constraint_degree in {1,2} required");
    for (uint32_t i = 0; i < n_comps; i++) {
        if (comps[i].n_pre < 2 || comps[i].n_main < 2 || comps[i].log_size < 1 || comps[i].log_size > 28) return set_err(ctx, NX_ERR_ARG, "synthetic component needs n_pre >= 2, n_main >= 2, 1 <= log_size <= 28");
        if (!comp_log_cd_ok(comps[i].log_constraint_degree_bound, ucfg->log_constraint_degree))
            return set_err(ctx, NX_ERR_ARG, "prove: a component's log_constraint_degree_bound exceeds the config's log_constraint_degree (the twiddle tree is sized by it)");
    }
    return NX_OK;
}

// one synthetic tree: every GPU fills and hands over its share of each component's columns (all of them on one GPU)
static int fill_and_extend(CommitmentSchemeProver& cs, TreeBuilder& tb, const nx_component_spec* comps, uint32_t n_comps, uint32_t tree, uint64_t seed, uint64_t inter_seed) {
    nx_ctx* ctx = cs.ctx;
    std::vector<std::pair<uint32_t, uint32_t>> groups, local;
    for (uint32_t i = 0; i < n_comps; i++) groups.push_back({tree == 0 ? comps[i].n_pre : tree == 1 ? comps[i].n_main : comps[i].n_inter, comps[i].log_size});
    plan_local_columns(groups, cs.dist, &local);
    for (uint32_t i = 0; i < n_comps; i++) {
        const uint32_t lo = local[i].first, hi = local[i].second, log = comps[i].log_size;
        DevBuf slab;
        if (hi > lo) {
            H_TRY(slab.alloc(ctx, (size_t)(hi - lo) << log));
            auto p = col_ptrs(slab.p, hi - lo, log);
            H_TRY(synth_fill_range(ctx, comps[i], i, tree, seed, inter_seed, lo, hi - lo, p.data(), 0, 1u << log));
        }
        tb.extend_evals_local(std::move(slab), groups[i].first, log, lo, hi);
    }
    return NX_OK;
}

static int prove_synth(nx_ctx* ctx, const nx_component_spec* comps, uint32_t n_comps, const nx_pcs_config* ucfg, uint64_t seed, const uint8_t* ad,
                       size_t ad_len, const nx_comm* comm, std::vector<uint32_t>* words, nx_prove_stats* st) {
    PcsConfig cfg = {ucfg->pow_bits, ucfg->log_blowup, ucfg->n_queries, ucfg->log_last_layer_degree_bound, ucfg->fri_alpha_mode, ucfg->log_constraint_degree};
    H_TRY(check_components(ctx, comps, n_comps, ucfg));
    for (uint32_t i = 0; i < n_comps; i++) if (comps[i].logup_mode) return set_err(ctx, NX_ERR_ARG, "nx_prove_synth: logup_mode is nx_prove_machine's (this machine's interaction trace is a synthetic fill)");
    H_TRY(nx_ctx_set_hash_mode(ctx, (int)ucfg->hash_mode));
    uint32_t max_log = 0;
    std::vector<Loc> locs = locations(comps, n_comps, &max_log);
    const bool timed = st != nullptr;
    nx_prove_stats local_stats;
    if (!st) st = &local_stats;
    memset(st, 0, sizeof *st);
    if (timed) { ctx->timing = true; timing_reset(ctx); }
    double t_start = 0;
    Lap lap{ctx, timed, 0};
    if (timed) { (void)nx_sync(ctx); t_start = lap.t0 = now_ms(); }

    // machine.rs:184-194 — twiddles for CanonicCoset(max_log + LOG_CONSTRAINT_DEGREE + log_blowup).half_coset
    nx_twiddles* tw = nullptr;
    H_TRY(nx_twiddles_create(ctx, max_log + cfg.log_constraint_degree + cfg.log_blowup - 1, &tw));
    struct TwGuard { nx_twiddles* t; ~TwGuard() { nx_twiddles_destroy(t); } } twg{tw};
    Blake2sChannel channel;
    for (size_t i = 0; i < ad_len; i++) channel.mix_u64(ad[i]);                       // machine.rs:198-200
    CommitmentSchemeProver cs(ctx, tw, cfg);                                          // machine.rs:202-203
    if (comm) { H_TRY(dist_init(ctx, comm, &cs.dist)); cs.dist.comm_ms = &st->comm_ms; cs.dist.comm_bytes = &st->comm_bytes; cs.dist.comm_calls = &st->n_alltoallv; }
    for (uint32_t i = 0; i < n_comps; i++) channel.mix_u64(comps[i].log_size);        // machine.rs:204-206
    lap(&st->commit);

    auto fill_and_commit = [&](uint32_t tree, uint64_t inter_seed) -> int {
        TreeBuilder tb = cs.tree_builder();
        H_TRY(fill_and_extend(cs, tb, comps, n_comps, tree, seed, inter_seed));
        lap(&st->trace_gen);
        H_TRY(tb.commit(channel));
        lap(&st->commit);
        return NX_OK;
    };
    H_TRY(fill_and_commit(0, 0));                                                     // machine.rs:208-228
    H_TRY(fill_and_commit(1, 0));                                                     // machine.rs:230-237
    QM31 z = channel.draw_secure_felt();                                              // machine.rs:239-240 (lookup elements)
    uint64_t inter_seed = ((u64)z.a.a << 32) ^ (u64)z.a.b ^ ((u64)z.b.a << 16) ^ ((u64)z.b.b << 48);
    channel.mix_felts(std::vector<QM31>(n_comps, q_zero()));                          // machine.rs:262 (claimed sums)
    H_TRY(fill_and_commit(2, inter_seed));                                            // machine.rs:249-263

    SynthAir air(comps, n_comps, locs);
    H_TRY(prove_core(ctx, cs, channel, cfg, tw, air, words, st, lap));
    if (timed) finish_stats(ctx, st, t_start);
    return NX_OK;
}

// ================================================================ nx_prove_machine: real logup + recorded AIR ===========
// Component (log_size, n_pre, n_main, n_inter = 4 L, bound, logup_mode): L logup columns holding F fractions — F = L (one per column),
// 2 L (NX_LOGUP_PAIRS) or 2 L - 1 (PAIRS | ODD).  Fraction f of a row:
//   den_f = t[a_f] - z                       (f even: a one-element tuple, like the reference's check_bytes limb, range256.rs:281-284)
//         = t[a_f] + alpha t[b_f] - z        (f odd: a two-element tuple)
//   num_f = 1,  or  -main[m_f] when f % 3 == 2  (the table side of a lookup: a negated multiplicity column)
//   t = the MAIN trace, a_f = (3 + 7 f) % n_main, b_f = (5 + 11 f) % n_main, m_f = (2 + 13 f) % n_main
//   NX_LOGUP_TABLE: t = the PREPROCESSED trace (get_preprocessed_column, reference extensions/multiplicity.rs:111-124),
//   a_f = (2 + 3 f) % n_pre, b_f = (1 + 5 f) % n_pre, and EVERY numerator is -main[m_f]
// Logup column j holds the sum of the fractions of batches <= j (LogupColGenerator::finalize_col; a batch is one fraction, or the pair
// (2 j, 2 j + 1) merged as (a d + b c) / (b d) by LogupTraceBuilder::add_to_relation_with, prover2/machine/src/lookups/
// logup_trace_builder.rs:86-101); the last one is finalised by LogupTraceGenerator::finalize_last (claimed sum; prefix sum over the rows
// of value - claimed / N).
// Constraints, in declaration order: the two transition constraints and the degree-2 constraints of the synthetic main trace (as
// nx_prove_synth), then per logup column stwo-constraint-framework's finalize_logup_batched — with (N_j, D_j) the batch's fraction
// num / den, or its pair sum (n0 d1 + n1 d0) / (d0 d1) (Fraction::add; finalize_logup_in_pairs, reference
// extensions/keccak/round/constraints.rs:116: degree 3 under the bound +1):
//   (S_j - S_{j-1}) D_j - N_j                                             j < L - 1
//   (S_j(row) - S_j(row - 1) - S_{j-1}(row) + claimed / N) D_j - N_j      j = L - 1 (mask [-1, 0] on the last column)
// Tuple schedules (NX_LOGUP_TUPLES(k), bits 4..7 of logup_mode; k = 0: the one- / two-element tuples above).  The reference's relations
// are wider and their entries are expressions: NX_TUPLES_V1 — widths cycling 1, 1, 4, 1, 9, 1, 3, 1 (range checks: chips/range_check/
// range256.rs:37; bit_op.rs:31,355 [op_type CONSTANT, b, c, a] with the flag column as numerator; register_mem_check.rs:34; the 3-wide
// one carries a constant and a SUM of two columns); NX_TUPLES_KECCAK — 3- / 4-wide bitwise lookups (chips/custom.rs:33-37) and, as the
// last two fractions, the 200-wide state lookups with the numerators (is_padding - 1) and (1 - is_padding) (custom.rs:45-46,
// extensions/keccak/round/constraints.rs:101-110); NX_TUPLES_V2 — prover2's relations, 9 / 21 / 14 / 10 / 4 / 12 / 8 wide
// (prover2/machine/src/lookups/relations.rs:33-90).  Entry k of fraction f reads column (3 + 7 f + 5 k) % n_main (a table: (2 + 3 f + 5 k)
// % n_pre); the second column of a sum is (5 + 11 f + 3 k) % n.
enum { E_COL = 0, E_CONST = 1, E_SUM = 2 };
enum { N_ONE = 0, N_NEG_MULT = 1, N_MULT_M1 = 2, N_ONE_M_MULT = 3, N_MULT = 4 };   // numerator: 1, -m, m - 1, 1 - m, m  (m = main[mult])
struct TupleEntry { uint32_t kind, a, b; };
struct FracDef {
    bool table; uint32_t w, col[2]; bool has_mult; uint32_t mult;     // (col[]: the tuple of schedule 0)
    uint32_t num = N_ONE; std::vector<TupleEntry> ent;                // every schedule
};
static uint32_t tuple_sched(const nx_component_spec& c) { return (c.logup_mode >> 4) & 15u; }
static bool logup_mode_ok(uint32_t m) { return (m >> 4) <= NX_TUPLES_V2 && !((m & NX_LOGUP_ODD) && !(m & NX_LOGUP_PAIRS)) && !(m & 8u); }
static uint32_t n_logup_fracs(const nx_component_spec& c) {
    const uint32_t L = c.n_inter / 4;
    if (!(c.logup_mode & NX_LOGUP_PAIRS) || L == 0) return L;
    return 2 * L - ((c.logup_mode & NX_LOGUP_ODD) ? 1 : 0);
}
static FracDef frac_def(const nx_component_spec& c, uint32_t f) {
    FracDef d; d.table = (c.logup_mode & NX_LOGUP_TABLE) != 0;
    const uint32_t sched = tuple_sched(c), n = d.table ? c.n_pre : c.n_main;
    d.mult = (2 + 13 * f) % c.n_main;
    if (sched == 0) {
        d.w = (f & 1) ? 2 : 1;
        if (d.table) { d.col[0] = (2 + 3 * f) % c.n_pre; d.col[1] = (1 + 5 * f) % c.n_pre; d.has_mult = true; }
        else { d.col[0] = (3 + 7 * f) % c.n_main; d.col[1] = (5 + 11 * f) % c.n_main; d.has_mult = f % 3 == 2; }
        d.num = d.has_mult ? N_NEG_MULT : N_ONE;
        for (uint32_t k = 0; k < d.w; k++) d.ent.push_back({E_COL, d.col[k], 0});
        return d;
    }
    const uint32_t F = n_logup_fracs(c);
    static const uint32_t W1[8] = {1, 1, 4, 1, 9, 1, 3, 1}, W3[7] = {9, 21, 14, 10, 4, 12, 8};
    const bool state = sched == NX_TUPLES_KECCAK && !d.table && F >= 2 && f + 2 >= F;
    d.w = sched == NX_TUPLES_V1 ? W1[f % 8] : sched == NX_TUPLES_V2 ? W3[f % 7] : state ? 200 : (f % 4 == 3 ? 4 : 3);
    for (uint32_t k = 0; k < d.w; k++) d.ent.push_back({E_COL, ((d.table ? 2 + 3 * f : 3 + 7 * f) + 5 * k) % n, 0});
    auto sum = [&](uint32_t k) { d.ent[k].kind = E_SUM; d.ent[k].b = (5 + 11 * f + 3 * k) % n; };
    if (sched == NX_TUPLES_V1 && d.w == 4) d.ent[0] = {E_CONST, 1 + f % 3, 0};
    if (sched == NX_TUPLES_V1 && d.w == 3) { d.ent[1] = {E_CONST, 5, 0}; sum(2); }
    if (sched == NX_TUPLES_V2 && d.w == 4) { d.ent[1] = {E_CONST, 7, 0}; sum(2); }
    d.num = d.table ? N_NEG_MULT : f % 3 == 2 ? N_NEG_MULT : N_ONE;
    if (!d.table) {
        if (sched == NX_TUPLES_V1 && f % 8 == 2) d.num = N_MULT;
        if (sched == NX_TUPLES_V2 && f % 5 == 1 && f % 3 != 2) d.num = N_MULT;
        if (state) d.num = f + 2 == F ? N_MULT_M1 : N_ONE_M_MULT;
    }
    d.has_mult = d.num != N_ONE; d.col[0] = d.col[1] = 0;
    return d;
}
static uint32_t max_tuple_width(const nx_component_spec& c) {
    uint32_t w = 2;
    if (tuple_sched(c)) for (uint32_t f = 0; f < n_logup_fracs(c); f++) w = std::max(w, frac_def(c, f).w);
    return w;
}
// the fractions of logup column j
static std::vector<uint32_t> batch_of(const nx_component_spec& c, uint32_t j) {
    if (!(c.logup_mode & NX_LOGUP_PAIRS)) return {j};
    std::vector<uint32_t> b{2 * j};
    if (2 * j + 1 < n_logup_fracs(c)) b.push_back(2 * j + 1);
    return b;
}

struct ProgEmit {
    std::vector<nx_cinstr> p;
    void op(uint32_t o, uint32_t d, uint32_t a = 0, uint32_t b = 0) { p.push_back(nx_cinstr{o, d, a, b}); }
};

// ---- the wide-tuple forms (tuple schedule != 0) -------------------------------------------------------------------------------------
// econsts of such a component: [z, alpha, claimed / N, 0, alpha^0, alpha^1, ..., alpha^(W - 1)] (W = its widest tuple) — LookupElements<N>
// holds its alpha powers as constants (stwo-constraint-framework relation!).  Registers: B 14..21 a group of 8 tuple values, 22..29 the second
// columns of sums, 30 / 31 the (negated) numerators, 32 the constant 1; E quads from 36.
enum { GV = 14, GW = 22, GN0 = 30, GN1 = 31, GONE = 32, GE = 36, GZ = GE, GNZ = GE + 4, GSH = GE + 8, GA = GE + 12, GTMP = GE + 16, GD0 = GE + 20, GD1 = GE + 24,
       GDD = GE + 28, GTT = GE + 32, GPN = GE + 36, GPR = GE + 40, GS0 = GE + 44, GS1 = GE + 48, GREGS = GE + 52, APOW0 = 4 };
// dst (an E quad) = sum_k alpha^k entry_k - z   (Relation::combine): the loads of 8 entries, then their arithmetic
static void emit_den(ProgEmit& e, const FracDef& d, uint32_t dst, uint32_t tuple_base) {
    for (uint32_t k0 = 0; k0 < d.w; k0 += 8) {
        const uint32_t k1 = std::min(d.w, k0 + 8);
        for (uint32_t k = k0; k < k1; k++) {
            const TupleEntry& t = d.ent[k];
            if (t.kind == E_CONST) e.op(NX_C_CONST, GV + (k - k0), t.a);
            else e.op(NX_C_LOAD, GV + (k - k0), tuple_base + t.a, 0);
            if (t.kind == E_SUM) e.op(NX_C_LOAD, GW + (k - k0), tuple_base + t.b, 0);
        }
        for (uint32_t k = k0; k < k1; k++) if (d.ent[k].kind == E_SUM) e.op(NX_C_ADD, GV + (k - k0), GV + (k - k0), GW + (k - k0));
        for (uint32_t k = k0; k < k1; k++) {
            if (k == 0) e.op(NX_C_ADDEB, dst, GNZ, GV);                                                   // alpha^0 entry_0 - z
            else { e.op(NX_C_CONSTE, GA, APOW0 + k); e.op(NX_C_MULEB, GTMP, GA, GV + (k - k0)); e.op(NX_C_ADDE, dst, dst, GTMP); }
        }
    }
}
// B[dst] = the numerator (neg: its negation, what the constraint adds)
static void emit_num(ProgEmit& e, const FracDef& d, uint32_t dst, uint32_t main_base, bool neg) {
    enum { T0 = 0 };
    switch (d.num) {
    case N_ONE: e.op(NX_C_CONST, dst, neg ? P - 1 : 1); break;
    case N_NEG_MULT: if (neg) e.op(NX_C_LOAD, dst, main_base + d.mult, 0); else { e.op(NX_C_LOAD, T0, main_base + d.mult, 0); e.op(NX_C_NEG, dst, T0); } break;
    case N_MULT: if (!neg) e.op(NX_C_LOAD, dst, main_base + d.mult, 0); else { e.op(NX_C_LOAD, T0, main_base + d.mult, 0); e.op(NX_C_NEG, dst, T0); } break;
    case N_MULT_M1: e.op(NX_C_LOAD, T0, main_base + d.mult, 0); if (neg) e.op(NX_C_SUB, dst, GONE, T0); else e.op(NX_C_SUB, dst, T0, GONE); break;
    default: e.op(NX_C_LOAD, T0, main_base + d.mult, 0); if (neg) e.op(NX_C_SUB, dst, T0, GONE); else e.op(NX_C_SUB, dst, GONE, T0); break;   // 1 - m
    }
}
// finalize_logup / finalize_logup_in_pairs over fractions of any width (see the constraint list above)
static void emit_wide_logup_constraints(ProgEmit& e, const nx_component_spec& c, uint32_t* nc) {
    const uint32_t L = c.n_inter / 4, MAIN = c.n_pre, INT = c.n_pre + c.n_main, TB = (c.logup_mode & NX_LOGUP_TABLE) ? 0u : MAIN;
    e.op(NX_C_CONSTE, GZ, 0); e.op(NX_C_CONSTE, GSH, 2); e.op(NX_C_CONSTE, GNZ, 3); e.op(NX_C_SUBE, GNZ, GNZ, GZ); e.op(NX_C_CONST, GONE, 1);
    for (uint32_t j = 0; j < L; j++) {
        const std::vector<uint32_t> fs = batch_of(c, j);
        const bool two = fs.size() == 2;
        for (size_t i = 0; i < fs.size(); i++) { const FracDef d = frac_def(c, fs[i]); emit_den(e, d, i ? GD1 : GD0, TB); emit_num(e, d, i ? GN1 : GN0, MAIN, true); }
        const uint32_t cur = (j & 1) ? GS1 : GS0, prev = (j & 1) ? GS0 : GS1;
        e.op(NX_C_LOADE, cur, INT + 4 * j, 0);
        uint32_t diff = GTT;
        if (j + 1 < L) { if (j == 0) diff = cur; else e.op(NX_C_SUBE, GTT, cur, prev); }
        else {
            e.op(NX_C_LOADE, GPR, INT + 4 * j, (uint32_t)-1);
            e.op(NX_C_SUBE, GTT, cur, GPR);
            if (j > 0) e.op(NX_C_SUBE, GTT, GTT, prev);
            e.op(NX_C_ADDE, GTT, GTT, GSH);
        }
        if (two) {
            e.op(NX_C_MULE, GDD, GD0, GD1); e.op(NX_C_MULE, GTT, diff, GDD);                             // diff d0 d1 - (n0 d1 + n1 d0)
            e.op(NX_C_MULEB, GPN, GD1, GN0); e.op(NX_C_ADDE, GTT, GTT, GPN);
            e.op(NX_C_MULEB, GPN, GD0, GN1); e.op(NX_C_ADDE, GTT, GTT, GPN);
        } else { e.op(NX_C_MULE, GTT, diff, GD0); e.op(NX_C_ADDEB, GTT, GTT, GN0); }
        e.op(NX_C_CONSTRAINT_E, 0, GTT); (*nc)++;
    }
}
// The component's relation entries as a fraction program (nx_logup_program): what a recording EvalAtRow hands over for
// eval.add_to_relation(RelationEntry::new(relation, multiplicity, &values)) — one NX_C_FRACB per entry, batch = its logup column.
// Columns are numbered like the component's AIR (preprocessed, main, interaction); econsts as above.
static std::vector<nx_cinstr> machine_fraction_program(const nx_component_spec& c, uint32_t* n_regs) {
    const uint32_t L = c.n_inter / 4, MAIN = c.n_pre, TB = (c.logup_mode & NX_LOGUP_TABLE) ? 0u : MAIN;
    ProgEmit e;
    e.op(NX_C_CONSTE, GZ, 0); e.op(NX_C_CONSTE, GNZ, 3); e.op(NX_C_SUBE, GNZ, GNZ, GZ); e.op(NX_C_CONST, GONE, 1);
    for (uint32_t j = 0; j < L; j++)
        for (uint32_t f : batch_of(c, j)) {
            FracDef d = frac_def(c, f);
            emit_den(e, d, GD0, TB); emit_num(e, d, GN0, MAIN, false);
            e.op(NX_C_FRACB, j, GN0, GD0);
        }
    *n_regs = GREGS;
    return std::move(e.p);
}

// econsts of a component: [z, alpha, claimed / N, 0]
// A component whose constraint-degree bound is 2 HAS constraints that need it (degree 4 or 5; v1's shift chips, reference
// prover/src/chips/instructions/i/sra.rs:267-307): every degree-2 main-trace constraint of such a component is multiplied by the two
// columns it squares, and each of the two transition constraints by main0 main1 — degree 4, satisfied by the same trace — so that all its
// main columns are needed on the 4x domain and two of them at a neighbour row there.  The logup constraints keep degree 2 (one fraction per
// column, the reference's finalize_logup, components/mod.rs:53) or 3 (pairs).
static GComponent machine_component(const nx_component_spec& c, const Loc& loc, const PcsConfig& cfg) {
    GComponent g;
    g.log_size = c.log_size;
    const bool quartic = comp_log_cd(c.log_constraint_degree_bound, cfg) >= 2;
    const uint32_t L = c.n_inter / 4, PRE = 0, MAIN = c.n_pre, INT = c.n_pre + c.n_main;
    for (uint32_t k = 0; k < c.n_pre; k++) { g.cols.push_back({0u, (uint32_t)loc.pre0 + k}); g.masks.push_back({0}); }
    for (uint32_t k = 0; k < c.n_main; k++) { g.cols.push_back({1u, (uint32_t)loc.main0 + k}); g.masks.push_back(k < 2 ? std::vector<int>{0, 1} : std::vector<int>{0}); }
    for (uint32_t k = 0; k < c.n_inter; k++) { g.cols.push_back({2u, (uint32_t)loc.inter0 + k}); g.masks.push_back(k / 4 + 1 == L ? std::vector<int>{-1, 0} : std::vector<int>{0}); }
    // Emission order: the column loads of every chunk of constraints come first, then the arithmetic — a run of independent loads
    // is what both the interpreter (it issues a run together) and the compiler (all requests of a chunk in flight before the first
    // use) turn into memory-level parallelism.  Registers: B 0..3 scratch, 4..13 a ring of main-column values (a chunk of 8 plus
    // the two predecessors the degree-2 constraints read), 14..25 tuple / multiplicity values of a logup chunk; E quads after that.
    enum { T0 = 0, T1 = 1, T2 = 2, T3 = 3, RING = 4, NRING = 10, TUP = 14, EBASE = 28,
           EZ = EBASE, EAL = EBASE + 4, ESH = EBASE + 8, ENZ = EBASE + 12, EDEN = EBASE + 16, ET = EBASE + 20, EPR = EBASE + 24, ES = EBASE + 28 /* 5 quads: S_{j-1} and a chunk of 4 */, NREGS = EBASE + 48 };
    g.n_regs = NREGS;
    ProgEmit e;
    uint32_t nc = 0;
    auto ring = [&](uint32_t k) { return (uint32_t)RING + k % NRING; };
    // (main0' - main0 - 1)(1 - is_last), (main1' - main1 - main0)(1 - is_last)
    e.op(NX_C_LOAD, ring(0), MAIN + 0, 0); e.op(NX_C_LOAD, ring(1), MAIN + 1, 0);
    e.op(NX_C_LOAD, T0, MAIN + 0, 1); e.op(NX_C_LOAD, T1, MAIN + 1, 1); e.op(NX_C_LOAD, T3, PRE + 1, 0);
    e.op(NX_C_CONST, T2, 1); e.op(NX_C_SUB, T3, T2, T3);
    // quartic: the two transition constraints times main0 main1 too — degree 4 AND a neighbour row, like the reference's constraints over
    // its masked columns Pc / IsPadding (prover/src/column.rs:13-20, trace/eval.rs:22-50): what the quarter domain's neighbour rows are for
    e.op(NX_C_SUB, T0, T0, ring(0)); e.op(NX_C_SUB, T0, T0, T2); e.op(NX_C_MUL, T0, T0, T3);
    if (quartic) { e.op(NX_C_MUL, T0, T0, ring(0)); e.op(NX_C_MUL, T0, T0, ring(1)); }
    e.op(NX_C_CONSTRAINT_B, 0, T0); nc++;
    e.op(NX_C_SUB, T1, T1, ring(1)); e.op(NX_C_SUB, T1, T1, ring(0)); e.op(NX_C_MUL, T1, T1, T3);
    if (quartic) { e.op(NX_C_MUL, T1, T1, ring(0)); e.op(NX_C_MUL, T1, T1, ring(1)); }
    e.op(NX_C_CONSTRAINT_B, 0, T1); nc++;
    for (uint32_t k0 = 2; k0 < c.n_main; k0 += 8) {
        const uint32_t k1 = std::min(c.n_main, k0 + 8);
        for (uint32_t k = k0; k < k1; k++) e.op(NX_C_LOAD, ring(k), MAIN + k, 0);
        for (uint32_t k = k0; k < k1; k++)
            if (!synth_col_is_free(k)) {
                e.op(NX_C_MUL, T0, ring(k - 1), ring(k - 1)); e.op(NX_C_MUL, T1, ring(k - 2), ring(k - 2)); e.op(NX_C_SUB, T2, ring(k), T0); e.op(NX_C_SUB, T2, T2, T1);
                if (quartic) { e.op(NX_C_MUL, T2, T2, ring(k - 1)); e.op(NX_C_MUL, T2, T2, ring(k - 2)); }
                e.op(NX_C_CONSTRAINT_B, 0, T2); nc++;
            }
    }
    const uint32_t TB = (c.logup_mode & NX_LOGUP_TABLE) ? (uint32_t)PRE : (uint32_t)MAIN;      // where the tuples are read
    if (L && tuple_sched(c)) {
        g.n_regs = std::max<uint32_t>(g.n_regs, GREGS);
        emit_wide_logup_constraints(e, c, &nc);
    } else if (L && !(c.logup_mode & NX_LOGUP_PAIRS)) {
        e.op(NX_C_CONSTE, EZ, 0); e.op(NX_C_CONSTE, EAL, 1); e.op(NX_C_CONSTE, ESH, 2); e.op(NX_C_CONSTE, ENZ, 3); e.op(NX_C_SUBE, ENZ, ENZ, EZ);   // -z
        auto S = [&](uint32_t j) { return (uint32_t)ES + 4 * (j % 5); };       // S_j lives in slot j % 5: a chunk of 4 never overwrites S_{j0 - 1}
        for (uint32_t j0 = 0; j0 < L; j0 += 4) {
            const uint32_t j1 = std::min(L, j0 + 4);
            for (uint32_t j = j0; j < j1; j++) {                               // the chunk's loads
                const FracDef d = frac_def(c, j);
                const uint32_t t = TUP + 3 * (j - j0);
                e.op(NX_C_LOAD, t, TB + d.col[0], 0);
                if (d.w == 2) e.op(NX_C_LOAD, t + 1, TB + d.col[1], 0);
                if (d.has_mult) e.op(NX_C_LOAD, t + 2, MAIN + d.mult, 0);
                e.op(NX_C_LOADE, S(j), INT + 4 * j, 0);
                if (j + 1 == L) e.op(NX_C_LOADE, EPR, INT + 4 * j, (uint32_t)-1);
            }
            for (uint32_t j = j0; j < j1; j++) {
                const FracDef d = frac_def(c, j);
                const uint32_t t = TUP + 3 * (j - j0), cur = S(j), prev = S(j + 4);   // (j - 1) % 5 == (j + 4) % 5
                if (d.w == 2) { e.op(NX_C_MULEB, EDEN, EAL, t + 1); e.op(NX_C_ADDEB, EDEN, EDEN, t); e.op(NX_C_ADDE, EDEN, EDEN, ENZ); }
                else e.op(NX_C_ADDEB, EDEN, ENZ, t);
                if (j + 1 < L) {
                    if (j == 0) e.op(NX_C_MULE, ET, cur, EDEN);
                    else { e.op(NX_C_SUBE, ET, cur, prev); e.op(NX_C_MULE, ET, ET, EDEN); }
                } else {
                    e.op(NX_C_SUBE, ET, cur, EPR);
                    if (j > 0) e.op(NX_C_SUBE, ET, ET, prev);
                    e.op(NX_C_ADDE, ET, ET, ESH);
                    e.op(NX_C_MULE, ET, ET, EDEN);
                }
                if (d.has_mult) e.op(NX_C_ADDEB, ET, ET, t + 2);              // - num = + main[m]
                else { e.op(NX_C_CONST, T2, P - 1); e.op(NX_C_ADDEB, ET, ET, T2); }   // - num = - 1
                e.op(NX_C_CONSTRAINT_E, 0, ET); nc++;
            }
        }
    } else if (L) {
        // finalize_logup_in_pairs: column j sums fractions 2 j and 2 j + 1.  Registers of this form: B 14..37 the tuple / multiplicity
        // values of a chunk of 4 columns (8 fractions x 3), E quads from 40.
        enum { PT = 14, PE = 40, PZ = PE, PAL = PE + 4, PSH = PE + 8, PNZ = PE + 12, PD0 = PE + 16, PD1 = PE + 20, PDD = PE + 24, PTT = PE + 28, PN = PE + 32, PPR = PE + 36,
               PS = PE + 40 /* 5 quads */, PREGS = PE + 60 };
        g.n_regs = PREGS;
        const uint32_t F = n_logup_fracs(c);
        e.op(NX_C_CONSTE, PZ, 0); e.op(NX_C_CONSTE, PAL, 1); e.op(NX_C_CONSTE, PSH, 2); e.op(NX_C_CONSTE, PNZ, 3); e.op(NX_C_SUBE, PNZ, PNZ, PZ);   // -z
        auto S = [&](uint32_t j) { return (uint32_t)PS + 4 * (j % 5); };
        for (uint32_t j0 = 0; j0 < L; j0 += 4) {
            const uint32_t j1 = std::min(L, j0 + 4);
            for (uint32_t j = j0; j < j1; j++) {
                for (uint32_t f = 2 * j; f < std::min(F, 2 * j + 2); f++) {
                    const FracDef d = frac_def(c, f);
                    const uint32_t t = PT + 3 * (f - 2 * j0);
                    e.op(NX_C_LOAD, t, TB + d.col[0], 0);
                    if (d.w == 2) e.op(NX_C_LOAD, t + 1, TB + d.col[1], 0);
                    if (d.has_mult) e.op(NX_C_LOAD, t + 2, MAIN + d.mult, 0);
                }
                e.op(NX_C_LOADE, S(j), INT + 4 * j, 0);
                if (j + 1 == L) e.op(NX_C_LOADE, PPR, INT + 4 * j, (uint32_t)-1);
            }
            for (uint32_t j = j0; j < j1; j++) {
                const uint32_t cur = S(j), prev = S(j + 4);
                const bool two = 2 * j + 1 < F;
                const FracDef d0 = frac_def(c, 2 * j), d1 = frac_def(c, two ? 2 * j + 1 : 2 * j);
                const uint32_t t0 = PT + 3 * (2 * j - 2 * j0), t1 = t0 + 3;
                auto den = [&](uint32_t dst, const FracDef& d, uint32_t t) {
                    if (d.w == 2) { e.op(NX_C_MULEB, dst, PAL, t + 1); e.op(NX_C_ADDEB, dst, dst, t); e.op(NX_C_ADDE, dst, dst, PNZ); }
                    else e.op(NX_C_ADDEB, dst, PNZ, t);
                };
                den(PD0, d0, t0);
                if (two) den(PD1, d1, t1);
                uint32_t diff = PTT;
                if (j + 1 < L) {
                    if (j == 0) diff = cur;                                    // S_0 - 0
                    else e.op(NX_C_SUBE, PTT, cur, prev);
                } else {
                    e.op(NX_C_SUBE, PTT, cur, PPR);
                    if (j > 0) e.op(NX_C_SUBE, PTT, PTT, prev);
                    e.op(NX_C_ADDE, PTT, PTT, PSH);
                }
                if (two) {
                    e.op(NX_C_MULE, PDD, PD0, PD1); e.op(NX_C_MULE, PTT, diff, PDD);            // diff d0 d1
                    // - (n0 d1 + n1 d0)
                    if (d0.has_mult) { e.op(NX_C_MULEB, PN, PD1, t0 + 2); e.op(NX_C_ADDE, PTT, PTT, PN); } else e.op(NX_C_SUBE, PTT, PTT, PD1);
                    if (d1.has_mult) { e.op(NX_C_MULEB, PN, PD0, t1 + 2); e.op(NX_C_ADDE, PTT, PTT, PN); } else e.op(NX_C_SUBE, PTT, PTT, PD0);
                } else {
                    e.op(NX_C_MULE, PTT, diff, PD0);
                    if (d0.has_mult) e.op(NX_C_ADDEB, PTT, PTT, t0 + 2);
                    else { e.op(NX_C_CONST, T2, P - 1); e.op(NX_C_ADDEB, PTT, PTT, T2); }
                }
                e.op(NX_C_CONSTRAINT_E, 0, PTT); nc++;
            }
        }
    }
    g.prog = std::move(e.p);
    g.n_constraints = nc;
    g.econsts.assign(tuple_sched(c) ? 4 * (size_t)(APOW0 + max_tuple_width(c)) : 16, 0);
    return g;
}

// The logup interaction trace of one component from the trace columns its fractions read — `mainv` / `prev`: main / preprocessed column
// index -> evaluations (bit-reversed circle-domain order; whole columns, or this GPU's row block of n_rows = 2^log_rows rows): inter =
// 4 L coordinate columns of n_rows words.
typedef std::map<uint32_t, const uint32_t*> ColMap;
// [z, alpha, shift, 0, alpha^0 .. alpha^(W-1)]: the E constants of a wide-tuple component (schedule 0: the first four)
static void fill_econsts(std::vector<uint32_t>& ec, const uint32_t z[4], const uint32_t alpha[4], const QM31& shift) {
    memcpy(&ec[0], z, 16); memcpy(&ec[4], alpha, 16); q_store(&ec[8], shift);
    QM31 pw = q_one(); const QM31 al = q_load(alpha);
    for (size_t k = APOW0; 4 * k + 4 <= ec.size(); k++) { q_store(&ec[4 * k], pw); pw = q_mul(pw, al); }
}
static int logup_columns(nx_ctx* ctx, const nx_component_spec& c, uint32_t log_rows, const ColMap& mainv, const ColMap& prev, const uint32_t z[4], const uint32_t alpha[4],
                         uint32_t* const* inter) {
    const uint32_t L = c.n_inter / 4, F = n_logup_fracs(c);
    const bool pairs = (c.logup_mode & NX_LOGUP_PAIRS) != 0;
    static const uint32_t one[4] = {1, 0, 0, 0}, minus_one[4] = {P - 1, 0, 0, 0};
    const bool per_column = ctx->opt.logup_per_column != 0;   // A/B: one nx_logup_col launch per column
    std::vector<FracDef> defs(F);
    bool affine = false;                                       // a numerator (m - 1) / (1 - m): not a scale times a column
    for (uint32_t f = 0; f < F; f++) { defs[f] = frac_def(c, f); affine = affine || defs[f].num == N_MULT_M1 || defs[f].num == N_ONE_M_MULT; }
    if (tuple_sched(c) && (affine || ctx->opt.machine_logup_program)) {
        // The route of a Rust-side prove without the chips' generators (reference_patch/machine_hip.rs): the interaction trace FROM THE
        // RECORDED relation entries (nx_logup_program) — any expression as tuple entry or numerator.
        uint32_t n_regs = 0;
        HostSpan hs_p("pm.machine_fraction_program");
        const std::vector<nx_cinstr> prog = machine_fraction_program(c, &n_regs);
        hs_p.stop();
        const uint32_t n_cols = c.n_pre + c.n_main + c.n_inter;
        std::vector<const uint32_t*> cols(n_cols, nullptr);
        for (auto& kv : prev) cols[kv.first] = kv.second;
        for (auto& kv : mainv) cols[c.n_pre + kv.first] = kv.second;
        std::vector<uint32_t> ec(4 * (size_t)(APOW0 + max_tuple_width(c)), 0);
        fill_econsts(ec, z, alpha, q_zero());
        return nx_logup_program(ctx, prog.data(), (uint32_t)prog.size(), n_regs, cols.data(), n_cols, ec.data(), (uint32_t)(ec.size() / 4), log_rows, L, inter, nullptr);
    }
    // nx_logup_frac takes columns: an entry that is a constant moves into z (z' = z - sum_k alpha^k const_k), a sum of two columns becomes
    // two tuple columns under the same alpha power — the same field element, by linearity of Relation::combine
    std::vector<QM31> apow;
    { QM31 pw = q_one(); const QM31 al = q_load(alpha); for (uint32_t k = 0; k < max_tuple_width(c); k++) { apow.push_back(pw); pw = q_mul(pw, al); } }
    std::vector<std::vector<const uint32_t*>> tuples(F);
    std::vector<std::vector<uint32_t>> aps(F);
    std::vector<std::array<uint32_t, 4>> zs(F);
    std::vector<nx_logup_frac> fr(F);
    for (uint32_t f = 0; f < F; f++) {
        const FracDef& d = defs[f];
        const ColMap& src = d.table ? prev : mainv;
        QM31 zf = q_load(z);
        auto push = [&](uint32_t col, uint32_t k) { tuples[f].push_back(src.at(col)); aps[f].resize(aps[f].size() + 4); q_store(&aps[f][aps[f].size() - 4], apow[k]); };
        for (uint32_t k = 0; k < d.w; k++) {
            const TupleEntry& t = d.ent[k];
            if (t.kind == E_CONST) zf = q_sub(zf, q_mul_m(apow[k], t.a));
            else { push(t.a, k); if (t.kind == E_SUM) push(t.b, k); }
        }
        q_store(zs[f].data(), zf);
        nx_logup_frac& q = fr[f];
        q.d_tuple_cols = tuples[f].data(); q.n_tuple_cols = (uint32_t)tuples[f].size(); q.alpha_powers = aps[f].data(); q.z = zs[f].data();
        q.d_mult = d.has_mult ? mainv.at(d.mult) : nullptr; q.scale = d.num == N_NEG_MULT ? minus_one : one;
    }
    if (per_column) {
        for (uint32_t j = 0; j < L; j++) {
            const uint32_t f0 = pairs ? 2 * j : j;
            const nx_logup_frac* fb = pairs && f0 + 1 < F ? &fr[f0 + 1] : nullptr;
            H_TRY(nx_logup_col(ctx, log_rows, &fr[f0], fb, j ? (const uint32_t* const*)(inter + 4 * (j - 1)) : nullptr, inter + 4 * j));
        }
        return NX_OK;
    }
    if (pairs) return nx_logup_cols_batched(ctx, log_rows, fr.data(), F, nullptr, L, inter);
    return nx_logup_cols(ctx, log_rows, fr.data(), L, inter);
}
// the columns of tree 0 (preprocessed) / 1 (main) the component's fractions read: their evaluations are needed after the commit
static std::set<uint32_t> logup_needed_columns(const nx_component_spec& c, uint32_t tree) {
    std::set<uint32_t> s;
    for (uint32_t f = 0; f < n_logup_fracs(c); f++) {
        const FracDef d = frac_def(c, f);
        if ((d.table ? 0u : 1u) == tree)
            for (const TupleEntry& t : d.ent) { if (t.kind != E_CONST) s.insert(t.a); if (t.kind == E_SUM) s.insert(t.b); }
        if (tree == 1 && d.has_mult) s.insert(d.mult);
    }
    return s;
}

// host: the preprocessed and the main trace handed over in HOST memory (component after component, column after column; NULL = generate
// them on the device from `seed`): the commits then upload them chunk by chunk under their own transforms (TreeBuilder::extend_evals_host)
struct HostTrace { const uint32_t* const* pre = nullptr; const uint32_t* const* main = nullptr; int coset_order = 0; };
static int prove_machine(nx_ctx* ctx, const nx_component_spec* comps, uint32_t n_comps, const nx_pcs_config* ucfg, uint64_t seed, const uint8_t* ad,
                         size_t ad_len, const nx_comm* comm, std::vector<uint32_t>* words, nx_prove_stats* st, const HostTrace* host = nullptr) {
    PcsConfig cfg = {ucfg->pow_bits, ucfg->log_blowup, ucfg->n_queries, ucfg->log_last_layer_degree_bound, ucfg->fri_alpha_mode, ucfg->log_constraint_degree};
    HostSpan hs_whole("pm.body (to the return statement)");
    HostSpan hs_pro("pm.prologue");
    H_TRY(check_components(ctx, comps, n_comps, ucfg));
    for (uint32_t i = 0; i < n_comps; i++) {
        if (comps[i].n_inter % 4) return set_err(ctx, NX_ERR_ARG, "nx_prove_machine: n_inter = 4 x (number of logup columns)");
        const uint32_t m = comps[i].logup_mode;
        if (!logup_mode_ok(m))
            return set_err(ctx, NX_ERR_ARG, "nx_prove_machine: logup_mode is a set of NX_LOGUP_PAIRS / NX_LOGUP_ODD (with PAIRS only) / NX_LOGUP_TABLE, plus NX_LOGUP_TUPLES(0 .. 3)");
    }
    H_TRY(nx_ctx_set_hash_mode(ctx, (int)ucfg->hash_mode));
    uint32_t max_log = 0;
    std::vector<Loc> locs = locations(comps, n_comps, &max_log);
    const bool timed = st != nullptr;
    nx_prove_stats local_stats;
    if (!st) st = &local_stats;
    memset(st, 0, sizeof *st);
    if (timed) { ctx->timing = true; timing_reset(ctx); }
    double t_start = 0;
    Lap lap{ctx, timed, 0};
    if (timed) { (void)nx_sync(ctx); t_start = lap.t0 = now_ms(); }

    hs_pro.stop();
    nx_twiddles* tw = nullptr;
    { HostSpan hs("pm.twiddles_create"); H_TRY(nx_twiddles_create(ctx, max_log + cfg.log_constraint_degree + cfg.log_blowup - 1, &tw)); }      // machine.rs:184-194
    struct TwGuard { nx_twiddles* t; ~TwGuard() { nx_twiddles_destroy(t); } } twg{tw};
    Blake2sChannel channel;
    for (size_t i = 0; i < ad_len; i++) channel.mix_u64(ad[i]);                       // machine.rs:198-200
    CommitmentSchemeProver cs(ctx, tw, cfg);                                          // machine.rs:202-203
    if (comm) { H_TRY(dist_init(ctx, comm, &cs.dist)); cs.dist.comm_ms = &st->comm_ms; cs.dist.comm_bytes = &st->comm_bytes; cs.dist.comm_calls = &st->n_alltoallv; }
    const Dist& D = cs.dist;
    for (uint32_t i = 0; i < n_comps; i++) channel.mix_u64(comps[i].log_size);        // machine.rs:204-206
    if (host && D.on()) return set_err(ctx, NX_ERR_ARG, "nx_prove_machine_host: one GPU (a row-sharded session takes host columns through nx_prover_tree_commit_host)");
    // The fills of the preprocessed and the main trace are queued FIRST (they depend on nothing), the host work of the prove's start —
    // assembling the components' programs, looking their kernels up, the vote of a row-sharded prove — runs while the GPU fills, and
    // only then the two commits (each ends in the synchronisation that fetches its root): the GPU does not idle through the set-up.
    TreeBuilder tb0 = cs.tree_builder(), tb1 = cs.tree_builder();
    GenericAir air; air.ctx = ctx;           // declared after cs: destroyed first
    // A trace tree is consumed by its commitment (the columns become coefficients), and the interaction trace needs evaluations
    // afterwards: the reference clones the whole finalized trace (machine.rs:232) and keeps the preprocessed one; here only the columns
    // the logup fractions read are kept — main columns, and the preprocessed columns of a table component (NX_LOGUP_TABLE).  One GPU:
    // clones of the filled columns (nx_copy).  Row-sharded: the logup is row-wise, so every GPU needs ITS ROWS of those columns — for
    // this synthetic trace the generator fills that block directly.  Host hand-over: cloned as the columns arrive (keep list).
    std::vector<DevBuf> kept[2]; kept[0].resize(n_comps); kept[1].resize(n_comps);
    std::vector<ColMap> kept_ptr[2]; kept_ptr[0].resize(n_comps); kept_ptr[1].resize(n_comps);
    auto stage_tree = [&](uint32_t tree, TreeBuilder& tb) -> int {              // machine.rs:208-228 (tree 0), :230-237 (tree 1)
        std::vector<std::pair<uint32_t, uint32_t>> groups, local;
        for (uint32_t i = 0; i < n_comps; i++) groups.push_back({tree == 0 ? comps[i].n_pre : comps[i].n_main, comps[i].log_size});
        plan_local_columns(groups, D, &local);
        for (uint32_t i = 0; i < n_comps; i++) {
            const nx_component_spec& c = comps[i];
            const uint32_t lo = local[i].first, hi = local[i].second, log = c.log_size, n_tree = groups[i].first;
            DevBuf slab;
            if (host) H_TRY(slab.alloc(ctx, (size_t)n_tree << log));
            else if (hi > lo) {
                H_TRY(slab.alloc(ctx, (size_t)(hi - lo) << log));
                auto p = col_ptrs(slab.p, hi - lo, log);
                H_TRY(synth_fill_range(ctx, c, i, tree, seed, 0, lo, hi - lo, p.data(), 0, 1u << log));
            }
            const std::set<uint32_t> need = logup_needed_columns(c, tree);
            DevBuf& kb = kept[tree][i]; ColMap& kp = kept_ptr[tree][i];
            if (host) {
                // the columns the logup fractions read are cloned as they arrive (keep list); the slab is filled by the commit
                std::vector<std::pair<uint32_t, uint32_t*>> keep;
                if (!need.empty()) {
                    H_TRY(kb.alloc(ctx, need.size() << log));
                    size_t q = 0;
                    for (uint32_t k : need) { uint32_t* dst = kb.p + (q++ << log); keep.push_back({k, dst}); kp[k] = dst; }
                }
                tb.extend_evals_host(std::move(slab), n_tree, log, (tree == 0 ? host->pre + locs[i].pre0 : host->main + locs[i].main0), host->coset_order, keep);
                continue;
            }
            if (!need.empty()) {
                if (D.on() && log < (uint32_t)D.log_w + 2) return set_err(ctx, NX_ERR_ARG, "row-sharded prove: every trace column needs at least 4 rows per GPU");
                const uint64_t nb = D.on() ? D.block(log) : ((uint64_t)1 << log);
                H_TRY(kb.alloc(ctx, need.size() * (size_t)nb));
                size_t q = 0;
                std::vector<uint32_t*> cd; std::vector<const uint32_t*> csrc;
                for (uint32_t k : need) {
                    uint32_t* dst = kb.p + q * (size_t)nb;
                    if (!D.on()) { cd.push_back(dst); csrc.push_back(slab.p + ((size_t)k << log)); }              // Column::clone (R4), batched below
                    else { uint32_t* one[1] = {dst}; H_TRY(synth_fill_range(ctx, c, i, tree, seed, 0, k, 1, one, (uint32_t)D.begin(log), (uint32_t)nb)); }
                    kp[k] = dst; q++;
                }
                H_TRY(copy_columns(ctx, cd.data(), csrc.data(), (uint32_t)cd.size(), (size_t)nb));
            }
            tb.extend_evals_local(std::move(slab), n_tree, log, lo, hi);
        }
        return NX_OK;
    };
    // "machine.reuse_preprocessed": the preprocessed tree of this statement shape was committed by an earlier proof on this context — the
    // columns do not depend on the seed (a program's preprocessed + program trace, reference machine.rs:208-228) — and is adopted instead
    // of being filled, transformed and hashed again (nx_prover_tree_adopt's rule: same root into the transcript, same proof bytes)
    std::string pre_key;
    std::shared_ptr<CommitmentTreeProver> pre_shared;
    if (ctx->opt.machine_reuse_pre && !host && !D.on()) {
        bool table = false;
        for (uint32_t i = 0; i < n_comps; i++) table = table || (comps[i].logup_mode & NX_LOGUP_TABLE);     // a table's fractions read preprocessed EVALUATIONS: those proofs fill the tree
        if (!table) {
            pre_key = "b" + std::to_string(cfg.log_blowup) + "h" + std::to_string(ctx->hash_mode);
            for (uint32_t i = 0; i < n_comps; i++) pre_key += "|" + std::to_string(comps[i].log_size) + ":" + std::to_string(comps[i].n_pre);
            auto it = ctx->machine_pre_cache.find(pre_key);
            if (it != ctx->machine_pre_cache.end()) pre_shared = std::static_pointer_cast<CommitmentTreeProver>(it->second);
        }
    }
    { HostSpan hs("pm.stage_trees(fill queued)"); if (!pre_shared) H_TRY(stage_tree(0, tb0));
    H_TRY(stage_tree(1, tb1)); }
    {
        HostSpan hs("pm.components+prepare #1");
        // Everything that can fail on ONE rank only before the exchanges start — the hiprtc compilation of the components' kernels
        // (the text does not depend on the proof: lookup elements and claimed sums are run-time constants) — happens here, followed
        // by a vote: a rank that failed must not leave its peers blocked in the first all-to-all.
        int rc_local = NX_OK;
        for (uint32_t i = 0; i < n_comps && rc_local == NX_OK; i++) {
            GComponent g = machine_component(comps[i], locs[i], cfg);
            g.log_cd = comps[i].log_constraint_degree_bound;
            rc_local = prepare_component_kernels(ctx, cfg, g, D.on());
            air.comps.push_back(std::move(g));             // kept: only the run-time constants (lookup elements, claimed-sum shift) are filled in later
        }
        H_TRY(vote_before_exchanges(ctx, D, rc_local, "nx_prove_machine"));
    }
    lap(&st->trace_gen);
    bool main_queued = false;
    if (pre_shared) { cs.trees.push_back(borrow_tree(pre_shared)); channel.mix_root(pre_shared->root); }
    else {
        // "machine.queue_trees": both tree builds are queued before the first root is fetched — the main tree's transforms do not depend
        // on the preprocessed root, so the GPU does not idle through that download (the roots still enter the transcript in order)
        main_queued = !host && !D.on() && ctx->opt.machine_queue_trees;
        HostSpan hs("pm.commit tree0 (+begin tree1)");
        if (main_queued) { H_TRY(tb0.commit_begin()); H_TRY(tb1.commit_begin()); H_TRY(tb0.commit_end(channel)); }
        else H_TRY(tb0.commit(channel));                                              // machine.rs:208-228
        if (!pre_key.empty()) {                                                       // keep it for the next proof of this shape
            auto sp = std::make_shared<CommitmentTreeProver>(std::move(cs.trees[0]));
            cs.trees[0] = borrow_tree(sp);
            // one entry: the tree of the statement shape proved last (a prover farm proves one program over and over; every further shape would
            // pin a whole preprocessed tree — coefficients, LDE, Merkle layers — in HBM until the context is destroyed: ADVICE r5)
            ctx->machine_pre_cache.clear();
            ctx->machine_pre_cache[pre_key] = sp;
        }
    }
    { HostSpan hs("pm.commit tree1 end (sync)"); if (main_queued) H_TRY(tb1.commit_end(channel)); else H_TRY(tb1.commit(channel)); }   // machine.rs:230-237
    lap(&st->commit);

    // machine.rs:239-247: draw_lookup_elements, generate_interaction_trace
    uint32_t z[4], alpha[4];
    HostSpan hs_int("pm.interaction (whole stage, host)");
    { std::vector<QM31> za = channel.draw_secure_felts(2); q_store(z, za[0]); q_store(alpha, za[1]); }
    std::vector<QM31> claimed(n_comps, q_zero());
    TreeBuilder tb2 = cs.tree_builder();
    {
        std::vector<std::pair<uint32_t, uint32_t>> groups, local;
        for (uint32_t i = 0; i < n_comps; i++) groups.push_back({comps[i].n_inter, comps[i].log_size});
        plan_local_columns(groups, D, &local);
        for (uint32_t i = 0; i < n_comps; i++) {
            const nx_component_spec& c = comps[i];
            const uint32_t log = c.log_size, L = c.n_inter / 4, lo = local[i].first, hi = local[i].second;
            DevBuf slab;
            if (L) {
                const uint32_t log_rows = D.on() ? log - (uint32_t)D.log_w : log;
                const uint64_t nb = (uint64_t)1 << log_rows;
                DevBuf rows; H_TRY(rows.alloc(ctx, (size_t)c.n_inter * nb));                  // every logup column, this GPU's rows
                std::vector<uint32_t*> ip(c.n_inter);
                for (uint32_t k = 0; k < c.n_inter; k++) ip[k] = rows.p + (size_t)k * nb;
                { HostSpan hs("pm.logup_columns"); H_TRY(logup_columns(ctx, c, log_rows, kept_ptr[1][i], kept_ptr[0][i], z, alpha, ip.data())); }
                uint32_t cs4[4];
                if (!D.on()) {
                    { HostSpan hs("pm.logup_finalize_last (sync)"); H_TRY(nx_logup_finalize_last(ctx, log, ip.data() + 4 * (L - 1), cs4)); }
                    slab = std::move(rows);
                } else {
                    // finalize_last is a prefix sum over ALL rows in natural order: the last column is all-gathered and finalised by
                    // everyone (the claimed sum enters the transcript); the other columns go back to column shards by one all-to-all
                    DevBuf last; H_TRY(last.alloc(ctx, (size_t)4 << log));
                    uint32_t* lp[4];
                    for (int q = 0; q < 4; q++) lp[q] = last.p + ((size_t)q << log);
                    H_TRY(D.allgather_cols(ctx, {ip[4 * (L - 1)], ip[4 * (L - 1) + 1], ip[4 * (L - 1) + 2], ip[4 * (L - 1) + 3]}, (size_t)nb, last.p, (uint64_t)1 << log));
                    H_TRY(nx_logup_finalize_last(ctx, log, lp, cs4));
                    const uint32_t n_loc = hi - lo;
                    DevBuf recv; H_TRY(recv.alloc(ctx, (size_t)std::max<uint32_t>(n_loc, 1) << log));
                    std::vector<size_t> soff(D.world), scnt(D.world), roff(D.world), rcnt(D.world);
                    {
                        // the share of component i's interaction columns each GPU transforms (the same plan on every GPU)
                        for (int r = 0; r < D.world; r++) {
                            Dist dr = D; dr.rank = r;
                            std::vector<std::pair<uint32_t, uint32_t>> lr; plan_local_columns(groups, dr, &lr);
                            soff[r] = (size_t)lr[i].first * nb; scnt[r] = (size_t)(lr[i].second - lr[i].first) * nb;
                            roff[r] = (size_t)r * n_loc * nb; rcnt[r] = (size_t)n_loc * nb;
                        }
                    }
                    H_TRY(D.alltoallv(ctx, rows.p, soff.data(), scnt.data(), recv.p, roff.data(), rcnt.data()));
                    if (n_loc) {
                        H_TRY(slab.alloc(ctx, (size_t)n_loc << log));
                        H_TRY(transpose_blocks(ctx, slab.p, (uint64_t)1 << log, recv.p, n_loc, (uint64_t)1 << log, (uint32_t)D.world, true));
                        for (uint32_t k = std::max(lo, 4 * (L - 1)); k < hi; k++)       // the finalised last column, for the GPUs that transform its coordinates
                            H_TRY(nx_copy(ctx, slab.p + ((size_t)(k - lo) << log), lp[k - 4 * (L - 1)], (size_t)1 << log));
                        H_TRY(nx_sync(ctx));   // `last` is released below
                    }
                }
                claimed[i] = q_load(cs4);
            }
            tb2.extend_evals_local(std::move(slab), c.n_inter, log, lo, hi);
        }
    }
    kept[0].clear(); kept[1].clear();
    hs_int.stop();
    lap(&st->interaction);
    channel.mix_felts(claimed);                                                       // machine.rs:262
    { HostSpan hs("pm.tb2.commit_begin"); H_TRY(tb2.commit_begin()); }                                                        // machine.rs:263, queued; its root is fetched below

    // machine.rs:265-285: the components (recorded programs; lookup elements and claimed-sum shifts are run-time constants), assembled
    // while the GPU builds the interaction tree
    for (uint32_t i = 0; i < n_comps; i++) {
        HostSpan hs("pm.components econsts");
        const QM31 shift = q_mul_m(claimed[i], m_inv((1u << comps[i].log_size) % P));
        fill_econsts(air.comps[i].econsts, z, alpha, shift);
    }
    // one GPU: the interaction tree's polynomials are listed once its build is queued, so the components are checked against the committed
    // trees while the GPU hashes (row-sharded: commit_end does the whole commit, the check follows it)
    if (!D.on()) { HostSpan hs("pm.air.check"); H_TRY(air.check(cs)); }
    { HostSpan hs("pm.tb2.commit_end (sync)"); H_TRY(tb2.commit_end(channel)); }
    lap(&st->commit);
    if (D.on()) { HostSpan hs("pm.air.check"); H_TRY(air.check(cs)); }
    H_TRY(prove_core(ctx, cs, channel, cfg, tw, air, words, st, lap));               // machine.rs:286-290
    ctx->last_claimed.resize(4 * (size_t)n_comps);                                    // Proof.claimed_sum (machine.rs:93-98, :291-296)
    for (uint32_t i = 0; i < n_comps; i++) q_store(&ctx->last_claimed[4 * (size_t)i], claimed[i]);
    if (timed) finish_stats(ctx, st, t_start);
    return NX_OK;
}

}  // namespace nxhip

using namespace nx;

// A rank whose sharded prove failed leaves its peers waiting in the next collective: tell the transport (nx_comm.abort, optional)
// — unless the failure is one every rank reached by itself (ctx->symmetric_failure: the vote, ConstraintsNotSatisfied) or one returned
// before this rank entered its first collective (ctx->comm_entered false: the argument checks ahead of dist_init and the vote — no peer
// waits for a rank that never joined): an abort cannot be undone, and an invalid trace must not cost a prover farm its group.  The
// error CODE does not decide: NX_ERR_ARG also comes from deep inside a sharded prove, one-sided.
static void abort_peers(nx_ctx* ctx, const nx_comm* comm, int rc) {
    if (rc != NX_OK && ctx->comm_entered && !ctx->symmetric_failure && comm && comm->world > 1 && comm->abort) comm->abort(comm->user);
}

static int hand_out(nx_ctx* ctx, int rc, std::vector<uint32_t>& w, uint32_t** proof_words, size_t* n_words, const char* who) {
    ctx->timing = false;
    if (rc != NX_OK) return rc;
    uint32_t* out = (uint32_t*)malloc(std::max<size_t>(w.size(), 1) * 4);
    if (!out) return set_err(ctx, NX_ERR_OOM, std::string(who) + ": malloc failed");
    memcpy(out, w.data(), w.size() * 4);
    *proof_words = out; *n_words = w.size();
    return NX_OK;
}

extern "C" {

int nx_prove_synth(nx_ctx* ctx, const nx_component_spec* comps, uint32_t n_comps, const nx_pcs_config* cfg, uint64_t seed, const uint8_t* ad,
                   size_t ad_len, uint32_t** proof_words, size_t* n_words, nx_prove_stats* stats) {
    NX_GUARD(ctx);
    if (!ctx || !comps || !cfg || !proof_words || !n_words) return set_err(ctx, NX_ERR_ARG, "nx_prove_synth: NULL argument");
    std::vector<uint32_t> w;
    return hand_out(ctx, nxhip::prove_synth(ctx, comps, n_comps, cfg, seed, ad, ad_len, nullptr, &w, stats), w, proof_words, n_words, "nx_prove_synth");
}

int nx_prove_synth_sharded(nx_ctx* ctx, const nx_component_spec* comps, uint32_t n_comps, const nx_pcs_config* cfg, uint64_t seed, const uint8_t* ad,
                           size_t ad_len, const nx_comm* comm, uint32_t** proof_words, size_t* n_words, nx_prove_stats* stats) {
    NX_GUARD(ctx);
    if (!ctx || !comps || !cfg || !proof_words || !n_words || !comm) return set_err(ctx, NX_ERR_ARG, "nx_prove_synth_sharded: NULL argument");
    std::vector<uint32_t> w;
    ctx->symmetric_failure = false; ctx->comm_entered = false;
    const int rc = nxhip::prove_synth(ctx, comps, n_comps, cfg, seed, ad, ad_len, comm, &w, stats);
    abort_peers(ctx, comm, rc);
    return hand_out(ctx, rc, w, proof_words, n_words, "nx_prove_synth_sharded");
}

// The HIP source nx_air_compile generates for the recorded AIR of one nx_prove_machine component (host only, no GPU): for offline
// inspection of the generated kernels (register use, code size) with hipcc.  *h_source: free with nx_free_host.
int nx_machine_air_source(const nx_component_spec* comp, char** h_source) {
    if (!comp || !h_source || comp->n_inter % 4 || comp->n_pre < 2 || comp->n_main < 2) return set_err(nullptr, NX_ERR_ARG, "nx_machine_air_source: bad argument");
    nxhip::PcsConfig one = {0, 1, 1, 0, 0, 1};      // a bound of 0 means "the config's": taken as 1 here (no config in this entry)
    nxhip::GComponent g = nxhip::machine_component(*comp, nxhip::Loc{0, 0, 0}, one);
    return nx_air_compile(nullptr, g.prog.data(), (uint32_t)g.prog.size(), g.n_regs, (uint32_t)g.cols.size(), (uint32_t)g.econsts.size() / 4, g.n_constraints, nullptr, h_source);
}

int nx_machine_air_program(const nx_component_spec* comp, uint32_t cfg_log_constraint_degree, nx_cinstr** h_program, uint32_t* n_instr, uint32_t* n_regs, uint32_t* n_constraints) {
    if (!comp || !h_program || !n_instr || !n_regs || !n_constraints || comp->n_inter % 4 || comp->n_pre < 2 || comp->n_main < 2 || cfg_log_constraint_degree < 1 ||
        cfg_log_constraint_degree > 2 || !nxhip::logup_mode_ok(comp->logup_mode))
        return set_err(nullptr, NX_ERR_ARG, "nx_machine_air_program: bad argument");
    nxhip::PcsConfig cfg = {0, 1, 1, 0, 0, cfg_log_constraint_degree};
    nxhip::GComponent g = nxhip::machine_component(*comp, nxhip::Loc{0, 0, 0}, cfg);
    nx_cinstr* out = (nx_cinstr*)malloc(std::max<size_t>(g.prog.size(), 1) * sizeof(nx_cinstr));
    if (!out) return set_err(nullptr, NX_ERR_OOM, "nx_machine_air_program: malloc failed");
    memcpy(out, g.prog.data(), g.prog.size() * sizeof(nx_cinstr));
    *h_program = out; *n_instr = (uint32_t)g.prog.size(); *n_regs = g.n_regs; *n_constraints = g.n_constraints;
    return NX_OK;
}

int nx_prove_machine(nx_ctx* ctx, const nx_component_spec* comps, uint32_t n_comps, const nx_pcs_config* cfg, uint64_t seed, const uint8_t* ad,
                     size_t ad_len, const nx_comm* comm, uint32_t** proof_words, size_t* n_words, nx_prove_stats* stats) {
    NX_GUARD(ctx);
    if (!ctx || !comps || !cfg || !proof_words || !n_words) return set_err(ctx, NX_ERR_ARG, "nx_prove_machine: NULL argument");
    std::vector<uint32_t> w;
    ctx->symmetric_failure = false; ctx->comm_entered = false;
    int rc;
    { nxhip::HostSpan hs("nx_prove_machine (whole call, destructors included)"); rc = nxhip::prove_machine(ctx, comps, n_comps, cfg, seed, ad, ad_len, comm, &w, stats); }
    nxhip::host_prof_dump("nx_prove_machine");
    abort_peers(ctx, comm, rc);
    return hand_out(ctx, rc, w, proof_words, n_words, "nx_prove_machine");
}

int nx_prove_machine_host(nx_ctx* ctx, const nx_component_spec* comps, uint32_t n_comps, const nx_pcs_config* cfg, const uint32_t* const* h_pre_cols,
                          const uint32_t* const* h_main_cols, int coset_order, const uint8_t* ad, size_t ad_len, uint32_t** proof_words, size_t* n_words,
                          nx_prove_stats* stats) {
    NX_GUARD(ctx);
    if (!ctx || !comps || !cfg || !proof_words || !n_words || !h_pre_cols || !h_main_cols) return set_err(ctx, NX_ERR_ARG, "nx_prove_machine_host: NULL argument");
    nxhip::HostTrace host; host.pre = h_pre_cols; host.main = h_main_cols; host.coset_order = coset_order;
    std::vector<uint32_t> w;
    return hand_out(ctx, nxhip::prove_machine(ctx, comps, n_comps, cfg, 0, ad, ad_len, nullptr, &w, stats, &host), w, proof_words, n_words, "nx_prove_machine_host");
}

int nx_machine_claimed_sums(const nx_ctx* ctx, uint32_t* claimed_sums, uint32_t cap_components, uint32_t* n_components) {
    if (!ctx || !n_components || (cap_components && !claimed_sums)) return set_err(nullptr, NX_ERR_ARG, "nx_machine_claimed_sums: NULL argument");
    const uint32_t n = (uint32_t)(ctx->last_claimed.size() / 4);
    if (n && cap_components) memcpy(claimed_sums, ctx->last_claimed.data(), (size_t)std::min(n, cap_components) * 16);
    *n_components = n;
    return NX_OK;
}

}  // extern "C"
