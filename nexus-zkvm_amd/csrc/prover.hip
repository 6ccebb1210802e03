// Host driver of the device prover: the C++ mirror of the Stwo objects the reference instantiates
//   CommitmentSchemeProver / TreeBuilder   (reference prover/src/machine.rs:202-263)
//   stwo::prover::prove                     (reference prover/src/machine.rs:286-290)
//   FriProver, prove_values, decommit       (inside prove)
// written against the C ABI of include/nexus_hip.h (plus the synthetic machine's own kernels), so
// that everything a Rust `HipBackend` shim needs is exercised through the same entry points.
// The reference's toolchain (Rust nightly) is absent from this image, hence C++ (task rule ②).
#include "internal.h"
#include "air.h"
#include "host/channel.h"
#include <algorithm>
#include <chrono>
#include <map>
#include <set>
#include <tuple>
#include <string.h>
#include <stdlib.h>

namespace nxhip {

using namespace nx;

#define H_TRY(call) do { int rc__ = (call); if (rc__ != NX_OK) return rc__; } while (0)

struct DevBuf {  // owned device allocation
    nx_ctx* ctx = nullptr; uint32_t* p = nullptr; size_t words = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept { ctx = o.ctx; p = o.p; words = o.words; o.p = nullptr; }
    DevBuf& operator=(DevBuf&& o) noexcept { release(); ctx = o.ctx; p = o.p; words = o.words; o.p = nullptr; return *this; }
    int alloc(nx_ctx* c, size_t w) { release(); ctx = c; words = w; return nx_alloc(c, w, &p); }
    void release() { if (p) { (void)nx_free(ctx, p); p = nullptr; } }
    ~DevBuf() { release(); }
};

struct ColumnRef { uint32_t* ptr; uint32_t log; };

struct PcsConfig { uint32_t pow_bits, log_blowup, n_queries, log_last_layer_degree_bound, fri_alpha_mode, log_constraint_degree; };

struct MerkleDecommitment { std::vector<Blake2sHash> hash_witness; std::vector<uint32_t> column_witness; };

// ---------------------------------------------------------------- MerkleProver::decommit ------
// Every word a decommitment needs depends only on the query positions and the layer structure, so all decommitments of a
// proof (4 trees + every FRI layer + the FRI witness values) first record their reads in one GatherBatch, ONE nx_gather
// fetches them (one kernel, one device->host copy, one synchronisation), then each plan picks its words up.
struct GatherBatch {
    std::vector<const uint32_t*> ptrs; std::vector<uint64_t> idx; std::vector<uint32_t> vals;
    size_t add(const uint32_t* p, uint64_t i) { ptrs.push_back(p); idx.push_back(i); return ptrs.size() - 1; }
    int run(nx_ctx* ctx) { vals.resize(ptrs.size()); return nx_gather(ctx, ptrs.data(), idx.data(), ptrs.size(), vals.data()); }
};
struct DecommitPlan { size_t first = 0; std::vector<uint8_t> kind; };  // kind: 0 hash word, 1 queried value, 2 column witness

static DecommitPlan merkle_decommit_plan(const nx_tree* tree, const std::map<uint32_t, std::vector<size_t>>& queries_per_log, std::vector<ColumnRef> cols,
                                         GatherBatch* gb) {
    std::stable_sort(cols.begin(), cols.end(), [](const ColumnRef& a, const ColumnRef& b) { return a.log > b.log; });
    DecommitPlan plan; plan.first = gb->ptrs.size();
    size_t ci = 0;
    std::vector<size_t> last;
    uint32_t n_layers = nx_merkle_n_layers(tree);
    for (int log = (int)n_layers - 1; log >= 0; log--) {
        std::vector<const uint32_t*> lc;
        while (ci < cols.size() && cols[ci].log == (uint32_t)log) lc.push_back(cols[ci++].ptr);
        const uint32_t* prev = (uint32_t)(log + 1) < n_layers ? nx_merkle_layer(tree, log + 1) : nullptr;
        static const std::vector<size_t> none;
        auto it = queries_per_log.find((uint32_t)log);
        const std::vector<size_t>& lq = it == queries_per_log.end() ? none : it->second;
        size_t pi = 0, qi = 0;
        std::vector<size_t> total;
        while (pi < last.size() || qi < lq.size()) {
            size_t node;
            if (pi < last.size() && qi < lq.size()) node = std::min(last[pi] / 2, lq[qi]);
            else if (pi < last.size()) node = last[pi] / 2;
            else node = lq[qi];
            if (prev) {
                for (size_t child = 2 * node; child <= 2 * node + 1; child++) {
                    if (pi < last.size() && last[pi] == child) { pi++; continue; }
                    for (int w = 0; w < 8; w++) { gb->add(prev, child * 8 + w); plan.kind.push_back(0); }
                }
            }
            uint8_t k = 2;
            if (qi < lq.size() && lq[qi] == node) { qi++; k = 1; }
            for (auto c : lc) { gb->add(c, node); plan.kind.push_back(k); }
            total.push_back(node);
        }
        last.swap(total);
    }
    return plan;
}
static void merkle_decommit_fill(const DecommitPlan& plan, const GatherBatch& gb, std::vector<uint32_t>* queried_values, MerkleDecommitment* d) {
    Blake2sHash cur; int hw = 0;
    for (size_t i = 0; i < plan.kind.size(); i++) {
        const uint32_t v = gb.vals[plan.first + i];
        if (plan.kind[i] == 0) { cur.w[hw++] = v; if (hw == 8) { d->hash_witness.push_back(cur); hw = 0; } }
        else if (plan.kind[i] == 1) { if (queried_values) queried_values->push_back(v); }
        else d->column_witness.push_back(v);
    }
}

// ---------------------------------------------------------------- commitment scheme ----------
struct CommitmentTreeProver {
    std::vector<ColumnRef> polys;  // coefficients, commit order
    std::vector<ColumnRef> evals;  // LDE (log = poly log + log_blowup)
    std::vector<DevBuf> bufs;
    nx_tree* merkle = nullptr;
    Blake2sHash root;
    CommitmentTreeProver() {}
    CommitmentTreeProver(CommitmentTreeProver&& o) noexcept
        : polys(std::move(o.polys)), evals(std::move(o.evals)), bufs(std::move(o.bufs)), merkle(o.merkle), root(o.root) { o.merkle = nullptr; }
    CommitmentTreeProver(const CommitmentTreeProver&) = delete;
    ~CommitmentTreeProver() { if (merkle) nx_tree_destroy(merkle); }
};

class CommitmentSchemeProver;

// TreeBuilder::{extend_evals, extend_polys, commit}
class TreeBuilder {
    struct Group { DevBuf slab; uint32_t n_cols, log; bool is_evals; };
    CommitmentSchemeProver& cs;
    std::vector<Group> groups;
  public:
    explicit TreeBuilder(CommitmentSchemeProver& c) : cs(c) {}
    // slab: n_cols contiguous columns of 2^log words (bit-reversed evaluations on CanonicCoset(log).circle_domain())
    void extend_evals(DevBuf&& slab, uint32_t n_cols, uint32_t log) { Group g; g.slab = std::move(slab); g.n_cols = n_cols; g.log = log; g.is_evals = true; groups.push_back(std::move(g)); }
    void extend_polys(DevBuf&& slab, uint32_t n_cols, uint32_t log) { Group g; g.slab = std::move(slab); g.n_cols = n_cols; g.log = log; g.is_evals = false; groups.push_back(std::move(g)); }
    int commit(Blake2sChannel& channel);
};

class CommitmentSchemeProver {
  public:
    nx_ctx* ctx; const nx_twiddles* tw; PcsConfig cfg;
    std::vector<CommitmentTreeProver> trees;
    CommitmentSchemeProver(nx_ctx* c, const nx_twiddles* t, PcsConfig f) : ctx(c), tw(t), cfg(f) {}
    TreeBuilder tree_builder() { return TreeBuilder(*this); }
};

static std::vector<uint32_t*> col_ptrs(uint32_t* base, uint32_t n, uint32_t log) {
    std::vector<uint32_t*> v(n);
    for (uint32_t i = 0; i < n; i++) v[i] = base + ((size_t)i << log);
    return v;
}

// Host words -> a device buffer the caller owns, through the pinned staging ring in chunks: every staged chunk is consumed by the
// copy enqueued right behind it, so — unlike a pointer into the ring — the result stays valid across any number of later stage()
// calls (alpha powers and vanishing denominators of a whole statement live across every component's kernels).
static int upload_owned(nx_ctx* ctx, const uint32_t* h, size_t n_words, DevBuf* out) {
    H_TRY(out->alloc(ctx, std::max<size_t>(n_words, 4)));
    const size_t chunk = (size_t)1 << 20;   // words (4 MiB of the 16 MiB ring)
    for (size_t off = 0; off < n_words; off += chunk) {
        const size_t n = std::min(chunk, n_words - off);
        void* st = nullptr;
        H_TRY(stage(ctx, h + off, n * 4, &st));
        NX_HIP(ctx, hipMemcpyAsync(out->p + off, st, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return NX_OK;
}

// TreeBuilder::commit = CommitmentTreeProver::new (LDE of every polynomial) + MerkleProver::commit.  The leaf layer is one
// Blake2s chain per row over the largest columns in commit order, so it is built incrementally: as soon as a group of
// columns is extended, its 16-column blocks are absorbed on the hash stream while the main stream already extends the next
// group (tree_pipe_* in merkle.hip).  Hashing is VALU-bound, the Circle FFT mostly waits on memory: they overlap well.
static uint32_t pipe_group_cols() {
    static const int v = []() { const char* e = getenv("NX_PIPE_COLS"); int x = e ? atoi(e) : 0; if (x < 16) x = 1 << 30; return (x / 16) * 16; }();   // thread-safe
    return (uint32_t)v;
}

int TreeBuilder::commit(Blake2sChannel& channel) {
    nx_ctx* ctx = cs.ctx;
    CommitmentTreeProver t;
    uint32_t max_el = 0, total_leaf_cols = 0;
    for (auto& g : groups) if (g.n_cols) max_el = std::max(max_el, g.log + cs.cfg.log_blowup);
    for (auto& g : groups) if (g.n_cols && g.log + cs.cfg.log_blowup == max_el) total_leaf_cols += g.n_cols;
    TreePipe tp;
    struct PipeGuard { nx_ctx* c; TreePipe* p; ~PipeGuard() { if (p->tree) { (void)nx_sync(c); nx_tree_destroy(p->tree); p->tree = nullptr; } } } guard{ctx, &tp};
    if (total_leaf_cols) H_TRY(tree_pipe_begin(ctx, max_el, total_leaf_cols, &tp));
    std::vector<const uint32_t*> small_cols; std::vector<uint32_t> small_logs;
    const uint32_t G = pipe_group_cols();
    // consecutive groups of one size (the components of a prover2-style statement) are extended by ONE batch call: a tree of 55 small
    // components is 6 calls, not 55 (each call forks/joins the FFT streams and is launch-bound below ~2^16 rows)
    for (size_t g0 = 0; g0 < groups.size();) {
        size_t g1 = g0 + 1;
        while (g1 < groups.size() && groups[g1].log == groups[g0].log && groups[g1].is_evals == groups[g0].is_evals) g1++;
        const uint32_t log = groups[g0].log, el = log + cs.cfg.log_blowup;
        const bool is_evals = groups[g0].is_evals;
        uint32_t n_run = 0;
        for (size_t g = g0; g < g1; g++) n_run += groups[g].n_cols;
        DevBuf lde;
        if (n_run) {
            H_TRY(lde.alloc(ctx, (size_t)n_run << el));
            std::vector<uint32_t*> in;
            for (size_t g = g0; g < g1; g++) { auto p = col_ptrs(groups[g].slab.p, groups[g].n_cols, log); in.insert(in.end(), p.begin(), p.end()); }
            auto out = col_ptrs(lde.p, n_run, el);
            const bool leaf = el == max_el;
            const uint32_t step = leaf ? G : n_run;
            for (uint32_t c0 = 0; c0 < n_run; c0 += step) {
                const uint32_t nb = std::min(step, n_run - c0);
                if (is_evals) H_TRY(nx_lde_batch(ctx, cs.tw, in.data() + c0, nb, log, cs.cfg.log_blowup, out.data() + c0));   // K3 + K4
                else H_TRY(nx_evaluate_batch(ctx, cs.tw, (const uint32_t* const*)in.data() + c0, nb, log, cs.cfg.log_blowup, out.data() + c0));  // K4
                if (leaf) H_TRY(tree_pipe_absorb(ctx, &tp, (const uint32_t* const*)out.data() + c0, nb, false));                   // K5, leaf layer
            }
            for (uint32_t i = 0; i < n_run; i++) {
                t.polys.push_back({in[i], log}); t.evals.push_back({out[i], el});
                if (!leaf) { small_cols.push_back(out[i]); small_logs.push_back(el); }
            }
        }
        for (size_t g = g0; g < g1; g++) t.bufs.push_back(std::move(groups[g].slab));
        t.bufs.push_back(std::move(lde));
        g0 = g1;
    }
    if (total_leaf_cols) H_TRY(tree_pipe_finish(ctx, &tp, small_cols.data(), small_logs.data(), (uint32_t)small_cols.size(), &t.merkle));   // K5, inner layers
    else H_TRY(nx_merkle_commit(ctx, nullptr, nullptr, 0, &t.merkle));
    H_TRY(nx_merkle_root(ctx, t.merkle, (uint8_t*)t.root.w));
    channel.mix_root(t.root);                                                                // K6
    cs.trees.push_back(std::move(t));
    groups.clear();
    return NX_OK;
}

// ---------------------------------------------------------------- FRI ------------------------
struct SecureColumn {  // SecureColumnByCoords on device
    DevBuf buf; uint32_t log = 0; uint32_t* c[4] = {nullptr, nullptr, nullptr, nullptr};
    int alloc(nx_ctx* ctx, uint32_t l) { log = l; H_TRY(buf.alloc(ctx, (size_t)4 << l)); for (int k = 0; k < 4; k++) c[k] = buf.p + ((size_t)k << l); return NX_OK; }
};
struct FriLayer { SecureColumn eval; nx_tree* merkle = nullptr; Blake2sHash root; };
struct FriLayerProof { std::vector<QM31> fri_witness; MerkleDecommitment decommitment; Blake2sHash commitment; };

struct Proof {
    std::vector<Blake2sHash> commitments;
    std::vector<std::vector<std::vector<QM31>>> sampled_values;
    std::vector<MerkleDecommitment> decommitments;
    std::vector<std::vector<uint32_t>> queried_values;
    uint64_t proof_of_work = 0;
    FriLayerProof first_layer;
    std::vector<FriLayerProof> inner_layers;
    std::vector<QM31> last_layer_poly;
};

static std::vector<size_t> queries_fold(const std::vector<size_t>& q, uint32_t n_folds) {
    std::vector<size_t> r;
    for (size_t p : q) { size_t f = p >> n_folds; if (r.empty() || r.back() != f) r.push_back(f); }
    return r;
}

class FriProver {
  public:
    nx_ctx* ctx; const nx_twiddles* tw; PcsConfig cfg;
    std::vector<SecureColumn> columns;  // first layer, decreasing size
    nx_tree* first_merkle = nullptr; Blake2sHash first_root;
    std::vector<FriLayer> inner;
    std::vector<QM31> last_layer_poly;
    FriProver(nx_ctx* c, const nx_twiddles* t, PcsConfig f) : ctx(c), tw(t), cfg(f) {}
    ~FriProver() { if (first_merkle) nx_tree_destroy(first_merkle); for (auto& l : inner) if (l.merkle) nx_tree_destroy(l.merkle); }

    static int commit_secure(nx_ctx* ctx, const std::vector<const SecureColumn*>& cols, nx_tree** tree, Blake2sHash* root) {
        std::vector<const uint32_t*> p; std::vector<uint32_t> logs;
        for (auto s : cols) for (int k = 0; k < 4; k++) { p.push_back(s->c[k]); logs.push_back(s->log); }
        H_TRY(nx_merkle_commit(ctx, p.data(), logs.data(), (uint32_t)p.size(), tree));
        return nx_merkle_root(ctx, *tree, (uint8_t*)root->w);
    }

    // FriProver::commit.  The whole commit phase is enqueued without a host round trip: the Blake2s channel lives on the device
    // (fri_channel_step after each tree, the folds read their alpha from the channel's records, fri_tail for the small layers);
    // the host reads the roots and the channel state back once, before the last layer.  NX_FRI_DEVICE_CHANNEL=0: the per-layer
    // host channel (same transcript; kept for A/B).
    int commit(Blake2sChannel& channel, std::vector<SecureColumn>&& cols) {
        static const bool dev_channel = []() { const char* e = getenv("NX_FRI_DEVICE_CHANNEL"); return !e || atoi(e) != 0; }();
        if (!dev_channel) return commit_host_channel(channel, std::move(cols));
        columns = std::move(cols);
        if (columns.empty()) return set_err(ctx, NX_ERR_ARG, "fri: no columns");
        const uint32_t last_log = cfg.log_last_layer_degree_bound + cfg.log_blowup;
        uint32_t layer_log = columns[0].log - 1;
        const uint32_t max_layers = layer_log > last_log ? layer_log - last_log : 0;
        const size_t st_words = FRI_STATE_HEAD + FRI_STATE_REC * (size_t)(1 + max_layers);
        DevBuf d_state; H_TRY(d_state.alloc(ctx, st_words));
        auto rec_alpha = [&](int j) { return (const uint32_t*)(d_state.p + FRI_STATE_HEAD + FRI_STATE_REC * j + 8); };
        {
            std::vector<uint32_t> h(FRI_STATE_HEAD, 0);
            memcpy(h.data(), channel.digest.w, 32);
            void* staged = nullptr; H_TRY(stage(ctx, h.data(), h.size() * 4, &staged));
            H_TRY(nx_copy(ctx, d_state.p, (const uint32_t*)staged, FRI_STATE_HEAD));
        }
        {   // first layer: every circle column in one mixed-degree tree
            std::vector<const uint32_t*> p; std::vector<uint32_t> logs;
            for (auto& c : columns) for (int k = 0; k < 4; k++) { p.push_back(c.c[k]); logs.push_back(c.log); }
            H_TRY(nx_merkle_commit(ctx, p.data(), logs.data(), (uint32_t)p.size(), &first_merkle));
            H_TRY(fri_channel_step(ctx, d_state.p, first_merkle->layers[0], 0));
        }
        int j_prev = 0;                         // record whose alpha is the current folding alpha
        SecureColumn layer; H_TRY(layer.alloc(ctx, layer_log));
        H_TRY(nx_memset_zero(ctx, layer.buf.p, layer.buf.words));
        size_t ci = 0;
        static const bool use_tail = []() { const char* e = getenv("NX_FRI_TAIL"); return !e || atoi(e) != 0; }();
        while (layer_log > last_log) {
            const int j = (int)inner.size() + 1;   // record of the layer committed in this iteration
            if (use_tail && ci == columns.size() && layer_log <= (uint32_t)FRI_TAIL_LOG && layer_log - last_log <= (uint32_t)FRI_TAIL_MAX_LAYERS) {
                const int n = (int)(layer_log - last_log);
                std::vector<FriLayer> tl(n);
                std::vector<uint32_t*> evals(n + 1), trees(n);
                tl[0].eval = std::move(layer);
                for (int t = 0; t < n; t++) {
                    if (t) H_TRY(tl[t].eval.alloc(ctx, layer_log - t));
                    for (int k = 0; k < 4; k++) tl[t].eval.c[k] = tl[t].eval.buf.p + ((size_t)k << (layer_log - t));
                    H_TRY(tree_alloc(ctx, layer_log - t, &tl[t].merkle));
                    evals[t] = tl[t].eval.buf.p; trees[t] = tl[t].merkle->layers[0];
                }
                SecureColumn fin; H_TRY(fin.alloc(ctx, last_log));
                evals[n] = fin.buf.p;
                H_TRY(fri_tail(ctx, tw, evals.data(), trees.data(), n, (int)layer_log, d_state.p, j));
                for (int t = 0; t < n; t++) inner.push_back(std::move(tl[t]));
                layer = std::move(fin);
                for (int k = 0; k < 4; k++) layer.c[k] = layer.buf.p + ((size_t)k << last_log);
                layer_log = last_log;
                break;
            }
            while (ci < columns.size() && columns[ci].log - 1 == layer_log) {
                const uint32_t* a = rec_alpha(cfg.fri_alpha_mode == NX_FRI_ALPHA_PREV ? j_prev : 0);
                H_TRY(fold_circle_dev(ctx, tw, layer.c, (const uint32_t* const*)columns[ci].c, columns[ci].log, a));
                ci++;
            }
            FriLayer L; L.eval = std::move(layer);
            for (int k = 0; k < 4; k++) L.eval.c[k] = L.eval.buf.p + ((size_t)k << layer_log);
            {
                std::vector<const uint32_t*> p; std::vector<uint32_t> logs;
                for (int k = 0; k < 4; k++) { p.push_back(L.eval.c[k]); logs.push_back(layer_log); }
                H_TRY(nx_merkle_commit(ctx, p.data(), logs.data(), 4, &L.merkle));
            }
            H_TRY(fri_channel_step(ctx, d_state.p, L.merkle->layers[0], j));
            SecureColumn next; H_TRY(next.alloc(ctx, layer_log - 1));
            H_TRY(fold_line_dev(ctx, tw, (const uint32_t* const*)L.eval.c, layer_log, rec_alpha(j), next.c));
            inner.push_back(std::move(L));
            layer = std::move(next);
            for (int k = 0; k < 4; k++) layer.c[k] = layer.buf.p + ((size_t)k << (layer_log - 1));
            layer_log--; j_prev = j;
        }
        {   // the one host round trip of the commit phase: channel state and every layer's root
            std::vector<uint32_t> st(st_words);
            H_TRY(nx_download(ctx, st.data(), d_state.p, FRI_STATE_HEAD + FRI_STATE_REC * (1 + inner.size())));
            memcpy(channel.digest.w, st.data(), 32);
            channel.n_challenges += (uint32_t)(1 + inner.size()); channel.n_sent = st[8];
            memcpy(first_root.w, &st[FRI_STATE_HEAD], 32);
            for (size_t i = 0; i < inner.size(); i++) memcpy(inner[i].root.w, &st[FRI_STATE_HEAD + FRI_STATE_REC * (i + 1)], 32);
        }
        return commit_last_layer(channel, layer, layer_log, ci);
    }

    int commit_host_channel(Blake2sChannel& channel, std::vector<SecureColumn>&& cols) {
        columns = std::move(cols);
        if (columns.empty()) return set_err(ctx, NX_ERR_ARG, "fri: no columns");
        { std::vector<const SecureColumn*> p; for (auto& c : columns) p.push_back(&c); H_TRY(commit_secure(ctx, p, &first_merkle, &first_root)); }
        channel.mix_root(first_root);
        QM31 folding_alpha = channel.draw_secure_felt();
        const QM31 first_alpha = folding_alpha;
        uint32_t layer_log = columns[0].log - 1;
        SecureColumn layer; H_TRY(layer.alloc(ctx, layer_log));
        H_TRY(nx_memset_zero(ctx, layer.buf.p, layer.buf.words));
        const uint32_t last_log = cfg.log_last_layer_degree_bound + cfg.log_blowup;
        size_t ci = 0; uint32_t n_doublings = 0;
        while (layer_log > last_log) {
            while (ci < columns.size() && columns[ci].log - 1 == layer_log) {
                QM31 a = cfg.fri_alpha_mode == NX_FRI_ALPHA_PREV ? folding_alpha : first_alpha;
                uint32_t aw[4]; q_store(aw, a);
                H_TRY(nx_fold_circle_into_line(ctx, tw, layer.c, (const uint32_t* const*)columns[ci].c, columns[ci].log, aw));
                ci++;
            }
            FriLayer L; L.eval = std::move(layer);
            for (int k = 0; k < 4; k++) L.eval.c[k] = L.eval.buf.p + ((size_t)k << layer_log);
            { std::vector<const SecureColumn*> p{&L.eval}; H_TRY(commit_secure(ctx, p, &L.merkle, &L.root)); }
            channel.mix_root(L.root);
            folding_alpha = channel.draw_secure_felt();
            SecureColumn next; H_TRY(next.alloc(ctx, layer_log - 1));
            uint32_t aw[4]; q_store(aw, folding_alpha);
            H_TRY(nx_fold_line(ctx, tw, (const uint32_t* const*)L.eval.c, layer_log, n_doublings, aw, next.c));
            inner.push_back(std::move(L));
            // moved FriLayer: fix coordinate pointers (buffer address is unchanged by the move)
            layer = std::move(next);
            for (int k = 0; k < 4; k++) layer.c[k] = layer.buf.p + ((size_t)k << (layer_log - 1));
            layer_log--; n_doublings++;
        }
        return commit_last_layer(channel, layer, layer_log, ci);
    }

    int commit_last_layer(Blake2sChannel& channel, SecureColumn& layer, uint32_t layer_log, size_t ci) {
        if (ci != columns.size()) return set_err(ctx, NX_ERR_PROTOCOL, "fri: first-layer columns not consumed (column smaller than the last layer)");
        // commit_last_layer: tiny — interpolate the line evaluation on the host
        size_t n = (size_t)1 << layer_log;
        std::vector<uint32_t> h(4 * n);
        H_TRY(nx_download(ctx, h.data(), layer.buf.p, 4 * n));
        std::vector<QM31> v(n);
        for (size_t i = 0; i < n; i++) v[i] = qm(h[i], h[n + i], h[2 * n + i], h[3 * n + i]);
        // LineEvaluation::interpolate (bit_reverse, line_ifft, scale) then into_ordered_coefficients
        for (size_t i = 0; i < n; i++) { size_t j = bitrev((u32)i, (int)layer_log); if (i < j) std::swap(v[i], v[j]); }
        {
            int dl = (int)layer_log;  // current line domain: half_odds(dl)
            while (dl > 0) {
                size_t ds = (size_t)1 << dl;
                for (size_t c0 = 0; c0 < n; c0 += ds)
                    for (size_t i = 0; i < ds / 2; i++) {
                        u32 x = pt_from_index(half_odds_index(dl, (u32)i)).x;
                        QM31 a = v[c0 + i], b = v[c0 + ds / 2 + i];
                        v[c0 + i] = q_add(a, b); v[c0 + ds / 2 + i] = q_mul_m(q_sub(a, b), m_inv(x));
                    }
                dl--;
            }
        }
        u32 len_inv = m_inv((u32)n);
        for (auto& x : v) x = q_mul_m(x, len_inv);
        for (size_t i = 0; i < n; i++) { size_t j = bitrev((u32)i, (int)layer_log); if (i < j) std::swap(v[i], v[j]); }
        size_t bound = (size_t)1 << cfg.log_last_layer_degree_bound;
        for (size_t i = bound; i < n; i++) if (!q_is_zero(v[i])) return set_err(ctx, NX_ERR_PROTOCOL, "fri: invalid degree in the last layer");
        v.resize(bound);
        channel.mix_felts(v);
        last_layer_poly = v;
        return NX_OK;
    }

    // compute_decommitment_positions_and_witness_evals: positions now, witness values gathered later
    static void decommit_positions(const std::vector<size_t>& queries, std::vector<size_t>* positions, std::vector<size_t>* witness_pos) {
        size_t i = 0;
        while (i < queries.size()) {
            size_t j = i;
            while (j < queries.size() && (queries[j] >> 1) == (queries[i] >> 1)) j++;
            size_t start = (queries[i] >> 1) << 1, qi = i;
            for (size_t pos = start; pos < start + 2; pos++) {
                positions->push_back(pos);
                if (qi < j && queries[qi] == pos) { qi++; continue; }
                witness_pos->push_back(pos);
            }
            i = j;
        }
    }
    struct SecurePlan { size_t first, count; };
    static SecurePlan plan_secure(const SecureColumn& col, const std::vector<size_t>& pos, GatherBatch* gb) {
        SecurePlan p{gb->ptrs.size(), pos.size()};
        for (size_t q : pos) for (int k = 0; k < 4; k++) gb->add(col.c[k], q);
        return p;
    }
    static void fill_secure(const SecurePlan& p, const GatherBatch& gb, std::vector<QM31>* out) {
        for (size_t i = 0; i < p.count; i++) { const uint32_t* v = &gb.vals[p.first + 4 * i]; out->push_back(qm(v[0], v[1], v[2], v[3])); }
    }

    // Queries::generate + the folded positions per column size (FriProver::decommit, first half)
    std::vector<size_t> queries;
    void draw_queries(Blake2sChannel& channel, std::map<uint32_t, std::vector<size_t>>* query_positions_per_log) {
        uint32_t max_log = columns[0].log;
        std::set<size_t> q; uint32_t cnt = 0; size_t mask = ((size_t)1 << max_log) - 1; bool done = false;
        while (!done) {
            uint32_t w[8]; channel.draw_u32s(w);
            for (int i = 0; i < 8 && !done; i++) { q.insert((size_t)w[i] & mask); if (++cnt == cfg.n_queries) done = true; }
        }
        queries.assign(q.begin(), q.end());
        for (auto& c : columns) (*query_positions_per_log)[c.log] = queries_fold(queries, max_log - c.log);
    }

    // FriProver::decommit in two steps around the proof-wide gather: plan (records every read) ... fill
    struct LayerPlan { std::vector<SecurePlan> witness; DecommitPlan merkle; };
    std::vector<LayerPlan> plans;   // [0] first layer, then the inner layers
    void decommit_plan(GatherBatch* gb) {
        uint32_t max_log = columns[0].log;
        {   // first layer
            LayerPlan lp;
            std::map<uint32_t, std::vector<size_t>> pos_by_log;
            std::vector<ColumnRef> refs;
            for (auto& c : columns) {
                std::vector<size_t> pos, wpos;
                decommit_positions(queries_fold(queries, max_log - c.log), &pos, &wpos);
                lp.witness.push_back(plan_secure(c, wpos, gb));
                pos_by_log[c.log] = pos;
                for (int k = 0; k < 4; k++) refs.push_back({c.c[k], c.log});
            }
            lp.merkle = merkle_decommit_plan(first_merkle, pos_by_log, refs, gb);
            plans.push_back(std::move(lp));
        }
        std::vector<size_t> lq = queries_fold(queries, 1);
        for (auto& L : inner) {
            LayerPlan lp;
            std::vector<size_t> pos, wpos;
            decommit_positions(lq, &pos, &wpos);
            lp.witness.push_back(plan_secure(L.eval, wpos, gb));
            std::map<uint32_t, std::vector<size_t>> pos_by_log; pos_by_log[L.eval.log] = pos;
            std::vector<ColumnRef> refs; for (int k = 0; k < 4; k++) refs.push_back({L.eval.c[k], L.eval.log});
            lp.merkle = merkle_decommit_plan(L.merkle, pos_by_log, refs, gb);
            plans.push_back(std::move(lp));
            lq = queries_fold(lq, 1);
        }
    }
    void decommit_fill(const GatherBatch& gb, Proof* proof) {
        for (auto& w : plans[0].witness) fill_secure(w, gb, &proof->first_layer.fri_witness);
        merkle_decommit_fill(plans[0].merkle, gb, nullptr, &proof->first_layer.decommitment);
        proof->first_layer.commitment = first_root;
        for (size_t i = 0; i < inner.size(); i++) {
            FriLayerProof lp;
            for (auto& w : plans[i + 1].witness) fill_secure(w, gb, &lp.fri_witness);
            merkle_decommit_fill(plans[i + 1].merkle, gb, nullptr, &lp.decommitment);
            lp.commitment = inner[i].root;
            proof->inner_layers.push_back(std::move(lp));
        }
        proof->last_layer_poly = last_layer_poly;
    }
};

// ---------------------------------------------------------------- proof wire format ("NXP1") --
static void ser_hash(std::vector<uint32_t>& o, const Blake2sHash& h) { o.insert(o.end(), h.w, h.w + 8); }
static void ser_q(std::vector<uint32_t>& o, QM31 q) { uint32_t w[4]; q_store(w, q); o.insert(o.end(), w, w + 4); }
static void ser_decommit(std::vector<uint32_t>& o, const MerkleDecommitment& d) {
    o.push_back((uint32_t)d.hash_witness.size()); for (auto& h : d.hash_witness) ser_hash(o, h);
    o.push_back((uint32_t)d.column_witness.size()); o.insert(o.end(), d.column_witness.begin(), d.column_witness.end());
}
static void ser_fri_layer(std::vector<uint32_t>& o, const FriLayerProof& l) {
    o.push_back((uint32_t)l.fri_witness.size()); for (auto& q : l.fri_witness) ser_q(o, q);
    ser_decommit(o, l.decommitment); ser_hash(o, l.commitment);
}
static std::vector<uint32_t> serialize(const Proof& p, const PcsConfig& cfg) {
    std::vector<uint32_t> o;
    o.push_back(0x3150584Eu);  // "NXP1"
    o.push_back(cfg.pow_bits); o.push_back(cfg.log_blowup); o.push_back(cfg.n_queries); o.push_back(cfg.log_last_layer_degree_bound);
    o.push_back((uint32_t)p.commitments.size());
    for (auto& h : p.commitments) ser_hash(o, h);
    for (auto& t : p.sampled_values) { o.push_back((uint32_t)t.size()); for (auto& c : t) { o.push_back((uint32_t)c.size()); for (auto& q : c) ser_q(o, q); } }
    for (auto& d : p.decommitments) ser_decommit(o, d);
    for (auto& v : p.queried_values) { o.push_back((uint32_t)v.size()); o.insert(o.end(), v.begin(), v.end()); }
    o.push_back((uint32_t)p.proof_of_work); o.push_back((uint32_t)(p.proof_of_work >> 32));
    ser_fri_layer(o, p.first_layer);
    o.push_back((uint32_t)p.inner_layers.size());
    for (auto& l : p.inner_layers) ser_fri_layer(o, l);
    o.push_back((uint32_t)p.last_layer_poly.size());
    for (auto& q : p.last_layer_poly) ser_q(o, q);
    return o;
}

// ---------------------------------------------------------------- synthetic machine ----------
struct Loc { size_t pre0, main0, inter0; };

static QPt get_random_point(Blake2sChannel& ch) {  // CirclePoint::get_random_point
    QM31 t = ch.draw_secure_felt(), t2 = q_sqr(t);
    QM31 inv = q_inv(q_add(t2, q_one()));
    QPt p; p.x = q_mul(q_sub(q_one(), t2), inv); p.y = q_mul(q_add(t, t), inv);
    return p;
}
static QM31 coset_vanishing_q(uint32_t n, QPt p) { QM31 x = p.x; for (uint32_t i = 1; i < n; i++) x = q_double_x(x); return x; }

// DomainEvaluationAccumulator::finalize — ascending size: lift the running polynomial, add, interpolate
static int finalize_accumulation(CommitmentSchemeProver& cs, std::map<uint32_t, SecureColumn>& sub, DevBuf* out_polys, uint32_t* out_log) {
    nx_ctx* ctx = cs.ctx;
    DevBuf cur; uint32_t cur_log = 0; bool have = false;
    for (auto& kv : sub) {
        uint32_t log = kv.first; SecureColumn& values = kv.second;
        if (have) {
            DevBuf lifted; H_TRY(lifted.alloc(ctx, (size_t)4 << log));
            auto src = col_ptrs(cur.p, 4, cur_log), dst = col_ptrs(lifted.p, 4, log);
            H_TRY(nx_evaluate_batch(ctx, cs.tw, (const uint32_t* const*)src.data(), 4, cur_log, log - cur_log, dst.data()));
            const u32* s4[4] = {dst[0], dst[1], dst[2], dst[3]};
            H_TRY(secure_accumulate(ctx, values.c, s4, 1u << log));
            H_TRY(nx_sync(ctx));
        }
        H_TRY(nx_interpolate_batch(ctx, cs.tw, values.c, 4, log));
        cur = std::move(values.buf); cur_log = log; have = true;
    }
    *out_polys = std::move(cur); *out_log = cur_log;
    return NX_OK;
}

// 1 / coset_vanishing(trace coset, eval_domain.at(i)) over the 2^(e - log_size) cosets of the evaluation domain, bit-reversed
static std::vector<uint32_t> vanishing_denominators(uint32_t log_size, uint32_t e) {
    const uint32_t log_expand = e - log_size;
    std::vector<uint32_t> den((size_t)1 << log_expand);
    for (uint32_t i = 0; i < den.size(); i++) {
        u32 x = pt_from_index(circle_domain_index((int)e, i)).x;
        for (uint32_t k = 1; k < log_size; k++) x = m_double_x(x);
        den[bitrev(i, (int)log_expand)] = m_inv(x);
    }
    return den;
}

// What stwo::prover::prove needs from the components (ComponentProvers): the synthetic machine and recorded AIRs provide it.
typedef std::vector<std::vector<std::vector<QM31>>> SampledValues;   // tree -> column -> mask
typedef std::vector<std::vector<std::vector<QPt>>> MaskPoints;
struct AirProver {
    virtual ~AirProver() {}
    virtual int compute_composition(CommitmentSchemeProver& cs, QM31 random_coeff, DevBuf* out_polys, uint32_t* out_log) = 0;
    virtual void mask_points(QPt oods, MaskPoints* points) = 0;          // the three trace trees
    virtual QM31 eval_composition_at_point(QPt point, const SampledValues& sv, QM31 random_coeff) = 0;
};

// ComponentProvers::compute_composition_polynomial for the synthetic machine.
static int compute_composition(CommitmentSchemeProver& cs, const nx_component_spec* comps, uint32_t n_comps, const std::vector<Loc>& locs,
                               QM31 random_coeff, DevBuf* out_polys, uint32_t* out_log) {
    nx_ctx* ctx = cs.ctx;
    const uint32_t lcd = cs.cfg.log_constraint_degree, blow = cs.cfg.log_blowup;
    size_t total = 0;
    for (uint32_t i = 0; i < n_comps; i++) total += synth_n_constraints(comps[i]);
    std::vector<QM31> powers(total);
    { QM31 a = q_one(); for (size_t i = 0; i < total; i++) { powers[i] = a; a = q_mul(a, random_coeff); } }
    std::map<uint32_t, SecureColumn> sub;  // evaluation-domain log size -> accumulation
    // alpha powers and vanishing denominators of ALL components in one stream-ordered copy (a prover2-style statement has dozens of
    // components: two staging calls each were most of their cost)
    std::vector<uint32_t> params; std::vector<size_t> off_pw(n_comps), off_den(n_comps);
    {
        size_t remaining = total;
        std::map<std::pair<uint32_t, uint32_t>, std::vector<uint32_t>> den_cache;
        for (uint32_t ci = 0; ci < n_comps; ci++) {
            const nx_component_spec& c = comps[ci];
            const uint32_t e = c.log_size + lcd;
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
        const uint32_t e = c.log_size + lcd;
        const u32* d_pw = d_params.p + off_pw[ci];
        const u32* d_den = d_params.p + off_den[ci];
        // trace on the evaluation domain
        ColSet pre, mainc, inter;
        DevBuf ext;
        auto slab_set = [](uint32_t* base, uint32_t log) { ColSet s; s.base = base; s.stride = (uint64_t)1 << log; s.table = nullptr; return s; };
        if (e == c.log_size + blow) {
            pre = slab_set(cs.trees[0].evals[locs[ci].pre0].ptr, e);
            mainc = slab_set(cs.trees[1].evals[locs[ci].main0].ptr, e);
            inter = slab_set(c.n_inter ? cs.trees[2].evals[locs[ci].inter0].ptr : nullptr, e);
        } else {  // "need_to_extend": re-evaluate the polynomials on the constraint domain
            uint32_t ncols = c.n_pre + c.n_main + c.n_inter;
            H_TRY(ext.alloc(ctx, (size_t)ncols << e));
            std::vector<const uint32_t*> src;
            for (uint32_t k = 0; k < c.n_pre; k++) src.push_back(cs.trees[0].polys[locs[ci].pre0 + k].ptr);
            for (uint32_t k = 0; k < c.n_main; k++) src.push_back(cs.trees[1].polys[locs[ci].main0 + k].ptr);
            for (uint32_t k = 0; k < c.n_inter; k++) src.push_back(cs.trees[2].polys[locs[ci].inter0 + k].ptr);
            auto dst = col_ptrs(ext.p, ncols, e);
            H_TRY(nx_evaluate_batch(ctx, cs.tw, src.data(), ncols, c.log_size, e - c.log_size, dst.data()));
            pre = slab_set(ext.p, e);
            mainc = slab_set(ext.p + ((size_t)c.n_pre << e), e);
            inter = slab_set(ext.p + ((size_t)(c.n_pre + c.n_main) << e), e);
        }
        if (!sub.count(e)) { H_TRY(sub[e].alloc(ctx, e)); H_TRY(nx_memset_zero(ctx, sub[e].buf.p, sub[e].buf.words)); }
        SynthRange rg{0, c.n_main, c.n_main, 0, c.n_inter, true};
        H_TRY(synth_constraints(ctx, pre, mainc, inter, rg, (int)c.log_size, (int)e, d_pw, d_den, sub[e].c));
    }
    return finalize_accumulation(cs, sub, out_polys, out_log);
}

// eval_composition_polynomial_at_point (the prover's OODS sanity check, stwo prover/mod.rs::prove)
static QM31 eval_composition_at_point(const nx_component_spec* comps, uint32_t n_comps, const std::vector<Loc>& locs, QPt point,
                                      const std::vector<std::vector<std::vector<QM31>>>& sv, QM31 random_coeff) {
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

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct SynthAir : AirProver {
    const nx_component_spec* comps; uint32_t n_comps; const std::vector<Loc>& locs;
    SynthAir(const nx_component_spec* c, uint32_t n, const std::vector<Loc>& l) : comps(c), n_comps(n), locs(l) {}
    int compute_composition(CommitmentSchemeProver& cs, QM31 rc, DevBuf* out, uint32_t* out_log) override { return nxhip::compute_composition(cs, comps, n_comps, locs, rc, out, out_log); }
    void mask_points(QPt oods, MaskPoints* points) override {
        points->assign(3, {});
        for (uint32_t i = 0; i < n_comps; i++) {
            QPt step; { Pt s = pt_from_index(1u << (31 - comps[i].log_size)); step.x = q_from_m(s.x); step.y = q_from_m(s.y); }
            for (uint32_t k = 0; k < comps[i].n_pre; k++) (*points)[0].push_back({oods});
            for (uint32_t k = 0; k < comps[i].n_main; k++) { if (k < 2) (*points)[1].push_back({oods, qpt_add(oods, step)}); else (*points)[1].push_back({oods}); }
            for (uint32_t k = 0; k < comps[i].n_inter; k++) (*points)[2].push_back({oods});
        }
    }
    QM31 eval_composition_at_point(QPt point, const SampledValues& sv, QM31 rc) override { return nxhip::eval_composition_at_point(comps, n_comps, locs, point, sv, rc); }
};

struct Lap {   // per-stage wall clock of nx_prove_stats (only when the caller asked for stats)
    nx_ctx* ctx; bool timed; double t0;
    void operator()(double* slot) { if (!timed) return; (void)nx_sync(ctx); double t = now_ms(); *slot += t - t0; t0 = t; }
};
static void finish_stats(nx_ctx* ctx, nx_prove_stats* st, double t_start) {
    (void)nx_sync(ctx);
    st->total = now_ms() - t_start;
    timing_flush(ctx);
    st->lde_kernel_ms = ctx->kind_ms[NX_T_LDE]; st->lde_algorithmic_bytes = ctx->kind_bytes[NX_T_LDE];
    st->merkle_kernel_ms = ctx->kind_ms[NX_T_MERKLE]; st->merkle_algorithmic_bytes = ctx->kind_bytes[NX_T_MERKLE];
    ctx->timing = false;
}

// stwo::prover::prove from the point where the three trace trees are committed: composition polynomial, OODS sampling, DEEP
// quotients, FRI, proof of work, decommitment.
static int prove_core(nx_ctx* ctx, CommitmentSchemeProver& cs, Blake2sChannel& channel, const PcsConfig& cfg, const nx_twiddles* tw, AirProver& air,
                      std::vector<uint32_t>* words, nx_prove_stats* st, Lap& lap) {
    // ---------------- stwo::prover::prove ----------------
    QM31 random_coeff = channel.draw_secure_felt();
    DevBuf comp_polys; uint32_t clog = 0;
    H_TRY(air.compute_composition(cs, random_coeff, &comp_polys, &clog));
    lap(&st->composition);
    { TreeBuilder tb = cs.tree_builder(); tb.extend_polys(std::move(comp_polys), 4, clog); H_TRY(tb.commit(channel)); }
    lap(&st->commit);
    QPt oods = get_random_point(channel);

    MaskPoints points;                                                    // tree -> column -> points
    air.mask_points(oods, &points);
    points.resize(4);
    points[3].assign(4, std::vector<QPt>{oods});

    // ---------------- prove_values ----------------
    Proof proof;
    proof.sampled_values.resize(4);
    {   // every tree and size is enqueued first; ONE synchronisation collects all the sampled values (K7)
        struct Pending { std::vector<uint32_t> out; std::vector<std::pair<uint32_t, uint32_t>> where; int t; };
        std::vector<Pending> pend; pend.reserve(64);
        std::vector<EvalJob> jobs;
        size_t n_req = 0;
        for (int t = 0; t < 4; t++) { std::set<uint32_t> logs; for (auto& c : cs.trees[t].polys) logs.insert(c.log); n_req += logs.size(); }
        pend.reserve(n_req);                                       // `out` buffers must not move while jobs point into them
        for (int t = 0; t < 4; t++) {
            auto& tr = cs.trees[t];
            proof.sampled_values[t].resize(tr.polys.size());
            std::map<uint32_t, std::vector<uint32_t>> by_log;  // poly log -> column indices
            for (uint32_t c = 0; c < tr.polys.size(); c++) by_log[tr.polys[c].log].push_back(c);
            for (auto& kv : by_log) {
                std::vector<const uint32_t*> pp; std::vector<uint32_t> pidx, pts;
                pend.emplace_back(); Pending& pd = pend.back(); pd.t = t;
                for (uint32_t li = 0; li < kv.second.size(); li++) {
                    uint32_t c = kv.second[li];
                    pp.push_back(tr.polys[c].ptr);
                    for (uint32_t s = 0; s < points[t][c].size(); s++) {
                        pidx.push_back(li);
                        uint32_t w[8]; q_store(w, points[t][c][s].x); q_store(w + 4, points[t][c][s].y);
                        pts.insert(pts.end(), w, w + 8);
                        pd.where.push_back({c, s});
                    }
                    proof.sampled_values[t][c].resize(points[t][c].size());
                }
                pd.out.assign(4 * pidx.size(), 0);
                H_TRY(eval_at_points_enqueue(ctx, pp.data(), kv.first, pidx.data(), pts.data(), (uint32_t)pidx.size(), pd.out.data(), &jobs));
            }
        }
        H_TRY(eval_at_points_collect(ctx, &jobs));
        for (auto& pd : pend)
            for (size_t i = 0; i < pd.where.size(); i++) proof.sampled_values[pd.t][pd.where[i].first][pd.where[i].second] = q_load(&pd.out[4 * i]);
    }
    { std::vector<QM31> flat; for (auto& t : proof.sampled_values) for (auto& c : t) for (auto& v : c) flat.push_back(v); channel.mix_felts(flat); }
    lap(&st->oods);
    QM31 q_coeff = channel.draw_secure_felt();
    // compute_fri_quotients: all columns flattened, stable-sorted by LDE size (descending), grouped by size
    struct Flat { const uint32_t* ptr; uint32_t log; int t; uint32_t c; };
    std::vector<Flat> all;
    for (int t = 0; t < 4; t++) for (uint32_t c = 0; c < cs.trees[t].evals.size(); c++) all.push_back({cs.trees[t].evals[c].ptr, cs.trees[t].evals[c].log, t, c});
    std::stable_sort(all.begin(), all.end(), [](const Flat& a, const Flat& b) { return a.log > b.log; });
    std::vector<SecureColumn> quotients;
    for (size_t i = 0; i < all.size();) {
        size_t j = i; while (j < all.size() && all[j].log == all[i].log) j++;
        // ColumnSampleBatch::new_vec — group by point, insertion ordered
        std::vector<QPt> bpts; std::vector<std::vector<std::pair<uint32_t, QM31>>> bcols;
        std::vector<const uint32_t*> gcols;
        for (size_t k = i; k < j; k++) {
            gcols.push_back(all[k].ptr);
            const auto& ps = points[all[k].t][all[k].c];
            for (size_t s = 0; s < ps.size(); s++) {
                size_t b = 0;
                for (; b < bpts.size(); b++) if (q_eq(bpts[b].x, ps[s].x) && q_eq(bpts[b].y, ps[s].y)) break;
                if (b == bpts.size()) { bpts.push_back(ps[s]); bcols.emplace_back(); }
                bcols[b].push_back({(uint32_t)(k - i), proof.sampled_values[all[k].t][all[k].c][s]});
            }
        }
        std::vector<uint32_t> fpts, counts, cidx, vals;
        for (size_t b = 0; b < bpts.size(); b++) {
            uint32_t w[8]; q_store(w, bpts[b].x); q_store(w + 4, bpts[b].y); fpts.insert(fpts.end(), w, w + 8);
            counts.push_back((uint32_t)bcols[b].size());
            for (auto& cv : bcols[b]) { cidx.push_back(cv.first); uint32_t q[4]; q_store(q, cv.second); vals.insert(vals.end(), q, q + 4); }
        }
        SecureColumn qc; H_TRY(qc.alloc(ctx, all[i].log));
        uint32_t aw[4]; q_store(aw, q_coeff);
        H_TRY(nx_accumulate_quotients(ctx, all[i].log, gcols.data(), (uint32_t)gcols.size(), aw, (uint32_t)bpts.size(), fpts.data(), counts.data(),
                                      cidx.data(), vals.data(), qc.c));                                                              // K8
        quotients.push_back(std::move(qc));
        for (int k = 0; k < 4; k++) quotients.back().c[k] = quotients.back().buf.p + ((size_t)k << quotients.back().log);
        i = j;
    }
    lap(&st->quotients);
    FriProver fri(ctx, tw, cfg);
    H_TRY(fri.commit(channel, std::move(quotients)));                                                                                 // K9
    lap(&st->fri);
    H_TRY(nx_grind(ctx, (const uint8_t*)channel.digest.w, cfg.pow_bits, &proof.proof_of_work));                                       // K10
    channel.mix_u64(proof.proof_of_work);
    lap(&st->pow);
    std::map<uint32_t, std::vector<size_t>> qpos;
    fri.draw_queries(channel, &qpos);
    GatherBatch gb;
    fri.decommit_plan(&gb);
    DecommitPlan tree_plans[4];
    for (int t = 0; t < 4; t++) tree_plans[t] = merkle_decommit_plan(cs.trees[t].merkle, qpos, cs.trees[t].evals, &gb);
    H_TRY(gb.run(ctx));
    fri.decommit_fill(gb, &proof);
    proof.decommitments.resize(4); proof.queried_values.resize(4);
    for (int t = 0; t < 4; t++) {
        merkle_decommit_fill(tree_plans[t], gb, &proof.queried_values[t], &proof.decommitments[t]);
        proof.commitments.push_back(cs.trees[t].root);
    }
    lap(&st->decommit);
    // ProvingError::ConstraintsNotSatisfied sanity check
    QM31 ce[4]; for (int k = 0; k < 4; k++) ce[k] = proof.sampled_values[3][k][0];
    QM31 lhs = q_add(q_add(ce[0], q_mul(ce[1], qm(0, 1, 0, 0))), q_add(q_mul(ce[2], qm(0, 0, 1, 0)), q_mul(ce[3], qm(0, 0, 0, 1))));
    if (!q_eq(lhs, air.eval_composition_at_point(oods, proof.sampled_values, random_coeff)))
        return set_err(ctx, NX_ERR_PROTOCOL, "ProvingError::ConstraintsNotSatisfied (composition OODS mismatch)");
    *words = serialize(proof, cfg);
    return NX_OK;
}

static int prove_synth(nx_ctx* ctx, const nx_component_spec* comps, uint32_t n_comps, const nx_pcs_config* ucfg, uint64_t seed, const uint8_t* ad,
                       size_t ad_len, std::vector<uint32_t>* words, nx_prove_stats* st) {
    PcsConfig cfg = {ucfg->pow_bits, ucfg->log_blowup, ucfg->n_queries, ucfg->log_last_layer_degree_bound, ucfg->fri_alpha_mode, ucfg->log_constraint_degree};
    if (n_comps == 0) return set_err(ctx, NX_ERR_ARG, "prove: no components");
    if (cfg.log_blowup < 1 || cfg.log_constraint_degree < 1 || cfg.log_constraint_degree > 2) return set_err(ctx, NX_ERR_ARG, "prove: log_blowup >= 1 and log_constraint_degree in {1,2} required");
    H_TRY(nx_ctx_set_hash_mode(ctx, (int)ucfg->hash_mode));
    uint32_t max_log = 0;
    std::vector<Loc> locs; { size_t a = 0, b = 0, c = 0; for (uint32_t i = 0; i < n_comps; i++) { locs.push_back({a, b, c}); a += comps[i].n_pre; b += comps[i].n_main; c += comps[i].n_inter; max_log = std::max(max_log, comps[i].log_size); } }
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
    for (uint32_t i = 0; i < n_comps; i++) channel.mix_u64(comps[i].log_size);        // machine.rs:204-206
    lap(&st->commit);

    auto fill_and_commit = [&](uint32_t tree, uint64_t inter_seed) -> int {
        TreeBuilder tb = cs.tree_builder();
        std::vector<uint32_t*> all;
        std::vector<DevBuf> slabs(n_comps);
        for (uint32_t i = 0; i < n_comps; i++) {
            uint32_t n = tree == 0 ? comps[i].n_pre : tree == 1 ? comps[i].n_main : comps[i].n_inter;
            if (n) H_TRY(slabs[i].alloc(ctx, (size_t)n << comps[i].log_size));
            auto p = col_ptrs(slabs[i].p, n, comps[i].log_size);
            all.insert(all.end(), p.begin(), p.end());
        }
        H_TRY(nx_synth_fill_tree(ctx, comps, n_comps, tree, seed, inter_seed, all.data()));
        lap(&st->trace_gen);
        for (uint32_t i = 0; i < n_comps; i++) {
            uint32_t n = tree == 0 ? comps[i].n_pre : tree == 1 ? comps[i].n_main : comps[i].n_inter;
            tb.extend_evals(std::move(slabs[i]), n, comps[i].log_size);
        }
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

// ================================================================ one proof, columns sharded over the GPUs of a node ===
// SURVEY.md §8(e), BASELINE config #4.  Same protocol, same transcript, same proof bytes as prove_synth; every rank holds a
// contiguous 16-column-aligned block of each trace tree.  What crosses the links (nx_comm): the Blake2s chaining state of
// the leaf layers (ring, 32 B per row per hop), two modular all-reduces (composition accumulator, DEEP quotient), the
// sampled and queried values (KBs) and roots / Merkle witnesses (broadcast).  FRI, grinding and the 4-column composition
// tree are replicated.  Restricted to one component (the shape of config #4).
struct Shard {
    const nx_comm* comm; int rank, world;
    uint32_t total[3];
    std::vector<std::pair<uint32_t, uint32_t>> ranges[3];   // per tree, per rank: [lo, hi)
    uint32_t lo(int t) const { return ranges[t][rank].first; }
    uint32_t hi(int t) const { return ranges[t][rank].second; }
    uint32_t n_local(int t) const { return hi(t) - lo(t); }
    int owner(int t, uint32_t c) const { for (int r = 0; r < world; r++) if (c >= ranges[t][r].first && c < ranges[t][r].second) return r; return -1; }
    std::vector<int> active(int t) const { std::vector<int> a; for (int r = 0; r < world; r++) if (ranges[t][r].second > ranges[t][r].first) a.push_back(r); return a; }
};
static void plan_column_shards(uint32_t n_cols, int world, std::vector<std::pair<uint32_t, uint32_t>>* out) {   // == nexus_zkvm_amd.sharded.plan_column_shards
    uint32_t blocks = (n_cols + 15) / 16, per = blocks / world, extra = blocks % world, b = 0;
    out->clear();
    for (int r = 0; r < world; r++) {
        uint32_t nb = per + ((uint32_t)r < extra ? 1 : 0);
        out->push_back({std::min(n_cols, 16 * b), std::min(n_cols, 16 * (b + nb))});
        b += nb;
    }
}
#define C_TRY(call) do { if ((call) != 0) return set_err(ctx, NX_ERR_HIP, "nx_comm callback failed: " #call); } while (0)

// TreeBuilder::commit of one sharded trace tree: local LDE, chaining-state ring over row chunks, inner layers on the last
// active rank, root to everybody.
static int commit_sharded_tree(CommitmentSchemeProver& cs, const Shard& sh, int t, DevBuf&& slab, uint32_t log, Blake2sChannel& channel) {
    nx_ctx* ctx = cs.ctx;
    const uint32_t el = log + cs.cfg.log_blowup, n_local = sh.n_local(t);
    CommitmentTreeProver tr;
    DevBuf lde;
    std::vector<uint32_t*> out;
    if (n_local) {
        H_TRY(lde.alloc(ctx, (size_t)n_local << el));
        auto in = col_ptrs(slab.p, n_local, log); out = col_ptrs(lde.p, n_local, el);
        H_TRY(nx_lde_batch(ctx, cs.tw, in.data(), n_local, log, cs.cfg.log_blowup, out.data()));
        for (uint32_t i = 0; i < n_local; i++) { tr.polys.push_back({in[i], log}); tr.evals.push_back({out[i], el}); }
    }
    const std::vector<int> act = sh.active(t);
    if (act.empty()) {   // a tree without columns: every rank commits the empty tree
        H_TRY(nx_merkle_commit(ctx, nullptr, nullptr, 0, &tr.merkle));
        H_TRY(nx_merkle_root(ctx, tr.merkle, (uint8_t*)tr.root.w));
    } else {
        const int last = act.back();
        if (n_local) {
            size_t pos = std::find(act.begin(), act.end(), sh.rank) - act.begin();
            const int prev = pos > 0 ? act[pos - 1] : -1, next = pos + 1 < act.size() ? act[pos + 1] : -1;
            const uint64_t n_rows = (uint64_t)1 << el;
            DevBuf st; H_TRY(st.alloc(ctx, (size_t)8 << el));
            const uint32_t n_chunks = (uint32_t)std::min<uint64_t>(32, std::max<uint64_t>(1, n_rows >> 10));   // up to 32 row chunks (>= 1024 rows each): fine-grained enough to keep seven hops busy
            for (uint32_t j = 0; j < n_chunks; j++) {
                const uint64_t rb = n_rows * j / n_chunks, re = n_rows * (j + 1) / n_chunks;
                uint32_t* sp = st.p + rb * 8;
                if (prev >= 0) C_TRY(sh.comm->recv(sh.comm->user, prev, sp, (size_t)(re - rb) * 8));
                H_TRY(nx_merkle_leaf_chain(ctx, (const uint32_t* const*)out.data(), n_local, el, sh.lo(t), sh.total[t], prev >= 0 ? sp : nullptr, sp, rb, re - rb));
                if (next >= 0) { H_TRY(nx_sync(ctx)); C_TRY(sh.comm->send(sh.comm->user, next, sp, (size_t)(re - rb) * 8)); }
            }
            if (sh.rank == last) {
                H_TRY(nx_merkle_from_leaves(ctx, st.p, el, &tr.merkle));
                H_TRY(nx_merkle_root(ctx, tr.merkle, (uint8_t*)tr.root.w));
            }
        }
        C_TRY(sh.comm->broadcast(sh.comm->user, tr.root.w, 32, last));
    }
    tr.bufs.push_back(std::move(slab));
    tr.bufs.push_back(std::move(lde));
    channel.mix_root(tr.root);
    cs.trees.push_back(std::move(tr));
    return NX_OK;
}

static int prove_synth_sharded(nx_ctx* ctx, const nx_component_spec* comps, uint32_t n_comps, const nx_pcs_config* ucfg, uint64_t seed, const uint8_t* ad,
                               size_t ad_len, const nx_comm* comm, std::vector<uint32_t>* words, nx_prove_stats* st) {
    PcsConfig cfg = {ucfg->pow_bits, ucfg->log_blowup, ucfg->n_queries, ucfg->log_last_layer_degree_bound, ucfg->fri_alpha_mode, ucfg->log_constraint_degree};
    if (n_comps != 1) return set_err(ctx, NX_ERR_ARG, "sharded prove: exactly one component (BASELINE config #4 shape)");
    if (cfg.log_blowup < 1 || cfg.log_constraint_degree < 1 || cfg.log_constraint_degree > 2) return set_err(ctx, NX_ERR_ARG, "prove: log_blowup >= 1 and log_constraint_degree in {1,2} required");
    if (!comm || comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world || !comm->send || !comm->recv || !comm->allreduce_m31 || !comm->allgather || !comm->broadcast)
        return set_err(ctx, NX_ERR_ARG, "sharded prove: incomplete nx_comm");
    const nx_component_spec& c = comps[0];
    if (c.n_pre < 2 || c.n_main < 2 || c.log_size < 1 || c.log_size > 28) return set_err(ctx, NX_ERR_ARG, "synthetic component needs n_pre >= 2, n_main >= 2, 1 <= log_size <= 28");
    H_TRY(nx_ctx_set_hash_mode(ctx, (int)ucfg->hash_mode));
    Shard sh; sh.comm = comm; sh.rank = comm->rank; sh.world = comm->world;
    sh.total[0] = c.n_pre; sh.total[1] = c.n_main; sh.total[2] = c.n_inter;
    for (int t = 0; t < 3; t++) plan_column_shards(sh.total[t], sh.world, &sh.ranges[t]);
    std::vector<Loc> locs{{0, 0, 0}};
    const bool timed = st != nullptr;
    nx_prove_stats local_stats;
    if (!st) st = &local_stats;
    memset(st, 0, sizeof *st);
    if (timed) { ctx->timing = true; timing_reset(ctx); }
    double t_start = 0, t0 = 0;
    auto lap = [&](double* slot) { if (!timed) return; (void)nx_sync(ctx); double t = now_ms(); *slot += t - t0; t0 = t; };
    if (timed) { (void)nx_sync(ctx); t_start = t0 = now_ms(); }

    nx_twiddles* tw = nullptr;
    H_TRY(nx_twiddles_create(ctx, c.log_size + cfg.log_constraint_degree + cfg.log_blowup - 1, &tw));
    struct TwGuard { nx_twiddles* t; ~TwGuard() { nx_twiddles_destroy(t); } } twg{tw};
    Blake2sChannel channel;
    for (size_t i = 0; i < ad_len; i++) channel.mix_u64(ad[i]);
    CommitmentSchemeProver cs(ctx, tw, cfg);
    channel.mix_u64(c.log_size);
    lap(&st->commit);

    auto fill_and_commit = [&](uint32_t tree, uint64_t inter_seed) -> int {
        const uint32_t n_local = sh.n_local((int)tree);
        DevBuf slab;
        if (n_local) {
            H_TRY(slab.alloc(ctx, (size_t)n_local << c.log_size));
            auto p = col_ptrs(slab.p, n_local, c.log_size);
            H_TRY(synth_fill_range(ctx, c, 0, tree, seed, inter_seed, sh.lo((int)tree), n_local, p.data()));
        }
        lap(&st->trace_gen);
        H_TRY(commit_sharded_tree(cs, sh, (int)tree, std::move(slab), c.log_size, channel));
        lap(&st->commit);
        return NX_OK;
    };
    H_TRY(fill_and_commit(0, 0));
    H_TRY(fill_and_commit(1, 0));
    QM31 z = channel.draw_secure_felt();
    uint64_t inter_seed = ((u64)z.a.a << 32) ^ (u64)z.a.b ^ ((u64)z.b.a << 16) ^ ((u64)z.b.b << 48);
    channel.mix_felts(std::vector<QM31>(1, q_zero()));
    H_TRY(fill_and_commit(2, inter_seed));

    // ---------------- composition polynomial: local constraints, one modular all-reduce, replicated interpolation ----------------
    QM31 random_coeff = channel.draw_secure_felt();
    const uint32_t lcd = cfg.log_constraint_degree, blow = cfg.log_blowup, e = c.log_size + lcd;
    DevBuf comp_polys;
    {
        const size_t nc = synth_n_constraints(c);
        std::vector<uint32_t> pw(4 * nc);
        { QM31 a = q_one(); std::vector<QM31> powers(nc); for (size_t i = 0; i < nc; i++) { powers[i] = a; a = q_mul(a, random_coeff); }
          for (size_t j = 0; j < nc; j++) q_store(&pw[4 * j], powers[nc - 1 - j]); }
        DevBuf d_pw_buf; H_TRY(upload_owned(ctx, pw.data(), pw.size(), &d_pw_buf)); const u32* d_pw = d_pw_buf.p;
        const uint32_t log_expand = e - c.log_size;
        std::vector<uint32_t> den((size_t)1 << log_expand);
        for (uint32_t i = 0; i < den.size(); i++) {
            u32 x = pt_from_index(circle_domain_index((int)e, i)).x;
            for (uint32_t k = 1; k < c.log_size; k++) x = m_double_x(x);
            den[bitrev(i, (int)log_expand)] = m_inv(x);
        }
        DevBuf d_den_buf; H_TRY(upload_owned(ctx, den.data(), den.size(), &d_den_buf)); const u32* d_den = d_den_buf.p;
        auto slab_set = [](uint32_t* base, uint32_t log) { ColSet s; s.base = base; s.stride = (uint64_t)1 << log; s.table = nullptr; return s; };
        ColSet pre, mainc, inter;
        DevBuf ext;
        const uint32_t np = sh.n_local(0), nm = sh.n_local(1), ni = sh.n_local(2);
        if (e == c.log_size + blow) {
            pre = slab_set(np ? cs.trees[0].evals[0].ptr : nullptr, e);
            mainc = slab_set(nm ? cs.trees[1].evals[0].ptr : nullptr, e);
            inter = slab_set(ni ? cs.trees[2].evals[0].ptr : nullptr, e);
        } else {
            const uint32_t ncols = np + nm + ni;
            H_TRY(ext.alloc(ctx, (size_t)std::max<uint32_t>(ncols, 1) << e));
            std::vector<const uint32_t*> src;
            for (int t = 0; t < 3; t++) for (auto& p : cs.trees[t].polys) src.push_back(p.ptr);
            auto dst = col_ptrs(ext.p, ncols, e);
            if (ncols) H_TRY(nx_evaluate_batch(ctx, cs.tw, src.data(), ncols, c.log_size, e - c.log_size, dst.data()));
            pre = slab_set(ext.p, e);
            mainc = slab_set(ext.p + ((size_t)np << e), e);
            inter = slab_set(ext.p + ((size_t)(np + nm) << e), e);
        }
        SecureColumn acc; H_TRY(acc.alloc(ctx, e)); H_TRY(nx_memset_zero(ctx, acc.buf.p, acc.buf.words));
        SynthRange rg{sh.lo(1), nm, c.n_main, sh.lo(2), ni, sh.lo(0) == 0 && sh.hi(0) >= 2};
        if (nm || ni) H_TRY(synth_constraints(ctx, pre, mainc, inter, rg, (int)c.log_size, (int)e, (const u32*)d_pw, (const u32*)d_den, acc.c));
        H_TRY(nx_sync(ctx));
        C_TRY(comm->allreduce_m31(comm->user, acc.buf.p, acc.buf.words));
        H_TRY(nx_interpolate_batch(ctx, cs.tw, acc.c, 4, e));
        comp_polys = std::move(acc.buf);
    }
    lap(&st->composition);
    { TreeBuilder tb = cs.tree_builder(); tb.extend_polys(std::move(comp_polys), 4, e); H_TRY(tb.commit(channel)); }   // replicated
    lap(&st->commit);
    QPt oods = get_random_point(channel);

    // mask points per GLOBAL column
    std::vector<std::vector<std::vector<QPt>>> points(4);
    {
        QPt step; { Pt s0 = pt_from_index(1u << (31 - c.log_size)); step.x = q_from_m(s0.x); step.y = q_from_m(s0.y); }
        for (uint32_t k = 0; k < c.n_pre; k++) points[0].push_back({oods});
        for (uint32_t k = 0; k < c.n_main; k++) { if (k < 2) points[1].push_back({oods, qpt_add(oods, step)}); else points[1].push_back({oods}); }
        for (uint32_t k = 0; k < c.n_inter; k++) points[2].push_back({oods});
        for (int k = 0; k < 4; k++) points[3].push_back({oods});
    }
    // ---------------- prove_values: local columns, then an all-gather of the (few KB of) sampled values ----------------
    Proof proof;
    proof.sampled_values.resize(4);
    for (int t = 0; t < 4; t++) { proof.sampled_values[t].resize(points[t].size()); for (size_t k = 0; k < points[t].size(); k++) proof.sampled_values[t][k].assign(points[t][k].size(), q_zero()); }
    for (int t = 0; t < 4; t++) {
        auto& tr = cs.trees[t];
        const uint32_t g0 = t < 3 ? sh.lo(t) : 0;
        if (tr.polys.empty()) continue;
        std::vector<const uint32_t*> pp; std::vector<uint32_t> pidx, pts; std::vector<std::pair<uint32_t, uint32_t>> where;
        for (uint32_t li = 0; li < tr.polys.size(); li++) {
            const uint32_t g = g0 + li;
            pp.push_back(tr.polys[li].ptr);
            for (uint32_t s2 = 0; s2 < points[t][g].size(); s2++) {
                pidx.push_back(li);
                uint32_t w[8]; q_store(w, points[t][g][s2].x); q_store(w + 4, points[t][g][s2].y);
                pts.insert(pts.end(), w, w + 8);
                where.push_back({g, s2});
            }
        }
        std::vector<uint32_t> out(4 * pidx.size());
        H_TRY(nx_eval_at_points(ctx, pp.data(), tr.polys[0].log, pidx.data(), pts.data(), (uint32_t)pidx.size(), out.data()));
        for (size_t i = 0; i < where.size(); i++) proof.sampled_values[t][where[i].first][where[i].second] = q_load(&out[4 * i]);
    }
    {   // exchange trees 0..2: every rank contributes the slots of its columns
        std::vector<uint32_t> mine;
        for (int t = 0; t < 3; t++) for (auto& col : proof.sampled_values[t]) for (auto& v : col) { uint32_t w[4]; q_store(w, v); mine.insert(mine.end(), w, w + 4); }
        std::vector<uint32_t> everyone(mine.size() * (size_t)sh.world);
        C_TRY(comm->allgather(comm->user, mine.data(), mine.size() * 4, everyone.data()));
        size_t slot = 0;
        for (int t = 0; t < 3; t++) for (uint32_t g = 0; g < proof.sampled_values[t].size(); g++) {
            const int r = sh.owner(t, g);
            for (auto& v : proof.sampled_values[t][g]) { v = q_load(&everyone[(size_t)r * mine.size() + 4 * slot]); slot++; }
        }
    }
    { std::vector<QM31> flat; for (auto& t : proof.sampled_values) for (auto& col : t) for (auto& v : col) flat.push_back(v); channel.mix_felts(flat); }
    lap(&st->oods);
    QM31 q_coeff = channel.draw_secure_felt();
    // ---------------- DEEP quotients: size groups over the GLOBAL column list; sharded groups = partial sums + all-reduce ----------------
    struct Flat { const uint32_t* ptr; uint32_t log; int t; uint32_t c; };
    std::vector<Flat> all;
    for (int t = 0; t < 3; t++) for (uint32_t g = 0; g < sh.total[t]; g++) {
        const bool local = g >= sh.lo(t) && g < sh.hi(t);
        all.push_back({local ? cs.trees[t].evals[g - sh.lo(t)].ptr : nullptr, c.log_size + blow, t, g});
    }
    for (uint32_t g = 0; g < 4; g++) all.push_back({cs.trees[3].evals[g].ptr, cs.trees[3].evals[g].log, 3, g});
    std::stable_sort(all.begin(), all.end(), [](const Flat& a, const Flat& b) { return a.log > b.log; });
    std::vector<SecureColumn> quotients;
    for (size_t i = 0; i < all.size();) {
        size_t j = i; while (j < all.size() && all[j].log == all[i].log) j++;
        std::vector<QPt> bpts; std::vector<std::vector<std::tuple<uint32_t, QM31, bool>>> bcols;
        std::vector<const uint32_t*> gcols;
        bool sharded_group = false;   // decided by the group's CONTENT (trace-tree columns), identically on every rank
        for (size_t k = i; k < j; k++) {
            const bool local = all[k].ptr != nullptr;
            uint32_t lidx = 0;
            if (local) { lidx = (uint32_t)gcols.size(); gcols.push_back(all[k].ptr); }
            if (all[k].t < 3) sharded_group = true;
            const auto& ps = points[all[k].t][all[k].c];
            for (size_t s2 = 0; s2 < ps.size(); s2++) {
                size_t b = 0;
                for (; b < bpts.size(); b++) if (q_eq(bpts[b].x, ps[s2].x) && q_eq(bpts[b].y, ps[s2].y)) break;
                if (b == bpts.size()) { bpts.push_back(ps[s2]); bcols.emplace_back(); }
                bcols[b].push_back(std::make_tuple(lidx, proof.sampled_values[all[k].t][all[k].c][s2], local));
            }
        }
        std::vector<uint32_t> fpts, counts, cidx, vals; std::vector<uint8_t> is_local;
        for (size_t b = 0; b < bpts.size(); b++) {
            uint32_t w[8]; q_store(w, bpts[b].x); q_store(w + 4, bpts[b].y); fpts.insert(fpts.end(), w, w + 8);
            counts.push_back((uint32_t)bcols[b].size());
            for (auto& cv : bcols[b]) { cidx.push_back(std::get<0>(cv)); uint32_t q[4]; q_store(q, std::get<1>(cv)); vals.insert(vals.end(), q, q + 4); is_local.push_back(std::get<2>(cv) ? 1 : 0); }
        }
        SecureColumn qc; H_TRY(qc.alloc(ctx, all[i].log));
        uint32_t aw[4]; q_store(aw, q_coeff);
        if (sharded_group) {
            H_TRY(nx_accumulate_quotients_partial(ctx, all[i].log, gcols.data(), (uint32_t)gcols.size(), aw, (uint32_t)bpts.size(), fpts.data(), counts.data(),
                                                  cidx.data(), vals.data(), is_local.data(), sh.rank == 0 ? 1 : 0, qc.c));
            H_TRY(nx_sync(ctx));
            C_TRY(comm->allreduce_m31(comm->user, qc.buf.p, qc.buf.words));
        } else {
            H_TRY(nx_accumulate_quotients(ctx, all[i].log, gcols.data(), (uint32_t)gcols.size(), aw, (uint32_t)bpts.size(), fpts.data(), counts.data(),
                                          cidx.data(), vals.data(), qc.c));
        }
        quotients.push_back(std::move(qc));
        for (int k = 0; k < 4; k++) quotients.back().c[k] = quotients.back().buf.p + ((size_t)k << quotients.back().log);
        i = j;
    }
    lap(&st->quotients);
    FriProver fri(ctx, tw, cfg);
    H_TRY(fri.commit(channel, std::move(quotients)));                    // replicated
    lap(&st->fri);
    H_TRY(nx_grind(ctx, (const uint8_t*)channel.digest.w, cfg.pow_bits, &proof.proof_of_work));
    channel.mix_u64(proof.proof_of_work);
    lap(&st->pow);
    // ---------------- decommit ----------------
    std::map<uint32_t, std::vector<size_t>> qpos;
    fri.draw_queries(channel, &qpos);
    GatherBatch gb;
    fri.decommit_plan(&gb);
    const std::vector<size_t>& tq = qpos[c.log_size + blow];             // query positions of the trace trees' single column size
    size_t value_first[3];
    for (int t = 0; t < 3; t++) { value_first[t] = gb.ptrs.size(); for (size_t q : tq) for (auto& ev : cs.trees[t].evals) gb.add(ev.ptr, q); }
    DecommitPlan hash_plans[4];
    for (int t = 0; t < 3; t++) if (cs.trees[t].merkle) hash_plans[t] = merkle_decommit_plan(cs.trees[t].merkle, qpos, std::vector<ColumnRef>(), &gb);
    hash_plans[3] = merkle_decommit_plan(cs.trees[3].merkle, qpos, cs.trees[3].evals, &gb);
    H_TRY(gb.run(ctx));
    fri.decommit_fill(gb, &proof);
    proof.decommitments.resize(4); proof.queried_values.resize(4);
    for (int t = 0; t < 3; t++) {
        // queried values: [query][global column]; every rank contributes its columns
        const uint32_t nl = sh.n_local(t), tot = sh.total[t];
        uint32_t max_local = 0; for (int r = 0; r < sh.world; r++) max_local = std::max(max_local, sh.ranges[t][r].second - sh.ranges[t][r].first);
        std::vector<uint32_t> mine(std::max<size_t>(1, tq.size() * (size_t)max_local), 0);
        for (size_t qi = 0; qi < tq.size(); qi++) for (uint32_t l = 0; l < nl; l++) mine[qi * max_local + l] = gb.vals[value_first[t] + qi * nl + l];
        std::vector<uint32_t> everyone(mine.size() * (size_t)sh.world);
        C_TRY(comm->allgather(comm->user, mine.data(), mine.size() * 4, everyone.data()));
        for (size_t qi = 0; qi < tq.size(); qi++) for (uint32_t g = 0; g < tot; g++) {
            const int r = sh.owner(t, g);
            proof.queried_values[t].push_back(everyone[(size_t)r * mine.size() + qi * max_local + (g - sh.ranges[t][r].first)]);
        }
        // hash witness: from the rank that holds the tree (the last active one; rank 0 for an empty tree, which is replicated)
        const std::vector<int> act = sh.active(t);
        const int holder = act.empty() ? 0 : act.back();
        std::vector<uint32_t> wit;
        if (sh.rank == holder) {
            MerkleDecommitment d; merkle_decommit_fill(hash_plans[t], gb, nullptr, &d);
            for (auto& h : d.hash_witness) wit.insert(wit.end(), h.w, h.w + 8);
        }
        uint32_t n_wit = (uint32_t)wit.size();
        C_TRY(comm->broadcast(comm->user, &n_wit, 4, holder));
        wit.resize(n_wit);
        if (n_wit) C_TRY(comm->broadcast(comm->user, wit.data(), (size_t)n_wit * 4, holder));
        for (uint32_t k = 0; k + 8 <= n_wit; k += 8) { Blake2sHash h; memcpy(h.w, &wit[k], 32); proof.decommitments[t].hash_witness.push_back(h); }
        proof.commitments.push_back(cs.trees[t].root);
    }
    merkle_decommit_fill(hash_plans[3], gb, &proof.queried_values[3], &proof.decommitments[3]);
    proof.commitments.push_back(cs.trees[3].root);
    lap(&st->decommit);
    QM31 ce[4]; for (int k = 0; k < 4; k++) ce[k] = proof.sampled_values[3][k][0];
    QM31 lhs = q_add(q_add(ce[0], q_mul(ce[1], qm(0, 1, 0, 0))), q_add(q_mul(ce[2], qm(0, 0, 1, 0)), q_mul(ce[3], qm(0, 0, 0, 1))));
    if (!q_eq(lhs, eval_composition_at_point(comps, n_comps, locs, oods, proof.sampled_values, random_coeff)))
        return set_err(ctx, NX_ERR_PROTOCOL, "ProvingError::ConstraintsNotSatisfied (composition OODS mismatch)");
    *words = serialize(proof, cfg);
    if (timed) {
        (void)nx_sync(ctx);
        st->total = now_ms() - t_start;
        timing_flush(ctx);
        st->lde_kernel_ms = ctx->kind_ms[NX_T_LDE]; st->lde_algorithmic_bytes = ctx->kind_bytes[NX_T_LDE];
        st->merkle_kernel_ms = ctx->kind_ms[NX_T_MERKLE]; st->merkle_algorithmic_bytes = ctx->kind_bytes[NX_T_MERKLE];
        ctx->timing = false;
    }
    return NX_OK;
}


// ================================================================ recorded AIRs: the nx_prover session ================
// The generic counterpart of prove_synth: components are RECORDED constraint programs (include/nexus_hip.h, nx_air_component —
// what FrameworkComponent<E> is to Stwo, reference prover/src/components/mod.rs:15-57), the caller drives the transcript
// prefix (reference machine.rs:198-263) through the session and nx_prover_prove runs stwo::prover::prove on the device.
struct GComponent {
    uint32_t log_size = 0, n_regs = 0, n_constraints = 0;
    std::vector<nx_cinstr> prog;
    std::vector<uint32_t> econsts;
    std::vector<std::pair<uint32_t, uint32_t>> cols;     // component column -> (tree, column in tree)
    std::vector<std::vector<int>> masks;                 // component column -> row offsets sampled
    const nx_air_kernel* kernel = nullptr; nx_air_kernel* owned = nullptr;
};

struct GenericAir : AirProver {
    nx_ctx* ctx; std::vector<GComponent> comps;
    std::vector<std::vector<std::vector<int>>> offs;     // tree -> column -> union of sampled offsets (first-appearance order)
    std::vector<std::vector<uint32_t>> tree_logs;
    ~GenericAir() override { for (auto& c : comps) if (c.owned) nx_air_kernel_destroy(c.owned); }

    // the consistency rules of oracle-side gair_check: every committed column claimed, sizes agree, loads inside the masks
    int check(const CommitmentSchemeProver& cs) {
        if (cs.trees.size() != 3) return set_err(ctx, NX_ERR_ARG, "nx_prover_prove: exactly three trace trees (preprocessed, main, interaction) must be committed first");
        if (comps.empty()) return set_err(ctx, NX_ERR_ARG, "nx_prover_prove: no components");
        tree_logs.assign(3, {}); offs.assign(3, {});
        std::vector<std::vector<char>> claimed(3);
        for (int t = 0; t < 3; t++) { for (auto& c : cs.trees[t].polys) tree_logs[t].push_back(c.log); claimed[t].assign(tree_logs[t].size(), 0); offs[t].resize(tree_logs[t].size()); }
        for (auto& c : comps) {
            for (size_t k = 0; k < c.cols.size(); k++) {
                const uint32_t t = c.cols[k].first, i = c.cols[k].second;
                if (t > 2 || i >= tree_logs[t].size()) return set_err(ctx, NX_ERR_ARG, "nx_prover_prove: component column outside the committed trees");
                if (tree_logs[t][i] != c.log_size) return set_err(ctx, NX_ERR_ARG, "nx_prover_prove: component column of a different log size than the component");
                claimed[t][i] = 1;
                for (int o : c.masks[k]) if (std::find(offs[t][i].begin(), offs[t][i].end(), o) == offs[t][i].end()) offs[t][i].push_back(o);
            }
            uint32_t n_c = 0;
            // the full instruction validation of nx_air_compile, ALWAYS (a caller-supplied kernel skips the compilation, and the
            // host interpreter of the OODS check indexes registers and secure constants with these fields)
            H_TRY(validate_air_program(ctx, c.prog.data(), (uint32_t)c.prog.size(), c.n_regs, (uint32_t)c.cols.size(), (uint32_t)c.econsts.size() / 4, &n_c));
            if (c.kernel) {
                uint32_t kc = 0, ke = 0, kn = 0;
                air_kernel_shape(c.kernel, &kc, &ke, &kn);
                if (kc != c.cols.size() || ke != c.econsts.size() / 4 || kn != c.n_constraints)
                    return set_err(ctx, NX_ERR_ARG, "nx_prover_prove: the supplied kernel was compiled for a different column / constant / constraint count than the component");
            }
            for (auto& in : c.prog) {
                if (in.op == NX_C_LOAD || in.op == NX_C_LOADE) {
                    const uint32_t w = in.op == NX_C_LOADE ? 4 : 1;
                    for (uint32_t j = 0; j < w; j++) {
                        if (in.a + j >= c.cols.size()) return set_err(ctx, NX_ERR_ARG, "nx_prover_prove: LOAD of a column the component does not claim");
                        const auto& m = c.masks[in.a + j];
                        if (std::find(m.begin(), m.end(), (int)in.b) == m.end()) return set_err(ctx, NX_ERR_ARG, "nx_prover_prove: LOAD at an offset missing from the column's mask");
                    }
                }
            }
            if (n_c != c.n_constraints) return set_err(ctx, NX_ERR_ARG, "nx_prover_prove: constraint count mismatch");
        }
        for (int t = 0; t < 3; t++) for (char x : claimed[t]) if (!x) return set_err(ctx, NX_ERR_ARG, "nx_prover_prove: a committed column is claimed by no component");
        return NX_OK;
    }

    int compute_composition(CommitmentSchemeProver& cs, QM31 random_coeff, DevBuf* out_polys, uint32_t* out_log) override {
        const uint32_t lcd = cs.cfg.log_constraint_degree, blow = cs.cfg.log_blowup;
        size_t total = 0;
        for (auto& c : comps) total += c.n_constraints;
        std::vector<QM31> powers(total);
        { QM31 a = q_one(); for (size_t i = 0; i < total; i++) { powers[i] = a; a = q_mul(a, random_coeff); } }
        std::map<uint32_t, SecureColumn> sub;
        size_t remaining = total;
        for (auto& c : comps) {
            const uint32_t e = c.log_size + lcd;
            const size_t nc = c.n_constraints;
            std::vector<uint32_t> pw(4 * nc);          // the LAST nc remaining powers, reversed (accumulator.columns())
            for (size_t j = 0; j < nc; j++) q_store(&pw[4 * j], powers[remaining - 1 - j]);
            remaining -= nc;
            const std::vector<uint32_t> den = vanishing_denominators(c.log_size, e);
            std::vector<const uint32_t*> ptrs(c.cols.size());
            DevBuf ext;
            if (e == c.log_size + blow) {
                for (size_t k = 0; k < c.cols.size(); k++) ptrs[k] = cs.trees[c.cols[k].first].evals[c.cols[k].second].ptr;
            } else {                                   // "need_to_extend": re-evaluate the polynomials on the constraint domain
                H_TRY(ext.alloc(ctx, c.cols.size() << e));
                std::vector<const uint32_t*> src(c.cols.size());
                for (size_t k = 0; k < c.cols.size(); k++) src[k] = cs.trees[c.cols[k].first].polys[c.cols[k].second].ptr;
                auto dst = col_ptrs(ext.p, (uint32_t)c.cols.size(), e);
                H_TRY(nx_evaluate_batch(ctx, cs.tw, src.data(), (uint32_t)c.cols.size(), c.log_size, e - c.log_size, dst.data()));
                for (size_t k = 0; k < c.cols.size(); k++) ptrs[k] = dst[k];
            }
            if (!sub.count(e)) { H_TRY(sub[e].alloc(ctx, e)); H_TRY(nx_memset_zero(ctx, sub[e].buf.p, sub[e].buf.words)); }
            if (!c.kernel) {
                H_TRY(nx_air_compile(ctx, c.prog.data(), (uint32_t)c.prog.size(), c.n_regs, (uint32_t)c.cols.size(), (uint32_t)c.econsts.size() / 4, c.n_constraints, &c.owned, nullptr));
                c.kernel = c.owned;
            }
            H_TRY(nx_air_eval(ctx, c.kernel, ptrs.data(), c.econsts.data(), pw.data(), den.data(), c.log_size, e, sub[e].c));
        }
        return finalize_accumulation(cs, sub, out_polys, out_log);
    }

    void mask_points(QPt oods, MaskPoints* points) override {
        points->assign(3, {});
        for (int t = 0; t < 3; t++)
            for (size_t c = 0; c < offs[t].size(); c++) {
                std::vector<QPt> pts;
                for (int o : offs[t][c]) {
                    if (o == 0) { pts.push_back(oods); continue; }
                    const int64_t idx = ((int64_t)o * ((int64_t)1 << (31 - tree_logs[t][c]))) & 0x7fffffffLL;
                    Pt s = pt_from_index((u32)idx);
                    QPt step; step.x = q_from_m(s.x); step.y = q_from_m(s.y);
                    pts.push_back(qpt_add(oods, step));
                }
                points->at(t).push_back(pts);
            }
    }

    // the recorded program over QM31: every register holds the value of its expression at the OODS point
    QM31 eval_composition_at_point(QPt point, const SampledValues& sv, QM31 rc) override {
        QM31 acc = q_zero();
        for (auto& c : comps) {
            const QM31 di = q_inv(coset_vanishing_q(c.log_size, point));
            std::vector<QM31> R(c.n_regs, q_zero());
            auto sampled = [&](uint32_t col, int off) {
                const uint32_t t = c.cols[col].first, i = c.cols[col].second;
                const size_t k = std::find(offs[t][i].begin(), offs[t][i].end(), off) - offs[t][i].begin();
                return sv[t][i][k];
            };
            for (auto& in : c.prog) {
                switch (in.op) {
                case NX_C_LOAD: R[in.dst] = sampled(in.a, (int)in.b); break;
                case NX_C_CONST: R[in.dst] = q_from_m(in.a); break;
                case NX_C_ADD: case NX_C_ADDE: case NX_C_ADDEB: R[in.dst] = q_add(R[in.a], R[in.b]); break;
                case NX_C_SUB: case NX_C_SUBE: R[in.dst] = q_sub(R[in.a], R[in.b]); break;
                case NX_C_MUL: case NX_C_MULE: case NX_C_MULEB: R[in.dst] = q_mul(R[in.a], R[in.b]); break;
                case NX_C_NEG: R[in.dst] = q_sub(q_zero(), R[in.a]); break;
                case NX_C_CONSTE: R[in.dst] = q_load(&c.econsts[4 * in.a]); break;
                case NX_C_LOADE: {
                    QM31 v = sampled(in.a, (int)in.b);
                    v = q_add(v, q_mul(sampled(in.a + 1, (int)in.b), qm(0, 1, 0, 0)));
                    v = q_add(v, q_mul(sampled(in.a + 2, (int)in.b), qm(0, 0, 1, 0)));
                    v = q_add(v, q_mul(sampled(in.a + 3, (int)in.b), qm(0, 0, 0, 1)));
                    R[in.dst] = v; break;
                }
                case NX_C_CONSTRAINT_B: case NX_C_CONSTRAINT_E: acc = q_add(q_mul(acc, rc), q_mul(di, R[in.a])); break;
                default: break;
                }
            }
        }
        return acc;
    }
};

}  // namespace nxhip

struct nx_prover {
    nx_ctx* ctx; nx_pcs_config ucfg; nxhip::PcsConfig cfg; nx_twiddles* tw = nullptr; uint32_t max_log;
    nxhip::Blake2sChannel channel;
    nxhip::CommitmentSchemeProver* cs = nullptr;
    struct Run { nxhip::DevBuf slab; uint32_t n_cols, log; };
    std::vector<Run> pending; bool open = false;
};

namespace nxhip {
}  // namespace nxhip

using namespace nx;

extern "C" {

int nx_lde_batch(nx_ctx* ctx, const nx_twiddles* tw, uint32_t* const* d_cols, uint32_t n_cols, uint32_t log_size, uint32_t log_blowup,
                 uint32_t* const* d_lde) {
    NX_GUARD(ctx);
    if (n_cols == 0) return NX_OK;
    ColSet c, o;
    NX_TRY(make_colset(ctx, d_cols, n_cols, &c));
    NX_TRY(make_colset(ctx, d_lde, n_cols, &o));
    return fft_lde(ctx, tw, c, n_cols, log_size, log_blowup, o);
}

int nx_lde_commit(nx_ctx* ctx, const nx_twiddles* tw, uint32_t* const* d_cols, uint32_t n_cols, uint32_t log_size, uint32_t log_blowup,
                  uint32_t* const* d_lde, uint8_t root[32]) {
    NX_GUARD(ctx);
    NX_TRY(nx_lde_batch(ctx, tw, d_cols, n_cols, log_size, log_blowup, d_lde));
    std::vector<uint32_t> logs(n_cols, log_size + log_blowup);
    nx_tree* t = nullptr;
    NX_TRY(nx_merkle_commit(ctx, (const uint32_t* const*)d_lde, logs.data(), n_cols, &t));
    int rc = nx_merkle_root(ctx, t, root);
    nx_tree_destroy(t);
    return rc;
}

int nx_prove_synth_sharded(nx_ctx* ctx, const nx_component_spec* comps, uint32_t n_comps, const nx_pcs_config* cfg, uint64_t seed, const uint8_t* ad,
                           size_t ad_len, const nx_comm* comm, uint32_t** proof_words, size_t* n_words, nx_prove_stats* stats) {
    NX_GUARD(ctx);
    if (!ctx || !comps || !cfg || !proof_words || !n_words || !comm) return set_err(ctx, NX_ERR_ARG, "nx_prove_synth_sharded: NULL argument");
    std::vector<uint32_t> w;
    int rc = nxhip::prove_synth_sharded(ctx, comps, n_comps, cfg, seed, ad, ad_len, comm, &w, stats);
    ctx->timing = false;
    if (rc != NX_OK) return rc;
    uint32_t* out = (uint32_t*)malloc(w.size() * 4);
    if (!out) return set_err(ctx, NX_ERR_OOM, "nx_prove_synth_sharded: malloc failed");
    memcpy(out, w.data(), w.size() * 4);
    *proof_words = out; *n_words = w.size();
    return NX_OK;
}

int nx_prove_synth(nx_ctx* ctx, const nx_component_spec* comps, uint32_t n_comps, const nx_pcs_config* cfg, uint64_t seed, const uint8_t* ad,
                   size_t ad_len, uint32_t** proof_words, size_t* n_words, nx_prove_stats* stats) {
    NX_GUARD(ctx);
    if (!ctx || !comps || !cfg || !proof_words || !n_words) return set_err(ctx, NX_ERR_ARG, "nx_prove_synth: NULL argument");
    std::vector<uint32_t> w;
    int rc = nxhip::prove_synth(ctx, comps, n_comps, cfg, seed, ad, ad_len, &w, stats);
    ctx->timing = false;
    if (rc != NX_OK) return rc;
    uint32_t* out = (uint32_t*)malloc(w.size() * 4);
    if (!out) return set_err(ctx, NX_ERR_OOM, "nx_prove_synth: malloc failed");
    memcpy(out, w.data(), w.size() * 4);
    *proof_words = out; *n_words = w.size();
    return NX_OK;
}


// MerkleProver::decommit(queries_per_log_size, columns) (inside stwo::prover::prove -> CommitmentSchemeProver::prove_values,
// reference machine.rs:286-290): the values of `columns` (commit order, any sizes) at the queried rows, plus the witness a
// verifier needs to recompute the root — sibling hashes not derivable from the queries and the column values of visited but
// unqueried nodes.  query_logs ascending or not, one (log, count) pair per queried layer; `queries` holds the sorted, deduplicated
// positions of every queried layer back to back.  All reads travel in ONE gather launch and one device-to-host copy.
// Outputs are malloc'd (nx_free_host); hash_witness has 8 words per hash.
int nx_merkle_decommit(nx_ctx* ctx, const nx_tree* tree, const uint32_t* const* d_cols, const uint32_t* log_sizes, uint32_t n_cols,
                       const uint32_t* query_logs, const uint32_t* query_counts, uint32_t n_query_logs, const uint64_t* queries,
                       uint32_t** queried_values, size_t* n_queried_values, uint32_t** hash_witness, size_t* n_hashes,
                       uint32_t** column_witness, size_t* n_column_witness) {
    NX_GUARD(ctx);
    if (!ctx || !tree || (n_cols && (!d_cols || !log_sizes)) || (n_query_logs && (!query_logs || !query_counts || !queries)) || !queried_values ||
        !n_queried_values || !hash_witness || !n_hashes || !column_witness || !n_column_witness)
        return set_err(ctx, NX_ERR_ARG, "nx_merkle_decommit: NULL argument");
    const uint32_t n_layers = nx_merkle_n_layers(tree);
    std::map<uint32_t, std::vector<size_t>> qpl;
    size_t off = 0;
    for (uint32_t i = 0; i < n_query_logs; i++) {
        if (query_logs[i] >= n_layers) return set_err(ctx, NX_ERR_ARG, "nx_merkle_decommit: queried layer outside the tree");
        std::vector<size_t>& v = qpl[query_logs[i]];
        for (uint32_t k = 0; k < query_counts[i]; k++) {
            const uint64_t q = queries[off + k];
            if (q >= ((uint64_t)1 << query_logs[i]) || (!v.empty() && q <= v.back())) return set_err(ctx, NX_ERR_ARG, "nx_merkle_decommit: queries must be sorted, distinct and inside their layer");
            v.push_back((size_t)q);
        }
        off += query_counts[i];
    }
    std::vector<nxhip::ColumnRef> cols(n_cols);
    for (uint32_t i = 0; i < n_cols; i++) {
        if (log_sizes[i] >= n_layers || !d_cols[i]) return set_err(ctx, NX_ERR_ARG, "nx_merkle_decommit: column larger than the tree (or NULL)");
        cols[i] = {const_cast<uint32_t*>(d_cols[i]), log_sizes[i]};
    }
    nxhip::GatherBatch gb;
    nxhip::DecommitPlan plan = nxhip::merkle_decommit_plan(tree, qpl, cols, &gb);
    NX_TRY(gb.run(ctx));
    std::vector<uint32_t> qv; nxhip::MerkleDecommitment d;
    nxhip::merkle_decommit_fill(plan, gb, &qv, &d);
    auto dup = [](const uint32_t* src, size_t n_words) -> uint32_t* { uint32_t* p = (uint32_t*)malloc(std::max<size_t>(n_words, 1) * 4); if (p && n_words) memcpy(p, src, n_words * 4); return p; };
    uint32_t* a = dup(qv.data(), qv.size());
    uint32_t* b = dup(d.hash_witness.empty() ? nullptr : d.hash_witness[0].w, d.hash_witness.size() * 8);
    uint32_t* c = dup(d.column_witness.data(), d.column_witness.size());
    if (!a || !b || !c) { free(a); free(b); free(c); return set_err(ctx, NX_ERR_OOM, "nx_merkle_decommit: malloc failed"); }
    *queried_values = a; *n_queried_values = qv.size();
    *hash_witness = b; *n_hashes = d.hash_witness.size();
    *column_witness = c; *n_column_witness = d.column_witness.size();
    return NX_OK;
}

// ---------------------------------------------------------------- nx_prover session (recorded AIRs) --
int nx_prover_create(nx_ctx* ctx, const nx_pcs_config* cfg, uint32_t max_log_size, nx_prover** out) {
    NX_GUARD(ctx);
    if (!ctx || !cfg || !out) return set_err(ctx, NX_ERR_ARG, "nx_prover_create: NULL argument");
    if (cfg->log_blowup < 1 || cfg->log_constraint_degree < 1 || cfg->log_constraint_degree > 2 || max_log_size < 1 || max_log_size + cfg->log_constraint_degree + cfg->log_blowup > 31)
        return set_err(ctx, NX_ERR_ARG, "nx_prover_create: log_blowup >= 1, log_constraint_degree in {1,2}, 1 <= max_log_size required");
    NX_TRY(nx_ctx_set_hash_mode(ctx, (int)cfg->hash_mode));
    nx_prover* p = new nx_prover();
    p->ctx = ctx; p->ucfg = *cfg; p->max_log = max_log_size;
    p->cfg = {cfg->pow_bits, cfg->log_blowup, cfg->n_queries, cfg->log_last_layer_degree_bound, cfg->fri_alpha_mode, cfg->log_constraint_degree};
    int rc = nx_twiddles_create(ctx, max_log_size + cfg->log_constraint_degree + cfg->log_blowup - 1, &p->tw);      // machine.rs:184-194
    if (rc != NX_OK) { delete p; return rc; }
    p->cs = new nxhip::CommitmentSchemeProver(ctx, p->tw, p->cfg);
    *out = p;
    return NX_OK;
}

void nx_prover_destroy(nx_prover* p) {
    NX_GUARD(p ? p->ctx : nullptr);
    if (!p) return;
    (void)nx_sync(p->ctx);
    p->pending.clear();
    delete p->cs;
    nx_twiddles_destroy(p->tw);
    delete p;
}

int nx_prover_mix_u64(nx_prover* p, uint64_t v) { if (!p) return set_err(nullptr, NX_ERR_ARG, "nx_prover_mix_u64: NULL prover"); p->channel.mix_u64(v); return NX_OK; }
int nx_prover_mix_felts(nx_prover* p, const uint32_t* felts, uint32_t n) {
    if (!p || (n && !felts)) return set_err(p ? p->ctx : nullptr, NX_ERR_ARG, "nx_prover_mix_felts: NULL argument");
    std::vector<QM31> v(n);
    for (uint32_t i = 0; i < n; i++) v[i] = q_load(felts + 4 * i);
    p->channel.mix_felts(v);
    return NX_OK;
}
int nx_prover_draw_felt(nx_prover* p, uint32_t out[4]) {
    if (!p || !out) return set_err(p ? p->ctx : nullptr, NX_ERR_ARG, "nx_prover_draw_felt: NULL argument");
    q_store(out, p->channel.draw_secure_felt());
    return NX_OK;
}
// Channel::draw_felts(n): consecutive secure felts taken from the stream of base felts (8 per Blake2s draw, i.e. TWO secure felts
// per draw) — what LookupElements::draw uses for (z, alpha) (reference machine.rs:239-240 draw_lookup_elements).
int nx_prover_draw_felts(nx_prover* p, uint32_t n_felts, uint32_t* out) {
    if (!p || (n_felts && !out)) return set_err(p ? p->ctx : nullptr, NX_ERR_ARG, "nx_prover_draw_felts: NULL argument");
    std::vector<QM31> v = p->channel.draw_secure_felts(n_felts);
    for (uint32_t i = 0; i < n_felts; i++) q_store(out + 4 * (size_t)i, v[i]);
    return NX_OK;
}
int nx_prover_channel_digest(const nx_prover* p, uint8_t digest[32]) {
    if (!p || !digest) return set_err(p ? p->ctx : nullptr, NX_ERR_ARG, "nx_prover_channel_digest: NULL argument");
    memcpy(digest, p->channel.digest.w, 32);
    return NX_OK;
}

int nx_prover_tree_begin(nx_prover* p, const uint32_t* log_sizes, uint32_t n_cols, uint32_t** d_cols_out) {
    NX_GUARD(p ? p->ctx : nullptr);
    if (!p || (n_cols && (!log_sizes || !d_cols_out))) return set_err(p ? p->ctx : nullptr, NX_ERR_ARG, "nx_prover_tree_begin: NULL argument");
    if (p->open) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_begin: the previous tree was not committed");
    if (p->cs->trees.size() >= 3) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_begin: the three trace trees are already committed");
    for (uint32_t i = 0; i < n_cols; i++) if (log_sizes[i] < 1 || log_sizes[i] > p->max_log) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_begin: column log size outside [1, max_log_size]");
    p->pending.clear();
    for (uint32_t i = 0; i < n_cols;) {             // one slab per run of equal sizes (TreeBuilder::extend_evals groups)
        uint32_t j = i; while (j < n_cols && log_sizes[j] == log_sizes[i]) j++;
        nx_prover::Run r; r.n_cols = j - i; r.log = log_sizes[i];
        int rc = r.slab.alloc(p->ctx, (size_t)r.n_cols << r.log);
        if (rc != NX_OK) { p->pending.clear(); return rc; }
        for (uint32_t k = 0; k < r.n_cols; k++) d_cols_out[i + k] = r.slab.p + ((size_t)k << r.log);
        p->pending.push_back(std::move(r));
        i = j;
    }
    p->open = true;
    return NX_OK;
}

int nx_prover_tree_commit(nx_prover* p, uint8_t root[32]) {
    NX_GUARD(p ? p->ctx : nullptr);
    if (!p) return set_err(nullptr, NX_ERR_ARG, "nx_prover_tree_commit: NULL prover");
    if (!p->open) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_commit: no tree was begun");
    nxhip::TreeBuilder tb = p->cs->tree_builder();
    for (auto& r : p->pending) tb.extend_evals(std::move(r.slab), r.n_cols, r.log);
    p->pending.clear(); p->open = false;
    NX_TRY(tb.commit(p->channel));
    if (root) memcpy(root, p->cs->trees.back().root.w, 32);
    return NX_OK;
}

int nx_prover_prove(nx_prover* p, const nx_air_component* comps, uint32_t n_comps, uint32_t** proof_words, size_t* n_words, nx_prove_stats* stats) {
    NX_GUARD(p ? p->ctx : nullptr);
    if (!p || !comps || !proof_words || !n_words) return set_err(p ? p->ctx : nullptr, NX_ERR_ARG, "nx_prover_prove: NULL argument");
    nx_ctx* ctx = p->ctx;
    if (p->open) return set_err(ctx, NX_ERR_ARG, "nx_prover_prove: a tree is begun but not committed");
    nxhip::GenericAir air; air.ctx = ctx;
    for (uint32_t i = 0; i < n_comps; i++) {
        const nx_air_component& u = comps[i];
        if (!u.program || (u.n_cols && (!u.col_tree || !u.col_index || !u.mask_count)) || (u.n_econsts && !u.econsts)) return set_err(ctx, NX_ERR_ARG, "nx_prover_prove: NULL pointer in a component");
        if (!u.mask_offsets) for (uint32_t k = 0; k < u.n_cols; k++) if (u.mask_count[k]) return set_err(ctx, NX_ERR_ARG, "nx_prover_prove: mask_offsets is NULL but a column has a nonzero mask_count");
        nxhip::GComponent g;
        g.log_size = u.log_size; g.n_regs = u.n_regs; g.n_constraints = u.n_constraints; g.kernel = u.kernel;
        g.prog.assign(u.program, u.program + u.n_instr);
        if (u.n_econsts) g.econsts.assign(u.econsts, u.econsts + 4 * (size_t)u.n_econsts);
        size_t m = 0;
        for (uint32_t k = 0; k < u.n_cols; k++) {
            g.cols.push_back({u.col_tree[k], u.col_index[k]});
            std::vector<int> o;
            for (uint32_t j = 0; j < u.mask_count[k]; j++) o.push_back((int)u.mask_offsets[m++]);
            g.masks.push_back(o);
        }
        air.comps.push_back(std::move(g));
    }
    NX_TRY(air.check(*p->cs));
    const bool timed = stats != nullptr;
    nx_prove_stats local;
    nx_prove_stats* st = stats ? stats : &local;
    memset(st, 0, sizeof *st);
    if (timed) { ctx->timing = true; timing_reset(ctx); }
    nxhip::Lap lap{ctx, timed, 0};
    double t_start = 0;
    if (timed) { (void)nx_sync(ctx); t_start = lap.t0 = nxhip::now_ms(); }
    std::vector<uint32_t> w;
    int rc = nxhip::prove_core(ctx, *p->cs, p->channel, p->cfg, p->tw, air, &w, st, lap);
    if (timed) nxhip::finish_stats(ctx, st, t_start);
    ctx->timing = false;
    if (rc != NX_OK) return rc;
    uint32_t* out = (uint32_t*)malloc(w.size() * 4);
    if (!out) return set_err(ctx, NX_ERR_OOM, "nx_prover_prove: malloc failed");
    memcpy(out, w.data(), w.size() * 4);
    *proof_words = out; *n_words = w.size();
    return NX_OK;
}

}  // extern "C"
