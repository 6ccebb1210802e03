// Host driver of the device prover: CommitmentSchemeProver / TreeBuilder (reference prover/src/machine.rs:202-263),
// stwo::prover::prove, FriProver, prove_values, decommit (reference prover/src/machine.rs:286-290) and the nx_prover session over
// recorded AIRs — see prover.h for the objects and for how one proof is spread over the GPUs of a node.
#include "prover.h"
#include <chrono>
#include <memory>
#include <mutex>
#include <string>
#include <numeric>
#include <tuple>

namespace nxhip {

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

#define C_TRY(call) do { if ((call) != 0) return set_err(ctx, NX_ERR_HIP, "nx_comm callback failed: " #call); } while (0)

// ---------------------------------------------------------------- Dist: the transport of a row-sharded prove ----------
int dist_init(nx_ctx* ctx, const nx_comm* comm, Dist* out) {
    if (!comm || comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world) return set_err(ctx, NX_ERR_ARG, "nx_comm: bad rank / world");
    int lw = 0; while ((1 << lw) < comm->world) lw++;
    if ((1 << lw) != comm->world) return set_err(ctx, NX_ERR_ARG, "nx_comm: the number of GPUs of one proof must be a power of two (row blocks are subtrees)");
    if (comm->world > 1 && (!comm->allgather || !comm->alltoallv || !comm->allgather_dev)) return set_err(ctx, NX_ERR_ARG, "nx_comm: allgather, alltoallv and allgather_dev are required");
    out->comm = comm; out->rank = comm->rank; out->world = comm->world; out->log_w = lw;
    return NX_OK;
}
int vote_before_exchanges(nx_ctx* ctx, const Dist& D, int rc_local, const char* who) {
    if (!D.on()) return rc_local;
    struct Symmetric { nx_ctx* c; bool armed = true; ~Symmetric() { if (armed) c->symmetric_failure = true; } } sym{ctx};   // every refusal below is every rank's (the all-gather itself failing is not)
    struct Ballot { int32_t rc; int32_t opt[5]; };
    const Ballot mine = {rc_local, {ctx->opt.air_degree_split, ctx->opt.air_half_domain, ctx->opt.air_quarter_domain, ctx->opt.fri_dist_min_log, ctx->opt.dist_chunks}};
    std::vector<Ballot> all((size_t)D.world);
    { const int rc = D.allgather_host(ctx, &mine, sizeof mine, all.data()); if (rc != NX_OK) { sym.armed = false; return rc; } }
    for (int r = 0; r < D.world; r++)
        if (all[r].rc != NX_OK) return rc_local != NX_OK ? rc_local : set_err(ctx, NX_ERR_HIP, std::string(who) + ": rank " + std::to_string(r) + " failed to prepare its kernels (code " + std::to_string(all[r].rc) + "); no rank proceeds");
    for (int r = 0; r < D.world; r++)
        if (memcmp(all[r].opt, all[0].opt, sizeof mine.opt) != 0)
            return set_err(ctx, NX_ERR_ARG, std::string(who) + ": the ranks of one proof run different context options (air.degree_split / air.half_domain / air.quarter_domain / fri.dist_min_log / dist.chunks: rank 0 vs rank " + std::to_string(r) + "); they shape the exchanges and must agree");
    sym.armed = false;
    return NX_OK;
}

struct CommClock { const Dist& d; double t0; CommClock(const Dist& x) : d(x), t0(now_ms()) {} ~CommClock() { if (d.comm_ms) *d.comm_ms += now_ms() - t0; } };
int Dist::allgather_host(nx_ctx* ctx, const void* send, size_t bytes, void* recv) const {
    CommClock clk(*this);
    ctx->comm_entered = true;
    C_TRY(comm->allgather(comm->user, send, bytes, recv));
    if (comm_bytes) *comm_bytes += bytes * (size_t)(world - 1);
    if (comm_calls) comm_calls[2]++;
    return NX_OK;
}
int Dist::allgather_dev(nx_ctx* ctx, const uint32_t* d_send, size_t words, uint32_t* d_recv) const {
    H_TRY(nx_sync(ctx));                      // the peers read this buffer: it must be complete
    CommClock clk(*this);
    ctx->comm_entered = true;
    C_TRY(comm->allgather_dev(comm->user, d_send, words, d_recv));
    if (comm_bytes) *comm_bytes += words * 4 * (size_t)(world - 1);
    if (comm_calls) comm_calls[1]++;
    return NX_OK;
}
int Dist::allgather_cols(nx_ctx* ctx, const std::vector<const uint32_t*>& blk, size_t rb, uint32_t* whole_base, uint64_t whole_stride) const {
    const size_t n = blk.size();
    if (n == 0) return NX_OK;
    if (n == 1) return allgather_dev(ctx, blk[0], rb, whole_base);
    bool contiguous = true;
    for (size_t k = 1; k < n; k++) contiguous = contiguous && blk[k] == blk[0] + k * rb;
    DevBuf send, tmp;
    const uint32_t* src = blk[0];
    if (!contiguous) {
        H_TRY(send.alloc(ctx, n * rb));
        for (size_t k = 0; k < n; k++) H_TRY(nx_copy(ctx, send.p + k * rb, blk[k], rb));
        src = send.p;
    }
    H_TRY(tmp.alloc(ctx, (size_t)world * n * rb));
    H_TRY(allgather_dev(ctx, src, n * rb, tmp.p));
    return transpose_blocks(ctx, whole_base, whole_stride, tmp.p, (uint32_t)n, (uint64_t)world * rb, (uint32_t)world, true);
}

int Dist::alltoallv(nx_ctx* ctx, const uint32_t* d_send, const size_t* soff, const size_t* scnt, uint32_t* d_recv, const size_t* roff, const size_t* rcnt,
                    hipEvent_t ready) const {
    if (ready) NX_HIP(ctx, hipEventSynchronize(ready));
    else H_TRY(nx_sync(ctx));
    CommClock clk(*this);
    ctx->comm_entered = true;
    C_TRY(comm->alltoallv(comm->user, d_send, soff, scnt, d_recv, roff, rcnt));
    if (comm_bytes) for (int r = 0; r < world; r++) if (r != rank) *comm_bytes += scnt[r] * 4;
    if (comm_calls) comm_calls[0]++;
    return NX_OK;
}

// node = H(left ‖ right) with no column values: the levels above the subtree roots of a row-sharded tree
static Blake2sHash hash_pair(int mode, const Blake2sHash& l, const Blake2sHash& r) {
    Blake2sHash h;
    if (mode == NX_HASH_BLAKE2S) { Blake2sState s; s.update(l.w, 32); s.update(r.w, 32); s.finalize((uint8_t*)h.w); }
    else { uint8_t block[64]; memcpy(block, l.w, 32); memcpy(block + 32, r.w, 32); memset(h.w, 0, 32); Blake2sState::compress(h.w, block, 0, false); }
    return h;
}

int merkle_commit_any(nx_ctx* ctx, const Dist& dist, const uint32_t* const* d_cols, const uint32_t* logs, uint32_t n_cols, TreeRef* out) {
    TreeRef t;
    uint32_t max_log = 0;
    for (uint32_t i = 0; i < n_cols; i++) max_log = std::max(max_log, logs[i]);
    t.n_layers = max_log + 1;
    if (!dist.on() || n_cols == 0) {   // a tree without columns is the hash of nothing on every GPU
        H_TRY(nx_merkle_commit(ctx, d_cols, logs, n_cols, &t.local));
        H_TRY(nx_merkle_root(ctx, t.local, (uint8_t*)t.root.w));
        *out = std::move(t);
        return NX_OK;
    }
    std::vector<uint32_t> ll(n_cols);
    for (uint32_t i = 0; i < n_cols; i++) {
        if (logs[i] < (uint32_t)dist.log_w + 2) return set_err(ctx, NX_ERR_ARG, "row-sharded commit: every column needs at least 4 rows per GPU");
        ll[i] = logs[i] - (uint32_t)dist.log_w;
    }
    H_TRY(nx_merkle_commit(ctx, d_cols, ll.data(), n_cols, &t.local));
    Blake2sHash mine;
    H_TRY(nx_merkle_root(ctx, t.local, (uint8_t*)mine.w));
    t.log_w = dist.log_w;
    t.top.resize(dist.log_w + 1);
    t.top[dist.log_w].resize(dist.world);
    H_TRY(dist.allgather_host(ctx, mine.w, 32, t.top[dist.log_w].data()));
    for (int k = dist.log_w - 1; k >= 0; k--) {
        t.top[k].resize((size_t)1 << k);
        for (size_t i = 0; i < t.top[k].size(); i++) t.top[k][i] = hash_pair(ctx->hash_mode, t.top[k + 1][2 * i], t.top[k + 1][2 * i + 1]);
    }
    t.root = t.top[0][0];
    *out = std::move(t);
    return NX_OK;
}

// ---------------------------------------------------------------- MerkleProver::decommit ------
int GatherBatch::run(nx_ctx* ctx) {
    vals.assign(ptrs.size(), 0);
    H_TRY(nx_gather(ctx, ptrs.data(), idx.data(), ptrs.size(), vals.data()));
    if (dist && dist->on() && !vals.empty()) {
        std::vector<uint32_t> all(vals.size() * (size_t)dist->world);
        H_TRY(dist->allgather_host(ctx, vals.data(), vals.size() * 4, all.data()));
        for (size_t i = 0; i < vals.size(); i++) if (owner[i] >= 0) vals[i] = all[(size_t)owner[i] * vals.size() + i];
    }
    for (auto& iv : imm) vals[iv.first] = iv.second;
    return NX_OK;
}

DecommitPlan merkle_decommit_plan(const TreeRef& tree, const std::map<uint32_t, std::vector<size_t>>& queries_per_log, std::vector<ColumnRef> cols, GatherBatch* gb) {
    std::stable_sort(cols.begin(), cols.end(), [](const ColumnRef& a, const ColumnRef& b) { return a.log > b.log; });
    DecommitPlan plan; plan.first = gb->ptrs.size();
    size_t ci = 0;
    std::vector<size_t> last;
    const uint32_t n_layers = tree.n_layers;
    for (int log = (int)n_layers - 1; log >= 0; log--) {
        std::vector<ColumnRef> lc;
        while (ci < cols.size() && cols[ci].log == (uint32_t)log) lc.push_back(cols[ci++]);
        const bool has_prev = (uint32_t)(log + 1) < n_layers;
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
            if (has_prev) {
                for (size_t child = 2 * node; child <= 2 * node + 1; child++) {
                    if (pi < last.size() && last[pi] == child) { pi++; continue; }
                    gb->add_node(tree, (uint32_t)log + 1, child);
                    for (int w = 0; w < 8; w++) plan.kind.push_back(0);
                }
            }
            uint8_t k = 2;
            if (qi < lq.size() && lq[qi] == node) { qi++; k = 1; }
            for (auto& c : lc) { gb->add_column(c, node); plan.kind.push_back(k); }
            total.push_back(node);
        }
        last.swap(total);
    }
    return plan;
}
void merkle_decommit_fill(const DecommitPlan& plan, const GatherBatch& gb, std::vector<uint32_t>* queried_values, MerkleDecommitment* d) {
    Blake2sHash cur; int hw = 0;
    for (size_t i = 0; i < plan.kind.size(); i++) {
        const uint32_t v = gb.vals[plan.first + i];
        if (plan.kind[i] == 0) { cur.w[hw++] = v; if (hw == 8) { d->hash_witness.push_back(cur); hw = 0; } }
        else if (plan.kind[i] == 1) { if (queried_values) queried_values->push_back(v); }
        else d->column_witness.push_back(v);
    }
}

// ---------------------------------------------------------------- commitment scheme ----------
CommitmentTreeProver borrow_tree(const std::shared_ptr<CommitmentTreeProver>& src) {
    CommitmentTreeProver t;
    t.polys = src->polys; t.owner = src->owner; t.evals = src->evals;
    t.merkle.local = src->merkle.local; t.merkle.borrowed = true; t.merkle.log_w = src->merkle.log_w; t.merkle.n_layers = src->merkle.n_layers;
    t.merkle.top = src->merkle.top; t.merkle.root = src->merkle.root;
    t.root = src->root;
    t.backing = src;
    return t;
}

std::vector<uint32_t*> col_ptrs(uint32_t* base, uint32_t n, uint32_t log) {
    std::vector<uint32_t*> v(n);
    for (uint32_t i = 0; i < n; i++) v[i] = base + ((size_t)i << log);
    return v;
}

// Host words -> a device buffer the caller owns, through the pinned staging ring in chunks: every staged chunk is consumed by the
// copy enqueued right behind it, so — unlike a pointer into the ring — the result stays valid across any number of later stage()
// calls (alpha powers and vanishing denominators of a whole statement live across every component's kernels).
int upload_owned(nx_ctx* ctx, const uint32_t* h, size_t n_words, DevBuf* out) {
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

void plan_local_columns(const std::vector<std::pair<uint32_t, uint32_t>>& groups, const Dist& dist, std::vector<std::pair<uint32_t, uint32_t>>* out) {
    out->assign(groups.size(), {0u, 0u});
    for (size_t g0 = 0; g0 < groups.size();) {
        size_t g1 = g0 + 1;
        while (g1 < groups.size() && groups[g1].second == groups[g0].second) g1++;
        uint32_t n_run = 0;
        for (size_t g = g0; g < g1; g++) n_run += groups[g].first;
        const uint32_t lo = Dist::cut(n_run, dist.rank, dist.world), hi = Dist::cut(n_run, dist.rank + 1, dist.world);
        uint32_t off = 0;
        for (size_t g = g0; g < g1; g++) {
            const uint32_t a = std::max(lo, off), b = std::min(hi, off + groups[g].first);
            (*out)[g] = a < b ? std::make_pair(a - off, b - off) : std::make_pair(0u, 0u);
            off += groups[g].first;
        }
        g0 = g1;
    }
}

// TreeBuilder::commit = CommitmentTreeProver::new (LDE of every polynomial) + MerkleProver::commit.  The leaf layer is one
// Blake2s chain per row over the largest columns in commit order, so it is built incrementally: as soon as a group of
// columns is extended, its 16-column blocks are absorbed on the hash stream while the main stream already extends the next
// group (tree_pipe_* in merkle.hip).
static uint32_t pipe_group_cols(const nx_ctx* ctx) { return ctx->opt.commit_pipe_cols >= 16 ? (uint32_t)ctx->opt.commit_pipe_cols : (1u << 30); }

int TreeBuilder::commit_end(Blake2sChannel& channel) {
    if (cs.dist.on()) return commit_dist(channel);
    nx_ctx* ctx = cs.ctx;
    if (!begun) return set_err(ctx, NX_ERR_ARG, "TreeBuilder::commit_end without commit_begin");
    begun = false;
    CommitmentTreeProver& t = cs.trees[tree_index];          // not necessarily the last: a caller may begin the next tree before ending this one
    H_TRY(nx_merkle_root(ctx, t.merkle.local, (uint8_t*)t.root.w));                          // the commit's synchronisation
    t.merkle.root = t.root;
    for (auto& f : feeds) H_TRY(f->finish());                                                // the host columns are the caller's again
    feeds.clear();
    channel.mix_root(t.root);                                                                // K6
    return NX_OK;
}

int TreeBuilder::commit_begin() {
    if (cs.dist.on()) return NX_OK;
    nx_ctx* ctx = cs.ctx;
    if (begun) return set_err(ctx, NX_ERR_ARG, "TreeBuilder::commit_begin twice");
    CommitmentTreeProver t;
    uint32_t max_el = 0, total_leaf_cols = 0;
    for (auto& g : groups) if (g.n_cols) max_el = std::max(max_el, g.log + cs.cfg.log_blowup);
    for (auto& g : groups) if (g.n_cols && g.log + cs.cfg.log_blowup == max_el) total_leaf_cols += g.n_cols;
    for (auto& g : groups) if (g.lo != 0 || g.hi != g.n_cols) return set_err(ctx, NX_ERR_ARG, "TreeBuilder: a column shard was handed to a single-GPU commitment scheme");
    TreePipe tp;
    struct PipeGuard { nx_ctx* c; TreePipe* p; ~PipeGuard() { if (p->tree) { (void)nx_sync(c); nx_tree_destroy(p->tree); p->tree = nullptr; } } } guard{ctx, &tp};
    if (total_leaf_cols) { H_TRY(tree_pipe_begin(ctx, max_el, total_leaf_cols, &tp)); tp.side_stream = pipe_group_cols(ctx) < (1u << 29); }
    std::vector<const uint32_t*> small_cols; std::vector<uint32_t> small_logs;
    const uint32_t G = pipe_group_cols(ctx);
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
            // host-resident columns of this run: per column its host source, and what to clone on arrival
            std::vector<const uint32_t*> hsrc(n_run, nullptr); std::vector<uint32_t*> keep_of(n_run, nullptr);
            bool any_host = false; int coset_order = 0;
            {
                uint32_t off = 0;
                for (size_t g = g0; g < g1; g++) {
                    if (!groups[g].host.empty()) {
                        if (any_host && coset_order != groups[g].coset_order) return set_err(ctx, NX_ERR_ARG, "TreeBuilder: the host columns of one run must share one order");
                        any_host = true; coset_order = groups[g].coset_order;
                        for (uint32_t k = 0; k < groups[g].n_cols; k++) hsrc[off + k] = groups[g].host[k];
                        for (auto& kv : groups[g].keep) { if (kv.first >= groups[g].n_cols) return set_err(ctx, NX_ERR_ARG, "TreeBuilder: keep index outside the group"); keep_of[off + kv.first] = kv.second; }
                    }
                    off += groups[g].n_cols;
                }
            }
            HostFeed* feed = nullptr;
            if (any_host) { feeds.emplace_back(new HostFeed()); feed = feeds.back().get(); H_TRY(feed->begin(ctx, log, coset_order)); }
            // chunks of 16 columns when the run is fed from the host (a chunk's transforms start when IT has arrived; 16 columns of 2^22
            // rows are 5 ms of PCIe and 0.7 ms of transforms), else the whole run (or NX_PIPE_COLS groups) per call
            const uint32_t step = any_host ? 16u : (leaf ? G : n_run);
            for (uint32_t c0 = 0; c0 < n_run; c0 += step) {
                const uint32_t nb = std::min(step, n_run - c0);
                if (any_host) {
                    // contiguous sub-runs with a host source are uploaded; columns without one were filled by the caller
                    for (uint32_t a = c0; a < c0 + nb;) {
                        if (!hsrc[a]) { a++; continue; }
                        uint32_t b2 = a; while (b2 < c0 + nb && hsrc[b2]) b2++;
                        hipEvent_t ready = nullptr;
                        H_TRY(feed->chunk(hsrc.data() + a, in.data() + a, b2 - a, &ready));
                        NX_HIP(ctx, hipStreamWaitEvent(ctx->stream, ready, 0));
                        a = b2;
                    }
                    for (uint32_t a = c0; a < c0 + nb; a++) if (keep_of[a]) H_TRY(nx_copy(ctx, keep_of[a], in[a], (size_t)1 << log));       // Column::clone (R4)
                }
                if (is_evals) H_TRY(nx_lde_batch(ctx, cs.tw, in.data() + c0, nb, log, cs.cfg.log_blowup, out.data() + c0));   // K3 + K4
                else H_TRY(nx_evaluate_batch(ctx, cs.tw, (const uint32_t* const*)in.data() + c0, nb, log, cs.cfg.log_blowup, out.data() + c0));  // K4
                if (leaf) H_TRY(tree_pipe_absorb(ctx, &tp, (const uint32_t* const*)out.data() + c0, nb, false));                   // K5, leaf layer (16-column blocks as they complete)
            }
            for (uint32_t i = 0; i < n_run; i++) {
                t.polys.push_back({in[i], log, false}); t.evals.push_back({out[i], el, false}); t.owner.push_back(-1);
                if (!leaf) { small_cols.push_back(out[i]); small_logs.push_back(el); }
            }
        }
        for (size_t g = g0; g < g1; g++) t.bufs.push_back(std::move(groups[g].slab));
        t.bufs.push_back(std::move(lde));
        g0 = g1;
    }
    t.merkle.n_layers = max_el + 1;
    if (total_leaf_cols) H_TRY(tree_pipe_finish(ctx, &tp, small_cols.data(), small_logs.data(), (uint32_t)small_cols.size(), &t.merkle.local));   // K5, inner layers
    else H_TRY(nx_merkle_commit(ctx, nullptr, nullptr, 0, &t.merkle.local));
    tree_index = cs.trees.size();
    cs.trees.push_back(std::move(t));
    groups.clear();
    begun = true;
    return NX_OK;
}

// Column chunks of a row-sharded commit (option "dist.chunks" overrides; the same on every GPU: it depends on the run, not on the rank).
// Chunks pay only where the exchange is worth hiding: >= 4 columns per GPU and chunk, >= 2^27 words in the run.
static uint32_t dist_chunks(const nx_ctx* ctx, uint32_t n_run, int world, uint32_t el) {
    const uint32_t min_loc = n_run / (uint32_t)world;
    if (ctx->opt.dist_chunks > 0) return (uint32_t)std::max(1, std::min(ctx->opt.dist_chunks, (int)std::max<uint32_t>(1, min_loc)));
    if (((uint64_t)n_run << el) < ((uint64_t)1 << 27)) return 1;
    return std::max<uint32_t>(1, std::min<uint32_t>(4, min_loc / 4));
}

// One proof on W GPUs: every GPU extends its share of each run of equally sized columns (iFFT + FFT need whole columns), one
// all-to-all per run turns the column shards into row blocks, the Merkle subtree over the block is local, the W subtree roots are
// all-gathered and the top log2 W levels computed by everyone.  Coefficients stay with the GPU that transformed them (OODS sampling
// is per column).  Polynomials handed in by extend_polys are on every GPU already: everyone evaluates them and keeps its rows.
int TreeBuilder::commit_dist(Blake2sChannel& channel) {
    nx_ctx* ctx = cs.ctx;
    const Dist& D = cs.dist;
    const uint32_t blow = cs.cfg.log_blowup;
    CommitmentTreeProver t;
    std::vector<const uint32_t*> all_cols; std::vector<uint32_t> all_logs;
    for (size_t g0 = 0; g0 < groups.size();) {
        size_t g1 = g0 + 1;
        while (g1 < groups.size() && groups[g1].log == groups[g0].log && groups[g1].is_evals == groups[g0].is_evals) g1++;
        const uint32_t log = groups[g0].log, el = log + blow;
        const bool is_evals = groups[g0].is_evals;
        uint32_t n_run = 0;
        for (size_t g = g0; g < g1; g++) n_run += groups[g].n_cols;
        if (n_run && el < (uint32_t)D.log_w + 2) return set_err(ctx, NX_ERR_ARG, "row-sharded commit: every LDE column needs at least 4 rows per GPU");
        const uint64_t mb = D.block(el);
        if (n_run && !is_evals) {
            // replicated coefficients: the whole LDE everywhere, this GPU's rows are a window into it
            DevBuf lde; H_TRY(lde.alloc(ctx, (size_t)n_run << el));
            std::vector<uint32_t*> in;
            for (size_t g = g0; g < g1; g++) {
                if (groups[g].lo != 0 || groups[g].hi != groups[g].n_cols) return set_err(ctx, NX_ERR_ARG, "TreeBuilder::extend_polys: the polynomials must be on every GPU");
                auto p = col_ptrs(groups[g].slab.p, groups[g].n_cols, log); in.insert(in.end(), p.begin(), p.end());
            }
            auto out = col_ptrs(lde.p, n_run, el);
            H_TRY(nx_evaluate_batch(ctx, cs.tw, (const uint32_t* const*)in.data(), n_run, log, blow, out.data()));
            for (uint32_t i = 0; i < n_run; i++) {
                t.polys.push_back({in[i], log, false}); t.owner.push_back(-1);
                t.evals.push_back({out[i] + D.begin(el), el, true});
                all_cols.push_back(out[i] + D.begin(el)); all_logs.push_back(el);
            }
            for (size_t g = g0; g < g1; g++) t.bufs.push_back(std::move(groups[g].slab));
            t.bufs.push_back(std::move(lde));
        } else if (n_run) {
            // this GPU's range of the run, as the caller planned it (plan_local_columns)
            const uint32_t lo = Dist::cut(n_run, D.rank, D.world), hi = Dist::cut(n_run, D.rank + 1, D.world), n_loc = hi - lo;
            std::vector<uint32_t*> in;
            {
                uint32_t off = 0;
                for (size_t g = g0; g < g1; g++) {
                    const uint32_t a = std::max(lo, off), b = std::min(hi, off + groups[g].n_cols);
                    const uint32_t glo = a < b ? a - off : 0, ghi = a < b ? b - off : 0;
                    if (groups[g].lo != glo || groups[g].hi != ghi) return set_err(ctx, NX_ERR_ARG, "TreeBuilder: the column shard does not match plan_local_columns");
                    auto p = col_ptrs(groups[g].slab.p, ghi - glo, log); in.insert(in.end(), p.begin(), p.end());
                    off += groups[g].n_cols;
                }
            }
            // The share is extended and exchanged in Q column chunks (the same Q on every GPU): chunk q+1's LDE is enqueued before the
            // host blocks in chunk q's all-to-all, so the transforms run while the links carry the previous chunk.  The receive slab
            // is chunk-major, rank-major inside a chunk: every exchange is contiguous on both sides.
            const uint32_t Q = dist_chunks(ctx, n_run, D.world, el);
            auto chunk_cols = [&](int r, uint32_t q) { const uint32_t nl = Dist::cut(n_run, r + 1, D.world) - Dist::cut(n_run, r, D.world); return Dist::cut(nl, (int)q + 1, (int)Q) - Dist::cut(nl, (int)q, (int)Q); };
            std::vector<size_t> chunk_base(Q + 1, 0);        // in columns
            for (uint32_t q = 0; q < Q; q++) { size_t n = 0; for (int r = 0; r < D.world; r++) n += chunk_cols(r, q); chunk_base[q + 1] = chunk_base[q] + n; }
            DevBuf rows; H_TRY(rows.alloc(ctx, (size_t)n_run * mb));
            {
                struct Chunk { DevBuf send; hipEvent_t ev = nullptr; uint32_t n = 0; };
                std::vector<Chunk> ch(Q);
                struct EvGuard { std::vector<Chunk>& c; ~EvGuard() { for (auto& x : c) if (x.ev) (void)hipEventDestroy(x.ev); } } ev_guard{ch};
                auto exchange = [&](uint32_t q) -> int {
                    std::vector<size_t> soff(D.world), scnt(D.world), roff(D.world), rcnt(D.world);
                    size_t at = chunk_base[q];
                    for (int r = 0; r < D.world; r++) {
                        soff[r] = (size_t)r * ch[q].n * mb; scnt[r] = (size_t)ch[q].n * mb;
                        roff[r] = at * mb; rcnt[r] = (size_t)chunk_cols(r, q) * mb; at += chunk_cols(r, q);
                    }
                    H_TRY(D.alltoallv(ctx, ch[q].send.p, soff.data(), scnt.data(), rows.p, roff.data(), rcnt.data(), Q > 1 ? ch[q].ev : nullptr));   // the transposition
                    ch[q].send.release();
                    return NX_OK;
                };
                for (uint32_t q = 0; q < Q; q++) {
                    const uint32_t c0 = Dist::cut(n_loc, (int)q, (int)Q), c1 = Dist::cut(n_loc, (int)q + 1, (int)Q);
                    ch[q].n = c1 - c0;
                    if (ch[q].n) {
                        DevBuf lde; H_TRY(lde.alloc(ctx, (size_t)ch[q].n << el));
                        auto out = col_ptrs(lde.p, ch[q].n, el);
                        H_TRY(nx_lde_batch(ctx, cs.tw, in.data() + c0, ch[q].n, log, blow, out.data()));                            // K3 + K4, column-parallel
                        H_TRY(ch[q].send.alloc(ctx, (size_t)ch[q].n << el));
                        H_TRY(transpose_blocks(ctx, lde.p, (uint64_t)1 << el, ch[q].send.p, ch[q].n, (uint64_t)1 << el, (uint32_t)D.world, false));
                    }
                    if (Q > 1) { NX_HIP(ctx, hipEventCreateWithFlags(&ch[q].ev, hipEventDisableTiming)); NX_HIP(ctx, hipEventRecord(ch[q].ev, ctx->cur)); }
                    if (q > 0) H_TRY(exchange(q - 1));
                }
                H_TRY(exchange(Q - 1));
            }
            // where column c of the run landed: its owner's chunk, then its place inside the owner's part of that chunk
            std::vector<uint32_t*> blk_of(n_run);
            for (int r = 0; r < D.world; r++) {
                const uint32_t rlo = Dist::cut(n_run, r, D.world), nl = Dist::cut(n_run, r + 1, D.world) - rlo;
                for (uint32_t q = 0; q < Q; q++) {
                    size_t at = chunk_base[q];
                    for (int r2 = 0; r2 < r; r2++) at += chunk_cols(r2, q);
                    const uint32_t k0 = Dist::cut(nl, (int)q, (int)Q), k1 = Dist::cut(nl, (int)q + 1, (int)Q);
                    for (uint32_t k = k0; k < k1; k++) blk_of[rlo + k] = rows.p + (at + (k - k0)) * mb;
                }
            }
            for (uint32_t c = 0; c < n_run; c++) {
                const bool local = c >= lo && c < hi;
                int own = 0; while (!(c >= Dist::cut(n_run, own, D.world) && c < Dist::cut(n_run, own + 1, D.world))) own++;
                t.polys.push_back({local ? in[c - lo] : nullptr, log, false}); t.owner.push_back(own);
                uint32_t* blk = blk_of[c];
                t.evals.push_back({blk, el, true});
                all_cols.push_back(blk); all_logs.push_back(el);
            }
            for (size_t g = g0; g < g1; g++) t.bufs.push_back(std::move(groups[g].slab));
            t.bufs.push_back(std::move(rows));
        }
        g0 = g1;
    }
    H_TRY(merkle_commit_any(ctx, D, all_cols.data(), all_logs.data(), (uint32_t)all_cols.size(), &t.merkle));                        // K5: local subtree + W roots
    t.root = t.merkle.root;
    channel.mix_root(t.root);
    cs.trees.push_back(std::move(t));
    groups.clear();
    return NX_OK;
}

// ---------------------------------------------------------------- FRI ------------------------
struct FriLayer { SecureColumn eval; TreeRef merkle; Blake2sHash root; };
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

// all-gather a row-sharded secure column into a whole one (the FRI tail, the composition accumulator)
static int gather_secure(nx_ctx* ctx, const Dist& D, const SecureColumn& blk, SecureColumn* whole) {
    H_TRY(whole->alloc(ctx, blk.log));
    return D.allgather_cols(ctx, {blk.c[0], blk.c[1], blk.c[2], blk.c[3]}, (size_t)blk.rows, whole->c[0], (uint64_t)1 << blk.log);
}

class FriProver {
  public:
    nx_ctx* ctx; const nx_twiddles* tw; PcsConfig cfg; const Dist& D;
    std::vector<SecureColumn> columns;  // first layer, decreasing size (row blocks when the prove is row-sharded)
    TreeRef first_merkle; Blake2sHash first_root;
    std::vector<FriLayer> inner;
    std::vector<QM31> last_layer_poly;
    FriProver(nx_ctx* c, const nx_twiddles* t, PcsConfig f, const Dist& d) : ctx(c), tw(t), cfg(f), D(d) {}

    static int commit_secure(nx_ctx* ctx, const Dist& D, const std::vector<const SecureColumn*>& cols, TreeRef* tree, Blake2sHash* root) {
        std::vector<const uint32_t*> p; std::vector<uint32_t> logs;
        const Dist single;
        bool blocks = false;
        for (auto s : cols) { blocks = blocks || s->block; for (int k = 0; k < 4; k++) { p.push_back(s->c[k]); logs.push_back(s->log); } }
        H_TRY(merkle_commit_any(ctx, blocks ? D : single, p.data(), logs.data(), (uint32_t)p.size(), tree));
        *root = tree->root;
        return NX_OK;
    }

    // FriProver::commit.  Single GPU: the whole commit phase is enqueued without a host round trip — the Blake2s channel lives on the
    // device (fri_channel_step after each tree, the folds read their alpha from the channel's records, fri_tail for the small layers);
    // the host reads the roots and the channel state back once, before the last layer.  NX_FRI_DEVICE_CHANNEL=0: the per-layer host
    // channel (same transcript; kept for A/B).  Row-sharded: every layer's root needs the W subtree roots, so the host channel.
    int commit(Blake2sChannel& channel, std::vector<SecureColumn>&& cols) {
        if (D.on() || !ctx->opt.fri_device_channel) return commit_host_channel(channel, std::move(cols));
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
        const uint32_t fused_min = ctx->opt.merkle_fused ? (uint32_t)std::max(ctx->opt.merkle_fused, 12) : 64u;   // "merkle.fused": from this size a layer's fold, leaf hash and first 6 levels are one launch
        {   // first layer: every circle column in one mixed-degree tree
            std::vector<const uint32_t*> p; std::vector<uint32_t> logs;
            for (auto& c : columns) for (int k = 0; k < 4; k++) { p.push_back(c.c[k]); logs.push_back(c.log); }
            H_TRY(nx_merkle_commit(ctx, p.data(), logs.data(), (uint32_t)p.size(), &first_merkle.local));
            first_merkle.n_layers = columns[0].log + 1;
            H_TRY(fri_channel_step(ctx, d_state.p, first_merkle.local->layers[0], 0));
        }
        int j_prev = 0;                         // record whose alpha is the current folding alpha
        SecureColumn layer; H_TRY(layer.alloc(ctx, layer_log));
        H_TRY(nx_memset_zero(ctx, layer.buf.p, layer.buf.words));
        size_t ci = 0;
        const bool use_tail = ctx->opt.fri_tail != 0;
        const uint32_t tail_log = ctx->opt.fri_tail == 1 ? (uint32_t)FRI_TAIL_LOG : (uint32_t)std::min(ctx->opt.fri_tail, FRI_TAIL_LOG);   // "fri.tail": 1 = from 2^11 points, 2 .. 11 = from that size
        // `pending`: `layer` is allocated but not computed yet — it is fold_line(inner.back().eval, alpha of record j_prev), which the fused
        // launch of the layer's commit computes on its way (or fold_line_dev, where the layer is needed before its commit)
        bool pending = false;
        auto materialize = [&]() -> int {
            if (!pending) return NX_OK;
            pending = false;
            return fold_line_dev(ctx, tw, (const uint32_t* const*)inner.back().eval.c, layer_log + 1, rec_alpha(j_prev), layer.c);
        };
        while (layer_log > last_log) {
            const int j = (int)inner.size() + 1;   // record of the layer committed in this iteration
            if (use_tail && ci == columns.size() && layer_log <= tail_log && layer_log - last_log <= (uint32_t)FRI_TAIL_MAX_LAYERS) {
                H_TRY(materialize());
                const int n = (int)(layer_log - last_log);
                std::vector<FriLayer> tl(n);
                std::vector<uint32_t*> evals(n + 1), trees(n);
                tl[0].eval = std::move(layer);
                for (int t = 0; t < n; t++) {
                    if (t) H_TRY(tl[t].eval.alloc(ctx, layer_log - t));
                    H_TRY(tree_alloc(ctx, layer_log - t, &tl[t].merkle.local));
                    tl[t].merkle.n_layers = layer_log - t + 1;
                    evals[t] = tl[t].eval.buf.p; trees[t] = tl[t].merkle.local->layers[0];
                }
                SecureColumn fin; H_TRY(fin.alloc(ctx, last_log));
                evals[n] = fin.buf.p;
                H_TRY(fri_tail(ctx, tw, evals.data(), trees.data(), n, (int)layer_log, d_state.p, j));
                for (int t = 0; t < n; t++) inner.push_back(std::move(tl[t]));
                layer = std::move(fin);
                layer_log = last_log;
                break;
            }
            if (ci < columns.size() && columns[ci].log - 1 == layer_log) H_TRY(materialize());     // a circle column joins here: it folds INTO the layer
            while (ci < columns.size() && columns[ci].log - 1 == layer_log) {
                const uint32_t* a = rec_alpha(cfg.fri_alpha_mode == NX_FRI_ALPHA_PREV ? j_prev : 0);
                H_TRY(fold_circle_dev(ctx, tw, layer.c, (const uint32_t* const*)columns[ci].c, columns[ci].log, a));
                ci++;
            }
            FriLayer L; L.eval = std::move(layer);
            if (pending && layer_log >= fused_min) {
                pending = false;
                H_TRY(fri_fold_commit_fused(ctx, tw, (const uint32_t* const*)inner.back().eval.c, layer_log + 1, rec_alpha(j_prev), L.eval.c, &L.merkle.local));
            } else {
                if (pending) { pending = false; H_TRY(fold_line_dev(ctx, tw, (const uint32_t* const*)inner.back().eval.c, layer_log + 1, rec_alpha(j_prev), L.eval.c)); }
                std::vector<const uint32_t*> p; std::vector<uint32_t> logs;
                for (int k = 0; k < 4; k++) { p.push_back(L.eval.c[k]); logs.push_back(layer_log); }
                H_TRY(nx_merkle_commit(ctx, p.data(), logs.data(), 4, &L.merkle.local));      // ("merkle.fused": leaf hash + 6 levels in one launch from that size up)
            }
            H_TRY(fri_channel_step(ctx, d_state.p, L.merkle.local->layers[0], j));
            L.merkle.n_layers = layer_log + 1;
            SecureColumn next; H_TRY(next.alloc(ctx, layer_log - 1));
            inner.push_back(std::move(L));
            layer = std::move(next);               // computed by the next commit's launch (or materialize)
            pending = true;
            layer_log--; j_prev = j;
        }
        H_TRY(materialize());
        {   // the one host round trip of the commit phase: channel state and every layer's root
            std::vector<uint32_t> st(st_words);
            H_TRY(nx_download(ctx, st.data(), d_state.p, FRI_STATE_HEAD + FRI_STATE_REC * (1 + inner.size())));
            memcpy(channel.digest.w, st.data(), 32);
            channel.n_challenges += (uint32_t)(1 + inner.size()); channel.n_sent = st[8];
            memcpy(first_root.w, &st[FRI_STATE_HEAD], 32); first_merkle.root = first_root;
            for (size_t i = 0; i < inner.size(); i++) { memcpy(inner[i].root.w, &st[FRI_STATE_HEAD + FRI_STATE_REC * (i + 1)], 32); inner[i].merkle.root = inner[i].root; }
        }
        return commit_last_layer(channel, layer, layer_log, ci);
    }

    // The per-layer host channel.  Row-sharded: the first-layer columns and the line layers are row blocks — trees are subtree + W
    // roots, folds are local (pairs are adjacent) — while that pays: a sharded layer costs a host all-gather of the W subtree roots
    // (a collective plus two host round trips, ~0.2 ms) and saves (1 - 1/W) of the layer's hashing, which is 0.14 ms for 2^21 rows on
    // the whole chip.  Below 2^"fri.dist_min_log" rows (context option, default 21; the tests set 0), or 2^FRI_DIST_MIN_LOCAL rows per GPU, the
    // layer and the remaining circle columns are all-gathered and the tail runs replicated on every GPU.
    static constexpr uint32_t FRI_DIST_MIN_LOCAL = 10;
    int commit_host_channel(Blake2sChannel& channel, std::vector<SecureColumn>&& cols) {
        columns = std::move(cols);
        if (columns.empty()) return set_err(ctx, NX_ERR_ARG, "fri: no columns");
        { std::vector<const SecureColumn*> p; for (auto& c : columns) p.push_back(&c); H_TRY(commit_secure(ctx, D, p, &first_merkle, &first_root)); }
        channel.mix_root(first_root);
        QM31 folding_alpha = channel.draw_secure_felt();
        const QM31 first_alpha = folding_alpha;
        uint32_t layer_log = columns[0].log - 1;
        bool sharded = D.on();
        const uint32_t min_log = (uint32_t)ctx->opt.fri_dist_min_log;
        SecureColumn layer;
        if (sharded) H_TRY(layer.alloc_rows(ctx, layer_log, D.block(layer_log), true)); else H_TRY(layer.alloc(ctx, layer_log));
        H_TRY(nx_memset_zero(ctx, layer.buf.p, layer.buf.words));
        const uint32_t last_log = cfg.log_last_layer_degree_bound + cfg.log_blowup;
        size_t ci = 0; uint32_t n_doublings = 0;
        std::vector<SecureColumn> gathered;   // whole copies of the circle columns folded in after the switch to the replicated tail
        while (layer_log > last_log) {
            if (sharded && (layer_log < (uint32_t)D.log_w + FRI_DIST_MIN_LOCAL || layer_log < min_log)) {
                SecureColumn whole; H_TRY(gather_secure(ctx, D, layer, &whole));
                layer = std::move(whole);
                gathered.resize(columns.size());
                for (size_t k = ci; k < columns.size(); k++) H_TRY(gather_secure(ctx, D, columns[k], &gathered[k]));
                sharded = false;
            }
            while (ci < columns.size() && columns[ci].log - 1 == layer_log) {
                QM31 a = cfg.fri_alpha_mode == NX_FRI_ALPHA_PREV ? folding_alpha : first_alpha;
                uint32_t aw[4]; q_store(aw, a);
                if (sharded) H_TRY(fold_circle_rows(ctx, tw, layer.c, (const uint32_t* const*)columns[ci].c, columns[ci].log, aw, (uint32_t)D.begin(layer_log), (uint32_t)D.block(layer_log)));
                else {
                    const SecureColumn& src = columns[ci].block && D.on() ? gathered[ci] : columns[ci];
                    H_TRY(nx_fold_circle_into_line(ctx, tw, layer.c, (const uint32_t* const*)src.c, src.log, aw));
                }
                ci++;
            }
            FriLayer L; L.eval = std::move(layer);
            { std::vector<const SecureColumn*> p{&L.eval}; H_TRY(commit_secure(ctx, D, p, &L.merkle, &L.root)); }
            channel.mix_root(L.root);
            folding_alpha = channel.draw_secure_felt();
            SecureColumn next;
            uint32_t aw[4]; q_store(aw, folding_alpha);
            if (sharded) {
                H_TRY(next.alloc_rows(ctx, layer_log - 1, D.block(layer_log - 1), true));
                H_TRY(fold_line_rows(ctx, tw, (const uint32_t* const*)L.eval.c, layer_log, aw, next.c, (uint32_t)D.begin(layer_log - 1), (uint32_t)D.block(layer_log - 1)));
            } else {
                H_TRY(next.alloc(ctx, layer_log - 1));
                H_TRY(nx_fold_line(ctx, tw, (const uint32_t* const*)L.eval.c, layer_log, n_doublings, aw, next.c));
            }
            inner.push_back(std::move(L));
            layer = std::move(next);
            layer_log--; n_doublings++;
        }
        if (sharded) { SecureColumn whole; H_TRY(gather_secure(ctx, D, layer, &whole)); layer = std::move(whole); }
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
        // Stwo mixes `LinePoly::from_ordered_coefficients(coeffs)`, whose storage is the BIT-REVERSED coefficient vector
        // [upstream-recollection; ADVICE r2]: the transcript sees that order (identical for bounds 0 and 1), the proof keeps the ordered one
        { std::vector<QM31> m(bound); for (size_t i = 0; i < bound; i++) m[i] = v[bitrev((u32)i, (int)cfg.log_last_layer_degree_bound)]; channel.mix_felts(m); }
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
    static ColumnRef coord_ref(const SecureColumn& col, int k) { return ColumnRef{col.c[k], col.log, col.block}; }
    static SecurePlan plan_secure(const SecureColumn& col, const std::vector<size_t>& pos, GatherBatch* gb) {
        SecurePlan p{gb->ptrs.size(), pos.size()};
        for (size_t q : pos) for (int k = 0; k < 4; k++) gb->add_column(coord_ref(col, k), q);
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
                for (int k = 0; k < 4; k++) refs.push_back(coord_ref(c, k));
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
            std::vector<ColumnRef> refs; for (int k = 0; k < 4; k++) refs.push_back(coord_ref(L.eval, k));
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

// ---------------------------------------------------------------- composition polynomial helpers ----------
QPt get_random_point(Blake2sChannel& ch) {  // CirclePoint::get_random_point
    QM31 t = ch.draw_secure_felt(), t2 = q_sqr(t);
    QM31 inv = q_inv(q_add(t2, q_one()));
    QPt p; p.x = q_mul(q_sub(q_one(), t2), inv); p.y = q_mul(q_add(t, t), inv);
    return p;
}
QM31 coset_vanishing_q(uint32_t n, QPt p) { QM31 x = p.x; for (uint32_t i = 1; i < n; i++) x = q_double_x(x); return x; }

// 1 / coset_vanishing(trace coset, eval_domain.at(i)) over the 2^(e - log_size) cosets of the evaluation domain, bit-reversed
std::vector<uint32_t> vanishing_denominators(uint32_t log_size, uint32_t e) {
    const uint32_t log_expand = e - log_size;
    std::vector<uint32_t> den((size_t)1 << log_expand);
    for (uint32_t i = 0; i < den.size(); i++) {
        u32 x = pt_from_index(circle_domain_index((int)e, i)).x;
        for (uint32_t k = 1; k < log_size; k++) x = m_double_x(x);
        den[bitrev(i, (int)log_expand)] = m_inv(x);
    }
    return den;
}

// Row-sharded prove: every GPU evaluates the polynomials IT holds of the selected columns on the 2^out_log-point domain of `tw_sub`
// (nx_evaluate_batch, expansion out_log - log_size), and one all-to-all hands out row blocks: (*blk)[i] = this GPU's rows
// [D.begin(out_log), + D.block(out_log)) of column sel[i] (not biased).  Receive layout: by source GPU, the source's columns in `sel` order.
static int sharded_evaluate_exchange(CommitmentSchemeProver& cs, const std::vector<std::pair<uint32_t, uint32_t>>& comp_cols, const std::vector<size_t>& sel, const nx_twiddles* tw_sub,
                                     uint32_t log_size, uint32_t out_log, std::vector<uint32_t*>* blk, std::vector<DevBuf>* keep) {
    nx_ctx* ctx = cs.ctx;
    const Dist& D = cs.dist;
    const size_t ns = sel.size();
    blk->assign(ns, nullptr);
    if (out_log < (uint32_t)D.log_w + 2) return set_err(ctx, NX_ERR_ARG, "row-sharded prove: the constraint domain needs at least 4 rows per GPU");
    const uint64_t mb = D.block(out_log);
    std::vector<int> own(ns, -1); std::vector<uint32_t> pos(ns, 0); std::vector<uint32_t> cnt(D.world, 0);
    for (size_t i = 0; i < ns; i++) {
        own[i] = cs.trees[comp_cols[sel[i]].first].owner[comp_cols[sel[i]].second];
        if (own[i] < 0) return set_err(ctx, NX_ERR_ARG, "row-sharded prove: a trace column is not column-sharded");
        pos[i] = cnt[own[i]]++;
    }
    const uint32_t n_loc = cnt[D.rank];
    DevBuf ext, send, rows;
    H_TRY(rows.alloc(ctx, std::max<size_t>(ns, 1) * mb));
    if (n_loc) {
        H_TRY(ext.alloc(ctx, (size_t)n_loc << out_log));
        std::vector<const uint32_t*> src;
        for (size_t i = 0; i < ns; i++) if (own[i] == D.rank) src.push_back(cs.trees[comp_cols[sel[i]].first].polys[comp_cols[sel[i]].second].ptr);
        auto dst = col_ptrs(ext.p, n_loc, out_log);
        H_TRY(nx_evaluate_batch(ctx, tw_sub, src.data(), n_loc, log_size, out_log - log_size, dst.data()));
        H_TRY(send.alloc(ctx, (size_t)n_loc << out_log));
        H_TRY(transpose_blocks(ctx, ext.p, (uint64_t)1 << out_log, send.p, n_loc, (uint64_t)1 << out_log, (uint32_t)D.world, false));
    }
    std::vector<size_t> soff(D.world), scnt(D.world), roff(D.world), rcnt(D.world);
    size_t acc = 0;
    for (int r = 0; r < D.world; r++) { soff[r] = (size_t)r * n_loc * mb; scnt[r] = (size_t)n_loc * mb; roff[r] = acc; rcnt[r] = (size_t)cnt[r] * mb; acc += rcnt[r]; }
    H_TRY(D.alltoallv(ctx, send.p, soff.data(), scnt.data(), rows.p, roff.data(), rcnt.data()));
    for (size_t i = 0; i < ns; i++) (*blk)[i] = rows.p + roff[own[i]] + (size_t)pos[i] * mb;
    keep->push_back(std::move(rows));
    return NX_OK;
}

// `used` (optional): the component columns the kernel will read; the others get a NULL pointer and cost nothing (no re-evaluation, no
// exchange) — the high-degree part of a degree-split component reads a fraction of the columns.
int columns_on_eval_domain(CommitmentSchemeProver& cs, const std::vector<std::pair<uint32_t, uint32_t>>& comp_cols, uint32_t log_size, uint32_t e,
                           const std::vector<char>& masked, EvalDomainCols* out, const std::vector<char>* used) {
    nx_ctx* ctx = cs.ctx;
    const Dist& D = cs.dist;
    const size_t n = comp_cols.size();
    out->ptrs.assign(n, nullptr);
    std::vector<size_t> sel;                                                          // the columns to provide
    for (size_t k = 0; k < n; k++) if (!used || (k < used->size() && (*used)[k])) sel.push_back(k);
    const size_t ns = sel.size();
    const bool committed = e == log_size + cs.cfg.log_blowup;
    if (!D.on()) {
        if (committed) { for (size_t k : sel) out->ptrs[k] = cs.trees[comp_cols[k].first].evals[comp_cols[k].second].ptr; return NX_OK; }
        DevBuf ext; H_TRY(ext.alloc(ctx, std::max<size_t>(ns, 1) << e));               // "need_to_extend": re-evaluate the polynomials on the constraint domain
        std::vector<const uint32_t*> src(ns);
        for (size_t i = 0; i < ns; i++) src[i] = cs.trees[comp_cols[sel[i]].first].polys[comp_cols[sel[i]].second].ptr;
        auto dst = col_ptrs(ext.p, (uint32_t)ns, e);
        if (ns) H_TRY(nx_evaluate_batch(ctx, cs.tw, src.data(), (uint32_t)ns, log_size, e - log_size, dst.data()));
        for (size_t i = 0; i < ns; i++) out->ptrs[sel[i]] = dst[i];
        out->keep.push_back(std::move(ext));
        return NX_OK;
    }
    if (e < (uint32_t)D.log_w + 2) return set_err(ctx, NX_ERR_ARG, "row-sharded prove: the constraint domain needs at least 4 rows per GPU");
    const uint64_t mb = D.block(e), rb = D.begin(e);
    std::vector<uint32_t*> blk(n, nullptr);
    if (committed) {
        for (size_t k : sel) blk[k] = cs.trees[comp_cols[k].first].evals[comp_cols[k].second].ptr;
    } else {
        std::vector<uint32_t*> b2;
        H_TRY(sharded_evaluate_exchange(cs, comp_cols, sel, cs.tw, log_size, e, &b2, &out->keep));
        for (size_t i = 0; i < ns; i++) blk[sel[i]] = b2[i];
    }
    {   // columns read at a non-zero mask offset: neighbour rows live in other blocks, so the whole columns — one all-gather for all of them
        std::vector<const uint32_t*> mblk; std::vector<size_t> mk;
        for (size_t k : sel) { if (k < masked.size() && masked[k]) { mblk.push_back(blk[k]); mk.push_back(k); } else out->ptrs[k] = bias_rows(blk[k], rb); }
        if (!mblk.empty()) {
            DevBuf whole; H_TRY(whole.alloc(ctx, mblk.size() << e));
            H_TRY(D.allgather_cols(ctx, mblk, (size_t)mb, whole.p, (uint64_t)1 << e));
            for (size_t i = 0; i < mk.size(); i++) out->ptrs[mk[i]] = whole.p + (i << e);
            out->keep.push_back(std::move(whole));
        }
    }
    return NX_OK;
}

int composition_accumulator(CommitmentSchemeProver& cs, std::map<uint32_t, SecureColumn>& sub, uint32_t e, SecureColumn** out) {
    if (!sub.count(e)) {
        SecureColumn& s = sub[e];
        if (cs.dist.on()) H_TRY(s.alloc_rows(cs.ctx, e, cs.dist.block(e), true)); else H_TRY(s.alloc(cs.ctx, e));
        H_TRY(nx_memset_zero(cs.ctx, s.buf.p, s.buf.words));
    }
    *out = &sub[e];
    return NX_OK;
}

// DomainEvaluationAccumulator::finalize — ascending size: lift the running polynomial, add, interpolate.  Contributions that are
// already coefficients (`coef`) are added to that size's coefficients; a size that has only such a contribution takes the running
// polynomial in coefficient form too (zero-extension = the same polynomial: its coefficients are added to the low part).
int finalize_accumulation(CommitmentSchemeProver& cs, std::map<uint32_t, SecureColumn>& sub, DevBuf* out_polys, uint32_t* out_log, std::map<uint32_t, SecureColumn>* coef) {
    nx_ctx* ctx = cs.ctx;
    DevBuf cur; uint32_t cur_log = 0; bool have = false;
    std::set<uint32_t> logs;
    for (auto& kv : sub) logs.insert(kv.first);
    if (coef) for (auto& kv : *coef) logs.insert(kv.first);
    for (uint32_t log : logs) {
        const bool has_evals = sub.count(log) != 0;
        SecureColumn values;
        if (has_evals) {
            SecureColumn& sv = sub[log];
            if (sv.block) { SecureColumn whole; H_TRY(gather_secure(ctx, cs.dist, sv, &whole)); sv = std::move(whole); }
            values = std::move(sv);
            if (have) {
                DevBuf lifted; H_TRY(lifted.alloc(ctx, (size_t)4 << log));
                auto src = col_ptrs(cur.p, 4, cur_log), dst = col_ptrs(lifted.p, 4, log);
                H_TRY(nx_evaluate_batch(ctx, cs.tw, (const uint32_t* const*)src.data(), 4, cur_log, log - cur_log, dst.data()));
                const u32* s4[4] = {dst[0], dst[1], dst[2], dst[3]};
                H_TRY(secure_accumulate(ctx, values.c, s4, 1u << log));
                H_TRY(nx_sync(ctx));
            }
            H_TRY(nx_interpolate_batch(ctx, cs.tw, values.c, 4, log));
            if (coef && coef->count(log)) {
                SecureColumn& cv = (*coef)[log];
                const u32* s4[4] = {cv.c[0], cv.c[1], cv.c[2], cv.c[3]};
                H_TRY(secure_accumulate(ctx, values.c, s4, 1u << log));
                H_TRY(nx_sync(ctx));                                  // cv is released with the map
            }
        } else {
            values = std::move((*coef)[log]);
            if (have) {
                auto src = col_ptrs(cur.p, 4, cur_log);
                const u32* s4[4] = {src[0], src[1], src[2], src[3]};
                H_TRY(secure_accumulate(ctx, values.c, s4, 1u << cur_log));
                H_TRY(nx_sync(ctx));
            }
        }
        cur = std::move(values.buf); cur_log = log; have = true;
    }
    *out_polys = std::move(cur); *out_log = cur_log;
    return NX_OK;
}

bool host_prof_on() { static const bool on = [] { const char* e = getenv("NX_HOST_PROF"); return e && *e && *e != '0'; }(); return on; }
namespace { struct HostProf { std::mutex mu; std::vector<std::pair<std::string, std::pair<double, unsigned>>> rows; }; HostProf& host_prof() { static HostProf h; return h; } }
void host_prof_add(const char* name, double ms) {
    HostProf& h = host_prof();
    std::lock_guard<std::mutex> lk(h.mu);
    for (auto& r : h.rows) if (r.first == name) { r.second.first += ms; r.second.second++; return; }
    h.rows.push_back({name, {ms, 1u}});
}
void host_prof_dump(const char* title) {
    if (!host_prof_on()) return;
    HostProf& h = host_prof();
    std::lock_guard<std::mutex> lk(h.mu);
    fprintf(stderr, "[NX_HOST_PROF] %s\n", title);
    for (auto& r : h.rows) fprintf(stderr, "[NX_HOST_PROF]   %-44s %9.3f ms  x%u\n", r.first.c_str(), r.second.first, r.second.second);
    h.rows.clear();
}

void finish_stats(nx_ctx* ctx, nx_prove_stats* st, double t_start) {
    (void)nx_sync(ctx);
    st->total = now_ms() - t_start;
    timing_flush(ctx);
    st->lde_kernel_ms = ctx->kind_ms[NX_T_LDE]; st->lde_algorithmic_bytes = ctx->kind_bytes[NX_T_LDE];
    st->merkle_kernel_ms = ctx->kind_ms[NX_T_MERKLE]; st->merkle_algorithmic_bytes = ctx->kind_bytes[NX_T_MERKLE];
    ctx->timing = false;
}

// stwo::prover::prove from the point where the three trace trees are committed: composition polynomial, OODS sampling, DEEP
// quotients, FRI, proof of work, decommitment.
int prove_core(nx_ctx* ctx, CommitmentSchemeProver& cs, Blake2sChannel& channel, const PcsConfig& cfg, const nx_twiddles* tw, AirProver& air,
               std::vector<uint32_t>* words, nx_prove_stats* st, Lap& lap) {
    const Dist& D = cs.dist;
    // T trace trees are committed (the reference: 3 — preprocessed, main, interaction, machine.rs:208-263; Stwo takes any TreeVec and so
    // does this prover: the count is whatever the session committed); the composition polynomial's tree is number T
    const int T = (int)cs.trees.size();
    // ---------------- stwo::prover::prove ----------------
    QM31 random_coeff = channel.draw_secure_felt();
    DevBuf comp_polys; uint32_t clog = 0;
    { HostSpan hs("pc.compute_composition"); H_TRY(air.compute_composition(cs, random_coeff, &comp_polys, &clog)); }
    lap(&st->composition);
    { HostSpan hs("pc.composition tree commit (sync)"); TreeBuilder tb = cs.tree_builder(); tb.extend_polys(std::move(comp_polys), 4, clog); H_TRY(tb.commit(channel)); }
    lap(&st->commit);
    QPt oods = get_random_point(channel);

    MaskPoints points;                                                    // tree -> column -> points
    { HostSpan hs("pc.mask_points"); air.mask_points(oods, &points); }
    points.resize(T + 1);
    points[T].assign(4, std::vector<QPt>{oods});

    // ---------------- prove_values ----------------
    Proof proof;
    proof.sampled_values.resize(T + 1);
    {   // every tree and size is enqueued first; ONE synchronisation collects all the sampled values (K7).  Row-sharded: each GPU
        // samples the polynomials it holds; the values (KBs) are all-gathered.
        // One request per polynomial SIZE, across the trees (the columns of the three trace trees of a component share size and point: one table
        // build and one sweep launch instead of one per tree — the launches of a small statement are latency, not bytes).
        struct Where { int t; uint32_t c, s; };
        struct Pending { std::vector<uint32_t> out; std::vector<Where> where; };
        HostSpan hs_enq("pc.oods enqueue");
        std::vector<Pending> pend;
        std::vector<EvalJob> jobs;
        std::map<uint32_t, std::vector<std::pair<int, uint32_t>>> by_log;   // poly log -> (tree, column) held by this GPU, in tree / column order
        for (int t = 0; t <= T; t++) {
            auto& tr = cs.trees[t];
            proof.sampled_values[t].resize(tr.polys.size());
            for (uint32_t c = 0; c < tr.polys.size(); c++) {
                proof.sampled_values[t][c].assign(points[t][c].size(), q_zero());
                if (tr.polys[c].ptr) by_log[tr.polys[c].log].push_back({t, c});
            }
        }
        pend.reserve(by_log.size());                               // `out` buffers must not move while jobs point into them
        for (auto& kv : by_log) {
            std::vector<const uint32_t*> pp; std::vector<uint32_t> pidx, pts;
            pend.emplace_back(); Pending& pd = pend.back();
            pp.reserve(kv.second.size()); pidx.reserve(kv.second.size()); pts.reserve(8 * kv.second.size()); pd.where.reserve(kv.second.size());
            for (uint32_t li = 0; li < kv.second.size(); li++) {
                const int t = kv.second[li].first; const uint32_t c = kv.second[li].second;
                pp.push_back(cs.trees[t].polys[c].ptr);
                for (uint32_t sidx = 0; sidx < points[t][c].size(); sidx++) {
                    pidx.push_back(li);
                    uint32_t w[8]; q_store(w, points[t][c][sidx].x); q_store(w + 4, points[t][c][sidx].y);
                    pts.insert(pts.end(), w, w + 8);
                    pd.where.push_back({t, c, sidx});
                }
            }
            pd.out.assign(4 * pidx.size(), 0);
            H_TRY(eval_at_points_enqueue(ctx, pp.data(), kv.first, pidx.data(), pts.data(), (uint32_t)pidx.size(), pd.out.data(), &jobs));
        }
        hs_enq.stop();
        { HostSpan hs("pc.oods collect (sync)"); H_TRY(eval_at_points_collect(ctx, &jobs)); }
        HostSpan hs_scatter("pc.oods scatter");
        for (auto& pd : pend)
            for (size_t i = 0; i < pd.where.size(); i++) proof.sampled_values[pd.where[i].t][pd.where[i].c][pd.where[i].s] = q_load(&pd.out[4 * i]);
        if (D.on()) {
            std::vector<uint32_t> mine;
            for (int t = 0; t <= T; t++) for (auto& col : proof.sampled_values[t]) for (auto& v : col) { uint32_t w[4]; q_store(w, v); mine.insert(mine.end(), w, w + 4); }
            std::vector<uint32_t> everyone(mine.size() * (size_t)D.world);
            H_TRY(D.allgather_host(ctx, mine.data(), mine.size() * 4, everyone.data()));
            size_t slot = 0;
            for (int t = 0; t <= T; t++) for (uint32_t c = 0; c < proof.sampled_values[t].size(); c++) {
                const int r = cs.trees[t].owner[c];
                for (auto& v : proof.sampled_values[t][c]) { if (r >= 0) v = q_load(&everyone[(size_t)r * mine.size() + 4 * slot]); slot++; }
            }
        }
    }
    { HostSpan hs("pc.oods mix_felts"); std::vector<QM31> flat; for (auto& t : proof.sampled_values) for (auto& c : t) for (auto& v : c) flat.push_back(v); channel.mix_felts(flat); }
    auto check_oods = [&]() -> int {
        // ProvingError::ConstraintsNotSatisfied: the composition polynomial at the OODS point against the constraints over the sampled
        // values.  Stwo checks it at the end of prove(); here once the DEEP quotients are QUEUED (same inputs, same verdict): the host
        // interprets every component's program at the point while the GPU computes the quotients (0.1 ms for the headline's AIR, 0.8 ms
        // for a keccak-shaped one — GPU-idle time when it ran between the sampling and the quotients), and a trace that violates its
        // constraints is still refused BEFORE FRI, the proof of work and the decommitment are paid for — under every evaluation
        // strategy: with "air.half_domain" / "air.quarter_domain" the composition is low-degree by construction even for an invalid
        // trace (Q0 + t Z is what the interpolation returns), so this equality — not the FRI degree check — is what catches it.
        HostSpan hs("pc.oods host check (interpreter)");
        QM31 ce[4]; for (int k = 0; k < 4; k++) ce[k] = proof.sampled_values[T][k][0];
        QM31 lhs = q_add(q_add(ce[0], q_mul(ce[1], qm(0, 1, 0, 0))), q_add(q_mul(ce[2], qm(0, 0, 1, 0)), q_mul(ce[3], qm(0, 0, 0, 1))));
        if (!q_eq(lhs, air.eval_composition_at_point(oods, proof.sampled_values, random_coeff))) {
            ctx->symmetric_failure = true;       // every rank holds the same all-gathered sampled values and reaches this verdict by itself
            return set_err(ctx, NX_ERR_PROTOCOL, "ProvingError::ConstraintsNotSatisfied (composition OODS mismatch)");
        }
        return NX_OK;
    };
    lap(&st->oods);
    QM31 q_coeff = channel.draw_secure_felt();
    HostSpan hs_q("pc.quotients (host grouping + launches)");
    // compute_fri_quotients: all columns flattened, stable-sorted by LDE size (descending), grouped by size
    struct Flat { const uint32_t* ptr; uint32_t log; int t; uint32_t c; };
    std::vector<Flat> all;
    for (int t = 0; t <= T; t++) for (uint32_t c = 0; c < cs.trees[t].evals.size(); c++) all.push_back({cs.trees[t].evals[c].ptr, cs.trees[t].evals[c].log, t, c});
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
        const uint32_t L = all[i].log;
        SecureColumn qc;
        if (D.on()) H_TRY(qc.alloc_rows(ctx, L, D.block(L), true)); else H_TRY(qc.alloc(ctx, L));
        uint32_t aw[4]; q_store(aw, q_coeff);
        // A wide group on one GPU goes through the coefficients (pcs.hip accumulate_quotients_coeffs: the numerators are linear in the
        // columns): half the bytes of the row-wise sum at blowup 2, worth it from ~48 columns per sample point (the combinations cost an
        // extension of 4 columns per point).  Row-sharded proves keep the row-wise path: their coefficients are column-sharded.
        bool via_coeffs = !D.on() && ctx->opt.quotients_coeffs && L >= cfg.log_blowup + 2 && gcols.size() >= 48 * bpts.size();
        std::vector<const uint32_t*> gpolys;
        if (via_coeffs) {
            for (size_t k = i; k < j && via_coeffs; k++) {
                const auto& pl = cs.trees[all[k].t].polys[all[k].c];
                if (!pl.ptr || pl.log + cfg.log_blowup != L) via_coeffs = false; else gpolys.push_back(pl.ptr);
            }
        }
        if (via_coeffs)
            H_TRY(accumulate_quotients_coeffs(ctx, tw, L, L - cfg.log_blowup, gpolys.data(), (uint32_t)gpolys.size(), aw, (uint32_t)bpts.size(), fpts.data(), counts.data(), cidx.data(),
                                              vals.data(), qc.c));                                                                     // K8
        else
            H_TRY(accumulate_quotients_rows(ctx, L, gcols.data(), (uint32_t)gcols.size(), aw, (uint32_t)bpts.size(), fpts.data(), counts.data(), cidx.data(), vals.data(), qc.c,
                                            D.on() ? D.begin(L) : 0, qc.rows));                                                        // K8
        quotients.push_back(std::move(qc));
        i = j;
    }
    hs_q.stop();
    H_TRY(check_oods());
    lap(&st->quotients);
    FriProver fri(ctx, tw, cfg, D);
    { HostSpan hs("pc.fri.commit (sync)"); H_TRY(fri.commit(channel, std::move(quotients))); }                                                                                 // K9
    lap(&st->fri);
    { HostSpan hs("pc.grind (sync)"); H_TRY(nx_grind(ctx, (const uint8_t*)channel.digest.w, cfg.pow_bits, &proof.proof_of_work)); }                                       // K10
    channel.mix_u64(proof.proof_of_work);
    lap(&st->pow);
    std::map<uint32_t, std::vector<size_t>> qpos;
    GatherBatch gb; gb.dist = &D;
    std::vector<DecommitPlan> tree_plans(T + 1);
    {
        HostSpan hs("pc.decommit plan");
        fri.draw_queries(channel, &qpos);
        fri.decommit_plan(&gb);
        for (int t = 0; t <= T; t++) tree_plans[t] = merkle_decommit_plan(cs.trees[t].merkle, qpos, cs.trees[t].evals, &gb);
    }
    { HostSpan hs("pc.decommit gather (sync)"); H_TRY(gb.run(ctx)); }
    HostSpan hs_fill("pc.decommit fill");
    fri.decommit_fill(gb, &proof);
    proof.decommitments.resize(T + 1); proof.queried_values.resize(T + 1);
    for (int t = 0; t <= T; t++) {
        merkle_decommit_fill(tree_plans[t], gb, &proof.queried_values[t], &proof.decommitments[t]);
        proof.commitments.push_back(cs.trees[t].root);
    }
    lap(&st->decommit);
    { HostSpan hs("pc.serialize"); *words = serialize(proof, cfg); }
    return NX_OK;
}

// ================================================================ recorded AIRs: the nx_prover session ================
// The generic counterpart of the synthetic machines: components are RECORDED constraint programs (include/nexus_hip.h,
// nx_air_component — what FrameworkComponent<E> is to Stwo, reference prover/src/components/mod.rs:15-57), the caller drives the
// transcript prefix (reference machine.rs:198-263) through the session and nx_prover_prove runs stwo::prover::prove on the device.

// the consistency rules of oracle-side gair_check: every committed column claimed, sizes agree, loads inside the masks
int GenericAir::check(const CommitmentSchemeProver& cs) {
    if (cs.trees.empty()) return set_err(ctx, NX_ERR_ARG, "nx_prover_prove: at least one trace tree must be committed first");
    const int NT = (int)cs.trees.size();                 // the reference commits 3 (preprocessed, main, interaction); the count is the session's
    if (comps.empty()) return set_err(ctx, NX_ERR_ARG, "nx_prover_prove: no components");
    tree_logs.assign(NT, {}); offs.assign(NT, {});
    std::vector<std::vector<char>> claimed(NT);
    for (int t = 0; t < NT; t++) { for (auto& c : cs.trees[t].polys) tree_logs[t].push_back(c.log); claimed[t].assign(tree_logs[t].size(), 0); offs[t].resize(tree_logs[t].size()); }
    for (auto& c : comps) {
        if (c.masks.size() != c.cols.size()) return set_err(ctx, NX_ERR_ARG, "nx_prover_prove: one mask list per component column required");
        for (size_t k = 0; k < c.cols.size(); k++) {
            const uint32_t t = c.cols[k].first, i = c.cols[k].second;
            if (t >= (uint32_t)NT || i >= tree_logs[t].size()) return set_err(ctx, NX_ERR_ARG, "nx_prover_prove: component column outside the committed trees");
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
    for (int t = 0; t < NT; t++) for (char x : claimed[t]) if (!x) return set_err(ctx, NX_ERR_ARG, "nx_prover_prove: a committed column is claimed by no component");
    return NX_OK;
}

static void release_split_cache(nx_ctx* ctx);
struct KernelCache { std::mutex mu; std::map<std::pair<nx_ctx*, std::string>, nx_air_kernel*> map; };
static KernelCache& kernel_cache() { static KernelCache c; return c; }
int cached_air_kernel(nx_ctx* ctx, const GComponent& g, const uint8_t* select, const nx_air_kernel** out) {
    std::string key((const char*)g.prog.data(), g.prog.size() * sizeof(nx_cinstr));
    key += "|" + std::to_string(g.cols.size()) + "|" + std::to_string(g.n_regs) + "|" + std::to_string(g.econsts.size() / 4);
    if (select) { key += "|"; key.append((const char*)select, g.n_constraints); }
    KernelCache& kc = kernel_cache();
    {
        std::lock_guard<std::mutex> lk(kc.mu);
        auto it = kc.map.find({ctx, key});
        if (it != kc.map.end()) { *out = it->second; return NX_OK; }
    }
    // compile OUTSIDE the lock: entries are per context and a context is driven by one thread, so nobody else can insert this key;
    // holding the process-wide mutex across hiprtc would serialise the GPUs of a thread-rank group and let one stuck compile stall all
    nx_air_kernel* k = nullptr;
    H_TRY(nx_air_compile_subset(ctx, g.prog.data(), (uint32_t)g.prog.size(), g.n_regs, (uint32_t)g.cols.size(), (uint32_t)g.econsts.size() / 4, g.n_constraints, select, &k, nullptr));
    std::lock_guard<std::mutex> lk(kc.mu);
    kc.map.insert({{ctx, key}, k});
    *out = k;
    return NX_OK;
}
void machine_kernels_release(nx_ctx* ctx) {   // nx_ctx_destroy: the modules belong to the context's device
    KernelCache& kc = kernel_cache();
    std::lock_guard<std::mutex> lk(kc.mu);
    for (auto it = kc.map.begin(); it != kc.map.end();) { if (it->first.first == ctx) { nx_air_kernel_destroy(it->second); it = kc.map.erase(it); } else ++it; }
    release_split_cache(ctx);
    (void)nx_sync(ctx);
    ctx->machine_pre_cache.clear();          // the kept preprocessed trees: their buffers go back to the context's allocator, which is still alive here
}

// A constraint of degree d <= 3 over columns of 2^n rows has its quotient in the FFT space of the 2^(n+1)-point domain (d <= 2^e + 1
// with e = 1): those constraints are evaluated there — on the committed evaluations when the blowup is 2 — and only the columns the
// remaining constraints read are re-extended to log_size + bound.  finalize_accumulation lifts the small accumulator exactly like a
// smaller component's, so the composition polynomial is the same, coefficient for coefficient.
struct SplitCache { std::mutex mu; std::map<std::pair<nx_ctx*, std::string>, std::vector<GComponent::Part>> map; };
static SplitCache& split_cache() { static SplitCache c; return c; }
static void release_split_cache(nx_ctx* ctx) {
    SplitCache& sc = split_cache();
    std::lock_guard<std::mutex> lk(sc.mu);
    for (auto it = sc.map.begin(); it != sc.map.end();) { if (it->first.first == ctx) it = sc.map.erase(it); else ++it; }
}
// The plan of a component: which constraints are evaluated where (analysis and kernels depend on the program and on three switches
// only: once per context and AIR).
//  * degree <= 2 and the log_size + 1 domain is the committed one (blowup 2), one GPU: on the FIRST HALF of that domain ("air.half_domain");
//  * degree <= 3 of a component whose bound exceeds 1: on the log_size + 1 domain ("air.degree_split"), unless the component's own
//    domain is the committed one;
//  * the rest (or everything): on the component's own domain.
int prepare_component_kernels(nx_ctx* ctx, const PcsConfig& cfg, GComponent& g, bool sharded) {
    if (g.prepared) return NX_OK;
    const uint32_t bound = comp_log_cd(g.log_cd, cfg);
    const bool can_half = ctx->opt.air_half_domain && !sharded && cfg.log_blowup == 1 && g.log_size >= 4 && g.n_constraints > 0;
    const bool can_low = ctx->opt.air_degree_split && bound > 1 && bound != cfg.log_blowup && g.n_constraints > 0;
    const bool can_quarter = ctx->opt.air_quarter_domain && cfg.log_blowup == 1 && bound == 2 && g.log_size >= 4 && g.n_constraints > 0;   // also row-sharded (round 6): the quarter's columns are an N-point transform on their owner and an N-row exchange instead of 4N / 4N
    if (!can_half && !can_low && !can_quarter) {
        GComponent::Part whole; whole.whole = true; whole.where = GComponent::ON_FULL;
        if (g.kernel) whole.kernel = g.kernel; else H_TRY(cached_air_kernel(ctx, g, nullptr, &whole.kernel));
        g.parts.clear(); g.parts.push_back(std::move(whole));
        g.prepared = true;
        return NX_OK;
    }
    std::string key((const char*)g.prog.data(), g.prog.size() * sizeof(nx_cinstr));
    key += "|" + std::to_string(g.cols.size()) + "|" + std::to_string(g.n_regs) + "|" + std::to_string(g.econsts.size() / 4) + "|" + std::to_string(bound) + (can_half ? "|h" : "|-") + (can_low ? "l" : "-") + (can_quarter ? (ctx->opt.air_quarter_domain >= 2 ? "Q" : "q") : "-");
    SplitCache& sc = split_cache();
    {
        std::lock_guard<std::mutex> lk(sc.mu);
        auto it = sc.map.find({ctx, key});
        if (it != sc.map.end()) { g.parts = it->second; g.prepared = true; return NX_OK; }
    }
    std::vector<uint32_t> deg;
    air_constraint_degrees(g.prog.data(), (uint32_t)g.prog.size(), g.n_regs, &deg);
    if (deg.size() != g.n_constraints) return set_err(ctx, NX_ERR_ARG, "recorded AIR: constraint count mismatch");
    // class of every constraint; a class nobody may take falls back to the next larger domain
    int where_of[3];          // by degree class: <= 2, == 3, >= 4
    where_of[2] = GComponent::ON_FULL;
    where_of[1] = can_low ? GComponent::ON_LOW : GComponent::ON_FULL;
    where_of[0] = can_half ? GComponent::ON_HALF : where_of[1];
    std::vector<int> where_c(g.n_constraints);
    for (uint32_t j = 0; j < g.n_constraints; j++) where_c[j] = where_of[deg[j] <= 2 ? 0 : deg[j] == 3 ? 1 : 2];
    if (can_quarter) {
        // degree 4 / 5 (d <= 2^2 + 1: the quotient has 3N + 1 coefficients)
        std::vector<char> nb;
        air_constraint_neighbours(g.prog.data(), (uint32_t)g.prog.size(), g.n_regs, &nb);
        // "air.quarter_domain" = 2 (the default): neighbour-reading constraints too — their columns get a 2N-point transform (compute_composition)
        const bool with_neighbours = ctx->opt.air_quarter_domain >= 2;
        for (uint32_t j = 0; j < g.n_constraints; j++) if (where_c[j] == GComponent::ON_FULL && deg[j] >= 4 && deg[j] <= 5 && (with_neighbours || !nb[j])) where_c[j] = GComponent::ON_QUARTER;
    }
    std::vector<GComponent::Part> parts;
    for (int where : {GComponent::ON_HALF, GComponent::ON_LOW, GComponent::ON_QUARTER, GComponent::ON_FULL}) {
        GComponent::Part part; part.where = where; part.select.assign(g.n_constraints, 0);
        bool any = false;
        for (uint32_t j = 0; j < g.n_constraints; j++) if (where_c[j] == where) { part.select[j] = 1; any = true; }
        if (any) parts.push_back(std::move(part));
    }
    if (parts.size() == 1 && parts[0].where == GComponent::ON_FULL) {        // nothing to move: the component is evaluated whole
        parts[0].whole = true; parts[0].select.clear();
        if (g.kernel) parts[0].kernel = g.kernel; else H_TRY(cached_air_kernel(ctx, g, nullptr, &parts[0].kernel));
    } else {
        for (auto& part : parts) {
            air_subset_columns(g.prog.data(), (uint32_t)g.prog.size(), g.n_regs, (uint32_t)g.cols.size(), part.select.data(), &part.used);
            bool all = true; for (uint8_t x : part.select) all = all && x;
            H_TRY(cached_air_kernel(ctx, g, all ? nullptr : part.select.data(), &part.kernel));
        }
    }
    g.parts = parts;
    {
        std::lock_guard<std::mutex> lk(sc.mu);
        sc.map.insert({{ctx, key}, parts});
    }
    g.prepared = true;
    return NX_OK;
}

// acc[k][0] -= t[k] z0, acc[k][n] = t[k]: the half-domain interpolant becomes the quotient's coefficients (compute_composition)
__global__ void half_fix_kernel(u32* c0, u32* c1, u32* c2, u32* c3, u32 n, u32 t0, u32 t1, u32 t2, u32 t3, u32 z0) {
    if (threadIdx.x || blockIdx.x) return;
    c0[0] = m_sub(c0[0], m_mul(t0, z0)); c1[0] = m_sub(c1[0], m_mul(t1, z0)); c2[0] = m_sub(c2[0], m_mul(t2, z0)); c3[0] = m_sub(c3[0], m_mul(t3, z0));
    c0[n] = t0; c1[n] = t1; c2[n] = t2; c3[n] = t3;
}

// The quotient of constraints of degree <= 2 over columns of N = 2^n rows is Q0 + t Z with Q0 in the FFT space of N points, Z the
// vanishing polynomial of the trace domain (the one basis function of the 2N-point space beyond the N-point one that a product of two
// trace polynomials divided by Z can reach) and t a secure-field scalar.  Z is CONSTANT (z0) on the first half of the bit-reversed
// 2N-point domain, and that half is itself an N-point circle domain (twiddles_first_half).  So the constraints are evaluated on the
// first N rows of the committed columns only — half the bytes —, interpolated there (= Q0 + t z0), and t comes from one further row
// w of the second half: t = (Q(w) - I(w)) / (Z(w) - z0).  The 2N coefficients [I - t z0 | t, 0 ...] are those of the plain
// evaluation on all 2N rows, exactly.
struct HalfGroup { SecureColumn acc; uint32_t n_extra = 0; };     // acc: N + n_extra rows per coordinate — the half and a few rows of the second half, one launch
static int half_group_finish(nx_ctx* ctx, CommitmentSchemeProver& cs, uint32_t n, HalfGroup& hg, SecureColumn* out_coef) {
    const uint32_t N = 1u << n;
    nx_twiddles* sub_tw = nullptr;
    H_TRY(twiddles_first_half(ctx, cs.tw, n, &sub_tw));
    struct Guard { nx_twiddles* t; ~Guard() { nx_twiddles_destroy(t); } } guard{sub_tw};
    H_TRY(nx_interpolate_batch(ctx, sub_tw, hg.acc.c, 4, n));
    // I(w) and Q(w) for w = row N of the 2N-point domain
    const Pt w = pt_from_index(circle_domain_index((int)n + 1, bitrev(N, (int)n + 1)));
    uint32_t pts[32], idx[4] = {0, 1, 2, 3}, iw[16], qw[4];
    for (int k = 0; k < 4; k++) { QPt qp; qp.x = q_from_m(w.x); qp.y = q_from_m(w.y); q_store(pts + 8 * k, qp.x); q_store(pts + 8 * k + 4, qp.y); }
    const uint32_t* polys[4] = {hg.acc.c[0], hg.acc.c[1], hg.acc.c[2], hg.acc.c[3]};
    H_TRY(nx_eval_at_points(ctx, polys, n, idx, pts, 4, iw));
    {   // Q(w): row N of the four coordinates, one gather and one synchronisation
        const uint32_t* gp[4] = {hg.acc.c[0], hg.acc.c[1], hg.acc.c[2], hg.acc.c[3]};
        const uint64_t gi[4] = {N, N, N, N};
        H_TRY(nx_gather(ctx, gp, gi, 4, qw));
    }
    const std::vector<uint32_t> den = vanishing_denominators(n, n + 1);      // 1/Z on the two halves
    const u32 z0 = m_inv(den[0]), z1 = m_inv(den[1]);
    const u32 dz = m_inv(m_sub(z1, z0));
    u32 t[4];
    for (int k = 0; k < 4; k++) t[k] = m_mul(m_sub(qw[k], iw[4 * k]), dz);
    H_TRY(out_coef->alloc(ctx, n + 1));
    H_TRY(nx_memset_zero(ctx, out_coef->buf.p, out_coef->buf.words));
    for (int k = 0; k < 4; k++) H_TRY(nx_copy(ctx, out_coef->c[k], hg.acc.c[k], N));
    hipLaunchKernelGGL(half_fix_kernel, dim3(1), dim3(64), 0, ctx->stream, out_coef->c[0], out_coef->c[1], out_coef->c[2], out_coef->c[3], N, t[0], t[1], t[2], t[3], z0);
    if (hipGetLastError() != hipSuccess) return set_err(ctx, NX_ERR_HIP, "half_fix_kernel launch failed");
    return NX_OK;
}

// out[i] = (a[i] + s1 b[i]) s2
__global__ void axpy_scale_kernel(u32* __restrict__ out, const u32* __restrict__ a, const u32* __restrict__ b, u32 s1, u32 s2, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = m_mul(m_add(a[i], m_mul(s1, b[i])), s2);
}
static int axpy_scale(nx_ctx* ctx, u32* out, const u32* a, const u32* b, u32 s1, u32 s2, u32 n) {
    hipLaunchKernelGGL(axpy_scale_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, out, a, b, s1, s2, n);
    if (hipGetLastError() != hipSuccess) return set_err(ctx, NX_ERR_HIP, "axpy_scale_kernel launch failed");
    return NX_OK;
}

// The same decomposition one level up (tests/test_exact_algebra_cpu.py::
// test_degree_four_quotient_from_3n_plus_1_samples states it on the oracle).  The quotient F of constraints of degree 4 / 5 over columns
// of N rows lives in the 4N-point space with 3N + 1 coefficients:  F = F0 + Z2 (F10 + t Z),  F0 in the 2N-point space, Z2 the vanishing
// polynomial of the committed 2N-point domain, F10 in the N-point space, Z the trace domain's vanishing polynomial, t a secure scalar.
//  * Z2 = 0 on the committed domain: evaluating the constraints on the COMMITTED 2N rows (no extension at all) and interpolating gives F0;
//  * the first QUARTER of the bit-reversed 4N-point domain is an N-point circle domain (twiddles_first_part, depth 2) on which Z2 and Z
//    are constant (z2, z0): the columns are evaluated there by an N-point transform each instead of a 4N-point one, the constraints
//    on N rows, and (F - F0) / z2 interpolated there is F10 + t z0;
//  * one further row w (row N, second quarter: Z = z1) gives t = (G(w) - I(w)) / (z1 - z0); the columns' values at w are one
//    eval_at_point sweep over their coefficients.
// 3N + 1 constraint evaluations instead of 4N, and per column an N-point transform + one sweep instead of a 4N-point transform.  The
// 4N coefficients [F0 | I - t z0 | t, 0 ...] are those of the plain evaluation on all 4N rows, exactly; they enter
// finalize_accumulation as a coefficient-form contribution.  Only for constraints that read no neighbour row (the quarter holds none).
struct QuarterGroup { SecureColumn acc2n, accq, rown; };     // accq: N + 4 rows per coordinate (rows [0, N) of the quarter and row N).  Row-sharded: this GPU's blocks of the 2N / N rows; rown: row N (every GPU computes it)
static int quarter_group_finish(nx_ctx* ctx, CommitmentSchemeProver& cs, uint32_t n, QuarterGroup& qg, SecureColumn* out_coef) {
    const uint32_t N = 1u << n;
    nx_twiddles* qtw = nullptr;
    H_TRY(twiddles_first_part(ctx, cs.tw, n, 2, &qtw));
    struct Guard { nx_twiddles* t; ~Guard() { nx_twiddles_destroy(t); } } guard{qtw};
    H_TRY(nx_interpolate_batch(ctx, cs.tw, qg.acc2n.c, 4, n + 1));                          // F0: 2N coefficients per coordinate
    const std::vector<uint32_t> d4 = vanishing_denominators(n, n + 2), d2 = vanishing_denominators(n + 1, n + 2);
    const u32 z0 = m_inv(d4[0]), z1 = m_inv(d4[1]), inv_z2 = d2[0];
    // F0 on the quarter: F0 = F0_lo + Z F0_hi with Z = z0 there
    SecureColumn fq, f0q, g;
    H_TRY(fq.alloc(ctx, n)); H_TRY(f0q.alloc(ctx, n)); H_TRY(g.alloc(ctx, n));
    for (int k = 0; k < 4; k++) H_TRY(axpy_scale(ctx, fq.c[k], qg.acc2n.c[k], qg.acc2n.c[k] + N, z0, 1, N));
    { const uint32_t* src[4] = {fq.c[0], fq.c[1], fq.c[2], fq.c[3]}; H_TRY(nx_evaluate_batch(ctx, qtw, src, 4, n, 0, f0q.c)); }
    for (int k = 0; k < 4; k++) H_TRY(axpy_scale(ctx, g.c[k], qg.accq.c[k], f0q.c[k], P - 1, inv_z2, N));      // G = (F - F0) / z2 on the quarter
    H_TRY(nx_interpolate_batch(ctx, qtw, g.c, 4, n));                                        // I = F10 + t z0
    // the scalars at w = row N of the 4N-point domain
    const Pt w = pt_from_index(circle_domain_index((int)n + 2, bitrev(N, (int)n + 2)));
    uint32_t pts[12 * 8], idx[12], ev[12 * 4], fw[4];
    const uint32_t* polys[12];
    for (int k = 0; k < 4; k++) { polys[k] = qg.acc2n.c[k]; polys[4 + k] = qg.acc2n.c[k] + N; polys[8 + k] = g.c[k]; }
    for (int i = 0; i < 12; i++) { idx[i] = (uint32_t)i; QPt qp; qp.x = q_from_m(w.x); qp.y = q_from_m(w.y); q_store(pts + 8 * i, qp.x); q_store(pts + 8 * i + 4, qp.y); }
    H_TRY(nx_eval_at_points(ctx, polys, n, idx, pts, 12, ev));
    {
        const uint32_t* gp[4] = {qg.accq.c[0], qg.accq.c[1], qg.accq.c[2], qg.accq.c[3]};
        const uint64_t gi[4] = {N, N, N, N};
        H_TRY(nx_gather(ctx, gp, gi, 4, fw));
    }
    const u32 dz = m_inv(m_sub(z1, z0));
    u32 t[4];
    for (int k = 0; k < 4; k++) {
        const u32 f0w = m_add(ev[4 * k], m_mul(z1, ev[4 * (4 + k)]));
        const u32 gw = m_mul(m_sub(fw[k], f0w), inv_z2);
        t[k] = m_mul(m_sub(gw, ev[4 * (8 + k)]), dz);
    }
    H_TRY(out_coef->alloc(ctx, n + 2));
    H_TRY(nx_memset_zero(ctx, out_coef->buf.p, out_coef->buf.words));
    for (int k = 0; k < 4; k++) { H_TRY(nx_copy(ctx, out_coef->c[k], qg.acc2n.c[k], 2 * (size_t)N)); H_TRY(nx_copy(ctx, out_coef->c[k] + 2 * (size_t)N, g.c[k], N)); }
    hipLaunchKernelGGL(half_fix_kernel, dim3(1), dim3(64), 0, ctx->stream, out_coef->c[0] + 2 * (size_t)N, out_coef->c[1] + 2 * (size_t)N, out_coef->c[2] + 2 * (size_t)N,
                       out_coef->c[3] + 2 * (size_t)N, N, t[0], t[1], t[2], t[3], z0);
    if (hipGetLastError() != hipSuccess) return set_err(ctx, NX_ERR_HIP, "half_fix_kernel launch failed");
    H_TRY(nx_sync(ctx));                                         // fq / f0q / g are released on return
    return NX_OK;
}

int GenericAir::compute_composition(CommitmentSchemeProver& cs, QM31 random_coeff, DevBuf* out_polys, uint32_t* out_log) {
    size_t total = 0;
    for (auto& c : comps) total += c.n_constraints;
    std::vector<QM31> powers(total);
    { QM31 a = q_one(); for (size_t i = 0; i < total; i++) { powers[i] = a; a = q_mul(a, random_coeff); } }
    std::map<uint32_t, SecureColumn> sub;
    std::map<uint32_t, HalfGroup> halves;                                     // by log_size: the half-domain parts of all components of that size
    std::map<uint32_t, QuarterGroup> quarters;                                // by log_size: the quarter-domain parts (degree 4 / 5) of all components of that size
    std::map<uint32_t, nx_twiddles*> qtw;                                     // quarter-domain twiddles by log_size
    struct QtwGuard { std::map<uint32_t, nx_twiddles*>& m; ~QtwGuard() { for (auto& kv : m) nx_twiddles_destroy(kv.second); } } qtw_guard{qtw};
    size_t remaining = total;
    for (auto& c : comps) {
        const uint32_t e = c.log_size + comp_log_cd(c.log_cd, cs.cfg);      // per component; finalize_accumulation lifts to the largest
        const size_t nc = c.n_constraints;
        std::vector<uint32_t> pw(4 * nc);          // the LAST nc remaining powers, reversed (accumulator.columns())
        for (size_t j = 0; j < nc; j++) q_store(&pw[4 * j], powers[remaining - 1 - j]);
        remaining -= nc;
        std::vector<char> masked(c.cols.size(), 0);
        for (size_t k = 0; k < c.cols.size(); k++) for (int o : c.masks[k]) if (o != 0) masked[k] = 1;
        { HostSpan hs("cc.prepare_component_kernels"); H_TRY(prepare_component_kernels(ctx, cs.cfg, c, cs.dist.on())); }
        {   // the composition keeps the size the bound declares, whatever domains the parts are evaluated on
            bool any_full = false;       // ... or a quarter part: its contribution has the declared size too (4N coefficients)
            for (auto& part : c.parts) any_full = any_full || part.where == GComponent::ON_FULL || part.where == GComponent::ON_QUARTER;
            SecureColumn* declared = nullptr;
            if (!any_full && e > c.log_size + 1) H_TRY(composition_accumulator(cs, sub, e, &declared));
        }
        for (const GComponent::Part& part : c.parts) {
            const std::vector<char>* used = part.whole ? nullptr : &part.used;
            if (part.where == GComponent::ON_HALF) {
                const uint32_t n = c.log_size, N = 1u << n, el = n + 1;
                HalfGroup& hg = halves[n];
                if (!hg.acc.buf.p) {
                    hg.n_extra = 4;                                  // row N is the one that is used; a multiple of 4 keeps the coordinates 16-byte aligned
                    H_TRY(hg.acc.alloc_rows(ctx, n, (uint64_t)N + hg.n_extra, false));
                    H_TRY(nx_memset_zero(ctx, hg.acc.buf.p, hg.acc.buf.words));
                }
                const std::vector<uint32_t> den = vanishing_denominators(n, el);
                EvalDomainCols cols;
                H_TRY(columns_on_eval_domain(cs, c.cols, n, el, masked, &cols, used));          // the committed evaluations (blowup 2)
                uint32_t* a4[4]; for (int k = 0; k < 4; k++) a4[k] = hg.acc.c[k];
                H_TRY(air_eval_rows(ctx, part.kernel, cols.ptrs.data(), c.econsts.data(), pw.data(), den.data(), n, el, a4, 0, N + hg.n_extra));
                continue;
            }
            if (part.where == GComponent::ON_QUARTER && cs.dist.on()) {
                // One proof on several GPUs.  The same three sample sets, placed where the data is: (1) the committed
                // 2N rows are row blocks already — no transform, no exchange; (2) the first quarter of the 4N-point domain: every GPU
                // transforms the polynomials IT holds on N points and one all-to-all hands out blocks of N / W rows — a quarter of the
                // transform work and of the bytes of the 4N-point re-evaluation this replaces; the neighbour-read columns (a handful: Pc,
                // IsPadding) are transformed on 2N points and all-gathered whole; (3) row N: one eval_at_point sweep per column on its
                // owner, the values all-gathered by the hosts, the row evaluated by every GPU.  quarter_group_finish then runs on the
                // gathered accumulators, replicated like finalize_accumulation.
                const Dist& D = cs.dist;
                const uint32_t n = c.log_size, N = 1u << n;
                QuarterGroup& qg = quarters[n];
                if (!qg.acc2n.buf.p) {
                    H_TRY(qg.acc2n.alloc_rows(ctx, n + 1, D.block(n + 1), true)); H_TRY(nx_memset_zero(ctx, qg.acc2n.buf.p, qg.acc2n.buf.words));
                    H_TRY(qg.accq.alloc_rows(ctx, n, D.block(n), true)); H_TRY(nx_memset_zero(ctx, qg.accq.buf.p, qg.accq.buf.words));
                    H_TRY(qg.rown.alloc_rows(ctx, n, 4, false)); H_TRY(nx_memset_zero(ctx, qg.rown.buf.p, qg.rown.buf.words));
                    H_TRY(twiddles_first_part(ctx, cs.tw, n, 2, &qtw[n]));
                }
                {
                    const std::vector<uint32_t> den = vanishing_denominators(n, n + 1);
                    EvalDomainCols cols;
                    H_TRY(columns_on_eval_domain(cs, c.cols, n, n + 1, masked, &cols, used));
                    const uint64_t rb = D.begin(n + 1);
                    uint32_t* a4[4]; for (int k = 0; k < 4; k++) a4[k] = bias_rows(qg.acc2n.c[k], rb);
                    H_TRY(air_eval_rows(ctx, part.kernel, cols.ptrs.data(), c.econsts.data(), pw.data(), den.data(), n, n + 1, a4, (uint32_t)rb, (uint32_t)D.block(n + 1)));
                }
                std::vector<size_t> sel, selm;
                for (size_t k = 0; k < c.cols.size(); k++) if (!used || (k < used->size() && (*used)[k])) (masked[k] ? selm : sel).push_back(k);
                const size_t ns = sel.size(), nm = selm.size();
                std::vector<DevBuf> keep;
                std::vector<const uint32_t*> ptrs(c.cols.size(), nullptr);
                const uint64_t rbq = D.begin(n), mbq = D.block(n);
                std::vector<uint32_t*> blk;
                if (ns) H_TRY(sharded_evaluate_exchange(cs, c.cols, sel, qtw[n], n, n, &blk, &keep));
                for (size_t i = 0; i < ns; i++) ptrs[sel[i]] = bias_rows((const uint32_t*)blk[i], rbq);
                DevBuf wholem;
                if (nm) {
                    nx_twiddles* htw = nullptr;                      // the first half of the 4N-point domain as a 2N-point circle domain
                    H_TRY(twiddles_first_part(ctx, cs.tw, n + 1, 1, &htw));
                    struct Guard { nx_twiddles* t; ~Guard() { nx_twiddles_destroy(t); } } guard{htw};
                    std::vector<uint32_t*> bm;
                    H_TRY(sharded_evaluate_exchange(cs, c.cols, selm, htw, n, n + 1, &bm, &keep));
                    H_TRY(wholem.alloc(ctx, nm << (n + 1)));
                    std::vector<const uint32_t*> mb(bm.begin(), bm.end());
                    H_TRY(D.allgather_cols(ctx, mb, (size_t)D.block(n + 1), wholem.p, (uint64_t)1 << (n + 1)));
                    for (size_t i = 0; i < nm; i++) ptrs[selm[i]] = wholem.p + (i << (n + 1));
                }
                const std::vector<uint32_t> den = vanishing_denominators(n, n + 2);
                {
                    uint32_t* a4[4]; for (int k = 0; k < 4; k++) a4[k] = bias_rows(qg.accq.c[k], rbq);
                    H_TRY(air_eval_rows(ctx, part.kernel, ptrs.data(), c.econsts.data(), pw.data(), den.data(), n, n + 2, a4, (uint32_t)rbq, (uint32_t)mbq));
                }
                DevBuf wv; H_TRY(wv.alloc(ctx, std::max<size_t>(ns, 4)));
                if (ns) {
                    // the columns' values at w = row N of the 4N-point domain: swept by their owners, all-gathered as host words
                    std::vector<uint32_t> mine(ns, 0), all((size_t)D.world * ns);
                    std::vector<const uint32_t*> src; std::vector<size_t> slot;
                    for (size_t i = 0; i < ns; i++)
                        if (cs.trees[c.cols[sel[i]].first].owner[c.cols[sel[i]].second] == D.rank) { src.push_back(cs.trees[c.cols[sel[i]].first].polys[c.cols[sel[i]].second].ptr); slot.push_back(i); }
                    if (!src.empty()) {
                        const Pt w = pt_from_index(circle_domain_index((int)n + 2, bitrev(N, (int)n + 2)));
                        const size_t nl = src.size();
                        std::vector<uint32_t> pidx(nl), pts(8 * nl), ev(4 * nl);
                        uint32_t pw8[8]; { QPt qp; qp.x = q_from_m(w.x); qp.y = q_from_m(w.y); q_store(pw8, qp.x); q_store(pw8 + 4, qp.y); }
                        for (size_t i = 0; i < nl; i++) { pidx[i] = (uint32_t)i; memcpy(&pts[8 * i], pw8, 32); }
                        H_TRY(nx_eval_at_points(ctx, src.data(), n, pidx.data(), pts.data(), (uint32_t)nl, ev.data()));
                        for (size_t i = 0; i < nl; i++) mine[slot[i]] = ev[4 * i];
                    }
                    H_TRY(D.allgather_host(ctx, mine.data(), ns * 4, all.data()));
                    std::vector<uint32_t> wh(ns);
                    for (size_t i = 0; i < ns; i++) wh[i] = all[(size_t)cs.trees[c.cols[sel[i]].first].owner[c.cols[sel[i]].second] * ns + i];
                    H_TRY(nx_upload(ctx, wv.p, wh.data(), ns));
                    for (size_t i = 0; i < ns; i++) ptrs[sel[i]] = bias_rows((const uint32_t*)(wv.p + i), N);
                }
                if (ns + nm) {
                    uint32_t* a4[4]; for (int k = 0; k < 4; k++) a4[k] = bias_rows(qg.rown.c[k], N);
                    H_TRY(air_eval_rows(ctx, part.kernel, ptrs.data(), c.econsts.data(), pw.data(), den.data(), n, n + 2, a4, N, 1));
                    H_TRY(nx_sync(ctx));                                                   // keep / wholem / wv are released at the end of this block
                }
                continue;
            }
            if (part.where == GComponent::ON_QUARTER) {
                const uint32_t n = c.log_size, N = 1u << n;
                QuarterGroup& qg = quarters[n];
                if (!qg.acc2n.buf.p) {
                    H_TRY(qg.acc2n.alloc(ctx, n + 1)); H_TRY(nx_memset_zero(ctx, qg.acc2n.buf.p, qg.acc2n.buf.words));
                    H_TRY(qg.accq.alloc_rows(ctx, n, (uint64_t)N + 4, false)); H_TRY(nx_memset_zero(ctx, qg.accq.buf.p, qg.accq.buf.words));
                    H_TRY(twiddles_first_part(ctx, cs.tw, n, 2, &qtw[n]));
                }
                {   // 1. the committed 2N rows (Z2 vanishes there: what comes out is F0)
                    const std::vector<uint32_t> den = vanishing_denominators(n, n + 1);
                    EvalDomainCols cols;
                    H_TRY(columns_on_eval_domain(cs, c.cols, n, n + 1, masked, &cols, used));
                    H_TRY(air_eval_rows(ctx, part.kernel, cols.ptrs.data(), c.econsts.data(), pw.data(), den.data(), n, n + 1, qg.acc2n.c, 0, 2 * N));
                }
                // 2. the columns the part reads, on the first quarter of the 4N-point domain (an N-point transform each) and at row N.
                // A column that is read at a neighbour row (`masked`: the reference's Pc / IsPadding, prover/src/column.rs:13-20) is
                // evaluated on the first HALF of the 4N-point domain instead — a 2N-point transform: the trace step maps the first quarter
                // +-(g + <g_{N/2}>) onto the second and back (tests/test_exact_algebra_cpu.py::test_neighbour_rows_of_the_first_quarter_
                // are_the_second_quarter), so the neighbour of every quarter row, and of row N, lies in rows [0, 2N) and the kernel's own
                // row arithmetic on the 4N-point domain indexes that buffer unchanged.
                std::vector<size_t> sel, selm;
                for (size_t k = 0; k < c.cols.size(); k++) if (!used || (k < used->size() && (*used)[k])) (masked[k] ? selm : sel).push_back(k);
                const size_t ns = sel.size(), nm = selm.size();
                std::vector<const uint32_t*> src(ns), srcm(nm);
                for (size_t i = 0; i < ns; i++) src[i] = cs.trees[c.cols[sel[i]].first].polys[c.cols[sel[i]].second].ptr;
                for (size_t i = 0; i < nm; i++) srcm[i] = cs.trees[c.cols[selm[i]].first].polys[c.cols[selm[i]].second].ptr;
                DevBuf ext, extm, wv; H_TRY(ext.alloc(ctx, std::max<size_t>(ns, 1) << n)); H_TRY(wv.alloc(ctx, std::max<size_t>(ns, 4)));
                auto dst = col_ptrs(ext.p, (uint32_t)ns, n);
                if (ns) H_TRY(nx_evaluate_batch(ctx, qtw[n], src.data(), (uint32_t)ns, n, 0, dst.data()));
                std::vector<uint32_t*> dstm;
                if (nm) {
                    nx_twiddles* htw = nullptr;                      // the first half of the 4N-point domain as a 2N-point circle domain
                    H_TRY(twiddles_first_part(ctx, cs.tw, n + 1, 1, &htw));
                    struct Guard { nx_twiddles* t; ~Guard() { nx_twiddles_destroy(t); } } guard{htw};
                    H_TRY(extm.alloc(ctx, nm << (n + 1)));
                    dstm = col_ptrs(extm.p, (uint32_t)nm, n + 1);
                    H_TRY(nx_evaluate_batch(ctx, htw, srcm.data(), (uint32_t)nm, n, 1, dstm.data()));
                    H_TRY(nx_sync(ctx));                             // the sub-tree is released here
                }
                const std::vector<uint32_t> den = vanishing_denominators(n, n + 2);
                std::vector<const uint32_t*> ptrs(c.cols.size(), nullptr);
                for (size_t i = 0; i < ns; i++) ptrs[sel[i]] = dst[i];
                for (size_t i = 0; i < nm; i++) ptrs[selm[i]] = dstm[i];
                uint32_t* a4[4]; for (int k = 0; k < 4; k++) a4[k] = qg.accq.c[k];
                H_TRY(air_eval_rows(ctx, part.kernel, ptrs.data(), c.econsts.data(), pw.data(), den.data(), n, n + 2, a4, 0, N));
                if (ns + nm) {
                    if (ns) {
                        const Pt w = pt_from_index(circle_domain_index((int)n + 2, bitrev(N, (int)n + 2)));
                        std::vector<uint32_t> pidx(ns), pts(8 * ns), ev(4 * ns), wh(ns);
                        uint32_t pw8[8]; { QPt qp; qp.x = q_from_m(w.x); qp.y = q_from_m(w.y); q_store(pw8, qp.x); q_store(pw8 + 4, qp.y); }
                        for (size_t i = 0; i < ns; i++) { pidx[i] = (uint32_t)i; memcpy(&pts[8 * i], pw8, 32); }
                        H_TRY(nx_eval_at_points(ctx, src.data(), n, pidx.data(), pts.data(), (uint32_t)ns, ev.data()));
                        for (size_t i = 0; i < ns; i++) wh[i] = ev[4 * i];                       // a base-field polynomial at a base-field point
                        H_TRY(nx_upload(ctx, wv.p, wh.data(), ns));
                        for (size_t i = 0; i < ns; i++) ptrs[sel[i]] = bias_rows((const uint32_t*)(wv.p + i), N);      // one-row "columns", indexed with the global row N
                    }
                    // the neighbour-read columns keep their 2N-row buffers: row N and its neighbours (first quarter) are both in them
                    H_TRY(air_eval_rows(ctx, part.kernel, ptrs.data(), c.econsts.data(), pw.data(), den.data(), n, n + 2, a4, N, 1));
                    H_TRY(nx_sync(ctx));                                                   // ext / extm / wv are released at the end of this block
                }
                continue;
            }
            const uint32_t pe = part.where == GComponent::ON_LOW ? c.log_size + 1 : e;
            const std::vector<uint32_t> den = vanishing_denominators(c.log_size, pe);
            EvalDomainCols cols;
            H_TRY(columns_on_eval_domain(cs, c.cols, c.log_size, pe, masked, &cols, used));
            SecureColumn* acc = nullptr;
            H_TRY(composition_accumulator(cs, sub, pe, &acc));
            const uint64_t rb = cs.dist.on() ? cs.dist.begin(pe) : 0;
            uint32_t* a4[4]; for (int k = 0; k < 4; k++) a4[k] = bias_rows(acc->c[k], rb);
            H_TRY(air_eval_rows(ctx, part.kernel, cols.ptrs.data(), c.econsts.data(), pw.data(), den.data(), c.log_size, pe, a4, (uint32_t)rb, (uint32_t)acc->rows));
            // `cols` (the re-evaluated columns) is released here: stream-ordered frees, the kernel above was enqueued first
        }
    }
    std::map<uint32_t, SecureColumn> coef;
    { HostSpan hs("cc.half_group_finish"); for (auto& kv : halves) H_TRY(half_group_finish(ctx, cs, kv.first, kv.second, &coef[kv.first + 1])); }
    for (auto& kv : quarters) {
        // components of 2N rows may have put a half-domain contribution at the same size: coefficient vectors add
        SecureColumn q4;
        if (cs.dist.on()) {                                     // the blocks become whole accumulators on every GPU; the finish is replicated
            QuarterGroup& qg = kv.second;
            const uint32_t n = kv.first, N = 1u << n;
            SecureColumn w2, wq, full;
            H_TRY(gather_secure(ctx, cs.dist, qg.acc2n, &w2));
            H_TRY(gather_secure(ctx, cs.dist, qg.accq, &wq));
            H_TRY(full.alloc_rows(ctx, n, (uint64_t)N + 4, false));
            H_TRY(nx_memset_zero(ctx, full.buf.p, full.buf.words));
            for (int k = 0; k < 4; k++) { H_TRY(nx_copy(ctx, full.c[k], wq.c[k], N)); H_TRY(nx_copy(ctx, full.c[k] + N, qg.rown.c[k], 1)); }
            H_TRY(nx_sync(ctx));                                // wq is released here
            qg.acc2n = std::move(w2); qg.accq = std::move(full);
        }
        H_TRY(quarter_group_finish(ctx, cs, kv.first, kv.second, &q4));
        auto it = coef.find(kv.first + 2);
        if (it == coef.end()) { coef[kv.first + 2] = std::move(q4); continue; }
        const u32* s4[4] = {q4.c[0], q4.c[1], q4.c[2], q4.c[3]};
        H_TRY(secure_accumulate(ctx, it->second.c, s4, 1u << (kv.first + 2)));
        H_TRY(nx_sync(ctx));                                   // q4 is released here
    }
    HostSpan hs("cc.finalize_accumulation");
    return finalize_accumulation(cs, sub, out_polys, out_log, coef.empty() ? nullptr : &coef);
}

void GenericAir::mask_points(QPt oods, MaskPoints* points) {
    points->assign(offs.size(), {});
    std::map<std::pair<uint32_t, int>, QPt> shifted;          // (column log size, offset) -> oods + offset x the trace step: a few distinct pairs for thousands of columns
    for (size_t t = 0; t < offs.size(); t++) {
        points->at(t).reserve(offs[t].size());
        for (size_t c = 0; c < offs[t].size(); c++) {
            std::vector<QPt> pts;
            pts.reserve(offs[t][c].size());
            for (int o : offs[t][c]) {
                if (o == 0) { pts.push_back(oods); continue; }
                auto it = shifted.find({tree_logs[t][c], o});
                if (it == shifted.end()) {
                    const int64_t idx = ((int64_t)o * ((int64_t)1 << (31 - tree_logs[t][c]))) & 0x7fffffffLL;
                    Pt s = pt_from_index((u32)idx);
                    QPt step; step.x = q_from_m(s.x); step.y = q_from_m(s.y);
                    it = shifted.insert({{tree_logs[t][c], o}, qpt_add(oods, step)}).first;
                }
                pts.push_back(it->second);
            }
            points->at(t).push_back(std::move(pts));
        }
    }
}

// the recorded program over QM31: every register holds the value of its expression at the OODS point
QM31 GenericAir::eval_composition_at_point(QPt point, const SampledValues& sv, QM31 rc) {
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

// MerkleProver::decommit(queries_per_log_size, columns) (inside stwo::prover::prove -> CommitmentSchemeProver::prove_values,
// reference machine.rs:286-290): the values of `columns` (commit order, any sizes) at the queried rows, plus the witness a
// verifier needs to recompute the root — sibling hashes not derivable from the queries and the column values of visited but
// unqueried nodes.  One (log, count) pair per queried layer; `queries` holds the sorted, deduplicated positions of every queried
// layer back to back.  All reads travel in ONE gather launch and one device-to-host copy.
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
        cols[i] = {const_cast<uint32_t*>(d_cols[i]), log_sizes[i], false};
    }
    nxhip::TreeRef ref; ref.local = const_cast<nx_tree*>(tree); ref.n_layers = n_layers;
    struct Release { nxhip::TreeRef& r; ~Release() { r.local = nullptr; } } release{ref};   // borrowed: the caller owns the tree
    nxhip::GatherBatch gb;
    nxhip::DecommitPlan plan = nxhip::merkle_decommit_plan(ref, qpl, cols, &gb);
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

// The share of a tree's columns each GPU of a row-sharded prove transforms (plan_local_columns): consecutive groups of one log size
// form a run, a run's columns are cut into `world` contiguous balanced ranges.  Pure host arithmetic (no context): a caller that
// fills its own trace uses it to know which columns of which component are its own.
int nx_plan_local_columns(const uint32_t* group_n_cols, const uint32_t* group_log_sizes, uint32_t n_groups, int32_t rank, int32_t world, uint32_t* lo, uint32_t* hi) {
    if ((n_groups && (!group_n_cols || !group_log_sizes || !lo || !hi)) || world < 1 || rank < 0 || rank >= world) return set_err(nullptr, NX_ERR_ARG, "nx_plan_local_columns: bad argument");
    std::vector<std::pair<uint32_t, uint32_t>> groups(n_groups), out;
    for (uint32_t i = 0; i < n_groups; i++) groups[i] = {group_n_cols[i], group_log_sizes[i]};
    nxhip::Dist d; d.rank = rank; d.world = world;
    nxhip::plan_local_columns(groups, d, &out);
    for (uint32_t i = 0; i < n_groups; i++) { lo[i] = out[i].first; hi[i] = out[i].second; }
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

// One proof on several GPUs: every GPU runs the same session (same transcript calls, same trees) with its own communicator; it
// fills only the columns nx_prover_tree_begin hands it (NULL for the others).  Call before the first tree.
int nx_prover_set_comm(nx_prover* p, const nx_comm* comm) {
    NX_GUARD(p ? p->ctx : nullptr);
    if (!p || !comm) return set_err(p ? p->ctx : nullptr, NX_ERR_ARG, "nx_prover_set_comm: NULL argument");
    if (!p->cs->trees.empty() || p->open) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_set_comm: must be called before the first tree");
    p->comm_copy = *comm; p->has_comm = true;
    return nxhip::dist_init(p->ctx, &p->comm_copy, &p->cs->dist);
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
    if (p->proved) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_begin: the session has proved; its trace trees are fixed");
    if (p->cs->trees.size() >= 16) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_begin: at most 16 trace trees");
    for (uint32_t i = 0; i < n_cols; i++) if (log_sizes[i] < 1 || log_sizes[i] > p->max_log) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_begin: column log size outside [1, max_log_size]");
    p->pending.clear();
    const nxhip::Dist& D = p->cs->dist;
    for (uint32_t i = 0; i < n_cols;) {             // one slab per run of equal sizes (TreeBuilder::extend_evals groups); row-sharded: this GPU's range of the run
        uint32_t j = i; while (j < n_cols && log_sizes[j] == log_sizes[i]) j++;
        nx_prover::Run r; r.n_cols = j - i; r.log = log_sizes[i];
        r.lo = nxhip::Dist::cut(r.n_cols, D.rank, D.world); r.hi = nxhip::Dist::cut(r.n_cols, D.rank + 1, D.world);
        int rc = r.slab.alloc(p->ctx, (size_t)std::max<uint32_t>(r.hi - r.lo, 1) << r.log);
        if (rc != NX_OK) { p->pending.clear(); return rc; }
        for (uint32_t k = 0; k < r.n_cols; k++) d_cols_out[i + k] = (k >= r.lo && k < r.hi) ? r.slab.p + ((size_t)(k - r.lo) << r.log) : nullptr;
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
    for (auto& r : p->pending) tb.extend_evals_local(std::move(r.slab), r.n_cols, r.log, r.lo, r.hi);
    p->pending.clear(); p->open = false;
    {
        p->ctx->symmetric_failure = false;
        const int rc = tb.commit(p->channel);
        if (rc != NX_OK) { if (p->has_comm && p->comm_copy.world > 1 && p->comm_copy.abort && !p->ctx->symmetric_failure) p->comm_copy.abort(p->comm_copy.user); return rc; }
    }
    if (root) memcpy(root, p->cs->trees.back().root.w, 32);
    return NX_OK;
}

// A committed tree as a handle other sessions of the same context can adopt: the session's entry becomes a view of the shared tree.
int nx_prover_tree_share(nx_prover* p, uint32_t tree_index, nx_committed_tree** out) {
    NX_GUARD(p ? p->ctx : nullptr);
    if (!p || !out) return set_err(p ? p->ctx : nullptr, NX_ERR_ARG, "nx_prover_tree_share: NULL argument");
    if (p->cs->dist.on()) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_share: one GPU (a row-sharded tree is spread over the ranks)");
    const size_t limit = p->proved ? p->pre_trees : p->cs->trees.size();
    if (tree_index >= limit) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_share: no such committed trace tree");
    nxhip::CommitmentTreeProver& cur = p->cs->trees[tree_index];
    std::shared_ptr<nxhip::CommitmentTreeProver> sp = cur.backing ? cur.backing : std::make_shared<nxhip::CommitmentTreeProver>(std::move(cur));
    if (!cur.backing) p->cs->trees[tree_index] = nxhip::borrow_tree(sp);
    *out = new nx_committed_tree{p->ctx, p->cfg.log_blowup, p->ctx->hash_mode, sp};
    return NX_OK;
}
// TreeBuilder::commit for a tree that was committed before: no upload, no transform, no hashing — the root enters the transcript as it
// would after a fresh commit, the prove reads the shared coefficients / extensions / nodes.
int nx_prover_tree_adopt(nx_prover* p, const nx_committed_tree* shared, uint8_t root[32]) {
    NX_GUARD(p ? p->ctx : nullptr);
    if (!p || !shared || !shared->tree) return set_err(p ? p->ctx : nullptr, NX_ERR_ARG, "nx_prover_tree_adopt: NULL argument");
    if (p->open) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_adopt: a tree is begun but not committed");
    if (p->proved) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_adopt: the session has proved; its trace trees are fixed");
    if (p->cs->dist.on()) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_adopt: one GPU");
    if (shared->ctx != p->ctx) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_adopt: the tree lives in another context (its buffers belong to that context's device and allocator)");
    if (shared->log_blowup != p->cfg.log_blowup || shared->hash_mode != p->ctx->hash_mode) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_adopt: the tree was committed under another blowup factor or node hash");
    if (p->cs->trees.size() >= 16) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_adopt: at most 16 trace trees");
    for (auto& c : shared->tree->polys) if (c.log > p->max_log) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_adopt: a column is larger than the session's max_log_size");
    p->cs->trees.push_back(nxhip::borrow_tree(shared->tree));
    p->channel.mix_root(shared->tree->root);
    if (root) memcpy(root, shared->tree->root.w, 32);
    return NX_OK;
}
void nx_committed_tree_release(nx_committed_tree* t) {
    if (!t) return;
    NX_GUARD(t->ctx);
    (void)nx_sync(t->ctx);                      // a prove that read the tree may still be queued
    delete t;
}
int nx_committed_tree_root(const nx_committed_tree* t, uint8_t root[32], uint32_t* n_cols) {
    if (!t || !t->tree || !root) return set_err(nullptr, NX_ERR_ARG, "nx_committed_tree_root: NULL argument");
    memcpy(root, t->tree->root.w, 32);
    if (n_cols) *n_cols = (uint32_t)t->tree->polys.size();
    return NX_OK;
}

int nx_prover_tree_commit_host(nx_prover* p, const uint32_t* const* h_cols, int coset_order, const uint32_t* keep_idx, uint32_t n_keep, uint32_t* const* d_keep,
                               uint8_t root[32]) {
    NX_GUARD(p ? p->ctx : nullptr);
    if (!p) return set_err(nullptr, NX_ERR_ARG, "nx_prover_tree_commit_host: NULL prover");
    if (!p->open) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_commit_host: no tree was begun");
    uint32_t n_total = 0;
    for (auto& r : p->pending) n_total += r.n_cols;
    if ((n_total && !h_cols) || (n_keep && (!keep_idx || !d_keep))) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_commit_host: NULL argument");
    for (uint32_t k = 0; k < n_keep; k++) if (keep_idx[k] >= n_total || !d_keep[k]) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_commit_host: keep entry outside the tree (or NULL)");
    const bool sharded = p->cs->dist.on();
    // a rank that fails here (before or inside the commit) tells its peers, which are in — or about to enter — the commit's exchanges
    auto run = [&]() -> int {
        nxhip::TreeBuilder tb = p->cs->tree_builder();
        uint32_t first = 0;
        for (auto& r : p->pending) {
            if (sharded) {
                // this GPU's columns of the run, uploaded before the commit (the exchange-bound sharded commit gains nothing from the overlap)
                std::vector<const uint32_t*> hs; std::vector<uint32_t*> ds;
                for (uint32_t k = r.lo; k < r.hi; k++) { if (!h_cols[first + k]) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_commit_host: NULL host column"); hs.push_back(h_cols[first + k]); ds.push_back(r.slab.p + ((size_t)(k - r.lo) << r.log)); }
                if (!hs.empty()) NX_TRY(nx_upload_columns(p->ctx, hs.data(), (uint32_t)hs.size(), r.log, ds.data(), coset_order));
                for (uint32_t k = 0; k < n_keep; k++)
                    if (keep_idx[k] >= first + r.lo && keep_idx[k] < first + r.hi) NX_TRY(nx_copy(p->ctx, d_keep[k], r.slab.p + ((size_t)(keep_idx[k] - first - r.lo) << r.log), (size_t)1 << r.log));
                tb.extend_evals_local(std::move(r.slab), r.n_cols, r.log, r.lo, r.hi);
            } else {
                for (uint32_t k = 0; k < r.n_cols; k++) if (!h_cols[first + k]) return set_err(p->ctx, NX_ERR_ARG, "nx_prover_tree_commit_host: NULL host column");
                std::vector<std::pair<uint32_t, uint32_t*>> keep;
                for (uint32_t k = 0; k < n_keep; k++) if (keep_idx[k] >= first && keep_idx[k] < first + r.n_cols) keep.push_back({keep_idx[k] - first, d_keep[k]});
                tb.extend_evals_host(std::move(r.slab), r.n_cols, r.log, h_cols + first, coset_order, keep);
            }
            first += r.n_cols;
        }
        return tb.commit(p->channel);
    };
    p->ctx->symmetric_failure = false;
    const int rc = run();
    p->pending.clear(); p->open = false;
    if (rc != NX_OK) { if (p->has_comm && p->comm_copy.world > 1 && p->comm_copy.abort && !p->ctx->symmetric_failure) p->comm_copy.abort(p->comm_copy.user); return rc; }
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
        if (!nxhip::comp_log_cd_ok(u.log_constraint_degree_bound, p->cfg.log_constraint_degree))
            return set_err(ctx, NX_ERR_ARG, "nx_prover_prove: a component's log_constraint_degree_bound exceeds the session config's log_constraint_degree (the twiddle tree is sized by it)");
        g.log_size = u.log_size; g.n_regs = u.n_regs; g.n_constraints = u.n_constraints; g.kernel = u.kernel; g.log_cd = u.log_constraint_degree_bound;
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
    if (p->proved) { while (p->cs->trees.size() > p->pre_trees) p->cs->trees.pop_back(); p->channel = p->pre_channel; }   // a second prove of the session
    ctx->symmetric_failure = false;
    if (p->cs->dist.on()) {
        // everything that can fail on one rank only (argument checks, hiprtc) happens before the first exchange, then the ranks vote
        int rc_local = air.check(*p->cs);
        for (auto& c : air.comps) if (rc_local == NX_OK) rc_local = nxhip::prepare_component_kernels(ctx, p->cfg, c, true);
        const int rcv = nxhip::vote_before_exchanges(ctx, p->cs->dist, rc_local, "nx_prover_prove");
        // a no-go of the vote is every rank's: nobody entered an exchange, nothing to abort (only a vote whose own all-gather failed is one-sided)
        if (rcv != NX_OK) { if (p->has_comm && p->comm_copy.abort && !ctx->symmetric_failure) p->comm_copy.abort(p->comm_copy.user); return rcv; }
    }
    NX_TRY(air.check(*p->cs));
    const bool timed = stats != nullptr;
    nx_prove_stats local;
    nx_prove_stats* st = stats ? stats : &local;
    memset(st, 0, sizeof *st);
    if (timed) { ctx->timing = true; timing_reset(ctx); }
    p->cs->dist.comm_ms = &st->comm_ms; p->cs->dist.comm_bytes = &st->comm_bytes; p->cs->dist.comm_calls = &st->n_alltoallv;
    nxhip::Lap lap{ctx, timed, 0};
    double t_start = 0;
    if (timed) { (void)nx_sync(ctx); t_start = lap.t0 = nxhip::now_ms(); }
    std::vector<uint32_t> w;
    // prove_core appends the composition tree and advances the channel (nx_prover_channel_digest afterwards is the transcript's final
    // state, as after stwo::prover::prove).  A later call starts again from the state the FIRST call found, and a failed call puts
    // that state back at once: the session can prove the same committed statement again (a retry after NX_ERR_PROTOCOL, other components)
    if (!p->proved) { p->pre_trees = p->cs->trees.size(); p->pre_channel = p->channel; p->proved = true; }
    int rc = nxhip::prove_core(ctx, *p->cs, p->channel, p->cfg, p->tw, air, &w, st, lap);
    if (rc != NX_OK) { while (p->cs->trees.size() > p->pre_trees) p->cs->trees.pop_back(); p->channel = p->pre_channel; p->proved = false; }
    if (rc != NX_OK && p->has_comm && p->comm_copy.world > 1 && p->comm_copy.abort && !ctx->symmetric_failure) p->comm_copy.abort(p->comm_copy.user);   // the peers wait in a collective this rank will not enter
    if (timed) nxhip::finish_stats(ctx, st, t_start);
    ctx->timing = false;
    p->cs->dist.comm_ms = nullptr; p->cs->dist.comm_bytes = nullptr; p->cs->dist.comm_calls = nullptr;
    if (rc != NX_OK) return rc;
    uint32_t* out = (uint32_t*)malloc(w.size() * 4);
    if (!out) return set_err(ctx, NX_ERR_OOM, "nx_prover_prove: malloc failed");
    memcpy(out, w.data(), w.size() * 4);
    *proof_words = out; *n_words = w.size();
    return NX_OK;
}

}  // extern "C"
