// Host-side prover objects shared by prover.hip (commitment scheme, FRI, stwo::prover::prove), machine.hip (the synthetic machines)
// and the nx_prover session: the C++ mirror of the Stwo objects the reference instantiates
//   CommitmentSchemeProver / TreeBuilder   (reference prover/src/machine.rs:202-263)
//   stwo::prover::prove, FriProver, prove_values, decommit   (reference prover/src/machine.rs:286-290)
// written against the C ABI of include/nexus_hip.h.  The reference's toolchain (Rust nightly) is absent from this image, hence C++.
//
// ONE proof on several GPUs (BASELINE configs #4/#5, `Dist`): the LDE is column-parallel (every GPU transforms its share of a tree's
// columns), then ONE all-to-all per tree turns column shards into ROW BLOCKS — GPU r holds rows [r M/W, (r+1) M/W) of every LDE
// column, a contiguous block of the bit-reversed domain, i.e. a subtree of the Merkle tree.  Leaf hashing, constraint evaluation,
// DEEP quotients and the FRI folds (pairs are adjacent in bit-reversed order) are then local; what crosses the links besides the
// transposition: the W subtree roots of every tree (32 B each), the sampled values (KBs), the few columns read at a non-zero mask
// offset (all-gather), the composition accumulator (all-gather of 4 columns) and the small FRI tail.  Proof bytes are those of the
// single-GPU prover.
#pragma once
#include "internal.h"
#include <memory>
#include "air.h"
#include "host/channel.h"
#include <algorithm>
#include <map>
#include <set>
#include <vector>
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

// ---------------------------------------------------------------- one proof over the GPUs of a node ----------
double now_ms();
struct Dist {
    const nx_comm* comm = nullptr; int rank = 0, world = 1, log_w = 0;
    double* comm_ms = nullptr; uint64_t* comm_bytes = nullptr;      // accounting sinks (nx_prove_stats), may be null
    uint32_t* comm_calls = nullptr;                                  // [alltoallv, allgather_dev, allgather_host] counters of nx_prove_stats, may be null
    bool on() const { return world > 1; }
    uint64_t block(uint32_t log) const { return ((uint64_t)1 << log) >> log_w; }        // rows per GPU of a column of 2^log rows
    uint64_t begin(uint32_t log) const { return block(log) * (uint64_t)rank; }            // first row of this GPU's block
    // [lo, hi) of n items owned by rank r (contiguous, balanced)
    static uint32_t cut(uint32_t n, int r, int world) { return (uint32_t)(((uint64_t)n * (uint64_t)r) / (uint64_t)world); }
    int allgather_host(nx_ctx* ctx, const void* send, size_t bytes, void* recv) const;
    int allgather_dev(nx_ctx* ctx, const uint32_t* d_send, size_t words, uint32_t* d_recv) const;
    // several row blocks (rb words each) -> their whole columns (whole_base + k * whole_stride, world * rb words each) with ONE collective:
    // the blocks travel as one buffer, the [rank][column][rb] result is unpacked by transpose_blocks
    int allgather_cols(nx_ctx* ctx, const std::vector<const uint32_t*>& blk, size_t rb, uint32_t* whole_base, uint64_t whole_stride) const;
    // ready: an event recorded on the context's stream after the send buffer was written — the host then waits for that event only,
    // so work enqueued behind it (the next column chunk's LDE) keeps the GPU busy during the exchange; null = drain the stream.
    int alltoallv(nx_ctx* ctx, const uint32_t* d_send, const size_t* soff, const size_t* scnt, uint32_t* d_recv, const size_t* roff, const size_t* rcnt,
                  hipEvent_t ready = nullptr) const;
};
int dist_init(nx_ctx* ctx, const nx_comm* comm, Dist* out);   // validates the callbacks; world must be a power of two
// The vote every sharded prove takes before its first exchange: each rank's local status (kernel compilation is the one thing that can
// fail on ONE rank only) and the context options that shape the exchanges ("air.degree_split", "air.half_domain", "air.quarter_domain",
// "fri.dist_min_log", "dist.chunks": which domains are evaluated, what is all-gathered, how many chunks an all-to-all has).  A rank that
// failed, or a rank configured differently, would otherwise meet its peers in mismatched collectives — a hang or garbage.
int vote_before_exchanges(nx_ctx* ctx, const Dist& D, int rc_local, const char* who);

// A column of 2^log words: the whole column (single GPU, or replicated on every GPU) or — block — this GPU's contiguous block of
// 2^log / W rows.  ptr == nullptr: another GPU holds it (coefficients stay column-sharded).
struct ColumnRef { uint32_t* ptr; uint32_t log; bool block; };

struct PcsConfig { uint32_t pow_bits, log_blowup, n_queries, log_last_layer_degree_bound, fri_alpha_mode, log_constraint_degree; };
// A component's log constraint-degree bound (nx_component_spec / nx_air_component .log_constraint_degree_bound): 0 = the config's.
// The reference's bound is per component — v1 main +2 (prover/src/components/mod.rs:12,44-45), extensions +1
// (prover/src/extensions/multiplicity.rs:108-110), prover2 1 / shifts 2 (framework/traits/builtin.rs:23) — and the composition
// polynomial takes the maximum (prover2/machine/src/prove.rs:44-48).
inline uint32_t comp_log_cd(uint32_t bound, const PcsConfig& cfg) { return bound ? bound : cfg.log_constraint_degree; }
inline bool comp_log_cd_ok(uint32_t bound, uint32_t cfg_lcd) { return bound <= cfg_lcd; }   // 0 (default) or 1 .. the config's (twiddle sizing)
struct MerkleDecommitment { std::vector<Blake2sHash> hash_witness; std::vector<uint32_t> column_witness; };

// A committed Merkle tree: whole on this GPU (log_w == 0), or row-sharded — the subtree over this GPU's row block plus host copies of
// the layers above the W subtree roots (identical on every GPU).
struct TreeRef {
    nx_tree* local = nullptr;
    bool borrowed = false;                               // the nodes belong to a shared tree (CommitmentTreeProver::backing keeps it alive)
    int log_w = 0;
    uint32_t n_layers = 0;                               // global layer count
    std::vector<std::vector<Blake2sHash>> top;           // top[k], k <= log_w
    Blake2sHash root;
    TreeRef() { memset(root.w, 0, 32); }
    TreeRef(const TreeRef&) = delete;
    TreeRef& operator=(const TreeRef&) = delete;
    TreeRef(TreeRef&& o) noexcept : local(o.local), borrowed(o.borrowed), log_w(o.log_w), n_layers(o.n_layers), top(std::move(o.top)), root(o.root) { o.local = nullptr; }
    TreeRef& operator=(TreeRef&& o) noexcept { if (local && !borrowed) nx_tree_destroy(local); local = o.local; borrowed = o.borrowed; log_w = o.log_w; n_layers = o.n_layers; top = std::move(o.top); root = o.root; o.local = nullptr; return *this; }
    ~TreeRef() { if (local && !borrowed) nx_tree_destroy(local); }
};
// MerkleProver::commit over columns of global log sizes `logs` (whole columns, or row blocks when dist is on): the tree, its root
// downloaded.  Row-sharded: local subtree + all-gather of the W subtree roots + the top log2 W levels on the host.
int merkle_commit_any(nx_ctx* ctx, const Dist& dist, const uint32_t* const* d_cols, const uint32_t* logs, uint32_t n_cols, TreeRef* out);

// ---------------------------------------------------------------- MerkleProver::decommit ------
// Every word a decommitment needs depends only on the query positions and the layer structure, so all decommitments of a proof
// (4 trees + every FRI layer + the FRI witness values) first record their reads in one GatherBatch, ONE nx_gather fetches them (one
// kernel, one device->host copy, one synchronisation), then each plan picks its words up.  Row-sharded: every GPU records the same
// reads, fetches the words it owns, and one all-gather of the (KB-sized) word arrays completes them.
struct GatherBatch {
    const Dist* dist = nullptr;
    std::vector<const uint32_t*> ptrs; std::vector<uint64_t> idx; std::vector<int> owner; std::vector<uint32_t> vals;
    std::vector<std::pair<size_t, uint32_t>> imm;   // words known on the host already (top layers of a sharded tree)
    // owner: the GPU that holds the word, -1 = this GPU (whole / replicated data)
    size_t add(const uint32_t* p, uint64_t i, int own = -1) {
        const bool mine = own < 0 || !dist || own == dist->rank;
        ptrs.push_back(mine ? p : nullptr); idx.push_back(mine ? i : 0); owner.push_back(own);
        return ptrs.size() - 1;
    }
    size_t add_imm(uint32_t v) { ptrs.push_back(nullptr); idx.push_back(0); owner.push_back(-1); imm.push_back({ptrs.size() - 1, v}); return ptrs.size() - 1; }
    void add_column(const ColumnRef& c, uint64_t row) {
        if (c.block && dist && dist->on()) { const uint64_t blk = dist->block(c.log); add(c.ptr, row % blk, (int)(row / blk)); }
        else add(c.ptr, row);
    }
    void add_node(const TreeRef& t, uint32_t layer, uint64_t node) {   // 8 words
        if (t.log_w == 0) { for (int w = 0; w < 8; w++) add(nx_merkle_layer(t.local, layer), node * 8 + w); return; }
        if (layer <= (uint32_t)t.log_w) { for (int w = 0; w < 8; w++) add_imm(t.top[layer][node].w[w]); return; }
        const uint32_t ll = layer - (uint32_t)t.log_w;
        const uint64_t per = (uint64_t)1 << ll;
        for (int w = 0; w < 8; w++) add(nx_merkle_layer(t.local, ll), (node % per) * 8 + w, (int)(node / per));
    }
    int run(nx_ctx* ctx);
};
struct DecommitPlan { size_t first = 0; std::vector<uint8_t> kind; };  // kind: 0 hash word, 1 queried value, 2 column witness
DecommitPlan merkle_decommit_plan(const TreeRef& tree, const std::map<uint32_t, std::vector<size_t>>& queries_per_log, std::vector<ColumnRef> cols, GatherBatch* gb);
void merkle_decommit_fill(const DecommitPlan& plan, const GatherBatch& gb, std::vector<uint32_t>* queried_values, MerkleDecommitment* d);

// ---------------------------------------------------------------- commitment scheme ----------
struct CommitmentTreeProver {
    std::vector<ColumnRef> polys;   // coefficients, commit order (ptr == nullptr: held by owner[c])
    std::vector<int> owner;         // GPU holding polys[c]; -1: this GPU / every GPU
    std::vector<ColumnRef> evals;   // LDE (log = poly log + log_blowup): whole columns or row blocks
    std::vector<DevBuf> bufs;
    TreeRef merkle;
    Blake2sHash root;
    // A tree ADOPTED from a shared one (nx_prover_tree_adopt: the preprocessed tree of a program that is proved again and again): the
    // column references and the Merkle layers point into `backing`, which owns the buffers; this entry owns nothing.  After a commit
    // every buffer of a tree is read-only (coefficients, extensions, nodes), so any number of sessions may read one backing tree.
    std::shared_ptr<CommitmentTreeProver> backing;
    CommitmentTreeProver() {}
    CommitmentTreeProver(CommitmentTreeProver&&) = default;
    CommitmentTreeProver& operator=(CommitmentTreeProver&&) = default;
    CommitmentTreeProver(const CommitmentTreeProver&) = delete;
};

class CommitmentSchemeProver;

// TreeBuilder::{extend_evals, extend_polys, commit}
class TreeBuilder {
  public:
    struct Group {
        DevBuf slab; uint32_t n_cols, log; bool is_evals; uint32_t lo, hi;   // slab: columns [lo, hi) of the group's n_cols (all of them on one GPU)
        // host-resident source (extend_evals_host): the slab is FILLED by the commit, from these host columns, while it transforms
        std::vector<const uint32_t*> host; int coset_order = 0;
        std::vector<std::pair<uint32_t, uint32_t*>> keep;                    // (column of the group, device buffer): its evaluations, cloned before the transform (R4)
    };
    explicit TreeBuilder(CommitmentSchemeProver& c) : cs(c) {}
    // slab: n_cols contiguous columns of 2^log words (bit-reversed evaluations on CanonicCoset(log).circle_domain())
    void extend_evals(DevBuf&& slab, uint32_t n_cols, uint32_t log) { push(std::move(slab), n_cols, log, true, 0, n_cols); }
    // The trace is in HOST memory (what the reference's trace builder hands over, prover/src/trace/trace_builder.rs:19-32): commit()
    // uploads it in column chunks on the copy stream — pinned in place, R3's permutation on the device when coset_order — and runs each
    // chunk's iFFT + LDE as soon as the chunk has arrived, so the PCIe transfer and the transforms overlap (one GPU).  keep: columns whose
    // evaluations are needed after the commit (the logup fractions read them): cloned on arrival, the reference's trace clone (machine.rs:232)
    void extend_evals_host(DevBuf&& slab, uint32_t n_cols, uint32_t log, const uint32_t* const* h_cols, int coset_order,
                           const std::vector<std::pair<uint32_t, uint32_t*>>& keep = {}) {
        push(std::move(slab), n_cols, log, true, 0, n_cols);
        groups.back().host.assign(h_cols, h_cols + n_cols); groups.back().coset_order = coset_order; groups.back().keep = keep;
    }
    // row-sharded prove: the slab holds this GPU's columns [lo, hi) of the group (plan_local_columns)
    void extend_evals_local(DevBuf&& slab, uint32_t n_cols, uint32_t log, uint32_t lo, uint32_t hi) { push(std::move(slab), n_cols, log, true, lo, hi); }
    // coefficients, all n_cols of them on every GPU (the composition polynomial)
    void extend_polys(DevBuf&& slab, uint32_t n_cols, uint32_t log) { push(std::move(slab), n_cols, log, false, 0, n_cols); }
    int commit(Blake2sChannel& channel) { int rc = commit_begin(); return rc != NX_OK ? rc : commit_end(channel); }
    // The two halves of commit, for callers with host work that does not depend on the root (assembling the next stage's descriptors,
    // looking kernels up): commit_begin queues the whole tree build — transforms, hashing, host-feed chunks — and returns; commit_end
    // downloads the root (the one synchronisation of a commit) and mixes it into the channel.  Whatever the caller mixes between the
    // two enters the transcript BEFORE the root, as it would before a one-piece commit.  Row-sharded: the exchanges synchronise with
    // the host anyway; commit_begin does nothing and commit_end is the whole commit.
    int commit_begin();
    int commit_end(Blake2sChannel& channel);
    TreeBuilder(TreeBuilder&&) = default;
    ~TreeBuilder() { for (auto& f : feeds) (void)f->finish(); }
  private:
    void push(DevBuf&& slab, uint32_t n, uint32_t log, bool ev, uint32_t lo, uint32_t hi) { Group g; g.slab = std::move(slab); g.n_cols = n; g.log = log; g.is_evals = ev; g.lo = lo; g.hi = hi; groups.push_back(std::move(g)); }
    int commit_dist(Blake2sChannel& channel);
    CommitmentSchemeProver& cs;
    std::vector<Group> groups;
    std::vector<std::unique_ptr<nx::HostFeed>> feeds;      // one per run with host-resident columns; drained (and the host unpinned) by commit_end
    bool begun = false;
    size_t tree_index = 0;                                   // of the tree commit_begin appended to cs.trees
};

// The share of every group of a tree this GPU transforms: consecutive groups of one size form a run, a run's columns are cut into W
// contiguous ranges.  groups: (n_cols, log) in commit order; out: [lo, hi) per group.  Single GPU: everything.
void plan_local_columns(const std::vector<std::pair<uint32_t, uint32_t>>& groups, const Dist& dist, std::vector<std::pair<uint32_t, uint32_t>>* out);

class CommitmentSchemeProver {
  public:
    nx_ctx* ctx; const nx_twiddles* tw; PcsConfig cfg; Dist dist;
    std::vector<CommitmentTreeProver> trees;
    CommitmentSchemeProver(nx_ctx* c, const nx_twiddles* t, PcsConfig f) : ctx(c), tw(t), cfg(f) {}
    TreeBuilder tree_builder() { return TreeBuilder(*this); }
};

// a non-owning view of a committed tree (same columns, same nodes, same root) that keeps `src` alive
CommitmentTreeProver borrow_tree(const std::shared_ptr<CommitmentTreeProver>& src);
std::vector<uint32_t*> col_ptrs(uint32_t* base, uint32_t n, uint32_t log);
int upload_owned(nx_ctx* ctx, const uint32_t* h, size_t n_words, DevBuf* out);

struct SecureColumn {  // SecureColumnByCoords on device: whole (rows == 2^log) or this GPU's row block
    DevBuf buf; uint32_t log = 0; uint64_t rows = 0; bool block = false; uint32_t* c[4] = {nullptr, nullptr, nullptr, nullptr};
    int alloc(nx_ctx* ctx, uint32_t l) { return alloc_rows(ctx, l, (uint64_t)1 << l, false); }
    int alloc_rows(nx_ctx* ctx, uint32_t l, uint64_t n_rows, bool is_block) {
        log = l; rows = n_rows; block = is_block;
        H_TRY(buf.alloc(ctx, (size_t)4 * n_rows)); fix(); return NX_OK;
    }
    void fix() { for (int k = 0; k < 4; k++) c[k] = buf.p + (size_t)k * rows; }   // after a move of the owning object
    SecureColumn() {}
    SecureColumn(SecureColumn&& o) noexcept : buf(std::move(o.buf)), log(o.log), rows(o.rows), block(o.block) { fix(); }
    SecureColumn& operator=(SecureColumn&& o) noexcept { buf = std::move(o.buf); log = o.log; rows = o.rows; block = o.block; fix(); return *this; }
};

typedef std::vector<std::vector<std::vector<QM31>>> SampledValues;   // tree -> column -> mask
typedef std::vector<std::vector<std::vector<QPt>>> MaskPoints;

// What stwo::prover::prove needs from the components (ComponentProvers): the synthetic machine and recorded AIRs provide it.
struct AirProver {
    virtual ~AirProver() {}
    virtual int compute_composition(CommitmentSchemeProver& cs, QM31 random_coeff, DevBuf* out_polys, uint32_t* out_log) = 0;
    virtual void mask_points(QPt oods, MaskPoints* points) = 0;          // the three trace trees
    virtual QM31 eval_composition_at_point(QPt point, const SampledValues& sv, QM31 random_coeff) = 0;
};

// The columns of one component on its constraint-evaluation domain (log e), one pointer per component column, indexable with the
// GLOBAL row: the committed LDE when e == log_size + log_blowup, else a re-evaluation of the polynomials ("need_to_extend").
// Row-sharded: row blocks (biased; re-evaluated columns cross the links in one all-to-all); columns read at a non-zero mask offset
// (`masked`) are all-gathered whole.
struct EvalDomainCols { std::vector<const uint32_t*> ptrs; std::vector<DevBuf> keep; };
int columns_on_eval_domain(CommitmentSchemeProver& cs, const std::vector<std::pair<uint32_t, uint32_t>>& comp_cols, uint32_t log_size, uint32_t e,
                           const std::vector<char>& masked, EvalDomainCols* out, const std::vector<char>* used = nullptr);
// per evaluation-domain size, this GPU's rows of the accumulation (whole on one GPU)
int composition_accumulator(CommitmentSchemeProver& cs, std::map<uint32_t, SecureColumn>& sub, uint32_t e, SecureColumn** out);
// DomainEvaluationAccumulator::finalize (row-sharded: all-gather of the accumulators first): 4 coefficient columns, replicated
// coef (optional): per log size, contributions already in COEFFICIENT form (the half-domain parts), added after that size's interpolation
int finalize_accumulation(CommitmentSchemeProver& cs, std::map<uint32_t, SecureColumn>& sub, DevBuf* out_polys, uint32_t* out_log,
                          std::map<uint32_t, SecureColumn>* coef = nullptr);
std::vector<uint32_t> vanishing_denominators(uint32_t log_size, uint32_t e);
QM31 coset_vanishing_q(uint32_t n, QPt p);
QPt get_random_point(Blake2sChannel& ch);

struct Lap {   // per-stage wall clock of nx_prove_stats (only when the caller asked for stats)
    nx_ctx* ctx; bool timed; double t0;
    void operator()(double* slot) { if (!timed) return; (void)nx_sync(ctx); double t = now_ms(); *slot += t - t0; t0 = t; }
};
void finish_stats(nx_ctx* ctx, nx_prove_stats* st, double t_start);


// stwo::prover::prove from the point where the three trace trees are committed
int prove_core(nx_ctx* ctx, CommitmentSchemeProver& cs, Blake2sChannel& channel, const PcsConfig& cfg, const nx_twiddles* tw, AirProver& air,
               std::vector<uint32_t>* words, nx_prove_stats* st, Lap& lap);

// ---------------------------------------------------------------- recorded AIRs (the nx_prover session, machine.hip) ---
struct GComponent {
    uint32_t log_size = 0, n_regs = 0, n_constraints = 0;
    uint32_t log_cd = 0;                                  // log constraint-degree bound of this component; 0 = cfg.log_constraint_degree
    std::vector<nx_cinstr> prog;
    std::vector<uint32_t> econsts;
    std::vector<std::pair<uint32_t, uint32_t>> cols;     // component column -> (tree, column in tree)
    std::vector<std::vector<int>> masks;                 // component column -> row offsets sampled
    const nx_air_kernel* kernel = nullptr; nx_air_kernel* owned = nullptr;
    // degree-aware composition (prepare_component_kernels): the parts of the constraints, the columns each reads, their kernels (context-cached)
    // where a part of the constraints is evaluated: on the component's own domain (log_size + bound), on the log_size + 1 domain
    // (degree <= 3), on the first half of the committed log_size + 1 domain (degree <= 2, see compute_composition), or — degree 4 / 5
    // without neighbour rows, bound 2 — on the committed domain plus the first quarter of the log_size + 2 domain (3N + 1 samples)
    enum { ON_FULL = 0, ON_LOW = 1, ON_HALF = 2, ON_QUARTER = 3 };   // ON_QUARTER: committed 2N rows + first quarter of the 4N-point domain (degree 4 / 5, no neighbour rows)
    struct Part { int where = ON_FULL; bool whole = false; std::vector<uint8_t> select; std::vector<char> used; const nx_air_kernel* kernel = nullptr; };
    bool prepared = false; std::vector<Part> parts;
};
// compiled kernels are cached per context and per (program, selection): one compilation serves every proof of an AIR
int cached_air_kernel(nx_ctx* ctx, const GComponent& g, const uint8_t* select, const nx_air_kernel** out);
// everything compute_composition will launch for this component, compiled now (a row-sharded prove votes on the result before its
// first exchange): the whole program, or its low / high parts when the component's bound exceeds 1 and "air.degree_split" is on
int prepare_component_kernels(nx_ctx* ctx, const PcsConfig& cfg, GComponent& g, bool sharded);
struct GenericAir : AirProver {
    nx_ctx* ctx; std::vector<GComponent> comps;
    std::vector<std::vector<std::vector<int>>> offs;     // tree -> column -> union of sampled offsets (first-appearance order)
    std::vector<std::vector<uint32_t>> tree_logs;
    ~GenericAir() override { for (auto& c : comps) if (c.owned) nx_air_kernel_destroy(c.owned); }
    int check(const CommitmentSchemeProver& cs);
    int compute_composition(CommitmentSchemeProver& cs, QM31 random_coeff, DevBuf* out_polys, uint32_t* out_log) override;
    void mask_points(QPt oods, MaskPoints* points) override;
    QM31 eval_composition_at_point(QPt point, const SampledValues& sv, QM31 rc) override;
};

void machine_kernels_release(nx_ctx* ctx);

}  // namespace nxhip

// A committed tree shared between sessions of one context (nx_prover_tree_share / nx_prover_tree_adopt)
struct nx_committed_tree { nx_ctx* ctx; uint32_t log_blowup; int hash_mode; std::shared_ptr<nxhip::CommitmentTreeProver> tree; };

struct nx_prover {
    nx_ctx* ctx; nx_pcs_config ucfg; nxhip::PcsConfig cfg; nx_twiddles* tw = nullptr; uint32_t max_log;
    nxhip::Blake2sChannel channel;
    nxhip::CommitmentSchemeProver* cs = nullptr;
    nx_comm comm_copy; bool has_comm = false;
    struct Run { nxhip::DevBuf slab; uint32_t n_cols, log, lo, hi; };
    std::vector<Run> pending; bool open = false;
    bool proved = false; size_t pre_trees = 0; nxhip::Blake2sChannel pre_channel;   // the state nx_prover_prove found (restored by the next call)
};
