// Internal declarations shared by the translation units of libnexus_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include <map>
#include <memory>
#include "../../include/nexus_hip.h"
#include "field.cuh"

// Per-context tuning / sharding policy (nx_ctx_set_option).  The environment variables of DESIGN.md §6.1 are only the DEFAULTS a new
// context starts from: two contexts of one process can run different settings.
struct nx_options {
    int fft_batch_cols;           // "fft.batch_cols": 2^22-row columns per launch (scaled up for smaller columns)
    int fft_streams;              // "fft.streams": concurrent streams of the column batches (1..4)
    int fri_dist_min_log;         // "fri.dist_min_log": row-sharded prove, FRI layers below this many rows run replicated
    int dist_chunks;              // "dist.chunks": 0 = automatic; else column chunks of a row-sharded commit's exchange
    int air_segment;              // "air.segment": estimated-instruction budget of one generated AIR kernel
    int quotients_coeffs;         // "quotients.coeffs": DEEP quotients of wide size groups from the coefficient columns (single GPU)
    int air_half_domain;          // "air.half_domain": constraints of degree <= 2 are evaluated on HALF of the committed 2N-point domain (single GPU, blowup 2)
    int air_quarter_domain;       // "air.quarter_domain": degree-4/5 constraints are evaluated on the committed 2N rows + the first QUARTER of the 4N-point domain (3N + 1 samples; single GPU, blowup 2, bound 2); 1: only those that read no neighbour row, 2: all (neighbour-read columns on the first half)
    int comm_timeout_ms;          // "comm.timeout_ms": native RCCL transport, longest wait for the peers in one collective (0 = for ever)
    // schedule choices that were A/B'd and settled (DESIGN.md section 6.1 lists the measurements); kept as options — not as getenv calls spread over
    // the kernels' launchers — so that a tool can still flip one on ONE context
    int fft_kmax;                 // "fft.kmax": most layers of a non-FIRST pass (runs of 2^(13-K) words), 1..11
    int fft_fused;                // "fft.fused": fused LDE middle launch (lde_mid_kernel)
    int merkle_fused;             // "merkle.fused": 0 = off, else the smallest log size from which the tree of <= 4 columns of one size (composition tree, FRI layers) gets its leaf hash — for a FRI layer also the fold — and 6 levels in one launch (merkle_fused_kernel)
    int merkle_top;               // "merkle.top": highest level the one-block top launch (merkle_top_kernel) starts from: 2^top nodes, <= 10
    int merkle_subtree; int merkle_pair_levels;           // "merkle.subtree": highest tree level built by the fused subtree launch (0 = one launch per level)
    int commit_pipe_cols;         // "commit.pipe_cols": leaf hashing of finished column groups of this many columns beside the next group's LDE (0 = off)
    int fri_device_channel;       // "fri.device_channel": FRI commit phase with the channel on the device
    int fri_tail;                 // "fri.tail": last FRI layers in one launch: 0 = off, 1 = from 2^11 points (FRI_TAIL_LOG), 2 .. 11 = from 2^that many
    int logup_scan_tiled;         // "logup.scan_tiled": finalize_last as coalesced tiles
    int logup_staged;             // "logup.staged": nx_logup_cols requests every read of a group of fractions up front (values parked in LDS)
    int logup_per_column;         // "logup.per_column": one nx_logup_col launch per column instead of nx_logup_cols
    int machine_logup_program;    // "machine.logup_program": nx_prove_machine builds the interaction trace of every wide-tuple component (NX_LOGUP_TUPLES != 0) from its recorded relation entries (nx_logup_program) instead of nx_logup_cols (same bytes; components with expression numerators always do)
    int machine_queue_trees;      // "machine.queue_trees": nx_prove_machine queues the preprocessed and the main tree builds before fetching the first root (1) or commits them one after the other (0; A/B)
    int machine_reuse_pre;        // "machine.reuse_preprocessed": nx_prove_machine keeps the committed preprocessed tree of a statement shape in the context and adopts it in later proofs (nx_prover_tree_adopt's rule; default 0: every proof commits it afresh, as the reference does)
    int air_degree_split;         // "air.degree_split": constraints of degree <= 3 of a component with a bound > 1 are evaluated on the log_size + 1 domain
};

struct nx_ctx {
    int device;
    nx_options opt;
    hipStream_t stream;
    // FFT column batches are spread over the main stream and these side streams (fork/join with events) so that
    // one batch's memory-heavy pass overlaps another batch's butterfly-heavy pass and launch tails are filled
    hipStream_t side[3];
    hipEvent_t fork_ev, join_ev[3];
    hipStream_t cur;   // stream the FFT launchers enqueue on (== stream outside a forked region)
    hipStream_t copy_stream, perm_stream;   // host-trace feed (HostFeed): PCIe copies / the R3 permutation behind them, next to the commit's kernels
    hipStream_t hash_stream;   // leaf hashing of finished column groups runs here, next to the LDE of the next group
    hipEvent_t hash_ev;
    int hash_mode;
    int n_cus;       // compute units of the device
    std::string err;
    // small device scratch for pointer tables / constants (ring, stream ordered)
    uint8_t* d_scratch;
    size_t scratch_size, scratch_off;
    uint8_t* h_scratch;  // pinned mirror of the ring
    // Bounce buffer of the blocking host <-> device copies (copy_h2d_blocking / copy_d2h_blocking): two pinned halves, allocated on first
    // use.  A copy of >= 1 MiB from / to PAGEABLE memory makes the HIP runtime pin the caller's pages in place and release them lazily;
    // when the caller's allocator hands the same heap pages out again (numpy arrays of one test after the other) the late release tears
    // the mapping down under the next copy: "Memory access fault by GPU ... on address <host heap>" (reproduced 1 run in 4, round 5).
    // Through the bounce buffer no caller page is ever pinned implicitly.
    uint8_t* h_bounce = nullptr; hipEvent_t bounce_ev[2] = {nullptr, nullptr};
    // caching allocator: freed device blocks are kept by exact size and handed back to later nx_alloc calls
    // (a prove repeats the same slab sizes); reuse is safe because all work is ordered on ctx->stream.
    std::multimap<size_t, void*> free_blocks;
    std::map<void*, size_t> live_blocks;
    size_t cached_bytes;
    // pinned host blocks cached by exact size: targets of asynchronous device-to-host copies that must outlive later stage() calls
    // (the partial sums of every OODS request of a proof are collected after ONE synchronisation)
    std::multimap<size_t, void*> free_pinned;
    std::map<void*, size_t> live_pinned;
    // kernel timing (HIP events on ctx->stream), resolved lazily by nx::timing_flush
    bool timing;
    struct Span { hipEvent_t e0, e1; int kind; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> event_pool;
    double kind_ms[4];
    uint64_t kind_bytes[4];
    std::vector<uint32_t> last_claimed;   // nx_machine_claimed_sums: 4 words per component of the last nx_prove_machine
    // The failure a sharded prove is returning was reached by EVERY rank on its own (the vote before the first exchange, an option
    // mismatch, ConstraintsNotSatisfied from the all-gathered sampled values): nobody is left waiting in a collective, so the entry
    // point must not abort the transport — an abort cannot be undone (ncclCommAbort; a broken thread-rank group) and an invalid trace
    // is an input error, not a reason to re-bootstrap a prover farm's communicator.
    bool symmetric_failure = false;
    // Set by the first collective a sharded prove of this context enters (Dist::allgather_host / allgather_dev / alltoallv), cleared by the
    // entry point.  While it is false no peer can be waiting for this rank, so a refusal needs no transport abort; once it is true every
    // failure that is not symmetric aborts — also NX_ERR_ARG, which stage(), TreeBuilder, nx_logup_cols* and air_eval_rows can return from
    // deep inside a prove, on one rank only.
    bool comm_entered = false;
    // "machine.reuse_preprocessed": committed preprocessed trees by statement shape (nxhip::CommitmentTreeProver, type-erased here)
    std::map<std::string, std::shared_ptr<void>> machine_pre_cache;
};
enum { NX_T_LDE = 0, NX_T_MERKLE = 1, NX_T_QUOT = 2, NX_T_OTHER = 3 };

struct nx_twiddles {
    nx_ctx* ctx;
    uint32_t log_half;  // root half coset log size; buffers hold 2^log_half words
    uint32_t* d_tw;
    uint32_t* d_itw;
    uint32_t* d_tw2;    // 2 * twiddle (fits 32 bits): what fft13's m_mul_dbl butterflies consume, saves the per-lane doubling
    uint32_t* d_itw2;
};

struct nx_tree {
    nx_ctx* ctx;
    std::vector<uint32_t*> layers;  // layers[k]: 2^k nodes x 8 words, device
};

namespace nx { void logup_kernels_release(nx_ctx* ctx); }       // air_jit.hip: compiled logup fraction programs cached per context
namespace nxhip { void machine_kernels_release(nx_ctx* ctx); }   // machine.hip: compiled AIR kernels cached per context
namespace nxhip {
double now_ms();
// NX_HOST_PROF=1: wall time of named HOST sections of a prove (a section that contains a synchronisation includes the wait), printed to
// stderr by host_prof_dump at the end of every prove entry point.  A diagnosis aid for the GPU-idle gaps of tools/kernel_sequence.py.
bool host_prof_on();
void host_prof_add(const char* name, double ms);
void host_prof_dump(const char* title);
struct HostSpan {
    const char* name; double t0; bool live;
    explicit HostSpan(const char* n) : name(n), t0(host_prof_on() ? now_ms() : 0), live(host_prof_on()) {}
    void stop() { if (live) { host_prof_add(name, now_ms() - t0); live = false; } }     // a section that ends before its scope does
    ~HostSpan() { stop(); }
    HostSpan(const HostSpan&) = delete;
    HostSpan& operator=(const HostSpan&) = delete;
};
}

namespace nx {

extern thread_local std::string g_last_error;

int set_err(nx_ctx* ctx, int code, const std::string& msg);
int hip_fail(nx_ctx* ctx, hipError_t e, const char* what, const char* file, int line);

#define NX_HIP(ctx, call)                                                         \
    do {                                                                          \
        hipError_t e__ = (call);                                                  \
        if (e__ != hipSuccess) return nx::hip_fail((ctx), e__, #call, __FILE__, __LINE__); \
    } while (0)
#define NX_TRY(call)                  \
    do {                              \
        int rc__ = (call);            \
        if (rc__ != NX_OK) return rc__; \
    } while (0)
#define NX_LAUNCH_CHECK(ctx) NX_HIP(ctx, hipGetLastError())

// Every extern "C" entry that takes a context (or a handle that owns one) makes the context's device current for the call:
// hipMalloc, hipHostRegister, hipFuncSetAttribute and kernel launches act on the CALLING THREAD's current device, which is 0 in a
// fresh worker thread whatever device the context was created on.  The previous device is restored on return.
struct DeviceGuard {
    int prev = -1; bool switched = false;
    explicit DeviceGuard(const nx_ctx* c) {
        if (!c) return;
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; }
        if (prev != c->device) { switched = hipSetDevice(c->device) == hipSuccess && prev >= 0; }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define NX_GUARD(ctxexpr) nx::DeviceGuard nx_device_guard__(ctxexpr)

// A set of equally sized columns: either base + c*stride (words) or a device pointer table.
struct ColSet {
    uint32_t* base;
    uint64_t stride;
    uint32_t* const* table;  // device memory, may be null
    __device__ __forceinline__ uint32_t* col(uint32_t c) const { return table ? table[c] : base + (uint64_t)c * stride; }
};

// Global-memory accessors for device code.  Column pointers reach the kernels inside by-value structs or pointer tables, so
// clang types them generic and emits flat_load/flat_store: those also count on lgkmcnt (every later s_load or LDS wait then
// waits for ALL outstanding column loads — the Merkle leaf kernel issued its 16 column loads one at a time) and are ordered
// against LDS traffic.  These helpers cast to address space 1 (global_load/global_store).
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define NX_GLOBAL_AS __attribute__((address_space(1)))
typedef uint32_t nx_v4u32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t gld(const uint32_t* p) { return *(NX_GLOBAL_AS const uint32_t*)p; }
// word at (uniform base) + a 32-bit BYTE offset: the `global_load_dword v, v_off, s[base]` form — no 64-bit address per load (a column is
// at most 2^30 words, so the offset of any row fits 32 bits)
__device__ __forceinline__ uint32_t gld_off(const uint32_t* base, uint32_t byte_off) { return *(NX_GLOBAL_AS const uint32_t*)((NX_GLOBAL_AS const char*)base + byte_off); }
__device__ __forceinline__ uint4 gld4(const uint32_t* p) { const nx_v4u32 v = *(NX_GLOBAL_AS const nx_v4u32*)p; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void gst(uint32_t* p, uint32_t v) { *(NX_GLOBAL_AS uint32_t*)p = v; }
__device__ __forceinline__ void gst4(uint32_t* p, uint4 v) { nx_v4u32 w = {v.x, v.y, v.z, v.w}; *(NX_GLOBAL_AS nx_v4u32*)p = w; }
#endif

// Row-sharded proves (one contiguous block of the bit-reversed rows per GPU): kernels index every column with the GLOBAL row, the
// host hands them the block pointer moved back by the block's first row.  Only addresses p[row_begin ...] are ever dereferenced.
template <class T> static inline T* bias_rows(T* p, uint64_t row_begin) { return p ? (T*)((uintptr_t)p - row_begin * sizeof(T)) : p; }

// Builds a ColSet from a host array of device pointers: constant stride is detected, otherwise the
// table is staged into the context's device scratch ring (stream-ordered).
int make_colset(nx_ctx* ctx, const uint32_t* const* h_ptrs, uint32_t n, ColSet* out);
// Stage arbitrary bytes into device scratch; returns device pointer (valid until the ring wraps).
int stage(nx_ctx* ctx, const void* h_src, size_t bytes, void** d_out);
int host_alloc(nx_ctx* ctx, size_t bytes, void** h_out);       // pinned host block owned by the caller until host_free (cached by size)
void host_free(nx_ctx* ctx, void* p);
// nx_eval_at_points in two phases, so that a prover samples every tree and size with ONE synchronisation: enqueue launches the
// kernels of one request and records where its partial sums will land; collect waits once and reduces them into the outputs.
struct EvalJob { uint8_t* blob; const uint32_t* h_part; uint32_t np, n_chunks; std::vector<uint32_t> evals; uint32_t* h_out; };
int eval_at_points_enqueue(nx_ctx* ctx, const uint32_t* const* d_polys, uint32_t log_size, const uint32_t* poly_idx, const uint32_t* h_points, uint32_t n_evals,
                           uint32_t* h_out, std::vector<EvalJob>* jobs);
int eval_at_points_collect(nx_ctx* ctx, std::vector<EvalJob>* jobs);
// cached device allocation (bytes); dev_free returns the block to the cache (no device sync)
int dev_alloc(nx_ctx* ctx, size_t bytes, void** out);
void dev_free(nx_ctx* ctx, void* p);
void dev_cache_release(nx_ctx* ctx);

// n_cols columns of n_words words each, dst[k] <- src[k], one launch (ctx.hip)
int copy_columns(nx_ctx* ctx, uint32_t* const* h_dst, const uint32_t* const* h_src, uint32_t n_cols, size_t n_words);
// A host-resident trace fed to the device WHILE the commitment transforms what has arrived (fft.hip): columns
// are pinned in place, copied on ctx->copy_stream and — coset_order — permuted into bit-reversed circle-domain order on
// ctx->perm_stream (R3 fused); chunk() returns an event that fires when its columns are in place, so the consumer orders its own
// stream behind it (hipStreamWaitEvent) and the host never waits.  finish() drains both streams and unpins (also on the error paths).
// host ranges their owner pinned (nx_host_pin; process-wide like the driver's registration): the host-column entry points neither pin nor
// unpin inside them
bool host_pinned_by_owner(const void* p, size_t bytes);
// blocking copies between caller (possibly pageable) host memory and the device: complete on return, never pin the caller's pages
int copy_h2d_blocking(nx_ctx* ctx, void* d_dst, const void* h_src, size_t bytes, hipStream_t stream = nullptr);   // stream: null = the context's
int copy_d2h_blocking(nx_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
// stream-ordered upload of caller host memory that may be freed as soon as this returns (through the staging ring, in chunks)
int upload_async_staged(nx_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
struct HostFeed {
    nx_ctx* ctx = nullptr; int coset_order = 0; uint32_t log = 0;
    uint32_t* d_tmp[2] = {nullptr, nullptr}; hipEvent_t copied[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
    uint64_t n_done = 0;
    std::vector<const uint32_t*> pinned; std::vector<hipEvent_t> events;
    int begin(nx_ctx* c, uint32_t log_size, int coset);
    int chunk(const uint32_t* const* h_cols, uint32_t* const* d_cols, uint32_t n_cols, hipEvent_t* ready);
    int finish();
    ~HostFeed() { (void)finish(); }
};
int transpose_blocks(nx_ctx* ctx, uint32_t* full, uint64_t col_stride, uint32_t* blocks, uint32_t n_cols, uint64_t rows, uint32_t world, bool unpack);

// Event-pair span on ctx->stream, recorded only when ctx->timing is on; resolved by timing_flush.
struct KTimer {
    nx_ctx* ctx; int idx; hipStream_t stream;
    KTimer(nx_ctx* c, int kind, uint64_t algorithmic_bytes, hipStream_t on_stream = nullptr);
    ~KTimer();
};
void timing_flush(nx_ctx* ctx);  // synchronises the stream and folds spans into kind_ms[]
void timing_reset(nx_ctx* ctx);

// ---- kernels' host-side launchers (device pointers; all stream-ordered on ctx->stream) ----
int fft_interpolate(nx_ctx* ctx, const nx_twiddles* tw, ColSet cols, uint32_t n_cols, uint32_t log_size);
int fft_evaluate(nx_ctx* ctx, const nx_twiddles* tw, ColSet polys, uint32_t n_cols, uint32_t log_size,
                 uint32_t log_expand, ColSet out);

// Pipelined tree build (merkle.hip): the leaf layer is a per-row Blake2s chain over the largest columns in commit order,
// so finished column groups can be absorbed (on ctx->hash_stream) while the main stream already transforms the next group.
// FRI tail (merkle.hip): the last FRI layers (line layers of <= 2^FRI_TAIL_LOG points) committed and folded in one launch with the
// channel on the device.
constexpr int FRI_TAIL_LOG = 11, FRI_TAIL_MAX_LAYERS = 16;   // 2^12 and up: one CU is slower than the per-layer launches (measured)
int tree_alloc(nx_ctx* ctx, uint32_t max_log, nx_tree** out);
// FRI channel state on the device: words [0,8) digest, [8] n_sent, then one record per committed layer at 9 + 12 j: root[8], alpha[4].
constexpr int FRI_STATE_HEAD = 9, FRI_STATE_REC = 12;
int merkle_commit_fused(nx_ctx* ctx, const uint32_t* const* d_cols, uint32_t n_cols, uint32_t log, nx_tree** out);   // <= 4 columns of >= 2^12 rows: leaf hash + 6 levels in one launch
int fri_fold_commit_fused(nx_ctx* ctx, const nx_twiddles* tw, const uint32_t* const* d_src4, uint32_t src_log, const uint32_t* d_alpha, uint32_t* const* d_dst4, nx_tree** out);   // fold_line + leaf hash + 6 levels of one FRI layer (>= 2^12 points) in one launch
int fri_channel_step(nx_ctx* ctx, uint32_t* d_state, const uint32_t* d_root, int j);   // mix_root(root) + draw_secure_felt -> record j
int fri_tail(nx_ctx* ctx, const nx_twiddles* tw, uint32_t* const* evals, uint32_t* const* trees, int n_layers, int log0, uint32_t* d_state, int j0);
int fold_circle_dev(nx_ctx* ctx, const nx_twiddles* tw, uint32_t* const* d_dst4, const uint32_t* const* d_src4, uint32_t src_log, const uint32_t* d_alpha);
int fold_line_dev(nx_ctx* ctx, const nx_twiddles* tw, const uint32_t* const* d_src4, uint32_t src_log, const uint32_t* d_alpha, uint32_t* const* d_dst4);
int fold_circle_rows(nx_ctx* ctx, const nx_twiddles* tw, uint32_t* const* d_dst4, const uint32_t* const* d_src4, uint32_t src_log, const uint32_t alpha[4], uint32_t i0, uint32_t n);
int fold_line_rows(nx_ctx* ctx, const nx_twiddles* tw, const uint32_t* const* d_src4, uint32_t src_log, const uint32_t alpha[4], uint32_t* const* d_dst4, uint32_t i0, uint32_t n);
int accumulate_quotients_rows(nx_ctx* ctx, uint32_t log_size, const uint32_t* const* d_cols, uint32_t n_cols, const uint32_t random_coeff[4], uint32_t n_batches,
                              const uint32_t* points, const uint32_t* batch_counts, const uint32_t* col_idx, const uint32_t* values, uint32_t* const* d_out4,
                              uint64_t row_begin, uint64_t n_rows);
int twiddles_first_part(nx_ctx* ctx, const nx_twiddles* tw, uint32_t n, uint32_t depth, nx_twiddles** out);   // fft.hip: the N-point domain that is the first 1 / 2^depth of the 2^depth N-point one
inline int twiddles_first_half(nx_ctx* ctx, const nx_twiddles* tw, uint32_t n, nx_twiddles** out) { return twiddles_first_part(ctx, tw, n, 1, out); }
int accumulate_quotients_coeffs(nx_ctx* ctx, const nx_twiddles* tw, uint32_t log_size, uint32_t log_coef, const uint32_t* const* d_polys, uint32_t n_cols,
                                const uint32_t random_coeff[4], uint32_t n_batches, const uint32_t* points, const uint32_t* batch_counts, const uint32_t* col_idx,
                                const uint32_t* values, uint32_t* const* d_out4);

struct TreePipe {
    nx_tree* tree = nullptr;
    uint32_t max_log = 0, total_leaf_cols = 0, absorbed = 0;
    std::vector<const uint32_t*> pending;   // columns handed in but not yet hashed (kept until a 16-column block is complete)
    bool any_launch = false;
    bool side_stream = false;              // hash on ctx->hash_stream behind the next column group's LDE (NX_PIPE_COLS); else in stream order
};
int tree_pipe_begin(nx_ctx* ctx, uint32_t max_log, uint32_t total_leaf_cols, TreePipe* tp);
int tree_pipe_absorb(nx_ctx* ctx, TreePipe* tp, const uint32_t* const* d_cols, uint32_t n_cols, bool flush);
// smaller columns (log < max_log) in commit order with their sizes; hands the finished tree to *out
int tree_pipe_finish(nx_ctx* ctx, TreePipe* tp, const uint32_t* const* d_small_cols, const uint32_t* small_logs, uint32_t n_small, nx_tree** out);
int fft_lde(nx_ctx* ctx, const nx_twiddles* tw, ColSet cols, uint32_t n_cols, uint32_t log_size, uint32_t log_expand, ColSet out);

// fork/join of the side streams around a loop over independent column batches
int streams_fork(nx_ctx* ctx, int n_streams);
void streams_pick(nx_ctx* ctx, int batch_index, int n_streams);
int streams_join(nx_ctx* ctx, int n_streams);

// fast path for transforms of >= 2^13 points (fft13.hip)
int fft13_interpolate(nx_ctx* ctx, const nx_twiddles* tw, ColSet cols, uint32_t n_cols, int n);
int fft13_evaluate(nx_ctx* ctx, const nx_twiddles* tw, ColSet polys, uint32_t n_cols, int log_in, int n, ColSet out);
int fft13_lde(nx_ctx* ctx, const nx_twiddles* tw, ColSet cols, uint32_t n_cols, int n, ColSet out);   // blow-up 2, n >= 14: middle passes fused


}  // namespace nx
