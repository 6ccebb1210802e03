/*
 * nexus_hip.h — C ABI of libnexus_hip.so, the MI355X (gfx950) backend for the Nexus zkVM
 * commit-and-prove hot path.
 *
 * The reference (nexus-xyz/nexus-zkvm) has no FFI today: it instantiates Stwo's CPU `SimdBackend`
 * directly (reference prover/src/machine.rs:16,186,203,286; prover2/machine/src/prove.rs:12,53,65,124).
 * The drop-in boundary is therefore the family of Stwo backend traits the reference relies on; each
 * entry point below names the trait method it replaces and the reference call site that reaches it.
 * A Rust `HipBackend` shim that binds these symbols is shown in INTEGRATION.md.
 *
 * Conventions
 *  - All field data are canonical M31 values in [0, 2^31-1), one uint32_t each, little endian.
 *    Secure-field (QM31) data are 4 coordinate words (a + bi) + (c + di)u  ->  {a, b, c, d};
 *    secure *columns* are 4 separate coordinate columns (Stwo `SecureColumnByCoords`).
 *  - Pointers named `d_*` / documented "device" are HIP device pointers (from nx_alloc, or any
 *    hipMalloc'd / torch-allocated memory on the context's device).  Pointer *arrays* such as
 *    `const uint32_t* const* cols` are HOST arrays holding device pointers.
 *  - Every function returns NX_OK (0) or a negative error code; nx_last_error() gives the text.
 *    Allocation failure is an error code, not a panic (reference: vec![] aborts, trace_builder.rs:29).
 *  - A context is single-threaded at the protocol level like the reference's prover
 *    (one &mut Blake2sChannel, machine.rs:197); all work is ordered on the context's HIP stream and
 *    is asynchronous; only functions that return data to the host synchronise.
 *  - Results are exact integers: bit-identical regardless of scheduling.
 */
#ifndef NEXUS_HIP_H
#define NEXUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NX_OK 0
#define NX_ERR_HIP (-1)       /* HIP runtime error (text in nx_last_error)                         */
#define NX_ERR_ARG (-2)       /* invalid argument                                                   */
#define NX_ERR_OOM (-3)       /* device or host allocation failed                                   */
#define NX_ERR_PROTOCOL (-4)  /* ProvingError::ConstraintsNotSatisfied / FRI invalid degree         */
#define NX_ERR_NO_DEVICE (-5) /* no gfx950 device visible — there is NO CPU fallback                */

/* Merkle hash rule (unverifiable upstream detail kept switchable). */
#define NX_HASH_BLAKE2S 0      /* standard Blake2s-256 of (left ‖ right ‖ column values)            */
#define NX_HASH_BLAKE2S_RAW0 1 /* zero-state raw compression chaining, t = f = 0 (older Stwo)       */

/* FRI circle-column folding alpha. */
#define NX_FRI_ALPHA_PREV 0  /* fold circle columns with the previous layer's alpha (newer Stwo)    */
#define NX_FRI_ALPHA_FIRST 1 /* fold all circle columns with the first alpha (older Stwo)           */

typedef struct nx_ctx nx_ctx;
typedef struct nx_twiddles nx_twiddles;
typedef struct nx_tree nx_tree;

/* ---------------------------------------------------------------- context ------------------- */
int nx_ctx_create(int device, nx_ctx** out);
void nx_ctx_destroy(nx_ctx* ctx);
const char* nx_last_error(const nx_ctx* ctx); /* ctx may be NULL: last global error               */
int nx_ctx_set_hash_mode(nx_ctx* ctx, int mode);
/* Per-context policy and tuning.  The NX_* environment variables only seed a new context's defaults; what a
 * context does is decided by its own options, so two contexts of one process may differ.  Names: "fft.batch_cols" (2^22-row columns per launch of the LDE),
 * "fft.streams" (1..4), "comm.timeout_ms" (both library transports — native RCCL and the in-process one: the longest a rank waits for its peers in one collective before it
 * aborts the communicator / breaks the group and fails the prove, default 120000; 0 = wait for ever), "fri.dist_min_log", "dist.chunks" and "air.degree_split" (1: degree-aware composition, see
 * nx_air_constraint_degrees) — row-sharded prove: every GPU of a proof must use the same values of these three, they shape the
 * exchanges —, "air.segment" (instruction budget of one generated AIR kernel), "quotients.coeffs" (1: the DEEP quotients of a wide
 * size group are accumulated from the coefficient columns — half the bytes at blowup 2; one GPU only), "air.half_domain" (1: constraints
 * of degree <= 2 are evaluated on the first half of the committed 2N-point domain; one GPU, blowup 2), "air.quarter_domain" (constraints
 * of degree 4 / 5 are evaluated on the committed 2N rows plus the first quarter of the 4N-point domain — 3N + 1 samples instead of 4N;
 * 1: only those that read no neighbour row, 2 (default): all of them, the columns read at a neighbour row — the reference's Pc /
 * IsPadding, prover/src/column.rs:13-20 — evaluated on the first HALF of that domain, which holds the neighbours of its first quarter;
 * one GPU, blowup 2, component bound 2).  Kernel-shape switches kept for A/B measurement (defaults are the
 * measured best): "fft.kmax" (most layers of a non-first FFT pass, 1..11), "fft.fused" (fused middle launch of the LDE), "merkle.subtree"
 * (highest level built by the fused sub-tree launch; 0 = one launch per level), "merkle.top" (the level, 1..10, from which ONE block
 * builds the rest of a tree), "merkle.pair_levels" (two node-only levels per launch
 * above it), "commit.pipe_cols" (leaf hashing beside the LDE in
 * groups of this many columns; 0 = off), "fri.device_channel", "fri.tail" (0 = off, 1 = the FRI layers of <= 2^11 points in one launch,
 * 2..11 = from 2^that many points), "logup.scan_tiled", "logup.per_column", "logup.staged" (nx_logup_cols requests
 * every read of a group of 8 fractions before it uses the first value; 0 = reads where they are used).
 * Unknown names and out-of-range values are NX_ERR_ARG.
 * None of them changes a result: proofs, roots and transforms are bit-identical under every setting. */
int nx_ctx_set_option(nx_ctx* ctx, const char* name, int64_t value);
int nx_ctx_get_option(const nx_ctx* ctx, const char* name, int64_t* value);
int nx_sync(nx_ctx* ctx);
/* Hands the context's cached device blocks (freed columns kept for the next prove of the same shape) back to the driver:
 * another context, process or library on the same GPU can then use that memory.  Synchronises the context. */
int nx_ctx_trim(nx_ctx* ctx);
void* nx_ctx_stream(nx_ctx* ctx); /* hipStream_t the context launches on                       */
const char* nx_version(void);

/* ------------------------------------------- columns: Column / ColumnOps ------------------- */
/* Column::zeros / from_iter / to_cpu (reference prover2/trace/src/builder.rs:103,
 * prover2/trace/src/component.rs:40-44; BaseColumn in prover/src/trace/utils.rs:100). */
int nx_alloc(nx_ctx* ctx, size_t n_words, uint32_t** d_out);
int nx_free(nx_ctx* ctx, uint32_t* d_ptr);
int nx_memset_zero(nx_ctx* ctx, uint32_t* d_ptr, size_t n_words);
int nx_upload(nx_ctx* ctx, uint32_t* d_dst, const uint32_t* h_src, size_t n_words);
int nx_download(nx_ctx* ctx, uint32_t* h_dst, const uint32_t* d_src, size_t n_words);
/* Column::clone on device (the reference clones whole traces before committing them: prover/src/machine.rs:210-214,232;
 * prover2/trace/src/component.rs:75).  Stream-ordered; src and dst must not overlap. */
int nx_copy(nx_ctx* ctx, uint32_t* d_dst, const uint32_t* d_src, size_t n_words);
/* Gather scattered words: out[i] = d_ptrs[i][index[i]] (decommitment reads, MerkleProver::decommit). */
int nx_gather(nx_ctx* ctx, const uint32_t* const* d_ptrs, const uint64_t* index, size_t n, uint32_t* h_out);

/* K1 — ColumnOps<M31>::bit_reverse_column (reference prover/src/trace/utils.rs:101,
 * prover2/trace/src/utils.rs:109).  In place. */
int nx_bit_reverse(nx_ctx* ctx, uint32_t* d_col, uint32_t log_size);

/* R3 — finalize_columns fused on device (reference prover/src/trace/utils.rs:94-106 +
 * prover/src/trace/utils_external.rs:24-39): natural coset order -> bit-reversed circle-domain
 * order, n_cols columns of 2^log_size words; src and dst must not alias. */
int nx_finalize_columns(nx_ctx* ctx, const uint32_t* const* d_src_natural, uint32_t* const* d_dst,
                        uint32_t n_cols, uint32_t log_size);
/* Same, source on the host (the trace the AIR layer filled, TracesBuilder cols, trace_builder.rs:19-32). */
int nx_upload_coset_order(nx_ctx* ctx, const uint32_t* h_natural, uint32_t log_size, uint32_t* d_dst);
/* The same for a whole host-resident trace: every host column is pinned in place, streamed over
 * PCIe on a side stream and permuted on device behind the copy (coset_order != 0) or stored as is (the host already holds
 * bit-reversed circle-domain order).  Blocking: the host columns are free on return.  Replaces the per-column CPU passes of
 * finalize_columns + into_circle_evaluation's clones (reference prover/src/trace/utils.rs:94-106, trace_builder.rs:156-164). */
int nx_upload_columns(nx_ctx* ctx, const uint32_t* const* h_cols, uint32_t n_cols, uint32_t log_size,
                      uint32_t* const* d_cols, int coset_order);
/* A trace buffer that is reused from proof to proof is pinned ONCE by its owner instead of once per upload: nx_host_pin registers
 * [h, h + bytes) with the driver (hipHostRegister; the memory stays where it is), nx_host_unpin releases it.  Every entry point that takes
 * host columns (nx_upload_columns, nx_prover_tree_commit_host, nx_prove_machine_host) recognises pinned columns and skips its own
 * pin / unpin of them — the per-proof cost of pinning 6 GB of trace is what separates 135 ms from the PCIe floor in bench.py's
 * host_trace block.  Pinning is an optimisation only: results never depend on it.  Pinned ranges must not overlap (NX_ERR_ARG; the
 * book is process-wide, any context recognises them); nx_host_unpin takes the start of a pinned range and waits for the calling
 * context's copies out of it. */
int nx_host_pin(nx_ctx* ctx, const void* h, size_t bytes);
int nx_host_unpin(nx_ctx* ctx, const void* h);

/* ----------------------------------------------------- K2-K4, K7: PolyOps ------------------ */
/* K2 — PolyOps::precompute_twiddles(CanonicCoset::new(log_half_coset + 1).half_coset())
 * (reference prover/src/machine.rs:186-194, prover2/machine/src/prove.rs:53-57, verify.rs:121). */
int nx_twiddles_create(nx_ctx* ctx, uint32_t log_half_coset, nx_twiddles** out);
void nx_twiddles_destroy(nx_twiddles* tw);
/* Test/debug access: copies the 2^log_half_coset forward and inverse twiddles to the host. */
int nx_twiddles_download(nx_ctx* ctx, const nx_twiddles* tw, uint32_t* h_tw, uint32_t* h_itw);

/* K3 — PolyOps::interpolate_columns via TreeBuilder::extend_evals (reference
 * prover/src/machine.rs:209,226,232,235,250,260).  In place: bit-reversed evaluations on
 * CanonicCoset(log_size).circle_domain() -> coefficients. */
int nx_interpolate_batch(nx_ctx* ctx, const nx_twiddles* tw, uint32_t* const* d_cols, uint32_t n_cols,
                         uint32_t log_size);
/* K4 — PolyOps::evaluate_polynomials via TreeBuilder::commit -> CommitmentTreeProver::new
 * (reference prover/src/machine.rs:228,237,263).  Coefficients (2^log_size) -> bit-reversed
 * evaluations on CanonicCoset(log_size + log_expand).circle_domain(); out-of-place. */
int nx_evaluate_batch(nx_ctx* ctx, const nx_twiddles* tw, const uint32_t* const* d_polys, uint32_t n_cols,
                      uint32_t log_size, uint32_t log_expand, uint32_t* const* d_out);
/* K3+K4 fused — what TreeBuilder::extend_evals followed by TreeBuilder::commit computes for one group of
 * equally sized columns (reference prover/src/machine.rs:232-237): d_cols holds bit-reversed evaluations on
 * entry and the coefficients on return; d_lde receives the evaluations on the blown-up domain.  Each column
 * batch runs iFFT and FFT back to back so the coefficients stay on chip between the two transforms. */
int nx_lde_batch(nx_ctx* ctx, const nx_twiddles* tw, uint32_t* const* d_cols, uint32_t n_cols, uint32_t log_size,
                 uint32_t log_blowup, uint32_t* const* d_lde);
/* K7 — PolyOps::eval_at_point (inside stwo::prover::prove, reference machine.rs:286):
 * out[i] = polys[poly_idx[i]] evaluated at points[i] (8 words x‖y each), all polys of one log_size. */
int nx_eval_at_points(nx_ctx* ctx, const uint32_t* const* d_polys, uint32_t log_size, const uint32_t* poly_idx,
                      const uint32_t* h_points, uint32_t n_evals, uint32_t* h_out /* 4 words each */);

/* ------------------------------------------- K5/K6: MerkleOps<Blake2sMerkleHasher> --------- */
/* MerkleProver::commit -> MerkleOps::commit_on_layer per layer (reference machine.rs:228,237,263).
 * Columns in commit order (stable-sorted by size inside); log_sizes are the column (LDE) sizes. */
int nx_merkle_commit(nx_ctx* ctx, const uint32_t* const* d_cols, const uint32_t* log_sizes, uint32_t n_cols,
                     nx_tree** out);
int nx_merkle_root(nx_ctx* ctx, const nx_tree* tree, uint8_t root[32]); /* roots(): machine.rs:411 */
uint32_t nx_merkle_n_layers(const nx_tree* tree);
/* Device pointer of layer k (2^k nodes x 8 words); layer 0 is the root. */
const uint32_t* nx_merkle_layer(const nx_tree* tree, uint32_t k);
void nx_tree_destroy(nx_tree* tree);

/* MerkleOps<Blake2sMerkleHasher>::commit_on_layer itself — ONE layer: node i = H(prev[2i] ‖ prev[2i+1] ‖ cols[0][i] ‖ cols[1][i] ...);
 * d_prev_layer: the 2^(log_size+1) nodes of the layer below (8 words each) or NULL for the leaf layer; d_cols: the columns of
 * 2^log_size words injected at this layer (commit order); d_out: 2^log_size nodes.  Hash rule = nx_ctx_set_hash_mode.
 * (MerkleProver::commit, the loop over layers, is nx_merkle_commit.) */
int nx_merkle_commit_on_layer(nx_ctx* ctx, uint32_t log_size, const uint32_t* d_prev_layer, const uint32_t* const* d_cols, uint32_t n_cols,
                              uint32_t* d_out);
/* MerkleProver::decommit(queries_per_log_size, columns) (inside stwo::prover::prove, reference machine.rs:286-290): the values of
 * `columns` (commit order, LDE log sizes) at the queried rows and the MerkleDecommitment {hash_witness, column_witness}.
 * One (query_logs[i], query_counts[i]) pair per queried layer; `queries` holds every layer's sorted, distinct positions back to
 * back.  Outputs are malloc'd (nx_free_host); hash_witness: 8 words per hash. */
int nx_merkle_decommit(nx_ctx* ctx, const nx_tree* tree, const uint32_t* const* d_cols, const uint32_t* log_sizes, uint32_t n_cols,
                       const uint32_t* query_logs, const uint32_t* query_counts, uint32_t n_query_logs, const uint64_t* queries,
                       uint32_t** queried_values, size_t* n_queried_values, uint32_t** hash_witness, size_t* n_hashes,
                       uint32_t** column_witness, size_t* n_column_witness);

/* Column-sharded commitment (BASELINE config #4).  A tree leaf is one sequential Blake2s chain
 * over ALL columns of the tree (16 columns = one 64-byte block), so column shards of one tree are chained:
 * the GPU that owns columns [col_offset, col_offset + n_cols) of a layer with total_cols columns continues the
 * 8-word chaining state of rows [row_begin, row_begin + n_rows) received from the previous GPU (d_state_in == NULL
 * for the shard that starts at column 0) and hands its state on; col_offset must be a multiple of 16.  The shard
 * that holds the last column applies the hash finalisation, so its d_state_out rows are the leaf digests.
 * d_state_in / d_state_out: n_rows x 8 words, indexed by row - row_begin (may alias).  d_cols: full columns
 * of 2^log_size words. */
int nx_merkle_leaf_chain(nx_ctx* ctx, const uint32_t* const* d_cols, uint32_t n_cols, uint32_t log_size,
                         uint32_t col_offset, uint32_t total_cols, const uint32_t* d_state_in, uint32_t* d_state_out,
                         uint64_t row_begin, uint64_t n_rows);
/* MerkleProver over already hashed leaves: builds the inner layers above 2^log_size leaf digests (8 words each,
 * copied in) — the part of a sharded commit the last GPU of the chain runs before broadcasting the root. */
int nx_merkle_from_leaves(nx_ctx* ctx, const uint32_t* d_leaf_digests, uint32_t log_size, nx_tree** out);

/* ------------------------------------------------------------- K8: QuotientOps ------------- */
/* QuotientOps::accumulate_quotients (inside prove).  One size group: n_cols columns of
 * 2^log_size words; sample batches flattened: batch b has batch_counts[b] (column, value) pairs,
 * column indices in col_idx, sampled values (4 words each) in values, point (8 words) in points.
 * random_coeff is the DEEP alpha (4 words).  d_out4: 4 coordinate columns of 2^log_size words. */
int nx_accumulate_quotients(nx_ctx* ctx, uint32_t log_size, const uint32_t* const* d_cols, uint32_t n_cols,
                            const uint32_t random_coeff[4], uint32_t n_batches, const uint32_t* points,
                            const uint32_t* batch_counts, const uint32_t* col_idx, const uint32_t* values,
                            uint32_t* const* d_out4);

/* Column-sharded variant: the quotient is a sum over columns, so each GPU accumulates the entries whose
 * columns it holds (entry_local[k] != 0; col_idx[k] then indexes ITS d_cols) with the alpha power of the entry's GLOBAL
 * position, exactly one GPU adds the line terms of all entries (include_line_terms), and the partial results are summed
 * mod p across GPUs (nx_comm.allreduce_m31).  Batches, counts, values describe ALL entries, on every GPU. */
int nx_accumulate_quotients_partial(nx_ctx* ctx, uint32_t log_size, const uint32_t* const* d_cols, uint32_t n_cols,
                                    const uint32_t random_coeff[4], uint32_t n_batches, const uint32_t* points,
                                    const uint32_t* batch_counts, const uint32_t* col_idx, const uint32_t* values,
                                    const uint8_t* entry_local, int include_line_terms, uint32_t* const* d_out4);

/* --------------------------------------------------------------- K9: FriOps ---------------- */
/* FriOps::fold_circle_into_line: dst (line evaluation, 2^(src_log-1)) = dst*alpha^2 + fold(src). */
int nx_fold_circle_into_line(nx_ctx* ctx, const nx_twiddles* tw, uint32_t* const* d_dst4,
                             const uint32_t* const* d_src4, uint32_t src_log, const uint32_t alpha[4]);
/* FriOps::fold_line: src on the line domain half_odds(src_log + n_doublings) doubled n_doublings
 * times (log size src_log) -> dst of log size src_log - 1. */
int nx_fold_line(nx_ctx* ctx, const nx_twiddles* tw, const uint32_t* const* d_src4, uint32_t src_log,
                 uint32_t n_doublings, const uint32_t alpha[4], uint32_t* const* d_dst4);

/* ------------------------------------------- the rest of the Backend supertraits -------------------
 * Called by stwo::prover::prove through `B: Backend` (reference machine.rs:286-290, prove.rs:124-128); not hot spots here (the
 * composition accumulator and the logup kernels fuse what they need) but required for a per-op HipBackend to type-check. */
/* AccumulationOps::accumulate: dst[k][i] += src[k][i] on 4 coordinate columns of 2^log_size words. */
int nx_secure_accumulate(nx_ctx* ctx, uint32_t* const* d_dst4, const uint32_t* const* d_src4, uint32_t log_size);
/* AccumulationOps::generate_secure_powers: out = [1, felt, felt^2, ...] (host; 4 words each). */
int nx_generate_secure_powers(const uint32_t felt[4], uint32_t n_powers, uint32_t* out);
/* FieldOps<BaseField>::batch_inverse / FieldOps<SecureField>::batch_inverse: element-wise inverses (inputs must be non-zero,
 * as Stwo requires); src and dst may alias. */
int nx_batch_inverse_m31(nx_ctx* ctx, const uint32_t* d_src, uint32_t* d_dst, size_t n);
int nx_batch_inverse_qm31(nx_ctx* ctx, const uint32_t* const* d_src4, uint32_t* const* d_dst4, size_t n);
/* ColumnOps<SecureField>::bit_reverse_column (a SecureColumnByCoords is 4 base columns). */
int nx_bit_reverse_secure(nx_ctx* ctx, uint32_t* const* d_col4, uint32_t log_size);
/* FriOps::decompose(eval) -> (g, lambda): lambda = (sum of the first half - sum of the second half) / 2^log_size of the
 * bit-reversed circle evaluation, g = eval -/+ lambda on the first / second half [upstream-recollection; not called by
 * FriProver::commit at the pinned revision as far as the reference's use of it shows — exported for trait completeness]. */
int nx_fri_decompose(nx_ctx* ctx, const uint32_t* const* d_src4, uint32_t log_size, uint32_t* const* d_g4, uint32_t lambda[4]);

/* --------------------------------------------------------------- K10: GrindOps ------------- */
/* GrindOps<Blake2sChannel>::grind: smallest nonce with >= pow_bits trailing zero bits of
 * Blake2s(digest ‖ nonce_le64) read as a little-endian u128. */
int nx_grind(nx_ctx* ctx, const uint8_t digest[32], uint32_t pow_bits, uint64_t* nonce);

/* ------------------------------------------- synthetic machine (BASELINE configs #2-#4) ---- */
/* The reference's AIR closures (MachineEval::evaluate, prover/src/components/mod.rs:48-56) are
 * generic Rust and cannot cross a C ABI (last row).  For measurement the library
 * carries the synthetic wide-Fibonacci-style machine of SURVEY.md §8(d): */
typedef struct nx_component_spec {
    uint32_t log_size, n_pre, n_main, n_inter;
    /* The component's log constraint-degree bound: its constraints are evaluated on CanonicCoset(log_size + bound) and the composition
     * polynomial has log size max over components of (log_size + bound) — per component as in the reference (v1: the main component
     * +2, prover/src/components/mod.rs:12,44-45, every extension +1, prover/src/extensions/multiplicity.rs:108-110; prover2: 1, the
     * shifts 2, prover2/machine/src/framework/traits/builtin.rs:23, composition size = the maximum, prove.rs:44-48).
     * 0 = nx_pcs_config.log_constraint_degree; otherwise 1 <= bound <= that field (the twiddle tree is sized by the config's). */
    uint32_t log_constraint_degree_bound;
    /* nx_prove_machine only — how the component's F logup fractions fill its L = n_inter / 4 logup columns (NX_LOGUP_* below). */
    uint32_t logup_mode;
} nx_component_spec;
/* nx_component_spec.logup_mode, a bit set:
 *   0                 one fraction per column, F = L: EvalAtRow::finalize_logup as the v1 main component and the multiplicity tables
 *                     use it (reference prover/src/components/mod.rs:53, prover/src/extensions/multiplicity.rs:123) — degree-2 constraints;
 *   NX_LOGUP_PAIRS    two fractions per column, F = 2 L: finalize_logup_in_pairs (reference prover/src/extensions/keccak/round/
 *                     constraints.rs:116, keccak/memory_check/constraints.rs:125, extensions/final_reg.rs:112, every prover2 component:
 *                     prover2/machine/src/components/.../mod.rs) over columns built pairwise, (a d + b c) / (b d), by
 *                     LogupTraceBuilder::add_to_relation_with (prover2/machine/src/lookups/logup_trace_builder.rs:86-101) — the
 *                     constraint (S_j - S_{j-1}) d0 d1 - (n0 d1 + n1 d0) has degree 3 under the bound +1;
 *   NX_LOGUP_ODD      with PAIRS: F = 2 L - 1, the last column holds the one left-over fraction (LogupTraceBuilder::finalize,
 *                     logup_trace_builder.rs:110-117; finalize_logup_in_pairs with an odd number of add_to_relation calls);
 *   NX_LOGUP_TABLE    the table side of a lookup: every tuple reads PREPROCESSED columns (get_preprocessed_column) and every numerator is
 *                     a negated multiplicity column of the main trace (reference prover/src/extensions/multiplicity.rs:111-124,
 *                     extensions/keccak/bitwise_table/mod.rs). */
enum { NX_LOGUP_PAIRS = 1, NX_LOGUP_ODD = 2, NX_LOGUP_TABLE = 4 };
/* Bits 4..7 of logup_mode: the TUPLE SCHEDULE — how wide the relation tuples of the component's fractions are and what their entries are.
 *   0                 one- / two-column tuples (rounds 2-5);
 *   NX_TUPLES_V1      v1's chips: widths cycling 1, 1, 4, 1, 9, 1, 3, 1 — range checks (reference prover/src/chips/range_check/range256.rs:37),
 *                     [op_type CONSTANT, b, c, a] with a flag COLUMN as numerator (chips/instructions/i/bit_op.rs:31,341-365), register memory
 *                     (memory_check/register_mem_check.rs:34), and a 3-wide entry list holding a constant and a SUM of two columns;
 *   NX_TUPLES_KECCAK  3- / 4-wide bitwise lookups (chips/custom.rs:33-37) and, as the component's last two fractions, the 200-wide state
 *                     lookups with the numerators (is_padding - 1) and (1 - is_padding) (chips/custom.rs:45-46,
 *                     extensions/keccak/round/constraints.rs:101-110): expression numerators — their interaction trace comes from the
 *                     recorded relation entries (nx_logup_program);
 *   NX_TUPLES_V2      prover2's relations: 9, 21, 14, 10, 4, 12, 8 wide (prover2/machine/src/lookups/relations.rs:33-90).
 * The context option "machine.logup_program" = 1 sends every wide-tuple component through nx_logup_program (same bytes). */
#define NX_LOGUP_TUPLES(k) ((uint32_t)(k) << 4)
enum { NX_TUPLES_V1 = 1, NX_TUPLES_KECCAK = 2, NX_TUPLES_V2 = 3 };

typedef struct nx_pcs_config {
    uint32_t pow_bits, log_blowup, n_queries, log_last_layer_degree_bound; /* PcsConfig / FriConfig */
    uint32_t hash_mode, fri_alpha_mode;                                    /* switchable rules       */
    uint32_t log_constraint_degree;   /* 1 or 2 (components/mod.rs:12): the default AND the largest per-component bound (twiddle sizing, machine.rs:184-194) */
} nx_pcs_config;

typedef struct nx_prove_stats { /* milliseconds, device-synchronised stage boundaries */
    double trace_gen, commit, composition, oods, quotients, fri, pow, decommit, total;
    double lde_kernel_ms;        /* sum of Circle-FFT kernel time inside commits (HIP events)   */
    uint64_t lde_algorithmic_bytes;
    double merkle_kernel_ms;
    uint64_t merkle_algorithmic_bytes;
    double interaction;          /* logup interaction-trace generation (nx_prove_machine)         */
    double comm_ms;              /* wall time inside nx_comm callbacks (one proof on several GPUs) */
    uint64_t comm_bytes;         /* bytes this GPU sent to its peers                               */
    /* collectives this GPU entered during the prove, by kind — each is also a host synchronisation of the stream (one proof on several
     * GPUs): all-to-all of column shards into row blocks, all-gathers of device buffers, all-gathers of host words (roots, sampled
     * values, votes).  DESIGN.md section 7 bounds them by the statement's shape; tests/test_gpu_machine.py holds the bound. */
    uint32_t n_alltoallv, n_allgather_dev, n_allgather_host, n_comm_reserved;
} nx_prove_stats;

/* Fill tree `tree` (0 preprocessed / 1 main / 2 interaction) of the synthetic trace directly in
 * bit-reversed circle-domain order: d_cols holds, component after component, n_* columns. */
int nx_synth_fill_tree(nx_ctx* ctx, const nx_component_spec* comps, uint32_t n_comps, uint32_t tree,
                       uint64_t seed, uint64_t inter_seed, uint32_t* const* d_cols);

/* Full prove of the synthetic machine: the orchestration of reference prover/src/machine.rs:184-296
 * and stwo::prover::prove on device.  Returns a malloc'd proof in the flat "NXP1" u32 wire format
 * (free with nx_free_host). stats may be NULL. */
int nx_prove_synth(nx_ctx* ctx, const nx_component_spec* comps, uint32_t n_comps, const nx_pcs_config* cfg,
                   uint64_t seed, const uint8_t* ad, size_t ad_len, uint32_t** proof_words, size_t* n_words,
                   nx_prove_stats* stats);
void nx_free_host(void* p);

/* EXPERIMENTAL until a real postcard dump of the reference has been replayed (tools/dump_reference.rs -> tools/replay_reference_dump.py):
 * the field order below is a recollection of Stwo @ 0790eba's derive(Serialize) declarations; a wrong order yields unparseable bytes.
 * The reference's proof bytes: `nexus_vm_prover::machine::Proof { stark_proof, claimed_sum, log_size }` (reference
 * prover/src/machine.rs:93-98) in postcard, the serde format the SDK ships proofs in (reference sdk/Cargo.toml:22,
 * sdk/src/stwo/seq.rs:60-64), from an NXP1 word stream plus the per-component claimed sums (4 words each) and log sizes.  Field
 * order = Stwo's derive(Serialize) declarations [upstream-recollection — pinned only once tools/dump_reference.rs has run on a box
 * with cargo].  Host arithmetic only (no context).  *bytes: free with nx_free_host. */
int nx_proof_serialize_stwo(const uint32_t* proof_words, size_t n_words, const uint32_t* claimed_sums, const uint32_t* log_sizes,
                            uint32_t n_components, uint8_t** bytes, size_t* n_bytes);

/* The reference-shaped machine: as nx_prove_synth, but the interaction tree is a REAL logup trace — lookup elements (z, alpha) drawn
 * after the main commit (reference machine.rs:239-240), one fraction per logup column over main-trace columns
 * (LogupTraceGenerator, reference traits.rs:124-145, chips/range_check/range256.rs:271-288: nx_logup_col per column, then
 * nx_logup_finalize_last), claimed sums mixed before the commit (machine.rs:262) — and the AIR (transition, degree-2 and logup
 * constraints with the [-1, 0] mask of the last logup column) is a recorded program compiled by nx_air_compile: the route a Rust
 * shim takes for the reference's own AIR.  n_inter = 4 x (logup columns of the component).  comm: NULL = one GPU; otherwise ONE
 * proof on the GPUs of the communicator (nx_comm below). */
struct nx_comm;
/* The HIP source nx_air_compile generates for the recorded AIR of one such component (host only; free with nx_free_host). */
int nx_machine_air_source(const nx_component_spec* comp, char** h_source);
/* The recorded program itself (host only): what nx_prove_machine hands to nx_air_compile for this component under a configuration
 * whose log_constraint_degree is cfg_log_constraint_degree — so that a CPU checker can run the product's emission through its own
 * interpreter.  Secure constants: [z, alpha, claimed / N, 0].  *h_program: free with nx_free_host. */
struct nx_cinstr;
int nx_machine_air_program(const nx_component_spec* comp, uint32_t cfg_log_constraint_degree, struct nx_cinstr** h_program,
                           uint32_t* n_instr, uint32_t* n_regs, uint32_t* n_constraints);
int nx_prove_machine(nx_ctx* ctx, const nx_component_spec* comps, uint32_t n_comps, const nx_pcs_config* cfg, uint64_t seed,
                     const uint8_t* ad, size_t ad_len, const struct nx_comm* comm, uint32_t** proof_words, size_t* n_words,
                     nx_prove_stats* stats);
/* The same machine with its preprocessed and main traces handed over in HOST memory — the hand-over of the reference, whose trace builder
 * fills `Vec<Vec<M31>>` on the CPU (prover/src/trace/trace_builder.rs:19-32) — instead of generated on the device: h_pre_cols / h_main_cols
 * hold, component after component, the n_pre / n_main columns of 2^log_size words (natural coset order when coset_order != 0, else
 * bit-reversed circle-domain order).  The two commits upload the columns in chunks and transform each chunk as it arrives
 * (nx_prover_tree_commit_host describes the pipeline); the interaction trace is generated on the device as in nx_prove_machine.  One GPU.
 * The proof equals nx_prove_machine's for the same trace, byte for byte.  This is what bench.py's `host_trace` block times. */
int nx_prove_machine_host(nx_ctx* ctx, const nx_component_spec* comps, uint32_t n_comps, const nx_pcs_config* cfg,
                          const uint32_t* const* h_pre_cols, const uint32_t* const* h_main_cols, int coset_order, const uint8_t* ad,
                          size_t ad_len, uint32_t** proof_words, size_t* n_words, nx_prove_stats* stats);
/* `Proof.claimed_sum` of the last successful nx_prove_machine on this context (reference prover/src/machine.rs:93-98: the proof the
 * reference returns carries the per-component logup claimed sums next to the StarkProof; a verifier mixes them before the interaction
 * commitment, machine.rs:262 / :448-450, and nx_proof_serialize_stwo takes them): 4 words per component, in component order.
 * Writes min(cap_components, n) entries and the count n.  Host only (no device work). */
int nx_machine_claimed_sums(const nx_ctx* ctx, uint32_t* claimed_sums, uint32_t cap_components, uint32_t* n_components);

/* ------------------------------------------- ONE proof on the GPUs of a node (BASELINE configs #4 / #5) ---------------
 * One process (or thread) per GPU calls the prove entry with the same arguments and its own nx_comm; W must be a power of two.
 * The LDE is column-parallel: every GPU transforms a contiguous share of each tree's columns.  One all-to-all per tree then turns
 * column shards into ROW BLOCKS — GPU r holds rows [r M/W, (r+1) M/W) of every LDE column (a contiguous block of the bit-reversed
 * domain = a subtree of the Merkle tree), so leaf hashing, constraint evaluation, DEEP quotients and the first FRI folds are
 * local.  Besides the transposition only these cross the links: the W subtree roots of every tree (all-gather, 32 B each), the
 * sampled and queried values (KBs), the columns read at a non-zero mask offset, the composition accumulator (all-gather of 4
 * columns) and the small FRI tail.  Every GPU returns the same proof, bit-identical to the single-GPU proof.
 * The transport is the caller's: RCCL over xGMI (nexus-zkvm_amd/sharded.py wraps torch.distributed) or anything else.  Device
 * buffers handed to a callback are complete when it is called and must be complete when it returns.  Counts/offsets are in
 * 32-bit words. */
typedef struct nx_comm {
    int32_t rank, world;
    void* user;
    int (*send)(void* user, int32_t dst, const uint32_t* d_buf, size_t n_words);          /* device buffer (ring commit protocol only) */
    int (*recv)(void* user, int32_t src, uint32_t* d_buf, size_t n_words);                /* device buffer (ring commit protocol only) */
    int (*allreduce_m31)(void* user, uint32_t* d_buf, size_t n_words);                    /* unused by the row-sharded prove          */
    int (*allgather)(void* user, const void* h_send, size_t bytes, void* h_recv);         /* host: recv = world x bytes               */
    int (*broadcast)(void* user, void* h_buf, size_t bytes, int32_t root);                /* host (unused by the row-sharded prove)   */
    /* device all-to-all: words [send_off[r], send_off[r] + send_cnt[r]) of d_send go to rank r; what rank r sent lands at
     * [recv_off[r], recv_off[r] + recv_cnt[r]) of d_recv (arrays of `world` entries; the own share is copied too).  d_send is complete
     * on entry, d_recv must be complete on return; OTHER work of the library may still be queued on the context's stream (the next
     * column chunk's transforms), so run the collective on a stream of the transport's own and wait for that stream only */
    int (*alltoallv)(void* user, const uint32_t* d_send, const size_t* send_off, const size_t* send_cnt, uint32_t* d_recv,
                     const size_t* recv_off, const size_t* recv_cnt);
    /* device all-gather: d_recv = world x n_words, rank r's contribution at r * n_words */
    int (*allgather_dev)(void* user, const uint32_t* d_send, size_t n_words, uint32_t* d_recv);
    /* optional (may be NULL).  The library calls it on a rank whose prove FAILED after the exchanges of a proof may have started (a
     * HIP error, out of memory, a failed callback): its peers are, or will be, waiting for it in a collective it will never enter.
     * The transport must make the pending and later collectives of this communicator fail on the peers instead of waiting — tear the
     * group down (the thread-rank transport breaks its barrier; the native RCCL transport calls ncclCommAbort and, since RCCL does
     * not propagate that to live peers, additionally bounds every wait by the context option "comm.timeout_ms").  A communicator
     * is unusable after abort. */
    void (*abort)(void* user);
} nx_comm;
/* The native transport: RCCL over xGMI (csrc/comm_rccl.hip; librccl is opened at run time, NX_ERR_HIP when it is not there).  Rank 0
 * calls nx_rccl_unique_id and ships the 128 bytes to the other ranks through any side channel (environment, file, socket — the
 * bootstrap every NCCL program has); every rank then creates its communicator on its own context and hands `*out` to the prove
 * entries / nx_prover_set_comm.  Collectives run on a stream of the transport's own (see alltoallv above).  One process or thread
 * per GPU; RCCL itself refuses two ranks on one device. */
int nx_rccl_unique_id(uint8_t id[128]);
int nx_comm_rccl_create(nx_ctx* ctx, const uint8_t unique_id[128], int32_t rank, int32_t world, nx_comm** out);
void nx_comm_rccl_destroy(nx_comm* comm);
/* The in-process transport (csrc/comm_local.hip): ONE process drives the GPUs of a node with one thread per GPU — each thread its own
 * context and its own communicator of one shared group — and the collectives are rendezvous of those threads plus device-to-device copies
 * every rank pulls from its peers (peer to peer over xGMI between two devices; plain copies when several contexts share one GPU, which is
 * how the test-suite runs the 2 / 4 / 8-rank proofs on a one-GPU box).  No RCCL, no second process.  abort() breaks the group: every
 * waiting and later collective fails on every rank; waits are bounded by "comm.timeout_ms".  The group outlives its communicators. */
typedef struct nx_comm_group nx_comm_group;
int nx_comm_group_create(int32_t world, nx_comm_group** out);
void nx_comm_group_destroy(nx_comm_group* group);
/* A group that a failure broke (abort, a timeout) stays broken until it is re-armed: call this when EVERY rank's thread has returned
 * from its prove call (refused while a rank is still copying).  The prove entries abort only on failures that can leave peers waiting
 * — a HIP / transport / allocation error on one rank —; a refusal every rank reaches by itself (an argument error, the vote before the
 * first exchange, ProvingError::ConstraintsNotSatisfied from the all-gathered sampled values) leaves the group and an RCCL
 * communicator usable for the next proof. */
int nx_comm_group_reset(nx_comm_group* group);
int nx_comm_group_broken(const nx_comm_group* group);
/* 1: rank `from_rank` pulls from `to_rank`'s device peer to peer (xGMI) or they share a device; 0: the runtime stages those copies
 * (peer access could not be enabled); -1: one of the two communicators does not exist yet. */
int nx_comm_group_peer_access(nx_comm_group* group, int32_t from_rank, int32_t to_rank);
int nx_comm_local_create(nx_comm_group* group, nx_ctx* ctx, int32_t rank, nx_comm** out);
void nx_comm_local_destroy(nx_comm* comm);
/* Which columns of a tree a GPU transforms: groups = (column count, log size) per component in commit order; consecutive groups
 * of one size form a run whose columns are cut into `world` contiguous balanced ranges; lo/hi[i] = this rank's [lo, hi) of group i.
 * Host arithmetic only (no context, no GPU). */
int nx_plan_local_columns(const uint32_t* group_n_cols, const uint32_t* group_log_sizes, uint32_t n_groups, int32_t rank,
                          int32_t world, uint32_t* lo, uint32_t* hi);
/* nx_prove_synth on the GPUs of `comm`. */
int nx_prove_synth_sharded(nx_ctx* ctx, const nx_component_spec* comps, uint32_t n_comps, const nx_pcs_config* cfg,
                           uint64_t seed, const uint8_t* ad, size_t ad_len, const nx_comm* comm, uint32_t** proof_words,
                           size_t* n_words, nx_prove_stats* stats);
/* Modular add / widening helpers for transports that sum M31 buffers (RCCL has no modular reduction). */
int nx_m31_add_into(nx_ctx* ctx, uint32_t* d_dst, const uint32_t* d_src, size_t n_words);
int nx_m31_widen(nx_ctx* ctx, uint64_t* d_dst, const uint32_t* d_src, size_t n_words);
int nx_m31_narrow(nx_ctx* ctx, uint32_t* d_dst, const uint64_t* d_src, size_t n_words);

/* ------------------------------------------- "next" row R9: recorded AIR constraints evaluated on device ---------------
 * The reference's AIR closures (MachineEval::evaluate, prover/src/components/mod.rs:39-57; BuiltInComponentEval::evaluate,
 * prover2/machine/src/framework/eval.rs:19-33) cannot cross a C ABI, but a recording EvalAtRow — the trick Stwo's
 * InfoEvaluator already plays to discover masks (prover/src/components/mod.rs:59-67) — turns them into a straight-line
 * program over the field tower, and that can.  Registers are indices of a per-row register file (a secure-field value takes
 * 4 consecutive registers); the host allocates them.  For every row of the evaluation domain (bit-reversed circle-domain
 * order, log size log_eval) the program runs once and  acc[row] += (sum_j alpha_powers[j] * C_j(row)) * denom_inv[row >> log_size]
 * (FrameworkComponent::evaluate_constraint_quotients_on_domain). */
enum {
    NX_C_LOAD = 0,          /* B[dst] = cols[a][row + (int32)b trace steps]                          */
    NX_C_CONST = 1,         /* B[dst] = a (canonical M31 immediate)                                  */
    NX_C_ADD = 2, NX_C_SUB = 3, NX_C_MUL = 4, /* B[dst] = B[a] op B[b]                               */
    NX_C_NEG = 5,           /* B[dst] = -B[a]                                                        */
    NX_C_CONSTE = 6,        /* E[dst] = econsts[a]   (QM31: lookup elements, ...)                    */
    NX_C_ADDE = 7, NX_C_SUBE = 8, NX_C_MULE = 9, /* E[dst] = E[a] op E[b]                            */
    NX_C_MULEB = 10,        /* E[dst] = E[a] * B[b]                                                  */
    NX_C_ADDEB = 11,        /* E[dst] = E[a] + B[b]                                                  */
    NX_C_LOADE = 12,        /* E[dst] = (cols[a], cols[a+1], cols[a+2], cols[a+3])[row + (int32)b steps]  (a secure column) */
    NX_C_CONSTRAINT_B = 13, /* add_constraint(B[a])                                                  */
    NX_C_CONSTRAINT_E = 14, /* add_constraint(E[a])                                                  */
    /* logup FRACTION programs only (nx_logup_program): a relation entry of the AIR, add_to_relation(RelationEntry { relation,
     * multiplicity, values }) — numerator = the multiplicity expression, denominator = relation.combine(values) */
    NX_C_FRAC = 15,         /* fraction E[a] / E[b] of logup column (batch) dst                       */
    NX_C_FRACB = 16         /* fraction B[a] / E[b] of logup column (batch) dst (a base-field numerator) */
};
typedef struct nx_cinstr { uint32_t op, dst, a, b; } nx_cinstr;
int nx_eval_constraint_program(nx_ctx* ctx, const nx_cinstr* program, uint32_t n_instr, uint32_t n_regs,
                               const uint32_t* const* d_cols, uint32_t n_cols, const uint32_t* econsts /* 4 words each */,
                               uint32_t n_econsts, const uint32_t* alpha_powers /* 4 words per constraint */,
                               uint32_t n_constraints, const uint32_t* denom_inv /* 2^(log_eval-log_size) words */,
                               uint32_t log_size, uint32_t log_eval, uint32_t* const* d_acc4);

/* The same program compiled into a gfx950 kernel at run time (hiprtc): straight-line code over VGPRs instead of an LDS
 * register file, ~3x faster than the interpreter; lookup elements and alpha powers stay run-time arguments, so one compilation
 * serves every proof of an AIR.  h_source_out (optional, may be the only output: then ctx may be NULL and no GPU is needed)
 * receives the generated HIP source (free with nx_free_host).  nx_air_eval is stream-ordered like the other kernels: it returns
 * once the launch is enqueued (nx_sync or any later entry that reads the accumulator orders after it). */
typedef struct nx_air_kernel nx_air_kernel;
int nx_air_compile(nx_ctx* ctx, const nx_cinstr* program, uint32_t n_instr, uint32_t n_regs, uint32_t n_cols,
                   uint32_t n_econsts, uint32_t n_constraints, nx_air_kernel** out, char** h_source_out);
int nx_air_eval(nx_ctx* ctx, const nx_air_kernel* kernel, const uint32_t* const* d_cols, const uint32_t* econsts,
                const uint32_t* alpha_powers, const uint32_t* denom_inv, uint32_t log_size, uint32_t log_eval,
                uint32_t* const* d_acc4);
void nx_air_kernel_destroy(nx_air_kernel* kernel);
/* Ahead-of-time kernels: the hiprtc compilation of a recorded AIR costs from 0.7 s (the bench's AIR) to seconds (a keccak-shaped one) and
 * would otherwise be paid by the first proof of every process.
 * nx_air_kernel_save: the kernel as a self-describing blob (header + gfx950 code object; free with nx_free_host) — build it once, in a
 * build step or on another box, ship it; nx_air_kernel_load: a kernel from such a blob, in milliseconds (checksummed; a blob of another
 * library version is refused).  Pass the loaded kernel in nx_air_component.kernel.
 * nx_air_cache_dir: a directory (created if its parent exists; NULL or "" = off; initial value: the environment variable
 * NX_AIR_CACHE_DIR) in which the library keeps those blobs by itself, keyed by a hash of the generated source, the target and the hiprtc
 * version: nx_air_compile / nx_air_compile_subset — and therefore the provers, which compile their components' kernels through them —
 * load from it when they can and store what they compile (write-then-rename, one temporary per writer: processes and threads may
 * share a directory).  Process-wide.  The directory is TRUSTED INPUT — its files are GPU code objects, accepted on a 64-bit checksum, not a
 * signature: it is created with mode 0700 and ignored (with one line on stderr) unless it is a directory owned by the calling user that
 * neither group nor others may write to.
 * nx_air_cache_stats: kernels compiled by hiprtc / loaded from the directory / stored into it by this process (any pointer may be NULL).
 * COMPILATION IN PROCESSES: a generated source of >= 4 kernels is cut into one part per kernel and the parts are compiled side by side by
 * helper processes — the executable nx_air_cc next to this library (environment NX_AIR_CC: another path), at most NX_AIR_COMPILE_PROCS at a
 * time (default: the host's hardware threads, at most 32; 1 = everything in this process).  hiprtc serialises the threads of a process, not
 * processes: the first proof of the keccak-shaped statement waits 2.1 s instead of 14 s (profiles/r06_compile_procs.jsonl).  The blob is
 * the same bytes however it was compiled; without the helper the parts go through hiprtc here. */
int nx_air_kernel_save(const nx_air_kernel* kernel, uint8_t** blob, size_t* n_bytes);
int nx_air_kernel_load(nx_ctx* ctx, const uint8_t* blob, size_t n_bytes, nx_air_kernel** out);
int nx_air_cache_dir(const char* dir);
int nx_air_cache_stats(uint64_t* n_compiled, uint64_t* n_disk_hits, uint64_t* n_stored);
/* Degree-aware evaluation.  FrameworkEval::max_constraint_log_degree_bound (reference prover/src/components/mod.rs:44-45: +2 for v1's
 * main component) is the bound of the component's HIGHEST-degree constraint: Stwo evaluates every constraint, and therefore
 * re-extends every column, on the domain of log_size + bound.  A constraint of degree d over columns of 2^n rows has its quotient in
 * the FFT space of 2^(n+e) points iff d <= 2^e + 1, so the constraints of degree <= 3 (v1: all the logup constraints of
 * finalize_logup, components/mod.rs:53, i.e. all 1000 interaction columns; most chip constraints) can be evaluated on the
 * 2^(n+1)-point domain — with blowup 2 the committed evaluations, no re-extension — and lifted like a smaller component
 * (DomainEvaluationAccumulator::finalize); the composition polynomial, and the proof, are the same bits.
 * nx_air_constraint_degrees: an upper bound of every constraint's degree (host only; ctx may be NULL).
 * nx_air_compile_subset: nx_air_compile for the constraints with select[j] != 0 (NULL = all) — same alpha-power and column
 * indices as the whole program, so kernels of disjoint subsets add up to the whole; columns no selected constraint loads may be
 * NULL in nx_air_eval's d_cols.  The prover does this by itself (context option "air.degree_split", default 1). */
int nx_air_constraint_degrees(nx_ctx* ctx, const nx_cinstr* program, uint32_t n_instr, uint32_t n_regs, uint32_t n_cols,
                              uint32_t n_econsts, uint32_t n_constraints, uint32_t* degrees /* n_constraints */);
int nx_air_compile_subset(nx_ctx* ctx, const nx_cinstr* program, uint32_t n_instr, uint32_t n_regs, uint32_t n_cols,
                          uint32_t n_econsts, uint32_t n_constraints, const uint8_t* select /* n_constraints, or NULL */,
                          nx_air_kernel** out, char** h_source_out);

/* ------------------------------------------- the prover session: stwo::prover::prove over recorded AIRs
 * What `nexus_vm_prover::prove` does around Stwo (reference prover/src/machine.rs:184-296; prover2/machine/src/prove.rs) with
 * the AIR supplied as recorded constraint programs instead of Rust closures.  The caller replays the reference's transcript
 * prefix through the session — mix the program description (machine.rs:198-206), commit the preprocessed tree (:208-228) and the
 * main tree (:230-237), draw lookup elements (:239-240), build the interaction trace (nx_logup_*), mix the claimed sums (:262),
 * commit the interaction tree (:263) — and nx_prover_prove runs CommitmentSchemeProver / stwo::prover::prove (:286-290) on the
 * device: composition polynomial (nx_air_eval), OODS sampling, DEEP quotients, FRI, proof of work, decommitment.  The proof is
 * in the NXP1 word format of nx_prove_synth.  The session proves the trace trees it was given — the reference commits three
 * (preprocessed, main, interaction), Stwo's prove takes any TreeVec; a component column names its tree by index —; the composition
 * polynomial's tree follows them. */
typedef struct nx_prover nx_prover;
int nx_prover_create(nx_ctx* ctx, const nx_pcs_config* cfg, uint32_t max_log_size, nx_prover** out);
void nx_prover_destroy(nx_prover* prover);
/* One proof on several GPUs: every GPU runs the same session calls with its own communicator (see nx_comm); nx_prover_tree_begin
 * then hands each GPU only ITS columns (NULL for the others).  Call right after nx_prover_create. */
int nx_prover_set_comm(nx_prover* prover, const nx_comm* comm);
/* Blake2sChannel of the session */
int nx_prover_mix_u64(nx_prover* prover, uint64_t v);
int nx_prover_mix_felts(nx_prover* prover, const uint32_t* felts, uint32_t n_felts); /* 4 words per QM31 */
int nx_prover_draw_felt(nx_prover* prover, uint32_t out[4]);
/* Channel::draw_felts(n): n secure felts from the base-felt stream (two per Blake2s draw) — LookupElements::draw's (z, alpha) */
int nx_prover_draw_felts(nx_prover* prover, uint32_t n_felts, uint32_t* out /* 4 words each */);
int nx_prover_channel_digest(const nx_prover* prover, uint8_t digest[32]);
/* TreeBuilder::extend_evals + commit (machine.rs:208-263).  tree_begin allocates the tree's columns in the session (one slab per
 * run of equal log sizes) and returns their device addresses; the caller fills them with bit-reversed circle-domain evaluations
 * (trace generation on the device, nx_upload_columns, nx_copy ...); tree_commit interpolates, extends, commits and mixes the root.
 * The columns then belong to the session (they hold the coefficients afterwards). */
int nx_prover_tree_begin(nx_prover* prover, const uint32_t* log_sizes, uint32_t n_cols, uint32_t** d_cols_out);
int nx_prover_tree_commit(nx_prover* prover, uint8_t root[32]);
/* The same for a tree whose columns are in HOST memory — what the reference's trace builder hands over (`Vec<Vec<M31>>`,
 * prover/src/trace/trace_builder.rs:19-32): h_cols[i] is the host source of column i of the tree begun with nx_prover_tree_begin
 * (2^log_sizes[i] words; natural coset order when coset_order != 0 — R3's permutation then runs on the device — else bit-reversed
 * circle-domain order).  The commit uploads the columns in chunks on a copy stream and transforms each chunk as soon as it has arrived:
 * PCIe transfer, iFFT + LDE and leaf hashing overlap.  keep_idx / d_keep (n_keep entries, may be 0):
 * columns whose EVALUATIONS are needed after the commit (the logup fractions read main-trace columns; the commit turns columns into
 * coefficients in place) are cloned into d_keep[k] (2^log words, caller-allocated) on arrival — the reference's whole-trace clone
 * (machine.rs:232), for the columns that need it.  Blocking like nx_prover_tree_commit; the host columns are free on return.
 * A row-sharded session uploads its own columns first (no overlap) and commits as usual. */
int nx_prover_tree_commit_host(nx_prover* prover, const uint32_t* const* h_cols, int coset_order, const uint32_t* keep_idx,
                               uint32_t n_keep, uint32_t* const* d_keep, uint8_t root[32]);
/* A committed tree that is proved over and over — the preprocessed tree: the reference commits the same preprocessed + program columns
 * in every proof of a program (prover/src/machine.rs:208-228) and once more in every verification (machine.rs:363-417, verify.rs:103-143,
 * which needs only the root) — need not be transformed and hashed again.  nx_prover_tree_share turns committed tree `tree_index` of a
 * session into a reference-counted handle (the session keeps proving with it); nx_prover_tree_adopt makes it the NEXT tree of another
 * session of the SAME context and mixes its root, exactly as nx_prover_tree_commit would after a fresh commit of the same columns: the
 * proof bytes are those of the fresh commit.  After a commit a tree's buffers are read-only, so any number of sessions may hold one.
 * One GPU; same blowup factor and node hash.  nx_committed_tree_release drops the handle (the buffers go when the last session that
 * adopted it is destroyed); nx_committed_tree_root: the root and the column count (a verifier's preprocessed-root check).
 * LIFETIME: a handle and every session that adopted it belong to the context that committed the tree — release / destroy them BEFORE
 * nx_ctx_destroy (their buffers return to that context's allocator; afterwards the handle dangles). */
typedef struct nx_committed_tree nx_committed_tree;
int nx_prover_tree_share(nx_prover* prover, uint32_t tree_index, nx_committed_tree** out);
int nx_prover_tree_adopt(nx_prover* prover, const nx_committed_tree* tree, uint8_t root[32]);
int nx_committed_tree_root(const nx_committed_tree* tree, uint8_t root[32], uint32_t* n_cols);
void nx_committed_tree_release(nx_committed_tree* tree);
/* A component: what FrameworkComponent<E> is to Stwo (reference prover/src/components/mod.rs:15-57).  Column k of the program is
 * column col_index[k] of tree col_tree[k] (TraceLocationAllocator); it is sampled at the mask_count[k] row offsets listed next in
 * mask_offsets (InfoEvaluator, components/mod.rs:59-67) — every LOAD offset must be listed, every committed column must be
 * claimed by some component.  kernel: the program compiled by nx_air_compile, or NULL to let nx_prover_prove compile it. */
typedef struct {
    uint32_t log_size;
    const nx_cinstr* program; uint32_t n_instr, n_regs;
    const uint32_t* econsts; uint32_t n_econsts;
    uint32_t n_constraints;
    const uint32_t* col_tree; const uint32_t* col_index; uint32_t n_cols;
    const uint32_t* mask_count; const int32_t* mask_offsets;
    const nx_air_kernel* kernel;
    uint32_t log_constraint_degree_bound;   /* this component's bound (see nx_component_spec): 0 = the session config's log_constraint_degree */
} nx_air_component;
/* stwo::prover::prove.  NX_ERR_PROTOCOL = ProvingError::ConstraintsNotSatisfied.  *proof_words: free with nx_free_host.  After a proof
 * nx_prover_channel_digest is the transcript's final state; another nx_prover_prove starts again from the state the first one found
 * (and a failed call restores it at once), so proving the committed statement again — or retrying — gives the same bytes. */
int nx_prover_prove(nx_prover* prover, const nx_air_component* components, uint32_t n_components, uint32_t** proof_words,
                    size_t* n_words, nx_prove_stats* stats);

/* ------------------------------------------- "next" row R8: logup interaction trace on device ------------------------
 * The reference fills the interaction trace on the CPU (prover/src/traits.rs:124-145 generate_interaction_trace -> per chip,
 * e.g. prover/src/chips/range_check/range256.rs:271-288; prover2/machine/src/lookups/logup_trace_builder.rs:22-121) through
 * stwo-constraint-framework's LogupTraceGenerator.  For main-trace columns that already live in HBM: */
/* Relation::combine: denom(row) = sum_i alpha_powers[i] * tuple_i(row) - z   (alpha_powers: 4 words per tuple column) */
int nx_logup_combine(nx_ctx* ctx, const uint32_t* const* d_tuple_cols, uint32_t n_cols, const uint32_t* alpha_powers,
                     const uint32_t z[4], uint32_t log_size, uint32_t* const* d_out4);
/* LogupColGenerator::{write_frac, finalize_col}: out(row) = num/den + prev(row).  A numerator is scale * mult(row)
 * (d_mult NULL: the constant `scale`, e.g. 1 or -1).  With a second fraction (d_den_b4 != NULL) the two are merged first,
 * (a d + b c) / (b d), as prover2's LogupTraceBuilder::add_to_relation_with does (logup_trace_builder.rs:93-97).
 * d_prev4 NULL for the first column.  out may alias prev. */
int nx_logup_finalize_col(nx_ctx* ctx, uint32_t log_size, const uint32_t* d_mult_a, const uint32_t scale_a[4],
                          const uint32_t* const* d_den_a4, const uint32_t* d_mult_b, const uint32_t scale_b[4],
                          const uint32_t* const* d_den_b4, const uint32_t* const* d_prev4, uint32_t* const* d_out4);
/* Fused form of the two calls above (the denominators never touch memory): one interaction column from one or two
 * fractions given by their tuple columns.  frac_b NULL = a single fraction; d_prev4 NULL = the first column. */
typedef struct nx_logup_frac {
    const uint32_t* const* d_tuple_cols; uint32_t n_tuple_cols;   /* relation tuple (main-trace columns)                */
    const uint32_t* alpha_powers;                                 /* 4 words per tuple column                           */
    const uint32_t* z;                                            /* 4 words                                            */
    const uint32_t* d_mult;                                       /* multiplicity column or NULL                        */
    const uint32_t* scale;                                        /* 4 words: numerator = scale * mult (or scale)       */
} nx_logup_frac;
int nx_logup_col(nx_ctx* ctx, uint32_t log_size, const nx_logup_frac* frac_a, const nx_logup_frac* frac_b,
                 const uint32_t* const* d_prev4, uint32_t* const* d_out4);
/* The n_cols logup columns of a component in ONE launch: column j = sum over i <= j of fraction i(row), i.e. what n_cols calls of
 * nx_logup_col with d_prev4 = the previous column produce, reading every tuple column once.  d_out: 4 n_cols coordinate columns. */
int nx_logup_cols(nx_ctx* ctx, uint32_t log_size, const nx_logup_frac* fracs, uint32_t n_cols, uint32_t* const* d_out);
/* The general form, stwo-constraint-framework's finalize_logup_batched on the trace side: fraction i belongs to batch batching[i]
 * (any order; every batch 0 .. n_cols - 1 must hold at least one fraction), column j = sum of the fractions of batches <= j — the
 * merged fraction (a d + b c) / (b d) of a pair (LogupTraceBuilder::add_to_relation_with, reference prover2/machine/src/lookups/
 * logup_trace_builder.rs:86-101) is the sum of its two fractions.  batching NULL = in pairs (i / 2: finalize_logup_in_pairs; an odd
 * n_fracs leaves the last column one fraction).  One launch; every tuple column is read once.  d_out: 4 n_cols coordinate columns. */
int nx_logup_cols_batched(nx_ctx* ctx, uint32_t log_size, const nx_logup_frac* fracs, uint32_t n_fracs, const uint32_t* batching,
                          uint32_t n_cols, uint32_t* const* d_out);
/* LogupTraceGenerator::finalize_last on the last column (in place): claimed_sum = sum over all rows; the column becomes
 * the inclusive prefix sum, in natural coset order, of (value - claimed_sum / 2^log_size). */
int nx_logup_finalize_last(nx_ctx* ctx, uint32_t log_size, uint32_t* const* d_col4, uint32_t claimed_sum[4]);
/* The same for n_cols secure columns of one size (d_cols4: n_cols x 4 coordinate pointers; claimed_sums: n_cols x 4 words) in three
 * launches and one device-to-host copy: the form for AIRs with many components / logup columns (BASELINE config #5). */
int nx_logup_finalize_last_batch(nx_ctx* ctx, uint32_t log_size, uint32_t* const* d_cols4, uint32_t n_cols,
                                 uint32_t* claimed_sums);

/* The interaction trace of a component FROM ITS RECORDED AIR.  A recording EvalAtRow sees every relation entry the AIR declares —
 * eval.add_to_relation(RelationEntry::new(relation, multiplicity, &values)), e.g. reference prover/src/components/mod.rs:48-56,
 * prover/src/extensions/keccak/round/constraints.rs:95-116, prover2's components — as two expressions over the trace columns: the
 * multiplicity and relation.combine(values) = sum_i alpha^i values_i - z.  Those expressions, lowered like constraints (same opcodes,
 * loads of the component's preprocessed / main columns, any row offset) with one NX_C_FRAC / NX_C_FRACB per entry, ARE the
 * interaction trace: logup column j (batch j of finalize_logup / finalize_logup_in_pairs / finalize_logup_batched) holds the sum over
 * the entries of batches <= j of multiplicity / denominator — what the reference's hand-written generators (prover/src/traits.rs:
 * 124-145 -> every chip's fill_interaction_trace; prover2/machine/src/lookups/logup_trace_builder.rs:22-121) must compute too, or
 * their own constraints fail.  So the generator of EVERY chip and component moves to the device without touching one of them.
 * program: the fractions in non-decreasing batch order, every batch 0 .. n_logup_cols - 1 present; d_cols: the component's columns
 * as EVALUATIONS on the trace domain (bit-reversed circle-domain order, 2^log_size words; columns the program does not load may be
 * NULL — the interaction columns themselves always are); d_out: 4 n_logup_cols coordinate columns.  Follow with
 * nx_logup_finalize_last on the last column.  Compiled by hiprtc (cached per context; nx_air_cache_dir applies).
 * h_source_out (optional; then ctx / d_cols / d_out may be NULL and no GPU is needed): the generated HIP source. */
int nx_logup_program(nx_ctx* ctx, const nx_cinstr* program, uint32_t n_instr, uint32_t n_regs, const uint32_t* const* d_cols,
                     uint32_t n_cols, const uint32_t* econsts, uint32_t n_econsts, uint32_t log_size, uint32_t n_logup_cols,
                     uint32_t* const* d_out, char** h_source_out);

/* Config #2: LDE + Blake2s commit of n_cols random columns of 2^log_size rows (already resident,
 * bit-reversed evaluations, overwritten by their coefficients); d_lde receives the LDE columns. */
int nx_lde_commit(nx_ctx* ctx, const nx_twiddles* tw, uint32_t* const* d_cols, uint32_t n_cols, uint32_t log_size,
                  uint32_t log_blowup, uint32_t* const* d_lde, uint8_t root[32]);

#ifdef __cplusplus
}
#endif
#endif /* NEXUS_HIP_H */
