/* The drop-in boundary used from plain C (C99): prove a small recorded AIR through the prover session of include/nexus_hip.h.
 *
 *   gcc -std=c99 -Iinclude examples/session_prove.c -Lnexus-zkvm_amd -lnexus_hip -Wl,-rpath,$PWD/nexus-zkvm_amd -o session_prove
 *
 * One component of 2^6 rows: main columns a, b, c with the constraint c - a*b - 3 = 0, recorded as the straight-line program a
 * recording EvalAtRow would emit (NX_C_* opcodes); a one-column preprocessed tree and an empty interaction tree complete the
 * three trace trees the protocol expects (reference prover/src/machine.rs:208-263).  Prints "ok <proof words>"; with the argument
 * "bad" it corrupts one trace cell first and expects NX_ERR_PROTOCOL (ProvingError::ConstraintsNotSatisfied). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nexus_hip.h"

#define P 0x7fffffffu
#define CHECK(call) do { int rc_ = (call); if (rc_ != NX_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, nx_last_error(ctx)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int bad = argc > 1 && !strcmp(argv[1], "bad");
    const uint32_t log = 6, n = 1u << 6;
    nx_ctx* ctx = NULL;
    if (nx_ctx_create(0, &ctx) != NX_OK) { fprintf(stderr, "nx_ctx_create: %s\n", nx_last_error(NULL)); return 2; }
    nx_pcs_config cfg = {2, 1, 3, 0, 0, 0, 1};   /* pow_bits, log_blowup, n_queries, last-layer bound, hash rule, alpha rule, constraint degree */
    nx_prover* pr = NULL;
    CHECK(nx_prover_create(ctx, &cfg, log, &pr));
    CHECK(nx_prover_mix_u64(pr, log));                                /* the transcript prefix: whatever identifies the statement */

    /* tree 0: one preprocessed column (zeros) */
    uint32_t logs[3] = {log, log, log};
    uint32_t* d_pre[1]; uint32_t* d_main[3];
    uint8_t root[32];
    CHECK(nx_prover_tree_begin(pr, logs, 1, d_pre));
    CHECK(nx_memset_zero(ctx, d_pre[0], n));
    CHECK(nx_prover_tree_commit(pr, root));

    /* tree 1: a, b, c = a*b + 3 (the constraint is row-local, so any row order is a valid trace) */
    uint32_t* h = (uint32_t*)malloc(3 * n * sizeof(uint32_t));
    for (uint32_t i = 0; i < n; i++) {
        uint64_t a = i + 1, b = 2 * i + 5;
        h[i] = (uint32_t)a; h[n + i] = (uint32_t)b; h[2 * n + i] = (uint32_t)((a * b + 3) % P);
    }
    if (bad) h[2 * n + 7] = (h[2 * n + 7] + 1) % P;
    CHECK(nx_prover_tree_begin(pr, logs, 3, d_main));
    for (int k = 0; k < 3; k++) CHECK(nx_upload(ctx, d_main[k], h + k * n, n));
    CHECK(nx_prover_tree_commit(pr, root));
    free(h);

    /* tree 2: no interaction columns */
    CHECK(nx_prover_tree_begin(pr, NULL, 0, NULL));
    CHECK(nx_prover_tree_commit(pr, root));

    /* the recorded constraint: r3 = a*b; r4 = 3; r3 = r3 + r4; r2 = c - r3; add_constraint(r2) */
    const nx_cinstr prog[] = {
        {NX_C_LOAD, 0, 0, 0}, {NX_C_LOAD, 1, 1, 0}, {NX_C_LOAD, 2, 2, 0},
        {NX_C_MUL, 3, 0, 1}, {NX_C_CONST, 4, 3, 0}, {NX_C_ADD, 3, 3, 4}, {NX_C_SUB, 2, 2, 3}, {NX_C_CONSTRAINT_B, 0, 2, 0},
    };
    const uint32_t col_tree[4] = {1, 1, 1, 0}, col_index[4] = {0, 1, 2, 0}, mask_count[4] = {1, 1, 1, 1};
    const int32_t mask_offsets[4] = {0, 0, 0, 0};
    nx_air_component comp;
    memset(&comp, 0, sizeof comp);
    comp.log_size = log; comp.program = prog; comp.n_instr = sizeof prog / sizeof prog[0]; comp.n_regs = 5; comp.n_constraints = 1;
    comp.col_tree = col_tree; comp.col_index = col_index; comp.n_cols = 4; comp.mask_count = mask_count; comp.mask_offsets = mask_offsets;

    uint32_t* proof = NULL; size_t n_words = 0;
    int rc = nx_prover_prove(pr, &comp, 1, &proof, &n_words, NULL);
    if (bad) {
        if (rc == NX_ERR_PROTOCOL) printf("refused: %s\n", nx_last_error(ctx)); else printf("UNEXPECTED rc %d\n", rc);
        nx_prover_destroy(pr); nx_ctx_destroy(ctx);
        return rc == NX_ERR_PROTOCOL ? 0 : 1;
    }
    if (rc != NX_OK) { fprintf(stderr, "nx_prover_prove failed (%d): %s\n", rc, nx_last_error(ctx)); return 1; }
    printf("ok %zu\n", n_words);
    nx_free_host(proof);
    nx_prover_destroy(pr);
    nx_ctx_destroy(ctx);
    return 0;
}
