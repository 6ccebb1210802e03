// Synthetic machine: shared host/device declarations (see air.hip).
#pragma once
#include <vector>
#include "internal.h"

namespace nx {

constexpr uint32_t SYNTH_GROUP = 16;  // every 16 columns, two are free witness inputs

inline bool synth_col_is_free(uint32_t k) { return (k % SYNTH_GROUP) < 2; }
inline uint32_t synth_n_constraints(const nx_component_spec& c) {
    uint32_t n = 2;
    for (uint32_t k = 2; k < c.n_main; k++) if (!synth_col_is_free(k)) n++;
    for (uint32_t k = 0; k < c.n_inter; k++) if (!synth_col_is_free(k)) n++;
    return n;
}

// the columns one rank holds of a component: [main_begin, main_begin + n_main) of n_main_total main columns, likewise the
// interaction columns; has_pre1: the preprocessed column 1 (is_last) is local.  A single GPU holds everything.
struct SynthRange { u32 main_begin, n_main, n_main_total, inter_begin, n_inter; bool has_pre1; };
// rows [row_begin, row_begin + n_rows) of the evaluation domain; every column / accumulator pointer already biased (bias_rows)
int synth_constraints(nx_ctx* ctx, ColSet pre, ColSet mainc, ColSet inter, const SynthRange& rg, int log_size, int e, const u32* d_pw,
                      const u32* d_denom_inv, u32* const acc4[4], u32 row_begin, u32 n_rows);
// columns [col_begin, col_begin + n_cols) of one tree of component ci, positions [pos_begin, pos_begin + n_pos) (d_cols: n_pos words each)
int synth_fill_range(nx_ctx* ctx, const nx_component_spec& c, uint32_t ci, uint32_t tree, uint64_t seed, uint64_t inter_seed, uint32_t col_begin,
                     uint32_t n_cols, uint32_t* const* d_cols, uint32_t pos_begin, uint32_t n_pos);
int secure_accumulate(nx_ctx* ctx, u32* const dst4[4], const u32* const src4[4], u32 n);
int fft_lde(nx_ctx* ctx, const nx_twiddles* tw, ColSet cols, uint32_t n_cols, uint32_t log_size, uint32_t log_expand, ColSet out);
int merkle_layer(nx_ctx* ctx, ColSet cols, u32 n_cols, const u32* prev, u32* out, u32 log);
// recorded AIR programs (air_jit.hip): bounds of every register / column / secure-constant index; counts the constraints
int validate_air_program(nx_ctx* ctx, const nx_cinstr* program, uint32_t n_instr, uint32_t n_regs, uint32_t n_cols, uint32_t n_econsts, uint32_t* n_constraints_out);
int air_eval_rows(nx_ctx* ctx, const nx_air_kernel* k, const uint32_t* const* d_cols, const uint32_t* econsts, const uint32_t* alpha_powers, const uint32_t* denom_inv,
                  uint32_t log_size, uint32_t log_eval, uint32_t* const* d_acc4, uint32_t row_begin, uint32_t n_rows);
void air_kernel_shape(const nx_air_kernel* k, uint32_t* n_cols, uint32_t* n_econsts, uint32_t* n_constraints);
// degree-aware composition (air_jit.hip): an upper bound of every constraint's degree; the columns a subset of the constraints reads
void air_constraint_degrees(const nx_cinstr* prog, uint32_t n_instr, uint32_t n_regs, std::vector<uint32_t>* out);
void air_constraint_neighbours(const nx_cinstr* prog, uint32_t n_instr, uint32_t n_regs, std::vector<char>* out);   // per constraint: reads a column at a non-zero row offset
void air_subset_columns(const nx_cinstr* prog, uint32_t n_instr, uint32_t n_regs, uint32_t n_cols, const uint8_t* select, std::vector<char>* used);

}  // namespace nx
