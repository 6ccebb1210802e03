// Generic constraint-quotient evaluation on device: SURVEY.md §8(f) rank 1 ("next" row R9).
//
// The reference's AIR is a generic Rust closure (`MachineEval::evaluate`, prover/src/components/mod.rs:39-57;
// `BuiltInComponentEval::evaluate`, prover2/machine/src/framework/eval.rs:19-33) that Stwo calls per 16-row vector on the
// constraint domain.  A closure cannot cross a C ABI, but what it DOES can: Stwo already runs it once over a recording
// evaluator to discover the column masks (`InfoEvaluator`, prover/src/components/mod.rs:59-67).  The same trick with an
// expression-recording `EvalAtRow` yields a straight-line program over the field tower — loads of (column, row offset),
// constants, + - *, secure-field (QM31) arithmetic for the logup constraints, and `add_constraint` — which this kernel
// interprets for every row of the evaluation domain:  acc[row] += (sum_j alpha^j C_j(row)) * denom_inv[row >> log_size],
// exactly the accumulation of stwo-constraint-framework's evaluate_constraint_quotients_on_domain.
//
// Execution model: one lane per row, all lanes run the same instruction (instruction fetch is wave-uniform, scalar).  The
// virtual registers live in LDS ([register][lane], conflict-free), the host allocates them (nexus-zkvm_amd/air_program.py);
// runs of LOADs are issued 8 at a time so the column reads overlap.  The accumulator is 4 lazy 64-bit sums.
#include "internal.h"
#include "air.h"
#include <atomic>
#include <algorithm>

namespace nx {

constexpr int CP_THREADS = 256;

// +offset trace steps in a bit-reversed circle-domain evaluation of log size e over a trace of log size log_size
// (stwo-constraint-framework offset_bit_reversed_circle_domain_index)
__device__ __forceinline__ u32 row_offset(u32 r, int log_size, int e, int offset) {
    if (offset == 0) return r;
    u32 idx = bitrev(r, e);
    const u32 half = 1u << (e - 1);
    const u32 step = (u32)offset << (e - log_size - 1);   // two's complement: negative offsets wrap below
    if (idx < half) idx = (idx + step) & (half - 1);
    else idx = ((idx - half - step) & (half - 1)) + half;
    return bitrev(idx, e);
}

__global__ __launch_bounds__(CP_THREADS) void constraint_program_kernel(const nx_cinstr* __restrict__ prog, u32 n_instr, const u32* const* __restrict__ cols,
                                                                        const u32* __restrict__ econst, const u32* __restrict__ pw,
                                                                        const u32* __restrict__ denom_inv, int log_size, int e,
                                                                        u32* a0, u32* a1, u32* a2, u32* a3) {
    extern __shared__ u32 cp_regs[];   // [register][lane]
    const u32 r = blockIdx.x * CP_THREADS + threadIdx.x;
    const bool live = r < (1u << e);
    const u32 row = live ? r : 0;
    u32* R = cp_regs + threadIdx.x;
#define RG(i) R[(size_t)(i) * CP_THREADS]
    u64 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    u32 j = 0, pending = 0;
    auto fold = [&]() { s0 = acc_fold(s0); s1 = acc_fold(s1); s2 = acc_fold(s2); s3 = acc_fold(s3); pending = 0; };
    for (u32 pc = 0; pc < n_instr; pc++) {
        const nx_cinstr in = prog[pc];
        switch (in.op) {
        case NX_C_LOAD: {
            u32 cnt = 1;
            while (cnt < 8 && pc + cnt < n_instr && prog[pc + cnt].op == NX_C_LOAD) cnt++;
            u32 v[8];
#pragma unroll
            for (u32 u = 0; u < 8; u++) if (u < cnt) { const nx_cinstr li = prog[pc + u]; v[u] = gld(cols[li.a] + row_offset(row, log_size, e, (int)li.b)); }
#pragma unroll
            for (u32 u = 0; u < 8; u++) if (u < cnt) RG(prog[pc + u].dst) = v[u];
            pc += cnt - 1;
            break;
        }
        case NX_C_CONST: RG(in.dst) = in.a; break;
        case NX_C_ADD: RG(in.dst) = m_add(RG(in.a), RG(in.b)); break;
        case NX_C_SUB: RG(in.dst) = m_sub(RG(in.a), RG(in.b)); break;
        case NX_C_MUL: RG(in.dst) = m_mul(RG(in.a), RG(in.b)); break;
        case NX_C_NEG: RG(in.dst) = m_neg(RG(in.a)); break;
        case NX_C_CONSTE: { RG(in.dst) = econst[4 * in.a]; RG(in.dst + 1) = econst[4 * in.a + 1]; RG(in.dst + 2) = econst[4 * in.a + 2]; RG(in.dst + 3) = econst[4 * in.a + 3]; break; }
        case NX_C_ADDE: case NX_C_SUBE: case NX_C_MULE: {
            const QM31 x = qm(RG(in.a), RG(in.a + 1), RG(in.a + 2), RG(in.a + 3)), y = qm(RG(in.b), RG(in.b + 1), RG(in.b + 2), RG(in.b + 3));
            const QM31 z = in.op == NX_C_ADDE ? q_add(x, y) : in.op == NX_C_SUBE ? q_sub(x, y) : q_mul(x, y);
            RG(in.dst) = z.a.a; RG(in.dst + 1) = z.a.b; RG(in.dst + 2) = z.b.a; RG(in.dst + 3) = z.b.b;
            break;
        }
        case NX_C_MULEB: {
            const u32 sc = RG(in.b);
            const u32 x0 = RG(in.a), x1 = RG(in.a + 1), x2 = RG(in.a + 2), x3 = RG(in.a + 3);
            RG(in.dst) = m_mul(x0, sc); RG(in.dst + 1) = m_mul(x1, sc); RG(in.dst + 2) = m_mul(x2, sc); RG(in.dst + 3) = m_mul(x3, sc);
            break;
        }
        case NX_C_ADDEB: {
            const u32 x0 = RG(in.a), x1 = RG(in.a + 1), x2 = RG(in.a + 2), x3 = RG(in.a + 3), sc = RG(in.b);
            RG(in.dst) = m_add(x0, sc); RG(in.dst + 1) = x1; RG(in.dst + 2) = x2; RG(in.dst + 3) = x3;
            break;
        }
        case NX_C_LOADE: {   // a secure column = 4 consecutive coordinate columns (SecureColumnByCoords)
            const u32 rr = row_offset(row, log_size, e, (int)in.b);
            const u32 x0 = gld(cols[in.a] + rr), x1 = gld(cols[in.a + 1] + rr), x2 = gld(cols[in.a + 2] + rr), x3 = gld(cols[in.a + 3] + rr);
            RG(in.dst) = x0; RG(in.dst + 1) = x1; RG(in.dst + 2) = x2; RG(in.dst + 3) = x3;
            break;
        }
        case NX_C_CONSTRAINT_B: {
            const u32 v = RG(in.a);
            s0 = acc_mad(s0, pw[4 * j], v); s1 = acc_mad(s1, pw[4 * j + 1], v); s2 = acc_mad(s2, pw[4 * j + 2], v); s3 = acc_mad(s3, pw[4 * j + 3], v);
            j++;
            if (++pending == 4) fold();
            break;
        }
        case NX_C_CONSTRAINT_E: {
            const QM31 t = q_mul(qm(pw[4 * j], pw[4 * j + 1], pw[4 * j + 2], pw[4 * j + 3]), qm(RG(in.a), RG(in.a + 1), RG(in.a + 2), RG(in.a + 3)));
            s0 += t.a.a; s1 += t.a.b; s2 += t.b.a; s3 += t.b.b;     // canonical values; counted like a product for the fold period
            j++;
            if (++pending == 4) fold();
            break;
        }
        default: break;
        }
    }
#undef RG
    if (!live) return;
    const u32 di = denom_inv[r >> log_size];
    a0[r] = m_add(a0[r], m_mul(acc_final(s0), di)); a1[r] = m_add(a1[r], m_mul(acc_final(s1), di));
    a2[r] = m_add(a2[r], m_mul(acc_final(s2), di)); a3[r] = m_add(a3[r], m_mul(acc_final(s3), di));
}

}  // namespace nx

using namespace nx;

extern "C" int nx_eval_constraint_program(nx_ctx* ctx, const nx_cinstr* program, uint32_t n_instr, uint32_t n_regs, const uint32_t* const* d_cols,
                                          uint32_t n_cols, const uint32_t* econsts, uint32_t n_econsts, const uint32_t* alpha_powers,
                                          uint32_t n_constraints, const uint32_t* denom_inv, uint32_t log_size, uint32_t log_eval,
                                          uint32_t* const* d_acc4) {
    NX_GUARD(ctx);
    if (!ctx || !program || !d_acc4 || (n_cols && !d_cols)) return set_err(ctx, NX_ERR_ARG, "nx_eval_constraint_program: NULL argument");
    if (log_size < 1 || log_eval <= log_size || log_eval > 30) return set_err(ctx, NX_ERR_ARG, "nx_eval_constraint_program: need 1 <= log_size < log_eval <= 30");
    if (n_regs == 0 || (size_t)n_regs * CP_THREADS * 4 > 160 * 1024) return set_err(ctx, NX_ERR_ARG, "nx_eval_constraint_program: register file does not fit the 160 KB LDS (max 160 registers)");
    // validate the program once on the host: register / column / constant / constraint indices in range
    uint32_t n_c = 0;
    NX_TRY(validate_air_program(ctx, program, n_instr, n_regs, n_cols, n_econsts, &n_c));
    if (n_c != n_constraints) return set_err(ctx, NX_ERR_ARG, "nx_eval_constraint_program: the program adds a different number of constraints than alpha powers were given");
    const size_t b_prog = (size_t)n_instr * sizeof(nx_cinstr), b_cols = (size_t)n_cols * 8, b_ec = (size_t)n_econsts * 16, b_pw = (size_t)n_constraints * 16,
                 b_den = (size_t)4 << (log_eval - log_size);
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t o_cols = al(b_prog), o_ec = o_cols + al(b_cols), o_pw = o_ec + al(b_ec), o_den = o_pw + al(b_pw), total = o_den + al(b_den) + 16;
    uint8_t* blob = nullptr;
    NX_TRY(dev_alloc(ctx, total, (void**)&blob));
    hipError_t er = hipSuccess;
    // through the staging ring: the caller's arrays may be freed as soon as this (stream-ordered) entry returns, and a large program
    // copied straight from pageable memory would have its pages pinned in place by the runtime (internal.h, h_bounce)
    int up_rc = NX_OK;
    auto up = [&](size_t off, const void* src, size_t bytes) { if (er == hipSuccess && up_rc == NX_OK && bytes) up_rc = upload_async_staged(ctx, blob + off, src, bytes); };
    up(0, program, b_prog); up(o_cols, d_cols, b_cols); up(o_ec, econsts, b_ec); up(o_pw, alpha_powers, b_pw); up(o_den, denom_inv, b_den);
    if (up_rc != NX_OK) { dev_free(ctx, blob); return up_rc; }
    if (er == hipSuccess) {
        static std::atomic<uint64_t> attr_set{0};   // one bit per device: function attributes are per device
        if (!(attr_set.load() & (1ull << (ctx->device & 63)))) {
            er = hipFuncSetAttribute((const void*)constraint_program_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set.fetch_or(1ull << (ctx->device & 63));
        }
        const uint32_t n = 1u << log_eval;
        if (er == hipSuccess) {
            hipLaunchKernelGGL(constraint_program_kernel, dim3((n + CP_THREADS - 1) / CP_THREADS), dim3(CP_THREADS), (size_t)n_regs * CP_THREADS * 4, ctx->stream,
                               (const nx_cinstr*)blob, n_instr, (const u32* const*)(blob + o_cols), (const u32*)(blob + o_ec), (const u32*)(blob + o_pw),
                               (const u32*)(blob + o_den), (int)log_size, (int)log_eval, d_acc4[0], d_acc4[1], d_acc4[2], d_acc4[3]);
            er = hipGetLastError();
        }
    }
    hipError_t e2 = hipStreamSynchronize(ctx->stream);   // the host arrays and the blob must outlive the kernel
    dev_free(ctx, blob);
    if (er != hipSuccess) return hip_fail(ctx, er, "nx_eval_constraint_program", __FILE__, __LINE__);
    if (e2 != hipSuccess) return hip_fail(ctx, e2, "nx_eval_constraint_program(sync)", __FILE__, __LINE__);
    return NX_OK;
}
