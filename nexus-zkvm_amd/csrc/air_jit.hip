// Recorded AIR constraints compiled to a gfx950 kernel at run time (hiprtc): the fast form of nx_eval_constraint_program
// ("a HIP interpreter/JIT that evaluates the recorded expression DAG per row").
//
// The interpreter (constraints.hip) pays an LDS round trip and a scalar dispatch per instruction; a recorded program is
// straight-line code, so it can simply be emitted as HIP source — one statement per instruction over named registers, the
// alpha-power folding schedule resolved at generation time — and handed to the compiler, which keeps values in VGPRs and
// schedules the column loads.  Same semantics, same operand encoding, same accumulation as the interpreter; the lookup
// elements (econsts) and alpha powers stay run-time arguments, so one compilation serves every proof of an AIR.
#include "internal.h"
#include "air.h"
#include <hip/hiprtc.h>
#include <algorithm>
#include <string>
#include <vector>
#include <string.h>
#include <errno.h>
#include <stdlib.h>

#include <atomic>
#include <map>
#include <set>
#include <mutex>
#include <stdio.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <condition_variable>
#include <thread>
#include <dlfcn.h>
#include <spawn.h>
#include <sys/wait.h>
extern char** environ;

struct nx_air_kernel {
    nx_ctx* ctx;
    std::vector<hipModule_t> modules;    // one code object, or one per PART of the program's kernels (compiled side by side in helper processes: compile_parts)
    std::vector<hipFunction_t> fns;      // one kernel per program segment (air_kernel, air_kernel_1, ...), launched back to back
    uint32_t n_cols, n_econsts, n_constraints;
    uint32_t kind = 0;                    // 0: constraint kernels (nx_air_eval's argument list); 1: fraction kernels of nx_logup_program (another argument list)
    std::vector<char> code;              // the gfx950 code object the module was loaded from, or a container of several (nx_air_kernel_save; the disk cache)
};

namespace nx {

static const char* AIR_PRELUDE = R"SRC(
typedef unsigned int u32; typedef unsigned long long u64;
#define P 0x7fffffffu
// header-free on purpose: only compiler builtins, so hiprtc and an offline hipcc see the same text
#define FI __attribute__((device)) __attribute__((always_inline)) inline
#define G(p) ((__attribute__((address_space(1))) const u32*)(p))   // the column pointers are global memory: global_load, not flat_load
FI u32 m_csub(u32 s) { u32 d; bool b = __builtin_usub_overflow(s, P, &d); return b ? s : d; }
FI u32 m_add(u32 a, u32 b) { return m_csub(a + b); }
FI u32 m_sub(u32 a, u32 b) { u32 d; bool br = __builtin_usub_overflow(a, b, &d); return br ? d + P : d; }
FI u32 m_neg(u32 a) { return a ? P - a : 0; }
FI u32 m_mul(u32 a, u32 b) { u64 p = (u64)a * (u64)(b << 1); return m_csub((u32)(p >> 32) + ((u32)p >> 1)); }
FI u64 acc_mad(u64 acc, u32 a, u32 b) { return acc + (u64)a * (u64)b; }
FI u64 acc_fold(u64 x) { return (x & (u64)P) + (x >> 31); }
FI u32 acc_final(u64 x) { x = acc_fold(x); u32 s = ((u32)x & P) + (u32)(x >> 31); return m_csub(s); }
struct Q { u32 a, b, c, d; };
FI Q q_add(Q x, Q y) { Q r = {m_add(x.a, y.a), m_add(x.b, y.b), m_add(x.c, y.c), m_add(x.d, y.d)}; return r; }
FI Q q_sub(Q x, Q y) { Q r = {m_sub(x.a, y.a), m_sub(x.b, y.b), m_sub(x.c, y.c), m_sub(x.d, y.d)}; return r; }
// (xa + xb u)(ya + yb u), u^2 = 2 + i, CM31 = M31[i]/(i^2+1)
FI void c_mul(u32 xa, u32 xb, u32 ya, u32 yb, u32& ra, u32& rb) { ra = m_sub(m_mul(xa, ya), m_mul(xb, yb)); rb = m_add(m_mul(xa, yb), m_mul(xb, ya)); }
)SRC"
#ifndef NX_Q_MUL_NAIVE
R"SRC(// every coordinate of the product is four raw 64-bit multiply-adds and ONE reduction once y.b (2 + i) = (e, f) is formed (field.cuh q_mul)
FI Q q_mul(Q x, Q y) {
    const u32 e = m_sub(m_add(y.c, y.c), y.d), f = m_add(m_add(y.d, y.d), y.c);
    const u32 nyb = P - y.b, nyd = P - y.d, nf = P - f;
    Q r = {acc_final(acc_mad(acc_mad(acc_mad((u64)x.a * y.a, x.b, nyb), x.c, e), x.d, nf)),
           acc_final(acc_mad(acc_mad(acc_mad((u64)x.a * y.b, x.b, y.a), x.c, f), x.d, e)),
           acc_final(acc_mad(acc_mad(acc_mad((u64)x.a * y.c, x.b, nyd), x.c, y.a), x.d, nyb)),
           acc_final(acc_mad(acc_mad(acc_mad((u64)x.a * y.d, x.b, y.c), x.c, y.b), x.d, y.a))};
    return r;
}
)SRC"
#else          // A/B build (tools/build_variant_lib.sh): the four reduced CM31 products of rounds 1 - 5
R"SRC(FI Q q_mul(Q x, Q y) {
    u32 aa0, aa1, bb0, bb1, ab0, ab1, ba0, ba1;
    c_mul(x.a, x.b, y.a, y.b, aa0, aa1); c_mul(x.c, x.d, y.c, y.d, bb0, bb1);
    c_mul(x.a, x.b, y.c, y.d, ab0, ab1); c_mul(x.c, x.d, y.a, y.b, ba0, ba1);
    u32 r0 = m_sub(m_add(bb0, bb0), bb1), r1 = m_add(m_add(bb1, bb1), bb0);      // bb * (2 + i)
    Q r = {m_add(aa0, r0), m_add(aa1, r1), m_add(ab0, ba0), m_add(ab1, ba1)};
    return r;
}
)SRC"
#endif
R"SRC(FI u32 bitrev(u32 i, int log) { return log ? (__builtin_bitreverse32(i) >> (32 - log)) : i; }
FI u32 row_offset(u32 r, int log_size, int e, int offset) {
    if (offset == 0) return r;
    u32 idx = bitrev(r, e);
    const u32 half = 1u << (e - 1);
    const u32 step = (u32)offset << (e - log_size - 1);
    if (idx < half) idx = (idx + step) & (half - 1);
    else idx = ((idx - half - step) & (half - 1)) + half;
    return bitrev(idx, e);
}
)SRC";

// Straight-line code has no loops, so its size grows with the AIR: the reference-shaped machine with 250 logup columns is > 1 MB of
// instructions, far beyond the 64 KB instruction cache — every wave then streams its code from L2 and the kernel runs at a tenth of
// the memory rate (measured: 173 ms for 92 GB of column reads).  The generator therefore cuts the program at constraint boundaries
// into segments of bounded estimated code size; each segment becomes its own kernel holding exactly the instructions its
// constraints depend on (backward slice over the register file, original order kept), and the segments accumulate into the same
// rows one launch after the other.  A program that fits is one segment: the same source as ever.
static uint32_t instr_cost(uint32_t op) {   // rough gfx950 instruction counts of the emitted statements
    switch (op) {
    case NX_C_LOAD: return 6; case NX_C_CONST: return 1; case NX_C_ADD: case NX_C_SUB: return 4; case NX_C_MUL: return 7; case NX_C_NEG: return 3;
    case NX_C_CONSTE: return 4; case NX_C_ADDE: case NX_C_SUBE: return 16; case NX_C_MULE: return 85; case NX_C_MULEB: return 30; case NX_C_ADDEB: return 4;      // MULE / CONSTRAINT_E: the lazy q_mul of round 6 (160 / 170 before)
    case NX_C_LOADE: return 24; case NX_C_CONSTRAINT_B: return 14; case NX_C_CONSTRAINT_E: return 100; default: return 1;
    }
}
// estimated-instruction budget of one generated kernel: ~70 KB of code per kernel at the default 9000; measured sweep 1500..60000
// (profiles/r02_segment_sweep.txt): 4500-13000 is flat and best.  Per context ("air.segment").
static uint32_t segment_budget(const nx_ctx* ctx) {
    if (ctx) return (uint32_t)ctx->opt.air_segment;
    const char* e = getenv("NX_AIR_SEGMENT");   // source-only generation without a context (CPU suite): the process default
    return (uint32_t)std::max(200, e ? atoi(e) : 9000);
}

// ---- dot-product peephole (round 6) ---------------------------------------------------------------------------------------------------
// `MULEB T, A, v` immediately followed by `ADDE D, D, T` (either operand order) where T dies there is one term of Relation::combine's
// sum_k alpha^k value_k — what machine.hip's emit_den and a recorder's lowering of LookupElements::combine produce, 200 terms long for the
// reference's widest relation.  Emitted literally a term is 4 reduced products and 4 reduced sums (~36 VALU instructions); fused it is 4
// raw 64-bit multiply-adds (v_mad_u64_u32 with its free 64-bit addend) into a lazy accumulator of D that is folded every 4 terms and
// reduced ONCE, when D is next read — the same residue, a third of the instructions.  `NX_AIR_FUSE_DOT=0` emits the literal form (A/B).
struct RegRange { uint32_t reg, w; };
static void instr_ranges(const nx_cinstr& in, std::vector<RegRange>* reads, RegRange* write) {
    reads->clear(); *write = {0, 0};
    switch (in.op) {
    case NX_C_LOAD: case NX_C_CONST: *write = {in.dst, 1}; break;
    case NX_C_ADD: case NX_C_SUB: case NX_C_MUL: reads->push_back({in.a, 1}); reads->push_back({in.b, 1}); *write = {in.dst, 1}; break;
    case NX_C_NEG: reads->push_back({in.a, 1}); *write = {in.dst, 1}; break;
    case NX_C_CONSTE: case NX_C_LOADE: *write = {in.dst, 4}; break;
    case NX_C_ADDE: case NX_C_SUBE: case NX_C_MULE: reads->push_back({in.a, 4}); reads->push_back({in.b, 4}); *write = {in.dst, 4}; break;
    case NX_C_MULEB: case NX_C_ADDEB: reads->push_back({in.a, 4}); reads->push_back({in.b, 1}); *write = {in.dst, 4}; break;
    case NX_C_CONSTRAINT_B: reads->push_back({in.a, 1}); break;
    case NX_C_CONSTRAINT_E: reads->push_back({in.a, 4}); break;
    case NX_C_FRAC: reads->push_back({in.a, 4}); reads->push_back({in.b, 4}); break;
    case NX_C_FRACB: reads->push_back({in.a, 1}); reads->push_back({in.b, 4}); break;
    default: break;
    }
}
static bool fuse_dot_enabled() { static const bool on = [] { const char* e = getenv("NX_AIR_FUSE_DOT"); return !(e && *e == '0'); }(); return on; }
struct DotFusion { std::vector<char> skip, fused; };      // by position in `keep`: the MULEB that is not emitted, the ADDE that becomes 4 multiply-adds
static DotFusion find_dot_fusions(const nx_cinstr* prog, const std::vector<uint32_t>& keep, uint32_t n_regs) {
    DotFusion f; f.skip.assign(keep.size(), 0); f.fused.assign(keep.size(), 0);
    if (!fuse_dot_enabled()) return f;
    std::vector<char> live(n_regs + 8, 0);             // backward: is the register read again before it is rewritten (within this kernel)?
    std::vector<RegRange> reads; RegRange wr;
    auto disjoint = [](uint32_t a, uint32_t b) { return a + 4 <= b || b + 4 <= a; };
    for (size_t pos = keep.size(); pos-- > 0;) {
        const nx_cinstr& in = prog[keep[pos]];
        if (in.op == NX_C_ADDE && pos >= 1 && prog[keep[pos - 1]].op == NX_C_MULEB) {
            const nx_cinstr& mu = prog[keep[pos - 1]];
            const uint32_t T = mu.dst, D = in.dst;
            const bool a_is_t = in.a == T, b_is_t = in.b == T;
            const uint32_t other = a_is_t ? in.b : in.a;
            bool dead = true; for (uint32_t k = 0; k < 4; k++) dead = dead && !live[T + k];
            if (a_is_t != b_is_t && other == D && disjoint(T, D) && disjoint(T, mu.a) && disjoint(D, mu.a) && !(mu.b >= D && mu.b < D + 4) && !(mu.b >= T && mu.b < T + 4) && dead) {
                f.fused[pos] = 1; f.skip[pos - 1] = 1;
            }
        }
        instr_ranges(in, &reads, &wr);
        for (uint32_t k = 0; k < wr.w; k++) live[wr.reg + k] = 0;
        for (const RegRange& r : reads) for (uint32_t k = 0; k < r.w; k++) live[r.reg + k] = 1;
    }
    return f;
}
// the lazy accumulators of a kernel being emitted
struct LazyAcc {
    std::map<uint32_t, int> pending;            // D -> products since the last fold
    std::set<uint32_t> declared;
    static std::string z(uint32_t d, int k) { return "z" + std::to_string(d) + "_" + std::to_string(k); }
    std::string materialize(uint32_t d) {
        std::string s = "  ";
        for (int k = 0; k < 4; k++) s += "r" + std::to_string(d + k) + " = acc_final(" + z(d, k) + "); ";
        pending.erase(d);
        return s + "\n";
    }
    // everything a (literal) instruction touches must be in its registers
    std::string before(const nx_cinstr& in) {
        std::string s;
        if (pending.empty()) return s;
        std::vector<RegRange> reads; RegRange wr;
        instr_ranges(in, &reads, &wr);
        reads.push_back(wr);
        std::vector<uint32_t> hit;
        for (auto& kv : pending) for (const RegRange& r : reads) if (r.w && r.reg < kv.first + 4 && kv.first < r.reg + r.w) { hit.push_back(kv.first); break; }
        for (uint32_t d : hit) s += materialize(d);
        return s;
    }
    std::string accumulate(const nx_cinstr& mu, const nx_cinstr& add) {
        std::string s;
        const uint32_t D = add.dst;
        {   // the factors themselves must not be lazy
            nx_cinstr probe = mu; probe.dst = mu.a;           // reads of A (4) and v (1) through MULEB's own ranges; the write range points at A: harmless
            s += before(probe);
        }
        if (!pending.count(D)) {
            s += "  ";
            if (!declared.count(D)) { s += "u64 "; declared.insert(D); }
            for (int k = 0; k < 4; k++) s += z(D, k) + " = r" + std::to_string(D + k) + (k < 3 ? ", " : ";\n");
            pending[D] = 0;
        }
        s += "  ";
        for (int k = 0; k < 4; k++) s += z(D, k) + " = acc_mad(" + z(D, k) + ", r" + std::to_string(mu.a + k) + ", r" + std::to_string(mu.b) + "); ";
        s += "\n";
        if (++pending[D] == 4) {
            s += "  ";
            for (int k = 0; k < 4; k++) s += z(D, k) + " = acc_fold(" + z(D, k) + "); ";
            s += "\n";
            pending[D] = 0;
        }
        return s;
    }
};

static std::string generate_kernel(const nx_cinstr* prog, const std::vector<uint32_t>& keep, const std::vector<uint32_t>& cons_index, uint32_t n_regs, const std::string& name) {
    std::string s;
    s += "extern \"C\" __attribute__((global)) __attribute__((amdgpu_flat_work_group_size(256, 256))) void " + name + "(const u32* const* __restrict__ cols, const u32* __restrict__ econst, const u32* __restrict__ pw,\n"
         "    const u32* __restrict__ denom_inv, int log_size, int e, u32* a0, u32* a1, u32* a2, u32* a3, u32 row_begin, u32 row_end) {\n"
         "  const u32 r = row_begin + __builtin_amdgcn_workgroup_id_x() * 256 + __builtin_amdgcn_workitem_id_x();\n  if (r >= row_end) return;\n"
         "  u64 s0 = 0, s1 = 0, s2 = 0, s3 = 0;\n";
    // which row offsets occur
    std::vector<int> offs;
    for (uint32_t i : keep)
        if (prog[i].op == NX_C_LOAD || prog[i].op == NX_C_LOADE) { int o = (int)prog[i].b; if (std::find(offs.begin(), offs.end(), o) == offs.end()) offs.push_back(o); }
    auto off_name = [](int o) { return std::string("row_") + (o < 0 ? "m" : "p") + std::to_string(o < 0 ? -o : o); };
    for (int o : offs) s += "  const u32 " + off_name(o) + " = row_offset(r, log_size, e, " + std::to_string(o) + ");\n";
    for (uint32_t k = 0; k < n_regs; k++) s += (k % 16 == 0 ? std::string("  u32 ") : std::string(", ")) + "r" + std::to_string(k) + ((k % 16 == 15 || k + 1 == n_regs) ? " = 0;\n" : " = 0");
    auto R = [](uint32_t i) { return "r" + std::to_string(i); };
    auto E = [&](uint32_t i) { return "Q{" + R(i) + ", " + R(i + 1) + ", " + R(i + 2) + ", " + R(i + 3) + "}"; };
    auto setE = [&](uint32_t d, const std::string& expr) {
        return "  { const Q t_ = " + expr + "; " + R(d) + " = t_.a; " + R(d + 1) + " = t_.b; " + R(d + 2) + " = t_.c; " + R(d + 3) + " = t_.d; }\n";
    };
    uint32_t pending = 0;
    const std::string fold = "  s0 = acc_fold(s0); s1 = acc_fold(s1); s2 = acc_fold(s2); s3 = acc_fold(s3);\n";
    const DotFusion fu = find_dot_fusions(prog, keep, n_regs);
    LazyAcc lazy;
    for (size_t pos = 0; pos < keep.size(); pos++) {
        const uint32_t i = keep[pos];
        const nx_cinstr& in = prog[i];
        if (fu.skip[pos]) continue;
        if (fu.fused[pos]) { s += lazy.accumulate(prog[keep[pos - 1]], in); continue; }
        s += lazy.before(in);
        switch (in.op) {
        case NX_C_LOAD: s += "  " + R(in.dst) + " = G(cols[" + std::to_string(in.a) + "])[" + off_name((int)in.b) + "];\n"; break;
        case NX_C_CONST: s += "  " + R(in.dst) + " = " + std::to_string(in.a) + "u;\n"; break;
        case NX_C_ADD: s += "  " + R(in.dst) + " = m_add(" + R(in.a) + ", " + R(in.b) + ");\n"; break;
        case NX_C_SUB: s += "  " + R(in.dst) + " = m_sub(" + R(in.a) + ", " + R(in.b) + ");\n"; break;
        case NX_C_MUL: s += "  " + R(in.dst) + " = m_mul(" + R(in.a) + ", " + R(in.b) + ");\n"; break;
        case NX_C_NEG: s += "  " + R(in.dst) + " = m_neg(" + R(in.a) + ");\n"; break;
        case NX_C_CONSTE: { std::string b = std::to_string(4 * in.a); s += setE(in.dst, "Q{econst[" + b + "], econst[" + b + " + 1], econst[" + b + " + 2], econst[" + b + " + 3]}"); break; }
        case NX_C_ADDE: s += setE(in.dst, "q_add(" + E(in.a) + ", " + E(in.b) + ")"); break;
        case NX_C_SUBE: s += setE(in.dst, "q_sub(" + E(in.a) + ", " + E(in.b) + ")"); break;
        case NX_C_MULE: s += setE(in.dst, "q_mul(" + E(in.a) + ", " + E(in.b) + ")"); break;
        case NX_C_MULEB: s += setE(in.dst, "Q{m_mul(" + R(in.a) + ", " + R(in.b) + "), m_mul(" + R(in.a + 1) + ", " + R(in.b) + "), m_mul(" + R(in.a + 2) + ", " + R(in.b) + "), m_mul(" + R(in.a + 3) + ", " + R(in.b) + ")}"); break;
        case NX_C_ADDEB: s += setE(in.dst, "Q{m_add(" + R(in.a) + ", " + R(in.b) + "), " + R(in.a + 1) + ", " + R(in.a + 2) + ", " + R(in.a + 3) + "}"); break;
        case NX_C_LOADE: {
            std::string o = off_name((int)in.b);
            s += setE(in.dst, "Q{G(cols[" + std::to_string(in.a) + "])[" + o + "], G(cols[" + std::to_string(in.a + 1) + "])[" + o + "], G(cols[" + std::to_string(in.a + 2) + "])[" + o + "], G(cols[" +
                                  std::to_string(in.a + 3) + "])[" + o + "]}");
            break;
        }
        case NX_C_CONSTRAINT_B: {
            std::string b = std::to_string(4 * cons_index[i]);
            s += "  s0 = acc_mad(s0, pw[" + b + "], " + R(in.a) + "); s1 = acc_mad(s1, pw[" + b + " + 1], " + R(in.a) + "); s2 = acc_mad(s2, pw[" + b + " + 2], " + R(in.a) + "); s3 = acc_mad(s3, pw[" + b +
                 " + 3], " + R(in.a) + ");\n";
            if (++pending == 4) { s += fold; pending = 0; }
            break;
        }
        case NX_C_CONSTRAINT_E: {
            std::string b = std::to_string(4 * cons_index[i]);
            s += "  { const Q t_ = q_mul(Q{pw[" + b + "], pw[" + b + " + 1], pw[" + b + " + 2], pw[" + b + " + 3]}, " + E(in.a) + "); s0 += t_.a; s1 += t_.b; s2 += t_.c; s3 += t_.d; }\n";
            if (++pending == 4) { s += fold; pending = 0; }
            break;
        }
        default: break;
        }
    }
    s += "  const u32 di = denom_inv[r >> log_size];\n"
         "  a0[r] = m_add(a0[r], m_mul(acc_final(s0), di)); a1[r] = m_add(a1[r], m_mul(acc_final(s1), di));\n"
         "  a2[r] = m_add(a2[r], m_mul(acc_final(s2), di)); a3[r] = m_add(a3[r], m_mul(acc_final(s3), di));\n}\n";
    return s;
}

// register-level dependencies of a straight-line program: deps[i] = the instructions that last wrote the registers instruction i
// reads; cons_index[i] = the ordinal of constraint instruction i (its alpha power)
struct ProgDeps { std::vector<std::vector<uint32_t>> deps; std::vector<uint32_t> cons_index; uint32_t n_constraints = 0; };
static ProgDeps program_deps(const nx_cinstr* prog, uint32_t n_instr, uint32_t n_regs) {
    ProgDeps pd;
    std::vector<int> last(n_regs, -1);
    pd.deps.resize(n_instr); pd.cons_index.assign(n_instr, 0);
    auto& deps = pd.deps; auto& cons_index = pd.cons_index; uint32_t& n_c = pd.n_constraints;
    auto use = [&](uint32_t i, uint32_t reg, uint32_t w) { for (uint32_t k = 0; k < w; k++) if (reg + k < n_regs && last[reg + k] >= 0) deps[i].push_back((uint32_t)last[reg + k]); };
    auto def = [&](uint32_t i, uint32_t reg, uint32_t w) { for (uint32_t k = 0; k < w; k++) if (reg + k < n_regs) last[reg + k] = (int)i; };
    for (uint32_t i = 0; i < n_instr; i++) {
        const nx_cinstr& in = prog[i];
        switch (in.op) {
        case NX_C_LOAD: case NX_C_CONST: def(i, in.dst, 1); break;
        case NX_C_ADD: case NX_C_SUB: case NX_C_MUL: use(i, in.a, 1); use(i, in.b, 1); def(i, in.dst, 1); break;
        case NX_C_NEG: use(i, in.a, 1); def(i, in.dst, 1); break;
        case NX_C_CONSTE: case NX_C_LOADE: def(i, in.dst, 4); break;
        case NX_C_ADDE: case NX_C_SUBE: case NX_C_MULE: use(i, in.a, 4); use(i, in.b, 4); def(i, in.dst, 4); break;
        case NX_C_MULEB: case NX_C_ADDEB: use(i, in.a, 4); use(i, in.b, 1); def(i, in.dst, 4); break;
        case NX_C_CONSTRAINT_B: use(i, in.a, 1); cons_index[i] = n_c++; break;
        case NX_C_CONSTRAINT_E: use(i, in.a, 4); cons_index[i] = n_c++; break;
        case NX_C_FRAC: use(i, in.a, 4); use(i, in.b, 4); break;
        case NX_C_FRACB: use(i, in.a, 1); use(i, in.b, 4); break;
        default: break;
        }
    }
    return pd;
}

// `select` (one flag per constraint, NULL = all): the kernels evaluate only the selected constraints — same alpha-power indices, same
// column indices as the whole program, so a component's constraints can be evaluated in parts on different domains (the
// degree-aware composition of prover.hip) and the parts add up to the whole.
static std::string generate_air_source(const nx_ctx* ctx, const nx_cinstr* prog, uint32_t n_instr, uint32_t n_regs, uint32_t* n_kernels = nullptr, const uint8_t* select = nullptr) {
    const ProgDeps pd = program_deps(prog, n_instr, n_regs);
    const auto& deps = pd.deps; const auto& cons_index = pd.cons_index;
    // segments: consecutive constraints whose slices fit the budget
    std::string s = AIR_PRELUDE;
    uint32_t n_seg = 0;
    std::vector<char> in_seg(n_instr, 0);
    std::vector<uint32_t> seg_cons;
    uint32_t cost = 0;
    auto add_slice = [&](uint32_t root) {        // marks the slice of `root`, returns the added cost
        uint32_t added = 0;
        std::vector<uint32_t> st{root};
        while (!st.empty()) {
            uint32_t i = st.back(); st.pop_back();
            if (in_seg[i]) continue;
            in_seg[i] = 1; added += instr_cost(prog[i].op);
            for (uint32_t d : deps[i]) if (!in_seg[d]) st.push_back(d);
        }
        return added;
    };
    auto flush = [&]() {
        std::vector<uint32_t> keep;
        for (uint32_t i = 0; i < n_instr; i++) if (in_seg[i]) keep.push_back(i);
        s += generate_kernel(prog, keep, cons_index, n_regs, n_seg == 0 ? std::string("air_kernel") : "air_kernel_" + std::to_string(n_seg));
        n_seg++;
        std::fill(in_seg.begin(), in_seg.end(), 0); cost = 0;
    };
    const uint32_t budget = segment_budget(ctx);
    bool any = false;
    for (uint32_t i = 0; i < n_instr; i++) {
        if (prog[i].op != NX_C_CONSTRAINT_B && prog[i].op != NX_C_CONSTRAINT_E) continue;
        if (select && !select[cons_index[i]]) continue;
        if (any && cost >= budget) { flush(); any = false; }
        cost += add_slice(i); any = true;
    }
    if (any || n_seg == 0) flush();
    if (n_kernels) *n_kernels = n_seg;
    return s;
}

}  // namespace nx

namespace nx {

// Upper bound of every constraint's degree in the trace columns (a validated program): a column load is 1, a constant 0, a sum the
// larger, a product the sum.  The rule behind FrameworkEval::max_constraint_log_degree_bound (reference
// prover/src/components/mod.rs:44-45): a constraint of degree d over columns of 2^n rows has a quotient in the FFT space of
// 2^(n+e) points iff d <= 2^e + 1.
void air_constraint_degrees(const nx_cinstr* prog, uint32_t n_instr, uint32_t n_regs, std::vector<uint32_t>* out) {
    std::vector<uint32_t> d(n_regs + 4, 0);
    out->clear();
    auto dE = [&](uint32_t r) { return std::max(std::max(d[r], d[r + 1]), std::max(d[r + 2], d[r + 3])); };
    auto setE = [&](uint32_t r, uint32_t v) { d[r] = d[r + 1] = d[r + 2] = d[r + 3] = v; };
    for (uint32_t i = 0; i < n_instr; i++) {
        const nx_cinstr& in = prog[i];
        switch (in.op) {
        case NX_C_LOAD: d[in.dst] = 1; break;
        case NX_C_CONST: d[in.dst] = 0; break;
        case NX_C_ADD: case NX_C_SUB: d[in.dst] = std::max(d[in.a], d[in.b]); break;
        case NX_C_MUL: d[in.dst] = d[in.a] + d[in.b]; break;
        case NX_C_NEG: d[in.dst] = d[in.a]; break;
        case NX_C_CONSTE: setE(in.dst, 0); break;
        case NX_C_LOADE: setE(in.dst, 1); break;
        case NX_C_ADDE: case NX_C_SUBE: setE(in.dst, std::max(dE(in.a), dE(in.b))); break;
        case NX_C_MULE: setE(in.dst, dE(in.a) + dE(in.b)); break;
        case NX_C_MULEB: setE(in.dst, dE(in.a) + d[in.b]); break;
        case NX_C_ADDEB: setE(in.dst, std::max(dE(in.a), d[in.b])); break;
        case NX_C_CONSTRAINT_B: out->push_back(d[in.a]); break;
        case NX_C_CONSTRAINT_E: out->push_back(dE(in.a)); break;
        default: break;
        }
    }
}

// Per constraint: does its value depend on a column read at a non-zero row offset (the same forward pass as the degrees)?  A constraint
// that reads only the current row can be evaluated on ANY set of points of the evaluation domain, e.g. an N-point sub-domain on
// which only those columns were evaluated (prover.hip, the quarter-domain part of a degree-4 component).
void air_constraint_neighbours(const nx_cinstr* prog, uint32_t n_instr, uint32_t n_regs, std::vector<char>* out) {
    std::vector<char> f(n_regs + 4, 0);
    out->clear();
    auto fE = [&](uint32_t r) { return (char)(f[r] | f[r + 1] | f[r + 2] | f[r + 3]); };
    auto setE = [&](uint32_t r, char v) { f[r] = f[r + 1] = f[r + 2] = f[r + 3] = v; };
    for (uint32_t i = 0; i < n_instr; i++) {
        const nx_cinstr& in = prog[i];
        switch (in.op) {
        case NX_C_LOAD: f[in.dst] = in.b != 0; break;
        case NX_C_CONST: f[in.dst] = 0; break;
        case NX_C_ADD: case NX_C_SUB: case NX_C_MUL: f[in.dst] = f[in.a] | f[in.b]; break;
        case NX_C_NEG: f[in.dst] = f[in.a]; break;
        case NX_C_CONSTE: setE(in.dst, 0); break;
        case NX_C_LOADE: setE(in.dst, in.b != 0); break;
        case NX_C_ADDE: case NX_C_SUBE: case NX_C_MULE: setE(in.dst, fE(in.a) | fE(in.b)); break;
        case NX_C_MULEB: case NX_C_ADDEB: setE(in.dst, fE(in.a) | f[in.b]); break;
        case NX_C_CONSTRAINT_B: out->push_back(f[in.a]); break;
        case NX_C_CONSTRAINT_E: out->push_back(fE(in.a)); break;
        default: break;
        }
    }
}

// The component columns the selected constraints read (the backward slices of generate_air_source).
void air_subset_columns(const nx_cinstr* prog, uint32_t n_instr, uint32_t n_regs, uint32_t n_cols, const uint8_t* select, std::vector<char>* used) {
    const ProgDeps pd = program_deps(prog, n_instr, n_regs);
    used->assign(n_cols, 0);
    std::vector<char> seen(n_instr, 0);
    std::vector<uint32_t> st;
    for (uint32_t i = 0; i < n_instr; i++)
        if ((prog[i].op == NX_C_CONSTRAINT_B || prog[i].op == NX_C_CONSTRAINT_E) && (!select || select[pd.cons_index[i]])) st.push_back(i);
    while (!st.empty()) {
        const uint32_t i = st.back(); st.pop_back();
        if (seen[i]) continue;
        seen[i] = 1;
        if (prog[i].op == NX_C_LOAD && prog[i].a < n_cols) (*used)[prog[i].a] = 1;
        if (prog[i].op == NX_C_LOADE) for (uint32_t k = 0; k < 4; k++) if (prog[i].a + k < n_cols) (*used)[prog[i].a + k] = 1;
        for (uint32_t d : pd.deps[i]) if (!seen[d]) st.push_back(d);
    }
}

// Shared by nx_air_compile, nx_eval_constraint_program's callers and the prover session (GenericAir::check): every register index
// (dst, a, b and the +3 of the secure-field ops), column index and secure-constant index of a recorded program is inside the
// announced bounds.  A caller mistake across the C ABI is NX_ERR_ARG, never an out-of-bounds access.
int validate_air_program(nx_ctx* ctx, const nx_cinstr* program, uint32_t n_instr, uint32_t n_regs, uint32_t n_cols, uint32_t n_econsts, uint32_t* n_constraints_out) {
    if (n_instr && !program) return set_err(ctx, NX_ERR_ARG, "recorded AIR: NULL program");
    if (n_regs == 0 || n_regs > 4096) return set_err(ctx, NX_ERR_ARG, "recorded AIR: register count out of range");
    uint32_t n_c = 0;
    for (uint32_t i = 0; i < n_instr; i++) {
        const nx_cinstr& in = program[i];
        auto reg_ok = [&](uint32_t rg, uint32_t width) { return rg <= n_regs && width <= n_regs - rg; };
        bool ok = true;
        switch (in.op) {
        case NX_C_LOAD: ok = reg_ok(in.dst, 1) && in.a < n_cols; break;
        case NX_C_CONST: ok = reg_ok(in.dst, 1) && in.a < P; break;
        case NX_C_ADD: case NX_C_SUB: case NX_C_MUL: ok = reg_ok(in.dst, 1) && reg_ok(in.a, 1) && reg_ok(in.b, 1); break;
        case NX_C_NEG: ok = reg_ok(in.dst, 1) && reg_ok(in.a, 1); break;
        case NX_C_CONSTE: ok = reg_ok(in.dst, 4) && in.a < n_econsts; break;
        case NX_C_ADDE: case NX_C_SUBE: case NX_C_MULE: ok = reg_ok(in.dst, 4) && reg_ok(in.a, 4) && reg_ok(in.b, 4); break;
        case NX_C_MULEB: case NX_C_ADDEB: ok = reg_ok(in.dst, 4) && reg_ok(in.a, 4) && reg_ok(in.b, 1); break;
        case NX_C_LOADE: ok = reg_ok(in.dst, 4) && in.a < n_cols && n_cols - in.a >= 4; break;
        case NX_C_CONSTRAINT_B: ok = reg_ok(in.a, 1); n_c++; break;
        case NX_C_CONSTRAINT_E: ok = reg_ok(in.a, 4); n_c++; break;
        default: ok = false;
        }
        if (!ok) return set_err(ctx, NX_ERR_ARG, "recorded AIR: malformed instruction " + std::to_string(i));
    }
    if (n_constraints_out) *n_constraints_out = n_c;
    return NX_OK;
}
// what a compiled kernel was built for (the prover session rejects a kernel that does not match its component)
void air_kernel_shape(const nx_air_kernel* k, uint32_t* n_cols, uint32_t* n_econsts, uint32_t* n_constraints) {
    *n_cols = k->n_cols; *n_econsts = k->n_econsts; *n_constraints = k->n_constraints;
}

}  // namespace nx

// ---- ahead-of-time kernels (the hiprtc compilation — 0.7 s for the headline AIR, 8 s for the keccak-shaped one — sat
// outside every number and every proof of a fresh process paid it).  A compiled kernel is a self-describing blob: a 48-byte header
// (magic, version, kernel / column / constant / constraint counts, code size, FNV-1a of the code) + the gfx950 code object.
//   nx_air_kernel_save / nx_air_kernel_load   the blob of a kernel / a kernel from a blob: build once (a build step, another box — the
//                                              compilation needs no GPU of the same process), ship next to the binary, load in ms
//   nx_air_cache_dir                           a directory the library keeps those blobs in, keyed by a hash of the generated source,
//                                              the target and the hiprtc version: nx_air_compile* (and the prover, which compiles the
//                                              components' kernels through it) look there first and store what they compile
namespace nx {
namespace {
struct BlobHeader { uint32_t magic, version, n_kernels, n_cols, n_econsts, n_constraints; uint64_t code_size, code_hash; uint64_t reserved; };
constexpr uint32_t BLOB_MAGIC = 0x4B41584Eu /* "NXAK" */, BLOB_VERSION = 1;
uint64_t fnv1a(const void* p, size_t n, uint64_t h = 0xcbf29ce484222325ull) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}
struct CacheState { std::mutex mu; bool init = false; std::string dir; std::atomic<uint64_t> compiled{0}, disk_hits{0}, stored{0}; };
CacheState& cache_state() { static CacheState s; return s; }
// The cache directory holds CODE OBJECTS that are loaded into the GPU on a 64-bit FNV match — a checksum, not a signature — so it is
// trusted input: created 0700, and refused (no disk cache, one line on stderr) when it is not a directory owned by this user or when
// group / others may write to it.
std::string trusted_cache_dir(const char* dir) {
    (void)mkdir(dir, 0700);                                        // one level; an existing directory is fine
    struct stat st;
    if (stat(dir, &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) {
        fprintf(stderr, "nexus_hip: the AIR kernel cache directory '%s' is not a directory owned by this user with mode go-w; kernels are not cached on disk\n", dir);
        return std::string();
    }
    return dir;
}
std::string cache_dir() {
    CacheState& s = cache_state();
    std::lock_guard<std::mutex> lk(s.mu);
    if (!s.init) { const char* e = getenv("NX_AIR_CACHE_DIR"); if (e && *e) s.dir = trusted_cache_dir(e); s.init = true; }
    return s.dir;
}
// The code region of a kernel blob: ONE gfx950 code object (an ELF image), or — a program of many kernels, compiled in parts — a container
// "NXMM", u32 n_parts, then per part {u32 first_kernel, u32 n_kernels, u64 size}, then the parts' code objects, each 8-byte aligned.
constexpr uint32_t PARTS_MAGIC = 0x4D4D584Eu /* "NXMM" */;
struct PartDesc { uint32_t first_kernel, n_kernels; uint64_t size; };
static void unload_modules(nx_air_kernel* k) { for (hipModule_t m : k->modules) (void)hipModuleUnload(m); k->modules.clear(); }
int load_code(nx_ctx* ctx, const BlobHeader& h, const char* code, nx_air_kernel** out) {
    nx_air_kernel* k = new nx_air_kernel();
    k->ctx = ctx; k->n_cols = h.n_cols; k->n_econsts = h.n_econsts; k->n_constraints = h.n_constraints; k->kind = (uint32_t)h.reserved;
    k->code.assign(code, code + h.code_size);
    std::vector<PartDesc> parts; std::vector<size_t> offs;
    uint32_t magic = 0; if (h.code_size >= 8) memcpy(&magic, k->code.data(), 4);
    if (magic == PARTS_MAGIC) {
        uint32_t n_parts = 0; memcpy(&n_parts, k->code.data() + 4, 4);
        size_t off = 8 + (size_t)n_parts * sizeof(PartDesc);
        if (n_parts == 0 || n_parts > 4096 || off > h.code_size) { delete k; return set_err(ctx, NX_ERR_ARG, "air kernel blob: malformed part table"); }
        parts.resize(n_parts);
        memcpy(parts.data(), k->code.data() + 8, (size_t)n_parts * sizeof(PartDesc));
        for (const PartDesc& pd : parts) {
            off = (off + 7) & ~(size_t)7;
            if (pd.size > h.code_size || off > h.code_size - pd.size || (uint64_t)pd.first_kernel + pd.n_kernels > h.n_kernels) { delete k; return set_err(ctx, NX_ERR_ARG, "air kernel blob: malformed part table"); }
            offs.push_back(off); off += pd.size;
        }
    } else { parts.push_back({0u, h.n_kernels, h.code_size}); offs.push_back(0); }
    k->fns.assign(h.n_kernels, nullptr);
    hipError_t e = hipSuccess;
    for (size_t q = 0; q < parts.size() && e == hipSuccess; q++) {
        hipModule_t m;
        e = hipModuleLoadData(&m, k->code.data() + offs[q]);
        if (e != hipSuccess) break;
        k->modules.push_back(m);
        for (uint32_t f = parts[q].first_kernel; f < parts[q].first_kernel + parts[q].n_kernels && e == hipSuccess; f++)
            e = hipModuleGetFunction(&k->fns[f], m, f == 0 ? "air_kernel" : ("air_kernel_" + std::to_string(f)).c_str());
    }
    for (hipFunction_t f : k->fns) if (e == hipSuccess && !f) e = hipErrorNotFound;
    if (e != hipSuccess) { unload_modules(k); delete k; return hip_fail(ctx, e, "hipModuleLoadData / hipModuleGetFunction(air kernel)", __FILE__, __LINE__); }
    *out = k;
    return NX_OK;
}
bool blob_ok(const uint8_t* blob, size_t n, BlobHeader* h) {
    if (!blob || n < sizeof(BlobHeader)) return false;
    memcpy(h, blob, sizeof *h);
    return h->magic == BLOB_MAGIC && h->version == BLOB_VERSION && h->n_kernels >= 1 && h->n_kernels <= 4096 && h->code_size == n - sizeof(BlobHeader) &&
           h->code_hash == fnv1a(blob + sizeof(BlobHeader), (size_t)h->code_size);
}
std::vector<uint8_t> make_blob(const nx_air_kernel* k) {
    BlobHeader h = {BLOB_MAGIC, BLOB_VERSION, (uint32_t)k->fns.size(), k->n_cols, k->n_econsts, k->n_constraints, (uint64_t)k->code.size(), fnv1a(k->code.data(), k->code.size()), k->kind};
    std::vector<uint8_t> b(sizeof h + k->code.size());
    memcpy(b.data(), &h, sizeof h); memcpy(b.data() + sizeof h, k->code.data(), k->code.size());
    return b;
}
// The optimisation level handed to hiprtc ("NX_AIR_HIPRTC_OPT"; part of the cache key).  Default -O1: the generated kernels are straight-line
// code whose schedule the generator already fixed — measured (round 5, same box): first prove of the keccak-shaped statement 15.7 s at -O3, 14.3 s
// at -O2, 10.8 s at -O1, with the same steady-state time (36.8 / 36.5 / 36.8 ms; headline 42.5 vs 42.2 ms, v1-shaped 142.9 vs 141.2 ms).
const char* hiprtc_opt_level() { static const std::string s = [] { const char* e = getenv("NX_AIR_HIPRTC_OPT"); return std::string(e && *e ? e : "-O1"); }(); return s.c_str(); }
std::string cache_path(const std::string& dir, const std::string& src) {
    int maj = 0, min = 0; (void)hiprtcVersion(&maj, &min);
    const std::string salt = std::string("|gfx950|") + hiprtc_opt_level() + "|hiprtc " + std::to_string(maj) + "." + std::to_string(min) + "|blob " + std::to_string(BLOB_VERSION);
    const uint64_t a = fnv1a(salt.data(), salt.size(), fnv1a(src.data(), src.size())), b = fnv1a(src.data(), src.size(), 0x9E3779B97F4A7C15ull ^ src.size());
    char name[64]; snprintf(name, sizeof name, "/nxair-%016llx%016llx.nxak", (unsigned long long)a, (unsigned long long)b);
    return dir + name;
}
}  // namespace
}  // namespace nx


namespace nx { int compile_source(nx_ctx* ctx, const std::string& src, uint32_t n_kernels, uint32_t n_cols, uint32_t n_econsts, uint32_t n_constraints, nx_air_kernel** out, uint32_t kind = 0); }
using namespace nx;

extern "C" {

// Validates like nx_eval_constraint_program, generates the source, compiles it for gfx950 and loads the module.
// h_source_out (optional): receives a malloc'd copy of the generated source (free with nx_free_host) — also usable without a GPU.
int nx_air_compile(nx_ctx* ctx, const nx_cinstr* program, uint32_t n_instr, uint32_t n_regs, uint32_t n_cols, uint32_t n_econsts, uint32_t n_constraints,
                   nx_air_kernel** out, char** h_source_out) {
    return nx_air_compile_subset(ctx, program, n_instr, n_regs, n_cols, n_econsts, n_constraints, nullptr, out, h_source_out);
}

int nx_air_constraint_degrees(nx_ctx* ctx, const nx_cinstr* program, uint32_t n_instr, uint32_t n_regs, uint32_t n_cols, uint32_t n_econsts, uint32_t n_constraints,
                              uint32_t* degrees) {
    NX_GUARD(ctx);
    if (!program || !degrees) return set_err(ctx, NX_ERR_ARG, "nx_air_constraint_degrees: NULL argument");
    uint32_t n_c = 0;
    NX_TRY(validate_air_program(ctx, program, n_instr, n_regs, n_cols, n_econsts, &n_c));
    if (n_c != n_constraints) return set_err(ctx, NX_ERR_ARG, "nx_air_constraint_degrees: the program adds a different number of constraints than announced");
    std::vector<uint32_t> d;
    air_constraint_degrees(program, n_instr, n_regs, &d);
    std::copy(d.begin(), d.end(), degrees);
    return NX_OK;
}

int nx_air_compile_subset(nx_ctx* ctx, const nx_cinstr* program, uint32_t n_instr, uint32_t n_regs, uint32_t n_cols, uint32_t n_econsts, uint32_t n_constraints,
                          const uint8_t* select, nx_air_kernel** out, char** h_source_out) {
    NX_GUARD(ctx);
    if (!program || (!out && !h_source_out)) return set_err(ctx, NX_ERR_ARG, "nx_air_compile: NULL argument");
    uint32_t n_c = 0;
    NX_TRY(validate_air_program(ctx, program, n_instr, n_regs, n_cols, n_econsts, &n_c));
    if (n_c != n_constraints) return set_err(ctx, NX_ERR_ARG, "nx_air_compile: the program adds a different number of constraints than announced");
    uint32_t n_kernels = 1;
    const std::string src = generate_air_source(ctx, program, n_instr, n_regs, &n_kernels, select);
    if (h_source_out) { *h_source_out = (char*)malloc(src.size() + 1); if (*h_source_out) std::copy(src.c_str(), src.c_str() + src.size() + 1, *h_source_out); }
    if (!out) return NX_OK;
    if (!ctx) return set_err(ctx, NX_ERR_ARG, "nx_air_compile: a context is needed to load the kernel");
    return compile_source(ctx, src, n_kernels, n_cols, n_econsts, n_constraints, out);
}

}  // extern "C"

namespace nx {
// ---- compilation: in this process, or — a program of many kernels — its parts side by side in helper PROCESSES ---------------------------
// hiprtc serialises the threads of a process (measured in round 5: 8 threads, no gain), and the first proof of a wide AIR waited 10 s (the
// keccak-shaped statement; 31 s at the reference's tuple widths) for ~80 kernels compiled one after the other.  The kernels of a generated
// source are independent: the source is cut into PARTS of PART_KERNELS kernels (prelude + that kernel: a function of the source alone, so
// the blob is the same bytes however it was compiled), and up to "NX_AIR_COMPILE_PROCS" helpers — the executable nx_air_cc beside this
// library (csrc/host/air_cc.cpp; NX_AIR_CC overrides the path) — compile one part each.  No helper, one process allowed, or a short
// program: the parts (or the whole source) go through hiprtc here.
constexpr uint32_t PART_KERNELS = 1, PARTS_MIN_KERNELS = 4;
static const char* KERNEL_MARK = "extern \"C\" __attribute__((global))";

static int hiprtc_compile(nx_ctx* ctx, const std::string& src, std::vector<char>* code) {
    hiprtcProgram rp;
    if (hiprtcCreateProgram(&rp, src.c_str(), "nx_air_kernel.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return set_err(ctx, NX_ERR_HIP, "hiprtcCreateProgram failed");
    const char* opts[] = {"--offload-arch=gfx950", hiprtc_opt_level()};
    hiprtcResult cr = hiprtcCompileProgram(rp, 2, opts);
    if (cr != HIPRTC_SUCCESS) {
        size_t ls = 0; (void)hiprtcGetProgramLogSize(rp, &ls);
        std::string log(ls, '\0'); if (ls) (void)hiprtcGetProgramLog(rp, &log[0]);
        (void)hiprtcDestroyProgram(&rp);
        return set_err(ctx, NX_ERR_HIP, "hiprtc compilation of the recorded AIR failed: " + log.substr(0, 400));
    }
    size_t cs = 0; (void)hiprtcGetCodeSize(rp, &cs);
    code->resize(cs);
    (void)hiprtcGetCode(rp, code->data());
    (void)hiprtcDestroyProgram(&rp);
    return NX_OK;
}
static int compile_procs() { const char* e = getenv("NX_AIR_COMPILE_PROCS"); const int hw = (int)std::thread::hardware_concurrency(); return e && *e ? atoi(e) : std::max(1, std::min(32, hw)); }   // read per compilation: a tool may flip it; more than 32 measured no faster (profiles/r06_compile_procs.jsonl)
static std::string helper_path() {
    static const std::string path = [] {
        const char* e = getenv("NX_AIR_CC");
        std::string p;
        if (e && *e) p = e;
        else { Dl_info info; if (dladdr((const void*)&helper_path, &info) && info.dli_fname) { p = info.dli_fname; const size_t sl = p.rfind('/'); p = (sl == std::string::npos ? std::string(".") : p.substr(0, sl)) + "/nx_air_cc"; } }
        return !p.empty() && access(p.c_str(), X_OK) == 0 ? p : std::string();
    }();
    return path;
}
// at most compile_procs() helpers of this process at a time, whichever thread started them
struct ProcGate { std::mutex mu; std::condition_variable cv; int running = 0; };
static ProcGate& proc_gate() { static ProcGate g; return g; }
static bool read_file(const std::string& path, std::vector<char>* out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[1 << 16]; size_t got; out->clear();
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) out->insert(out->end(), buf, buf + got);
    fclose(f);
    return true;
}
// src -> the code region of its blob (one code object, or the "NXMM" container of its parts)
static int compile_parts(nx_ctx* ctx, const std::string& src, uint32_t n_kernels, std::vector<char>* code) {
    std::vector<size_t> marks;
    for (size_t p = src.find(KERNEL_MARK); p != std::string::npos; p = src.find(KERNEL_MARK, p + 1)) marks.push_back(p);
    if (n_kernels < PARTS_MIN_KERNELS || marks.size() != n_kernels) return hiprtc_compile(ctx, src, code);
    const std::string prelude = src.substr(0, marks[0]);
    std::vector<std::string> part_src; std::vector<PartDesc> parts;
    for (uint32_t k0 = 0; k0 < n_kernels; k0 += PART_KERNELS) {
        const uint32_t k1 = std::min(n_kernels, k0 + PART_KERNELS);
        part_src.push_back(prelude + src.substr(marks[k0], (k1 < n_kernels ? marks[k1] : src.size()) - marks[k0]));
        parts.push_back({k0, k1 - k0, 0});
    }
    std::vector<std::vector<char>> objs(parts.size());
    const std::string helper = helper_path();
    const int procs = compile_procs();
    // the hiprtc this process runs (PyTorch carries its own): the helpers load the same file, so a part is the same bytes wherever it is compiled
    const std::string rtc_lib = [] { Dl_info info; return dladdr((const void*)&hiprtcCompileProgram, &info) && info.dli_fname ? std::string(info.dli_fname) : std::string(); }();
    std::vector<char> done(parts.size(), 0);
    if (procs > 1 && !helper.empty()) {
        const char* tmp = getenv("TMPDIR");
        std::string dir = std::string(tmp && *tmp ? tmp : "/tmp") + "/nxaircc.XXXXXX";
        if (mkdtemp(&dir[0])) {
            std::vector<pid_t> pid(parts.size(), -1);
            ProcGate& g = proc_gate();
            auto reap = [&](size_t q) {
                int st = 0;
                if (pid[q] > 0) {
                    pid_t w;
                    do w = waitpid(pid[q], &st, 0); while (w < 0 && errno == EINTR);
                    // the slot is given back whatever waitpid said (ECHILD: the embedding process ignores SIGCHLD and the kernel reaped the
                    // helper — its part is then compiled here): a slot that is never returned would park a later compilation for ever
                    { std::lock_guard<std::mutex> lk(g.mu); g.running--; } g.cv.notify_one();
                    if (w == pid[q] && WIFEXITED(st) && WEXITSTATUS(st) == 0 && read_file(dir + "/p" + std::to_string(q) + ".co", &objs[q]) && !objs[q].empty()) done[q] = 1;
                }
                pid[q] = -1;
            };
            size_t next_reap = 0;
            for (size_t q = 0; q < parts.size(); q++) {
                const std::string in = dir + "/p" + std::to_string(q) + ".hip", outp = dir + "/p" + std::to_string(q) + ".co";
                FILE* f = fopen(in.c_str(), "wb");
                if (!f) continue;
                const bool ok = fwrite(part_src[q].data(), 1, part_src[q].size(), f) == part_src[q].size();
                if (fclose(f) != 0 || !ok) continue;
                {   // wait for a slot; a thread that holds every slot itself reaps its oldest helper first
                    std::unique_lock<std::mutex> lk(g.mu);
                    while (g.running >= procs) {
                        if (next_reap < q) { lk.unlock(); while (next_reap < q && pid[next_reap] <= 0) next_reap++; if (next_reap < q) reap(next_reap++); lk.lock(); }
                        else g.cv.wait(lk);
                    }
                    g.running++;
                }
                const char* argv[] = {helper.c_str(), in.c_str(), outp.c_str(), hiprtc_opt_level(), rtc_lib.c_str(), nullptr};
                pid_t child = -1;
                if (posix_spawn(&child, helper.c_str(), nullptr, nullptr, (char* const*)argv, environ) != 0) { std::lock_guard<std::mutex> lk(g.mu); g.running--; child = -1; }
                pid[q] = child;
            }
            for (size_t q = 0; q < parts.size(); q++) if (pid[q] > 0) reap(q);
            for (size_t q = 0; q < parts.size(); q++) { (void)remove((dir + "/p" + std::to_string(q) + ".hip").c_str()); (void)remove((dir + "/p" + std::to_string(q) + ".co").c_str()); (void)remove((dir + "/p" + std::to_string(q) + ".co.log").c_str()); }
            (void)rmdir(dir.c_str());
        }
    }
    for (size_t q = 0; q < parts.size(); q++)            // whatever no helper delivered (no helper at all; a compile error: hiprtc reports it here)
        if (!done[q]) NX_TRY(hiprtc_compile(ctx, part_src[q], &objs[q]));
    const uint32_t n_parts = (uint32_t)parts.size();
    code->clear();
    code->resize(8 + (size_t)n_parts * sizeof(PartDesc));
    memcpy(code->data(), &PARTS_MAGIC, 4); memcpy(code->data() + 4, &n_parts, 4);
    for (size_t q = 0; q < parts.size(); q++) {
        parts[q].size = objs[q].size();
        code->resize((code->size() + 7) & ~(size_t)7, 0);
        code->insert(code->end(), objs[q].begin(), objs[q].end());
    }
    memcpy(code->data() + 8, parts.data(), (size_t)n_parts * sizeof(PartDesc));
    return NX_OK;
}
}  // namespace nx

namespace nx {
// generated source -> loaded kernels: the cache directory first, else hiprtc (and the directory is fed)
int compile_source(nx_ctx* ctx, const std::string& src, uint32_t n_kernels, uint32_t n_cols, uint32_t n_econsts, uint32_t n_constraints, nx_air_kernel** out, uint32_t kind) {
    const std::string dir = cache_dir();
    const std::string path = dir.empty() ? std::string() : cache_path(dir, src);
    if (!path.empty()) {                                  // the disk cache: a blob stored by an earlier process (or a build step)
        FILE* f = fopen(path.c_str(), "rb");
        if (f) {
            std::vector<uint8_t> b;
            uint8_t buf[1 << 16]; size_t got;
            while ((got = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + got);
            fclose(f);
            BlobHeader h;
            if (blob_ok(b.data(), b.size(), &h) && h.n_kernels == n_kernels && h.n_cols == n_cols && h.n_econsts == n_econsts && h.n_constraints == n_constraints && h.reserved == kind &&
                load_code(ctx, h, (const char*)b.data() + sizeof h, out) == NX_OK) { cache_state().disk_hits++; return NX_OK; }
            // a damaged or foreign file: fall through to the compiler (and overwrite it)
        }
    }
    std::vector<char> code;
    NX_TRY(compile_parts(ctx, src, n_kernels, &code));
    cache_state().compiled++;
    BlobHeader h = {BLOB_MAGIC, BLOB_VERSION, n_kernels, n_cols, n_econsts, n_constraints, (uint64_t)code.size(), 0, kind};
    NX_TRY(load_code(ctx, h, code.data(), out));
    if (!path.empty()) {                                  // store: write beside, then rename — a concurrent reader sees the old file or the whole new one
        const std::vector<uint8_t> b = make_blob(*out);
        // one temporary per WRITER: several contexts of a process (thread ranks, concurrent sessions) compile the same source at the same time
        static std::atomic<uint64_t> writer{0};
        const std::string tmp = path + ".tmp" + std::to_string((long)getpid()) + "." + std::to_string((unsigned long long)writer.fetch_add(1));
        const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL, 0600);
        FILE* f = fd >= 0 ? fdopen(fd, "wb") : nullptr;
        if (!f && fd >= 0) close(fd);
        if (f) {
            const bool ok = fwrite(b.data(), 1, b.size(), f) == b.size();
            if (fclose(f) == 0 && ok && rename(tmp.c_str(), path.c_str()) == 0) cache_state().stored++;
            else (void)remove(tmp.c_str());
        }
    }
    return NX_OK;
}
}  // namespace nx

extern "C" {

int nx_air_cache_dir(const char* dir) {
    CacheState& s = cache_state();
    std::lock_guard<std::mutex> lk(s.mu);
    s.init = true; s.dir = dir && *dir ? trusted_cache_dir(dir) : std::string();
    return NX_OK;
}
int nx_air_cache_stats(uint64_t* n_compiled, uint64_t* n_disk_hits, uint64_t* n_stored) {
    CacheState& s = cache_state();
    if (n_compiled) *n_compiled = s.compiled.load();
    if (n_disk_hits) *n_disk_hits = s.disk_hits.load();
    if (n_stored) *n_stored = s.stored.load();
    return NX_OK;
}
int nx_air_kernel_save(const nx_air_kernel* k, uint8_t** blob, size_t* n_bytes) {
    if (!k || !blob || !n_bytes) return set_err(k ? k->ctx : nullptr, NX_ERR_ARG, "nx_air_kernel_save: NULL argument");
    const std::vector<uint8_t> b = make_blob(k);
    uint8_t* out = (uint8_t*)malloc(b.size());
    if (!out) return set_err(k->ctx, NX_ERR_OOM, "nx_air_kernel_save: malloc failed");
    memcpy(out, b.data(), b.size());
    *blob = out; *n_bytes = b.size();
    return NX_OK;
}
int nx_air_kernel_load(nx_ctx* ctx, const uint8_t* blob, size_t n_bytes, nx_air_kernel** out) {
    NX_GUARD(ctx);
    if (!ctx || !blob || !out) return set_err(ctx, NX_ERR_ARG, "nx_air_kernel_load: NULL argument");
    BlobHeader h;
    if (!blob_ok(blob, n_bytes, &h)) return set_err(ctx, NX_ERR_ARG, "nx_air_kernel_load: not a kernel blob of this library version (magic / size / checksum)");
    if (h.reserved != 0) return set_err(ctx, NX_ERR_ARG, "nx_air_kernel_load: the blob holds the fraction kernels of nx_logup_program (a cache-directory file), not constraint kernels");
    return load_code(ctx, h, (const char*)blob + sizeof h, out);
}

void nx_air_kernel_destroy(nx_air_kernel* k) {
    if (!k) return;
    NX_GUARD(k->ctx);
    (void)hipStreamSynchronize(k->ctx->stream);
    for (hipModule_t m : k->modules) (void)hipModuleUnload(m);
    delete k;
}

}  // extern "C"

namespace nx {
// Rows [row_begin, row_begin + n_rows) of the evaluation domain.  The generated kernel indexes columns and accumulators with the
// GLOBAL row: in a row-sharded prove the caller passes block pointers moved back by the block's first row (bias_rows) — columns read
// at a non-zero mask offset must be whole (their neighbour rows live in other blocks).
int air_eval_rows(nx_ctx* ctx, const nx_air_kernel* k, const uint32_t* const* d_cols, const uint32_t* econsts, const uint32_t* alpha_powers, const uint32_t* denom_inv,
                  uint32_t log_size, uint32_t log_eval, uint32_t* const* d_acc4, uint32_t row_begin, uint32_t n_rows) {
    if (!ctx || !k || !d_acc4 || (k->n_cols && !d_cols)) return set_err(ctx, NX_ERR_ARG, "nx_air_eval: NULL argument");
    if (k->kind != 0) return set_err(ctx, NX_ERR_ARG, "nx_air_eval: not a constraint kernel");
    if (log_size < 1 || log_eval <= log_size || log_eval > 30) return set_err(ctx, NX_ERR_ARG, "nx_air_eval: need 1 <= log_size < log_eval <= 30");
    if ((uint64_t)row_begin + n_rows > ((uint64_t)1 << log_eval)) return set_err(ctx, NX_ERR_ARG, "nx_air_eval: row block outside the evaluation domain");
    if (!n_rows) return NX_OK;
    const size_t b_cols = (size_t)k->n_cols * 8, b_ec = (size_t)k->n_econsts * 16, b_pw = (size_t)k->n_constraints * 16, b_den = (size_t)4 << (log_eval - log_size);
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t o_ec = al(b_cols), o_pw = o_ec + al(b_ec), o_den = o_pw + al(b_pw), total = o_den + al(b_den) + 16;
    // one stream-ordered copy through the context's pinned staging ring: no allocation, no synchronisation — a statement with
    // dozens of components enqueues their kernels back to back
    std::vector<uint8_t> host(total, 0);
    if (b_cols) memcpy(host.data(), d_cols, b_cols);
    if (b_ec) memcpy(host.data() + o_ec, econsts, b_ec);
    if (b_pw) memcpy(host.data() + o_pw, alpha_powers, b_pw);
    if (b_den) memcpy(host.data() + o_den, denom_inv, b_den);
    void* staged = nullptr;
    NX_TRY(stage(ctx, host.data(), total, &staged));
    const uint8_t* blob = (const uint8_t*)staged;
    const void* p_cols = blob; const void* p_ec = blob + o_ec; const void* p_pw = blob + o_pw; const void* p_den = blob + o_den;
    int ls = (int)log_size, le = (int)log_eval;
    uint32_t* a0 = d_acc4[0]; uint32_t* a1 = d_acc4[1]; uint32_t* a2 = d_acc4[2]; uint32_t* a3 = d_acc4[3];
    uint32_t rb = row_begin, re = row_begin + n_rows;
    void* args[] = {&p_cols, &p_ec, &p_pw, &p_den, &ls, &le, &a0, &a1, &a2, &a3, &rb, &re};
    for (hipFunction_t fn : k->fns) {       // the segments of a large program accumulate into the same rows, one launch after the other
        hipError_t er = hipModuleLaunchKernel(fn, (n_rows + 255) / 256, 1, 1, 256, 1, 1, 0, ctx->stream, args, nullptr);
        if (er != hipSuccess) return hip_fail(ctx, er, "nx_air_eval", __FILE__, __LINE__);
    }
    return NX_OK;
}
}  // namespace nx

extern "C" {

int nx_air_eval(nx_ctx* ctx, const nx_air_kernel* k, const uint32_t* const* d_cols, const uint32_t* econsts, const uint32_t* alpha_powers, const uint32_t* denom_inv,
                uint32_t log_size, uint32_t log_eval, uint32_t* const* d_acc4) {
    NX_GUARD(ctx);
    if (log_eval > 30) return set_err(ctx, NX_ERR_ARG, "nx_air_eval: need 1 <= log_size < log_eval <= 30");
    return air_eval_rows(ctx, k, d_cols, econsts, alpha_powers, denom_inv, log_size, log_eval, d_acc4, 0, 1u << log_eval);
}

}  // extern "C"

// ================================================================ logup fraction programs (nx_logup_program) ================
// The interaction trace of a component from the relation entries its recorded AIR declares (include/nexus_hip.h).  Same lowering
// as the constraint kernels — straight-line code per row, cut into segments at BATCH boundaries so that no kernel outgrows the
// instruction cache; a segment starts from the running sum the previous one left in its last column.  Inside a segment the
// denominators are inverted in groups of up to 8 with ONE M31 inverse per group (Montgomery's trick on the QM31 norms, as
// logup_cols_kernel does: logup.hip).  Rows are the trace domain's own (2^log_size evaluations, bit-reversed circle-domain
// order); a row offset is taken in natural coset order, where the trace step is +1.
namespace nx {

static const char* LOGUP_PRELUDE = R"SRC(
FI u32 m_sqr(u32 a) { return m_mul(a, a); }
FI u32 m_sqn(u32 a, int n) { for (int i = 0; i < n; i++) a = m_sqr(a); return a; }
FI u32 m_inv(u32 a) {
    const u32 t2 = m_mul(m_sqr(a), a), t4 = m_mul(m_sqn(t2, 2), t2), t8 = m_mul(m_sqn(t4, 4), t4), t16 = m_mul(m_sqn(t8, 8), t8);
    const u32 t24 = m_mul(m_sqn(t16, 8), t8), t28 = m_mul(m_sqn(t24, 4), t4), t29 = m_mul(m_sqr(t28), a);
    return m_mul(m_sqn(t29, 2), a);
}
// natural coset row <-> position in bit-reversed circle-domain order (reference prover/src/trace/utils_external.rs:24-39)
FI u32 pos_of_coset_row(u32 c, int log) { const u32 N = 1u << log; const u32 d = (c & 1) ? N - 1 - (c >> 1) : (c >> 1); return bitrev(d, log); }
FI u32 coset_row_of_pos(u32 p, int log) { const u32 N = 1u << log, d = bitrev(p, log); return d < N / 2 ? 2 * d : 2 * (N - 1 - d) + 1; }
FI u32 trace_row_offset(u32 r, int log, int off) { if (off == 0) return r; return pos_of_coset_row((coset_row_of_pos(r, log) + (u32)off) & ((1u << log) - 1), log); }
// the norm part of a QM31 inverse: x^-1 = (a, -b) D^-1, D = a^2 - (2 + i) b^2 in CM31, D^-1 = conj(D) / (D.a^2 + D.b^2)
)SRC"
#ifndef NX_Q_MUL_NAIVE
R"SRC(// D = x.a^2 - (2 + i) x.b^2 and x^-1 = (x.a, -x.b) conj(D) / |D|^2: every coordinate one lazy sum of at most four products (field.cuh q_norm_cm / q_conj_times)
FI void q_norm(Q x, u32& da, u32& db, u32& nrm) {
    const u32 a0d = m_add(x.a, x.a), b0d = m_add(x.c, x.c), b1d = m_add(x.d, x.d), sd = m_add(b0d, b1d), nb0 = P - x.c;
    da = acc_final(acc_mad(acc_mad(acc_mad((u64)x.a * x.a, x.b, P - x.b), b0d, nb0), sd, x.d));
    db = acc_final(acc_mad(acc_mad(acc_mad((u64)a0d * x.b, b0d, P - b1d), x.c, nb0), x.d, x.d));
    nrm = acc_final(acc_mad((u64)da * da, db, db));
}
FI Q q_frac_add(Q run, Q x, u32 da, u32 db, u32 s) {    // run + num / x with s = num / nrm (a base-field numerator rides on the norm's inverse; the sum's words ride in the accumulators)
    const u32 ia = m_mul(da, s), ib = m_mul(m_neg(db), s), na = P - ia, nb = P - ib;
    Q r = {acc_final(acc_mad(acc_mad((u64)run.a, x.a, ia), x.b, nb)), acc_final(acc_mad(acc_mad((u64)run.b, x.a, ib), x.b, ia)),
           acc_final(acc_mad(acc_mad((u64)run.c, x.c, na), x.d, ib)), acc_final(acc_mad(acc_mad((u64)run.d, x.c, nb), x.d, na))};
    return r;
}
FI Q q_inv_from(Q x, u32 da, u32 db, u32 ninv) {       // ninv = 1 / nrm
    const u32 ia = m_mul(da, ninv), ib = m_mul(m_neg(db), ninv), na = P - ia, nb = P - ib;
    Q r = {acc_final(acc_mad((u64)x.a * ia, x.b, nb)), acc_final(acc_mad((u64)x.a * ib, x.b, ia)), acc_final(acc_mad((u64)x.c * na, x.d, ib)), acc_final(acc_mad((u64)x.c * nb, x.d, na))};
    return r;
}
)SRC"
#else
R"SRC(FI void q_norm(Q x, u32& da, u32& db, u32& nrm) {
    u32 a0, a1, b0, b1; c_mul(x.a, x.b, x.a, x.b, a0, a1); c_mul(x.c, x.d, x.c, x.d, b0, b1);
    da = m_sub(a0, m_sub(m_add(b0, b0), b1)); db = m_sub(a1, m_add(m_add(b1, b1), b0));
    nrm = m_add(m_sqr(da), m_sqr(db));
}
FI Q q_inv_from(Q x, u32 da, u32 db, u32 ninv) {       // ninv = 1 / nrm
    const u32 ia = m_mul(da, ninv), ib = m_mul(m_neg(db), ninv);
    Q r; c_mul(x.a, x.b, ia, ib, r.a, r.b); c_mul(m_neg(x.c), m_neg(x.d), ia, ib, r.c, r.d);
    return r;
}
FI Q q_frac_add(Q run, Q x, u32 da, u32 db, u32 s) { return q_add(run, q_inv_from(x, da, db, s)); }
)SRC"
#endif
;

constexpr uint32_t LOGUP_PROG_GROUP = 8;

static std::string generate_logup_kernel(const nx_cinstr* prog, const std::vector<uint32_t>& keep, uint32_t n_regs, uint32_t first_col, const std::vector<char>& batch_end, const std::string& name) {
    std::string s;
    s += "extern \"C\" __attribute__((global)) __attribute__((amdgpu_flat_work_group_size(256, 256))) void " + name +
         "(const u32* const* __restrict__ cols, const u32* __restrict__ econst, u32* const* __restrict__ out, int log_size, u32 n) {\n"
         "  const u32 r = __builtin_amdgcn_workgroup_id_x() * 256 + __builtin_amdgcn_workitem_id_x();\n  if (r >= n) return;\n";
    std::vector<int> offs;
    for (uint32_t i : keep)
        if (prog[i].op == NX_C_LOAD || prog[i].op == NX_C_LOADE) { int o = (int)prog[i].b; if (std::find(offs.begin(), offs.end(), o) == offs.end()) offs.push_back(o); }
    auto off_name = [](int o) { return std::string("row_") + (o < 0 ? "m" : "p") + std::to_string(o < 0 ? -o : o); };
    for (int o : offs) s += "  const u32 " + off_name(o) + " = trace_row_offset(r, log_size, " + std::to_string(o) + ");\n";
    for (uint32_t k = 0; k < n_regs; k++) s += (k % 16 == 0 ? std::string("  u32 ") : std::string(", ")) + "r" + std::to_string(k) + ((k % 16 == 15 || k + 1 == n_regs) ? " = 0;\n" : " = 0");
    auto R = [](uint32_t i) { return "r" + std::to_string(i); };
    auto E = [&](uint32_t i) { return "Q{" + R(i) + ", " + R(i + 1) + ", " + R(i + 2) + ", " + R(i + 3) + "}"; };
    auto setE = [&](uint32_t d, const std::string& expr) {
        return "  { const Q t_ = " + expr + "; " + R(d) + " = t_.a; " + R(d + 1) + " = t_.b; " + R(d + 2) + " = t_.c; " + R(d + 3) + " = t_.d; }\n";
    };
    if (first_col == 0) s += "  Q run = {0, 0, 0, 0};\n";
    else {
        const std::string b = std::to_string(4 * (first_col - 1));
        s += "  Q run = {out[" + b + "][r], out[" + b + " + 1][r], out[" + b + " + 2][r], out[" + b + " + 3][r]};\n";
    }
    struct Pending { uint32_t instr; bool base; };
    std::vector<Pending> group;
    uint32_t uid = 0;
    auto resolve = [&]() {            // invert the group's denominators together, add the fractions in order, store finished columns
        if (group.empty()) return;
        const size_t G = group.size();
        auto v = [&](const char* p, size_t k) { return std::string(p) + std::to_string(uid) + "_" + std::to_string(k); };
        s += "  {\n";
        for (size_t k = 0; k < G; k++) {
            s += "    u32 " + v("da", k) + ", " + v("db", k) + ", " + v("nr", k) + "; q_norm(" + v("fd", k) + ", " + v("da", k) + ", " + v("db", k) + ", " + v("nr", k) + ");\n";
            s += "    const u32 " + v("nz", k) + " = " + v("nr", k) + " ? " + v("nr", k) + " : 1u, " + v("pp", k) + " = " + (k ? "m_mul(" + v("pp", k - 1) + ", " + v("nz", k) + ")" : v("nz", k)) + ";\n";
        }
        s += "    u32 inv = m_inv(" + v("pp", G - 1) + ");\n";
        for (size_t k = G; k-- > 0;) {
            s += "    const u32 " + v("ni", k) + " = " + v("nr", k) + " ? " + (k ? "m_mul(inv, " + v("pp", k - 1) + ")" : std::string("inv")) + " : 0u;" + (k ? " inv = m_mul(inv, " + v("nz", k) + ");" : "") + "\n";
        }
        for (size_t k = 0; k < G; k++) {
            const nx_cinstr& in = prog[group[k].instr];
            if (group[k].base) s += "    run = q_frac_add(run, " + v("fd", k) + ", " + v("da", k) + ", " + v("db", k) + ", m_mul(" + v("ni", k) + ", " + v("fn", k) + "));\n";
            else s += "    { const Q qi = q_inv_from(" + v("fd", k) + ", " + v("da", k) + ", " + v("db", k) + ", " + v("ni", k) + "); run = q_add(run, q_mul(" + v("fn", k) + ", qi)); }\n";
            if (batch_end[group[k].instr]) {
                const std::string b = std::to_string(4 * in.dst);
                s += "    out[" + b + "][r] = run.a; out[" + b + " + 1][r] = run.b; out[" + b + " + 2][r] = run.c; out[" + b + " + 3][r] = run.d;\n";
            }
        }
        s += "  }\n";
        group.clear(); uid++;
    };
    const DotFusion fu = find_dot_fusions(prog, keep, n_regs);
    LazyAcc lazy;
    for (size_t pos = 0; pos < keep.size(); pos++) {
        const uint32_t i = keep[pos];
        const nx_cinstr& in = prog[i];
        if (fu.skip[pos]) continue;
        if (fu.fused[pos]) { s += lazy.accumulate(prog[keep[pos - 1]], in); continue; }
        s += lazy.before(in);
        switch (in.op) {
        case NX_C_LOAD: s += "  " + R(in.dst) + " = G(cols[" + std::to_string(in.a) + "])[" + off_name((int)in.b) + "];\n"; break;
        case NX_C_CONST: s += "  " + R(in.dst) + " = " + std::to_string(in.a) + "u;\n"; break;
        case NX_C_ADD: s += "  " + R(in.dst) + " = m_add(" + R(in.a) + ", " + R(in.b) + ");\n"; break;
        case NX_C_SUB: s += "  " + R(in.dst) + " = m_sub(" + R(in.a) + ", " + R(in.b) + ");\n"; break;
        case NX_C_MUL: s += "  " + R(in.dst) + " = m_mul(" + R(in.a) + ", " + R(in.b) + ");\n"; break;
        case NX_C_NEG: s += "  " + R(in.dst) + " = m_neg(" + R(in.a) + ");\n"; break;
        case NX_C_CONSTE: { std::string b = std::to_string(4 * in.a); s += setE(in.dst, "Q{econst[" + b + "], econst[" + b + " + 1], econst[" + b + " + 2], econst[" + b + " + 3]}"); break; }
        case NX_C_ADDE: s += setE(in.dst, "q_add(" + E(in.a) + ", " + E(in.b) + ")"); break;
        case NX_C_SUBE: s += setE(in.dst, "q_sub(" + E(in.a) + ", " + E(in.b) + ")"); break;
        case NX_C_MULE: s += setE(in.dst, "q_mul(" + E(in.a) + ", " + E(in.b) + ")"); break;
        case NX_C_MULEB: s += setE(in.dst, "Q{m_mul(" + R(in.a) + ", " + R(in.b) + "), m_mul(" + R(in.a + 1) + ", " + R(in.b) + "), m_mul(" + R(in.a + 2) + ", " + R(in.b) + "), m_mul(" + R(in.a + 3) + ", " + R(in.b) + ")}"); break;
        case NX_C_ADDEB: s += setE(in.dst, "Q{m_add(" + R(in.a) + ", " + R(in.b) + "), " + R(in.a + 1) + ", " + R(in.a + 2) + ", " + R(in.a + 3) + "}"); break;
        case NX_C_LOADE: {
            std::string o = off_name((int)in.b);
            s += setE(in.dst, "Q{G(cols[" + std::to_string(in.a) + "])[" + o + "], G(cols[" + std::to_string(in.a + 1) + "])[" + o + "], G(cols[" + std::to_string(in.a + 2) + "])[" + o + "], G(cols[" +
                                  std::to_string(in.a + 3) + "])[" + o + "]}");
            break;
        }
        case NX_C_FRAC: case NX_C_FRACB: {
            // the registers may be reused by later instructions: the fraction is copied out until its group is resolved
            const std::string k = std::to_string(uid) + "_" + std::to_string(group.size());
            s += "  const Q fd" + k + " = " + E(in.b) + "; ";
            s += in.op == NX_C_FRACB ? "const u32 fn" + k + " = " + R(in.a) + ";\n" : "const Q fn" + k + " = " + E(in.a) + ";\n";
            group.push_back({i, in.op == NX_C_FRACB});
            if (group.size() == LOGUP_PROG_GROUP) resolve();
            break;
        }
        default: break;
        }
    }
    resolve();
    s += "}\n";
    return s;
}

int validate_logup_program(nx_ctx* ctx, const nx_cinstr* program, uint32_t n_instr, uint32_t n_regs, uint32_t n_cols, uint32_t n_econsts, uint32_t n_logup_cols) {
    if (n_instr && !program) return set_err(ctx, NX_ERR_ARG, "logup program: NULL program");
    if (n_regs == 0 || n_regs > 4096) return set_err(ctx, NX_ERR_ARG, "logup program: register count out of range");
    std::vector<nx_cinstr> compute;
    uint32_t next_batch = 0; bool any = false;
    for (uint32_t i = 0; i < n_instr; i++) {
        const nx_cinstr& in = program[i];
        if (in.op == NX_C_CONSTRAINT_B || in.op == NX_C_CONSTRAINT_E) return set_err(ctx, NX_ERR_ARG, "logup program: a constraint instruction (the fractions are the roots of this program)");
        if (in.op == NX_C_FRAC || in.op == NX_C_FRACB) {
            const uint32_t wa = in.op == NX_C_FRAC ? 4 : 1;
            if (!(in.a <= n_regs && wa <= n_regs - in.a && in.b <= n_regs && 4 <= n_regs - in.b)) return set_err(ctx, NX_ERR_ARG, "logup program: malformed instruction " + std::to_string(i));
            if (in.dst >= n_logup_cols) return set_err(ctx, NX_ERR_ARG, "logup program: fraction of a logup column the component does not have");
            if (any ? (in.dst != next_batch && in.dst != next_batch + 1) : in.dst != 0)
                return set_err(ctx, NX_ERR_ARG, "logup program: fractions must come in non-decreasing batch order and every batch 0 .. last must hold one (finalize_logup_batched)");
            next_batch = in.dst; any = true;
            continue;
        }
        compute.push_back(in);
    }
    if (n_logup_cols && (!any || next_batch + 1 != n_logup_cols)) return set_err(ctx, NX_ERR_ARG, "logup program: a logup column without fractions");
    uint32_t n_c = 0;
    return validate_air_program(ctx, compute.data(), (uint32_t)compute.size(), n_regs, n_cols, n_econsts, &n_c);
}

static std::string generate_logup_source(const nx_ctx* ctx, const nx_cinstr* prog, uint32_t n_instr, uint32_t n_regs, uint32_t* n_kernels) {
    const ProgDeps pd = program_deps(prog, n_instr, n_regs);
    std::string s = std::string(AIR_PRELUDE) + LOGUP_PRELUDE;
    std::vector<char> batch_end(n_instr, 0);
    { int last = -1; for (uint32_t i = 0; i < n_instr; i++) if (prog[i].op == NX_C_FRAC || prog[i].op == NX_C_FRACB) { if (last >= 0 && prog[last].dst != prog[i].dst) batch_end[last] = 1; last = (int)i; } if (last >= 0) batch_end[last] = 1; }
    uint32_t n_seg = 0, cost = 0, first_col = 0;
    std::vector<char> in_seg(n_instr, 0);
    auto add_slice = [&](uint32_t root) {
        uint32_t added = 0;
        std::vector<uint32_t> st{root};
        while (!st.empty()) {
            uint32_t i = st.back(); st.pop_back();
            if (in_seg[i]) continue;
            in_seg[i] = 1; added += prog[i].op == NX_C_FRAC || prog[i].op == NX_C_FRACB ? 120 : instr_cost(prog[i].op);
            for (uint32_t d : pd.deps[i]) if (!in_seg[d]) st.push_back(d);
        }
        return added;
    };
    bool any = false;
    auto flush = [&]() {
        std::vector<uint32_t> keep;
        for (uint32_t i = 0; i < n_instr; i++) if (in_seg[i]) keep.push_back(i);
        s += generate_logup_kernel(prog, keep, n_regs, first_col, batch_end, n_seg == 0 ? std::string("air_kernel") : "air_kernel_" + std::to_string(n_seg));
        n_seg++;
        std::fill(in_seg.begin(), in_seg.end(), 0); cost = 0; any = false;
    };
    const uint32_t budget = segment_budget(ctx);
    for (uint32_t i = 0; i < n_instr; i++) {
        if (prog[i].op != NX_C_FRAC && prog[i].op != NX_C_FRACB) continue;
        if (!any) first_col = prog[i].dst;
        cost += add_slice(i); any = true;
        if (batch_end[i] && cost >= budget) flush();            // cut only where a column is complete: the next segment reads it back
    }
    if (any || n_seg == 0) flush();
    *n_kernels = n_seg;
    return s;
}

struct LogupKernelCache { std::mutex mu; std::map<std::pair<nx_ctx*, std::string>, nx_air_kernel*> map; };
static LogupKernelCache& logup_kernel_cache() { static LogupKernelCache c; return c; }
void logup_kernels_release(nx_ctx* ctx) {
    LogupKernelCache& kc = logup_kernel_cache();
    std::lock_guard<std::mutex> lk(kc.mu);
    for (auto it = kc.map.begin(); it != kc.map.end();) { if (it->first.first == ctx) { nx_air_kernel_destroy(it->second); it = kc.map.erase(it); } else ++it; }
}

}  // namespace nx

extern "C" int nx_logup_program(nx_ctx* ctx, const nx_cinstr* program, uint32_t n_instr, uint32_t n_regs, const uint32_t* const* d_cols, uint32_t n_cols, const uint32_t* econsts,
                                uint32_t n_econsts, uint32_t log_size, uint32_t n_logup_cols, uint32_t* const* d_out, char** h_source_out) {
    NX_GUARD(ctx);
    if (!program || (n_econsts && !econsts)) return set_err(ctx, NX_ERR_ARG, "nx_logup_program: NULL argument");
    if (log_size < 1 || log_size > 30) return set_err(ctx, NX_ERR_ARG, "nx_logup_program: log_size out of range");
    { nxhip::HostSpan hs("lp.validate"); NX_TRY(validate_logup_program(ctx, program, n_instr, n_regs, n_cols, n_econsts, n_logup_cols)); }
    nxhip::HostSpan hs_key("lp.key+lookup");
    // The kernels are cached per context by the PROGRAM (its bytes and shape, the segment budget): the generated source of a wide component
    // is megabytes of text, and building it only to look the kernels up cost a keccak-shaped prove 5 ms of GPU idle time per round
    // component (profiles/r06_keccak_tuples_sequence.txt, before).  The source is generated on a miss (and for h_source_out).
    std::string key((const char*)program, (size_t)n_instr * sizeof(nx_cinstr));
    key += "|" + std::to_string(n_regs) + "|" + std::to_string(n_cols) + "|" + std::to_string(n_econsts) + "|" + std::to_string(n_logup_cols) + "|" + std::to_string(segment_budget(ctx));
    const nx_air_kernel* k = nullptr;
    if (ctx) { LogupKernelCache& kc = logup_kernel_cache(); std::lock_guard<std::mutex> lk(kc.mu); auto it = kc.map.find({ctx, key}); if (it != kc.map.end()) k = it->second; }
    hs_key.stop();
    nxhip::HostSpan hs_rest("lp.checks+stage+launch");
    uint32_t n_kernels = 1;
    std::string src;
    if (!k || h_source_out) src = generate_logup_source(ctx, program, n_instr, n_regs, &n_kernels);
    if (h_source_out) { *h_source_out = (char*)malloc(src.size() + 1); if (*h_source_out) std::copy(src.c_str(), src.c_str() + src.size() + 1, *h_source_out); }
    if (!ctx || !d_out) { if (h_source_out) return NX_OK; return set_err(ctx, NX_ERR_ARG, "nx_logup_program: a context and output columns are needed to run"); }
    if (n_logup_cols == 0) return NX_OK;
    if (n_cols && !d_cols) return set_err(ctx, NX_ERR_ARG, "nx_logup_program: NULL column table");
    for (uint32_t i = 0; i < n_instr; i++) {                  // a loaded column must be there
        const nx_cinstr& in = program[i];
        if (in.op == NX_C_LOAD && !d_cols[in.a]) return set_err(ctx, NX_ERR_ARG, "nx_logup_program: the program loads a column that was passed as NULL");
        if (in.op == NX_C_LOADE) for (uint32_t k = 0; k < 4; k++) if (!d_cols[in.a + k]) return set_err(ctx, NX_ERR_ARG, "nx_logup_program: the program loads a column that was passed as NULL");
    }
    for (size_t k = 0; k < 4 * (size_t)n_logup_cols; k++) if (!d_out[k]) return set_err(ctx, NX_ERR_ARG, "nx_logup_program: NULL output column");
    if (!k) {
        LogupKernelCache& kc = logup_kernel_cache();
        nx_air_kernel* nk = nullptr;
        NX_TRY(compile_source(ctx, src, n_kernels, n_cols, n_econsts, n_logup_cols, &nk, 1));      // outside the lock: entries are per context
        std::lock_guard<std::mutex> lk(kc.mu);
        kc.map.insert({{ctx, key}, nk});
        k = nk;
    }
    const size_t b_cols = (size_t)n_cols * 8, b_ec = (size_t)n_econsts * 16, b_out = (size_t)n_logup_cols * 32;
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t o_ec = al(b_cols), o_out = o_ec + al(b_ec), total = o_out + al(b_out) + 16;
    std::vector<uint8_t> host(total, 0);
    if (b_cols) memcpy(host.data(), d_cols, b_cols);
    if (b_ec) memcpy(host.data() + o_ec, econsts, b_ec);
    memcpy(host.data() + o_out, d_out, b_out);
    uint8_t* blob = nullptr;
    NX_TRY(dev_alloc(ctx, total, (void**)&blob));
    hipError_t e = hipSuccess;
    for (size_t off = 0; off < total && e == hipSuccess; off += (size_t)4 << 20) {
        const size_t nb = std::min(total - off, (size_t)4 << 20);
        void* st = nullptr;
        int rc = stage(ctx, host.data() + off, nb, &st);
        if (rc != NX_OK) { dev_free(ctx, blob); return rc; }
        e = hipMemcpyAsync(blob + off, st, nb, hipMemcpyDeviceToDevice, ctx->stream);
    }
    const void* p_cols = blob; const void* p_ec = blob + o_ec; const void* p_out = blob + o_out;
    int ls = (int)log_size; uint32_t n = 1u << log_size;
    void* args[] = {&p_cols, &p_ec, &p_out, &ls, &n};
    for (hipFunction_t fn : k->fns) {
        if (e != hipSuccess) break;
        e = hipModuleLaunchKernel(fn, (n + 255) / 256, 1, 1, 256, 1, 1, 0, ctx->stream, args, nullptr);
    }
    dev_free(ctx, blob);
    if (e != hipSuccess) return hip_fail(ctx, e, "nx_logup_program", __FILE__, __LINE__);
    return NX_OK;
}
