// Circle FFT / iFFT over M31 for gfx950 (K1-K4 of SURVEY.md §8(a)).
//
// Replaces Stwo `PolyOps::{precompute_twiddles, interpolate_columns, evaluate_polynomials}` and
// `ColumnOps::bit_reverse_column`, reached from reference prover/src/machine.rs:186,209-263 and
// prover/src/trace/utils.rs:94-106.
//
// Design (MI355X-first, not a port of Stwo's u32x16 SIMD loops):
//  * A transform of 2^n points is split into 2-3 *passes*; a pass owns a contiguous range of
//    butterfly layers [lo, hi) and processes tiles that hold every index bit of that range plus B
//    low bits, so each global access is a >=2^B*4-byte contiguous run (coalesced 64-128 B segments).
//  * Inside a pass the tile lives in LDS (padded 1 word per 16 against bank conflicts); each lane
//    pulls 16 values into VGPRs and runs 4 butterfly layers in registers (radix-16) between LDS
//    round trips, so LDS traffic is ~1/4 of a layer-per-sync schedule.
//  * Mersenne reduction stays in registers (field.cuh); all stores are canonical.
//  * Columns are processed in small batches through *all* passes before the next batch starts, so a
//    batch's inter-pass traffic stays resident in the 256 MB Infinity Cache instead of going to HBM.
//  * The zero-extension of the LDE ("extend") is folded into the first evaluate pass: source words
//    at index >= 2^log_in read as zero, nothing is materialised.
#include "internal.h"
#include <mutex>
#include <atomic>
#include <stdlib.h>
#include <algorithm>

namespace nx {

// ------------------------------------------------------------------ twiddles (K2) -----------
// pt_from_index as a sum of table entries (field.cuh GenTable: no doubling chain per element)
__global__ void twiddle_kernel(u32* tw, u32* itw, int h, GenTable gen) {
    u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    u32 total = 1u << h;
    if (p >= total) return;
    u32 q = total - p;
    u32 val;
    if (q == 1) {
        val = 1;  // padding element
    } else {
        int m = 31 - __clz(q - 1);          // this layer holds 2^m twiddles
        int l = h - 1 - m;                  // doubling depth
        u32 k = (2u << m) - q;              // index within the layer
        u32 init = (1u << (31 - h - 2)) << l, step = (1u << (31 - h)) << l;
        u32 idx = (init + step * bitrev(k, m)) & 0x7fffffffu;
        val = pt_from_index_tbl(gen, idx).x;
    }
    tw[p] = val;
    itw[p] = m_inv(val);
}

// ------------------------------------------------------------------ FFT passes (K3/K4) ------
struct FftPass {
    ColSet src, dst;
    const u32* tw;   // forward or inverse twiddle buffer, 2^tw_log words
    u32 tw_log;
    int n;           // log size of the transform
    int log_in;      // source holds 2^log_in words; higher indices read as zero
    int lo, hi;      // butterfly layers [lo, hi)
    int B;           // low contiguous index bits carried by a tile (0 when lo == 0)
    u32 scale;       // multiply outputs by this on store (0: no scaling)
    u32 n_cols;
};

__device__ __forceinline__ u32 lds_pad(u32 t) { return t + (t >> 4); }

// butterfly / inverse butterfly with the twiddle's sign folded into the output wiring (no negation op)
template <bool INV>
__device__ __forceinline__ void bfly(u32& x0, u32& x1, u32 t, bool neg) {
    if (INV) {
        u32 s = m_add(x0, x1);
        u32 d = neg ? m_sub(x1, x0) : m_sub(x0, x1);
        x0 = s; x1 = m_mul(d, t);
    } else {
        u32 m = m_mul(x1, t);
        u32 a = m_add(x0, m), b = m_sub(x0, m);
        x0 = neg ? b : a; x1 = neg ? a : b;
    }
}

template <int CNT>
__device__ __forceinline__ void load_tw(const u32* __restrict__ p, u32* dst) {
    if (CNT == 8) {
        uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 4);
        dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w; dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
    } else if (CNT == 4) {
        uint4 a = *reinterpret_cast<const uint4*>(p);
        dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w;
    } else if (CNT == 2) {
        uint2 a = *reinterpret_cast<const uint2*>(p);
        dst[0] = a.x; dst[1] = a.y;
    } else {
        dst[0] = p[0];
    }
}

// loads the twiddles of layers Q..R-1 of a lane's block (compile-time recursion: counts are template constants)
template <int R, int Q, bool CIRCLE>
__device__ __forceinline__ void load_round_tw(const u32* const __restrict__* twl, u32 g0, int l0, u32* tw) {
    if constexpr (Q < R) {
        if constexpr (!(CIRCLE && Q == 0))
            load_tw<(1 << (R - 1 - Q))>(twl[Q] + (g0 >> (l0 + Q + 1)), tw + ((1 << R) - (1 << (R - Q))));
        load_round_tw<R, Q + 1, CIRCLE>(twl, g0, l0, tw);
    }
}

// One LDS round trip: every lane owns 2^R tile elements that differ in R consecutive layer bits and runs those
// R butterfly layers in registers.  Twiddles of layer q are 2^(R-1-q) consecutive, aligned words -> one vector
// load per layer.  CIRCLE: the round's first layer is the circle layer, whose twiddles are derived from the
// next (first line) layer's values as [x, y] -> [y, -y, -x, x].
template <int R, bool INV, bool CIRCLE>
__device__ __forceinline__ void radix_round(u32* lds, const FftPass& a, int j, u32 tile_base, int s) {
    static_assert(!CIRCLE || R >= 3, "circle rounds need the pair (x, y) of the first line layer");
    const int bp = a.B + j;  // 0 or >= 4 (rounds start at multiples of 4; B is 0 or >= 5)
    const u32 estride = bp ? ((1u << bp) + ((1u << bp) >> 4)) : 1u;  // padded distance between a lane's elements
    const u32 n_items = (1u << s) >> R;
    const u32 maskB = (1u << a.B) - 1;
    const int l0 = a.lo + j;
    const u32* __restrict__ twl[R];
#pragma unroll
    for (int q = 0; q < R; q++) twl[q] = a.tw + ((1u << a.tw_log) - (1u << (a.n - (l0 + q))));
    for (u32 w = threadIdx.x; w < n_items; w += blockDim.x) {
        const u32 wl = w & ((1u << bp) - 1), wh = w >> bp;
        const u32 t0 = (wh << (bp + R)) | wl;
        const u32 p0 = lds_pad(t0);
        u32 v[1 << R];
#pragma unroll
        for (int e = 0; e < (1 << R); e++) v[e] = lds[p0 + e * estride];
        const u32 g0 = tile_base + ((t0 >> a.B) << a.lo) + (t0 & maskB);
        u32 tw[(1 << R)];  // layer q at offset 2^R - 2^(R-q), 2^(R-1-q) entries
        load_round_tw<R, 0, CIRCLE>(twl, g0, l0, tw);
#pragma unroll
        for (int qq = 0; qq < R; qq++) {
            const int q = INV ? qq : R - 1 - qq;
#pragma unroll
            for (int e = 0; e < (1 << R); e++) {
                if (e & (1 << q)) continue;
                const int h = e >> (q + 1);  // pair index within the lane's block
                if (CIRCLE && q == 0) {
                    const int c = h >> 2, sel = h & 3;
                    const u32 t = tw[((1 << R) - (1 << (R - 1))) + 2 * c + ((sel & 2) ? 0 : 1)];  // x for sel 2,3; y for sel 0,1
                    bfly<INV>(v[e], v[e | 1], t, sel == 1 || sel == 2);
                } else {
                    bfly<INV>(v[e], v[e | (1 << q)], tw[((1 << R) - (1 << (R - q))) + h], false);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < (1 << R); e++) lds[p0 + e * estride] = v[e];
    }
    __syncthreads();
}

template <bool INV, bool FIRST>
__device__ __forceinline__ void remainder_round(u32* lds, const FftPass& a, int rem, int j, bool circle, u32 tile_base, int s) {
    if (rem == 3) { if (FIRST && circle) radix_round<3, INV, true>(lds, a, j, tile_base, s); else radix_round<3, INV, false>(lds, a, j, tile_base, s); }
    else if (rem == 2) radix_round<2, INV, false>(lds, a, j, tile_base, s);
    else if (rem == 1) radix_round<1, INV, false>(lds, a, j, tile_base, s);
}

// FIRST: the pass that holds layers [0, s0) as a contiguous tile (lo == 0, B == 0) and therefore the circle layer.
template <bool INV, bool FIRST>
__global__ void fft_pass_kernel(FftPass a) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int K = a.hi - a.lo, s = a.B + K;
    const u32 S = 1u << s;
    const u32 col = blockIdx.x % a.n_cols, tile = blockIdx.x / a.n_cols;
    const int lb = a.lo - a.B;  // bits of the low-block part of the tile id
    const u32 lowblock = tile & ((1u << lb) - 1), high = tile >> lb;
    const u32 tile_base = (high << a.hi) | (lowblock << a.B);
    const u32 maskB = (1u << a.B) - 1;
    const u32* __restrict__ src = a.src.col(col);
    u32* __restrict__ dst = a.dst.col(col);
    const u32 in_limit = 1u << a.log_in;

    // ---- global -> LDS, 16-byte accesses (a tile row of 2^B words is contiguous; B>=2 or lo==0) ----
    for (u32 t = threadIdx.x * 4; t < S; t += blockDim.x * 4) {
        u32 g = tile_base + ((t >> a.B) << a.lo) + (t & maskB);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g + 3 < in_limit) v = *reinterpret_cast<const uint4*>(src + g);
        else if (g < in_limit) { v.x = src[g]; if (g + 1 < in_limit) v.y = src[g + 1]; if (g + 2 < in_limit) v.z = src[g + 2]; }
        u32 p = lds_pad(t);
        lds[p] = v.x; lds[p + 1] = v.y; lds[p + 2] = v.z; lds[p + 3] = v.w;
    }
    __syncthreads();

    // ---- butterfly rounds: 4 layers per LDS round trip, rounds start at layer offsets that are multiples of 4 ----
    const int nfull = K >> 2, rem = K & 3;
    if (INV) {
        for (int i = 0; i < nfull; i++) {
            if (FIRST && i == 0) radix_round<4, true, true>(lds, a, 0, tile_base, s);
            else radix_round<4, true, false>(lds, a, 4 * i, tile_base, s);
        }
        if (rem) remainder_round<true, FIRST>(lds, a, rem, 4 * nfull, nfull == 0, tile_base, s);
    } else {
        if (rem) remainder_round<false, FIRST>(lds, a, rem, 4 * nfull, nfull == 0, tile_base, s);
        for (int i = nfull - 1; i >= 0; i--) {
            if (FIRST && i == 0) radix_round<4, false, true>(lds, a, 0, tile_base, s);
            else radix_round<4, false, false>(lds, a, 4 * i, tile_base, s);
        }
    }

    // ---- LDS -> global ----
    for (u32 t = threadIdx.x * 4; t < S; t += blockDim.x * 4) {
        u32 g = tile_base + ((t >> a.B) << a.lo) + (t & maskB);
        u32 p = lds_pad(t);
        uint4 v = make_uint4(lds[p], lds[p + 1], lds[p + 2], lds[p + 3]);
        if (a.scale) { v.x = m_mul(v.x, a.scale); v.y = m_mul(v.y, a.scale); v.z = m_mul(v.z, a.scale); v.w = m_mul(v.w, a.scale); }
        *reinterpret_cast<uint4*>(dst + g) = v;
    }
}

// n = 1, 2 (Stwo special-cases them too: prover/backend/cpu/circle.rs interpolate/evaluate)
__global__ void fft_tiny_kernel(ColSet src, ColSet dst, u32 n_cols, int n, int log_in, bool inv) {
    u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cols) return;
    const u32* s = src.col(c); u32* d = dst.col(c);
    u32 v[4];
    for (int i = 0; i < (1 << n); i++) v[i] = i < (1 << log_in) ? s[i] : 0;
    Pt p0 = pt_from_index(half_odds_index(n - 1, 0));
    if (n == 1) {
        if (inv) { u32 a = v[0], b = v[1]; u32 hi = m_inv(2); v[0] = m_mul(m_add(a, b), hi); v[1] = m_mul(m_mul(m_sub(a, b), m_inv(p0.y)), hi); }
        else { u32 m = m_mul(v[1], p0.y); u32 a = v[0]; v[0] = m_add(a, m); v[1] = m_sub(a, m); }
    } else {
        if (inv) {
            u32 xi = m_inv(p0.x), yi = m_inv(p0.y), qi = m_inv(4);
            u32 a0 = m_add(v[0], v[1]), a1 = m_mul(m_sub(v[0], v[1]), yi);
            u32 a2 = m_add(v[2], v[3]), a3 = m_mul(m_sub(v[2], v[3]), m_neg(yi));
            v[0] = m_mul(m_add(a0, a2), qi); v[2] = m_mul(m_mul(m_sub(a0, a2), xi), qi);
            v[1] = m_mul(m_add(a1, a3), qi); v[3] = m_mul(m_mul(m_sub(a1, a3), xi), qi);
        } else {
            u32 m = m_mul(v[2], p0.x); u32 b0 = m_add(v[0], m), b2 = m_sub(v[0], m);
            m = m_mul(v[3], p0.x); u32 b1 = m_add(v[1], m), b3 = m_sub(v[1], m);
            m = m_mul(b1, p0.y); v[0] = m_add(b0, m); v[1] = m_sub(b0, m);
            m = m_mul(b3, m_neg(p0.y)); v[2] = m_add(b2, m); v[3] = m_sub(b2, m);
        }
    }
    for (int i = 0; i < (1 << n); i++) d[i] = v[i];
}

// ---- planning ----
// The pass shapes of the transforms below 2^13 points (everything larger is fft13.hip): one LDS-resident pass of up to 2^13 rows,
// later passes with 5 contiguity bits, 256-lane blocks.  (Rounds 1-2 tuned these through NX_FFT_SMAX / _B / _THREADS and kept the whole
// pre-fft13 path behind NX_FFT_LEGACY; the measurements are settled — DESIGN.md section 6 — and the knobs are gone.)
struct FftTune { int smax, bmax, threads, legacy; };
static constexpr FftTune g_tune = {13, 5, 256, 0};
static inline void tune_init() {}

struct PassPlan { int lo, hi, B; };
static int rounds_of(int k) { return (k + 3) / 4; }
// Layer ranges in ascending order; the first holds layers [0, s0) as a contiguous tile, later passes
// hold k <= smax - bmax layers plus bmax low bits.  Fewest passes first (each pass is one HBM/MALL
// round trip), then fewest LDS rounds (4 layers per round), then the largest first tile.
static std::vector<PassPlan> plan_passes(int n) {
    tune_init();
    std::vector<PassPlan> p;
    if (n <= g_tune.smax) { p.push_back({0, n, 0}); return p; }
    int kmax = g_tune.smax - g_tune.bmax;
    int extra = (n - g_tune.smax + kmax - 1) / kmax;
    // choose s0 in [max(bmax, n - extra*kmax), smax]; split the rest as evenly as the round count allows
    int best_s0 = -1, best_rounds = 1 << 30;
    std::vector<int> best_ks;
    for (int s0 = g_tune.smax; s0 >= std::max(g_tune.bmax, n - extra * kmax); s0--) {
        int rest = n - s0;
        // greedy: fill passes with multiples of 4 where possible
        std::vector<int> ks(extra, 0);
        int left = rest;
        for (int i = 0; i < extra; i++) {
            int remaining_passes = extra - i - 1;
            int k = std::min(kmax, left - remaining_passes);          // leave >= 1 layer for each later pass
            int need_min = left - remaining_passes * kmax;            // later passes cannot absorb more
            int k4 = (k / 4) * 4;
            if (k4 >= std::max(1, need_min) && k4 > 0) k = k4;
            ks[i] = k; left -= k;
        }
        if (left != 0) continue;
        int r = rounds_of(s0);
        for (int k : ks) r += rounds_of(k);
        if (r < best_rounds) { best_rounds = r; best_s0 = s0; best_ks = ks; }
    }
    p.push_back({0, best_s0, 0});
    int lo = best_s0;
    for (int k : best_ks) {
        // tile >= 2^11 words where the layer count is small: longer contiguous runs, fewer tiny blocks
        int B = std::min(lo, std::max(g_tune.bmax, std::min(11, g_tune.smax) - k));
        p.push_back({lo, lo + k, B}); lo += k;
    }
    return p;
}

static int launch_pass(nx_ctx* ctx, bool inv, const FftPass& a) {
    int s = a.B + (a.hi - a.lo);
    u32 tiles = 1u << (a.n - s);
    size_t lds_bytes = (((size_t)1 << s) + ((size_t)1 << s >> 4) + 4) * 4;
    int threads = std::min<int>(g_tune.threads, std::max(64, (1 << s) / 4));
    dim3 grid(tiles * a.n_cols), block(threads);
    const bool first = a.lo == 0;
    if (lds_bytes > 48 * 1024) {
        static std::atomic<uint64_t> attr_set{0};   // one bit per device: function attributes are per device
        if (!(attr_set.load() & (1ull << (ctx->device & 63)))) {
            NX_HIP(ctx, hipFuncSetAttribute((const void*)fft_pass_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            NX_HIP(ctx, hipFuncSetAttribute((const void*)fft_pass_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            NX_HIP(ctx, hipFuncSetAttribute((const void*)fft_pass_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            NX_HIP(ctx, hipFuncSetAttribute((const void*)fft_pass_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set.fetch_or(1ull << (ctx->device & 63));
        }
    }
    if (inv && first) hipLaunchKernelGGL((fft_pass_kernel<true, true>), grid, block, lds_bytes, ctx->cur, a);
    else if (inv) hipLaunchKernelGGL((fft_pass_kernel<true, false>), grid, block, lds_bytes, ctx->cur, a);
    else if (first) hipLaunchKernelGGL((fft_pass_kernel<false, true>), grid, block, lds_bytes, ctx->cur, a);
    else hipLaunchKernelGGL((fft_pass_kernel<false, false>), grid, block, lds_bytes, ctx->cur, a);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

static ColSet sub_colset(const ColSet& c, u32 first) {
    ColSet r = c;
    if (c.table) r.table = c.table + first; else r.base = c.base + (uint64_t)first * c.stride;
    return r;
}

static int check_tw(nx_ctx* ctx, const nx_twiddles* tw, int n) {
    if (!tw) return set_err(ctx, NX_ERR_ARG, "twiddles are NULL");
    if (n < 1 || n - 1 > (int)tw->log_half) return set_err(ctx, NX_ERR_ARG, "domain larger than the twiddle tree (or log_size < 1)");
    return NX_OK;
}

static int interpolate_cols(nx_ctx* ctx, const nx_twiddles* tw, ColSet cols, u32 n_cols, int n) {
    if (n >= 13 && !g_tune.legacy) return fft13_interpolate(ctx, tw, cols, n_cols, n);
    if (n <= 2) {
        hipLaunchKernelGGL(fft_tiny_kernel, dim3((n_cols + 63) / 64), dim3(64), 0, ctx->cur, cols, cols, n_cols, n, n, true);
        NX_LAUNCH_CHECK(ctx);
        return NX_OK;
    }
    std::vector<PassPlan> plan = plan_passes(n);
    u32 scale = m_inv(1u << n);
    for (size_t i = 0; i < plan.size(); i++) {
        FftPass a; a.src = cols; a.dst = cols; a.tw = tw->d_itw; a.tw_log = tw->log_half; a.n = n; a.log_in = n;
        a.lo = plan[i].lo; a.hi = plan[i].hi; a.B = plan[i].B; a.scale = i + 1 == plan.size() ? scale : 0; a.n_cols = n_cols;
        NX_TRY(launch_pass(ctx, true, a));
    }
    return NX_OK;
}

static int evaluate_cols(nx_ctx* ctx, const nx_twiddles* tw, ColSet polys, u32 n_cols, int log_in, int n, ColSet out) {
    if (log_in >= 13 && !g_tune.legacy) return fft13_evaluate(ctx, tw, polys, n_cols, log_in, n, out);
    if (n <= 2) {
        hipLaunchKernelGGL(fft_tiny_kernel, dim3((n_cols + 63) / 64), dim3(64), 0, ctx->cur, polys, out, n_cols, n, log_in, false);
        NX_LAUNCH_CHECK(ctx);
        return NX_OK;
    }
    std::vector<PassPlan> plan = plan_passes(n);
    for (size_t k = plan.size(); k-- > 0;) {
        bool first = k + 1 == plan.size();
        FftPass a; a.src = first ? polys : out; a.dst = out; a.tw = tw->d_tw; a.tw_log = tw->log_half; a.n = n;
        a.log_in = first ? log_in : n; a.lo = plan[k].lo; a.hi = plan[k].hi; a.B = plan[k].B; a.scale = 0; a.n_cols = n_cols;
        NX_TRY(launch_pass(ctx, false, a));
    }
    return NX_OK;
}

// Columns per launch: g_tune.batch_cols (2) is sized for 2^22-row columns — four of them in flight fill the Infinity Cache.  Smaller
// columns get proportionally more per launch (the same bytes in flight): at 2^18 rows a 2-column launch is ~10 us of work behind
// ~5 us of launch latency, and a 438-column tree needs 657 of them.
// Larger columns get fewer: at 2^23 / 2^24 rows one column per launch on the two streams measured 6 % faster than two
// (profiles/r06_big_batch.txt) — four 2^24-row columns in flight are 3x the Infinity Cache.
static u32 batch_for(const nx_ctx* ctx, uint32_t log_size) {
    if (log_size > 22) return (u32)std::max(1, ctx->opt.fft_batch_cols >> std::min<uint32_t>(log_size - 22, 8));
    const int shift = log_size < 22 ? (int)(22 - log_size) : 0;
    return (u32)std::min<uint64_t>((uint64_t)ctx->opt.fft_batch_cols << std::min(shift, 8), 256);
}

int fft_interpolate(nx_ctx* ctx, const nx_twiddles* tw, ColSet cols, uint32_t n_cols, uint32_t log_size) {
    NX_TRY(check_tw(ctx, tw, (int)log_size));
    tune_init();
    KTimer t(ctx, NX_T_LDE, (uint64_t)n_cols * 8ull << log_size);
    const u32 bc = batch_for(ctx, log_size);
    const int ns = n_cols > bc ? ctx->opt.fft_streams : 1;
    NX_TRY(streams_fork(ctx, ns));
    int rc = NX_OK;
    for (u32 c0 = 0, bi = 0; c0 < n_cols && rc == NX_OK; c0 += bc, bi++) {
        u32 nb = std::min<u32>(bc, n_cols - c0);
        streams_pick(ctx, (int)bi, ns);
        rc = interpolate_cols(ctx, tw, sub_colset(cols, c0), nb, (int)log_size);
    }
    int rj = streams_join(ctx, ns);
    return rc != NX_OK ? rc : rj;
}

int fft_evaluate(nx_ctx* ctx, const nx_twiddles* tw, ColSet polys, uint32_t n_cols, uint32_t log_size, uint32_t log_expand, ColSet out) {
    int n = (int)(log_size + log_expand);
    NX_TRY(check_tw(ctx, tw, n));
    tune_init();
    KTimer t(ctx, NX_T_LDE, (uint64_t)n_cols * ((4ull << log_size) + (4ull << n)));
    const u32 bc = batch_for(ctx, log_size);
    const int ns = n_cols > bc ? ctx->opt.fft_streams : 1;
    NX_TRY(streams_fork(ctx, ns));
    int rc = NX_OK;
    for (u32 c0 = 0, bi = 0; c0 < n_cols && rc == NX_OK; c0 += bc, bi++) {
        u32 nb = std::min<u32>(bc, n_cols - c0);
        streams_pick(ctx, (int)bi, ns);
        rc = evaluate_cols(ctx, tw, sub_colset(polys, c0), nb, (int)log_size, n, sub_colset(out, c0));
    }
    int rj = streams_join(ctx, ns);
    return rc != NX_OK ? rc : rj;
}

// iFFT + LDE per column batch (coefficients stay cache-resident between the two transforms).
int fft_lde(nx_ctx* ctx, const nx_twiddles* tw, ColSet cols, uint32_t n_cols, uint32_t log_size, uint32_t log_expand, ColSet out) {
    int n = (int)(log_size + log_expand);
    NX_TRY(check_tw(ctx, tw, n));
    tune_init();
    // algorithmic bytes: read N evals, write N coeffs, write M = 2^n LDE words
    KTimer t(ctx, NX_T_LDE, (uint64_t)n_cols * ((8ull << log_size) + (4ull << n)));
    const u32 bc = batch_for(ctx, log_size);
    const int ns = n_cols > bc ? ctx->opt.fft_streams : 1;
    NX_TRY(streams_fork(ctx, ns));
    int rc = NX_OK;
    for (u32 c0 = 0, bi = 0; c0 < n_cols && rc == NX_OK; c0 += bc, bi++) {
        u32 nb = std::min<u32>(bc, n_cols - c0);
        streams_pick(ctx, (int)bi, ns);
        if (log_expand == 1 && log_size >= 14 && ctx->opt.fft_fused) {
            rc = fft13_lde(ctx, tw, sub_colset(cols, c0), nb, (int)log_size, sub_colset(out, c0));   // middle passes fused (fft13.hip)
            continue;
        }
        rc = interpolate_cols(ctx, tw, sub_colset(cols, c0), nb, (int)log_size);
        if (rc == NX_OK) rc = evaluate_cols(ctx, tw, sub_colset(cols, c0), nb, (int)log_size, n, sub_colset(out, c0));
    }
    int rj = streams_join(ctx, ns);
    return rc != NX_OK ? rc : rj;
}

// ------------------------------------------------------------------ permutations (K1, R3) ---
__global__ void bit_reverse_kernel(u32* col, int log) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << log)) return;
    u32 j = bitrev(i, log);
    if (i < j) { u32 a = col[i], b = col[j]; col[i] = b; col[j] = a; }
}

// out[i] = natural[ coset_index( bitrev(i) ) ]: natural coset order -> bit-reversed circle-domain order
__device__ __forceinline__ u32 natural_index_of(u32 i, int log) {
    u32 N = 1u << log, d = bitrev(i, log);
    return d < N / 2 ? 2 * d : 2 * N - 1 - 2 * d;
}
__global__ void finalize_kernel(ColSet src, ColSet dst, u32 n_cols, int log) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    u32 c = blockIdx.y;
    if (i >= (1u << log) || c >= n_cols) return;
    dst.col(c)[i] = src.col(c)[natural_index_of(i, log)];
}

__global__ void twiddle_double_kernel(const u32* tw, const u32* itw, u32* tw2, u32* itw2, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { tw2[i] = tw[i] << 1; itw2[i] = itw[i] << 1; }
}

}  // namespace nx

using namespace nx;

extern "C" {

int nx_twiddles_create(nx_ctx* ctx, uint32_t log_half_coset, nx_twiddles** out) {
    NX_GUARD(ctx);
    if (!ctx || !out) return set_err(ctx, NX_ERR_ARG, "nx_twiddles_create: NULL argument");
    if (log_half_coset < 1 || log_half_coset > 28) return set_err(ctx, NX_ERR_ARG, "nx_twiddles_create: log_half_coset out of range [1,28]");
    nx_twiddles* t = new nx_twiddles();
    t->ctx = ctx; t->log_half = log_half_coset; t->d_tw = nullptr; t->d_itw = nullptr; t->d_tw2 = nullptr; t->d_itw2 = nullptr;
    size_t bytes = (size_t)4 << log_half_coset;
    int rc0 = dev_alloc(ctx, bytes, (void**)&t->d_tw);
    if (rc0 == NX_OK) rc0 = dev_alloc(ctx, bytes, (void**)&t->d_itw);
    if (rc0 == NX_OK) rc0 = dev_alloc(ctx, bytes, (void**)&t->d_tw2);
    if (rc0 == NX_OK) rc0 = dev_alloc(ctx, bytes, (void**)&t->d_itw2);
    if (rc0 != NX_OK) { dev_free(ctx, t->d_tw); dev_free(ctx, t->d_itw); dev_free(ctx, t->d_tw2); dev_free(ctx, t->d_itw2); delete t; return rc0; }
    hipError_t e;
    u32 total = 1u << log_half_coset;
    const GenTable gen = gen_table();
    hipLaunchKernelGGL(twiddle_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, t->d_tw, t->d_itw, (int)log_half_coset, gen);
    hipLaunchKernelGGL(twiddle_double_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, t->d_tw, t->d_itw, t->d_tw2, t->d_itw2, total);
    e = hipGetLastError();
    if (e != hipSuccess) { dev_free(ctx, t->d_tw); dev_free(ctx, t->d_itw); dev_free(ctx, t->d_tw2); dev_free(ctx, t->d_itw2); delete t; return hip_fail(ctx, e, "twiddle_kernel", __FILE__, __LINE__); }
    *out = t;
    return NX_OK;
}

}  // extern "C" (re-opened below)
namespace nx {
// Twiddles of the N = 2^n points that are the FIRST HALF of the bit-reversed CanonicCoset(n + 1) circle domain: an N-point circle
// domain (a half coset of size N/2 and its conjugate) whose layer-l twiddles are the first half of layer l of the size-2^(n+1) tables —
// after the top layer of the 2^(n+1)-point transform of N coefficients both halves hold the same data, and the remaining layers act
// on each half with its half of every table.  The result is laid out like a tree of log_half = n - 1, so every transform entry point
// works on it unchanged: interpolating the first half of an extension returns the N coefficients.
// depth d in general: the first 1 / 2^d of the bit-reversed CanonicCoset(n + d) domain, with the first 2^-d of every layer of the
// size-2^(n+d) tables (d = 2: the first QUARTER of the 4N-point domain — the quarter-domain composition of prover.hip;
// tests/test_exact_algebra_cpu.py::test_degree_four_quotient_from_3n_plus_1_samples checks the statement on the oracle).
__global__ void sub_twiddle_kernel(const u32* __restrict__ s0, const u32* __restrict__ s1, const u32* __restrict__ s2, const u32* __restrict__ s3,
                                   u32* __restrict__ d0, u32* __restrict__ d1, u32* __restrict__ d2, u32* __restrict__ d3, int n, int L, int depth) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x, half = 1u << (n - 1);
    if (i >= half) return;
    if (i + 1 == half) { d0[i] = 1; d1[i] = 1; d2[i] = 2; d3[i] = 2; return; }     // the pad word of a tree (never a twiddle)
    const u32 m = half - i;                                   // in (2^(K-1), 2^K], K = n - layer
    const int K = 32 - __builtin_clz(m - 1);
    const u32 h = (1u << K) - m;
    const u32 src = (1u << L) - (1u << (K + depth)) + h;
    d0[i] = s0[src]; d1[i] = s1[src]; d2[i] = s2[src]; d3[i] = s3[src];
}
int twiddles_first_part(nx_ctx* ctx, const nx_twiddles* tw, uint32_t n, uint32_t depth, nx_twiddles** out) {
    if (depth < 1 || depth > 4 || n < 3 || n + depth - 1 > tw->log_half) return set_err(ctx, NX_ERR_ARG, "twiddles_first_part: need 1 <= depth <= 4 and 3 <= n <= log_half + 1 - depth of the source tree");
    nx_twiddles* t = new nx_twiddles();
    t->ctx = ctx; t->log_half = n - 1; t->d_tw = t->d_itw = t->d_tw2 = t->d_itw2 = nullptr;
    const size_t bytes = (size_t)4 << (n - 1);
    int rc = dev_alloc(ctx, bytes, (void**)&t->d_tw);
    if (rc == NX_OK) rc = dev_alloc(ctx, bytes, (void**)&t->d_itw);
    if (rc == NX_OK) rc = dev_alloc(ctx, bytes, (void**)&t->d_tw2);
    if (rc == NX_OK) rc = dev_alloc(ctx, bytes, (void**)&t->d_itw2);
    if (rc != NX_OK) { dev_free(ctx, t->d_tw); dev_free(ctx, t->d_itw); dev_free(ctx, t->d_tw2); dev_free(ctx, t->d_itw2); delete t; return rc; }
    const u32 half = 1u << (n - 1);
    hipLaunchKernelGGL(sub_twiddle_kernel, dim3((half + 255) / 256), dim3(256), 0, ctx->stream, tw->d_tw, tw->d_itw, tw->d_tw2, tw->d_itw2, t->d_tw, t->d_itw, t->d_tw2, t->d_itw2,
                       (int)n, (int)tw->log_half, (int)depth);
    if (hipGetLastError() != hipSuccess) { nx_twiddles_destroy(t); return set_err(ctx, NX_ERR_HIP, "sub_twiddle_kernel launch failed"); }
    *out = t;
    return NX_OK;
}
}  // namespace nx
extern "C" {

void nx_twiddles_destroy(nx_twiddles* tw) {
    NX_GUARD(tw ? tw->ctx : nullptr);
    if (!tw) return;
    dev_free(tw->ctx, tw->d_tw); dev_free(tw->ctx, tw->d_itw); dev_free(tw->ctx, tw->d_tw2); dev_free(tw->ctx, tw->d_itw2);
    delete tw;
}

int nx_twiddles_download(nx_ctx* ctx, const nx_twiddles* tw, uint32_t* h_tw, uint32_t* h_itw) {
    NX_GUARD(ctx);
    if (!ctx || !tw || !h_tw || !h_itw) return set_err(ctx, NX_ERR_ARG, "nx_twiddles_download: NULL argument");
    NX_TRY(nx_download(ctx, h_tw, tw->d_tw, (size_t)1 << tw->log_half));
    return nx_download(ctx, h_itw, tw->d_itw, (size_t)1 << tw->log_half);
}

int nx_interpolate_batch(nx_ctx* ctx, const nx_twiddles* tw, uint32_t* const* d_cols, uint32_t n_cols, uint32_t log_size) {
    NX_GUARD(ctx);
    if (n_cols == 0) return NX_OK;
    ColSet cs; NX_TRY(make_colset(ctx, d_cols, n_cols, &cs));
    return fft_interpolate(ctx, tw, cs, n_cols, log_size);
}

int nx_evaluate_batch(nx_ctx* ctx, const nx_twiddles* tw, const uint32_t* const* d_polys, uint32_t n_cols, uint32_t log_size,
                      uint32_t log_expand, uint32_t* const* d_out) {
    NX_GUARD(ctx);
    if (n_cols == 0) return NX_OK;
    ColSet p, o;
    NX_TRY(make_colset(ctx, d_polys, n_cols, &p));
    NX_TRY(make_colset(ctx, d_out, n_cols, &o));
    return fft_evaluate(ctx, tw, p, n_cols, log_size, log_expand, o);
}

int nx_bit_reverse(nx_ctx* ctx, uint32_t* d_col, uint32_t log_size) {
    NX_GUARD(ctx);
    if (!ctx || !d_col) return set_err(ctx, NX_ERR_ARG, "nx_bit_reverse: NULL argument");
    if (log_size > 31) return set_err(ctx, NX_ERR_ARG, "nx_bit_reverse: log_size too large");
    u32 n = 1u << log_size;
    hipLaunchKernelGGL(bit_reverse_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_col, (int)log_size);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

int nx_finalize_columns(nx_ctx* ctx, const uint32_t* const* d_src_natural, uint32_t* const* d_dst, uint32_t n_cols, uint32_t log_size) {
    NX_GUARD(ctx);
    if (n_cols == 0) return NX_OK;
    if (log_size < 1) return set_err(ctx, NX_ERR_ARG, "nx_finalize_columns: log_size < 1");
    ColSet s, d;
    NX_TRY(make_colset(ctx, d_src_natural, n_cols, &s));
    NX_TRY(make_colset(ctx, d_dst, n_cols, &d));
    u32 n = 1u << log_size;
    for (u32 c0 = 0; c0 < n_cols; c0 += 32768) {
        u32 nb = std::min<u32>(32768, n_cols - c0);
        hipLaunchKernelGGL(finalize_kernel, dim3((n + 255) / 256, nb), dim3(256), 0, ctx->stream, sub_colset(s, c0), sub_colset(d, c0), nb, (int)log_size);
        NX_LAUNCH_CHECK(ctx);
    }
    return NX_OK;
}


// R3 + R4 for a whole host-resident trace: pin every host column in place (hipHostRegister, no
// staging copy), stream it over PCIe on a side stream and run the coset-order -> bit-reversed-circle-domain permutation
// (or nothing, when the host already holds that order) behind it on the main stream, two columns in flight.  This replaces
// the reference's per-column CPU passes (coset_order_to_circle_domain_order + from_iter + bit_reverse_column + clone).
int nx_upload_columns(nx_ctx* ctx, const uint32_t* const* h_cols, uint32_t n_cols, uint32_t log_size, uint32_t* const* d_cols, int coset_order) {
    NX_GUARD(ctx);
    if (!ctx || (n_cols && (!h_cols || !d_cols))) return set_err(ctx, NX_ERR_ARG, "nx_upload_columns: NULL argument");
    if (log_size < 1 || log_size > 30) return set_err(ctx, NX_ERR_ARG, "nx_upload_columns: bad log_size");
    const size_t n = (size_t)1 << log_size, bytes = n * 4;
    uint32_t* d_tmp[2] = {nullptr, nullptr};
    hipEvent_t copied[2], consumed[2];
    int rc = NX_OK;
    if (coset_order) for (int k = 0; k < 2 && rc == NX_OK; k++) rc = dev_alloc(ctx, bytes, (void**)&d_tmp[k]);
    for (int k = 0; k < 2; k++) { (void)hipEventCreateWithFlags(&copied[k], hipEventDisableTiming); (void)hipEventCreateWithFlags(&consumed[k], hipEventDisableTiming); }
    hipStream_t copy_stream = ctx->side[0];
    hipError_t e = hipEventRecord(ctx->fork_ev, ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(copy_stream, ctx->fork_ev, 0);
    std::vector<bool> registered(n_cols, false);
    for (uint32_t c = 0; c < n_cols && rc == NX_OK && e == hipSuccess; c++) {
        const int k = c & 1;
        registered[c] = !host_pinned_by_owner(h_cols[c], bytes) && hipHostRegister((void*)h_cols[c], bytes, hipHostRegisterDefault) == hipSuccess;   // else pinned by its owner, or a pageable copy
        if (!registered[c]) (void)hipGetLastError();
        uint32_t* dst = coset_order ? d_tmp[k] : d_cols[c];
        if (coset_order && c >= 2) e = hipStreamWaitEvent(copy_stream, consumed[k], 0);      // the permute kernel of column c-2 has read d_tmp[k]
        if (e == hipSuccess) {
            // a column that could not be pinned (already registered by someone else, ...) goes through the context's bounce buffer: a
            // straight copy from pageable memory would be pinned in place by the runtime and released lazily (internal.h, h_bounce)
            if (registered[c] || host_pinned_by_owner(h_cols[c], bytes)) e = hipMemcpyAsync(dst, h_cols[c], bytes, hipMemcpyHostToDevice, copy_stream);
            else if (copy_h2d_blocking(ctx, dst, h_cols[c], bytes, copy_stream) != NX_OK) e = hipErrorUnknown;
        }
        if (e == hipSuccess) e = hipEventRecord(copied[k], copy_stream);
        if (e == hipSuccess && coset_order) {
            e = hipStreamWaitEvent(ctx->stream, copied[k], 0);
            if (e == hipSuccess) {
                ColSet s1, d1; s1.base = d_tmp[k]; s1.stride = 0; s1.table = nullptr; d1.base = d_cols[c]; d1.stride = 0; d1.table = nullptr;
                hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((n + 255) / 256), 1), dim3(256), 0, ctx->stream, s1, d1, 1u, (int)log_size);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipEventRecord(consumed[k], ctx->stream);
        }
    }
    // the caller may reuse / free the host columns on return: everything must have left them
    hipError_t e2 = hipStreamSynchronize(copy_stream);
    hipError_t e3 = hipStreamSynchronize(ctx->stream);
    for (uint32_t c = 0; c < n_cols; c++) if (registered[c]) (void)hipHostUnregister((void*)h_cols[c]);
    for (int k = 0; k < 2; k++) { (void)hipEventDestroy(copied[k]); (void)hipEventDestroy(consumed[k]); dev_free(ctx, d_tmp[k]); }
    if (rc != NX_OK) return rc;
    if (e != hipSuccess) return hip_fail(ctx, e, "nx_upload_columns", __FILE__, __LINE__);
    if (e2 != hipSuccess) return hip_fail(ctx, e2, "nx_upload_columns(copy sync)", __FILE__, __LINE__);
    if (e3 != hipSuccess) return hip_fail(ctx, e3, "nx_upload_columns(sync)", __FILE__, __LINE__);
    return NX_OK;
}

}  // extern "C" (re-opened below)
namespace nx {
int HostFeed::begin(nx_ctx* c, uint32_t log_size, int coset) {
    ctx = c; log = log_size; coset_order = coset; n_done = 0;
    if (log_size < 1 || log_size > 30) return set_err(c, NX_ERR_ARG, "host feed: bad log_size");
    for (int k = 0; k < 2; k++) {
        if (coset_order) NX_TRY(dev_alloc(ctx, (size_t)4 << log, (void**)&d_tmp[k]));
        NX_HIP(ctx, hipEventCreateWithFlags(&copied[k], hipEventDisableTiming));
        NX_HIP(ctx, hipEventCreateWithFlags(&consumed[k], hipEventDisableTiming));
    }
    // the destination columns (and d_tmp) come from the context's block cache, whose frees are ordered on ctx->stream only: the feed's
    // streams start behind everything the context has queued so far
    hipEvent_t here = nullptr;
    NX_HIP(ctx, hipEventCreateWithFlags(&here, hipEventDisableTiming));
    events.push_back(here);
    NX_HIP(ctx, hipEventRecord(here, ctx->stream));
    NX_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, here, 0));
    NX_HIP(ctx, hipStreamWaitEvent(ctx->perm_stream, here, 0));
    return NX_OK;
}
int HostFeed::chunk(const uint32_t* const* h_cols, uint32_t* const* d_cols, uint32_t n_cols, hipEvent_t* ready) {
    const size_t n = (size_t)1 << log, bytes = n * 4;
    for (uint32_t c = 0; c < n_cols; c++) {
        if (!h_cols[c] || !d_cols[c]) return set_err(ctx, NX_ERR_ARG, "host feed: NULL column");
        bool dma_ok = true;
        if (host_pinned_by_owner(h_cols[c], bytes)) {}                      // pinned once by its owner (nx_host_pin)
        else if (hipHostRegister((void*)h_cols[c], bytes, hipHostRegisterDefault) == hipSuccess) pinned.push_back(h_cols[c]);
        else { (void)hipGetLastError(); dma_ok = false; }                         // not pinnable: through the bounce buffer (never a pin-in-place copy: internal.h, h_bounce)
        const int k = (int)(n_done & 1);
        auto h2d = [&](uint32_t* dst, hipStream_t st) -> int {
            if (dma_ok) { NX_HIP(ctx, hipMemcpyAsync(dst, h_cols[c], bytes, hipMemcpyHostToDevice, st)); return NX_OK; }
            return copy_h2d_blocking(ctx, dst, h_cols[c], bytes, st);
        };
        if (!coset_order) {
            // no permutation to run: the second stream carries every other column's copy (two DMA queues in flight: 136.6 -> 132.7 ms
            // for the 374-column headline trace, bench.py host_trace)
            NX_TRY(h2d(d_cols[c], (n_done & 1) ? ctx->perm_stream : ctx->copy_stream));
        } else {
            if (n_done >= 2) NX_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, consumed[k], 0));      // the permutation of column n_done - 2 has read d_tmp[k]
            NX_TRY(h2d(d_tmp[k], ctx->copy_stream));
            NX_HIP(ctx, hipEventRecord(copied[k], ctx->copy_stream));
            NX_HIP(ctx, hipStreamWaitEvent(ctx->perm_stream, copied[k], 0));
            ColSet s1, d1; s1.base = d_tmp[k]; s1.stride = 0; s1.table = nullptr; d1.base = d_cols[c]; d1.stride = 0; d1.table = nullptr;
            hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((n + 255) / 256), 1), dim3(256), 0, ctx->perm_stream, s1, d1, 1u, (int)log);
            NX_LAUNCH_CHECK(ctx);
            NX_HIP(ctx, hipEventRecord(consumed[k], ctx->perm_stream));
        }
        n_done++;
    }
    hipEvent_t ev = nullptr;
    NX_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    events.push_back(ev);
    if (!coset_order) {                                     // the chunk is ready when both queues are through it
        hipEvent_t evp = nullptr;
        NX_HIP(ctx, hipEventCreateWithFlags(&evp, hipEventDisableTiming));
        events.push_back(evp);
        NX_HIP(ctx, hipEventRecord(evp, ctx->perm_stream));
        NX_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, evp, 0));
    }
    NX_HIP(ctx, hipEventRecord(ev, coset_order ? ctx->perm_stream : ctx->copy_stream));
    *ready = ev;
    return NX_OK;
}
int HostFeed::finish() {
    if (!ctx) return NX_OK;
    hipError_t e1 = hipStreamSynchronize(ctx->copy_stream), e2 = hipStreamSynchronize(ctx->perm_stream);
    for (const uint32_t* h : pinned) (void)hipHostUnregister((void*)h);
    pinned.clear();
    for (hipEvent_t ev : events) (void)hipEventDestroy(ev);
    events.clear();
    for (int k = 0; k < 2; k++) {
        if (copied[k]) (void)hipEventDestroy(copied[k]);
        if (consumed[k]) (void)hipEventDestroy(consumed[k]);
        copied[k] = consumed[k] = nullptr;
        if (d_tmp[k]) { dev_free(ctx, d_tmp[k]); d_tmp[k] = nullptr; }
    }
    nx_ctx* c = ctx; ctx = nullptr;
    if (e1 != hipSuccess) return hip_fail(c, e1, "host feed (copy stream)", __FILE__, __LINE__);
    if (e2 != hipSuccess) return hip_fail(c, e2, "host feed (permutation stream)", __FILE__, __LINE__);
    return NX_OK;
}
}  // namespace nx
extern "C" {

}  // extern "C" (re-opened below)
namespace nx {
// (the runtime accepts a second registration of a range, and the first hipHostUnregister then drops both: the book is kept here,
// process-wide because the registration is)
struct PinBook { std::mutex mu; std::map<const uint8_t*, size_t> ranges; };
static PinBook& pin_book() { static PinBook b; return b; }
bool host_pinned_by_owner(const void* p, size_t bytes) {
    PinBook& b = pin_book();
    std::lock_guard<std::mutex> lk(b.mu);
    auto it = b.ranges.upper_bound((const uint8_t*)p);
    if (it == b.ranges.begin()) return false;
    --it;
    return (const uint8_t*)p + bytes <= it->first + it->second;
}
}  // namespace nx
extern "C" {

int nx_host_pin(nx_ctx* ctx, const void* h, size_t bytes) {
    NX_GUARD(ctx);
    if (!ctx || !h || !bytes) return set_err(ctx, NX_ERR_ARG, "nx_host_pin: NULL / empty range");
    PinBook& b = pin_book();
    std::lock_guard<std::mutex> lk(b.mu);
    auto it = b.ranges.upper_bound((const uint8_t*)h);
    if (it != b.ranges.end() && it->first < (const uint8_t*)h + bytes) return set_err(ctx, NX_ERR_ARG, "nx_host_pin: the range overlaps a pinned one");
    if (it != b.ranges.begin()) { --it; if (it->first + it->second > (const uint8_t*)h) return set_err(ctx, NX_ERR_ARG, "nx_host_pin: the range overlaps a pinned one"); }
    NX_HIP(ctx, hipHostRegister((void*)h, bytes, hipHostRegisterDefault));
    b.ranges[(const uint8_t*)h] = bytes;
    return NX_OK;
}
int nx_host_unpin(nx_ctx* ctx, const void* h) {
    NX_GUARD(ctx);
    if (!ctx || !h) return set_err(ctx, NX_ERR_ARG, "nx_host_unpin: NULL argument");
    NX_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));      // nothing of this context may still be reading it
    NX_HIP(ctx, hipStreamSynchronize(ctx->perm_stream));
    NX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PinBook& b = pin_book();
    std::lock_guard<std::mutex> lk(b.mu);
    auto it = b.ranges.find((const uint8_t*)h);
    if (it == b.ranges.end()) return set_err(ctx, NX_ERR_ARG, "nx_host_unpin: not the start of a range pinned with nx_host_pin");
    b.ranges.erase(it);
    NX_HIP(ctx, hipHostUnregister((void*)h));
    return NX_OK;
}

int nx_upload_coset_order(nx_ctx* ctx, const uint32_t* h_natural, uint32_t log_size, uint32_t* d_dst) {
    NX_GUARD(ctx);
    if (!ctx || !h_natural || !d_dst) return set_err(ctx, NX_ERR_ARG, "nx_upload_coset_order: NULL argument");
    if (log_size < 1 || log_size > 30) return set_err(ctx, NX_ERR_ARG, "nx_upload_coset_order: bad log_size");
    uint32_t* d_tmp = nullptr;
    size_t n = (size_t)1 << log_size;
    NX_TRY(dev_alloc(ctx, n * 4, (void**)&d_tmp));
    int rc = copy_h2d_blocking(ctx, d_tmp, h_natural, n * 4);
    if (rc == NX_OK) { const uint32_t* sp = d_tmp; uint32_t* dp = d_dst; rc = nx_finalize_columns(ctx, &sp, &dp, 1, log_size); }
    (void)hipStreamSynchronize(ctx->stream);
    dev_free(ctx, d_tmp);
    return rc;
}

}  // extern "C"
