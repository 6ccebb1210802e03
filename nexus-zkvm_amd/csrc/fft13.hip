// Circle FFT / iFFT passes for transforms of >= 2^13 points: the fast path of K3/K4.
//
// Same math as fft.hip (Stwo CpuBackend circle.rs butterflies, bit-reversed order), different schedule:
//  * a pass owns 13 index bits: K butterfly layers [lo, lo+K) plus B = 13-K low bits that only make
//    the global accesses contiguous (runs of 2^B words, B = 0 for the pass that holds layers [0,13)).
//    2^22 points = 2 passes (13 + 9 layers); each pass is one HBM round trip.
//  * a block transforms one 2^13-row tile of one column (34 KB of LDS, 58-70 VGPRs, 3-4 blocks per CU).  (Column PAIRS per block
//    with 64-bit LDS accesses were measured slower in rounds 1-2 and are gone.)
//  * 4 layers per LDS round trip (radix-16 in registers); the pass's top layer is fused into the
//    global<->LDS staging (its twiddle is uniform over the tile); the layer count K is a template constant.
//  * the twiddles of a round are requested one round ahead from DOUBLED tables (2t: what the doubled-factor
//    product consumes), so a round waits on LDS only; all global accesses are address-space-1.
//  * an LDE with blow-up 2 runs its middle (inverse last pass + forward first pass of both replicas) as ONE
//    launch, lde_mid_kernel: the coefficients are written once and never re-read.
//  * the trivial top layers of an LDE (inputs that are zero by construction) are not computed: the
//    coefficient tile is replicated, each replica runs the remaining layers with its own twiddles.
//  * blockIdx -> (tile, column group) keeps all column groups of a tile on one XCD (same twiddles,
//    neighbouring runs of the same cache lines) — for speed only, correctness never depends on it.
#include "internal.h"
#include <atomic>
#include <algorithm>
#include <stdlib.h>

namespace nx {

#ifndef NX_FFT_MINWAVES   // A/B knob: 5 forces <= 96 VGPRs (measured: spills, slower)
#define NX_FFT_MINWAVES 1
#endif
#ifndef NX_FFT_MINWAVES1  // single-column tiles: forcing 8 waves/SIMD (<= 64 VGPRs, 4 blocks per CU) measured 9 % slower than the natural 58-70
#define NX_FFT_MINWAVES1 1
#endif
constexpr int T13_S = 13;
constexpr u32 T13_ROWS = 1u << T13_S;
constexpr u32 T13_HALF = T13_ROWS / 2;

struct Pass13 {
    ColSet src, dst;
    const u32* tw;   // forward or inverse DOUBLED twiddle buffer (2^tw_log words, 2 * twiddle each)
    u32 tw_log;
    int n;           // log size of the whole transform (index space of dst)
    int log_in;      // src holds 2^log_in words; dst index bits >= log_in select a replica
    int lo, K, B;    // layers [lo, lo+K), K + B == 13
    u32 scale;       // inverse passes: multiply the outputs by this (0 = none)
    u32 n_cols, n_groups, tiles;
    u32 rep_log;     // log2 of the replica count of this pass's source tiles (LDE top pass), else 0
};

template <int CB> struct alignas(4 * CB) Row { u32 c[CB]; };

// Global-memory accessors.  The column and twiddle pointers reach the kernel inside a by-value struct, so clang types them as
// generic ("flat"): a flat access may alias LDS, is ordered against every ds_read/ds_write and counts on lgkmcnt as well — the
// staging loads of a tile were issued one at a time, each waiting for the previous tile quarter's LDS write.  Casting to
// address space 1 turns them into global_load/global_store that the compiler is free to issue together.
#define NX_GLOBAL __attribute__((address_space(1)))
typedef u32 v4u32 __attribute__((ext_vector_type(4)));
typedef u32 v2u32 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 gload4(const u32* p) { const v4u32 v = *(NX_GLOBAL const v4u32*)p; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint2 gload2(const u32* p) { const v2u32 v = *(NX_GLOBAL const v2u32*)p; return make_uint2(v.x, v.y); }
__device__ __forceinline__ u32 gload1(const u32* p) { return *(NX_GLOBAL const u32*)p; }
__device__ __forceinline__ void gstore4(u32* p, uint4 v) { v4u32 w = {v.x, v.y, v.z, v.w}; *(NX_GLOBAL v4u32*)p = w; }


__device__ __forceinline__ u32 pad13(u32 t) { return t + (t >> 4); }

// t2 = 2 * twiddle (read from the doubled tables, see field.cuh m_mul_dbl)
template <bool INV>
__device__ __forceinline__ void bfly13(u32& x0, u32& x1, u32 t2, bool neg) {
    if (INV) {
        u32 s = m_add(x0, x1);
        u32 d = neg ? m_sub(x1, x0) : m_sub(x0, x1);
        x0 = s; x1 = m_mul_dbl(d, t2);
    } else {
        u32 m = m_mul_dbl(x1, t2);
        u32 a = m_add(x0, m), b = m_sub(x0, m);
        x0 = neg ? b : a; x1 = neg ? a : b;
    }
}

#ifndef NX_FFT_SCALAR_TW   // A/B knob: 0 = the twiddles of every round through vector loads
#define NX_FFT_SCALAR_TW 1
#endif
// Twiddle tables are never written while a transform runs: through the constant address space a WAVE-UNIFORM address becomes a scalar load
// (s_load_dwordx8 ...: no VMEM instruction, no VGPRs), any other address an ordinary global load.
#define NX_CONSTANT __attribute__((address_space(4)))
template <int CNT>
__device__ __forceinline__ void load_tw13_uniform(const u32* __restrict__ p, u32* dst) {
    NX_CONSTANT const u32* q = (NX_CONSTANT const u32*)p;
#pragma unroll
    for (int k = 0; k < CNT; k++) dst[k] = q[k];
}
template <int CNT>
__device__ __forceinline__ void load_tw13(const u32* __restrict__ p, u32* dst) {
    if constexpr (CNT == 8) {
        uint4 a = gload4(p), b = gload4(p + 4);
        dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w; dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
    } else if constexpr (CNT == 4) {
        uint4 a = gload4(p);
        dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w;
    } else if constexpr (CNT == 2) {
        uint2 a = gload2(p);
        dst[0] = a.x; dst[1] = a.y;
    } else {
        dst[0] = gload1(p);
    }
}

// twiddles of layers l0+Q .. l0+R-1 for the 2^R rows starting at global index g0: layer q at tw[2^R - 2^(R-q)], 2^(R-1-q) words
template <int R, int Q, bool CIRCLE, bool UNIFORM = false>
__device__ __forceinline__ void load_round_tw13(const u32* const __restrict__* twl, u32 g0, int l0, u32* tw) {
    if constexpr (Q < R) {
        if constexpr (!(CIRCLE && Q == 0)) {
            if constexpr (UNIFORM) load_tw13_uniform<(1 << (R - 1 - Q))>(twl[Q] + (g0 >> (l0 + Q + 1)), tw + ((1 << R) - (1 << (R - Q))));
            else load_tw13<(1 << (R - 1 - Q))>(twl[Q] + (g0 >> (l0 + Q + 1)), tw + ((1 << R) - (1 << (R - Q))));
        }
        load_round_tw13<R, Q + 1, CIRCLE, UNIFORM>(twl, g0, l0, tw);
    }
}

// the R butterfly layers of 2^R rows held in registers; CIRCLE: layer 0 is the circle layer, its twiddles are
// derived from the first line layer's (x, y) pairs as [y, -y, -x, x]
template <int R, int CB, bool INV, bool CIRCLE>
__device__ __forceinline__ void butterflies(Row<CB>* v, const u32* tw) {
    static_assert(!CIRCLE || R >= 3, "circle rounds need the (x, y) pair of the first line layer");
#pragma unroll
    for (int qq = 0; qq < R; qq++) {
        const int q = INV ? qq : R - 1 - qq;
#pragma unroll
        for (int e = 0; e < (1 << R); e++) {
            if (e & (1 << q)) continue;
            const int h = e >> (q + 1);
            if (CIRCLE && q == 0) {
                const int c = h >> 2, sel = h & 3;
                const u32 t = tw[((1 << R) - (1 << (R - 1))) + 2 * c + ((sel & 2) ? 0 : 1)];
                const bool neg = sel == 1 || sel == 2;
#pragma unroll
                for (int k = 0; k < CB; k++) bfly13<INV>(v[e].c[k], v[e | 1].c[k], t, neg);
            } else {
                const u32 t = tw[((1 << R) - (1 << (R - q))) + h];
#pragma unroll
                for (int k = 0; k < CB; k++) bfly13<INV>(v[e].c[k], v[e | (1 << q)].c[k], t, false);
            }
        }
    }
}

// twiddle request for the full round at tile bit `bp`, for the 2^RB rows this lane owns
// From tile bit 6 up a round's twiddles are the same for the 64 lanes of a wave: the lane bits below `bp` are shifted out of every twiddle
// index (they address rows INSIDE a butterfly group) and the bits from `bp` up are the wave's.  Lane 0's index then stands for the wave, the
// addresses are wave-uniform and the loads scalar (round 6; NX_FFT_SCALAR_TW).
template <int RB, bool CIRCLE>
__device__ __forceinline__ void tw_request(const Pass13& a, int bp, u32 tile_base, u32* tw) {
    const bool uniform = NX_FFT_SCALAR_TW && !CIRCLE && bp >= 6;
    const u32 w = uniform ? (u32)__builtin_amdgcn_readfirstlane((int)threadIdx.x) : threadIdx.x;
    const u32 wl = w & ((1u << bp) - 1), wh = w >> bp;
    const u32 t0 = (wh << (bp + RB)) | wl;
    const u32 g0 = tile_base + ((t0 >> a.B) << a.lo) + (t0 & ((1u << a.B) - 1));
    const int l0 = a.lo + (bp - a.B);
    const u32* __restrict__ twl[RB];
#pragma unroll
    for (int q = 0; q < RB; q++) twl[q] = a.tw + ((1u << a.tw_log) - (1u << (a.n - (l0 + q))));
    if (uniform) load_round_tw13<RB, 0, CIRCLE, true>(twl, g0, l0, tw);
    else load_round_tw13<RB, 0, CIRCLE, false>(twl, g0, l0, tw);
}

// One full LDS round trip (every lane owns 2^RB rows): tile bits [bp, bp+RB), twiddles already in registers.
template <int RB, int CB, bool INV, bool CIRCLE>
__device__ __forceinline__ void round_full(Row<CB>* lds, int bp, const u32* tw) {
    const u32 w = threadIdx.x;
    const u32 wl = w & ((1u << bp) - 1), wh = w >> bp;
    const u32 t0 = (wh << (bp + RB)) | wl;
    Row<CB> v[1 << RB];
    if (bp == 0 || bp >= 4) {   // the padding is linear in the row step
        const u32 estride = bp ? ((1u << bp) + ((1u << bp) >> 4)) : 1u;
        const u32 p0 = pad13(t0);
#pragma unroll
        for (int e = 0; e < (1 << RB); e++) v[e] = lds[p0 + e * estride];
        butterflies<RB, CB, INV, CIRCLE>(v, tw);
#pragma unroll
        for (int e = 0; e < (1 << RB); e++) lds[p0 + e * estride] = v[e];
    } else {
#pragma unroll
        for (int e = 0; e < (1 << RB); e++) v[e] = lds[pad13(t0 + ((u32)e << bp))];
        butterflies<RB, CB, INV, CIRCLE>(v, tw);
#pragma unroll
        for (int e = 0; e < (1 << RB); e++) lds[pad13(t0 + ((u32)e << bp))] = v[e];
    }
}

// Remainder round (R < RB layers at the bottom of a non-FIRST pass, bp >= 4): several row groups per lane.
template <int R, int CB, int NT, bool INV>
__device__ __forceinline__ void round_rem(Row<CB>* lds, const Pass13& a, int bp, u32 tile_base) {
    constexpr u32 NBLK = T13_ROWS >> R;
    const u32 estride = (1u << bp) + ((1u << bp) >> 4);
    const u32 maskB = (1u << a.B) - 1;
    const int l0 = a.lo + (bp - a.B);
    const u32* __restrict__ twl[R];
#pragma unroll
    for (int q = 0; q < R; q++) twl[q] = a.tw + ((1u << a.tw_log) - (1u << (a.n - (l0 + q))));
#pragma unroll 1
    for (u32 w = threadIdx.x; w < NBLK; w += NT) {
        const u32 wl = w & ((1u << bp) - 1), wh = w >> bp;
        const u32 t0 = (wh << (bp + R)) | wl;
        const u32 p0 = pad13(t0);
        Row<CB> v[1 << R];
#pragma unroll
        for (int e = 0; e < (1 << R); e++) v[e] = lds[p0 + e * estride];
        const u32 g0 = tile_base + ((t0 >> a.B) << a.lo) + (t0 & maskB);
        u32 tw[1 << R];
        load_round_tw13<R, 0, false>(twl, g0, l0, tw);
        butterflies<R, CB, INV, false>(v, tw);
#pragma unroll
        for (int e = 0; e < (1 << R); e++) lds[p0 + e * estride] = v[e];
    }
}

template <int RB, int CB, int NT, bool INV>
__device__ __forceinline__ void rem_round13(Row<CB>* lds, const Pass13& a, int rem, int bp, u32 tile_base) {
    if (rem == 3) { if constexpr (RB > 3) round_rem<3, CB, NT, INV>(lds, a, bp, tile_base); }
    else if (rem == 2) round_rem<2, CB, NT, INV>(lds, a, bp, tile_base);
    else if (rem == 1) round_rem<1, CB, NT, INV>(lds, a, bp, tile_base);
    __syncthreads();
}

__device__ __forceinline__ u32 get4(const uint4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// FIRST: the pass that owns layers [0, 13) on a contiguous tile (lo == 0, B == 0), including the circle layer.
// One block = one (tile, group of CB columns), 2^(13-RB) lanes.
// KT: the pass's layer count as a compile-time constant (0 = read it from the pass descriptor).  With K known the round
// sequence is straight-line code; with a run-time K the twiddle registers are merged across branches, the register allocator
// inserts copies right after the prefetch loads and every round waits for its successor's twiddles (measured: the specialised
// kernels are the ones the 2^20..2^24 transforms use).
template <bool INV, bool FIRST, int CB, int RB, int KT>
__global__ __launch_bounds__(T13_ROWS >> RB, CB == 1 ? NX_FFT_MINWAVES1 : NX_FFT_MINWAVES) void fft13_kernel(Pass13 a) {
    constexpr int NT = T13_ROWS >> RB;
    constexpr int MAXR = (T13_S - 1 + RB - 1) / RB;       // full rounds in a 13-layer pass
    extern __shared__ __attribute__((aligned(16))) u32 lds13[];
    Row<CB>* lds = reinterpret_cast<Row<CB>*>(lds13);

    const int K = FIRST ? T13_S : (KT ? KT : a.K), B = FIRST ? 0 : (KT ? T13_S - KT : a.B), lo = FIRST ? 0 : a.lo;
    const int lb = lo - B;
    const u32 maskB = (1u << B) - 1;
    auto goff = [&](u32 t) -> u32 { return FIRST ? t : (((t >> B) << lo) + (t & maskB)); };
    // full rounds sit at tile bits bp(i) = B + rem + RB i, i < nfull; the remainder round (K-1 mod RB layers) at bit B
    const int K1 = K - 1, nfull = K1 / RB, rem = K1 % RB;
    auto bp_of = [&](int i) -> int { return B + rem + RB * i; };

    struct Work { u32 grp, tile_base, src_base, high; };
    // work item -> (tile, column group).  XCD x owns source tiles [x, x+1) * src_tiles/8; the replicas of a source tile and its
    // column groups run back to back on that XCD, so the tile's coefficients and twiddles are fetched into one L2 once.
    auto decode = [&](u32 b) -> Work {
        u32 tile, grp;
        const u32 src_tiles = a.tiles >> a.rep_log;
        if (src_tiles >= 8) {
            const u32 xcd = b & 7, y = b >> 3, idx = y / a.n_groups;
            grp = y % a.n_groups;
            tile = ((idx & ((1u << a.rep_log) - 1)) * src_tiles) + xcd * (src_tiles >> 3) + (idx >> a.rep_log);
        } else { grp = b % a.n_groups; tile = b / a.n_groups; }
        const u32 lowblock = tile & ((1u << lb) - 1), high = tile >> lb;
        Work w; w.grp = grp; w.high = high;
        w.tile_base = (high << (lo + K)) | (lowblock << B);
        w.src_base = w.tile_base & ((1u << a.log_in) - 1);
        return w;
    };

    // every global load of a tile is issued before the first LDS write (one memory round trip, not one per quarter)
    constexpr int IT = INV ? (int)(T13_ROWS / 4) / NT : (int)(T13_HALF / 4) / NT;
    uint4 xa[IT][CB], xb[INV ? 1 : IT][CB];
    auto issue = [&](const Work& wk) {
        const u32* __restrict__ s[CB];
#pragma unroll
        for (int k = 0; k < CB; k++) s[k] = a.src.col(min(wk.grp * CB + k, a.n_cols - 1));
#pragma unroll
        for (int it = 0; it < IT; it++) {
            const u32 t = (threadIdx.x + it * NT) * 4;
            if (INV) {
                const u32 g = wk.src_base + goff(t);
#pragma unroll
                for (int k = 0; k < CB; k++) xa[it][k] = gload4(s[k] + g);
            } else {
                const u32 ga = wk.src_base + goff(t), gb = wk.src_base + goff(t + T13_HALF);
#pragma unroll
                for (int k = 0; k < CB; k++) { xa[it][k] = gload4(s[k] + ga); xb[it][k] = gload4(s[k] + gb); }
            }
        }
    };

    const Work wk = decode(blockIdx.x);
    issue(wk);
    {
        u32* __restrict__ d[CB];
        bool live[CB];
#pragma unroll
        for (int k = 0; k < CB; k++) {
            const u32 c = min(wk.grp * CB + k, a.n_cols - 1);
            live[k] = wk.grp * CB + k < a.n_cols;
            d[k] = a.dst.col(c);
        }
        const u32 tile_base = wk.tile_base, high = wk.high;
        u32 twA[1 << RB], twB[1 << RB];
        if (nfull) {
            const int i0 = INV ? 0 : nfull - 1;
            if (FIRST && i0 == 0) tw_request<RB, true>(a, 0, tile_base, twA); else tw_request<RB, false>(a, bp_of(i0), tile_base, twA);
        }
        // the pass's top layer (tile bit 12): one twiddle for the whole tile
        const int le = lo + K - 1;
        const u32 te2 = a.tw[((1u << a.tw_log) - (1u << (a.n - le))) + high], te = te2 >> 1;

#pragma unroll
        for (int it = 0; it < IT; it++) {
            const u32 t = (threadIdx.x + it * NT) * 4;
            if (INV) {
                const u32 p = pad13(t);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    Row<CB> r;
#pragma unroll
                    for (int k = 0; k < CB; k++) r.c[k] = get4(xa[it][k], i);
                    lds[p + i] = r;
                }
            } else {
                const u32 pa = pad13(t), pb = pad13(t + T13_HALF);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    Row<CB> ra, rb;
#pragma unroll
                    for (int k = 0; k < CB; k++) {
                        u32 u = get4(xa[it][k], i), v = get4(xb[it][k], i);
                        bfly13<false>(u, v, te2, false);
                        ra.c[k] = u; rb.c[k] = v;
                    }
                    lds[pa + i] = ra;
                    lds[pb + i] = rb;
                }
            }
        }
        __syncthreads();

        // step j of the round sequence uses twiddle registers (j even ? twA : twB) and requests step j+1's into the other set
        if (INV) {
            if (rem) rem_round13<RB, CB, NT, true>(lds, a, rem, B, tile_base);
#pragma unroll
            for (int j = 0; j < MAXR; j++) {
                if (j < nfull) {
                    u32* cur = (j & 1) ? twB : twA;
                    u32* nxt = (j & 1) ? twA : twB;
                    if (j + 1 < nfull) tw_request<RB, false>(a, bp_of(j + 1), tile_base, nxt);
                    __builtin_amdgcn_sched_barrier(0);   // keep the requests ahead of this round's butterflies
                    if (FIRST && j == 0) round_full<RB, CB, true, true>(lds, 0, cur); else round_full<RB, CB, true, false>(lds, bp_of(j), cur);
                    __syncthreads();
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < MAXR; j++) {
                if (j < nfull) {
                    const int i = nfull - 1 - j;
                    u32* cur = (j & 1) ? twB : twA;
                    u32* nxt = (j & 1) ? twA : twB;
                    if (i > 0) { if (FIRST && i == 1) tw_request<RB, true>(a, 0, tile_base, nxt); else tw_request<RB, false>(a, bp_of(i - 1), tile_base, nxt); }
                    __builtin_amdgcn_sched_barrier(0);   // keep the requests ahead of this round's butterflies
                    if (FIRST && i == 0) round_full<RB, CB, false, true>(lds, 0, cur); else round_full<RB, CB, false, false>(lds, bp_of(i), cur);
                    __syncthreads();
                }
            }
            if (rem) rem_round13<RB, CB, NT, false>(lds, a, rem, B, tile_base);
        }

        if (INV) {
            // top layer fused into the store; the 1/N scale rides on it (one multiply per output instead of two)
            const u32 sc2 = a.scale << 1, tes2 = (a.scale ? m_mul(te, a.scale) : te) << 1;
#pragma unroll
            for (int it = 0; it < (int)(T13_HALF / 4) / NT; it++) {
                const u32 t = (threadIdx.x + it * NT) * 4;
                const u32 pa = pad13(t), pb = pad13(t + T13_HALF);
                u32 oa[CB][4], ob[CB][4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const Row<CB> x = lds[pa + i], y = lds[pb + i];
#pragma unroll
                    for (int k = 0; k < CB; k++) {
                        const u32 sm = m_add(x.c[k], y.c[k]), df = m_sub(x.c[k], y.c[k]);
                        oa[k][i] = sc2 ? m_mul_dbl(sm, sc2) : sm; ob[k][i] = m_mul_dbl(df, tes2);
                    }
                }
                const u32 ga = tile_base + goff(t), gb = tile_base + goff(t + T13_HALF);
#pragma unroll
                for (int k = 0; k < CB; k++) {
                    if (live[k]) {
                        gstore4(d[k] + ga, make_uint4(oa[k][0], oa[k][1], oa[k][2], oa[k][3]));
                        gstore4(d[k] + gb, make_uint4(ob[k][0], ob[k][1], ob[k][2], ob[k][3]));
                    }
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < (int)(T13_ROWS / 4) / NT; it++) {
                const u32 t = (threadIdx.x + it * NT) * 4;
                const u32 p = pad13(t);
                const Row<CB> x0 = lds[p], x1 = lds[p + 1], x2 = lds[p + 2], x3 = lds[p + 3];
                const u32 g = tile_base + goff(t);
#pragma unroll
                for (int k = 0; k < CB; k++)
                    if (live[k]) gstore4(d[k] + g, make_uint4(x0.c[k], x1.c[k], x2.c[k], x3.c[k]));
            }
        }
    }
}

// ---- the fused middle of an LDE with blow-up 2 ---------------------------------------------------------------------------------
// The inverse transform's LAST pass and the forward transform's FIRST pass have the same tile geometry (layers [lo, lo+K) with
// lo + K = n; the forward one runs on 2^(n+1) points whose index bit n selects a replica of the coefficient tile).  One block
// therefore does both: 9 inverse layers on the tile, the 1/N scale, the coefficients out to the column (the commitment scheme keeps
// them for the OODS evaluation), then — from registers, without a trip through memory — the forward edge layer of BOTH replicas
// into two LDS tiles, their remaining layers, and the two output tiles.  Against the separate passes this saves one read of the
// coefficients and two of the four block prologues/epilogues per tile; LDS: two single-column tiles (68 KB, 2 blocks per CU).
struct PassMid {
    ColSet cols, out;            // evaluations in / coefficients out (in place, 2^n words); LDE out (2^(n+1) words)
    const u32* itw; const u32* tw;   // DOUBLED inverse / forward twiddle tables
    u32 tw_log;
    int n, lo, K, B;
    u32 scale, n_cols, tiles;    // tiles = 2^(n - 13); one column per block (n_groups == n_cols)
};

template <int RB, int KT>
__global__ __launch_bounds__(T13_ROWS >> RB, 1) void lde_mid_kernel(PassMid m) {
    constexpr int CB = 1;
    constexpr int NT = T13_ROWS >> RB;
    constexpr int MAXR = (T13_S - 1 + RB - 1) / RB;
    constexpr u32 TILE_ROWS_PADDED = T13_ROWS + (T13_ROWS >> 4);
    extern __shared__ __attribute__((aligned(16))) u32 lds13[];
    Row<CB>* ldsA = reinterpret_cast<Row<CB>*>(lds13);
    Row<CB>* ldsB = ldsA + TILE_ROWS_PADDED;

    const int K = KT ? KT : m.K, B = T13_S - K, lo = m.lo, lb = lo - B;
    const u32 maskB = (1u << B) - 1;
    auto goff = [&](u32 t) -> u32 { return ((t >> B) << lo) + (t & maskB); };
    const int K1 = K - 1, nfull = K1 / RB, rem = K1 % RB;
    auto bp_of = [&](int i) -> int { return B + rem + RB * i; };
    // descriptors for the shared round helpers (they read tw, tw_log, n, lo, B)
    Pass13 ai; ai.tw = m.itw; ai.tw_log = m.tw_log; ai.n = m.n; ai.log_in = m.n; ai.lo = lo; ai.K = K; ai.B = B;
    ai.scale = 0; ai.n_cols = m.n_cols; ai.n_groups = m.n_cols; ai.tiles = m.tiles; ai.rep_log = 0;
    Pass13 af = ai; af.tw = m.tw; af.n = m.n + 1;

    u32 tile, col;
    {
        const u32 b = blockIdx.x;
        if (m.tiles >= 8) { const u32 xcd = b & 7, y = b >> 3; col = y % m.n_cols; tile = xcd * (m.tiles >> 3) + y / m.n_cols; }
        else { col = b % m.n_cols; tile = b / m.n_cols; }
    }
    const u32 lowblock = tile & ((1u << lb) - 1), high = tile >> lb;
    const u32 tile_base = (high << (lo + K)) | (lowblock << B);
    u32* __restrict__ cc = m.cols.col(col);
    u32* __restrict__ oc = m.out.col(col);

    constexpr int IT = (int)(T13_ROWS / 4) / NT;
    uint4 xa[IT];
#pragma unroll
    for (int it = 0; it < IT; it++) xa[it] = gload4(cc + tile_base + goff((threadIdx.x + it * NT) * 4));
    u32 twA[1 << RB], twB[1 << RB];
    if (nfull) tw_request<RB, false>(ai, bp_of(0), tile_base, twA);
    const int le = lo + K - 1;
    const u32 tei2 = m.itw[((1u << m.tw_log) - (1u << (m.n - le))) + high], tei = tei2 >> 1;
    // forward geometry of replica r: tile index r * tiles + tile in the 2^(n+1) index space
    u32 fbase[2], tef2[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const u32 tf = (u32)r * m.tiles + tile, lbk = tf & ((1u << lb) - 1), hf = tf >> lb;
        fbase[r] = (hf << (lo + K)) | (lbk << B);
        tef2[r] = m.tw[((1u << m.tw_log) - (1u << (m.n + 1 - le))) + hf];
    }
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const u32 p = pad13((threadIdx.x + it * NT) * 4);
#pragma unroll
        for (int i = 0; i < 4; i++) { Row<CB> r; r.c[0] = get4(xa[it], i); ldsA[p + i] = r; }
    }
    __syncthreads();

    // ---- inverse layers [lo, lo + K - 1)
    if (rem) rem_round13<RB, CB, NT, true>(ldsA, ai, rem, B, tile_base);
#pragma unroll
    for (int j = 0; j < MAXR; j++) {
        if (j < nfull) {
            u32* cur = (j & 1) ? twB : twA;
            u32* nxt = (j & 1) ? twA : twB;
            if (j + 1 < nfull) tw_request<RB, false>(ai, bp_of(j + 1), tile_base, nxt);
            __builtin_amdgcn_sched_barrier(0);
            round_full<RB, CB, true, false>(ldsA, bp_of(j), cur);
            __syncthreads();
        }
    }
    // forward step s = replica * nfull + j runs round i = nfull - 1 - j; step s uses (s odd ? twB : twA)
    if (nfull) tw_request<RB, false>(af, bp_of(nfull - 1), fbase[0], twA);
    __builtin_amdgcn_sched_barrier(0);

    // ---- inverse top layer + scale -> coefficients (to the column); forward edge layer of both replicas -> the two LDS tiles
    {
        const u32 sc2 = m.scale << 1, tes2 = m_mul(tei, m.scale) << 1;
#pragma unroll
        for (int it = 0; it < (int)(T13_HALF / 4) / NT; it++) {
            const u32 t = (threadIdx.x + it * NT) * 4;
            const u32 pa = pad13(t), pb = pad13(t + T13_HALF);
            u32 ca[4], cb[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const u32 x = ldsA[pa + i].c[0], y = ldsA[pb + i].c[0];
                ca[i] = m_mul_dbl(m_add(x, y), sc2); cb[i] = m_mul_dbl(m_sub(x, y), tes2);
            }
            gstore4(cc + tile_base + goff(t), make_uint4(ca[0], ca[1], ca[2], ca[3]));
            gstore4(cc + tile_base + goff(t + T13_HALF), make_uint4(cb[0], cb[1], cb[2], cb[3]));
#pragma unroll
            for (int i = 0; i < 4; i++) {
                u32 u0 = ca[i], w0 = cb[i], u1 = ca[i], w1 = cb[i];
                bfly13<false>(u0, w0, tef2[0], false);
                bfly13<false>(u1, w1, tef2[1], false);
                Row<CB> r;
                r.c[0] = u0; ldsA[pa + i] = r; r.c[0] = w0; ldsA[pb + i] = r;
                r.c[0] = u1; ldsB[pa + i] = r; r.c[0] = w1; ldsB[pb + i] = r;
            }
        }
    }
    __syncthreads();

    // ---- forward layers of the two replicas, then their tiles out
#pragma unroll
    for (int r = 0; r < 2; r++) {
        Row<CB>* lds = r ? ldsB : ldsA;
#pragma unroll
        for (int j = 0; j < MAXR; j++) {
            if (j < nfull) {
                const int i = nfull - 1 - j;
                const int sidx = r * nfull + j;
                u32* cur = (sidx & 1) ? twB : twA;
                u32* nxt = (sidx & 1) ? twA : twB;
                if (i > 0) tw_request<RB, false>(af, bp_of(i - 1), fbase[r], nxt);
                else if (r == 0) tw_request<RB, false>(af, bp_of(nfull - 1), fbase[1], nxt);
                __builtin_amdgcn_sched_barrier(0);
                round_full<RB, CB, false, false>(lds, bp_of(i), cur);
                __syncthreads();
            }
        }
        if (rem) rem_round13<RB, CB, NT, false>(lds, af, rem, B, fbase[r]);
#pragma unroll
        for (int it = 0; it < (int)(T13_ROWS / 4) / NT; it++) {
            const u32 t = (threadIdx.x + it * NT) * 4;
            const u32 p = pad13(t);
            const Row<CB> x0 = lds[p], x1 = lds[p + 1], x2 = lds[p + 2], x3 = lds[p + 3];
            gstore4(oc + fbase[r] + goff(t), make_uint4(x0.c[0], x1.c[0], x2.c[0], x3.c[0]));
        }
    }
}

// ---- planning: layers [0, m) -> the FIRST pass [0, 13) plus passes of <= kmax layers (runs of 2^(13-kmax) words) ----
struct Plan13 { int lo, K, B; };
static std::vector<Plan13> plan13(int m, int kmax) {
    std::vector<Plan13> p;
    p.push_back({0, T13_S, 0});
    int rest = m - T13_S;
    if (rest <= 0) return p;
    int nhi = (rest + kmax - 1) / kmax, lo = T13_S;
    for (int i = 0; i < nhi; i++) {
        int k = rest / nhi + (i < rest % nhi ? 1 : 0);
        p.push_back({lo, k, T13_S - k});
        lo += k;
    }
    return p;
}

// one column per block (CB = 1: 34 KB tiles, 58-70 VGPRs, 3-4 blocks per CU) measured 3 % faster than column pairs (CB = 2, b64 LDS, 2 blocks per
// CU); radix-8 rounds (RB = 3) slower in every shape: neither is instantiated any more (the kernel stays generic in CB / RB)
template <bool INV, bool FIRST, int CB, int RB, int KT>
static int launch13_t(nx_ctx* ctx, const Pass13& a) {
    static std::atomic<uint64_t> attr_set{0};   // one bit per device: function attributes are per device
    const size_t lds_bytes = ((size_t)T13_ROWS + (T13_ROWS >> 4)) * 4 * CB;
    if (!(attr_set.load() & (1ull << (ctx->device & 63)))) {
        NX_HIP(ctx, hipFuncSetAttribute((const void*)fft13_kernel<INV, FIRST, CB, RB, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.fetch_or(1ull << (ctx->device & 63));
    }
    dim3 grid(a.tiles * a.n_groups), block(T13_ROWS >> RB);
    hipLaunchKernelGGL((fft13_kernel<INV, FIRST, CB, RB, KT>), grid, block, lds_bytes, ctx->cur, a);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

template <bool INV, int CB, int RB>
static int launch13_k(nx_ctx* ctx, bool first, const Pass13& a) {
    if (first) return launch13_t<INV, true, CB, RB, 0>(ctx, a);
    switch (a.K) {
    case 1: return launch13_t<INV, false, CB, RB, 1>(ctx, a);
    case 2: return launch13_t<INV, false, CB, RB, 2>(ctx, a);
    case 3: return launch13_t<INV, false, CB, RB, 3>(ctx, a);
    case 4: return launch13_t<INV, false, CB, RB, 4>(ctx, a);
    case 5: return launch13_t<INV, false, CB, RB, 5>(ctx, a);
    case 6: return launch13_t<INV, false, CB, RB, 6>(ctx, a);
    case 7: return launch13_t<INV, false, CB, RB, 7>(ctx, a);
    case 8: return launch13_t<INV, false, CB, RB, 8>(ctx, a);
    case 9: return launch13_t<INV, false, CB, RB, 9>(ctx, a);
    default: return launch13_t<INV, false, CB, RB, 0>(ctx, a);   // NX_FFT_KMAX > 9
    }
}

static int launch13(nx_ctx* ctx, bool inv, bool first, Pass13 a) {
    a.n_groups = a.n_cols;
    return inv ? launch13_k<true, 1, 4>(ctx, first, a) : launch13_k<false, 1, 4>(ctx, first, a);
}

// in-place iFFT of n_cols columns of 2^n words (n >= 13)
int fft13_interpolate(nx_ctx* ctx, const nx_twiddles* tw, ColSet cols, u32 n_cols, int n) {
    std::vector<Plan13> plan = plan13(n, ctx->opt.fft_kmax);
    for (size_t i = 0; i < plan.size(); i++) {
        Pass13 a; a.src = cols; a.dst = cols; a.tw = tw->d_itw2; a.tw_log = tw->log_half; a.n = n; a.log_in = n;
        a.lo = plan[i].lo; a.K = plan[i].K; a.B = plan[i].B; a.scale = i + 1 == plan.size() ? m_inv(1u << n) : 0;
        a.n_cols = n_cols; a.n_groups = 0; a.tiles = 1u << (n - T13_S); a.rep_log = 0;
        NX_TRY(launch13(ctx, true, i == 0, a));
    }
    return NX_OK;
}

// FFT of 2^log_in coefficients per column onto 2^n points (n >= log_in >= 13); out may alias polys when n == log_in
int fft13_evaluate(nx_ctx* ctx, const nx_twiddles* tw, ColSet polys, u32 n_cols, int log_in, int n, ColSet out) {
    std::vector<Plan13> plan = plan13(log_in, ctx->opt.fft_kmax);
    for (size_t k = plan.size(); k-- > 0;) {
        const bool top = k + 1 == plan.size();
        Pass13 a; a.src = top ? polys : out; a.dst = out; a.tw = tw->d_tw2; a.tw_log = tw->log_half; a.n = n; a.log_in = top ? log_in : n;
        a.lo = plan[k].lo; a.K = plan[k].K; a.B = plan[k].B; a.scale = 0;
        a.n_cols = n_cols; a.n_groups = 0; a.tiles = 1u << (n - T13_S); a.rep_log = top ? (u32)(n - log_in) : 0;
        NX_TRY(launch13(ctx, false, k == 0, a));
    }
    return NX_OK;
}

template <int RB, int KT>
static int launch_mid_t(nx_ctx* ctx, const PassMid& m) {
    static std::atomic<uint64_t> attr_set{0};   // one bit per device: function attributes are per device
    const size_t lds_bytes = ((size_t)T13_ROWS + (T13_ROWS >> 4)) * 4 * 2;
    if (!(attr_set.load() & (1ull << (ctx->device & 63)))) {
        NX_HIP(ctx, hipFuncSetAttribute((const void*)lde_mid_kernel<RB, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.fetch_or(1ull << (ctx->device & 63));
    }
    hipLaunchKernelGGL((lde_mid_kernel<RB, KT>), dim3(m.tiles * m.n_cols), dim3(T13_ROWS >> RB), lds_bytes, ctx->cur, m);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}
static int launch_mid(nx_ctx* ctx, const PassMid& m) {
    switch (m.K) {
    case 1: return launch_mid_t<4, 1>(ctx, m);
    case 2: return launch_mid_t<4, 2>(ctx, m);
    case 3: return launch_mid_t<4, 3>(ctx, m);
    case 4: return launch_mid_t<4, 4>(ctx, m);
    case 5: return launch_mid_t<4, 5>(ctx, m);
    case 6: return launch_mid_t<4, 6>(ctx, m);
    case 7: return launch_mid_t<4, 7>(ctx, m);
    case 8: return launch_mid_t<4, 8>(ctx, m);
    case 9: return launch_mid_t<4, 9>(ctx, m);
    default: return launch_mid_t<4, 0>(ctx, m);
    }
}

// iFFT in place (coefficients stay in `cols`) + FFT onto 2^(n+1) points in `out`, n >= 14: every pass as in fft13_interpolate /
// fft13_evaluate except the inverse transform's last pass and the forward transform's first pass, which are one launch.
int fft13_lde(nx_ctx* ctx, const nx_twiddles* tw, ColSet cols, u32 n_cols, int n, ColSet out) {
    std::vector<Plan13> plan = plan13(n, ctx->opt.fft_kmax);
    const size_t P = plan.size();
    if (P < 2) return set_err(ctx, NX_ERR_ARG, "fft13_lde: needs at least two passes");
    for (size_t i = 0; i + 1 < P; i++) {
        Pass13 a; a.src = cols; a.dst = cols; a.tw = tw->d_itw2; a.tw_log = tw->log_half; a.n = n; a.log_in = n;
        a.lo = plan[i].lo; a.K = plan[i].K; a.B = plan[i].B; a.scale = 0;
        a.n_cols = n_cols; a.n_groups = 0; a.tiles = 1u << (n - T13_S); a.rep_log = 0;
        NX_TRY(launch13(ctx, true, i == 0, a));
    }
    {
        PassMid m; m.cols = cols; m.out = out; m.itw = tw->d_itw2; m.tw = tw->d_tw2; m.tw_log = tw->log_half; m.n = n;
        m.lo = plan[P - 1].lo; m.K = plan[P - 1].K; m.B = plan[P - 1].B; m.scale = m_inv(1u << n); m.n_cols = n_cols; m.tiles = 1u << (n - T13_S);
        NX_TRY(launch_mid(ctx, m));
    }
    for (size_t k = P - 1; k-- > 0;) {
        Pass13 a; a.src = out; a.dst = out; a.tw = tw->d_tw2; a.tw_log = tw->log_half; a.n = n + 1; a.log_in = n + 1;
        a.lo = plan[k].lo; a.K = plan[k].K; a.B = plan[k].B; a.scale = 0;
        a.n_cols = n_cols; a.n_groups = 0; a.tiles = 1u << (n + 1 - T13_S); a.rep_log = 0;
        NX_TRY(launch13(ctx, false, k == 0, a));
    }
    return NX_OK;
}

}  // namespace nx
