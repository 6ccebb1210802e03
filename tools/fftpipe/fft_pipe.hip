// Pipelined Circle-FFT kernels for the LDE of columns of 2^17 .. 2^22 rows (K3/K4 of SURVEY.md §8(a); reference call sites
// prover/src/machine.rs:209-263 via TreeBuilder::extend_evals / commit, Stwo PolyOps::interpolate_columns + evaluate_polynomials).
//
// Why a second schedule next to fft13.hip: there one block = one tile, all ~1000 resident blocks of a launch load, transform and
// store in lock step (profiles/r02_fft_sq_counters.json: VALU busy 44-60 %, memory idle while the butterflies run).  Here a block
// is PERSISTENT and software-pipelined over its work items:
//     iteration k:   [tile k+1: LDS-DMA in flight ......................................]
//                    [tile k: butterfly rounds in LDS, fused store]   [tile k-1: stores draining]
//  * two LDS tile buffers per block, two blocks per CU (4 waves per SIMD, <= 128 VGPRs — the registers pay for whole-tile twiddle
//    sets and hoisted LDS addresses);
//  * global -> LDS by global_load_lds_dwordx4 (no staging registers, no ds_write); the XOR swizzle of fft_pipe.cuh is applied to the
//    GLOBAL address each lane fetches, so the landed tile is bank-conflict free for every round;
//  * vmcnt is an in-order counter per wave and hipcc's bookkeeping of it is conservative around loops (it answered vmcnt(0) — i.e.
//    "wait for the prefetch" — for every register load of a first version), so NO load of these kernels has a register
//    destination: tiles AND twiddles arrive by LDS-DMA (the FIRST passes keep the tile's twiddle slices in LDS "slabs", re-filled
//    for tile k+1 as soon as the round that reads them is over; the middle launch's twiddles are tile independent and loaded once;
//    wave-uniform twiddles come through the scalar cache).  Every DMA of an iteration is issued BEFORE the iteration's four
//    global stores, so "everything for tile k+1 has landed" is exactly `s_waitcnt vmcnt(4)` at the top of iteration k+1;
//  * block barriers are bare s_barrier + lgkmcnt(0): __syncthreads() would drain vmcnt (the DMA writes LDS) and with it the prefetch.
// The lane-level code is shared with the CPU replay (tests/fftpipe_emul.cpp); results are bit-identical to fft13.hip.
//
// STATUS (round 3, measured on MI355X: profiles/r03_fft_pipe_*): correct at every size, LDS bank conflicts 0 (SQ_LDS_BANK_CONFLICT),
// the prefetch hides the loads — and the LDE is still 7-9 % SLOWER than fft13.hip at 2^22 rows (64 columns: 3.33 ms vs 2.93-3.2).
// Ablations of the same build: no global traffic 2.72 ms, no butterfly arithmetic 2.36 ms, neither (LDS round trips + barriers +
// addressing only) 1.5 ms: the LDS instruction stream (8.5 k wave-instructions per column and CU, ~4 cycles each) and the VALU work
// (~35 us per column and CU at the measured instruction rates) do not overlap with only 4 waves per SIMD — which is what the two
// 80-KB blocks per CU allow — and the iteration's four store instructions block their waves while they are issued.  The schedule
// therefore stays OPT-IN (context option "fft.pipe", environment default NX_FFT_PIPE=1); DESIGN.md §6 item 17 has the numbers.
#include "internal.h"
#include "fft_pipe.cuh"
#include <atomic>
#include <algorithm>
#include <stdlib.h>

namespace nx {
using namespace pipe;

#define NX_LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ void blk_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Between two rounds that touch only the wave's own rows (a wave of 64 lanes x 16 rows owns rows [1024 w, 1024 w + 1024) in every round
// at tile bits < 10): the wave's LDS writes have landed, no other wave is involved.  Waves of a block then drift apart, so their LDS
// and VALU phases overlap instead of hitting each pipe together.
__device__ __forceinline__ void wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// One LDS-DMA instruction: every active lane fetches the 16 bytes at base + byte_off; lane l's land at LDS byte address lds_addr + 16 l.
// Inline asm on purpose: hipcc treats the builtin as "may write any LDS" and puts s_waitcnt vmcnt(0) in front of the next ds_read —
// i.e. it waits for the prefetch just issued.  M0 (the LDS base) is compiler-reserved: saved and restored inside the statement; the
// leading s_nop covers a base SGPR fresh from v_readfirstlane (guide §5.7).  Completion is counted by hand (wait_prefetch below).
__device__ __forceinline__ void dma16(const u32* base, u32 byte_off, u32 lds_addr) {
#if NX_PIPE_ABL & 2
    asm volatile("" ::"v"(byte_off), "s"(base), "s"(lds_addr));
    return;
#endif
    u32 keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(byte_off), "s"(base), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ u32 lds_addr_of(const u32* p) { return (u32)(uintptr_t)(NX_LDS_AS const u32*)p; }

// LDS-DMA of one tile: the lane that fills slot s fetches the 4-word group swz(s).  `wv` = wave index (an SGPR: the LDS base of a
// DMA instruction is wave-uniform).
template <class Goff>
__device__ __forceinline__ void dma_tile(u32* lds_buf, const u32* src, u32 tid, u32 wv, Goff goff) {
    const u32 l0 = lds_addr_of(lds_buf) + wv * 1024u;
#pragma unroll
    for (int i = 0; i < 4; i++) dma16(src, goff(4 * swz((u32)i * NT + tid)) * 4u, l0 + (u32)i * (NT * 16u));
}
// LDS-DMA of a twiddle slab (fft_pipe.cuh): layers L0..L1 of the doubled table, the tile's slices one after the other; the 1-KiB
// chunks are dealt to the block's 8 waves, lanes beyond a short slice stay idle.
template <int L0, int L1, int L = L0, int C0 = 0>
__device__ __forceinline__ void dma_slab(u32* slab, const u32* tbl2, u32 tw_log, int n, u32 tile, u32 lane, u32 wv) {
    if constexpr (L <= L1) {
        constexpr u32 cnt = 1u << (12 - L);
        constexpr int nch = cnt >= 256 ? (int)(cnt / 256) : 1;
        const u32* src = tbl2 + lvl_off(tw_log, n, L) + tile * cnt;
        const u32 dst = lds_addr_of(slab + slab_off(L0, L));
#pragma unroll
        for (int j = 0; j < nch; j++)
            if (wv == (u32)((C0 + j) & 7) && (cnt >= 256 || lane * 4 < cnt)) dma16(src + j * 256, lane * 16u, dst + (u32)j * 1024u);
        dma_slab<L0, L1, L + 1, C0 + nch>(slab, tbl2, tw_log, n, tile, lane, wv);
    }
}
// "every DMA this wave issued before its last 4 (store) instructions has landed" / "everything has landed"
#if NX_PIPE_ABL & 2
__device__ __forceinline__ void wait_prefetch() { asm volatile("" ::: "memory"); }
#else
__device__ __forceinline__ void wait_prefetch() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
#endif
__device__ __forceinline__ void wait_all_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// keeps the LDS addresses of a round from being hoisted out of the item loop and held in registers for the whole kernel
__device__ __forceinline__ u32 opaque(u32 v) { asm volatile("" : "+v"(v)); return v; }
#ifdef NX_PIPE_HOIST   // A/B: let the FIRST passes keep their LDS addresses in registers across items (93 / 112 VGPRs)
__device__ __forceinline__ u32 opaque1(u32 v) { return v; }
#else
__device__ __forceinline__ u32 opaque1(u32 v) { return opaque(v); }
#endif
__device__ __forceinline__ u32* col_ptr(const ColSet& c, u32 col) { return c.table ? ldsp(c.table + col) : c.base + (uint64_t)col * c.stride; }

struct PipeFirst {
    ColSet cols;       // transformed in place
    const u32* tw2;    // DOUBLED twiddle table (inverse for the inverse pass)
    u32 tw_log;
    int n;             // log size of the transform
    u32 n_cols, tiles, n_items;
};

// ---- inverse FIRST pass: layers [0, 13) of the iFFT on contiguous tiles ------------------------------------------------------------
// LDS (words): tile buffers [0, 8192) [8192, 16384); slab A = layers 1..3 (the circle round derives layer 0 from layer 1), slab B =
// layers 4..7: 80 KB per block, two blocks per CU.  Layers 8..11 have wave-uniform twiddles (scalar loads), layer 12 is fused
// into the store.
constexpr u32 P1_SA = 2 * T_ROWS, P1_SB = P1_SA + slab_words(1, 3), P1_WORDS = P1_SB + slab_words(4, 7);
static_assert(P1_WORDS * 4 <= 80 * 1024, "two blocks per CU");

__global__ __launch_bounds__(NT, 4) void pipe_ifirst_kernel(PipeFirst a) {
    extern __shared__ __attribute__((aligned(16))) u32 plds[];
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 utid = __builtin_amdgcn_readfirstlane(tid & ~255u);   // the round at bit 8 sees tid >> 8 only: wave-uniform twiddles
    u32 item = blockIdx.x;
    if (item >= a.n_items) return;
    const u32 stride = gridDim.x;
    auto ident = [](u32 t) { return t; };
    u32* SA = plds + P1_SA;
    u32* SB = plds + P1_SB;
    Item it = decode_item(item, a.tiles, a.n_cols);
    u32* base = col_ptr(a.cols, it.col) + ((size_t)it.tile << T_S);
    dma_tile(plds, base, tid, wv, ident);
    dma_slab<1, 3>(SA, a.tw2, a.tw_log, a.n, it.tile, lane, wv);
    dma_slab<4, 7>(SB, a.tw2, a.tw_log, a.n, it.tile, lane, wv);
    u32 twC[16];
    first_tw_fetch<8, 4, false, MEM_SCALAR>(a.tw2, a.tw_log, a.n, it.tile, utid, twC);
    u32 te2 = lds1(a.tw2 + lvl_off(a.tw_log, a.n, 12) + it.tile);
    wait_all_vmem();
    u32 cur = 0;
    for (;;) {
        const u32 nxt = item + stride;
        const bool has_next = nxt < a.n_items;
        u32* X = plds + cur * T_ROWS;
        wait_prefetch();         // this wave's DMAs for tile k have landed (its stores of tile k-1 may still be in flight) ...
        blk_barrier();           // ... and everybody's; every lane is past the previous tile's LDS reads
        Item nit = it;
        u32* nbase = base;
        if (has_next) {
            nit = decode_item(nxt, a.tiles, a.n_cols);
            nbase = col_ptr(a.cols, nit.col) + ((size_t)nit.tile << T_S);
            dma_tile(plds + (cur ^ 1u) * T_ROWS, nbase, tid, wv, ident);
        }
        u32 tw[16];
        { const u32 t = opaque1(tid); slab_tw_fetch<0, 4, true>(SA, t, tw); round16_low<true, true>(X, t, tw); }
        wave_sync();             // bits [0, 8) stay inside the wave's 1024 rows
        { const u32 t = opaque1(tid); slab_tw_fetch<4, 4, false>(SB, t, tw); round16<4, 4, true>(X, t, tw); }
        blk_barrier();
        if (has_next) { dma_slab<1, 3>(SA, a.tw2, a.tw_log, a.n, nit.tile, lane, wv); dma_slab<4, 7>(SB, a.tw2, a.tw_log, a.n, nit.tile, lane, wv); }
        round16<8, 4, true>(X, opaque1(tid), twC);
        blk_barrier();
        if (has_next) first_tw_fetch<8, 4, false, MEM_SCALAR>(a.tw2, a.tw_log, a.n, nit.tile, utid, twC);
        ifirst_store(X, opaque1(tid), base, te2);     // the iteration's 4 global stores
        if (!has_next) break;
        te2 = lds1(a.tw2 + lvl_off(a.tw_log, a.n, 12) + nit.tile);
        item = nxt; it = nit; base = nbase; cur ^= 1u;
    }
}

// ---- forward FIRST pass: layers [13, 0) of the FFT on contiguous tiles -------------------------------------------------------------
// LDS (words): two tile buffers; slab A = layers 2..4, slab 1 = layer 1 (both fused store layers read it: the circle layer's twiddles
// are derived from it), slab B = layers 5..8.  Layers 9..12 have tile-uniform twiddles (scalar loads).
constexpr u32 P3_SA = 2 * T_ROWS, P3_S1 = P3_SA + slab_words(2, 4), P3_SB = P3_S1 + slab_words(1, 1), P3_WORDS = P3_SB + slab_words(5, 8);
static_assert(P3_WORDS * 4 <= 80 * 1024, "two blocks per CU");

__global__ __launch_bounds__(NT, 4) void pipe_ffirst_kernel(PipeFirst a) {
    extern __shared__ __attribute__((aligned(16))) u32 plds[];
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    u32 item = blockIdx.x;
    if (item >= a.n_items) return;
    const u32 stride = gridDim.x;
    auto ident = [](u32 t) { return t; };
    u32* SA = plds + P3_SA;
    u32* S1 = plds + P3_S1;
    u32* SB = plds + P3_SB;
    Item it = decode_item(item, a.tiles, a.n_cols);
    u32* base = col_ptr(a.cols, it.col) + ((size_t)it.tile << T_S);
    dma_tile(plds, base, tid, wv, ident);
    dma_slab<5, 8>(SB, a.tw2, a.tw_log, a.n, it.tile, lane, wv);
    dma_slab<2, 4>(SA, a.tw2, a.tw_log, a.n, it.tile, lane, wv);
    dma_slab<1, 1>(S1, a.tw2, a.tw_log, a.n, it.tile, lane, wv);
    u32 twC[16];
    first_tw_fetch<9, 4, false, MEM_SCALAR>(a.tw2, a.tw_log, a.n, it.tile, 0, twC);   // tile-uniform
    wait_all_vmem();
    u32 cur = 0;
    for (;;) {
        const u32 nxt = item + stride;
        const bool has_next = nxt < a.n_items;
        u32* X = plds + cur * T_ROWS;
        wait_prefetch();
        blk_barrier();
        Item nit = it;
        u32* nbase = base;
        if (has_next) {
            nit = decode_item(nxt, a.tiles, a.n_cols);
            nbase = col_ptr(a.cols, nit.col) + ((size_t)nit.tile << T_S);
            dma_tile(plds + (cur ^ 1u) * T_ROWS, nbase, tid, wv, ident);
        }
        round16<9, 4, false>(X, opaque1(tid), twC);
        blk_barrier();
        if (has_next) first_tw_fetch<9, 4, false, MEM_SCALAR>(a.tw2, a.tw_log, a.n, nit.tile, 0, twC);
        u32 tw[16];
        { const u32 t = opaque1(tid); slab_tw_fetch<5, 4, false>(SB, t, tw); round16<5, 4, false>(X, t, tw); }
        wave_sync();             // bits [0, 9) stay inside the wave's 1024 rows: no block barrier until the slabs are re-filled
        { const u32 t = opaque1(tid); slab_tw_fetch<2, 3, false>(SA, t, tw); round16<2, 3, false>(X, t, tw); }
        wave_sync();
        uint4 x[4];
        u32 tws[8];
        ffirst_store_load(X, S1, opaque1(tid), x, tws);
        blk_barrier();           // every wave is done with the three slabs: they may be re-filled
        if (has_next) {
            dma_slab<5, 8>(SB, a.tw2, a.tw_log, a.n, nit.tile, lane, wv);
            dma_slab<2, 4>(SA, a.tw2, a.tw_log, a.n, nit.tile, lane, wv);
            dma_slab<1, 1>(S1, a.tw2, a.tw_log, a.n, nit.tile, lane, wv);
        }
        ffirst_store_finish(x, tws, tid, base, it.tile);     // the iteration's 4 global stores
        if (!has_next) break;
        item = nxt; it = nit; base = nbase; cur ^= 1u;
    }
}

// ---- the middle of an LDE with blow-up 2: inverse layers [13, n), 1/N, coefficients out, forward layers [n, 13) of both replicas ----
struct PipeMid {
    ColSet cols, out;            // evaluations in / coefficients out (in place, 2^n words); LDE out (2^(n+1) words)
    const u32* itw2; const u32* tw2;   // DOUBLED inverse / forward twiddle tables
    u32 tw_log;
    int n;
    u32 n_cols, n_items, scale;  // tiles = 2^(n - 13)
};

// the rounds of tile bits [B, 11) on buffer Z, a block barrier after each; twiddles from the block's LDS tables of one direction
// `last_is_wave_local`: the caller follows the forward rounds with a wave-local step (the tile store), so the barrier after the last
// round may be the wave's own when that round stayed inside the wave's rows.
template <int K, bool INV>
__device__ __forceinline__ void mid_rounds(u32* Z, const u32* dir_tw, u32 tid) {
    using P = MidPlan<K>;
    constexpr bool REM_LOCAL = P::B + 4 <= 10;     // the remainder round's 16 rows per lane span tile bits [B, B + 4): inside 1024 rows?
    u32 tw[16];
    auto rem = [&](bool barrier) {
        if constexpr (P::REM > 0) {
            const u32 t = opaque(tid); mid_tw_fetch<P::B, P::REM>(dir_tw, P::B, t, tw); round16<P::B, P::REM, INV>(Z, t, tw);
            if (barrier) blk_barrier(); else wave_sync();
        }
    };
    auto full = [&]() {
        if constexpr (P::NFULL > 0) { const u32 t = opaque(tid); mid_tw_fetch<P::BPF, 4>(dir_tw, P::B, t, tw); round16<P::BPF, 4, INV>(Z, t, tw); blk_barrier(); }
    };
    if constexpr (INV) { rem(true); full(); } else { full(); rem(!REM_LOCAL); }
}

// LDS (words): X [0, 8192), Y [8192, 16384), then the twiddle tables of the inverse direction and of the two forward replicas
// (2^K words each, tile independent, filled once per block).  Per iteration: 4 coefficient stores + 4 stores of replica 0, then the
// next tile's DMA into X, then the 4 stores of replica 1 — so `vmcnt(4)` at the top again means "the tile has landed".
template <int K>
__global__ __launch_bounds__(NT, 4) void pipe_mid_kernel(PipeMid m) {
    using P = MidPlan<K>;
    constexpr int B = P::B;
    constexpr u32 W = 1u << (T_S - B);
    extern __shared__ __attribute__((aligned(16))) u32 plds[];
    u32* X = plds;
    u32* Y = plds + T_ROWS;
    u32* LTW = plds + 2 * T_ROWS;
    const u32 tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    u32 item = blockIdx.x;
    if (item >= m.n_items) return;
    const u32 stride = gridDim.x, tiles = 1u << K;
    auto goff = [](u32 t) { return mid_goff(t, B); };
    Item it = decode_item(item, tiles, m.n_cols);
    u32* ct = col_ptr(m.cols, it.col) + ((size_t)it.tile << B);
    dma_tile(X, ct, tid, wv, goff);
    // the block's twiddle tables and the constants of the radix-4 hand-over
#pragma unroll
    for (int b = B; b < 11; b++) {
        const u32 cnt = 1u << (12 - b);
        if (tid < cnt) {
            LTW[mid_tw_off(B, b) + tid] = ldg1(m.itw2 + lvl_off(m.tw_log, m.n, b + K) + tid);
            LTW[W + mid_tw_off(B, b) + tid] = ldg1(m.tw2 + lvl_off(m.tw_log, m.n + 1, b + K) + tid);
            LTW[2 * W + mid_tw_off(B, b) + tid] = ldg1(m.tw2 + lvl_off(m.tw_log, m.n + 1, b + K) + cnt + tid);
        }
    }
    MidConsts k;
    k.i11[0] = lds1(m.itw2 + lvl_off(m.tw_log, m.n, 11 + K)); k.i11[1] = lds1(m.itw2 + lvl_off(m.tw_log, m.n, 11 + K) + 1);
    k.i12 = lds1(m.itw2 + lvl_off(m.tw_log, m.n, 12 + K));
#pragma unroll
    for (int r = 0; r < 2; r++) {
        k.f12[r] = lds1(m.tw2 + lvl_off(m.tw_log, m.n + 1, 12 + K) + r);
        k.f11[2 * r] = lds1(m.tw2 + lvl_off(m.tw_log, m.n + 1, 11 + K) + 2 * r);
        k.f11[2 * r + 1] = lds1(m.tw2 + lvl_off(m.tw_log, m.n + 1, 11 + K) + 2 * r + 1);
    }
    k.scale = m.scale;
    wait_all_vmem();
    for (;;) {
        const u32 nxt = item + stride;
        const bool has_next = nxt < m.n_items;
        u32* oc = col_ptr(m.out, it.col) + ((size_t)it.tile << B);
        wait_prefetch();
        blk_barrier();
        mid_rounds<K, true>(X, LTW, tid);
        mid_handover(X, Y, opaque(tid), k, ct, B);
        blk_barrier();
        mid_rounds<K, false>(X, LTW + W, tid);
        mid_store(X, opaque(tid), oc, B);
        blk_barrier();           // X is free: the next tile lands in it while replica 1 is transformed
        Item nit = it;
        if (has_next) {
            nit = decode_item(nxt, tiles, m.n_cols);
            ct = col_ptr(m.cols, nit.col) + ((size_t)nit.tile << B);
            dma_tile(X, ct, tid, wv, goff);
        }
        mid_rounds<K, false>(Y, LTW + 2 * W, tid);
        mid_store(Y, opaque(tid), oc + ((size_t)1 << m.n), B);     // the 4 stores behind the DMA
        if (!has_next) break;
        item = nxt; it = nit;
    }
}

// ======================================================================================================================================
// Tile kernels: the SAME lane-level rounds (conflict-free swizzled tile, b128 low round, radix-4 hand-over, wave-local syncs) under
// fft13.hip's schedule — one work item per block, loads staged through registers, twiddles from the global tables — so that 3-4
// blocks per CU (one 32-KB tile each, <= 64-80 VGPRs) overlap each other's phases instead of one block pipelining its own.
// ======================================================================================================================================
#ifndef NX_TILE_WAVES     // waves per SIMD the FIRST tile kernels are compiled for: 8 = 4 blocks per CU (<= 64 VGPRs), 6 = 3 blocks (<= 80)
#define NX_TILE_WAVES 6
#endif

// this wave's 1024 rows of the tile, coalesced (64 lanes x 16 B = 1 KiB per load), into their swizzled slots
__device__ __forceinline__ void tile_stage_load(const u32* base, u32 tid, uint4* x) {
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = ldg4(base + 4 * wave_group(tid, i));
}
__device__ __forceinline__ void tile_stage_write(u32* X, u32 tid, const uint4* x) {
#pragma unroll
    for (int i = 0; i < 4; i++) *reinterpret_cast<uint4*>(X + 4 * swz(wave_group(tid, i))) = x[i];
}

__global__ __launch_bounds__(NT, NX_TILE_WAVES) void tile_ifirst_kernel(PipeFirst a) {
    extern __shared__ __attribute__((aligned(16))) u32 plds[];
    u32* X = plds;
    const u32 tid = threadIdx.x;
    const u32 utid = __builtin_amdgcn_readfirstlane(tid & ~255u);
    const Item it = decode_item(blockIdx.x, a.tiles, a.n_cols);
    u32* base = col_ptr(a.cols, it.col) + ((size_t)it.tile << T_S);
    uint4 x[4];
    tile_stage_load(base, tid, x);
    u32 twA[16], twB[16], twC[16];
    first_tw_fetch<0, 4, true>(a.tw2, a.tw_log, a.n, it.tile, tid, twA);
    first_tw_fetch<4, 4, false>(a.tw2, a.tw_log, a.n, it.tile, tid, twB);
    const u32 te2 = lds1(a.tw2 + lvl_off(a.tw_log, a.n, 12) + it.tile);
    tile_stage_write(X, tid, x);
    wave_sync();                 // a wave stages, and then transforms, its own 1024 rows: bits [0, 8) need no block barrier
    round16_low<true, true>(X, tid, twA);
    wave_sync();
    round16<4, 4, true>(X, tid, twB);
    first_tw_fetch<8, 4, false, MEM_SCALAR>(a.tw2, a.tw_log, a.n, it.tile, utid, twC);
    blk_barrier();
    round16<8, 4, true>(X, tid, twC);
    blk_barrier();
    ifirst_store(X, tid, base, te2);
}

__global__ __launch_bounds__(NT, NX_TILE_WAVES) void tile_ffirst_kernel(PipeFirst a) {
    extern __shared__ __attribute__((aligned(16))) u32 plds[];
    u32* X = plds;
    const u32 tid = threadIdx.x;
    const Item it = decode_item(blockIdx.x, a.tiles, a.n_cols);
    u32* base = col_ptr(a.cols, it.col) + ((size_t)it.tile << T_S);
    uint4 x[4];
    tile_stage_load(base, tid, x);
    u32 twC[16], twB[16], twA[16];
    first_tw_fetch<9, 4, false, MEM_SCALAR>(a.tw2, a.tw_log, a.n, it.tile, 0, twC);
    first_tw_fetch<5, 4, false>(a.tw2, a.tw_log, a.n, it.tile, tid, twB);
    tile_stage_write(X, tid, x);
    blk_barrier();
    round16<9, 4, false>(X, tid, twC);
    first_tw_fetch<2, 3, false>(a.tw2, a.tw_log, a.n, it.tile, tid, twA);
    blk_barrier();
    round16<5, 4, false>(X, tid, twB);
    wave_sync();                 // bits [0, 9): the wave's own rows
    // the layer-1 pairs of the fused store come from the global table (the slab of the pipelined kernel is its LDS copy)
    u32 tws[8];
#pragma unroll
    for (int i = 0; i < 4; i++) ld_words<2, MEM_GLOBAL>(a.tw2 + lvl_off(a.tw_log, a.n, 1) + ((it.tile * T_GROUPS + wave_group(tid, i)) & ~1u), tws + 2 * i);
    round16<2, 3, false>(X, tid, twA);
    wave_sync();
    uint4 y[4];
#pragma unroll
    for (int i = 0; i < 4; i++) y[i] = *reinterpret_cast<const uint4*>(X + 4 * swz(wave_group(tid, i)));
    ffirst_store_finish(y, tws, tid, base, it.tile);
}

template <int K, bool INV>
__device__ __forceinline__ void tile_mid_rounds(u32* Z, const u32* tbl2, u32 tw_log, int n_dir, u32 r, u32 tid) {
    using P = MidPlan<K>;
    constexpr bool REM_LOCAL = P::B + 4 <= 10;
    u32 twr[16], twf[16];
    if constexpr (P::REM > 0) mid_tw_fetch_global<P::B, P::REM>(tbl2, tw_log, n_dir, K, r, tid, twr);
    if constexpr (P::NFULL > 0) mid_tw_fetch_global<P::BPF, 4>(tbl2, tw_log, n_dir, K, r, tid, twf);
    if constexpr (INV) {
        if constexpr (P::REM > 0) { round16<P::B, P::REM, true>(Z, tid, twr); blk_barrier(); }
        if constexpr (P::NFULL > 0) { round16<P::BPF, 4, true>(Z, tid, twf); blk_barrier(); }
    } else {
        if constexpr (P::NFULL > 0) { round16<P::BPF, 4, false>(Z, tid, twf); blk_barrier(); }
        if constexpr (P::REM > 0) { round16<P::B, P::REM, false>(Z, tid, twr); if (REM_LOCAL) wave_sync(); else blk_barrier(); }
    }
}

template <int K>
__global__ __launch_bounds__(NT, 4) void tile_mid_kernel(PipeMid m) {
    using P = MidPlan<K>;
    constexpr int B = P::B;
    extern __shared__ __attribute__((aligned(16))) u32 plds[];
    u32* X = plds;
    u32* Y = plds + T_ROWS;
    const u32 tid = threadIdx.x;
    const Item it = decode_item(blockIdx.x, 1u << K, m.n_cols);
    u32* ct = col_ptr(m.cols, it.col) + ((size_t)it.tile << B);
    u32* oc = col_ptr(m.out, it.col) + ((size_t)it.tile << B);
    uint4 x[4];
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = ldg4(ct + mid_goff(4 * ((u32)i * NT + tid), B));
    MidConsts k;
    k.i11[0] = lds1(m.itw2 + lvl_off(m.tw_log, m.n, 11 + K)); k.i11[1] = lds1(m.itw2 + lvl_off(m.tw_log, m.n, 11 + K) + 1);
    k.i12 = lds1(m.itw2 + lvl_off(m.tw_log, m.n, 12 + K));
#pragma unroll
    for (int r = 0; r < 2; r++) {
        k.f12[r] = lds1(m.tw2 + lvl_off(m.tw_log, m.n + 1, 12 + K) + r);
        k.f11[2 * r] = lds1(m.tw2 + lvl_off(m.tw_log, m.n + 1, 11 + K) + 2 * r);
        k.f11[2 * r + 1] = lds1(m.tw2 + lvl_off(m.tw_log, m.n + 1, 11 + K) + 2 * r + 1);
    }
    k.scale = m.scale;
#pragma unroll
    for (int i = 0; i < 4; i++) *reinterpret_cast<uint4*>(X + 4 * swz((u32)i * NT + tid)) = x[i];
    blk_barrier();
    tile_mid_rounds<K, true>(X, m.itw2, m.tw_log, m.n, 0, tid);
    mid_handover(X, Y, tid, k, ct, B);
    blk_barrier();
#pragma unroll
    for (int r = 0; r < 2; r++) {
        u32* Z = r ? Y : X;
        tile_mid_rounds<K, false>(Z, m.tw2, m.tw_log, m.n + 1, (u32)r, tid);
        mid_store(Z, tid, oc + ((size_t)r << m.n), B);
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------------
bool fft_pipe_supports(int n) { return n >= 17 && n <= 22; }

static u32 pipe_grid(const nx_ctx* ctx, u32 n_items) {
    const int want = ctx->opt.fft_pipe_grid > 0 ? ctx->opt.fft_pipe_grid : ctx->opt.fft_pipe_blocks_per_cu * std::max(1, ctx->n_cus);
    const u32 cap = (u32)std::max(8, want) & ~7u;
    return std::min(n_items, cap);   // n_items is a multiple of 8 (tiles >= 16): blocks stay on the XCD that owns their tiles
}

template <class KernelT>
static int pipe_set_lds(nx_ctx* ctx, KernelT kernel, std::atomic<uint64_t>& done) {
    if (!(done.load() & (1ull << (ctx->device & 63)))) {
        NX_HIP(ctx, hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        done.fetch_or(1ull << (ctx->device & 63));
    }
    return NX_OK;
}

static int launch_pipe_first(nx_ctx* ctx, bool inv, const PipeFirst& a) {
    static std::atomic<uint64_t> set_i{0}, set_f{0};
    const size_t lds_bytes = (size_t)(inv ? P1_WORDS : P3_WORDS) * 4;
    if (inv) {
        NX_TRY(pipe_set_lds(ctx, pipe_ifirst_kernel, set_i));
        hipLaunchKernelGGL(pipe_ifirst_kernel, dim3(pipe_grid(ctx, a.n_items)), dim3(NT), lds_bytes, ctx->cur, a);
    } else {
        NX_TRY(pipe_set_lds(ctx, pipe_ffirst_kernel, set_f));
        hipLaunchKernelGGL(pipe_ffirst_kernel, dim3(pipe_grid(ctx, a.n_items)), dim3(NT), lds_bytes, ctx->cur, a);
    }
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

template <int K>
static int launch_pipe_mid_t(nx_ctx* ctx, const PipeMid& m) {
    static std::atomic<uint64_t> set{0};
    const size_t lds_bytes = (2 * (size_t)T_ROWS + 3 * ((size_t)1 << (T_S - MidPlan<K>::B))) * 4;
    NX_TRY(pipe_set_lds(ctx, pipe_mid_kernel<K>, set));
    hipLaunchKernelGGL(pipe_mid_kernel<K>, dim3(pipe_grid(ctx, m.n_items)), dim3(NT), lds_bytes, ctx->cur, m);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

template <class KernelT, class ArgT>
static int launch_tile(nx_ctx* ctx, KernelT kernel, std::atomic<uint64_t>& set, const ArgT& a, u32 n_items, size_t lds_bytes) {
    NX_TRY(pipe_set_lds(ctx, kernel, set));
    hipLaunchKernelGGL(kernel, dim3(n_items), dim3(NT), lds_bytes, ctx->cur, a);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}
template <int K>
static int launch_tile_mid_t(nx_ctx* ctx, const PipeMid& m) {
    static std::atomic<uint64_t> set{0};
    return launch_tile(ctx, tile_mid_kernel<K>, set, m, m.n_items, 2 * (size_t)T_ROWS * 4);
}
// The LDE under the tile schedule (one item per block): iFFT FIRST, fused middle, FFT FIRST — three launches, 17 <= n <= 22.
int fft_tile_lde(nx_ctx* ctx, const nx_twiddles* tw, ColSet cols, u32 n_cols, int n, ColSet out) {
    if (!fft_pipe_supports(n)) return set_err(ctx, NX_ERR_ARG, "fft_tile_lde: 17 <= log_size <= 22");
    static std::atomic<uint64_t> set_i{0}, set_f{0};
    PipeFirst a; a.cols = cols; a.tw2 = tw->d_itw2; a.tw_log = tw->log_half; a.n = n; a.n_cols = n_cols; a.tiles = 1u << (n - T_S); a.n_items = a.tiles * n_cols;
    NX_TRY(launch_tile(ctx, tile_ifirst_kernel, set_i, a, a.n_items, (size_t)T_ROWS * 4));
    PipeMid m; m.cols = cols; m.out = out; m.itw2 = tw->d_itw2; m.tw2 = tw->d_tw2; m.tw_log = tw->log_half; m.n = n; m.n_cols = n_cols;
    m.n_items = n_cols << (n - T_S); m.scale = m_inv(1u << n);
    switch (n - T_S) {
    case 4: NX_TRY(launch_tile_mid_t<4>(ctx, m)); break;
    case 5: NX_TRY(launch_tile_mid_t<5>(ctx, m)); break;
    case 6: NX_TRY(launch_tile_mid_t<6>(ctx, m)); break;
    case 7: NX_TRY(launch_tile_mid_t<7>(ctx, m)); break;
    case 8: NX_TRY(launch_tile_mid_t<8>(ctx, m)); break;
    default: NX_TRY(launch_tile_mid_t<9>(ctx, m)); break;
    }
    PipeFirst f; f.cols = out; f.tw2 = tw->d_tw2; f.tw_log = tw->log_half; f.n = n + 1; f.n_cols = n_cols; f.tiles = 1u << (n + 1 - T_S); f.n_items = f.tiles * n_cols;
    return launch_tile(ctx, tile_ffirst_kernel, set_f, f, f.n_items, (size_t)T_ROWS * 4);
}

// inverse FIRST pass of a 2^n-point iFFT (n >= 17), in place (never the last pass: no 1/N)
int fft_pipe_ifirst(nx_ctx* ctx, const nx_twiddles* tw, ColSet cols, u32 n_cols, int n) {
    PipeFirst a; a.cols = cols; a.tw2 = tw->d_itw2; a.tw_log = tw->log_half; a.n = n; a.n_cols = n_cols; a.tiles = 1u << (n - T_S);
    a.n_items = a.tiles * n_cols;
    return launch_pipe_first(ctx, true, a);
}
// forward FIRST pass (the last launch) of a 2^n-point FFT, in place
int fft_pipe_ffirst(nx_ctx* ctx, const nx_twiddles* tw, ColSet cols, u32 n_cols, int n) {
    PipeFirst a; a.cols = cols; a.tw2 = tw->d_tw2; a.tw_log = tw->log_half; a.n = n; a.n_cols = n_cols; a.tiles = 1u << (n - T_S);
    a.n_items = a.tiles * n_cols;
    return launch_pipe_first(ctx, false, a);
}

// iFFT in place (coefficients stay in `cols`) + FFT onto 2^(n+1) points in `out`, 17 <= n <= 22: three launches.
int fft_pipe_lde(nx_ctx* ctx, const nx_twiddles* tw, ColSet cols, u32 n_cols, int n, ColSet out) {
    if (!fft_pipe_supports(n)) return set_err(ctx, NX_ERR_ARG, "fft_pipe_lde: 17 <= log_size <= 22");
    NX_TRY(fft_pipe_ifirst(ctx, tw, cols, n_cols, n));
    PipeMid m; m.cols = cols; m.out = out; m.itw2 = tw->d_itw2; m.tw2 = tw->d_tw2; m.tw_log = tw->log_half; m.n = n; m.n_cols = n_cols;
    m.n_items = n_cols << (n - T_S); m.scale = m_inv(1u << n);
    switch (n - T_S) {
    case 4: NX_TRY(launch_pipe_mid_t<4>(ctx, m)); break;
    case 5: NX_TRY(launch_pipe_mid_t<5>(ctx, m)); break;
    case 6: NX_TRY(launch_pipe_mid_t<6>(ctx, m)); break;
    case 7: NX_TRY(launch_pipe_mid_t<7>(ctx, m)); break;
    case 8: NX_TRY(launch_pipe_mid_t<8>(ctx, m)); break;
    default: NX_TRY(launch_pipe_mid_t<9>(ctx, m)); break;
    }
    return fft_pipe_ffirst(ctx, tw, out, n_cols, n + 1);
}

}  // namespace nx
