// Lane-level building blocks of the pipelined Circle-FFT kernels (fft_pipe.hip): tile layout, row ownership, twiddle indexing and
// the butterfly rounds.  Everything here is __host__ __device__ and free of HIP runtime state, so the CPU suite replays the
// kernels' data flow lane by lane against the oracle (tests/fftpipe_emul.hip, tests/test_fft_pipe_cpu.py) before a GPU sees them.
//
// Same math as fft13.hip (Stwo CpuBackend circle.rs butterflies on bit-reversed data), different schedule:
//  * a block is persistent: it walks work items (tile, column) with two LDS tile buffers; the NEXT tile is fetched by LDS-DMA
//    (global_load_lds_dwordx4: no VGPRs, no ds_write) while the current one is transformed, the previous tile's stores drain behind.
//  * the 2^13-word tile is stored in 16-byte slots, slot s holding the 4-word group swz(s): an XOR swizzle applied on the GLOBAL
//    address side of the DMA (LDS-DMA writes 64 consecutive slots per wave-instruction).  Every access pattern of the rounds
//    (stride 2^bp rows per lane, b128 rows, consecutive groups) is then bank-conflict free: tools/fftpipe/lds_conflicts.py.
//  * layers next to a global access are fused into it from registers: the inverse FIRST pass's top layer into its store, the
//    forward FIRST pass's two bottom layers into its store, the middle launch's two top layers (radix-4) into the hand-over from the
//    inverse half to the two forward replicas.
#pragma once
#include "field.cuh"

namespace nx {
namespace pipe {

constexpr int T_S = 13;
constexpr u32 T_ROWS = 1u << T_S;       // words per tile
constexpr u32 T_GROUPS = T_ROWS / 4;    // 16-byte groups per tile
constexpr int NT = 512;                 // lanes per block: 16 rows per lane

// logical 4-word group (11 bits) <-> LDS slot.  Bits 4, 3^5, 6 are folded into bits 0, 1, 2: an involution.
NX_HD u32 swz(u32 g) { return g ^ ((g >> 4) & 1u) ^ ((((g >> 3) ^ (g >> 5)) & 1u) << 1) ^ (((g >> 6) & 1u) << 2); }
NX_HD u32 phys(u32 t) { return (swz(t >> 2) << 2) | (t & 3u); }   // tile row -> LDS word

// offset of layer l's table inside a twiddle buffer of 2^tw_log words, for a transform of 2^n points: 2^(n-l-1) entries
NX_HD u32 lvl_off(u32 tw_log, int n, int l) { return (1u << tw_log) - (1u << (n - l)); }

// ---- twiddle register file of one round: layer q (0 = lowest tile bit of the round) at [16 - (16 >> q), +8 >> q) -----------------
// t2 = 2 * twiddle (read from the doubled tables, see field.cuh m_mul_dbl)
template <bool INV>
NX_HD void bfly2(u32& x0, u32& x1, u32 t2, bool neg) {
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

// R butterfly layers over the 16 rows a lane holds (rows differ in tile bits [bp, bp+4); layers act on the low R of them, the
// upper 4-R bits only select independent sub-blocks).  CIRCLE: layer 0 is the circle layer, its twiddles are derived from the
// first line layer's (x, y) pairs as [y, -y, -x, x] (Stwo circle.rs / oracle poly.h circle_twiddle).
// NX_PIPE_ABL (tools/fftpipe/build_abl.sh, never the product build): 1 = no butterfly arithmetic (one XOR per row keeps the LDS round
// trips and the twiddle reads alive), 2 = no global traffic (fft_pipe.hip: no DMA, no stores).  Ablation timing only.
#ifndef NX_PIPE_ABL
#define NX_PIPE_ABL 0
#endif
template <int R, bool INV, bool CIRCLE>
NX_HD void butterflies16(u32* v, const u32* tw) {
    static_assert(!CIRCLE || R >= 2, "circle rounds need the (x, y) pair of the first line layer");
#if NX_PIPE_ABL & 1
    u32 x = 0;
#pragma unroll
    for (int q = CIRCLE ? 1 : 0; q < R; q++)
#pragma unroll
        for (int h = 0; h < (8 >> q); h++) x ^= tw[(16 - (16 >> q)) + h];
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] ^= x;
    return;
#endif
#pragma unroll
    for (int qq = 0; qq < R; qq++) {
        const int q = INV ? qq : R - 1 - qq;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            if (e & (1 << q)) continue;
            const int h = e >> (q + 1);
            if (CIRCLE && q == 0) {
                const int c = h >> 2, sel = h & 3;
                const u32 t = tw[8 + 2 * c + ((sel & 2) ? 0 : 1)];
                bfly2<INV>(v[e], v[e | 1], t, sel == 1 || sel == 2);
            } else {
                bfly2<INV>(v[e], v[e | (1 << q)], tw[(16 - (16 >> q)) + h], false);
            }
        }
    }
}

template <int BP> NX_HD u32 row0(u32 tid) { return BP >= 9 ? tid : (((tid >> BP) << (BP + 4)) | (tid & ((1u << BP) - 1u))); }

// One LDS round trip with 4-byte accesses: rows row0 + (e << BP).
template <int BP, int R, bool INV>
NX_HD void round16(u32* lds, u32 tid, const u32* tw) {
    static_assert(BP >= 2 && BP + 4 <= T_S, "b32 rounds need the two low tile bits outside the round");
    const u32 t0 = row0<BP>(tid);
    u32 v[16];
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = lds[phys(t0 + ((u32)e << BP))];
    butterflies16<R, INV, false>(v, tw);
#pragma unroll
    for (int e = 0; e < 16; e++) lds[phys(t0 + ((u32)e << BP))] = v[e];
}

// The round at tile bits [0, 4): a lane owns rows 16 tid .. 16 tid + 15 = four 16-byte groups.
template <bool INV, bool CIRCLE>
NX_HD void round16_low(u32* lds, u32 tid, const u32* tw) {
    u32 v[16];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint4 x = *reinterpret_cast<const uint4*>(lds + 4 * swz(4 * tid + c));
        v[4 * c] = x.x; v[4 * c + 1] = x.y; v[4 * c + 2] = x.z; v[4 * c + 3] = x.w;
    }
    butterflies16<4, INV, CIRCLE>(v, tw);
#pragma unroll
    for (int c = 0; c < 4; c++)
        *reinterpret_cast<uint4*>(lds + 4 * swz(4 * tid + c)) = make_uint4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
}

// ---- twiddle indices ----------------------------------------------------------------------------------------------------------------
// FIRST passes (contiguous tile T of a 2^n-point transform, layer = tile bit): the R layers of the round at BP for the lane's rows.
// Layer q needs the 8 >> q words tbl[lvl_off(n, BP+q) + first(q) ...]; first(q) is aligned to 8 >> q.
template <int BP> NX_HD u32 first_tw_index(u32 tile, u32 tid, int q) { return ((tile << T_S) + row0<BP>(tid)) >> (BP + q + 1); }

// Middle launch (layers [13, n) of a 2^n-point transform <-> tile bits [B, 13), B = 13 - K, K = n - 13): the twiddle of tile bit b
// at tile row t is entry t >> (b + 1) of layer b + K — the same for every tile.  The block keeps the tables of bits [B, 11) in LDS,
// bit b at word offset mid_tw_off(B, b), 2^(12-b) entries; the forward tables of replica r are the r-th halves of the 2^(n+1)-point
// transform's tables.
NX_HD u32 mid_tw_off(int B, int b) { return (1u << (T_S - B)) - (1u << (T_S - b)); }
NX_HD u32 mid_tw_words(int B) { return 1u << (T_S - B); }   // per direction (4 words of slack at the end)
template <int BP> NX_HD u32 mid_tw_index(u32 tid, int q) { return (BP >= 9 ? 0u : (tid >> BP)) << (3 - q); }

// global word offset of tile row t of middle-launch tile T (runs of 2^B words, 2^13 words apart)
NX_HD u32 mid_goff(u32 t, int B) { return ((t >> B) << T_S) + (t & ((1u << B) - 1u)); }

// ---- memory accessors: address-space-1 vector accesses on the device (never flat: see internal.h), plain ones in the CPU replay -------
// Wave-uniform reads of kernel-constant data (twiddle tables, pointer tables) go through the constant address space: s_load into
// SGPRs — no VGPRs, and no entry in the wave's in-order vmcnt queue behind a tile's DMA.
#define NX_PIPE_AS1 __attribute__((address_space(1)))
#define NX_PIPE_AS4 __attribute__((address_space(4)))
#if defined(__HIP_DEVICE_COMPILE__)
typedef u32 pv4 __attribute__((ext_vector_type(4)));
typedef u32 pv2 __attribute__((ext_vector_type(2)));
NX_HD uint4 lds4(const u32* p) { const pv4 v = *(NX_PIPE_AS4 const pv4*)p; return make_uint4(v.x, v.y, v.z, v.w); }
NX_HD uint2 lds2(const u32* p) { const pv2 v = *(NX_PIPE_AS4 const pv2*)p; return make_uint2(v.x, v.y); }
NX_HD u32 lds1(const u32* p) { return *(NX_PIPE_AS4 const u32*)p; }
NX_HD u32* ldsp(u32* const* p) { return *(u32* NX_PIPE_AS4 const*)p; }
NX_HD uint4 ldg4(const u32* p) { const pv4 v = *(NX_PIPE_AS1 const pv4*)p; return make_uint4(v.x, v.y, v.z, v.w); }
NX_HD uint2 ldg2(const u32* p) { const pv2 v = *(NX_PIPE_AS1 const pv2*)p; return make_uint2(v.x, v.y); }
NX_HD u32 ldg1(const u32* p) { return *(NX_PIPE_AS1 const u32*)p; }
#if NX_PIPE_ABL & 2
NX_HD void stg4(u32* p, uint4 v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(p)); }
#else
NX_HD void stg4(u32* p, uint4 v) { pv4 w = {v.x, v.y, v.z, v.w}; *(NX_PIPE_AS1 pv4*)p = w; }
#endif
#else
NX_HD uint4 ldg4(const u32* p) { return make_uint4(p[0], p[1], p[2], p[3]); }
NX_HD uint2 ldg2(const u32* p) { return make_uint2(p[0], p[1]); }
NX_HD u32 ldg1(const u32* p) { return p[0]; }
NX_HD void stg4(u32* p, uint4 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w; }
NX_HD uint4 lds4(const u32* p) { return make_uint4(p[0], p[1], p[2], p[3]); }
NX_HD uint2 lds2(const u32* p) { return make_uint2(p[0], p[1]); }
NX_HD u32 lds1(const u32* p) { return p[0]; }
NX_HD u32* ldsp(u32* const* p) { return *p; }
#endif
enum { MEM_LDS = 0, MEM_GLOBAL = 1, MEM_SCALAR = 2 };

template <int CNT, int MEM>
NX_HD void ld_words(const u32* p, u32* dst) {
    if constexpr (CNT == 8) {
        const uint4 a = MEM == MEM_GLOBAL ? ldg4(p) : MEM == MEM_SCALAR ? lds4(p) : *reinterpret_cast<const uint4*>(p);
        const uint4 b = MEM == MEM_GLOBAL ? ldg4(p + 4) : MEM == MEM_SCALAR ? lds4(p + 4) : *reinterpret_cast<const uint4*>(p + 4);
        dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w; dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
    } else if constexpr (CNT == 4) {
        const uint4 a = MEM == MEM_GLOBAL ? ldg4(p) : MEM == MEM_SCALAR ? lds4(p) : *reinterpret_cast<const uint4*>(p);
        dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w;
    } else if constexpr (CNT == 2) {
        const uint2 a = MEM == MEM_GLOBAL ? ldg2(p) : MEM == MEM_SCALAR ? lds2(p) : *reinterpret_cast<const uint2*>(p);
        dst[0] = a.x; dst[1] = a.y;
    } else {
        dst[0] = MEM == MEM_GLOBAL ? ldg1(p) : MEM == MEM_SCALAR ? lds1(p) : p[0];
    }
}

// twiddle registers of the round at BP of a FIRST pass (from the doubled global table tbl2).  MEM_SCALAR: the caller passes a
// wave-uniform tid (rounds at BP >= 8: the index depends on tid >> 8 only).
template <int BP, int R, bool CIRCLE, int MEM = MEM_GLOBAL, int Q = 0>
NX_HD void first_tw_fetch(const u32* tbl2, u32 tw_log, int n, u32 tile, u32 tid, u32* tw) {
    static_assert(MEM != MEM_SCALAR || BP >= 8, "only the rounds at tile bits >= 8 have wave-uniform twiddles");
    if constexpr (Q < R) {
        if constexpr (!(CIRCLE && Q == 0))
            ld_words<(8 >> Q), MEM>(tbl2 + lvl_off(tw_log, n, BP + Q) + first_tw_index<BP>(tile, tid, Q), tw + (16 - (16 >> Q)));
        first_tw_fetch<BP, R, CIRCLE, MEM, Q + 1>(tbl2, tw_log, n, tile, tid, tw);
    }
}
// FIRST passes keep a tile's twiddles in LDS "slabs" (fetched by LDS-DMA one item ahead, like the tile): a slab holds, for the layers
// l = L0 .. L1 one after the other, the tile's slice of layer l's table — 2^(12-l) words starting at entry tile * 2^(12-l).
NX_HD constexpr u32 slab_off(int L0, int l) { return (1u << (T_S - L0)) - (1u << (T_S - l)); }      // word offset of layer l's slice
NX_HD constexpr u32 slab_words(int L0, int L1) { return slab_off(L0, L1 + 1); }
// twiddle registers of the round at BP from the slab whose first layer is L0 (= BP, or BP + 1 for the circle round)
template <int BP, int R, bool CIRCLE, int Q = 0>
NX_HD void slab_tw_fetch(const u32* slab, u32 tid, u32* tw) {
    if constexpr (Q < R) {
        if constexpr (!(CIRCLE && Q == 0))
            ld_words<(8 >> Q), MEM_LDS>(slab + slab_off(CIRCLE ? BP + 1 : BP, BP + Q) + (row0<BP>(tid) >> (BP + Q + 1)), tw + (16 - (16 >> Q)));
        slab_tw_fetch<BP, R, CIRCLE, Q + 1>(slab, tid, tw);
    }
}

// twiddle registers of the round at BP of the middle launch (from the block's LDS copy of one direction's tables)
template <int BP, int R, int Q = 0>
NX_HD void mid_tw_fetch(const u32* dir_tw, int B, u32 tid, u32* tw) {
    if constexpr (Q < R) {
        ld_words<(8 >> Q), MEM_LDS>(dir_tw + mid_tw_off(B, BP + Q) + mid_tw_index<BP>(tid, Q), tw + (16 - (16 >> Q)));
        mid_tw_fetch<BP, R, Q + 1>(dir_tw, B, tid, tw);
    }
}

// round plan of the middle launch for K = n - 13 layers: tile bits 11, 12 belong to the radix-4 hand-over, bits [B, 11) to one
// remainder round at B (REM layers) and at most one full round at B + REM
template <int K> struct MidPlan {
    static constexpr int B = T_S - K, NLAY = K - 2, REM = NLAY % 4, NFULL = NLAY / 4, BPF = B + REM;
    static_assert(K >= 4 && K <= 9 && NFULL <= 1, "middle launch: 4 <= K <= 9");
};

// the same twiddle registers straight from the doubled global table (tile kernels: the middle pass's tables are a few KB shared by
// every tile — L1 / L2 hits): layers BP+Q+K of the 2^n_dir-point transform, replica r's half
template <int BP, int R, int Q = 0>
NX_HD void mid_tw_fetch_global(const u32* tbl2, u32 tw_log, int n_dir, int K, u32 r, u32 tid, u32* tw) {
    if constexpr (Q < R) {
        ld_words<(8 >> Q), MEM_GLOBAL>(tbl2 + lvl_off(tw_log, n_dir, BP + Q + K) + (r << (12 - BP - Q)) + mid_tw_index<BP>(tid, Q), tw + (16 - (16 >> Q)));
        mid_tw_fetch_global<BP, R, Q + 1>(tbl2, tw_log, n_dir, K, r, tid, tw);
    }
}

// ---- fused edges -----------------------------------------------------------------------------------------------------------------------
struct MidConsts {          // doubled twiddles of the two top tile bits (11, 12), tile independent
    u32 i11[2], i12;        // inverse: bit 11 entries 0/1, bit 12
    u32 f12[2], f11[4];     // forward replica r: bit 12 entry r, bit 11 entries 2r, 2r+1
    u32 scale;              // 1/N (plain, canonical)
};

// Inverse radix-4 over tile bits 11, 12 of the four rows x[m] (m = bits 12:11), scaled by 1/N: the coefficients.
NX_HD void mid_inverse4(u32* x, const MidConsts& k, u32 sc2, u32 i12s2) {
    bfly2<true>(x[0], x[1], k.i11[0], false);
    bfly2<true>(x[2], x[3], k.i11[1], false);
    const u32 s0 = m_add(x[0], x[2]), d0 = m_sub(x[0], x[2]), s1 = m_add(x[1], x[3]), d1 = m_sub(x[1], x[3]);
    x[0] = m_mul_dbl(s0, sc2); x[2] = m_mul_dbl(d0, i12s2);
    x[1] = m_mul_dbl(s1, sc2); x[3] = m_mul_dbl(d1, i12s2);
}
// Forward radix-4 (bit 12, then bit 11) of replica r on a copy of the coefficients.
NX_HD void mid_forward4(u32* z, const MidConsts& k, int r) {
    bfly2<false>(z[0], z[2], k.f12[r], false);
    bfly2<false>(z[1], z[3], k.f12[r], false);
    bfly2<false>(z[0], z[1], k.f11[2 * r], false);
    bfly2<false>(z[2], z[3], k.f11[2 * r + 1], false);
}

// Forward FIRST pass, layers 1 and 0 of the four rows of global group gp (= global row / 4): (a, b) = the aligned pair of layer-1
// table entries (gp & ~1, gp | 1).  Layer 1 pairs rows (0,2), (1,3) with entry gp; the circle layer pairs (0,1), (2,3) with the
// OTHER entry of the pair, sign pattern by the parity of gp ([y, -y, -x, x] per four butterflies).
NX_HD void first_forward_low2(u32* x, u32 a, u32 b, u32 gp) {
    const bool odd = gp & 1u;
    const u32 t1 = odd ? b : a, tc = odd ? a : b;
    bfly2<false>(x[0], x[2], t1, false);
    bfly2<false>(x[1], x[3], t1, false);
    bfly2<false>(x[0], x[1], tc, odd);
    bfly2<false>(x[2], x[3], tc, !odd);
}

// The wave that ran a tile's last wave-local rounds also stores it: wave w owns rows [1024 w, 1024 w + 1024) = groups [256 w, 256 w + 256);
// pass `it` of a store covers 64 consecutive groups (1 KiB of the tile) per wave.
NX_HD u32 wave_group(u32 tid, int it) { return ((tid >> 6) << 8) + ((u32)it << 6) + (tid & 63u); }

NX_HD u32 comp4(const uint4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// Inverse FIRST pass, store: tile bit 12 (one twiddle for the whole tile, te2 = 2 * twiddle) fused.  No 1/N here: a transform of
// >= 2^14 points never ends with this pass.
NX_HD void ifirst_store(const u32* X, u32 tid, u32* dst_tile, u32 te2) {
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const u32 g = tid + (u32)NT * it;
        const uint4 a = *reinterpret_cast<const uint4*>(X + 4 * swz(g)), b = *reinterpret_cast<const uint4*>(X + 4 * swz(g + T_GROUPS / 2));
        u32 xa[4] = {a.x, a.y, a.z, a.w}, xb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 4; i++) bfly2<true>(xa[i], xb[i], te2, false);
        stg4(dst_tile + 4 * g, make_uint4(xa[0], xa[1], xa[2], xa[3]));
        stg4(dst_tile + 4 * g + T_ROWS / 2, make_uint4(xb[0], xb[1], xb[2], xb[3]));
    }
}

// Forward FIRST pass, store with layers 1 and 0 fused, in two phases (the layer-1 slab is re-filled for the next tile in between):
// load: the lane's four groups and, per group, the aligned pair of layer-1 entries (slab1 = the tile's 2048-word layer-1 slice);
// store: the two layers from registers, then the tile's rows out.
NX_HD void ffirst_store_load(const u32* X, const u32* slab1, u32 tid, uint4* x, u32* tws) {
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const u32 g = wave_group(tid, it);
        x[it] = *reinterpret_cast<const uint4*>(X + 4 * swz(g));
        ld_words<2, MEM_LDS>(slab1 + (g & ~1u), tws + 2 * it);
    }
}
NX_HD void ffirst_store_finish(const uint4* xin, const u32* tws, u32 tid, u32* dst_tile, u32 tile) {
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const u32 g = wave_group(tid, it);
        u32 x[4] = {xin[it].x, xin[it].y, xin[it].z, xin[it].w};
        first_forward_low2(x, tws[2 * it], tws[2 * it + 1], tile * T_GROUPS + g);
        stg4(dst_tile + 4 * g, make_uint4(x[0], x[1], x[2], x[3]));
    }
}

// Middle launch, hand-over: X holds the tile after the inverse layers of bits [B, 11).  Inverse radix-4 over bits 11, 12 + 1/N ->
// the coefficients, written to the column; then from registers the forward radix-4 of both replicas -> X (replica 0, in place: a lane
// rewrites exactly the slots it read) and Y (replica 1).
NX_HD void mid_handover(u32* X, u32* Y, u32 tid, const MidConsts& k, u32* coef_tile, int B) {
    const u32 sc2 = k.scale << 1, i12s2 = m_mul(k.i12 >> 1, k.scale) << 1;
    uint4 x[4];
#pragma unroll
    for (int m = 0; m < 4; m++) x[m] = *reinterpret_cast<const uint4*>(X + 4 * swz(tid + (u32)NT * m));
    u32 c[4][4];   // [m][component]
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u32 r[4] = {comp4(x[0], i), comp4(x[1], i), comp4(x[2], i), comp4(x[3], i)};
        mid_inverse4(r, k, sc2, i12s2);
#pragma unroll
        for (int m = 0; m < 4; m++) c[m][i] = r[m];
    }
#pragma unroll
    for (int m = 0; m < 4; m++) stg4(coef_tile + mid_goff(4 * (tid + (u32)NT * m), B), make_uint4(c[m][0], c[m][1], c[m][2], c[m][3]));
#pragma unroll
    for (int r = 0; r < 2; r++) {
        u32 z[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            u32 w[4] = {c[0][i], c[1][i], c[2][i], c[3][i]};
            mid_forward4(w, k, r);
#pragma unroll
            for (int m = 0; m < 4; m++) z[m][i] = w[m];
        }
        u32* Z = r ? Y : X;
#pragma unroll
        for (int m = 0; m < 4; m++) *reinterpret_cast<uint4*>(Z + 4 * swz(tid + (u32)NT * m)) = make_uint4(z[m][0], z[m][1], z[m][2], z[m][3]);
    }
}
NX_HD void mid_store(const u32* Z, u32 tid, u32* out_tile, int B) {
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const u32 g = wave_group(tid, it);
        stg4(out_tile + mid_goff(4 * g, B), *reinterpret_cast<const uint4*>(Z + 4 * swz(g)));
    }
}

// work item -> (tile, column): XCD x (= item % 8, blocks are dealt round-robin to the XCDs) owns the tiles [x, x+1) * tiles/8; the
// columns of a tile run back to back on that XCD, so the tile's twiddles reach one L2 once.  Speed only, never correctness.
struct Item { u32 tile, col; };
NX_HD Item decode_item(u32 i, u32 tiles, u32 n_cols) {
    Item r;
    if (tiles >= 8) { const u32 x = i & 7u, y = i >> 3; r.col = y % n_cols; r.tile = x * (tiles >> 3) + y / n_cols; }
    else { r.col = i % n_cols; r.tile = i / n_cols; }
    return r;
}

}  // namespace pipe
}  // namespace nx
