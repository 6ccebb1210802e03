// Blake2s mixed-degree Merkle commitment and proof-of-work grinding for gfx950 (K5, K10).
//
// Replaces Stwo `MerkleOps<Blake2sMerkleHasher>::commit_on_layer` (driven by MerkleProver::commit,
// reached from reference prover/src/machine.rs:228,237,263 and, on the verifier side,
// machine.rs:363-417) and `GrindOps<Blake2sChannel>::grind`.
//
// Design: one lane per tree node; a node's message is streamed 16 columns (one 64-byte Blake2s
// block) at a time — for each column the 64 lanes of a wave read 64 consecutive rows, a 256-byte
// coalesced access — with the 8-word chaining state and the 16-word block in VGPRs, the 10 rounds
// fully unrolled with compile-time message schedule (rotates lower to v_alignbit_b32).  The kernel
// is VALU-issue bound (≈1000 integer ops per block), not HBM bound.
// Where a tree level has fewer nodes than the machine has lanes (the top of every tree, the small FRI layers) the LATENCY of
// a compression is what counts: there one compression runs on a quad of lanes (b2s_compress_quad) and the nodes stay in LDS
// (merkle_top_kernel); the last FRI layers are committed, mixed into the channel and folded in one launch (fri_tail_kernel),
// the larger ones get their channel step from fri_channel_kernel — the FRI commit phase never waits for the host.
#include "internal.h"
#include <atomic>
#include <algorithm>
#include <numeric>
#include <string.h>
#include <stdlib.h>

namespace nx {

__device__ __constant__ const u32 B2S_IV_D[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                                  0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};

__device__ __forceinline__ u32 rotr(u32 x, int r) { return __builtin_amdgcn_alignbit(x, x, r); }

// a + b + x stays one v_add3_u32: splitting it into two v_add_u32 wins 7 % in a register-only loop
// (tools/ubench/b2s_rates.hip) but loses 14 % in the real leaf kernel (A/B on the same box, 2.31 vs 2.67 ms / 128 columns).
#define B2S_G(a, b, c, d, x, y)                         \
    a = a + b + (x); d = rotr(d ^ a, 16);               \
    c = c + d;       b = rotr(b ^ c, 12);               \
    a = a + b + (y); d = rotr(d ^ a, 8);                \
    c = c + d;       b = rotr(b ^ c, 7);

#define B2S_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
    B2S_G(v0, v4, v8, v12, m[s0], m[s1]);   B2S_G(v1, v5, v9, v13, m[s2], m[s3]);       \
    B2S_G(v2, v6, v10, v14, m[s4], m[s5]);  B2S_G(v3, v7, v11, v15, m[s6], m[s7]);      \
    B2S_G(v0, v5, v10, v15, m[s8], m[s9]);  B2S_G(v1, v6, v11, v12, m[s10], m[s11]);    \
    B2S_G(v2, v7, v8, v13, m[s12], m[s13]); B2S_G(v3, v4, v9, v14, m[s14], m[s15]);

__device__ __forceinline__ void b2s_compress(u32 h[8], const u32 m[16], u32 t0, u32 f0) {
    u32 v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
    u32 v8 = 0x6A09E667u, v9 = 0xBB67AE85u, v10 = 0x3C6EF372u, v11 = 0xA54FF53Au;
    u32 v12 = 0x510E527Fu ^ t0, v13 = 0x9B05688Cu, v14 = 0x1F83D9ABu ^ f0, v15 = 0x5BE0CD19u;
    B2S_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    B2S_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    B2S_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
    B2S_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    B2S_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
    B2S_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    B2S_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
    B2S_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    B2S_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
    B2S_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
    h[0] ^= v0 ^ v8;  h[1] ^= v1 ^ v9;  h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11;
    h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
}

// 16 consecutive columns of one 64-byte message block at row i: the column addresses first (one uniform branch, wide scalar loads of
// the pointer table), then 16 global loads issued back to back; columns past n_cols read as zero (the padded last block)
__device__ __forceinline__ void load_block16(const ColSet& cols, u32 n_cols, u32 c0, u64 i, u32* dst) {
    const u32* p[16];
    if (c0 + 16 <= n_cols) {
        if (cols.table) {
#pragma unroll
            for (int k = 0; k < 16; k++) p[k] = cols.table[c0 + k];
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) p[k] = cols.base + (uint64_t)(c0 + k) * cols.stride;
        }
        const u32 off = (u32)i << 2;          // rows < 2^30: one 32-bit byte offset for all 16 loads, the column bases stay in SGPRs
#pragma unroll
        for (int k = 0; k < 16; k++) dst[k] = gld_off(p[k], off);
    } else {
        const u32 off = (u32)i << 2;
#pragma unroll
        for (int k = 0; k < 16; k++) p[k] = cols.col(min(c0 + k, n_cols - 1));
#pragma unroll
        for (int k = 0; k < 16; k++) dst[k] = (c0 + k < n_cols) ? gld_off(p[k], off) : 0u;
    }
}

// One Merkle layer: node i = H(prev[2i] ‖ prev[2i+1] ‖ col_0[i] ‖ col_1[i] ‖ ...).
//   MODE 0: standard Blake2s-256 (byte counter, final-block flag, IV ^ parameter block)
//   MODE 1: zero-state raw compression chaining, t = f = 0, zero-padded 64-byte blocks
template <int MODE>
__global__ __launch_bounds__(256) void merkle_layer_kernel(ColSet cols, u32 n_cols, const u32* __restrict__ prev,
                                                           u32* __restrict__ out, u32 n_nodes) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    u32 h[8];
    if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) h[k] = B2S_IV_D[k];
        h[0] ^= 0x01010020u;
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) h[k] = 0;
    }
    const u32 total_bytes = (prev ? 64u : 0u) + 4u * n_cols;
    u32 m[16];
    u32 t = 0;
    if (prev) {
        const u32* p = prev + (size_t)i * 16;
        uint4 a = gld4(p), b = gld4(p + 4), c = gld4(p + 8), d = gld4(p + 12);
        m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
        m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w; m[12] = d.x; m[13] = d.y; m[14] = d.z; m[15] = d.w;
        t = 64;
        if (MODE == 0) b2s_compress(h, m, t, n_cols == 0 ? 0xFFFFFFFFu : 0u);
        else b2s_compress(h, m, 0, 0);
    } else if (MODE == 0 && n_cols == 0) {
#pragma unroll
        for (int k = 0; k < 16; k++) m[k] = 0;
        b2s_compress(h, m, 0, 0xFFFFFFFFu);  // Blake2s of the empty message
    }
    // stream the columns 16 at a time, double-buffered: the 16 loads of chunk k+1 are in flight while the
    // ~1000 VALU ops of chunk k's compression run
    u32 nx_[16];
    auto load_chunk = [&](u32 c0, u32* dst) { load_block16(cols, n_cols, c0, i, dst); };
    if (n_cols) load_chunk(0, nx_);
    for (u32 c0 = 0; c0 < n_cols; c0 += 16) {
#pragma unroll
        for (int k = 0; k < 16; k++) m[k] = nx_[k];
        bool last = c0 + 16 >= n_cols;
        if (!last) load_chunk(c0 + 16, nx_);
        if (MODE == 0) { t = last ? total_bytes : t + 64; b2s_compress(h, m, t, last ? 0xFFFFFFFFu : 0u); }
        else b2s_compress(h, m, 0, 0);
    }
    uint4* o = reinterpret_cast<uint4*>(out + (size_t)i * 8);
    o[0] = make_uint4(h[0], h[1], h[2], h[3]);
    o[1] = make_uint4(h[4], h[5], h[6], h[7]);
}


// TWO node-only levels in one launch: lane i reads the four nodes prev[4i .. 4i + 3] (128 contiguous bytes), writes the two nodes of the
// level in between and node i of the level above them.  The same three compressions as two launches of merkle_layer_kernel, without reading
// the middle level back and with half the launches' ramps and tails (the level launches sit at 29-32 G compressions/s against the 39.5 of
// the compression loop: DESIGN.md section 6 item 23).
template <int MODE>
__global__ __launch_bounds__(256) void merkle_pair_levels_kernel(const u32* __restrict__ prev, u32* __restrict__ mid, u32* __restrict__ out, u32 n_nodes) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const u32* p = prev + (size_t)i * 32;
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = gld4(p + 4 * k);
    u32 top[16];
#pragma unroll
    for (int half = 0; half < 2; half++) {
        u32 h[8], m[16];
#pragma unroll
        for (int k = 0; k < 8; k++) h[k] = MODE == 0 ? B2S_IV_D[k] : 0u;
        if (MODE == 0) h[0] ^= 0x01010020u;
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint4 w = v[4 * half + k]; m[4 * k] = w.x; m[4 * k + 1] = w.y; m[4 * k + 2] = w.z; m[4 * k + 3] = w.w; }
        if (MODE == 0) b2s_compress(h, m, 64, 0xFFFFFFFFu); else b2s_compress(h, m, 0, 0);
        uint4* o = reinterpret_cast<uint4*>(mid + ((size_t)2 * i + half) * 8);
        o[0] = make_uint4(h[0], h[1], h[2], h[3]);
        o[1] = make_uint4(h[4], h[5], h[6], h[7]);
#pragma unroll
        for (int k = 0; k < 8; k++) top[8 * half + k] = h[k];
    }
    u32 h[8];
#pragma unroll
    for (int k = 0; k < 8; k++) h[k] = MODE == 0 ? B2S_IV_D[k] : 0u;
    if (MODE == 0) h[0] ^= 0x01010020u;
    if (MODE == 0) b2s_compress(h, top, 64, 0xFFFFFFFFu); else b2s_compress(h, top, 0, 0);
    uint4* o = reinterpret_cast<uint4*>(out + (size_t)i * 8);
    o[0] = make_uint4(h[0], h[1], h[2], h[3]);
    o[1] = make_uint4(h[4], h[5], h[6], h[7]);
}

// One column shard of a leaf layer: continue (or start) the per-row chaining state over this shard's columns.
// `first`: the shard starts at column 0 (state = IV); `last`: it holds the layer's last column (finalisation).
template <int MODE>
__global__ __launch_bounds__(256) void merkle_leaf_chain_kernel(ColSet cols, u32 n_cols, u32 col_offset, u32 total_cols,
                                                                const u32* __restrict__ state_in, u32* __restrict__ state_out,
                                                                u64 row_begin, u64 n_rows) {
    const u64 r = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const u64 i = row_begin + r;
    const bool last_shard = col_offset + n_cols == total_cols;
    u32 h[8];
    if (state_in) {
        const uint4* p = reinterpret_cast<const uint4*>(state_in + r * 8);
        uint4 a = p[0], b = p[1];
        h[0] = a.x; h[1] = a.y; h[2] = a.z; h[3] = a.w; h[4] = b.x; h[5] = b.y; h[6] = b.z; h[7] = b.w;
    } else if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) h[k] = B2S_IV_D[k];
        h[0] ^= 0x01010020u;
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) h[k] = 0;
    }
    const u32 total_bytes = 4u * total_cols;
    u32 t = 4u * col_offset;   // bytes hashed by the previous shards
    u32 m[16], nx_[16];
    auto load_chunk = [&](u32 c0, u32* dst) { load_block16(cols, n_cols, c0, i, dst); };
    if (n_cols) load_chunk(0, nx_);
    for (u32 c0 = 0; c0 < n_cols; c0 += 16) {
#pragma unroll
        for (int k = 0; k < 16; k++) m[k] = nx_[k];
        const bool last_chunk = c0 + 16 >= n_cols;
        if (!last_chunk) load_chunk(c0 + 16, nx_);
        const bool fin = last_chunk && last_shard;
        if (MODE == 0) { t = fin ? total_bytes : t + 64; b2s_compress(h, m, t, fin ? 0xFFFFFFFFu : 0u); }
        else b2s_compress(h, m, 0, 0);
    }
    uint4* o = reinterpret_cast<uint4*>(state_out + r * 8);
    o[0] = make_uint4(h[0], h[1], h[2], h[3]);
    o[1] = make_uint4(h[4], h[5], h[6], h[7]);
}


// ---- one Blake2s compression spread over the 4 lanes of a quad ----
// The top of a tree is a chain of dependent compressions (a level has fewer nodes than the machine has lanes), so what matters
// there is the LATENCY of one compression, ~1000 dependent-ish VALU ops for a single lane.  Lane q of a quad holds column q of the
// 4x4 state (a_q, b_q, c_q, d_q): the column step is 4 G functions in parallel, the diagonal step the same after rotating b, c, d
// by 1, 2, 3 lanes (DPP quad_perm) — ~300 ops per compression.  The message words each lane needs (4 per round) are fetched
// from LDS through per-lane offsets prepared once per kernel.
__device__ __constant__ const unsigned char B2S_SIGMA_D[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};

struct QuadLane { u32 off[10]; u32 iv_lo, iv_hi; int q; };   // per-lane constants of the quad scheme; off[r]: the 4 message-word BYTE offsets of round r, one per byte

__device__ __forceinline__ void quad_lane_init(QuadLane& L) {
    const int q = threadIdx.x & 3;
    L.q = q;
#pragma unroll
    for (int r = 0; r < 10; r++)
        L.off[r] = 4u * ((u32)B2S_SIGMA_D[r][2 * q] | ((u32)B2S_SIGMA_D[r][2 * q + 1] << 8) | ((u32)B2S_SIGMA_D[r][8 + 2 * q] << 16) | ((u32)B2S_SIGMA_D[r][9 + 2 * q] << 24));
    L.iv_lo = B2S_IV_D[q]; L.iv_hi = B2S_IV_D[4 + q];
}

#define QROT(x, ctrl) (u32) __builtin_amdgcn_mov_dpp((int)(x), ctrl, 0xF, 0xF, true)

__device__ __forceinline__ u32 quad_word(const u32* msg, u32 packed, int k) {
    return *reinterpret_cast<const u32*>(reinterpret_cast<const char*>(msg) + ((packed >> (8 * k)) & 0xFFu));
}
// the 10 rounds on a quad-distributed state; the message words of round r + 1 are requested while round r computes (requesting
// all 40 up front spills at the 128-VGPR limit of a 1024-lane block: measured 40 % slower)
__device__ __forceinline__ void quad_rounds(const QuadLane& L, const u32* msg, u32& a, u32& b, u32& c, u32& d) {
    u32 w0 = quad_word(msg, L.off[0], 0), w1 = quad_word(msg, L.off[0], 1), w2 = quad_word(msg, L.off[0], 2), w3 = quad_word(msg, L.off[0], 3);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        u32 n0 = 0, n1 = 0, n2 = 0, n3 = 0;
        if (r < 9) { n0 = quad_word(msg, L.off[r + 1], 0); n1 = quad_word(msg, L.off[r + 1], 1); n2 = quad_word(msg, L.off[r + 1], 2); n3 = quad_word(msg, L.off[r + 1], 3); }
        B2S_G(a, b, c, d, w0, w1);
        b = QROT(b, 0x39); c = QROT(c, 0x4E); d = QROT(d, 0x93);
        B2S_G(a, b, c, d, w2, w3);
        b = QROT(b, 0x93); c = QROT(c, 0x4E); d = QROT(d, 0x39);
        w0 = n0; w1 = n1; w2 = n2; w3 = n3;
    }
}

// msg: the node's 16 message words in LDS; returns digest words q (lo) and 4 + q (hi).  MODE 0: standard Blake2s-256 with
// counter t0 and final flag f0; MODE 1: raw compression on a zero state.
template <int MODE>
__device__ __forceinline__ void b2s_compress_quad(const QuadLane& L, const u32* msg, u32 t0, u32 f0, u32& lo, u32& hi) {
    const u32 ha = MODE == 0 ? (L.iv_lo ^ (L.q == 0 ? 0x01010020u : 0u)) : 0u, hb = MODE == 0 ? L.iv_hi : 0u;
    u32 a = ha, b = hb, c = L.iv_lo, d = L.iv_hi ^ (L.q == 0 ? t0 : 0u) ^ (L.q == 2 ? f0 : 0u);
    quad_rounds(L, msg, a, b, c, d);
    lo = ha ^ a ^ c; hi = hb ^ b ^ d;
}

// Levels from_log .. 0 of a tree whose level from_log + 1 is already in LDS (`nodes`: level k at word offset (2^k - 1) * 8), by a
// 1024-lane block.  A level with 1024 nodes uses one lane per node, narrower levels one quad per node (4 nodes in flight per
// 16 lanes, the compression ~3x shorter).  Every node is also written to the tree in global memory (`base`, same layout): the
// decommitment reads it later.  Ends with a barrier.
template <int MODE>
__device__ __forceinline__ void tree_levels_lds(const QuadLane& L, u32* nodes, u32* __restrict__ base, int from_log) {
    for (int log = from_log; log >= 0; log--) {
        const u32 n = 1u << log;
        u32* out_g = base + ((size_t)n - 1) * 8;
        u32* out_l = nodes + ((size_t)n - 1) * 8;
        const u32* in_l = nodes + (((size_t)2 << log) - 1) * 8;
        if (n >= 1024) {
            for (u32 i = threadIdx.x; i < n; i += 1024) {
                u32 h[8], m[16];
#pragma unroll
                for (int k = 0; k < 8; k++) h[k] = MODE == 0 ? B2S_IV_D[k] : 0u;
                if (MODE == 0) h[0] ^= 0x01010020u;
#pragma unroll
                for (int k = 0; k < 16; k++) m[k] = in_l[(size_t)i * 16 + k];
                if (MODE == 0) b2s_compress(h, m, 64, 0xFFFFFFFFu); else b2s_compress(h, m, 0, 0);
#pragma unroll
                for (int k = 0; k < 8; k++) out_l[(size_t)i * 8 + k] = h[k];
                gst4(out_g + (size_t)i * 8, make_uint4(h[0], h[1], h[2], h[3]));
                gst4(out_g + (size_t)i * 8 + 4, make_uint4(h[4], h[5], h[6], h[7]));
            }
        } else {
            for (u32 i = threadIdx.x >> 2; i < n; i += 256) {   // whole quads are active or idle together (DPP stays inside the quad)
                u32 lo, hi;
                b2s_compress_quad<MODE>(L, in_l + (size_t)i * 16, MODE == 0 ? 64u : 0u, MODE == 0 ? 0xFFFFFFFFu : 0u, lo, hi);
                out_l[(size_t)i * 8 + L.q] = lo; out_l[(size_t)i * 8 + 4 + L.q] = hi;
                gst(out_g + (size_t)i * 8 + L.q, lo); gst(out_g + (size_t)i * 8 + 4 + L.q, hi);
            }
        }
        __syncthreads();
    }
}

// Top of the tree in ONE launch: layers `top`..0 (2^top <= 1024 nodes) when no columns are injected there.
// `base` is the start of the tree allocation (layer k at node offset 2^k - 1).  Dynamic LDS: (2^(top+2) - 1) * 32 bytes.
template <int MODE>
__global__ __launch_bounds__(1024) void merkle_top_kernel(u32* __restrict__ base, int top) {
    extern __shared__ __attribute__((aligned(16))) u32 top_nodes[];
    QuadLane L;
    quad_lane_init(L);
    {   // level top + 1 (written by the previous launch) into LDS
        const size_t w0 = (((size_t)2 << top) - 1) * 8, nw = ((size_t)2 << top) * 8;
        uint4 v[4];   // <= 2048 nodes = 4 x uint4 per lane: all requested before the first LDS write
#pragma unroll
        for (int it = 0; it < 4; it++) { const size_t i = (size_t)threadIdx.x * 4 + (size_t)it * 4096; if (i < nw) v[it] = gld4(base + w0 + i); }
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const size_t i = (size_t)threadIdx.x * 4 + (size_t)it * 4096;
            if (i < nw) { top_nodes[w0 + i] = v[it].x; top_nodes[w0 + i + 1] = v[it].y; top_nodes[w0 + i + 2] = v[it].z; top_nodes[w0 + i + 3] = v[it].w; }
        }
    }
    __syncthreads();
    tree_levels_lds<MODE>(L, top_nodes, base, top);
}

static int launch_merkle_top(nx_ctx* ctx, u32* buf, int top) {
    static std::atomic<uint64_t> attr_set{0};   // one bit per device: function attributes are per device
    if (!(attr_set.load() & (1ull << (ctx->device & 63)))) {
        NX_HIP(ctx, hipFuncSetAttribute((const void*)merkle_top_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        NX_HIP(ctx, hipFuncSetAttribute((const void*)merkle_top_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.fetch_or(1ull << (ctx->device & 63));
    }
    const size_t lds = (((size_t)4 << top) - 1) * 32;
    if (ctx->hash_mode == NX_HASH_BLAKE2S) hipLaunchKernelGGL(merkle_top_kernel<0>, dim3(1), dim3(1024), lds, ctx->stream, buf, top);
    else hipLaunchKernelGGL(merkle_top_kernel<1>, dim3(1), dim3(1024), lds, ctx->stream, buf, top);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}


// Levels in_log-1 .. in_log-d of a tree in ONE launch (no columns injected there): block b owns the 128 nodes [128 b, 128 b + 128) of
// level in_log (written by the previous launch) and the subtrees above them — d <= 7 levels, one quad of lanes per node, the nodes
// LDS-resident between levels.  Replaces d launches of 5-6 us each whose work is a few hundred thousand compressions in total.
constexpr int SUBTREE_LEAVES = 128;
template <int MODE>
__global__ __launch_bounds__(2 * SUBTREE_LEAVES) void merkle_subtree_kernel(u32* __restrict__ base, int in_log, int d) {
    __shared__ __attribute__((aligned(16))) u32 lv[(2 * SUBTREE_LEAVES - 1) * 8];   // level k of the block at node offset 256 - (256 >> k)
    QuadLane L;
    quad_lane_init(L);
    const u32 tid = threadIdx.x, b = blockIdx.x;
    {
        const u32* src = base + (((size_t)1 << in_log) - 1 + (size_t)b * SUBTREE_LEAVES) * 8;
        const uint4 v = gld4(src + 4 * tid);                                          // 128 nodes x 8 words = 256 lanes x 4 words
        lv[4 * tid] = v.x; lv[4 * tid + 1] = v.y; lv[4 * tid + 2] = v.z; lv[4 * tid + 3] = v.w;
    }
    __syncthreads();
    for (int k = 1; k <= d; k++) {
        const u32 n = SUBTREE_LEAVES >> k, i = tid >> 2;
        const u32* in_l = lv + (size_t)(2 * SUBTREE_LEAVES - ((2 * SUBTREE_LEAVES) >> (k - 1))) * 8;
        u32* out_l = lv + (size_t)(2 * SUBTREE_LEAVES - ((2 * SUBTREE_LEAVES) >> k)) * 8;
        u32* out_g = base + (((size_t)1 << (in_log - k)) - 1 + (size_t)b * n) * 8;
        if (i < n) {                                                                  // whole quads are active or idle together
            u32 lo, hi;
            b2s_compress_quad<MODE>(L, in_l + (size_t)i * 16, MODE == 0 ? 64u : 0u, MODE == 0 ? 0xFFFFFFFFu : 0u, lo, hi);
            out_l[(size_t)i * 8 + L.q] = lo; out_l[(size_t)i * 8 + 4 + L.q] = hi;
            gst(out_g + (size_t)i * 8 + L.q, lo); gst(out_g + (size_t)i * 8 + 4 + L.q, hi);
        }
        __syncthreads();
    }
}

static int launch_merkle_subtree(nx_ctx* ctx, u32* buf, int in_log, int d) {
    const unsigned blocks = 1u << (in_log - 7);
    if (ctx->hash_mode == NX_HASH_BLAKE2S) hipLaunchKernelGGL(merkle_subtree_kernel<0>, dim3(blocks), dim3(2 * SUBTREE_LEAVES), 0, ctx->stream, buf, in_log, d);
    else hipLaunchKernelGGL(merkle_subtree_kernel<1>, dim3(blocks), dim3(2 * SUBTREE_LEAVES), 0, ctx->stream, buf, in_log, d);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

// ---------------------------------------------------------------- FRI tail: the last layers of FriProver::commit in ONE launch
// Below ~2^12 points a FRI layer costs ~60 us of launches and host round trips (Merkle layers, root download, channel,
// alpha upload, fold) for a few microseconds of work, and the layers are strictly sequential (alpha_{k+1} = H(root_k)).  This
// kernel keeps the Blake2sChannel on the device: per layer, one 1024-lane block builds the layer's Merkle tree (same node rule
// as merkle_layer_kernel), lane 0 mixes the root into the channel and draws the folding alpha (Blake2sChannel::mix_root /
// draw_secure_felt, host/channel.h), then all lanes fold the line (fold_line_kernel's rule) into the next layer.
struct FriTailArgs {
    u32* eval[FRI_TAIL_MAX_LAYERS + 1];   // eval[j]: 4 coordinate columns of 2^(log0 - j) words, contiguous; eval[0] is the input
    u32* tree[FRI_TAIL_MAX_LAYERS];       // tree[j]: (2^(log0 - j + 1) - 1) nodes x 8 words, layer k at node offset 2^k - 1
    int n_layers, log0;
    const u32* itw; u32 tw_log;
    u32* state;                           // device FRI channel state (internal.h): digest[8], n_sent, records (root[8], alpha[4])
    int j0;                               // record index of this launch's first layer
};

__device__ __forceinline__ void b2s_init_std(u32 h[8]) {
#pragma unroll
    for (int k = 0; k < 8; k++) h[k] = B2S_IV_D[k];
    h[0] ^= 0x01010020u;
}

// Blake2sChannel::mix_root(root) followed by draw_secure_felt, by the four lanes of quad 0.  chan (LDS): [0,8) digest (updated),
// [8,24) scratch message block, [24,28) the drawn alpha.  root: 8 words (LDS or global).  Always standard Blake2s.
__device__ __forceinline__ void channel_mix_draw_quad(const QuadLane& L, u32* chan, const u32* root, u32& n_sent) {
    const u32 tid = threadIdx.x;
    u32 lo, hi;
    chan[8 + tid] = chan[tid]; chan[12 + tid] = chan[4 + tid]; chan[16 + tid] = root[tid]; chan[20 + tid] = root[4 + tid];
    b2s_compress_quad<0>(L, chan + 8, 64, 0xFFFFFFFFu, lo, hi);      // digest = Blake2s(digest ‖ root)
    chan[tid] = lo; chan[4 + tid] = hi;
    n_sent = 0;
    // draw_secure_felt: Blake2s(digest ‖ n_sent_le32 ‖ 29 zero bytes) until all 8 words are < 2P; the first 4, reduced
    for (;;) {
        chan[8 + tid] = lo; chan[12 + tid] = hi; chan[16 + tid] = tid == 0 ? n_sent : 0u; chan[20 + tid] = 0;
        // two blocks: 64 bytes (not final), then 1 zero byte (final, t = 65), chained through the quad-distributed state
        const u32 ha = L.iv_lo ^ (L.q == 0 ? 0x01010020u : 0u), hb = L.iv_hi;
        u32 x = ha, y = hb, z = L.iv_lo, w = L.iv_hi ^ (L.q == 0 ? 64u : 0u);
        quad_rounds(L, chan + 8, x, y, z, w);
        const u32 h1a = ha ^ x ^ z, h1b = hb ^ y ^ w;
        x = h1a; y = h1b; z = L.iv_lo; w = L.iv_hi ^ (L.q == 0 ? 65u : 0u) ^ (L.q == 2 ? 0xFFFFFFFFu : 0u);
#pragma unroll
        for (int r = 0; r < 10; r++) {
            B2S_G(x, y, z, w, 0u, 0u);
            y = QROT(y, 0x39); z = QROT(z, 0x4E); w = QROT(w, 0x93);
            B2S_G(x, y, z, w, 0u, 0u);
            y = QROT(y, 0x93); z = QROT(z, 0x4E); w = QROT(w, 0x39);
        }
        const u32 g_lo = h1a ^ x ^ z, g_hi = h1b ^ y ^ w;
        n_sent++;
        int ok = (g_lo < 2u * P && g_hi < 2u * P) ? 1 : 0;           // all four lanes must agree: AND over the quad
        ok &= __builtin_amdgcn_mov_dpp(ok, 0x39, 0xF, 0xF, true);
        ok &= __builtin_amdgcn_mov_dpp(ok, 0x4E, 0xF, 0xF, true);
        if (ok) { chan[24 + tid] = g_lo >= P ? g_lo - P : g_lo; break; }
    }
}

// One FRI layer's channel step for the layers whose trees are built by ordinary launches: record j <- (root, alpha).
__global__ __launch_bounds__(64) void fri_channel_kernel(u32* __restrict__ state, const u32* __restrict__ root, int j) {
    __shared__ u32 chan[28];
    __shared__ u32 rt[8];
    QuadLane L;
    quad_lane_init(L);
    const u32 tid = threadIdx.x;
    if (tid < 8) { chan[tid] = state[tid]; rt[tid] = root[tid]; }
    __syncthreads();
    if (tid < 4) {
        u32 n_sent = 0;
        channel_mix_draw_quad(L, chan, rt, n_sent);
        u32* rec = state + FRI_STATE_HEAD + FRI_STATE_REC * j;
        rec[tid] = rt[tid]; rec[4 + tid] = rt[4 + tid]; rec[8 + tid] = chan[24 + tid];
        state[tid] = chan[tid]; state[4 + tid] = chan[4 + tid];
        if (tid == 0) state[8] = n_sent;
    }
}

template <int MODE>
__global__ __launch_bounds__(1024) void fri_tail_kernel(FriTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32 tail_lds[];   // tree nodes of the current layer (level k at (2^k - 1) * 8), then 32 channel words
    u32* nodes = tail_lds;
    u32* chan = tail_lds + (((size_t)2 << a.log0) - 1) * 8;          // [0,8) digest, [8,24) message block, [24,28) alpha
    const u32 tid = threadIdx.x;
    QuadLane L;
    quad_lane_init(L);
    if (tid < 8) chan[tid] = a.state[tid];
    u32 n_sent = 0;
    __syncthreads();
    for (int j = 0; j < a.n_layers; j++) {
        const int l = a.log0 - j;
        const u32 n = 1u << l;
        const u32* c0 = a.eval[j]; const u32* c1 = c0 + n; const u32* c2 = c1 + n; const u32* c3 = c2 + n;
        u32* tree = a.tree[j];
        // leaf layer: node i = H(c0[i], c1[i], c2[i], c3[i])  (merkle_layer_kernel with prev == NULL, n_cols == 4)
        for (u32 i = tid; i < n; i += 1024) {
            u32 h[8], m[16];
            if (MODE == 0) b2s_init_std(h); else { for (int k = 0; k < 8; k++) h[k] = 0; }
#pragma unroll
            for (int k = 4; k < 16; k++) m[k] = 0;
            m[0] = gld(c0 + i); m[1] = gld(c1 + i); m[2] = gld(c2 + i); m[3] = gld(c3 + i);
            if (MODE == 0) b2s_compress(h, m, 16, 0xFFFFFFFFu); else b2s_compress(h, m, 0, 0);
            u32* ol = nodes + ((size_t)n - 1 + i) * 8;
            u32* og = tree + ((size_t)n - 1 + i) * 8;
#pragma unroll
            for (int k = 0; k < 8; k++) ol[k] = h[k];
            gst4(og, make_uint4(h[0], h[1], h[2], h[3])); gst4(og + 4, make_uint4(h[4], h[5], h[6], h[7]));
        }
        __syncthreads();
        tree_levels_lds<MODE>(L, nodes, tree, l - 1);
        // the channel, on quad 0 (always standard Blake2s): mix_root, then draw_secure_felt
        if (tid < 4) {
            channel_mix_draw_quad(L, chan, nodes, n_sent);
            u32* rec = a.state + FRI_STATE_HEAD + FRI_STATE_REC * (a.j0 + j);
            rec[tid] = nodes[tid]; rec[4 + tid] = nodes[4 + tid]; rec[8 + tid] = chan[24 + tid];
        }
        __syncthreads();
        // fold_line: next[i] = (f0 + f1) + alpha * ((f0 - f1) / x_i), pairs (2i, 2i+1), 1/x_i = itw layer (H - l), entry i
        const QM31 alpha = qm(chan[24], chan[25], chan[26], chan[27]);
        u32* d0 = a.eval[j + 1]; const u32 nh = n >> 1;
        for (u32 i = tid; i < nh; i += 1024) {
            const u32 xi = gld(a.itw + ((1u << a.tw_log) - (1u << l) + i));
            const QM31 f0 = qm(gld(c0 + 2 * i), gld(c1 + 2 * i), gld(c2 + 2 * i), gld(c3 + 2 * i));
            const QM31 f1 = qm(gld(c0 + 2 * i + 1), gld(c1 + 2 * i + 1), gld(c2 + 2 * i + 1), gld(c3 + 2 * i + 1));
            const QM31 sum = q_add(f0, f1), t = q_mul_m(q_sub(f0, f1), xi);
            const QM31 o = q_add(sum, q_mul(alpha, t));
            gst(d0 + i, o.a.a); gst(d0 + nh + i, o.a.b); gst(d0 + 2 * nh + i, o.b.a); gst(d0 + 3 * nh + i, o.b.b);
        }
        __threadfence_block();
        __syncthreads();
    }
    if (tid < 8) a.state[tid] = chan[tid];
    if (tid == 0) a.state[8] = n_sent;
}

// ---------------------------------------------------------------- leaf hash + 6 levels of a secure column's tree in ONE launch ----
// Round 6.  A 1024-lane block owns 4096 consecutive rows of <= 4 columns of one size (the composition tree, a FRI
// layer: the 4 coordinate columns of a secure column):
//   SRC_COLS     the rows are read from the columns;
//   SRC_FOLD     the rows ARE FriProver::commit's line fold of the previous layer (fold_line_dev_kernel's rule, alpha from the device
//                channel's record): the folded evaluations are written out (the next fold and the decommitment read them) and hashed.
// A lane hashes its 4 leaves and the 3 nodes above them in registers (7 compressions); the block's 1024 nodes two levels up shrink
// through LDS, one lane per node, while a level still fills whole waves (512 ... 64 nodes).  Every node goes to the tree; the levels
// above (64 nodes per block) are the level-by-level path's (build_inner_layers).  Replaces fold + leaf + two pair-level launches by one.
// What was measured on the way (profiles/r06_merkle_fused.txt): the block's cascade continued down to its root (10 barrier-separated
// levels, quads or lanes) makes a block live 17 compressions long with most of its waves parked — 28.8 G compressions/s on 2^24 leaf
// digests against 34 G/s of the pair-level kernels, and 73 us for ANY layer below 2^18 points against ~50 us of subtree + top launches:
// the level-by-level path was already at the compression rate, its launches are not what a big layer waits for.
enum { SRC_COLS = 1, SRC_FOLD = 2 };
struct FusedTreeArgs {
    u32* tree; int leaf_log;
    const u32* col[4]; u32 n_cols;                                                          // SRC_COLS (1 ... 4 columns)
    const u32* fsrc[4]; u32* fdst[4]; const u32* itw; u32 tw_log; const u32* alpha;         // SRC_FOLD (fsrc: 2^(leaf_log + 1) words each)
};
constexpr u32 FUSED_LDS_WORDS = 2047 * 8, FUSED_MIN_LOG = 12, FUSED_LEVELS = 6;   // LDS: 1024 + 512 + ... nodes, level of m nodes at word (m - 1) * 8; the launch builds levels leaf_log ... leaf_log - 6

template <int MODE>
__device__ __forceinline__ void node_hash(const u32* m, u32* h) {
#pragma unroll
    for (int k = 0; k < 8; k++) h[k] = MODE == 0 ? B2S_IV_D[k] : 0u;
    if (MODE == 0) { h[0] ^= 0x01010020u; b2s_compress(h, m, 64, 0xFFFFFFFFu); } else b2s_compress(h, m, 0, 0);
}
__device__ __forceinline__ void store_node(u32* tree, int level, u32 idx, const u32* h) {
    u32* o = tree + (((size_t)1 << level) - 1 + idx) * 8;
    gst4(o, make_uint4(h[0], h[1], h[2], h[3])); gst4(o + 4, make_uint4(h[4], h[5], h[6], h[7]));
}

#ifndef NX_FUSED_MINWAVES   // 8 waves per SIMD = two 1024-lane blocks per CU: a block's narrow cascade levels run beside the other block's wide ones
#define NX_FUSED_MINWAVES 8
#endif
template <int MODE, int SRC>
__global__ __launch_bounds__(1024, NX_FUSED_MINWAVES) void merkle_fused_kernel(FusedTreeArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32 fz[];
    const u32 tid = threadIdx.x, g = blockIdx.x * 1024 + tid;       // this lane's node two levels above the leaves
    const int in_log = a.leaf_log;
    u32 n1[16], n2[8];                                               // the two nodes one level up (one message), the node two levels up
    {
        u32 val[4][4];                                               // [column / coordinate][row]
        const u32 nc = SRC == SRC_FOLD ? 4u : a.n_cols;
        if (SRC == SRC_FOLD) {
            const QM31 alpha = qm(a.alpha[0], a.alpha[1], a.alpha[2], a.alpha[3]);
            const uint4 x4 = gld4(a.itw + ((1u << a.tw_log) - (2u << in_log) + 4 * g));
            const u32 xs[4] = {x4.x, x4.y, x4.z, x4.w};
            uint4 lo[4], hi[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { lo[q] = gld4(a.fsrc[q] + (size_t)8 * g); hi[q] = gld4(a.fsrc[q] + (size_t)8 * g + 4); }
#pragma unroll
            for (int r = 0; r < 4; r++) {                           // outputs 4 g + r: the pair (2 i, 2 i + 1) of every coordinate
                u32 e0[4], e1[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint4 w = r < 2 ? lo[q] : hi[q];
                    e0[q] = (r & 1) ? w.z : w.x; e1[q] = (r & 1) ? w.w : w.y;
                }
                const QM31 f0 = qm(e0[0], e0[1], e0[2], e0[3]), f1 = qm(e1[0], e1[1], e1[2], e1[3]);
                const QM31 o = q_add(q_add(f0, f1), q_mul(alpha, q_mul_m(q_sub(f0, f1), xs[r])));
                val[0][r] = o.a.a; val[1][r] = o.a.b; val[2][r] = o.b.a; val[3][r] = o.b.b;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) gst4(a.fdst[q] + (size_t)4 * g, make_uint4(val[q][0], val[q][1], val[q][2], val[q][3]));
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if ((u32)c < nc) { const uint4 v = gld4(a.col[c] + (size_t)4 * g); val[c][0] = v.x; val[c][1] = v.y; val[c][2] = v.z; val[c][3] = v.w; }
                else { val[c][0] = val[c][1] = val[c][2] = val[c][3] = 0; }
            }
        }
#pragma unroll
        for (int hlf = 0; hlf < 2; hlf++) {
            u32 pr[16];                                              // the two leaf digests under node hlf
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                const int r = 2 * hlf + s2;
                u32 m[16];
#pragma unroll
                for (int c = 0; c < 16; c++) m[c] = c < 4 ? val[c][r] : 0u;
                u32* d = pr + 8 * s2;
#pragma unroll
                for (int k = 0; k < 8; k++) d[k] = MODE == 0 ? B2S_IV_D[k] : 0u;
                if (MODE == 0) { d[0] ^= 0x01010020u; b2s_compress(d, m, 4u * nc, 0xFFFFFFFFu); } else b2s_compress(d, m, 0, 0);
                store_node(a.tree, in_log, 4 * g + r, d);
            }
            node_hash<MODE>(pr, n1 + 8 * hlf);
            store_node(a.tree, in_log - 1, 2 * g + hlf, n1 + 8 * hlf);
        }
    }
    node_hash<MODE>(n1, n2);
    store_node(a.tree, in_log - 2, g, n2);
    {
        u32* o = fz + (size_t)(1023 + tid) * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) o[k] = n2[k];
    }
    __syncthreads();
    int lvl = in_log - 3;                                            // levels in_log - 3 ... in_log - 6: 512 ... 64 nodes of the block, one lane per node
    for (u32 m = 512; m >= 64; m >>= 1, lvl--) {
        const u32* in_l = fz + (size_t)(2 * m - 1) * 8;
        u32* out_l = fz + (size_t)(m - 1) * 8;
        if (tid < m) {
            u32 msg[16], h[8];
#pragma unroll
            for (int k = 0; k < 16; k++) msg[k] = in_l[(size_t)tid * 16 + k];
            node_hash<MODE>(msg, h);
            if (m > 64) {
#pragma unroll
                for (int k = 0; k < 8; k++) out_l[(size_t)tid * 8 + k] = h[k];
            }
            store_node(a.tree, lvl, blockIdx.x * m + tid, h);
        }
        if (m > 64) __syncthreads();
    }
}

// An empty tree of 2^max_log leaves (one device allocation, layer k at node offset 2^k - 1).
int tree_alloc(nx_ctx* ctx, uint32_t max_log, nx_tree** out) {
    nx_tree* t = new nx_tree();
    t->ctx = ctx;
    uint32_t* buf = nullptr;
    int rc = dev_alloc(ctx, (((size_t)2 << max_log) - 1) * 32, (void**)&buf);
    if (rc != NX_OK) { delete t; return rc; }
    t->layers.resize(max_log + 1);
    for (uint32_t k = 0; k <= max_log; k++) t->layers[k] = buf + (((size_t)1 << k) - 1) * 8;
    *out = t;
    return NX_OK;
}

// The last n_layers layers in one launch; d_state is the device channel state (the launch reads and updates it), j0 the record index
// of the first of these layers.  evals[j] / trees[j]: device buffers the caller allocated (evals[0] = the input layer of log size
// log0; evals has n_layers + 1 entries).  Asynchronous on the context's stream.
int fri_tail(nx_ctx* ctx, const nx_twiddles* tw, u32* const* evals, u32* const* trees, int n_layers, int log0, u32* d_state, int j0) {
    if (n_layers < 1 || n_layers > FRI_TAIL_MAX_LAYERS || log0 < n_layers || log0 > 11 || (u32)log0 > tw->log_half) return set_err(ctx, NX_ERR_ARG, "fri_tail: bad layer range");
    FriTailArgs a;
    for (int j = 0; j <= n_layers; j++) a.eval[j] = evals[j];
    for (int j = 0; j < n_layers; j++) a.tree[j] = trees[j];
    a.n_layers = n_layers; a.log0 = log0; a.itw = tw->d_itw; a.tw_log = tw->log_half; a.state = d_state; a.j0 = j0;
    static std::atomic<uint64_t> attr_set{0};   // one bit per device: function attributes are per device
    if (!(attr_set.load() & (1ull << (ctx->device & 63)))) {
        NX_HIP(ctx, hipFuncSetAttribute((const void*)fri_tail_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        NX_HIP(ctx, hipFuncSetAttribute((const void*)fri_tail_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.fetch_or(1ull << (ctx->device & 63));
    }
    const size_t lds = ((((size_t)2 << log0) - 1) * 8 + 32) * 4;
    if (ctx->hash_mode == NX_HASH_BLAKE2S) hipLaunchKernelGGL(fri_tail_kernel<0>, dim3(1), dim3(1024), lds, ctx->stream, a);
    else hipLaunchKernelGGL(fri_tail_kernel<1>, dim3(1), dim3(1024), lds, ctx->stream, a);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

int fri_channel_step(nx_ctx* ctx, u32* d_state, const u32* d_root, int j) {
    hipLaunchKernelGGL(fri_channel_kernel, dim3(1), dim3(64), 0, ctx->stream, d_state, d_root, j);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

// GrindOps: nonce = base + thread; H(digest ‖ nonce_le64) is a single final 40-byte block.
__global__ void grind_kernel(const u32* __restrict__ digest, u32 pow_bits, u64 base, unsigned long long* result) {
    u64 nonce = base + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 h[8], m[16];
#pragma unroll
    for (int k = 0; k < 8; k++) { h[k] = B2S_IV_D[k]; m[k] = digest[k]; }
    h[0] ^= 0x01010020u;
    m[8] = (u32)nonce; m[9] = (u32)(nonce >> 32);
#pragma unroll
    for (int k = 10; k < 16; k++) m[k] = 0;
    b2s_compress(h, m, 40, 0xFFFFFFFFu);
    u32 tz;
    if (h[0]) tz = __ffs(h[0]) - 1;
    else if (h[1]) tz = 32 + __ffs(h[1]) - 1;
    else if (h[2]) tz = 64 + __ffs(h[2]) - 1;
    else if (h[3]) tz = 96 + __ffs(h[3]) - 1;
    else tz = 128;
    if (tz >= pow_bits) atomicMin(result, (unsigned long long)nonce);
}

int merkle_layer(nx_ctx* ctx, ColSet cols, u32 n_cols, const u32* prev, u32* out, u32 log);

int leaf_chain_launch(nx_ctx* ctx, hipStream_t stream, ColSet cs, u32 n_cols, u32 col_offset, u32 total_cols, const u32* state_in, u32* state_out,
                      u64 row_begin, u64 n_rows) {
    KTimer timer(ctx, NX_T_MERKLE, n_rows * (4ull * n_cols + 64), stream);
    dim3 block(256);
    dim3 grid((unsigned)((n_rows + 255) / 256));
    if (ctx->hash_mode == NX_HASH_BLAKE2S)
        hipLaunchKernelGGL(merkle_leaf_chain_kernel<0>, grid, block, 0, stream, cs, n_cols, col_offset, total_cols, state_in, state_out, row_begin, n_rows);
    else
        hipLaunchKernelGGL(merkle_leaf_chain_kernel<1>, grid, block, 0, stream, cs, n_cols, col_offset, total_cols, state_in, state_out, row_begin, n_rows);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}


static int build_inner_layers(nx_ctx* ctx, nx_tree* t, uint32_t max_log, const std::vector<const uint32_t*>& sorted, const std::vector<uint32_t>& logs);
static int fused_launch(nx_ctx* ctx, int src, FusedTreeArgs a, nx_tree* t) {
    static std::atomic<uint64_t> attr_set{0};   // one bit per device: function attributes are per device
    if (!(attr_set.load() & (1ull << (ctx->device & 63)))) {
        const void* fns[] = {(const void*)merkle_fused_kernel<0, SRC_COLS>, (const void*)merkle_fused_kernel<0, SRC_FOLD>, (const void*)merkle_fused_kernel<1, SRC_COLS>, (const void*)merkle_fused_kernel<1, SRC_FOLD>};
        for (const void* f : fns) NX_HIP(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.fetch_or(1ull << (ctx->device & 63));
    }
    const dim3 grid(1u << (a.leaf_log - 12)), block(1024);
    const size_t lds = (size_t)FUSED_LDS_WORDS * 4;
    const bool std_hash = ctx->hash_mode == NX_HASH_BLAKE2S;
    if (src == SRC_COLS) { if (std_hash) hipLaunchKernelGGL((merkle_fused_kernel<0, SRC_COLS>), grid, block, lds, ctx->stream, a); else hipLaunchKernelGGL((merkle_fused_kernel<1, SRC_COLS>), grid, block, lds, ctx->stream, a); }
    else { if (std_hash) hipLaunchKernelGGL((merkle_fused_kernel<0, SRC_FOLD>), grid, block, lds, ctx->stream, a); else hipLaunchKernelGGL((merkle_fused_kernel<1, SRC_FOLD>), grid, block, lds, ctx->stream, a); }
    NX_LAUNCH_CHECK(ctx);
    return build_inner_layers(ctx, t, (uint32_t)a.leaf_log - FUSED_LEVELS, {}, {});      // level leaf_log - 6 is hashed: the rest level by level
}
// MerkleProver::commit of n_cols <= 4 columns of one size: leaf hash and 6 levels in one launch, then the level-by-level path
int merkle_commit_fused(nx_ctx* ctx, const uint32_t* const* d_cols, uint32_t n_cols, uint32_t log, nx_tree** out) {
    if (n_cols < 1 || n_cols > 4 || log < FUSED_MIN_LOG || log > 30) return set_err(ctx, NX_ERR_ARG, "merkle_commit_fused: 1 .. 4 columns of 2^12 .. 2^30 rows");
    nx_tree* t = nullptr;
    NX_TRY(tree_alloc(ctx, log, &t));
    KTimer timer(ctx, NX_T_MERKLE, ((uint64_t)4 * n_cols + 128) << log);
    FusedTreeArgs a; memset(&a, 0, sizeof a);
    a.tree = t->layers[0]; a.leaf_log = (int)log; a.n_cols = n_cols;
    for (uint32_t k = 0; k < n_cols; k++) a.col[k] = d_cols[k];
    const int rc = fused_launch(ctx, SRC_COLS, a, t);
    if (rc != NX_OK) { nx_tree_destroy(t); return rc; }
    *out = t;
    return NX_OK;
}
// One inner layer of FriProver::commit: dst = fold_line(src, alpha) (2^(src_log - 1) points) and its Merkle tree; the fold, the leaf hash and 6 levels are one launch
int fri_fold_commit_fused(nx_ctx* ctx, const nx_twiddles* tw, const uint32_t* const* d_src4, uint32_t src_log, const uint32_t* d_alpha, uint32_t* const* d_dst4, nx_tree** out) {
    if (src_log < FUSED_MIN_LOG + 1 || src_log > tw->log_half) return set_err(ctx, NX_ERR_ARG, "fri_fold_commit_fused: layer size out of range");
    nx_tree* t = nullptr;
    NX_TRY(tree_alloc(ctx, src_log - 1, &t));
    KTimer timer(ctx, NX_T_MERKLE, (uint64_t)(16 + 128) << (src_log - 1));
    FusedTreeArgs a; memset(&a, 0, sizeof a);
    a.tree = t->layers[0]; a.leaf_log = (int)src_log - 1;
    for (int q = 0; q < 4; q++) { a.fsrc[q] = d_src4[q]; a.fdst[q] = d_dst4[q]; }
    a.itw = tw->d_itw; a.tw_log = tw->log_half; a.alpha = d_alpha;
    const int rc = fused_launch(ctx, SRC_FOLD, a, t);
    if (rc != NX_OK) { nx_tree_destroy(t); return rc; }
    *out = t;
    return NX_OK;
}

// inner layers [max_log-1 .. 0] above an already hashed layer max_log; `sorted`/`logs`: the smaller columns, size-descending
static int build_inner_layers(nx_ctx* ctx, nx_tree* t, uint32_t max_log, const std::vector<const uint32_t*>& sorted, const std::vector<uint32_t>& logs) {
    uint32_t* buf = t->layers[0];
    size_t ci = 0;
    const size_t n = sorted.size();
    int smallest_col_log = n ? (int)logs[n - 1] : (int)max_log;
    int top_fused = std::min(ctx->opt.merkle_top, std::min((int)max_log - 1, smallest_col_log - 1));
    // levels [top_fused + 1, SUBTREE_TOP] without injected columns: one launch (merkle_subtree_kernel) instead of one per level
    const int SUBTREE_TOP = ctx->opt.merkle_subtree;   // 0 = off
    for (int log = (int)max_log - 1; log >= 0; log--) {
        if (log == top_fused && log >= 1) {
            NX_TRY(launch_merkle_top(ctx, buf, log));
            break;
        }
        if (top_fused >= 1 && log <= SUBTREE_TOP && log >= 6 && log - top_fused >= 2 && log - top_fused <= 7 && smallest_col_log > log) {      // log >= 6: level log + 1 holds at least one block's 128 nodes ("merkle.top" below 5)
            NX_TRY(launch_merkle_subtree(ctx, buf, log + 1, log - top_fused));       // levels log .. top_fused + 1
            log = top_fused + 1;
            continue;
        }
        // two node-only levels (log and log - 1) above the fused region: one launch for both
        if (ctx->opt.merkle_pair_levels && log >= 13 && log - 1 > top_fused && !(top_fused >= 1 && log - 1 <= SUBTREE_TOP && log - 1 - top_fused >= 2 && log - 1 - top_fused <= 7)
            && (ci >= n || (int)logs[ci] < log - 1)) {
            const u32 nn = 1u << (log - 1);
            if (ctx->hash_mode == NX_HASH_BLAKE2S) hipLaunchKernelGGL(merkle_pair_levels_kernel<0>, dim3((nn + 255) / 256), dim3(256), 0, ctx->stream, t->layers[log + 1], t->layers[log], t->layers[log - 1], nn);
            else hipLaunchKernelGGL(merkle_pair_levels_kernel<1>, dim3((nn + 255) / 256), dim3(256), 0, ctx->stream, t->layers[log + 1], t->layers[log], t->layers[log - 1], nn);
            NX_LAUNCH_CHECK(ctx);
            log -= 1;
            continue;
        }
        size_t c0 = ci;
        while (ci < n && logs[ci] == (uint32_t)log) ci++;
        ColSet cs;
        NX_TRY(make_colset(ctx, (const uint32_t* const*)(sorted.data() + c0), (uint32_t)(ci - c0), &cs));
        NX_TRY(merkle_layer(ctx, cs, (uint32_t)(ci - c0), t->layers[log + 1], t->layers[log], (uint32_t)log));
    }
    return NX_OK;
}

int tree_pipe_begin(nx_ctx* ctx, uint32_t max_log, uint32_t total_leaf_cols, TreePipe* tp) {
    if (max_log > 30 || total_leaf_cols == 0) return set_err(ctx, NX_ERR_ARG, "tree_pipe_begin: bad shape");
    nx_tree* t = new nx_tree();
    t->ctx = ctx;
    uint32_t* buf = nullptr;
    size_t total_nodes = ((size_t)2 << max_log) - 1;
    { int rc0 = dev_alloc(ctx, total_nodes * 32, (void**)&buf); if (rc0 != NX_OK) { delete t; return rc0; } }
    t->layers.resize(max_log + 1);
    for (uint32_t k = 0; k <= max_log; k++) t->layers[k] = buf + (((size_t)1 << k) - 1) * 8;
    tp->tree = t; tp->max_log = max_log; tp->total_leaf_cols = total_leaf_cols; tp->absorbed = 0; tp->pending.clear(); tp->any_launch = false;
    return NX_OK;
}

// Hash every complete 16-column block handed in so far (all of them when `flush`): ordered after the work already on the
// main stream (the LDE that produced the columns), executed on the hash stream.
int tree_pipe_absorb(nx_ctx* ctx, TreePipe* tp, const uint32_t* const* d_cols, uint32_t n_cols, bool flush) {
    for (uint32_t i = 0; i < n_cols; i++) tp->pending.push_back(d_cols[i]);
    if (tp->absorbed + tp->pending.size() > tp->total_leaf_cols) return set_err(ctx, NX_ERR_ARG, "tree_pipe_absorb: more columns than announced");
    const bool is_end = tp->absorbed + tp->pending.size() == tp->total_leaf_cols;
    size_t take = (flush || is_end) && is_end ? tp->pending.size() : (tp->pending.size() / 16) * 16;
    if (take == 0) return NX_OK;
    ColSet cs; NX_TRY(make_colset(ctx, tp->pending.data(), (uint32_t)take, &cs));   // pointer table staged on the main stream
    hipStream_t hs = ctx->stream;
    if (tp->side_stream) {
        NX_HIP(ctx, hipEventRecord(ctx->hash_ev, ctx->stream));
        NX_HIP(ctx, hipStreamWaitEvent(ctx->hash_stream, ctx->hash_ev, 0));
        hs = ctx->hash_stream;
    }
    u32* leaves = tp->tree->layers[tp->max_log];
    NX_TRY(leaf_chain_launch(ctx, hs, cs, (u32)take, tp->absorbed, tp->total_leaf_cols, tp->absorbed ? leaves : nullptr, leaves, 0,
                             (u64)1 << tp->max_log));
    tp->absorbed += (uint32_t)take;
    tp->pending.erase(tp->pending.begin(), tp->pending.begin() + take);
    tp->any_launch = true;
    return NX_OK;
}

int tree_pipe_finish(nx_ctx* ctx, TreePipe* tp, const uint32_t* const* d_small_cols, const uint32_t* small_logs, uint32_t n_small, nx_tree** out) {
    int rc = tree_pipe_absorb(ctx, tp, nullptr, 0, true);
    if (rc == NX_OK && tp->absorbed != tp->total_leaf_cols) rc = set_err(ctx, NX_ERR_ARG, "tree_pipe_finish: fewer leaf columns than announced");
    if (rc == NX_OK && tp->side_stream) {
        hipError_t e = hipEventRecord(ctx->hash_ev, ctx->hash_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->hash_ev, 0);
        if (e != hipSuccess) rc = hip_fail(ctx, e, "tree_pipe_finish", __FILE__, __LINE__);
    }
    if (rc == NX_OK) {
        std::vector<uint32_t> order(n_small);
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return small_logs[a] > small_logs[b]; });
        std::vector<const uint32_t*> sorted(n_small); std::vector<uint32_t> logs(n_small);
        for (uint32_t i = 0; i < n_small; i++) { sorted[i] = d_small_cols[order[i]]; logs[i] = small_logs[order[i]]; }
        KTimer timer(ctx, NX_T_MERKLE, (uint64_t)96 << tp->max_log);
        rc = build_inner_layers(ctx, tp->tree, tp->max_log, sorted, logs);
    }
    if (rc != NX_OK) { (void)hipStreamSynchronize(ctx->hash_stream); nx_tree_destroy(tp->tree); tp->tree = nullptr; return rc; }
    *out = tp->tree; tp->tree = nullptr;
    return NX_OK;
}

int merkle_layer(nx_ctx* ctx, ColSet cols, u32 n_cols, const u32* prev, u32* out, u32 log) {
    u32 n = 1u << log;
    dim3 grid((n + 255) / 256), block(256);
    if (ctx->hash_mode == NX_HASH_BLAKE2S) hipLaunchKernelGGL(merkle_layer_kernel<0>, grid, block, 0, ctx->stream, cols, n_cols, prev, out, n);
    else hipLaunchKernelGGL(merkle_layer_kernel<1>, grid, block, 0, ctx->stream, cols, n_cols, prev, out, n);
    NX_LAUNCH_CHECK(ctx);
    return NX_OK;
}

}  // namespace nx

using namespace nx;

extern "C" {

int nx_merkle_commit(nx_ctx* ctx, const uint32_t* const* d_cols, const uint32_t* log_sizes, uint32_t n_cols, nx_tree** out) {
    NX_GUARD(ctx);
    if (!ctx || !out) return set_err(ctx, NX_ERR_ARG, "nx_merkle_commit: NULL argument");
    // stable sort by size, descending (MerkleProver::commit)
    std::vector<uint32_t> order(n_cols);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return log_sizes[a] > log_sizes[b]; });
    uint32_t max_log = n_cols ? log_sizes[order[0]] : 0;
    if (max_log > 30) return set_err(ctx, NX_ERR_ARG, "nx_merkle_commit: column too large");
    std::vector<const uint32_t*> sorted(n_cols);
    for (uint32_t i = 0; i < n_cols; i++) sorted[i] = d_cols[order[i]];

    bool aligned16 = true;                                         // the fused launch reads 4 rows of a column as one 16-byte word
    for (uint32_t i = 0; i < n_cols; i++) aligned16 = aligned16 && ((uintptr_t)sorted[i] & 15u) == 0;
    if (ctx->opt.merkle_fused && aligned16 && n_cols >= 1 && n_cols <= 4 && max_log >= (uint32_t)std::max(ctx->opt.merkle_fused, (int)FUSED_MIN_LOG) && log_sizes[order[n_cols - 1]] == max_log)
        return merkle_commit_fused(ctx, sorted.data(), n_cols, max_log, out);      // a secure column (composition tree, a FRI layer): leaf hash and 6 levels in one launch
    nx_tree* t = new nx_tree();
    t->ctx = ctx;
    uint32_t* buf = nullptr;
    size_t total_nodes = ((size_t)2 << max_log) - 1;
    { int rc0 = dev_alloc(ctx, total_nodes * 32, (void**)&buf); if (rc0 != NX_OK) { delete t; return rc0; } }
    // layers[k] at node offset 2^k - 1 (root first) inside one allocation
    t->layers.resize(max_log + 1);
    for (uint32_t k = 0; k <= max_log; k++) t->layers[k] = buf + (((size_t)1 << k) - 1) * 8;

    uint64_t alg_bytes = 0;
    for (uint32_t i = 0; i < n_cols; i++) alg_bytes += 4ull << log_sizes[i];
    alg_bytes += (128ull << max_log);  // 32 B written per leaf + ~96 B per leaf for all inner layers
    KTimer timer(ctx, NX_T_MERKLE, alg_bytes);

    int rc = NX_OK;
    // leaf layer: the columns of the largest size (or the empty message), then the inner layers with the smaller columns
    size_t n_leaf = 0;
    while (n_leaf < n_cols && log_sizes[order[n_leaf]] == max_log) n_leaf++;
    {
        ColSet cs;
        rc = make_colset(ctx, (const uint32_t* const*)sorted.data(), (uint32_t)n_leaf, &cs);
        if (rc == NX_OK) rc = merkle_layer(ctx, cs, (uint32_t)n_leaf, nullptr, t->layers[max_log], max_log);
    }
    if (rc == NX_OK && max_log > 0) {
        std::vector<const uint32_t*> small(sorted.begin() + n_leaf, sorted.end());
        std::vector<uint32_t> small_logs(n_cols - n_leaf);
        for (size_t i = n_leaf; i < n_cols; i++) small_logs[i - n_leaf] = log_sizes[order[i]];
        rc = build_inner_layers(ctx, t, max_log, small, small_logs);
    }
    if (rc != NX_OK) { dev_free(ctx, buf); delete t; return rc; }
    *out = t;
    return NX_OK;
}


int nx_merkle_leaf_chain(nx_ctx* ctx, const uint32_t* const* d_cols, uint32_t n_cols, uint32_t log_size, uint32_t col_offset,
                         uint32_t total_cols, const uint32_t* d_state_in, uint32_t* d_state_out, uint64_t row_begin, uint64_t n_rows) {
    NX_GUARD(ctx);
    if (!ctx || !d_state_out || (n_cols && !d_cols)) return set_err(ctx, NX_ERR_ARG, "nx_merkle_leaf_chain: NULL argument");
    if (n_cols == 0 || col_offset + (uint64_t)n_cols > total_cols) return set_err(ctx, NX_ERR_ARG, "nx_merkle_leaf_chain: empty shard or column range outside the layer");
    if (col_offset % 16) return set_err(ctx, NX_ERR_ARG, "nx_merkle_leaf_chain: col_offset must be a multiple of 16 (one Blake2s block)");
    if (col_offset + n_cols != total_cols && n_cols % 16) return set_err(ctx, NX_ERR_ARG, "nx_merkle_leaf_chain: an inner shard must hold a multiple of 16 columns");
    if ((col_offset == 0) != (d_state_in == nullptr)) return set_err(ctx, NX_ERR_ARG, "nx_merkle_leaf_chain: d_state_in must be NULL exactly for the shard at column 0");
    if (log_size > 30 || row_begin + n_rows > ((uint64_t)1 << log_size)) return set_err(ctx, NX_ERR_ARG, "nx_merkle_leaf_chain: row range outside the column");
    if (n_rows == 0) return NX_OK;
    ColSet cs; NX_TRY(make_colset(ctx, d_cols, n_cols, &cs));
    return leaf_chain_launch(ctx, ctx->stream, cs, n_cols, col_offset, total_cols, d_state_in, d_state_out, row_begin, n_rows);
}

int nx_merkle_from_leaves(nx_ctx* ctx, const uint32_t* d_leaf_digests, uint32_t log_size, nx_tree** out) {
    NX_GUARD(ctx);
    if (!ctx || !out || !d_leaf_digests) return set_err(ctx, NX_ERR_ARG, "nx_merkle_from_leaves: NULL argument");
    if (log_size > 30) return set_err(ctx, NX_ERR_ARG, "nx_merkle_from_leaves: layer too large");
    nx_tree* t = new nx_tree();
    t->ctx = ctx;
    uint32_t* buf = nullptr;
    size_t total_nodes = ((size_t)2 << log_size) - 1;
    { int rc0 = dev_alloc(ctx, total_nodes * 32, (void**)&buf); if (rc0 != NX_OK) { delete t; return rc0; } }
    t->layers.resize(log_size + 1);
    for (uint32_t k = 0; k <= log_size; k++) t->layers[k] = buf + (((size_t)1 << k) - 1) * 8;
    int rc = nx_copy(ctx, t->layers[log_size], d_leaf_digests, (size_t)8 << log_size);
    KTimer timer(ctx, NX_T_MERKLE, (uint64_t)96 << log_size);
    const int top_fused = std::min(ctx->opt.merkle_top, (int)log_size - 1);
    ColSet none; none.base = nullptr; none.stride = 0; none.table = nullptr;
    for (int log = (int)log_size - 1; log >= 0 && rc == NX_OK; log--) {
        if (log == top_fused && log >= 1) {
            rc = launch_merkle_top(ctx, buf, log);
            break;
        }
        rc = merkle_layer(ctx, none, 0, t->layers[log + 1], t->layers[log], (uint32_t)log);
    }
    if (rc != NX_OK) { dev_free(ctx, buf); delete t; return rc; }
    *out = t;
    return NX_OK;
}

int nx_merkle_root(nx_ctx* ctx, const nx_tree* tree, uint8_t root[32]) {
    NX_GUARD(ctx);
    if (!ctx || !tree || !root || tree->layers.empty()) return set_err(ctx, NX_ERR_ARG, "nx_merkle_root: NULL argument");
    return nx_download(ctx, (uint32_t*)root, tree->layers[0], 8);
}
uint32_t nx_merkle_n_layers(const nx_tree* tree) { return tree ? (uint32_t)tree->layers.size() : 0; }
const uint32_t* nx_merkle_layer(const nx_tree* tree, uint32_t k) { return tree && k < tree->layers.size() ? tree->layers[k] : nullptr; }

void nx_tree_destroy(nx_tree* tree) {
    NX_GUARD(tree ? tree->ctx : nullptr);
    if (!tree) return;
    if (!tree->layers.empty()) dev_free(tree->ctx, tree->layers[0]);
    delete tree;
}

int nx_grind(nx_ctx* ctx, const uint8_t digest[32], uint32_t pow_bits, uint64_t* nonce) {
    NX_GUARD(ctx);
    if (!ctx || !digest || !nonce) return set_err(ctx, NX_ERR_ARG, "nx_grind: NULL argument");
    if (pow_bits > 64) return set_err(ctx, NX_ERR_ARG, "nx_grind: pow_bits > 64");
    struct { uint32_t d[8]; unsigned long long res; } h;
    memcpy(h.d, digest, 32);
    h.res = ~0ull;
    uint8_t* d = nullptr;
    NX_TRY(dev_alloc(ctx, sizeof h, (void**)&d));
    hipError_t e = hipMemcpyAsync(d, &h, sizeof h, hipMemcpyHostToDevice, ctx->stream);
    // the first launch tries 64 x the expected number of nonces (2^pow_bits): enough with probability 1 - e^-64, ~10 us for the
    // default 10 bits instead of the 108 us of a fixed 2^22 batch; later launches (never needed in practice) double up to 2^24
    uint64_t batch = std::min<uint64_t>(1ull << 24, std::max<uint64_t>(1ull << 16, 64ull << std::min<uint32_t>(pow_bits, 18)));
    uint64_t base = 0;
    unsigned long long res = ~0ull;
    while (e == hipSuccess) {
        hipLaunchKernelGGL(grind_kernel, dim3((unsigned)(batch / 256)), dim3(256), 0, ctx->stream, (const u32*)d, pow_bits, base,
                           (unsigned long long*)(d + 32));
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(&res, d + 32, 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (res != ~0ull) break;
        base += batch;
        batch = std::min<uint64_t>(1ull << 24, batch * 2);
    }
    (void)hipStreamSynchronize(ctx->stream);
    dev_free(ctx, d);
    if (e != hipSuccess) return hip_fail(ctx, e, "nx_grind", __FILE__, __LINE__);
    *nonce = res;
    return NX_OK;
}

}  // extern "C"
