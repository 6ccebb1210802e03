// ORACLE — TEST INFRASTRUCTURE ONLY (see fields.h header).  PARITY UNPINNED.
//
// Mixed-degree Blake2s Merkle tree and the Blake2s Fiat–Shamir channel.  Restates Stwo
//   core/vcs/blake2_merkle.rs   (Blake2sMerkleHasher::hash_node, Blake2sMerkleChannel::mix_root)
//   prover/vcs/prover.rs        (MerkleProver::{commit, decommit})
//   core/vcs/verifier.rs        (MerkleVerifier::verify)
//   core/channel/blake2s.rs     (Blake2sChannel)
//   prover/backend/cpu/grind.rs (GrindOps::grind)
// Reference call sites: prover/src/machine.rs:197-206 (channel seeding), :228,:237,:263 (commit),
// :411-416 (root comparison in verify).
//
// UNVERIFIED UPSTREAM RULE kept switchable: HASH_STD = standard
// Blake2s-256 over (left ‖ right ‖ column values as LE u32); HASH_RAW0 = the older rule, raw
// compression chaining from an all-zero state with t = f = 0 over 64-byte zero-padded blocks.
#pragma once
#include <thread>
#include <vector>
#include <map>
#include <algorithm>
#include <cstring>
#include <string>
#include "fields.h"
#include "blake2s.h"

namespace orc {

enum HashMode { HASH_STD = 0, HASH_RAW0 = 1 };

struct Hash { uint32_t w[8]; };
static inline bool hash_eq(const Hash& a, const Hash& b) { return memcmp(a.w, b.w, 32) == 0; }

static inline Hash hash_node(const Hash* left, const Hash* right, const u32* vals, size_t nvals, int mode) {
    Hash out;
    if (mode == HASH_STD) {
        Blake2s s;
        if (left) { s.update(left->w, 32); s.update(right->w, 32); }
        if (nvals) s.update(vals, nvals * 4);  // little-endian host
        s.finalize((uint8_t*)out.w);
    } else {
        uint32_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t m[16];
        if (left) { memcpy(m, left->w, 32); memcpy(m + 8, right->w, 32); b2s_compress(st, m, 0, 0, 0, 0); }
        for (size_t i = 0; i < nvals; i += 16) {
            size_t k = nvals - i < 16 ? nvals - i : 16;
            memset(m, 0, sizeof m); memcpy(m, vals + i, k * 4);
            b2s_compress(st, m, 0, 0, 0, 0);
        }
        memcpy(out.w, st, 32);
    }
    return out;
}

struct ColRef { const u32* data; int log; };

struct MerkleTree {
    std::vector<std::vector<Hash>> layers;  // layers[k] has 2^k nodes; layers[0][0] is the root
    Hash root() const { return layers[0][0]; }
};

// MerkleProver::commit — columns in commit order; stable sort by size descending.
// host threads used by merkle_commit for layers of >= 4096 nodes (set by the prove drivers; the result does not depend on it)
static inline int& merkle_n_threads() { static int v = 1; return v; }

static inline MerkleTree merkle_commit(std::vector<ColRef> cols, int mode) {
    MerkleTree t;
    if (cols.empty()) { t.layers.push_back({hash_node(nullptr, nullptr, nullptr, 0, mode)}); return t; }
    std::stable_sort(cols.begin(), cols.end(), [](const ColRef& a, const ColRef& b) { return a.log > b.log; });
    int max_log = cols[0].log;
    t.layers.resize(max_log + 1);
    size_t ci = 0;
    std::vector<u32> vals;
    for (int log = max_log; log >= 0; log--) {
        std::vector<const u32*> lc;
        while (ci < cols.size() && cols[ci].log == log) lc.push_back(cols[ci++].data);
        size_t n = (size_t)1 << log;
        t.layers[log].resize(n);
        const std::vector<Hash>* prev = log < max_log ? &t.layers[log + 1] : nullptr;
        auto range = [&](size_t i0, size_t i1) {
            std::vector<u32> v(lc.size());
            for (size_t i = i0; i < i1; i++) {
                for (size_t c = 0; c < lc.size(); c++) v[c] = lc[c][i];
                t.layers[log][i] = hash_node(prev ? &(*prev)[2 * i] : nullptr, prev ? &(*prev)[2 * i + 1] : nullptr, v.data(), v.size(), mode);
            }
        };
        const int nt = n >= 4096 ? merkle_n_threads() : 1;
        if (nt <= 1) range(0, n);
        else {
            std::vector<std::thread> th;
            for (int k = 0; k < nt; k++) th.emplace_back(range, n * k / nt, n * (k + 1) / nt);
            for (auto& x : th) x.join();
        }
    }
    return t;
}

struct MerkleDecommitment {
    std::vector<Hash> hash_witness;
    std::vector<u32> column_witness;
};

// MerkleProver::decommit.  queries_per_log: log_size -> sorted, deduplicated positions.
static inline void merkle_decommit(const MerkleTree& t, const std::map<int, std::vector<size_t>>& queries_per_log,
                                   std::vector<ColRef> cols, std::vector<u32>& queried_values, MerkleDecommitment& d) {
    std::stable_sort(cols.begin(), cols.end(), [](const ColRef& a, const ColRef& b) { return a.log > b.log; });
    size_t ci = 0;
    std::vector<size_t> last_layer_queries;
    for (int log = (int)t.layers.size() - 1; log >= 0; log--) {
        std::vector<const u32*> lc;
        while (ci < cols.size() && cols[ci].log == log) lc.push_back(cols[ci++].data);
        const std::vector<Hash>* prev = (size_t)(log + 1) < t.layers.size() ? &t.layers[log + 1] : nullptr;
        static const std::vector<size_t> empty;
        auto it = queries_per_log.find(log);
        const std::vector<size_t>& lq = it == queries_per_log.end() ? empty : it->second;
        size_t pi = 0, qi = 0;
        std::vector<size_t> total;
        while (pi < last_layer_queries.size() || qi < lq.size()) {
            size_t node;
            if (pi < last_layer_queries.size() && qi < lq.size()) node = std::min(last_layer_queries[pi] / 2, lq[qi]);
            else if (pi < last_layer_queries.size()) node = last_layer_queries[pi] / 2;
            else node = lq[qi];
            if (prev) {
                if (pi < last_layer_queries.size() && last_layer_queries[pi] == 2 * node) pi++;
                else d.hash_witness.push_back((*prev)[2 * node]);
                if (pi < last_layer_queries.size() && last_layer_queries[pi] == 2 * node + 1) pi++;
                else d.hash_witness.push_back((*prev)[2 * node + 1]);
            }
            if (qi < lq.size() && lq[qi] == node) { qi++; for (auto c : lc) queried_values.push_back(c[node]); }
            else for (auto c : lc) d.column_witness.push_back(c[node]);
            total.push_back(node);
        }
        last_layer_queries.swap(total);
    }
}

// MerkleVerifier::verify.  column_log_sizes in commit order; queried_values as produced by
// merkle_decommit.  Returns "" on success or the name of the failure.
static inline std::string merkle_verify(const Hash& root, const std::vector<int>& column_log_sizes,
                                        const std::map<int, std::vector<size_t>>& queries_per_log,
                                        const std::vector<u32>& queried_values, const MerkleDecommitment& d, int mode) {
    if (column_log_sizes.empty()) {  // tree without columns: single node = hash of nothing
        if (!d.hash_witness.empty() || !d.column_witness.empty()) return "WitnessTooLong";
        if (!queried_values.empty()) return "TooManyQueriedValues";
        return hash_eq(hash_node(nullptr, nullptr, nullptr, 0, mode), root) ? "" : "RootMismatch";
    }
    int max_log = 0;
    for (int l : column_log_sizes) max_log = std::max(max_log, l);
    std::map<int, size_t> n_cols_per_log;
    for (int l : column_log_sizes) n_cols_per_log[l]++;
    size_t qv = 0, hw = 0, cw = 0;
    std::vector<std::pair<size_t, Hash>> last;  // (node index, hash) of the previous (larger) layer
    for (int log = max_log; log >= 0; log--) {
        size_t n_cols = n_cols_per_log.count(log) ? n_cols_per_log[log] : 0;
        static const std::vector<size_t> empty;
        auto it = queries_per_log.find(log);
        const std::vector<size_t>& lq = it == queries_per_log.end() ? empty : it->second;
        bool has_prev = log < max_log;
        size_t pi = 0, qi = 0;
        std::vector<std::pair<size_t, Hash>> total;
        std::vector<u32> vals(n_cols);
        while (pi < last.size() || qi < lq.size()) {
            size_t node;
            if (pi < last.size() && qi < lq.size()) node = std::min(last[pi].first / 2, lq[qi]);
            else if (pi < last.size()) node = last[pi].first / 2;
            else node = lq[qi];
            Hash l, r;
            if (has_prev) {
                if (pi < last.size() && last[pi].first == 2 * node) l = last[pi++].second;
                else { if (hw >= d.hash_witness.size()) return "WitnessTooShort"; l = d.hash_witness[hw++]; }
                if (pi < last.size() && last[pi].first == 2 * node + 1) r = last[pi++].second;
                else { if (hw >= d.hash_witness.size()) return "WitnessTooShort"; r = d.hash_witness[hw++]; }
            }
            if (qi < lq.size() && lq[qi] == node) {
                qi++;
                if (qv + n_cols > queried_values.size()) return "TooFewQueriedValues";
                for (size_t c = 0; c < n_cols; c++) vals[c] = queried_values[qv++];
            } else {
                if (cw + n_cols > d.column_witness.size()) return "WitnessTooShort";
                for (size_t c = 0; c < n_cols; c++) vals[c] = d.column_witness[cw++];
            }
            total.push_back({node, hash_node(has_prev ? &l : nullptr, has_prev ? &r : nullptr, vals.data(), n_cols, mode)});
        }
        last.swap(total);
    }
    if (hw != d.hash_witness.size() || cw != d.column_witness.size()) return "WitnessTooLong";
    if (qv != queried_values.size()) return "TooManyQueriedValues";
    if (last.size() != 1 || !hash_eq(last[0].second, root)) return "RootMismatch";
    return "";
}

// ---- Blake2sChannel (core/channel/blake2s.rs) ----
struct Channel {
    Hash digest;
    u32 n_challenges, n_sent;
    Channel() { memset(digest.w, 0, 32); n_challenges = 0; n_sent = 0; }
    void update_digest(const Hash& h) { digest = h; n_challenges++; n_sent = 0; }
    void mix_root(const Hash& root) {  // Blake2sMerkleChannel::mix_root = H(digest ‖ root)
        Blake2s s; s.update(digest.w, 32); s.update(root.w, 32);
        Hash h; s.finalize((uint8_t*)h.w); update_digest(h);
    }
    void mix_u32s(const u32* data, size_t n) {
        Blake2s s; s.update(digest.w, 32); s.update(data, n * 4);
        Hash h; s.finalize((uint8_t*)h.w); update_digest(h);
    }
    void mix_u64(u64 v) { u32 d[2] = {(u32)v, (u32)(v >> 32)}; mix_u32s(d, 2); }
    void mix_felts(const QM31* f, size_t n) {
        std::vector<u32> w(n * 4);
        for (size_t i = 0; i < n; i++) qm31_store(&w[4 * i], f[i]);
        mix_u32s(w.data(), w.size());
    }
    // draw_random_bytes: H(digest ‖ n_sent as LE padded to 32 bytes ‖ 0x00)
    void draw_u32s(u32 out[8]) {
        uint8_t in[65]; memset(in, 0, sizeof in);
        memcpy(in, digest.w, 32);
        memcpy(in + 32, &n_sent, 4);
        in[64] = 0;
        n_sent++;
        blake2s_hash(in, 65, (uint8_t*)out);
    }
    void draw_base_felts(u32 out[8]) {
        for (;;) {
            u32 w[8]; draw_u32s(w);
            bool ok = true;
            for (int i = 0; i < 8; i++) if (w[i] >= 2 * P) ok = false;
            if (ok) { for (int i = 0; i < 8; i++) out[i] = m31_reduce(w[i]); return; }
        }
    }
    QM31 draw_secure_felt() { u32 f[8]; draw_base_felts(f); return qm31(f[0], f[1], f[2], f[3]); }
    std::vector<QM31> draw_secure_felts(size_t n) {
        std::vector<QM31> r; u32 f[8]; int have = 0, pos = 0;
        while (r.size() < n) {
            u32 q[4];
            for (int k = 0; k < 4; k++) { if (pos == have) { draw_base_felts(f); have = 8; pos = 0; } q[k] = f[pos++]; }
            r.push_back(qm31(q[0], q[1], q[2], q[3]));
        }
        return r;
    }
    // trailing zeros of the first 16 digest bytes read as a LE u128
    static u32 trailing_zeros(const Hash& h) {
        for (int i = 0; i < 4; i++) if (h.w[i]) return 32 * i + (u32)__builtin_ctz(h.w[i]);
        return 128;
    }
    bool verify_pow_nonce(u32 n_bits, u64 nonce) const {
        Channel c = *this; c.mix_u64(nonce); return trailing_zeros(c.digest) >= n_bits;
    }
    u64 grind(u32 pow_bits) const {  // smallest valid nonce
        for (u64 nonce = 0;; nonce++) if (verify_pow_nonce(pow_bits, nonce)) return nonce;
    }
};

}  // namespace orc
