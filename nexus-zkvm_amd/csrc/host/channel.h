// Host-side Fiat–Shamir transcript of the product: Blake2s-256 and Blake2sChannel.
//
// Mirrors Stwo `core/channel/blake2s.rs::Blake2sChannel` and `Blake2sMerkleChannel::mix_root`
// (constructed at reference prover/src/machine.rs:197, seeded at :198-206, claimed sums mixed at
// :262).  The transcript is tiny and inherently sequential, so it stays on the host
// ("negligible, host").  This is product code: it shares nothing with oracle/.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../field.cuh"

namespace nxhip {

using nx::QM31;
using nx::u32;
using nx::u64;

struct Blake2sState {
    uint32_t h[8];
    uint8_t buf[64];
    size_t buflen;
    uint64_t t;

    static inline uint32_t rotr(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }
    static void compress(uint32_t h[8], const uint8_t block[64], uint64_t t, bool last) {
        static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
        static const uint8_t S[10][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
        uint32_t m[16], v[16];
        memcpy(m, block, 64);
        for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = IV[i]; }
        v[12] ^= (uint32_t)t; v[13] ^= (uint32_t)(t >> 32);
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint32_t x, uint32_t y) {
            v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 16);
            v[c] = v[c] + v[d]; v[b] = rotr(v[b] ^ v[c], 12);
            v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 8);
            v[c] = v[c] + v[d]; v[b] = rotr(v[b] ^ v[c], 7);
        };
        for (int r = 0; r < 10; r++) {
            const uint8_t* s = S[r];
            G(0, 4, 8, 12, m[s[0]], m[s[1]]); G(1, 5, 9, 13, m[s[2]], m[s[3]]);
            G(2, 6, 10, 14, m[s[4]], m[s[5]]); G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]); G(1, 6, 11, 12, m[s[10]], m[s[11]]);
            G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
    }
    Blake2sState() {
        static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
        for (int i = 0; i < 8; i++) h[i] = IV[i];
        h[0] ^= 0x01010020u;
        buflen = 0; t = 0;
    }
    void update(const void* data, size_t len) {
        const uint8_t* p = (const uint8_t*)data;
        while (len) {
            if (buflen == 64) { t += 64; compress(h, buf, t, false); buflen = 0; }
            size_t take = 64 - buflen; if (take > len) take = len;
            memcpy(buf + buflen, p, take); buflen += take; p += take; len -= take;
        }
    }
    void finalize(uint8_t out[32]) {
        t += buflen;
        memset(buf + buflen, 0, 64 - buflen);
        compress(h, buf, t, true);
        memcpy(out, h, 32);
    }
};

struct Blake2sHash { uint32_t w[8]; };

class Blake2sChannel {
  public:
    Blake2sHash digest;
    uint32_t n_challenges = 0, n_sent = 0;
    Blake2sChannel() { memset(digest.w, 0, 32); }
    void update_digest(const Blake2sHash& h) { digest = h; n_challenges++; n_sent = 0; }
    void mix_root(const Blake2sHash& root) {
        Blake2sState s; s.update(digest.w, 32); s.update(root.w, 32);
        Blake2sHash h; s.finalize((uint8_t*)h.w); update_digest(h);
    }
    void mix_u32s(const uint32_t* d, size_t n) {
        Blake2sState s; s.update(digest.w, 32); s.update(d, n * 4);
        Blake2sHash h; s.finalize((uint8_t*)h.w); update_digest(h);
    }
    void mix_u64(uint64_t v) { uint32_t d[2] = {(uint32_t)v, (uint32_t)(v >> 32)}; mix_u32s(d, 2); }
    void mix_felts(const std::vector<QM31>& f) {
        std::vector<uint32_t> w(f.size() * 4);
        for (size_t i = 0; i < f.size(); i++) nx::q_store(&w[4 * i], f[i]);
        mix_u32s(w.data(), w.size());
    }
    void draw_u32s(uint32_t out[8]) {
        uint8_t in[65]; memset(in, 0, sizeof in);
        memcpy(in, digest.w, 32); memcpy(in + 32, &n_sent, 4);
        n_sent++;
        Blake2sState s; s.update(in, 65); s.finalize((uint8_t*)out);
    }
    void draw_base_felts(uint32_t out[8]) {
        for (;;) {
            uint32_t w[8]; draw_u32s(w);
            bool ok = true;
            for (int i = 0; i < 8; i++) ok = ok && w[i] < 2u * nx::P;
            if (ok) { for (int i = 0; i < 8; i++) out[i] = w[i] >= nx::P ? w[i] - nx::P : w[i]; return; }
        }
    }
    QM31 draw_secure_felt() { uint32_t f[8]; draw_base_felts(f); return nx::qm(f[0], f[1], f[2], f[3]); }
    // Channel::draw_felts(n): chunks of 4 from the concatenated base-felt draws (two secure felts per draw) [upstream-recollection]
    std::vector<QM31> draw_secure_felts(size_t n) {
        std::vector<QM31> out;
        while (out.size() < n) {
            uint32_t f[8]; draw_base_felts(f);
            out.push_back(nx::qm(f[0], f[1], f[2], f[3]));
            if (out.size() < n) out.push_back(nx::qm(f[4], f[5], f[6], f[7]));
        }
        return out;
    }
};

}  // namespace nxhip
