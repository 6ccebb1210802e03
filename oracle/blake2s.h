// ORACLE — TEST INFRASTRUCTURE ONLY (see fields.h header).  PARITY UNPINNED.
//
// Blake2s-256 (RFC 7693) plus the raw compression function.  Restates what Stwo's
// `Blake2sHasher` (core/vcs/blake2_hash.rs: wraps the `blake2` crate's Blake2s256) and
// `compress` (prover/backend/cpu/blake2s.rs) compute.  Pinned here against the RFC 7693
// known-answer vectors (tests/test_oracle_primitives.py).
#pragma once
#include <cstdint>
#include <cstring>
#include <cstddef>

namespace orc {

static const uint32_t B2S_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                   0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const uint8_t B2S_SIGMA[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};

static inline uint32_t rotr32(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }

// compress(h, m, t0, t1, f0, f1): one Blake2s compression; h updated in place.
static inline void b2s_compress(uint32_t h[8], const uint32_t m[16], uint32_t t0, uint32_t t1, uint32_t f0, uint32_t f1) {
    uint32_t v[16];
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = B2S_IV[i]; }
    v[12] ^= t0; v[13] ^= t1; v[14] ^= f0; v[15] ^= f1;
#define B2S_G(a, b, c, d, x, y)                 \
    v[a] = v[a] + v[b] + (x); v[d] = rotr32(v[d] ^ v[a], 16); \
    v[c] = v[c] + v[d];       v[b] = rotr32(v[b] ^ v[c], 12); \
    v[a] = v[a] + v[b] + (y); v[d] = rotr32(v[d] ^ v[a], 8);  \
    v[c] = v[c] + v[d];       v[b] = rotr32(v[b] ^ v[c], 7);
    for (int r = 0; r < 10; r++) {
        const uint8_t* s = B2S_SIGMA[r];
        B2S_G(0, 4, 8, 12, m[s[0]], m[s[1]]);
        B2S_G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        B2S_G(2, 6, 10, 14, m[s[4]], m[s[5]]);
        B2S_G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        B2S_G(0, 5, 10, 15, m[s[8]], m[s[9]]);
        B2S_G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        B2S_G(2, 7, 8, 13, m[s[12]], m[s[13]]);
        B2S_G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#undef B2S_G
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}

// Incremental standard Blake2s-256 (no key).
struct Blake2s {
    uint32_t h[8];
    uint8_t buf[64];
    size_t buflen;
    uint64_t t;
    Blake2s() { reset(); }
    void reset() {
        for (int i = 0; i < 8; i++) h[i] = B2S_IV[i];
        h[0] ^= 0x01010020u;  // digest length 32, fanout 1, depth 1
        buflen = 0; t = 0;
    }
    void update(const void* data, size_t len) {
        const uint8_t* p = (const uint8_t*)data;
        while (len) {
            if (buflen == 64) {  // buffer full and more data follows: not the last block
                t += 64;
                uint32_t m[16]; memcpy(m, buf, 64);
                b2s_compress(h, m, (uint32_t)t, (uint32_t)(t >> 32), 0, 0);
                buflen = 0;
            }
            size_t take = 64 - buflen; if (take > len) take = len;
            memcpy(buf + buflen, p, take);
            buflen += take; p += take; len -= take;
        }
    }
    void finalize(uint8_t out[32]) {
        t += buflen;
        memset(buf + buflen, 0, 64 - buflen);
        uint32_t m[16]; memcpy(m, buf, 64);
        b2s_compress(h, m, (uint32_t)t, (uint32_t)(t >> 32), 0xFFFFFFFFu, 0);
        memcpy(out, h, 32);
    }
};

static inline void blake2s_hash(const void* data, size_t len, uint8_t out[32]) {
    Blake2s s; s.update(data, len); s.finalize(out);
}

}  // namespace orc
