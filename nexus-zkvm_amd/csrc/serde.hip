// The reference's proof bytes: `nexus_vm_prover::machine::Proof { stark_proof, claimed_sum, log_size }`
// (reference prover/src/machine.rs:93-98) in the serde format the SDK ships proofs in — postcard (reference sdk/Cargo.toml:22,
// sdk/src/stwo/seq.rs:60-64) — produced from the library's NXP1 word stream.  Host arithmetic only: no context, no GPU.
//
// Field order follows Stwo's derive(Serialize) declarations [upstream-recollection, stwo @ 0790eba — to be pinned by
// tools/dump_reference.rs on a box with cargo]:
//   StarkProof(CommitmentSchemeProof { config: PcsConfig { pow_bits, fri_config: FriConfig { log_blowup_factor,
//     log_last_layer_degree_bound, n_queries } }, commitments, sampled_values, decommitments, queried_values, proof_of_work,
//     fri_proof: FriProof { first_layer, inner_layers, last_layer_poly: LinePoly { coeffs (bit-reversed), log_size } } })
//   FriLayerProof { fri_witness, decommitment: MerkleDecommitment { hash_witness, column_witness }, commitment }
// postcard: u32 / u64 / usize as LEB128 varints, Vec<T> as varint length + items, [u8; 32] as 32 raw bytes, M31 as its u32,
// QM31 as its 4 M31 coordinates.
#include "internal.h"
#include <string.h>
#include <stdlib.h>

namespace {

struct Writer {
    std::vector<uint8_t> b;
    void varint(uint64_t v) { while (v >= 0x80) { b.push_back((uint8_t)(v | 0x80)); v >>= 7; } b.push_back((uint8_t)v); }
    void hash(const uint32_t* w) { const uint8_t* p = (const uint8_t*)w; b.insert(b.end(), p, p + 32); }
    void qm31(const uint32_t* w) { for (int k = 0; k < 4; k++) varint(w[k]); }
};
struct Reader {
    const uint32_t* p; size_t n, i = 0; bool ok = true;
    uint32_t u() { if (i >= n) { ok = false; return 0; } return p[i++]; }
    const uint32_t* take(size_t k) { if (k > n - i) { ok = false; return nullptr; } const uint32_t* r = p + i; i += k; return r; }
    size_t count(size_t unit) { uint32_t c = u(); if (unit && (size_t)c > (n - i) / unit) { ok = false; return 0; } return c; }
};

bool decommitment(Reader& r, Writer& w) {
    size_t nh = r.count(8); w.varint(nh);
    for (size_t k = 0; k < nh && r.ok; k++) { const uint32_t* h = r.take(8); if (h) w.hash(h); }
    size_t nc = r.count(1); w.varint(nc);
    for (size_t k = 0; k < nc && r.ok; k++) w.varint(r.u());
    return r.ok;
}
bool fri_layer(Reader& r, Writer& w) {
    size_t nw = r.count(4); w.varint(nw);
    for (size_t k = 0; k < nw && r.ok; k++) { const uint32_t* q = r.take(4); if (q) w.qm31(q); }
    if (!decommitment(r, w)) return false;
    const uint32_t* h = r.take(8); if (h) w.hash(h);
    return r.ok;
}

}  // namespace

extern "C" int nx_proof_serialize_stwo(const uint32_t* proof_words, size_t n_words, const uint32_t* claimed_sums, const uint32_t* log_sizes,
                                       uint32_t n_components, uint8_t** bytes, size_t* n_bytes) {
    using nx::set_err;
    if (!proof_words || !bytes || !n_bytes || (n_components && (!claimed_sums || !log_sizes))) return set_err(nullptr, NX_ERR_ARG, "nx_proof_serialize_stwo: NULL argument");
    Reader r{proof_words, n_words};
    Writer w;
    if (r.u() != 0x3150584Eu) return set_err(nullptr, NX_ERR_ARG, "nx_proof_serialize_stwo: not an NXP1 proof");
    const uint32_t pow_bits = r.u(), log_blowup = r.u(), n_queries = r.u(), log_last = r.u();
    w.varint(pow_bits); w.varint(log_blowup); w.varint(log_last); w.varint(n_queries);          // PcsConfig { pow_bits, FriConfig { blowup, last layer bound, n_queries } }
    const size_t nt = r.count(8);
    w.varint(nt);
    for (size_t t = 0; t < nt && r.ok; t++) { const uint32_t* h = r.take(8); if (h) w.hash(h); }  // commitments
    w.varint(nt);
    for (size_t t = 0; t < nt && r.ok; t++) {                                                     // sampled_values
        size_t ncol = r.count(1); w.varint(ncol);
        for (size_t c = 0; c < ncol && r.ok; c++) { size_t ns = r.count(4); w.varint(ns); for (size_t s = 0; s < ns && r.ok; s++) { const uint32_t* q = r.take(4); if (q) w.qm31(q); } }
    }
    w.varint(nt);
    for (size_t t = 0; t < nt && r.ok; t++) if (!decommitment(r, w)) break;                        // decommitments
    w.varint(nt);
    for (size_t t = 0; t < nt && r.ok; t++) { size_t nv = r.count(1); w.varint(nv); for (size_t k = 0; k < nv && r.ok; k++) w.varint(r.u()); }   // queried_values
    { const uint64_t lo = r.u(), hi = r.u(); w.varint(lo | (hi << 32)); }                          // proof_of_work
    if (r.ok) fri_layer(r, w);                                                                     // fri_proof.first_layer
    { size_t nl = r.count(1); w.varint(nl); for (size_t l = 0; l < nl && r.ok; l++) if (!fri_layer(r, w)) break; }
    {   // last_layer_poly: LinePoly { coeffs in bit-reversed order, log_size }; NXP1 holds the ordered coefficients
        size_t nc = r.count(4);
        const uint32_t* c = r.take(4 * nc);
        int lg = 0; while (((size_t)1 << lg) < nc) lg++;
        if (r.ok && nc != ((size_t)1 << lg)) return set_err(nullptr, NX_ERR_ARG, "nx_proof_serialize_stwo: last layer polynomial length is not a power of two");
        w.varint(nc);
        for (size_t k = 0; k < nc && r.ok; k++) w.qm31(c + 4 * nx::bitrev((uint32_t)k, lg));
        w.varint((uint64_t)lg);
    }
    if (!r.ok || r.i != n_words) return set_err(nullptr, NX_ERR_ARG, "nx_proof_serialize_stwo: malformed NXP1 proof");
    w.varint(n_components);                                                                        // Proof.claimed_sum
    for (uint32_t k = 0; k < n_components; k++) w.qm31(claimed_sums + 4 * (size_t)k);
    w.varint(n_components);                                                                        // Proof.log_size
    for (uint32_t k = 0; k < n_components; k++) w.varint(log_sizes[k]);
    uint8_t* out = (uint8_t*)malloc(w.b.size() ? w.b.size() : 1);
    if (!out) return set_err(nullptr, NX_ERR_OOM, "nx_proof_serialize_stwo: malloc failed");
    memcpy(out, w.b.data(), w.b.size());
    *bytes = out; *n_bytes = w.b.size();
    return NX_OK;
}
