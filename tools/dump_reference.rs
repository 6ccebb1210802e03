//! tools/dump_reference.rs — golden vectors from the REAL reference (nexus-zkvm @ this checkout + stwo @ 0790eba), to pin the
//! in-repo oracle and the GPU library.  Needs a box with the Rust toolchain of `rust-toolchain.toml` (nightly-2025-05-09) and
//! network access for the stwo git dependency; it could not be built or run in the build container (no cargo, no network).
//!
//! Crate layout: the reference's pin is the POST-SPLIT Stwo — one crate `stwo` with `stwo::core::…` (fields, circle, channel, pcs,
//! vcs, verifier) and `stwo::prover::…` (backend, poly, CommitmentSchemeProver, prove), as the reference's own imports show
//! (prover/src/machine.rs:4-19, prover/src/test_utils.rs:1-17, prover2/machine/src/prove.rs:2-15).  Every path below that the
//! reference itself names is checked against it by tests/test_rust_shim_cpu.py; the ones it never names (the traits Stwo calls
//! internally: FriOps, QuotientOps, MerkleOps, GrindOps, …) are listed in rust/UNOBSERVED_PATHS.txt as upstream recollection — a wrong
//! one is a one-line `use` fix on first compile, the arithmetic below does not depend on it.
//!
//! Install:  cp tools/dump_reference.rs <reference>/prover/tests/dump_reference.rs
//!           add to prover/Cargo.toml [dev-dependencies] (it holds only `rand` and `rand_chacha` today, prover/Cargo.toml:27-29):
//!             serde_json = "1", hex = "0.4", postcard = { version = "1.0.10", features = ["alloc", "use-std"] }
//!           (`nexus-vm`, `nexus-common`, `stwo` are regular dependencies of the crate already, prover/Cargo.toml:13-24)
//! Run:      cd <reference>/prover && cargo test --release --test dump_reference -- --nocapture > reference_dump.json
//! Replay:   python tools/replay_reference_dump.py reference_dump.json          (CPU oracle; add --gpu on an MI355X box)
//!
//! What it dumps (one JSON object):
//!   "kat": Stwo-level known answers on seeded inputs (SplitMix64, the generator tests/ and bench.py use), independent of any AIR:
//!     twiddles of a log-5 half coset; interpolate / evaluate (blow-up 2) of one 2^6 column; Blake2s Merkle root + every layer of a
//!     mixed-degree commit (columns of 2^6, 2^6 (x17), 2^4); Blake2sChannel digests after mix_u64 / mix_felts / mix_root and the felts /
//!     bytes it draws; grind nonces; eval_at_point; accumulate_quotients of 3 columns / 2 sample batches; fold_circle_into_line and
//!     fold_line; FriOps::decompose if the trait still has it (Appendix B.5); a LogupTraceGenerator driven in pairs + finalize_last
//!     ("logup_pairs"), and a 200-element relation with constant / sum entries and an expression numerator ("logup_wide": the
//!     reference's widest tuple).  These settle SURVEY.md Appendix B.1-B.6 and pin the logup forms of VERDICT r4 #2.
//!   "prove": for the `stark_prove` bench program (prover-benches/benches/stark_prove.rs:55-82) at log sizes 8, 12, 16:
//!     the 4 commitments, claimed sums, log sizes, proof_of_work, every FRI layer commitment, the last layer polynomial, the
//!     size of sampled / queried values, a hash + (for log 8) the full hex of postcard::to_stdvec(&proof)  (nx_proof_serialize_stwo).
use nexus_vm::emulator::View;
use nexus_vm::riscv::{BasicBlock, BuiltinOpcode, Instruction, Opcode};
use nexus_vm::trace::{k_trace_direct, UniformTrace};
use nexus_vm_prover::prove;
use serde_json::{json, Value};
use stwo::core::channel::{Blake2sChannel, Channel, MerkleChannel};
use stwo::core::circle::CirclePoint;
use stwo::core::fields::m31::BaseField;
use stwo::core::fields::qm31::SecureField;
use stwo::core::pcs::quotients::ColumnSampleBatch;
use stwo::core::poly::circle::CanonicCoset;
use stwo::core::poly::line::LineDomain;
use stwo::core::vcs::blake2_merkle::{Blake2sMerkleChannel, Blake2sMerkleHasher};
use num_traits::One;
use stwo::prover::backend::simd::m31::{PackedBaseField, LOG_N_LANES};
use stwo::prover::backend::simd::qm31::PackedSecureField;
use stwo::prover::backend::simd::SimdBackend;
use stwo::prover::backend::{Col, Column};
use stwo_constraint_framework::LogupTraceGenerator;
stwo_constraint_framework::relation!(WideLookupElements, 200);
use stwo::prover::fri::FriOps;
use stwo::prover::line::LineEvaluation;
use stwo::prover::pcs::quotient_ops::QuotientOps;
use stwo::prover::poly::circle::{CircleEvaluation, PolyOps, SecureEvaluation};
use stwo::prover::poly::BitReversedOrder;
use stwo::prover::proof_of_work::GrindOps;
use stwo::prover::vcs::ops::MerkleOps;
use stwo::prover::vcs::prover::MerkleProver;

fn splitmix64(mut x: u64) -> u64 {
    x = x.wrapping_add(0x9E3779B97F4A7C15);
    x = (x ^ (x >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
    x = (x ^ (x >> 27)).wrapping_mul(0x94D049BB133111EB);
    x ^ (x >> 31)
}
/// column `c`, row `r` of the seeded test data: (splitmix64(seed ^ c << 32 ^ r) >> 33) with P mapped to 0
fn val(seed: u64, c: u64, r: u64) -> BaseField {
    let v = (splitmix64(seed ^ (c << 32) ^ r) >> 33) as u32;
    BaseField::from_u32_unchecked(if v == (1 << 31) - 1 { 0 } else { v })
}
fn m31s(v: impl IntoIterator<Item = BaseField>) -> Value { json!(v.into_iter().map(|x| x.0).collect::<Vec<u32>>()) }
fn qm31(q: SecureField) -> Value { let a = q.to_m31_array(); json!([a[0].0, a[1].0, a[2].0, a[3].0]) }
fn hexs(b: impl AsRef<[u8]>) -> Value { json!(hex::encode(b)) }
fn col(seed: u64, c: u64, log: u32) -> Col<SimdBackend, BaseField> { (0..1u64 << log).map(|r| val(seed, c, r)).collect() }

fn kat() -> Value {
    let mut out = serde_json::Map::new();
    // K2: twiddle tree of CanonicCoset(6).half_coset (log 5)
    let tw = SimdBackend::precompute_twiddles(CanonicCoset::new(6).half_coset());
    out.insert("twiddles_log5".into(), json!({"twiddles": m31s(tw.twiddles.to_cpu()), "itwiddles": m31s(tw.itwiddles.to_cpu())}));
    // K3 / K4: one column of 2^6 seeded values (bit-reversed circle-domain order as given), interpolate, evaluate on the 2^7 domain
    let tw7 = SimdBackend::precompute_twiddles(CanonicCoset::new(7).half_coset());
    let dom6 = CanonicCoset::new(6).circle_domain();
    let eval = CircleEvaluation::<SimdBackend, BaseField, BitReversedOrder>::new(dom6, col(0xC0FFEE, 0, 6));
    let poly = eval.clone().interpolate_with_twiddles(&tw7);
    let lde = poly.evaluate_with_twiddles(CanonicCoset::new(7).circle_domain(), &tw7);
    out.insert("lde_log6".into(), json!({"seed": "0xC0FFEE", "coeffs": m31s(poly.coeffs.to_cpu()), "lde": m31s(lde.values.to_cpu())}));
    // K7
    let mut ch = Blake2sChannel::default();
    ch.mix_u64(7);
    let p = CirclePoint::<SecureField>::get_random_point(&mut ch);
    out.insert("eval_at_point".into(), json!({"point": [qm31(p.x), qm31(p.y)], "value": qm31(poly.eval_at_point(p))}));
    // K5: mixed-degree Merkle commit — 18 columns of 2^6 (one full 16-column block + 2), 1 column of 2^4
    let cols: Vec<Col<SimdBackend, BaseField>> = (0..18).map(|c| col(1, c, 6)).chain(std::iter::once(col(1, 99, 4))).collect();
    let merkle = MerkleProver::<SimdBackend, Blake2sMerkleHasher>::commit(cols.iter().collect());
    out.insert("merkle".into(), json!({"root": hexs(merkle.root().0), "layers": merkle.layers.iter().map(|l| l.to_cpu().iter().map(|h| hex::encode(h.0)).collect::<Vec<_>>()).collect::<Vec<_>>()}));
    // K6: channel
    let mut ch = Blake2sChannel::default();
    let mut steps = vec![];
    ch.mix_u64(0x0123456789ABCDEF); steps.push(json!({"after": "mix_u64(0x0123456789ABCDEF)", "digest": hexs(ch.digest().0)}));
    // (`draw_felt` / `draw_felts` / `draw_random_bytes` before the rename; the dump keys keep the old names the replay tool reads)
    let f = ch.draw_secure_felt(); steps.push(json!({"draw_felt": qm31(f), "digest": hexs(ch.digest().0)}));
    let fs = ch.draw_secure_felts(3); steps.push(json!({"draw_felts(3)": fs.iter().map(|x| qm31(*x)).collect::<Vec<_>>()}));
    ch.mix_felts(&[f, fs[0]]); steps.push(json!({"after": "mix_felts([first drawn felt, first of draw_felts(3)])", "digest": hexs(ch.digest().0)}));
    Blake2sMerkleChannel::mix_root(&mut ch, merkle.root()); steps.push(json!({"after": "mix_root(merkle root above)", "digest": hexs(ch.digest().0)}));
    steps.push(json!({"draw_random_bytes": hexs(ch.draw_u32s().iter().flat_map(|w| w.to_le_bytes()).collect::<Vec<u8>>())}));
    for bits in [0u32, 5, 10, 16] { let mut c2 = ch.clone(); steps.push(json!({"grind_bits": bits, "nonce": SimdBackend::grind(&c2, bits)})); let _ = &mut c2; }
    out.insert("channel".into(), json!(steps));
    // R8: LogupTraceGenerator driven in PAIRS, the way prover2's LogupTraceBuilder::add_to_relation_with drives it (reference
    // prover2/machine/src/lookups/logup_trace_builder.rs:86-101) and finalize_logup_in_pairs constrains it: column 0 = the fractions
    // 1 / (t0 - z) and -t3 / (t1 + alpha t2 - z) merged as (a d + b c) / (b d), column 1 = the left-over 1 / (t2 + alpha t0 + alpha^2 t1 - z),
    // finalize_last (claimed sum; prefix sum in natural coset order).  Pins VERDICT r4 #2's forms and nx_logup_cols_batched / nx_logup_program.
    {
        let log = 6u32;
        let (z, alpha) = (ch.draw_secure_felt(), ch.draw_secure_felt());
        let t: Vec<Col<SimdBackend, BaseField>> = (0..4).map(|c| col(3, c, log)).collect();
        let (pz, pa, pa2) = (PackedSecureField::broadcast(z), PackedSecureField::broadcast(alpha), PackedSecureField::broadcast(alpha * alpha));
        let n_vec = 1usize << (log - LOG_N_LANES);
        let mut gen = LogupTraceGenerator::new(log);
        let mut c0 = gen.new_col();
        for vr in 0..n_vec {
            let (a, b): (PackedSecureField, PackedSecureField) = (PackedSecureField::one(), PackedSecureField::from(t[0].data[vr]) - pz);
            let (c, d): (PackedSecureField, PackedSecureField) = (-PackedSecureField::from(t[3].data[vr]), PackedSecureField::from(t[1].data[vr]) + pa * t[2].data[vr] - pz);
            c0.write_frac(vr, a * d + b * c, b * d);
        }
        c0.finalize_col();
        let mut c1 = gen.new_col();
        for vr in 0..n_vec {
            c1.write_frac(vr, PackedSecureField::one(), PackedSecureField::from(t[2].data[vr]) + pa * t[0].data[vr] + pa2 * t[1].data[vr] - pz);
        }
        c1.finalize_col();
        let (trace, claimed) = gen.finalize_last();
        out.insert("logup_pairs".into(), json!({"seed": 3, "z": qm31(z), "alpha": qm31(alpha), "columns": trace.iter().map(|e| m31s(e.values.to_cpu())).collect::<Vec<_>>(),
                                                 "claimed_sum": qm31(claimed)}));
    }
    // R8 at the reference's tuple WIDTH: a 200-element relation — the keccak chips' state lookup (reference
    // prover/src/chips/custom.rs:45-46 `relation!(RawStateLookupElements, 25 * 8)`, extensions/keccak/round/constraints.rs:101-110) — combined
    // by Stwo's own LookupElements::combine, with a CONSTANT entry and a SUM-of-two-columns entry in the tuple (bit_op.rs:355; bitwise_table/
    // constraints.rs:50-71) and the expression numerator (is_padding - 1), one fraction per column, finalize_last.  Pins nx_logup_program /
    // nx_logup_cols at 200 tuple columns and the alpha powers alpha^0 .. alpha^199 ("logup_wide").
    {
        let log = 6u32;
        let wide = WideLookupElements::draw(&mut ch);
        let t: Vec<Col<SimdBackend, BaseField>> = (0..5).map(|c| col(4, c, log)).collect();
        let n_vec = 1usize << (log - LOG_N_LANES);
        let mut gen = LogupTraceGenerator::new(log);
        let mut c0 = gen.new_col();
        for vr in 0..n_vec {
            let tuple: Vec<PackedBaseField> = (0..200usize)
                .map(|k| match k {
                    1 => PackedBaseField::broadcast(BaseField::from(5u32)),
                    2 => t[1].data[vr] + t[2].data[vr],
                    _ => t[k % 5].data[vr],
                })
                .collect();
            let denom: PackedSecureField = wide.combine(&tuple);
            c0.write_frac(vr, PackedSecureField::from(t[3].data[vr]) - PackedSecureField::one(), denom);
        }
        c0.finalize_col();
        let (trace, claimed) = gen.finalize_last();
        out.insert("logup_wide".into(), json!({"seed": 4, "z": qm31(wide.z), "alpha": qm31(wide.alpha), "columns": trace.iter().map(|e| m31s(e.values.to_cpu())).collect::<Vec<_>>(),
                                                "claimed_sum": qm31(claimed)}));
    }
    // K8: DEEP quotients of 3 LDE columns, two sample batches (points p and p + step; values = the true evaluations, so the result is low degree)
    let polys: Vec<_> = (0..3).map(|c| CircleEvaluation::<SimdBackend, BaseField, BitReversedOrder>::new(dom6, col(2, c, 6)).interpolate_with_twiddles(&tw7)).collect();
    let ldes: Vec<_> = polys.iter().map(|p| p.evaluate_with_twiddles(CanonicCoset::new(7).circle_domain(), &tw7)).collect();
    let step = CanonicCoset::new(6).step().into_ef();
    let batches = vec![
        ColumnSampleBatch { point: p, columns_and_values: (0..3).map(|c| (c, polys[c].eval_at_point(p))).collect() },
        ColumnSampleBatch { point: p + step, columns_and_values: vec![(0, polys[0].eval_at_point(p + step)), (1, polys[1].eval_at_point(p + step))] },
    ];
    let alpha = ch.draw_secure_felt();
    let q = SimdBackend::accumulate_quotients(CanonicCoset::new(7).circle_domain(), &ldes.iter().collect::<Vec<_>>(), alpha, &batches, 1);
    out.insert("quotients".into(), json!({"seed": 2, "alpha": qm31(alpha), "out": (0..4).map(|k| m31s(q.values.columns[k].to_cpu())).collect::<Vec<_>>()}));
    // K9: folds of the quotient column
    let a2 = ch.draw_secure_felt();
    let mut line = LineEvaluation::<SimdBackend>::new_zero(LineDomain::new(CanonicCoset::new(7).half_coset()));
    SimdBackend::fold_circle_into_line(&mut line, &q, a2, &tw7);
    let folded = SimdBackend::fold_line(&line, a2, &tw7);
    out.insert("folds".into(), json!({"alpha": qm31(a2), "circle_into_line": (0..4).map(|k| m31s(line.values.columns[k].to_cpu())).collect::<Vec<_>>(),
                                       "fold_line": (0..4).map(|k| m31s(folded.values.columns[k].to_cpu())).collect::<Vec<_>>()}));
    // Appendix B.5 — delete this block if FriOps::decompose is gone at the pinned revision (that answers the question)
    let (g, lambda) = SimdBackend::decompose(&SecureEvaluation::<SimdBackend, BitReversedOrder>::new(CanonicCoset::new(7).circle_domain(), q.values.clone()));
    out.insert("decompose".into(), json!({"lambda": qm31(lambda), "g0": m31s(g.values.columns[0].to_cpu())}));
    Value::Object(out)
}

/// The bench's program and trace, built the way prover-benches/benches/stark_prove.rs:55-82 builds them: one ADDI, then ADDs cycling
/// through the registers, 2^log_size instructions, one basic block, `k_trace_direct(&blocks, 1)`.  (`nexus_common_testing::
/// program_trace` returns only the `Vec<BasicBlock>` — common-testing/src/lib.rs:5 — and is not a dependency of the prover crate.)
fn program_trace(log_size: u32) -> (View, UniformTrace) {
    const NUM_REGISTERS: u8 = nexus_common::constants::NUM_REGISTERS as u8;
    let (mut i, mut j, mut k) = (0u8, 1u8, 2u8);
    let first = Instruction::new_ir(Opcode::from(BuiltinOpcode::ADDI), 1, 0, 1);
    let rest = std::iter::from_fn(|| {
        let inst = Instruction::new_ir(Opcode::from(BuiltinOpcode::ADD), k, j, i.into());
        i = (i + 1) % NUM_REGISTERS;
        j = (j + 1) % NUM_REGISTERS;
        k = (k + 1) % NUM_REGISTERS;
        Some(inst)
    });
    let insts: Vec<Instruction> = std::iter::once(first).chain(rest).take(1 << log_size).collect();
    k_trace_direct(&vec![BasicBlock::new(insts)], 1).expect("error generating trace")
}

fn prove_dump(log_size: u32) -> Value {
    let (view, trace) = program_trace(log_size);
    let proof = prove(&trace, &view).expect("prove");
    let bytes = postcard::to_stdvec(&proof).expect("postcard");
    let sp = &proof.stark_proof.0;
    let sha = { use std::hash::Hasher; let mut h = std::collections::hash_map::DefaultHasher::new(); h.write(&bytes); h.finish() };
    json!({
        "log_size": log_size,
        "config": {"pow_bits": sp.config.pow_bits, "log_blowup_factor": sp.config.fri_config.log_blowup_factor,
                   "log_last_layer_degree_bound": sp.config.fri_config.log_last_layer_degree_bound, "n_queries": sp.config.fri_config.n_queries},
        "commitments": sp.commitments.iter().map(|h| hex::encode(h.0)).collect::<Vec<_>>(),
        "claimed_sum": proof.claimed_sum.iter().map(|x| qm31(*x)).collect::<Vec<_>>(),
        "component_log_sizes": proof.log_size,
        "n_sampled_columns": sp.sampled_values.iter().map(|t| t.len()).collect::<Vec<_>>(),
        "n_queried_values": sp.queried_values.iter().map(|t| t.len()).collect::<Vec<_>>(),
        "proof_of_work": sp.proof_of_work,
        "fri_first_layer": hex::encode(sp.fri_proof.first_layer.commitment.0),
        "fri_inner_layers": sp.fri_proof.inner_layers.iter().map(|l| hex::encode(l.commitment.0)).collect::<Vec<_>>(),
        "last_layer_poly": serde_json::to_value(&sp.fri_proof.last_layer_poly).unwrap(),
        "postcard_len": bytes.len(), "postcard_hash64": sha,
        "postcard_hex": if log_size <= 8 { hex::encode(&bytes) } else { String::new() },
    })
}

#[test]
fn dump_reference() {
    let out = json!({"reference": "nexus-zkvm v0.3.6 + stwo 0790eba", "kat": kat(), "prove": [prove_dump(8), prove_dump(12), prove_dump(16)]});
    println!("{}", serde_json::to_string(&out).unwrap());
}
