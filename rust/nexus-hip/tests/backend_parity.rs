//! `cargo test -p nexus-hip --features …` on a box with a gfx950 GPU: HipBackend against SimdBackend, the way the reference tests its own
//! use of the backend (prover/src/trace/utils.rs:113-128 `test_order`; prover/src/test_utils.rs:32-110 `test_params` / `commit_traces`).
//! Every test runs the same inputs through both backends and compares VALUES (field elements, roots, proof bytes) — never timings.
//!
//! NOT COMPILED in the build image (no Rust toolchain).  The Stwo items used here are the ones the reference itself imports
//! (tests/golden/reference_use_paths.txt) plus the entries of rust/UNOBSERVED_PATHS.txt; tests/test_rust_shim_cpu.py holds this file to
//! that list like the rest of rust/.
#![cfg(stwo_traits)]

use stwo::core::{
    channel::Blake2sChannel,
    fields::{m31::BaseField, qm31::SecureField},
    pcs::PcsConfig,
    poly::circle::CanonicCoset,
    vcs::blake2_merkle::Blake2sMerkleChannel,
};
use stwo::prover::{
    backend::{
        simd::{column::BaseColumn, SimdBackend},
        BackendForChannel, Col, Column, ColumnOps,
    },
    poly::{
        circle::{CircleEvaluation, PolyOps},
        BitReversedOrder,
    },
    CommitmentSchemeProver,
};
use stwo_constraint_framework::{preprocessed_columns::PreProcessedColumnId, EvalAtRow, FrameworkComponent, FrameworkEval, TraceLocationAllocator};

use nexus_hip::record::{record_component, TraceLocations};
use nexus_hip::{proof_bytes, HipBackend, HipColumn, Session};
use nexus_hip_sys as sys;

/// a fixed pseudo-random column (xorshift; no `rand` needed): canonical M31 values, both boundary values included
fn column_values(log_size: u32, seed: u64) -> Vec<BaseField> {
    let mut s = seed | 1;
    let p = (1u32 << 31) - 1;
    (0..1usize << log_size)
        .map(|i| {
            s ^= s << 13;
            s ^= s >> 7;
            s ^= s << 17;
            let v = match i { 0 => 0, 1 => p - 1, _ => (s % p as u64) as u32 };
            BaseField::from_u32_unchecked(v)
        })
        .collect()
}

/// reference prover/src/trace/utils.rs:117-128: the bit-reversal of a column is the backend's, whatever the backend
#[test]
fn bit_reverse_column_matches_simd() {
    for log_size in [4u32, 9, 13, 16] {
        let values = column_values(log_size, 7 + log_size as u64);
        let mut simd = BaseColumn::from_iter(values.clone());
        <SimdBackend as ColumnOps<BaseField>>::bit_reverse_column(&mut simd);
        let mut hip: HipColumn<BaseField> = values.into_iter().collect();
        <HipBackend as ColumnOps<BaseField>>::bit_reverse_column(&mut hip);
        assert_eq!(simd.to_cpu(), hip.to_cpu(), "log_size {log_size}");
    }
}

fn evaluation<B: ColumnOps<BaseField>>(log_size: u32, seed: u64) -> CircleEvaluation<B, BaseField, BitReversedOrder>
where
    Col<B, BaseField>: FromIterator<BaseField>,
{
    let domain = CanonicCoset::new(log_size).circle_domain();
    CircleEvaluation::new(domain, column_values(log_size, seed).into_iter().collect())
}

/// PolyOps: interpolate (K3) and evaluate on the blown-up domain (K4) with the twiddles the reference builds (machine.rs:186-194)
#[test]
fn interpolate_and_extend_match_simd() {
    for log_size in [5u32, 10, 13, 15] {
        let ext = CanonicCoset::new(log_size + 1).circle_domain();
        let tw_s = SimdBackend::precompute_twiddles(ext.half_coset);
        let tw_h = HipBackend::precompute_twiddles(ext.half_coset);
        let poly_s = evaluation::<SimdBackend>(log_size, 3).interpolate_with_twiddles(&tw_s);
        let poly_h = evaluation::<HipBackend>(log_size, 3).interpolate_with_twiddles(&tw_h);
        assert_eq!(poly_s.coeffs.to_cpu(), poly_h.coeffs.to_cpu(), "coefficients, log_size {log_size}");
        let lde_s = poly_s.evaluate_with_twiddles(ext, &tw_s);
        let lde_h = poly_h.evaluate_with_twiddles(ext, &tw_h);
        assert_eq!(lde_s.values.to_cpu(), lde_h.values.to_cpu(), "extension, log_size {log_size}");
    }
}

/// reference prover/src/test_utils.rs:63-110 (`commit_traces`), generic over the backend: three trees of different widths and two column
/// sizes, committed in the reference's order; what comes back is what the verifier sees of them
fn commit_three_trees<B>(log_size: u32) -> (Vec<[u8; 32]>, [u8; 32])
where
    B: BackendForChannel<Blake2sMerkleChannel> + PolyOps,
    Col<B, BaseField>: FromIterator<BaseField>,
{
    let config = PcsConfig::default();
    let twiddles = B::precompute_twiddles(CanonicCoset::new(log_size + config.fri_config.log_blowup_factor + 1).circle_domain().half_coset);
    let mut commitment_scheme = CommitmentSchemeProver::<B, Blake2sMerkleChannel>::new(config, &twiddles);
    let mut channel = Blake2sChannel::default();
    for (tree, widths) in [(0u64, [3usize, 2]), (1, [20, 5]), (2, [8, 4])] {
        let mut tree_builder = commitment_scheme.tree_builder();
        let evals: Vec<_> = (0..widths[0])
            .map(|c| evaluation::<B>(log_size, 100 * tree + c as u64))
            .chain((0..widths[1]).map(|c| evaluation::<B>(log_size - 3, 1000 * tree + c as u64)))
            .collect();
        tree_builder.extend_evals(evals);
        tree_builder.commit(&mut channel);
    }
    (commitment_scheme.roots().iter().map(|r| r.0).collect(), channel.digest().0)
}

#[test]
fn commitments_match_simd() {
    for log_size in [8u32, 13] {
        assert_eq!(commit_three_trees::<SimdBackend>(log_size), commit_three_trees::<HipBackend>(log_size), "log_size {log_size}");
    }
}

/// a component small enough to read: a preprocessed selector, column b the square of column a, column c stepping by one (a neighbour-row
/// read, mask [-1, 0]), constraints of degree 2
#[derive(Clone)]
struct SquaresEval {
    log_size: u32,
}
fn selector_id() -> PreProcessedColumnId {
    PreProcessedColumnId { id: "parity_selector".to_owned() }
}
impl FrameworkEval for SquaresEval {
    fn log_size(&self) -> u32 {
        self.log_size
    }
    fn max_constraint_log_degree_bound(&self) -> u32 {
        self.log_size + 1
    }
    fn evaluate<E: EvalAtRow>(&self, mut eval: E) -> E {
        let sel = eval.get_preprocessed_column(selector_id());
        let a = eval.next_trace_mask();
        let b = eval.next_trace_mask();
        let [c_prev, c] = eval.next_interaction_mask(1, [-1, 0]);
        eval.add_constraint(a.clone() * a.clone() - b.clone());
        eval.add_constraint((c - c_prev - E::F::from(BaseField::from_u32_unchecked(1))) * (a.clone() - b.clone()));
        eval.add_constraint(sel * (a - b));
        eval
    }
}

/// rows in TRACE order: a is 0 / 1 valued, so a^2 == a == b and the factor (a - b) of the other two constraints vanishes on every row
fn squares_trace(log_size: u32) -> [Vec<BaseField>; 3] {
    let n = 1usize << log_size;
    let a: Vec<BaseField> = (0..n).map(|i| BaseField::from_u32_unchecked(((i as u32).wrapping_mul(2654435761) >> 31) & 1)).collect();
    let b = a.clone();
    let c: Vec<BaseField> = (0..n).map(|i| BaseField::from_u32_unchecked(i as u32)).collect();
    [a, b, c]
}

/// `stwo::prover::prove::<SimdBackend, Blake2sMerkleChannel>` (reference machine.rs:286-290) against the session's whole prove on the
/// device for the same component, trace and transcript prefix: the proofs are the same BYTES (postcard, as the SDK ships them: sdk/Cargo.toml:22)
#[test]
fn session_prove_matches_stwo_prove_on_simd() {
    let log_size = 10u32;
    let config = PcsConfig::default();
    let eval = SquaresEval { log_size };
    let trace = squares_trace(log_size);

    // ---- the reference's route (machine.rs:184-290 with one component, one preprocessed column, no interaction tree)
    let twiddles = SimdBackend::precompute_twiddles(CanonicCoset::new(log_size + config.fri_config.log_blowup_factor + 1).circle_domain().half_coset);
    let mut commitment_scheme = CommitmentSchemeProver::<SimdBackend, Blake2sMerkleChannel>::new(config, &twiddles);
    let mut channel = Blake2sChannel::default();
    let finalize = |col: &Vec<BaseField>| {
        let mut c = BaseColumn::from_iter(stwo_order(col));
        <SimdBackend as ColumnOps<BaseField>>::bit_reverse_column(&mut c);
        c
    };
    let selector = finalize(&column_values(log_size, 99));
    let finalized: Vec<BaseColumn> = trace.iter().map(finalize).collect();
    let domain = CanonicCoset::new(log_size).circle_domain();
    let mut tree_builder = commitment_scheme.tree_builder();
    tree_builder.extend_evals([CircleEvaluation::new(domain, selector.clone())]);
    tree_builder.commit(&mut channel);
    let mut tree_builder = commitment_scheme.tree_builder();
    tree_builder.extend_evals(finalized.iter().map(|c| CircleEvaluation::new(domain, c.clone())));
    tree_builder.commit(&mut channel);
    let component = FrameworkComponent::new(&mut TraceLocationAllocator::default(), eval.clone(), SecureField::default());
    let stark_proof = stwo::prover::prove::<SimdBackend, Blake2sMerkleChannel>(&[&component], &mut channel, commitment_scheme).expect("the trace satisfies the constraints");
    let expected = postcard::to_allocvec(&(stark_proof, vec![SecureField::default()], vec![log_size])).unwrap();

    // ---- the device route: the same columns through the session, the component recorded instead of instantiated
    let cfg = sys::nx_pcs_config {
        pow_bits: config.pow_bits,
        log_blowup: config.fri_config.log_blowup_factor,
        n_queries: config.fri_config.n_queries as u32,
        log_last_layer_degree_bound: config.fri_config.log_last_layer_degree_bound,
        fri_alpha_mode: 0,
        hash_mode: 0,
        log_constraint_degree: 1,
    };
    let mut session = Session::new(&cfg, log_size, 0).unwrap();
    let words_of = |c: &BaseColumn| -> Vec<u32> { c.to_cpu().into_iter().map(|x| x.0).collect() };
    let sel_host = words_of(&selector);
    let _ = session.tree_begin(&[log_size]).unwrap();
    session.tree_commit_host(&[sel_host.as_ptr()], false, &[]).unwrap();
    let host: Vec<Vec<u32>> = finalized.iter().map(words_of).collect();
    let ptrs: Vec<*const u32> = host.iter().map(|c| c.as_ptr()).collect();
    let _ = session.tree_begin(&[log_size; 3]).unwrap();
    session.tree_commit_host(&ptrs, false, &[]).unwrap();
    let mut locations = TraceLocations::default();
    let recorded = record_component(&eval, &mut locations, SecureField::default());
    let words = session.prove(&[recorded]).unwrap();
    let got = proof_bytes(&words, &[0, 0, 0, 0], &[log_size]).unwrap();
    assert_eq!(expected, got);
}

/// the reference's `coset_order_to_circle_domain_order` (prover/src/trace/utils_external.rs:24-39), restated for the test: first the even
/// trace rows, then the odd ones reversed
fn stwo_order(values: &[BaseField]) -> Vec<BaseField> {
    let n = values.len();
    let mut out = Vec::with_capacity(n);
    for i in 0..n / 2 {
        out.push(values[i << 1]);
    }
    for i in 0..n / 2 {
        out.push(values[n - 1 - (i << 1)]);
    }
    out
}
