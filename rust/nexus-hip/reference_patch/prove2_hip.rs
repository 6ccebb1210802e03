//! `prover2/machine/src/prove_hip.rs` — drop-in for the reference's v2 prover, a sibling of `prove` in `prover2/machine/src/lib.rs` (see
//! README.md next to this file, "prover2", for the edits that wire it in).  `prove_hip` is `prove` (prover2/machine/src/prove.rs:34-135) with
//! the Stwo objects replaced by the device session of `nexus-hip`:
//!
//!   prove.rs                                                          here
//!   :35-42    generate_component_trace per component (CPU, unchanged)  the same statements
//!   :44-58    max_constraint_log_degree_bound, precompute_twiddles     inside `Session::new` (sized for the largest per-component bound)
//!   :59-68    Blake2sChannel, mix_u64 per AD byte / per log size       `Session::mix_u64`
//!   :70-84    tree_builder.extend_evals(to_circle_evaluation(..)) x2   `commit_tree_keeping_evaluations` (upload under the transforms;
//!             + commit(channel)                                        ~55 components of differing log sizes = one mixed-degree tree each)
//!   :86-89    c.draw_lookup_elements(&mut lookup_elements, channel)    a host `Blake2sChannel` set to the session's digest (`host_channel_at`)
//!   :91-105   c.generate_interaction_trace(..), mix_felts, commit      ON THE DEVICE: every component's relation entries, recorded by the same
//!                                                                      evaluator that records its constraints, run as a fraction program
//!                                                                      (`interaction_tree_on_device`); `LogupTraceBuilder` (lookups/
//!                                                                      logup_trace_builder.rs:22-121) is not called, tree 2 never crosses PCIe
//!   :107-121  c.to_component_prover(..)                                `c.to_recorded_component(..)` (README.md edit 6): the same
//!                                                                      `BuiltInComponentEval` (framework/eval.rs:7-33), recorded instead of boxed
//!   :123-127  stwo::prover::prove(..)                                  `Session::prove` (composition, OODS, DEEP quotients, FRI, PoW, decommit)
//!   :129-133  Proof { stark_proof, claimed_sums, log_sizes }           `proof_bytes` (postcard) -> `Proof` (same field order as v1's: prove.rs:27-32)
//!
//! NOT COMPILED in the build image (no Rust toolchain).  The `use` block below is prove.rs:1-25's, minus what the session replaces — the
//! structural test holds every `crate::` / `super::` path here to the ones prove.rs itself imports.  Every v2 component finalises its logup
//! columns in pairs (`finalize_logup_in_pairs`, e.g. components/execution/add/mod.rs) — the recorder infers the batching from the
//! interaction columns the eval asks for (rust/nexus-hip/src/record.rs), so nothing here names it.
use num_traits::Zero;
use stwo::{
    core::{fields::qm31::SecureField, pcs::PcsConfig},
    prover::ProvingError,
};

use nexus_vm::{emulator::View, trace::Trace};
use nexus_vm_prover_trace::{
    component::ComponentTrace,
    eval::{ORIGINAL_TRACE_IDX, PREPROCESSED_TRACE_IDX},
};

use super::BASE_COMPONENTS;
use crate::prove::Proof;
use crate::{lookups::AllLookupElements, side_note::SideNote};

use nexus_hip::record::TraceLocations;
// the steps both reference patches share (this one and machine_hip.rs): rust/nexus-hip/src/simd_host.rs
use nexus_hip::simd_host::{columns_read_by_fractions, commit_tree_keeping_evaluations, host_channel_at, interaction_tree_on_device, pcs_config, secure_from_words, SimdEval};
use nexus_hip::{proof_bytes, HipError, RecordedComponent, Session};

fn to_proving_error(e: HipError) -> ProvingError {
    match e {
        HipError::ConstraintsNotSatisfied => ProvingError::ConstraintsNotSatisfied,
        // the reference has no error channel for resources either: `vec![..]` aborts (prover2/trace/src/builder.rs:103)
        other => panic!("nexus-hip: {other:?}"),
    }
}

pub fn prove_hip(trace: &impl Trace, view: &View) -> Result<Proof, ProvingError> {
    // ---- prove.rs:35-42: component traces on the CPU, the reference's own statements
    let mut prover_side_note = SideNote::new(trace, view);
    let components = BASE_COMPONENTS;

    let traces: Vec<ComponentTrace> = components
        .iter()
        .map(|c| c.generate_component_trace(&mut prover_side_note))
        .collect();
    let log_sizes: Vec<u32> = traces.iter().map(ComponentTrace::log_size).collect();

    // ---- prove.rs:44-68: the largest constraint domain sizes the twiddles (inside the session); channel seeding.  The session takes
    // (largest log size, largest bound over it): a tower at least as tall as CanonicCoset::new(max bound + blowup)'s half coset
    let max_log = log_sizes.iter().copied().max().unwrap_or(0);
    let log_constraint_degree = components
        .iter()
        .zip(&log_sizes)
        .map(|(c, &log_size)| c.max_constraint_log_degree_bound(log_size) - log_size)
        .max()
        .unwrap_or(1);
    let config = PcsConfig::default();
    let cfg = pcs_config(&config, log_constraint_degree);
    let device: i32 = std::env::var("NEXUS_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
    let mut session = Session::new(&cfg, max_log, device).map_err(to_proving_error)?;
    for byte in view.view_associated_data().unwrap_or_default() {
        session.mix_u64(byte.into());
    }
    log_sizes.iter().for_each(|log_size| session.mix_u64(*log_size as u64));

    // ---- prove.rs:70-84: preprocessed tree, main tree — every component's columns, component after component
    let tree0: Vec<SimdEval> = traces.iter().flat_map(|t| t.to_circle_evaluation(PREPROCESSED_TRACE_IDX)).collect();
    let tree1: Vec<SimdEval> = traces.iter().flat_map(|t| t.to_circle_evaluation(ORIGINAL_TRACE_IDX)).collect();
    // Shape pass: the columns the relation entries read are a property of the AIR, not of the lookup elements — the components recorded
    // once with dummy elements (lookups/mod.rs `AllLookupElements::dummy`) say which evaluations must outlive their commit.
    let reads = {
        let dummy = AllLookupElements::dummy();
        let mut shape_pass = TraceLocations::default();
        let shape: Vec<RecordedComponent> = components
            .iter()
            .zip(&log_sizes)
            .map(|(c, log_size)| c.to_recorded_component(&mut shape_pass, &dummy, *log_size, SecureField::zero()))
            .collect();
        columns_read_by_fractions(&shape, [tree0.len(), tree1.len()])
    };
    let kept0 = commit_tree_keeping_evaluations(&mut session, &tree0, Some(&reads[0])).map_err(to_proving_error)?;
    drop(tree0);
    let kept1 = commit_tree_keeping_evaluations(&mut session, &tree1, Some(&reads[1])).map_err(to_proving_error)?;
    drop(tree1);
    drop(traces);                                            // the reference moves them into generate_interaction_trace (prove.rs:96-99)

    // ---- prove.rs:86-89: lookup elements, every component drawing from the one channel in component order
    let mut lookup_elements = AllLookupElements::default();
    let mut host_channel = host_channel_at(&session);
    components
        .iter()
        .for_each(|c| c.draw_lookup_elements(&mut lookup_elements, &mut host_channel));

    // ---- prove.rs:91-105: the interaction trace from the components' own relation entries on the device.  The components are recorded a
    // first time with zero claimed sums — their fraction programs are what generates the trace; the constraints are recorded again below,
    // once the sums are known (the reference, too, builds its component provers after the interaction trace: prove.rs:107-119).
    let mut first_pass = TraceLocations::default();
    let generators: Vec<RecordedComponent> = components
        .iter()
        .zip(&log_sizes)
        .map(|(c, log_size)| c.to_recorded_component(&mut first_pass, &lookup_elements, *log_size, SecureField::zero()))
        .collect();
    let claimed = interaction_tree_on_device(&mut session, &generators, [&kept0, &kept1]).map_err(to_proving_error)?;
    drop(generators);
    session.free_columns();                                      // the kept evaluations: nothing reads them after the interaction tree
    let claimed_sums: Vec<SecureField> = claimed.iter().map(secure_from_words).collect();
    let claimed_words: Vec<u32> = claimed.iter().flatten().copied().collect();
    session.mix_felts(&claimed_words);
    session.tree_commit().map_err(to_proving_error)?;

    // ---- prove.rs:107-121: the components, recorded instead of boxed (same evals, same order, same claimed sums)
    let mut locations = TraceLocations::default();
    let recorded: Vec<RecordedComponent> = components
        .iter()
        .zip(&log_sizes)
        .zip(&claimed_sums)
        .map(|((c, log_size), claimed_sum)| c.to_recorded_component(&mut locations, &lookup_elements, *log_size, *claimed_sum))
        .collect();

    // ---- prove.rs:123-133: stwo::prover::prove on the device; the reference's Proof from its postcard bytes
    let words = session.prove(&recorded).map_err(to_proving_error)?;
    let bytes = proof_bytes(&words, &claimed_words, &log_sizes).map_err(to_proving_error)?;
    let proof: Proof = postcard::from_bytes(&bytes).expect("nx_proof_serialize_stwo emits the serde layout of prove::Proof");
    debug_assert!(proof.claimed_sums == claimed_sums && proof.log_sizes == log_sizes);
    Ok(proof)
}

#[cfg(test)]
mod tests {
    use super::*;
    use crate::{prove, verify};
    use nexus_vm::{
        riscv::{BasicBlock, BuiltinOpcode, Instruction, Opcode},
        trace::k_trace_direct,
    };

    /// prove.rs:146-161's program; the device proof verifies with the reference's own verifier and is the SimdBackend proof byte for byte
    #[test]
    fn prove_hip_verify_and_equals_simd() {
        let basic_block = vec![BasicBlock::new(vec![
            Instruction::new_ir(Opcode::from(BuiltinOpcode::ADDI), 1, 0, 1),
            Instruction::new_ir(Opcode::from(BuiltinOpcode::ADD), 2, 1, 0),
            Instruction::new_ir(Opcode::from(BuiltinOpcode::ADD), 3, 2, 1),
        ])];
        let (view, program_trace) = k_trace_direct(&basic_block, 1).expect("error generating trace");
        let device = prove_hip(&program_trace, &view).unwrap();
        let host = prove(&program_trace, &view).unwrap();
        assert_eq!(postcard::to_allocvec(&device).unwrap(), postcard::to_allocvec(&host).unwrap());
        verify(device, &view).unwrap();
    }
}
