//! `prover/src/machine_hip.rs` — drop-in for the reference tree, a child module of `machine` (see README.md next to this file for
//! the small edits that wire it in).  `Machine::<C>::prove_hip_with_extensions` is `Machine::<C>::prove_with_extensions` (prover/src/machine.rs:130-297)
//! with the Stwo objects replaced by the device session of `nexus-hip`:
//!
//!   machine.rs                                                        here
//!   :135-183  trace generation (CPU, unchanged)                        the same statements
//!   :184-194  SimdBackend::precompute_twiddles(..)                     inside `Session::new` (sized max_log + LOG_CONSTRAINT_DEGREE + blowup - 1)
//!   :197-206  Blake2sChannel, mix_u64 per AD byte / per log size       `Session::mix_u64`
//!   :208-237  tree_builder.extend_evals(..) / .commit(channel) x2      `Session::tree_begin` + `tree_commit_host` (upload under the transforms)
//!   :239-240  C::draw_lookup_elements(&mut lookup_elements, channel)   a host `Blake2sChannel` set to the session's digest (`host_channel_at`)
//!   :242-263  generate_interaction_trace, mix_felts, commit            ON THE DEVICE: every component's relation entries, recorded by the same
//!                                                                      evaluator that records its constraints, run as a fraction program
//!                                                                      (`Session::logup_trace` = nx_logup_program + finalize_last) over the
//!                                                                      kept evaluations of the preprocessed / main columns, straight into the
//!                                                                      interaction tree's columns; `mix_felts`; `tree_commit` — no chip's
//!                                                                      `fill_interaction_trace` is called, nothing of tree 2 crosses PCIe
//!   :264-285  FrameworkComponent::new(..) / to_component_prover(..)    `record_component(..)` over the same `MachineEval` / extension evals
//!   :286-290  stwo::prover::prove(..)                                  `Session::prove` (composition, OODS, DEEP quotients, FRI, PoW, decommit)
//!   :292-296  Proof { stark_proof, claimed_sum, log_size }             `proof_bytes` (postcard) -> `Proof`
//!
//! NOT COMPILED in the build image (no Rust toolchain); the `use` block below is machine.rs:1-47's (its `super::` is this file's `crate::`) — the
//! structural test holds every `crate::` path here to the ones machine.rs itself imports.  The uploads are the preprocessed and the main tree only
//! (with the CPU generator the ~1000 interaction columns of a 2^22-row proof were 16.8 GB over PCIe); the reference's own
//! `generate_interaction_trace` (traits.rs:124-145) stays the cross-check of a debug build (`NEXUS_HIP_CHECK_LOGUP=1`).
use num_traits::Zero;
use stwo::{
    core::{fields::qm31::SecureField, pcs::PcsConfig},
    prover::ProvingError,
};

// a CHILD module of `machine` (`#[path = "machine_hip.rs"] mod hip;` inside machine.rs): `BASE_EXTENSIONS` and `Machine::max_log_size`
// are private to that module (machine.rs:82, :488)
use super::{GeneratedTraces, Machine, Proof, BASE_EXTENSIONS};
use crate::trace::eval::{INTERACTION_TRACE_IDX, ORIGINAL_TRACE_IDX, PREPROCESSED_TRACE_IDX};
use nexus_vm::{emulator::View, trace::Trace};

use crate::components::{MachineEval, LOG_CONSTRAINT_DEGREE};
use crate::traits::MachineChip;
use crate::{
    components::AllLookupElements,
    extensions::{ComponentTrace, ExtensionComponent, ExtensionsConfig},
    traits::generate_interaction_trace,
};

use nexus_hip::record::{record_component, TraceLocations};
// the steps both reference patches share (this one and prove2_hip.rs): rust/nexus-hip/src/simd_host.rs
use nexus_hip::simd_host::{columns_read_by_fractions, commit_tree_keeping_evaluations, host_channel_at, interaction_tree_on_device, pcs_config, secure_from_words, SimdEval};
use nexus_hip::{proof_bytes, HipError, RecordedComponent, Session};

fn to_proving_error(e: HipError) -> ProvingError {
    match e {
        HipError::ConstraintsNotSatisfied => ProvingError::ConstraintsNotSatisfied,
        // the reference has no error channel for resources either: `vec![..]` aborts (trace_builder.rs:29)
        other => panic!("nexus-hip: {other:?}"),
    }
}

impl<C: MachineChip + Sync> Machine<C> {
    pub fn prove_hip(trace: &impl Trace, view: &View) -> Result<Proof, ProvingError> {
        Self::prove_hip_with_extensions(&[], trace, view)
    }

    pub fn prove_hip_with_extensions(
        extensions: &[ExtensionComponent],
        trace: &impl Trace,
        view: &View,
    ) -> Result<Proof, ProvingError> {
        // ---- machine.rs:135-183: sizes, preprocessed / main / program traces on the CPU — the reference's own statements, hoisted
        // (README.md edit 2) into `Machine::generate_traces` so that both provers run the same ones
        let init_memory = [
            view.get_ro_initial_memory(),
            view.get_rw_initial_memory(),
            view.get_public_input(),
        ]
        .concat();
        let GeneratedTraces {
            log_size,
            extensions_config,
            preprocessed_trace,
            finalized_trace,
            finalized_program_trace,
            all_log_sizes,
            mut prover_side_note,
            program_trace_ref,
        } = Self::generate_traces(extensions, trace, view, &init_memory);
        let extensions_iter = BASE_EXTENSIONS.iter().chain(extensions);

        // ---- machine.rs:184-206: config, twiddles (inside the session), channel seeding
        let config = PcsConfig::default();
        let max_log = log_size.max(all_log_sizes.iter().copied().max().unwrap_or(0));
        let cfg = pcs_config(&config, LOG_CONSTRAINT_DEGREE);
        let device: i32 = std::env::var("NEXUS_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut session = Session::new(&cfg, max_log, device).map_err(to_proving_error)?;
        for byte in view.view_associated_data().unwrap_or_default() {
            session.mix_u64(byte.into());
        }
        all_log_sizes.iter().for_each(|log_size| session.mix_u64(*log_size as u64));

        // ---- machine.rs:208-228: preprocessed tree (base preprocessed + program columns, then the extensions')
        let extension_traces: Vec<ComponentTrace> = extensions_iter
            .clone()
            .zip(all_log_sizes.get(1..).unwrap_or_default())
            .map(|(ext, log_size)| {
                ext.generate_component_trace(*log_size, program_trace_ref, &mut prover_side_note)
            })
            .collect();
        let mut tree0: Vec<SimdEval> = preprocessed_trace
            .clone()
            .into_circle_evaluation()
            .into_iter()
            .chain(finalized_program_trace.clone().into_circle_evaluation())
            .collect();
        for extension_trace in &extension_traces {
            tree0.extend(extension_trace.to_circle_evaluation(PREPROCESSED_TRACE_IDX));
        }
        // Shape pass: which columns of trees 0 / 1 the relation entries read is a property of the AIR, not of the lookup elements — the
        // components recorded once with dummy elements say it, and only those columns' evaluations are kept beyond their commit.
        let reads = {
            let dummy = AllLookupElements::dummy();
            let mut shape_pass = TraceLocations::default();
            let mut shape: Vec<RecordedComponent> = vec![record_component(
                &MachineEval::<C>::new(log_size, dummy.clone(), extensions_config.clone()),
                &mut shape_pass,
                SecureField::zero(),
            )];
            for (ext, log_size) in extensions_iter.clone().zip(all_log_sizes.get(1..).unwrap_or_default()) {
                shape.push(ext.to_recorded_component(&mut shape_pass, &dummy, *log_size, SecureField::zero()));
            }
            let n_main = finalized_trace.clone().into_circle_evaluation().len()
                + extension_traces.iter().map(|t| t.to_circle_evaluation(ORIGINAL_TRACE_IDX).len()).sum::<usize>();
            columns_read_by_fractions(&shape, [tree0.len(), n_main])
        };
        let kept0 = commit_tree_keeping_evaluations(&mut session, &tree0, Some(&reads[0])).map_err(to_proving_error)?;
        drop(tree0);

        // ---- machine.rs:230-237: main tree
        let mut tree1: Vec<SimdEval> = finalized_trace.clone().into_circle_evaluation();
        for extension_trace in &extension_traces {
            tree1.extend(extension_trace.to_circle_evaluation(ORIGINAL_TRACE_IDX));
        }
        let kept1 = commit_tree_keeping_evaluations(&mut session, &tree1, Some(&reads[1])).map_err(to_proving_error)?;
        drop(tree1);

        // ---- machine.rs:239-263: lookup elements from the session's channel; the interaction trace from the components' own relation
        // entries on the device; claimed sums; interaction tree.  The components are recorded a first time with zero claimed sums: their
        // fraction programs are what generates the trace (the constraints are recorded again below, once the sums are known — the
        // reference, too, builds its components after the interaction trace: machine.rs:264-285).
        let mut lookup_elements = AllLookupElements::default();
        C::draw_lookup_elements(&mut lookup_elements, &mut host_channel_at(&session), &extensions_config);
        let mut first_pass = TraceLocations::default();
        let mut generators: Vec<RecordedComponent> = vec![record_component(
            &MachineEval::<C>::new(log_size, lookup_elements.clone(), extensions_config.clone()),
            &mut first_pass,
            SecureField::zero(),
        )];
        for (ext, log_size) in extensions_iter.clone().zip(all_log_sizes.get(1..).unwrap_or_default()) {
            generators.push(ext.to_recorded_component(&mut first_pass, &lookup_elements, *log_size, SecureField::zero()));
        }
        let claimed = interaction_tree_on_device(&mut session, &generators, [&kept0, &kept1]).map_err(to_proving_error)?;
        drop(generators);
        session.free_columns();                                  // the kept evaluations: nothing reads them after the interaction tree
        let all_claimed_sums: Vec<SecureField> = claimed
            .iter()
            .map(secure_from_words)
            .collect();
        let claimed_sum = all_claimed_sums[0];
        if std::env::var("NEXUS_HIP_CHECK_LOGUP").is_ok() {
            // the reference's CPU generator as a cross-check of the device trace (debug only: it is the 16.8 GB this route avoids)
            let (_, cpu_sum) = generate_interaction_trace::<C>(&finalized_trace, &preprocessed_trace, &finalized_program_trace, &lookup_elements);
            assert_eq!(cpu_sum, claimed_sum, "device logup trace and fill_interaction_trace disagree on the main component's claimed sum");
        }
        drop(extension_traces);
        let claimed_words: Vec<u32> = claimed.iter().flatten().copied().collect();
        session.mix_felts(&claimed_words);
        session.tree_commit().map_err(to_proving_error)?;

        // ---- machine.rs:264-285: the components, recorded instead of instantiated (same evals, same order, same claimed sums)
        let mut locations = TraceLocations::default();
        let mut components: Vec<RecordedComponent> = vec![record_component(
            &MachineEval::<C>::new(log_size, lookup_elements.clone(), extensions_config.clone()),
            &mut locations,
            claimed_sum,
        )];
        for ((ext, claimed_sum), log_size) in extensions_iter
            .zip(all_claimed_sums.get(1..).unwrap_or_default())
            .zip(all_log_sizes.get(1..).unwrap_or_default())
        {
            components.push(ext.to_recorded_component(&mut locations, &lookup_elements, *log_size, *claimed_sum));
        }

        // ---- machine.rs:286-296: stwo::prover::prove on the device; the reference's Proof from its postcard bytes
        let words = session.prove(&components).map_err(to_proving_error)?;
        let bytes = proof_bytes(&words, &claimed_words, &all_log_sizes).map_err(to_proving_error)?;
        let proof: Proof = postcard::from_bytes(&bytes).expect("nx_proof_serialize_stwo emits the serde layout of machine::Proof");
        debug_assert!(proof.claimed_sum == all_claimed_sums && proof.log_size == all_log_sizes);
        debug_assert!(INTERACTION_TRACE_IDX == 2 && SecureField::zero().is_zero());
        Ok(proof)
    }
}
