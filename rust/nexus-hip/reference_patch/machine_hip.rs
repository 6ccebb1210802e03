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
    core::{
        channel::Blake2sChannel,
        fields::{m31::BaseField, qm31::SecureField},
        pcs::PcsConfig,
        vcs::blake2_hash::Blake2sHash,
    },
    prover::{
        backend::simd::SimdBackend,
        poly::{circle::CircleEvaluation, BitReversedOrder},
        ProvingError,
    },
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
use nexus_hip::{proof_bytes, HipError, RecordedComponent, Session};
use nexus_hip_sys as sys;

type SimdEval = CircleEvaluation<SimdBackend, BaseField, BitReversedOrder>;

fn q4(s: SecureField) -> [u32; 4] {
    let a = s.to_m31_array();
    [a[0].0, a[1].0, a[2].0, a[3].0]
}
/// `C::draw_lookup_elements(&mut lookup_elements, prover_channel, ..)` (machine.rs:239-240) takes `&mut impl Channel`, and `Channel`
/// is `Default + Clone`: it has to be a real channel.  A host `Blake2sChannel` standing where the session's transcript stands does it:
/// the lookup elements are DRAWN (draws hash the digest with a counter and leave the digest alone), and what follows — `mix_felts` of
/// the claimed sums, machine.rs:262 — replaces the digest by H(digest ‖ felts) and resets the counter, so the session's transcript
/// needs no replay of the draws.  [upstream-recollection: `Blake2sChannel::update_digest` is public; the oracle's channel
/// (oracle/blake2s.h, csrc/host/channel.h) restates the same rule and the parity suite runs the draw-then-mix sequence through it]
fn host_channel_at(session: &Session) -> Blake2sChannel {
    let mut ch = Blake2sChannel::default();
    ch.update_digest(Blake2sHash(session.channel_digest()));
    ch
}

/// host pointers of a batch of SimdBackend evaluations (bit-reversed circle-domain order already: `finalize_columns` ran on the CPU,
/// trace/utils.rs:94-106) and their log sizes, in commit order
fn host_columns(evals: &[SimdEval]) -> (Vec<*const u32>, Vec<u32>) {
    let ptrs = evals.iter().map(|e| e.values.as_slice().as_ptr() as *const u32).collect();
    let logs = evals.iter().map(|e| e.domain.log_size()).collect();
    (ptrs, logs)
}

/// TreeBuilder::extend_evals(..) + commit(channel) for one trace tree (machine.rs:208-228, :230-237): the columns stay in the
/// SimdBackend evaluations' memory and go up in chunks under the commit's own transforms (nx_prover_tree_commit_host).  Every column's
/// EVALUATIONS are also kept on the device (cloned as the chunks arrive: the reference's `finalized_trace.clone()`, machine.rs:232) —
/// the fraction programs read them after the commit has turned the tree's own columns into coefficients.  Returns the kept columns.
fn commit_tree_keeping_evaluations(session: &mut Session, evals: &[SimdEval]) -> Result<Vec<*const u32>, HipError> {
    let (host, logs) = host_columns(evals);
    session.tree_begin(&logs)?;
    let mut keep: Vec<(u32, *mut u32)> = Vec::with_capacity(logs.len());
    let mut i = 0;
    while i < logs.len() {                                   // one allocation per run of equally sized columns
        let mut j = i;
        while j < logs.len() && logs[j] == logs[i] { j += 1; }
        for (k, p) in session.alloc_columns(j - i, logs[i])?.into_iter().enumerate() { keep.push(((i + k) as u32, p)); }
        i = j;
    }
    session.tree_commit_host(&host, false, &keep)?;
    Ok(keep.iter().map(|k| k.1 as *const u32).collect())
}

/// The interaction tree (machine.rs:242-263) from the components' recorded relation entries: `recorded[c]` is component c recorded with a
/// ZERO claimed sum (the fractions do not depend on it); its columns of trees 0 / 1 are looked up in the kept evaluations, its columns of
/// tree 2 are the session's own (tree_begin), filled in place.  Returns the claimed sums in component order.
fn interaction_tree_on_device(session: &mut Session, recorded: &[RecordedComponent], kept: [&[*const u32]; 2]) -> Result<Vec<[u32; 4]>, HipError> {
    let mut logs: Vec<u32> = Vec::new();
    for c in recorded { logs.extend(std::iter::repeat(c.log_size).take(4 * c.n_logup_cols as usize)); }
    let tree2 = session.tree_begin(&logs)?;
    let mut claimed = Vec::with_capacity(recorded.len());
    for c in recorded {
        let cols: Vec<*const u32> = c.col_tree.iter().zip(&c.col_index).map(|(&t, &i)| if t < 2 { kept[t as usize][i as usize] } else { std::ptr::null() }).collect();
        // the component's interaction columns, in the order it declared them (TraceLocations hands them out consecutively)
        let out: Vec<*mut u32> = c.col_tree.iter().zip(&c.col_index).filter(|(&t, _)| t == 2).map(|(_, &i)| tree2[i as usize]).collect();
        claimed.push(session.logup_trace(c, &cols, &out)?);
    }
    Ok(claimed)
}

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
        let cfg = sys::nx_pcs_config {
            pow_bits: config.pow_bits,
            log_blowup: config.fri_config.log_blowup_factor,
            n_queries: config.fri_config.n_queries as u32,
            log_last_layer_degree_bound: config.fri_config.log_last_layer_degree_bound,
            hash_mode: sys::NX_HASH_BLAKE2S as u32,
            fri_alpha_mode: sys::NX_FRI_ALPHA_PREV as u32,
            log_constraint_degree: LOG_CONSTRAINT_DEGREE,
        };
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
        let kept0 = commit_tree_keeping_evaluations(&mut session, &tree0).map_err(to_proving_error)?;
        drop(tree0);

        // ---- machine.rs:230-237: main tree
        let mut tree1: Vec<SimdEval> = finalized_trace.clone().into_circle_evaluation();
        for extension_trace in &extension_traces {
            tree1.extend(extension_trace.to_circle_evaluation(ORIGINAL_TRACE_IDX));
        }
        let kept1 = commit_tree_keeping_evaluations(&mut session, &tree1).map_err(to_proving_error)?;
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
        let all_claimed_sums: Vec<SecureField> = claimed
            .iter()
            .map(|w| SecureField::from_m31_array([BaseField::from(w[0]), BaseField::from(w[1]), BaseField::from(w[2]), BaseField::from(w[3])]))
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
