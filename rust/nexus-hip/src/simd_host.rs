//! What the two reference patches (`reference_patch/machine_hip.rs` for `prover/src/machine.rs`, `reference_patch/prove2_hip.rs` for
//! `prover2/machine/src/prove.rs`) share: the handful of steps that take the reference's `SimdBackend` evaluations — the trace columns
//! its chips filled on the CPU — onto a `Session`.  Compiled only with `RUSTFLAGS="--cfg stwo_traits"` (it names Stwo types).
//!
//!   reference                                                                here
//!   tree_builder.extend_evals(..) + .commit(channel)                         `commit_tree_keeping_evaluations`
//!       (machine.rs:208-237, prove.rs:70-84)
//!   draw_lookup_elements(.., channel)  (machine.rs:239-240, prove.rs:86-89)  a host `Blake2sChannel` at the session's digest: `host_channel_at`
//!   generate_interaction_trace + extend_evals                                `interaction_tree_on_device`: the recorded relation entries run as
//!       (machine.rs:242-263, prove.rs:91-105)                                fraction programs straight into the interaction tree's columns
//!   PcsConfig::default() -> CommitmentSchemeProver::new (machine.rs:184-203) `pcs_config`
//!   the verifier's re-commitment of the preprocessed tree (R10:               `preprocessed_root_on_device`
//!       machine.rs:363-411, prover2 verify.rs:118-135)
use crate::{HipError, RecordedComponent, Session};
use nexus_hip_sys as sys;

use stwo::core::channel::Blake2sChannel;
use stwo::core::fields::m31::BaseField;
use stwo::core::fields::qm31::SecureField;
use stwo::core::pcs::PcsConfig;
use stwo::core::vcs::blake2_hash::Blake2sHash;
use stwo::prover::backend::simd::SimdBackend;
use stwo::prover::poly::circle::CircleEvaluation;
use stwo::prover::poly::BitReversedOrder;

pub type SimdEval = CircleEvaluation<SimdBackend, BaseField, BitReversedOrder>;

/// the reference's `PcsConfig` as the C ABI's; `log_constraint_degree`: the largest `max_constraint_log_degree_bound - log_size` of the
/// statement's components (v1: `LOG_CONSTRAINT_DEGREE`, components/mod.rs:12; v2: the largest `C::LOG_CONSTRAINT_DEGREE_BOUND`,
/// framework/eval.rs:14-16) — it sizes the session's twiddles the way machine.rs:186-194 / prove.rs:53-58 size Stwo's
pub fn pcs_config(config: &PcsConfig, log_constraint_degree: u32) -> sys::nx_pcs_config {
    sys::nx_pcs_config {
        pow_bits: config.pow_bits,
        log_blowup: config.fri_config.log_blowup_factor,
        n_queries: config.fri_config.n_queries as u32,
        log_last_layer_degree_bound: config.fri_config.log_last_layer_degree_bound,
        hash_mode: sys::NX_HASH_BLAKE2S as u32,
        fri_alpha_mode: sys::NX_FRI_ALPHA_PREV as u32,
        log_constraint_degree,
    }
}

pub fn secure_words(s: SecureField) -> [u32; 4] {
    let a = s.to_m31_array();
    [a[0].0, a[1].0, a[2].0, a[3].0]
}
pub fn secure_from_words(w: &[u32; 4]) -> SecureField {
    SecureField::from_m31_array([BaseField::from(w[0]), BaseField::from(w[1]), BaseField::from(w[2]), BaseField::from(w[3])])
}

/// `draw_lookup_elements(&mut lookup_elements, prover_channel, ..)` (machine.rs:239-240; prove.rs:86-89 — there `&mut Blake2sChannel`
/// by name) takes a real channel.  A host `Blake2sChannel` standing where the session's transcript stands does it: the lookup elements
/// are DRAWN (draws hash the digest with a counter and leave the digest alone), and what follows — `mix_felts` of the claimed sums,
/// machine.rs:262 / prove.rs:104 — replaces the digest by H(digest ‖ felts) and resets the counter, so the session's transcript needs
/// no replay of the draws.  [upstream-recollection: `Blake2sChannel::update_digest` is public; the oracle's channel (oracle/blake2s.h,
/// csrc/host/channel.h) restates the same rule and the parity suite runs the draw-then-mix sequence through it]
pub fn host_channel_at(session: &Session) -> Blake2sChannel {
    let mut ch = Blake2sChannel::default();
    ch.update_digest(Blake2sHash(session.channel_digest()));
    ch
}

/// host pointers of a batch of SimdBackend evaluations (bit-reversed circle-domain order already: `finalize_columns` ran on the CPU,
/// trace/utils.rs:94-106, prover2/trace/src/utils.rs:102-114) and their log sizes, in commit order
pub fn host_columns(evals: &[SimdEval]) -> (Vec<*const u32>, Vec<u32>) {
    let ptrs = evals.iter().map(|e| e.values.as_slice().as_ptr() as *const u32).collect();
    let logs = evals.iter().map(|e| e.domain.log_size()).collect();
    (ptrs, logs)
}

/// Which columns of trees 0 / 1 the components' FRACTION PROGRAMS load (`NX_C_LOAD` / `NX_C_LOADE` of a component column whose
/// `col_tree` is < 2): the only evaluations that have to outlive their tree's commit.  The load set of a recorded component does not
/// depend on the lookup elements (they are E constants of the program), so a "shape pass" — the components recorded once with
/// `AllLookupElements::dummy()` before anything is committed (reference prover/src/components/lookups.rs:66-68, prover2/machine/src/
/// lookups/mod.rs) — yields it.  `n_cols[t]`: the column count of tree t.  (keeping EVERY column of both trees added ~50 % to
/// the peak HBM of the commit and prove phases of a v1-sized trace.)
pub fn columns_read_by_fractions(recorded: &[RecordedComponent], n_cols: [usize; 2]) -> [Vec<bool>; 2] {
    let mut reads = [vec![false; n_cols[0]], vec![false; n_cols[1]]];
    for c in recorded {
        for ins in &c.logup_program {
            let width = if ins.op == sys::NX_C_LOAD as u32 { 1 } else if ins.op == sys::NX_C_LOADE as u32 { 4 } else { 0 };
            for k in 0..width {
                let col = ins.a as usize + k;
                let (t, i) = (c.col_tree[col] as usize, c.col_index[col] as usize);
                if t < 2 { reads[t][i] = true; }
            }
        }
    }
    reads
}

/// TreeBuilder::extend_evals(..) + commit(channel) for one trace tree: the columns stay in the SimdBackend evaluations' memory and go up
/// in chunks under the commit's own transforms (nx_prover_tree_commit_host).  The EVALUATIONS of the columns marked in `keep_mask`
/// (`columns_read_by_fractions`; `None`: every column) are also kept on the device (cloned as the chunks arrive: the reference's
/// `finalized_trace.clone()`, machine.rs:232; `to_circle_evaluation`'s clone, prover2/trace/src/component.rs:63-79) — the fraction
/// programs read them after the commit has turned the tree's own columns into coefficients.  Returns one pointer per column of the tree,
/// in commit order: the kept evaluations, null for a column that was not kept.  `Session::free_columns` releases them (call it right
/// after `interaction_tree_on_device`: they are not needed by `Session::prove`).
pub fn commit_tree_keeping_evaluations(session: &mut Session, evals: &[SimdEval], keep_mask: Option<&[bool]>) -> Result<Vec<*const u32>, HipError> {
    let (host, logs) = host_columns(evals);
    session.tree_begin(&logs)?;
    let wanted = |i: usize| keep_mask.map_or(true, |m| m.get(i).copied().unwrap_or(false));
    let mut keep: Vec<(u32, *mut u32)> = Vec::new();
    let mut kept: Vec<*const u32> = vec![std::ptr::null(); logs.len()];
    let mut i = 0;
    while i < logs.len() {                                   // one allocation per run of equally sized columns: the kept ones of the run
        let mut j = i;
        while j < logs.len() && logs[j] == logs[i] { j += 1; }
        let run: Vec<usize> = (i..j).filter(|&k| wanted(k)).collect();
        if !run.is_empty() {
            for (k, p) in run.iter().zip(session.alloc_columns(run.len(), logs[i])?) { keep.push((*k as u32, p)); kept[*k] = p as *const u32; }
        }
        i = j;
    }
    session.tree_commit_host(&host, false, &keep)?;
    Ok(kept)
}

/// The interaction tree from the components' recorded relation entries: `recorded[c]` is component c recorded with a ZERO claimed sum
/// (the fractions do not depend on it); its columns of trees 0 / 1 are looked up in the kept evaluations, its columns of tree 2 are
/// the session's own (tree_begin), filled in place.  Returns the claimed sums in component order; the caller mixes them and commits
/// (`mix_felts`, `tree_commit`: machine.rs:262-263, prove.rs:104-105).  Nothing of tree 2 crosses PCIe.
pub fn interaction_tree_on_device(session: &mut Session, recorded: &[RecordedComponent], kept: [&[*const u32]; 2]) -> Result<Vec<[u32; 4]>, HipError> {
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

/// The verifier's "simulate the prover and compute expected commitment to preprocessed trace" (machine.rs:363-411; prover2
/// verify.rs:118-135): a `CommitmentSchemeProver::<SimdBackend, _>` built only to interpolate, extend and hash the preprocessed columns
/// and read `roots()[PREPROCESSED_TRACE_IDX]` — on the CPU the dominant cost of `verify`.  Here: the same
/// columns through a session's first tree; the root is the 32 bytes the reference compares with `proof.commitments[0]`.
/// `max_log_size` / `log_constraint_degree`: what the reference passes to `precompute_twiddles` there (largest log size; v1
/// `LOG_CONSTRAINT_DEGREE`, v2 the largest per-component bound over its log size) — the tower only has to be tall enough.
pub fn preprocessed_root_on_device(config: &PcsConfig, evals: &[SimdEval], max_log_size: u32, log_constraint_degree: u32, device: i32) -> Result<Blake2sHash, HipError> {
    let cfg = pcs_config(config, log_constraint_degree);
    let mut session = Session::new(&cfg, max_log_size, device)?;
    let (host, logs) = host_columns(evals);
    session.tree_begin(&logs)?;
    Ok(Blake2sHash(session.tree_commit_host(&host, false, &[])?))
}
