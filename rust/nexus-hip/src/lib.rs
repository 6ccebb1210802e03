//! `nexus-hip` — the Rust host side of the MI355X prover backend: Stwo's backend traits over `libnexus_hip.so` (`HipBackend`,
//! `backend.rs`), the recording `EvalAtRow` that carries the reference's AIR closures across the C ABI (`record.rs`), and the
//! whole-prove session (`Session`) that `reference_patch/machine_hip.rs` drives in place of `prover/src/machine.rs:184-296`.
//!
//! NOT COMPILED in the repository's build image (no Rust toolchain there, no network for the `stwo` git dependency).  What is
//! checked without a compiler (tests/test_rust_shim_cpu.py): every `sys::nx_*` call names a function `nexus-hip-sys` declares, with
//! the declared number of arguments; every Stwo / nexus path named here is one the reference itself names (tests/golden/
//! reference_use_paths.txt) or is listed with its reason in rust/UNOBSERVED_PATHS.txt; every backend trait of INTEGRATION.md §2 has
//! its `impl … for HipBackend`; brackets balance.  Signatures marked [upstream-recollection] must meet the pinned crate
//! (stwo @ 0790eba) on first compile.
//!
//! Two routes, both keeping `nexus_vm_prover::prove(&impl Trace, &View) -> Result<Proof, ProvingError>` unchanged:
//!  1. per operation: `HipBackend` implements the traits `stwo::prover::prove` is generic over (`backend.rs`); the AIR reaches the
//!     device as a `RecordedComponent` whose `ComponentProver<HipBackend>` evaluates the recorded program with `nx_air_eval`;
//!  2. session: after trace generation the whole of machine.rs:184-296 runs on the device — commits, recorded AIR, OODS, DEEP
//!     quotients, FRI, PoW, decommitment — and only the proof comes back (`reference_patch/machine_hip.rs`).
#![allow(clippy::missing_safety_doc)]

use nexus_hip_sys as sys;
use std::ffi::CStr;
use std::marker::PhantomData;
use std::sync::{Mutex, MutexGuard, OnceLock};

pub mod record;
#[cfg(stwo_traits)]
pub mod backend;
#[cfg(stwo_traits)]
pub mod simd_host;

// ------------------------------------------------------------------------------------------------ context ----
/// Stwo's `Backend` is a zero-sized type whose methods are called from whatever (rayon) thread Stwo likes, while an `nx_ctx` is
/// single-threaded at the protocol level.  One process-wide context behind a mutex: every backend op takes the lock for the
/// duration of its (asynchronous, stream-ordered) enqueue — ops of one prove are sequential anyway (one `&mut Blake2sChannel`,
/// reference machine.rs:197), and a second concurrent prove uses a `Session` with a context of its own instead.
pub(crate) struct Ctx(pub(crate) *mut sys::nx_ctx);
unsafe impl Send for Ctx {}
static CTX: OnceLock<Mutex<Ctx>> = OnceLock::new();

pub(crate) fn ctx() -> MutexGuard<'static, Ctx> {
    CTX.get_or_init(|| {
        let dev: i32 = std::env::var("NEXUS_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut c = std::ptr::null_mut();
        let rc = unsafe { sys::nx_ctx_create(dev, &mut c) };
        if rc != sys::NX_OK {
            panic!("nx_ctx_create({dev}) failed ({rc}): {}", last_error(std::ptr::null()));
        }
        Mutex::new(Ctx(c))
    })
    .lock()
    .expect("nexus-hip context poisoned")
}

pub(crate) fn last_error(c: *const sys::nx_ctx) -> String {
    unsafe { CStr::from_ptr(sys::nx_last_error(c)) }.to_string_lossy().into_owned()
}

/// What can go wrong below the C ABI, as the caller can act on it.  `ConstraintsNotSatisfied` is the reference's
/// `ProvingError::ConstraintsNotSatisfied` (core/src/lib.rs:22-24) and is what `prove` returns; `Argument` is a caller mistake
/// (unknown option, sizes that do not fit) and is returned, not a panic; `Device` (HIP / RCCL / out of memory) aborts a prove the
/// way the reference's `vec![..]` allocation failure does (trace_builder.rs:29) — `check` panics on it, `try_check` returns it.
#[derive(Debug, Clone, PartialEq, Eq)]
pub enum HipError {
    ConstraintsNotSatisfied,
    Argument(String),
    Device(i32, String),
}

pub(crate) fn try_check(c: *const sys::nx_ctx, rc: i32) -> Result<(), HipError> {
    match rc {
        sys::NX_OK => Ok(()),
        sys::NX_ERR_PROTOCOL => Err(HipError::ConstraintsNotSatisfied),
        sys::NX_ERR_ARG => Err(HipError::Argument(last_error(c))),
        _ => Err(HipError::Device(rc, last_error(c))),
    }
}
/// Backend-trait methods have no error channel (Stwo's signatures return values): device failures panic there.
pub(crate) fn check(c: *const sys::nx_ctx, rc: i32) {
    if let Err(e) = try_check(c, rc) {
        panic!("libnexus_hip: {e:?}");
    }
}

// ------------------------------------------------------------------------------------------------ columns ----
/// A device column of `len` elements of `T`, `words_per` u32 words each, from `nx_alloc`; `Drop` returns them (Stwo moves evaluations
/// into the tree builder by value — reference machine.rs:209-215 — so ownership maps onto a handle with an explicit free).
/// Layout: base field = `len` words; secure field = 4 coordinate blocks of `len` words (`SecureColumnByCoords` order);
/// Blake2s hash = `len` x 8 words.
pub struct HipColumn<T> {
    pub(crate) ptr: *mut u32,
    pub(crate) len: usize,
    pub(crate) words_per: usize,
    _t: PhantomData<T>,
}
unsafe impl<T> Send for HipColumn<T> {}
unsafe impl<T> Sync for HipColumn<T> {}

impl<T> HipColumn<T> {
    pub fn zeros_words(len: usize, words_per: usize) -> Self {
        let g = ctx();
        let mut p = std::ptr::null_mut();
        check(g.0, unsafe { sys::nx_alloc(g.0, (len * words_per).max(1), &mut p) });
        check(g.0, unsafe { sys::nx_memset_zero(g.0, p, len * words_per) });
        Self { ptr: p, len, words_per, _t: PhantomData }
    }
    pub fn from_host_words(words: &[u32], words_per: usize) -> Self {
        let g = ctx();
        let mut p = std::ptr::null_mut();
        check(g.0, unsafe { sys::nx_alloc(g.0, words.len().max(1), &mut p) });
        check(g.0, unsafe { sys::nx_upload(g.0, p, words.as_ptr(), words.len()) });
        Self { ptr: p, len: words.len() / words_per, words_per, _t: PhantomData }
    }
    pub fn to_host_words(&self) -> Vec<u32> {
        let g = ctx();
        let mut v = vec![0u32; self.len * self.words_per];
        check(g.0, unsafe { sys::nx_download(g.0, v.as_mut_ptr(), self.ptr, v.len()) });
        v
    }
    pub fn word_at(&self, i: usize) -> u32 {
        let g = ctx();
        let (p, idx, mut out) = (self.ptr as *const u32, i as u64, 0u32);
        check(g.0, unsafe { sys::nx_gather(g.0, &p, &idx, 1, &mut out) });
        out
    }
    pub fn set_word(&mut self, i: usize, w: u32) {
        let g = ctx();
        check(g.0, unsafe { sys::nx_upload(g.0, self.ptr.add(i), &w, 1) });
    }
    pub fn clone_on_device(&self) -> Self {
        let g = ctx();
        let mut p = std::ptr::null_mut();
        let n = self.len * self.words_per;
        check(g.0, unsafe { sys::nx_alloc(g.0, n.max(1), &mut p) });
        check(g.0, unsafe { sys::nx_copy(g.0, p, self.ptr, n) });
        Self { ptr: p, len: self.len, words_per: self.words_per, _t: PhantomData }
    }
    /// coordinate block `k` of a secure column (`k` < 4), or the column itself for `words_per == 1`
    pub fn coord_ptr(&self, k: usize) -> *mut u32 { unsafe { self.ptr.add(k * self.len) } }
    pub fn as_ptr(&self) -> *const u32 { self.ptr }
    pub fn as_mut_ptr(&mut self) -> *mut u32 { self.ptr }
    pub fn len(&self) -> usize { self.len }
    pub fn is_empty(&self) -> bool { self.len == 0 }
}
impl<T> Clone for HipColumn<T> { fn clone(&self) -> Self { self.clone_on_device() } }
impl<T> std::fmt::Debug for HipColumn<T> {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result { write!(f, "HipColumn {{ len: {}, words_per: {} }}", self.len, self.words_per) }
}
impl<T> Drop for HipColumn<T> {
    fn drop(&mut self) {
        if let Some(m) = CTX.get() {
            if let Ok(g) = m.lock() { unsafe { sys::nx_free(g.0, self.ptr) }; }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backend ----
#[derive(Copy, Clone, Debug, Default, serde::Serialize, serde::Deserialize)]
pub struct HipBackend;

/// `TwiddleTree<HipBackend>`'s payload: ONE device object holds the forward and the inverse tables (and their doubled forms), so
/// `twiddles` and `itwiddles` of the tree are the same `Arc<HipTwiddles>`.
pub struct HipTwiddles(pub *mut sys::nx_twiddles);
unsafe impl Send for HipTwiddles {}
unsafe impl Sync for HipTwiddles {}
impl Drop for HipTwiddles {
    fn drop(&mut self) {
        // the context's allocator cache is touched: same lock as every other op
        if let Some(m) = CTX.get() {
            if let Ok(_g) = m.lock() { unsafe { sys::nx_twiddles_destroy(self.0) }; }
        }
    }
}

/// The operations behind the Stwo trait methods, on raw device pointers: each `impl <Trait> for HipBackend` in `backend.rs` is a thin
/// adapter over one of these (INTEGRATION.md §2 lists trait method -> export).
pub mod ops {
    use super::*;

    /// `ColumnOps<BaseField>::bit_reverse_column` (reference prover/src/trace/utils.rs:101)
    pub fn bit_reverse(col: *mut u32, log_size: u32) { let g = ctx(); check(g.0, unsafe { sys::nx_bit_reverse(g.0, col, log_size) }); }
    /// `ColumnOps<SecureField>::bit_reverse_column` on the 4 coordinate columns
    pub fn bit_reverse_secure(col4: &[*mut u32; 4], log_size: u32) { let g = ctx(); check(g.0, unsafe { sys::nx_bit_reverse_secure(g.0, col4.as_ptr(), log_size) }); }
    /// `PolyOps::new_canonical_ordered` / reference `finalize_columns` (prover/src/trace/utils.rs:94-106): natural coset order ->
    /// bit-reversed circle-domain order, on the device
    pub fn finalize_columns(src_natural: &[*const u32], dst: &[*mut u32], log_size: u32) {
        let g = ctx(); check(g.0, unsafe { sys::nx_finalize_columns(g.0, src_natural.as_ptr(), dst.as_ptr(), src_natural.len() as u32, log_size) });
    }
    /// `PolyOps::precompute_twiddles(coset)` for a half coset of log size `log_half_coset` (reference machine.rs:186-194)
    pub fn precompute_twiddles(log_half_coset: u32) -> HipTwiddles {
        let g = ctx(); let mut t = std::ptr::null_mut();
        check(g.0, unsafe { sys::nx_twiddles_create(g.0, log_half_coset, &mut t) });
        HipTwiddles(t)
    }
    /// `PolyOps::interpolate_columns` for one group of equally sized columns (in place: evaluations -> coefficients)
    pub fn interpolate(tw: &HipTwiddles, cols: &[*mut u32], log_size: u32) { let g = ctx(); check(g.0, unsafe { sys::nx_interpolate_batch(g.0, tw.0, cols.as_ptr(), cols.len() as u32, log_size) }); }
    /// `PolyOps::evaluate_polynomials` for one group
    pub fn evaluate(tw: &HipTwiddles, polys: &[*const u32], log_size: u32, log_expand: u32, out: &[*mut u32]) {
        let g = ctx(); check(g.0, unsafe { sys::nx_evaluate_batch(g.0, tw.0, polys.as_ptr(), polys.len() as u32, log_size, log_expand, out.as_ptr()) });
    }
    /// `TreeBuilder::extend_evals` + `commit` fused for one group: evaluations -> coefficients (in place) + LDE
    pub fn lde(tw: &HipTwiddles, cols: &[*mut u32], log_size: u32, log_blowup: u32, lde: &[*mut u32]) {
        let g = ctx(); check(g.0, unsafe { sys::nx_lde_batch(g.0, tw.0, cols.as_ptr(), cols.len() as u32, log_size, log_blowup, lde.as_ptr()) });
    }
    /// `PolyOps::eval_at_point`, batched: out[i] = polys[poly_idx[i]](points[i]); points are 8 words (x, y), results 4 words
    pub fn eval_at_points(polys: &[*const u32], log_size: u32, poly_idx: &[u32], points: &[u32], out: &mut [u32]) {
        let g = ctx(); check(g.0, unsafe { sys::nx_eval_at_points(g.0, polys.as_ptr(), log_size, poly_idx.as_ptr(), points.as_ptr(), poly_idx.len() as u32, out.as_mut_ptr()) });
    }
    /// `MerkleOps::<Blake2sMerkleHasher>::commit_on_layer` — one layer, the trait's own shape
    pub fn commit_on_layer(log_size: u32, prev_layer: *const u32, cols: &[*const u32], out: *mut u32) {
        let g = ctx(); check(g.0, unsafe { sys::nx_merkle_commit_on_layer(g.0, log_size, prev_layer, cols.as_ptr(), cols.len() as u32, out) });
    }
    /// `MerkleProver::commit` as one call (whole tree), and its root
    pub fn merkle_commit(cols: &[*const u32], log_sizes: &[u32]) -> (*mut sys::nx_tree, [u8; 32]) {
        let g = ctx(); let mut t = std::ptr::null_mut(); let mut root = [0u8; 32];
        check(g.0, unsafe { sys::nx_merkle_commit(g.0, cols.as_ptr(), log_sizes.as_ptr(), cols.len() as u32, &mut t) });
        check(g.0, unsafe { sys::nx_merkle_root(g.0, t, root.as_mut_ptr()) });
        (t, root)
    }
    /// `QuotientOps::accumulate_quotients`
    #[allow(clippy::too_many_arguments)]
    pub fn accumulate_quotients(log_size: u32, cols: &[*const u32], random_coeff: &[u32; 4], points: &[u32], batch_counts: &[u32], col_idx: &[u32], values: &[u32], out4: &[*mut u32; 4]) {
        let g = ctx();
        check(g.0, unsafe { sys::nx_accumulate_quotients(g.0, log_size, cols.as_ptr(), cols.len() as u32, random_coeff.as_ptr(), batch_counts.len() as u32, points.as_ptr(), batch_counts.as_ptr(), col_idx.as_ptr(), values.as_ptr(), out4.as_ptr()) });
    }
    /// `FriOps::fold_circle_into_line` / `fold_line` / `decompose`
    pub fn fold_circle_into_line(tw: &HipTwiddles, dst4: &[*mut u32; 4], src4: &[*const u32; 4], src_log: u32, alpha: &[u32; 4]) {
        let g = ctx(); check(g.0, unsafe { sys::nx_fold_circle_into_line(g.0, tw.0, dst4.as_ptr(), src4.as_ptr(), src_log, alpha.as_ptr()) });
    }
    /// the line domain of log size `src_log` is always `half_odds(src_log)` (`half_odds(k).double() == half_odds(k - 1)`), so the
    /// export's `n_doublings` argument carries no information: 0
    pub fn fold_line(tw: &HipTwiddles, src4: &[*const u32; 4], src_log: u32, alpha: &[u32; 4], dst4: &[*mut u32; 4]) {
        let g = ctx(); check(g.0, unsafe { sys::nx_fold_line(g.0, tw.0, src4.as_ptr(), src_log, 0, alpha.as_ptr(), dst4.as_ptr()) });
    }
    pub fn fri_decompose(src4: &[*const u32; 4], log_size: u32, g4: &[*mut u32; 4]) -> [u32; 4] {
        let g = ctx(); let mut lambda = [0u32; 4];
        check(g.0, unsafe { sys::nx_fri_decompose(g.0, src4.as_ptr(), log_size, g4.as_ptr(), lambda.as_mut_ptr()) });
        lambda
    }
    /// `AccumulationOps::{accumulate, generate_secure_powers}`
    pub fn secure_accumulate(dst4: &[*mut u32; 4], src4: &[*const u32; 4], log_size: u32) { let g = ctx(); check(g.0, unsafe { sys::nx_secure_accumulate(g.0, dst4.as_ptr(), src4.as_ptr(), log_size) }); }
    pub fn generate_secure_powers(felt: &[u32; 4], n: u32) -> Vec<u32> { let mut v = vec![0u32; 4 * n as usize]; unsafe { sys::nx_generate_secure_powers(felt.as_ptr(), n, v.as_mut_ptr()) }; v }
    /// `FieldOps::batch_inverse` (M31 / QM31)
    pub fn batch_inverse_m31(src: *const u32, dst: *mut u32, n: usize) { let g = ctx(); check(g.0, unsafe { sys::nx_batch_inverse_m31(g.0, src, dst, n) }); }
    pub fn batch_inverse_qm31(src4: &[*const u32; 4], dst4: &[*mut u32; 4], n: usize) { let g = ctx(); check(g.0, unsafe { sys::nx_batch_inverse_qm31(g.0, src4.as_ptr(), dst4.as_ptr(), n) }); }
    /// `GrindOps::<Blake2sChannel>::grind`
    pub fn grind(digest: &[u8; 32], pow_bits: u32) -> u64 { let g = ctx(); let mut nonce = 0u64; check(g.0, unsafe { sys::nx_grind(g.0, digest.as_ptr(), pow_bits, &mut nonce) }); nonce }
    /// `ComponentProver::evaluate_constraint_quotients_on_domain` of a recorded component: compile once (cached by the caller), then
    /// accumulate  sum_j alpha_powers[j] C_j(row) / Z(row)  into the four coordinate columns of the domain accumulator
    pub fn air_compile(program: &[sys::nx_cinstr], n_regs: u32, n_cols: u32, n_econsts: u32, n_constraints: u32) -> *mut sys::nx_air_kernel {
        let g = ctx(); let mut k = std::ptr::null_mut();
        check(g.0, unsafe { sys::nx_air_compile(g.0, program.as_ptr(), program.len() as u32, n_regs, n_cols, n_econsts, n_constraints, &mut k, std::ptr::null_mut()) });
        k
    }
    #[allow(clippy::too_many_arguments)]
    pub fn air_eval(kernel: *const sys::nx_air_kernel, cols: &[*const u32], econsts: &[u32], alpha_powers: &[u32], denom_inv: &[u32], log_size: u32, log_eval: u32, acc4: &[*mut u32; 4]) {
        let g = ctx(); check(g.0, unsafe { sys::nx_air_eval(g.0, kernel, cols.as_ptr(), econsts.as_ptr(), alpha_powers.as_ptr(), denom_inv.as_ptr(), log_size, log_eval, acc4.as_ptr()) });
    }
}

// ------------------------------------------------------------------------------------------------ session ----
/// One recorded component: what `FrameworkComponent<E>` is to Stwo, as data (see include/nexus_hip.h `nx_air_component`);
/// produced by `record::record_component` from any `FrameworkEval`.
#[derive(Clone, Debug, Default)]
pub struct RecordedComponent {
    pub log_size: u32,
    pub program: Vec<sys::nx_cinstr>, pub n_regs: u32, pub n_constraints: u32,
    /// secure-field constants of the program, 4 words each (lookup elements, claimed-sum shifts: run-time values — the instruction
    /// stream names them by index, so one compiled kernel serves every proof of the AIR)
    pub econsts: Vec<u32>,
    pub col_tree: Vec<u32>, pub col_index: Vec<u32>,
    pub mask_count: Vec<u32>, pub mask_offsets: Vec<i32>,
    /// the reference's bound is per component: main +2 (components/mod.rs:12,44-45), extensions +1 (extensions/multiplicity.rs:108-110)
    pub log_constraint_degree_bound: u32,
    /// the component's relation entries as a fraction program (`NX_C_FRAC` / `NX_C_FRACB`): `Session::logup_trace` runs it on the device
    /// to produce the component's `n_logup_cols` secure interaction columns — the reference's `generate_interaction_trace`
    /// (traits.rs:124-145) / `LogupTraceBuilder` (prover2 lookups/logup_trace_builder.rs:22-121) without any chip-side code
    pub logup_program: Vec<sys::nx_cinstr>, pub logup_n_regs: u32, pub logup_econsts: Vec<u32>, pub n_logup_cols: u32,
}

/// machine.rs:184-296 on the device, on a context of the session's own (several sessions may prove concurrently, one per thread).
/// Order of calls = the reference's transcript: `mix_u64` (:198-206), `tree_begin` / fill / `tree_commit` per trace tree (:208-263)
/// with `draw_felts` (:239-240) and `mix_felts` (:262) in between, `prove` (:286-290).
pub struct Session { ctx: *mut sys::nx_ctx, p: *mut sys::nx_prover, comm: *mut sys::nx_comm, comm_is_local: bool, owned: Vec<*mut u32> }

/// The shared board of the in-process transport (csrc/comm_local.hip): one per proof, shared (`Arc`) by the threads that prove it — one
/// thread per GPU, each with its own `Session` and `Session::set_local_comm(&group, rank)`.  No RCCL, no second process.
pub struct LocalGroup(*mut sys::nx_comm_group, pub i32);
unsafe impl Send for LocalGroup {}
unsafe impl Sync for LocalGroup {}
impl LocalGroup {
    pub fn new(world: i32) -> Result<Self, HipError> {
        let mut g = std::ptr::null_mut();
        try_check(std::ptr::null(), unsafe { sys::nx_comm_group_create(world, &mut g) })?;
        Ok(Self(g, world))
    }
}
impl LocalGroup {
    /// re-arm a group that an abort or a timeout broke — when every rank's thread has returned from its prove (`nx_comm_group_reset`)
    pub fn reset(&self) -> Result<(), HipError> { try_check(std::ptr::null(), unsafe { sys::nx_comm_group_reset(self.0) }) }
    pub fn is_broken(&self) -> bool { unsafe { sys::nx_comm_group_broken(self.0) != 0 } }
}
impl Drop for LocalGroup { fn drop(&mut self) { unsafe { sys::nx_comm_group_destroy(self.0) } } }
impl Session {
    pub fn new(cfg: &sys::nx_pcs_config, max_log_size: u32, device: i32) -> Result<Self, HipError> {
        let mut c = std::ptr::null_mut();
        try_check(std::ptr::null(), unsafe { sys::nx_ctx_create(device, &mut c) })?;
        let mut p = std::ptr::null_mut();
        if let Err(e) = try_check(c, unsafe { sys::nx_prover_create(c, cfg, max_log_size, &mut p) }) {
            unsafe { sys::nx_ctx_destroy(c) };
            return Err(e);
        }
        Ok(Self { ctx: c, p, comm: std::ptr::null_mut(), comm_is_local: false, owned: Vec::new() })
    }
    /// one proof on the GPUs of a node: the native RCCL transport (csrc/comm_rccl.hip); `unique_id` from rank 0's `rccl_unique_id()`.
    /// The communicator belongs to the session and is destroyed with it (after the prover, which refers to it).
    pub fn set_rccl_comm(&mut self, unique_id: &[u8; 128], rank: i32, world: i32) -> Result<(), HipError> {
        if !self.comm.is_null() { return Err(HipError::Argument("the session already has a communicator".into())); }
        let mut comm = std::ptr::null_mut();
        try_check(self.ctx, unsafe { sys::nx_comm_rccl_create(self.ctx, unique_id.as_ptr(), rank, world, &mut comm) })?;
        if let Err(e) = try_check(self.ctx, unsafe { sys::nx_prover_set_comm(self.p, comm) }) {
            unsafe { sys::nx_comm_rccl_destroy(comm) };
            return Err(e);
        }
        self.comm = comm;
        Ok(())
    }
    /// one proof on the GPUs of a node from ONE process: this session is rank `rank` of `group` (the group must outlive the session)
    pub fn set_local_comm(&mut self, group: &LocalGroup, rank: i32) -> Result<(), HipError> {
        if !self.comm.is_null() { return Err(HipError::Argument("the session already has a communicator".into())); }
        let mut comm = std::ptr::null_mut();
        try_check(self.ctx, unsafe { sys::nx_comm_local_create(group.0, self.ctx, rank, &mut comm) })?;
        if let Err(e) = try_check(self.ctx, unsafe { sys::nx_prover_set_comm(self.p, comm) }) {
            unsafe { sys::nx_comm_local_destroy(comm) };
            return Err(e);
        }
        self.comm = comm; self.comm_is_local = true;
        Ok(())
    }
    pub fn mix_u64(&mut self, v: u64) { check(self.ctx, unsafe { sys::nx_prover_mix_u64(self.p, v) }); }
    pub fn mix_felts(&mut self, felts: &[u32]) { check(self.ctx, unsafe { sys::nx_prover_mix_felts(self.p, felts.as_ptr(), (felts.len() / 4) as u32) }); }
    pub fn draw_felts(&mut self, n: u32) -> Vec<u32> { let mut v = vec![0u32; 4 * n as usize]; check(self.ctx, unsafe { sys::nx_prover_draw_felts(self.p, n, v.as_mut_ptr()) }); v }
    pub fn channel_digest(&self) -> [u8; 32] { let mut d = [0u8; 32]; check(self.ctx, unsafe { sys::nx_prover_channel_digest(self.p, d.as_mut_ptr()) }); d }
    /// TreeBuilder::extend_evals: session-owned columns to fill (NULL for columns another GPU transforms)
    pub fn tree_begin(&mut self, log_sizes: &[u32]) -> Result<Vec<*mut u32>, HipError> {
        let mut ptrs = vec![std::ptr::null_mut(); log_sizes.len()];
        try_check(self.ctx, unsafe { sys::nx_prover_tree_begin(self.p, log_sizes.as_ptr(), log_sizes.len() as u32, ptrs.as_mut_ptr()) })?;
        Ok(ptrs)
    }
    /// host columns -> the session's columns.  `coset_order`: the reference's `Vec<Vec<M31>>` before `finalize_columns` (natural coset
    /// order; R3 runs on the device behind the copy); otherwise bit-reversed circle-domain evaluations as a `CircleEvaluation` holds them
    pub fn upload(&mut self, host_cols: &[*const u32], log_size: u32, dev_cols: &[*mut u32], coset_order: bool) -> Result<(), HipError> {
        try_check(self.ctx, unsafe { sys::nx_upload_columns(self.ctx, host_cols.as_ptr(), host_cols.len() as u32, log_size, dev_cols.as_ptr(), coset_order as i32) })
    }
    /// A trace slab that is reused from proof to proof: pinned once by its owner (`nx_host_pin`); `upload` / `tree_commit_host` then skip
    /// their per-call pinning of columns inside it.  Undo with `unpin_host` before the slab is dropped.
    pub fn pin_host(&mut self, slab: &[u32]) -> Result<(), HipError> {
        try_check(self.ctx, unsafe { sys::nx_host_pin(self.ctx, slab.as_ptr() as *const core::ffi::c_void, core::mem::size_of_val(slab)) })
    }
    pub fn unpin_host(&mut self, slab: &[u32]) -> Result<(), HipError> {
        try_check(self.ctx, unsafe { sys::nx_host_unpin(self.ctx, slab.as_ptr() as *const core::ffi::c_void) })
    }
    /// TreeBuilder::commit: interpolate, extend, Merkle-commit, mix the root
    pub fn tree_commit(&mut self) -> Result<[u8; 32], HipError> { let mut r = [0u8; 32]; try_check(self.ctx, unsafe { sys::nx_prover_tree_commit(self.p, r.as_mut_ptr()) })?; Ok(r) }
    /// TreeBuilder::extend_evals + commit for a tree whose columns are in HOST memory (begun with `tree_begin`): pinned in place, uploaded
    /// in chunks on a copy stream WHILE the commit transforms and hashes the chunks that have arrived.  `keep`: (column of the tree, device
    /// buffer of 2^log words) — the evaluations of these columns are cloned on arrival (the commit turns the tree's columns into coefficients).
    pub fn tree_commit_host(&mut self, host_cols: &[*const u32], coset_order: bool, keep: &[(u32, *mut u32)]) -> Result<[u8; 32], HipError> {
        let idx: Vec<u32> = keep.iter().map(|k| k.0).collect();
        let dst: Vec<*mut u32> = keep.iter().map(|k| k.1).collect();
        let mut r = [0u8; 32];
        try_check(self.ctx, unsafe { sys::nx_prover_tree_commit_host(self.p, host_cols.as_ptr(), coset_order as i32, idx.as_ptr(), idx.len() as u32, dst.as_ptr(), r.as_mut_ptr()) })?;
        Ok(r)
    }
    /// `n_cols` device columns of 2^log_size words that live as long as the session (the kept evaluations of the trace columns the
    /// logup fractions read: `tree_commit_host`'s `keep` targets)
    pub fn alloc_columns(&mut self, n_cols: usize, log_size: u32) -> Result<Vec<*mut u32>, HipError> {
        let mut base = std::ptr::null_mut();
        let words = n_cols.max(1) << log_size;
        try_check(self.ctx, unsafe { sys::nx_alloc(self.ctx, words, &mut base) })?;
        self.owned.push(base);
        Ok((0..n_cols).map(|k| unsafe { base.add(k << log_size) }).collect())
    }
    /// Releases every buffer handed out by `alloc_columns` (the kept evaluations): after the interaction tree is built nothing reads
    /// them, and `prove` needs the HBM.  The pointers dangle afterwards.
    pub fn free_columns(&mut self) {
        for p in self.owned.drain(..) { unsafe { sys::nx_free(self.ctx, p) }; }
    }
    /// The component's interaction trace from its recorded relation entries, on the device (`nx_logup_program`), finalised
    /// (`LogupTraceGenerator::finalize_last`: `nx_logup_finalize_last`).  `cols[k]`: the evaluations (bit-reversed circle-domain order)
    /// of component column k, null where the fraction program loads nothing (the interaction columns themselves); `out`: the
    /// component's 4 x n_logup_cols coordinate columns — typically the session's own, from `tree_begin` of the interaction tree, so
    /// nothing is copied.  Returns the claimed sum (machine.rs:249-262).
    pub fn logup_trace(&mut self, comp: &RecordedComponent, cols: &[*const u32], out: &[*mut u32]) -> Result<[u32; 4], HipError> {
        if comp.n_logup_cols == 0 { return Ok([0; 4]); }
        if out.len() != 4 * comp.n_logup_cols as usize || cols.len() != comp.col_tree.len() { return Err(HipError::Argument("logup_trace: column table of the wrong length".into())); }
        try_check(self.ctx, unsafe { sys::nx_logup_program(self.ctx, comp.logup_program.as_ptr(), comp.logup_program.len() as u32, comp.logup_n_regs, cols.as_ptr(), cols.len() as u32,
                                                           comp.logup_econsts.as_ptr(), (comp.logup_econsts.len() / 4) as u32, comp.log_size, comp.n_logup_cols, out.as_ptr(), std::ptr::null_mut()) })?;
        let mut claimed = [0u32; 4];
        try_check(self.ctx, unsafe { sys::nx_logup_finalize_last(self.ctx, comp.log_size, out[out.len() - 4..].as_ptr(), claimed.as_mut_ptr()) })?;
        Ok(claimed)
    }
    /// Compiled AIR / fraction kernels are kept in `dir` across processes (`nx_air_cache_dir`): the first proof of a process loads
    /// them in milliseconds instead of paying hiprtc (seconds for an AIR of the reference's size).  Process-wide.
    pub fn kernel_cache_dir(dir: &str) -> Result<(), HipError> {
        let c = std::ffi::CString::new(dir).map_err(|_| HipError::Argument("path holds a NUL byte".into()))?;
        try_check(std::ptr::null(), unsafe { sys::nx_air_cache_dir(c.as_ptr()) })
    }
    /// stwo::prover::prove (machine.rs:286-290): NXP1 proof words
    pub fn prove(&mut self, comps: &[RecordedComponent]) -> Result<Vec<u32>, HipError> {
        let raw: Vec<sys::nx_air_component> = comps.iter().map(|c| sys::nx_air_component {
            log_size: c.log_size, program: c.program.as_ptr(), n_instr: c.program.len() as u32, n_regs: c.n_regs,
            econsts: c.econsts.as_ptr(), n_econsts: (c.econsts.len() / 4) as u32, n_constraints: c.n_constraints,
            col_tree: c.col_tree.as_ptr(), col_index: c.col_index.as_ptr(), n_cols: c.col_tree.len() as u32,
            mask_count: c.mask_count.as_ptr(), mask_offsets: c.mask_offsets.as_ptr(), kernel: std::ptr::null(),
            log_constraint_degree_bound: c.log_constraint_degree_bound,
        }).collect();
        let (mut words, mut n) = (std::ptr::null_mut(), 0usize);
        try_check(self.ctx, unsafe { sys::nx_prover_prove(self.p, raw.as_ptr(), raw.len() as u32, &mut words, &mut n, std::ptr::null_mut()) })?;
        let v = unsafe { std::slice::from_raw_parts(words, n) }.to_vec();
        unsafe { sys::nx_free_host(words as *mut std::ffi::c_void) };
        Ok(v)
    }
    pub fn ctx(&self) -> *mut sys::nx_ctx { self.ctx }
    /// Per-context policy (include/nexus_hip.h `nx_ctx_set_option`), e.g. `("air.degree_split", 0)` to evaluate every constraint on the
    /// component's full domain like Stwo does.  No option changes a proof byte.  An unknown name or a value out of range is `Err(Argument)`.
    pub fn set_option(&mut self, name: &str, value: i64) -> Result<(), HipError> {
        let c = std::ffi::CString::new(name).map_err(|_| HipError::Argument("option name holds a NUL byte".into()))?;
        try_check(self.ctx, unsafe { sys::nx_ctx_set_option(self.ctx, c.as_ptr(), value) })
    }
}
impl RecordedComponent {
    /// An upper bound of every constraint's degree in the trace columns: the smallest sound `log_constraint_degree_bound` is the e with
    /// max degree <= 2^e + 1 (what a `FrameworkEval::max_constraint_log_degree_bound` must return, components/mod.rs:44-45).
    pub fn constraint_degrees(&self) -> Vec<u32> {
        let mut d = vec![0u32; self.n_constraints as usize];
        let rc = unsafe { sys::nx_air_constraint_degrees(std::ptr::null_mut(), self.program.as_ptr(), self.program.len() as u32, self.n_regs, self.col_tree.len() as u32,
                                                         (self.econsts.len() / 4) as u32, self.n_constraints, d.as_mut_ptr()) };
        assert_eq!(rc, sys::NX_OK, "{}", last_error(std::ptr::null()));
        d
    }
}
impl Drop for Session {
    fn drop(&mut self) {
        unsafe {
            sys::nx_prover_destroy(self.p);
            for p in self.owned.drain(..) { sys::nx_free(self.ctx, p); }
            if !self.comm.is_null() { if self.comm_is_local { sys::nx_comm_local_destroy(self.comm) } else { sys::nx_comm_rccl_destroy(self.comm) } }
            sys::nx_ctx_destroy(self.ctx);
        }
    }
}

pub fn rccl_unique_id() -> Result<[u8; 128], HipError> { let mut id = [0u8; 128]; try_check(std::ptr::null(), unsafe { sys::nx_rccl_unique_id(id.as_mut_ptr()) })?; Ok(id) }

/// The postcard bytes of the reference's `Proof { stark_proof, claimed_sum, log_size }` (machine.rs:93-98) from NXP1 words —
/// EXPERIMENTAL: the field order is upstream-recollection until tools/dump_reference.rs has run against the pinned Stwo.
pub fn proof_bytes(words: &[u32], claimed_sums: &[u32], log_sizes: &[u32]) -> Result<Vec<u8>, HipError> {
    let (mut b, mut n) = (std::ptr::null_mut(), 0usize);
    try_check(std::ptr::null(), unsafe { sys::nx_proof_serialize_stwo(words.as_ptr(), words.len(), claimed_sums.as_ptr(), log_sizes.as_ptr(), log_sizes.len() as u32, &mut b, &mut n) })?;
    let v = unsafe { std::slice::from_raw_parts(b, n) }.to_vec();
    unsafe { sys::nx_free_host(b as *mut std::ffi::c_void) };
    Ok(v)
}

/// The swap at reference core/src/lib.rs:22-24: `pub use nexus_vm_prover::{prove, ...}` stays; inside `nexus_vm_prover`,
/// `prove` (prover/src/lib.rs:26-31) calls `Machine::<BaseComponent>::prove_hip` under the `hip` feature.  The body lives in the
/// reference tree because it uses crate-private items (`TracesBuilder`, `SideNote`, `MachineEval::new`, `generate_interaction_trace`):
/// `rust/nexus-hip/reference_patch/machine_hip.rs`, a drop-in `prover/src/machine_hip.rs` that follows machine.rs:130-297 line by
/// line with `Session` in place of `CommitmentSchemeProver<SimdBackend, _>` and `record::record_component` in place of
/// `FrameworkComponent::new`.
pub const SWAP_POINT: &str = "core/src/lib.rs:22-24";
