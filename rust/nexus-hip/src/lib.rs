//! `HipBackend` — Stwo's backend traits over `libnexus_hip.so` (MI355X / gfx950), and the whole-prove session route.
//!
//! NOT COMPILED in the repository's build image (no Rust toolchain there): written against Stwo @ 0790eba as the reference uses it
//! (reference `prover/src/machine.rs:4-19,184-290`); signatures marked [upstream-recollection] must be checked against the pinned
//! crate on first compile.  What IS checked without a compiler (tests/test_rust_shim_cpu.py): every `sys::nx_*` call below names a
//! function `nexus-hip-sys` declares, with the declared number of arguments.
//!
//! Two routes, both keeping `nexus_vm_prover::prove(&impl Trace, &View) -> Result<Proof, ProvingError>` unchanged:
//!  1. per-operation: `HipBackend` implements the traits `stwo::prover::prove` is generic over; the reference swaps the type
//!     argument `SimdBackend` -> `HipBackend` (machine.rs:16,186,203,271,283,286);
//!  2. session (`prove_on_device`): after trace generation the whole of machine.rs:184-296 runs on the device — commits,
//!     logup interaction trace, recorded AIR, OODS, DEEP quotients, FRI, PoW, decommitment — and only the proof comes back.
#![allow(clippy::missing_safety_doc)]

use nexus_hip_sys as sys;
use std::ffi::CStr;
use std::marker::PhantomData;
use std::sync::{Mutex, MutexGuard, OnceLock};

// ------------------------------------------------------------------------------------------------ context ----
/// Stwo's `Backend` is a zero-sized type whose methods are called from whatever (rayon) thread Stwo likes, while an `nx_ctx` is
/// single-threaded at the protocol level.  One process-wide context behind a mutex: every backend op takes the lock for the
/// duration of its (asynchronous, stream-ordered) enqueue — ops of one prove are sequential anyway (one `&mut Blake2sChannel`,
/// reference machine.rs:197), and a second concurrent prove uses `prove_on_device` with its own context instead.
struct Ctx(*mut sys::nx_ctx);
unsafe impl Send for Ctx {}
static CTX: OnceLock<Mutex<Ctx>> = OnceLock::new();

fn ctx() -> MutexGuard<'static, Ctx> {
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

fn last_error(c: *const sys::nx_ctx) -> String {
    unsafe { CStr::from_ptr(sys::nx_last_error(c)) }.to_string_lossy().into_owned()
}

/// Allocation / HIP failures panic (the reference's `vec![..]` aborts too, trace_builder.rs:29); a protocol error is the
/// reference's `ProvingError::ConstraintsNotSatisfied` (core/src/lib.rs:22-24) and is returned.
fn check(c: *const sys::nx_ctx, rc: i32) -> Result<(), ProvingErrorKind> {
    match rc {
        sys::NX_OK => Ok(()),
        sys::NX_ERR_PROTOCOL => Err(ProvingErrorKind::ConstraintsNotSatisfied),
        _ => panic!("libnexus_hip error {rc}: {}", last_error(c)),
    }
}
#[derive(Debug)]
pub enum ProvingErrorKind { ConstraintsNotSatisfied }

// ------------------------------------------------------------------------------------------------ columns ----
/// A device column: `len` u32 words from `nx_alloc`; `Drop` returns them (Stwo moves evaluations into the tree builder by value —
/// reference machine.rs:209-215 — so ownership maps onto a handle with an explicit free).
pub struct HipColumn<T> { ptr: *mut u32, len: usize, _t: PhantomData<T> }
unsafe impl<T> Send for HipColumn<T> {}
unsafe impl<T> Sync for HipColumn<T> {}

impl<T> HipColumn<T> {
    pub fn zeros_words(len: usize) -> Self {
        let g = ctx();
        let mut p = std::ptr::null_mut();
        check(g.0, unsafe { sys::nx_alloc(g.0, len.max(1), &mut p) }).unwrap();
        check(g.0, unsafe { sys::nx_memset_zero(g.0, p, len) }).unwrap();
        Self { ptr: p, len, _t: PhantomData }
    }
    pub fn from_host_words(words: &[u32]) -> Self {
        let g = ctx();
        let mut p = std::ptr::null_mut();
        check(g.0, unsafe { sys::nx_alloc(g.0, words.len().max(1), &mut p) }).unwrap();
        check(g.0, unsafe { sys::nx_upload(g.0, p, words.as_ptr(), words.len()) }).unwrap();
        Self { ptr: p, len: words.len(), _t: PhantomData }
    }
    pub fn to_host_words(&self) -> Vec<u32> {
        let g = ctx();
        let mut v = vec![0u32; self.len];
        check(g.0, unsafe { sys::nx_download(g.0, v.as_mut_ptr(), self.ptr, self.len) }).unwrap();
        v
    }
    pub fn word_at(&self, i: usize) -> u32 {
        let g = ctx();
        let (p, idx, mut out) = (self.ptr as *const u32, i as u64, 0u32);
        check(g.0, unsafe { sys::nx_gather(g.0, &p, &idx, 1, &mut out) }).unwrap();
        out
    }
    pub fn clone_on_device(&self) -> Self {
        let g = ctx();
        let mut p = std::ptr::null_mut();
        check(g.0, unsafe { sys::nx_alloc(g.0, self.len.max(1), &mut p) }).unwrap();
        check(g.0, unsafe { sys::nx_copy(g.0, p, self.ptr, self.len) }).unwrap();
        Self { ptr: p, len: self.len, _t: PhantomData }
    }
    pub fn as_ptr(&self) -> *const u32 { self.ptr }
    pub fn as_mut_ptr(&mut self) -> *mut u32 { self.ptr }
    pub fn len(&self) -> usize { self.len }
    pub fn is_empty(&self) -> bool { self.len == 0 }
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

pub struct HipTwiddles(pub *mut sys::nx_twiddles);
unsafe impl Send for HipTwiddles {}
unsafe impl Sync for HipTwiddles {}
impl Drop for HipTwiddles { fn drop(&mut self) { unsafe { sys::nx_twiddles_destroy(self.0) } } }

/// The operations behind the Stwo trait methods, on raw device pointers: each `impl <Trait> for HipBackend` below (gated behind
/// the `stwo-traits` cfg until it has been compiled against the pinned crate) is a thin adapter over one of these.
pub mod ops {
    use super::*;

    /// `ColumnOps<BaseField>::bit_reverse_column` (reference prover/src/trace/utils.rs:101)
    pub fn bit_reverse(col: *mut u32, log_size: u32) { let g = ctx(); check(g.0, unsafe { sys::nx_bit_reverse(g.0, col, log_size) }).unwrap(); }
    /// `ColumnOps<SecureField>::bit_reverse_column` on the 4 coordinate columns
    pub fn bit_reverse_secure(col4: &[*mut u32; 4], log_size: u32) { let g = ctx(); check(g.0, unsafe { sys::nx_bit_reverse_secure(g.0, col4.as_ptr(), log_size) }).unwrap(); }
    /// `PolyOps::precompute_twiddles(CanonicCoset::new(log + 1).half_coset())` (reference machine.rs:186-194)
    pub fn precompute_twiddles(log_half_coset: u32) -> HipTwiddles {
        let g = ctx(); let mut t = std::ptr::null_mut();
        check(g.0, unsafe { sys::nx_twiddles_create(g.0, log_half_coset, &mut t) }).unwrap();
        HipTwiddles(t)
    }
    /// `PolyOps::interpolate_columns` for one group of equally sized columns (in place: evaluations -> coefficients)
    pub fn interpolate(tw: &HipTwiddles, cols: &[*mut u32], log_size: u32) { let g = ctx(); check(g.0, unsafe { sys::nx_interpolate_batch(g.0, tw.0, cols.as_ptr(), cols.len() as u32, log_size) }).unwrap(); }
    /// `PolyOps::evaluate_polynomials` for one group
    pub fn evaluate(tw: &HipTwiddles, polys: &[*const u32], log_size: u32, log_expand: u32, out: &[*mut u32]) {
        let g = ctx(); check(g.0, unsafe { sys::nx_evaluate_batch(g.0, tw.0, polys.as_ptr(), polys.len() as u32, log_size, log_expand, out.as_ptr()) }).unwrap();
    }
    /// `TreeBuilder::extend_evals` + `commit` fused for one group: evaluations -> coefficients (in place) + LDE
    pub fn lde(tw: &HipTwiddles, cols: &[*mut u32], log_size: u32, log_blowup: u32, lde: &[*mut u32]) {
        let g = ctx(); check(g.0, unsafe { sys::nx_lde_batch(g.0, tw.0, cols.as_ptr(), cols.len() as u32, log_size, log_blowup, lde.as_ptr()) }).unwrap();
    }
    /// `PolyOps::eval_at_point`, batched: out[i] = polys[poly_idx[i]](points[i]); points are 8 words (x, y), results 4 words
    pub fn eval_at_points(polys: &[*const u32], log_size: u32, poly_idx: &[u32], points: &[u32], out: &mut [u32]) {
        let g = ctx(); check(g.0, unsafe { sys::nx_eval_at_points(g.0, polys.as_ptr(), log_size, poly_idx.as_ptr(), points.as_ptr(), poly_idx.len() as u32, out.as_mut_ptr()) }).unwrap();
    }
    /// `MerkleOps::<Blake2sMerkleHasher>::commit_on_layer` — one layer, the trait's own shape
    pub fn commit_on_layer(log_size: u32, prev_layer: *const u32, cols: &[*const u32], out: *mut u32) {
        let g = ctx(); check(g.0, unsafe { sys::nx_merkle_commit_on_layer(g.0, log_size, prev_layer, cols.as_ptr(), cols.len() as u32, out) }).unwrap();
    }
    /// `MerkleProver::commit` as one call (whole tree), and its root
    pub fn merkle_commit(cols: &[*const u32], log_sizes: &[u32]) -> (*mut sys::nx_tree, [u8; 32]) {
        let g = ctx(); let mut t = std::ptr::null_mut(); let mut root = [0u8; 32];
        check(g.0, unsafe { sys::nx_merkle_commit(g.0, cols.as_ptr(), log_sizes.as_ptr(), cols.len() as u32, &mut t) }).unwrap();
        check(g.0, unsafe { sys::nx_merkle_root(g.0, t, root.as_mut_ptr()) }).unwrap();
        (t, root)
    }
    /// `QuotientOps::accumulate_quotients`
    #[allow(clippy::too_many_arguments)]
    pub fn accumulate_quotients(log_size: u32, cols: &[*const u32], random_coeff: &[u32; 4], points: &[u32], batch_counts: &[u32], col_idx: &[u32], values: &[u32], out4: &[*mut u32; 4]) {
        let g = ctx();
        check(g.0, unsafe { sys::nx_accumulate_quotients(g.0, log_size, cols.as_ptr(), cols.len() as u32, random_coeff.as_ptr(), batch_counts.len() as u32, points.as_ptr(), batch_counts.as_ptr(), col_idx.as_ptr(), values.as_ptr(), out4.as_ptr()) }).unwrap();
    }
    /// `FriOps::fold_circle_into_line` / `fold_line` / `decompose`
    pub fn fold_circle_into_line(tw: &HipTwiddles, dst4: &[*mut u32; 4], src4: &[*const u32; 4], src_log: u32, alpha: &[u32; 4]) {
        let g = ctx(); check(g.0, unsafe { sys::nx_fold_circle_into_line(g.0, tw.0, dst4.as_ptr(), src4.as_ptr(), src_log, alpha.as_ptr()) }).unwrap();
    }
    /// `n_doublings`: how often the line domain of the twiddle tree's root half coset has been doubled to reach `src_log`
    pub fn fold_line(tw: &HipTwiddles, src4: &[*const u32; 4], src_log: u32, n_doublings: u32, alpha: &[u32; 4], dst4: &[*mut u32; 4]) {
        let g = ctx(); check(g.0, unsafe { sys::nx_fold_line(g.0, tw.0, src4.as_ptr(), src_log, n_doublings, alpha.as_ptr(), dst4.as_ptr()) }).unwrap();
    }
    pub fn fri_decompose(src4: &[*const u32; 4], log_size: u32, g4: &[*mut u32; 4]) -> [u32; 4] {
        let g = ctx(); let mut lambda = [0u32; 4];
        check(g.0, unsafe { sys::nx_fri_decompose(g.0, src4.as_ptr(), log_size, g4.as_ptr(), lambda.as_mut_ptr()) }).unwrap();
        lambda
    }
    /// `AccumulationOps::{accumulate, generate_secure_powers}`
    pub fn secure_accumulate(dst4: &[*mut u32; 4], src4: &[*const u32; 4], log_size: u32) { let g = ctx(); check(g.0, unsafe { sys::nx_secure_accumulate(g.0, dst4.as_ptr(), src4.as_ptr(), log_size) }).unwrap(); }
    pub fn generate_secure_powers(felt: &[u32; 4], n: u32) -> Vec<u32> { let mut v = vec![0u32; 4 * n as usize]; unsafe { sys::nx_generate_secure_powers(felt.as_ptr(), n, v.as_mut_ptr()) }; v }
    /// `FieldOps::batch_inverse` (M31 / QM31)
    pub fn batch_inverse_m31(src: *const u32, dst: *mut u32, n: usize) { let g = ctx(); check(g.0, unsafe { sys::nx_batch_inverse_m31(g.0, src, dst, n) }).unwrap(); }
    pub fn batch_inverse_qm31(src4: &[*const u32; 4], dst4: &[*mut u32; 4], n: usize) { let g = ctx(); check(g.0, unsafe { sys::nx_batch_inverse_qm31(g.0, src4.as_ptr(), dst4.as_ptr(), n) }).unwrap(); }
    /// `GrindOps::<Blake2sChannel>::grind`
    pub fn grind(digest: &[u8; 32], pow_bits: u32) -> u64 { let g = ctx(); let mut nonce = 0u64; check(g.0, unsafe { sys::nx_grind(g.0, digest.as_ptr(), pow_bits, &mut nonce) }).unwrap(); nonce }
    /// R3 fused with the upload: the host trace (`Vec<Vec<M31>>`, natural coset order) -> device columns in bit-reversed
    /// circle-domain order (reference prover/src/trace/utils.rs:94-106 + utils_external.rs:24-39), no CPU pass
    pub fn upload_trace(host_cols: &[*const u32], log_size: u32, dev_cols: &[*mut u32]) {
        let g = ctx(); check(g.0, unsafe { sys::nx_upload_columns(g.0, host_cols.as_ptr(), host_cols.len() as u32, log_size, dev_cols.as_ptr(), 1) }).unwrap();
    }
}

/// The trait impls proper.  Behind a cfg until they have met the pinned Stwo's exact signatures on a box with cargo:
/// `RUSTFLAGS="--cfg stwo_traits" cargo build -p nexus-hip`.
#[cfg(stwo_traits)]
mod stwo_impls {
    use super::*;
    use stwo::core::fields::m31::BaseField;
    use stwo::core::fields::qm31::SecureField;
    use stwo::prover::backend::{Backend, BackendForChannel, Column, ColumnOps};
    use stwo::core::vcs::blake2_merkle::{Blake2sMerkleChannel, Blake2sMerkleHasher};

    impl Backend for HipBackend {}
    impl BackendForChannel<Blake2sMerkleChannel> for HipBackend {}

    impl ColumnOps<BaseField> for HipBackend {
        type Column = HipColumn<BaseField>;
        fn bit_reverse_column(col: &mut Self::Column) { ops::bit_reverse(col.as_mut_ptr(), col.len().ilog2()); }
    }
    impl Column<BaseField> for HipColumn<BaseField> {
        fn zeros(len: usize) -> Self { HipColumn::zeros_words(len) }
        unsafe fn uninitialized(len: usize) -> Self { HipColumn::zeros_words(len) }
        fn to_cpu(&self) -> Vec<BaseField> { self.to_host_words().into_iter().map(BaseField::from_u32_unchecked).collect() }
        fn len(&self) -> usize { self.len }
        fn at(&self, index: usize) -> BaseField { BaseField::from_u32_unchecked(self.word_at(index)) }
        fn set(&mut self, index: usize, value: BaseField) {
            let g = ctx(); let w = value.0;
            check(g.0, unsafe { sys::nx_upload(g.0, self.ptr.add(index), &w, 1) }).unwrap();
        }
    }
    impl FromIterator<BaseField> for HipColumn<BaseField> {
        fn from_iter<I: IntoIterator<Item = BaseField>>(it: I) -> Self { let v: Vec<u32> = it.into_iter().map(|x| x.0).collect(); HipColumn::from_host_words(&v) }
    }
    // PolyOps, MerkleOps<Blake2sMerkleHasher>, QuotientOps, FriOps, AccumulationOps, FieldOps<BaseField>, FieldOps<SecureField>,
    // ColumnOps<SecureField>, GrindOps<Blake2sChannel>: one adapter each over `ops::*` (INTEGRATION.md §2 lists method -> export);
    // GkrOps: `unimplemented!()` — the reference has no GKR lookups (`grep -ri gkr` over the reference: empty).
    #[allow(dead_code)] fn _types(_: SecureField, _: Blake2sMerkleHasher) {}
}

// ------------------------------------------------------------------------------------------------ session ----
/// One recorded component: what `FrameworkComponent<E>` is to Stwo, as data (see include/nexus_hip.h `nx_air_component`).
pub struct RecordedComponent {
    pub log_size: u32,
    pub program: Vec<sys::nx_cinstr>, pub n_regs: u32, pub n_constraints: u32,
    pub econsts: Vec<u32>,
    pub col_tree: Vec<u32>, pub col_index: Vec<u32>,
    pub mask_count: Vec<u32>, pub mask_offsets: Vec<i32>,
    /// the reference's bound is per component: main +2 (components/mod.rs:12,44-45), extensions +1 (extensions/multiplicity.rs:108-110)
    pub log_constraint_degree_bound: u32,
}

/// machine.rs:184-296 on the device.  `fill_tree(tree, device column pointers)` writes the tree's columns (bit-reversed
/// circle-domain evaluations) into session-owned memory — `ops::upload_trace` for a host trace, kernels for a device one;
/// `interaction(z_alpha) -> claimed sums` is called between the main and the interaction tree with the drawn lookup elements.
pub struct Session { ctx: *mut sys::nx_ctx, p: *mut sys::nx_prover }
impl Session {
    pub fn new(cfg: &sys::nx_pcs_config, max_log_size: u32, device: i32) -> Self {
        let mut c = std::ptr::null_mut();
        let rc = unsafe { sys::nx_ctx_create(device, &mut c) };
        if rc != sys::NX_OK { panic!("nx_ctx_create failed ({rc}): {}", last_error(std::ptr::null())); }
        let mut p = std::ptr::null_mut();
        check(c, unsafe { sys::nx_prover_create(c, cfg, max_log_size, &mut p) }).unwrap();
        Self { ctx: c, p }
    }
    /// one proof on the GPUs of a node: the native RCCL transport (csrc/comm_rccl.hip); `unique_id` from rank 0's `rccl_unique_id()`
    pub fn set_rccl_comm(&mut self, unique_id: &[u8; 128], rank: i32, world: i32) -> *mut sys::nx_comm {
        let mut comm = std::ptr::null_mut();
        check(self.ctx, unsafe { sys::nx_comm_rccl_create(self.ctx, unique_id.as_ptr(), rank, world, &mut comm) }).unwrap();
        check(self.ctx, unsafe { sys::nx_prover_set_comm(self.p, comm) }).unwrap();
        comm
    }
    pub fn mix_u64(&mut self, v: u64) { check(self.ctx, unsafe { sys::nx_prover_mix_u64(self.p, v) }).unwrap(); }
    pub fn mix_felts(&mut self, felts: &[u32]) { check(self.ctx, unsafe { sys::nx_prover_mix_felts(self.p, felts.as_ptr(), (felts.len() / 4) as u32) }).unwrap(); }
    pub fn draw_felts(&mut self, n: u32) -> Vec<u32> { let mut v = vec![0u32; 4 * n as usize]; check(self.ctx, unsafe { sys::nx_prover_draw_felts(self.p, n, v.as_mut_ptr()) }).unwrap(); v }
    /// TreeBuilder::extend_evals: session-owned columns to fill (NULL for columns another GPU transforms)
    pub fn tree_begin(&mut self, log_sizes: &[u32]) -> Vec<*mut u32> {
        let mut ptrs = vec![std::ptr::null_mut(); log_sizes.len()];
        check(self.ctx, unsafe { sys::nx_prover_tree_begin(self.p, log_sizes.as_ptr(), log_sizes.len() as u32, ptrs.as_mut_ptr()) }).unwrap();
        ptrs
    }
    /// TreeBuilder::commit: interpolate, extend, Merkle-commit, mix the root
    pub fn tree_commit(&mut self) -> [u8; 32] { let mut r = [0u8; 32]; check(self.ctx, unsafe { sys::nx_prover_tree_commit(self.p, r.as_mut_ptr()) }).unwrap(); r }
    /// stwo::prover::prove (machine.rs:286-290): NXP1 proof words
    pub fn prove(&mut self, comps: &[RecordedComponent]) -> Result<Vec<u32>, ProvingErrorKind> {
        let raw: Vec<sys::nx_air_component> = comps.iter().map(|c| sys::nx_air_component {
            log_size: c.log_size, program: c.program.as_ptr(), n_instr: c.program.len() as u32, n_regs: c.n_regs,
            econsts: c.econsts.as_ptr(), n_econsts: (c.econsts.len() / 4) as u32, n_constraints: c.n_constraints,
            col_tree: c.col_tree.as_ptr(), col_index: c.col_index.as_ptr(), n_cols: c.col_tree.len() as u32,
            mask_count: c.mask_count.as_ptr(), mask_offsets: c.mask_offsets.as_ptr(), kernel: std::ptr::null(),
            log_constraint_degree_bound: c.log_constraint_degree_bound,
        }).collect();
        let (mut words, mut n) = (std::ptr::null_mut(), 0usize);
        check(self.ctx, unsafe { sys::nx_prover_prove(self.p, raw.as_ptr(), raw.len() as u32, &mut words, &mut n, std::ptr::null_mut()) })?;
        let v = unsafe { std::slice::from_raw_parts(words, n) }.to_vec();
        unsafe { sys::nx_free_host(words as *mut std::ffi::c_void) };
        Ok(v)
    }
    pub fn ctx(&self) -> *mut sys::nx_ctx { self.ctx }
    /// Per-context policy (include/nexus_hip.h `nx_ctx_set_option`), e.g. `("air.degree_split", 0)` to evaluate every constraint on the
    /// component's full domain like Stwo does.  No option changes a proof byte.
    pub fn set_option(&mut self, name: &str, value: i64) -> Result<(), ProvingErrorKind> {
        let c = std::ffi::CString::new(name).expect("option name");
        check(self.ctx, unsafe { sys::nx_ctx_set_option(self.ctx, c.as_ptr(), value) })
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
impl Drop for Session { fn drop(&mut self) { unsafe { sys::nx_prover_destroy(self.p); sys::nx_ctx_destroy(self.ctx); } } }

pub fn rccl_unique_id() -> [u8; 128] { let mut id = [0u8; 128]; let rc = unsafe { sys::nx_rccl_unique_id(id.as_mut_ptr()) }; assert_eq!(rc, sys::NX_OK, "{}", last_error(std::ptr::null())); id }

/// The postcard bytes of the reference's `Proof { stark_proof, claimed_sum, log_size }` (machine.rs:93-98) from NXP1 words —
/// EXPERIMENTAL: the field order is upstream-recollection until tools/dump_reference.rs has run against the pinned Stwo.
pub fn proof_bytes(words: &[u32], claimed_sums: &[u32], log_sizes: &[u32]) -> Vec<u8> {
    let (mut b, mut n) = (std::ptr::null_mut(), 0usize);
    let rc = unsafe { sys::nx_proof_serialize_stwo(words.as_ptr(), words.len(), claimed_sums.as_ptr(), log_sizes.as_ptr(), log_sizes.len() as u32, &mut b, &mut n) };
    assert_eq!(rc, sys::NX_OK, "{}", last_error(std::ptr::null()));
    let v = unsafe { std::slice::from_raw_parts(b, n) }.to_vec();
    unsafe { sys::nx_free_host(b as *mut std::ffi::c_void) };
    v
}

/// The swap at reference core/src/lib.rs:22-24: `pub use nexus_vm_prover::{prove, ...}` becomes, under the `hip` feature,
/// `pub use nexus_hip::prove;` with the signature `fn prove(trace: &impl Trace, view: &View) -> Result<Proof, ProvingError>`.
/// The body (in the reference tree, where `Trace`, `View`, `BaseComponent` and the chips live): trace generation as today
/// (machine.rs:135-183), then `Session` — `mix_u64` per program byte and component log size (:198-206), `tree_begin` +
/// `ops::upload_trace` + `tree_commit` for the preprocessed and main trees (:208-237), `draw_felts` for the lookup elements
/// (:239-240), `sys::nx_logup_cols` + `sys::nx_logup_finalize_last` into the interaction tree's columns and `mix_felts` of the claimed
/// sums (:249-263), `Session::prove` over the components recorded once through a recording `EvalAtRow`, and
/// `proof_bytes` -> `Proof` (postcard).
pub const SWAP_POINT: &str = "core/src/lib.rs:22-24";
