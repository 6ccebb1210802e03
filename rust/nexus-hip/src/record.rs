//! The recording `EvalAtRow`: the reference's AIR closures cross the C ABI as straight-line programs.
//!
//! `FrameworkEval::evaluate<E: EvalAtRow>(&self, eval: E) -> E` (reference prover/src/components/mod.rs:39-57,
//! prover2/machine/src/framework/eval.rs:19-33) is generic Rust: it cannot be called from a kernel.  But it is generic over the FIELD
//! TYPE too, and Stwo already runs it over types that do not compute — `InfoEvaluator` (mask discovery, components/mod.rs:59-67).
//! `RecordingEval` is such an evaluator: its `F` / `EF` are expression handles, every operator builds a DAG node, `add_constraint`
//! notes a root.  `finish()` then does what nexus-zkvm_amd/air_program.py::ProgramBuilder::build does on the Python side (the two are
//! kept rule for rule the same, so the CPU parity tests of the Python recorder speak for this lowering): share common subexpressions,
//! emit every constraint's not-yet-emitted nodes in post-order with the column loads of each group of 8 constraints hoisted in front
//! of the group, allocate registers by linear scan (base values one register, secure values four consecutive ones), and write
//! `nx_cinstr` words.  The result is a `RecordedComponent`: program + the component's columns in the three trace trees + the mask
//! offsets each column is sampled at — what `FrameworkComponent<E>` is to Stwo, as data.
//!
//! The relation entries are recorded too: `add_to_relation(RelationEntry::new(relation, multiplicity, &values))` (components/mod.rs:48-56,
//! extensions/keccak/round/constraints.rs:95-116, every prover2 component) leaves the pair (multiplicity, relation.combine(values)) —
//! two more DAG roots — and `finish()` lowers them to the FRACTION program (`NX_C_FRAC` / `NX_C_FRACB`, include/nexus_hip.h) that
//! `nx_logup_program` turns into the component's interaction trace on the device: the reference's hand-written generators
//! (traits.rs:124-145 -> every chip's `fill_interaction_trace`; prover2 `LogupTraceBuilder`) compute exactly these fractions, so none
//! of them has to be touched or ported (air_program.py `add_to_relation` / `build_logup` is the Python twin; tests/test_logup_cpu.py).
//!
//! NOT COMPILED here (no Rust toolchain in the build image).  The trait surface implemented below — `EvalAtRow`'s associated-type
//! bounds, `logup_proxy!`, `LogupAtRow::new` — is [upstream-recollection] of stwo-constraint-framework @ 0790eba; the reference shows
//! its USE (`eval.next_interaction_mask(ORIGINAL_TRACE_IDX, [0, 1])`, `eval.get_preprocessed_column(PreProcessedColumnId { id })`:
//! prover/src/trace/eval.rs:22-50; `eval.add_constraint`, `eval.add_to_relation`, `eval.finalize_logup()`: components/mod.rs:48-56).
use crate::RecordedComponent;
use nexus_hip_sys as sys;
use std::collections::HashMap;
use std::ops::{Add, AddAssign, Mul, MulAssign, Neg, Sub};
use std::rc::Rc;

pub const P: u32 = (1 << 31) - 1;

// ------------------------------------------------------------------------------------------------ expression handles ----
#[derive(Debug)]
pub enum BNode {
    /// component column `col` (index into the component's column table) at row offset `off`
    Col { col: u32, off: i32 },
    Const(u32),
    Add(F, F),
    Sub(F, F),
    Mul(F, F),
    Neg(F),
}
/// `EvalAtRow::F` of the recorder: a base-field expression
#[derive(Clone, Debug)]
pub struct F(pub Rc<BNode>);

#[derive(Debug)]
pub enum ENode {
    /// `combine_ef`: v0 + i v1 + u v2 + iu v3
    Combine([F; 4]),
    Const([u32; 4]),
    FromBase(F),
    Add(EF, EF),
    Sub(EF, EF),
    Mul(EF, EF),
    AddB(EF, F),
    MulB(EF, F),
}
/// `EvalAtRow::EF` of the recorder: a secure-field expression
#[derive(Clone, Debug)]
pub struct EF(pub Rc<ENode>);

fn b(n: BNode) -> F { F(Rc::new(n)) }
fn e(n: ENode) -> EF { EF(Rc::new(n)) }
fn bconst(v: u32) -> F { b(BNode::Const(v % P)) }
fn econst(v: [u32; 4]) -> EF { e(ENode::Const(v)) }
fn is_bconst(x: &F, v: u32) -> bool { matches!(*x.0, BNode::Const(c) if c == v) }

impl Add<F> for F { type Output = F; fn add(self, o: F) -> F { b(BNode::Add(self, o)) } }
impl Sub<F> for F { type Output = F; fn sub(self, o: F) -> F { b(BNode::Sub(self, o)) } }
impl Mul<F> for F { type Output = F; fn mul(self, o: F) -> F { b(BNode::Mul(self, o)) } }
impl Neg for F { type Output = F; fn neg(self) -> F { b(BNode::Neg(self)) } }
impl AddAssign<F> for F { fn add_assign(&mut self, o: F) { *self = self.clone() + o; } }
impl MulAssign<F> for F { fn mul_assign(&mut self, o: F) { *self = self.clone() * o; } }

impl Add<EF> for EF { type Output = EF; fn add(self, o: EF) -> EF { e(ENode::Add(self, o)) } }
impl Sub<EF> for EF { type Output = EF; fn sub(self, o: EF) -> EF { e(ENode::Sub(self, o)) } }
impl Mul<EF> for EF { type Output = EF; fn mul(self, o: EF) -> EF { e(ENode::Mul(self, o)) } }
impl Neg for EF { type Output = EF; fn neg(self) -> EF { e(ENode::Sub(econst([0; 4]), self)) } }     // air_program.py _neg_e: 0 - x
impl Add<F> for EF { type Output = EF; fn add(self, o: F) -> EF { e(ENode::AddB(self, o)) } }
impl Mul<F> for EF { type Output = EF; fn mul(self, o: F) -> EF { e(ENode::MulB(self, o)) } }
impl From<F> for EF { fn from(x: F) -> EF { e(ENode::FromBase(x)) } }

/// The impls that name Stwo's field types: behind the same cfg as the backend traits (the recorder's lowering above and below them is
/// plain Rust and is what the structural tests read).
#[cfg(stwo_traits)]
mod stwo_glue {
    use super::*;
    use num_traits::{One, Zero};
    use stwo::core::fields::m31::BaseField;
    use stwo::core::fields::qm31::SecureField;
    use stwo::core::fields::FieldExpOps;
    use stwo_constraint_framework::preprocessed_columns::PreProcessedColumnId;
    use stwo::prover::lookups::utils::Fraction;
    use stwo_constraint_framework::{EvalAtRow, FrameworkEval, LogupAtRow, Relation, RelationEntry, INTERACTION_TRACE_IDX, PREPROCESSED_TRACE_IDX};

    fn q4(s: SecureField) -> [u32; 4] { let a = s.to_m31_array(); [a[0].0, a[1].0, a[2].0, a[3].0] }

    impl Zero for F { fn zero() -> F { bconst(0) } fn is_zero(&self) -> bool { is_bconst(self, 0) } }
    impl One for F { fn one() -> F { bconst(1) } }
    /// a recorded expression has no inverse (no opcode divides: constraints are polynomial identities); Stwo's own recording
    /// evaluators refuse it the same way
    impl FieldExpOps for F { fn inverse(&self) -> F { unimplemented!("a recorded constraint cannot invert a trace expression") } }
    impl From<BaseField> for F { fn from(v: BaseField) -> F { bconst(v.0) } }
    impl AddAssign<BaseField> for F { fn add_assign(&mut self, o: BaseField) { *self = self.clone() + bconst(o.0); } }
    impl Mul<BaseField> for F { type Output = F; fn mul(self, o: BaseField) -> F { self * bconst(o.0) } }
    impl Add<SecureField> for F { type Output = EF; fn add(self, o: SecureField) -> EF { econst(q4(o)) + self } }
    impl Mul<SecureField> for F { type Output = EF; fn mul(self, o: SecureField) -> EF { econst(q4(o)) * self } }

    impl Zero for EF { fn zero() -> EF { econst([0; 4]) } fn is_zero(&self) -> bool { matches!(*self.0, ENode::Const(c) if c == [0; 4]) } }
    impl One for EF { fn one() -> EF { econst([1, 0, 0, 0]) } }
    impl From<SecureField> for EF { fn from(v: SecureField) -> EF { econst(q4(v)) } }
    impl Add<SecureField> for EF { type Output = EF; fn add(self, o: SecureField) -> EF { self + econst(q4(o)) } }
    impl Sub<SecureField> for EF { type Output = EF; fn sub(self, o: SecureField) -> EF { self - econst(q4(o)) } }
    impl Mul<SecureField> for EF { type Output = EF; fn mul(self, o: SecureField) -> EF { self * econst(q4(o)) } }

    /// The evaluator handed to `FrameworkEval::evaluate`.
    pub struct RecordingEval<'a> {
        pub rec: Recorder,
        pub loc: &'a mut TraceLocations,
        /// stwo-constraint-framework's logup state: `add_to_relation` / `finalize_logup*` are the trait's own (`logup_proxy!`) and end in
        /// `next_extension_interaction_mask(INTERACTION_TRACE_IDX, [-1, 0])` + `add_constraint` calls on this evaluator
        pub logup: LogupAtRow<Self>,
    }
    impl<'a> RecordingEval<'a> {
        pub fn new(log_size: u32, claimed_sum: SecureField, loc: &'a mut TraceLocations) -> Self {
            Self { rec: Recorder::default(), loc, logup: LogupAtRow::new(INTERACTION_TRACE_IDX, claimed_sum, log_size) }
        }
    }
    impl<'a> EvalAtRow for RecordingEval<'a> {
        type F = F;
        type EF = EF;
        fn next_interaction_mask<const N: usize>(&mut self, interaction: usize, offsets: [isize; N]) -> [F; N] {
            let offs: Vec<i32> = offsets.iter().map(|&o| o as i32).collect();
            let col = self.rec.new_column(interaction as u32, self.loc.take(interaction), &offs);
            offsets.map(|o| b(BNode::Col { col, off: o as i32 }))
        }
        fn get_preprocessed_column(&mut self, column: PreProcessedColumnId) -> F {
            let index = self.loc.preprocessed_index(&column.id);
            let col = self.rec.preprocessed_column(PREPROCESSED_TRACE_IDX as u32, index);
            b(BNode::Col { col, off: 0 })
        }
        fn add_constraint<G>(&mut self, constraint: G)
        where
            EF: Mul<G, Output = EF> + From<G>,
        {
            self.rec.constraints.push(EF::from(constraint));
        }
        fn combine_ef(values: [F; 4]) -> EF { e(ENode::Combine(values)) }
        /// `EvalAtRow::add_to_relation` — the trait's default body (Fraction::new(multiplicity, relation.combine(values)) ->
        /// write_logup_frac) with the fraction ALSO kept as two roots of the recorder: the interaction trace is generated from them
        /// [upstream-recollection: `RelationEntry { relation, multiplicity, values }`, `Fraction::new`, `Relation::combine`]
        fn add_to_relation<R: Relation<F, EF>>(&mut self, entry: RelationEntry<'_, F, EF, R>) {
            let frac = Fraction::new(entry.multiplicity.clone(), entry.relation.combine(entry.values));
            self.rec.fracs.push((frac.numerator.clone(), frac.denominator.clone()));
            self.write_logup_frac(frac);
        }
        // write_logup_frac / finalize_logup / finalize_logup_in_pairs / finalize_logup_batched: Stwo's own (the logup CONSTRAINTS are
        // its statements, not a restatement); they end in next_extension_interaction_mask + add_constraint calls on this evaluator
        stwo_constraint_framework::logup_proxy!();
    }

    /// `FrameworkComponent::new(tree_span_provider, eval, claimed_sum)` (reference machine.rs:265-270) for the device route: runs the
    /// component's `evaluate` once over the recorder.  `loc` plays `TraceLocationAllocator`: components are recorded in the order the
    /// reference creates them, each takes the next free columns of the main and interaction trees; preprocessed columns are shared
    /// by id.  The bound is `eval.max_constraint_log_degree_bound() - eval.log_size()` (components/mod.rs:44-45: +2 for the machine).
    pub fn record_component<E: FrameworkEval>(eval: &E, loc: &mut TraceLocations, claimed_sum: SecureField) -> RecordedComponent {
        let log_size = eval.log_size();
        let done = eval.evaluate(RecordingEval::new(log_size, claimed_sum, loc));
        done.rec.finish(log_size, eval.max_constraint_log_degree_bound() - log_size)
    }
}
#[cfg(stwo_traits)]
pub use stwo_glue::{record_component, RecordingEval};

// ------------------------------------------------------------------------------------------------ column bookkeeping ----
/// `TraceLocationAllocator` for recorded components (reference machine.rs:264: `TraceLocationAllocator::default()`): the next free
/// column of each trace tree, and the preprocessed columns by id in first-request order — the order the reference commits them in
/// (trace/eval.rs:24-33 requests `PreprocessedColumn::STRING_IDS` then `ProgramColumn::STRING_IDS`; machine.rs:208-217 commits the
/// same sequence).
#[derive(Default, Debug)]
pub struct TraceLocations {
    pub next: [u32; 3],
    pub preprocessed_ids: Vec<String>,
}
impl TraceLocations {
    pub fn take(&mut self, interaction: usize) -> u32 { let i = self.next[interaction]; self.next[interaction] += 1; i }
    pub fn preprocessed_index(&mut self, id: &str) -> u32 {
        if let Some(i) = self.preprocessed_ids.iter().position(|x| x == id) { return i as u32; }
        self.preprocessed_ids.push(id.to_owned());
        (self.preprocessed_ids.len() - 1) as u32
    }
}

/// What one `evaluate` run leaves behind: the component's columns (tree, index in tree, mask offsets) and the constraint roots.
#[derive(Default, Debug)]
pub struct Recorder {
    pub col_tree: Vec<u32>,
    pub col_index: Vec<u32>,
    pub masks: Vec<Vec<i32>>,
    pub constraints: Vec<EF>,
    /// the relation entries in declaration order: (multiplicity, relation.combine(values))
    pub fracs: Vec<(EF, EF)>,
    preprocessed: HashMap<u32, u32>,        // preprocessed tree index -> component column
}
impl Recorder {
    pub fn new_column(&mut self, tree: u32, index: u32, offsets: &[i32]) -> u32 {
        self.col_tree.push(tree); self.col_index.push(index); self.masks.push(offsets.to_vec());
        (self.col_tree.len() - 1) as u32
    }
    pub fn preprocessed_column(&mut self, tree: u32, index: u32) -> u32 {
        if let Some(&c) = self.preprocessed.get(&index) { return c; }
        let c = self.new_column(tree, index, &[0]);
        self.preprocessed.insert(index, c);
        c
    }

    /// Lower the recorded DAG (air_program.py ProgramBuilder.build, rule for rule).
    pub fn finish(self, log_size: u32, log_constraint_degree_bound: u32) -> RecordedComponent {
        let mut lo = Lowering::default();
        let roots: Vec<Root> = self.constraints.iter().map(|c| { let (i, secure) = lo.constraint_root(c); Root::Cons(i, secure) }).collect();
        let (program, n_regs) = lo.emit(&roots);
        let mut mask_count = Vec::new();
        let mut mask_offsets = Vec::new();
        for m in &self.masks { mask_count.push(m.len() as u32); mask_offsets.extend_from_slice(m); }
        // The fraction program.  Which batch an entry belongs to is read off what finalize_logup* did: it took one secure interaction
        // column per batch, so (interaction columns) / 4 batches — one entry each (finalize_logup) or two (finalize_logup_in_pairs, the
        // last batch one when the count is odd); the reference uses no other batching (a custom `Batching` would have to be passed in).
        let n_logup_cols = (self.col_tree.iter().filter(|&&t| t == 2).count() / 4) as u32;
        let f = self.fracs.len() as u32;
        let per_batch = if f == n_logup_cols { 1 } else if f.div_ceil(2) == n_logup_cols { 2 } else {
            panic!("{f} relation entries in {n_logup_cols} logup columns: neither finalize_logup nor finalize_logup_in_pairs")
        };
        let mut fl = Lowering::default();
        let froots: Vec<Root> = self.fracs.iter().enumerate().map(|(i, (num, den))| {
            let (n, secure) = fl.constraint_root(num);           // a lifted base multiplicity (1, -m) stays a base numerator: NX_C_FRACB
            Root::Frac { num: n, num_secure: secure, den: fl.ext(den), batch: i as u32 / per_batch }
        }).collect();
        let (logup_program, logup_n_regs) = if froots.is_empty() { (Vec::new(), 1) } else { fl.emit(&froots) };
        RecordedComponent {
            log_size, program, n_regs, n_constraints: roots.len() as u32,
            econsts: lo.econsts.iter().flat_map(|q| q.iter().copied()).collect(),
            col_tree: self.col_tree, col_index: self.col_index, mask_count, mask_offsets, log_constraint_degree_bound,
            logup_program, logup_n_regs, logup_econsts: fl.econsts.iter().flat_map(|q| q.iter().copied()).collect(), n_logup_cols,
        }
    }
}

// ------------------------------------------------------------------------------------------------ lowering ----
#[derive(Clone, Copy, PartialEq, Eq, Hash, Debug)]
enum Key {
    Load(u32, i32), LoadE(u32, i32), Const(u32), ConstE(u32),
    Add(usize, usize), Sub(usize, usize), Mul(usize, usize), Neg(usize),
    AddE(usize, usize), SubE(usize, usize), MulE(usize, usize), MulEB(usize, usize), AddEB(usize, usize),
}
impl Key {
    fn children(&self) -> Vec<usize> {
        match *self {
            Key::Add(a, b) | Key::Sub(a, b) | Key::Mul(a, b) | Key::AddE(a, b) | Key::SubE(a, b) | Key::MulE(a, b) | Key::MulEB(a, b) | Key::AddEB(a, b) => vec![a, b],
            Key::Neg(a) => vec![a],
            _ => vec![],
        }
    }
    fn is_load(&self) -> bool { matches!(self, Key::Load(..) | Key::LoadE(..)) }
    fn is_secure(&self) -> bool { matches!(self, Key::LoadE(..) | Key::ConstE(..) | Key::AddE(..) | Key::SubE(..) | Key::MulE(..) | Key::MulEB(..) | Key::AddEB(..)) }
}

/// What a lowered program ends in: a constraint (node, secure?) or a relation entry's fraction (include/nexus_hip.h NX_C_FRAC / NX_C_FRACB)
#[derive(Clone, Copy, Debug)]
enum Root { Cons(usize, bool), Frac { num: usize, num_secure: bool, den: usize, batch: u32 } }
impl Root {
    fn nodes(&self) -> Vec<usize> { match *self { Root::Cons(i, _) => vec![i], Root::Frac { num, den, .. } => vec![num, den] } }
}

#[derive(Default)]
struct Lowering {
    nodes: Vec<Key>,
    cse: HashMap<Key, usize>,
    seen_b: HashMap<*const BNode, usize>,       // DAG nodes already lowered (shared sub-expressions are walked once)
    seen_e: HashMap<*const ENode, usize>,
    econsts: Vec<[u32; 4]>,
}
impl Lowering {
    fn node(&mut self, k: Key) -> usize {
        if let Some(&i) = self.cse.get(&k) { return i; }
        self.nodes.push(k); self.cse.insert(k, self.nodes.len() - 1);
        self.nodes.len() - 1
    }
    fn conste(&mut self, q: [u32; 4]) -> usize {
        let q = [q[0] % P, q[1] % P, q[2] % P, q[3] % P];
        let i = match self.econsts.iter().position(|x| *x == q) { Some(i) => i, None => { self.econsts.push(q); self.econsts.len() - 1 } };
        self.node(Key::ConstE(i as u32))
    }
    fn base(&mut self, x: &F) -> usize {
        let p = Rc::as_ptr(&x.0);
        if let Some(&i) = self.seen_b.get(&p) { return i; }
        let i = match &*x.0 {
            BNode::Col { col, off } => self.node(Key::Load(*col, *off)),
            BNode::Const(v) => self.node(Key::Const(*v % P)),
            BNode::Add(l, r) => { let (a, c) = (self.base(l), self.base(r)); self.node(Key::Add(a.min(c), a.max(c))) }     // commutative: canonical operand order
            BNode::Mul(l, r) => { let (a, c) = (self.base(l), self.base(r)); self.node(Key::Mul(a.min(c), a.max(c))) }
            BNode::Sub(l, r) => { let (a, c) = (self.base(l), self.base(r)); self.node(Key::Sub(a, c)) }
            BNode::Neg(v) => { let a = self.base(v); self.node(Key::Neg(a)) }
        };
        self.seen_b.insert(p, i);
        i
    }
    fn ext(&mut self, x: &EF) -> usize {
        let p = Rc::as_ptr(&x.0);
        if let Some(&i) = self.seen_e.get(&p) { return i; }
        let i = match &*x.0 {
            ENode::Const(q) => self.conste(*q),
            ENode::Combine(v) => {
                // four consecutive columns at one offset are ONE secure load (what next_extension_interaction_mask produces); anything
                // else is spelt out over the basis 1, i, u, iu
                let cols: Vec<Option<(u32, i32)>> = v.iter().map(|f| match &*f.0 { BNode::Col { col, off } => Some((*col, *off)), _ => None }).collect();
                let consecutive = cols.iter().all(|c| c.is_some()) && (1..4).all(|k| cols[k].unwrap() == (cols[0].unwrap().0 + k as u32, cols[0].unwrap().1));
                if consecutive { let (c, o) = cols[0].unwrap(); self.node(Key::LoadE(c, o)) } else {
                    let basis = [[0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]];
                    let zero = self.conste([0; 4]);
                    let v0 = self.base(&v[0]);
                    let mut acc = self.node(Key::AddEB(zero, v0));
                    for k in 0..3 {
                        let (bk, vk) = (self.conste(basis[k]), self.base(&v[k + 1]));
                        let t = self.node(Key::MulEB(bk, vk));
                        acc = self.node(Key::AddE(acc.min(t), acc.max(t)));
                    }
                    acc
                }
            }
            ENode::FromBase(f) => { let (z, v) = (self.conste([0; 4]), self.base(f)); self.node(Key::AddEB(z, v)) }
            ENode::Add(l, r) => { let (a, c) = (self.ext(l), self.ext(r)); self.node(Key::AddE(a.min(c), a.max(c))) }
            ENode::Mul(l, r) => { let (a, c) = (self.ext(l), self.ext(r)); self.node(Key::MulE(a.min(c), a.max(c))) }
            ENode::Sub(l, r) => { let (a, c) = (self.ext(l), self.ext(r)); self.node(Key::SubE(a, c)) }
            ENode::AddB(l, r) => { let (a, c) = (self.ext(l), self.base(r)); self.node(Key::AddEB(a, c)) }
            ENode::MulB(l, r) => { let (a, c) = (self.ext(l), self.base(r)); self.node(Key::MulEB(a, c)) }
        };
        self.seen_e.insert(p, i);
        i
    }
    /// a constraint that is a lifted base expression stays a base constraint (NX_C_CONSTRAINT_B: one multiply-add per coordinate of
    /// the accumulator instead of a secure product)
    fn constraint_root(&mut self, c: &EF) -> (usize, bool) {
        if let ENode::FromBase(f) = &*c.0 { return (self.base(f), false); }
        (self.ext(c), true)
    }

    fn emit(&mut self, roots: &[Root]) -> (Vec<sys::nx_cinstr>, u32) {
        #[derive(Clone, Copy)]
        enum Item { Node(usize), Root(Root) }
        const CHUNK: usize = 8;
        let nodes = self.nodes.clone();
        let mut order: Vec<Item> = Vec::new();
        let mut emitted = vec![false; nodes.len()];
        // post-order emission of the not-yet-emitted part of a root; loads_only: just its column loads (hoisted in front of a chunk)
        let emit_root = |root: usize, loads_only: bool, order: &mut Vec<Item>, emitted: &mut Vec<bool>| {
            let mut stack: Vec<(usize, bool)> = vec![(root, false)];
            while let Some((n, done)) = stack.pop() {
                if emitted[n] { continue; }
                let kids = nodes[n].children();
                if done || kids.is_empty() {
                    if loads_only && !nodes[n].is_load() { continue; }
                    order.push(Item::Node(n)); emitted[n] = true;
                    continue;
                }
                if !loads_only { stack.push((n, true)); }
                for &k in kids.iter().rev() { stack.push((k, false)); }
            }
        };
        for (j, root) in roots.iter().enumerate() {
            if j % CHUNK == 0 { for r in &roots[j..(j + CHUNK).min(roots.len())] { for n in r.nodes() { emit_root(n, true, &mut order, &mut emitted); } } }
            for n in root.nodes() { emit_root(n, false, &mut order, &mut emitted); }
            order.push(Item::Root(*root));
        }
        // last use of every node
        let mut last: HashMap<usize, usize> = HashMap::new();
        for (pos, it) in order.iter().enumerate() {
            match *it { Item::Root(r) => { for n in r.nodes() { last.insert(n, pos); } } Item::Node(i) => { for k in nodes[i].children() { last.insert(k, pos); } } }
        }
        // linear scan: base registers and secure quads from separate pools
        let (mut free_b, mut free_e, mut n_b, mut n_e): (Vec<u32>, Vec<u32>, u32, u32) = (vec![], vec![], 0, 0);
        let mut slot: HashMap<usize, (bool, u32)> = HashMap::new();
        for (pos, it) in order.iter().enumerate() {
            if let Item::Node(i) = *it {
                let s = if nodes[i].is_secure() { (true, free_e.pop().unwrap_or_else(|| { n_e += 1; n_e - 1 })) } else { (false, free_b.pop().unwrap_or_else(|| { n_b += 1; n_b - 1 })) };
                slot.insert(i, s);
            }
            let mut touched: Vec<usize> = match *it { Item::Root(r) => r.nodes(), Item::Node(i) => nodes[i].children() };
            touched.sort_unstable(); touched.dedup();
            for a in touched {
                if last.get(&a) == Some(&pos) { if let Some(&(sec, idx)) = slot.get(&a) { if sec { free_e.push(idx) } else { free_b.push(idx) } } }
            }
            if let Item::Node(i) = *it { if !last.contains_key(&i) { let (sec, idx) = slot[&i]; if sec { free_e.push(idx) } else { free_b.push(idx) } } }
        }
        let n_regs = (n_b + 4 * n_e).max(1);
        let reg = |i: usize| -> u32 { let (sec, idx) = slot[&i]; if sec { n_b + 4 * idx } else { idx } };
        let ins = |op: u32, dst: u32, a: u32, b: u32| sys::nx_cinstr { op, dst, a, b };
        let mut out = Vec::with_capacity(order.len());
        for it in &order {
            match *it {
                Item::Root(Root::Cons(i, secure)) => out.push(ins(if secure { sys::NX_C_CONSTRAINT_E } else { sys::NX_C_CONSTRAINT_B }, 0, reg(i), 0)),
                Item::Root(Root::Frac { num, num_secure, den, batch }) => out.push(ins(if num_secure { sys::NX_C_FRAC } else { sys::NX_C_FRACB }, batch, reg(num), reg(den))),
                Item::Node(i) => out.push(match nodes[i] {
                    Key::Load(c, o) => ins(sys::NX_C_LOAD, reg(i), c, o as u32),
                    Key::LoadE(c, o) => ins(sys::NX_C_LOADE, reg(i), c, o as u32),
                    Key::Const(v) => ins(sys::NX_C_CONST, reg(i), v, 0),
                    Key::ConstE(k) => ins(sys::NX_C_CONSTE, reg(i), k, 0),
                    Key::Neg(a) => ins(sys::NX_C_NEG, reg(i), reg(a), 0),
                    Key::Add(a, c) => ins(sys::NX_C_ADD, reg(i), reg(a), reg(c)),
                    Key::Sub(a, c) => ins(sys::NX_C_SUB, reg(i), reg(a), reg(c)),
                    Key::Mul(a, c) => ins(sys::NX_C_MUL, reg(i), reg(a), reg(c)),
                    Key::AddE(a, c) => ins(sys::NX_C_ADDE, reg(i), reg(a), reg(c)),
                    Key::SubE(a, c) => ins(sys::NX_C_SUBE, reg(i), reg(a), reg(c)),
                    Key::MulE(a, c) => ins(sys::NX_C_MULE, reg(i), reg(a), reg(c)),
                    Key::MulEB(a, c) => ins(sys::NX_C_MULEB, reg(i), reg(a), reg(c)),
                    Key::AddEB(a, c) => ins(sys::NX_C_ADDEB, reg(i), reg(a), reg(c)),
                }),
            }
        }
        (out, n_regs)
    }
}
