"""Recording evaluator for AIR constraints -> the straight-line program `nx_eval_constraint_program` runs on device.

This is the Python twin of the Rust piece a `HipBackend` shim needs for SURVEY.md §8(f) rank 1: an `EvalAtRow`
implementation whose field type records instead of computing — exactly what Stwo's `InfoEvaluator` does to discover the
column masks (reference prover/src/components/mod.rs:59-67; prover2/machine/src/framework/traits/erased.rs:94-104).
Running the reference's `add_constraints` closure once over it yields the expression DAG; `build()` orders it, shares
common subexpressions, allocates the per-row registers (a secure-field value = 4 consecutive registers) and emits the
`nx_cinstr` words of include/nexus_hip.h.

    pb = ProgramBuilder()
    a, a_next = pb.next_trace_mask(col=0, offsets=(0, 1))      # EvalAtRow::next_interaction_mask
    (b,) = pb.next_trace_mask(col=1)
    pb.add_constraint((a_next - a - pb.const(1)) * b)           # EvalAtRow::add_constraint
    prog = pb.build()

Lookups are recorded the way the reference declares them (prover/src/components/mod.rs:48-56, extensions/keccak/round/constraints.rs:95-116):
    rel = pb.relation(z, alpha, 3)                               # LookupElements<3>: z, alpha (run-time secure constants)
    pb.add_to_relation(rel, 1, [a, b, c])                        # RelationEntry::new(relation, multiplicity, &values)
    pb.add_to_relation(rel, -m, [x, y, w])
    pb.finalize_logup_in_pairs(first_interaction_col, claimed_sum / N)   # or finalize_logup / finalize_logup_batched
and yield BOTH the logup constraints (in `build()`, exactly stwo-constraint-framework's finalize_logup_batched) and the fraction
program `build_logup()` that nx_logup_program turns into the interaction trace on the device — the same relation entries, so the
trace satisfies the constraints by construction.
"""
import numpy as np

P = (1 << 31) - 1
(LOAD, CONST, ADD, SUB, MUL, NEG, CONSTE, ADDE, SUBE, MULE, MULEB, ADDEB, LOADE, CONSTRAINT_B, CONSTRAINT_E, FRAC, FRACB) = range(17)


class Expr:
    __slots__ = ("pb", "id", "kind")

    def __init__(self, pb, nid, kind):
        self.pb, self.id, self.kind = pb, nid, kind

    def _lift(self, o):
        if isinstance(o, Expr):
            return o
        return self.pb.const(int(o))

    def __add__(self, o):
        return self.pb._bin("add", self, self._lift(o))
    __radd__ = __add__

    def __sub__(self, o):
        return self.pb._bin("sub", self, self._lift(o))

    def __rsub__(self, o):
        return self.pb._bin("sub", self._lift(o), self)

    def __mul__(self, o):
        return self.pb._bin("mul", self, self._lift(o))
    __rmul__ = __mul__

    def __neg__(self):
        return self.pb._neg(self)


class Program:
    def __init__(self, instrs, n_regs, econsts, n_constraints, masks=None):
        self.instrs, self.n_regs, self.econsts, self.n_constraints = instrs, n_regs, econsts, n_constraints
        self.masks = masks or {}     # column -> row offsets, in declaration order (what InfoEvaluator collects)


class Component:
    """What FrameworkComponent<E> is to Stwo (reference prover/src/components/mod.rs:15-57): a trace log size, the recorded
    constraints, the place of the component's columns in the three trace trees (TraceLocationAllocator) and the mask offsets
    each column is sampled at.  cols[k] = (tree, column index in that tree) for the program's column k; a column the program
    never loads is still sampled at offset 0 (every committed column must be claimed by a component)."""

    def __init__(self, log_size, program, cols, masks=None, log_constraint_degree_bound=0):
        self.log_size, self.program, self.cols = int(log_size), program, [(int(t), int(i)) for t, i in cols]
        # the component's own bound (the reference's is per component: components/mod.rs:12, extensions/multiplicity.rs:108-110); 0 = the session config's
        self.log_constraint_degree_bound = int(log_constraint_degree_bound)
        self.masks = [list(masks[k]) if masks is not None else list(program.masks.get(k, (0,))) for k in range(len(self.cols))]


class Relation:
    """LookupElements<N> as the recorder sees them: z and the alpha powers are secure constants of the program (run-time values)."""

    def __init__(self, z, alpha_powers):
        self.z, self.alpha_powers = z, alpha_powers

    def combine(self, values):
        """Relation::combine: sum_i alpha^i values_i - z"""
        assert 1 <= len(values) <= len(self.alpha_powers)
        acc = None
        for a, v in zip(self.alpha_powers, values):
            term = a * v
            acc = term if acc is None else acc + term
        return acc - self.z


class ProgramBuilder:
    def __init__(self):
        self.nodes = []          # (op, kind, args...)
        self.cse = {}
        self.constraints = []    # node ids in declaration order
        self.econsts = []
        self.masks = {}
        self.entries = []        # relation entries since the last finalize_logup*: (numerator Expr, denominator Expr)
        self.fractions = []      # finalized: (numerator id, denominator id, logup column)
        self.n_logup_cols = 0

    def _node(self, key, kind):
        if key in self.cse:
            return Expr(self, self.cse[key], kind)
        self.nodes.append((key, kind))
        self.cse[key] = len(self.nodes) - 1
        return Expr(self, len(self.nodes) - 1, kind)

    # ---- leaves
    def next_trace_mask(self, col, offsets=(0,)):
        """Base-field column `col` (index into the column table handed to nx_eval_constraint_program) at the given row offsets."""
        m = self.masks.setdefault(int(col), [])
        m.extend(int(o) for o in offsets if int(o) not in m)
        return [self._node(("load", int(col), int(o)), "B") for o in offsets]

    def next_secure_mask(self, first_col, offsets=(0,)):
        """A secure (QM31) column stored as 4 consecutive coordinate columns starting at first_col (logup columns)."""
        for k in range(4):
            m = self.masks.setdefault(int(first_col) + k, [])
            m.extend(int(o) for o in offsets if int(o) not in m)
        return [self._node(("loade", int(first_col), int(o)), "E") for o in offsets]

    def const(self, v):
        return self._node(("const", int(v) % P), "B")

    def econst(self, q4):
        q = tuple(int(x) % P for x in q4)
        if q not in self.econsts:
            self.econsts.append(q)
        return self._node(("conste", self.econsts.index(q)), "E")

    # ---- arithmetic
    def _bin(self, op, x, y):
        if x.kind == "B" and y.kind == "B":
            if op in ("add", "mul") and y.id < x.id:
                x, y = y, x                      # commutative: canonical operand order for CSE
            return self._node((op, x.id, y.id), "B")
        if x.kind == "E" and y.kind == "E":
            if op in ("add", "mul") and y.id < x.id:
                x, y = y, x
            return self._node((op + "e", x.id, y.id), "E")
        # mixed: secure (op) base
        if op == "mul":
            e, b = (x, y) if x.kind == "E" else (y, x)
            return self._node(("muleb", e.id, b.id), "E")
        if op == "add":
            e, b = (x, y) if x.kind == "E" else (y, x)
            return self._node(("addeb", e.id, b.id), "E")
        if x.kind == "E":                         # E - B = E + (-B)
            return self._node(("addeb", x.id, self._neg(y).id), "E")
        return self._node(("addeb", self._neg_e(y).id, x.id), "E")   # B - E = (-E) + B

    def _neg(self, x):
        if x.kind == "B":
            return self._node(("neg", x.id), "B")
        return self._neg_e(x)

    def _neg_e(self, x):
        zero = self.econst((0, 0, 0, 0))
        return self._node(("sube", zero.id, x.id), "E")

    def add_constraint(self, expr):
        self.constraints.append(expr.id)

    # ---- lookups (stwo-constraint-framework: RelationEntry, EvalAtRow::add_to_relation, finalize_logup*)
    def relation(self, z, alpha, n):
        """LookupElements<n> with the drawn z and alpha (4 words each): alpha powers 1, alpha, ..., alpha^(n-1) as secure constants"""
        from_q = lambda q: tuple(int(x) % P for x in q)
        pw, cur = [], (1, 0, 0, 0)
        for _ in range(n):
            pw.append(self.econst(cur))
            cur = _qm31_mul(cur, from_q(alpha))
        return Relation(self.econst(from_q(z)), pw)

    def add_to_relation(self, relation, multiplicity, values):
        """eval.add_to_relation(RelationEntry::new(relation, multiplicity, &values)): one fraction multiplicity / relation.combine(values)"""
        num = multiplicity if isinstance(multiplicity, Expr) else self.const(int(multiplicity))
        self.entries.append((num, relation.combine([v if isinstance(v, Expr) else self.const(int(v)) for v in values])))

    def finalize_logup(self, first_interaction_col, cumsum_shift):
        self.finalize_logup_batched(first_interaction_col, cumsum_shift, list(range(len(self.entries))))

    def finalize_logup_in_pairs(self, first_interaction_col, cumsum_shift):
        self.finalize_logup_batched(first_interaction_col, cumsum_shift, [i // 2 for i in range(len(self.entries))])

    def finalize_logup_batched(self, first_interaction_col, cumsum_shift, batching):
        """stwo-constraint-framework's finalize_logup_batched: entry i belongs to batch batching[i]; batch j owns the secure interaction
        column first_interaction_col + 4 j; per batch the constraint (S_j - S_{j-1}) D - N with N / D the batch's fraction sum, the
        last one over the [-1, 0] mask with the claimed-sum shift.  cumsum_shift: claimed_sum / N (4 words, a run-time constant)."""
        assert len(batching) == len(self.entries) and self.entries
        last = max(batching)
        assert sorted(set(batching)) == list(range(last + 1)), "every batch 0 .. last must hold an entry"
        shift = self.econst(tuple(int(x) % P for x in cumsum_shift))
        base = self.n_logup_cols
        prev = None
        for j in range(last + 1):
            num = den = None
            for (n, d), b in zip(self.entries, batching):
                if b != j:
                    continue
                if den is None:
                    num, den = n, d
                else:
                    num, den = num * d + n * den, den * d            # Fraction::add
            col = int(first_interaction_col) + 4 * j
            if j < last:
                (cur,) = self.next_secure_mask(col)
                diff = cur if prev is None else cur - prev
            else:
                prow, cur = self.next_secure_mask(col, (-1, 0))
                diff = cur - prow
                if prev is not None:
                    diff = diff - prev
                diff = diff + shift
            self.add_constraint(diff * den - num)
            prev = cur
            for (n, d), b in zip(self.entries, batching):
                if b == j:
                    self.fractions.append((n.id, d.id, base + j))
        self.n_logup_cols = base + last + 1
        self.entries = []

    def build_logup(self):
        """The fraction program of the finalized relation entries (nx_logup_program / oracle logup_program): the interaction trace."""
        assert not self.entries, "relation entries without a finalize_logup*"
        prog = self._lower([("frac", f) for f in self.fractions])
        prog.n_logup_cols = self.n_logup_cols
        return prog

    # ---- lowering
    def build(self):
        return self._lower([("cons", c) for c in self.constraints])

    def _lower(self, roots):
        nodes = self.nodes
        root_ids = lambda r: [r[1]] if r[0] == "cons" else [r[1][0], r[1][1]]
        # liveness: only nodes reachable from a root are emitted
        needed = [False] * len(nodes)
        stack = [i for r in roots for i in root_ids(r)]
        while stack:
            i = stack.pop()
            if needed[i]:
                continue
            needed[i] = True
            key = nodes[i][0]
            if key[0] in ("add", "sub", "mul", "adde", "sube", "mule", "muleb", "addeb"):
                stack += [key[1], key[2]]
            elif key[0] == "neg":
                stack.append(key[1])
        # emission order: constraints in declaration order (alpha power j belongs to the j-th add_constraint), each preceded
        # by the not-yet-emitted part of its expression (post-order), so values are computed right before their first use and
        # the register file stays small; the column loads of every group of 8 constraints are hoisted in front of the
        # group so that runs of LOADs form (the kernel issues a run's reads together).
        BIN = ("add", "sub", "mul", "adde", "sube", "mule", "muleb", "addeb")

        def children(i):
            key = nodes[i][0]
            return [key[1], key[2]] if key[0] in BIN else [key[1]] if key[0] == "neg" else []

        order, emitted = [], set()

        def emit(i, loads_only=False):
            stack = [(i, False)]
            while stack:
                n, done = stack.pop()
                if n in emitted:
                    continue
                is_load = nodes[n][0][0] in ("load", "loade")
                if done or not children(n):
                    if loads_only and not is_load:
                        continue
                    order.append(("node", n)); emitted.add(n)
                    continue
                if not loads_only:
                    stack.append((n, True))
                for c in reversed(children(n)):
                    stack.append((c, False))

        CHUNK = 8
        for j, root in enumerate(roots):
            if j % CHUNK == 0:
                for nxt in roots[j:j + CHUNK]:
                    for nid in root_ids(nxt):
                        emit(nid, loads_only=True)
            for nid in root_ids(root):
                emit(nid)
            order.append(root)
        # last use of every node (position in `order`)
        last = {}
        for pos, (what, i) in enumerate(order):
            if what == "cons":
                last[i] = pos
            elif what == "frac":
                last[i[0]] = pos; last[i[1]] = pos
            else:
                key = nodes[i][0]
                for a in ([key[1], key[2]] if key[0] in ("add", "sub", "mul", "adde", "sube", "mule", "muleb", "addeb") else [key[1]] if key[0] == "neg" else []):
                    last[a] = pos
        # linear-scan allocation: B registers and E register quads from separate pools
        free_b, free_e, n_b, n_e = [], [], 0, 0
        slot = {}
        alloc_log = []
        for pos, (what, i) in enumerate(order):
            if what == "node":
                if nodes[i][1] == "B":
                    if free_b:
                        slot[i] = ("B", free_b.pop())
                    else:
                        slot[i] = ("B", n_b); n_b += 1
                else:
                    if free_e:
                        slot[i] = ("E", free_e.pop())
                    else:
                        slot[i] = ("E", n_e); n_e += 1
            alloc_log.append(None)
            # free the operands whose last use is this position (after the instruction has read them)
            touched = []
            if what == "cons":
                touched = [i]
            elif what == "frac":
                touched = [i[0], i[1]]
            else:
                key = nodes[i][0]
                touched = [key[1], key[2]] if key[0] in ("add", "sub", "mul", "adde", "sube", "mule", "muleb", "addeb") else [key[1]] if key[0] == "neg" else []
            for a in set(touched):
                if last.get(a) == pos and a in slot:
                    kind, idx = slot[a]
                    (free_b if kind == "B" else free_e).append(idx)
            if what == "node" and i not in last:      # dead on arrival cannot happen (needed[] filter), keep for safety
                kind, idx = slot[i]
                (free_b if kind == "B" else free_e).append(idx)
        n_regs = max(1, n_b + 4 * n_e)

        def reg(i):
            kind, idx = slot[i]
            return idx if kind == "B" else n_b + 4 * idx

        out = []
        opmap = {"add": ADD, "sub": SUB, "mul": MUL, "adde": ADDE, "sube": SUBE, "mule": MULE, "muleb": MULEB, "addeb": ADDEB}
        for what, i in order:
            if what == "cons":
                out.append((CONSTRAINT_B if nodes[i][1] == "B" else CONSTRAINT_E, 0, reg(i), 0))
                continue
            if what == "frac":
                assert nodes[i[1]][1] == "E", "a relation's denominator is a secure-field value"
                out.append((FRACB if nodes[i[0]][1] == "B" else FRAC, i[2], reg(i[0]), reg(i[1])))
                continue
            key = nodes[i][0]
            if key[0] == "load":
                out.append((LOAD, reg(i), key[1], key[2] & 0xFFFFFFFF))
            elif key[0] == "loade":
                out.append((LOADE, reg(i), key[1], key[2] & 0xFFFFFFFF))
            elif key[0] == "const":
                out.append((CONST, reg(i), key[1], 0))
            elif key[0] == "conste":
                out.append((CONSTE, reg(i), key[1], 0))
            elif key[0] == "neg":
                out.append((NEG, reg(i), reg(key[1]), 0))
            else:
                out.append((opmap[key[0]], reg(i), reg(key[1]), reg(key[2])))
        instrs = np.array(out, dtype=np.uint32).reshape(-1, 4)
        econsts = np.array(self.econsts, dtype=np.uint32).reshape(-1, 4) if self.econsts else np.zeros((0, 4), np.uint32)
        n_cons = sum(1 for r in roots if r[0] == "cons")
        return Program(instrs, n_regs, econsts, n_cons, {k: list(v) for k, v in self.masks.items()})


def _qm31_mul(x, y):
    """(a + b u)(c + d u) over CM31 = M31[i] / (i^2 + 1), u^2 = 2 + i — host arithmetic for the alpha powers of a relation"""
    def cmul(p, q):
        return ((p[0] * q[0] - p[1] * q[1]) % P, (p[0] * q[1] + p[1] * q[0]) % P)
    a, b, c, d = (x[0], x[1]), (x[2], x[3]), (y[0], y[1]), (y[2], y[3])
    ac, bd, ad, bc = cmul(a, c), cmul(b, d), cmul(a, d), cmul(b, c)
    r = ((2 * bd[0] - bd[1]) % P, (2 * bd[1] + bd[0]) % P)
    return ((ac[0] + r[0]) % P, (ac[1] + r[1]) % P, (ad[0] + bc[0]) % P, (ad[1] + bc[1]) % P)
