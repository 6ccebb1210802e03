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
"""
import numpy as np

P = (1 << 31) - 1
(LOAD, CONST, ADD, SUB, MUL, NEG, CONSTE, ADDE, SUBE, MULE, MULEB, ADDEB, LOADE, CONSTRAINT_B, CONSTRAINT_E) = range(15)


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


class ProgramBuilder:
    def __init__(self):
        self.nodes = []          # (op, kind, args...)
        self.cse = {}
        self.constraints = []    # node ids in declaration order
        self.econsts = []
        self.masks = {}

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

    # ---- lowering
    def build(self):
        nodes = self.nodes
        # liveness: only nodes reachable from a constraint are emitted
        needed = [False] * len(nodes)
        stack = list(self.constraints)
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
        for j, nid in enumerate(self.constraints):
            if j % CHUNK == 0:
                for nxt in self.constraints[j:j + CHUNK]:
                    emit(nxt, loads_only=True)
            emit(nid)
            order.append(("cons", nid))
        # last use of every node (position in `order`)
        last = {}
        for pos, (what, i) in enumerate(order):
            if what == "cons":
                last[i] = pos
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
        return Program(instrs, n_regs, econsts, len(self.constraints), {k: list(v) for k, v in self.masks.items()})
