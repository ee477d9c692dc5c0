"""Fused native-form objective programs for unconstrained elementwise-sum problems.

BASELINE config C2 ("unconstrained Rosenbrock chain, n = 1e5 — tape f / grad f eval + line
search only", SURVEY.md §8d: kernel = fused f + grad f, 16 n algorithmic bytes) is an objective of
the form

    f(x) = c0 + sum_t  w_t * sum_i  phi_t( x[o_1 + s_1 i], x[o_2 + s_2 i], ..., constants[i] )

with phi_t a small tree of elementwise atoms.  The canonical form the reference hands to IPOPT
(dnlp2smooth.py:42-111) spreads it over 4n variables and 3n equalities; evaluating f and grad f
there costs several sweeps of the tape.  This module compiles the USER's expression tree into a
per-element register program (a handful of instructions) that one HIP kernel interprets with
its register file in LDS: x is read once, grad f is accumulated once (csrc/fused_obj.h).

The program is pure data (no code generation): opcodes below, unary atoms by their tape opcode
and the same `unary_rules` arithmetic as the tape kernels.  Anything outside the supported forms
raises NotFusable and the caller keeps the tape path.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np

from . import atoms as at
from .expressions import Constant, Expression, Parameter, Variable
from .lowering import OP_POWER, UNARY_OPS

F_LOADV, F_LOADC, F_UNARY, F_ADD, F_SUB, F_MUL, F_SCALE, F_ADDC, F_DIV = range(9)
MAX_INSTR = 32


class NotFusable(Exception):
    pass


def _const_value(e) -> Optional[np.ndarray]:
    if isinstance(e, (Constant, Parameter)) or (e.is_constant() and not e.variables()):
        v = e.value
        if v is None:
            raise NotFusable("constant without a value")
        if hasattr(v, "toarray"):
            v = v.toarray()
        return np.asarray(v, dtype=float)
    return None


class _Program:
    def __init__(self, nelem: int):
        self.nelem = int(nelem)
        self.instr: List[tuple] = []      # (op, a, b, off, stride, p, p2, var_id)
        self.consts: List[np.ndarray] = []
        self.nconst = 0

    def add(self, op, a=-1, b=-1, off=0, stride=0, p=0.0, p2=0.0, var=None) -> int:
        self.instr.append((op, a, b, int(off), int(stride), float(p), float(p2), var))
        return len(self.instr) - 1

    def const_vec(self, v: np.ndarray) -> int:
        off = self.nconst
        self.consts.append(np.asarray(v, dtype=float).reshape(-1, order="F"))
        self.nconst += v.size
        return off


class FusedBuilder:
    """Objective expression (minimisation form) -> list of per-element programs."""

    def __init__(self):
        self.c0 = 0.0
        self.programs: List[_Program] = []

    # ---- term level: scalar combinations of full sums ------------------------------------
    def term(self, e: Expression, w: float):
        cv = _const_value(e)
        if cv is not None:
            if cv.size != 1:
                raise NotFusable("non-scalar constant term")
            self.c0 += w * float(cv.reshape(-1)[0])
            return
        if isinstance(e, at.AddExpression):
            for a in e.args:
                self.term(a, w)
            return
        if isinstance(e, at.NegExpression):
            self.term(e.args[0], -w)
            return
        if isinstance(e, at.DivExpression):
            d = _const_value(e.args[1])
            if d is None or d.size != 1:
                raise NotFusable("division by a non-constant")
            self.term(e.args[0], w / float(d.reshape(-1)[0]))
            return
        if isinstance(e, at.MulExpression) and e.size == 1:
            c0, c1 = _const_value(e.args[0]), _const_value(e.args[1])
            if c0 is not None and c0.size == 1:
                self.term(e.args[1], w * float(c0.reshape(-1)[0]))
                return
            if c1 is not None and c1.size == 1:
                self.term(e.args[0], w * float(c1.reshape(-1)[0]))
                return
        if isinstance(e, at.Sum) and e.axis is None:
            self.sum_of(e.args[0], w)
            return
        if isinstance(e, at.quad_over_lin):
            y = _const_value(e.args[1])
            if y is None or y.size != 1:
                raise NotFusable("quad_over_lin with a variable denominator")
            self.sum_of(at.power(e.args[0], 2), w / float(y.reshape(-1)[0]))
            return
        if e.size == 1:
            self.sum_of(e, w)
            return
        raise NotFusable("objective term %s" % type(e).__name__)

    def sum_of(self, e: Expression, w: float):
        prog = _Program(e.size)
        root = self.emit(prog, e)
        if w != 1.0:
            root = prog.add(F_SCALE, a=root, p=w)
        if len(prog.instr) > MAX_INSTR:
            raise NotFusable("element program longer than %d instructions" % MAX_INSTR)
        # programs over the same index range share one pass when they fit together
        for q in self.programs:
            if q.nelem == prog.nelem and len(q.instr) + len(prog.instr) + 1 <= MAX_INSTR:
                base = len(q.instr)
                qroot = base - 1
                coff = q.nconst
                for (op, a, b, off, stride, p, p2, var) in prog.instr:
                    if op == F_LOADC:
                        off += coff
                    if op in (F_ADD, F_SUB, F_MUL, F_DIV):      # b is an operand only for binary ops
                        b += base
                    q.instr.append((op, a + base if a >= 0 else a, b, off, stride, p, p2, var))
                q.consts.extend(prog.consts)
                q.nconst += prog.nconst
                q.add(F_ADD, a=qroot, b=len(q.instr) - 1)
                return
        self.programs.append(prog)

    # ---- element level ----------------------------------------------------------------------
    def emit(self, prog: _Program, e: Expression) -> int:
        n = prog.nelem
        if isinstance(e, at.Promote) and e.args[0].is_constant() and e.args[0].size == 1:
            return self._scalar_operand(prog, e.args[0])        # broadcast scalar: no n-vector is built
        cv = _const_value(e)
        if cv is not None:
            if cv.size == 1:
                return prog.add(F_LOADC, off=prog.const_vec(cv.reshape(1)), stride=0)
            if cv.size == n:
                return prog.add(F_LOADC, off=prog.const_vec(cv), stride=1)
            raise NotFusable("constant of size %d in a length-%d term" % (cv.size, n))
        if e.size != n:
            raise NotFusable("operand of size %d in a length-%d term" % (e.size, n))
        if isinstance(e, Variable):
            return prog.add(F_LOADV, off=0, stride=1, var=e)
        if isinstance(e, at.index) and isinstance(e.args[0], Variable):
            idx = np.asarray(e._select()).reshape(-1, order="F")
            step = int(idx[1] - idx[0]) if idx.size > 1 else 1
            if idx.size > 1 and not np.array_equal(idx, idx[0] + step * np.arange(idx.size)):
                raise NotFusable("index that is not an arithmetic progression")
            return prog.add(F_LOADV, off=int(idx[0]), stride=step, var=e.args[0])
        if isinstance(e, at.Promote):
            return self.emit(prog, e.args[0]) if e.args[0].size == n else self._scalar_operand(prog, e.args[0])
        if isinstance(e, at.NegExpression):
            return prog.add(F_SCALE, a=self.emit(prog, e.args[0]), p=-1.0)
        if isinstance(e, at.AddExpression):
            acc = None
            shift = 0.0
            for a in e.args:
                if isinstance(a, at.Promote) and a.args[0].is_constant() and a.args[0].size == 1:
                    shift += float(_const_value(a.args[0]).reshape(-1)[0])
                    continue
                c = _const_value(a) if a.size == 1 else None
                if c is not None:
                    shift += float(c.reshape(-1)[0])
                    continue
                if isinstance(a, at.NegExpression) and acc is not None:
                    acc = prog.add(F_SUB, a=acc, b=self.emit(prog, a.args[0]))
                    continue
                r = self.emit(prog, a)
                acc = r if acc is None else prog.add(F_ADD, a=acc, b=r)
            if acc is None:
                raise NotFusable("sum of constants")
            return prog.add(F_ADDC, a=acc, p=shift) if shift != 0.0 else acc
        if isinstance(e, at.DivExpression):
            d = _const_value(e.args[1])
            if d is not None and d.size == 1:
                return prog.add(F_SCALE, a=self.emit(prog, e.args[0]), p=1.0 / float(d.reshape(-1)[0]))
            return prog.add(F_DIV, a=self.emit(prog, e.args[0]), b=self.emit(prog, e.args[1]))
        if isinstance(e, at.MulExpression):          # includes `multiply`
            l, r = e.args
            cl, cr = _const_value(l), _const_value(r)
            if cl is not None and cl.size == 1:
                return prog.add(F_SCALE, a=self.emit(prog, r), p=float(cl.reshape(-1)[0]))
            if cr is not None and cr.size == 1:
                return prog.add(F_SCALE, a=self.emit(prog, l), p=float(cr.reshape(-1)[0]))
            if not isinstance(e, at.multiply) and not (l.size == 1 or r.size == 1 or n == 1):
                raise NotFusable("matrix product inside an elementwise term")
            return prog.add(F_MUL, a=self.emit(prog, l), b=self.emit(prog, r))
        if isinstance(e, at.power):
            p = e.p_rational
            if p == 1:
                return self.emit(prog, e.args[0])
            if p == 0:
                return prog.add(F_LOADC, off=prog.const_vec(np.ones(1)), stride=0)
            return prog.add(F_UNARY, a=self.emit(prog, e.args[0]), b=OP_POWER, p=float(p), p2=float(e.p_value))
        for cls, op in UNARY_OPS.items():
            if type(e) is cls:
                return prog.add(F_UNARY, a=self.emit(prog, e.args[0]), b=op)
        raise NotFusable("atom %s inside an elementwise term" % type(e).__name__)

    def _scalar_operand(self, prog, e):
        cv = _const_value(e)
        if cv is None or cv.size != 1:
            raise NotFusable("promotion of a non-constant scalar")
        return prog.add(F_LOADC, off=prog.const_vec(cv.reshape(1)), stride=0)


def build_fused_spec(problem_min) -> Optional[FusedBuilder]:
    """problem_min: the user's problem in minimisation form.  None when it is not an
    unconstrained elementwise-sum objective."""
    if problem_min.constraints:
        return None
    for v in problem_min.variables():
        if v.attributes.get("nonneg") or v.attributes.get("nonpos") or v.bounds is not None:
            return None
    fb = FusedBuilder()
    try:
        fb.term(problem_min.objective.expr, 1.0)
    except NotFusable:
        return None
    if not fb.programs:
        return None
    return fb


def fused_arrays(fb: FusedBuilder, var_free_base: Dict[int, int], nfree: int) -> Dict[str, np.ndarray]:
    """Resolve variable references to positions in the free-variable vector and flatten the
    programs into tape arrays (`fz_*`, parsed by csrc/fused_obj.h)."""
    op, a, b, off, stride, p, p2 = [], [], [], [], [], [], []
    start, nelem, consts, cbase = [0], [], [], 0
    for prog in fb.programs:
        for (o, ia, ib, of, st, pp, pp2, var) in prog.instr:
            if o == F_LOADV:
                of += var_free_base[var.id]
            elif o == F_LOADC:
                of += cbase
            op.append(o); a.append(ia); b.append(ib); off.append(of); stride.append(st); p.append(pp); p2.append(pp2)
        start.append(len(op))
        nelem.append(prog.nelem)
        consts.extend(prog.consts)
        cbase += prog.nconst
    cc = np.concatenate(consts) if consts else np.zeros(0)
    return {
        "fz_dims": np.array([len(fb.programs), len(op), cc.size, nfree], dtype=np.int64),
        "fz_prog_start": np.asarray(start, dtype=np.int64),
        "fz_prog_nelem": np.asarray(nelem, dtype=np.int64),
        "fz_op": np.asarray(op, dtype=np.int32), "fz_a": np.asarray(a, dtype=np.int32),
        "fz_b": np.asarray(b, dtype=np.int32),
        "fz_off": np.asarray(off, dtype=np.int64), "fz_stride": np.asarray(stride, dtype=np.int64),
        "fz_p": np.asarray(p, dtype=np.float64), "fz_p2": np.asarray(p2, dtype=np.float64),
        "fz_consts": cc.astype(np.float64), "fz_c0": np.array([fb.c0], dtype=np.float64),
    }
