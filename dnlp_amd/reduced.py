"""Reduced-space structure of a canonicalised problem whose constraints only DEFINE auxiliary
variables (BASELINE config C2: "unconstrained ... tape f / grad f eval + line search only").

dnlp2smooth turns an unconstrained smooth objective such as the Rosenbrock chain into
`min f(x, t)  s.t.  t_k == expr_k(x, t_<k)` (reference dnlp2smooth.py:42-111 and the `t == arg`
rules of Appendix B).  Those equalities are explicit definitions: given the user's variables the
auxiliary ones follow by forward substitution, and the gradient of the reduced objective by one
adjoint substitution.  Both are expressed with the SAME tape kernels (sweep, G spmv, J^T
products) as fixed-point passes whose count is the nesting depth, so the device evaluates
f / grad f of the user's unconstrained problem without any linear algebra.

`reduction_arrays` returns the extra tape arrays, or None when the problem is not of this form
(user constraints, epigraph inequalities, bilinear `z*y == f` definitions ...).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .constraints import Zero
from .expressions import Variable


def reduction_arrays(new_problem, tape, user_var_ids):
    """new_problem: the lowered canonical problem (constraints already `expr == 0` / `>= 0`).
    user_var_ids: set of id() of the variables of the problem the user wrote."""
    variables = new_problem.variables()
    offsets, off = {}, 0
    for v in variables:
        offsets[id(v)] = off
        off += v.size
    N, m = tape.N, tape.m
    if m == 0:
        return {"def_var": np.zeros(0, np.int32), "free_idx": np.arange(N, dtype=np.int32),
                "red_depth": np.array([0], np.int64)}
    def_var = np.full(m, -1, dtype=np.int64)
    row = 0
    for c in new_problem.constraints:
        size = c.size
        if not isinstance(c, Zero):
            return None
        e = c.args[0]
        # lower_equality builds  lhs - rhs = AddExpression([t, -(arg)])  with t a bare Variable
        lhs = e.args[0] if getattr(e, "args", None) else None
        if not isinstance(lhs, Variable) or id(lhs) in user_var_ids or lhs.size != size:
            return None
        if any(v is lhs for v in e.args[1].variables()) if len(e.args) > 1 else True:
            return None
        def_var[row:row + size] = offsets[id(lhs)] + np.arange(size)
        row += size
    if np.any(def_var < 0):
        return None
    is_def = np.zeros(N, bool)
    is_def[def_var] = True
    if int(is_def.sum()) != m:                     # two rows define the same variable
        return None
    free = np.nonzero(~is_def)[0]
    is_user = np.zeros(N, bool)
    for v in variables:
        if id(v) in user_var_ids:
            is_user[offsets[id(v)]:offsets[id(v)] + v.size] = True
    if not np.array_equal(is_user, ~is_def):
        return None
    # the defining row of t must carry coefficient +1 on t; nesting depth from the dependency pattern
    # of the Jacobian
    G = tape.G.tocsr()
    rows_of = np.repeat(np.arange(m), np.diff(G.indptr))
    on_def = G.indices == def_var[rows_of]
    diag = np.zeros(m)
    diag[rows_of[on_def]] = G.data[on_def]
    if not np.allclose(diag, 1.0):
        return None
    level = np.zeros(N, dtype=np.int64)
    jr = np.asarray(tape.jac_rows, dtype=np.int64)
    jc = np.asarray(tape.jac_cols, dtype=np.int64)
    keep = jc != def_var[jr]
    dep_rows, dep_cols = def_var[jr[keep]], jc[keep]
    depth = 0
    for _ in range(256):
        new = np.zeros(N, dtype=np.int64)
        if dep_rows.size:
            np.maximum.at(new, dep_rows, level[dep_cols] + 1)
        new[~is_def] = 0
        new = np.maximum(new, np.where(is_def, 1, 0))
        if np.array_equal(new, level):
            break
        level = new
        depth += 1
    else:
        return None        # cyclic definitions
    return {"def_var": def_var.astype(np.int32), "free_idx": free.astype(np.int32),
            "red_depth": np.array([int(level.max())], np.int64)}
