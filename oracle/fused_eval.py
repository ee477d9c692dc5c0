"""TEST INFRASTRUCTURE — numpy interpreter of the fused element programs (dnlp_amd/fused.py).

Only tests/ may import this.  It restates the forward / reverse sweep that
dnlp_amd/csrc/fused_obj.h runs per element, vectorised over the elements, with the unary atom
rules of oracle/tape_eval.py (which cite the reference's atom files).
"""
import numpy as np

from .tape_eval import unary_rules

F_LOADV, F_LOADC, F_UNARY, F_ADD, F_SUB, F_MUL, F_SCALE, F_ADDC, F_DIV = range(9)


def numpy_eval(arrays, xfree):
    """Reference interpreter of the program arrays (tests): returns (f, grad)."""
    xfree = np.asarray(xfree, dtype=float)
    grad = np.zeros_like(xfree)
    f = float(arrays["fz_c0"][0])
    for pidx in range(int(arrays["fz_dims"][0])):
        s0, s1 = int(arrays["fz_prog_start"][pidx]), int(arrays["fz_prog_start"][pidx + 1])
        n = int(arrays["fz_prog_nelem"][pidx])
        i = np.arange(n)
        vals, idxs = {}, {}
        for k in range(s0, s1):
            o, ia, ib = int(arrays["fz_op"][k]), int(arrays["fz_a"][k]) + s0, int(arrays["fz_b"][k]) + s0
            of, st, p, p2 = int(arrays["fz_off"][k]), int(arrays["fz_stride"][k]), arrays["fz_p"][k], arrays["fz_p2"][k]
            if o == F_LOADV:
                idxs[k] = of + st * i
                vals[k] = xfree[idxs[k]]
            elif o == F_LOADC:
                vals[k] = arrays["fz_consts"][of + st * i]
            elif o == F_UNARY:
                vals[k] = unary_rules(int(arrays["fz_b"][k]), vals[ia], p, p2)[0]
            elif o == F_ADD:
                vals[k] = vals[ia] + vals[ib]
            elif o == F_SUB:
                vals[k] = vals[ia] - vals[ib]
            elif o == F_MUL:
                vals[k] = vals[ia] * vals[ib]
            elif o == F_SCALE:
                vals[k] = p * vals[ia]
            elif o == F_ADDC:
                vals[k] = vals[ia] + p
            elif o == F_DIV:
                vals[k] = vals[ia] / vals[ib]
        f += float(np.sum(vals[s1 - 1]))
        adj = {s1 - 1: np.ones(n)}
        for k in range(s1 - 1, s0 - 1, -1):
            g = adj.get(k)
            if g is None:
                continue
            o, ia, ib = int(arrays["fz_op"][k]), int(arrays["fz_a"][k]) + s0, int(arrays["fz_b"][k]) + s0
            p, p2 = arrays["fz_p"][k], arrays["fz_p2"][k]
            if o == F_LOADV:
                np.add.at(grad, idxs[k], g)
            elif o == F_UNARY:
                adj[ia] = g * unary_rules(int(arrays["fz_b"][k]), vals[ia], p, p2)[1]
            elif o == F_ADD:
                adj[ia] = g; adj[ib] = g
            elif o == F_SUB:
                adj[ia] = g; adj[ib] = -g
            elif o == F_MUL:
                adj[ia] = g * vals[ib]; adj[ib] = g * vals[ia]
            elif o == F_SCALE:
                adj[ia] = g * p
            elif o == F_ADDC:
                adj[ia] = g
            elif o == F_DIV:
                adj[ia] = g / vals[ib]; adj[ib] = -g * vals[ia] / (vals[ib] ** 2)
    return f, grad
