"""TEST INFRASTRUCTURE — numpy restatement of the device tape evaluation (f, ∇f, g, J, ∇²L).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The product path (dnlp_amd/) never does: it calls libdnlp_hip.so and fails loudly without
it.

What is restated.  The arithmetic of the reference's seven `Oracles` callbacks
(reference cvxpy/reductions/solvers/nlp_solvers/nlp_solver.py:212-421) and the per-atom
derivative rules they recurse into (file:line per opcode below), in
the flattened normal form that dnlp_amd/lowering.py produces and the HIP kernels in
dnlp_amd/csrc/ evaluate.  Parity of this file with the reference is pinned by the golden
vectors under tests/golden/ (captured by tools/make_golden.py from the reference itself).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

# opcodes (dnlp_amd/lowering.py, dnlp_amd/csrc/tape.h)
OP_EXP, OP_LOG, OP_ENTR, OP_LOGISTIC, OP_POWER = 1, 2, 3, 4, 5
OP_SIN, OP_COS, OP_TAN, OP_SINH, OP_TANH, OP_ASINH, OP_ATANH, OP_XEXP = 6, 7, 8, 9, 10, 11, 12, 13
OP_MUL, OP_REL_ENTR = 20, 21
OP_QUAD_FORM_DENSE, OP_QUAD_FORM_SPARSE, OP_QUAD_OVER_LIN, OP_MATMUL = 30, 31, 32, 33


def unary_rules(op, u, p_der, p_fwd):
    """(value, first derivative, second derivative) of the smooth elementwise atoms."""
    with np.errstate(all="ignore"):
        if op == OP_EXP:        # reference elementwise/exp.py:34-35,112-121,102-107
            e = np.exp(u)
            return e, e, e
        if op == OP_LOG:        # elementwise/log.py:33-36,118-127,108-113
            return np.log(u), 1.0 / u, -1.0 / (u * u)
        if op == OP_ENTR:       # elementwise/entr.py:35-44,116-120,106-111
            val = np.where(u > 0, -u * np.log(np.where(u > 0, u, 1.0)),
                           np.where(u == 0, 0.0, -np.inf))
            return val, -np.log(u) - 1.0, -1.0 / u
        if op == OP_LOGISTIC:   # elementwise/logistic.py:36-39,108-113,97-103
            e = np.exp(u)
            return np.logaddexp(0, u), e / (1 + e), e / ((1 + e) ** 2)
        if op == OP_POWER:      # elementwise/power.py:187-188 (value), :433-450, :408-422
            return (np.power(u, p_fwd), p_der * np.power(u, p_der - 1),
                    p_der * (p_der - 1) * np.power(u, p_der - 2))
        if op == OP_SIN:        # elementwise/trig.py:33-36,99-103,90-94
            return np.sin(u), np.cos(u), -np.sin(u)
        if op == OP_COS:        # trig.py:113-116,179-183,170-174
            return np.cos(u), -np.sin(u), -np.cos(u)
        if op == OP_TAN:        # trig.py:194-197,261-265,251-256
            c = np.cos(u)
            return np.tan(u), 1.0 / (c * c), 2 * np.tan(u) / (c * c)
        if op == OP_SINH:       # elementwise/hyperbolic.py:33-36,94-98,85-89
            return np.sinh(u), np.cosh(u), np.sinh(u)
        if op == OP_TANH:       # hyperbolic.py:108-111,169-173,160-164
            ch = np.cosh(u)
            return np.tanh(u), 1.0 / (ch * ch), -2 * np.tanh(u) / (ch * ch)
        if op == OP_ASINH:      # hyperbolic.py:183-186,228-232,219-223
            return np.arcsinh(u), 1.0 / np.sqrt(1 + u * u), -u / np.power(1 + u * u, 1.5)
        if op == OP_ATANH:      # hyperbolic.py:242-245,287-291,278-282
            return np.arctanh(u), 1.0 / (1 - u * u), 2 * u / ((1 - u * u) ** 2)
        if op == OP_XEXP:       # elementwise/xexp.py:35-36,108-112,117-121
            e = np.exp(u)
            return u * e, e * (1 + u), e * (2 + u)
    raise ValueError("unknown unary opcode %d" % op)


def _csr(a, name, shape):
    return sp.csr_matrix((a[name + "_val"], a[name + "_idx"], a[name + "_ptr"]), shape=shape)


class TapeEvaluator:
    """Evaluates a tape (dict of named arrays from dnlp_amd.tape.tape_arrays)."""

    def __init__(self, arrays, dense_overrides=None):
        a = self.a = arrays
        d = a["dims"]
        (self.N, self.m, self.Z, self.nseg, self.nd, self.nh, self.nnzJ, self.nnzH,
         self.ndense, self.nsparse, self.nblk, self.coo_complete) = [int(v) for v in d]
        N, m, Z = self.N, self.m, self.Z
        self.G = _csr(a, "G", (m, N + Z))
        self.Mg = _csr(a, "Mg", (N, self.nd))
        self.Mw = _csr(a, "Mw", (Z, 1 + m))
        self.MJ = _csr(a, "MJ", (self.nnzJ, self.nd))
        self.MH = _csr(a, "MH", (self.nnzH, self.nh))
        self.dense = {}
        for k in range(self.ndense):
            n = int(a["dense_n"][k])
            if "dense%d" % k in a:
                self.dense[k] = a["dense%d" % k].reshape((n, n), order="F")
        if dense_overrides:
            self.dense.update(dense_overrides)
        self.sparse = {}
        for k in range(self.nsparse):
            n = None
            for sidx in range(self.nseg):
                if a["seg_op"][sidx] == OP_QUAD_FORM_SPARSE and a["seg_aux"][sidx] == k:
                    n = int(a["seg_n"][sidx])
            self.sparse[k] = (_csr(a, "sp%d" % k, (n, n)), a["sp%d_hr" % k], a["sp%d_hc" % k],
                              a["sp%d_hv" % k])
        self.blocks = a["dense_blocks"].reshape(-1, 6) if self.nblk else np.zeros((0, 6), int)

    def _arg(self, s, which):
        a = self.a
        off, ln = int(a["seg_%s_off" % which][s]), int(a["seg_%s_len" % which][s])
        if off < 0:
            return None
        return a["gidx"][off:off + ln].astype(np.int64)

    def sweep(self, x, w=None):
        """One pass over the segments: z, dvals and (if w is given) hvals."""
        a = self.a
        x = np.asarray(x, dtype=float)
        z = np.zeros(self.Z)
        dv = np.zeros(self.nd)
        hv = np.zeros(self.nh) if w is not None else None
        for s in range(self.nseg):
            op = int(a["seg_op"][s])
            n = int(a["seg_n"][s])
            zo, do, ho = int(a["seg_zoff"][s]), int(a["seg_doff"][s]), int(a["seg_hoff"][s])
            i0 = self._arg(s, "a0")
            i1 = self._arg(s, "a1")
            u = x[i0]
            if op < OP_MUL:
                val, d1, d2 = unary_rules(op, u, a["seg_param"][s], a["seg_param2"][s])
                z[zo:zo + n] = val
                dv[do:do + n] = d1
                if hv is not None:
                    hv[ho:ho + n] = w[zo:zo + n] * d2
            elif op == OP_MUL:
                # binary_operators.py:586-591 (jacobian), :543-546 (hessian cross block)
                v = x[i1]
                z[zo:zo + n] = u * v
                dv[do:do + n] = v
                dv[do + n:do + 2 * n] = u
                if hv is not None:
                    hv[ho:ho + n] = w[zo:zo + n]
            elif op == OP_REL_ENTR:
                # elementwise/rel_entr.py:37-40, :129-148, :150-179
                v = x[i1]
                with np.errstate(all="ignore"):
                    z[zo:zo + n] = u * np.log(u / v)
                    dv[do:do + n] = np.log(u / v) + 1.0
                    dv[do + n:do + 2 * n] = -u / v
                    if hv is not None:
                        ww = w[zo:zo + n]
                        hv[ho:ho + n] = ww / u
                        hv[ho + n:ho + 2 * n] = ww * u / (v * v)
                        hv[ho + 2 * n:ho + 3 * n] = -ww / v
            elif op == OP_QUAD_FORM_DENSE:
                # quad_form.py:41-47 (value), :154-160 (2 P x); Hessian block 2 w P (:143-149)
                P = self.dense[int(a["seg_aux"][s])]
                Pu = P @ u
                z[zo] = u @ Pu
                dv[do:do + n] = 2.0 * Pu
            elif op == OP_QUAD_FORM_SPARSE:
                P, hr, hc, hvv = self.sparse[int(a["seg_aux"][s])]
                z[zo] = u @ (P @ u)
                dv[do:do + n] = (P @ u) + (P.T @ u)
                if hv is not None:
                    hv[ho:ho + hvv.size] = w[zo] * hvv
            elif op == OP_QUAD_OVER_LIN:
                # quad_over_lin.py:38-45, :178-185, :162-173
                y = x[i1][0]
                ss = float(u @ u)
                z[zo] = ss / y
                dv[do:do + n] = 2 * u / y
                dv[do + n] = -ss / (y * y)
                if hv is not None:
                    ww = w[zo]
                    hv[ho:ho + n] = 2 * ww / y
                    hv[ho + n] = 2 * ww * ss / (y ** 3)
                    hv[ho + n + 1:ho + 2 * n + 1] = -2 * ww * u / (y * y)
            elif op == OP_MATMUL:
                # binary_operators.py:309-371 (kron Jacobians), :278-282 (cross Hessian)
                mm, kk, pp = int(a["seg_d0"][s]), int(a["seg_d1"][s]), int(a["seg_d2"][s])
                U = u.reshape((mm, kk), order="F")
                V = x[i1].reshape((kk, pp), order="F")
                z[zo:zo + mm * pp] = (U @ V).reshape(-1, order="F")
                # entry order: (i,j) F-order, l fastest: dU block then dV block
                cnt = mm * pp * kk
                I = np.tile(np.repeat(np.arange(mm), kk), pp)
                J = np.repeat(np.arange(pp), mm * kk)
                L = np.tile(np.arange(kk), mm * pp)
                dv[do:do + cnt] = V[L, J]
                dv[do + cnt:do + 2 * cnt] = U[I, L]
                if hv is not None:
                    hv[ho:ho + cnt] = w[zo + I + J * mm]
            else:
                raise ValueError("unknown opcode %d" % op)
        return z, dv, hv

    # ---- the reference's callback set --------------------------------------------
    def objective(self, x):
        z, _, _ = self.sweep(x)
        return float(self.a["c0"][0] + self.a["c"] @ np.concatenate([x, z]))

    def gradient(self, x):
        _, dv, _ = self.sweep(x)
        return self.a["c"][:self.N] + self.Mg @ dv

    def constraints(self, x):
        z, _, _ = self.sweep(x)
        return self.a["b"] + self.G @ np.concatenate([x, z])

    def jacobianstructure(self):
        return self.a["jac_rows"], self.a["jac_cols"]

    def jacobian(self, x):
        _, dv, _ = self.sweep(x)
        return self.a["Jc"] + self.MJ @ dv

    def hessianstructure(self):
        return self.a["hess_rows"], self.a["hess_cols"]

    def weights(self, lagrange, obj_factor):
        return self.Mw @ np.concatenate([[obj_factor], np.asarray(lagrange, float)])

    def hessian(self, x, lagrange, obj_factor):
        if not self.coo_complete:
            raise ValueError("dense quad_form block too large for a COO Hessian")
        w = self.weights(lagrange, obj_factor)
        _, _, hv = self.sweep(x, w)
        H = self.MH @ hv
        for k, (seg, cid, x0, n, zi, has_pos) in enumerate(self.blocks):
            P = self.dense[int(cid)]
            ii, jj = np.tril_indices(int(n))
            pos = self.a["dense_blk%d_pos" % k]
            if int(has_pos) == 2:                     # contiguous run: base + q
                pos = int(pos[0]) + np.arange(ii.size)
            np.add.at(H, pos, 2.0 * w[int(zi)] * P[ii, jj])
        return H

    def hessian_dense(self, x, lagrange, obj_factor):
        """Full symmetric N x N Hessian of the Lagrangian (for KKT checks)."""
        w = self.weights(lagrange, obj_factor)
        _, _, hv = self.sweep(x, w)
        vals = self.MH @ hv
        H = np.zeros((self.N, self.N))
        r, c = self.a["hess_rows"], self.a["hess_cols"]
        H[r, c] = vals
        for k, (seg, cid, x0, n, zi, has_pos) in enumerate(self.blocks):
            P = self.dense[int(cid)]
            x0, n = int(x0), int(n)
            if has_pos:
                H[x0:x0 + n, x0:x0 + n] += np.tril(2.0 * w[int(zi)] * P)
            else:
                H[x0:x0 + n, x0:x0 + n] += np.tril(2.0 * w[int(zi)] * P)
        return H + np.tril(H, -1).T
