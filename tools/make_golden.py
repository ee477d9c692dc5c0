"""Generate tests/golden/*.npz from the REFERENCE (runs only in the build container).

For every problem in tests/problem_zoo.py: build it with the reference's cvxpy, run the
reference's own reduction chain (FlipObjective -> CvxAttr2Constr -> Dnlp2Smooth ->
NLPsolver.apply, problems/problem.py:1220-1243) and record what its `Oracles` return:
N, m, per-variable (shape, offset), x0, lb, ub, cl, cu, Jacobian / Hessian structures and,
at K seeded points inside the bounds, f, grad f, g, Jacobian values, Hessian values for
seeded (lambda, sigma).  The fixtures are data only; no reference source travels.

    python tools/make_golden.py            # rewrites tests/golden/
    python tools/make_golden.py nb_power_flow nb_path_planning    # only these
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "tests"))

from ref_import import import_reference, ref_reductions  # noqa: E402

K_POINTS = 3


def sample_point(rng, x0, lb, ub):
    x = x0 + 0.05 * rng.standard_normal(x0.size)
    lo = np.where(np.isfinite(lb), lb + 0.02, -np.inf)
    hi = np.where(np.isfinite(ub), ub - 0.02, np.inf)
    bad = lo > hi
    mid = 0.5 * (np.where(np.isfinite(lb), lb, 0) + np.where(np.isfinite(ub), ub, 0))
    x = np.minimum(np.maximum(x, lo), hi)
    x[bad] = mid[bad]
    return x


def main():
    cp = import_reference()
    from problem_zoo import GOLDEN_ZOO
    out_dir = os.path.join(HERE, "..", "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    only = set(sys.argv[1:])
    for name, builder in GOLDEN_ZOO.items():
        if only and name not in only:
            continue
        prob = builder(cp)
        # the reference's chain (problems/problem.py:1220-1243), reduction by reduction, so that the
        # smooth problem NLPsolver.apply receives (whose variables() order its Bounds use) is at hand
        smooth = prob
        for red in ref_reductions(cp, prob)[:-1]:
            smooth, _ = red.apply(smooth)
        data, inv = ref_reductions(cp, prob)[-1].apply(smooth)
        o = data["oracles"]
        x0 = np.asarray(data["x0"], dtype=float)
        N, m = x0.size, len(data["cl"])
        jr, jc = o.jacobianstructure()
        hr, hc = o.hessianstructure()
        rec = {
            "N": N, "m": m, "x0": x0, "lb": data["lb"], "ub": data["ub"],
            "cl": data["cl"], "cu": data["cu"],
            "jac_rows": np.asarray(jr, np.int32), "jac_cols": np.asarray(jc, np.int32),
            "hess_rows": np.asarray(hr, np.int32), "hess_cols": np.asarray(hc, np.int32),
            "var_sizes": np.array([v.size for v in data["problem"].variables()], np.int64),
            "var_ndims": np.array([v.ndim for v in data["problem"].variables()], np.int64),
            "is_max": int(type(prob.objective) == cp.Maximize),
        }
        # The reference lays lb / ub / x0 out in the order of the PRE-lowering problem's variables
        # (nlp_solver.py:84,116,163) but evaluates its oracles in the lowered problem's order (:200).
        # bounds_perm[k] = position, in the bounds order, of the k-th variable of the oracle order, so
        # that the per-variable bounds can be compared whatever flat order an implementation uses.
        pre_ids = [v.id for v in smooth.variables()]
        rec["bounds_var_sizes"] = np.array([v.size for v in smooth.variables()], np.int64)
        rec["bounds_perm"] = np.array([pre_ids.index(v.id) for v in data["problem"].variables()], np.int64)
        rng = np.random.default_rng(0)
        for k in range(K_POINTS):
            x = sample_point(rng, x0, np.asarray(data["lb"]), np.asarray(data["ub"])) if k else x0.copy()
            lam = rng.standard_normal(m)
            sigma = float(rng.uniform(0.5, 1.5))
            rec["x_%d" % k] = x
            rec["lam_%d" % k] = lam
            rec["sigma_%d" % k] = sigma
            rec["f_%d" % k] = float(o.objective(x))
            rec["grad_%d" % k] = np.array(o.gradient(x), dtype=float).copy()
            rec["g_%d" % k] = np.asarray(o.constraints(x), dtype=float) if m else np.zeros(0)
            rec["jac_%d" % k] = np.asarray(o.jacobian(x), dtype=float).ravel() if m else np.zeros(0)
            rec["hess_%d" % k] = np.asarray(o.hessian(x, lam, sigma), dtype=float).ravel()
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
        print("%-20s N=%-5d m=%-5d nnzJ=%-6d nnzH=%-6d" % (name, N, m, len(jr), len(hr)))


if __name__ == "__main__":
    main()
