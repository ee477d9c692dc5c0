"""Single-launch batch path (csrc/batch.h, dnlp_amd.batch.ParametricBatch / solve_batch):
BASELINE config C5.  CPU: the affine parameter -> tape-data map against direct lowering.
GPU: every instance of a batch against the CPU oracle solving that instance's own tape."""
import numpy as np
import pytest

import batch_problems as bp
from dnlp_amd.batch import (BATCH_DATA_KEYS, ParametricBatch, arrays_with_data, instance_data,
                            lower_arrays, same_structure)

TEMPLATES = {"localization": bp.template_localization, "circle_packing": bp.template_circle_packing}


@pytest.mark.parametrize("name", sorted(TEMPLATES))
def test_parameter_to_tape_data_map_is_affine_and_exact(name):
    prob, params, sample, _ = TEMPLATES[name]()
    pb = ParametricBatch(prob, params)
    assert pb.affine
    thetas = np.stack([sample(i) for i in range(12)])
    mat = pb.data(thetas)
    assert mat.shape == (12, pb.d0.size)
    for i in (0, 5, 11):
        pb._set(thetas[i])
        a, _, _, _ = lower_arrays(prob)
        assert same_structure(pb.arrays0, a)
        d = instance_data(a)
        fin = np.isfinite(d)
        assert np.array_equal(d[~fin], mat[i][~fin])
        np.testing.assert_allclose(mat[i][fin], d[fin], rtol=1e-13, atol=1e-13)
        back = arrays_with_data(pb.arrays0, mat[i])
        for k in BATCH_DATA_KEYS:
            assert back[k].shape == a[k].shape


def test_non_affine_parameter_use_falls_back_to_lowering():
    import dnlp_amd as cp
    p = cp.Parameter(2, name="p", value=np.array([1.0, 2.0]))
    x = cp.Variable(2)
    prob = cp.Problem(cp.Minimize(cp.sum_squares(x - cp.multiply(p, p))), [cp.sum(cp.exp(x)) <= 50])
    pb = ParametricBatch(prob, [p])
    assert not pb.affine
    mat = pb.data(np.array([[1.0, 2.0], [3.0, 0.5]]))
    p.value = np.array([3.0, 0.5])
    a, _, _, _ = lower_arrays(prob)
    np.testing.assert_allclose(mat[1][np.isfinite(mat[1])], instance_data(a)[np.isfinite(mat[1])])


def _oracle(arrays, opts=None, check_point=None):
    """The host build's solve of one instance; with `check_point` also the instance's constraint functions at that
    point (key "g_at_point": how far another optimum of the same objective value is checked for feasibility)."""
    from dnlp_amd.nlp_solver import HIPNLP
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    orc = OracleProblem(serialize(arrays))
    o = dict(HIPNLP.DEFAULT_OPTIONS)
    o.update(opts or {})
    for k, v in o.items():
        orc.set_option(k, v)
    info = orc.solve(arrays["x0"])
    if check_point is not None:
        info["g_at_point"] = np.array(orc.eval_g(np.asarray(check_point, dtype=float)))
    return info


@pytest.mark.gpu
@pytest.mark.parametrize("name,batch", [("localization", 96), ("circle_packing", 64)])
def test_batch_kernel_execution_space_agreement_per_instance(name, batch, gpu_required):
    """Execution-space agreement (the oracle is the host build of the same algorithm text; reference-
    held answers for the batch path are in test_full_size_configs.py): every instance must land on the oracle's
    optimum (1e-6 relative objective, 1e-5 primal point) with the oracle's status, and (up to
    reduction order) in the same number of iterations."""
    prob, params, sample, var = TEMPLATES[name]()
    pb = ParametricBatch(prob, params)
    thetas = np.stack([sample(i) for i in range(batch)])
    res = pb.solve(thetas, want_duals=True)
    mat = pb.data(thetas)
    assert res.x.shape == (batch, pb.arrays0["dims"][0])
    same_iters = 0
    other_kkt_point = 0
    loose_circle = 0
    for i in range(batch):
        oi = _oracle(arrays_with_data(pb.arrays0, mat[i]))
        assert res.status[i] == oi["status"]
        if oi["status"] != 0:
            continue
        if name == "circle_packing" and abs(res.raw["obj_val"][i] - oi["obj_val"]) > 1e-6 * max(1.0, abs(oi["obj_val"])):
            # non-convex: on a long path (dozens of iterations) summation order can tip an instance
            # into another local optimum; such an instance must still be a KKT point (status 0 above)
            # and there may only be a few of them
            other_kkt_point += 1
            continue
        assert abs(res.raw["obj_val"][i] - oi["obj_val"]) <= 1e-6 * max(1.0, abs(oi["obj_val"]))
        if name == "circle_packing" and not np.allclose(res.x[i], oi["x"], rtol=1e-5, atol=1e-6):
            # a circle that touches nothing can slide: the optimum is a face, and with the barrier going down
            # to IPOPT's 1e-11 the two builds' rounding picks different points of it.  Same objective (above) and
            # feasible for the instance's own constraint functions and bounds = another point of the same optimal face
            # (how far it slid is not a property of the solver: 0.0085 .. 0.066 seen across builds).
            a = arrays_with_data(pb.arrays0, mat[i])
            g = _oracle(a, check_point=res.x[i])["g_at_point"]
            assert np.all(g >= a["cl"] - 1e-6) and np.all(g <= a["cu"] + 1e-6)
            assert np.all(res.x[i] >= a["lb"] - 1e-8) and np.all(res.x[i] <= a["ub"] + 1e-8)
            assert np.max(np.abs(res.x[i] - oi["x"])) <= 0.25
            loose_circle += 1
            same_iters += int(res.iterations[i] == oi["iterations"])
            continue
        np.testing.assert_allclose(res.x[i], oi["x"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(res.raw["mult_g"][i], oi["mult_g"], rtol=1e-4, atol=1e-5)
        same_iters += int(res.iterations[i] == oi["iterations"])
    assert other_kkt_point <= batch // 16 and loose_circle <= batch // 8
    assert same_iters >= int(0.85 * batch)
    assert np.sum(res.status == 0) >= batch - 1
    vals = res.value_of(var)
    assert vals.shape == (batch,) + tuple(var.shape)


@pytest.mark.gpu
def test_batch_kernel_large_instances_four_wavefronts(gpu_required):
    """KKT order 1636 (path planning): the 256-lane form of the kernel (workgroup Bunch-Kaufman,
    matrix in global memory).  Instance 0 is the notebook's problem: published optimum."""
    prob, params, sample, x = bp.template_path_planning()
    pb = ParametricBatch(prob, params)
    assert pb.affine
    thetas = np.stack([sample(i) for i in range(3)])
    res = pb.solve(thetas)
    assert np.all(res.status == 0)
    assert abs(res.obj_val[0] - 1.3136882319337619e+01) <= 1e-6 * 13.14      # path_planning.ipynb:103
    oi = _oracle(arrays_with_data(pb.arrays0, pb.data(thetas)[2]))
    assert oi["status"] == 0
    assert abs(res.raw["obj_val"][2] - oi["obj_val"]) <= 1e-6 * abs(oi["obj_val"])
    np.testing.assert_allclose(res.x[2], oi["x"], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_power_flow_batch_base_case_is_the_published_optimum(gpu_required):
    """9-bus OPF with the loads as parameters: instance 0 carries the notebook's loads and must reach
    the published IPOPT objective (power_flow.ipynb:101); scaled-load instances match the oracle."""
    prob, params, sample, p = bp.template_power_flow()
    pb = ParametricBatch(prob, params)
    assert pb.affine
    thetas = np.stack([sample(i) for i in range(16)])
    res = pb.solve(thetas, least_square_init_duals="no")
    assert np.all(res.status == 0)
    assert abs(res.obj_val[0] - 3.0878422284732592e+03) <= 1e-6 * 3.0878e3
    mat = pb.data(thetas)
    oi = _oracle(arrays_with_data(pb.arrays0, mat[5]), {"least_square_init_duals": "no"})
    assert oi["status"] == 0 and abs(res.raw["obj_val"][5] - oi["obj_val"]) <= 1e-6 * abs(oi["obj_val"])
    # heavier load -> higher generation cost
    order = np.argsort(thetas[:, :3].sum(axis=1))
    assert res.obj_val[order[-1]] > res.obj_val[order[0]]


@pytest.mark.gpu
def test_localization_batch_recovers_true_positions(gpu_required):
    """Noise-free ranges: the optimum is the true position (test_nlp_solvers.py:175-189 analogue)."""
    prob, params, sample, x = bp.template_localization()
    pb = ParametricBatch(prob, params)
    ids = np.arange(200)
    res = pb.solve(np.stack([sample(i) for i in ids]))
    ok = res.status == 0
    assert ok.sum() >= 198
    truth = np.stack([np.random.default_rng(i).uniform(-3, 3, 2) for i in ids])
    got = res.value_of(x)
    # a few instances end in a different local minimum of the non-convex range fit
    good = np.linalg.norm(got - truth, axis=1) < 1e-4
    assert good[ok].mean() > 0.8
    assert np.all(np.abs(res.obj_val[ok & good]) < 1e-8)


@pytest.mark.gpu
def test_solve_batch_generic_path_equals_template_path(gpu_required):
    from dnlp_amd.batch import solve_batch
    ids = list(range(24))
    res_g = solve_batch([bp.build_localization(i) for i in ids])
    prob, params, sample, x = bp.template_localization()
    res_t = ParametricBatch(prob, params).solve(np.stack([sample(i) for i in ids]))
    assert np.array_equal(res_g.status, res_t.status)
    np.testing.assert_allclose(res_g.obj_val, res_t.obj_val, rtol=0, atol=1e-9)
    np.testing.assert_allclose(res_g.x, res_t.x, rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
def test_batch_rejects_wrong_stride_and_dense_blocks(gpu_required):
    import dnlp_amd as cp
    from dnlp_amd.batch import _device_handle
    prob, params, sample, _ = bp.template_localization()
    pb = ParametricBatch(prob, params)
    h = _device_handle(pb.arrays0, pb.data0["tape"], 0, {})
    with pytest.raises(ValueError):
        h.solve_batch(np.zeros((2, pb.d0.size + 1)))
    h.close()
    n = 60                                                   # dense quad_form block (n > 48)
    A = np.random.default_rng(0).standard_normal((n, n))
    xq = cp.Variable(n)
    xq.value = np.ones(n) / np.sqrt(n)
    q = cp.Problem(cp.Minimize(-cp.quad_form(xq, A @ A.T)), [cp.sum_squares(xq) == 1])
    arrays, data, _, _ = lower_arrays(q)
    hq = _device_handle(arrays, data["tape"], 0, {})
    with pytest.raises(RuntimeError):
        hq.solve_batch(instance_data(arrays)[None, :])
    hq.close()


@pytest.mark.gpu
def test_best_of_multistart_is_one_batched_launch(gpu_required):
    """Problem.solve(nlp=True, best_of=N) (reference problem.py:1256-1269): the N starts go through
    dnlp_solve_batch; per-start objectives and the selected optimum equal the serial loop's."""
    import dnlp_amd as cp

    def build():
        rng = np.random.default_rng(3)
        n = 4
        radius = rng.uniform(1.0, 3.0, n)
        centers = cp.Variable((n, 2), name="c")
        cons = []
        for i in range(n - 1):
            cons += [cp.sum((centers[i, :] - centers[i + 1:, :]) ** 2, axis=1) >= (radius[i] + radius[i + 1:]) ** 2]
        centers.sample_bounds = [-5.0, 5.0]
        return cp.Problem(cp.Minimize(cp.max(cp.norm_inf(centers, axis=1) + radius)), cons), centers

    # a solved variable counts as user-initialised in later best_of calls (reference
    # problem.py:1650-1656), so each mode gets a fresh problem
    prob, centers = build()
    np.random.seed(0)
    prob.solve(nlp=True, best_of=12)
    objs_b = np.array(prob.solver_stats.extra_stats["all_objs_from_best_of"])
    val_b, c_b = prob.value, centers.value.copy()
    prob, centers = build()
    np.random.seed(0)
    prob.solve(nlp=True, best_of=12, batch=False)
    objs_s = np.array(prob.solver_stats.extra_stats["all_objs_from_best_of"])
    assert len(set(np.round(objs_s, 6))) > 1                 # the starts really differ
    np.testing.assert_allclose(objs_b, objs_s, rtol=1e-6, atol=1e-8)
    assert abs(val_b - prob.value) <= 1e-6 * abs(val_b)
    np.testing.assert_allclose(c_b, centers.value, rtol=1e-5, atol=1e-5)
    # prob.value is the solver's (epigraph) objective, all_objs the original objective at the point
    assert abs(val_b - objs_b.min()) <= 1e-6 * abs(val_b)


@pytest.mark.gpu
def test_small_silent_solves_run_the_loop_on_the_device(gpu_required):
    """Problem.solve(nlp=True) of a small problem = batch of one (no host round trips); the host-
    driven loop (device_loop='host', or verbose) gives the same answer."""
    import dnlp_amd as cp
    from problem_zoo import hs071, localization
    for build in (hs071, localization):
        pa, pb_ = build(cp), build(cp)
        pa.solve(nlp=True)
        pb_.solve(nlp=True, device_loop="host")
        assert pa.status == pb_.status == "optimal"
        assert abs(pa.value - pb_.value) <= 1e-7 * max(1.0, abs(pb_.value))
        for va, vb in zip(pa.variables(), pb_.variables()):
            np.testing.assert_allclose(va.value, vb.value, rtol=1e-5, atol=1e-6)
        assert pa.solver_stats.num_iters == pb_.solver_stats.num_iters


@pytest.mark.gpu
def test_batch_status_codes_and_options(gpu_required):
    """IPOPT status integers per instance: an infeasible instance and an iteration limit."""
    import dnlp_amd as cp
    from dnlp_amd.batch import solve_batch
    def make(ub):
        x = cp.Variable(2)
        x.value = np.array([0.5, 0.5])
        return cp.Problem(cp.Minimize(cp.sum(cp.exp(x))), [cp.sum(cp.square(x)) >= 4.0, x <= ub, x >= -ub])
    res = solve_batch([make(3.0), make(0.5), make(2.5)])
    assert res.status[0] == 0 and res.status[2] == 0
    assert res.status[1] in (2, -2)                      # Infeasible_Problem_Detected / Restoration_Failed
    res2 = solve_batch([make(3.0), make(2.5)], max_iter=2)
    assert np.all(res2.status == -1) and np.all(res2.iterations == 2)
    res3 = solve_batch([make(3.0)], mu_strategy="monotone")
    assert res3.status[0] == 0 and abs(res3.obj_val[0] - res.obj_val[0]) <= 1e-6


@pytest.mark.gpu
def test_instance_data_generated_on_the_device_equals_host_rows(gpu_required):
    """Affine templates hand the parameter -> data map to the device once (dnlp_batch_set_affine_map) and a
    solve moves only the parameter rows (dnlp_solve_batch_theta).  Same instances through the host-generated
    rows (dnlp_solve_batch): same statuses and iteration counts, solutions equal to rounding of the row
    generation (d0 + D dtheta summed in another order)."""
    for tmpl, nb in ((bp.template_localization, 300), (lambda: bp.template_circle_packing(4), 200)):
        prob, params, sample, var = tmpl()
        pb = ParametricBatch(prob, params)
        assert pb.affine
        thetas = np.stack([sample(i) for i in range(nb)])
        res_dev = pb.solve(thetas, want_duals=True)
        assert pb._map_on_device
        raw = pb._handle.solve_batch(pb.data(thetas), want_duals=True)
        assert np.array_equal(res_dev.status, raw["status"])
        assert np.mean(res_dev.iterations == raw["iterations"]) >= 0.97
        same = res_dev.iterations == raw["iterations"]
        np.testing.assert_allclose(res_dev.raw["obj_val"][same], raw["obj_val"][same], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(res_dev.x[same], raw["x"][same], rtol=1e-6, atol=1e-7)
        pb.close()


@pytest.mark.gpu
def test_best_of_64_on_a_dense_kkt_of_order_600_is_one_launch(gpu_required, capsys):
    """best_of on a mid-size DENSE problem (reference problem.py:1256-1269): 400 variables under 200 dense equality
    rows — KKT order 600, no sparsity to plan — runs its 64 starts as one batched launch (four wavefronts per
    instance, the factorisation by the workgroup on the instance's matrix in global memory); objectives per start
    and the selected optimum equal the serial loop's."""
    import dnlp_amd as cp

    def build():
        rng = np.random.default_rng(5)
        n, m = 400, 200
        A = rng.standard_normal((m, n))
        xs = rng.uniform(-1.0, 1.0, n)
        x = cp.Variable(n, name="x")
        x.sample_bounds = [-2.0, 2.0]
        obj = cp.sum(cp.power(x, 4)) - 3.0 * cp.sum(cp.square(x))          # a double well per coordinate
        return cp.Problem(cp.Minimize(obj), [A @ x == A @ xs]), x

    prob, x = build()
    np.random.seed(1)
    prob.solve(nlp=True, best_of=64)
    objs_b = np.array(prob.solver_stats.extra_stats["all_objs_from_best_of"])
    assert "Solving 64 NLP starts in one batched launch" in capsys.readouterr().out
    val_b = prob.value
    prob, x = build()
    np.random.seed(1)
    prob.solve(nlp=True, best_of=64, batch=False)
    objs_s = np.array(prob.solver_stats.extra_stats["all_objs_from_best_of"])
    assert len(set(np.round(objs_s, 5))) > 4                 # the starts end in different local minima
    np.testing.assert_allclose(objs_b, objs_s, rtol=1e-6, atol=1e-7)
    assert abs(val_b - prob.value) <= 1e-6 * max(1.0, abs(val_b))


@pytest.mark.gpu
@pytest.mark.parametrize("nv", [200, 600, 900])
def test_dense_instances_above_order_256_in_one_launch(nv, gpu_required):
    """Dense batch instances of KKT order 300 / 900 / 1350 (two, four and six rows per lane of the in-workgroup
    panel-blocked Bunch-Kaufman, csrc/exec_block.h bk_panels): each instance of a two-start launch ends where the
    host-driven solve of the same start ends."""
    import dnlp_amd as cp
    from dnlp_amd.batch import _device_handle, instance_data
    rng = np.random.default_rng(nv)
    m = nv // 2
    A = rng.standard_normal((m, nv))
    xs = rng.uniform(-1.0, 1.0, nv)
    x = cp.Variable(nv, name="x")
    x.sample_bounds = [-2.0, 2.0]
    prob = cp.Problem(cp.Minimize(cp.sum(cp.power(x, 4)) - 3.0 * cp.sum(cp.square(x))), [A @ x == A @ xs])
    chain = prob._build_chain(None)
    np.random.seed(nv)
    rows, starts = [], []
    for run in range(2):
        prob.set_random_NLP_initial_point(run)
        starts.append(x.value.copy())
        data, _ = chain.apply(prob, make_handle=False)
        rows.append(instance_data(data["tape_arrays"]))
    h = _device_handle(data["tape_arrays"], data["tape"], None, {"print_level": 0})
    raw = h.solve_batch(np.stack(rows), want_duals=True)
    h.close()
    assert list(raw["status"]) == [0, 0]
    for k in range(2):
        x.value = starts[k]
        prob.solve(nlp=True)
        assert abs(raw["obj_val"][k] - prob.value) <= 1e-6 * max(1.0, abs(prob.value)), (k, raw["obj_val"][k], prob.value)
        np.testing.assert_allclose(A @ raw["x"][k][:nv], A @ xs, rtol=0, atol=1e-6)


@pytest.mark.gpu
def test_solve_many_overlaps_launches_and_changes_no_result(gpu_required):
    """ParametricBatch.solve_many: six batches of 512 localization instances with two launches in flight (one handle,
    dnlp_batch_stream_* with two slots inside the library) give, batch by batch, the bits of six solve() calls one after
    the other."""
    import batch_problems as bp
    from dnlp_amd.batch import ParametricBatch
    prob, params, sample, _ = bp.template_localization()
    pb = ParametricBatch(prob, params)
    batches = [np.stack([sample(1000 * k + i) for i in range(512)]) for k in range(6)]
    one_by_one = [pb.solve(t) for t in batches]
    overlapped = pb.solve_many(batches, in_flight=2)
    assert len(overlapped) == 6
    for a, b in zip(one_by_one, overlapped):
        assert np.array_equal(a.status, b.status) and np.array_equal(a.iterations, b.iterations)
        assert np.array_equal(a.obj_val, b.obj_val) and np.array_equal(a.x, b.x)
    pb.close()


@pytest.mark.gpu
def test_batch_stream_tickets_and_refusals(gpu_required):
    """dnlp_batch_stream_* at its edges (include/dnlp_hip.h): more submissions than slots (the first slot to finish takes the
    next), batches of different sizes incl. an empty one through the same stream, a ticket that was never issued, results
    waited for out of order and a ticket waited for twice, parameter rows of the wrong width (refused at wait, the stream
    stays usable), and a stream on a handle without an affine map (refused at create)."""
    import ctypes as C
    import batch_problems as bp
    from dnlp_amd import _capi
    from dnlp_amd.batch import ParametricBatch
    prob, params, sample, _ = bp.template_localization()
    pb = ParametricBatch(prob, params)
    sizes = [64, 0, 7, 300, 1]
    batches = [np.stack([sample(50 * k + i) for i in range(n)]).reshape(n, -1) if n else np.zeros((0, 22)) for k, n in enumerate(sizes)]
    P = batches[0].shape[1]
    batches[1] = np.zeros((0, P))
    ref = [pb.solve(t) for t in batches]
    many = pb.solve_many(batches, in_flight=2)
    for a, b in zip(ref, many):
        assert a.x.shape == b.x.shape and np.array_equal(a.x, b.x) and np.array_equal(a.status, b.status)
    # the raw entry points
    h = pb._handle
    api = h.api
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))       # noqa: E731
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))          # noqa: E731
    st = api.batch_stream_create(h.ptr, 2)
    assert st
    n = 16
    th = np.ascontiguousarray(batches[0][:n])
    outs = [dict(x=np.zeros((n, h.n)), obj=np.zeros(n), st=np.zeros(n, np.int32), it=np.zeros(n, np.int32), nf=np.zeros(n, np.int32)) for _ in range(3)]

    def submit(o, theta, width):
        return api.batch_stream_submit(st, n, dp(theta), width, dp(o["x"]), dp(o["obj"]), None, None, None, ip(o["st"]), ip(o["it"]), ip(o["nf"]))

    sec = C.c_double()
    secp = C.cast(C.byref(sec), C.POINTER(C.c_double))
    t0, t1, t2 = submit(outs[0], th, P), submit(outs[1], th, P), submit(outs[2], th, P)
    assert (t0, t1, t2) == (0, 1, 2)
    assert api.batch_stream_wait(st, 7, secp) == -1 and "no such ticket" in api.error()
    # three submissions through two slots: the third took whichever slot finished first; every result is on record until it
    # is waited for — in any order, however many submissions came after it — and is handed out once
    assert api.batch_stream_wait(st, t2, secp) == 0 and api.batch_stream_wait(st, t0, secp) == 0 and api.batch_stream_wait(st, t1, secp) == 0
    assert np.array_equal(outs[1]["x"], ref[0].x[:n]) and np.array_equal(outs[2]["x"], ref[0].x[:n])
    assert np.array_equal(outs[0]["x"], ref[0].x[:n])
    assert api.batch_stream_wait(st, t0, secp) == -2 and "waited for already" in api.error()
    bad = np.zeros((n, P + 1))
    t3 = submit(outs[0], bad, P + 1)
    assert t3 == 3 and api.batch_stream_wait(st, t3, secp) != 0 and api.error()
    t4 = submit(outs[0], th, P)
    assert api.batch_stream_wait(st, t4, secp) == 0 and np.array_equal(outs[0]["x"], ref[0].x[:n])
    api.batch_stream_destroy(st)
    pb.close()
    from dnlp_amd.tape import serialize
    bare = _capi.DeviceProblem(serialize(pb.arrays0), None, device=0)
    assert not api.batch_stream_create(bare.ptr, 2) and "affine" in api.error()
    bare.close()
