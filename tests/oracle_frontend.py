"""TEST INFRASTRUCTURE: run the dnlp_amd front-end (`Problem.solve(nlp=True)`: reduction chain,
Bounds, best_of loop, invert / unpack) with the CPU oracle (oracle/) as the engine behind the C-ABI
shaped handle, so that the host logic is covered without a GPU.  The product never does this: its
solver interface creates device handles only and fails loudly without an MI355X."""
import contextlib

from dnlp_amd.nlp_solver import DeviceOracles, HIPNLP, build_nlp_data
from dnlp_amd.tape import serialize


class OracleNLP(HIPNLP):
    def apply(self, problem, user_variables=None, make_handle=True, fused_spec=None):
        from oracle.oracle_capi import OracleProblem
        data, inverse_data = build_nlp_data(problem, user_variables, fused_spec)
        if not make_handle:
            return data, inverse_data
        handle = OracleProblem(serialize(data["tape_arrays"]))
        oracles = DeviceOracles(handle, len(data["x0"]), len(data["cl"]))
        data["handle"] = handle
        data["oracles"] = oracles
        for k in ("objective", "gradient", "constraints", "jacobian", "jacobianstructure", "hessian",
                  "hessianstructure"):
            data[k] = getattr(oracles, k)
        return data, inverse_data

    def _use_device_loop(self, data, options, mode):
        return False

    solve_batch_via_data = None          # best_of runs the reference's serial loop


@contextlib.contextmanager
def oracle_engine():
    """Within the block, dnlp_amd.Problem.solve(nlp=True) is answered by the CPU oracle."""
    import dnlp_amd as cp
    from dnlp_amd.problem import Maximize, NLPChain
    orig = cp.Problem._build_chain
    cp.Problem._build_chain = lambda self, solver: NLPChain(type(self.objective) == Maximize, OracleNLP())
    try:
        yield
    finally:
        cp.Problem._build_chain = orig
