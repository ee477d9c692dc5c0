"""Exceptions of the nlp=True path (reference: cvxpy/error.py:21-47)."""


class SolverError(Exception):
    """The solver reported a failure status (reference problem.py:1624-1629)."""


class DNLPError(Exception):
    """The problem does not follow the disciplined-NLP rules (reference problem.py:1276-1277)."""


class DeviceUnavailableError(RuntimeError):
    """libdnlp_hip.so is missing or no MI355X device is present. There is no CPU fallback."""
