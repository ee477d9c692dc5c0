"""pytest plugin for tools/run_reference_tests.py: py3.10 typing shim + generated version module
 + the cyipopt stand-in, then the reference is imported in place."""
import os
import sys
import types
import typing

import typing_extensions

typing.Self = typing_extensions.Self
v = types.ModuleType("cvxpy.version")
v.version = v.full_version = "1.8.0.dev0"
v.short_version = "1.8.0"
v.git_revision = "Unknown"
v.commit_count = "0"
v.release = False
sys.modules["cvxpy.version"] = v
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (HERE, ROOT, "/root/reference"):
    if p not in sys.path:
        sys.path.insert(0, p)
