"""Run the REFERENCE's own NLP test-suite (in place, from /root/reference) against this
repository's solver through the cyipopt stand-in (tools/refshim).  Build-container only.

    python tools/run_reference_tests.py [pytest args ...]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1",
           PYTHONPATH=os.pathsep.join([os.path.join(HERE, "refshim"), os.path.dirname(HERE)]))
args = sys.argv[1:] or ["/root/reference/cvxpy/tests/NLP_tests"]
cmd = [sys.executable, "-m", "pytest", "-p", "dnlp_ref_plugin", "-p", "no:cacheprovider", "-q",
       "--rootdir", "/tmp", "-W", "ignore"] + args
sys.exit(subprocess.call(cmd, cwd="/tmp", env=env))
