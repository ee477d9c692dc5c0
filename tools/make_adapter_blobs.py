"""Tape blobs produced by dnlp_amd.cvxpy_adapter from the REFERENCE's own objects (build container
only): for every problem of tests/problem_zoo.GOLDEN_ZOO the reference's cvxpy builds the problem,
the reference's own reduction chain canonicalises it (problems/problem.py:1220-1243), and
`tape_from_cvxpy` translates the canonical cvxpy trees into the device tape.  The serialised tapes
are committed under tests/golden/adapter/ (data: index arrays and constants, no reference source) so
that the GPU box — where cvxpy does not exist — can run the adapter's output through the C ABI
(tests/test_adapter_blobs.py).

    python tools/make_adapter_blobs.py [name ...]
"""
import gzip
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ref_import import import_reference, ref_chain_apply  # noqa: E402


def main():
    cp = import_reference()
    from dnlp_amd.cvxpy_adapter import tape_blob_from_cvxpy
    from problem_zoo import GOLDEN_ZOO
    out_dir = os.path.join(ROOT, "tests", "golden", "adapter")
    os.makedirs(out_dir, exist_ok=True)
    only = set(sys.argv[1:])
    for name, builder in GOLDEN_ZOO.items():
        if only and name not in only:
            continue
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            prob = builder(cp)
            data, inv, chain = ref_chain_apply(cp, prob)
            blob = tape_blob_from_cvxpy(data)
        with gzip.GzipFile(os.path.join(out_dir, name + ".blob.gz"), "wb", mtime=0) as fh:
            fh.write(blob)
        print("%-22s %8d bytes  N=%d m=%d" % (name, len(blob), len(data["x0"]), len(data["cl"])))


if __name__ == "__main__":
    main()
