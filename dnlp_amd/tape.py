"""Serialise the lowered problem into the flat tape blob that `dnlp_create` parses.

Blob layout (little endian, mirrored by csrc/tape.h):

    char   magic[8] = "DNLPTAPE"
    u32    version  = 1
    u32    n_arrays
    n_arrays x { char name[40]; u32 dtype (0=f64, 1=i32, 2=i64); u32 pad; u64 count;
                 u64 offset }                     # offset from the start of the blob, 64-B aligned
    raw array data

Every quantity is a named 1-D array; scalars are length-1 arrays.  Sparse maps are CSR
triplets `<name>_ptr (i64)`, `<name>_idx (i32)`, `<name>_val (f64)`.  Dense quad_form
matrices travel as column-major f64 (`dense<k>`) unless they are device resident, in which
case only their order is recorded and the caller binds the HBM pointer with
`dnlp_bind_dense`.
"""
from __future__ import annotations

import struct
from typing import Dict

import numpy as np
import scipy.sparse as sp

from .lowering import Tape

MAGIC = b"DNLPTAPE"
VERSION = 1
_DT = {np.dtype("float64"): 0, np.dtype("int32"): 1, np.dtype("int64"): 2}
DTYPE_CODES = _DT          # (csrc/tape.h: BlobArray::dtype)


def _csr(arrs: Dict[str, np.ndarray], name: str, M):
    from ._capi import CsrArrays
    if not isinstance(M, CsrArrays):        # (the C ABI's views are canonical already and are taken as they are)
        M = sp.csr_matrix(M)
        M.sort_indices()
    arrs[name + "_ptr"] = M.indptr.astype(np.int64, copy=False)
    arrs[name + "_idx"] = M.indices.astype(np.int32, copy=False)
    arrs[name + "_val"] = M.data.astype(np.float64, copy=False)


def tape_arrays(t: Tape, x0, lb, ub, cl, cu) -> Dict[str, np.ndarray]:
    """Named host arrays of the tape (also what oracle/tape_eval.py consumes)."""
    a: Dict[str, np.ndarray] = {}
    nseg = len(t.segments)
    nd, nh = int(t.drow.size), int(t.hrow.size)
    a["dims"] = np.array([t.N, t.m, t.Z, nseg, nd, nh, t.nnzJ, t.nnzH,
                          len(t.dense_consts), len(t.sparse_consts), len(t.dense_blocks),
                          1 if t.hess_coo_complete else 0], dtype=np.int64)
    for k, v in (("x0", x0), ("lb", lb), ("ub", ub), ("cl", cl), ("cu", cu)):
        a[k] = np.asarray(v, dtype=np.float64).reshape(-1)
    # segments (SoA); argument index lists are concatenated into one gather array and
    # contiguous runs are flagged so kernels can use base+i addressing (coalesced)
    gidx = []
    goff = 0
    seg = {k: [] for k in ("op", "n", "a0_base", "a0_off", "a0_len", "a1_base", "a1_off",
                           "a1_len", "zoff", "zcount", "doff", "dcount", "hoff", "hcount",
                           "aux", "d0", "d1", "d2")}
    par, par2 = [], []

    def add_arg(prefix, idx):
        nonlocal goff
        if idx is None:
            seg[prefix + "_base"].append(-1)
            seg[prefix + "_off"].append(-1)
            seg[prefix + "_len"].append(0)
            return
        idx = np.asarray(idx, dtype=np.int64)
        contiguous = idx.size > 0 and (idx.size == 1 or np.all(np.diff(idx) == 1))
        seg[prefix + "_base"].append(int(idx[0]) if contiguous else -1)
        seg[prefix + "_off"].append(goff)
        seg[prefix + "_len"].append(int(idx.size))
        gidx.append(idx.astype(np.int32, copy=False))
        goff += idx.size

    for s in t.segments:
        seg["op"].append(s.op)
        seg["n"].append(s.n)
        add_arg("a0", s.a0)
        add_arg("a1", s.a1)
        for k in ("zoff", "zcount", "doff", "dcount", "hoff", "hcount", "aux"):
            seg[k].append(getattr(s, k))
        seg["d0"].append(s.dims[0])
        seg["d1"].append(s.dims[1])
        seg["d2"].append(s.dims[2])
        par.append(s.param)
        par2.append(s.param2)
    for k, v in seg.items():
        a["seg_" + k] = np.asarray(v, dtype=np.int64)
    a["seg_param"] = np.asarray(par, dtype=np.float64)
    a["seg_param2"] = np.asarray(par2, dtype=np.float64)
    a["gidx"] = np.concatenate(gidx).astype(np.int32, copy=False) if gidx else np.zeros(0, np.int32)
    # linear parts
    a["c0"] = np.array([t.c0])
    a["c"] = t.c.astype(np.float64, copy=False)
    _csr(a, "G", t.G)
    a["b"] = t.b.astype(np.float64, copy=False)
    # derivative maps
    a["drow"] = t.drow.astype(np.int32, copy=False)
    a["dcol"] = t.dcol.astype(np.int32, copy=False)
    a["hrow"] = t.hrow.astype(np.int32, copy=False)
    a["hcol"] = t.hcol.astype(np.int32, copy=False)
    a["hz"] = t.hz.astype(np.int32, copy=False)
    _csr(a, "Mg", t.Mg)
    _csr(a, "Mw", t.Mw)
    _csr(a, "MJ", t.MJ)
    a["Jc"] = t.Jc.astype(np.float64, copy=False)
    a["jac_rows"] = t.jac_rows.astype(np.int32, copy=False)
    a["jac_cols"] = t.jac_cols.astype(np.int32, copy=False)
    _csr(a, "MH", t.MH)
    a["hess_rows"] = t.hess_rows.astype(np.int32, copy=False)
    a["hess_cols"] = t.hess_cols.astype(np.int32, copy=False)
    # constants
    dn = []
    for k, dc in enumerate(t.dense_consts):
        dn.append(dc.n)
        if dc.host is not None:
            a["dense%d" % k] = np.asfortranarray(dc.host).reshape(-1, order="F").astype(np.float64, copy=False)
    a["dense_n"] = np.asarray(dn, dtype=np.int64)
    for k, (P, r, c, v) in enumerate(t.sparse_consts):
        _csr(a, "sp%d" % k, P)
        _csr(a, "sp%dT" % k, sp.csr_matrix(P).T)
        a["sp%d_hr" % k] = r.astype(np.int32, copy=False)
        a["sp%d_hc" % k] = c.astype(np.int32, copy=False)
        a["sp%d_hv" % k] = v.astype(np.float64, copy=False)
    blk = np.array([[b["seg"], b["const"], b["x0"], b["n"], b["z"],
                     (2 if b.get("coo_pos_identity") else 1) if b.get("coo_pos") is not None else 0] for b in t.dense_blocks],
                   dtype=np.int64).reshape(-1)
    a["dense_blocks"] = blk
    for k, b in enumerate(t.dense_blocks):
        if b.get("coo_pos") is not None:
            a["dense_blk%d_pos" % k] = np.asarray(b["coo_pos"], dtype=np.int64)
    return a


def serialize(arrays: Dict[str, np.ndarray]):
    """Named arrays -> one contiguous blob (a bytearray: every array is copied exactly once into
    its 64-byte aligned place; an 800 MB tape used to cost five copies)."""
    names = list(arrays.keys())
    header_size = 16 + len(names) * (40 + 4 + 4 + 8 + 8)
    off = (header_size + 63) // 64 * 64
    entries = []
    flat = []
    for name in names:
        arr = np.ascontiguousarray(arrays[name])
        if arr.dtype not in _DT:
            raise TypeError("array %s has unsupported dtype %s" % (name, arr.dtype))
        nm = name.encode()
        if len(nm) > 39:
            raise ValueError("array name too long: %s" % nm)
        nbytes = arr.size * arr.dtype.itemsize
        entries.append((nm, _DT[arr.dtype], arr.size, off))
        flat.append((off, arr.reshape(-1)))
        off += (nbytes + 63) // 64 * 64
    out = bytearray(off)
    struct.pack_into("<8sII", out, 0, MAGIC, VERSION, len(names))
    pos = 16
    for nm, dt, cnt, o in entries:
        struct.pack_into("<40sIIQQ", out, pos, nm, dt, 0, cnt, o)
        pos += 64
    view = np.frombuffer(out, dtype=np.uint8)
    for o, arr in flat:
        if arr.size:
            view[o:o + arr.size * arr.dtype.itemsize] = arr.view(np.uint8)
    return out


def deserialize(blob: bytes) -> Dict[str, np.ndarray]:
    if blob[:8] != MAGIC:
        raise ValueError("not a DNLP tape blob")
    version, n = struct.unpack_from("<II", blob, 8)
    if version != VERSION:
        raise ValueError("unsupported tape version %d" % version)
    inv = {0: np.float64, 1: np.int32, 2: np.int64}
    out = {}
    pos = 16
    for _ in range(n):
        nm = blob[pos:pos + 40].split(b"\0", 1)[0].decode()
        dt, _pad, cnt, off = struct.unpack_from("<IIQQ", blob, pos + 40)
        pos += 64
        out[nm] = np.frombuffer(blob, dtype=inv[dt], count=cnt, offset=off).copy()
    return out
