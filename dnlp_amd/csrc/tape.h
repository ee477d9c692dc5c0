// Tape blob parser and the exec-space resident copy of the lowered problem.
// Blob layout: dnlp_amd/tape.py.  Normal form: dnlp_amd/lowering.py.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "exec.h"
#include "wave_hdr.h"      // (kCooHeavy)

namespace dnlp {

struct BlobArray {
  int dtype = 0;          // 0 f64, 1 i32, 2 i64
  uint64_t count = 0;
  const void* data = nullptr;
};

// One named array of a tape handed over without the blob around it (include/dnlp_hip.h: dnlp_tape_array).
#ifdef DNLP_HIP_H
using TapeArrayDesc = ::dnlp_tape_array;       // the product library: the public header's struct
#else
struct TapeArrayDesc {                         // (same layout; the test oracle library has no public header)
  const char* name;
  int32_t dtype;          // 0 f64, 1 i32, 2 i64
  int32_t reserved;
  uint64_t count;
  const void* data;
};
#endif

// A VIEW of the caller's memory: everything the model needs is copied into the execution space while the
// problem is created (Tape::load, FusedObjective::load), so the blob is parsed in place -- a 1.16 GB tape
// (BASELINE C3) used to be copied once more here -- and the view is dropped when creation is done.  Only a
// blob that is not 8-byte aligned is copied first.
class TapeBlob {
 public:
  TapeBlob(const void* blob, size_t len) {
    const char* base = static_cast<const char*>(blob);
    if (reinterpret_cast<uintptr_t>(blob) & 7) {
      buf_.assign(base, base + len);
      base = buf_.data();
    }
    if (len < 16 || std::memcmp(base, "DNLPTAPE", 8) != 0) throw std::runtime_error("not a DNLP tape blob");
    uint32_t version, n;
    std::memcpy(&version, base + 8, 4);
    std::memcpy(&n, base + 12, 4);
    if (version != 1) throw std::runtime_error("unsupported tape version");
    size_t pos = 16;
    for (uint32_t k = 0; k < n; ++k) {
      if (pos + 64 > len) throw std::runtime_error("truncated tape header");
      char name[41];
      std::memcpy(name, base + pos, 40);
      name[40] = 0;
      uint32_t dt;
      uint64_t cnt, off;
      std::memcpy(&dt, base + pos + 40, 4);
      std::memcpy(&cnt, base + pos + 48, 8);
      std::memcpy(&off, base + pos + 56, 8);
      pos += 64;
      size_t esz = dt == 1 ? 4 : 8;
      if (off + cnt * esz > len) throw std::runtime_error(std::string("tape array out of range: ") + name);
      BlobArray a;
      a.dtype = static_cast<int>(dt);
      a.count = cnt;
      a.data = base + off;
      arrays_[name] = a;
    }
  }
  // the same tape as separate arrays (no blob was built: the arrays stay where the front-end has them)
  TapeBlob(const TapeArrayDesc* arr, int n) {
    if (!arr || n <= 0) throw std::runtime_error("empty tape array list");
    for (int k = 0; k < n; ++k) {
      if (!arr[k].name) throw std::runtime_error("tape array without a name");
      if (arr[k].dtype < 0 || arr[k].dtype > 2) throw std::runtime_error(std::string("tape array has an unknown dtype: ") + arr[k].name);
      if (arr[k].count && !arr[k].data) throw std::runtime_error(std::string("tape array without data: ") + arr[k].name);
      if (reinterpret_cast<uintptr_t>(arr[k].data) & (arr[k].dtype == 1 ? 3 : 7))
        throw std::runtime_error(std::string("tape array is not aligned to its element size: ") + arr[k].name);
      BlobArray a;
      a.dtype = arr[k].dtype;
      a.count = arr[k].count;
      a.data = arr[k].data;
      arrays_[arr[k].name] = a;
    }
  }
  bool has(const std::string& k) const { return arrays_.count(k) != 0; }
  const BlobArray& get(const std::string& k, int dtype) const {
    auto it = arrays_.find(k);
    if (it == arrays_.end()) throw std::runtime_error("tape array missing: " + k);
    if (it->second.dtype != dtype) throw std::runtime_error("tape array has wrong dtype: " + k);
    return it->second;
  }
  const double* f64(const std::string& k) const { return static_cast<const double*>(get(k, 0).data); }
  const i32* i32s(const std::string& k) const { return static_cast<const i32*>(get(k, 1).data); }
  const i64* i64s(const std::string& k) const { return static_cast<const i64*>(get(k, 2).data); }
  uint64_t count(const std::string& k) const {
    auto it = arrays_.find(k);
    if (it == arrays_.end()) throw std::runtime_error("tape array missing: " + k);
    return it->second.count;
  }

 private:
  std::vector<char> buf_;
  std::map<std::string, BlobArray> arrays_;
};

// CSR matrix resident in an execution space
struct Csr {
  i64 rows = 0, cols = 0, nnz = 0;
  i64* ptr = nullptr;
  i32* idx = nullptr;
  double* val = nullptr;
};

// segment table (host copy; small) — one row per nonlinear atom block
struct SegHost {
  int op;
  i64 n;
  i64 a0_base, a0_off, a0_len, a1_base, a1_off, a1_len;
  i64 zoff, zcount, doff, dcount, hoff, hcount, aux, d0, d1, d2;
  double param, param2;
};

struct DenseBlock {
  i64 seg, cid, x0, n, z, has_pos;
  i64* coo_pos = nullptr;    // exec space, n(n+1)/2 entries in tril_indices order (row-major)
  i64 coo_base = 0;          // has_pos == 2: the positions are coo_base + q (contiguous run), no table
};

struct SparseConst { Csr P, PT; i64 nh = 0; double* hv = nullptr; };

// Index of a COO pattern BY OUTPUT, built once when the tape is loaded: the entries that feed output g are the
// segment ptr[g] .. ptr[g+1] of (ent = COO entry, src = index into the multiplied vector), in storage order.
// A product that walks the segments sums every output in one fixed order: no floating-point atomics, the same
// bits on every run and in every execution space that sums a segment serially (host loops, the in-kernel solver).
// ptr == nullptr: no index (pattern above Tape::coo_index_max entries); the exec space falls back to its scatter form.
struct CooIdx {
  i64 nout = 0, total = 0;
  i32* ptr = nullptr;
  i32* ent = nullptr;
  i32* src = nullptr;
  // outputs with more than kHeavy entries (a dense column under many rows): the in-kernel solver sums those with all
  // its lanes and a fixed reduction tree instead of one lane walking the whole segment
  static constexpr i64 kHeavy = kCooHeavy;
  i64 nheavy = 0;
  i32* heavy = nullptr;
};

// Everything the evaluators (model.h) and the interior-point loop (ipm_core.h) read, as plain
// pointers and counts: the same struct describes a tape loaded by Tape<E>::load below and one
// instance of a batch inside the batched-solve kernel (exec_block.h).  "exec space" arrays live
// where E::map runs; "control space" arrays are read by the code that drives E (host memory for
// the host / HIP spaces, device memory inside the batch kernel).
struct TapeView {
  i64 N = 0, m = 0, Z = 0, nseg = 0, nd = 0, nh = 0, nnzJ = 0, nnzH = 0, ndense = 0, nsparse = 0,
      nblk = 0, coo_complete = 1;
  // exec space: bounds and start of the canonical problem
  double *d_x0 = nullptr, *d_lb = nullptr, *d_ub = nullptr, *d_cl = nullptr, *d_cu = nullptr;
  // flat (elementwise-class) segment table in exec space
  i64 nflat = 0, flat_units = 0;
  i64* flat_start = nullptr;   // nflat+1 prefix of work units
  i32* flat_op = nullptr;
  i64 *flat_a0b = nullptr, *flat_a0o = nullptr, *flat_a1b = nullptr, *flat_a1o = nullptr;
  i64 *flat_a0l = nullptr, *flat_a1l = nullptr;
  i64 *flat_zoff = nullptr, *flat_doff = nullptr, *flat_hoff = nullptr, *flat_n = nullptr;
  i64 *flat_d0 = nullptr, *flat_d1 = nullptr, *flat_d2 = nullptr;
  double *flat_p = nullptr, *flat_p2 = nullptr;
  i32* gidx = nullptr;
  double c0 = 0.0;
  double *c = nullptr, *b = nullptr, *Jc = nullptr;
  Csr G, Mg, Mw, MJ, MH;
  i32 *jac_rows = nullptr, *jac_cols = nullptr, *hess_rows = nullptr, *hess_cols = nullptr;
  i64* jac_rowptr = nullptr;   // m+1: the Jacobian COO is row-major sorted
  CooIdx jac_by_row, jac_by_col, hess_sym;   // order-fixed products J v, J^T v, sym(H) v (model.h)
  // A Jacobian too large to index whose rows all carry the SAME column list (a dense constraint block: BASELINE C3's
  // A, 1e3 x 1e4) is a rectangular matrix stored row-major: jac_rect_cols = its row length (0: not rectangular), the
  // columns are jac_cols[0 .. jac_rect_cols).  The host-driven device space multiplies with it in a fixed order
  // without any index (exec_hip.h rect_mult / rect_tmult).
  i64 jac_rect_cols = 0;
  // control space: reduction-class segments, constants, dense Hessian blocks
  const SegHost* segs = nullptr;
  const i64* red_segs = nullptr;
  i64 nred = 0;
  const double** dense_ptr = nullptr;   // per dense constant: exec-space column-major matrix (or null until bound)
  i64* dense_ld = nullptr;
  const SparseConst* sparse = nullptr;
  const DenseBlock* blocks = nullptr;
  // reduced-space structure (dnlp_amd/reduced.py): every constraint row defines one auxiliary variable
  bool reducible = false;
  i64 nfree = 0, red_depth = 0;
  i32 *def_var = nullptr, *free_idx = nullptr;

  DNLP_HD bool dense_bound() const {
    for (i64 k = 0; k < ndense; ++k) if (!dense_ptr[k]) return false;
    return true;
  }
};

// The lowered problem resident in exec space E: owner of the arrays behind a TapeView.
template <class E>
struct Tape : TapeView {
  E* ex = nullptr;
  std::vector<double> h_x0, h_lb, h_ub, h_cl, h_cu;      // host copies
  std::vector<SegHost> h_segs;
  std::vector<i64> h_red_segs;   // indices of reduction-class segments
  std::vector<i64> h_flat_seg;   // segment index of every flat-table row
  std::vector<i32> h_jac_rows, h_jac_cols, h_hess_rows, h_hess_cols;
  std::vector<double> h_jac_const;   // |coefficient| of Jacobian entries that are constant (affine rows), else 0
  // constants
  std::vector<i64> dense_n;
  std::vector<const double*> h_dense_ptr;   // exec space, column-major
  std::vector<i64> h_dense_ld;
  std::vector<bool> dense_owned;
  std::vector<SparseConst> h_sparse;
  std::vector<DenseBlock> h_blocks;

  template <class T> T* up(const T* src, size_t n) {
    T* d = ex->template alloc<T>(n);
    ex->h2d(d, src, n * sizeof(T));
    return d;
  }
  // patterns above this many entries keep the exec space's scatter product (BASELINE C3's dense Jacobian, 1e7 entries:
  // the column index alone would cost 0.16 s of its first call); DNLP_COO_DET_MAX overrides
  i64 coo_index_max = 4000000;
  // mode 0: outputs = rows (src = column), 1: outputs = columns (src = row), 2: both sides of a lower triangle
  CooIdx build_coo_index(const std::vector<i32>& hr, const std::vector<i32>& hc, i64 nout, int mode) {
    CooIdx ix;
    const i64 nnz = static_cast<i64>(hr.size());
    if (const char* e = std::getenv("DNLP_COO_DET_MAX")) coo_index_max = std::atoll(e);
    if (nnz > coo_index_max || nnz > (i64{1} << 30)) return ix;      // (entry ids are 32-bit, two per entry in mode 2)
    std::vector<i32> ptr(static_cast<size_t>(nout) + 1, 0);
    auto each = [&](auto&& f) {                  // (entry, output, source) in storage order
      for (i64 p = 0; p < nnz; ++p) {
        const i32 rp = hr[static_cast<size_t>(p)], cp = hc[static_cast<size_t>(p)];
        if (mode != 1) f(p, rp, cp);
        if (mode == 1 || (mode == 2 && rp != cp)) f(p, cp, rp);
      }
    };
    each([&](i64, i32 o, i32) { ++ptr[static_cast<size_t>(o) + 1]; });
    for (i64 g = 0; g < nout; ++g) ptr[static_cast<size_t>(g) + 1] += ptr[static_cast<size_t>(g)];
    const i64 total = ptr[static_cast<size_t>(nout)];
    std::vector<i32> ent(static_cast<size_t>(total)), src(static_cast<size_t>(total));
    std::vector<i32> fill(ptr.begin(), ptr.end() - 1);
    each([&](i64 p, i32 o, i32 sidx) {
      const i64 at = fill[static_cast<size_t>(o)]++;
      ent[static_cast<size_t>(at)] = static_cast<i32>(p);
      src[static_cast<size_t>(at)] = sidx;
    });
    ix.nout = nout; ix.total = total;
    std::vector<i32> heavy;
    for (i64 g = 0; g < nout; ++g)
      if (ptr[static_cast<size_t>(g) + 1] - ptr[static_cast<size_t>(g)] > CooIdx::kHeavy) heavy.push_back(static_cast<i32>(g));
    ix.nheavy = static_cast<i64>(heavy.size());
    ix.heavy = up(heavy.data(), heavy.size());
    ix.ptr = up(ptr.data(), ptr.size());
    ix.ent = up(ent.data(), ent.size());
    ix.src = up(src.data(), src.size());
    return ix;
  }
  Csr up_csr(const TapeBlob& tb, const std::string& name, i64 rows, i64 cols) {
    Csr m;
    m.rows = rows;
    m.cols = cols;
    m.nnz = static_cast<i64>(tb.count(name + "_idx"));
    m.ptr = up(tb.i64s(name + "_ptr"), static_cast<size_t>(rows + 1));
    m.idx = up(tb.i32s(name + "_idx"), static_cast<size_t>(m.nnz));
    m.val = up(tb.f64(name + "_val"), static_cast<size_t>(m.nnz));
    return m;
  }

  void load(E* e, const TapeBlob& tb) {
    ex = e;
    const i64* d = tb.i64s("dims");
    N = d[0]; m = d[1]; Z = d[2]; nseg = d[3]; nd = d[4]; nh = d[5]; nnzJ = d[6]; nnzH = d[7];
    ndense = d[8]; nsparse = d[9]; nblk = d[10]; coo_complete = d[11];
    auto vec = [&](const char* k, i64 n) { const double* p = tb.f64(k); return std::vector<double>(p, p + n); };
    h_x0 = vec("x0", N); h_lb = vec("lb", N); h_ub = vec("ub", N); h_cl = vec("cl", m); h_cu = vec("cu", m);
    d_x0 = up(h_x0.data(), h_x0.size()); d_lb = up(h_lb.data(), h_lb.size()); d_ub = up(h_ub.data(), h_ub.size());
    d_cl = up(h_cl.data(), h_cl.size()); d_cu = up(h_cu.data(), h_cu.size());
    h_segs.resize(static_cast<size_t>(nseg));
    auto S = [&](const char* k) { return tb.i64s(std::string("seg_") + k); };
    for (i64 s = 0; s < nseg; ++s) {
      SegHost& g = h_segs[static_cast<size_t>(s)];
      g.op = static_cast<int>(S("op")[s]); g.n = S("n")[s];
      g.a0_base = S("a0_base")[s]; g.a0_off = S("a0_off")[s]; g.a0_len = S("a0_len")[s];
      g.a1_base = S("a1_base")[s]; g.a1_off = S("a1_off")[s]; g.a1_len = S("a1_len")[s];
      g.zoff = S("zoff")[s]; g.zcount = S("zcount")[s]; g.doff = S("doff")[s]; g.dcount = S("dcount")[s];
      g.hoff = S("hoff")[s]; g.hcount = S("hcount")[s]; g.aux = S("aux")[s];
      g.d0 = S("d0")[s]; g.d1 = S("d1")[s]; g.d2 = S("d2")[s];
      g.param = tb.f64("seg_param")[s]; g.param2 = tb.f64("seg_param2")[s];
    }
    // flat table
    std::vector<i64> fs{0}, a0b, a0o, a0l, a1b, a1o, a1l, zo, dof, ho, nn, e0, e1, e2;
    std::vector<i32> fop;
    std::vector<double> fp, fp2;
    // OP_MATMUL (33) is elementwise-class (one unit per output entry) despite its opcode
    for (i64 s = 0; s < nseg; ++s) {
      const SegHost& g = h_segs[static_cast<size_t>(s)];
      bool flat = (g.op < 30) || (g.op == 33);
      if (!flat) { h_red_segs.push_back(s); continue; }
      i64 units = (g.op == 33) ? g.d0 * g.d2 : g.n;
      fs.push_back(fs.back() + units);
      h_flat_seg.push_back(s);
      fop.push_back(g.op); a0b.push_back(g.a0_base); a0o.push_back(g.a0_off); a0l.push_back(g.a0_len);
      a1b.push_back(g.a1_base); a1o.push_back(g.a1_off); a1l.push_back(g.a1_len);
      zo.push_back(g.zoff); dof.push_back(g.doff); ho.push_back(g.hoff); nn.push_back(g.n);
      e0.push_back(g.d0); e1.push_back(g.d1); e2.push_back(g.d2); fp.push_back(g.param); fp2.push_back(g.param2);
    }
    nflat = static_cast<i64>(fop.size());
    flat_units = fs.back();
    flat_start = up(fs.data(), fs.size());
    flat_op = up(fop.data(), fop.size());
    flat_a0b = up(a0b.data(), a0b.size()); flat_a0o = up(a0o.data(), a0o.size()); flat_a0l = up(a0l.data(), a0l.size());
    flat_a1b = up(a1b.data(), a1b.size()); flat_a1o = up(a1o.data(), a1o.size()); flat_a1l = up(a1l.data(), a1l.size());
    flat_zoff = up(zo.data(), zo.size()); flat_doff = up(dof.data(), dof.size()); flat_hoff = up(ho.data(), ho.size());
    flat_n = up(nn.data(), nn.size());
    flat_d0 = up(e0.data(), e0.size()); flat_d1 = up(e1.data(), e1.size()); flat_d2 = up(e2.data(), e2.size());
    flat_p = up(fp.data(), fp.size()); flat_p2 = up(fp2.data(), fp2.size());
    gidx = up(tb.i32s("gidx"), tb.count("gidx"));
    c0 = tb.f64("c0")[0];
    c = up(tb.f64("c"), static_cast<size_t>(N + Z));
    b = up(tb.f64("b"), static_cast<size_t>(m));
    Jc = up(tb.f64("Jc"), static_cast<size_t>(nnzJ));
    G = up_csr(tb, "G", m, N + Z);
    Mg = up_csr(tb, "Mg", N, nd);
    Mw = up_csr(tb, "Mw", Z, 1 + m);
    MJ = up_csr(tb, "MJ", nnzJ, nd);
    MH = up_csr(tb, "MH", nnzH, nh);
    h_jac_rows.assign(tb.i32s("jac_rows"), tb.i32s("jac_rows") + nnzJ);
    h_jac_cols.assign(tb.i32s("jac_cols"), tb.i32s("jac_cols") + nnzJ);
    {
      const i64* mjp = tb.i64s("MJ_ptr");
      const double* jc0 = tb.f64("Jc");
      h_jac_const.assign(static_cast<size_t>(nnzJ), 0.0);
      for (i64 p = 0; p < nnzJ; ++p)
        if (mjp[p + 1] == mjp[p]) h_jac_const[static_cast<size_t>(p)] = std::fabs(jc0[p]);
    }
    h_hess_rows.assign(tb.i32s("hess_rows"), tb.i32s("hess_rows") + nnzH);
    h_hess_cols.assign(tb.i32s("hess_cols"), tb.i32s("hess_cols") + nnzH);
    {
      std::vector<i64> rp(static_cast<size_t>(m + 1), 0);
      for (i32 r : h_jac_rows) rp[static_cast<size_t>(r) + 1]++;
      for (i64 i = 0; i < m; ++i) rp[static_cast<size_t>(i + 1)] += rp[static_cast<size_t>(i)];
      for (i64 p = 1; p < nnzJ; ++p)
        if (h_jac_rows[static_cast<size_t>(p)] < h_jac_rows[static_cast<size_t>(p - 1)]) throw std::runtime_error("tape Jacobian pattern is not row-major sorted");
      jac_rowptr = up(rp.data(), rp.size());
    }
    jac_rows = up(h_jac_rows.data(), h_jac_rows.size());
    jac_cols = up(h_jac_cols.data(), h_jac_cols.size());
    hess_rows = up(h_hess_rows.data(), h_hess_rows.size());
    hess_cols = up(h_hess_cols.data(), h_hess_cols.size());
    jac_by_row = build_coo_index(h_jac_rows, h_jac_cols, m, 0);
    jac_by_col = build_coo_index(h_jac_rows, h_jac_cols, N, 1);
    hess_sym = build_coo_index(h_hess_rows, h_hess_cols, N, 2);
    jac_rect_cols = 0;
    if (!jac_by_col.ptr && m > 0 && nnzJ > 0 && nnzJ % m == 0) {
      const i64 L = nnzJ / m;
      bool rect = true;
      for (i64 i = 0; i < m && rect; ++i) {
        const i32* row = h_jac_cols.data() + i * L;
        rect = h_jac_rows[static_cast<size_t>(i * L)] == i && h_jac_rows[static_cast<size_t>(i * L + L - 1)] == i &&
               (i == 0 || std::memcmp(row, h_jac_cols.data(), sizeof(i32) * static_cast<size_t>(L)) == 0);
      }
      if (rect) jac_rect_cols = L;
    }
    dense_n.assign(tb.i64s("dense_n"), tb.i64s("dense_n") + ndense);
    h_dense_ptr.assign(static_cast<size_t>(ndense), nullptr);
    h_dense_ld.assign(static_cast<size_t>(ndense), 0);
    dense_owned.assign(static_cast<size_t>(ndense), false);
    for (i64 k = 0; k < ndense; ++k) {
      std::string nm = "dense" + std::to_string(k);
      if (tb.has(nm)) {
        i64 n = dense_n[static_cast<size_t>(k)];
        h_dense_ptr[static_cast<size_t>(k)] = up(tb.f64(nm), static_cast<size_t>(n * n));
        h_dense_ld[static_cast<size_t>(k)] = n;
        dense_owned[static_cast<size_t>(k)] = true;
      }
    }
    h_sparse.resize(static_cast<size_t>(nsparse));
    for (i64 k = 0; k < nsparse; ++k) {
      std::string nm = "sp" + std::to_string(k);
      i64 n = 0;
      for (auto& g : h_segs) if (g.op == 31 && g.aux == k) n = g.n;
      h_sparse[static_cast<size_t>(k)].P = up_csr(tb, nm, n, n);
      h_sparse[static_cast<size_t>(k)].PT = up_csr(tb, nm + "T", n, n);
      h_sparse[static_cast<size_t>(k)].nh = static_cast<i64>(tb.count(nm + "_hv"));
      h_sparse[static_cast<size_t>(k)].hv = up(tb.f64(nm + "_hv"), tb.count(nm + "_hv"));
    }
    h_blocks.resize(static_cast<size_t>(nblk));
    const i64* bl = nblk ? tb.i64s("dense_blocks") : nullptr;
    for (i64 k = 0; k < nblk; ++k) {
      DenseBlock& B = h_blocks[static_cast<size_t>(k)];
      B.seg = bl[6 * k]; B.cid = bl[6 * k + 1]; B.x0 = bl[6 * k + 2]; B.n = bl[6 * k + 3];
      B.z = bl[6 * k + 4]; B.has_pos = bl[6 * k + 5];
      if (B.has_pos == 2) {
        B.coo_base = tb.i64s("dense_blk" + std::to_string(k) + "_pos")[0];
      } else if (B.has_pos) {
        std::string nm = "dense_blk" + std::to_string(k) + "_pos";
        B.coo_pos = up(tb.i64s(nm), tb.count(nm));
      }
    }
    // control-space views of the small host tables
    segs = h_segs.data(); red_segs = h_red_segs.data(); nred = static_cast<i64>(h_red_segs.size());
    dense_ptr = h_dense_ptr.data(); dense_ld = h_dense_ld.data();
    sparse = h_sparse.data(); blocks = h_blocks.data();
    load_reduction(tb);
  }

  void load_reduction(const TapeBlob& tb) {
    if (!tb.has("def_var") || !tb.has("free_idx") || !tb.has("red_depth")) return;
    reducible = true;
    nfree = static_cast<i64>(tb.count("free_idx"));
    red_depth = tb.i64s("red_depth")[0];
    def_var = up(tb.i32s("def_var"), tb.count("def_var"));
    free_idx = up(tb.i32s("free_idx"), tb.count("free_idx"));
  }

};

}  // namespace dnlp
