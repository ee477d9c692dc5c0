// Affine forms of the lowering, in C++ (the role of cvxcore's lin_ops builder, cvxpy/cvxcore/src/LinOpOperations.cpp and
// cvxcore.cpp:161-215, for the nlp=True path): value = A [x; z] + b with A a CSR matrix over the problem's columns.
// The front-end's DAG walk (dnlp_amd/lowering.py) keeps a handle per sub-expression and composes them with the few
// operations affine atoms need — selection of rows, sum, negation, row scaling, left-multiplication by a constant
// sparse matrix, stacking — instead of building scipy.sparse objects (tens of microseconds of Python per object: a
// small problem spent 5-30 ms in them, BASELINE C2's canonical form 0.11 s).  Host code only; no device is touched.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#include "lower_maps.h"

namespace dnlp {

// The matrix part is shared between forms that differ only in their constants (A x - b keeps the arrays of A x: for
// BASELINE C3's 1e3 x 1e4 dense block that is 120 MB not copied), and exported to the front-end as views.
struct LfMat {
  std::vector<long long> ptr;      // rows + 1
  std::vector<int32_t> idx;
  std::vector<double> val;
  bool canonical = true;           // every row sorted by column, no duplicate columns
};
struct LinFormH {
  long long rows = 0, ncol = 0;
  std::shared_ptr<LfMat> m = std::make_shared<LfMat>();
  std::vector<long long>& ptr;
  std::vector<int32_t>& idx;
  std::vector<double>& val;
  bool& canonical;
  std::vector<double> b;           // rows
  LinFormH() : ptr(m->ptr), idx(m->idx), val(m->val), canonical(m->canonical) {}
  // a form over the matrix part of another one (the constants are the caller's to fill)
  explicit LinFormH(const LinFormH& o) : rows(o.rows), ncol(o.ncol), m(o.m), ptr(m->ptr), idx(m->idx), val(m->val), canonical(m->canonical) {}
  LinFormH& operator=(const LinFormH&) = delete;
  long long nnz() const { return static_cast<long long>(idx.size()); }
};

inline LinFormH* lf_const(long long ncol, long long n, const double* b) {
  auto* f = new LinFormH();
  f->rows = n; f->ncol = ncol;
  f->ptr.assign(static_cast<size_t>(n) + 1, 0);
  f->b.assign(b, b + n);
  return f;
}
// rows r = 0 .. n-1 with the single coefficient 1 at column col0 + r (a variable, or a run of z entries)
inline LinFormH* lf_range(long long ncol, long long n, long long col0) {
  auto* f = new LinFormH();
  f->rows = n; f->ncol = ncol;
  f->ptr.resize(static_cast<size_t>(n) + 1);
  f->idx.resize(static_cast<size_t>(n));
  f->val.assign(static_cast<size_t>(n), 1.0);
  f->b.assign(static_cast<size_t>(n), 0.0);
  for (long long r = 0; r <= n; ++r) f->ptr[static_cast<size_t>(r)] = r;
  for (long long r = 0; r < n; ++r) f->idx[static_cast<size_t>(r)] = static_cast<int32_t>(col0 + r);
  return f;
}
inline LinFormH* lf_select(const LinFormH& a, const long long* sel, long long n) {
  auto* f = new LinFormH();
  f->rows = n; f->ncol = a.ncol; f->canonical = a.canonical;
  f->ptr.resize(static_cast<size_t>(n) + 1);
  f->b.resize(static_cast<size_t>(n));
  long long nnz = 0;
  for (long long r = 0; r < n; ++r) {
    const long long s = sel[r];
    if (s < 0 || s >= a.rows) { delete f; throw std::runtime_error("linform select: row index out of range"); }
    f->ptr[static_cast<size_t>(r)] = nnz;
    nnz += a.ptr[static_cast<size_t>(s) + 1] - a.ptr[static_cast<size_t>(s)];
    f->b[static_cast<size_t>(r)] = a.b[static_cast<size_t>(s)];
  }
  f->ptr[static_cast<size_t>(n)] = nnz;
  f->idx.resize(static_cast<size_t>(nnz));
  f->val.resize(static_cast<size_t>(nnz));
  for (long long r = 0; r < n; ++r) {
    const long long s = sel[r], o = f->ptr[static_cast<size_t>(r)], p0 = a.ptr[static_cast<size_t>(s)];
    const long long len = a.ptr[static_cast<size_t>(s) + 1] - p0;
    if (len) {
      std::memcpy(&f->idx[static_cast<size_t>(o)], &a.idx[static_cast<size_t>(p0)], static_cast<size_t>(len) * sizeof(int32_t));
      std::memcpy(&f->val[static_cast<size_t>(o)], &a.val[static_cast<size_t>(p0)], static_cast<size_t>(len) * sizeof(double));
    }
  }
  return f;
}
// sort the rows by column and sum duplicate columns (in place)
inline void lf_canonicalize(LinFormH& f) {
  if (f.canonical) return;
  std::vector<std::pair<int32_t, double>> row;
  long long w = 0;
  std::vector<long long> nptr(f.ptr.size());
  for (long long r = 0; r < f.rows; ++r) {
    const long long p0 = f.ptr[static_cast<size_t>(r)], p1 = f.ptr[static_cast<size_t>(r) + 1];
    bool sorted = true;
    for (long long e = p0 + 1; e < p1 && sorted; ++e) sorted = f.idx[static_cast<size_t>(e)] > f.idx[static_cast<size_t>(e - 1)];
    nptr[static_cast<size_t>(r)] = w;
    if (sorted) {
      if (w != p0) for (long long e = p0; e < p1; ++e) { f.idx[static_cast<size_t>(w + e - p0)] = f.idx[static_cast<size_t>(e)]; f.val[static_cast<size_t>(w + e - p0)] = f.val[static_cast<size_t>(e)]; }
      w += p1 - p0;
      continue;
    }
    row.clear();
    for (long long e = p0; e < p1; ++e) row.emplace_back(f.idx[static_cast<size_t>(e)], f.val[static_cast<size_t>(e)]);
    std::stable_sort(row.begin(), row.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
    for (size_t k = 0; k < row.size(); ++k) {
      if (k && f.idx[static_cast<size_t>(w - 1)] == row[k].first && w > nptr[static_cast<size_t>(r)]) f.val[static_cast<size_t>(w - 1)] += row[k].second;
      else { f.idx[static_cast<size_t>(w)] = row[k].first; f.val[static_cast<size_t>(w)] = row[k].second; ++w; }
    }
  }
  nptr[static_cast<size_t>(f.rows)] = w;
  f.ptr.swap(nptr);
  f.idx.resize(static_cast<size_t>(w));
  f.val.resize(static_cast<size_t>(w));
  f.canonical = true;
}
// a + b (same row count): rows merged by column, coefficients of a common column added
inline LinFormH* lf_add(LinFormH& a, LinFormH& b) {
  if (a.rows != b.rows) throw std::runtime_error("linform add: row counts differ");
  if (a.nnz() == 0 || b.nnz() == 0) {
    auto* f = new LinFormH(a.nnz() == 0 ? b : a);
    f->b.resize(static_cast<size_t>(a.rows));
    for (long long r = 0; r < a.rows; ++r) f->b[static_cast<size_t>(r)] = a.b[static_cast<size_t>(r)] + b.b[static_cast<size_t>(r)];
    return f;
  }
  lf_canonicalize(a);
  lf_canonicalize(b);
  auto* f = new LinFormH();
  f->rows = a.rows; f->ncol = a.ncol;
  f->ptr.assign(static_cast<size_t>(a.rows) + 1, 0);
  f->b.resize(static_cast<size_t>(a.rows));
  // rows are independent: count, prefix, fill — in row chunks on a few threads (the canonical form of BASELINE C2 adds
  // forms of 1e5 ... 3e5 rows six times)
  // (a coefficient that comes out exactly zero is dropped, as scipy's csr + csr does: theta - theta.T has +1 and -1
  //  on the same position of its diagonal rows, and the Jacobian pattern must not carry that entry)
  auto merge_row = [&](long long r, int32_t* oi, double* ov) -> long long {
    long long i = a.ptr[static_cast<size_t>(r)], j = b.ptr[static_cast<size_t>(r)], w = 0;
    const long long i1 = a.ptr[static_cast<size_t>(r) + 1], j1 = b.ptr[static_cast<size_t>(r) + 1];
    while (i < i1 || j < j1) {
      const int32_t ci = i < i1 ? a.idx[static_cast<size_t>(i)] : INT32_MAX, cj = j < j1 ? b.idx[static_cast<size_t>(j)] : INT32_MAX;
      int32_t c;
      double v;
      if (ci == cj) { c = ci; v = a.val[static_cast<size_t>(i)] + b.val[static_cast<size_t>(j)]; ++i; ++j; }
      else if (ci < cj) { c = ci; v = a.val[static_cast<size_t>(i)]; ++i; }
      else { c = cj; v = b.val[static_cast<size_t>(j)]; ++j; }
      if (v != 0.0) { if (oi) { oi[w] = c; ov[w] = v; } ++w; }
    }
    return w;
  };
  lm_par_for(a.rows, 8192, [&](lm_i64 lo, lm_i64 hi) {
    for (lm_i64 r = lo; r < hi; ++r) {
      f->ptr[static_cast<size_t>(r) + 1] = merge_row(r, nullptr, nullptr);
      f->b[static_cast<size_t>(r)] = a.b[static_cast<size_t>(r)] + b.b[static_cast<size_t>(r)];
    }
  });
  for (long long r = 0; r < a.rows; ++r) f->ptr[static_cast<size_t>(r) + 1] += f->ptr[static_cast<size_t>(r)];
  f->idx.resize(static_cast<size_t>(f->ptr[static_cast<size_t>(a.rows)]));
  f->val.resize(f->idx.size());
  lm_par_for(a.rows, 8192, [&](lm_i64 lo, lm_i64 hi) {
    for (lm_i64 r = lo; r < hi; ++r) {
      const long long o = f->ptr[static_cast<size_t>(r)];
      merge_row(r, f->idx.data() + o, f->val.data() + o);
    }
  });
  return f;
}
// diag(s) a   (s == nullptr: -a); coefficients that become exactly zero are dropped (scipy's diags(s) @ A does)
inline LinFormH* lf_scale_rows(const LinFormH& a, const double* s) {
  auto* f = new LinFormH();
  f->rows = a.rows; f->ncol = a.ncol; f->canonical = a.canonical;
  f->ptr.resize(static_cast<size_t>(a.rows) + 1);
  f->b.resize(static_cast<size_t>(a.rows));
  f->idx.reserve(a.idx.size());
  f->val.reserve(a.idx.size());
  for (long long r = 0; r < a.rows; ++r) {
    const double c = s ? s[r] : -1.0;
    f->ptr[static_cast<size_t>(r)] = static_cast<long long>(f->idx.size());
    for (long long e = a.ptr[static_cast<size_t>(r)]; e < a.ptr[static_cast<size_t>(r) + 1]; ++e) {
      const double v = c * a.val[static_cast<size_t>(e)];
      if (v != 0.0 || !s) { f->idx.push_back(a.idx[static_cast<size_t>(e)]); f->val.push_back(v); }
    }
    f->b[static_cast<size_t>(r)] = c * a.b[static_cast<size_t>(r)];
  }
  f->ptr[static_cast<size_t>(a.rows)] = static_cast<long long>(f->idx.size());
  return f;
}
// is `a` a plain selection of columns (one coefficient 1 per row, no constant)?
inline bool lf_is_selection(const LinFormH& a) {
  if (a.nnz() != a.rows) return false;
  for (long long r = 0; r < a.rows; ++r) {
    if (a.ptr[static_cast<size_t>(r) + 1] - a.ptr[static_cast<size_t>(r)] != 1 || a.val[static_cast<size_t>(r)] != 1.0 || a.b[static_cast<size_t>(r)] != 0.0) return false;
  }
  return true;
}
// S a with S a constant CSR matrix (srows x a.rows)
inline LinFormH* lf_apply_csr(const LinFormH& a, long long srows, const long long* sp, const int32_t* si, const double* sv) {
  auto* f = new LinFormH();
  f->rows = srows; f->ncol = a.ncol;
  f->ptr.resize(static_cast<size_t>(srows) + 1);
  f->b.assign(static_cast<size_t>(srows), 0.0);
  const long long snnz = sp[srows];
  if (lf_is_selection(a)) {
    // S with its columns renamed: one gather (a dense 1e3 x 1e4 constraint block is 1e7 entries), rows re-sorted only
    // when the renaming is not monotone
    f->idx.resize(static_cast<size_t>(snnz));
    f->val.assign(sv, sv + snnz);
    std::memcpy(f->ptr.data(), sp, (static_cast<size_t>(srows) + 1) * sizeof(long long));
    bool mono = true;
    for (long long r = 1; r < a.rows && mono; ++r) mono = a.idx[static_cast<size_t>(r)] > a.idx[static_cast<size_t>(r - 1)];
    for (long long e = 0; e < snnz; ++e) f->idx[static_cast<size_t>(e)] = a.idx[static_cast<size_t>(si[e])];
    bool s_sorted = true;
    for (long long r = 0; r < srows && s_sorted; ++r)
      for (long long e = sp[r] + 1; e < sp[r + 1]; ++e) if (si[e] <= si[e - 1]) { s_sorted = false; break; }
    f->canonical = mono && s_sorted;
    return f;
  }
  // general product with a dense accumulator over the columns touched by a row (marker array)
  std::vector<double> acc(static_cast<size_t>(a.ncol), 0.0);
  std::vector<long long> mark(static_cast<size_t>(a.ncol), -1);
  std::vector<int32_t> touched;
  for (long long r = 0; r < srows; ++r) {
    f->ptr[static_cast<size_t>(r)] = static_cast<long long>(f->idx.size());
    touched.clear();
    double bb = 0.0;
    for (long long e = sp[r]; e < sp[r + 1]; ++e) {
      const long long k = si[e];
      if (k < 0 || k >= a.rows) { delete f; throw std::runtime_error("linform apply: column index out of range"); }
      const double s = sv[e];
      bb += s * a.b[static_cast<size_t>(k)];
      for (long long q = a.ptr[static_cast<size_t>(k)]; q < a.ptr[static_cast<size_t>(k) + 1]; ++q) {
        const int32_t c = a.idx[static_cast<size_t>(q)];
        if (mark[static_cast<size_t>(c)] != r) { mark[static_cast<size_t>(c)] = r; acc[static_cast<size_t>(c)] = 0.0; touched.push_back(c); }
        acc[static_cast<size_t>(c)] += s * a.val[static_cast<size_t>(q)];
      }
    }
    std::sort(touched.begin(), touched.end());
    for (int32_t c : touched) if (acc[static_cast<size_t>(c)] != 0.0) { f->idx.push_back(c); f->val.push_back(acc[static_cast<size_t>(c)]); }   // (zero sums dropped: scipy's csr @ csr)
    f->b[static_cast<size_t>(r)] = bb;
  }
  f->ptr[static_cast<size_t>(srows)] = static_cast<long long>(f->idx.size());
  return f;
}
// S a with S a dense constant (srows x a.rows, row-major): exact zeros of S carry no entry (scipy's csr_matrix(dense))
inline LinFormH* lf_apply_dense(const LinFormH& a, long long srows, const double* S) {
  const long long k = a.rows;
  std::vector<long long> sp(static_cast<size_t>(srows) + 1, 0);
  lm_par_for(srows, 16, [&](lm_i64 lo, lm_i64 hi) {
    for (lm_i64 r = lo; r < hi; ++r) {
      long long c = 0;
      const double* row = S + r * k;
      for (long long j = 0; j < k; ++j) c += row[j] != 0.0;
      sp[static_cast<size_t>(r) + 1] = c;
    }
  });
  for (long long r = 0; r < srows; ++r) sp[static_cast<size_t>(r) + 1] += sp[static_cast<size_t>(r)];
  const long long snnz = sp[static_cast<size_t>(srows)];
  if (lf_is_selection(a)) {
    auto* f = new LinFormH();
    f->rows = srows; f->ncol = a.ncol;
    f->b.assign(static_cast<size_t>(srows), 0.0);
    f->idx.resize(static_cast<size_t>(snnz));
    f->val.resize(static_cast<size_t>(snnz));
    bool mono = true;
    for (long long r = 1; r < k && mono; ++r) mono = a.idx[static_cast<size_t>(r)] > a.idx[static_cast<size_t>(r - 1)];
    lm_par_for(srows, 16, [&](lm_i64 lo, lm_i64 hi) {
      for (lm_i64 r = lo; r < hi; ++r) {
        const double* row = S + r * k;
        long long w = sp[static_cast<size_t>(r)];
        for (long long j = 0; j < k; ++j) if (row[j] != 0.0) { f->idx[static_cast<size_t>(w)] = a.idx[static_cast<size_t>(j)]; f->val[static_cast<size_t>(w)] = row[j]; ++w; }
      }
    });
    f->ptr.swap(sp);
    f->canonical = mono;
    return f;
  }
  std::vector<int32_t> si(static_cast<size_t>(snnz));
  std::vector<double> sv(static_cast<size_t>(snnz));
  lm_par_for(srows, 16, [&](lm_i64 lo, lm_i64 hi) {
    for (lm_i64 r = lo; r < hi; ++r) {
      const double* row = S + r * k;
      long long w = sp[static_cast<size_t>(r)];
      for (long long j = 0; j < k; ++j) if (row[j] != 0.0) { si[static_cast<size_t>(w)] = static_cast<int32_t>(j); sv[static_cast<size_t>(w)] = row[j]; ++w; }
    }
  });
  return lf_apply_csr(a, srows, sp.data(), si.data(), sv.data());
}
inline LinFormH* lf_vstack(LinFormH* const* parts, int n) {
  auto* f = new LinFormH();
  if (n > 0) f->ncol = parts[0]->ncol;
  long long rows = 0, nnz = 0;
  for (int k = 0; k < n; ++k) { rows += parts[k]->rows; nnz += parts[k]->nnz(); f->canonical = f->canonical && parts[k]->canonical; }
  f->rows = rows;
  f->ptr.reserve(static_cast<size_t>(rows) + 1);
  f->idx.reserve(static_cast<size_t>(nnz));
  f->val.reserve(static_cast<size_t>(nnz));
  f->b.reserve(static_cast<size_t>(rows));
  long long off = 0;
  for (int k = 0; k < n; ++k) {
    const LinFormH& p = *parts[k];
    for (long long r = 0; r < p.rows; ++r) f->ptr.push_back(off + p.ptr[static_cast<size_t>(r)]);
    f->idx.insert(f->idx.end(), p.idx.begin(), p.idx.end());
    f->val.insert(f->val.end(), p.val.begin(), p.val.end());
    f->b.insert(f->b.end(), p.b.begin(), p.b.end());
    off += p.nnz();
  }
  f->ptr.push_back(off);
  return f;
}

}  // namespace dnlp
