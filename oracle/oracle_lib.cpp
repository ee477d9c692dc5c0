// TEST INFRASTRUCTURE — host build of the solver core behind the same C surface as the
// product library, exported with the prefix orc_ (liboracle_dnlp.so).  See host_exec.h.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include "host_exec.h"
#include "../dnlp_amd/csrc/capi_impl.h"

DNLP_DEFINE_CAPI(orc_, dnlp::HostExec, orc_problem)

// CPU-baseline switch (bench.py): route the pivoted dense factorisation through LAPACK DSYTRF/DSYTRS
// of the given shared library; returns the BLAS thread count (>= 1) or 0 when the library or its
// symbols are missing.  threads > 0 sets the BLAS thread count first.
extern "C" int orc_use_lapack(const char* path, int threads) {
  dnlp::HostLapack& L = dnlp::HostLapack::get();
  if (!path || !*path) { L.sytrf = nullptr; L.sytrs = nullptr; return 0; }
  if (!L.load(path)) return 0;
  if (threads > 0 && L.set_threads) L.set_threads(threads);
  return L.get_threads ? L.get_threads() : 1;
}

// Seconds of one DSYTRF of a seeded symmetric indefinite matrix of order n with `threads` BLAS
// threads (bench.py picks the thread count that factors fastest on this box: OpenBLAS's DSYTRF does
// not scale to every core of a large host).  Negative when no LAPACK is loaded.
extern "C" double orc_lapack_probe(int n, int threads) {
  dnlp::HostLapack& L = dnlp::HostLapack::get();
  if (!L.sytrf) return -1.0;
  if (threads > 0 && L.set_threads) L.set_threads(threads);
  std::vector<double> A(static_cast<size_t>(n) * n);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (int j = 0; j < n; ++j)
    for (int i = j; i < n; ++i) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      A[static_cast<size_t>(i) + static_cast<size_t>(j) * n] = static_cast<double>(s >> 11) / 9007199254740992.0 - 0.5;
    }
  std::vector<int> ipiv(static_cast<size_t>(n));
  int info = 0, lwork = -1;
  double wq = 0.0;
  L.sytrf("L", &n, A.data(), &n, ipiv.data(), &wq, &lwork, &info);
  lwork = static_cast<int>(wq) > n ? static_cast<int>(wq) : n;
  std::vector<double> work(static_cast<size_t>(lwork));
  const double t0 = dnlp::now_sec();
  L.sytrf("L", &n, A.data(), &n, ipiv.data(), work.data(), &lwork, &info);
  return dnlp::now_sec() - t0;
}

// CPU-baseline switch: unpivoted dense factorisations of order >= 512 take the blocked LDL^T whose trailing update is
// DGEMM (the device's algorithm on the host's BLAS); needs orc_use_lapack first.  Returns 1 when the BLAS has DGEMM / DTRSM.
extern "C" int orc_use_blocked_ldlt(int on) {
  dnlp::HostLapack& L = dnlp::HostLapack::get();
  L.blocked_unpivoted = on != 0 && L.gemm && L.trsm;
  return L.blocked_unpivoted ? 1 : 0;
}

// GFLOP/s of one square DGEMM of order n with `threads` BLAS threads: the rate the blocked factorisation can approach.
extern "C" double orc_dgemm_probe(int n, int threads) {
  dnlp::HostLapack& L = dnlp::HostLapack::get();
  if (!L.gemm) return -1.0;
  if (threads > 0 && L.set_threads) L.set_threads(threads);
  std::vector<double> A(static_cast<size_t>(n) * n, 0.5), B(static_cast<size_t>(n) * n, 0.25), C(static_cast<size_t>(n) * n, 0.0);
  const double one = 1.0, zero = 0.0;
  L.gemm("N", "T", &n, &n, &n, &one, A.data(), &n, B.data(), &n, &zero, C.data(), &n);      // warm the threads
  const double t0 = dnlp::now_sec();
  L.gemm("N", "T", &n, &n, &n, &one, A.data(), &n, B.data(), &n, &zero, C.data(), &n);
  return 2.0 * n * n * n / (dnlp::now_sec() - t0) / 1e9;
}

// Seconds of the four phases of the last blocked unpivoted LDL^T (diagonal blocks, DTRSM, W copy, DGEMM); zeros unless
// DNLP_HOST_LDLT_TIMING is set.
extern "C" void orc_blocked_ldlt_phases(double* out4) {
  dnlp::HostLapack& L = dnlp::HostLapack::get();
  for (int k = 0; k < 4; ++k) out4[k] = L.last_phases[k];
}
