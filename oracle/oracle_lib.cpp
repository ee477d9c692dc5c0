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
