// TEST INFRASTRUCTURE — host build of the solver core behind the same C surface as the
// product library, exported with the prefix orc_ (liboracle_dnlp.so).  See host_exec.h.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include "host_exec.h"
#include "../dnlp_amd/csrc/capi_impl.h"

DNLP_DEFINE_CAPI(orc_, dnlp::HostExec)
