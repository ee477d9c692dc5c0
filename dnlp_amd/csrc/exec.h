// Execution-space abstraction shared by the HIP product build and the host reference build.
//
// The interior-point driver (ipm_core.h) and the elementwise atom rules (atom_math.h) are
// single-source: every per-element formula is a DNLP_HD lambda.  hipcc instantiates them as
// gfx950 kernels through HipExec (exec_hip.h).  The test oracle (oracle/host_exec.h, outside
// this package) instantiates the SAME text as host loops so the algorithm can be pinned
// against known answers without a GPU; libdnlp_hip.so contains no host execution path.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DNLP_HD __host__ __device__
#else
#define DNLP_HD
#endif

namespace dnlp {

using i64 = int64_t;
using i32 = int32_t;

constexpr double kInf = std::numeric_limits<double>::infinity();

}  // namespace dnlp
