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
#include <stdexcept>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DNLP_HD __host__ __device__
#else
#define DNLP_HD
#endif

// DNLP_DEVICE_PASS is 1 while hipcc compiles the gfx950 side of a translation unit.  The
// single-source classes are __host__ __device__ so that the batched solver (exec_block.h) can
// run the whole interior-point loop inside one kernel, one workgroup per problem instance;
// host-only facilities (exceptions, clocks, stdio) are compiled out of that pass.
#if defined(__HIP_DEVICE_COMPILE__)
#define DNLP_DEVICE_PASS 1
#define DNLP_FAIL(msg) __builtin_trap()
#else
#define DNLP_DEVICE_PASS 0
#define DNLP_FAIL(msg) throw std::runtime_error(msg)
#endif

// The in-kernel execution space keeps its solver objects (exec space, model, KKT, interior-point state) in LDS, but a
// member function that is not inlined receives a GENERIC `this`: every field read is then a flat_load (address-space
// check, both wait counters, ~2x the latency of ds_read) — Ipm::step() alone had 351 of them and not one ds_read.
// DNLP_THIS_IN_LDS(E) at the top of such a function tells the compiler what the kernel knows (LLVM's InferAddressSpaces
// takes llvm.assume(is.shared(p)) as proof): field accesses through `this` become LDS instructions.
#if DNLP_DEVICE_PASS
#define DNLP_THIS_IN_LDS(E) do { if constexpr (E::objects_in_lds) __builtin_assume(__builtin_amdgcn_is_shared(this)); } while (0)
#define DNLP_PTR_IN_LDS(E, p) do { if constexpr (E::objects_in_lds) __builtin_assume(__builtin_amdgcn_is_shared(p)); } while (0)
#else
#define DNLP_THIS_IN_LDS(E) do { } while (0)
#define DNLP_PTR_IN_LDS(E, p) do { } while (0)
#endif

namespace dnlp {

using i64 = int64_t;
using i32 = int32_t;

// A vector of the solver (iterate, direction, residual, factor values ...) as the solver objects hold it: a plain
// double* in every execution space, plus — in a space whose vectors ALL live in LDS (E::vectors_in_lds: the packed batch
// kernel of batch.h) — the knowledge that it does: every read of the member tells the compiler so, and the element
// accesses through the copy (`const double* xx = x;` before a lambda, as everywhere in ipm_core.h / model.h) become
// ds_read / ds_write instead of flat instructions (address-space check, both wait counters, the texture path).
// The member is STORED as an LDS pointer (32 bits) and converted to a generic one where it is read: LLVM's address-space
// inference follows a generic pointer that comes out of an addrspacecast from LDS and rewrites the accesses through it.
// (Measured on a ten-line kernel, tools/micro/lds_infer.hip: a generic -> LDS -> generic round trip of a loaded generic
//  pointer is folded away and infers nothing; llvm.assume(is.shared(p)) on every read works but crashed this compiler's
//  loop-deletion pass on Ipm::update_mu, where the reads sit inside loops after inlining.)
template <class E, bool InLds = E::vectors_in_lds>
struct VecP {
  double* p = nullptr;
  DNLP_HD VecP() {}
  DNLP_HD VecP(double* q) : p(q) {}
  DNLP_HD VecP& operator=(double* q) { p = q; return *this; }
  DNLP_HD operator double*() const { return p; }
};
#if defined(__HIPCC__)
template <class E>
struct VecP<E, true> {
  typedef __attribute__((address_space(3))) double lds_t;
  lds_t* p = nullptr;
  __device__ VecP() {}
  __device__ VecP(double* q) : p((lds_t*)q) {}
  __device__ VecP& operator=(double* q) { p = (lds_t*)q; return *this; }
  __device__ operator double*() const { return (double*)p; }
};
#endif
// the same statement for a vector that arrives as a function argument (callers pass solver vectors only); used at the
// top of functions, outside loops
#if DNLP_DEVICE_PASS
#define DNLP_VEC_IN_LDS(E, p) do { if constexpr (E::vectors_in_lds) __builtin_assume(__builtin_amdgcn_is_shared(p)); } while (0)
#else
#define DNLP_VEC_IN_LDS(E, p) do { } while (0)
#endif

constexpr double kInf = std::numeric_limits<double>::infinity();
struct D2 { double first, second; };     // pair of reduction results (E::min2)
// several reductions in ONE pass (E::reduce_multi<NM, NS>): NM maxima (NaN -> +inf, as E::max) and NS sums
struct RMulti { double mx[4]; double sm[4]; };

}  // namespace dnlp

#include <string>
#include <vector>

namespace dnlp {

// Traits shared by the execution spaces that are driven from the host (HostExec in oracle/,
// HipExec in exec_hip.h): an iteration log, host-side control memory, host-only algorithm
// variants (Lanczos bound, condensed multiplier start).
struct HostControlled {
  static constexpr bool has_log = true;
  static constexpr bool has_host_control = true;
  static constexpr bool objects_in_lds = false;
  static constexpr bool vectors_in_lds = false;
  static constexpr int kFilterCap = 1024;
  // largest order the space's pivoted (Bunch-Kaufman) factorisation accepts; HipExec narrows it (its solve keeps
  // the vector in LDS), the host space's LAPACK backend has no such limit
  static constexpr long long kPivotedMaxOrder = 1LL << 40;
  // a dense m x m condensed least-squares multiplier start on the space's GEMM (HipExec: FP64 MFMA)
  static constexpr bool has_condensed_ls = false;
  struct Log {
    std::vector<std::string> lines;
    void append(const Log& o) { lines.insert(lines.end(), o.lines.begin(), o.lines.end()); }
  };
  template <class T> T* ctl_alloc(size_t n) { return static_cast<T*>(std::calloc(n ? n : 1, sizeof(T))); }
};

}  // namespace dnlp
