// Which of three ways makes LLVM emit LDS instructions for accesses through a pointer that was loaded from an object in
// LDS (round 4, exec.h VecP): workA plain (flat), workB generic->LDS->generic round trip (flat: folded away), workC the
// member stored as an LDS pointer (ds_read / ds_write), workD llvm.assume(is.shared) on the loaded value (ds_*).
//   hipcc --offload-arch=gfx950 -O3 -c lds_infer.hip --save-temps; grep -E "flat_|ds_" lds_infer-hip-amdgcn-amd-amdhsa-gfx950.s
#include <hip/hip_runtime.h>
typedef __attribute__((address_space(3))) double lds_double;
template <class F> __device__ void map(int n, F f) { for (int i = threadIdx.x; i < n; i += 64) f(i); __syncthreads(); }
struct ObjA { double* x; double* y; int n; };          // plain
struct ObjC { lds_double* x; lds_double* y; int n; };  // typed members
__device__ __attribute__((noinline)) void workA(ObjA* o) {
  __builtin_assume(__builtin_amdgcn_is_shared(o));
  double* x = o->x; const double* y = o->y;
  map(o->n, [=](int i) { x[i] = 2.0 * y[i] + x[i]; });
}
__device__ __attribute__((noinline)) void workB(ObjA* o) {   // round trip
  __builtin_assume(__builtin_amdgcn_is_shared(o));
  double* x = (double*)(lds_double*)o->x; const double* y = (const double*)(const lds_double*)o->y;
  map(o->n, [=](int i) { x[i] = 2.0 * y[i] + x[i]; });
}
__device__ __attribute__((noinline)) void workC(ObjC* o) {   // typed member, generic locals
  __builtin_assume(__builtin_amdgcn_is_shared(o));
  double* x = (double*)o->x; const double* y = (const double*)o->y;
  map(o->n, [=](int i) { x[i] = 2.0 * y[i] + x[i]; });
}
__device__ __attribute__((noinline)) void workD(ObjA* o) {   // assume on values
  __builtin_assume(__builtin_amdgcn_is_shared(o));
  double* x = o->x; const double* y = o->y;
  __builtin_assume(__builtin_amdgcn_is_shared(x)); __builtin_assume(__builtin_amdgcn_is_shared(y));
  map(o->n, [=](int i) { x[i] = 2.0 * y[i] + x[i]; });
}
__global__ void k(double* out, int n) {
  extern __shared__ double pool[];
  __shared__ ObjA a; __shared__ ObjC c;
  if (threadIdx.x == 0) { a.x = pool; a.y = pool + n; a.n = n; c.x = (lds_double*)pool; c.y = (lds_double*)(pool + n); c.n = n; }
  __syncthreads();
  workA(&a); workB(&a); workC(&c); workD(&a);
  out[threadIdx.x] = pool[threadIdx.x];
}
