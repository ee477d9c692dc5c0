// Code generation for fused objective programs (BASELINE config C2): the slot program of
// fused_obj.h becomes the text of ONE specialised HIP kernel, compiled at lowering time with hiprtc
// for gfx950 and loaded with hipModuleLoadData.  The interpreter of exec_hip.h stays as the fallback
// (strided programs, no hiprtc).
//
// Why: the interpreter ends at ~20 % of HBM (decode + LDS round trip per op).  Generated code keeps
// every slot in a VGPR with a static name, constant-folds the unary rule, and — because the index
// maps are known — needs neither atomics nor a zeroed gradient:
//
//   OWNER COMPUTES.  A lane owns E consecutive entries [c, c + E) of grad f.  With unit-stride loads
//   x[off_k + i], off_k in [lo, hi], the elements that touch those entries are
//   i in [c - hi, c + E - 1 - lo]: E + W of them (W = hi - lo, the "halo", 1 for the Rosenbrock
//   chain).  The lane evaluates all of them, keeps only the adjoints that land in its own entries
//   (which ones is known at generation time: static register indices), counts an element in f only
//   when it owns the element's lowest entry, and stores its E entries once, as one vector.
//   x is read once (16-byte lanes; halo values come from neighbouring lanes' cache lines),
//   grad f is written once: the 16 n algorithmic bytes of SURVEY.md 8d, no memset, no atomics,
//   bit-reproducible.  The W redundant element evaluations per lane are FP64 flops on an HBM-bound
//   kernel.
//
// The same generator emits the element function for the persistent L-BFGS kernel (lbfgs_codegen.h).
#pragma once
#include <cinttypes>
#include <cstdio>
#include <string>
#include <vector>

#include "fused_obj.h"

namespace dnlp {

struct FusedCodegenInfo {
  int E = 4;           // grad entries per lane
  i64 lo = 0, hi = 0;  // offset range over all programs
  bool ok = false;
  std::string why;
};

inline std::string cg_double(double v) {
  char buf[64];
  if (v != v) return "__builtin_nan(\"\")";
  if (v == kInf) return "__builtin_inf()";
  if (v == -kInf) return "(-__builtin_inf())";
  std::snprintf(buf, sizeof buf, "%a", v);      // hex float: exact
  return std::string("(") + buf + ")";
}

// Can every program of the objective be generated?  (unit-stride variable loads within a small
// window; per-element constants with stride 0 or 1)
inline FusedCodegenInfo fused_codegen_plan(const std::vector<FusedSlotProg>& progs, int E) {
  FusedCodegenInfo info;
  info.E = E;
  bool any = false;
  for (const FusedSlotProg& P : progs) {
    if (P.win_extra < 0) { info.why = "strided or far-apart variable loads"; return info; }
    for (int i = 0; i < P.nops; ++i) {
      const int op = static_cast<int>(P.rec[i].code & 0xffu);
      if (op == S_LOADC && P.rec[i].stride != 0 && P.rec[i].stride != 1) { info.why = "strided constants"; return info; }
    }
    const i64 plo = P.win_lo, phi = P.win_lo + P.win_extra;
    if (!any) { info.lo = plo; info.hi = phi; any = true; }
    info.lo = std::min(info.lo, plo);
    info.hi = std::max(info.hi, phi);
  }
  if (!any) { info.why = "no program"; return info; }
  if (info.hi - info.lo > 16) { info.why = "window too wide"; return info; }
  info.ok = true;
  return info;
}

// Text of   template <bool GUARD> __device__ void dnlp_chunk(i64 c, x, consts, nfree, double (&g)[E], double& f)
// : evaluates every element that touches grad entries [c, c + E) and accumulates the owned adjoints
// into g[0..E) and the owned elements' values into f.
inline std::string fused_codegen_chunk(const std::vector<FusedSlotProg>& progs, const FusedCodegenInfo& info) {
  const int E = info.E;
  const i64 lo = info.lo, hi = info.hi;
  const int W = static_cast<int>(hi - lo);
  const int NX = E + 2 * W;            // x values a lane can touch: indices c - W .. c + E - 1 + W
  std::string s;
  char b[512];
  auto add = [&](const char* fmt, auto... a) { std::snprintf(b, sizeof b, fmt, a...); s += b; };
  // the window xr[k] = x[c - W + k], k < E + 2W, is loaded by dnlp_chunk below; the L-BFGS kernels
  // (lbfgs_codegen.h) hand in a window they computed themselves (trial point x + step * dir)
  add("#define DNLP_W %d\n#define DNLP_NX %d\n", W, NX);
  {
    i64 min_nelem = progs[0].nelem;
    for (const auto& P : progs) min_nelem = std::min(min_nelem, P.nelem);
    // every window entry inside [0, nfree) and every touched element valid in every program
    add("#define DNLP_INTERIOR(c, nfree) ((c) - DNLP_W >= 0 && (c) + DNLP_E + DNLP_W <= (nfree) && (c) - %lldLL >= 0 && "
        "(c) + DNLP_E - 1 - %lldLL < %lldLL)\n", static_cast<long long>(hi), static_cast<long long>(lo),
        static_cast<long long>(min_nelem));
  }
  s += "template <bool GUARD>\n__device__ __forceinline__ void dnlp_chunk_w(const i64 c, const double (&xr)[DNLP_NX],\n"
       "    const double* __restrict__ consts, double (&g)[DNLP_E], double& facc) {\n";
  int pi = 0;
  for (const FusedSlotProg& P : progs) {
    add("  // ---- program %d: %d ops, %d slots, %lld elements\n", pi, P.nops, P.nslots, static_cast<long long>(P.nelem));
    // elements e_j = c - hi + j, j = 0 .. E + W - 1 (all programs share the global [lo, hi] frame)
    for (int j = 0; j < E + W; ++j) {
      // the lane that owns the element's own lowest entry e_j + P.win_lo (always a valid index of
      // grad f) counts the element in f: exactly one lane per element
      const i64 t_own = j - hi + P.win_lo;
      const bool counted = t_own >= 0 && t_own < E;
      add("  { const i64 e = c - %lld + %d;\n", static_cast<long long>(hi), j);
      add("    if (!GUARD || (e >= 0 && e < %lldLL)) {\n", static_cast<long long>(P.nelem));
      for (int k = 0; k < P.nslots; ++k) add("      double s%d;\n", k);
      bool any_effect = false;
      std::string body;
      auto addb = [&](const char* fmt, auto... a) { std::snprintf(b, sizeof b, fmt, a...); body += b; };
      for (int i = 0; i < P.nops; ++i) {
        const FusedOpRec& R = P.rec[i];
        const int op = static_cast<int>(R.code & 0xffu), d = static_cast<int>((R.code >> 8) & 0xffu),
                  s1 = static_cast<int>((R.code >> 16) & 0xffu), s2 = static_cast<int>(R.code >> 24);
        switch (op) {
          case S_LOADV:   // x[e + off] = xr[j + off - lo]
            addb("      s%d = xr[%d];\n", d, static_cast<int>(j + R.off - lo));
            break;
          case S_LOADC:
            if (R.stride == 0) addb("      s%d = consts[%lld];\n", d, static_cast<long long>(R.off));
            else addb("      s%d = consts[%lld + e];\n", d, static_cast<long long>(R.off));
            break;
          case S_UNARY: {
            const int u = static_cast<int>(R.u);
            const bool sq = u == OP_POWER && R.p == 2.0 && R.q == 2.0;
            const bool to_f = R.w != 0.0;
            if (sq) {
              if (to_f) { if (counted) { addb("      facc += %s * (s%d * s%d);\n", cg_double(R.w).c_str(), s1, s1); any_effect = true; } }
              else addb("      { const double t = s%d; s%d = t * t; s%d = 2.0 * t; }\n", s1, d, s2);
              if (to_f) addb("      s%d = 2.0 * s%d;\n", s2, s1);
            } else {
              addb("      { double v, g1, g2; dnlp::unary_rules(%d, s%d, %s, %s, v, g1, g2); (void)g2;\n", u, s1,
                   cg_double(R.p).c_str(), cg_double(R.q).c_str());
              if (to_f) { if (counted) { addb("        facc += %s * v;\n", cg_double(R.w).c_str()); any_effect = true; } }
              else addb("        s%d = v;\n", d);
              addb("        s%d = g1; }\n", s2);
            }
            break; }
          case S_ADD: addb("      s%d = s%d + s%d;\n", d, s1, s2); break;
          case S_SUB: addb("      s%d = s%d - s%d;\n", d, s1, s2); break;
          case S_MUL: addb("      s%d = s%d * s%d;\n", d, s1, s2); break;
          case S_DIV: addb("      s%d = s%d / s%d;\n", d, s1, s2); break;
          case S_SCALE: addb("      s%d = %s * s%d;\n", d, cg_double(R.p).c_str(), s1); break;
          case S_ADDC: addb("      s%d = s%d + %s;\n", d, s1, cg_double(R.p).c_str()); break;
          case S_SET: addb("      s%d = %s;\n", d, cg_double(R.p).c_str()); break;
          case S_AXPB: addb("      s%d = %s * s%d + %s;\n", d, cg_double(R.p).c_str(), s1, cg_double(R.q).c_str()); break;
          case S_ACCF:
            if (counted) { addb("      facc += %s * s%d;\n", cg_double(R.p).c_str(), s1); any_effect = true; }
            break;
          default: {   // S_SCATTER to index e + off: owned entry t = j - hi + off
            const i64 t = j - hi + R.off;
            if (t >= 0 && t < E) { addb("      g[%d] += %s * s%d;\n", static_cast<int>(t), cg_double(R.p).c_str(), s1); any_effect = true; }
            break; }
        }
      }
      if (any_effect) s += body;       // an element none of whose results this lane keeps is not evaluated
      s += "    }\n  }\n";
    }
    ++pi;
  }
  s += "}\n";
  s += "template <bool GUARD>\n__device__ __forceinline__ void dnlp_chunk(const i64 c, const double* __restrict__ x,\n"
       "    const double* __restrict__ consts, const i64 nfree, double (&g)[DNLP_E], double& facc) {\n";
  s += "  double xr[DNLP_NX];\n  if (!GUARD) {\n";
  // aligned middle part as 16-byte vectors, halo as scalars
  for (int k = 0; k < W; ++k) add("    xr[%d] = x[c - %d];\n", k, W - k);
  for (int k = 0; k + 1 < E; k += 2)
    add("    { const double2 v = *reinterpret_cast<const double2*>(x + c + %d); xr[%d] = v.x; xr[%d] = v.y; }\n", k, W + k, W + k + 1);
  if (E & 1) add("    xr[%d] = x[c + %d];\n", W + E - 1, E - 1);
  for (int k = 0; k < W; ++k) add("    xr[%d] = x[c + %d];\n", W + E + k, E + k);
  s += "  } else {\n";
  s += "    for (int k = 0; k < DNLP_NX; ++k) { const i64 q = c - DNLP_W + k; xr[k] = (q >= 0 && q < nfree) ? x[q] : 0.0; }\n";
  s += "  }\n  dnlp_chunk_w<GUARD>(c, xr, consts, g, facc);\n}\n";
  return s;
}

// Common preamble of every generated translation unit: integer types, the unary rules of
// atom_math.h (text embedded at build time, atom_math_src.inc) and wavefront helpers.
inline std::string fused_codegen_preamble(int E) {
  static const char* atom_math_text =
#include "atom_math_src.inc"
      ;
  std::string s;
  s += "typedef long long i64;\ntypedef int i32;\n#define DNLP_HD __device__\n";
  s += "namespace dnlp { constexpr double kInf = __builtin_inf(); }\n";
  s += atom_math_text;
  s += "\n#define DNLP_E " + std::to_string(E) + "\n";
  // (DNLP_EMULATE: tests/test_fused_codegen.py compiles this very text with g++ and runs the lanes one
  // after the other on the host to check the generated arithmetic without a GPU)
  s += "#ifndef DNLP_EMULATE\n"
       // wavefront sum on DPP row shifts / row broadcasts, result to every lane through lane 63 (wave_ops.h has the
       // same text for the library's own kernels): a __shfl_xor butterfly is twelve ds_bpermute round trips, ~1 500
       // cycles; this is ~200
       "template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dnlp_dpp_f64(double v, double keep) {\n"
       "  const int lo = __builtin_amdgcn_update_dpp(__double2loint(keep), __double2loint(v), CTRL, ROW_MASK, 0xf, false);\n"
       "  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(keep), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);\n"
       "  return __hiloint2double(hi, lo);\n}\n"
       "__device__ __forceinline__ double dnlp_lane63(double v) {\n"
       "  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));\n}\n"
       "__device__ __forceinline__ double dnlp_wave_sum(double v) {\n"
       "  v += dnlp_dpp_f64<0x111, 0xf>(v, 0.0);\n  v += dnlp_dpp_f64<0x112, 0xf>(v, 0.0);\n"
       "  v += dnlp_dpp_f64<0x114, 0xf>(v, 0.0);\n  v += dnlp_dpp_f64<0x118, 0xf>(v, 0.0);\n"
       "  v += dnlp_dpp_f64<0x142, 0xa>(v, 0.0);\n  v += dnlp_dpp_f64<0x143, 0xc>(v, 0.0);\n"
       "  return dnlp_lane63(v);\n}\n"
       "__device__ __forceinline__ void dnlp_store_f(double facc, double* __restrict__ partial) {\n"
       "  facc = dnlp_wave_sum(facc);\n"
       "  if ((threadIdx.x & 63) == 0) partial[blockIdx.x * 4 + (threadIdx.x >> 6)] = facc;\n}\n"
       "#endif\n";
  return s;
}

// The evaluation kernel:  grad[c .. c+E) for every chunk, f partial per wavefront.
//   extern "C" __global__ void dnlp_fused_eval(x, consts, grad, partial, nfree, nchunks)
inline std::string fused_codegen_eval_source(const std::vector<FusedSlotProg>& progs, const FusedCodegenInfo& info) {
  const int E = info.E;
  std::string s = fused_codegen_preamble(E);
  s += fused_codegen_chunk(progs, info);
  char b[1024];
  s += "extern \"C\" __global__ void __launch_bounds__(256) dnlp_fused_eval(const double* __restrict__ x,\n"
       "    const double* __restrict__ consts, double* __restrict__ grad, double* __restrict__ partial,\n"
       "    const i64 nfree, const i64 nchunks) {\n"
       "  double facc = 0.0;\n"
       "  for (i64 q = static_cast<i64>(blockIdx.x) * 256 + threadIdx.x; q < nchunks; q += static_cast<i64>(gridDim.x) * 256) {\n"
       "    const i64 c = q * DNLP_E;\n"
       "    double g[DNLP_E];\n"
       "#pragma unroll\n"
       "    for (int t = 0; t < DNLP_E; ++t) g[t] = 0.0;\n"
       "    const bool interior = DNLP_INTERIOR(c, nfree);\n";
  s += "    if (interior) {\n      dnlp_chunk<false>(c, x, consts, nfree, g, facc);\n";
  // vector stores of the owned entries
  s += "      double* gp = grad + c;\n";
  for (int k = 0; k + 1 < E; k += 2) {
    std::snprintf(b, sizeof b, "      *reinterpret_cast<double2*>(gp + %d) = make_double2(g[%d], g[%d]);\n", k, k, k + 1);
    s += b;
  }
  if (E & 1) { std::snprintf(b, sizeof b, "      gp[%d] = g[%d];\n", E - 1, E - 1); s += b; }
  s += "    } else {\n      dnlp_chunk<true>(c, x, consts, nfree, g, facc);\n"
       "#pragma unroll\n      for (int t = 0; t < DNLP_E; ++t) if (c + t < nfree) grad[c + t] = g[t];\n    }\n  }\n"
       "  dnlp_store_f(facc, partial);\n}\n";
  return s;
}

}  // namespace dnlp
