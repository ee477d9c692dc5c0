// Template-specialised batch solver, part 5: the per-template kernel as TEXT.
//
// wave_batch.h's kernels serve every template: what a template fixes — sizes, where each table of the plan block and each
// vector of an instance lies — they read from the wavefront's state record in LDS, a pointer load and an address
// computation in front of every loop, loop bounds in registers.  For the templates a rank solves by the thousand (the C5
// members of BASELINE.json; reference role: the serial loop cvxpy/problems/problem.py:1256-1269 into IPOPT,
// ipopt_nlpif.py:140-170) the library compiles ONE kernel per template at run time (hiprtc, cached on disk by source
// hash: fused_rtc.h) from
//     a prelude                 what the embedded headers expect from exec.h / <cmath> / <algorithm>, hiprtc-clean
//     atom_math.h               the unary atom rules            }
//     ipm_options.h             options, status codes           }  the very text the library's own kernels and the
//     wave_hdr.h, wave_args.h   plan header, launch arguments   }  host lane of the test oracle are compiled from
//     namespace wspec           THE TEMPLATE: every size, table offset and vector place as a literal (this file)
//     wave_ops.h, wave_ipm.h    reductions, the interior-point loop (its WK / WT / WV accessors resolve to wspec::)
//     wave_spec_kernel.h        static LDS for plan + work tables + shares, the lane policy, the kernel
//     wave_gen_rt.h + generated the static-pattern LDL^T (factorisation, both substitutions) as straight-line phases for
//                               THIS template's levels (wave_gen.h), in place of the walk over the plan's level tables
// The algorithm text is not forked: the same wave_ipm.h is pinned bit for bit on the CPU (tests/test_wave_ipm_cpu.py) and
// the per-template kernel is compared with the library's own on the device (tests/test_wave_spec.py).
//
// This header is plain host C++ (no HIP): the test oracle prints the same text (oracle_lib.cpp orc_wave_spec_source) so
// that a CPU test can hand it to hiprtc without a GPU.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "wave_gen.h"
#include "wave_ipm.h"

namespace dnlp {

// host stand-in for a lane policy: only its types matter (WaveIpm<P>::layout is run once on the host to learn the offsets)
struct WaveProbeLanes {
  typedef double D;
  typedef const i32 I;
  static constexpr int lanes = 1;
  static constexpr bool hoist = false;
  static int lane() { return 0; }
  static void sync() {}
  static double sum(double v) { return v; }
  static double vmax(double v) { return v; }
  template <int N> static void sum_n(double (&)[N]) {}
  template <int N> static void vmax_n(double (&)[N]) {}
  static double now() { return dnlp::now_sec(); }
  static int tab_load(const i32*, int) { return 0; }
  static int tab_at(const i32* tab, int, int idx, int) { return tab[idx]; }
  static int uni(int v) { return v; }
  template <int SL> static double row_get(const double (&a)[SL], int row) { return a[row]; }
};

constexpr int kWaveSpecRecBytesMax = 1344;      // (what the host sizes a launch with; the kernel asserts its record — 1334 B today — fits;
                                                //  a profile build's record carries 224 B of cycle counters more: kWaveSpecRecBytesProf)
constexpr int kWaveSpecRecBytesProf = 1568;

// the largest number of wavefronts per workgroup whose shares fit a compute unit's LDS beside the 16-bit plan (0: none)
// (tables_global: the plan prefix and the work tables stay in global memory — wave_spec_kernel.h kTabGlobal)
inline int wave_spec_max_waves(const WaveHdr& h, size_t gen_words, bool prof = false, bool tables_global = false) {
  const size_t cap = 160 * 1024 - 512;
  const size_t plan_b = tables_global ? 64 : ((static_cast<size_t>(h.keep_gen) + 7) & ~static_cast<size_t>(7)) * 2 + ((gen_words + 3) & ~static_cast<size_t>(3)) * 4;
  const size_t share_b = (prof ? kWaveSpecRecBytesProf : kWaveSpecRecBytesMax) + static_cast<size_t>(h.state_doubles) * 8 + 16;
  if (plan_b + share_b > cap) return 0;
  const size_t k = (cap - plan_b) / share_b;
  return static_cast<int>(k > 8 ? 8 : k);
}

// namespace wspec of a template: the literals behind WK / WT / WV / WDIR / WCSR / WCOO (wave_ipm.h)
inline std::string wave_spec_constants(const std::vector<i32>& blk, int nw, size_t gen_words = 0, bool prof = false, bool tables_global = false) {
  const WaveHdr& h = *reinterpret_cast<const WaveHdr*>(blk.data());
  typedef WaveIpm<WaveProbeLanes> W;
  W::WState S;
  std::vector<double> vecs(static_cast<size_t>(h.state_doubles) + 8, 0.0);
  const i32 laid = W::layout((W::WS*)&S, &h, blk.data(), vecs.data());      // (the cast matters only to the device pass, where WS is LDS-qualified)
  if (laid != h.state_doubles) throw std::runtime_error("wave codegen: layout and wave_state_doubles disagree");
  std::string s = "namespace wspec {\n";
  char b[256];
  auto put = [&](const char* pre, const char* name, long long v) { std::snprintf(b, sizeof b, "constexpr int %s%s = %lld;\n", pre, name, v); s += b; };
#define W_PUT_K(f) put("k_", #f, static_cast<long long>(S.f));
#define W_PUT_T(f) put("t_", #f, static_cast<long long>(S.f - blk.data()));
#define W_PUT_V(f) put("v_", #f, static_cast<long long>(S.f - vecs.data()));
  WAVE_SIZE_FIELDS(W_PUT_K)
  WAVE_TAB_FIELDS(W_PUT_T)
  WAVE_VEC_FIELDS(W_PUT_V)
#undef W_PUT_K
#undef W_PUT_T
#undef W_PUT_V
  s += "constexpr int v_dir[3][7] = {";
  for (int a = 0; a < 3; ++a) {
    s += a ? ", {" : "{";
    for (int k = 0; k < 7; ++k) { std::snprintf(b, sizeof b, "%s%lld", k ? ", " : "", static_cast<long long>(S.dir[a][k] - vecs.data())); s += b; }
    s += "}";
  }
  s += "};\n";
  auto csr = [&](const char* name, const W::WCsr& M) {
    std::snprintf(b, sizeof b, "constexpr int t_%s_ptr = %lld, t_%s_idx = %lld, k_%s_rows = %d, k_%s_val = %d;\n", name,
                  static_cast<long long>(M.ptr - blk.data()), name, static_cast<long long>(M.idx - blk.data()), name, M.rows, name, M.val);
    s += b;
  };
  csr("G", S.G); csr("Mg", S.Mg); csr("MJ", S.MJ); csr("Mw", S.Mw); csr("MH", S.MH);
  s += "constexpr int k_G_id = 0, k_Mg_id = 1, k_MJ_id = 2, k_Mw_id = 3, k_MH_id = 4;\n";
  auto coo = [&](const char* name, const W::WCoo& M) {
    std::snprintf(b, sizeof b, "constexpr int t_%s_ptr = %lld, t_%s_ent = %lld, t_%s_src = %lld, t_%s_heavy = %lld, k_%s_nout = %d, k_%s_nheavy = %d;\n", name,
                  static_cast<long long>(M.ptr - blk.data()), name, static_cast<long long>(M.ent - blk.data()), name, static_cast<long long>(M.src - blk.data()),
                  name, static_cast<long long>(M.heavy - blk.data()), name, M.nout, name, M.nheavy);
    s += b;
  };
  coo("jr", S.jr); coo("jc", S.jc); coo("hs", S.hs);
  put("", "kStateDoubles", h.state_doubles);
  put("", "kPlanInts", h.keep_gen);      // (what a kernel with generated LDL^T phases stages of the block)
  put("", "kGenWords", static_cast<long long>(gen_words));
  put("", "kNW", nw);
  s += tables_global ? "constexpr bool kTabGlobal = true;\n" : "constexpr bool kTabGlobal = false;\n";
  put("", "kRecBytesMax", prof ? kWaveSpecRecBytesProf : kWaveSpecRecBytesMax);
  s += "}  // namespace wspec\n";
  return s;
}

// what the embedded headers expect around them, without a standard library (hiprtc)
inline const char* wave_spec_prelude() {
  return
      "#define DNLP_RTC 1\n"
      "#define DNLP_WAVE_SPEC 1\n"
      "#define DNLP_WAVE_GEN 1\n"
      "#define DNLP_HD __device__\n"
      "#define DNLP_DEVICE_PASS 1\n"
      "typedef __INT16_TYPE__ int16_t;\ntypedef __INT32_TYPE__ int32_t;\ntypedef __INT64_TYPE__ int64_t;\ntypedef __UINT64_TYPE__ uint64_t;\n"
      "namespace dnlp {\n"
      "using i64 = int64_t;\nusing i32 = int32_t;\n"
      "constexpr double kInf = __builtin_inf();\n"
      "struct D2 { double first, second; };\n"
      "}  // namespace dnlp\n"
      // <algorithm> / <cmath> as far as wave_ipm.h uses them (std::max / std::min: libstdc++'s definitions — the second
      // argument wins only when it compares greater / less, NaN conventions included)
      "namespace dnlp_std {\n"
      "template <class T> __device__ constexpr const T& max(const T& a, const T& b) { return (a < b) ? b : a; }\n"
      "template <class T> __device__ constexpr const T& min(const T& a, const T& b) { return (b < a) ? b : a; }\n"
      "__device__ inline bool isfinite(double v) { return __builtin_isfinite(v); }\n"
      "__device__ inline double pow(double a, double b) { return ::pow(a, b); }\n"
      "__device__ inline double sqrt(double a) { return ::sqrt(a); }\n"
      "__device__ inline double log(double a) { return ::log(a); }\n"
      "__device__ inline double exp(double a) { return ::exp(a); }\n"
      "__device__ inline double fabs(double a) { return ::fabs(a); }\n"
      "}  // namespace dnlp_std\n"
      "#define std dnlp_std\n";
}

// development aid: with DNLP_WAVE_SRC_DIR set, a header's text is read from <dir>/<name> (stripped of its #pragma once and
// #include lines, as the build does for the embedded copy) instead of taken from the library — the kernel text can then be
// edited without rebuilding the library
inline std::string wave_spec_text(const char* name, const char* embedded) {
  const char* dir = std::getenv("DNLP_WAVE_SRC_DIR");
  if (!dir || !*dir) return embedded;
  FILE* fp = std::fopen((std::string(dir) + "/" + name).c_str(), "r");
  if (!fp) return embedded;
  std::string out, line;
  char buf[4096];
  while (std::fgets(buf, sizeof buf, fp)) {
    line = buf;
    size_t k = 0;
    while (k < line.size() && (line[k] == ' ' || line[k] == '\t')) ++k;
    if (line.compare(0, 12, "#pragma once") == 0 || line.compare(k, 8, "#include") == 0) continue;
    out += line;
  }
  std::fclose(fp);
  return out;
}

// the translation unit of a template's kernel (entry point: dnlp_wave_spec_kernel)
inline std::string wave_spec_source(const std::vector<i32>& blk, int nw, const WaveGen& gen, bool prof = false, bool tables_global = false) {
  static const char* atom_math_text =
#include "atom_math_src.inc"
      ;
  static const char* ipm_options_text =
#include "ipm_options_src.inc"
      ;
  static const char* wave_hdr_text =
#include "wave_hdr_src.inc"
      ;
  static const char* wave_args_text =
#include "wave_args_src.inc"
      ;
  static const char* wave_ops_text =
#include "wave_ops_src.inc"
      ;
  static const char* wave_ipm_text =
#include "wave_ipm_src.inc"
      ;
  static const char* wave_gen_rt_text =
#include "wave_gen_rt_src.inc"
      ;
  static const char* wave_spec_kernel_text =
#include "wave_spec_kernel_src.inc"
      ;
  std::string s = prof ? "#define DNLP_WAVE_PROF 1\n" : "";
  s += wave_spec_prelude();
  s += wave_spec_text("atom_math.h", atom_math_text);
  s += wave_spec_text("ipm_options.h", ipm_options_text);
  s += wave_spec_text("wave_hdr.h", wave_hdr_text);
  s += wave_spec_text("wave_args.h", wave_args_text);
  s += wave_spec_constants(blk, nw, gen.G.size(), prof, tables_global);
  s += "namespace wspec { constexpr bool kSolve2Lds = false; }\n";      // (a workgroup-kernel matter: wave_wg_lds_ranges)
  s += wave_spec_text("wave_ops.h", wave_ops_text);
  s += wave_spec_text("wave_ipm.h", wave_ipm_text);
  s += wave_spec_text("wave_spec_kernel.h", wave_spec_kernel_text);      // (the lane policy P: the generated functions below are templates over it)
  s += wave_spec_text("wave_gen_rt.h", wave_gen_rt_text);
  s += gen.code;
  return s;
}

// the translation unit of a template's WORKGROUP-per-instance kernel (wave_wg_kernel.h; entry point: dnlp_wave_wg_kernel):
// `gen` must have been generated for 64 x nwg lanes per phase
// Which vectors of an instance the workgroup-per-instance kernel keeps in LDS (the rest: the workgroup's slab of global
// memory): by priority the three arrays a single linear solve runs on (rhs sol res), the factor's values (svals), the three
// arrays of the mu oracle's second system in front of them (the centering direction's cx czL czU: contiguous with rhs sol res
// in wave_ipm.h layout) — as far as 160 KB minus the wavefronts' records hold them.  Two ranges of offsets.
// (window_doubles: the factorisation's LDS windows — wave_gen.h; free_doubles: what the chosen ranges leave)
// (share: workgroups meant to share a compute unit — each gets 160 KB / share)
inline std::string wave_wg_lds_ranges(const std::vector<i32>& blk, int nwg, bool enable, int stage_words = 0, int window_doubles = 0, long long* free_doubles = nullptr,
                                      int share = 1) {
  const WaveHdr& h = *reinterpret_cast<const WaveHdr*>(blk.data());
  typedef WaveIpm<WaveProbeLanes> W;
  W::WState S;
  std::vector<double> vecs(static_cast<size_t>(h.state_doubles) + 8, 0.0);
  W::layout((W::WS*)&S, &h, blk.data(), vecs.data());
  auto ev = [](long long n) { return (n + 1) & ~1LL; };
  // (beside them: the wavefronts' records, the staged kernel arguments and reduction partials, the narrow phases' staging buffer)
  const long long cap = (160 * 1024 / (share > 0 ? share : 1) - 2048 - static_cast<long long>(nwg) * 1600 - 1024 - 4LL * stage_words) / 8 - window_doubles;
  const long long nm = ev(h.N + h.m), a0 = S.svals - vecs.data(), a1 = a0 + ev(h.sp_nvals);
  const long long c0 = S.rhs - vecs.data(), c1 = (S.res - vecs.data()) + nm, b0 = S.dir[2][0] - vecs.data();
  long long r0a = 0, r0b = 0, r1a = 0, r1b = 0;
  if (enable) {
    const bool six = b0 + 3 * nm == c0;          // (cx czL czU rhs sol res one block)
    if (six && (a1 - a0) + (c1 - b0) <= cap) { r0a = a0; r0b = a1; r1a = b0; r1b = c1; }
    else if ((a1 - a0) + (c1 - c0) <= cap) { r0a = a0; r0b = a1; r1a = c0; r1b = c1; }
    else if (c1 - c0 <= cap) { r0a = c0; r0b = c1; }
  }
  if (free_doubles) *free_doubles = cap - ((r0b - r0a) + (r1b - r1a));
  char b[256];
  std::snprintf(b, sizeof b, "namespace wspec { constexpr int kLds0a = %lld, kLds0b = %lld, kLds1a = %lld, kLds1b = %lld, kLdsDoubles = %lld; }\n", r0a, r0b, r1a, r1b,
                (r0b - r0a) + (r1b - r1a));
  // (kSolve2Lds: BOTH systems' solve arrays are in LDS — then every right-hand side of ldl_solve is: wave_ipm.h)
  const bool solve2 = r1b > r1a && r1a == b0 && r1b == c1;
  return std::string(b) + (solve2 ? "namespace wspec { constexpr bool kSolve2Lds = true; }\n" : "namespace wspec { constexpr bool kSolve2Lds = false; }\n");
}

// the generated phases of a template's workgroup kernel: 64 x nwg lanes per phase, the factorisation's windows as large as the LDS
// that the vectors' ranges leave (half each, at most 2048 doubles)
inline WaveGen wave_wg_generate(const std::vector<i32>& blk, int nwg, bool lds_vectors = true, int share = 1) {
  const bool on = !(std::getenv("DNLP_WAVE_WG_WINDOWS") && std::atoi(std::getenv("DNLP_WAVE_WG_WINDOWS")) == 0);
  WaveGen plain = wave_generate(blk, 64 * nwg);          // (what the staging buffer takes is known only from a generation)
  if (!on) return plain;
  long long free_d = 0;
  (void)wave_wg_lds_ranges(blk, nwg, lds_vectors, plain.stage_words, 0, &free_d, share);
  // the rows' window first (a level's span is short), the products' window gets the rest
  const int ww = static_cast<int>(std::max(0LL, std::min(free_d / 4, 2048LL))) & ~1;
  const int sw = static_cast<int>(std::max(0LL, std::min(free_d - ww, 2048LL))) & ~1;
  if (ww < 16) return plain;
  return wave_generate(blk, 64 * nwg, ww, sw);
}

inline std::string wave_wg_source(const std::vector<i32>& blk, int nwg, const WaveGen& gen, bool prof = false, int bound_threads = 0, bool lds_vectors = true, int share = 1) {
  static const char* atom_math_text =
#include "atom_math_src.inc"
      ;
  static const char* ipm_options_text =
#include "ipm_options_src.inc"
      ;
  static const char* wave_hdr_text =
#include "wave_hdr_src.inc"
      ;
  static const char* wave_args_text =
#include "wave_args_src.inc"
      ;
  static const char* wave_ops_text =
#include "wave_ops_src.inc"
      ;
  static const char* wave_ipm_text =
#include "wave_ipm_src.inc"
      ;
  static const char* wave_gen_rt_text =
#include "wave_gen_rt_src.inc"
      ;
  static const char* wave_wg_kernel_text =
#include "wave_wg_kernel_src.inc"
      ;
  std::string s = prof ? "#define DNLP_WAVE_PROF 1\n" : "";
  s += wave_spec_prelude();
  s += wave_spec_text("atom_math.h", atom_math_text);
  s += wave_spec_text("ipm_options.h", ipm_options_text);
  s += wave_spec_text("wave_hdr.h", wave_hdr_text);
  s += wave_spec_text("wave_args.h", wave_args_text);
  s += wave_spec_constants(blk, nwg, gen.G.size(), prof);
  s += "namespace wspec { constexpr int kWgBound = " + std::to_string(bound_threads > 64 * nwg ? bound_threads : 64 * nwg) + "; }\n";
  s += "namespace wspec { constexpr int kStageWords = " + std::to_string(gen.stage_words) + ", kWwin = " + std::to_string(gen.wwin_doubles) +
       ", kSwin = " + std::to_string(gen.swin_doubles) + "; }\n";
  s += wave_wg_lds_ranges(blk, nwg, lds_vectors, gen.stage_words, gen.wwin_doubles + gen.swin_doubles, nullptr, share);
  s += wave_spec_text("wave_ops.h", wave_ops_text);
  s += wave_spec_text("wave_ipm.h", wave_ipm_text);
  s += wave_spec_text("wave_wg_kernel.h", wave_wg_kernel_text);
  s += wave_spec_text("wave_gen_rt.h", wave_gen_rt_text);
  s += gen.code;
  return s;
}

}  // namespace dnlp
