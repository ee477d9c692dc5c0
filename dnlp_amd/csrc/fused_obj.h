// Fused native-form objective: f and grad f of an unconstrained elementwise-sum problem in ONE
// pass over the user's variables (BASELINE config C2, SURVEY.md §8d: "kernel = fused f + grad f,
// algorithmic bytes 16 n").  The per-element register program comes from dnlp_amd/fused.py
// (tape arrays fz_*); it is data, not generated code.
//
// Per element i: forward sweep over the instructions (values into the element's slots), the
// last value is the element's contribution to f; reverse sweep turns every slot into the adjoint
// of its value (each value has exactly one consumer: the programs are trees) and scatters the
// adjoints of the loads into grad.  The arithmetic of the unary atoms is `unary_rules`
// (atom_math.h) — the same rules the tape kernels use (reference cvxpy/atoms/elementwise/*.py).
//
// fused_elements() is the single source: HipExec runs it with the slots in LDS
// (slot[k][lane]: conflict-free, register file of the interpreter) and atomic adds into grad;
// the host space of the test oracle runs the same function with a local array.
#pragma once
#include <algorithm>

#include "atom_math.h"
#include "tape.h"

namespace dnlp {

constexpr int kFusedMaxInstr = 32;
enum FusedOp : int { F_LOADV = 0, F_LOADC, F_UNARY, F_ADD, F_SUB, F_MUL, F_SCALE, F_ADDC, F_DIV };

// one program, passed by value to the kernel (scalar loads, uniform control flow)
struct FusedProg {
  int n = 0;
  i64 nelem = 0;
  int op[kFusedMaxInstr], a[kFusedMaxInstr], b[kFusedMaxInstr];
  i64 off[kFusedMaxInstr], stride[kFusedMaxInstr];
  double p[kFusedMaxInstr], p2[kFusedMaxInstr];
  // gradient window: when every variable load is x[off_j + i] (unit stride, offsets within a small
  // range) the adjoints of a tile of elements land in [tile + win_lo, tile + win_lo + tile_size +
  // win_extra): they are summed in LDS first and flushed with ONE global atomic per entry
  i64 win_lo = 0;
  int win_extra = -1;        // -1: no window (strided / far-apart loads): global atomics per load
};

// NE elements per call (instruction-major: one opcode decode serves NE independent element
// chains).  S: slot accessor  double& S(int k, int e);  G: scatter  void G(i64 index, double value).
// Elements i0 + e * estride, e < NE; elements >= P.nelem are skipped by the caller's `valid` mask.
template <int NE, class S, class G>
DNLP_HD inline double fused_elements(const FusedProg& P, i64 i0, i64 estride, const bool (&valid)[NE],
                                     const double* __restrict__ x, const double* __restrict__ consts,
                                     S slot, G scatter) {
  const int n = P.n;
  // opcode dispatch OUTSIDE the element loop: one decode serves NE independent element chains
#define DNLP_FZ_EACH for (int e = 0; e < NE; ++e) if (valid[e])
  for (int k = 0; k < n; ++k) {
    const int op = P.op[k], a = P.a[k], b = P.b[k];
    const i64 off = P.off[k], st = P.stride[k];
    const double p = P.p[k], p2 = P.p2[k];
    switch (op) {
      case F_LOADV:
#pragma unroll
        DNLP_FZ_EACH slot(k, e) = x[off + st * (i0 + e * estride)];
        break;
      case F_LOADC:
#pragma unroll
        DNLP_FZ_EACH slot(k, e) = consts[off + st * (i0 + e * estride)];
        break;
      case F_UNARY:
#pragma unroll
        DNLP_FZ_EACH { double v, g1, g2; unary_rules(b, slot(a, e), p, p2, v, g1, g2); slot(k, e) = v; }
        break;
      case F_ADD:
#pragma unroll
        DNLP_FZ_EACH slot(k, e) = slot(a, e) + slot(b, e);
        break;
      case F_SUB:
#pragma unroll
        DNLP_FZ_EACH slot(k, e) = slot(a, e) - slot(b, e);
        break;
      case F_MUL:
#pragma unroll
        DNLP_FZ_EACH slot(k, e) = slot(a, e) * slot(b, e);
        break;
      case F_SCALE:
#pragma unroll
        DNLP_FZ_EACH slot(k, e) = p * slot(a, e);
        break;
      case F_ADDC:
#pragma unroll
        DNLP_FZ_EACH slot(k, e) = slot(a, e) + p;
        break;
      default:
#pragma unroll
        DNLP_FZ_EACH slot(k, e) = slot(a, e) / slot(b, e);
        break;
    }
  }
  double fsum = 0.0;
#pragma unroll
  DNLP_FZ_EACH { fsum += slot(n - 1, e); slot(n - 1, e) = 1.0; }
  for (int k = n - 1; k >= 0; --k) {
    const int op = P.op[k], a = P.a[k], b = P.b[k];
    const i64 off = P.off[k], st = P.stride[k];
    const double p = P.p[k], p2 = P.p2[k];
    switch (op) {
      case F_LOADV:
#pragma unroll
        DNLP_FZ_EACH scatter(off + st * (i0 + e * estride), slot(k, e));
        break;
      case F_LOADC: break;
      case F_UNARY:
#pragma unroll
        DNLP_FZ_EACH { double v, g1, g2; unary_rules(b, slot(a, e), p, p2, v, g1, g2); slot(a, e) = slot(k, e) * g1; }
        break;
      case F_ADD:
#pragma unroll
        DNLP_FZ_EACH { const double g = slot(k, e); slot(a, e) = g; slot(b, e) = g; }
        break;
      case F_SUB:
#pragma unroll
        DNLP_FZ_EACH { const double g = slot(k, e); slot(a, e) = g; slot(b, e) = -g; }
        break;
      case F_MUL:
#pragma unroll
        DNLP_FZ_EACH { const double g = slot(k, e), va = slot(a, e), vb = slot(b, e); slot(a, e) = g * vb; slot(b, e) = g * va; }
        break;
      case F_SCALE:
#pragma unroll
        DNLP_FZ_EACH slot(a, e) = slot(k, e) * p;
        break;
      case F_ADDC:
#pragma unroll
        DNLP_FZ_EACH slot(a, e) = slot(k, e);
        break;
      default:
#pragma unroll
        DNLP_FZ_EACH { const double g = slot(k, e), va = slot(a, e), vb = slot(b, e); slot(a, e) = g / vb; slot(b, e) = -g * va / (vb * vb); }
        break;
    }
  }
#undef DNLP_FZ_EACH
  return fsum;
}

template <class E>
struct FusedObjective {
  E* ex = nullptr;
  std::vector<FusedProg> progs;
  double* consts = nullptr;   // exec space
  double c0 = 0.0;
  i64 nfree = 0;
  bool present = false;

  void load(E* e, const TapeBlob& tb) {
    ex = e;
    if (!tb.has("fz_dims")) return;
    const i64* d = tb.i64s("fz_dims");
    const i64 nprog = d[0], ninstr = d[1], nconst = d[2];
    nfree = d[3];
    const i64* ps = tb.i64s("fz_prog_start");
    const i64* pn = tb.i64s("fz_prog_nelem");
    for (i64 q = 0; q < nprog; ++q) {
      FusedProg P;
      P.n = static_cast<int>(ps[q + 1] - ps[q]);
      P.nelem = pn[q];
      if (P.n < 1 || P.n > kFusedMaxInstr || ps[q + 1] > ninstr) throw std::runtime_error("bad fused program");
      for (int k = 0; k < P.n; ++k) {
        const i64 s = ps[q] + k;
        P.op[k] = tb.i32s("fz_op")[s]; P.a[k] = tb.i32s("fz_a")[s]; P.b[k] = tb.i32s("fz_b")[s];
        P.off[k] = tb.i64s("fz_off")[s]; P.stride[k] = tb.i64s("fz_stride")[s];
        P.p[k] = tb.f64("fz_p")[s]; P.p2[k] = tb.f64("fz_p2")[s];
        // every value must have exactly one consumer that comes later (tree programs)
        if (P.op[k] >= F_UNARY && (P.a[k] < 0 || P.a[k] >= k)) throw std::runtime_error("bad fused operand");
        if ((P.op[k] == F_ADD || P.op[k] == F_SUB || P.op[k] == F_MUL || P.op[k] == F_DIV) && (P.b[k] < 0 || P.b[k] >= k))
          throw std::runtime_error("bad fused operand");
      }
      {
        i64 lo = 0, hi = 0;
        bool unit = true, any = false;
        for (int k = 0; k < P.n; ++k) {
          if (P.op[k] != F_LOADV) continue;
          if (P.stride[k] != 1) unit = false;
          if (!any) { lo = hi = P.off[k]; any = true; }
          lo = std::min(lo, P.off[k]); hi = std::max(hi, P.off[k]);
        }
        if (any && unit && hi - lo <= 64) { P.win_lo = lo; P.win_extra = static_cast<int>(hi - lo); }
      }
      progs.push_back(P);
    }
    consts = ex->template alloc<double>(static_cast<size_t>(nconst > 0 ? nconst : 1));
    if (nconst > 0) ex->h2d(consts, tb.f64("fz_consts"), sizeof(double) * static_cast<size_t>(nconst));
    c0 = tb.f64("fz_c0")[0];
    present = true;
  }

  // x, grad: exec space, nfree entries.  Returns f.
  double eval(const double* x, double* grad) {
    ex->zero(grad, sizeof(double) * static_cast<size_t>(nfree));
    double f = c0;
    for (const FusedProg& P : progs) f += ex->fused_eval(P, x, consts, grad);
    return f;
  }
};

}  // namespace dnlp
