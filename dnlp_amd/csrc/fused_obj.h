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
#include <stdexcept>
#include <utility>
#include <vector>

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

// ---- slot program ---------------------------------------------------------------------------
// The tree program is differentiated ON THE HOST, once: forward ops, then the reverse sweep written
// out as ordinary three-address ops (adjoint of a unary = adjoint * stored derivative, of a product =
// adjoint * the other factor, ...).  A linear-scan register allocation over that straight-line code
// maps every value / derivative / adjoint to a physical slot; a slot is reused as soon as its last
// reader has run (in place when the reader is the writer).  Rosenbrock: 11 tree instructions ->
// 21 ops over 4 slots instead of 11 slots, which is what lets the device kernel keep FOUR elements
// per lane in LDS and amortise one opcode decode over four independent element chains.
constexpr int kFusedMaxOps = 112;     // 112 x 32 B of records + header stay below the 4 KB kernel-argument limit
enum FusedSlotOp : unsigned char {
  S_LOADV = 0, S_LOADC, S_UNARY, S_ADD, S_SUB, S_MUL, S_DIV, S_SCALE, S_ADDC, S_SET, S_ACCF, S_SCATTER, S_AXPB
};

// One op = one 32-byte record, so the interpreter's decode is a single scalar load:
// code = op | d << 8 | s1 << 16 | s2 << 24 (S_UNARY: s2 = slot of the derivative); u = unary opcode;
// p = scale / shift / weight / exponent; q = second parameter (S_UNARY exponent of the derivative,
// S_AXPB shift) or, for S_LOADV / S_LOADC / S_SCATTER, the index offset; stride = index stride.
struct alignas(32) FusedOpRec {
  unsigned code, u;
  double p;
  union { double q; i64 off; };
  union { i64 stride; double w; };     // S_UNARY: w != 0 -> the value goes straight into f with weight w
};

struct FusedSlotProg {
  int nops = 0, nslots = 0;
  i64 nelem = 0;
  i64 win_lo = 0;            // gradient window (see FusedProg)
  int win_extra = -1;
  FusedOpRec rec[kFusedMaxOps];
};

// Host: tree program -> slot program.  Throws when the expansion exceeds the fixed capacities.
inline FusedSlotProg fused_compile(const FusedProg& T) {
  const int n = T.n;
  struct VOp { int op, d, d2, s1, s2, u; double p, p2; };
  std::vector<VOp> ops;
  std::vector<double> accw;          // per final op: weight of the fused f accumulation (S_UNARY), else 0
  std::vector<i64> roff, rstride;
  auto ref = [&](i64 off, i64 st) {
    for (size_t k = 0; k < roff.size(); ++k) if (roff[k] == off && rstride[k] == st) return static_cast<int>(k);
    roff.push_back(off); rstride.push_back(st);
    return static_cast<int>(roff.size() - 1);
  };
  // virtual registers: value k, derivative n + k, adjoint 2n + k, temporary 3n + k
  auto V = [&](int k) { return k; };
  auto G = [&](int k) { return n + k; };
  std::vector<int> adj(static_cast<size_t>(n), -1);       // adjoint register of instruction k (-1: a constant)
  // forward peephole: scale followed by shift (or shift followed by scale) is one op  d = p s + q
  std::vector<int> cons(static_cast<size_t>(n), -1);
  for (int k = 0; k < n; ++k) {
    if (T.op[k] >= F_UNARY) cons[static_cast<size_t>(T.a[k])] = k;
    if (T.op[k] == F_ADD || T.op[k] == F_SUB || T.op[k] == F_MUL || T.op[k] == F_DIV) cons[static_cast<size_t>(T.b[k])] = k;
  }
  std::vector<char> deferred(static_cast<size_t>(n), 0);
  // the root's add / sub / scale chain is never evaluated: its leaves are summed into f with
  // weights (S_ACCF carries the weight), and their adjoints are the weights
  std::vector<char> peeled(static_cast<size_t>(n), 0);
  std::vector<std::pair<int, double>> roots;
  {
    std::vector<std::pair<int, double>> st{{n - 1, 1.0}};
    while (!st.empty()) {
      const auto [k, w] = st.back();
      st.pop_back();
      const int o = T.op[k];
      if (o == F_ADD || o == F_SUB) {
        peeled[static_cast<size_t>(k)] = 1;
        st.push_back({T.b[k], o == F_SUB ? -w : w});
        st.push_back({T.a[k], w});
      } else if (o == F_SCALE) {
        peeled[static_cast<size_t>(k)] = 1;
        st.push_back({T.a[k], w * T.p[k]});
      } else {
        roots.push_back({k, w});
      }
    }
  }
  // a root leaf goes into f as soon as it exists: its slot is free for the rest of the program
  auto accf = [&](int k) {
    for (const auto& rw : roots) if (rw.first == k) ops.push_back({S_ACCF, -1, -1, k, -1, 0, rw.second, 0});
  };
  for (int k = 0; k < n; ++k) {
    const int o = T.op[k], a = T.a[k], b = T.b[k];
    const int ck = cons[static_cast<size_t>(k)];
    if (peeled[static_cast<size_t>(k)]) continue;
    if ((o == F_SCALE || o == F_ADDC) && ck >= 0 && T.op[ck] == (o == F_SCALE ? F_ADDC : F_SCALE) &&
        !peeled[static_cast<size_t>(ck)] && !deferred[static_cast<size_t>(a)]) {
      deferred[static_cast<size_t>(k)] = 1;       // emitted together with its consumer
      continue;
    }
    if ((o == F_SCALE || o == F_ADDC) && deferred[static_cast<size_t>(a)]) {
      const int src = V(T.a[a]);
      if (o == F_ADDC) ops.push_back({S_AXPB, V(k), -1, src, -1, 0, T.p[a], T.p[k]});            // p_a x + q_k
      else ops.push_back({S_AXPB, V(k), -1, src, -1, 0, T.p[k], T.p[k] * T.p[a]});                // p_k (x + q_a)
      accf(k);
      continue;
    }
    switch (o) {
      case F_LOADV: ops.push_back({S_LOADV, V(k), -1, -1, -1, ref(T.off[k], T.stride[k]), 0, 0}); break;
      case F_LOADC: ops.push_back({S_LOADC, V(k), -1, -1, -1, ref(T.off[k], T.stride[k]), 0, 0}); break;
      case F_UNARY: ops.push_back({S_UNARY, V(k), G(k), V(a), -1, b, T.p[k], T.p2[k]}); break;
      case F_ADD: ops.push_back({S_ADD, V(k), -1, V(a), V(b), 0, 0, 0}); break;
      case F_SUB: ops.push_back({S_SUB, V(k), -1, V(a), V(b), 0, 0, 0}); break;
      case F_MUL: ops.push_back({S_MUL, V(k), -1, V(a), V(b), 0, 0, 0}); break;
      case F_DIV: ops.push_back({S_DIV, V(k), -1, V(a), V(b), 0, 0, 0}); break;
      case F_SCALE: ops.push_back({S_SCALE, V(k), -1, V(a), -1, 0, T.p[k], 0}); break;
      case F_ADDC: ops.push_back({S_ADDC, V(k), -1, V(a), -1, 0, T.p[k], 0}); break;
      default: throw std::runtime_error("bad fused opcode");
    }
    accf(k);
  }
  // Reverse sweep.  The adjoint of instruction k is carried as  fac[k] * (register adj[k]), with
  // adj[k] = -1 meaning the pure constant fac[k]: constant factors (the root's 1, scales, signs of
  // subtractions) never cost an op — they travel to the scatter, which applies them (S_SCATTER
  // carries the weight); a product rule whose incoming adjoint is a pure constant is just an alias
  // to the stored derivative / other factor.  Rosenbrock: 11 tree instructions -> 14 ops, 5 slots.
  std::vector<double> fac(static_cast<size_t>(n), 0.0);
  std::vector<char> seen(static_cast<size_t>(n), 0);
  auto give = [&](int i, int reg, double c) { adj[static_cast<size_t>(i)] = reg; fac[static_cast<size_t>(i)] = c; seen[static_cast<size_t>(i)] = 1; };
  give(n - 1, -1, 1.0);
  for (int k = n - 1; k >= 0; --k) {
    const int o = T.op[k], a = T.a[k], b = T.b[k];
    if (!seen[static_cast<size_t>(k)]) throw std::runtime_error("fused program is not a tree");
    int ak = adj[static_cast<size_t>(k)];
    double c = fac[static_cast<size_t>(k)];
    // adjoint register holding exactly the adjoint (factor folded in), for the rules that need it
    auto materialise = [&]() {
      if (ak < 0) { ak = 2 * n + k; ops.push_back({S_SET, ak, -1, -1, -1, 0, c, 0}); }
      else if (c != 1.0) { ops.push_back({S_SCALE, 2 * n + k, -1, ak, -1, 0, c, 0}); ak = 2 * n + k; }
      c = 1.0;
    };
    // child adjoint = c * ak * (register other)
    auto times = [&](int child, int other) {
      if (ak < 0) { give(child, other, c); return; }
      ops.push_back({S_MUL, 2 * n + child, -1, ak, other, 0, 0, 0});
      give(child, 2 * n + child, c);
    };
    switch (o) {
      case F_LOADV:
        if (ak < 0) materialise();
        ops.push_back({S_SCATTER, -1, -1, ak, -1, ref(T.off[k], T.stride[k]), c, 0});
        break;
      case F_LOADC: break;
      case F_UNARY: times(a, G(k)); break;
      case F_ADD: give(a, ak, c); give(b, ak, c); break;
      case F_SUB: give(a, ak, c); give(b, ak, -c); break;
      case F_MUL: times(a, V(b)); times(b, V(a)); break;
      case F_SCALE: give(a, ak, c * T.p[k]); break;
      case F_ADDC: give(a, ak, c); break;
      default: {   // k = a / b:  adj_a = adj_k / b ;  adj_b = -adj_a * k
        materialise();
        const int Aa = 2 * n + a, Tm = 3 * n + k;
        ops.push_back({S_DIV, Aa, -1, ak, V(b), 0, 0, 0}); give(a, Aa, 1.0);
        ops.push_back({S_MUL, Tm, -1, Aa, V(k), 0, 0, 0}); give(b, Tm, -1.0); break; }
    }
  }
  // a unary root leaf is summed into f by the op itself (no separate S_ACCF decode, no value slot)
  {
    std::vector<VOp> fusedops;
    for (size_t i = 0; i < ops.size(); ++i) {
      if (ops[i].op == S_UNARY && i + 1 < ops.size() && ops[i + 1].op == S_ACCF && ops[i + 1].s1 == ops[i].d && ops[i + 1].p != 0.0) {
        fusedops.push_back(ops[i]);
        fusedops.back().d = -1;       // the value is not stored: it goes into f with the weight kept in accw
        accw.push_back(ops[i + 1].p);
        ++i;
      } else {
        fusedops.push_back(ops[i]);
        accw.push_back(0.0);
      }
    }
    ops.swap(fusedops);
  }
  if (ops.size() > static_cast<size_t>(kFusedMaxOps))
    throw std::runtime_error("fused program too long for the slot form");
  // liveness: last op that reads each virtual register
  const int nv = 4 * n;
  std::vector<int> last(static_cast<size_t>(nv), -1), slot(static_cast<size_t>(nv), -1);
  for (size_t i = 0; i < ops.size(); ++i) {
    if (ops[i].s1 >= 0) last[static_cast<size_t>(ops[i].s1)] = static_cast<int>(i);
    if (ops[i].s2 >= 0) last[static_cast<size_t>(ops[i].s2)] = static_cast<int>(i);
  }
  std::vector<int> free_slots;
  int nslots = 0;
  auto take = [&]() {
    if (!free_slots.empty()) {
      auto it = std::min_element(free_slots.begin(), free_slots.end());
      const int sl = *it; free_slots.erase(it); return sl;
    }
    return nslots++;
  };
  FusedSlotProg S;
  for (size_t i = 0; i < ops.size(); ++i) {
    const VOp& o = ops[i];
    const int ps1 = o.s1 >= 0 ? slot[static_cast<size_t>(o.s1)] : 0, ps2 = o.s2 >= 0 ? slot[static_cast<size_t>(o.s2)] : 0;
    // sources read for the last time here give their slots back BEFORE the destinations are placed
    // (every op reads all its sources into registers before it writes)
    if (o.s1 >= 0 && last[static_cast<size_t>(o.s1)] == static_cast<int>(i)) free_slots.push_back(ps1);
    if (o.s2 >= 0 && o.s2 != o.s1 && last[static_cast<size_t>(o.s2)] == static_cast<int>(i)) free_slots.push_back(ps2);
    int pd = 0, pd2 = 0;
    if (o.d >= 0) { pd = take(); slot[static_cast<size_t>(o.d)] = pd; }
    if (o.d2 >= 0) { pd2 = take(); slot[static_cast<size_t>(o.d2)] = pd2; }
    // a destination nobody reads (adjoint of a constant branch) is released at once
    if (o.d >= 0 && last[static_cast<size_t>(o.d)] < 0) free_slots.push_back(pd);
    if (o.d2 >= 0 && last[static_cast<size_t>(o.d2)] < 0) free_slots.push_back(pd2);
    FusedOpRec& R = S.rec[i];
    const unsigned ps2f = static_cast<unsigned>(o.op == S_UNARY ? pd2 : ps2);
    R.code = static_cast<unsigned>(o.op) | static_cast<unsigned>(pd) << 8 | static_cast<unsigned>(ps1) << 16 | ps2f << 24;
    R.u = static_cast<unsigned>(o.op == S_UNARY ? o.u : 0);
    R.p = o.p;
    R.stride = 0;
    if (o.op == S_UNARY) R.w = accw[i];
    if (o.op == S_LOADV || o.op == S_LOADC || o.op == S_SCATTER) {
      R.off = roff[static_cast<size_t>(o.u)];
      R.stride = rstride[static_cast<size_t>(o.u)];
    } else {
      R.q = o.p2;
    }
  }
  if (nslots > 255) throw std::runtime_error("fused program needs too many slots");
  S.nops = static_cast<int>(ops.size());
  S.nslots = nslots;
  S.nelem = T.nelem;
  S.win_lo = T.win_lo;
  S.win_extra = T.win_extra;
  return S;
}

// NE elements per call (op-major: one opcode decode serves NE independent element chains).
// S: slot accessor  double& S(int slot, int e);  G: scatter  void G(i64 index, double value).
// Elements i0 + e * estride, e < NE; elements >= P.nelem are skipped through `valid`.
// ALL: every element of the tile is valid (all tiles but the last): no per-element predicate, so the
// NE element chains of an op stay in ONE basic block and their LDS / global accesses overlap.
template <int NE, bool ALL = false, class S, class G>
DNLP_HD inline double fused_elements(const FusedSlotProg& P, i64 i0, i64 estride, const bool (&valid)[NE],
                                     const double* __restrict__ x, const double* __restrict__ consts,
                                     S slot, G scatter) {
  double fsum = 0.0;
  const int nops = P.nops;
#define DNLP_FZ_EACH for (int e = 0; e < NE; ++e) if (ALL || valid[e])
  for (int i = 0; i < nops; ++i) {
    const FusedOpRec R = P.rec[i];      // (fetching record i + 1 ahead was measured: 1.02 -> 1.09 ms)
    const int op = static_cast<int>(R.code & 0xffu), d = static_cast<int>((R.code >> 8) & 0xffu),
              s1 = static_cast<int>((R.code >> 16) & 0xffu), s2 = static_cast<int>(R.code >> 24);
    const double p = R.p;
    switch (op) {
      case S_LOADV: {
        const i64 off = R.off, st = R.stride;
#pragma unroll
        DNLP_FZ_EACH slot(d, e) = x[off + st * (i0 + e * estride)];
        break; }
      case S_LOADC: {
        const i64 off = R.off, st = R.stride;
#pragma unroll
        DNLP_FZ_EACH slot(d, e) = consts[off + st * (i0 + e * estride)];
        break; }
      case S_UNARY: {
        const int u = static_cast<int>(R.u);
        const double p2 = R.q, w = R.w;
        const bool sq = u == OP_POWER && p == 2.0 && p2 == 2.0;      // squares: no libm, no compare chain
        if (w != 0.0) {                                              // root leaf: value straight into f
          if (sq) {
#pragma unroll
            DNLP_FZ_EACH { const double t = slot(s1, e); fsum += w * (t * t); slot(s2, e) = 2.0 * t; }
          } else {
#pragma unroll
            DNLP_FZ_EACH { double v, g1, g2; unary_rules(u, slot(s1, e), p, p2, v, g1, g2); fsum += w * v; slot(s2, e) = g1; }
          }
        } else if (sq) {
#pragma unroll
          DNLP_FZ_EACH { const double t = slot(s1, e); slot(d, e) = t * t; slot(s2, e) = 2.0 * t; }
        } else {
#pragma unroll
          DNLP_FZ_EACH { double v, g1, g2; unary_rules(u, slot(s1, e), p, p2, v, g1, g2); slot(d, e) = v; slot(s2, e) = g1; }
        }
        break; }
      case S_ADD:
#pragma unroll
        DNLP_FZ_EACH { const double a = slot(s1, e), b = slot(s2, e); slot(d, e) = a + b; }
        break;
      case S_SUB:
#pragma unroll
        DNLP_FZ_EACH { const double a = slot(s1, e), b = slot(s2, e); slot(d, e) = a - b; }
        break;
      case S_MUL:
#pragma unroll
        DNLP_FZ_EACH { const double a = slot(s1, e), b = slot(s2, e); slot(d, e) = a * b; }
        break;
      case S_DIV:
#pragma unroll
        DNLP_FZ_EACH { const double a = slot(s1, e), b = slot(s2, e); slot(d, e) = a / b; }
        break;
      case S_SCALE:
#pragma unroll
        DNLP_FZ_EACH slot(d, e) = p * slot(s1, e);
        break;
      case S_ADDC:
#pragma unroll
        DNLP_FZ_EACH slot(d, e) = slot(s1, e) + p;
        break;
      case S_SET:
#pragma unroll
        DNLP_FZ_EACH slot(d, e) = p;
        break;
      case S_AXPB: {
        const double q = R.q;
#pragma unroll
        DNLP_FZ_EACH slot(d, e) = p * slot(s1, e) + q;
        break; }
      case S_ACCF:
#pragma unroll
        DNLP_FZ_EACH fsum += p * slot(s1, e);
        break;
      default: {   // S_SCATTER
        const i64 off = R.off, st = R.stride;
#pragma unroll
        DNLP_FZ_EACH scatter(off + st * (i0 + e * estride), p * slot(s1, e));
        break; }
    }
  }
#undef DNLP_FZ_EACH
  return fsum;
}

// fz_* arrays of the tape blob -> slot programs (no execution space involved)
inline bool fused_parse_programs(const TapeBlob& tb, std::vector<FusedSlotProg>& progs, i64& nconst, double& c0, i64& nfree) {
  if (!tb.has("fz_dims")) return false;
  const i64* d = tb.i64s("fz_dims");
  const i64 nprog = d[0], ninstr = d[1];
  nconst = d[2];
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
    progs.push_back(fused_compile(P));
  }
  c0 = tb.f64("fz_c0")[0];
  return true;
}

template <class E>
struct FusedObjective {
  E* ex = nullptr;
  std::vector<FusedSlotProg> progs;
  double* consts = nullptr;   // exec space
  double c0 = 0.0;
  i64 nfree = 0;
  bool present = false;

  void load(E* e, const TapeBlob& tb) {
    ex = e;
    i64 nconst = 0;
    if (!fused_parse_programs(tb, progs, nconst, c0, nfree)) return;
    consts = ex->template alloc<double>(static_cast<size_t>(nconst > 0 ? nconst : 1));
    if (nconst > 0) ex->h2d(consts, tb.f64("fz_consts"), sizeof(double) * static_cast<size_t>(nconst));
    present = true;
  }

  // x, grad: exec space, nfree entries.  Returns f.
  double eval(const double* x, double* grad) {
    // specialised kernel generated from the programs (fused_codegen.h): grad is written, not accumulated
    double fg = 0.0;
    if (ex->fused_generated_eval(progs, x, consts, grad, nfree, fg)) return c0 + fg;
    ex->zero(grad, sizeof(double) * static_cast<size_t>(nfree));
    double f = c0;
    for (const FusedSlotProg& P : progs) f += ex->fused_eval(P, x, consts, grad);
    return f;
  }
};

}  // namespace dnlp
