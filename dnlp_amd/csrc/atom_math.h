// Per-element value / first / second derivative of the smooth elementwise atoms.
// One rule per reference atom: forward `numeric`, `_jacobian`
// diagonal and `_hess_vec` diagonal of cvxpy/atoms/elementwise/*.py (file:line per case).
// Shared by the HIP tape kernels (tape_kernels.hip) and the host reference backend.
#pragma once
#include "exec.h"

namespace dnlp {

enum Op : int {
  OP_EXP = 1, OP_LOG = 2, OP_ENTR = 3, OP_LOGISTIC = 4, OP_POWER = 5, OP_SIN = 6, OP_COS = 7,
  OP_TAN = 8, OP_SINH = 9, OP_TANH = 10, OP_ASINH = 11, OP_ATANH = 12, OP_XEXP = 13,
  OP_MUL = 20, OP_REL_ENTR = 21,
  OP_QUAD_FORM_DENSE = 30, OP_QUAD_FORM_SPARSE = 31, OP_QUAD_OVER_LIN = 32, OP_MATMUL = 33
};

DNLP_HD inline bool op_is_flat(int op) { return op < OP_QUAD_FORM_DENSE; }

// integer-exponent fast paths keep x^2 etc. exact and cheap (pow() is ~50 instructions)
DNLP_HD inline double pow_fast(double u, double p) {
  if (p == 2.0) return u * u;
  if (p == 1.0) return u;
  if (p == 0.0) return 1.0;
  if (p == 3.0) return u * u * u;
  if (p == 0.5) return sqrt(u);
  if (p == -0.5) return 1.0 / sqrt(u);
  if (p == -1.0) return 1.0 / u;
  if (p == 1.5) return u * sqrt(u);
  if (p == -1.5) return 1.0 / (u * sqrt(u));
  if (p == 4.0) { double t = u * u; return t * t; }
  return pow(u, p);
}

// value, d/du, d2/du2 of a unary atom.  p_der is the derivative exponent (reference uses
// p_rational there, power.py:410-419,433-450), p_fwd the forward one (power.py:187-188).
DNLP_HD inline void unary_rules(int op, double u, double p_der, double p_fwd, double& val,
                                double& d1, double& d2) {
  switch (op) {
    case OP_EXP: {                       // exp.py:34-35,112-121,102-107
      double e = exp(u); val = e; d1 = e; d2 = e; break; }
    case OP_LOG: {                       // log.py:33-36,118-127,108-113
      val = log(u); d1 = 1.0 / u; d2 = -1.0 / (u * u); break; }
    case OP_ENTR: {                      // entr.py:35-44,116-120,106-111
      double lg = log(u);
      val = (u > 0.0) ? -u * lg : (u == 0.0 ? 0.0 : -kInf);
      d1 = -lg - 1.0; d2 = -1.0 / u; break; }
    case OP_LOGISTIC: {                  // logistic.py:36-39,108-113,97-103
      double e = exp(u);
      val = (u > 0.0) ? u + log1p(exp(-u)) : log1p(e);
      d1 = e / (1.0 + e); d2 = e / ((1.0 + e) * (1.0 + e)); break; }
    case OP_POWER: {                     // power.py:187-188,433-450,408-422
      val = pow_fast(u, p_fwd);
      d1 = p_der * pow_fast(u, p_der - 1.0);
      d2 = p_der * (p_der - 1.0) * pow_fast(u, p_der - 2.0); break; }
    case OP_SIN: {                       // trig.py:33-36,99-103,90-94
      double sv = sin(u), cv = cos(u); val = sv; d1 = cv; d2 = -sv; break; }
    case OP_COS: {                       // trig.py:113-116,179-183,170-174
      double sv = sin(u), cv = cos(u); val = cv; d1 = -sv; d2 = -cv; break; }
    case OP_TAN: {                       // trig.py:194-197,261-265,251-256
      double t = tan(u), c = cos(u); val = t; d1 = 1.0 / (c * c); d2 = 2.0 * t / (c * c); break; }
    case OP_SINH: {                      // hyperbolic.py:33-36,94-98,85-89
      double sh = sinh(u), ch = cosh(u); val = sh; d1 = ch; d2 = sh; break; }
    case OP_TANH: {                      // hyperbolic.py:108-111,169-173,160-164
      double th = tanh(u), ch = cosh(u); val = th; d1 = 1.0 / (ch * ch);
      d2 = -2.0 * th / (ch * ch); break; }
    case OP_ASINH: {                     // hyperbolic.py:183-186,228-232,219-223
      double q = 1.0 + u * u; val = asinh(u); d1 = 1.0 / sqrt(q); d2 = -u / (q * sqrt(q)); break; }
    case OP_ATANH: {                     // hyperbolic.py:242-245,287-291,278-282
      double q = 1.0 - u * u; val = atanh(u); d1 = 1.0 / q; d2 = 2.0 * u / (q * q); break; }
    case OP_XEXP: {                      // xexp.py:35-36,108-112,117-121
      double e = exp(u); val = u * e; d1 = e * (1.0 + u); d2 = e * (2.0 + u); break; }
    default: val = d1 = d2 = 0.0;
  }
}

}  // namespace dnlp
