/* exa_oracle.c — TEST ORACLE.  CPU restatement of the reference's evaluation algorithm for
 * obj / cons_nln! / grad! / jac_coord! / hess_coord! / jac_structure! / hess_structure!.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; nothing under examodels.jl_amd/ links, imports or calls it.
 *
 * Pinning status: the reference is pure Julia and cannot run in the build container (no julia, no
 * network), so the oracle is pinned against (a) the reference's own known-answer tests
 * (test/NLPTest/conaug_test.jl, feature_test.jl, test/ConcreteModeTest.jl, docs/src/develop.md KKT point)
 * and (b) golden vectors produced by an independent symbolic differentiator (sympy) over the reference's
 * ADTest expression list (test/ADTest/ADTest.jl:6-121) — see tests/golden/.  Bit-level agreement with
 * Julia's libm is unpinned; the parity bar is 1e-10 relative.
 *
 * It is a tree-walking interpreter that performs, per data point, exactly the reference's
 *   forward sweep  (src/register.jl:65-68,174-266; leaves src/graph.jl:305-323,397-400,491-494)
 *   reverse sweeps (src/gradient.jl:11-26,64-86; src/jacobian.jl:16-40,69-83;
 *                   src/hessian.jl:16-268,337-380,382-517,580-642)
 * with zero-filled outputs and one `+=` per contribution, in the reference's order, using the derivative
 * formulas of src/functionlist.jl:6-81 verbatim.  Slot maps follow src/simdfunction.jl:78-100
 * (first-appearance order under structural identity), offsets src/nlp.jl:1474-1482,1597-1611,1730-1738,
 * 1980-2015.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/exahip_ir.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------------ */
/* typed scalars: Julia keeps Int and Float64 apart until promotion                                  */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { int is_int; int64_t i; double f; } val_t;
static inline val_t VI(int64_t i) { val_t v = {1, i, (double)i}; return v; }
static inline val_t VF(double f) { val_t v = {0, 0, f}; return v; }
static inline double asf(val_t v) { return v.is_int ? (double)v.i : v.f; }

/* x^n, integer n — Base.^(::Float64, ::Integer): n==0 -> 1, n==3 -> x*x*x, n<0 -> via inv, else by squaring */
static double pow_int(double x, int64_t n) {
    if (n == 0) return 1.0;
    if (n == 1) return x;
    if (n == 2) return x * x;
    if (n == 3) return x * x * x;
    if (n < 0) { x = 1.0 / x; n = -n; if (n == 2) return x * x; }
    double y = 1.0;
    while (n > 1) { if (n & 1) y *= x; x *= x; n >>= 1; }
    return x * y;
}
static int64_t ipow_int(int64_t b, int64_t n) { int64_t y = 1; while (n > 0) { if (n & 1) y *= b; b *= b; n >>= 1; } return y; }
static double powv(double x, val_t e) { return e.is_int ? pow_int(x, e.i) : pow(x, e.f); }

/* ------------------------------------------------------------------------------------------------ */
/* derivative tables — src/functionlist.jl:6-81, formula shapes kept                                  */
/* ------------------------------------------------------------------------------------------------ */
static const double CLOG2 = 0.69314718055994530942, CLOG10 = 2.30258509299404568402;
static const double CD2R = M_PI / 180.0, CR2D = 180.0 / M_PI;
static inline double sq(double x) { return x * x; }
static inline double cb(double x) { return x * x * x; }
static inline double sec_(double x) { return 1.0 / cos(x); }
static inline double csc_(double x) { return 1.0 / sin(x); }
static inline double cot_(double x) { return 1.0 / tan(x); }
static inline double sech_(double x) { return 1.0 / cosh(x); }
static inline double csch_(double x) { return 1.0 / sinh(x); }
static inline double coth_(double x) { return 1.0 / tanh(x); }
static inline double d2r(double x) { return x * CD2R; }
/* Base.sind / Base.cosd / Base.tand (Julia base/special/trig.jl — outside /root/reference, restated from its published
 * algorithm): the argument is reduced with an EXACT rem(x, 360) and the quadrant is selected BEFORE the conversion to
 * radians, so the functions are exact at the multiples of 90: sind(180) = 0, cosd(90) = 0, tand(90) = Inf,
 * cscd(180) = Inf.  (Julia evaluates its sin/cos kernels at a double-double deg2rad; plain deg2rad here — a 1-ulp
 * matter, inside the 1e-10 bar.  Julia throws DomainError at +-Inf; a kernel cannot: NaN.) */
static inline double sind_(double x) {
    if (isinf(x)) return NAN;
    if (isnan(x)) return x;
    const double rx = copysign(fmod(x, 360.0), x), arx = fabs(rx);
    if (rx == 0.0) return rx;
    if (arx < 45.0) return sin(d2r(rx));
    if (arx <= 135.0) return copysign(cos(d2r(90.0 - arx)), rx);
    if (arx == 180.0) return copysign(0.0, rx);
    if (arx < 225.0) return sin(d2r((180.0 - arx) * (rx > 0 ? 1.0 : -1.0)));
    if (arx <= 315.0) return -copysign(cos(d2r(270.0 - arx)), rx);
    return sin(d2r(rx - copysign(360.0, rx)));
}
static inline double cosd_(double x) {
    if (isinf(x)) return NAN;
    if (isnan(x)) return x;
    const double rx = fabs(fmod(x, 360.0));
    if (rx <= 45.0) return cos(d2r(rx));
    if (rx < 135.0) return sin(d2r(90.0 - rx));
    if (rx <= 225.0) return -cos(d2r(180.0 - rx));
    if (rx < 315.0) return sin(d2r(rx - 270.0));
    return cos(d2r(360.0 - rx));
}
static inline double tand_(double x) { return sind_(x) / cosd_(x); }      /* Base: tand(x) = sind(x) / cosd(x) */
static inline double cscd_(double x) { return 1.0 / sind_(x); }
static inline double secd_(double x) { return 1.0 / cosd_(x); }
static inline double cotd_(double x) { return 1.0 / tand_(x); }
/* Base.sinpi / Base.cospi semantics: exact at integers and half-integers (the argument is reduced BEFORE it is multiplied
 * by pi): n = nearest integer, r = x - n is exact and |r| <= 1/2; sin(pi x) = (-1)^n sin(pi r). */
static inline double sinpi_(double x) {
    if (!isfinite(x)) return NAN;
    const double n = nearbyint(x), r = x - n, a = fabs(r);
    double s = a <= 0.25 ? sin(M_PI * r) : copysign(cos(M_PI * (0.5 - a)), r);
    if (fmod(n, 2.0) != 0.0) s = -s;
    return s;
}
static inline double cospi_(double x) {
    if (!isfinite(x)) return NAN;
    const double n = nearbyint(x), r = x - n, a = fabs(r);
    double c = a <= 0.25 ? cos(M_PI * r) : sin(M_PI * (0.5 - a));
    if (fmod(n, 2.0) != 0.0) c = -c;
    return c;
}
static inline double sinc_(double x) { return x == 0.0 ? 1.0 : sinpi_(x) / (M_PI * x); }
static inline double sign_(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : x); }

#include "exa_special.h"

static double un_f(int fn, double x) {
    if (fn >= EXA_U_FIRST_SPECIAL) return sf_f(fn, x);
    switch (fn) {
    case EXA_U_PLUS: return x;            case EXA_U_MINUS: return -x;
    case EXA_U_INV: return 1.0 / x;       case EXA_U_SQRT: return sqrt(x);
    case EXA_U_CBRT: return cbrt(x);      case EXA_U_ABS: return fabs(x);
    case EXA_U_ABS2: return x * x;        case EXA_U_SIGN: return sign_(x);
    case EXA_U_EXP: return exp(x);        case EXA_U_EXP2: return exp2(x);
    case EXA_U_EXP10: return exp10(x);    case EXA_U_EXPM1: return expm1(x);
    case EXA_U_LOG: return log(x);        case EXA_U_LOG2: return log2(x);
    case EXA_U_LOG1P: return log1p(x);    case EXA_U_LOG10: return log10(x);
    case EXA_U_SIN: return sin(x);        case EXA_U_COS: return cos(x);
    case EXA_U_TAN: return tan(x);        case EXA_U_ASIN: return asin(x);
    case EXA_U_ACOS: return acos(x);      case EXA_U_ATAN: return atan(x);
    case EXA_U_ACOT: return atan(1.0 / x);
    case EXA_U_CSC: return csc_(x);       case EXA_U_SEC: return sec_(x);
    case EXA_U_COT: return cot_(x);       case EXA_U_SINH: return sinh(x);
    case EXA_U_COSH: return cosh(x);      case EXA_U_TANH: return tanh(x);
    case EXA_U_ASINH: return asinh(x);    case EXA_U_ACOSH: return acosh(x);
    case EXA_U_CSCH: return csch_(x);     case EXA_U_SECH: return sech_(x);
    case EXA_U_COTH: return coth_(x);     case EXA_U_SIND: return sind_(x);
    case EXA_U_COSD: return cosd_(x);     case EXA_U_TAND: return tand_(x);
    case EXA_U_CSCD: return cscd_(x);     case EXA_U_SECD: return secd_(x);
    case EXA_U_COTD: return cotd_(x);     case EXA_U_ATAND: return CR2D * atan(x);
    case EXA_U_ACOTD: return CR2D * atan(1.0 / x);
    case EXA_U_SINPI: return sinpi_(x);   case EXA_U_COSPI: return cospi_(x);
    case EXA_U_SINC: return sinc_(x);     case EXA_U_DEG2RAD: return d2r(x);
    case EXA_U_RAD2DEG: return x * CR2D;  case EXA_U_SIGNBIT: return signbit(x) ? 1.0 : 0.0;
    case EXA_U_FLOOR: return floor(x);    case EXA_U_CEIL: return ceil(x);
    case EXA_U_ATANH: return atanh(x);    case EXA_U_ACOTH: return atanh(1.0 / x);
    }
    return NAN;
}

static double un_df(int fn, double x) {
    if (fn >= EXA_U_FIRST_SPECIAL) return sf_df(fn, x);
    switch (fn) {
    case EXA_U_PLUS: return 1.0;
    case EXA_U_MINUS: return -1.0;
    case EXA_U_INV: return -1 / (sq(x));
    case EXA_U_SQRT: return 1 / (2 * sqrt(x));
    case EXA_U_CBRT: return 1 / (3 * (sq(cbrt(x))));
    case EXA_U_ABS: return signbit(x) ? -1.0 : 1.0;
    case EXA_U_ABS2: return 2 * x;
    case EXA_U_SIGN: return 0.0;
    case EXA_U_EXP: return exp(x);
    case EXA_U_EXP2: return CLOG2 * exp2(x);
    case EXA_U_EXP10: return CLOG10 * exp10(x);
    case EXA_U_EXPM1: return exp(x);
    case EXA_U_LOG: return 1 / x;
    case EXA_U_LOG2: return 1 / (CLOG2 * x);
    case EXA_U_LOG1P: return 1 / (1 + x);
    case EXA_U_LOG10: return 1 / (CLOG10 * x);
    case EXA_U_SIN: return cos(x);
    case EXA_U_COS: return -sin(x);
    case EXA_U_TAN: return sq(sec_(x));
    case EXA_U_ASIN: return 1 / sqrt(1 - (sq(x)));
    case EXA_U_ACOS: return -1 / sqrt(1 - (sq(x)));
    case EXA_U_ATAN: return 1 / (1 + sq(x));
    case EXA_U_ACOT: return -1 / (1 + sq(x));
    case EXA_U_CSC: return -cot_(x) * csc_(x);
    case EXA_U_SEC: return sec_(x) * tan(x);
    case EXA_U_COT: return -1 - (sq(cot_(x)));
    case EXA_U_SINH: return cosh(x);
    case EXA_U_COSH: return sinh(x);
    case EXA_U_TANH: return 1 - (sq(tanh(x)));
    case EXA_U_ASINH: return 1 / sqrt(1 + sq(x));
    case EXA_U_ACOSH: return 1 / sqrt(-1 + sq(x));
    case EXA_U_CSCH: return -csch_(x) * coth_(x);
    case EXA_U_SECH: return -tanh(x) * sech_(x);
    case EXA_U_COTH: return -(sq(csch_(x)));
    case EXA_U_SIND: return d2r(cosd_(x));
    case EXA_U_COSD: return -d2r(sind_(x));
    case EXA_U_TAND: return d2r(1 + sq(tand_(x)));
    case EXA_U_CSCD: return -d2r(cscd_(x) * cotd_(x));
    case EXA_U_SECD: return d2r(tand_(x) * secd_(x));
    case EXA_U_COTD: return -d2r(1 + sq(cotd_(x)));
    case EXA_U_ATAND: return 1 / d2r(1 + sq(x));
    case EXA_U_ACOTD: return -1 / d2r(1 + sq(x));
    case EXA_U_SINPI: return M_PI * cospi_(x);
    case EXA_U_COSPI: return -M_PI * sinpi_(x);
    case EXA_U_SINC: return (-sinpi_(x) + M_PI * x * cospi_(x)) / (M_PI * (sq(x)));
    case EXA_U_DEG2RAD: return CD2R;
    case EXA_U_RAD2DEG: return CR2D;
    case EXA_U_SIGNBIT: case EXA_U_FLOOR: case EXA_U_CEIL: return 0.0;
    case EXA_U_ATANH: return fabs(x) > 1.0 ? NAN : 1.0 / (1 - sq(x));
    case EXA_U_ACOTH: return fabs(x) < 1.0 ? NAN : 1.0 / (1 - sq(x));
    }
    return NAN;
}

static double un_ddf(int fn, double x) {
    if (fn >= EXA_U_FIRST_SPECIAL) return sf_ddf(fn, x);
    switch (fn) {
    case EXA_U_PLUS: case EXA_U_MINUS: return 0.0;
    case EXA_U_INV: return 2 / (cb(x));
    case EXA_U_SQRT: return -1 / (4 * (cb(sqrt(x))));
    case EXA_U_CBRT: return -2 / (9 * (pow_int(cbrt(x), 5)));
    case EXA_U_ABS: return 0.0;
    case EXA_U_ABS2: return 2.0;
    case EXA_U_SIGN: return 0.0;
    case EXA_U_EXP: return exp(x);
    case EXA_U_EXP2: return sq(CLOG2) * exp2(x);
    case EXA_U_EXP10: return sq(CLOG10) * exp10(x);
    case EXA_U_EXPM1: return exp(x);
    case EXA_U_LOG: return -1 / (sq(x));
    case EXA_U_LOG2: return -CLOG2 / (sq(CLOG2) * (sq(x)));
    case EXA_U_LOG1P: return -1 / (sq(1 + x));
    case EXA_U_LOG10: return -CLOG10 / (sq(CLOG10) * (sq(x)));
    case EXA_U_SIN: return -sin(x);
    case EXA_U_COS: return -cos(x);
    case EXA_U_TAN: return 2 * (sq(sec_(x))) * tan(x);
    case EXA_U_ASIN: return x / ((1 - (sq(x))) * sqrt(1 - (sq(x))));
    case EXA_U_ACOS: return (-x) / ((1 - (sq(x))) * sqrt(1 - (sq(x))));
    case EXA_U_ATAN: return (-2 * x) / (sq(1 + sq(x)));
    case EXA_U_ACOT: return (2 * x) / (sq(1 + sq(x)));
    case EXA_U_CSC: return -(-1 - (sq(cot_(x)))) * csc_(x) + (sq(cot_(x))) * csc_(x);
    case EXA_U_SEC: return cb(sec_(x)) + sec_(x) * (sq(tan(x)));
    case EXA_U_COT: return -2 * cot_(x) * (-1 - (sq(cot_(x))));
    case EXA_U_SINH: return sinh(x);
    case EXA_U_COSH: return cosh(x);
    case EXA_U_TANH: return -2 * tanh(x) * (1 - (sq(tanh(x))));
    case EXA_U_ASINH: return (-x) / ((1 + sq(x)) * sqrt(1 + sq(x)));
    case EXA_U_ACOSH: return (-x) / ((-1 + sq(x)) * sqrt(-1 + sq(x)));
    case EXA_U_CSCH: return cb(csch_(x)) + csch_(x) * (sq(coth_(x)));
    case EXA_U_SECH: return -(1 - (sq(tanh(x)))) * sech_(x) + (sq(tanh(x))) * sech_(x);
    case EXA_U_COTH: return 2 * (sq(csch_(x))) * coth_(x);
    case EXA_U_SIND: return -CD2R * d2r(sind_(x));
    case EXA_U_COSD: return -CD2R * d2r(cosd_(x));
    case EXA_U_TAND: return (2 * CD2R) * tand_(x) * d2r(1 + sq(tand_(x)));
    case EXA_U_CSCD: return -CD2R * (-d2r(cscd_(x) * cotd_(x)) * cotd_(x) - cscd_(x) * d2r(1 + sq(cotd_(x))));
    case EXA_U_SECD: return CD2R * (d2r(tand_(x) * secd_(x)) * tand_(x) + d2r(1 + sq(tand_(x))) * secd_(x));
    case EXA_U_COTD: return (2 * CD2R) * cotd_(x) * d2r(1 + sq(cotd_(x)));
    case EXA_U_ATAND: return (-(2 * CD2R) * x) / (sq(d2r(1 + sq(x))));
    case EXA_U_ACOTD: return ((2 * CD2R) * x) / (sq(d2r(1 + sq(x))));
    case EXA_U_SINPI: return -sq(M_PI) * sinpi_(x);
    case EXA_U_COSPI: return -sq(M_PI) * cospi_(x);
    case EXA_U_SINC:
        return ((2 * sq(M_PI)) * sinpi_(x) - (2 * cb(M_PI)) * x * cospi_(x) - pow_int(M_PI, 4) * (sq(x)) * sinpi_(x)) /
               (cb(M_PI) * (cb(x)));
    case EXA_U_DEG2RAD: case EXA_U_RAD2DEG: case EXA_U_SIGNBIT: case EXA_U_FLOOR: case EXA_U_CEIL: return 0.0;
    case EXA_U_ATANH: return fabs(x) > 1.0 ? NAN : (-sq(1.0 / (1 - sq(x)))) * (-2 * x);
    case EXA_U_ACOTH: return fabs(x) < 1.0 ? NAN : (-sq(1.0 / (1 - sq(x)))) * (-2 * x);
    }
    return NAN;
}

/* bivariate: operands typed because `^` distinguishes Int exponents (Base.^ dispatch) */
static double bin_f(int fn, val_t a, val_t b) {
    double x1 = asf(a), x2 = asf(b);
    if (fn >= EXA_B_FIRST_SPECIAL) return sfb_f(fn, x1, x2);
    switch (fn) {
    case EXA_B_ADD: return x1 + x2;   case EXA_B_SUB: return x1 - x2;
    case EXA_B_MUL: return x1 * x2;   case EXA_B_DIV: return x1 / x2;
    case EXA_B_POW: return powv(x1, b);
    case EXA_B_ATAN2: return atan2(x1, x2);
    case EXA_B_HYPOT: return hypot(x1, x2);
    case EXA_B_MAX: return (x1 > x2 || x1 != x1) ? x1 : x2;
    case EXA_B_MIN: return (x1 < x2 || x1 != x1) ? x1 : x2;
    }
    return NAN;
}
static inline val_t vadd(val_t v, int64_t k) { return v.is_int ? VI(v.i + k) : VF(v.f + (double)k); }
static double bin_d1(int fn, val_t a, val_t b) {
    double x1 = asf(a), x2 = asf(b);
    if (fn >= EXA_B_FIRST_SPECIAL) return sfb_d1(fn, x1, x2);
    switch (fn) {
    case EXA_B_ADD: return 1.0;       case EXA_B_SUB: return 1.0;
    case EXA_B_MUL: return x2;        case EXA_B_DIV: return 1 / x2;
    case EXA_B_POW: return x2 * powv(x1, vadd(b, -1));
    case EXA_B_ATAN2: return x2 / (sq(x1) + sq(x2));
    case EXA_B_HYPOT: return x1 / hypot(x1, x2);
    case EXA_B_MAX: return x1 > x2 ? 1.0 : 0.0;
    case EXA_B_MIN: return x1 < x2 ? 1.0 : 0.0;
    }
    return NAN;
}
static double bin_d2(int fn, val_t a, val_t b) {
    double x1 = asf(a), x2 = asf(b);
    if (fn >= EXA_B_FIRST_SPECIAL) return sfb_d2(fn, x1, x2);
    switch (fn) {
    case EXA_B_ADD: return 1.0;       case EXA_B_SUB: return -1.0;
    case EXA_B_MUL: return x1;        case EXA_B_DIV: return (-x1) / (sq(x2));
    case EXA_B_POW: return log(x1) * powv(x1, b);
    case EXA_B_ATAN2: return (-x1) / (sq(x1) + sq(x2));
    case EXA_B_HYPOT: return x2 / hypot(x1, x2);
    case EXA_B_MAX: return x1 > x2 ? 0.0 : 1.0;
    case EXA_B_MIN: return x1 < x2 ? 0.0 : 1.0;
    }
    return NAN;
}
static double bin_d11(int fn, val_t a, val_t b) {
    double x1 = asf(a), x2 = asf(b);
    if (fn >= EXA_B_FIRST_SPECIAL) return sfb_d11(fn, x1, x2);
    switch (fn) {
    case EXA_B_POW: return asf(vadd(b, -1)) * x2 * powv(x1, vadd(b, -2));
    case EXA_B_ATAN2: return (-2 * x1 * x2) / (sq(sq(x1) + sq(x2)));
    case EXA_B_HYPOT: return (-(sq(x1)) + sq(hypot(x1, x2))) / (cb(hypot(x1, x2)));
    default: return 0.0;
    }
}
static double bin_d12(int fn, val_t a, val_t b) {
    double x1 = asf(a), x2 = asf(b);
    if (fn >= EXA_B_FIRST_SPECIAL) return sfb_d12(fn, x1, x2);
    switch (fn) {
    case EXA_B_MUL: return 1.0;
    case EXA_B_DIV: return -1 / (sq(x2));
    case EXA_B_POW: return powv(x1, vadd(b, -1)) + x2 * powv(x1, vadd(b, -1)) * log(x1);
    case EXA_B_ATAN2: return (sq(x1) - (sq(x2))) / (pow_int(x1, 4) + 2 * (sq(x1)) * (sq(x2)) + pow_int(x2, 4));
    case EXA_B_HYPOT: return (-x1 * x2) / (cb(hypot(x1, x2)));
    default: return 0.0;
    }
}
static double bin_d22(int fn, val_t a, val_t b) {
    double x1 = asf(a), x2 = asf(b);
    if (fn >= EXA_B_FIRST_SPECIAL) return sfb_d22(fn, x1, x2);
    switch (fn) {
    case EXA_B_DIV: return (2 * x1) / (cb(x2));
    case EXA_B_POW: return (sq(log(x1))) * powv(x1, b);
    case EXA_B_ATAN2: return (2 * x1 * x2) / (sq(sq(x1) + sq(x2)));
    case EXA_B_HYPOT: return (-(sq(x2)) + sq(hypot(x1, x2))) / (cb(hypot(x1, x2)));
    default: return 0.0;
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* model storage                                                                                     */
/* ------------------------------------------------------------------------------------------------ */
enum { K_CONST = 0, K_VAR = 1, K_UN = 2, K_BIN = 3, K_NULL = 4 };
enum { FX_NONE = 0, FX_FIRST = 1, FX_SECOND = 2 };

typedef struct adnode {
    int kind, fn, fixed;
    int ir;              /* K_CONST: IR root of the Real subtree; K_VAR: IR root of the index expression */
    int cir;             /* fixed binary: IR root of the constant operand */
    int key;             /* K_VAR: id of the structural key of the index expression */
    struct adnode *l, *r;
    /* per-evaluation state */
    double x, y1, y2, h11, h12, h22;
    int64_t vi;
} adnode;

typedef struct {
    int kind;
    int n_nodes; exa_node_t *nodes; int root, target, base;
    int n_cols; exa_column_t *cols;
    int64_t n;
    int64_t o0, o1, o2; int o1step, o2step;
    int n1, n2;            /* number of leaf visits in the 1st / 2nd order traversals */
    int *comp1, *comp2;    /* 1-based slot maps */
    int *isconst;          /* per IR node: subtree has no VAR */
    int nad;               /* number of AD nodes in the template */
} pattern;

typedef struct {
    int64_t nvar, npar, ncon, nnzj, nnzh, nnzg, nobj, nconaug;
    int npat; pattern *pat;
    double *theta;
    double *x0, *lvar, *uvar, *lcon, *ucon;
    int nthreads;
    int rank, world;   /* iterator shard of every pattern: [part_lo(n, rank), part_lo(n, rank + 1)) (tests of the N>1 host logic) */
} ora_model;

/* ------------------------------------------------------------------------------------------------ */
/* constant / index sub-expression evaluation (primal `node(i, x, θ)`, src/graph.jl:305-323)           */
/* ------------------------------------------------------------------------------------------------ */
static val_t col_val(const pattern *p, int c, int64_t I) {
    const exa_column_t *col = &p->cols[c];
    if (col->type == EXA_COL_RANGE) return VI(col->start + col->step * I);
    if (col->type == EXA_COL_I64) return VI(((const int64_t *)col->data)[I]);
    return VF(((const double *)col->data)[I]);
}

static val_t ev(const pattern *p, int k, int64_t I, const double *x, const double *theta) {
    const exa_node_t *nd = &p->nodes[k];
    switch (nd->op) {
    case EXA_OP_CONST_F: return VF(nd->fval);
    case EXA_OP_CONST_I: return VI(nd->ival);
    case EXA_OP_NULLV: return VF(nd->fval);
    case EXA_OP_DATA: return col_val(p, nd->a, I);
    case EXA_OP_PAR: { val_t i = ev(p, nd->a, I, x, theta); return VF(theta[i.i - 1]); }
    case EXA_OP_VAR: { val_t i = ev(p, nd->a, I, x, theta); return VF(x[i.i - 1]); }
    case EXA_OP_UN: {
        val_t a = ev(p, nd->a, I, x, theta);
        if (a.is_int) {
            if (nd->fn == EXA_U_PLUS) return a;
            if (nd->fn == EXA_U_MINUS) return VI(-a.i);
            if (nd->fn == EXA_U_ABS) return VI(a.i < 0 ? -a.i : a.i);
            if (nd->fn == EXA_U_ABS2) return VI(a.i * a.i);
        }
        return VF(un_f(nd->fn, asf(a)));
    }
    case EXA_OP_BIN: {
        val_t a = ev(p, nd->a, I, x, theta), b = ev(p, nd->b, I, x, theta);
        if (a.is_int && b.is_int) {
            switch (nd->fn) {
            case EXA_B_ADD: return VI(a.i + b.i);
            case EXA_B_SUB: return VI(a.i - b.i);
            case EXA_B_MUL: return VI(a.i * b.i);
            case EXA_B_POW: if (b.i >= 0) return VI(ipow_int(a.i, b.i)); break;
            case EXA_B_MAX: return VI(a.i > b.i ? a.i : b.i);
            case EXA_B_MIN: return VI(a.i < b.i ? a.i : b.i);
            default: break;
            }
        }
        return VF(bin_f(nd->fn, a, b));
    }
    }
    return VF(NAN);
}

/* ------------------------------------------------------------------------------------------------ */
/* AD-tree template construction                                                                     */
/* ------------------------------------------------------------------------------------------------ */
static void mark_const(pattern *p) {
    p->isconst = (int *)calloc(p->n_nodes, sizeof(int));
    for (int k = 0; k < p->n_nodes; k++) {
        const exa_node_t *nd = &p->nodes[k];
        switch (nd->op) {
        case EXA_OP_VAR: p->isconst[k] = 0; break;
        case EXA_OP_UN: p->isconst[k] = p->isconst[nd->a]; break;
        case EXA_OP_BIN: p->isconst[k] = p->isconst[nd->a] && p->isconst[nd->b]; break;
        default: p->isconst[k] = 1; break;  /* CONST, DATA, PAR (theta is constant for AD), NULLV */
        }
    }
}

/* structural key of an index expression: canonical text of the IR subtree */
static void key_text(const pattern *p, int k, char **buf, size_t *len, size_t *cap) {
    const exa_node_t *nd = &p->nodes[k];
    char tmp[64];
    int w = 0;
    switch (nd->op) {
    case EXA_OP_CONST_I: w = snprintf(tmp, sizeof tmp, "i%lld", (long long)nd->ival); break;
    case EXA_OP_CONST_F: w = snprintf(tmp, sizeof tmp, "f%a", nd->fval); break;
    case EXA_OP_DATA: w = snprintf(tmp, sizeof tmp, "d%d", nd->a); break;
    default: w = snprintf(tmp, sizeof tmp, "(%d.%d", nd->op, nd->fn); break;
    }
    if (*len + w + 4 > *cap) { *cap = (*cap + w + 4) * 2; *buf = (char *)realloc(*buf, *cap); }
    memcpy(*buf + *len, tmp, w); *len += w;
    if (nd->op == EXA_OP_UN || nd->op == EXA_OP_BIN || nd->op == EXA_OP_VAR || nd->op == EXA_OP_PAR) {
        (*buf)[(*len)++] = ' ';
        key_text(p, nd->a, buf, len, cap);
        if (nd->op == EXA_OP_BIN) { (*buf)[(*len)++] = ' '; key_text(p, nd->b, buf, len, cap); }
        if (*len + 2 > *cap) { *cap *= 2; *buf = (char *)realloc(*buf, *cap); }
        (*buf)[(*len)++] = ')';
    }
    (*buf)[*len] = 0;
}

typedef struct { char **keys; int nkeys; } keytab;
static int key_id(keytab *kt, const pattern *p, int k) {
    char *buf = (char *)malloc(64); size_t len = 0, cap = 64; buf[0] = 0;
    key_text(p, k, &buf, &len, &cap);
    for (int i = 0; i < kt->nkeys; i++) if (!strcmp(kt->keys[i], buf)) { free(buf); return i; }
    kt->keys = (char **)realloc(kt->keys, sizeof(char *) * (kt->nkeys + 1));
    kt->keys[kt->nkeys] = buf;
    return kt->nkeys++;
}

static adnode *build_ad(pattern *p, int k, keytab *kt) {
    const exa_node_t *nd = &p->nodes[k];
    adnode *a = (adnode *)calloc(1, sizeof(adnode));
    p->nad++;
    a->ir = k; a->cir = -1; a->key = -1;
    if (nd->op == EXA_OP_NULLV) { a->kind = K_NULL; return a; }
    if (p->isconst[k]) { a->kind = K_CONST; return a; }
    if (nd->op == EXA_OP_VAR) { a->kind = K_VAR; a->ir = nd->a; a->key = key_id(kt, p, nd->a); return a; }
    if (nd->op == EXA_OP_UN) { a->kind = K_UN; a->fn = nd->fn; a->l = build_ad(p, nd->a, kt); return a; }
    /* BIN */
    a->fn = nd->fn;
    int ca = p->isconst[nd->a] && p->nodes[nd->a].op != EXA_OP_NULLV;
    int cbb = p->isconst[nd->b] && p->nodes[nd->b].op != EXA_OP_NULLV;
    if (cbb) { a->kind = K_UN; a->fixed = FX_SECOND; a->cir = nd->b; a->l = build_ad(p, nd->a, kt); }        /* register.jl:231-248 */
    else if (ca) { a->kind = K_UN; a->fixed = FX_FIRST; a->cir = nd->a; a->l = build_ad(p, nd->b, kt); }   /* register.jl:249-266 */
    else { a->kind = K_BIN; a->l = build_ad(p, nd->a, kt); a->r = build_ad(p, nd->b, kt); }                 /* register.jl:209-230 */
    return a;
}

static adnode *clone_ad(const adnode *a) {
    if (!a) return NULL;
    adnode *c = (adnode *)malloc(sizeof(adnode));
    *c = *a; c->l = clone_ad(a->l); c->r = clone_ad(a->r);
    return c;
}
static void free_ad(adnode *a) { if (!a) return; free_ad(a->l); free_ad(a->r); free(a); }

/* ------------------------------------------------------------------------------------------------ */
/* forward sweeps                                                                                    */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { const pattern *p; int64_t I; const double *x, *theta; } ctx;

/* order = 1: value + first partials; order = 2: + second partials (register.jl:65-68, 209-266) */
static void fwd(adnode *a, const ctx *c, int order) {
    switch (a->kind) {
    case K_NULL: a->x = c->p->nodes[a->ir].fval; return;
    case K_CONST: a->x = asf(ev(c->p, a->ir, c->I, c->x, c->theta)); return;
    case K_VAR: a->vi = ev(c->p, a->ir, c->I, c->x, c->theta).i; a->x = c->x ? c->x[a->vi - 1] : NAN; return;
    case K_UN:
        fwd(a->l, c, order);
        if (a->fixed == FX_NONE) {
            double u = a->l->x;
            a->x = un_f(a->fn, u); a->y1 = un_df(a->fn, u);
            if (order > 1) a->h11 = un_ddf(a->fn, u);
        } else if (a->fixed == FX_SECOND) {
            val_t x1 = VF(a->l->x), x2 = ev(c->p, a->cir, c->I, c->x, c->theta);
            a->x = bin_f(a->fn, x1, x2); a->y1 = bin_d1(a->fn, x1, x2);
            if (order > 1) a->h11 = bin_d11(a->fn, x1, x2);
        } else {
            val_t x1 = ev(c->p, a->cir, c->I, c->x, c->theta), x2 = VF(a->l->x);
            a->x = bin_f(a->fn, x1, x2); a->y1 = bin_d2(a->fn, x1, x2);
            if (order > 1) a->h11 = bin_d22(a->fn, x1, x2);
        }
        return;
    case K_BIN: {
        fwd(a->l, c, order); fwd(a->r, c, order);
        val_t x1 = VF(a->l->x), x2 = VF(a->r->x);
        a->x = bin_f(a->fn, x1, x2); a->y1 = bin_d1(a->fn, x1, x2); a->y2 = bin_d2(a->fn, x1, x2);
        if (order > 1) { a->h11 = bin_d11(a->fn, x1, x2); a->h12 = bin_d12(a->fn, x1, x2); a->h22 = bin_d22(a->fn, x1, x2); }
        return;
    }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* reverse sweeps with pluggable leaf actions                                                         */
/* ------------------------------------------------------------------------------------------------ */
enum { S_COLLECT, S_VALUES, S_DENSE, S_STRUCT, S_JV, S_JTV, S_HV };
typedef struct {
    int mode; int cnt;
    /* S_COLLECT */ int *raw; int nraw, capraw;             /* 1st order: key; 2nd order: key1*65536+key2 */
    /* S_VALUES  */ double *out; int64_t base; const int *comp;
    /* S_DENSE   */ double *dense;
    /* S_STRUCT  */ int64_t *rows, *cols; int64_t row;
    /* S_JV/S_JTV/S_HV: matrix-free products (jacobian.jl:41-68, hessian.jl:291-315, 566-579) */ double *py; const double *pv;
} sink;

static void push_raw(sink *s, int v) {
    if (s->nraw == s->capraw) { s->capraw = s->capraw ? 2 * s->capraw : 32; s->raw = (int *)realloc(s->raw, sizeof(int) * s->capraw); }
    s->raw[s->nraw++] = v;
}

static void leaf1(sink *s, adnode *v, double adj) {
    switch (s->mode) {
    case S_COLLECT: push_raw(s, v->key); s->cnt++; break;
    case S_VALUES: s->out[s->base + s->comp[s->cnt++] - 1] += adj; break;                 /* gradient.jl:83-86, jacobian.jl:37-40 */
    case S_DENSE: s->dense[v->vi - 1] += adj; s->cnt++; break;                              /* gradient.jl:23-26 */
    case S_STRUCT: { int64_t ind = s->base + s->comp[s->cnt++] - 1; s->rows[ind] = s->row; s->cols[ind] = v->vi; break; } /* jacobian.jl:69-83 */
    case S_JV: s->py[s->row - 1] += adj * s->pv[v->vi - 1]; s->cnt++; break;      /* jacobian.jl:41-54 */
    case S_JTV: s->py[v->vi - 1] += adj * s->pv[s->row - 1]; s->cnt++; break;     /* jacobian.jl:55-68 */
    default: break;
    }
}

/* gradient.jl:64-86 / jacobian.jl:16-40 */
static void grpass(adnode *a, sink *s, double adj) {
    switch (a->kind) {
    case K_CONST: case K_NULL: return;
    case K_UN: grpass(a->l, s, adj * a->y1); return;
    case K_BIN: grpass(a->l, s, adj * a->y1); grpass(a->r, s, adj * a->y2); return;
    case K_VAR: leaf1(s, a, adj); return;
    }
}

static void leaf2(sink *s, adnode *v1, adnode *v2, double val) {
    switch (s->mode) {
    case S_COLLECT: push_raw(s, v1->key * 65536 + v2->key); s->cnt++; break;
    case S_VALUES: s->out[s->base + s->comp[s->cnt++] - 1] += val; break;
    case S_STRUCT: {                                                                         /* hessian.jl:593-642 */
        int64_t ind = s->base + s->comp[s->cnt++] - 1, i = v1->vi, j = v2->vi;
        if (i >= j) { s->rows[ind] = i; s->cols[ind] = j; } else { s->rows[ind] = j; s->cols[ind] = i; }
        break;
    }
    case S_HV: {   /* hessian.jl:291-315 (cross) / 566-579 (leaf): val is adj (cross) or adj2 (leaf) */
        const int64_t i = v1->vi, j = v2->vi;
        if (v1 == v2) s->py[i - 1] += val * s->pv[i - 1];
        else if (i == j) s->py[i - 1] += 2 * val * s->pv[i - 1];
        else { s->py[i - 1] += val * s->pv[j - 1]; s->py[j - 1] += val * s->pv[i - 1]; }
        s->cnt++;
        break;
    }
    default: break;
    }
}

/* hessian.jl:16-268, 317-320 */
static void hdrpass(adnode *t1, adnode *t2, sink *s, double adj) {
    if (t1->kind == K_NULL || t2->kind == K_NULL || t1->kind == K_CONST || t2->kind == K_CONST) return;
    if (t1->kind == K_UN && t2->kind == K_UN) { hdrpass(t1->l, t2->l, s, adj * t1->y1 * t2->y1); return; }
    if (t1->kind == K_VAR && t2->kind == K_UN) { hdrpass(t1, t2->l, s, adj * t2->y1); return; }
    if (t1->kind == K_UN && t2->kind == K_VAR) { hdrpass(t1->l, t2, s, adj * t1->y1); return; }
    if (t1->kind == K_BIN && t2->kind == K_BIN) {
        hdrpass(t1->l, t2->l, s, adj * t1->y1 * t2->y1);
        hdrpass(t1->l, t2->r, s, adj * t1->y1 * t2->y2);
        hdrpass(t1->r, t2->l, s, adj * t1->y2 * t2->y1);
        hdrpass(t1->r, t2->r, s, adj * t1->y2 * t2->y2);
        return;
    }
    if (t1->kind == K_UN && t2->kind == K_BIN) {
        hdrpass(t1->l, t2->l, s, adj * t1->y1 * t2->y1);
        hdrpass(t1->l, t2->r, s, adj * t1->y1 * t2->y2);
        return;
    }
    if (t1->kind == K_BIN && t2->kind == K_UN) {
        hdrpass(t1->l, t2->l, s, adj * t1->y1 * t2->y1);
        hdrpass(t1->r, t2->l, s, adj * t1->y2 * t2->y1);
        return;
    }
    if (t1->kind == K_VAR && t2->kind == K_BIN) {
        hdrpass(t1, t2->l, s, adj * t2->y1);
        hdrpass(t1, t2->r, s, adj * t2->y2);
        return;
    }
    if (t1->kind == K_BIN && t2->kind == K_VAR) {
        hdrpass(t1->l, t2, s, adj * t1->y1);
        hdrpass(t1->r, t2, s, adj * t1->y2);
        return;
    }
    /* VAR x VAR (hessian.jl:251-268) */
    leaf2(s, t1, t2, (s->mode == S_VALUES && t1->vi == t2->vi) ? 2 * adj : adj);   /* S_HV applies the 2x itself */
}

/* hessian.jl:337-380, 580-592 */
static void hrpass(adnode *t, sink *s, double adj, double adj2) {
    switch (t->kind) {
    case K_CONST: case K_NULL: return;
    case K_UN: hrpass(t->l, s, adj * t->y1, adj2 * (t->y1 * t->y1) + adj * t->h11); return;
    case K_BIN: {
        double adj2y1y2 = adj2 * t->y1 * t->y2;
        double adjh12 = adj * t->h12;
        hrpass(t->l, s, adj * t->y1, adj2 * (t->y1 * t->y1) + adj * t->h11);
        hrpass(t->r, s, adj * t->y2, adj2 * (t->y2 * t->y2) + adj * t->h22);
        hdrpass(t->l, t->r, s, adj2y1y2 + adjh12);
        return;
    }
    case K_VAR: leaf2(s, t, t, adj2); return;
    }
}

/* hessian.jl:382-517: top-level linear peeling */
static void hrpass0(adnode *t, sink *s, double adj, double adj2) {
    if (t->kind == K_UN && t->fixed != FX_NONE) {
        if (t->fn == EXA_B_MUL) { hrpass0(t->l, s, adj * t->y1, adj2 * (t->y1 * t->y1)); return; }
        if (t->fn == EXA_B_ADD) { hrpass0(t->l, s, adj, adj2); return; }
        if (t->fn == EXA_B_SUB) { hrpass0(t->l, s, t->fixed == FX_FIRST ? -adj : adj, adj2); return; }
    } else if (t->kind == K_UN) {
        if (t->fn == EXA_U_PLUS) { hrpass0(t->l, s, adj, adj2); return; }
        if (t->fn == EXA_U_MINUS) { hrpass0(t->l, s, -adj, adj2); return; }
    } else if (t->kind == K_BIN) {
        if (t->fn == EXA_B_ADD) { hrpass0(t->l, s, adj, adj2); hrpass0(t->r, s, adj, adj2); return; }
        if (t->fn == EXA_B_SUB) { hrpass0(t->l, s, adj, adj2); hrpass0(t->r, s, -adj, adj2); return; }
    } else if (t->kind == K_VAR) {
        return;
    }
    hrpass(t, s, adj, adj2);
}

/* ------------------------------------------------------------------------------------------------ */
/* planning: slot maps and offsets                                                                    */
/* ------------------------------------------------------------------------------------------------ */
static int *dedup(const int *raw, int n, int *nuniq) {   /* simdfunction.jl:63-76, 89-97 */
    int *comp = (int *)malloc(sizeof(int) * (n ? n : 1));
    int *uniq = (int *)malloc(sizeof(int) * (n ? n : 1));
    int nu = 0;
    for (int i = 0; i < n; i++) {
        int f = -1;
        for (int j = 0; j < nu; j++) if (uniq[j] == raw[i]) { f = j; break; }
        if (f < 0) { uniq[nu] = raw[i]; f = nu++; }
        comp[i] = f + 1;
    }
    free(uniq);
    *nuniq = nu;
    return comp;
}

static adnode *plan_pattern(pattern *p) {
    mark_const(p);
    keytab kt = {0, 0};
    p->nad = 0;
    adnode *t = build_ad(p, p->root, &kt);
    sink s1; memset(&s1, 0, sizeof s1); s1.mode = S_COLLECT;
    grpass(t, &s1, NAN);
    p->n1 = s1.nraw; p->comp1 = dedup(s1.raw, s1.nraw, &p->o1step); free(s1.raw);
    sink s2; memset(&s2, 0, sizeof s2); s2.mode = S_COLLECT;
    hrpass0(t, &s2, NAN, NAN);
    p->n2 = s2.nraw; p->comp2 = dedup(s2.raw, s2.nraw, &p->o2step); free(s2.raw);
    for (int i = 0; i < kt.nkeys; i++) free(kt.keys[i]);
    free(kt.keys);
    return t;
}

/* ------------------------------------------------------------------------------------------------ */
/* public API                                                                                        */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { ora_model m; adnode **tmpl; } ora_handle;

static double *dupd(const double *src, int64_t n, double fill) {
    double *d = (double *)malloc(sizeof(double) * (n ? n : 1));
    for (int64_t i = 0; i < n; i++) d[i] = src ? src[i] : fill;
    return d;
}

void *ora_new(const exa_model_desc_t *d) {
    ora_handle *h = (ora_handle *)calloc(1, sizeof(ora_handle));
    ora_model *m = &h->m;
    m->nvar = d->nvar; m->npar = d->npar; m->npat = d->n_patterns; m->nthreads = 1; m->rank = 0; m->world = 1;
    m->pat = (pattern *)calloc(m->npat ? m->npat : 1, sizeof(pattern));
    h->tmpl = (adnode **)calloc(m->npat ? m->npat : 1, sizeof(adnode *));
    m->theta = dupd(d->theta0, d->npar, 0.0);
    m->x0 = dupd(d->x0, d->nvar, 0.0); m->lvar = dupd(d->lvar, d->nvar, -INFINITY); m->uvar = dupd(d->uvar, d->nvar, INFINITY);
    for (int k = 0; k < m->npat; k++) {
        const exa_pattern_t *s = &d->patterns[k];
        pattern *p = &m->pat[k];
        p->kind = s->kind; p->n_nodes = s->n_nodes; p->root = s->root; p->target = s->target; p->base = s->base;
        p->n_cols = s->n_cols; p->n = s->n;
        p->nodes = (exa_node_t *)malloc(sizeof(exa_node_t) * s->n_nodes);
        memcpy(p->nodes, s->nodes, sizeof(exa_node_t) * s->n_nodes);
        p->cols = (exa_column_t *)calloc(s->n_cols ? s->n_cols : 1, sizeof(exa_column_t));
        for (int c = 0; c < s->n_cols; c++) {
            p->cols[c] = s->cols[c];
            if (s->cols[c].type != EXA_COL_RANGE) {
                void *cp = malloc(8 * (s->n ? s->n : 1));
                memcpy(cp, s->cols[c].data, 8 * s->n);
                p->cols[c].data = cp;
            }
        }
        h->tmpl[k] = plan_pattern(p);
        /* running counters in insertion order (nlp.jl:1474-1482, 1597-1611, 1730-1738) */
        if (p->kind == EXA_PAT_OBJ) {
            p->o0 = m->nobj; p->o1 = m->nnzg; p->o2 = m->nnzh;
            m->nobj += p->n; m->nnzg += p->n * p->o1step; m->nnzh += p->n * p->o2step;
        } else if (p->kind == EXA_PAT_CON) {
            p->o0 = m->ncon; p->o1 = m->nnzj; p->o2 = m->nnzh;
            m->ncon += p->n; m->nnzj += p->n * p->o1step; m->nnzh += p->n * p->o2step;
        } else {
            p->o0 = m->pat[p->base].o0; p->o1 = m->nnzj; p->o2 = m->nnzh;   /* nlp.jl:1683: offset0(c1, 0) */
            m->nconaug += p->n; m->nnzj += p->n * p->o1step; m->nnzh += p->n * p->o2step;
        }
    }
    m->lcon = dupd(d->lcon, m->ncon, 0.0); m->ucon = dupd(d->ucon, m->ncon, 0.0);
    return h;
}

void ora_free(void *hh) {
    ora_handle *h = (ora_handle *)hh;
    ora_model *m = &h->m;
    for (int k = 0; k < m->npat; k++) {
        pattern *p = &m->pat[k];
        for (int c = 0; c < p->n_cols; c++) if (p->cols[c].type != EXA_COL_RANGE) free((void *)p->cols[c].data);
        free(p->cols); free(p->nodes); free(p->comp1); free(p->comp2); free(p->isconst);
        free_ad(h->tmpl[k]);
    }
    free(h->tmpl); free(m->pat); free(m->theta); free(m->x0); free(m->lvar); free(m->uvar); free(m->lcon); free(m->ucon);
    free(h);
}

int64_t ora_nvar(void *h) { return ((ora_handle *)h)->m.nvar; }
int64_t ora_ncon(void *h) { return ((ora_handle *)h)->m.ncon; }
int64_t ora_nnzj(void *h) { return ((ora_handle *)h)->m.nnzj; }
int64_t ora_nnzh(void *h) { return ((ora_handle *)h)->m.nnzh; }
int64_t ora_nnzg(void *h) { return ((ora_handle *)h)->m.nnzg; }
int ora_npatterns(void *h) { return ((ora_handle *)h)->m.npat; }
/* the library's partition (examodels.jl_amd/csrc/exa_internal.hpp part_lo): floor(n / G) items per rank, the last rank the remainder on top — equal
 * pieces: one in-place all-gather completes an owner-sharded vector —; fewer than 16 items per rank: floor(n r / G) */
static inline int64_t ora_part_lo(int64_t n, int r, int G) {
    if (r >= G) return n;
    return n >= 16LL * G ? (n / G) * (int64_t)r : (int64_t)((__int128)n * r / G);
}
static inline int64_t shard_lo(const ora_model *m, const pattern *p) { return ora_part_lo(p->n, m->rank, m->world); }
static inline int64_t shard_hi(const ora_model *m, const pattern *p) { return ora_part_lo(p->n, m->rank + 1, m->world); }
void ora_set_shard(void *h, int rank, int world) { ((ora_handle *)h)->m.rank = rank; ((ora_handle *)h)->m.world = world; }
void ora_set_threads(void *h, int n) { ((ora_handle *)h)->m.nthreads = n < 1 ? 1 : n; }
void ora_set_theta(void *h, int64_t off, const double *v, int64_t len) { memcpy(((ora_handle *)h)->m.theta + off, v, sizeof(double) * len); }

void ora_pattern_info(void *hh, int k, int64_t out[9]) {
    pattern *p = &((ora_handle *)hh)->m.pat[k];
    out[0] = p->kind; out[1] = p->n; out[2] = p->o0; out[3] = p->o1; out[4] = p->o2;
    out[5] = p->o1step; out[6] = p->o2step; out[7] = p->n1; out[8] = p->n2;
}
void ora_pattern_comp(void *hh, int k, int order, int32_t *out) {
    pattern *p = &((ora_handle *)hh)->m.pat[k];
    if (order == 1) for (int i = 0; i < p->n1; i++) out[i] = p->comp1[i];
    else for (int i = 0; i < p->n2; i++) out[i] = p->comp2[i];
}
void ora_meta(void *hh, double *x0, double *lvar, double *uvar, double *lcon, double *ucon) {
    ora_model *m = &((ora_handle *)hh)->m;
    memcpy(x0, m->x0, 8 * m->nvar); memcpy(lvar, m->lvar, 8 * m->nvar); memcpy(uvar, m->uvar, 8 * m->nvar);
    memcpy(lcon, m->lcon, 8 * m->ncon); memcpy(ucon, m->ucon, 8 * m->ncon);
}

/* row of data point I: offset0 (nlp.jl:1980-2001), 0-based result */
static inline int64_t row_of(const pattern *p, int64_t I, const double *theta) {
    if (p->kind == EXA_PAT_CONAUG) return p->o0 + ev(p, p->target, I, NULL, theta).i - 1;
    return p->o0 + I;
}

/* obj (nlp.jl:1827-1839): sequential sum, patterns in insertion order */
double ora_obj(void *hh, const double *x) {
    ora_model *m = &((ora_handle *)hh)->m;
    double s = 0.0;
    for (int k = 0; k < m->npat; k++) {
        pattern *p = &m->pat[k];
        if (p->kind != EXA_PAT_OBJ) continue;
        for (int64_t I = shard_lo(m, p); I < shard_hi(m, p); I++) s += asf(ev(p, p->root, I, x, m->theta));
    }
    return s;
}

/* cons_nln! (nlp.jl:1841-1854) */
void ora_cons(void *hh, const double *x, double *g) {
    ora_model *m = &((ora_handle *)hh)->m;
    for (int64_t i = 0; i < m->ncon; i++) g[i] = 0.0;
    for (int k = 0; k < m->npat; k++) {
        pattern *p = &m->pat[k];
        if (p->kind == EXA_PAT_OBJ) continue;
        for (int64_t I = shard_lo(m, p); I < shard_hi(m, p); I++) g[row_of(p, I, m->theta)] += asf(ev(p, p->root, I, x, m->theta));
    }
}

/* cons_nln! once more in __float128, with the magnitude of what each row summed (exa_quad.h): the arbiter for rows that cancel.
 * g_q[i] = the quad result rounded to double, mag[i] = the first-order running error bound of the row's double-precision evaluation in
 * units of eps (the contributions' bounds + one rounding per addition into the row).
 * Returns 1, or 0 when a pattern uses a function without a quad restatement (the outputs are then meaningless). */
#include "exa_quad.h"
int ora_cons_quad(void *hh, const double *x, double *g_q, double *mag) {
    ora_model *m = &((ora_handle *)hh)->m;
    int ok = 1;
    q128 *acc = (q128 *)calloc((size_t)(m->ncon > 0 ? m->ncon : 1), sizeof(q128));
    q128 *mg = (q128 *)calloc((size_t)(m->ncon > 0 ? m->ncon : 1), sizeof(q128));
    for (int k = 0; k < m->npat; k++) {
        pattern *p = &m->pat[k];
        if (p->kind == EXA_PAT_OBJ) continue;
        const int64_t lo = shard_lo(m, p), hi = shard_hi(m, p);
        /* base rows: one row per data point, disjoint — over the model's threads; augmentation terms share rows: sequential */
        const int nt = p->kind == EXA_PAT_CON ? m->nthreads : 1;
        int okp = 1;
        (void)nt;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nt) reduction(&& : okp) if (nt > 1)
#endif
        for (int64_t I = lo; I < hi; I++) {
            int ok1 = 1;
            const qval_t v = qev(p, p->root, I, x, m->theta, &ok1);
            const int64_t r = row_of(p, I, m->theta);
            const int first = acc[r] == 0 && mg[r] == 0;
            acc[r] += v.f;
            mg[r] += v.mag + (first ? 0.0Q : fabsq(acc[r]));      /* every further addition into the row rounds once more */
            okp = okp && ok1;
        }
        ok = ok && okp;
    }
    for (int64_t i = 0; i < m->ncon; i++) { g_q[i] = (double)acc[i]; mag[i] = (double)mg[i]; }
    free(acc); free(mg);
    return ok;
}

/* generic driver over data points for the derivative callbacks */
typedef void (*point_fn)(const pattern *p, adnode *t, const ctx *c, void *user);
static void drive(ora_handle *h, int k, const double *x, point_fn fn, void *user, int parallel) {
    ora_model *m = &h->m;
    pattern *p = &m->pat[k];
    int nt = parallel ? m->nthreads : 1;
    const int64_t lo = shard_lo(m, p), hi = shard_hi(m, p);
    if (nt <= 1) {
        adnode *t = h->tmpl[k];
        for (int64_t I = lo; I < hi; I++) { ctx c = {p, I, x, m->theta}; fn(p, t, &c, user); }
        return;
    }
#ifdef _OPENMP
#pragma omp parallel num_threads(nt)
    {
        adnode *t = clone_ad(h->tmpl[k]);
#pragma omp for schedule(static)
        for (int64_t I = lo; I < hi; I++) { ctx c = {p, I, x, m->theta}; fn(p, t, &c, user); }
        free_ad(t);
    }
#else
    { adnode *t = h->tmpl[k]; for (int64_t I = lo; I < hi; I++) { ctx c = {p, I, x, m->theta}; fn(p, t, &c, user); } }
#endif
}

/* grad! CPU path: dense scatter (nlp.jl:1858-1868, gradient.jl:11-49) — sequential (shared targets) */
static void pt_grad(const pattern *p, adnode *t, const ctx *c, void *user) {
    fwd(t, c, 1);
    sink s; memset(&s, 0, sizeof s); s.mode = S_DENSE; s.dense = (double *)user;
    grpass(t, &s, 1.0);
}
void ora_grad(void *hh, const double *x, double *g) {
    ora_handle *h = (ora_handle *)hh; ora_model *m = &h->m;
    for (int64_t i = 0; i < m->nvar; i++) g[i] = 0.0;
    for (int k = 0; k < m->npat; k++) if (m->pat[k].kind == EXA_PAT_OBJ) drive(h, k, x, pt_grad, g, 0);
}

/* sparse objective gradient buffer (KA ext :310-336 first stage; gradient.jl:170-181) */
static void pt_sgrad(const pattern *p, adnode *t, const ctx *c, void *user) {
    fwd(t, c, 1);
    sink s; memset(&s, 0, sizeof s); s.mode = S_VALUES; s.out = (double *)user;
    s.base = p->o1 + (int64_t)p->o1step * c->I; s.comp = p->comp1;
    grpass(t, &s, 1.0);
}
void ora_sgrad(void *hh, const double *x, double *buf) {
    ora_handle *h = (ora_handle *)hh; ora_model *m = &h->m;
    for (int64_t i = 0; i < m->nnzg; i++) buf[i] = 0.0;
    for (int k = 0; k < m->npat; k++) if (m->pat[k].kind == EXA_PAT_OBJ) drive(h, k, x, pt_sgrad, buf, 1);
}

/* jac_coord! (nlp.jl:1870-1880, jacobian.jl:112-132) */
void ora_jac(void *hh, const double *x, double *jac) {
    ora_handle *h = (ora_handle *)hh; ora_model *m = &h->m;
    for (int64_t i = 0; i < m->nnzj; i++) jac[i] = 0.0;
    for (int k = 0; k < m->npat; k++) if (m->pat[k].kind != EXA_PAT_OBJ) drive(h, k, x, pt_sgrad, jac, 1);
}

/* hess_coord! (nlp.jl:1906-1940, hessian.jl:681-717) */
typedef struct { double *out; const double *y; double sigma; const double *theta; } hess_user;
static void pt_hess(const pattern *p, adnode *t, const ctx *c, void *user) {
    hess_user *u = (hess_user *)user;
    fwd(t, c, 2);
    sink s; memset(&s, 0, sizeof s); s.mode = S_VALUES; s.out = u->out;
    s.base = p->o2 + (int64_t)p->o2step * c->I; s.comp = p->comp2;
    double adj = (p->kind == EXA_PAT_OBJ) ? u->sigma : u->y[row_of(p, c->I, u->theta)];
    hrpass0(t, &s, adj, 0.0);
}
void ora_hess(void *hh, const double *x, const double *y, double sigma, double *hess) {
    ora_handle *h = (ora_handle *)hh; ora_model *m = &h->m;
    for (int64_t i = 0; i < m->nnzh; i++) hess[i] = 0.0;
    hess_user u = {hess, y, sigma, m->theta};
    for (int k = 0; k < m->npat; k++) if (m->pat[k].kind == EXA_PAT_OBJ) drive(h, k, x, pt_hess, &u, 1);
    for (int k = 0; k < m->npat; k++) if (m->pat[k].kind != EXA_PAT_OBJ) drive(h, k, x, pt_hess, &u, 1);
}

/* jprod_nln! / jtprod_nln! (nlp.jl:1882-1904) and hprod! (nlp.jl:1942-1978): sequential, shared targets */
typedef struct { double *out; const double *v; const double *y; double sigma; const double *theta; int mode; } prod_user;
static void pt_jprod(const pattern *p, adnode *t, const ctx *c, void *user) {
    prod_user *u = (prod_user *)user;
    fwd(t, c, 1);
    sink s; memset(&s, 0, sizeof s); s.mode = u->mode; s.py = u->out; s.pv = u->v; s.row = row_of(p, c->I, u->theta) + 1;
    grpass(t, &s, 1.0);
}
void ora_jprod(void *hh, const double *x, const double *v, double *Jv) {
    ora_handle *h = (ora_handle *)hh; ora_model *m = &h->m;
    for (int64_t i = 0; i < m->ncon; i++) Jv[i] = 0.0;
    prod_user u = {Jv, v, NULL, 0.0, m->theta, S_JV};
    for (int k = 0; k < m->npat; k++) if (m->pat[k].kind != EXA_PAT_OBJ) drive(h, k, x, pt_jprod, &u, 0);
}
void ora_jtprod(void *hh, const double *x, const double *v, double *Jtv) {
    ora_handle *h = (ora_handle *)hh; ora_model *m = &h->m;
    for (int64_t i = 0; i < m->nvar; i++) Jtv[i] = 0.0;
    prod_user u = {Jtv, v, NULL, 0.0, m->theta, S_JTV};
    for (int k = 0; k < m->npat; k++) if (m->pat[k].kind != EXA_PAT_OBJ) drive(h, k, x, pt_jprod, &u, 0);
}
static void pt_hprod(const pattern *p, adnode *t, const ctx *c, void *user) {
    prod_user *u = (prod_user *)user;
    fwd(t, c, 2);
    sink s; memset(&s, 0, sizeof s); s.mode = S_HV; s.py = u->out; s.pv = u->v;
    double adj = (p->kind == EXA_PAT_OBJ) ? u->sigma : u->y[row_of(p, c->I, u->theta)];
    hrpass0(t, &s, adj, 0.0);
}
void ora_hprod(void *hh, const double *x, const double *y, const double *v, double sigma, double *Hv) {
    ora_handle *h = (ora_handle *)hh; ora_model *m = &h->m;
    for (int64_t i = 0; i < m->nvar; i++) Hv[i] = 0.0;
    prod_user u = {Hv, v, y, sigma, m->theta, S_HV};
    for (int k = 0; k < m->npat; k++) if (m->pat[k].kind == EXA_PAT_OBJ) drive(h, k, x, pt_hprod, &u, 0);
    for (int k = 0; k < m->npat; k++) if (m->pat[k].kind != EXA_PAT_OBJ) drive(h, k, x, pt_hprod, &u, 0);
}

/* structures (nlp.jl:1798-1825) */
typedef struct { int64_t *rows, *cols; const double *theta; } st_user;
static void pt_jst(const pattern *p, adnode *t, const ctx *c, void *user) {
    st_user *u = (st_user *)user;
    fwd(t, c, 1);
    sink s; memset(&s, 0, sizeof s); s.mode = S_STRUCT; s.rows = u->rows; s.cols = u->cols;
    s.base = p->o1 + (int64_t)p->o1step * c->I; s.comp = p->comp1; s.row = row_of(p, c->I, u->theta) + 1;
    grpass(t, &s, NAN);
}
void ora_jac_structure(void *hh, int64_t *rows, int64_t *cols) {
    ora_handle *h = (ora_handle *)hh; ora_model *m = &h->m;
    st_user u = {rows, cols, m->theta};
    for (int k = 0; k < m->npat; k++) if (m->pat[k].kind != EXA_PAT_OBJ) drive(h, k, NULL, pt_jst, &u, 1);
}
static void pt_hst(const pattern *p, adnode *t, const ctx *c, void *user) {
    st_user *u = (st_user *)user;
    fwd(t, c, 2);
    sink s; memset(&s, 0, sizeof s); s.mode = S_STRUCT; s.rows = u->rows; s.cols = u->cols;
    s.base = p->o2 + (int64_t)p->o2step * c->I; s.comp = p->comp2;
    hrpass0(t, &s, NAN, NAN);
}
void ora_hess_structure(void *hh, int64_t *rows, int64_t *cols) {
    ora_handle *h = (ora_handle *)hh; ora_model *m = &h->m;
    st_user u = {rows, cols, m->theta};
    for (int k = 0; k < m->npat; k++) drive(h, k, NULL, pt_hst, &u, 1);
}

/* ------------------------------------------------------------------------------------------------ */
/* scalar expression helpers for the table tests (ADTest.jl:298-342): f, f', f'' of one table entry    */
/* ------------------------------------------------------------------------------------------------ */
void ora_un_table(int fn, double x, double out[3]) { out[0] = un_f(fn, x); out[1] = un_df(fn, x); out[2] = un_ddf(fn, x); }
void ora_bin_table(int fn, double x1, double x2, double out[6]) {
    val_t a = VF(x1), b = VF(x2);
    out[0] = bin_f(fn, a, b); out[1] = bin_d1(fn, a, b); out[2] = bin_d2(fn, a, b);
    out[3] = bin_d11(fn, a, b); out[4] = bin_d12(fn, a, b); out[5] = bin_d22(fn, a, b);
}
int ora_has_openmp(void) {
#ifdef _OPENMP
    return 1;
#else
    return 0;
#endif
}

/* ------------------------------------------------------------------------------------------------ */
/* Luksan-Vlcek hess_coord!, hand-specialised straight-line C — TEST/BASELINE ONLY.                    */
/* What Julia's compiler makes of shessian! for the two LV patterns (benchmark/runbenchmark.jl:163-169):*/
/* the generic interpreter above pays tree-walking overhead per node that compiled Julia does not, so   */
/* bench.py's cpu_baseline times THIS as the proxy for `backend = nothing`.  Same algorithm: zero-fill  */
/* (nlp.jl:1913) then one `+=` per contribution in hrpass0/hrpass/hdrpass order (hessian.jl), constraint*/
/* block first.  Checked against the interpreter in tests/test_known_answers.py.                        */
/* ------------------------------------------------------------------------------------------------ */
void ora_lv_hess_compiled(int64_t N, const double *x, const double *y, double sigma, double *H, int threads) {
    const int64_t ncon = N - 2, nobj = N - 1;
    const int64_t nnzh = 6 * ncon + 3 * nobj;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t k = 0; k < nnzh; k++) H[k] = 0.0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t I = 0; I < ncon; I++) {            /* i = I+1: a = x[i+1], b = x[i+2], c = x[i] */
        const double a = x[I + 1], b = x[I + 2], c = x[I], lam = y[I];
        double *h = H + 6 * I;
        /* 3a^3: FirstFixed{*}(3, a^3): adj = 3 lam; a^3: y = 3a^2, h = 6a */
        const double y3 = 3.0 * (a * a), h3 = (2 * 3) * a;
        h[0] += 0.0 * (y3 * y3) + (lam * 3.0) * h3;                                   /* (a,a) */
        /* sin(a-b) * sin(a+b) */
        const double u = a - b, v = a + b;
        const double su = sin(u), cu = cos(u), sv = sin(v), cv = cos(v);
        /* Node2{*}(S1, S2): y1 = S2.x, y2 = S1.x, h12 = 1 */
        const double adj1 = lam * sv, adj2 = lam * su;
        /* hrpass(S1 = sin(u)): child u = a - b with adj = adj1*cu, adj2' = 0*.. + adj1*(-su) */
        { const double ad = adj1 * cu, a2 = 0.0 * (cu * cu) + adj1 * (-su);
          /* u = a - b (Node2{-}): hrpass(a, ad*1, a2*1 + ad*0), hrpass(b, ad*-1, a2*1 + ad*0), hdrpass(a,b, a2*1*-1 + ad*0) */
          h[0] += a2 * (1.0 * 1.0) + (ad * 1.0) * 0.0;                                 /* (a,a) */
          h[1] += a2 * (-1.0 * -1.0) + (ad * -1.0) * 0.0;                              /* (b,b) */
          h[2] += a2 * 1.0 * -1.0 + ad * 0.0; }                                        /* (a,b) */
        { const double ad = adj2 * cv, a2 = 0.0 * (cv * cv) + adj2 * (-sv);
          h[0] += a2 * (1.0 * 1.0) + (ad * 1.0) * 0.0;                                 /* (a,a) */
          h[1] += a2 * (1.0 * 1.0) + (ad * 1.0) * 0.0;                                 /* (b,b) */
          h[2] += a2 * 1.0 * 1.0 + ad * 0.0; }                                         /* (a,b) */
        /* hdrpass(S1, S2, adj = 0*.. + lam*1): (u-leaves) x (v-leaves) with adj*cu*cv */
        { const double ad = (0.0 * sv * su + lam * 1.0) * cu * cv;
          h[0] += 2.0 * (ad * 1.0 * 1.0);                                              /* (a,a): i == j -> 2 adj */
          h[2] += ad * 1.0 * 1.0;                                                      /* (a,b) */
          h[3] += ad * -1.0 * 1.0;                                                     /* (b,a) */
          h[1] += 2.0 * (ad * -1.0 * 1.0); }                                           /* (b,b) */
        /* - c * exp(c - a): binary '-' at the top passes -lam to Node2{*}(c, E) */
        { const double e = exp(c - a), nl = -lam;
          /* hrpass(c, nl*e, 0*e*e + nl*0) */
          h[4] += 0.0 * (e * e) + nl * 0.0;                                            /* (c,c) */
          /* hrpass(E = exp(w), adj = nl*c, adj2 = 0*c*c + nl*0) -> w = c - a: ad = adj*e, a2 = adj2*e*e + adj*e */
          const double adE = nl * c, a2E = 0.0 * (c * c) + nl * 0.0;
          const double ad = adE * e, a2 = a2E * (e * e) + adE * e;
          h[4] += a2 * (1.0 * 1.0) + (ad * 1.0) * 0.0;                                 /* (c,c) */
          h[0] += a2 * (-1.0 * -1.0) + (ad * -1.0) * 0.0;                              /* (a,a) */
          h[5] += a2 * 1.0 * -1.0 + ad * 0.0;                                          /* (c,a) */
          /* hdrpass(c, E, adj = 0*e*c + nl*1): var x unary -> adj*e; then var x Node2{-}: (c,c) 2adj, (c,a) -adj */
          const double adx = (0.0 * e * c + nl * 1.0) * e;
          h[4] += 2.0 * (adx * 1.0);                                                   /* (c,c) */
          h[5] += adx * -1.0; }                                                        /* (c,a) */
    }
    double *Ho = H + 6 * ncon;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t I = 0; I < nobj; I++) {            /* i = I+2: p = x[i-1], q = x[i] */
        const double p = x[I], q = x[I + 1];
        double *h = Ho + 3 * I;
        /* 100 * (p^2 - q)^2: FirstFixed{*} -> adj = 100 sigma; abs2(w): y = 2w, h = 2; w = abs2(p) - q */
        const double w = p * p - q, adj = sigma * 100.0;
        const double ad = adj * (2.0 * w), a2 = 0.0 * ((2.0 * w) * (2.0 * w)) + adj * 2.0;
        /* Node2{-}(abs2(p), q): hrpass(abs2(p), ad, a2) -> p: adj2 = a2*(2p)^2 + ad*2 ; hrpass(q, -ad, a2) ; hdrpass -> -a2 * 2p */
        h[0] += a2 * ((2.0 * p) * (2.0 * p)) + ad * 2.0;                               /* (p,p) */
        h[1] += a2 * (-1.0 * -1.0) + (ad * -1.0) * 0.0;                                /* (q,q) */
        h[2] += (a2 * 1.0 * -1.0 + ad * 0.0) * (2.0 * p);                              /* (p,q) -> (i, i-1) */
        /* (p - 1)^2 */
        h[0] += 0.0 * ((2.0 * (p - 1.0)) * (2.0 * (p - 1.0))) + sigma * 2.0;           /* (p,p) */
    }
}
