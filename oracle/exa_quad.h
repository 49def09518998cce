/* exa_quad.h — the higher-precision ARBITER of the test oracle (included by exa_oracle.c; TEST INFRASTRUCTURE, never linked into the
 * product).  `north_star` asks for results "within 1e-10 relative"; component-wise that is unattainable BY ANY DOUBLE-PRECISION
 * EVALUATOR — the reference included — for an entry that is a cancelling sum: a power-balance row of size 5e-7 among terms of size
 * 1e4 carries an absolute error of a few ulp OF THE TERMS, i.e. 1e-10 .. 1e-9 of itself, in the kernel and in the double oracle
 * alike.  To say so with evidence instead of with a floored denominator, the primal of a pattern (`node(i, x, theta)`,
 * src/graph.jl:305-323) is evaluated here a second time in __float128 (libquadmath: 113-bit significands, 1e-34), together with a
 * first-order RUNNING ERROR BOUND of its double-precision evaluation, in units of eps (Higham, Accuracy and Stability of Numerical
 * Algorithms, sec. 3.3): inputs and literals are exact (e = 0); e(a +- b) = e(a) + e(b) + |r|; e(a b) = |b| e(a) + |a| e(b) + |r|;
 * e(a / b) = e(a) / |b| + |a / b^2| e(b) + |r|; e(f(a)) = |f'(a)| e(a) + 2 |r| (a library function is good to an ulp or two), the
 * derivatives from the oracle's own tables.  A test then asserts
 *     |a - q| <= 1e-10 |q|     or     |a - q| <= K eps e          (K: a small constant stated in the test)
 * for the kernel's value a AND for the double oracle's: no double evaluation of the expression can promise more than the second
 * clause, so the exception is explicit, bounded, and tested on both.
 * Covers the 52 + 9 entries of src/functionlist.jl; the SpecialFunctions extension has no quad restatement (quad_ok = 0). */
#include <quadmath.h>

typedef __float128 q128;
typedef struct { int is_int; int64_t i; q128 f; q128 mag; } qval_t;     /* mag: the running error bound e, in units of eps */
static inline qval_t QI(int64_t i) { qval_t v = {1, i, (q128)i, 0.0Q}; return v; }
static inline qval_t QF(q128 f) { qval_t v = {0, 0, f, 0.0Q}; return v; }      /* exact input / literal */
static inline qval_t QR(q128 f, q128 e) { qval_t v = {0, 0, f, e}; return v; }     /* a computed result */
static inline double qd(q128 f) { return (double)f; }
static const q128 QPI = M_PIq;

static q128 qpow_int(q128 x, int64_t n) {     /* as pow_int above: Base.^(::Float64, ::Integer) by repeated multiplication */
    if (n < 0) return 1.0Q / qpow_int(x, -n);
    q128 y = 1.0Q;
    while (n > 0) { if (n & 1) y *= x; x *= x; n >>= 1; }
    return y;
}
static q128 qsign(q128 x) { return x > 0 ? 1.0Q : (x < 0 ? -1.0Q : x); }

static q128 qun_f(int fn, q128 x, int *ok) {
    switch (fn) {
    case EXA_U_PLUS: return x;             case EXA_U_MINUS: return -x;
    case EXA_U_INV: return 1.0Q / x;       case EXA_U_SQRT: return sqrtq(x);
    case EXA_U_CBRT: return cbrtq(x);      case EXA_U_ABS: return fabsq(x);
    case EXA_U_ABS2: return x * x;         case EXA_U_SIGN: return qsign(x);
    case EXA_U_EXP: return expq(x);        case EXA_U_EXP2: return exp2q(x);
    case EXA_U_EXP10: return powq(10.0Q, x); case EXA_U_EXPM1: return expm1q(x);
    case EXA_U_LOG: return logq(x);        case EXA_U_LOG2: return log2q(x);
    case EXA_U_LOG1P: return log1pq(x);    case EXA_U_LOG10: return log10q(x);
    case EXA_U_SIN: return sinq(x);        case EXA_U_COS: return cosq(x);
    case EXA_U_TAN: return tanq(x);        case EXA_U_ASIN: return asinq(x);
    case EXA_U_ACOS: return acosq(x);      case EXA_U_ATAN: return atanq(x);
    case EXA_U_ACOT: return atanq(1.0Q / x);
    case EXA_U_CSC: return 1.0Q / sinq(x); case EXA_U_SEC: return 1.0Q / cosq(x);
    case EXA_U_COT: return 1.0Q / tanq(x); case EXA_U_SINH: return sinhq(x);
    case EXA_U_COSH: return coshq(x);      case EXA_U_TANH: return tanhq(x);
    case EXA_U_ASINH: return asinhq(x);    case EXA_U_ACOSH: return acoshq(x);
    case EXA_U_CSCH: return 1.0Q / sinhq(x); case EXA_U_SECH: return 1.0Q / coshq(x);
    case EXA_U_COTH: return 1.0Q / tanhq(x);
    case EXA_U_SIND: return sinq(x * QPI / 180.0Q);  case EXA_U_COSD: return cosq(x * QPI / 180.0Q);
    case EXA_U_TAND: return tanq(x * QPI / 180.0Q);  case EXA_U_CSCD: return 1.0Q / sinq(x * QPI / 180.0Q);
    case EXA_U_SECD: return 1.0Q / cosq(x * QPI / 180.0Q); case EXA_U_COTD: return 1.0Q / tanq(x * QPI / 180.0Q);
    case EXA_U_ATAND: return atanq(x) * 180.0Q / QPI;  case EXA_U_ACOTD: return atanq(1.0Q / x) * 180.0Q / QPI;
    case EXA_U_SINPI: return sinq(QPI * x); case EXA_U_COSPI: return cosq(QPI * x);
    case EXA_U_SINC: return x == 0 ? 1.0Q : sinq(QPI * x) / (QPI * x);
    case EXA_U_DEG2RAD: return x * QPI / 180.0Q;   case EXA_U_RAD2DEG: return x * 180.0Q / QPI;
    case EXA_U_SIGNBIT: return signbitq(x) ? 1.0Q : 0.0Q;
    case EXA_U_FLOOR: return floorq(x);    case EXA_U_CEIL: return ceilq(x);
    case EXA_U_ATANH: return atanhq(x);    case EXA_U_ACOTH: return atanhq(1.0Q / x);
    }
    *ok = 0;
    return 0.0Q;
}

static qval_t qev(const pattern *p, int k, int64_t I, const double *x, const double *theta, int *ok) {
    const exa_node_t *nd = &p->nodes[k];
    switch (nd->op) {
    case EXA_OP_CONST_F: return QF((q128)nd->fval);
    case EXA_OP_CONST_I: return QI(nd->ival);
    case EXA_OP_NULLV: return QF((q128)nd->fval);
    case EXA_OP_DATA: { val_t v = col_val(p, nd->a, I); return v.is_int ? QI(v.i) : QF((q128)v.f); }
    case EXA_OP_PAR: { qval_t i = qev(p, nd->a, I, x, theta, ok); return QF((q128)theta[i.i - 1]); }
    case EXA_OP_VAR: { qval_t i = qev(p, nd->a, I, x, theta, ok); return QF((q128)x[i.i - 1]); }
    case EXA_OP_UN: {
        qval_t a = qev(p, nd->a, I, x, theta, ok);
        if (a.is_int) {
            if (nd->fn == EXA_U_PLUS) return a;
            if (nd->fn == EXA_U_MINUS) return QI(-a.i);
            if (nd->fn == EXA_U_ABS) return QI(a.i < 0 ? -a.i : a.i);
            if (nd->fn == EXA_U_ABS2) return QI(a.i * a.i);
        }
        const q128 r = qun_f(nd->fn, a.f, ok);
        if (nd->fn == EXA_U_PLUS) return QR(r, a.mag);
        if (nd->fn == EXA_U_MINUS || nd->fn == EXA_U_ABS) return QR(r, a.mag);       /* exact operations */
        return QR(r, fabsq((q128)un_df(nd->fn, qd(a.f))) * a.mag + 2.0Q * fabsq(r));
    }
    case EXA_OP_BIN: {
        qval_t a = qev(p, nd->a, I, x, theta, ok), b = qev(p, nd->b, I, x, theta, ok);
        if (a.is_int && b.is_int) {
            switch (nd->fn) {
            case EXA_B_ADD: return QI(a.i + b.i);
            case EXA_B_SUB: return QI(a.i - b.i);
            case EXA_B_MUL: return QI(a.i * b.i);
            case EXA_B_POW: if (b.i >= 0) return QI(ipow_int(a.i, b.i)); break;
            case EXA_B_MAX: return QI(a.i > b.i ? a.i : b.i);
            case EXA_B_MIN: return QI(a.i < b.i ? a.i : b.i);
            default: break;
            }
        }
        const q128 x1 = a.f, x2 = b.f;
        q128 r;
        switch (nd->fn) {
        case EXA_B_ADD: r = x1 + x2; return QR(r, a.mag + b.mag + fabsq(r));
        case EXA_B_SUB: r = x1 - x2; return QR(r, a.mag + b.mag + fabsq(r));
        case EXA_B_MUL: r = x1 * x2; return QR(r, fabsq(x2) * a.mag + fabsq(x1) * b.mag + fabsq(r));
        case EXA_B_DIV: r = x1 / x2; return QR(r, a.mag / fabsq(x2) + fabsq(x1 / (x2 * x2)) * b.mag + fabsq(r));
        case EXA_B_MAX: return (x1 > x2 || x1 != x1) ? QR(x1, a.mag) : QR(x2, b.mag);
        case EXA_B_MIN: return (x1 < x2 || x1 != x1) ? QR(x1, a.mag) : QR(x2, b.mag);
        case EXA_B_POW: case EXA_B_ATAN2: case EXA_B_HYPOT: {
            r = nd->fn == EXA_B_POW ? (b.is_int ? qpow_int(x1, b.i) : powq(x1, x2)) : nd->fn == EXA_B_ATAN2 ? atan2q(x1, x2) : hypotq(x1, x2);
            const val_t va = a.is_int ? VI(a.i) : VF(qd(x1)), vb = b.is_int ? VI(b.i) : VF(qd(x2));
            const double d1 = bin_d1(nd->fn, va, vb), d2 = b.is_int ? 0.0 : bin_d2(nd->fn, va, vb);
            /* an integer power is |n| - 1 multiplications, each with its rounding */
            const q128 own = nd->fn == EXA_B_POW && b.is_int ? (q128)(b.i < 0 ? -b.i + 1 : (b.i > 1 ? b.i - 1 : 1)) : 2.0Q;
            return QR(r, fabsq((q128)d1) * a.mag + (b.mag > 0 ? fabsq((q128)d2) * b.mag : 0.0Q) + own * fabsq(r));
        }
        }
        *ok = 0;
        return QF(0.0Q);
    }
    }
    *ok = 0;
    return QF(0.0Q);
}
