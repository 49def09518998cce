// exa_codegen.cpp — emits one HIP module per model: straight-line, register-resident AD code per pattern.
//
// What it replaces: the KernelAbstractions kernels kerf/kerf2/kerg/kerj/kerh/kerh2 of
// ext/ExaModelsKernelAbstractions.jl:608-684 together with Julia's type-specialisation of the whole
// expression tree into each of them.  Design differences (MI355X-first, not a translation):
//   * one FUSED launch per callback for the whole model: blockIdx -> (pattern, data-point tile) through a
//     cumulative block table, so a 15-pattern ACOPF Hessian is 1 launch instead of 1 memset + 9 kernels;
//   * every COO slot is accumulated in a VGPR in the reference's contribution order and stored ONCE
//     (no zero-fill, no read-modify-write on HBM: KA ext :521,:533 + hessian.jl:580-592 do fill! and `+=`);
//   * the iterator is struct-of-arrays, lane I reads column[I] (coalesced); UnitRange iterators cost no load;
//   * forward sweep, partials and reverse sweep are symbolic here: constants fold, x*1 / x+0 vanish, common
//     sub-expressions (one sincos per argument, exp reused for f=f'=f'') are shared by construction.
// Derivative formulas follow src/functionlist.jl:6-81 (algebraically identical, a few rewritten through the
// already-computed primal to save FP64 divides; parity bar 1e-10 relative).
#include <algorithm>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <sstream>
#include <stdexcept>

#include "exa_internal.hpp"
#include "exa_traverse.hpp"

namespace exa {
namespace {

[[noreturn]] void fail(const std::string &m) { throw BadInput(m); }
int env_int(const char *name, int dflt) { const char *v = getenv(name); return v && *v ? atoi(v) : dflt; }

// ---------------------------------------------------------------------------------------------------
// symbolic values
// ---------------------------------------------------------------------------------------------------
struct Val {
    enum K { LF, LI, SF, SI } k = LF;   // literal float / literal int / SSA float / SSA int
    double f = 0.0;
    int64_t i = 0;
    int id = -1;
    bool is_lit() const { return k == LF || k == LI; }
    bool is_int() const { return k == LI || k == SI; }
    double litv() const { return k == LI ? (double)i : f; }
    bool lit_eq(double v) const { return is_lit() && litv() == v; }
};

std::string fmt_double(double v) {
    if (std::isnan(v)) return "__builtin_nan(\"\")";
    if (std::isinf(v)) return v > 0 ? "__builtin_inf()" : "(-__builtin_inf())";
    char buf[64];
    if (v == std::floor(v) && std::fabs(v) < 1e15) snprintf(buf, sizeof buf, "%.1f", v);
    else snprintf(buf, sizeof buf, "%.17g", v);
    std::string s = buf;
    if (s.find_first_of(".en") == std::string::npos) s += ".0";
    if (v < 0 || (v == 0 && std::signbit(v))) s = "(" + s + ")";
    return s;
}

struct Emitter {
    std::vector<std::string> lines;
    std::map<std::string, Val> memo;
    int next = 0;
    struct Def { int line; std::string name, expr; bool is_int; };
    std::vector<Def> defs;        // the lines that are plain SSA definitions `const T name = expr;` (raw), in order
    std::map<std::string, std::string> expr_of;   // SSA name -> its defining expression text

    static Val litf(double v) { Val r; r.k = Val::LF; r.f = v; return r; }
    static Val liti(int64_t v) { Val r; r.k = Val::LI; r.i = v; r.f = (double)v; return r; }

    std::string s(const Val &v) const {
        switch (v.k) {
        case Val::LF: return fmt_double(v.f);
        case Val::LI: { char b[40]; snprintf(b, sizeof b, v.i < 0 ? "(%" PRId64 "L)" : "%" PRId64 "L", v.i); return b; }
        case Val::SF: return "t" + std::to_string(v.id);
        case Val::SI: return "k" + std::to_string(v.id);
        }
        return "?";
    }
    // text of v in a floating-point context
    std::string sd(const Val &v) const {
        if (v.k == Val::LI) return fmt_double((double)v.i);
        if (v.k == Val::SI) return "(double)" + s(v);
        return s(v);
    }
    Val tod(const Val &v) {
        if (v.k == Val::LI) return litf((double)v.i);
        if (v.k == Val::SI) return raw("(double)" + s(v), false);
        return v;
    }
    // memoised SSA definition of an expression text
    Val raw(const std::string &expr, bool is_int) {
        auto it = memo.find(expr);
        if (it != memo.end()) return it->second;
        Val r;
        r.k = is_int ? Val::SI : Val::SF;
        r.id = next++;
        lines.push_back(std::string(is_int ? "const long k" : "const double t") + std::to_string(r.id) + " = " + expr + ";");
        defs.push_back({(int)lines.size() - 1, std::string(is_int ? "k" : "t") + std::to_string(r.id), expr, is_int});
        expr_of[defs.back().name] = expr;
        memo[expr] = r;
        return r;
    }
    Val neg(const Val &a) {
        if (a.k == Val::LF) return litf(-a.f);
        if (a.k == Val::LI) return liti(-a.i);
        return raw("-" + s(a), a.is_int());
    }
    Val bin(char op, Val a, Val b) {
        const bool ii = a.is_int() && b.is_int() && op != '/';
        if (a.is_lit() && b.is_lit()) {
            if (ii) {
                switch (op) { case '+': return liti(a.i + b.i); case '-': return liti(a.i - b.i); case '*': return liti(a.i * b.i); }
            }
            const double x = a.litv(), y = b.litv();
            switch (op) { case '+': return litf(x + y); case '-': return litf(x - y); case '*': return litf(x * y); case '/': return litf(x / y); }
        }
        // identities on exact literals.  1*z, z/1 and z-0 are exact for every z; 0*z -> 0, 0/z -> 0 (differ from IEEE
        // when z is Inf/NaN: the reference, which multiplies at run time, has NaN there) and z+0 -> z (differs for
        // z = -0.0) are dropped under EXAHIP_STRICT_IEEE=1 — the reference's special values entry for entry, at the
        // price of the multiplications by literal zeros the reverse sweep is full of (DESIGN.md §4 has the numbers).
        const bool strict = env_int("EXAHIP_STRICT_IEEE", 0) != 0;
        if (strict && !ii) {
            if (op == '*' && (a.lit_eq(1) || b.lit_eq(1))) return a.lit_eq(1) ? tod(b) : tod(a);
            if (op == '/' && b.lit_eq(1)) return tod(a);
            if (op == '-' && b.lit_eq(0)) return tod(a);
            return raw(sd(a) + " " + op + " " + sd(b), false);
        }
        switch (op) {
        case '+': if (a.lit_eq(0)) return ii ? b : tod(b); if (b.lit_eq(0)) return ii ? a : tod(a); break;
        case '-': if (b.lit_eq(0)) return ii ? a : tod(a); if (a.lit_eq(0)) return neg(ii ? b : tod(b)); break;
        case '*':
            if (a.lit_eq(0) || b.lit_eq(0)) return ii ? liti(0) : litf(0.0);
            if (a.lit_eq(1)) return ii ? b : tod(b);
            if (b.lit_eq(1)) return ii ? a : tod(a);
            if (a.lit_eq(-1)) return neg(ii ? b : tod(b));
            if (b.lit_eq(-1)) return neg(ii ? a : tod(a));
            break;
        case '/':
            if (b.lit_eq(1)) return tod(a);
            if (a.lit_eq(0)) return litf(0.0);
            break;
        }
        if (ii) return raw(s(a) + " " + op + " " + s(b), true);
        return raw(sd(a) + " " + op + " " + sd(b), false);
    }
    Val add(Val a, Val b) { return bin('+', a, b); }
    Val sub(Val a, Val b) { return bin('-', a, b); }
    Val mul(Val a, Val b) { return bin('*', a, b); }
    Val div(Val a, Val b) { return bin('/', a, b); }
    Val sq(Val a) { return mul(a, a); }
    // template call: $1 $2 $3 replaced by operand texts (floating context)
    Val call(const std::string &tmpl, std::initializer_list<Val> args) {
        std::string out;
        std::vector<Val> av(args);
        for (size_t i = 0; i < tmpl.size(); i++) {
            if (tmpl[i] == '$' && i + 1 < tmpl.size() && tmpl[i + 1] >= '1' && tmpl[i + 1] <= '9') {
                size_t k = (size_t)(tmpl[i + 1] - '1');
                if (k >= av.size()) fail("bad template " + tmpl);
                out += sd(av[k]);
                i++;
            } else out += tmpl[i];
        }
        return raw(out, false);
    }
};

// ---------------------------------------------------------------------------------------------------
// function rules: (x, y, h) of a univariate; $1 = argument, $2 = primal f, $3 = first derivative
// ---------------------------------------------------------------------------------------------------
struct UnSpec { const char *f, *df, *ddf; };
// A leading '=' marks an exact literal (lets the reverse sweep fold it).
const double kPi = 3.14159265358979323846;
const double kD2R = kPi / 180.0, kR2D = 180.0 / kPi;

// The generator keeps per-module state in file-level variables (g_handover, g_lit_idx, g_lds_need, ...): ONE lock for
// generate_module and generate_window_module (models may be built and compressed from several host threads).
std::mutex g_gen_mu;

const UnSpec *un_spec(int fn) {
    static UnSpec T[EXA_U_COUNT];
    static bool init = false;
    if (!init) {
        init = true;
        T[EXA_U_PLUS] = {"$1", "=1", "=0"};
        T[EXA_U_MINUS] = {"-$1", "=-1", "=0"};
        T[EXA_U_INV] = {"1.0 / $1", "-($2 * $2)", "2.0 * $2 * $2 * $2"};
        T[EXA_U_SQRT] = {"sqrt($1)", "0.5 / $2", "-0.25 / ($2 * $2 * $2)"};
        T[EXA_U_CBRT] = {"cbrt($1)", "1.0 / (3.0 * $2 * $2)", "-2.0 / (9.0 * $2 * $2 * $2 * $2 * $2)"};
        T[EXA_U_ABS] = {"fabs($1)", "(__builtin_signbit($1) ? -1.0 : 1.0)", "=0"};
        T[EXA_U_ABS2] = {"$1 * $1", "2.0 * $1", "=2"};
        T[EXA_U_SIGN] = {"exa_sign($1)", "=0", "=0"};
        T[EXA_U_EXP] = {"exp($1)", "$2", "$2"};
        T[EXA_U_EXP2] = {"exp2($1)", "EXA_LOG2 * $2", "EXA_LOG2 * EXA_LOG2 * $2"};
        T[EXA_U_EXP10] = {"exp10($1)", "EXA_LOG10 * $2", "EXA_LOG10 * EXA_LOG10 * $2"};
        T[EXA_U_EXPM1] = {"expm1($1)", "exp($1)", "$3"};
        T[EXA_U_LOG] = {"log($1)", "1.0 / $1", "-($3 * $3)"};
        T[EXA_U_LOG2] = {"log2($1)", "1.0 / (EXA_LOG2 * $1)", "-$3 / $1"};
        T[EXA_U_LOG1P] = {"log1p($1)", "1.0 / (1.0 + $1)", "-($3 * $3)"};
        T[EXA_U_LOG10] = {"log10($1)", "1.0 / (EXA_LOG10 * $1)", "-$3 / $1"};
        T[EXA_U_SIN] = {nullptr, nullptr, nullptr};   // handled through sincos
        T[EXA_U_COS] = {nullptr, nullptr, nullptr};
        T[EXA_U_TAN] = {"tan($1)", "1.0 + $2 * $2", "2.0 * $3 * $2"};
        T[EXA_U_ASIN] = {"asin($1)", "1.0 / sqrt(1.0 - $1 * $1)", "$1 * $3 / (1.0 - $1 * $1)"};
        T[EXA_U_ACOS] = {"acos($1)", "-1.0 / sqrt(1.0 - $1 * $1)", "$1 * $3 / (1.0 - $1 * $1)"};
        T[EXA_U_ATAN] = {"atan($1)", "1.0 / (1.0 + $1 * $1)", "-2.0 * $1 * $3 * $3"};
        T[EXA_U_ACOT] = {"atan(1.0 / $1)", "-1.0 / (1.0 + $1 * $1)", "2.0 * $1 * $3 * $3"};
        T[EXA_U_CSC] = {"1.0 / sin($1)", "-$2 / tan($1)", "(1.0 + 2.0 * exa_sq(1.0 / tan($1))) * $2"};
        T[EXA_U_SEC] = {"1.0 / cos($1)", "$2 * tan($1)", "$2 * $2 * $2 + $2 * exa_sq(tan($1))"};
        T[EXA_U_COT] = {"1.0 / tan($1)", "-1.0 - $2 * $2", "-2.0 * $2 * $3"};
        T[EXA_U_SINH] = {"sinh($1)", "cosh($1)", "$2"};
        T[EXA_U_COSH] = {"cosh($1)", "sinh($1)", "$2"};
        T[EXA_U_TANH] = {"tanh($1)", "1.0 - $2 * $2", "-2.0 * $2 * $3"};
        T[EXA_U_ASINH] = {"asinh($1)", "1.0 / sqrt(1.0 + $1 * $1)", "-$1 * $3 / (1.0 + $1 * $1)"};
        T[EXA_U_ACOSH] = {"acosh($1)", "1.0 / sqrt($1 * $1 - 1.0)", "-$1 * $3 / ($1 * $1 - 1.0)"};
        T[EXA_U_CSCH] = {"1.0 / sinh($1)", "-$2 / tanh($1)", "$2 * $2 * $2 + $2 * exa_sq(1.0 / tanh($1))"};
        T[EXA_U_SECH] = {"1.0 / cosh($1)", "-tanh($1) * $2", "(2.0 * exa_sq(tanh($1)) - 1.0) * $2"};
        T[EXA_U_COTH] = {"1.0 / tanh($1)", "-exa_sq(1.0 / sinh($1))", "-2.0 * $3 * $2"};
        T[EXA_U_SIND] = {"exa_sind($1)", "EXA_D2R * exa_cosd($1)", "-(EXA_D2R * EXA_D2R) * $2"};
        T[EXA_U_COSD] = {"exa_cosd($1)", "-EXA_D2R * exa_sind($1)", "-(EXA_D2R * EXA_D2R) * $2"};
        T[EXA_U_TAND] = {"exa_tand($1)", "EXA_D2R * (1.0 + $2 * $2)", "2.0 * EXA_D2R * $2 * $3"};
        T[EXA_U_CSCD] = {"1.0 / exa_sind($1)", "-EXA_D2R * $2 / exa_tand($1)", "(EXA_D2R * EXA_D2R) * $2 * (1.0 + 2.0 * exa_sq(1.0 / exa_tand($1)))"};
        T[EXA_U_SECD] = {"1.0 / exa_cosd($1)", "EXA_D2R * exa_tand($1) * $2", "(EXA_D2R * EXA_D2R) * $2 * (1.0 + 2.0 * exa_sq(exa_tand($1)))"};
        T[EXA_U_COTD] = {"1.0 / exa_tand($1)", "-EXA_D2R * (1.0 + $2 * $2)", "-2.0 * EXA_D2R * $2 * $3"};
        T[EXA_U_ATAND] = {"EXA_R2D * atan($1)", "1.0 / (EXA_D2R * (1.0 + $1 * $1))", "-2.0 * EXA_D2R * $1 * $3 * $3"};
        T[EXA_U_ACOTD] = {"EXA_R2D * atan(1.0 / $1)", "-1.0 / (EXA_D2R * (1.0 + $1 * $1))", "2.0 * EXA_D2R * $1 * $3 * $3"};
        T[EXA_U_SINPI] = {"sinpi($1)", "EXA_PI * cospi($1)", "-(EXA_PI * EXA_PI) * $2"};
        T[EXA_U_COSPI] = {"cospi($1)", "-EXA_PI * sinpi($1)", "-(EXA_PI * EXA_PI) * $2"};
        T[EXA_U_SINC] = {"exa_sinc($1)",
                         "(-sinpi($1) + EXA_PI * $1 * cospi($1)) / (EXA_PI * ($1 * $1))",
                         "((2.0 * EXA_PI * EXA_PI) * sinpi($1) - (2.0 * EXA_PI * EXA_PI * EXA_PI) * $1 * cospi($1) - "
                         "(EXA_PI * EXA_PI * EXA_PI * EXA_PI) * ($1 * $1) * sinpi($1)) / ((EXA_PI * EXA_PI * EXA_PI) * ($1 * $1 * $1))"};
        T[EXA_U_DEG2RAD] = {"EXA_D2R * $1", "=D2R", "=0"};
        T[EXA_U_RAD2DEG] = {"EXA_R2D * $1", "=R2D", "=0"};
        T[EXA_U_SIGNBIT] = {"(__builtin_signbit($1) ? 1.0 : 0.0)", "=0", "=0"};
        T[EXA_U_FLOOR] = {"floor($1)", "=0", "=0"};
        T[EXA_U_CEIL] = {"ceil($1)", "=0", "=0"};
        T[EXA_U_ATANH] = {"atanh($1)", "(fabs($1) > 1.0 ? __builtin_nan(\"\") : 1.0 / (1.0 - $1 * $1))",
                          "(fabs($1) > 1.0 ? __builtin_nan(\"\") : 2.0 * $1 * exa_sq(1.0 / (1.0 - $1 * $1)))"};
        T[EXA_U_ACOTH] = {"atanh(1.0 / $1)", "(fabs($1) < 1.0 ? __builtin_nan(\"\") : 1.0 / (1.0 - $1 * $1))",
                          "(fabs($1) < 1.0 ? __builtin_nan(\"\") : 2.0 * $1 * exa_sq(1.0 / (1.0 - $1 * $1)))"};
    }
    return &T[fn];
}

double host_un(int fn, double x) {   // folding of literal arguments, primal only
    switch (fn) {
    case EXA_U_PLUS: return x; case EXA_U_MINUS: return -x; case EXA_U_ABS2: return x * x; case EXA_U_ABS: return std::fabs(x);
    case EXA_U_INV: return 1.0 / x; case EXA_U_SQRT: return std::sqrt(x);
    default: return NAN;
    }
}
bool host_un_ok(int fn) {
    return fn == EXA_U_PLUS || fn == EXA_U_MINUS || fn == EXA_U_ABS2 || fn == EXA_U_ABS || fn == EXA_U_INV || fn == EXA_U_SQRT;
}

struct Triple { Val x, y, h; };

Val lit_or_tmpl(Emitter &e, const char *spec, Val u, Val f, Val d) {
    if (spec[0] == '=') {
        std::string v = spec + 1;
        if (v == "D2R") return Emitter::litf(kD2R);
        if (v == "R2D") return Emitter::litf(kR2D);
        return Emitter::litf(atof(v.c_str()));
    }
    return e.call(spec, {u, f, d});
}

Triple un_rule(Emitter &e, int fn, Val u, int order) {
    Triple r;
    u = e.tod(u);
    if (u.is_lit() && host_un_ok(fn) && order == 0) { r.x = Emitter::litf(host_un(fn, u.f)); return r; }
    if (fn == EXA_U_SIN || fn == EXA_U_COS) {
        if (order == 0 && !env_int("EXAHIP_FAST_TRIG", 1)) {
            r.x = e.call(fn == EXA_U_SIN ? "sin($1)" : "cos($1)", {u});
            return r;
        }
        // (value-only contexts too: exa_sin / exa_cos each run the whole sincos, so a pattern — or a fused group — that
        // needs both of one argument pays once)
        // one sincos per argument serves value and both derivatives (functionlist.jl:22-23)
        const std::string key = "sincos|" + e.s(u);
        Val sv, cv;
        auto it = e.memo.find(key);
        if (it == e.memo.end() && env_int("EXAHIP_SYM_TRIG", 1) && !env_int("EXAHIP_WHATIF_NOTRIG", 0)) {
            // u = a - b where sincos(b - a) is already there (the two ends of an ACOPF branch: va_f - va_t and va_t - va_f):
            // a - b == -(b - a) exactly and exa_sincos / ocml sincos are exactly odd / even, so sin u = 0.0 - sin(b - a)
            // (written as a subtraction from +0.0: bit-identical to the direct evaluation for a == b too, where
            // -(+0.0) would be -0.0) and cos u = cos(b - a).
            auto ex = e.expr_of.find(e.s(u));
            if (ex != e.expr_of.end()) {
                const std::string &t = ex->second;
                const size_t at = t.find(" - ");
                if (at != std::string::npos && t.find(' ', at + 3) == std::string::npos && t.find(' ') == at) {
                    auto rev = e.memo.find(t.substr(at + 3) + " - " + t.substr(0, at));
                    if (rev != e.memo.end()) {
                        const std::string rkey = "sincos|" + e.s(rev->second);
                        auto rs = e.memo.find(rkey);
                        if (rs != e.memo.end()) {
                            sv = e.raw("0.0 - " + e.s(rs->second), false);
                            cv = e.memo[rkey + "|c"];
                            e.memo[key] = sv;
                            e.memo[key + "|c"] = cv;
                            it = e.memo.find(key);
                        }
                    }
                }
            }
        }
        if (it == e.memo.end()) {
            sv.k = Val::SF; sv.id = e.next++;
            cv.k = Val::SF; cv.id = e.next++;
            if (env_int("EXAHIP_WHATIF_NOTRIG", 0))   // diagnostic only (wrong numbers): how much of the kernel is FP64 transcendental issue?
                e.lines.push_back("const double t" + std::to_string(sv.id) + " = " + e.s(u) + ", t" + std::to_string(cv.id) + " = 1.0 - " + e.s(u) + ";");
            else
            e.lines.push_back("double t" + std::to_string(sv.id) + ", t" + std::to_string(cv.id) + "; " +
                              (env_int("EXAHIP_FAST_TRIG", 1) ? "exa_sincos(" : "sincos(") + e.s(u) + ", &t" +
                              std::to_string(sv.id) + ", &t" + std::to_string(cv.id) + ");");
            e.memo[key] = sv;
            e.memo[key + "|c"] = cv;
        } else { sv = it->second; cv = e.memo[key + "|c"]; }
        if (fn == EXA_U_SIN) { r.x = sv; r.y = cv; r.h = e.neg(sv); }
        else { r.x = cv; r.y = e.neg(sv); r.h = e.neg(cv); }
        return r;
    }
    if (fn == EXA_U_MINUS) { r.x = e.neg(u); r.y = Emitter::litf(-1); r.h = Emitter::litf(0); return r; }
    if (fn == EXA_U_PLUS) { r.x = u; r.y = Emitter::litf(1); r.h = Emitter::litf(0); return r; }
    if (fn == EXA_U_ABS2) { r.x = e.mul(u, u); r.y = e.mul(Emitter::litf(2), u); r.h = Emitter::litf(2); return r; }
    const UnSpec *sp = un_spec(fn);
    if (!sp->f) fail("univariate function without rule");
    r.x = e.call(sp->f, {u});
    if (order >= 1) r.y = lit_or_tmpl(e, sp->df, u, r.x, r.x);
    if (order >= 2) r.h = lit_or_tmpl(e, sp->ddf, u, r.x, r.y);
    return r;
}

// x^n for a literal integer n by repeated multiplication (Base.^(::Float64, ::Integer); n==3 -> x*x*x)
Val powi_lit(Emitter &e, Val x, int64_t n) {
    x = e.tod(x);
    if (n == 0) return Emitter::litf(1.0);
    if (x.is_lit()) return Emitter::litf(std::pow(x.f, (double)n));
    if (n < 0) { Val r = e.div(Emitter::litf(1.0), x); return powi_lit(e, r, -n); }
    if (n == 1) return x;
    if (n == 2) return e.mul(x, x);
    if (n == 3) return e.mul(e.mul(x, x), x);
    Val y; bool has = false;
    Val b = x;
    while (n > 1) {
        if (n & 1) { y = has ? e.mul(y, b) : b; has = true; }
        b = e.mul(b, b);
        n >>= 1;
    }
    return has ? e.mul(b, y) : b;
}

// x1 ^ x2 where the exponent is a typed value
Val pow_any(Emitter &e, Val x1, Val x2) {
    if (x2.k == Val::LI) return powi_lit(e, x1, x2.i);
    if (x2.k == Val::SI) return e.raw("exa_powi(" + e.sd(x1) + ", " + e.s(x2) + ")", false);
    x1 = e.tod(x1);
    if (x1.is_lit() && x2.is_lit()) return Emitter::litf(std::pow(x1.f, x2.f));
    return e.call("pow($1, $2)", {x1, x2});
}
Val add_i(Emitter &e, Val v, int64_t k) { return e.add(v, v.is_int() ? Emitter::liti(k) : Emitter::litf((double)k)); }

struct Six { Val x, y1, y2, h11, h12, h22; };

// full bivariate rule (both operands differentiable), functionlist.jl:71-81
Six bin_rule(Emitter &e, int fn, Val x1, Val x2, int order) {
    Six r;
    const Val Z = Emitter::litf(0), O = Emitter::litf(1);
    r.h11 = r.h12 = r.h22 = Z;
    switch (fn) {
    case EXA_B_ADD: r.x = e.add(x1, x2); r.y1 = O; r.y2 = O; return r;
    case EXA_B_SUB: r.x = e.sub(x1, x2); r.y1 = O; r.y2 = Emitter::litf(-1); return r;
    case EXA_B_MUL: r.x = e.mul(x1, x2); r.y1 = e.tod(x2); r.y2 = e.tod(x1); r.h12 = O; return r;
    case EXA_B_DIV: {
        r.x = e.div(x1, x2);
        if (order >= 1 && env_int("EXAHIP_STRICT_IEEE", 0)) {
            // the table's own forms (functionlist.jl:75): 1/x2, -x1/x2^2, -1/x2^2, 2x1/x2^3 — three more divisions; they
            // differ from the quotient forms below only where x2^2 or x2^3 over/underflows (|x2| > 1.3e154, < 1e-103)
            r.y1 = e.div(O, x2);
            r.y2 = e.div(e.neg(x1), e.sq(x2));
            if (order >= 2) {
                r.h12 = e.div(Emitter::litf(-1), e.sq(x2));
                r.h22 = e.div(e.mul(Emitter::litf(2), x1), e.mul(e.sq(x2), x2));
            }
            return r;
        }
        if (order >= 1) {
            Val inv = e.div(O, x2);
            r.y1 = inv;
            r.y2 = e.neg(e.mul(r.x, inv));                       // -x1/x2^2
            if (order >= 2) {
                r.h12 = e.neg(e.mul(inv, inv));                  // -1/x2^2
                r.h22 = e.mul(Emitter::litf(-2), e.mul(r.y2, inv));   // 2 x1 / x2^3
            }
        }
        return r;
    }
    case EXA_B_POW: {
        r.x = pow_any(e, x1, x2);
        if (order >= 1) {
            Val pm1 = pow_any(e, x1, add_i(e, x2, -1));
            Val lg = e.call("log($1)", {x1});
            r.y1 = e.mul(x2, pm1);
            r.y2 = e.mul(lg, r.x);
            if (order >= 2) {
                r.h11 = e.mul(e.mul(add_i(e, x2, -1), x2), pow_any(e, x1, add_i(e, x2, -2)));
                r.h12 = e.add(pm1, e.mul(e.mul(x2, pm1), lg));
                r.h22 = e.mul(e.mul(lg, lg), r.x);
            }
        }
        return r;
    }
    case EXA_B_ATAN2: {
        r.x = e.call("atan2($1, $2)", {x1, x2});
        if (order >= 1) {
            Val d = e.add(e.sq(x1), e.sq(x2));
            r.y1 = e.div(x2, d);
            r.y2 = e.div(e.neg(x1), d);
            if (order >= 2) {
                Val d2 = e.sq(d);
                r.h11 = e.div(e.mul(e.mul(Emitter::litf(-2), x1), x2), d2);
                r.h12 = e.div(e.sub(e.sq(x1), e.sq(x2)), d2);     // x1^4 + 2x1^2x2^2 + x2^4 == (x1^2+x2^2)^2
                r.h22 = e.div(e.mul(e.mul(Emitter::litf(2), x1), x2), d2);
            }
        }
        return r;
    }
    case EXA_B_HYPOT: {
        r.x = e.call("hypot($1, $2)", {x1, x2});
        if (order >= 1) {
            r.y1 = e.div(x1, r.x);
            r.y2 = e.div(x2, r.x);
            if (order >= 2) {
                Val h3 = e.mul(e.sq(r.x), r.x);
                r.h11 = e.div(e.sub(e.sq(r.x), e.sq(x1)), h3);
                r.h12 = e.div(e.neg(e.mul(x1, x2)), h3);
                r.h22 = e.div(e.sub(e.sq(r.x), e.sq(x2)), h3);
            }
        }
        return r;
    }
    case EXA_B_MAX:
        r.x = e.call("(($1 > $2 || $1 != $1) ? $1 : $2)", {x1, x2});
        r.y1 = e.call("($1 > $2 ? 1.0 : 0.0)", {x1, x2});
        r.y2 = e.call("($1 > $2 ? 0.0 : 1.0)", {x1, x2});
        return r;
    case EXA_B_MIN:
        r.x = e.call("(($1 < $2 || $1 != $1) ? $1 : $2)", {x1, x2});
        r.y1 = e.call("($1 < $2 ? 1.0 : 0.0)", {x1, x2});
        r.y2 = e.call("($1 < $2 ? 0.0 : 1.0)", {x1, x2});
        return r;
    }
    fail("unknown bivariate function");
}

// one operand constant: SecondFixed (constant is 2nd: uses d1, d11) / FirstFixed (constant is 1st: d2, d22)
Triple fixed_rule(Emitter &e, int fn, int fixed, Val v, Val c, int order) {
    Triple r;
    const Val Z = Emitter::litf(0), O = Emitter::litf(1);
    const bool second = fixed == FX_SECOND;   // v OP c
    switch (fn) {
    case EXA_B_ADD: r.x = second ? e.add(v, c) : e.add(c, v); r.y = O; r.h = Z; return r;
    case EXA_B_SUB:
        r.x = second ? e.sub(v, c) : e.sub(c, v);
        r.y = second ? O : Emitter::litf(-1); r.h = Z; return r;
    case EXA_B_MUL: r.x = second ? e.mul(v, c) : e.mul(c, v); r.y = e.tod(c); r.h = Z; return r;
    case EXA_B_DIV:
        if (second) {     // v / c
            r.x = e.div(v, c);
            r.y = e.div(O, c); r.h = Z;
        } else {          // c / v : d2 = -c/v^2, d22 = 2c/v^3
            r.x = e.div(c, v);
            if (order >= 1) {
                Val inv = e.div(O, v);
                r.y = e.neg(e.mul(r.x, inv));
                if (order >= 2) r.h = e.mul(Emitter::litf(-2), e.mul(r.y, inv));
            }
        }
        return r;
    case EXA_B_POW:
        if (second) {     // v ^ c
            r.x = pow_any(e, v, c);
            if (order >= 1) r.y = e.mul(c, pow_any(e, v, add_i(e, c, -1)));
            if (order >= 2) r.h = e.mul(e.mul(add_i(e, c, -1), c), pow_any(e, v, add_i(e, c, -2)));
        } else {          // c ^ v
            r.x = pow_any(e, c, e.tod(v));
            if (order >= 1) { Val lg = e.call("log($1)", {c}); r.y = e.mul(lg, r.x); if (order >= 2) r.h = e.mul(e.mul(lg, lg), r.x); }
        }
        return r;
    default: {
        Six s = second ? bin_rule(e, fn, v, e.tod(c), order) : bin_rule(e, fn, e.tod(c), v, order);
        r.x = s.x;
        r.y = second ? s.y1 : s.y2;
        r.h = second ? s.h11 : s.h22;
        return r;
    }
    }
}

// ---------------------------------------------------------------------------------------------------
// per-pattern body generator
// ---------------------------------------------------------------------------------------------------
struct FV { Val x, y1, y2, h11, h12, h22, vidx; };

struct Body {
    const Model &m;
    const Pattern &p;
    int pi;
    const ParamLayout &L;
    Emitter own_;
    Emitter &e;                 // own_, or the emitter shared by the patterns of a fused group (one memo: common loads and
                                // common subexpressions of co-indexed patterns are emitted once)
    std::vector<FV> fv;
    std::map<int, Val> cmemo;   // IR node -> value of constant subtree

    Body(const Model &mm, int pidx, const ParamLayout &ll, Emitter *shared = nullptr)
        : m(mm), p(mm.pats[pidx]), pi(pidx), L(ll), e(shared ? *shared : own_) { fv.resize(p.ad.size()); }

    std::string P(int w) const { return "P[" + std::to_string(w) + "]"; }

    Val column(int c) {
        const Column &col = p.cols[c];
        const int w = L.pat[pi].col[c];
        if (col.type == EXA_COL_RANGE) {
            if (col.step == 1) return e.raw(P(w) + " + I", true);
            return e.raw(P(w) + " + " + std::to_string(col.step) + "L * I", true);
        }
        if (col.type == EXA_COL_I64) return e.raw("((const long*)" + P(w) + ")[I]", true);
        return e.raw("((const double*)" + P(w) + ")[I]", false);
    }

    // value of a Real (non-differentiable) subtree: primal evaluation, Int kept apart from Float64
    Val cval(int k) {
        auto it = cmemo.find(k);
        if (it != cmemo.end()) return it->second;
        const exa_node_t &nd = p.nodes[k];
        Val r;
        switch (nd.op) {
        case EXA_OP_CONST_F: r = Emitter::litf(nd.fval); break;
        case EXA_OP_CONST_I: r = Emitter::liti(nd.ival); break;
        case EXA_OP_NULLV: r = Emitter::litf(nd.fval); break;
        case EXA_OP_DATA: r = column(nd.a); break;
        case EXA_OP_PAR: {
            Val i = cval(nd.a);
            if (!i.is_int()) fail("parameter index expression is not integer-typed");
            r = e.raw("th[" + e.s(e.sub(i, Emitter::liti(1))) + "]", false);
            break;
        }
        case EXA_OP_VAR: r = var_load(cval(nd.a)); break;   // primal-only contexts (obj/cons)
        case EXA_OP_UN: {
            Val a = cval(nd.a);
            if (a.is_int() && (nd.fn == EXA_U_PLUS || nd.fn == EXA_U_MINUS || nd.fn == EXA_U_ABS || nd.fn == EXA_U_ABS2)) {
                if (nd.fn == EXA_U_PLUS) r = a;
                else if (nd.fn == EXA_U_MINUS) r = e.neg(a);
                else if (nd.fn == EXA_U_ABS2) r = e.mul(a, a);
                else r = a.is_lit() ? Emitter::liti(a.i < 0 ? -a.i : a.i) : e.raw("(" + e.s(a) + " < 0 ? -" + e.s(a) + " : " + e.s(a) + ")", true);
            } else r = un_rule(e, nd.fn, a, 0).x;
            break;
        }
        case EXA_OP_BIN: {
            Val a = cval(nd.a), b = cval(nd.b);
            if (a.is_int() && b.is_int() && (nd.fn == EXA_B_ADD || nd.fn == EXA_B_SUB || nd.fn == EXA_B_MUL)) {
                r = e.bin(nd.fn == EXA_B_ADD ? '+' : nd.fn == EXA_B_SUB ? '-' : '*', a, b);
            } else if (a.is_int() && b.is_int() && (nd.fn == EXA_B_MAX || nd.fn == EXA_B_MIN)) {
                const char *op = nd.fn == EXA_B_MAX ? ">" : "<";
                r = e.raw("(" + e.s(a) + " " + op + " " + e.s(b) + " ? " + e.s(a) + " : " + e.s(b) + ")", true);
            } else if (nd.fn == EXA_B_POW) {
                r = pow_any(e, a, b);
            } else {
                r = bin_rule(e, nd.fn, e.tod(a), e.tod(b), 0).x;
            }
            break;
        }
        default: fail("bad opcode");
        }
        cmemo[k] = r;
        return r;
    }

    Val var_load(Val idx) {
        if (!idx.is_int()) fail("variable index expression is not integer-typed");
        return e.raw("x[" + e.s(e.sub(idx, Emitter::liti(1))) + "]", false);
    }

    // forward sweep over the AD tree (register.jl:65-68, 209-266); `structure` => indices only
    void forward(int n, int order, bool structure) {
        const ADNode &t = p.ad[n];
        FV &v = fv[n];
        switch (t.kind) {
        case AD_NULL: v.x = Emitter::litf(p.nodes[t.ir].fval); return;
        case AD_CONST: if (!structure) v.x = cval(t.ir); return;
        case AD_VAR:
            v.vidx = cval(t.ir);
            if (!structure) v.x = var_load(v.vidx);
            return;
        case AD_UN: {
            forward(t.l, order, structure);
            if (structure) return;
            Triple r = (t.fixed == FX_NONE) ? un_rule(e, t.fn, fv[t.l].x, order) : fixed_rule(e, t.fn, t.fixed, fv[t.l].x, cval(t.cir), order);
            v.x = r.x; v.y1 = r.y; v.h11 = r.h;
            return;
        }
        case AD_BIN: {
            forward(t.l, order, structure);
            forward(t.r, order, structure);
            if (structure) return;
            Six r = bin_rule(e, t.fn, fv[t.l].x, fv[t.r].x, order);
            v.x = r.x; v.y1 = r.y1; v.y2 = r.y2; v.h11 = r.h11; v.h12 = r.h12; v.h22 = r.h22;
            return;
        }
        }
    }

    // 0-based row of this data point: offset0 (nlp.jl:1980-2001)
    std::string row0() {
        const int w = L.pat[pi].o0;
        if (p.kind == EXA_PAT_CONAUG) {
            Val t = cval(p.target);
            return P(w) + " + " + e.s(e.sub(t, Emitter::liti(1)));
        }
        return P(w) + " + I";
    }
};

// Second-order adjoint at the root.  The reference seeds it with a RUN-TIME zero (shessian!, hessian.jl:714-717:
// `adj2 = zero(T)`) and its generic node rule computes adj2 * y^2 + adj * h (hessian.jl:346-360) — so wherever a first
// partial is Inf or NaN (log'(0), exp overflow, ...) the reference's Hessian entry is NaN (0 * Inf), not +-Inf.  A
// literal zero would be folded away here and give +-Inf instead; the seed is therefore an SSA value the compiler
// must multiply with (no fast-math), which costs one multiply-add per first generic node under the root.
Val zero_seed(Body &b) {
    if (env_int("EXAHIP_FOLD_ZERO_SEED", 0)) return Emitter::litf(0.0);     // experiment knob: the folded form
    return b.e.raw("0.0", false);
}

// symbolic algebra for the reverse sweeps
struct GenAlg {
    using T = Val;
    Body &b;
    const std::vector<int> &comp;
    int cnt = 0;
    std::vector<Val> acc;
    std::vector<char> has;
    GenAlg(Body &bb, const std::vector<int> &c, int nslots) : b(bb), comp(c), acc(nslots), has(nslots, 0) {}
    T y1(int n) { return b.fv[n].y1; }
    T y2(int n) { return b.fv[n].y2; }
    T h11(int n) { return b.fv[n].h11; }
    T h12(int n) { return b.fv[n].h12; }
    T h22(int n) { return b.fv[n].h22; }
    T mul(T x, T y) { return b.e.mul(x, y); }
    T add(T x, T y) { return b.e.add(x, y); }
    T neg(T x) { return b.e.neg(x); }
    void put(T v) {
        const int s = comp[cnt++] - 1;
        v = b.e.tod(v);
        if (!has[s]) { acc[s] = v; has[s] = 1; }
        else acc[s] = b.e.add(acc[s], v);
    }
    void leaf1(int, T adj) { put(adj); }
    void leaf2(int n1, int n2, T val, bool cross) {
        if (cross) {
            // hessian.jl:251-268: i == j ? 2adj : adj, compared on run-time indices
            if (b.p.ad[n1].key == b.p.ad[n2].key) val = b.e.mul(Emitter::litf(2), val);
            else if (!val.lit_eq(0)) {
                Val i = b.fv[n1].vidx, j = b.fv[n2].vidx;
                if (i.is_lit() && j.is_lit()) { if (i.i == j.i) val = b.e.mul(Emitter::litf(2), val); }
                else val = b.e.raw("(" + b.e.s(i) + " == " + b.e.s(j) + " ? 2.0 * " + b.e.sd(val) + " : " + b.e.sd(val) + ")", false);
            }
        }
        put(val);
    }
};

void emit_lines(std::ostringstream &os, const Emitter &e, const char *indent = "    ") {
    for (const auto &l : e.lines) os << indent << l << "\n";
}

// ---- load stage / evaluation stage of a pattern body (chained, software-pipelined callbacks) ------------------------
// The SSA lines of a body are split into what touches memory — the loads of x, y, theta and the iterator columns,
// together with the integer index arithmetic they need — and the arithmetic that consumes the loaded values.  The
// load stage hands its values over in `in[]` (doubles) and `ik[]` (integers read from data columns); the evaluation
// stage re-derives the pure index arithmetic (scalar / cheap integer work) and reads everything else from there.
struct Split {
    std::vector<std::string> load, eval;
    int nin = 0, nik = 0;
};
bool is_memory_read(const std::string &expr) {
    for (const char *pre : {"x[", "y[", "th[", "v[", "((const long*)P[", "((const double*)P["})
        if (expr.compare(0, strlen(pre), pre) == 0) return true;
    return false;
}
Split split_body(const Emitter &e) {
    Split sp;
    size_t d = 0;
    for (size_t li = 0; li < e.lines.size(); li++) {
        const std::string &line = e.lines[li];
        if (d < e.defs.size() && e.defs[d].line == (int)li) {
            const Emitter::Def &df = e.defs[d++];
            if (is_memory_read(df.expr)) {
                if (env_int("EXAHIP_NT_LOADS", 0) && (df.expr.compare(0, 2, "x[") == 0 || df.expr.compare(0, 2, "y[") == 0))
                    sp.load.push_back(std::string(df.is_int ? "const long " : "const double ") + df.name + " = __builtin_nontemporal_load(&" + df.expr + ");");
                else
                sp.load.push_back(line);
                if (df.is_int) {
                    sp.load.push_back("ik[" + std::to_string(sp.nik) + "] = " + df.name + ";");
                    sp.eval.push_back("const long " + df.name + " = ik[" + std::to_string(sp.nik++) + "];");
                } else {
                    sp.load.push_back("in[" + std::to_string(sp.nin) + "] = " + df.name + ";");
                    sp.eval.push_back("const double " + df.name + " = in[" + std::to_string(sp.nin++) + "];");
                }
            } else if (df.is_int) {
                sp.load.push_back(line);       // index arithmetic: needed by the loads, recomputed by the evaluation
                sp.eval.push_back(line);
            } else sp.eval.push_back(line);
        } else sp.eval.push_back(line);        // multi-value statements (sincos): arithmetic
    }
    return sp;
}
// doubles / integers the load stage of (callback, pattern) hands over (per module generation)
std::map<std::pair<int, int>, std::pair<int, int>> g_handover;
int chain_len(int cb) {
    if (cb == CB_HESSC) return env_int("EXAHIP_CHAIN", 4);       // tiles per workgroup of exa_hessc; 0 = no such kernel
    return 0;
}

const char *kPrelude = R"HIP(// Generated by libexahip (examodels.jl_amd/csrc/exa_codegen.cpp) for gfx950.  Do not edit.
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
#define EXA_LOG2 0.69314718055994530942
#define EXA_LOG10 2.30258509299404568402
#define EXA_PI 3.14159265358979323846
#define EXA_D2R (EXA_PI / 180.0)
#define EXA_R2D (180.0 / EXA_PI)
#define EXA_BLOCK @BLOCK@
#define EXA_PULL_PPT @PULLPPT@
#define EXA_AUG_LONG 512
#define EXA_AUG_CHUNK 8192
static __device__ __forceinline__ double exa_sq(double x) { return x * x; }
// FP64 add to memory as ONE hardware instruction (global_atomic_add_f64 / ds_add_f64), by builtin: what
// unsafeAtomicAdd becomes depends on the compiler's header and flags (the hiprtc bundled with PyTorch's ROCm 7.0 turns
// it into a compare-and-swap LOOP — 10x slower on contended targets)
static __device__ __forceinline__ void exa_atomic_add(double* p, double v) {
    (void)__builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double*)p, v);
}
static __device__ __forceinline__ void exa_lds_add(double* p, double v) {
    (void)__builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)p, v);
}
static __device__ __forceinline__ double exa_sign(double x) { return x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : x); }
// Base.sind / cosd / tand: exact rem(x, 360), quadrant selected before the conversion to radians — exact zeros and poles
// at the multiples of 90 (sind(180) = 0, cosd(90) = 0, tand(90) = Inf), like Julia's; +-Inf -> NaN.
static __device__ __forceinline__ double exa_sind(double x) {
    if (!(fabs(x) < __builtin_inf())) return x != x ? x : __builtin_nan("");
    const double rx = copysign(fmod(x, 360.0), x), arx = fabs(rx);
    if (rx == 0.0) return rx;
    if (arx < 45.0) return sin(EXA_D2R * rx);
    if (arx <= 135.0) return copysign(cos(EXA_D2R * (90.0 - arx)), rx);
    if (arx == 180.0) return copysign(0.0, rx);
    if (arx < 225.0) return sin(EXA_D2R * ((180.0 - arx) * (rx > 0.0 ? 1.0 : -1.0)));
    if (arx <= 315.0) return -copysign(cos(EXA_D2R * (270.0 - arx)), rx);
    return sin(EXA_D2R * (rx - copysign(360.0, rx)));
}
static __device__ __forceinline__ double exa_cosd(double x) {
    if (!(fabs(x) < __builtin_inf())) return x != x ? x : __builtin_nan("");
    const double rx = fabs(fmod(x, 360.0));
    if (rx <= 45.0) return cos(EXA_D2R * rx);
    if (rx < 135.0) return sin(EXA_D2R * (90.0 - rx));
    if (rx <= 225.0) return -cos(EXA_D2R * (180.0 - rx));
    if (rx < 315.0) return sin(EXA_D2R * (rx - 270.0));
    return cos(EXA_D2R * (360.0 - rx));
}
static __device__ __forceinline__ double exa_tand(double x) { return exa_sind(x) / exa_cosd(x); }
static __device__ __forceinline__ double exa_sinc(double x) { return x == 0.0 ? 1.0 : sinpi(x) / (EXA_PI * x); }
// sin and cos together, FP64.  FP64 transcendentals are software sequences on CDNA4 and the ocml pair costs ~50 FP64
// instructions (it carries double-double terms for < 1 ulp); for |x| < 2^19 * pi/2 this version does a 3-term Cody-Waite
// reduction with FMAs (the products k*PIO2_1, k*PIO2_2 are exact for |k| < 2^20: 33-bit constants) and the fdlibm
// minimax kernels on [-pi/4, pi/4] — ~28 FP64 instructions, <= 1.5 ulp.  Larger arguments take the ocml path.
static __device__ __forceinline__ void exa_sincos(double x, double* sp, double* cp) {
    if (!(fabs(x) < 823549.6)) { sincos(x, sp, cp); return; }            // also routes NaN/Inf to ocml
    const double kd = __builtin_rint(x * 6.36619772367581382433e-01);
    double r = __builtin_fma(-kd, 1.57079632673412561417e+00, x);   // pio2_1  (first 33 bits of pi/2)
    r = __builtin_fma(-kd, 6.07710050630396597660e-11, r);     // pio2_2  (next 33 bits)
    r = __builtin_fma(-kd, 2.02226624879595063154e-21, r);     // pio2_2t (tail): pi/2 to ~119 bits in total
    const double z = r * r;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double sn = __builtin_fma(r * z, ps, r);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double cs = w + (((1.0 - w) - hz) + z * z * pc);
    const int q = (int)kd & 3;
    const double s0 = (q & 1) ? cs : sn, c0 = (q & 1) ? sn : cs;
    *sp = (q & 2) ? -s0 : s0;
    *cp = ((q + 1) & 2) ? -c0 : c0;
}
static __device__ __forceinline__ double exa_sin(double x) { double s, c; exa_sincos(x, &s, &c); return s; }
static __device__ __forceinline__ double exa_cos(double x) { double s, c; exa_sincos(x, &s, &c); return c; }
// x^n, run-time integer n (Base.^(::Float64, ::Integer)): by squaring; n < 0 through the reciprocal
static __device__ double exa_powi(double x, long n) {
    if (n == 0) return 1.0;
    if (n < 0) { x = 1.0 / x; n = -n; }
    double y = 1.0;
    while (n > 1) { if (n & 1) y *= x; x *= x; n >>= 1; }
    return x * y;
}
// COO store epilogue.  A wavefront owns 64*S CONTIGUOUS output doubles (lane l owns S of them).  Storing them
// lane-by-lane is a 8*S-byte-strided pattern (3.8 TB/s measured); instead the wavefront stages its block in LDS
// slot-major (conflict-free ds_write_b64) and streams it out lane-interleaved: every store instruction is one
// fully coalesced 512 B burst, non-temporal (measured 5.7 TB/s, tools/store_bench.hip).
// Flushes the S slots of PP consecutive points (point group g of the wavefront): one CONTIGUOUS run of PP*S doubles,
// so every store instruction is a full 512-B burst.  PP = 64 stages the whole wavefront at once; wide patterns use
// PP = 32/16/8 (several passes) to bound LDS per workgroup.  tile is slot-major with leading dimension LD, chosen by
// the generator so that the transposed ds_read_b64 of a 32-lane half hits 32 distinct bank pairs.
template <int S, int PP, int LD>
static __device__ __forceinline__ void exa_flush_points(double* __restrict__ out, long obase, long npts, const double* tile, int lane, int g) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int CNT = S * PP;
    double* __restrict__ dst = out + obase + (long)CNT * g;
    if (npts >= (long)(g + 1) * PP) {
        // all PP points of the group are in range (every wavefront but the last of a pattern): no per-element test
#pragma unroll
        for (int k = 0; k * 64 < CNT; k++) {
            const int j = k * 64 + lane;
            const int l2 = j / S, s2 = j - l2 * S;
            if ((k + 1) * 64 <= CNT || j < CNT) __builtin_nontemporal_store(tile[s2 * LD + l2], dst + j);
        }
    } else {
#pragma unroll
        for (int k = 0; k * 64 < CNT; k++) {
            const int j = k * 64 + lane;
            const int l2 = j / S, s2 = j - l2 * S;
            if (j < CNT && g * PP + l2 < npts) __builtin_nontemporal_store(tile[s2 * LD + l2], dst + j);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// The same flush with a FIXED number of store instructions and no branch around any of them (chained callbacks): lanes
// without a slot of their own store to `sink` (a scratch line per lane) instead of being masked off.  On gfx9 loads and
// stores retire through ONE in-order counter (vmcnt); only when the number of stores between a load and its use is the
// same on every path can the compiler wait for the load alone (vmcnt(N > 0)) instead of draining the stores as well.
template <int S, int PP, int LD>
static __device__ __forceinline__ void exa_flush_points_nb(double* __restrict__ out, double* __restrict__ sink, long obase, long npts, const double* tile,
                                                            int lane, int g) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int CNT = S * PP;
    double* __restrict__ dst = out + obase + (long)CNT * g;
    const long left = npts - (long)g * PP;          // points of this group that exist
#pragma unroll
    for (int k = 0; k * 64 < CNT; k++) {
        const int j = k * 64 + lane;
        int l2 = j / S;
        const int s2 = j - l2 * S;
        const bool ok = ((k + 1) * 64 <= CNT || j < CNT) && l2 < left;
        if ((k + 1) * 64 > CNT) l2 = l2 < PP ? l2 : PP - 1;
        double* __restrict__ p = ok ? dst + j : sink + lane;
        __builtin_nontemporal_store(tile[s2 * LD + l2], p);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// Scatter to a target that is THE SAME for every data point of the pattern (a literal variable index, e.g. the shared
// step length dt[1] of the rocket model): the wavefront adds its 64 contributions with DPP/shuffle butterflies and issues
// ONE atomic — 64x fewer same-address FP64 atomics, which otherwise serialise at the memory side.
static __device__ __forceinline__ void exa_wave_atomic_add(double* p, double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) exa_atomic_add(p, v);
}
// Scatter through a DATA index: the target may be one variable shared by every data point (a step length, a slack, a
// reference bus reached through a table column) or one of a FEW shared variables in any order (design variables behind a
// scenario column) — up to 64 same-address atomics per wavefront that serialise chip-wide at ~12 ns each (1e7 points on
// one variable: 120 ms; on two, interleaved at random: 81 ms; on sixteen: 26 ms).  The wavefront therefore peels groups
// of lanes that name the same variable — the first pending lane's target, a ballot, a butterfly over the group's values,
// ONE atomic by its first lane — for as long as a group has at least two lanes; what is left (all lanes, when every
// lane names its own variable: one ballot wasted) goes lane by lane.  Must be reached by the whole wavefront.
static __device__ __forceinline__ void exa_scatter_add(double* __restrict__ out, long idx, double v, bool act) {
    unsigned long long todo = __ballot(act);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const long t = __shfl(idx, leader, 64);
        const unsigned long long grp = __ballot(act && idx == t) & todo;
        if (__popcll(grp) < 2) break;
        double s = ((grp >> lane) & 1ull) ? v : 0.0;
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == leader) exa_atomic_add(&out[t], s);
        todo &= ~grp;
    }
    if ((todo >> lane) & 1ull) exa_atomic_add(&out[idx], v);
}
// The loop-free form (all lanes on ONE variable -> one atomic, else lane by lane) for bodies of thousands of SSA values:
// such kernels run at the 512-VGPR limit with scratch spills, and the compiler bundled with PyTorch's ROCm 7.0 (hiprtc)
// miscompiles the peeling loop there (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION in exa_hprod of
// tests/test_random_expressions.py seed 541; the same source built by ROCm 7.2's hipcc is correct).
static __device__ __forceinline__ void exa_scatter_add1(double* __restrict__ out, long idx, double v, bool act) {
    const unsigned long long m = __ballot(act);
    if (m == 0) return;
    const long first = __shfl(idx, __ffsll((long long)m) - 1, 64);
    if (__ballot(act && idx != first) == 0) {
        double s = act ? v : 0.0;
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0) exa_atomic_add(&out[first], s);
    } else if (act) {
        exa_atomic_add(&out[idx], v);
    }
}
// sum over the 256-thread workgroup: 64-lane wavefront butterflies, then 4 partials through LDS
static __device__ __forceinline__ double exa_block_sum(double v) {
    __shared__ double red[EXA_BLOCK / 64];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) { for (int w = 0; w < EXA_BLOCK / 64; w++) s += red[w]; }
    return s;
}
// cons_nln! augmentation, second stage (the reference's compress_to_dense, KA ext :691-697): one thread per distinct
// target row adds its contributions, listed in insertion order, to the base value written by exa_cons.
extern "C" __global__ void __launch_bounds__(EXA_BLOCK) exa_aug_gather(const long* __restrict__ rows, const long* __restrict__ ptr,
        const long* __restrict__ perm, const double* __restrict__ buf, double* __restrict__ c, long nrows) {
    const long t = (long)blockIdx.x * EXA_BLOCK + threadIdx.x;
    if (t >= nrows) return;
    const long b = ptr[t], e = ptr[t + 1];
    if (e - b > EXA_AUG_LONG) return;          // rows collecting very many terms: exa_aug_long + exa_aug_fold
    const long r = rows[t];
    double s = c[r];
    long j = b;
    for (; j + 4 <= e; j += 4) {               // insertion order kept; four gathers in flight instead of one
        const long q0 = perm[j], q1 = perm[j + 1], q2 = perm[j + 2], q3 = perm[j + 3];
        const double a0 = buf[q0], a1 = buf[q1], a2 = buf[q2], a3 = buf[q3];
        s += a0; s += a1; s += a2; s += a3;
    }
    for (; j < e; j++) s += buf[perm[j]];
    c[r] = s;
}
// A row that collects thousands of terms (a coupling constraint summing over every data point) would be one thread's
// sequential loop above — 1.6 s for 1e7 terms.  Such rows are summed cooperatively in a FIXED order instead: a partial
// sum per chunk of EXA_AUG_CHUNK terms (blockIdx.x = long row, blockIdx.y = chunk), then one thread folds the chunks.
extern "C" __global__ void __launch_bounds__(EXA_BLOCK) exa_aug_long(const long* __restrict__ list, const long* __restrict__ ptr,
        const long* __restrict__ perm, const double* __restrict__ buf, double* __restrict__ partial, int chunks) {
    const long t = list[blockIdx.x];
    const long beg = ptr[t] + (long)blockIdx.y * EXA_AUG_CHUNK;
    const long end = beg + EXA_AUG_CHUNK < ptr[t + 1] ? beg + EXA_AUG_CHUNK : ptr[t + 1];
    double s = 0.0;
    for (long j = beg + threadIdx.x; j < end; j += EXA_BLOCK) s += buf[perm[j]];
    const double tot = exa_block_sum(s);
    if (threadIdx.x == 0) partial[(long)blockIdx.x * chunks + blockIdx.y] = tot;
}
extern "C" __global__ void __launch_bounds__(EXA_BLOCK) exa_aug_fold(const long* __restrict__ list, const long* __restrict__ rows,
        const double* __restrict__ partial, int chunks, double* __restrict__ c, long nlong) {
    const long l = (long)blockIdx.x * EXA_BLOCK + threadIdx.x;
    if (l >= nlong) return;
    const long r = rows[list[l]];
    double s = c[r];
    for (int k = 0; k < chunks; k++) s += partial[l * chunks + k];
    c[r] = s;
}
// second stage of obj: one workgroup folds the per-workgroup partial sums in a fixed order (deterministic)
extern "C" __global__ void __launch_bounds__(1024) exa_reduce_partials(const double* __restrict__ part, long n, double* __restrict__ out) {
    __shared__ double red[16];
    // eight independent accumulators: eight loads in flight per thread (the fused sweep leaves one partial per workgroup
    // of EVERY pattern — 78 000 for LV 1e7 — and a single dependent chain took 26 us); the order is fixed all the same
    double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    long i = threadIdx.x;
    for (; i + 7 * 1024 < n; i += 8 * 1024) {
#pragma unroll
        for (int q = 0; q < 8; q++) a[q] += part[i + q * 1024];
    }
    for (; i < n; i += 1024) a[0] += part[i];
    double v = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { double s = 0.0; for (int w = 0; w < 16; w++) s += red[w]; out[0] = s; }
}
)HIP";

// LDS budget per 256-thread workgroup for the store staging (tuning knob; changes the source and so the cache key)
int lds_budget() { return env_int("EXAHIP_LDS_BUDGET", 40960); }
int tile_doubles(int S);
// a pattern too wide to stage even 8 points per pass within the 160 KB of LDS falls back to direct per-lane stores
bool use_tile(int S) { return S >= 2 && (long)(kBlock / 64) * tile_doubles(S) * 8 <= env_int("EXAHIP_LDS_MAX", 150000); }
// points staged per pass: the largest of 64/32/16/8 whose tile (4 wavefronts) fits the budget
int tile_pp(int S) {
    for (int pp = 64; pp > 8; pp >>= 1)
        if ((kBlock / 64) * S * (pp + 1) * 8 <= lds_budget()) return pp;
    return 8;
}
// Leading dimension of the slot-major tile.  Writes (lane-consecutive) are conflict-free for any LD; the transposed
// read of lane j fetches element (j % S) * LD + j / S, and a ds_read_b64 is serviced per 32-lane half with 32 bank
// pairs (MI355X_MICROARCH.md §LDS) — pick the LD in [PP, PP+32] with the fewest extra cycles.
int tile_ld(int S) {
    const int pp = tile_pp(S), cnt = S * pp;
    int best = pp + 1;
    long best_cost = -1;
    for (int ld = pp; ld <= pp + 32; ld++) {
        long cost = 0;
        for (int k = 0; k * 64 < cnt; k++)
            for (int half = 0; half < 2; half++) {
                int mult[32] = {0};
                int mx = 0;
                for (int l = 0; l < 32; l++) {
                    const int j = k * 64 + half * 32 + l;
                    if (j >= cnt) continue;
                    const int d = (j % S) * ld + j / S;
                    mx = std::max(mx, ++mult[d & 31]);
                }
                cost += mx > 0 ? mx - 1 : 0;
            }
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = ld; }
    }
    return best;
}
int tile_doubles(int S) { return S * tile_ld(S); }   // per wavefront

// prologue of a COO-writing pattern function: tail lanes are clamped (they recompute the last point and their
// stores are masked) so that the whole wavefront reaches the cooperative store epilogue
void emit_coo_prologue(std::ostringstream &os, const Body &b, const ParamLayout &L, int pi, bool tile) {
    os << "    const long I0 = " << b.P(L.pat[pi].lo) << " + tid;\n    const long hi = " << b.P(L.pat[pi].hi) << ";\n";
    if (tile) {
        os << "    const int lane = threadIdx.x & 63;\n    if (I0 - lane >= hi) return;\n"
           << "    const long I = I0 < hi ? I0 : hi - 1;\n";
    } else {
        os << "    if (I0 >= hi) return;\n    const long I = I0;\n";
    }
}
void emit_coo_stores(std::ostringstream &os, const Body &b, int word_o, int S, const std::vector<std::string> &vals, bool tile,
                     const std::string &out = "out", const std::string &tag = "", bool no_branch = false) {
    if (no_branch && !tile) {
        // narrow patterns (S < 2), chained: one store per slot, lanes beyond the shard store to the sink
        os << "    {\n    double* __restrict__ po = I0 < hi ? " << out << " + " << b.P(word_o) << " + " << S << "L * I : sink + (threadIdx.x & 63);\n";
        for (int s = 0; s < S; s++) os << "    po[" << (s == 0 ? "0" : "(I0 < hi ? " + std::to_string(s) + " : 0)") << "] = " << vals[s] << ";\n";
        os << "    }\n";
        return;
    }
    if (tile) {
        const int pp = tile_pp(S), ld = tile_ld(S);
        // `lds` is this WAVEFRONT's private staging region (sized for the widest tile of the kernel): wavefronts never
        // share LDS words, so no workgroup barrier is needed even when a function flushes two tiles of different shape
        os << "    {\n    double* tile = lds;\n"
           << "    const long obase = " << b.P(word_o) << " + " << S << "L * (I0 - lane);\n    const long npts = hi - (I0 - lane);\n";
        for (int g = 0; g < 64 / pp; g++) {
            if (pp == 64) {
                for (int s = 0; s < S; s++) os << "    tile[" << s * ld << " + lane] = " << vals[s] << ";\n";
            } else {
                os << "    if ((lane / " << pp << ") == " << g << ") {\n";
                for (int s = 0; s < S; s++) os << "        tile[" << s * ld << " + (lane % " << pp << ")] = " << vals[s] << ";\n";
                os << "    }\n";
            }
            if (no_branch) os << "    exa_flush_points_nb<" << S << ", " << pp << ", " << ld << ">(" << out << ", sink, obase, npts, tile, lane, " << g << ");\n";
            else os << "    exa_flush_points<" << S << ", " << pp << ", " << ld << ">(" << out << ", obase, npts, tile, lane, " << g << ");\n";
        }
        os << "    }\n";
    } else {
        os << "    if (I0 < hi) {\n    const long o" << tag << " = " << b.P(word_o) << " + " << S << "L * I;\n";
        for (int s = 0; s < S; s++) os << "    " << out << "[o" << tag << " + " << s << "] = " << vals[s] << ";\n";
        os << "    }\n";
    }
}

// slots stored through a position table (exa_c*p): slot q of the uncompressed COO goes to out[pos[q]]
void emit_coo_stores_permuted(std::ostringstream &os, const Body &b, int word_o, int S, const std::vector<std::string> &vals, const std::string &tag) {
    os << "    if (I0 < hi) {\n    const long o" << tag << " = " << b.P(word_o) << " + " << S << "L * I;\n";
    for (int s = 0; s < S; s++) os << "    out[pos[o" << tag << " + " << s << "]] = " << vals[s] << ";\n";
    os << "    }\n";
}

// ---- merged slots of a fused group (compressed Hessian of data-indexed models) -------------------------------------
// The patterns of a group put many of their Hessian slots on the SAME matrix entry for every data point: ACOPF's four
// branch-flow constraints have 40 slots on the 10 pairs of {va_f, va_t, vm_f, vm_t}.  For the COMPRESSED Hessian those can
// be added in registers before anything is stored: a merged slot per distinct unordered pair of index expressions.
// Index expressions compare by a canonical text that does not depend on SSA numbering (columns by their aliased
// parameter word), so the value kernel and the structure kernel — generated separately — agree on the merged slots.
std::string index_key(const Pattern &p, const ParamLayout &L, int pi, int k) {
    const exa_node_t &nd = p.nodes[k];
    switch (nd.op) {
    case EXA_OP_CONST_I: return "i" + std::to_string(nd.ival);
    case EXA_OP_DATA: return "c" + std::to_string(L.pat[pi].col[nd.a]);
    case EXA_OP_UN: return "u" + std::to_string(nd.fn) + "(" + index_key(p, L, pi, nd.a) + ")";
    case EXA_OP_BIN: return "b" + std::to_string(nd.fn) + "(" + index_key(p, L, pi, nd.a) + "," + index_key(p, L, pi, nd.b) + ")";
    default: return "?" + std::to_string(pi) + ":" + std::to_string(k);      // never equal to anything of another pattern
    }
}
struct MergedSlot { std::string key; Val ia, ib, sum; bool has = false; };
// slot s of pattern b.p (accumulated value `acc`, or structure only) joins the merged slot of its pair
void merge_slot(std::vector<MergedSlot> &ms, Body &b, int s, const Val *acc) {
    const Pattern &p = b.p;
    const int la = p.slotvar2[s].first, lb = p.slotvar2[s].second;
    std::string ka = index_key(p, b.L, b.pi, p.ad[la].ir), kb = index_key(p, b.L, b.pi, p.ad[lb].ir);
    if (kb < ka) std::swap(ka, kb);
    const std::string key = ka + "|" + kb;
    for (MergedSlot &q : ms)
        if (q.key == key) { if (acc) q.sum = q.has ? b.e.add(q.sum, *acc) : b.e.tod(*acc); q.has = q.has || acc; return; }
    MergedSlot q;
    q.key = key; q.ia = b.fv[la].vidx; q.ib = b.fv[lb].vidx;
    if (acc) { q.sum = b.e.tod(*acc); q.has = true; }
    ms.push_back(q);
}
int merged_slot_count(const Model &m, const ParamLayout &L, const std::vector<int> &grp) {
    Emitter E;
    std::vector<MergedSlot> ms;
    for (int pk : grp) {
        Body b(m, pk, L, &E);
        b.forward(b.p.ad_root, 0, true);
        for (int s = 0; s < b.p.o2step; s++) merge_slot(ms, b, s, nullptr);
    }
    return (int)ms.size();
}
// values of the merged slots of group gi, stored through pos[] (sorted order of the MERGED slot space); mo = first merged
// slot of this group's data point 0
void gen_merged_hess_fn(std::ostringstream &os, const Model &m, const ParamLayout &L, int gi) {
    const auto &grp = L.groups[CB_HESS][gi];
    os << "static __device__ __forceinline__ void g" << gi << "_hessm(const long* __restrict__ P, const double* __restrict__ x, "
          "const double* __restrict__ y, const double* __restrict__ th, double* __restrict__ out, double sigma, long tid, "
          "const unsigned* __restrict__ pos, long mo) {\n";
    { Body b0(m, grp.front(), L); emit_coo_prologue(os, b0, L, grp.front(), false); }
    Emitter E;
    std::vector<MergedSlot> ms;
    for (int pk : grp) {
        Body b(m, pk, L, &E);
        const Pattern &p = b.p;
        b.forward(p.ad_root, 2, false);
        Val adj = p.kind == EXA_PAT_OBJ ? E.raw("sigma", false) : E.raw("y[" + b.row0() + "]", false);
        GenAlg a(b, p.comp2, p.o2step);
        hrpass0(p, p.ad_root, a, adj, zero_seed(b));
        for (int s = 0; s < p.o2step; s++) merge_slot(ms, b, s, &a.acc[s]);
    }
    emit_lines(os, E);
    const size_t S = ms.size();
    os << "    const long o_ = mo + " << S << "L * (I - " << Body(m, grp.front(), L).P(L.pat[grp.front()].lo) << ");\n";
    for (size_t j = 0; j < S; j++) os << "    out[pos[o_ + " << j << "]] = " << E.sd(ms[j].sum) << ";\n";
    os << "}\n";
}
void gen_merged_struct_fn(std::ostringstream &os, const Model &m, const ParamLayout &L, int gi) {
    const auto &grp = L.groups[CB_HESS][gi];
    os << "static __device__ __forceinline__ void g" << gi << "_hstm(const long* __restrict__ P, long* __restrict__ rows, long* __restrict__ cols, "
          "long tid, long mo) {\n";
    { Body b0(m, grp.front(), L); emit_coo_prologue(os, b0, L, grp.front(), false); }
    Emitter E;
    std::vector<MergedSlot> ms;
    for (int pk : grp) {
        Body b(m, pk, L, &E);
        b.forward(b.p.ad_root, 0, true);
        for (int s = 0; s < b.p.o2step; s++) merge_slot(ms, b, s, nullptr);
    }
    emit_lines(os, E);
    os << "    const long o_ = mo + " << ms.size() << "L * (I - " << Body(m, grp.front(), L).P(L.pat[grp.front()].lo) << ");\n";
    for (size_t j = 0; j < ms.size(); j++) {
        const std::string si = E.s(ms[j].ia), sj = E.s(ms[j].ib);
        os << "    rows[o_ + " << j << "] = " << si << " >= " << sj << " ? " << si << " : " << sj << "; cols[o_ + " << j << "] = " << si << " >= " << sj
           << " ? " << sj << " : " << si << ";\n";
    }
    os << "}\n";
}

// ---- gather ("pull") formulation of the objective gradient ------------------------------------------------
// index expression == a * (RANGE column) + c ?
struct Affine { bool ok = false; int col = -1; int64_t a = 0, c = 0; };
Affine affine(const Pattern &p, int k) {
    const exa_node_t &nd = p.nodes[k];
    Affine r;
    if (nd.op == EXA_OP_CONST_I) { r.ok = true; r.c = nd.ival; return r; }
    if (nd.op == EXA_OP_DATA) {
        if (p.cols[nd.a].type != EXA_COL_RANGE) return r;
        r.ok = true; r.col = nd.a; r.a = 1; return r;
    }
    if (nd.op == EXA_OP_UN && (nd.fn == EXA_U_PLUS || nd.fn == EXA_U_MINUS)) {
        Affine x = affine(p, nd.a);
        if (!x.ok) return r;
        if (nd.fn == EXA_U_MINUS) { x.a = -x.a; x.c = -x.c; }
        return x;
    }
    if (nd.op == EXA_OP_BIN && (nd.fn == EXA_B_ADD || nd.fn == EXA_B_SUB || nd.fn == EXA_B_MUL)) {
        Affine x = affine(p, nd.a), y = affine(p, nd.b);
        if (!x.ok || !y.ok) return r;
        if (nd.fn == EXA_B_MUL) {
            if (x.col >= 0 && y.col >= 0) return r;
            if (y.col >= 0) std::swap(x, y);
            r.ok = true; r.col = x.col; r.a = x.a * y.c; r.c = x.c * y.c; return r;
        }
        const int64_t sg = nd.fn == EXA_B_ADD ? 1 : -1;
        if (x.col >= 0 && y.col >= 0 && x.col != y.col) return r;
        r.ok = true; r.col = x.col >= 0 ? x.col : y.col; r.a = x.a + sg * y.a; r.c = x.c + sg * y.c;
        if (r.a == 0) r.col = -1;
        return r;
    }
    return r;
}

// ---- scattered `out[idx-1] += val` (grad of data-indexed patterns, J'v, Hv) ------------------------------------
// Three mechanisms, picked per target at generation time:
//   * literal index (same target for every data point)      -> wavefront butterfly + ONE atomic (exa_wave_atomic_add);
//   * index = (unit-step range value) + c for >= 2 targets   -> the wavefront's contributions fall into a window of
//     64 + span consecutive variables: accumulate them in LDS (ds_add_f64), then 64 + span global atomics instead of
//     64 per target (LV J'v: 192 -> 66 per wavefront);
//   * anything else                                           -> one FP64 hardware atomic per lane.
int g_lds_need[CB_COUNT];   // doubles of LDS per wavefront needed by the scatter windows of each callback (per module)
// literal scatter targets of every (callback, pattern): 0-based variable indices, in the order of the pattern's `lit[]`
std::map<std::pair<int, int>, std::vector<std::string>> g_lit_idx;
// largest scatter body (SSA lines) per callback: bodies of thousands of values run at the 512-VGPR limit with scratch
// spills, and such kernels have produced wrong sums / memory faults whenever a loop sat around or inside the body (the
// 16-tile loop, the peeling loop of exa_scatter_add) — with the hiprtc of ROCm 7.0 and, less often, the hipcc of 7.2
// (sweep over 120 random depth-5/6 models, tests/test_random_expressions.py).  Past kHugeBody lines the generator emits
// no loop: one tile per workgroup, the loop-free exa_scatter_add1.
std::map<int, size_t> g_scatter_lines;
int huge_body() { return env_int("EXAHIP_HUGE_BODY", 1000); }
bool g_loopfree[CB_COUNT];      // set from a dry pass over the callback's scatter bodies (generate_module)

struct Scatter {
    struct Item { const Pattern *p; int pi; int ir; Val vidx, val; };
    Emitter &e;
    const ParamLayout &L;
    std::vector<Item> items;
    std::vector<std::string> lit_idx;
    bool loopfree = false;        // this callback has a huge body somewhere: no peeling loop (exa_scatter_add1)
    Scatter(Emitter &ee, const ParamLayout &ll) : e(ee), L(ll) {}
    explicit Scatter(Body &bb) : e(bb.e), L(bb.L) {}
    void add(Body &b, int ad_leaf, Val val) {
        if (val.lit_eq(0)) return;
        items.push_back({&b.p, b.pi, b.p.ad[ad_leaf].ir, b.fv[ad_leaf].vidx, val});
    }
    // Contributions of ONE thread to the same variable are added in registers first: within a pattern (hprod: one item
    // per distinct variable) and — fused groups — ACROSS patterns: the four branch-flow constraints of ACOPF, the
    // angle-difference and the thermal-limit constraints all scatter to the voltage variables of the same two buses,
    // 26 same-target atomics per branch that become 8 (index expressions compare by text: the patterns name the same
    // aliased table column, exa_plan.cpp).
    void merge() {
        std::vector<Item> out;
        for (const Item &it : items) {
            const std::string key = e.s(it.vidx);
            bool found = false;
            for (Item &o : out)
                if (e.s(o.vidx) == key) { o.val = e.add(o.val, it.val); found = true; break; }
            if (!found) out.push_back(it);
        }
        items.swap(out);
    }
    // returns the LDS doubles needed per wavefront; fills `lines`; sets full_wave
    int emit(std::vector<std::string> &lines, bool &full_wave) {
        merge();
        // candidates: index = (unit-step range value) + c, all on the same range column; clustered into windows of
        // offsets that lie within 64 of each other (one window per variable block the pattern touches)
        struct Cand { int item; int64_t c; };
        std::vector<Cand> cand;
        int colword = -1;
        if (env_int("EXAHIP_LDS_SCATTER", 1)) {
            for (size_t k = 0; k < items.size(); k++) {
                Affine a = affine(*items[k].p, items[k].ir);
                if (!a.ok || a.col < 0 || a.a != 1 || items[k].p->cols[a.col].step != 1) continue;
                const int w = L.pat[items[k].pi].col[a.col];
                if (colword >= 0 && w != colword) continue;
                colword = w;
                cand.push_back({(int)k, a.c});
            }
            std::stable_sort(cand.begin(), cand.end(), [](const Cand &x, const Cand &y) { return x.c < y.c; });
        }
        std::vector<char> inwin(items.size(), 0);
        int total = 0;
        bool first = true;
        for (size_t g0 = 0; g0 < cand.size();) {
            size_t g1 = g0 + 1;
            while (g1 < cand.size() && cand[g1].c - cand[g0].c <= 64) g1++;
            if (g1 - g0 >= 2) {
                const int64_t cmin = cand[g0].c, cmax = cand[g1 - 1].c;
                const int W = 64 + (int)(cmax - cmin);
                const std::string reg = "(lds + " + std::to_string(total) + ")";
                full_wave = true;
                if (first) lines.push_back("// scatter windows in LDS (one per variable block)");
                first = false;
                lines.push_back("for (int j = lane; j < " + std::to_string(W) + "; j += 64) " + reg + "[j] = 0.0;");
                lines.push_back("__builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier(); "
                                "__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");");
                std::string body = "if (act) {";
                for (size_t q = g0; q < g1; q++) {
                    body += " exa_lds_add(&" + reg + "[lane + " + std::to_string(cand[q].c - cmin) + "], " + e.sd(items[cand[q].item].val) + ");";
                    inwin[cand[q].item] = 1;
                }
                lines.push_back(body + " }");
                lines.push_back("__builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier(); "
                                "__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");");
                // variable (0-based) held by the window's first word: range value of the wavefront's first point + cmin - 1
                lines.push_back("{ const long wb = P[" + std::to_string(colword) + "] + (I0 - lane) + (" + std::to_string(cmin) + "L) - 1L;");
                lines.push_back("  for (int j = lane; j < " + std::to_string(W) + "; j += 64) { const double t_ = " + reg +
                                "[j]; if (t_ != 0.0) exa_atomic_add(&out[wb + j], t_); } }");
                total += W;
            }
            g0 = g1;
        }
        const int W = total;
        for (size_t k = 0; k < items.size(); k++) {
            if (inwin[k]) continue;
            const std::string idx = e.s(e.sub(items[k].vidx, Emitter::liti(1)));
            if (items[k].vidx.is_lit() && env_int("EXAHIP_WAVE_REDUCE", 1)) {
                // same target for every data point: accumulate in a register across this thread's tiles; the kernel
                // adds it to memory ONCE per wavefront after the tile loop (pK_*_fin)
                full_wave = true;
                lines.push_back("lit[" + std::to_string(lit_idx.size()) + "] += act ? " + e.sd(items[k].val) + " : 0.0;");
                lit_idx.push_back(idx);
            } else if (!affine(*items[k].p, items[k].ir).ok && env_int("EXAHIP_WAVE_REDUCE", 1)) {
                // reached through a data column: possibly the same variable for the whole wavefront (exa_scatter_add)
                full_wave = true;
                const bool peel = !loopfree && (int)e.lines.size() <= huge_body();
                lines.push_back(std::string(peel ? "exa_scatter_add" : "exa_scatter_add1") + "(out, " + idx + ", " + e.sd(items[k].val) + ", act);");
            } else {
                lines.push_back("if (act) exa_atomic_add(&out[" + idx + "], " + e.sd(items[k].val) + ");");
            }
        }
        return W;
    }
};
// prologue of a scattering pattern function (grad / jtprod / hprod)
void emit_scatter_prologue(std::ostringstream &os, const Body &b, const ParamLayout &L, int pi, bool full_wave) {
    os << "    const long I0 = " << b.P(L.pat[pi].lo) << " + tid;\n    const long hi = " << b.P(L.pat[pi].hi) << ";\n";
    if (full_wave)
        // A wavefront whose 64 points all lie beyond the pattern skips the body.  Its first point, I0 - lane, is the same in
        // every lane but the compiler cannot know: it would mask the lanes out (exec) around the whole body instead of
        // branching — and values that live ACROSS the body in registers (the per-lane sums of shared targets, added up by a
        // butterfly after the tile loop) were then spilled inside the masked region and reloaded behind it for all lanes:
        // garbage in the lanes that were masked out whenever the body is large enough to spill (AGPRs / scratch), i.e.
        // wrong or astronomically wrong J'v / Hv entries, intermittently, on random depth-6 models.  readfirstlane makes
        // the test scalar: a real branch, no masking.
        os << "    const int lane = threadIdx.x & 63;\n"
              "    { const long w0_ = I0 - lane;\n"
              "      const long wf_ = ((long)__builtin_amdgcn_readfirstlane((int)(w0_ >> 32)) << 32) | (long)(unsigned int)__builtin_amdgcn_readfirstlane((int)w0_);\n"
              "      if (wf_ >= hi) return; }\n"
              "    const bool act = I0 < hi;\n    const long I = act ? I0 : hi - 1;\n";
    else
        os << "    if (I0 >= hi) return;\n    const bool act = true;\n    const long I = I0;\n";
}

std::string fn_name(int pi, const char *cb) { return "p" + std::to_string(pi) + "_" + cb; }
void emit_two_stage(std::ostringstream &os, Body &b, const ParamLayout &L, int pi, int cb, const char *name, bool hess, bool tile,
                    int word_o, int S, const std::vector<std::string> &vals);

// ---- per-pattern device functions -----------------------------------------------------------------
void gen_value_fn(std::ostringstream &os, const Model &m, int pi, const ParamLayout &L) {
    Body b(m, pi, L);
    Val v = b.e.tod(b.cval(b.p.root));
    os << "static __device__ __forceinline__ double " << fn_name(pi, "val")
       << "(const long* __restrict__ P, const double* __restrict__ x, const double* __restrict__ th, long I) {\n";
    emit_lines(os, b.e);
    os << "    return " << b.e.s(v) << ";\n}\n";
}

void gen_cons_fn(std::ostringstream &os, const Model &m, int pi, const ParamLayout &L) {
    Body b(m, pi, L);
    // Two pieces so that a thread handling several points evaluates ALL of them before storing any: the value at an
    // index clamped into the shard (no branch -> one basic block -> the loads of all points are in flight together),
    // then the guarded store.  Base rows: plain store into c; augmentation terms: into the value buffer, gathered per
    // row by exa_aug_gather.
    os << "static __device__ __forceinline__ double " << fn_name(pi, "consv")
       << "(const long* __restrict__ P, const double* __restrict__ x, const double* __restrict__ th, long tid) {\n"
       << "    const long I_ = " << b.P(L.pat[pi].lo) << " + tid, h_ = " << b.P(L.pat[pi].hi) << " - 1;\n"
       << "    return " << fn_name(pi, "val") << "(P, x, th, I_ < h_ ? I_ : h_);\n}\n";
    os << "static __device__ __forceinline__ void " << fn_name(pi, "conss")
       << "(const long* __restrict__ P, double* __restrict__ c, double* __restrict__ aug, long tid, double v) {\n"
       << "    const long I = " << b.P(L.pat[pi].lo) << " + tid;\n    if (I >= " << b.P(L.pat[pi].hi) << ") return;\n";
    if (b.p.kind == EXA_PAT_CONAUG) os << "    aug[" << b.P(L.pat[pi].oa) << " + I] = v;\n";
    else os << "    c[" << b.P(L.pat[pi].o0) << " + I] = v;\n";
    os << "}\n";
}

void gen_first_fn(std::ostringstream &os, const Model &m, int pi, const ParamLayout &L, bool grad) {
    Body b(m, pi, L);
    const Pattern &p = b.p;
    b.forward(p.ad_root, 1, false);
    GenAlg a(b, p.comp1, p.o1step);
    grpass(p, p.ad_root, a, Emitter::litf(1.0));
    const bool tile = !grad && use_tile(p.o1step);
    // index texts are computed first so that they land in e.lines
    std::vector<std::string> stores, vals;
    bool full_wave = false;
    Scatter sc(b);
    sc.loopfree = g_loopfree[CB_GRAD];
    for (int s = 0; s < p.o1step; s++) {
        if (grad) sc.add(b, p.slotvar1[s], a.acc[s]);
        else vals.push_back(b.e.sd(a.acc[s]));
    }
    if (grad) { g_lds_need[CB_GRAD] = std::max(g_lds_need[CB_GRAD], sc.emit(stores, full_wave)); g_lit_idx[{CB_GRAD, pi}] = sc.lit_idx;
                g_scatter_lines[CB_GRAD] = std::max(g_scatter_lines[CB_GRAD], b.e.lines.size()); }
    os << "static __device__ __forceinline__ void " << fn_name(pi, grad ? "grad" : "jac")
       << "(const long* __restrict__ P, const double* __restrict__ x, const double* __restrict__ th, double* __restrict__ out, long tid"
       << ", double* lds" << (grad ? ", double* lit" : "") << ") {\n";
    if (grad) emit_scatter_prologue(os, b, L, pi, full_wave);
    else emit_coo_prologue(os, b, L, pi, tile);
    emit_lines(os, b.e);
    if (grad) { for (auto &s : stores) os << "    " << s << "\n"; }
    else emit_coo_stores(os, b, L.pat[pi].o1, p.o1step, vals, tile);
    os << "}\n";
}

// ---- gather ("pull") formulation of the objective gradient ------------------------------------------------
// An objective pattern can be gathered when every first-order slot's variable index is (range value) + c: variable
// v then receives slot s from exactly one data point, I = (v - c_s - start) / step.  One thread per VARIABLE
// re-evaluates the (cheap) pattern at those points: coalesced store, no zero-fill, no atomics, deterministic.
// (The reference's KA path resolves the same contention with a sorted gather list, KA ext :310-336.)
bool pull_ok(const Pattern &p, std::vector<Affine> &slots) {
    if (p.kind != EXA_PAT_OBJ || p.n == 0 || p.o1step < 1 || p.o1step > env_int("EXAHIP_PULL_MAX_SLOTS", 4)) return false;
    if (!env_int("EXAHIP_GRAD_PULL", 1)) return false;
    slots.clear();
    int col = -1;
    for (int s = 0; s < p.o1step; s++) {
        Affine a = affine(p, p.ad[p.slotvar1[s]].ir);
        if (!a.ok || a.col < 0 || a.a != 1) return false;
        if (col >= 0 && a.col != col) return false;
        col = a.col;
        if (p.cols[col].step < 1) return false;
        slots.push_back(a);
    }
    return true;
}

void gen_pull_fn(std::ostringstream &os, const Model &m, int pi, const ParamLayout &L) {
    const Pattern &p = m.pats[pi];
    std::vector<Affine> slots;
    pull_ok(p, slots);
    os << "static __device__ __forceinline__ double " << fn_name(pi, "pull")
       << "(const long* __restrict__ P, const double* __restrict__ x, const double* __restrict__ th, long v) {\n    double g = 0.0;\n";
    // every slot is evaluated unconditionally at an index clamped into the shard and its contribution selected
    // afterwards: no data-dependent branch, so a thread handling several variables has all its loads in flight at once
    os << "    const long lo_ = " << Body(m, pi, L).P(L.pat[pi].lo) << ", hi_ = " << Body(m, pi, L).P(L.pat[pi].hi) << ";\n"
       << "    if (lo_ >= hi_) return 0.0;\n";
    for (int s = 0; s < p.o1step; s++) {
        Body b(m, pi, L);
        b.forward(p.ad_root, 1, false);
        GenAlg a(b, p.comp1, p.o1step);
        grpass(p, p.ad_root, a, Emitter::litf(1.0));
        const int64_t step = p.cols[slots[s].col].step;
        os << "    {   // slot " << s << ": x[range" << (slots[s].c >= 0 ? " + " : " - ") << std::llabs(slots[s].c) << "]\n"
           << "        const long r = v - (" << slots[s].c << "L) - " << b.P(L.pat[pi].col[slots[s].col]) << ";\n"
           << "        const long J = r / " << step << "L;\n"
           << "        const bool ok = r >= 0 && J * " << step << "L == r && J >= lo_ && J < hi_;\n"
           << "        const long I = J < lo_ ? lo_ : (J >= hi_ ? hi_ - 1 : J);\n";
        emit_lines(os, b.e, "        ");
        os << "        g += ok ? " << b.e.sd(a.acc[s]) << " : 0.0;\n    }\n";
    }
    os << "    return g;\n}\n";
}

// COO-writing pattern function in two stages (see split_body): pK_<cb>L loads, pK_<cb>E evaluates and stores.
void emit_two_stage(std::ostringstream &os, Body &b, const ParamLayout &L, int pi, int cb, const char *name, bool hess, bool tile,
                    int word_o, int S, const std::vector<std::string> &vals) {
    const Split sp = split_body(b.e);
    g_handover[{cb, pi}] = {sp.nin, sp.nik};
    os << "static __device__ __forceinline__ void " << fn_name(pi, name) << "L(const long* __restrict__ P, const double* __restrict__ x, "
       << (hess ? "const double* __restrict__ y, " : "") << "const double* __restrict__ th, long tid, double* in, long* ik) {\n"
       << "    const long I0 = " << b.P(L.pat[pi].lo) << " + tid;\n    const long hi = " << b.P(L.pat[pi].hi) << ";\n"
       // no branch: lanes (and whole tiles) beyond the shard re-read its last point — or point 0 when the shard holds
       // nothing of this pattern (active patterns have n >= 1; exa_shard_var_range counts that point in)
       << "    const long I = I0 < hi ? I0 : (hi > 0 ? hi - 1 : 0);\n";
    for (const auto &l : sp.load) os << "    " << l << "\n";
    os << "}\n";
    os << "static __device__ __forceinline__ void " << fn_name(pi, name) << "E(const long* __restrict__ P, const double* in, const long* ik, "
       << "double* __restrict__ out, double* __restrict__ sink, " << (hess ? "double sigma, " : "") << "long tid, double* lds) {\n"
       << "    const long I0 = " << b.P(L.pat[pi].lo) << " + tid;\n    const long hi = " << b.P(L.pat[pi].hi) << ";\n"
       << "    const int lane = threadIdx.x & 63;\n    const long I = I0 < hi ? I0 : (hi > 0 ? hi - 1 : 0);\n";
    for (const auto &l : sp.eval) os << "    " << l << "\n";
    if (env_int("EXAHIP_NB", 1)) emit_coo_stores(os, b, word_o, S, vals, tile, "out", "", true);
    else { os << "    const long npts_ = hi - (I0 - lane); (void)npts_;\n"; emit_coo_stores(os, b, word_o, S, vals, tile); }
    os << "}\n";
}

void gen_hess_fn(std::ostringstream &os, const Model &m, int pi, const ParamLayout &L) {
    Body b(m, pi, L);
    const Pattern &p = b.p;
    b.forward(p.ad_root, 2, false);
    Val adj;
    if (p.kind == EXA_PAT_OBJ) adj = b.e.raw("sigma", false);
    else adj = b.e.raw("y[" + b.row0() + "]", false);
    GenAlg a(b, p.comp2, p.o2step);
    hrpass0(p, p.ad_root, a, adj, zero_seed(b));
    const bool tile = use_tile(p.o2step);
    std::vector<std::string> vals;
    for (int s = 0; s < p.o2step; s++) vals.push_back(b.e.sd(a.acc[s]));
    if (L.chain[CB_HESSC] > 0) emit_two_stage(os, b, L, pi, CB_HESSC, "hessc", true, tile, L.pat[pi].o2, p.o2step, vals);
}

// jac_coord! / hess_coord! (exa_jac / exa_hess): one device function per FUSED GROUP — the patterns of exactly the same
// length, evaluated by thread I one after the other inside ONE emitter (shared loads, shared gathers, one sincos per
// argument for all of them), each pattern's slots staged and flushed to ITS OWN contiguous COO range as soon as they are
// complete (so only one pattern's values are live at a time).  A singleton group is the plain per-pattern function.
void gen_coo_group_fn(std::ostringstream &os, const Model &m, const ParamLayout &L, int cb, int gi, bool permuted = false) {
    const auto &grp = L.groups[cb][gi];
    const bool hess = cb == CB_HESS;
    bool any_tile = false;
    for (int pk : grp) any_tile = any_tile || (!permuted && use_tile(hess ? m.pats[pk].o2step : m.pats[pk].o1step));
    os << "static __device__ __forceinline__ void g" << gi << "_" << (hess ? "hess" : "jac") << (permuted ? "p" : "")
       << "(const long* __restrict__ P, const double* __restrict__ x, " << (hess ? "const double* __restrict__ y, " : "")
       << "const double* __restrict__ th, double* __restrict__ out, " << (hess ? "double sigma, " : "") << "long tid, "
       << (permuted ? "const unsigned* __restrict__ pos" : "double* lds") << ") {\n";
    {
        Body b0(m, grp.front(), L);
        emit_coo_prologue(os, b0, L, grp.front(), any_tile);      // the group shares lo / hi
    }
    Emitter E;
    size_t emitted = 0;
    for (int pk : grp) {
        Body b(m, pk, L, &E);
        const Pattern &p = b.p;
        std::vector<std::string> vals;
        int S, word;
        if (hess) {
            b.forward(p.ad_root, 2, false);
            Val adj = p.kind == EXA_PAT_OBJ ? E.raw("sigma", false) : E.raw("y[" + b.row0() + "]", false);
            GenAlg a(b, p.comp2, p.o2step);
            hrpass0(p, p.ad_root, a, adj, zero_seed(b));
            S = p.o2step; word = L.pat[pk].o2;
            for (int s = 0; s < S; s++) vals.push_back(E.sd(a.acc[s]));
        } else {
            b.forward(p.ad_root, 1, false);
            GenAlg a(b, p.comp1, p.o1step);
            grpass(p, p.ad_root, a, Emitter::litf(1.0));
            S = p.o1step; word = L.pat[pk].o1;
            for (int s = 0; s < S; s++) vals.push_back(E.sd(a.acc[s]));
        }
        for (; emitted < E.lines.size(); emitted++) os << "    " << E.lines[emitted] << "\n";
        if (permuted) emit_coo_stores_permuted(os, b, word, S, vals, "_" + std::to_string(pk));
        else emit_coo_stores(os, b, word, S, vals, use_tile(S), "out", "_" + std::to_string(pk));
    }
    os << "}\n";
}

// ---- fused sweep: value + Jacobian slots + Hessian slots from ONE second-order forward sweep --------
// A solver iteration asks for cons!, jac_coord! and hess_coord! at the same x; the second-order forward sweep already
// holds the value and the first partials (graph.jl:416-447), so one kernel emits c, J and H (and the objective
// partial sums) and the transcendental work is done once instead of three times.
// One device function per fused group (objective patterns stay alone: their workgroups also produce the partial sums of
// obj): thread I evaluates every pattern of the group in one emitter, storing each pattern's row value, Jacobian slots
// and Hessian slots as soon as they are complete.
void gen_fused_group_fn(std::ostringstream &os, const Model &m, const ParamLayout &L, int gi) {
    const auto &grp = L.groups[CB_FUSED][gi];
    os << "static __device__ __forceinline__ double g" << gi << "_fused"
       << "(const long* __restrict__ P, const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ th, "
          "double* __restrict__ cout, double* __restrict__ augout, double* __restrict__ jout, double* __restrict__ hout, double sigma, "
          "long tid, double* lds, const long* __restrict__ augptr, const long* __restrict__ augsrc, const double* __restrict__ augcoef) {\n";
    {
        Body b0(m, grp.front(), L);
        os << "    const long I0 = " << b0.P(L.pat[grp.front()].lo) << " + tid;\n    const long hi = " << b0.P(L.pat[grp.front()].hi) << ";\n"
           << "    const int lane = threadIdx.x & 63;\n    if (I0 - lane >= hi) return 0.0;\n    const long I = I0 < hi ? I0 : hi - 1;\n";
    }
    Emitter E;
    size_t emitted = 0;
    std::string ret = "0.0";
    for (int pk : grp) {
        Body b(m, pk, L, &E);
        const Pattern &p = b.p;
        b.forward(p.ad_root, 2, false);
        Val value = E.tod(b.fv[p.ad_root].x);
        if (p.ad[p.ad_root].kind == AD_CONST) value = E.tod(b.cval(p.root));
        const bool isobj = p.kind == EXA_PAT_OBJ;
        GenAlg a1(b, p.comp1, p.o1step);
        if (!isobj && p.o1step > 0) grpass(p, p.ad_root, a1, Emitter::litf(1.0));
        Val adj = isobj ? E.raw("sigma", false) : E.raw("y[" + b.row0() + "]", false);
        GenAlg a2(b, p.comp2, p.o2step);
        if (p.o2step > 0) hrpass0(p, p.ad_root, a2, adj, zero_seed(b));
        const std::string rowtxt = isobj ? "" : (p.kind == EXA_PAT_CONAUG ? b.P(L.pat[pk].oa) + " + I" : b.P(L.pat[pk].o0) + " + I");
        for (; emitted < E.lines.size(); emitted++) os << "    " << E.lines[emitted] << "\n";
        // Linear augmentation terms (c * x[k], evaluated at build): with the row lists of exa_cons1 at hand (augptr != null:
        // unsharded, no long rows) the thread that owns a base row adds the row's terms itself, in list order — no value
        // buffer, no exa_aug_gather launch behind the sweep.  Otherwise the terms' values go to the buffer as before.
        bool target = false;
        for (const Pattern &q : m.pats) target = target || (q.kind == EXA_PAT_CONAUG && q.n > 0 && q.base == pk);
        if (isobj) {}
        else if (m.aug_linear && p.kind == EXA_PAT_CONAUG) os << "    if (I0 < hi && !augptr) augout[" << rowtxt << "] = " << E.sd(value) << ";\n";
        else if (m.aug_linear && target)
            os << "    if (I0 < hi) {\n        double v = " << E.sd(value) << ";\n        const long r_ = " << rowtxt << ";\n"
                  "        if (augptr) {\n            long j = augptr[r_];\n            const long je = augptr[r_ + 1];\n"
                  "            for (; j + 4 <= je; j += 4) {\n"
                  "                const long i0 = augsrc[j], i1 = augsrc[j + 1], i2 = augsrc[j + 2], i3 = augsrc[j + 3];\n"
                  "                const double c0 = augcoef[j], c1 = augcoef[j + 1], c2 = augcoef[j + 2], c3 = augcoef[j + 3];\n"
                  "                const double x0 = x[i0], x1 = x[i1], x2 = x[i2], x3 = x[i3];\n"
                  "                v += __dmul_rn(c0, x0); v += __dmul_rn(c1, x1); v += __dmul_rn(c2, x2); v += __dmul_rn(c3, x3);\n            }\n"
                  "            for (; j < je; j++) v += __dmul_rn(augcoef[j], x[augsrc[j]]);\n        }\n        cout[r_] = v;\n    }\n";
        else os << "    if (I0 < hi) " << (p.kind == EXA_PAT_CONAUG ? "augout" : "cout") << "[" << rowtxt << "] = " << E.sd(value) << ";\n";
        const std::string tag = "_" + std::to_string(pk);
        if (!isobj && p.o1step > 0) {
            std::vector<std::string> vals;
            for (int s = 0; s < p.o1step; s++) vals.push_back(E.sd(a1.acc[s]));
            emit_coo_stores(os, b, L.pat[pk].o1, p.o1step, vals, use_tile(p.o1step), "jout", "j" + tag);
        }
        if (p.o2step > 0) {
            std::vector<std::string> vals;
            for (int s = 0; s < p.o2step; s++) vals.push_back(E.sd(a2.acc[s]));
            emit_coo_stores(os, b, L.pat[pk].o2, p.o2step, vals, use_tile(p.o2step), "hout", "h" + tag);
        }
        if (isobj) ret = "(I0 < hi ? " + E.sd(value) + " : 0.0)";
    }
    os << "    return " << ret << ";\n}\n";
}

// ---- matrix-free products: same sweeps, different leaf actions ---------------------------------
// Jv: row value = sum_s acc_s * v[k_s] (jacobian.jl:41-54) — a base row is owned by one data point (plain store),
// augmentation terms go through the value buffer + exa_aug_gather like cons_nln!.
void gen_jprod_fn(std::ostringstream &os, const Model &m, int pi, const ParamLayout &L) {
    Body b(m, pi, L);
    const Pattern &p = b.p;
    Val sum = Emitter::litf(0.0);
    if (p.o1step > 0) {
        b.forward(p.ad_root, 1, false);
        GenAlg a(b, p.comp1, p.o1step);
        grpass(p, p.ad_root, a, Emitter::litf(1.0));
        for (int s = 0; s < p.o1step; s++) {
            Val vi = b.fv[p.slotvar1[s]].vidx;
            Val vv = b.e.raw("v[" + b.e.s(b.e.sub(vi, Emitter::liti(1))) + "]", false);
            sum = b.e.add(sum, b.e.mul(a.acc[s], vv));
        }
    }
    const std::string dst = p.kind == EXA_PAT_CONAUG ? "aug[" + b.P(L.pat[pi].oa) + " + I]" : "out[" + b.P(L.pat[pi].o0) + " + I]";
    os << "static __device__ __forceinline__ void " << fn_name(pi, "jprod")
       << "(const long* __restrict__ P, const double* __restrict__ x, const double* __restrict__ th, const double* __restrict__ v, "
          "double* __restrict__ out, double* __restrict__ aug, long tid) {\n"
       << "    const long I = " << b.P(L.pat[pi].lo) << " + tid;\n    if (I >= " << b.P(L.pat[pi].hi) << ") return;\n";
    emit_lines(os, b.e);
    os << "    " << dst << " = " << b.e.sd(sum) << ";\n}\n";
}

// J'v: out[k_s] += acc_s * v[row] (jacobian.jl:55-68) — shared targets, FP64 hardware atomics on a zeroed vector
void jtprod_items(Body &b, Scatter &sc) {
    const Pattern &p = b.p;
    b.forward(p.ad_root, 1, false);
    GenAlg a(b, p.comp1, p.o1step);
    grpass(p, p.ad_root, a, Emitter::litf(1.0));
    Val w = b.e.raw("v[" + b.row0() + "]", false);
    for (int s = 0; s < p.o1step; s++) sc.add(b, p.slotvar1[s], b.e.mul(a.acc[s], w));
}

// Hv: for a lower-triangular COO entry (i, j, A): i == j -> Hv[i] += A v[i]; else Hv[i] += A v[j], Hv[j] += A v[i]
// (hessian.jl:291-315, 566-579).  Contributions are merged per variable in registers first: one atomic per
// distinct variable of the data point instead of one or two per slot.
void hprod_items(Body &b, Scatter &sc) {
    const Pattern &p = b.p;
    b.forward(p.ad_root, 2, false);
    Val adj;
    if (p.kind == EXA_PAT_OBJ) adj = b.e.raw("sigma", false);
    else adj = b.e.raw("y[" + b.row0() + "]", false);
    GenAlg a(b, p.comp2, p.o2step);
    hrpass0(p, p.ad_root, a, adj, zero_seed(b));
    const int nk = (int)p.keys.size();
    std::vector<int> rep(nk, -1);
    for (size_t n = 0; n < p.ad.size(); n++)
        if (p.ad[n].kind == AD_VAR && rep[p.ad[n].key] < 0) rep[p.ad[n].key] = (int)n;
    std::vector<Val> hv(nk, Emitter::litf(0.0));
    std::vector<char> used(nk, 0);
    auto vload = [&](int key) {
        Val vi = b.fv[rep[key]].vidx;
        return b.e.raw("v[" + b.e.s(b.e.sub(vi, Emitter::liti(1))) + "]", false);
    };
    for (int s = 0; s < p.o2step; s++) {
        const int k1 = p.ad[p.slotvar2[s].first].key, k2 = p.ad[p.slotvar2[s].second].key;
        Val A = a.acc[s];
        if (A.lit_eq(0)) continue;
        if (k1 == k2) {
            hv[k1] = b.e.add(hv[k1], b.e.mul(A, vload(k1)));
            used[k1] = 1;
        } else {
            // A already carries the i == j ? 2adj : adj rule; when the two keys alias at run time only one update applies
            Val i = b.fv[rep[k1]].vidx, j = b.fv[rep[k2]].vidx;
            hv[k1] = b.e.add(hv[k1], b.e.mul(A, vload(k2)));
            Val second = b.e.mul(A, vload(k1));
            if (!(i.is_lit() && j.is_lit()))
                second = b.e.raw("(" + b.e.s(i) + " == " + b.e.s(j) + " ? 0.0 : " + b.e.sd(second) + ")", false);
            else if (i.i == j.i) second = Emitter::litf(0.0);
            hv[k2] = b.e.add(hv[k2], second);
            used[k1] = used[k2] = 1;
        }
    }
    for (int k = 0; k < nk; k++)
        if (used[k]) sc.add(b, rep[k], hv[k]);
}

// One device function per GROUP of a scattering product (J'v / Hv): the patterns of a group iterate over the same data
// points (equal length, same shard), thread I evaluates ALL of them at point I inside one emitter — loads of aliased
// table columns, gathers of x and common subexpressions (one sincos(va_f - va_t) for the four branch flows) are shared —
// and their contributions are merged per target before anything is added to memory (Scatter::merge).
void gen_scatter_group_fn(std::ostringstream &os, const Model &m, const ParamLayout &L, int cb, int gi) {
    const auto &grp = L.groups[cb][gi];
    const bool hp = cb == CB_HPROD;
    Emitter E;
    Scatter sc(E, L);
    sc.loopfree = g_loopfree[cb];
    std::vector<std::unique_ptr<Body>> bodies;
    for (int pk : grp) {
        bodies.emplace_back(new Body(m, pk, L, &E));
        if (hp) hprod_items(*bodies.back(), sc); else jtprod_items(*bodies.back(), sc);
    }
    std::vector<std::string> stores;
    bool full_wave = false;
    g_lds_need[cb] = std::max(g_lds_need[cb], sc.emit(stores, full_wave));
    g_lit_idx[{cb, gi}] = sc.lit_idx;
    g_scatter_lines[cb] = std::max(g_scatter_lines[cb], E.lines.size());
    const char *name = hp ? "hprod" : "jtprod";
    os << "static __device__ __forceinline__ void g" << gi << "_" << name
       << "(const long* __restrict__ P, const double* __restrict__ x, " << (hp ? "const double* __restrict__ y, " : "")
       << "const double* __restrict__ th, const double* __restrict__ v, double* __restrict__ out, " << (hp ? "double sigma, " : "")
       << "long tid, double* lds, double* lit) {\n";
    emit_scatter_prologue(os, *bodies.front(), L, grp.front(), full_wave);     // the group shares lo / hi
    emit_lines(os, E);
    for (auto &st : stores) os << "    " << st << "\n";
    os << "}\n";
}

void gen_struct_fn(std::ostringstream &os, const Model &m, int pi, const ParamLayout &L, bool hess) {
    Body b(m, pi, L);
    const Pattern &p = b.p;
    b.forward(p.ad_root, 0, true);
    std::vector<std::string> stores;
    if (hess) {
        for (int s = 0; s < p.o2step; s++) {
            Val i = b.fv[p.slotvar2[s].first].vidx, j = b.fv[p.slotvar2[s].second].vidx;
            const std::string si = b.e.s(i), sj = b.e.s(j);
            // lower triangle: (max, min) (hessian.jl:622-642)
            stores.push_back("rows[o + " + std::to_string(s) + "] = (IT)(" + si + " >= " + sj + " ? " + si + " : " + sj + "); cols[o + " +
                             std::to_string(s) + "] = (IT)(" + si + " >= " + sj + " ? " + sj + " : " + si + ");");
        }
    } else {
        const std::string row = b.row0();
        for (int s = 0; s < p.o1step; s++) {
            Val i = b.fv[p.slotvar1[s]].vidx;
            stores.push_back("rows[o + " + std::to_string(s) + "] = (IT)(" + row + " + 1); cols[o + " + std::to_string(s) + "] = (IT)(" + b.e.s(i) + ");");
        }
    }
    os << "template <typename IT> static __device__ __forceinline__ void " << fn_name(pi, hess ? "hst" : "jst")
       << "(const long* __restrict__ P, IT* __restrict__ rows, IT* __restrict__ cols, long tid) {\n"
       << "    const long I = " << b.P(L.pat[pi].lo) << " + tid;\n    if (I >= " << b.P(L.pat[pi].hi) << ") return;\n";
    emit_lines(os, b.e);
    os << "    const long o = " << b.P(hess ? L.pat[pi].o2 : L.pat[pi].o1) << " + " << (hess ? p.o2step : p.o1step) << "L * I;\n";
    for (auto &s : stores) os << "    " << s << "\n";
    os << "}\n";
}

// grad! by the reference's scheme (KA ext :310-336): the first partials of every objective pattern go to their slots of a
// gradient COO (ExaCore.nnzg entries, slot o1 + o1step * I + s), pK_gst names the variable of each slot; the runtime sorts
// (variable, slot) once and adds each variable's slots in slot order — deterministic, and one variable shared by millions
// of data points is summed cooperatively instead of by millions of atomics on one cache line.
void gen_gradv_fn(std::ostringstream &os, const Model &m, int pi, const ParamLayout &L) {
    Body b(m, pi, L);
    const Pattern &p = b.p;
    b.forward(p.ad_root, 1, false);
    GenAlg a(b, p.comp1, p.o1step);
    grpass(p, p.ad_root, a, Emitter::litf(1.0));
    os << "static __device__ __forceinline__ void " << fn_name(pi, "gradv")
       << "(const long* __restrict__ P, const double* __restrict__ x, const double* __restrict__ th, double* __restrict__ gout, long I) {\n";
    emit_lines(os, b.e);
    os << "    const long o = " << b.P(L.pat[pi].o1) << " + " << p.o1step << "L * I;\n";
    for (int s = 0; s < p.o1step; s++) os << "    gout[o + " << s << "] = " << b.e.sd(a.acc[s]) << ";\n";
    os << "}\n";
    Body c(m, pi, L);
    c.forward(p.ad_root, 0, true);
    os << "static __device__ __forceinline__ void " << fn_name(pi, "gst") << "(const long* __restrict__ P, long* __restrict__ cols, long I) {\n";
    emit_lines(os, c.e);
    os << "    const long o = " << c.P(L.pat[pi].o1) << " + " << p.o1step << "L * I;\n";
    for (int s = 0; s < p.o1step; s++) os << "    cols[o + " << s << "] = " << c.e.s(c.fv[p.slotvar1[s]].vidx) << ";\n";
    os << "}\n";
}

// ---- fused kernels: blockIdx -> (pattern, tile) ------------------------------------------------------
void gen_dispatch(std::ostringstream &os, const ParamLayout &L, int cb, const std::string &call_prefix, const std::string &call_args,
                  const std::string &tail_args = "") {
    const auto &act = L.active[cb];
    const int ppt = L.ppt[cb];
    if (env_int("EXAHIP_XCD_REMAP", 0))
        // workgroup b runs on XCD b % 8 (observed placement): give every XCD one contiguous range of tiles
        // (measured SLOWER than the default interleaving on all three configs; kept only as an experiment knob)
        os << "    const long nb_ = gridDim.x, q_ = nb_ >> 3, r_ = nb_ & 7, xcd_ = blockIdx.x & 7, i_ = blockIdx.x >> 3;\n"
              "    const long b = xcd_ * q_ + (xcd_ < r_ ? xcd_ : r_) + i_;\n";
    else
        os << "    const long b = blockIdx.x;\n";
    // block map: which pattern and which tile this workgroup evaluates (interleaved by the runtime so that patterns
    // reading the same x ranges run on the same XCD at about the same time)
    os << "    const long e_ = ((const long*)P[" << L.blk[cb] << "])[b];\n    const int ps_ = (int)(e_ >> 40);\n"
          "    const long tid0 = (e_ & ((1L << 40) - 1)) * (EXA_BLOCK * " << ppt << ") + threadIdx.x;\n";
    const bool scatter = cb == CB_GRAD || cb == CB_JTPROD || cb == CB_HPROD;
    const bool grouped = cb == CB_JTPROD || cb == CB_HPROD || cb == CB_JAC || cb == CB_HESS;     // dispatch units are fused groups
    const size_t nunits = grouped ? L.groups[cb].size() : act.size();
    auto unit_key = [&](size_t k) { return grouped ? (int)k : act[k]; };
    size_t maxlit = 0;
    if (scatter) for (size_t k = 0; k < nunits; k++) maxlit = std::max(maxlit, g_lit_idx[{cb, unit_key(k)}].size());
    if (scatter) os << "    double lit[" << std::max<size_t>(maxlit, 1) << "] = {0.0};\n";
    for (size_t k = 0; k < nunits; k++) {
        os << "    " << (k ? "else " : "") << "if (ps_ == " << k << ") {\n";
        // no unrolling for the scatter kernels: two inlined copies of a large Hessian body exhaust the register file
        // (512 VGPRs + scratch spills were observed, and a spilling exa_hprod produced wrong sums on gfx950)
        if (ppt > 1) os << "#pragma unroll " << (scatter ? 1 : 2) << "\n        for (int u = 0; u < " << ppt << "; u++) ";
        else os << "        { const int u = 0; ";
        os << (grouped ? "g" : "p") << unit_key(k) << "_" << call_prefix << "(" << call_args << ", tid0 + u * EXA_BLOCK" << tail_args << (scatter ? ", lit" : "") << ");"
           << (ppt > 1 ? "" : " }") << "\n";
        if (scatter) {
            // targets shared by all data points: one wavefront reduction + one atomic per wavefront AFTER the tile loop
            const auto &li = g_lit_idx[{cb, unit_key(k)}];
            for (size_t q = 0; q < li.size(); q++) os << "        exa_wave_atomic_add(&out[" << li[q] << "], lit[" << q << "]);\n";
        }
        os << "    }\n";
    }
}

// Chained dispatch (jac / hess), see ParamLayout::chain: entry = (group, first tile); T tiles, all patterns of the
// group per tile, the loads of the next (pattern, tile) issued before the current one is evaluated and stored.  Every
// pattern of a group keeps its own hand-over registers, so no value ever merges across patterns.
void gen_dispatch_chained(std::ostringstream &os, const ParamLayout &L, int cb, const char *name, bool hess) {
    const int T = L.chain[cb];
    const auto &groups = L.groups[cb];
    auto ld = [&](int pk, const std::string &tid, const std::string &sfx) {
        os << "p" << pk << "_" << name << "L(P, x, " << (hess ? "y, " : "") << "th, " << tid << ", in" << pk << sfx << ", ik" << pk << sfx << ");";
    };
    auto ev = [&](int pk, const std::string &tid) {
        os << "p" << pk << "_" << name << "E(P, in" << pk << ", ik" << pk << ", out, sink, " << (hess ? "sigma, " : "") << tid << ", lds);";
    };
    os << "    const long e_ = ((const long*)P[" << L.blk[cb] << "])[blockIdx.x];\n    const int gs_ = (int)(e_ >> 40);\n"
       << "    const long t0_ = (e_ & ((1L << 40) - 1)) * " << T << ";\n";
    for (size_t g = 0; g < groups.size(); g++) {
        const auto &grp = groups[g];
        const int first = grp.front(), last = grp.back();
        os << "    " << (g ? "else " : "") << "if (gs_ == " << g << ") {\n        const long tend_ = t0_ + " << T << " < P[" << L.gtiles[cb][g] << "] ? t0_ + " << T
           << " : P[" << L.gtiles[cb][g] << "];\n";
        for (int pk : grp) {
            const auto ho = g_handover[{cb, pk}];
            os << "        double in" << pk << "[" << std::max(1, ho.first) << "]; long ik" << pk << "[" << std::max(1, ho.second) << "];\n";
        }
        const auto h0 = g_handover[{cb, first}];
        os << "        double in" << first << "n[" << std::max(1, h0.first) << "]; long ik" << first << "n[" << std::max(1, h0.second) << "];\n        ";
        ld(first, "t0_ * EXA_BLOCK + threadIdx.x", "");
        // (claimed before the loop too: at the loop header the two incoming paths must agree that these loads are done)
        os << "\n#pragma unroll\n        for (int q = 0; q < " << std::max(1, h0.first) << "; q++) asm volatile(\"\" : \"+v\"(in" << first << "[q]));\n"
           << "#pragma unroll\n        for (int q = 0; q < " << std::max(1, h0.second) << "; q++) asm volatile(\"\" : \"+v\"(ik" << first << "[q]));";
        os << "\n#pragma unroll 1\n        for (long t = t0_; t < tend_; t++) {\n            const long tid = t * EXA_BLOCK + threadIdx.x;\n";
        // (the next tile's load is unconditional — the last tile loads itself again —: the number of memory instructions
        // per iteration is fixed)
        if (env_int("EXAHIP_CHAIN_EARLY", 1)) {
            // all loads of the iteration first — the other patterns of this tile AND the first pattern of the next tile —,
            // then all evaluations: every load has at least one evaluation's arithmetic to land in
            for (size_t j = 1; j < grp.size(); j++) { os << "            "; ld(grp[j], "tid", ""); os << "\n"; }
            os << "            "; ld(first, "(t + 1 < tend_ ? t + 1 : t) * EXA_BLOCK + threadIdx.x", "n"); os << "\n";
            for (size_t j = 0; j < grp.size(); j++) { os << "            "; ev(grp[j], "tid"); if (j + 1 < grp.size()) os << "\n"; }
        } else {
            for (size_t j = 1; j < grp.size(); j++) {
                os << "            "; ld(grp[j], "tid", ""); os << "\n            "; ev(grp[j - 1], "tid"); os << "\n";
            }
            os << "            "; ld(first, "(t + 1 < tend_ ? t + 1 : t) * EXA_BLOCK + threadIdx.x", "n"); os << "\n            "; ev(last, "tid");
        }
        // The hand-over registers are claimed HERE, at the bottom of the iteration, where the only memory instructions
        // issued after the loads are the fixed number of stores of the last pattern: the wait is vmcnt(#stores).  Left to
        // itself the compiler merges in*n into in* and waits at the loop header, where the first entry (no stores behind
        // its loads) forces vmcnt(0) — draining every store of the previous tile before the next evaluation starts.
        os << "\n#pragma unroll\n            for (int q = 0; q < " << std::max(1, h0.first) << "; q++) { asm volatile(\"\" : \"+v\"(in" << first << "n[q])); in" << first << "[q] = in" << first << "n[q]; }\n"
           << "#pragma unroll\n            for (int q = 0; q < " << std::max(1, h0.second) << "; q++) { asm volatile(\"\" : \"+v\"(ik" << first << "n[q])); ik" << first << "[q] = ik" << first << "n[q]; }\n        }\n    }\n";
    }
}

}  // namespace

Generated generate_module(const Model &m, bool loopfree_scatter) {
    // the scatter bookkeeping above (g_lds_need, g_lit_idx) is module-level state of one generation: serialise
    // concurrent model builds here (planning and hipcc still run in parallel)
    std::lock_guard<std::mutex> gen_lock(g_gen_mu);
    Generated g;
    ParamLayout &L = g.layout;
    const int np = (int)m.pats.size();
    int w = 0;
    L.pat.resize(np);
    for (int k = 0; k < np; k++) {
        auto &pp = L.pat[k];
        pp.lo = w++; pp.hi = w++; pp.o0 = w++; pp.o1 = w++; pp.o2 = w++; pp.oa = w++; pp.ob = w++;
        for (size_t c = 0; c < m.pats[k].cols.size(); c++) {
            const Column &col = m.pats[k].cols[c];
            pp.col.push_back(col.alias_pat >= 0 ? L.pat[col.alias_pat].col[col.alias_col] : w++);      // one word per DISTINCT column
        }
    }
    for (int k = 0; k < np; k++) {
        const Pattern &p = m.pats[k];
        if (p.n == 0) continue;
        if (p.kind == EXA_PAT_OBJ) {
            L.active[CB_OBJ].push_back(k);
            std::vector<Affine> sl;
            if (pull_ok(p, sl)) L.pull.push_back(k);
            else if (p.o1step > 0) L.active[CB_GRAD].push_back(k);
        } else {
            L.active[CB_CONS].push_back(k);      // base rows and augmentation terms share one launch
            if (p.kind == EXA_PAT_CON) L.active[CB_CONS1].push_back(k);
            L.active[CB_JPROD].push_back(k);
            if (p.o1step > 0) L.active[CB_JTPROD].push_back(k);
            if (p.o1step > 0) { L.active[CB_JAC].push_back(k); L.active[CB_JSTRUCT].push_back(k); }
        }
        if (p.o2step > 0) { L.active[CB_HESS].push_back(k); L.active[CB_HESSC].push_back(k); L.active[CB_HSTRUCT].push_back(k); L.active[CB_HPROD].push_back(k); }
        L.active[CB_FUSED].push_back(k);
    }
    for (int cb = 0; cb < CB_COUNT; cb++) { L.blk[cb] = w++; L.ppt[cb] = 1; L.chain[cb] = std::max(0, chain_len(cb)); }
    g_handover.clear();
    // groups of co-indexed patterns (iterator lengths within 2 of each other), in dispatch order.  The lengths are what
    // decides, so instances of a model family share one module unless two unrelated blocks happen to be equally long.
    for (int cb : {CB_HESSC}) {
        if (L.chain[cb] == 0) continue;
        const int gmax = std::max(1, env_int("EXAHIP_GROUP_MAX", 8));
        for (int k : L.active[cb]) {
            bool placed = false;
            if (env_int("EXAHIP_GROUP", 1))
                for (auto &g : L.groups[cb])
                    if ((int)g.size() < gmax && std::llabs(m.pats[g.front()].n - m.pats[k].n) <= 2) { g.push_back(k); placed = true; break; }
            if (!placed) L.groups[cb].push_back({k});
        }
        for (size_t g = 0; g < L.groups[cb].size(); g++) L.gtiles[cb].push_back(w++);
    }
    // fused groups of the scattering products and of the one-launch cons_nln!: patterns of EXACTLY the same length (one
    // thread evaluates point I of all)
    for (int cb : {CB_JTPROD, CB_HPROD, CB_CONS1, CB_JAC, CB_HESS, CB_FUSED}) {
        const int gmax = std::max(1, env_int("EXAHIP_GROUP_MAX", 8));
        for (int k : L.active[cb]) {
            bool placed = false;
            const bool coo = cb == CB_JAC || cb == CB_HESS || cb == CB_FUSED;
            // (fused sweep: an objective pattern stays alone — its workgroups also write the partial sums of obj)
            const bool alone = cb == CB_FUSED && m.pats[k].kind == EXA_PAT_OBJ;
            // (a group's body is the concatenation of its patterns' bodies: bounded by the slots it computes, so that a
            // model with many equally long wide patterns does not produce one register-starved monster)
            auto slots = [&](int q) { const Pattern &t = m.pats[q]; return cb == CB_JAC || cb == CB_JTPROD || cb == CB_CONS1 ? t.o1step : t.o1step + t.o2step; };
            const int cap = env_int("EXAHIP_GROUP_SLOTS", 128);
            // (the scattering products also by the RAW contributions their reverse sweeps walk — what the body's length
            // follows: fusing is for small bodies that share loads (ACOPF's branch rows: 48 first-order / 90 second-order
            // contributions per group; the rocket: 25 / 103); bodies of thousands of SSA values gain nothing from it and,
            // fused, were miscompiled by the hiprtc of ROCm 7.0 once wavefront operations sat in them — wrong entries of
            // J'v / Hv in 4 of 60 random depth-6 models, none with the patterns on their own)
            auto raw = [&](int q) { const Pattern &t = m.pats[q]; return (int)(cb == CB_JTPROD ? t.comp1.size() : cb == CB_HPROD ? t.comp2.size() : 0); };
            const int rawcap = cb == CB_JTPROD ? env_int("EXAHIP_GROUP_RAW1", 64) : env_int("EXAHIP_GROUP_RAW2", 160);
            if (!alone && env_int(coo ? "EXAHIP_GROUP_COO" : "EXAHIP_GROUP_SCATTER", 1))
                for (auto &g : L.groups[cb]) {
                    int have = 0, have_raw = 0;
                    for (int q : g) { have += slots(q); have_raw += raw(q); }
                    if ((int)g.size() < gmax && have + slots(k) <= cap && have_raw + raw(k) <= rawcap && m.pats[g.front()].n == m.pats[k].n &&
                        !(cb == CB_FUSED && m.pats[g.front()].kind == EXA_PAT_OBJ)) {
                        g.push_back(k); placed = true; break;
                    }
                }
            if (!placed) L.groups[cb].push_back({k});
        }
    }

    // streaming value kernels keep more loads in flight per wavefront with several points per thread (measured)
    L.ppt[CB_OBJ] = env_int("EXAHIP_PPT_OBJ", 8);
    L.ppt[CB_CONS] = env_int("EXAHIP_PPT_CONS", 1);
    L.ppt[CB_GRAD] = env_int("EXAHIP_PPT_GRAD", 1);
    L.ppt[CB_JAC] = env_int("EXAHIP_PPT_JAC", 1);
    L.ppt[CB_HESS] = env_int("EXAHIP_PPT_HESS", 1);
    L.nwords = w;

    for (int &v : g_lds_need) v = 0;
    g_lit_idx.clear();
    g_scatter_lines.clear();
    std::ostringstream os;
    L.pull_ppt = std::max(1, env_int("EXAHIP_PPT_PULL", 2));
    {
        std::string pre = kPrelude;
        const std::string tag = "@BLOCK@", tag2 = "@PULLPPT@";
        pre.replace(pre.find(tag), tag.size(), std::to_string(kBlock));
        pre.replace(pre.find(tag2), tag2.size(), std::to_string(L.pull_ppt));
        os << pre;
    }
    os << "// patterns=" << np << " (sizes, offsets and column pointers are run-time parameters in P[])\n";
    if (loopfree_scatter) os << "// scatter kernels without loops: the first build of this module spilled registers there\n";
    {
        // dry pass over the scatter bodies: which callbacks hold a huge body (g_scatter_lines) and must be generated
        // without loops; its output and bookkeeping are discarded
        for (int cb = 0; cb < CB_COUNT; cb++) g_loopfree[cb] = false;
        std::ostringstream dry;
        for (int k = 0; k < np; k++) {
            const Pattern &p = m.pats[k];
            if (p.n > 0 && p.kind == EXA_PAT_OBJ && p.o1step > 0 && std::find(L.pull.begin(), L.pull.end(), k) == L.pull.end()) gen_first_fn(dry, m, k, L, true);
        }
        for (int cb : {CB_JTPROD, CB_HPROD})
            for (size_t g = 0; g < L.groups[cb].size(); g++) gen_scatter_group_fn(dry, m, L, cb, (int)g);
        for (int cb : {CB_GRAD, CB_JTPROD, CB_HPROD}) g_loopfree[cb] = loopfree_scatter || (int)g_scatter_lines[cb] > huge_body();
        g_lit_idx.clear();
        g_scatter_lines.clear();
        for (int cb = 0; cb < CB_COUNT; cb++) g_lds_need[cb] = 0;
    }
    for (int k = 0; k < np; k++) {
        const Pattern &p = m.pats[k];
        if (p.n == 0) continue;
        os << "// ---- pattern " << k << ": kind=" << p.kind << " o1step=" << p.o1step << " o2step=" << p.o2step << " ----\n";
        gen_value_fn(os, m, k, L);
        if (p.kind == EXA_PAT_OBJ) {
            if (std::find(L.pull.begin(), L.pull.end(), k) != L.pull.end()) gen_pull_fn(os, m, k, L);
            else if (p.o1step > 0) gen_first_fn(os, m, k, L, true);
            gen_gradv_fn(os, m, k, L);
        }
        else {
            gen_cons_fn(os, m, k, L);
            gen_jprod_fn(os, m, k, L);
            if (p.o1step > 0) gen_struct_fn(os, m, k, L, false);
        }
        if (p.o2step > 0) { gen_hess_fn(os, m, k, L); gen_struct_fn(os, m, k, L, true); }
    }
    for (int cb : {CB_JTPROD, CB_HPROD})
        for (size_t g = 0; g < L.groups[cb].size(); g++) gen_scatter_group_fn(os, m, L, cb, (int)g);
    for (int cb : {CB_JAC, CB_HESS})
        for (size_t g = 0; g < L.groups[cb].size(); g++) gen_coo_group_fn(os, m, L, cb, (int)g);
    for (size_t g = 0; g < L.groups[CB_FUSED].size(); g++) gen_fused_group_fn(os, m, L, (int)g);
    // scatter kernels whose patterns have targets shared by ALL data points process 16 tiles per workgroup: the shared
    // target then receives one atomic per wavefront per 16 tiles (same-address atomics serialise chip-wide at ~10 ns:
    // the rocket's step variable took 47 000 of them per J'v, 0.47 ms)
    for (int cb : {CB_GRAD, CB_JTPROD, CB_HPROD}) {
        bool any = false;
        for (const auto &kv : g_lit_idx) any = any || (kv.first.first == cb && !kv.second.empty());
        if (any && !g_loopfree[cb]) L.ppt[cb] = env_int("EXAHIP_PPT_LITERAL", 16);
    }
    // obj: per-workgroup partial sums
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_obj(const long* __restrict__ P, const double* __restrict__ x, "
          "const double* __restrict__ th, double* __restrict__ part) {\n    const long b = blockIdx.x;\n    double v = 0.0;\n";
    {
        const auto &act = L.active[CB_OBJ];
        const int ppt = L.ppt[CB_OBJ];
        os << "    const long e_ = ((const long*)P[" << L.blk[CB_OBJ] << "])[b];\n    const int ps_ = (int)(e_ >> 40);\n"
              "    const long t0_ = (e_ & ((1L << 40) - 1)) * (EXA_BLOCK * " << ppt << ") + threadIdx.x;\n";
        for (size_t k = 0; k < act.size(); k++) {
            const auto &pp = L.pat[act[k]];
            os << "    " << (k ? "else " : "") << "if (ps_ == " << k << ") {\n        const long I0 = P[" << pp.lo << "] + t0_;\n#pragma unroll\n"
               << "        for (int u = 0; u < " << ppt << "; u++) { const long I = I0 + u * EXA_BLOCK, h_ = P[" << pp.hi << "] - 1; "
               << "const double t_ = p" << act[k] << "_val(P, x, th, I < h_ ? I : h_); v += I <= h_ ? t_ : 0.0; }\n    }\n";
        }
    }
    os << "    const double s = exa_block_sum(v);\n    if (threadIdx.x == 0) part[b] = s;\n}\n";
    // gradient COO + its structure (sorted grad!, gen_gradv_fn): the dispatch of exa_obj
    for (int which = 0; which < 2; which++) {
        const auto &act = L.active[CB_OBJ];
        const int ppt = L.ppt[CB_OBJ];
        if (which == 0)
            os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_gradv(const long* __restrict__ P, const double* __restrict__ x, "
                  "const double* __restrict__ th, double* __restrict__ gout) {\n";
        else
            os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_gstruct(const long* __restrict__ P, long* __restrict__ cols) {\n";
        os << "    const long e_ = ((const long*)P[" << L.blk[CB_OBJ] << "])[blockIdx.x];\n    const int ps_ = (int)(e_ >> 40);\n"
              "    const long t0_ = (e_ & ((1L << 40) - 1)) * (EXA_BLOCK * " << ppt << ") + threadIdx.x;\n";
        for (size_t k = 0; k < act.size(); k++) {
            const auto &pp = L.pat[act[k]];
            os << "    " << (k ? "else " : "") << "if (ps_ == " << k << ") {\n#pragma unroll 1\n        for (int u = 0; u < " << ppt
               << "; u++) { const long I = P[" << pp.lo << "] + t0_ + u * EXA_BLOCK; if (I < P[" << pp.hi << "]) "
               << (which == 0 ? fn_name(act[k], "gradv") + "(P, x, th, gout, I)" : fn_name(act[k], "gst") + "(P, cols, I)") << "; }\n    }\n";
        }
        if (act.empty()) os << "    (void)ps_; (void)t0_;\n";
        os << "}\n";
    }
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_grad(const long* __restrict__ P, const double* __restrict__ x, "
          "const double* __restrict__ th, double* __restrict__ out) {\n";
    auto scatter_lds = [&](int cb) {
        if (g_lds_need[cb]) os << "    __shared__ double lds_all[(EXA_BLOCK / 64) * " << g_lds_need[cb] << "];\n    double* lds = lds_all + (threadIdx.x >> 6) * "
                               << g_lds_need[cb] << ";\n";
        else os << "    double* lds = nullptr;\n";
    };
    scatter_lds(CB_GRAD);
    gen_dispatch(os, L, CB_GRAD, "grad", "P, x, th, out", ", lds");
    os << "}\n";
    // grad!, gather part: one thread per variable; also provides the zero of untouched variables (no memset)
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_grad_pull(const long* __restrict__ P, const double* __restrict__ x, "
          "const double* __restrict__ th, double* __restrict__ out, long nvar) {\n"
          "    const long v0 = (long)blockIdx.x * (EXA_BLOCK * EXA_PULL_PPT) + threadIdx.x;\n    double g[EXA_PULL_PPT];\n"
          "#pragma unroll\n    for (int u = 0; u < EXA_PULL_PPT; u++) {\n        const long v_ = v0 + u * EXA_BLOCK, v = v_ < nvar ? v_ : nvar - 1;\n        g[u] = 0.0;\n";
    for (int k : L.pull) os << "        g[u] += p" << k << "_pull(P, x, th, v + 1);\n";
    os << "    }\n#pragma unroll\n    for (int u = 0; u < EXA_PULL_PPT; u++) { const long v = v0 + u * EXA_BLOCK; if (v < nvar) out[v] = g[u]; }\n}\n";
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_cons(const long* __restrict__ P, const double* __restrict__ x, "
          "const double* __restrict__ th, double* __restrict__ out, double* __restrict__ aug) {\n";
    {
        const auto &act = L.active[CB_CONS];
        const int ppt = L.ppt[CB_CONS];
        os << "    const long e_ = ((const long*)P[" << L.blk[CB_CONS] << "])[blockIdx.x];\n    const int ps_ = (int)(e_ >> 40);\n"
              "    const long tid0 = (e_ & ((1L << 40) - 1)) * (EXA_BLOCK * " << ppt << ") + threadIdx.x;\n    double v_[" << ppt << "];\n";
        for (size_t k = 0; k < act.size(); k++) {
            os << "    " << (k ? "else " : "") << "if (ps_ == " << k << ") {\n#pragma unroll\n        for (int u = 0; u < " << ppt
               << "; u++) v_[u] = p" << act[k] << "_consv(P, x, th, tid0 + u * EXA_BLOCK);\n#pragma unroll\n        for (int u = 0; u < " << ppt
               << "; u++) p" << act[k] << "_conss(P, out, aug, tid0 + u * EXA_BLOCK, v_[u]);\n    }\n";
        }
    }
    os << "}\n";
    // cons_nln! in ONE launch (unsharded models whose rows collect at most EXA_AUG_LONG terms).  The reference runs the base
    // kernel, the augmentation kernels and compress_to_dense (KA ext :273-308, :691-697); exa_cons + exa_aug_gather are two
    // dependent launches.  Here the thread that owns base row r walks the row's augmentation terms — listed at build time
    // in insertion order as (pattern, data point) — and EVALUATES them itself: same terms, same order of additions, no
    // buffer round trip, no second launch.  augptr [ncon + 1] / augsrc [nconaug]: CSR over constraint rows.
    // When every term is coefficient * x[index] (aug_linear: evaluated at build) the walk is two loads per term, four
    // terms in flight, the additions still in insertion order; otherwise (pattern, point) entries and a switch.
    // Dispatch units are FUSED GROUPS (patterns of exactly the same length): thread I evaluates the rows of all of them in
    // one emitter — the branch table's columns, the gathered voltages and sincos(va_f - va_t) are loaded / computed once
    // for ACOPF's four flow, one angle-difference and two thermal-limit rows of branch I.
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_cons1(const long* __restrict__ P, const double* __restrict__ x, "
          "const double* __restrict__ th, double* __restrict__ out, const long* __restrict__ augptr, const long* __restrict__ augsrc, "
          "const double* __restrict__ augcoef) {\n";
    {
        std::vector<int> augs;
        for (int k = 0; k < np; k++) if (m.pats[k].n > 0 && m.pats[k].kind == EXA_PAT_CONAUG) augs.push_back(k);
        os << "    const long e_ = ((const long*)P[" << L.blk[CB_CONS1] << "])[blockIdx.x];\n    const int ps_ = (int)(e_ >> 40);\n"
              "    const long tid0 = (e_ & ((1L << 40) - 1)) * EXA_BLOCK + threadIdx.x;\n";
        for (size_t g = 0; g < L.groups[CB_CONS1].size(); g++) {
            const auto &grp = L.groups[CB_CONS1][g];
            const auto &pp0 = L.pat[grp.front()];
            os << "    " << (g ? "else " : "") << "if (ps_ == " << g << ") {\n        const long I = P[" << pp0.lo << "] + tid0;\n        if (I >= P[" << pp0.hi
               << "]) return;\n";
            Emitter E;
            std::vector<std::unique_ptr<Body>> bodies;
            std::vector<Val> vals;
            for (int pk : grp) {
                bodies.emplace_back(new Body(m, pk, L, &E));
                vals.push_back(E.tod(bodies.back()->cval(m.pats[pk].root)));
            }
            emit_lines(os, E, "        ");
            for (size_t q = 0; q < grp.size(); q++) {
                const int pk = grp[q];
                const auto &pp = L.pat[pk];
                bool target = false;
                for (int a : augs) target = target || m.pats[a].base == pk;
                if (!target) { os << "        out[P[" << pp.o0 << "] + I] = " << E.sd(vals[q]) << ";\n"; continue; }
                os << "        {\n        double v = " << E.sd(vals[q]) << ";\n        const long r_ = P[" << pp.o0 << "] + I;\n";
                if (m.aug_linear) {
                    os << "        long j = augptr[r_];\n        const long je = augptr[r_ + 1];\n"
                          "        for (; j + 4 <= je; j += 4) {\n"
                          "            const long i0 = augsrc[j], i1 = augsrc[j + 1], i2 = augsrc[j + 2], i3 = augsrc[j + 3];\n"
                          "            const double c0 = augcoef[j], c1 = augcoef[j + 1], c2 = augcoef[j + 2], c3 = augcoef[j + 3];\n"
                          "            const double x0 = x[i0], x1 = x[i1], x2 = x[i2], x3 = x[i3];\n"
                          // (products rounded on their own, like the reference's c * x followed by +=: no FMA contraction)
                          "            v += __dmul_rn(c0, x0); v += __dmul_rn(c1, x1); v += __dmul_rn(c2, x2); v += __dmul_rn(c3, x3);\n        }\n"
                          "        for (; j < je; j++) v += __dmul_rn(augcoef[j], x[augsrc[j]]);\n";
                } else {
                    os << "        for (long j = augptr[r_], je = augptr[r_ + 1]; j < je; j++) {\n"
                          "            const long s_ = augsrc[j];\n            const int ap_ = (int)(s_ >> 40);\n            const long J = s_ & ((1L << 40) - 1);\n";
                    bool first = true;
                    for (int a : augs) {
                        if (m.pats[a].base != pk) continue;
                        os << "            " << (first ? "" : "else ") << "if (ap_ == " << a << ") v += p" << a << "_val(P, x, th, J);\n";
                        first = false;
                    }
                    os << "        }\n";
                }
                os << "        out[r_] = v;\n        }\n";
            }
            os << "    }\n";
        }
    }
    os << "}\n";
    // jprod_nln! in ONE launch, when every augmentation term is c * x[k] (its Jacobian entry is the constant c): same
    // dispatch units and row lists as exa_cons1; row r = sum_s J[r, k_s] v[k_s] + sum_terms c_j v[var_j]
    if (m.aug_linear || m.nconaug == 0) {
        os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_jprod1(const long* __restrict__ P, const double* __restrict__ x, "
              "const double* __restrict__ th, const double* __restrict__ v, double* __restrict__ out, const long* __restrict__ augptr, "
              "const long* __restrict__ augsrc, const double* __restrict__ augcoef) {\n";
        std::vector<int> augs;
        for (int k = 0; k < np; k++) if (m.pats[k].n > 0 && m.pats[k].kind == EXA_PAT_CONAUG) augs.push_back(k);
        os << "    const long e_ = ((const long*)P[" << L.blk[CB_CONS1] << "])[blockIdx.x];\n    const int ps_ = (int)(e_ >> 40);\n"
              "    const long tid0 = (e_ & ((1L << 40) - 1)) * EXA_BLOCK + threadIdx.x;\n";
        for (size_t g = 0; g < L.groups[CB_CONS1].size(); g++) {
            const auto &grp = L.groups[CB_CONS1][g];
            const auto &pp0 = L.pat[grp.front()];
            os << "    " << (g ? "else " : "") << "if (ps_ == " << g << ") {\n        const long I = P[" << pp0.lo << "] + tid0;\n        if (I >= P[" << pp0.hi
               << "]) return;\n";
            Emitter E;
            std::vector<std::unique_ptr<Body>> bodies;
            std::vector<Val> sums;
            for (int pk : grp) {
                bodies.emplace_back(new Body(m, pk, L, &E));
                Body &b = *bodies.back();
                const Pattern &p = b.p;
                Val sum = Emitter::litf(0.0);
                if (p.o1step > 0) {
                    b.forward(p.ad_root, 1, false);
                    GenAlg a(b, p.comp1, p.o1step);
                    grpass(p, p.ad_root, a, Emitter::litf(1.0));
                    for (int sl = 0; sl < p.o1step; sl++) {
                        Val vi = b.fv[p.slotvar1[sl]].vidx;
                        Val vv = E.raw("v[" + E.s(E.sub(vi, Emitter::liti(1))) + "]", false);
                        sum = E.add(sum, E.mul(a.acc[sl], vv));
                    }
                }
                sums.push_back(sum);
            }
            emit_lines(os, E, "        ");
            for (size_t q = 0; q < grp.size(); q++) {
                const int pk = grp[q];
                bool target = false;
                for (int a : augs) target = target || m.pats[a].base == pk;
                if (!target) { os << "        out[P[" << L.pat[pk].o0 << "] + I] = " << E.sd(sums[q]) << ";\n"; continue; }
                os << "        {\n        double s_ = " << E.sd(sums[q]) << ";\n        const long r_ = P[" << L.pat[pk].o0 << "] + I;\n"
                      "        long j = augptr[r_];\n        const long je = augptr[r_ + 1];\n"
                      "        for (; j + 4 <= je; j += 4) {\n"
                      "            const long i0 = augsrc[j], i1 = augsrc[j + 1], i2 = augsrc[j + 2], i3 = augsrc[j + 3];\n"
                      "            const double c0 = augcoef[j], c1 = augcoef[j + 1], c2 = augcoef[j + 2], c3 = augcoef[j + 3];\n"
                      "            const double v0 = v[i0], v1 = v[i1], v2 = v[i2], v3 = v[i3];\n"
                      "            s_ += __dmul_rn(c0, v0); s_ += __dmul_rn(c1, v1); s_ += __dmul_rn(c2, v2); s_ += __dmul_rn(c3, v3);\n        }\n"
                      "        for (; j < je; j++) s_ += __dmul_rn(augcoef[j], v[augsrc[j]]);\n        out[r_] = s_;\n        }\n";
            }
            os << "    }\n";
        }
        os << "}\n";
    }
    auto lds_decl = [&](int cb, bool hess) {
        int mx = 0;
        for (int k : L.active[cb]) { const int S = hess ? m.pats[k].o2step : m.pats[k].o1step; if (use_tile(S)) mx = std::max(mx, tile_doubles(S)); }
        if (mx) os << "    __shared__ double lds_all[(EXA_BLOCK / 64) * " << mx << "];\n    double* lds = lds_all + (threadIdx.x >> 6) * " << mx << ";\n";
        else os << "    double* lds = nullptr;\n";
    };
    // `sink`: 64 doubles nobody reads, the target of lanes that have no slot to store (chained kernels: exa_flush_points_nb)
    auto occupancy_hint = [&](int cb) -> std::string {
        const int w = L.chain[cb] > 0 ? env_int("EXAHIP_CHAIN_WAVES", 0) : 0;
        return w > 0 ? "__attribute__((amdgpu_waves_per_eu(" + std::to_string(w) + "))) " : "";
    };
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) " << occupancy_hint(CB_JAC) << "exa_jac(const long* __restrict__ P, const double* __restrict__ x, "
          "const double* __restrict__ th, double* __restrict__ out) {\n";
    lds_decl(CB_JAC, false);
    gen_dispatch(os, L, CB_JAC, "jac", "P, x, th, out", ", lds");
    os << "}\n";
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) " << occupancy_hint(CB_HESS) << "exa_hess(const long* __restrict__ P, const double* __restrict__ x, "
          "const double* __restrict__ y, const double* __restrict__ th, double* __restrict__ out, double sigma) {\n";
    lds_decl(CB_HESS, true);
    gen_dispatch(os, L, CB_HESS, "hess", "P, x, y, th, out, sigma", ", lds");
    os << "}\n";
    if (L.chain[CB_HESSC] > 0) {
        os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) " << occupancy_hint(CB_HESSC) << "exa_hessc(const long* __restrict__ P, const double* __restrict__ x, "
              "const double* __restrict__ y, const double* __restrict__ th, double* __restrict__ out, double sigma, double* __restrict__ sink) {\n";
        lds_decl(CB_HESS, true);
        gen_dispatch_chained(os, L, CB_HESSC, "hessc", true);
        os << "}\n";
    }
    // fused cons + jac + hess (+ objective partial sums)
    {
        int mx = 0;
        for (int k : L.active[CB_FUSED]) {
            const Pattern &p = m.pats[k];
            if (p.kind != EXA_PAT_OBJ && use_tile(p.o1step)) mx = std::max(mx, tile_doubles(p.o1step));
            if (use_tile(p.o2step)) mx = std::max(mx, tile_doubles(p.o2step));
        }
        os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_fused(const long* __restrict__ P, const double* __restrict__ x, "
              "const double* __restrict__ y, const double* __restrict__ th, double* __restrict__ part, double* __restrict__ cout, "
              "double* __restrict__ augout, double* __restrict__ jout, double* __restrict__ hout, double sigma, "
              "const long* __restrict__ augptr, const long* __restrict__ augsrc, const double* __restrict__ augcoef) {\n";
        if (mx) os << "    __shared__ double lds_all[(EXA_BLOCK / 64) * " << mx << "];\n    double* lds = lds_all + (threadIdx.x >> 6) * " << mx << ";\n";
        else os << "    double* lds = nullptr;\n";
        // only the workgroups of OBJECTIVE patterns have something to add to obj: they write one partial sum each, at a
        // compact index (pattern's first slot + tile), so the reduction reads 1/3 of the workgroup count on LV
        os << "    const long b = blockIdx.x;\n"
           << "    const long e_ = ((const long*)P[" << L.blk[CB_FUSED] << "])[b];\n    const int ps_ = (int)(e_ >> 40);\n"
              "    const long tile_ = e_ & ((1L << 40) - 1);\n    const long tid0 = tile_ * EXA_BLOCK + threadIdx.x;\n";
        const auto &grps = L.groups[CB_FUSED];
        for (size_t k = 0; k < grps.size(); k++) {
            os << "    " << (k ? "else " : "") << "if (ps_ == " << k << ") { const double v = g" << k
               << "_fused(P, x, y, th, cout, augout, jout, hout, sigma, tid0, lds, augptr, augsrc, augcoef);";
            if (m.pats[grps[k].front()].kind == EXA_PAT_OBJ)
                os << " const double s = exa_block_sum(v); if (threadIdx.x == 0) part[P[" << L.pat[grps[k].front()].ob << "] + tile_] = s;";
            else os << " (void)v;";
            os << " }\n";
        }
        os << "}\n";
    }
    const char *prod_sig = "(const long* __restrict__ P, const double* __restrict__ x, const double* __restrict__ th, "
                           "const double* __restrict__ v, double* __restrict__ out) {\n";
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_jprod(const long* __restrict__ P, const double* __restrict__ x, "
          "const double* __restrict__ th, const double* __restrict__ v, double* __restrict__ out, double* __restrict__ aug) {\n";
    gen_dispatch(os, L, CB_JPROD, "jprod", "P, x, th, v, out, aug");
    os << "}\n";
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_jtprod" << prod_sig;
    scatter_lds(CB_JTPROD);
    gen_dispatch(os, L, CB_JTPROD, "jtprod", "P, x, th, v, out", ", lds");
    os << "}\n";
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_hprod(const long* __restrict__ P, const double* __restrict__ x, "
          "const double* __restrict__ y, const double* __restrict__ th, const double* __restrict__ v, double* __restrict__ out, double sigma) {\n";
    scatter_lds(CB_HPROD);
    gen_dispatch(os, L, CB_HPROD, "hprod", "P, x, y, th, v, out, sigma", ", lds");
    os << "}\n";
    for (int wide = 0; wide < 2; wide++) {
        const char *it = wide ? "long" : "int";
        const char *sfx = wide ? "64" : "32";
        os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_jstruct" << sfx << "(const long* __restrict__ P, " << it
           << "* __restrict__ rows, " << it << "* __restrict__ cols) {\n";
        gen_dispatch(os, L, CB_JSTRUCT, std::string("jst<") + it + ">", "P, rows, cols");
        os << "}\n";
        os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_hstruct" << sfx << "(const long* __restrict__ P, " << it
           << "* __restrict__ rows, " << it << "* __restrict__ cols) {\n";
        gen_dispatch(os, L, CB_HSTRUCT, std::string("hst<") + it + ">", "P, rows, cols");
        os << "}\n";
    }
    g.source = os.str();
    return g;
}

// Smallest and largest 1-based variable index the data points [lo, hi) of a pattern read, when every index expression
// is affine in a range column (stencil models); false = some index comes from a data column (anywhere in 1..nvar).
// Used by exa_shard_var_range: a rank of a sharded stencil model needs only that stretch of x (plus nothing else).
bool pattern_var_range(const Pattern &p, int64_t lo, int64_t hi, int64_t *vmin, int64_t *vmax) {
    int64_t a = INT64_MAX, b = INT64_MIN;
    for (const ADNode &n : p.ad) {
        if (n.kind != AD_VAR) continue;
        const Affine f = affine(p, n.ir);
        if (!f.ok) return false;
        if (f.col < 0) { a = std::min(a, f.c); b = std::max(b, f.c); continue; }
        const Column &c = p.cols[f.col];
        const int64_t v0 = f.a * (c.start + c.step * lo) + f.c, v1 = f.a * (c.start + c.step * (hi - 1)) + f.c;
        a = std::min(a, std::min(v0, v1)); b = std::max(b, std::max(v0, v1));
    }
    *vmin = a; *vmax = b;
    return true;
}

// ---- windowed compressed COO ------------------------------------------------------------------
// exa_chess / exa_cjac without the uncompressed round trip.  For a pattern whose data point I puts slot s on compressed
// entry a_s + b*I (checked against the sorted structure at exa_compress time), a workgroup OWNS a window of W
// consecutive compressed entries: it evaluates, for every pattern, exactly the points that touch the window (the few
// points straddling two windows are evaluated by both, each keeping its own entries), adds the values into an LDS copy
// of the window — slot groups in a fixed order, a barrier between groups that could meet in one word, so the sum order
// is fixed and the result bit-reproducible — and streams the window out with plain coalesced stores: no zero-fill, no
// atomics, 8 B of HBM traffic per COMPRESSED entry instead of 16 B + 12 B per uncompressed one.
// The handful of points at a pattern's ends where the structure is irregular (first columns holding fewer rows) are
// left out of the windows and added afterwards by exa_c*x, sequentially.
static void gen_window_value_fn(std::ostringstream &os, const Model &m, const ParamLayout &L, int k, bool hess) {
    Body b(m, k, L);
    const Pattern &p = b.p;
    const int S = hess ? p.o2step : p.o1step;
    std::vector<Val> acc;
    if (hess) {
        b.forward(p.ad_root, 2, false);
        Val adj;
        if (p.kind == EXA_PAT_OBJ) adj = b.e.raw("sigma", false);
        else adj = b.e.raw("y[" + b.row0() + "]", false);
        GenAlg a(b, p.comp2, p.o2step);
        hrpass0(p, p.ad_root, a, adj, zero_seed(b));
        acc = a.acc;
    } else {
        b.forward(p.ad_root, 1, false);
        GenAlg a(b, p.comp1, p.o1step);
        grpass(p, p.ad_root, a, Emitter::litf(1.0));
        acc = a.acc;
    }
    std::vector<std::string> vals;
    for (int s = 0; s < S; s++) vals.push_back(b.e.sd(acc[s]));
    const char *tag = hess ? "hessv" : "jacv";
    // values of one data point, in slot order
    os << "static __device__ __forceinline__ void " << fn_name(k, tag)
       << "(const long* __restrict__ P, const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ th, "
          "double sigma, long I, double* v) {\n";
    emit_lines(os, b.e);
    for (int s = 0; s < S; s++) os << "    v[" << s << "] = " << vals[s] << ";\n";
    os << "}\n";
}

// adds one chunk's values of pass j into the window: groups of a phase never meet in one word, a barrier between phases
static void gen_window_fn(std::ostringstream &os, const WindowPat &wp, int j, bool hess, int S) {
    const int ngroups = (int)wp.phase.size();
    int nphase = 0;
    for (int ph : wp.phase) nphase = std::max(nphase, ph + 1);
    os << "static __device__ __forceinline__ void w" << j << (hess ? "_hessa" : "_jaca")
       << "(const long* __restrict__ Q, long I, bool act, long c0, int W, double* win, const double* v) {\n"
       << "    const long cb_ = Q[" << wp.qbase << "] * I - c0;\n";
    for (int ph = 0; ph < nphase; ph++) {
        if (ph) os << "    __syncthreads();\n";
        for (int g = 0; g < ngroups; g++) {
            if (wp.phase[g] != ph) continue;
            std::string sum;
            for (int s = 0; s < S; s++)
                if (wp.group[s] == g) sum += (sum.empty() ? "" : " + ") + ("v[" + std::to_string(s) + "]");
            os << "    { const long c = Q[" << wp.qbase + 5 + g << "] + cb_; if (act && (unsigned long)c < (unsigned long)W) win[EXA_WPOS((int)c)] += " << sum << "; }\n";
        }
    }
    os << "}\n";
}

static void emit_window_shared_body(std::ostringstream &os, const Model &m, const std::vector<WindowShared> &sh, bool hess);
static void gen_window_kernels(std::ostringstream &os, const Model &m, const std::vector<WindowPat> &pats, const std::vector<WindowShared> &sh,
                               bool hess, bool single) {
    const char *nm = hess ? "chess" : "cjac";
    const char *fa = hess ? "hessa" : "jaca";
    const char *fv = hess ? "hessv" : "jacv";
    const int np = (int)pats.size();
    // R[window][pass] = first and one-past-last data point touching the window (host-computed: no 64-bit divisions at
    // the head of every workgroup's dependency chain).
    // Occupancy hint: the straight-line kernel is latency-bound between barriers (LV 1e7: 0.141 ms unhinted at 124
    // VGPRs, 0.10 ms at 8 waves per SIMD); only for small bodies, which fit 64 / 80 registers without spilling — the
    // chunk loops did spill under it (LV, two chunks per window: 0.118 -> 0.375 ms)
    int waves = env_int("EXAHIP_CW_WAVES", -1);
    if (waves < 0) {
        int slots = 0;
        for (const auto &wp : pats) slots += hess ? m.pats[wp.k].o2step : m.pats[wp.k].o1step;
        waves = !single ? 0 : (slots <= 16 ? 8 : (slots <= 40 ? 6 : 0));
    }
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) ";
    if (waves > 0) os << "__attribute__((amdgpu_waves_per_eu(" << waves << "))) ";
    os << "exa_" << nm << "w(const long* __restrict__ P, const long* __restrict__ Q, "
          "const int* __restrict__ R, const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ th, "
          "double* __restrict__ cout, double sigma, long ncomp, int W) {\n"
          "    extern __shared__ double win[];\n";
    os << "    const long c0 = (long)blockIdx.x * W;\n"
          "    const int* r_ = R + (long)blockIdx.x * " << 2 * np << ";\n"
          "    for (int w = threadIdx.x; w < W; w += EXA_BLOCK) win[w] = 0.0;\n";
    for (int j = 0; j < np; j++)
        os << "    const long lo" << j << " = r_[" << 2 * j << "], hi" << j << " = r_[" << 2 * j + 1 << "];\n";
    if (single) {
        // every pass fits one chunk: all values first (the loads of all passes overlap), then the additions
        os << "    {\n";
        for (int j = 0; j < np; j++) {
            const int S = hess ? m.pats[pats[j].k].o2step : m.pats[pats[j].k].o1step;
            os << "        const bool act" << j << " = lo" << j << " + threadIdx.x < hi" << j << ";\n        const long I" << j << " = act" << j << " ? lo" << j
               << " + threadIdx.x : 0;\n        double v" << j << "[" << S << "];\n        " << fn_name(pats[j].k, fv) << "(P, x, y, th, sigma, I" << j << ", v" << j << ");\n";
        }
        for (int j = 0; j < np; j++)
            os << "        __syncthreads();\n        w" << j << "_" << fa << "(Q, I" << j << ", act" << j << ", c0, W, win, v" << j << ");\n";
    } else {
        // chunk loops, software-pipelined: the next chunk's values are computed (its loads issued) before the current
        // chunk's additions wait at the barrier
        os << "    {\n";
        for (int j = 0; j < np; j++) {
            const int S = hess ? m.pats[pats[j].k].o2step : m.pats[pats[j].k].o1step;
            const std::string ev = fn_name(pats[j].k, fv);
            os << "        if (lo" << j << " < hi" << j << ") {\n            long base = lo" << j << ";\n            bool act = base + threadIdx.x < hi" << j
               << ";\n            long I = act ? base + threadIdx.x : 0;\n            double v[" << S << "], vn[" << S << "];\n            " << ev
               << "(P, x, y, th, sigma, I, v);\n            while (base < hi" << j << ") {\n                const long nb = base + EXA_BLOCK;\n"
               << "                const bool actn = nb + threadIdx.x < hi" << j << ";\n                const long In = actn ? nb + threadIdx.x : 0;\n"
               << "                if (nb < hi" << j << ") " << ev << "(P, x, y, th, sigma, In, vn);\n                __syncthreads();\n                w" << j << "_" << fa
               << "(Q, I, act, c0, W, win, v);\n                for (int s = 0; s < " << S << "; s++) v[s] = vn[s];\n                act = actn; I = In; base = nb;\n"
               << "            }\n        }\n";
        }
    }
    os << "    }\n    __syncthreads();\n"
          "    for (int w = threadIdx.x; w < W; w += EXA_BLOCK) if (c0 + w < ncomp) __builtin_nontemporal_store(win[EXA_WPOS(w)], &cout[c0 + w]);\n}\n";
}

// block-owned variant: see WindowSpec.  R[block][pattern] = the points of the pattern with a slot in one of the block's
// windows (at most EXA_BLOCK of them: one chunk); every pattern is evaluated once, then each pass adds its slots into the
// window of its space; windows are clipped to their space when streamed out
static void gen_window_kernel_blocks(std::ostringstream &os, const Model &m, const std::vector<WindowPat> &pats, bool hess, int nspaces, int zs) {
    const char *nm = hess ? "chess" : "cjac";
    const char *fa = hess ? "hessa" : "jaca";
    const char *fv = hess ? "hessv" : "jacv";
    std::vector<int> pk;
    for (const auto &wp : pats) if (std::find(pk.begin(), pk.end(), wp.k) == pk.end()) pk.push_back(wp.k);
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_" << nm << "w(const long* __restrict__ P, const long* __restrict__ Q, "
          "const int* __restrict__ R, const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ th, "
          "double* __restrict__ cout, double sigma, long ncomp, int W) {\n"
          "    extern __shared__ double win[];\n    const long j_ = blockIdx.x;\n"
          "    const int* r_ = R + j_ * " << 2 * pk.size() << ";\n"
          "    for (int w = threadIdx.x; w < W; w += EXA_BLOCK) win[w] = 0.0;\n";
    for (size_t q = 0; q < pk.size(); q++) {
        const int S = hess ? m.pats[pk[q]].o2step : m.pats[pk[q]].o1step;
        os << "    const bool act" << q << " = r_[" << 2 * q << "] + (long)threadIdx.x < r_[" << 2 * q + 1 << "];\n    const long I" << q << " = act" << q
           << " ? r_[" << 2 * q << "] + (long)threadIdx.x : 0;\n    double v" << q << "[" << S << "];\n    " << fn_name(pk[q], fv) << "(P, x, y, th, sigma, I" << q
           << ", v" << q << ");\n";
    }
    for (size_t j = 0; j < pats.size(); j++) {
        const size_t q = std::find(pk.begin(), pk.end(), pats[j].k) - pk.begin();
        const int z = zs + 4 * pats[j].space;
        os << "    __syncthreads();\n    w" << j << "_" << fa << "(Q, I" << q << ", act" << q << ", Q[" << z << "] + j_ * Q[" << z + 2 << "], (int)Q[" << z + 2
           << "], win + Q[" << z + 3 << "], v" << q << ");\n";
    }
    os << "    __syncthreads();\n";
    for (int sp = 0; sp < nspaces; sp++) {
        const int z = zs + 4 * sp;
        os << "    {\n        const long c0 = Q[" << z << "] + j_ * Q[" << z + 2 << "], end = Q[" << z + 1 << "];\n        const int We = (int)Q[" << z + 2
           << "];\n        const double* wn = win + Q[" << z + 3 << "];\n"
           << "        for (int w = threadIdx.x; w < We; w += EXA_BLOCK) if (c0 + w < end) __builtin_nontemporal_store(wn[EXA_WPOS(w)], &cout[c0 + w]);\n    }\n";
    }
    os << "    (void)ncomp;\n}\n";
}

// irregular end points: X = [pattern, I] per point (up to EXA_BLOCK of them); values go through xbuf; then one thread
// per DISTINCT compressed target adds that target's values in (point, slot) order: T = [ntargets, then per target:
// compressed entry, first, one-past-last position in E], E = positions in xbuf
static void gen_window_x(std::ostringstream &os, const Model &m, const std::vector<int> &pk, bool hess) {
    const char *nm = hess ? "chess" : "cjac";
    const char *fv = hess ? "hessv" : "jacv";
    int smax = 1;
    for (int k : pk) smax = std::max(smax, hess ? m.pats[k].o2step : m.pats[k].o1step);
    os << "extern \"C\" __global__ void __launch_bounds__(1024) exa_" << nm << "x(const long* __restrict__ P, const long* __restrict__ X, "
          "const int* __restrict__ T, const int* __restrict__ E, const double* __restrict__ x, const double* __restrict__ y, "
          "const double* __restrict__ th, double* __restrict__ xbuf, double* __restrict__ cout, double sigma, int nx, "
          "const double* __restrict__ part, const long* __restrict__ F) {\n"
          "    const int t = threadIdx.x;\n    if (t < nx) {\n        const long pk_ = X[2 * t], I = X[2 * t + 1];\n        double v[" << smax << "];\n"
          "        for (int s = 0; s < " << smax << "; s++) v[s] = 0.0;\n";
    for (size_t j = 0; j < pk.size(); j++)
        os << "        " << (j ? "else " : "") << "if (pk_ == " << pk[j] << ") " << fn_name(pk[j], fv) << "(P, x, y, th, sigma, I, v);\n";
    os << "        for (int s = 0; s < " << smax << "; s++) xbuf[t * " << smax << " + s] = v[s];\n    }\n    __syncthreads();\n"
          "    for (int q = t; q < (nx > 0 ? T[0] : 0); q += 1024) {\n        const int c = T[1 + 3 * q];\n        double s = cout[c];\n"
          "        for (int e = T[2 + 3 * q]; e < T[3 + 3 * q]; e++) s += xbuf[E[e]];\n        cout[c] = s;\n    }\n"
          // fold of the shared-entry partial sums: F = [ngroups, then per group: first partial, count, compressed entry];
          // groups in order (several may share an entry), fixed summation order
          "    __shared__ double red[16];\n"
          "    for (long g = 0; g < F[0]; g++) {\n        __syncthreads();\n        const long off = F[1 + 3 * g], n = F[2 + 3 * g];\n        double a = 0.0;\n"
          "        for (long i = t; i < n; i += 1024) a += part[off + i];\n        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);\n"
          "        if ((t & 63) == 0) red[t >> 6] = a;\n        __syncthreads();\n"
          "        if (t == 0) { double s = 0.0; for (int w = 0; w < 16; w++) s += red[w]; cout[F[3 + 3 * g]] += s; }\n    }\n}\n";
}
static void emit_window_shared_body(std::ostringstream &os, const Model &m, const std::vector<WindowShared> &sh, bool hess) {
    const char *fv = hess ? "hessv" : "jacv";
    for (size_t j = 0; j < sh.size(); j++) {
        const int St = hess ? m.pats[sh[j].k].o2step : m.pats[sh[j].k].o1step;
        const size_t ng = sh[j].groups.size();
        os << "        " << (j ? "else " : "") << "if (b < S[" << 4 * (j + 1) + 2 << "]) {\n            const long tile = b - S[" << 4 * j + 2 << "], nt = S["
           << 4 * (j + 1) + 2 << "] - S[" << 4 * j + 2 << "];\n            double acc[" << ng << "];\n            for (int g = 0; g < " << ng << "; g++) acc[g] = 0.0;\n"
           << "#pragma unroll 1\n            for (int u = 0; u < " << kSharedTiles << "; u++) {\n"
           << "                const long I0 = S[" << 4 * j << "] + (tile * " << kSharedTiles << " + u) * EXA_BLOCK + threadIdx.x;\n"
           << "                if (I0 - threadIdx.x >= S[" << 4 * j + 1 << "]) break;\n"
           << "                const bool act = I0 < S[" << 4 * j + 1 << "];\n                const long I = act ? I0 : 0;\n                double v[" << St << "];\n                "
           << fn_name(sh[j].k, fv) << "(P, x, y, th, sigma, I, v);\n";
        for (size_t g = 0; g < ng; g++) {
            std::string sum;
            for (int s : sh[j].groups[g]) sum += (sum.empty() ? "" : " + ") + ("v[" + std::to_string(s) + "]");
            os << "                acc[" << g << "] += act ? " << sum << " : 0.0;\n";
        }
        os << "            }\n";
        for (size_t g = 0; g < ng; g++)
            os << "            { const double s = exa_block_sum(acc[" << g << "]); if (threadIdx.x == 0) part[S[" << 4 * j + 3 << "] + " << g
               << " * nt + tile] = s; __syncthreads(); }\n";
        os << "        }\n";
    }
}

// entries EVERY point adds to (b = 0: the rocket's step length): per-workgroup sums over the regular points (S = [per
// pattern j: e_lo, e_hi, first workgroup, first partial] + sentinel), folded by the tail kernel.  A launch of its own:
// as extra workgroups of the window kernel they each reserved a window's LDS and cost more than the launch (rocket chess
// 0.125 -> 0.137 ms)
static void gen_window_shared(std::ostringstream &os, const Model &m, const std::vector<WindowShared> &sh, bool hess) {
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_" << (hess ? "chess" : "cjac") << "s(const long* __restrict__ P, const long* __restrict__ S, "
          "const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ th, double* __restrict__ part, double sigma) {\n"
          "    {\n        const long b = blockIdx.x;\n";
    emit_window_shared_body(os, m, sh, hess);
    os << "    }\n}\n";
}

std::vector<int> merged_hess_slots(const Model &m, const ParamLayout &L) {
    std::vector<int> out;
    for (const auto &grp : L.groups[CB_HESS]) out.push_back(merged_slot_count(m, L, grp));
    return out;
}

std::string generate_window_module(const Model &m, const ParamLayout &L, const WindowSpec &spec) {
    std::lock_guard<std::mutex> gen_lock(g_gen_mu);
    std::ostringstream os;
    {
        std::string pre = kPrelude;
        const std::string tag = "@BLOCK@", tag2 = "@PULLPPT@";
        pre.replace(pre.find(tag), tag.size(), std::to_string(kBlock));
        pre.replace(pre.find(tag2), tag2.size(), std::to_string(L.pull_ppt));
        os << pre;
    }
    // LDS position of window entry c: the low four bits (the 64-bit bank) are XOR-ed with the next four, so that lanes
    // striding through the window by 2, 3, 12 ... entries (the stride of a pass) spread over the banks instead of
    // hitting the same 2-4 of them (rocket, stride 12: 8-way conflicts on every read-modify-write); a bijection within
    // each aligned block of 16 entries, W is a multiple of 16
    os << "// windowed compressed-COO kernels\n#define EXA_WPOS(c) " << (env_int("EXAHIP_CW_SWIZZLE", 1) ? "((c) ^ (((c) >> 4) & 15))" : "(c)") << "\n";
    if (spec.hess_merged) {
        // exa_chessm / exa_hstructm: the merged slot space (see merge_slot); M[g] = first merged slot of group g
        const auto &groups = L.groups[CB_HESS];
        for (size_t g = 0; g < groups.size(); g++) { gen_merged_hess_fn(os, m, L, (int)g); gen_merged_struct_fn(os, m, L, (int)g); }
        const std::string head = "    const long e_ = ((const long*)P[" + std::to_string(L.blk[CB_HESS]) + "])[blockIdx.x];\n    const int ps_ = (int)(e_ >> 40);\n"
                                 "    const long tid0 = (e_ & ((1L << 40) - 1)) * EXA_BLOCK + threadIdx.x;\n";
        os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_chessm(const long* __restrict__ P, const double* __restrict__ x, "
              "const double* __restrict__ y, const double* __restrict__ th, double* __restrict__ out, double sigma, const unsigned* __restrict__ pos, "
              "const long* __restrict__ M) {\n" << head;
        for (size_t g = 0; g < groups.size(); g++)
            os << "    " << (g ? "else " : "") << "if (ps_ == " << g << ") g" << g << "_hessm(P, x, y, th, out, sigma, tid0, pos, M[" << g << "]);\n";
        os << "}\nextern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_hstructm(const long* __restrict__ P, long* __restrict__ rows, "
              "long* __restrict__ cols, const long* __restrict__ M) {\n" << head;
        for (size_t g = 0; g < groups.size(); g++)
            os << "    " << (g ? "else " : "") << "if (ps_ == " << g << ") g" << g << "_hstm(P, rows, cols, tid0, M[" << g << "]);\n";
        os << "}\n";
    }
    for (int hess = 1; hess >= 0; hess--) {
        if (!(hess ? spec.hess_scatter : spec.jac_scatter)) continue;
        const int cb = hess ? CB_HESS : CB_JAC;
        for (size_t g = 0; g < L.groups[cb].size(); g++) gen_coo_group_fn(os, m, L, cb, (int)g, true);
        os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_" << (hess ? "chessp" : "cjacp")
           << "(const long* __restrict__ P, const double* __restrict__ x, " << (hess ? "const double* __restrict__ y, " : "")
           << "const double* __restrict__ th, double* __restrict__ out, " << (hess ? "double sigma, " : "") << "const unsigned* __restrict__ pos) {\n"
           << "    const long e_ = ((const long*)P[" << L.blk[cb] << "])[blockIdx.x];\n    const int ps_ = (int)(e_ >> 40);\n"
              "    const long tid0 = (e_ & ((1L << 40) - 1)) * EXA_BLOCK + threadIdx.x;\n";
        for (size_t g = 0; g < L.groups[cb].size(); g++)
            os << "    " << (g ? "else " : "") << "if (ps_ == " << g << ") g" << g << "_" << (hess ? "hessp" : "jacp") << "(P, x, " << (hess ? "y, " : "")
               << "th, out, " << (hess ? "sigma, " : "") << "tid0, pos);\n";
        os << "}\n";
    }
    for (int hess = 1; hess >= 0; hess--) {
        const auto &pats = hess ? spec.hess : spec.jac;
        const auto &sh = hess ? spec.hess_shared : spec.jac_shared;
        if (pats.empty()) continue;
        // value function once per pattern, accumulate function per (pattern, stride class) pass
        std::vector<int> pk;
        for (const auto &wp : pats) if (std::find(pk.begin(), pk.end(), wp.k) == pk.end()) pk.push_back(wp.k);
        for (const auto &q : sh) if (std::find(pk.begin(), pk.end(), q.k) == pk.end()) pk.push_back(q.k);
        for (int k : pk) gen_window_value_fn(os, m, L, k, hess != 0);
        for (size_t j = 0; j < pats.size(); j++) gen_window_fn(os, pats[j], (int)j, hess != 0, hess ? m.pats[pats[j].k].o2step : m.pats[pats[j].k].o1step);
        if ((hess ? spec.hess_nspaces : spec.jac_nspaces) > 0)
            gen_window_kernel_blocks(os, m, pats, hess != 0, hess ? spec.hess_nspaces : spec.jac_nspaces, hess ? spec.hess_zs : spec.jac_zs);
        else
            gen_window_kernels(os, m, pats, sh, hess != 0, hess ? spec.hess_single : spec.jac_single);
        // every active pattern may own irregular end points
        std::vector<int> all;
        for (int k : L.active[hess ? CB_HESS : CB_JAC]) all.push_back(k);
        for (int k : all) if (std::find(pk.begin(), pk.end(), k) == pk.end()) { gen_window_value_fn(os, m, L, k, hess != 0); pk.push_back(k); }
        gen_window_x(os, m, all, hess != 0);
        if (!sh.empty()) gen_window_shared(os, m, sh, hess != 0);
    }
    return os.str();
}

}  // namespace exa
