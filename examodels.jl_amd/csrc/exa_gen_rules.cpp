// exa_gen_rules.cpp — the derivative tables of the reference as symbolic rules.
//
// What it restates: _UNIVARIATES / _BIVARIATES of src/functionlist.jl:6-81 (closed-form f', f'' and the five bivariate
// partials), the integer-power rewrite of src/specialization.jl:193-202 and the FirstFixed / SecondFixed forms of
// src/register.jl:231-266.  Formulas are algebraically the table's; a few are written through the already-computed primal
// to save FP64 divides (parity bar 1e-10 relative; EXAHIP_STRICT_IEEE=1 restores the table's own forms
// where they differ in special values).
#include "exa_gen.hpp"

namespace exa {

// ---- user-registered functions ---------------------------------------------------------------------------------------------------
static std::mutex g_user_mu;
static std::vector<std::unique_ptr<UserFn>> g_user_un, g_user_bin;      // id = EXA_USER_FN_BASE + index; entries are never removed
static bool ident_ok(const std::string &n) {
    if (n.empty() || n.size() > 64 || (n[0] >= '0' && n[0] <= '9')) return false;
    for (char c : n) if (!((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_')) return false;
    return true;
}
int register_user_fn(bool bivariate, const UserFn &fn, std::string *err, bool dry_run, bool *known) {
    if (known) *known = false;
    if (!ident_ok(fn.name)) { *err = "function name must be an identifier of at most 64 characters"; return -1; }
    // placeholders: univariate f: $1 | df: $1 $2(primal) | ddf: $1 $2 $3(first derivative);  bivariate f: $1 $2 | partials: $1 $2 $3(primal)
    struct Rule { const std::string *t; int nph; };
    std::vector<Rule> rules;
    if (bivariate) rules = {{&fn.f, 2}, {&fn.d1, 3}, {&fn.d2, 3}, {&fn.d11, 3}, {&fn.d12, 3}, {&fn.d22, 3}};
    else if (!fn.fused.empty()) {
        rules = {{&fn.fused, 4}};      // $1 argument; $2 $3 $4 receive f, f', f''
        for (const char *ph : {"$1", "$2", "$3", "$4"})
            if (fn.fused.find(ph) == std::string::npos) { *err = "the fused rule of `" + fn.name + "` must use $1 and assign $2, $3 and $4"; return -1; }
    } else rules = {{&fn.f, 1}, {&fn.d1, 2}, {&fn.d11, 3}};
    if (bivariate && !fn.fused.empty()) { *err = "`" + fn.name + "`: a bivariate registration has no fused form"; return -1; }
    for (const Rule &r : rules) {
        if (r.t->empty()) { *err = "every rule of `" + fn.name + "` needs an expression ('=0' for an exact zero)"; return -1; }
        for (size_t i = 0; i < r.t->size(); i++) {
            if ((*r.t)[i] != '$') continue;
            // a placeholder is '$' + ONE digit the rule has: a bare '$' at the end and "$10" (would paste as $1 followed by 0) are refused here,
            // not by the compiler at model build
            const char d = i + 1 < r.t->size() ? (*r.t)[i + 1] : '\0', d2 = i + 2 < r.t->size() ? (*r.t)[i + 2] : '\0';
            if (d < '1' || d > char('0' + r.nph) || (d2 >= '0' && d2 <= '9'))
                { *err = "rule `" + *r.t + "` of `" + fn.name + "` refers to a placeholder it does not have"; return -1; }
        }
    }
    std::lock_guard<std::mutex> lk(g_user_mu);
    auto &tab = bivariate ? g_user_bin : g_user_un;
    for (size_t i = 0; i < tab.size(); i++)
        if (tab[i]->name == fn.name) {
            const UserFn &o = *tab[i];      // registering the same rules again returns the same id; different rules are refused
            if (o.f == fn.f && o.d1 == fn.d1 && o.d2 == fn.d2 && o.d11 == fn.d11 && o.d12 == fn.d12 && o.d22 == fn.d22 && o.helpers == fn.helpers && o.fused == fn.fused)
                { if (known) *known = true; return EXA_USER_FN_BASE + (int)i; }
            *err = "`" + fn.name + "` is already registered with other rules"; return -1;
        }
    if (dry_run) return EXA_USER_FN_BASE + (int)tab.size();
    tab.push_back(std::make_unique<UserFn>(fn));
    return EXA_USER_FN_BASE + (int)tab.size() - 1;
}
const UserFn *user_fn(bool bivariate, int fn) {
    std::lock_guard<std::mutex> lk(g_user_mu);
    auto &tab = bivariate ? g_user_bin : g_user_un;
    const int k = fn - EXA_USER_FN_BASE;
    return k >= 0 && k < (int)tab.size() ? tab[(size_t)k].get() : nullptr;
}

namespace gen {

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
// ---------------------------------------------------------------------------------------------------
// function rules: (x, y, h) of a univariate; $1 = argument, $2 = primal f, $3 = first derivative
// ---------------------------------------------------------------------------------------------------
struct UnSpec { const char *f, *df, *ddf; };
// A leading '=' marks an exact literal (lets the reverse sweep fold it).
const double kPi = 3.14159265358979323846;
const double kD2R = kPi / 180.0, kR2D = 180.0 / kPi;

std::mutex g_gen_mu;

static const UnSpec *un_spec(int fn) {
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
        // ---- the SpecialFunctions extension (ext/functionlist.jl:6-102).  Primal functions: ocml where it has them (erf, erfc,
        // erfcx, erfinv, erfcinv, tgamma, j0/j1/jn, y0/y1/yn), else the exa_* routines of the special prelude (exa_gen_prelude.cpp).
        // The table's derivative formulas, written through the primal ($2) / first derivative ($3) where that is the same algebra.
        T[EXA_U_ERF] = {"erf($1)", "(2.0 * EXA_INVSQRTPI) * exp(-($1 * $1))", "-2.0 * $1 * $3"};
        T[EXA_U_ERFC] = {"erfc($1)", "-(2.0 * EXA_INVSQRTPI) * exp(-($1 * $1))", "-2.0 * $1 * $3"};
        T[EXA_U_ERFI] = {"exa_erfi($1)", "(2.0 * EXA_INVSQRTPI) * exp($1 * $1)", "2.0 * $1 * $3"};
        T[EXA_U_ERFCX] = {"erfcx($1)", "2.0 * (-EXA_INVSQRTPI + $1 * $2)", "2.0 * ($2 + $1 * $3)"};
        T[EXA_U_DIGAMMA] = {"exa_polygamma<0>($1)", "exa_polygamma<1>($1)", "exa_polygamma<2>($1)"};
        T[EXA_U_TRIGAMMA] = {"exa_polygamma<1>($1)", "exa_polygamma<2>($1)", "exa_polygamma<3>($1)"};
        T[EXA_U_INVDIGAMMA] = {"exa_invdigamma($1)", "1.0 / exa_polygamma<1>($2)", "-exa_polygamma<2>($2) * ($3 * $3 * $3)"};
        T[EXA_U_GAMMA] = {"tgamma($1)", "$2 * exa_polygamma<0>($1)", "$2 * (exa_polygamma<1>($1) + exa_sq(exa_polygamma<0>($1)))"};
        T[EXA_U_AIRYAI] = {"exa_airy<0>($1)", "exa_airy<1>($1)", "$1 * $2"};
        T[EXA_U_AIRYBI] = {"exa_airy<2>($1)", "exa_airy<3>($1)", "$1 * $2"};
        T[EXA_U_AIRYAIPRIME] = {"exa_airy<1>($1)", "$1 * exa_airy<0>($1)", "exa_airy<0>($1) + $1 * $2"};
        T[EXA_U_AIRYBIPRIME] = {"exa_airy<3>($1)", "$1 * exa_airy<2>($1)", "exa_airy<2>($1) + $1 * $2"};
        T[EXA_U_BESSELJ0] = {"j0($1)", "-j1($1)", "0.5 * (exa_jn(2, $1) - $2)"};
        T[EXA_U_BESSELY0] = {"y0($1)", "-y1($1)", "0.5 * (yn(2, $1) - $2)"};
        T[EXA_U_BESSELJ1] = {"j1($1)", "0.5 * (j0($1) - exa_jn(2, $1))", "0.5 * (0.5 * (exa_jn(3, $1) - $2) - $2)"};
        T[EXA_U_BESSELY1] = {"y1($1)", "0.5 * (y0($1) - yn(2, $1))", "0.5 * (0.5 * (yn(3, $1) - $2) - $2)"};
        T[EXA_U_DAWSON] = {"exa_dawson($1)", "1.0 - 2.0 * $1 * $2", "-2.0 * $2 - 2.0 * $1 * $3"};
        T[EXA_U_ERFINV] = {"erfinv($1)", "EXA_SQRTPIHALF * exp($2 * $2)", "2.0 * $2 * $3 * $3"};
        T[EXA_U_ERFCINV] = {"erfcinv($1)", "-EXA_SQRTPIHALF * exp($2 * $2)", "2.0 * $2 * $3 * $3"};
    }
    return &T[fn];
}

static double host_un(int fn, double x) {   // folding of literal arguments, primal only
    switch (fn) {
    case EXA_U_PLUS: return x; case EXA_U_MINUS: return -x; case EXA_U_ABS2: return x * x; case EXA_U_ABS: return std::fabs(x);
    case EXA_U_INV: return 1.0 / x; case EXA_U_SQRT: return std::sqrt(x);
    default: return NAN;
    }
}
static bool host_un_ok(int fn) {
    return fn == EXA_U_PLUS || fn == EXA_U_MINUS || fn == EXA_U_ABS2 || fn == EXA_U_ABS || fn == EXA_U_INV || fn == EXA_U_SQRT;
}

static Val lit_or_tmpl(Emitter &e, const char *spec, Val u, Val f, Val d) {
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
        // EXAHIP_FAST_TRIG=2: a value-only context takes the sine or the cosine ALONE (exa_sin1 / exa_cos1: one polynomial, <= 2.5 ulp)
        // unless the other one of the same argument is already there
        if (order == 0 && env_int("EXAHIP_FAST_TRIG", 1) == 2 && e.memo.find("sincos|" + e.s(u)) == e.memo.end()) {
            r.x = e.call(fn == EXA_U_SIN ? "exa_sin1($1)" : "exa_cos1($1)", {u});
            return r;
        }
        // (value-only contexts too: exa_sin / exa_cos each run the whole sincos, so a pattern — or a fused group — that
        // needs both of one argument pays once)
        // one sincos per argument serves value and both derivatives (functionlist.jl:22-23)
        const std::string key = "sincos|" + e.s(u);
        Val sv, cv;
        auto it = e.memo.find(key);
        if (it == e.memo.end() && env_int("EXAHIP_SYM_TRIG", 1)) {
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
    UnSpec user;
    const UnSpec *sp;
    const UserFn *uf = user_fn(false, fn);
    if (uf && !uf->fused.empty()) {
        // exa_register_univariate_fused: one statement per distinct argument gives value and both derivatives (what the table does for
        // sin / cos through sincos); value-only kernels let the compiler drop what they do not read
        const std::string key = "userfused|" + std::to_string(fn) + "|" + e.s(u);
        Val v[3];
        if (e.memo.find(key) == e.memo.end()) {
            std::string decl = "double ", stmt;
            for (int k = 0; k < 3; k++) { v[k].k = Val::SF; v[k].id = e.next++; decl += (k ? ", t" : "t") + std::to_string(v[k].id); }
            for (size_t i = 0; i < uf->fused.size(); i++) {
                const char c = uf->fused[i], d = i + 1 < uf->fused.size() ? uf->fused[i + 1] : 0;
                if (c == '$' && d >= '1' && d <= '4') { stmt += d == '1' ? "(" + e.sd(u) + ")" : e.s(v[d - '2']); i++; }
                else stmt += c;
            }
            // the statement in its own block: temporaries it declares do not collide between two uses in one kernel
            e.lines.push_back(decl + "; { " + stmt + (stmt.empty() || stmt.back() == ';' ? "" : ";") + " }");
            e.memo[key] = v[0]; e.memo[key + "|d"] = v[1]; e.memo[key + "|h"] = v[2];
        }
        r.x = e.memo[key]; r.y = e.memo[key + "|d"]; r.h = e.memo[key + "|h"];
        return r;
    }
    if (uf) {          // exa_register_univariate: $1 argument, $2 primal, $3 first derivative (ddf only)
        user = {uf->f.c_str(), uf->d1.c_str(), uf->d11.c_str()};
        sp = &user;
    } else if (fn == EXA_U_EXP && env_int("EXAHIP_FAST_EXP", 1)) {
        // the lean FP64 exponential of the prelude (exa_exp: 24 vector instructions where ocml's takes 40, < 1 ulp); 0 = ocml's
        user = {"exa_exp($1)", "$2", "$2"};
        sp = &user;
    } else sp = un_spec(fn);
    if (!sp->f) fail("univariate function without rule");
    r.x = e.call(sp->f, {u});
    if (order >= 1) r.y = lit_or_tmpl(e, sp->df, u, r.x, r.x);
    if (order >= 2) r.h = lit_or_tmpl(e, sp->ddf, u, r.x, r.y);
    return r;
}

// x^n for a literal integer n by repeated multiplication (Base.^(::Float64, ::Integer); n==3 -> x*x*x)
static Val powi_lit(Emitter &e, Val x, int64_t n) {
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
static Val add_i(Emitter &e, Val v, int64_t k) { return e.add(v, v.is_int() ? Emitter::liti(k) : Emitter::litf((double)k)); }

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
    case EXA_B_BETA:
    case EXA_B_LOGBETA: {
        // ext/functionlist.jl:109-124: with p1 = digamma(x1) - digamma(x1 + x2), p2 = digamma(x2) - digamma(x1 + x2),
        // t. = trigamma(.):  logbeta: (p1, p2, t1 - t12, -t12, t2 - t12);  beta: B * (p1, p2, t1 - t12 + p1^2, -t12 + p1 p2, t2 - t12 + p2^2)
        const bool lg = fn == EXA_B_LOGBETA;
        r.x = e.call(lg ? "exa_logbeta($1, $2)" : "exa_beta($1, $2)", {x1, x2});
        if (order >= 1) {
            Val s = e.add(x1, x2);
            Val ds = e.call("exa_polygamma<0>($1)", {s});
            Val p1 = e.sub(e.call("exa_polygamma<0>($1)", {x1}), ds), p2 = e.sub(e.call("exa_polygamma<0>($1)", {x2}), ds);
            r.y1 = lg ? p1 : e.mul(r.x, p1);
            r.y2 = lg ? p2 : e.mul(r.x, p2);
            if (order >= 2) {
                Val ts = e.call("exa_polygamma<1>($1)", {s});
                Val a11 = e.sub(e.call("exa_polygamma<1>($1)", {x1}), ts), a22 = e.sub(e.call("exa_polygamma<1>($1)", {x2}), ts);
                if (lg) { r.h11 = a11; r.h12 = e.neg(ts); r.h22 = a22; }
                else {
                    r.h11 = e.mul(r.x, e.add(a11, e.sq(p1)));
                    r.h12 = e.add(e.neg(e.mul(r.x, ts)), e.mul(e.mul(r.x, p1), p2));
                    r.h22 = e.mul(r.x, e.add(a22, e.sq(p2)));
                }
            }
        }
        return r;
    }
    }
    if (const UserFn *uf = user_fn(true, fn)) {       // exa_register_bivariate: $1, $2 the arguments, $3 the primal
        r.x = e.call(uf->f, {x1, x2});
        if (order >= 1) {
            r.y1 = lit_or_tmpl(e, uf->d1.c_str(), x1, x2, r.x);
            r.y2 = lit_or_tmpl(e, uf->d2.c_str(), x1, x2, r.x);
            if (order >= 2) {
                r.h11 = lit_or_tmpl(e, uf->d11.c_str(), x1, x2, r.x);
                r.h12 = lit_or_tmpl(e, uf->d12.c_str(), x1, x2, r.x);
                r.h22 = lit_or_tmpl(e, uf->d22.c_str(), x1, x2, r.x);
            }
        }
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

}  // namespace gen
}  // namespace exa
