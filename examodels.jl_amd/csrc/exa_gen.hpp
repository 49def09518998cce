// exa_gen.hpp — internals of the HIP code generator, shared by its translation units:
//   exa_gen_rules.cpp    derivative tables (src/functionlist.jl:6-81) as symbolic rules
//   exa_gen_prelude.cpp  the fixed text every generated module starts with (device helpers, reduction kernels)
//   exa_gen_coo.cpp      value / COO-writing pattern functions (obj, cons, jac, hess, hessc, fused sweep, structures)
//   exa_gen_scatter.cpp  scattering pattern functions (grad!, J'v, Hv), the gathered gradient, the gradient COO
//   exa_gen_module.cpp   generate_module: parameter layout, fused groups, the __global__ kernels
//   exa_gen_window.cpp   the second module of a model: owner-computes window kernels (compressed COO, products)
#pragma once
#include <algorithm>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "exa_internal.hpp"
#include "exa_traverse.hpp"

namespace exa {
namespace gen {

[[noreturn]] inline void fail(const std::string &m) { throw BadInput(m); }
inline int env_int(const char *name, int dflt) { const char *v = getenv(name); return v && *v ? atoi(v) : dflt; }

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

std::string fmt_double(double v);

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
// function rules (exa_gen_rules.cpp): (x, y, h) of a univariate; $1 = argument, $2 = primal f, $3 = first derivative
// ---------------------------------------------------------------------------------------------------
struct Triple { Val x, y, h; };
struct Six { Val x, y1, y2, h11, h12, h22; };
Triple un_rule(Emitter &e, int fn, Val u, int order);
Six bin_rule(Emitter &e, int fn, Val x1, Val x2, int order);
Triple fixed_rule(Emitter &e, int fn, int fixed, Val v, Val c, int order);
Val pow_any(Emitter &e, Val x1, Val x2);

// The generator keeps per-module state in file-level variables (g_handover, g_lit_idx, g_lds_need, ...): ONE lock for
// generate_module and generate_window_module (models may be built and compressed from several host threads).
extern std::mutex g_gen_mu;
extern const char *kPrelude;                 // exa_gen_prelude.cpp
extern const char *kSpecialPrelude;          // the SpecialFunctions routines (only for models that use them)
std::string prelude_text(const Model &m, const ParamLayout &L);   // kPrelude with its @TAGS@ filled in (+ kSpecialPrelude)

// ---------------------------------------------------------------------------------------------------
// per-pattern body generator
// ---------------------------------------------------------------------------------------------------
struct Affine { bool ok = false; int col = -1; int64_t a = 0, c = 0; };      // index expression == a * (RANGE column) + c ?
Affine affine(const Pattern &p, int k);                                        // exa_gen_scatter.cpp
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
    // x loads whose index is (unit-step range column) + literal: SSA name of the loaded value -> (column, literal); and
    // whether some x load is NOT of that form (staged chained kernels, ParamLayout::stage)
    std::map<std::string, std::pair<int, int64_t>> xoff;
    bool xother = false;
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
            if (!structure) {
                v.x = var_load(v.vidx);
                const Affine f = affine(p, t.ir);
                if (f.ok && f.col >= 0 && f.a == 1 && p.cols[f.col].step == 1 && v.x.k == Val::SF) xoff[e.s(v.x)] = {f.col, f.c};
                else xother = true;
            }
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
inline Val zero_seed(Body &b) { return b.e.raw("0.0", false); }

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

inline void emit_lines(std::ostringstream &os, const Emitter &e, const char *indent = "    ") {
    for (const auto &l : e.lines) os << indent << l << "\n";
}

// ---- load stage / evaluation stage of a pattern body (exa_gen_coo.cpp) ---------------------------------------------
struct Split {
    std::vector<std::string> load, eval;
    int nin = 0, nik = 0;
};
Split split_body(const Emitter &e);
extern std::map<std::pair<int, int>, std::pair<int, int>> g_handover;   // doubles / integers the load stage of (callback, pattern) hands over
constexpr int kChainTiles = 4;               // tiles per workgroup of exa_hessc (T = 2 / 8 / 16 measured: profiles/NOTES.md)

// ---- COO store staging (exa_gen_coo.cpp) -----------------------------------------------------------------------------
bool use_tile(int S);
int tile_pp(int S);
int tile_ld(int S);
int tile_doubles(int S);
void emit_coo_prologue(std::ostringstream &os, const Body &b, const ParamLayout &L, int pi, bool tile);
void emit_coo_stores(std::ostringstream &os, const Body &b, int word_o, int S, const std::vector<std::string> &vals, bool tile,
                     const std::string &out = "out", const std::string &tag = "", bool no_branch = false);
void emit_coo_stores_permuted(std::ostringstream &os, const Body &b, int word_o, int S, const std::vector<std::string> &vals, const std::string &tag);
inline std::string fn_name(int pi, const char *cb) { return "p" + std::to_string(pi) + "_" + cb; }

// pattern / group device functions
void gen_value_fn(std::ostringstream &os, const Model &m, int pi, const ParamLayout &L);
void gen_cons_fn(std::ostringstream &os, const Model &m, int pi, const ParamLayout &L);
void gen_cons_two_stage(std::ostringstream &os, const Model &m, int pi, const ParamLayout &L);      // pK_consL / pK_consE (exa_consl)
void gen_jac_group_two_stage(std::ostringstream &os, const Model &m, const ParamLayout &L, int gi); // gK_jacL / gK_jacE (exa_jacl)
void gen_dispatch_looped(std::ostringstream &os, const ParamLayout &L, int cb);                     // the pipelined tile loop of exa_consl / exa_jacl
void gen_hess_fn(std::ostringstream &os, const Model &m, int pi, const ParamLayout &L);
void gen_coo_group_fn(std::ostringstream &os, const Model &m, const ParamLayout &L, int cb, int gi, bool permuted = false);
void gen_fused_group_fn(std::ostringstream &os, const Model &m, const ParamLayout &L, int gi);
void gen_struct_fn(std::ostringstream &os, const Model &m, int pi, const ParamLayout &L, bool hess);
void gen_merged_hess_fn(std::ostringstream &os, const Model &m, const ParamLayout &L, int gi);
void gen_merged_struct_fn(std::ostringstream &os, const Model &m, const ParamLayout &L, int gi);
int merged_slot_count(const Model &m, const ParamLayout &L, const std::vector<int> &grp);
// looped: the body as a LOOP over `ppt` consecutive block-map entries of `nent` (kernel arguments of exa_jacl / exa_consl), one tile each
void gen_dispatch(std::ostringstream &os, const ParamLayout &L, int cb, const std::string &call_prefix, const std::string &call_args,
                  const std::string &tail_args = "", bool looped = false);
void gen_dispatch_chained(std::ostringstream &os, const ParamLayout &L, int cb, const char *name, bool hess);
void gen_dispatch_chained_staged(std::ostringstream &os, const Model &m, const ParamLayout &L);
bool pattern_stage(const Model &m, int pi, const ParamLayout &L, ParamLayout::Stage *out);      // ParamLayout::stage of one pattern


// ---- scattered `out[idx-1] += val` (grad of data-indexed patterns, J'v, Hv) ------------------------------------
// Three mechanisms, picked per target at generation time:
//   * literal index (same target for every data point)      -> wavefront butterfly + ONE atomic (exa_wave_atomic_add);
//   * index = (unit-step range value) + c for >= 2 targets   -> the wavefront's contributions fall into a window of
//     64 + span consecutive variables: accumulate them in LDS (ds_add_f64), then 64 + span global atomics instead of
//     64 per target (LV J'v: 192 -> 66 per wavefront);
//   * anything else                                           -> one FP64 hardware atomic per lane.
extern int g_lds_need[CB_COUNT];   // doubles of LDS per wavefront needed by the scatter windows of each callback (per module)
// literal scatter targets of every (callback, unit): 0-based variable indices, in the order of the unit's `lit[]`
extern std::map<std::pair<int, int>, std::vector<std::string>> g_lit_idx;
// largest scatter body (SSA lines) per callback, and the callbacks generated without loops because of it (see kHugeBody)
extern std::map<int, size_t> g_scatter_lines;
extern bool g_loopfree[CB_COUNT];
// Bodies of thousands of SSA values run at the 512-VGPR limit with scratch spills, and such kernels have produced wrong
// sums / memory faults whenever a loop sat around or inside the body (the 16-tile loop, the peeling loop of
// exa_scatter_add).  Past kHugeBody lines the generator emits no loop: one tile per workgroup, exa_scatter_add1.
constexpr int kHugeBody = 1000;

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
    int emit(std::vector<std::string> &lines, bool &full_wave);
};
void emit_scatter_prologue(std::ostringstream &os, const Body &b, const ParamLayout &L, int pi, bool full_wave);
void gen_first_fn(std::ostringstream &os, const Model &m, int pi, const ParamLayout &L, bool grad);
bool pull_ok(const Pattern &p, std::vector<Affine> &slots);
void gen_pull_fn(std::ostringstream &os, const Model &m, int pi, const ParamLayout &L);
void gen_jprod_fn(std::ostringstream &os, const Model &m, int pi, const ParamLayout &L);
void jtprod_items(Body &b, Scatter &sc);
void hprod_items(Body &b, Scatter &sc);
void gen_scatter_group_fn(std::ostringstream &os, const Model &m, const ParamLayout &L, int cb, int gi);
void gen_gradv_fn(std::ostringstream &os, const Model &m, int pi, const ParamLayout &L);

}  // namespace gen
}  // namespace exa
