// exa_recipe.cpp — recipes: parse the wire format (include/exahip_recipe.h), bind data through the builder ABI,
// evaluate the deferred sizes and hand the concrete pattern table to the planner.
//
// Role in the reference: `instantiate` of a core built against ArgSource placeholders (src/argument.jl:150-185,
// src/nlp.jl:809-863) + the generated builder of ExaModelsCompiler (ExaModelsCompiler.jl:1197-1330).  Nothing here
// is on the evaluation path: an instance is an ordinary model of exa_runtime.cpp afterwards.
#include <algorithm>
#include <cstring>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/exahip_recipe.h"
#include "exa_internal.hpp"

namespace exa {
namespace {

enum { SYM_CONST = 0, SYM_SCALAR, SYM_LEN, SYM_ADD, SYM_SUB, SYM_MUL, SYM_FLOORDIV, SYM_MAX0, SYM_NEG,
       // real-valued expressions of the sizes (a deferred coefficient such as 1 / (N + 1), ArgumentTest.jl:231-280)
       SYM_FCONST, SYM_ITOF, SYM_FADD, SYM_FSUB, SYM_FMUL, SYM_FDIV, SYM_FNEG };
inline bool sym_is_real(int op) { return op >= SYM_FCONST; }
enum { F_SCALAR = 0, F_ARRAY = 1, F_TABLE = 2 };
enum { T_I64 = 0, T_F64 = 1 };
enum { SRC_CONST = 0, SRC_INLINE, SRC_FIELD, SRC_COL };
enum { RC_RANGE = 0, RC_INLINE_I64, RC_INLINE_F64, RC_FIELD, RC_COL, RC_AXIS_RANGE, RC_AXIS_FIELD, RC_AXIS_INLINE_I64,
       RC_AXIS_INLINE_F64 };

struct RField {
    std::string name;
    int kind = 0, type = 0;
    std::vector<std::pair<std::string, int>> cols;
};
struct RSym { int op = 0; int64_t a = 0, b = 0; };
struct IVal { int sym = -1; int64_t v = 0; };
struct RSeg {
    IVal n;
    int src = SRC_CONST;
    double c = 0.0;
    std::vector<double> inl;
    int field = -1, col = -1;
};
struct RBlock {
    std::string name;
    int kind = 0;
    IVal off, len;
    std::vector<IVal> dims;
};
struct RCol {
    int kind = RC_RANGE;
    IVal a, b, c, d;
    std::vector<int64_t> idata;
    std::vector<double> fdata;
    int field = -1, col = -1;
};
struct RPattern {
    int kind = 0, root = -1, target = -1, base = -1;
    IVal n;
    std::vector<exa_node_t> nodes;
    std::vector<int> nodesym;
    std::vector<RCol> cols;
};
struct Recipe {
    int minimize = 1;
    std::vector<RField> fields;
    std::vector<RSym> syms;
    IVal nvar, npar;
    std::vector<RSeg> vec[7];      // x0 lvar uvar theta y0 lcon ucon
    std::vector<RBlock> blocks;
    std::vector<RPattern> pats;
    std::string schema, argtype;
};

// bound data of one instantiation
struct Slot {
    bool set = false;
    int64_t iscalar = 0;
    double fscalar = 0.0;
    std::vector<int64_t> iarr;
    std::vector<double> farr;
    std::vector<Slot> cols;        // table: one per column
    int64_t len() const { return (int64_t)std::max(iarr.size(), farr.size()); }
};
struct Builder {
    int recipe = 0;
    std::vector<Slot> slots;
};

std::mutex g_mu;
std::vector<std::unique_ptr<Recipe>> g_recipes;
std::vector<std::unique_ptr<Builder>> g_builders;

// ---- reader ---------------------------------------------------------------------------------------------
struct Reader {
    const unsigned char *p, *end;
    const std::vector<char> *real = nullptr;     // per size expression: is it real-valued? (set once the table is read)
    void need(size_t n) const { if ((size_t)(end - p) < n) throw BadInput("recipe: truncated"); }
    int32_t i32() { need(4); int32_t v; std::memcpy(&v, p, 4); p += 4; return v; }
    int64_t i64() { need(8); int64_t v; std::memcpy(&v, p, 8); p += 8; return v; }
    double f64() { need(8); double v; std::memcpy(&v, p, 8); p += 8; return v; }
    int count(int32_t limit = 1 << 28) { int32_t n = i32(); if (n < 0 || n > limit) throw BadInput("recipe: bad count"); return n; }
    std::string str() { int n = count(1 << 20); need((size_t)n); std::string s((const char *)p, (size_t)n); p += n; return s; }
    template <class T> std::vector<T> arr() {
        int64_t n = i64();
        if (n < 0 || (uint64_t)n > (uint64_t)(end - p) / sizeof(T)) throw BadInput("recipe: bad array length");
        std::vector<T> v((size_t)n);
        if (n) std::memcpy(v.data(), p, sizeof(T) * (size_t)n);
        p += sizeof(T) * (size_t)n;
        return v;
    }
    IVal ival(size_t nsyms) {
        IVal v;
        const int is = i32();
        const int64_t x = i64();
        if (is) { if (x < 0 || (uint64_t)x >= nsyms || (real && (*real)[(size_t)x])) throw BadInput("recipe: size expression out of range or not an integer"); v.sym = (int)x; }
        else v.v = x;
        return v;
    }
};

const char *tname(int t) { return t == T_I64 ? "i64" : "f64"; }

// may model files bring device code of their own? -1 = not said (the environment decides), 0 / 1 = exa_recipe_trust_code
std::atomic<int> g_trust_code{-1};
// ... and for ONE load on ONE thread (exa_recipe_load_trusted: the loader of a packed library, whose embedded recipe is native code
// already).  Not the process-wide switch: a concurrent exa_recipe_load of an untrusted file on another thread stays refused, and the
// process's own setting (unset / 0 / 1) is never touched.
thread_local int t_trusted_load = 0;
bool trust_model_code() {
    if (t_trusted_load > 0) return true;
    const int v = g_trust_code.load();
    if (v >= 0) return v > 0;
    const char *e = getenv("EXAHIP_TRUST_MODEL_CODE");
    return e && *e && std::string(e) != "0";
}

void describe(Recipe &r) {
    // schema JSON exactly in the shape the cnlp consumers parse (ExaModelsCompiler.jl:567-576)
    std::string j = "{\"fields\":[", a;
    for (size_t k = 0; k < r.fields.size(); k++) {
        const RField &f = r.fields[k];
        if (k) { j += ","; a += ","; }
        if (f.kind == F_TABLE) {
            j += "{\"name\":\"" + f.name + "\",\"kind\":\"table\",\"columns\":[";
            a += "Table{";
            for (size_t c = 0; c < f.cols.size(); c++) {
                if (c) { j += ","; a += " "; }
                j += "{\"name\":\"" + f.cols[c].first + "\",\"type\":\"" + tname(f.cols[c].second) + "\"}";
                a += f.cols[c].first + "::" + (f.cols[c].second == T_I64 ? "int" : "f64");
            }
            j += "]}";
            a += "}|" + f.name;
        } else {
            j += "{\"name\":\"" + f.name + "\",\"kind\":\"" + (f.kind == F_SCALAR ? "scalar" : "array") + "\",\"type\":\"" +
                 tname(f.type) + "\"}";
            // signature form of P_argtype (Compiler test :655-657): `int|arg1,Vector{f64}|v0,...`
            if (f.kind == F_SCALAR) a += std::string(f.type == T_I64 ? "int" : "f64") + "|" + f.name;
            else a += std::string("Vector{") + (f.type == T_I64 ? "int" : "f64") + "}|" + f.name;   // _jtype, Compiler :936-943
        }
    }
    j += "]}";
    r.schema = j;
    // a lone integer is the one-knob model: `int|size` (Compiler test :651)
    if (r.fields.size() == 1 && r.fields[0].kind == F_SCALAR && r.fields[0].type == T_I64) a = "int|size";
    r.argtype = a;
}

std::unique_ptr<Recipe> parse(const void *bytes, size_t len) {
    Reader rd{(const unsigned char *)bytes, (const unsigned char *)bytes + len};
    rd.need(8);
    if (std::memcmp(rd.p, "EXARCP01", 8) != 0) throw BadInput("recipe: bad magic");
    rd.p += 8;
    auto r = std::make_unique<Recipe>();
    r->minimize = rd.i32();
    const int nf = rd.count(1 << 16);
    for (int k = 0; k < nf; k++) {
        RField f;
        f.name = rd.str();
        f.kind = rd.i32();
        f.type = rd.i32();
        const int nc = rd.count(1 << 16);
        for (int c = 0; c < nc; c++) { std::string cn = rd.str(); int t = rd.i32(); f.cols.emplace_back(cn, t); }
        if (f.kind < F_SCALAR || f.kind > F_TABLE || (f.type != T_I64 && f.type != T_F64)) throw BadInput("recipe: bad field");
        for (auto &c : f.cols) if (c.second != T_I64 && c.second != T_F64) throw BadInput("recipe: bad column type");
        r->fields.push_back(std::move(f));
    }
    const int ns = rd.count();
    for (int k = 0; k < ns; k++) {
        RSym s;
        s.op = rd.i32(); s.a = rd.i64(); s.b = rd.i64();
        const bool leaf = s.op == SYM_CONST || s.op == SYM_FCONST, fld = s.op == SYM_SCALAR || s.op == SYM_LEN,
                   un = s.op == SYM_MAX0 || s.op == SYM_NEG || s.op == SYM_ITOF || s.op == SYM_FNEG;
        if (s.op < SYM_CONST || s.op > SYM_FNEG) throw BadInput("recipe: bad size opcode");
        if (fld && (s.a < 0 || s.a >= nf)) throw BadInput("recipe: size expression names an unknown field");
        if (fld && s.op == SYM_SCALAR && (r->fields[s.a].kind != F_SCALAR || r->fields[s.a].type != T_I64))
            throw BadInput("recipe: SCALAR size must be an i64 scalar field");
        if (fld && s.op == SYM_LEN && r->fields[s.a].kind == F_SCALAR) throw BadInput("recipe: LEN of a scalar field");
        if (!leaf && !fld && (s.a < 0 || s.a >= k || (!un && (s.b < 0 || s.b >= k)))) throw BadInput("recipe: size expression not in SSA order");
        if (!leaf && !fld) {
            // integer operators take integer operands; ITOF takes an integer; the real operators take reals
            const bool want_real = sym_is_real(s.op) && s.op != SYM_ITOF;
            if (sym_is_real(r->syms[s.a].op) != want_real || (!un && sym_is_real(r->syms[s.b].op) != want_real))
                throw BadInput("recipe: size expression mixes integer and real operands");
        }
        r->syms.push_back(s);
    }
    const size_t nsy = r->syms.size();
    std::vector<char> is_real(nsy);
    for (size_t k = 0; k < nsy; k++) is_real[k] = sym_is_real(r->syms[k].op);
    rd.real = &is_real;
    auto field_ok = [&](int f, int kind) { return f >= 0 && f < nf && r->fields[f].kind == kind; };
    r->nvar = rd.ival(nsy);
    r->npar = rd.ival(nsy);
    for (int v = 0; v < 7; v++) {
        const int n = rd.count();
        for (int k = 0; k < n; k++) {
            RSeg s;
            s.n = rd.ival(nsy);
            s.src = rd.i32();
            if (s.src == SRC_CONST) s.c = rd.f64();
            else if (s.src == SRC_INLINE) s.inl = rd.arr<double>();
            else if (s.src == SRC_FIELD) { s.field = rd.i32(); if (!field_ok(s.field, F_ARRAY)) throw BadInput("recipe: segment names a non-array field"); }
            else if (s.src == SRC_COL) {
                s.field = rd.i32(); s.col = rd.i32();
                if (!field_ok(s.field, F_TABLE) || s.col < 0 || s.col >= (int)r->fields[s.field].cols.size()) throw BadInput("recipe: segment names an unknown column");
            } else throw BadInput("recipe: bad segment source");
            r->vec[v].push_back(std::move(s));
        }
    }
    const int nb = rd.count(1 << 20);
    for (int k = 0; k < nb; k++) {
        RBlock b;
        b.name = rd.str();
        b.kind = rd.i32();
        b.off = rd.ival(nsy);
        b.len = rd.ival(nsy);
        const int nd = rd.count(64);
        for (int d = 0; d < nd; d++) b.dims.push_back(rd.ival(nsy));
        if (b.kind < 0 || b.kind > 2) throw BadInput("recipe: bad block kind");
        r->blocks.push_back(std::move(b));
    }
    const int np = rd.count(1 << 20);
    for (int k = 0; k < np; k++) {
        RPattern p;
        p.kind = rd.i32(); p.root = rd.i32(); p.target = rd.i32(); p.base = rd.i32();
        p.n = rd.ival(nsy);
        const int nn = rd.count();
        for (int i = 0; i < nn; i++) {
            exa_node_t nd{};
            nd.op = rd.i32(); nd.fn = rd.i32(); nd.a = rd.i32(); nd.b = rd.i32(); nd.fval = rd.f64(); nd.ival = rd.i64();
            const int sy = rd.i32();
            if (sy >= (int)nsy || (sy >= 0 && nd.op != (is_real[(size_t)sy] ? EXA_OP_CONST_F : EXA_OP_CONST_I)))
                throw BadInput("recipe: bad node size reference");
            p.nodes.push_back(nd);
            p.nodesym.push_back(sy);
        }
        const int nc = rd.count(1 << 20);
        for (int c = 0; c < nc; c++) {
            RCol col;
            col.kind = rd.i32();
            switch (col.kind) {
            case RC_RANGE: col.a = rd.ival(nsy); col.b = rd.ival(nsy); break;
            case RC_INLINE_I64: col.idata = rd.arr<int64_t>(); break;
            case RC_INLINE_F64: col.fdata = rd.arr<double>(); break;
            case RC_FIELD: col.field = rd.i32(); if (!field_ok(col.field, F_ARRAY)) throw BadInput("recipe: column names a non-array field"); break;
            case RC_COL:
                col.field = rd.i32(); col.col = rd.i32();
                if (!field_ok(col.field, F_TABLE) || col.col < 0 || col.col >= (int)r->fields[col.field].cols.size()) throw BadInput("recipe: column names an unknown table column");
                break;
            case RC_AXIS_RANGE: col.a = rd.ival(nsy); col.b = rd.ival(nsy); col.c = rd.ival(nsy); col.d = rd.ival(nsy); break;
            case RC_AXIS_FIELD:
                col.field = rd.i32(); col.col = rd.i32(); col.c = rd.ival(nsy); col.d = rd.ival(nsy);
                if (col.col < 0 ? !field_ok(col.field, F_ARRAY)
                                : (!field_ok(col.field, F_TABLE) || col.col >= (int)r->fields[col.field].cols.size()))
                    throw BadInput("recipe: axis names an unknown field");
                break;
            case RC_AXIS_INLINE_I64: col.idata = rd.arr<int64_t>(); col.c = rd.ival(nsy); col.d = rd.ival(nsy); break;
            case RC_AXIS_INLINE_F64: col.fdata = rd.arr<double>(); col.c = rd.ival(nsy); col.d = rd.ival(nsy); break;
            default: throw BadInput("recipe: bad column kind");
            }
            p.cols.push_back(std::move(col));
        }
        r->pats.push_back(std::move(p));
    }
    // optional trailing section: the user-registered functions the patterns use (exa_register_univariate / _bivariate), by the ids the
    // WRITER's process had given them.  Loading registers them here (the same rules again return the id they already have; a name that
    // is taken by other rules refuses the file) and renumbers the nodes, so a model file does not depend on what the loading process
    // registered before, nor in which order.
    // TRUST BOUNDARY: the rule texts and `helpers` are HIP device source that hipcc / hiprtc will compile and the GPU will run.  A file
    // whose entries are not, rule for rule, registrations this process has already made itself is therefore REFUSED unless the host
    // opted in (exa_recipe_trust_code(1), or EXAHIP_TRUST_MODEL_CODE=1 in the environment): without that a model file stays what it
    // was before files could carry registrations — validated data.
    struct Entry { int biv, fid; UserFn u; };
    std::vector<Entry> entries;
    std::map<std::pair<int, int>, int> remap;       // (bivariate, id in the file) -> id in this process
    if (rd.p != rd.end) {
        std::map<std::pair<int, std::string>, size_t> by_name;
        std::string fresh;
        const int nu = rd.count(1 << 16);
        for (int k = 0; k < nu; k++) {
            Entry en;
            en.biv = rd.i32(); en.fid = rd.i32();
            UserFn &u = en.u;
            u.name = rd.str(); u.f = rd.str(); u.d1 = rd.str(); u.d2 = rd.str(); u.d11 = rd.str(); u.d12 = rd.str(); u.d22 = rd.str();
            u.helpers = rd.str(); u.fused = rd.str();
            if ((en.biv != 0 && en.biv != 1) || en.fid < EXA_USER_FN_BASE || remap.count({en.biv, en.fid}) || by_name.count({en.biv, u.name}))
                throw BadInput("recipe: bad user-function entry");
            remap[{en.biv, en.fid}] = -1;
            by_name[{en.biv, u.name}] = entries.size();
            // every check of a registration, nothing entered yet: a file that is refused leaves the process as it was
            std::string err;
            bool known = false;
            if (register_user_fn(en.biv == 1, u, &err, true, &known) < 0) throw BadInput("recipe: user function: " + err);
            if (!known) fresh += (fresh.empty() ? "`" : ", `") + u.name + "`";
            entries.push_back(std::move(en));
        }
        if (rd.p != rd.end) throw BadInput("recipe: trailing bytes");
        if (!fresh.empty() && !trust_model_code())
            throw BadInput("recipe: the file carries device code (registered function " + fresh + ") that this process has not registered itself: "
                           "refused unless exa_recipe_trust_code(1) was called or EXAHIP_TRUST_MODEL_CODE=1 is set");
    }
    // (a file WITHOUT the section must not name registered functions at all: the ids are the writer's, and binding them to whatever
    // this process happens to have registered under those numbers could silently mean another function)
    for (const RPattern &p : r->pats)
        for (const exa_node_t &nd : p.nodes)
            if ((nd.op == EXA_OP_UN || nd.op == EXA_OP_BIN) && nd.fn >= EXA_USER_FN_BASE && !remap.count({nd.op == EXA_OP_BIN ? 1 : 0, nd.fn}))
                throw BadInput("recipe: a node uses a registered function the file does not define");
    describe(*r);
    // every check has passed: only now is anything entered into the process-wide tables
    for (const Entry &en : entries) {
        std::string err;
        const int id = register_user_fn(en.biv == 1, en.u, &err);
        if (id < 0) throw BadInput("recipe: user function: " + err);       // (another thread took the name in between)
        remap[{en.biv, en.fid}] = id;
    }
    for (RPattern &p : r->pats)
        for (exa_node_t &nd : p.nodes)
            if ((nd.op == EXA_OP_UN || nd.op == EXA_OP_BIN) && nd.fn >= EXA_USER_FN_BASE) nd.fn = remap[{nd.op == EXA_OP_BIN ? 1 : 0, nd.fn}];
    return r;
}

// ---- instantiation --------------------------------------------------------------------------------------
struct Instance {
    std::vector<double> vec[7];
    std::vector<std::vector<exa_node_t>> nodes;
    std::vector<std::vector<std::vector<int64_t>>> icol;
    std::vector<std::vector<std::vector<double>>> fcol;
    std::vector<std::vector<exa_column_t>> cols;
    std::vector<exa_pattern_t> pats;
    exa_model_desc_t desc{};
    std::vector<BlockInfo> blocks;
};

int64_t floordiv(int64_t a, int64_t b) {
    if (b == 0) throw BadInput("recipe: division by zero in a size expression");
    int64_t q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) q--;
    return q;
}

void instantiate(const Recipe &r, const Builder &B, Instance &I) {
    // 1. size expressions
    std::vector<int64_t> sv(r.syms.size());
    std::vector<double> sf(r.syms.size(), 0.0);       // values of the real-valued expressions
    for (size_t k = 0; k < r.syms.size(); k++) {
        const RSym &s = r.syms[k];
        switch (s.op) {
        case SYM_FCONST: std::memcpy(&sf[k], &s.a, 8); break;
        case SYM_ITOF: sf[k] = (double)sv[s.a]; break;
        case SYM_FADD: sf[k] = sf[s.a] + sf[s.b]; break;
        case SYM_FSUB: sf[k] = sf[s.a] - sf[s.b]; break;
        case SYM_FMUL: sf[k] = sf[s.a] * sf[s.b]; break;
        case SYM_FDIV: sf[k] = sf[s.a] / sf[s.b]; break;
        case SYM_FNEG: sf[k] = -sf[s.a]; break;
        case SYM_CONST: sv[k] = s.a; break;
        case SYM_SCALAR: sv[k] = B.slots[s.a].iscalar; break;
        case SYM_LEN: {
            const Slot &sl = B.slots[s.a];
            sv[k] = r.fields[s.a].kind == F_TABLE ? (sl.cols.empty() ? 0 : sl.cols[0].len()) : sl.len();
            break;
        }
        case SYM_ADD: sv[k] = sv[s.a] + sv[s.b]; break;
        case SYM_SUB: sv[k] = sv[s.a] - sv[s.b]; break;
        case SYM_MUL: sv[k] = sv[s.a] * sv[s.b]; break;
        case SYM_FLOORDIV: sv[k] = floordiv(sv[s.a], sv[s.b]); break;
        case SYM_MAX0: sv[k] = std::max<int64_t>(0, sv[s.a]); break;
        default: sv[k] = -sv[s.a]; break;
        }
    }
    auto ev = [&](const IVal &v) { return v.sym >= 0 ? sv[v.sym] : v.v; };
    auto field_f64 = [&](int field, int col, const char *what) {
        // a data field as doubles (start / bound vectors accept integer data too)
        const Slot &sl = col < 0 ? B.slots[field] : B.slots[field].cols[col];
        std::vector<double> out;
        if (!sl.farr.empty() || sl.iarr.empty()) out = sl.farr;
        else out.assign(sl.iarr.begin(), sl.iarr.end());
        (void)what;
        return out;
    };
    // 2. start / bound / parameter vectors
    static const char *vname[7] = {"x0", "lvar", "uvar", "theta", "y0", "lcon", "ucon"};
    for (int v = 0; v < 7; v++) {
        for (const RSeg &s : r.vec[v]) {
            const int64_t n = ev(s.n);
            if (n < 0) throw BadInput(std::string("recipe: negative block length in ") + vname[v]);
            if (s.src == SRC_CONST) I.vec[v].insert(I.vec[v].end(), (size_t)n, s.c);
            else {
                std::vector<double> d = s.src == SRC_INLINE ? s.inl : field_f64(s.field, s.src == SRC_COL ? s.col : -1, vname[v]);
                if ((int64_t)d.size() != n)
                    throw BadInput(std::string("instantiation data: ") + vname[v] + " block expects " + std::to_string(n) +
                                   " values, the bound field has " + std::to_string(d.size()));
                I.vec[v].insert(I.vec[v].end(), d.begin(), d.end());
            }
        }
    }
    const int64_t nvar = ev(r.nvar), npar = ev(r.npar);
    if ((int64_t)I.vec[0].size() != nvar || (int64_t)I.vec[3].size() != npar) throw BadInput("recipe: vector segments do not add up to nvar / npar");
    // 3. patterns
    const size_t np = r.pats.size();
    I.nodes.resize(np); I.icol.resize(np); I.fcol.resize(np); I.cols.resize(np); I.pats.resize(np);
    int64_t ncon = 0;
    for (size_t k = 0; k < np; k++) {
        const RPattern &p = r.pats[k];
        const int64_t n = ev(p.n);
        if (n < 0) throw BadInput("recipe: negative iterator length");
        I.nodes[k] = p.nodes;
        for (size_t i = 0; i < p.nodes.size(); i++)
            if (p.nodesym[i] >= 0) {
                if (p.nodes[i].op == EXA_OP_CONST_F) I.nodes[k][i].fval = sf[p.nodesym[i]];
                else I.nodes[k][i].ival = sv[p.nodesym[i]];
            }
        I.icol[k].resize(p.cols.size());
        I.fcol[k].resize(p.cols.size());
        for (size_t c = 0; c < p.cols.size(); c++) {
            const RCol &rc = p.cols[c];
            exa_column_t out{};
            std::vector<int64_t> &iv = I.icol[k][c];
            std::vector<double> &fv = I.fcol[k][c];
            auto bind = [&](const Slot &sl, int type) {       // a whole field / column, one value per data point
                if (type == T_I64) iv = sl.iarr; else fv = sl.farr;
                if ((int64_t)std::max(iv.size(), fv.size()) != n)
                    throw BadInput("instantiation data: an iterator column has " + std::to_string(std::max(iv.size(), fv.size())) +
                                   " entries, the pattern iterates " + std::to_string(n));
                out.type = type == T_I64 ? EXA_COL_I64 : EXA_COL_F64;
            };
            auto axis = [&](const std::vector<int64_t> *ai, const std::vector<double> *af, int64_t start, int64_t step) {
                const int64_t al = ev(rc.c), inner = ev(rc.d);
                if (al < 0 || inner <= 0) throw BadInput("recipe: bad product axis");
                if ((ai && (int64_t)ai->size() != al) || (af && (int64_t)af->size() != al)) throw BadInput("instantiation data: product axis length mismatch");
                if (af) { fv.resize((size_t)n); for (int64_t q = 0; q < n; q++) fv[q] = (*af)[(q / inner) % al]; out.type = EXA_COL_F64; }
                else {
                    iv.resize((size_t)n);
                    for (int64_t q = 0; q < n; q++) { const int64_t e = (q / inner) % al; iv[q] = ai ? (*ai)[e] : start + step * e; }
                    out.type = EXA_COL_I64;
                }
            };
            switch (rc.kind) {
            case RC_RANGE: out.type = EXA_COL_RANGE; out.start = ev(rc.a); out.step = ev(rc.b); break;
            case RC_INLINE_I64: iv = rc.idata; out.type = EXA_COL_I64; if ((int64_t)iv.size() != n) throw BadInput("recipe: inline column length mismatch"); break;
            case RC_INLINE_F64: fv = rc.fdata; out.type = EXA_COL_F64; if ((int64_t)fv.size() != n) throw BadInput("recipe: inline column length mismatch"); break;
            case RC_FIELD: bind(B.slots[rc.field], r.fields[rc.field].type); break;
            case RC_COL: bind(B.slots[rc.field].cols[rc.col], r.fields[rc.field].cols[rc.col].second); break;
            case RC_AXIS_RANGE: axis(nullptr, nullptr, ev(rc.a), ev(rc.b)); break;
            case RC_AXIS_FIELD: {
                const Slot &sl = rc.col < 0 ? B.slots[rc.field] : B.slots[rc.field].cols[rc.col];
                const int t = rc.col < 0 ? r.fields[rc.field].type : r.fields[rc.field].cols[rc.col].second;
                axis(t == T_I64 ? &sl.iarr : nullptr, t == T_F64 ? &sl.farr : nullptr, 0, 0);
                break;
            }
            case RC_AXIS_INLINE_I64: axis(&rc.idata, nullptr, 0, 0); break;
            default: axis(nullptr, &rc.fdata, 0, 0); break;
            }
            out.data = out.type == EXA_COL_I64 ? (const void *)iv.data() : out.type == EXA_COL_F64 ? (const void *)fv.data() : nullptr;
            I.cols[k].push_back(out);
        }
        exa_pattern_t &o = I.pats[k];
        o.kind = p.kind; o.n_nodes = (int)I.nodes[k].size(); o.nodes = I.nodes[k].data();
        o.root = p.root; o.target = p.target; o.base = p.base;
        o.n_cols = (int)I.cols[k].size(); o.cols = I.cols[k].data(); o.n = n;
        if (p.kind == EXA_PAT_CON) ncon += n;
    }
    if ((int64_t)I.vec[4].size() != ncon) throw BadInput("recipe: constraint vector segments do not add up to ncon");
    exa_model_desc_t &d = I.desc;
    d.nvar = nvar; d.npar = npar;
    d.x0 = I.vec[0].data(); d.lvar = I.vec[1].data(); d.uvar = I.vec[2].data(); d.theta0 = I.vec[3].data();
    d.n_patterns = (int)np; d.minimize = r.minimize; d.patterns = I.pats.data();
    d.y0 = I.vec[4].data(); d.lcon = I.vec[5].data(); d.ucon = I.vec[6].data();
    for (const RBlock &b : r.blocks) {
        BlockInfo o;
        o.name = b.name; o.kind = b.kind; o.offset = ev(b.off); o.length = ev(b.len);
        for (const IVal &v : b.dims) o.dims.push_back(ev(v));
        I.blocks.push_back(std::move(o));
    }
}

Recipe *get_recipe(int id) { return id >= 1 && id <= (int)g_recipes.size() ? g_recipes[id - 1].get() : nullptr; }
Builder *get_builder(int id) { return id >= 1 && id <= (int)g_builders.size() ? g_builders[id - 1].get() : nullptr; }

bool ready(const Recipe &r, const Builder &b) {
    for (size_t k = 0; k < r.fields.size(); k++) {
        const Slot &s = b.slots[k];
        if (r.fields[k].kind != F_TABLE) { if (!s.set) return false; continue; }
        for (size_t c = 0; c < s.cols.size(); c++) {
            if (!s.cols[c].set) return false;
            if (s.cols[c].len() != s.cols[0].len()) return false;   // columns must agree before rows can be reassembled
        }
    }
    return true;
}

int build(const Recipe &r, const Builder &b, bool device) {
    try {
        Instance I;
        instantiate(r, b, I);
        int id = 0;
        if (create_model(&I.desc, &id, device) != 0) return 0;      // last-error text already set
        attach_blocks(id, std::move(I.blocks));
        return id;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return 0;
    }
}

int copyout(const std::string &s, char *buf, int cap) {
    const int n = (int)s.size(), c = std::min(cap, n);
    if (c > 0 && buf) std::memcpy(buf, s.data(), (size_t)c);
    return n;
}

int new_scalar(int recipe, int n, bool device) {
    std::lock_guard<std::mutex> lk(g_mu);
    Recipe *r = get_recipe(recipe);
    if (!r) return 0;
    Builder b;
    b.slots.resize(r->fields.size());
    if (r->fields.empty()) return build(*r, b, device);
    if (r->fields.size() == 1 && r->fields[0].kind == F_SCALAR && r->fields[0].type == T_I64) {
        b.slots[0].set = true;
        b.slots[0].iscalar = n;
        return build(*r, b, device);
    }
    set_last_error("this recipe instantiates through the builder (exa_data_begin ... exa_new_from_data)");
    return 0;
}

// locate (field [, column]) by name with the expected kind/type: 0 ok, 1 unknown
int find(const Recipe &r, const char *field, const char *column, int kind, int type, int *fo, int *co) {
    if (!field) return 1;
    for (size_t k = 0; k < r.fields.size(); k++) {
        const RField &f = r.fields[k];
        if (f.name != field || f.kind != kind) continue;
        if (kind != F_TABLE) { if (f.type != type) return 1; *fo = (int)k; return 0; }
        if (!column) return 1;
        for (size_t c = 0; c < f.cols.size(); c++)
            if (f.cols[c].first == column && f.cols[c].second == type) { *fo = (int)k; *co = (int)c; return 0; }
    }
    return 1;
}

template <class F> int with_builder(int builder, F &&f) {
    std::lock_guard<std::mutex> lk(g_mu);
    Builder *b = get_builder(builder);
    Recipe *r = b ? get_recipe(b->recipe) : nullptr;
    if (!b || !r) return 1;
    try { return f(*r, *b); }
    catch (const std::exception &e) { set_last_error(e.what()); return 2; }
}

}  // namespace
}  // namespace exa

using namespace exa;

extern "C" {

int exa_recipe_trust_code(int on) {
    const int before = trust_model_code() ? 1 : 0;
    if (on >= 0) g_trust_code.store(on ? 1 : 0);
    return before;
}
int exa_recipe_load_trusted(const void *bytes, size_t len) {
    struct Scope { Scope() { t_trusted_load++; } ~Scope() { t_trusted_load--; } } scope;
    return exa_recipe_load(bytes, len);
}
int exa_recipe_load(const void *bytes, size_t len) {
    if (!bytes) return 0;
    try {
        auto r = parse(bytes, len);
        std::lock_guard<std::mutex> lk(g_mu);
        for (size_t i = 0; i < g_recipes.size(); i++)
            if (!g_recipes[i]) { g_recipes[i] = std::move(r); return (int)i + 1; }
        g_recipes.push_back(std::move(r));
        return (int)g_recipes.size();
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return 0;
    }
}
int exa_recipe_free(int recipe) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!get_recipe(recipe)) return 1;
    g_recipes[recipe - 1].reset();
    return 0;
}
int exa_recipe_nargs(int recipe) {
    std::lock_guard<std::mutex> lk(g_mu);
    Recipe *r = get_recipe(recipe);
    return r ? (int)r->fields.size() : -1;
}
int exa_recipe_argtype(int recipe, char *buf, int cap) {
    std::lock_guard<std::mutex> lk(g_mu);
    Recipe *r = get_recipe(recipe);
    return r ? copyout(r->argtype, buf, cap) : -1;
}
int exa_recipe_schema(int recipe, char *buf, int cap) {
    std::lock_guard<std::mutex> lk(g_mu);
    Recipe *r = get_recipe(recipe);
    return r ? copyout(r->schema, buf, cap) : -1;
}
int exa_recipe_new(int recipe, int n) { return new_scalar(recipe, n, true); }
int exa_recipe_plan(int recipe, int n) { return new_scalar(recipe, n, false); }

int exa_data_begin(int recipe) {
    std::lock_guard<std::mutex> lk(g_mu);
    Recipe *r = get_recipe(recipe);
    if (!r) return 0;
    auto b = std::make_unique<Builder>();
    b->recipe = recipe;
    b->slots.resize(r->fields.size());
    for (size_t k = 0; k < r->fields.size(); k++) b->slots[k].cols.resize(r->fields[k].cols.size());
    for (size_t i = 0; i < g_builders.size(); i++)
        if (!g_builders[i]) { g_builders[i] = std::move(b); return (int)i + 1; }
    g_builders.push_back(std::move(b));
    return (int)g_builders.size();
}
int exa_data_free(int builder) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!get_builder(builder)) return 1;
    g_builders[builder - 1].reset();
    return 0;
}
int exa_set_scalar_i64(int builder, const char *field, int64_t v) {
    return with_builder(builder, [&](Recipe &r, Builder &b) {
        int f = -1, c = -1;
        if (find(r, field, nullptr, F_SCALAR, T_I64, &f, &c)) return 1;
        b.slots[f].iscalar = v; b.slots[f].set = true;
        return 0;
    });
}
int exa_set_scalar_f64(int builder, const char *field, double v) {
    return with_builder(builder, [&](Recipe &r, Builder &b) {
        int f = -1, c = -1;
        if (find(r, field, nullptr, F_SCALAR, T_F64, &f, &c)) return 1;
        b.slots[f].fscalar = v; b.slots[f].set = true;
        return 0;
    });
}
int exa_set_array_i64(int builder, const char *field, const int64_t *v, int len) {
    return with_builder(builder, [&](Recipe &r, Builder &b) {
        int f = -1, c = -1;
        if (len < 0 || (len && !v) || find(r, field, nullptr, F_ARRAY, T_I64, &f, &c)) return 1;
        b.slots[f].iarr.assign(v, v + len); b.slots[f].set = true;
        return 0;
    });
}
int exa_set_array_f64(int builder, const char *field, const double *v, int len) {
    return with_builder(builder, [&](Recipe &r, Builder &b) {
        int f = -1, c = -1;
        if (len < 0 || (len && !v) || find(r, field, nullptr, F_ARRAY, T_F64, &f, &c)) return 1;
        b.slots[f].farr.assign(v, v + len); b.slots[f].set = true;
        return 0;
    });
}
int exa_set_col_i64(int builder, const char *table, const char *column, const int64_t *v, int len) {
    return with_builder(builder, [&](Recipe &r, Builder &b) {
        int f = -1, c = -1;
        if (len < 0 || (len && !v) || find(r, table, column, F_TABLE, T_I64, &f, &c)) return 1;
        b.slots[f].cols[c].iarr.assign(v, v + len); b.slots[f].cols[c].set = true;
        return 0;
    });
}
int exa_set_col_f64(int builder, const char *table, const char *column, const double *v, int len) {
    return with_builder(builder, [&](Recipe &r, Builder &b) {
        int f = -1, c = -1;
        if (len < 0 || (len && !v) || find(r, table, column, F_TABLE, T_F64, &f, &c)) return 1;
        b.slots[f].cols[c].farr.assign(v, v + len); b.slots[f].cols[c].set = true;
        return 0;
    });
}
int exa_data_ready(int builder) {
    std::lock_guard<std::mutex> lk(g_mu);
    Builder *b = get_builder(builder);
    Recipe *r = b ? get_recipe(b->recipe) : nullptr;
    return b && r && ready(*r, *b) ? 1 : 0;
}
static int from_data(int builder, bool device) {
    std::lock_guard<std::mutex> lk(g_mu);
    Builder *b = get_builder(builder);
    Recipe *r = b ? get_recipe(b->recipe) : nullptr;
    if (!b || !r) return 0;
    if (!ready(*r, *b)) { set_last_error("builder: not every schema field is set (or a table's columns differ in length)"); return 0; }
    return build(*r, *b, device);
}
int exa_new_from_data(int builder) { return from_data(builder, true); }
int exa_plan_from_data(int builder) { return from_data(builder, false); }

}  // extern "C"
