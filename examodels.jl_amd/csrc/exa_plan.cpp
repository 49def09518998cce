// exa_plan.cpp — layout planner: from the pattern table to AD trees, slot maps and running offsets.
//
// Replaces the build-time half of src/simdfunction.jl:78-100 (sparsity probe with NaN inputs -> raw1/raw2 ->
// identity-dedup -> Compressor tuples, o1step/o2step) and the counter bookkeeping of src/nlp.jl:1474-1482
// (_add_obj), :1597-1611 (_add_con), :1730-1738 (_add_con!).  Instead of probing with NaNs it classifies every
// IR subtree statically (constant for AD <=> contains no VAR) and replays the traversal order symbolically.
#include <algorithm>
#include <map>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <stdexcept>

#include "exa_internal.hpp"
#include "exa_traverse.hpp"

namespace exa {

namespace {

[[noreturn]] void fail(const std::string &msg) { throw BadInput(msg); }

void check_pattern(const Pattern &p, int pi) {
    const int n = (int)p.nodes.size();
    auto bad = [&](const char *what, int k) {
        std::ostringstream os;
        os << "pattern " << pi << ": " << what << " at node " << k;
        fail(os.str());
    };
    if (p.root < 0 || p.root >= n) bad("root out of range", p.root);
    for (int k = 0; k < n; k++) {
        const exa_node_t &nd = p.nodes[k];
        switch (nd.op) {
        case EXA_OP_CONST_F: case EXA_OP_CONST_I: case EXA_OP_NULLV: break;
        case EXA_OP_DATA:
            if (nd.a < 0 || nd.a >= (int)p.cols.size()) bad("column id out of range", k);
            break;
        case EXA_OP_PAR: case EXA_OP_VAR: case EXA_OP_UN:
            if (nd.a < 0 || nd.a >= k) bad("child must precede parent", k);
            if (nd.op == EXA_OP_UN && (nd.fn < 0 || nd.fn >= EXA_U_COUNT) && !user_fn(false, nd.fn)) bad("unknown univariate function", k);
            break;
        case EXA_OP_BIN:
            if (nd.a < 0 || nd.a >= k || nd.b < 0 || nd.b >= k) bad("child must precede parent", k);
            if ((nd.fn < 0 || nd.fn >= EXA_B_COUNT) && !user_fn(true, nd.fn)) bad("unknown bivariate function", k);
            break;
        default: bad("unknown opcode", k);
        }
    }
}

// static classification: constant-for-AD, and Int-typed (Julia keeps Int and Float64 apart until promotion)
void classify(Pattern &p) {
    const int n = (int)p.nodes.size();
    p.isconst.assign(n, 1);
    p.isint.assign(n, 0);
    for (int k = 0; k < n; k++) {
        const exa_node_t &nd = p.nodes[k];
        switch (nd.op) {
        case EXA_OP_CONST_I: p.isint[k] = 1; break;
        case EXA_OP_DATA: p.isint[k] = p.cols[nd.a].type != EXA_COL_F64; break;
        case EXA_OP_VAR: p.isconst[k] = 0; break;
        case EXA_OP_UN:
            p.isconst[k] = p.isconst[nd.a];
            p.isint[k] = p.isint[nd.a] && (nd.fn == EXA_U_PLUS || nd.fn == EXA_U_MINUS || nd.fn == EXA_U_ABS || nd.fn == EXA_U_ABS2);
            break;
        case EXA_OP_BIN:
            p.isconst[k] = p.isconst[nd.a] && p.isconst[nd.b];
            p.isint[k] = p.isint[nd.a] && p.isint[nd.b] &&
                         (nd.fn == EXA_B_ADD || nd.fn == EXA_B_SUB || nd.fn == EXA_B_MUL || nd.fn == EXA_B_MAX || nd.fn == EXA_B_MIN);
            break;
        default: break;
        }
    }
}

// canonical text of an index expression: two VAR leaves are the same Jacobian/Hessian key iff their index
// trees are structurally identical (`===` on immutable node structs, simdfunction.jl:63-76)
void key_text(const Pattern &p, int k, std::ostringstream &os) {
    const exa_node_t &nd = p.nodes[k];
    switch (nd.op) {
    case EXA_OP_CONST_I: os << 'i' << nd.ival; return;
    case EXA_OP_CONST_F: { uint64_t b; std::memcpy(&b, &nd.fval, 8); os << 'f' << std::hex << b << std::dec; return; }
    case EXA_OP_DATA: os << 'd' << nd.a; return;
    default: break;
    }
    os << '(' << nd.op << '.' << nd.fn << ' ';
    key_text(p, nd.a, os);
    if (nd.op == EXA_OP_BIN) { os << ' '; key_text(p, nd.b, os); }
    os << ')';
}

int key_id(Pattern &p, int k) {
    std::ostringstream os;
    key_text(p, k, os);
    const std::string s = os.str();
    for (size_t i = 0; i < p.keys.size(); i++)
        if (p.keys[i] == s) return (int)i;
    p.keys.push_back(s);
    return (int)p.keys.size() - 1;
}

int build_ad(Pattern &p, int k) {
    const exa_node_t nd = p.nodes[k];
    ADNode a;
    a.ir = k;
    if (nd.op == EXA_OP_NULLV) {
        a.kind = AD_NULL;
    } else if (p.isconst[k]) {
        a.kind = AD_CONST;
    } else if (nd.op == EXA_OP_VAR) {
        if (!p.isint[nd.a]) fail("variable index expression is not integer-typed");
        a.kind = AD_VAR; a.ir = nd.a; a.key = key_id(p, nd.a);
    } else if (nd.op == EXA_OP_UN) {
        a.kind = AD_UN; a.fn = nd.fn; a.l = build_ad(p, nd.a);
    } else {   // BIN with at least one differentiable operand
        a.fn = nd.fn;
        const bool ca = p.isconst[nd.a] && p.nodes[nd.a].op != EXA_OP_NULLV;
        const bool cb = p.isconst[nd.b] && p.nodes[nd.b].op != EXA_OP_NULLV;
        if (cb) { a.kind = AD_UN; a.fixed = FX_SECOND; a.cir = nd.b; a.l = build_ad(p, nd.a); }        // register.jl:231-248
        else if (ca) { a.kind = AD_UN; a.fixed = FX_FIRST; a.cir = nd.a; a.l = build_ad(p, nd.b); }   // register.jl:249-266
        else { a.kind = AD_BIN; a.l = build_ad(p, nd.a); a.r = build_ad(p, nd.b); }                    // register.jl:209-230
    }
    p.ad.push_back(a);
    return (int)p.ad.size() - 1;
}

// dummy algebra: only the visit order matters
struct ListAlg {
    using T = char;
    const Pattern &p;
    std::vector<int> raw1;
    std::vector<std::pair<int, int>> raw2;       // ordered key pairs
    std::vector<int> leaf1s;
    std::vector<std::pair<int, int>> leaf2s;     // ordered AD-leaf pairs
    explicit ListAlg(const Pattern &pp) : p(pp) {}
    T y1(int) { return 0; } T y2(int) { return 0; } T h11(int) { return 0; } T h12(int) { return 0; } T h22(int) { return 0; }
    T mul(T, T) { return 0; } T add(T, T) { return 0; } T neg(T) { return 0; }
    void leaf1(int n, T) { raw1.push_back(p.ad[n].key); leaf1s.push_back(n); }
    void leaf2(int n1, int n2, T, bool) { raw2.push_back({p.ad[n1].key, p.ad[n2].key}); leaf2s.push_back({n1, n2}); }
};

template <class K>
int dedup(const std::vector<K> &raw, std::vector<int> &comp, std::vector<int> &first_visit) {
    std::vector<K> uniq;
    comp.clear();
    first_visit.clear();
    for (size_t i = 0; i < raw.size(); i++) {
        int f = -1;
        for (size_t j = 0; j < uniq.size(); j++)
            if (uniq[j] == raw[i]) { f = (int)j; break; }
        if (f < 0) { uniq.push_back(raw[i]); first_visit.push_back((int)i); f = (int)uniq.size() - 1; }
        comp.push_back(f + 1);
    }
    return (int)uniq.size();
}

void plan_pattern(Pattern &p, int pi) {
    check_pattern(p, pi);
    classify(p);
    if (p.kind == EXA_PAT_CONAUG) {
        if (p.target < 0 || p.target >= (int)p.nodes.size() || !p.isint[p.target] || !p.isconst[p.target])
            fail("augmentation target must be an integer expression of the data point");
    }
    p.ad.clear();
    p.keys.clear();
    p.ad_root = build_ad(p, p.root);
    ListAlg a1(p);
    grpass(p, p.ad_root, a1, 0);
    std::vector<int> fv;
    p.o1step = dedup(a1.raw1, p.comp1, fv);
    p.slotvar1.clear();
    for (int v : fv) p.slotvar1.push_back(a1.leaf1s[v]);
    ListAlg a2(p);
    hrpass0(p, p.ad_root, a2, 0, 0);
    p.o2step = dedup(a2.raw2, p.comp2, fv);
    p.slotvar2.clear();
    for (int v : fv) p.slotvar2.push_back(a2.leaf2s[v]);
}

// host evaluation of an integer-typed expression of the data point (index arithmetic only)
int64_t eval_int(const Pattern &p, int k, int64_t I) {
    const exa_node_t &nd = p.nodes[k];
    switch (nd.op) {
    case EXA_OP_CONST_I: return nd.ival;
    case EXA_OP_DATA: {
        const Column &c = p.cols[nd.a];
        return c.type == EXA_COL_RANGE ? c.start + c.step * I : c.idata[(size_t)I];
    }
    case EXA_OP_UN: {
        const int64_t a = eval_int(p, nd.a, I);
        switch (nd.fn) { case EXA_U_PLUS: return a; case EXA_U_MINUS: return -a; case EXA_U_ABS: return a < 0 ? -a : a; case EXA_U_ABS2: return a * a; }
        break;
    }
    case EXA_OP_BIN: {
        const int64_t a = eval_int(p, nd.a, I), b = eval_int(p, nd.b, I);
        switch (nd.fn) {
        case EXA_B_ADD: return a + b; case EXA_B_SUB: return a - b; case EXA_B_MUL: return a * b;
        case EXA_B_MAX: return a > b ? a : b; case EXA_B_MIN: return a < b ? a : b;
        }
        break;
    }
    default: break;
    }
    fail("non-integer node inside an index expression");
}

// ---- index bounds -----------------------------------------------------------------------------------------
// Every x[...] / theta[...] a kernel will read is checked against 1..nvar / 1..npar when the model is built: the
// reference checks concrete indices at build time (nlp.jl:990-995) and leaves symbolic ones unchecked; here an index
// comes from an external table and an out-of-range one would be an out-of-bounds device read.  Index expressions over
// ranges only (+, -, * by a constant) are monotone in the data point and are checked at both ends; anything reading a
// stored column is scanned point by point (budget: 4e8 node visits per model, beyond that the rest is left unchecked).
bool range_affine(const Pattern &p, int k, bool &has_point) {
    const exa_node_t &nd = p.nodes[k];
    switch (nd.op) {
    case EXA_OP_CONST_I: return true;
    case EXA_OP_DATA:
        if (p.cols[nd.a].type != EXA_COL_RANGE) return false;
        has_point = true;
        return true;
    case EXA_OP_UN:
        return (nd.fn == EXA_U_PLUS || nd.fn == EXA_U_MINUS) && range_affine(p, nd.a, has_point);
    case EXA_OP_BIN: {
        bool la = false, lb = false;
        if (!range_affine(p, nd.a, la) || !range_affine(p, nd.b, lb)) return false;
        if (nd.fn == EXA_B_MUL && la && lb) return false;      // point * point: not monotone in general
        if (nd.fn != EXA_B_ADD && nd.fn != EXA_B_SUB && nd.fn != EXA_B_MUL) return false;
        has_point = has_point || la || lb;
        return true;
    }
    default: return false;
    }
}
int tree_size(const Pattern &p, int k) {
    const exa_node_t &nd = p.nodes[k];
    if (nd.op == EXA_OP_UN) return 1 + tree_size(p, nd.a);
    if (nd.op == EXA_OP_BIN) return 1 + tree_size(p, nd.a) + tree_size(p, nd.b);
    return 1;
}
void check_index_bounds(const Model &m) {
    double budget = 4e8;
    for (size_t pk = 0; pk < m.pats.size(); pk++) {
        const Pattern &p = m.pats[pk];
        if (p.n == 0) continue;
        std::vector<std::pair<int, bool>> roots;    // (index expression root, is a parameter index)
        for (const exa_node_t &nd : p.nodes)
            if (nd.op == EXA_OP_VAR || nd.op == EXA_OP_PAR) {
                const std::pair<int, bool> r(nd.a, nd.op == EXA_OP_PAR);
                if (std::find(roots.begin(), roots.end(), r) == roots.end()) roots.push_back(r);
            }
        for (const auto &r : roots) {
            const int64_t limit = r.second ? m.npar : m.nvar;
            auto check = [&](int64_t I) {
                const int64_t v = eval_int(p, r.first, I);
                if (v < 1 || v > limit)
                    fail("pattern " + std::to_string(pk) + ": " + (r.second ? "parameter" : "variable") + " index " + std::to_string(v) +
                         " at data point " + std::to_string(I + 1) + " is outside 1.." + std::to_string(limit));
            };
            bool has_point = false;
            if (range_affine(p, r.first, has_point)) { check(0); check(p.n - 1); continue; }
            const double cost = (double)p.n * tree_size(p, r.first);
            if (cost > budget) continue;
            budget -= cost;
            for (int64_t I = 0; I < p.n; I++) check(I);
        }
    }
}

// Is the pattern's expression  c * x[index]  with c a literal or a Float64 data column (+x, -x, c*x, x*c)?  Then
// (index node, coefficient node or -1, sign) describe it.
bool linear_leaf(const Pattern &p, int k, int &var_idx, int &coef_node, double &lit) {
    const exa_node_t &nd = p.nodes[k];
    if (nd.op == EXA_OP_VAR) { var_idx = nd.a; coef_node = -1; lit = 1.0; return true; }
    if (nd.op == EXA_OP_UN && (nd.fn == EXA_U_MINUS || nd.fn == EXA_U_PLUS)) {
        if (p.nodes[nd.a].op != EXA_OP_VAR) return false;
        var_idx = p.nodes[nd.a].a; coef_node = -1; lit = nd.fn == EXA_U_MINUS ? -1.0 : 1.0;
        return true;
    }
    if (nd.op == EXA_OP_BIN && nd.fn == EXA_B_MUL) {
        for (int side = 0; side < 2; side++) {
            const int cv = side ? nd.b : nd.a, vv = side ? nd.a : nd.b;
            if (p.nodes[vv].op != EXA_OP_VAR) continue;
            const exa_node_t &c = p.nodes[cv];
            if (c.op == EXA_OP_CONST_F) { var_idx = p.nodes[vv].a; coef_node = -1; lit = c.fval; return true; }
            if (c.op == EXA_OP_CONST_I) { var_idx = p.nodes[vv].a; coef_node = -1; lit = (double)c.ival; return true; }
            if (c.op == EXA_OP_DATA && p.cols[c.a].type == EXA_COL_F64) { var_idx = p.nodes[vv].a; coef_node = cv; lit = 1.0; return true; }
        }
    }
    return false;
}

// sorted (target row -> contributing buffer entries) lists for the constraint augmentations
void build_aug_lists(Model &m) {
    if (m.nconaug == 0) return;
    {   // all terms of the form c * x[index]?  (evaluated per data point on the host, once)
        bool lin = true;
        for (const Pattern &p : m.pats) {
            if (p.kind != EXA_PAT_CONAUG || p.n == 0) continue;
            int vi, cn; double lit;
            lin = lin && linear_leaf(p, p.root, vi, cn, lit);
        }
        if (lin) {
            m.aug_var.resize((size_t)m.nconaug);
            m.aug_coef.resize((size_t)m.nconaug);
            for (const Pattern &p : m.pats) {
                if (p.kind != EXA_PAT_CONAUG || p.n == 0) continue;
                int vi = -1, cn = -1; double lit = 1.0;
                linear_leaf(p, p.root, vi, cn, lit);
                for (int64_t I = 0; I < p.n; I++) {
                    m.aug_var[(size_t)(p.oa + I)] = eval_int(p, vi, I) - 1;
                    m.aug_coef[(size_t)(p.oa + I)] = cn < 0 ? lit : p.cols[p.nodes[cn].a].fdata[(size_t)I];
                }
            }
        }
        m.aug_linear = lin;
    }
    std::vector<int64_t> row((size_t)m.nconaug);
    for (const Pattern &p : m.pats) {
        if (p.kind != EXA_PAT_CONAUG) continue;
        const int64_t nbase = m.pats[p.base].n;
        for (int64_t I = 0; I < p.n; I++) {
            const int64_t t = eval_int(p, p.target, I);
            if (t < 1 || t > nbase) fail("augmentation target row outside the base constraint block");
            row[(size_t)(p.oa + I)] = p.o0 + t - 1;
        }
    }
    m.aug_perm.resize((size_t)m.nconaug);
    for (int64_t q = 0; q < m.nconaug; q++) m.aug_perm[(size_t)q] = q;
    std::stable_sort(m.aug_perm.begin(), m.aug_perm.end(), [&](int64_t a, int64_t b) { return row[(size_t)a] < row[(size_t)b]; });
    for (int64_t j = 0; j < m.nconaug; j++) {
        const int64_t r = row[(size_t)m.aug_perm[(size_t)j]];
        if (m.aug_rows.empty() || m.aug_rows.back() != r) { m.aug_rows.push_back(r); m.aug_ptr.push_back(j); }
    }
    m.aug_ptr.push_back(m.nconaug);
}

template <class T>
std::vector<T> copy_or(const T *src, int64_t n, T fill) {
    std::vector<T> v((size_t)n, fill);
    if (src) std::memcpy(v.data(), src, sizeof(T) * (size_t)n);
    return v;
}

}  // namespace

// A locality-improving order of the data points of pattern pk (exa_locality_order): stable by the smallest variable any x[...] of
// the pattern names at the point — for a branch table: by from-bus, which is how a MATPOWER / PGLIB case file lists its branches.
// The gathers of the data-indexed kernels are line-granular (profiles/r3_acopf_topology.json: the bus-ordered ACOPF runs 35-40 %
// faster than the same network with its branches in random order).  Needs the host columns (a plan-only handle).
std::vector<int64_t> locality_keys(const Model &m, int pk) {
    const Pattern &p = m.pats[pk];
    std::vector<int> roots;
    for (const exa_node_t &nd : p.nodes)
        if (nd.op == EXA_OP_VAR && std::find(roots.begin(), roots.end(), nd.a) == roots.end()) roots.push_back(nd.a);
    std::vector<int64_t> key((size_t)p.n, INT64_MAX);
    for (int64_t I = 0; I < p.n; I++)
        for (int r : roots) key[(size_t)I] = std::min(key[(size_t)I], eval_int(p, r, I));
    return key;
}
bool scatter_bitmap(const Model &m, const std::vector<int> &pats, std::vector<uint64_t> &bits) {
    bits.assign((size_t)(m.nvar + 63) / 64, 0);
    for (int pk : pats) {
        const Pattern &p = m.pats[pk];
        // the distinct index EXPRESSIONS of the first-order slots (slots with the same expression are merged in registers, Scatter::merge)
        std::vector<int> roots;
        for (int s = 0; s < p.o1step; s++) {
            const int ir = p.ad[p.slotvar1[s]].ir;
            bool seen = false;
            for (int q : roots) seen = seen || p.ad[p.slotvar1[s]].key == p.ad[q].key;
            (void)ir;
            if (!seen) roots.push_back(p.slotvar1[s]);
        }
        for (int64_t I = 0; I < p.n; I++)
            for (int leaf : roots) {
                const int64_t v = eval_int(p, p.ad[leaf].ir, I) - 1;
                if (v < 0 || v >= m.nvar) return false;
                uint64_t &w = bits[(size_t)(v >> 6)];
                const uint64_t bit = 1ull << (v & 63);
                if (w & bit) return false;
                w |= bit;
            }
    }
    return true;
}
std::vector<int64_t> locality_order(const Model &m, int pk) {
    const Pattern &p = m.pats[pk];
    std::vector<int64_t> key = locality_keys(m, pk), perm((size_t)p.n);
    for (int64_t I = 0; I < p.n; I++) {
        if (key[(size_t)I] == INT64_MAX) key[(size_t)I] = I;      // (a pattern that reaches no variable keeps its order)
        perm[(size_t)I] = I;
    }
    std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return key[(size_t)a] < key[(size_t)b]; });
    return perm;
}

std::unique_ptr<Model> plan_model(const exa_model_desc_t *d) {
    if (!d) fail("null model description");
    if (d->nvar < 0 || d->npar < 0 || d->n_patterns < 0) fail("negative size in model description");
    auto m = std::make_unique<Model>();
    m->nvar = d->nvar; m->npar = d->npar; m->minimize = d->minimize;
    // a NULL start / bound vector means "the default constant" (exahip_ir.h) and stays unmaterialised (empty): at
    // N = 1e8 these six vectors would be 4.8 GB of host memory nobody reads
    auto keep = [](const double *src, int64_t n) { return src ? std::vector<double>(src, src + n) : std::vector<double>(); };
    m->x0 = keep(d->x0, d->nvar);
    m->lvar = keep(d->lvar, d->nvar);
    m->uvar = keep(d->uvar, d->nvar);
    m->theta = copy_or<double>(d->theta0, d->npar, 0.0);
    m->pats.resize(d->n_patterns);
    std::map<uint64_t, std::vector<std::pair<int, int>>> seen_cols;
    for (int k = 0; k < d->n_patterns; k++) {
        const exa_pattern_t &s = d->patterns[k];
        Pattern &p = m->pats[k];
        if (s.n < 0 || s.n_nodes <= 0 || !s.nodes) fail("pattern without nodes");
        p.kind = s.kind; p.root = s.root; p.target = s.target; p.base = s.base; p.n = s.n;
        p.nodes.assign(s.nodes, s.nodes + s.n_nodes);
        p.cols.resize(s.n_cols);
        for (int c = 0; c < s.n_cols; c++) {
            const exa_column_t &sc = s.cols[c];
            Column &col = p.cols[c];
            col.type = sc.type; col.start = sc.start; col.step = sc.step;
            if (sc.type == EXA_COL_I64) {
                if (!sc.data && s.n) fail("I64 column without data");
                col.idata.assign((const int64_t *)sc.data, (const int64_t *)sc.data + s.n);
            } else if (sc.type == EXA_COL_F64) {
                if (!sc.data && s.n) fail("F64 column without data");
                col.fdata.assign((const double *)sc.data, (const double *)sc.data + s.n);
            } else if (sc.type != EXA_COL_RANGE) {
                fail("unknown column type");
            }
        }
        // columns this model already holds (same type, same contents): alias them to the first copy
        for (int c = 0; c < s.n_cols; c++) {
            Column &col = p.cols[c];
            uint64_t hsh = 1469598103934665603ull ^ (uint64_t)col.type;
            auto mix = [&](const void *data, size_t bytes) {
                const unsigned char *q = (const unsigned char *)data;
                for (size_t i = 0; i < bytes; i++) { hsh ^= q[i]; hsh *= 1099511628211ull; }
            };
            const int64_t rng_[2] = {col.start, col.step};
            if (col.type == EXA_COL_RANGE) mix(rng_, sizeof rng_);
            else if (col.type == EXA_COL_I64) mix(col.idata.data(), 8 * col.idata.size());
            else mix(col.fdata.data(), 8 * col.fdata.size());
            mix(&p.n, sizeof p.n);
            auto &bucket = seen_cols[hsh];
            for (const auto &kc : bucket) {
                const Pattern &q = m->pats[kc.first];
                const Column &o = q.cols[kc.second];
                const bool same = o.type == col.type && q.n == p.n &&
                                  (col.type == EXA_COL_RANGE ? (o.start == col.start && o.step == col.step)
                                   : col.type == EXA_COL_I64 ? o.idata == col.idata : std::memcmp(o.fdata.data(), col.fdata.data(), 8 * col.fdata.size()) == 0);
                if (same) { col.alias_pat = kc.first; col.alias_col = kc.second; break; }
            }
            if (col.alias_pat < 0) bucket.push_back({k, c});
        }
        if (p.kind != EXA_PAT_OBJ && p.kind != EXA_PAT_CON && p.kind != EXA_PAT_CONAUG) fail("unknown pattern kind");
        if (p.kind == EXA_PAT_CONAUG) {
            if (p.base < 0 || p.base >= k || m->pats[p.base].kind != EXA_PAT_CON) fail("augmentation base must be an earlier CON pattern");
        }
        plan_pattern(p, k);
        // running counters, insertion order (nlp.jl:1474-1482, 1597-1611, 1730-1738)
        if (p.kind == EXA_PAT_OBJ) {
            p.o0 = m->nobj; p.o1 = m->nnzg; p.o2 = m->nnzh;
            m->nobj += p.n; m->nnzg += p.n * p.o1step; m->nnzh += p.n * p.o2step;
        } else if (p.kind == EXA_PAT_CON) {
            p.o0 = m->ncon; p.o1 = m->nnzj; p.o2 = m->nnzh;
            m->ncon += p.n; m->nnzj += p.n * p.o1step; m->nnzh += p.n * p.o2step;
        } else {
            p.o0 = m->pats[p.base].o0;   // offset0(c1, 0) (nlp.jl:1683)
            p.o1 = m->nnzj; p.o2 = m->nnzh; p.oa = m->nconaug;
            m->nconaug += p.n; m->nnzj += p.n * p.o1step; m->nnzh += p.n * p.o2step;
        }
    }
    build_aug_lists(*m);
    check_index_bounds(*m);
    m->y0 = keep(d->y0, m->ncon);
    m->lcon = keep(d->lcon, m->ncon);
    m->ucon = keep(d->ucon, m->ncon);
    return m;
}

}  // namespace exa
