// exa_gen_scatter.cpp — pattern functions whose outputs are SHARED between data points: grad! (gradient.jl:64-86,
// KA ext :310-336), J'v and Hv (jacobian.jl:55-68, hessian.jl:291-315, 566-579; KA ext :389-511), and J v.
// Mechanisms: gather per variable for affine objective patterns (exa_grad_pull), FP64 hardware atomics with per-wavefront
// LDS windows / register sums / lane peeling (exa_grad, exa_jtprod, exa_hprod), the gradient COO of the reference's scheme
// (exa_gradv).  The owner-computes window form of the products lives in exa_gen_window.cpp.
#include "exa_gen.hpp"

namespace exa {
namespace gen {

// index expression == a * (RANGE column) + c ?
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

int g_lds_need[CB_COUNT];
std::map<std::pair<int, int>, std::vector<std::string>> g_lit_idx;
std::map<int, size_t> g_scatter_lines;
bool g_loopfree[CB_COUNT];

int Scatter::emit(std::vector<std::string> &lines, bool &full_wave) {
    merge();
    // candidates: index = (unit-step range value) + c, all on the same range column; clustered into windows of
    // offsets that lie within 64 of each other (one window per variable block the pattern touches)
    struct Cand { int item; int64_t c; };
    std::vector<Cand> cand;
    int colword = -1;
    {
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
        if (items[k].vidx.is_lit()) {
            // same target for every data point: accumulate in a register across this thread's tiles; the kernel
            // adds it to memory ONCE per wavefront after the tile loop (pK_*_fin)
            full_wave = true;
            lines.push_back("lit[" + std::to_string(lit_idx.size()) + "] += act ? " + e.sd(items[k].val) + " : 0.0;");
            lit_idx.push_back(idx);
        } else if (!affine(*items[k].p, items[k].ir).ok) {
            // reached through a data column: possibly the same variable for the whole wavefront (exa_scatter_add)
            full_wave = true;
            const bool peel = !loopfree && (int)e.lines.size() <= kHugeBody;
            lines.push_back(std::string(peel ? "exa_scatter_add" : "exa_scatter_add1") + "(out, " + idx + ", " + e.sd(items[k].val) + ", act);");
        } else {
            lines.push_back("if (act) exa_atomic_add(&out[" + idx + "], " + e.sd(items[k].val) + ");");
        }
    }
    return W;
}

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
    // one-launch grad! (ParamLayout::gbits, as in the fused sweep): where the model build has proven that no variable is named twice by the
    // objective's items, each item IS its variable's gradient entry — a plain store (the zero tiles of the same launch cover the rest)
    std::vector<std::string> direct;
    if (grad && L.gbits >= 0)
        for (const Scatter::Item &it : sc.items) direct.push_back("if (act) out[" + b.e.s(b.e.sub(it.vidx, Emitter::liti(1))) + "] = " + b.e.sd(it.val) + ";");
    if (grad) emit_scatter_prologue(os, b, L, pi, full_wave || !direct.empty());
    else emit_coo_prologue(os, b, L, pi, tile);
    emit_lines(os, b.e);
    if (grad) {
        if (!direct.empty()) {
            os << "    if (P[" << L.gbits << "]) {\n";
            for (auto &s : direct) os << "        " << s << "\n";
            os << "    } else {\n";
        }
        for (auto &s : stores) os << "    " << s << "\n";
        if (!direct.empty()) os << "    }\n";
    }
    else emit_coo_stores(os, b, L.pat[pi].o1, p.o1step, vals, tile);
    os << "}\n";
}

// ---- gather ("pull") formulation of the objective gradient ------------------------------------------------
// An objective pattern can be gathered when every first-order slot's variable index is (range value) + c: variable
// v then receives slot s from exactly one data point, I = (v - c_s - start) / step.  One thread per VARIABLE
// re-evaluates the (cheap) pattern at those points: coalesced store, no zero-fill, no atomics, deterministic.
// (The reference's KA path resolves the same contention with a sorted gather list, KA ext :310-336.)
bool pull_ok(const Pattern &p, std::vector<Affine> &slots) {
    if (p.kind != EXA_PAT_OBJ || p.n == 0 || p.o1step < 1 || p.o1step > 4) return false;      // (each slot re-evaluates the pattern)
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

}  // namespace gen
void pull_point_range(const Pattern &p, int64_t v_lo, int64_t v_hi, int64_t *jlo, int64_t *jhi) {
    std::vector<gen::Affine> slots;
    *jlo = 0; *jhi = 0;
    if (!gen::pull_ok(p, slots) || v_hi < v_lo) return;
    auto fdiv = [](int64_t a, int64_t b) { int64_t q = a / b; return (a % b != 0 && a < 0) ? q - 1 : q; };   // b > 0
    int64_t lo = INT64_MAX, hi = INT64_MIN;
    for (const gen::Affine &s : slots) {
        const Column &c = p.cols[s.col];
        // variable = c.start + c.step * J + s.c  (s.a == 1, c.step >= 1)
        lo = std::min(lo, -fdiv(-(v_lo - s.c - c.start), c.step));
        hi = std::max(hi, fdiv(v_hi - s.c - c.start, c.step) + 1);
    }
    lo = std::max<int64_t>(lo, 0); hi = std::min(hi, p.n);
    if (hi > lo) { *jlo = lo; *jhi = hi; }
}
namespace gen {
void gen_pull_fn(std::ostringstream &os, const Model &m, int pi, const ParamLayout &L) {
    const Pattern &p = m.pats[pi];
    std::vector<Affine> slots;
    pull_ok(p, slots);
    os << "static __device__ __forceinline__ double " << fn_name(pi, "pull")
       << "(const long* __restrict__ P, const double* __restrict__ x, const double* __restrict__ th, long v) {\n    double g = 0.0;\n";
    // every slot is evaluated unconditionally at an index clamped into the shard and its contribution selected
    // afterwards: no data-dependent branch, so a thread handling several variables has all its loads in flight at once
    // (the points of the WHOLE pattern that touch this rank's variables, not the shard's own points: the variable's owner
    // evaluates whatever touches it; clamped indices stay inside that range, i.e. inside the stretch of x the rank holds)
    os << "    const long lo_ = " << Body(m, pi, L).P(L.pat[pi].qlo) << ", hi_ = " << Body(m, pi, L).P(L.pat[pi].qhi) << ";\n"
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

}  // namespace gen
}  // namespace exa
