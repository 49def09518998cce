// exa_gen_coo.cpp — pattern functions that evaluate values or write COO slots: obj / cons_nln! values, jac_coord! and
// hess_coord! (plain and chained), the fused sweep, the structure kernels; and the dispatch of a fused launch.
//
// What it replaces: the KernelAbstractions kernels kerf/kerf2/kerj/kerh/kerh2 of ext/ExaModelsKernelAbstractions.jl:608-684
// together with Julia's type-specialisation of the whole expression tree into each of them.  Design differences
// (MI355X-first, not a translation):
//   * one FUSED launch per callback for the whole model: blockIdx -> (pattern, data-point tile) through a block map;
//   * every COO slot is accumulated in a VGPR in the reference's contribution order and stored ONCE (no zero-fill, no
//     read-modify-write on HBM: KA ext :521,:533 + hessian.jl:580-592 do fill! and `+=`);
//   * the iterator is struct-of-arrays, lane I reads column[I] (coalesced); UnitRange iterators cost no load;
//   * forward sweep, partials and reverse sweep are symbolic here: constants fold, x*1 / x+0 vanish, common
//     sub-expressions (one sincos per argument, exp reused for f=f'=f'') are shared by construction.
#include "exa_gen.hpp"

namespace exa {
namespace gen {

// ---- load stage / evaluation stage of a pattern body (chained, software-pipelined callbacks) ------------------------
// The SSA lines of a body are split into what touches memory — the loads of x, y, theta and the iterator columns,
// together with the integer index arithmetic they need — and the arithmetic that consumes the loaded values.  The
// load stage hands its values over in `in[]` (doubles) and `ik[]` (integers read from data columns); the evaluation
// stage re-derives the pure index arithmetic (scalar / cheap integer work) and reads everything else from there.
static bool is_memory_read(const std::string &expr) {
    for (const char *pre : {"x[", "y[", "th[", "v[", "((const long*)P[", "((const double*)P["})
        if (expr.compare(0, strlen(pre), pre) == 0) return true;
    return false;
}
// staged != null: x loads listed there (name -> LDS index expression: stretch base + the cluster's lane offset + literal distance) are
// NOT loaded: the evaluation stage reads them from the wavefront's staged stretches, xs[...]
// line_marks / eval_marks: for every line index in line_marks (ascending), the number of evaluation-stage lines emitted before that line
// (a fused group's function stores a pattern's slots right behind the lines that complete them)
static Split split_body_staged(const Emitter &e, const std::map<std::string, std::string> *staged, const std::vector<size_t> *line_marks = nullptr,
                               std::vector<size_t> *eval_marks = nullptr);
Split split_body(const Emitter &e) { return split_body_staged(e, nullptr); }
static Split split_body_staged(const Emitter &e, const std::map<std::string, std::string> *staged, const std::vector<size_t> *line_marks,
                               std::vector<size_t> *eval_marks) {
    Split sp;
    size_t d = 0, mk = 0;
    for (size_t li = 0; li <= e.lines.size(); li++) {
        while (line_marks && mk < line_marks->size() && (*line_marks)[mk] == li) { eval_marks->push_back(sp.eval.size()); mk++; }
        if (li == e.lines.size()) break;
        const std::string &line = e.lines[li];
        if (d < e.defs.size() && e.defs[d].line == (int)li) {
            const Emitter::Def &df = e.defs[d++];
            if (staged && df.expr.compare(0, 2, "x[") == 0 && staged->count(df.name)) {
                sp.eval.push_back("const double " + df.name + " = xs[" + staged->at(df.name) + "];");
            } else if (is_memory_read(df.expr)) {
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
std::map<std::pair<int, int>, std::pair<int, int>> g_handover;

// LDS budget per 256-thread workgroup for the store staging
static constexpr int kLdsBudget = 40960;
// a pattern too wide to stage even 8 points per pass within the 160 KB of LDS falls back to direct per-lane stores
bool use_tile(int S) { return S >= 2 && (long)(kBlock / 64) * tile_doubles(S) * 8 <= env_int("EXAHIP_LDS_MAX", 150000); }
// points staged per pass: the largest of 64/32/16/8 whose tile (4 wavefronts) fits the budget
int tile_pp(int S) {
    for (int pp = 64; pp > 8; pp >>= 1)
        if ((kBlock / 64) * S * (pp + 1) * 8 <= kLdsBudget) return pp;
    return 8;
}
// Leading dimension of the slot-major tile.  Writes (lane-consecutive) are conflict-free for any LD; the transposed
// read of lane j fetches element (j % S) * LD + j / S, and a ds_read_b64 is serviced per 32-lane half with 32 bank
// pairs (MI355X_MICROARCH.md §LDS) — pick the LD in [PP, PP+32] with the fewest extra cycles.
int tile_ld(int S) {
    const int pp = tile_pp(S), cnt = S * pp;
    int best = pp + 1;
    long best_cost = -1;
    // (the search is NOT bounded by the tile's footprint: the conflict-free LD = 47 of the rocket's 47-slot pattern is a 70 KB tile, two workgroups
    // per CU where the registers allow three; LD = 17 gives the third and LOSES — hess_coord! 0.0881 -> 0.0917 ms, profiles/r5_rocket_tile_ld_ab.txt)
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
                     const std::string &out, const std::string &tag, bool no_branch) {
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
static std::string index_key(const Pattern &p, const ParamLayout &L, int pi, int k) {
    const exa_node_t &nd = p.nodes[k];
    switch (nd.op) {
    case EXA_OP_CONST_I: return "i" + std::to_string(nd.ival);
    case EXA_OP_DATA: return "c" + std::to_string(L.pat[pi].col[nd.a]);
    case EXA_OP_UN: return "u" + std::to_string(nd.fn) + "(" + index_key(p, L, pi, nd.a) + ")";
    case EXA_OP_BIN: return "b" + std::to_string(nd.fn) + "(" + index_key(p, L, pi, nd.a) + "," + index_key(p, L, pi, nd.b) + ")";
    default: return "?" + std::to_string(pi) + ":" + std::to_string(k);      // never equal to anything of another pattern
    }
}
namespace {
struct MergedSlot { std::string key; Val ia, ib, sum; bool has = false; };
}  // namespace
// slot s of pattern b.p (accumulated value `acc`, or structure only) joins the merged slot of its pair
static void merge_slot(std::vector<MergedSlot> &ms, Body &b, int s, const Val *acc) {
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
    else os << "    __builtin_nontemporal_store(v, &c[" << b.P(L.pat[pi].o0) << " + I]);\n";
    os << "}\n";
}

// cons_nln! value in two stages for the pipelined tile loop (exa_consl): pK_consL issues the loads of point tid (clamped into the shard) and
// hands them over, pK_consE evaluates from the hand-over registers.  The value is the same expression as pK_val: bitwise equal.
void gen_cons_two_stage(std::ostringstream &os, const Model &m, int pi, const ParamLayout &L) {
    Body b(m, pi, L);
    Val v = b.e.tod(b.cval(b.p.root));
    const Split sp = split_body(b.e);
    g_handover[{CB_CONS, pi}] = {sp.nin, sp.nik};
    const std::string head = "    const long I_ = " + b.P(L.pat[pi].lo) + " + tid, hi_ = " + b.P(L.pat[pi].hi) + ";\n    const long I = I_ < hi_ ? I_ : (hi_ > 0 ? hi_ - 1 : 0);\n";
    os << "static __device__ __forceinline__ void " << fn_name(pi, "consL")
       << "(const long* __restrict__ P, const double* __restrict__ x, const double* __restrict__ th, long tid, double* in, long* ik) {\n" << head;
    for (const auto &l : sp.load) os << "    " << l << "\n";
    os << "    (void)I;\n}\n";
    os << "static __device__ __forceinline__ double " << fn_name(pi, "consE")
       << "(const long* __restrict__ P, const double* in, const long* ik, long tid) {\n" << head;
    for (const auto &l : sp.eval) os << "    " << l << "\n";
    os << "    (void)I;\n    return " << b.e.s(v) << ";\n}\n";
}

// COO-writing pattern function in two stages (see split_body): pK_<cb>L loads, pK_<cb>E evaluates and stores.
static void emit_two_stage(std::ostringstream &os, Body &b, const ParamLayout &L, int pi, int cb, const char *name, bool hess, bool tile,
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
    emit_coo_stores(os, b, word_o, S, vals, tile, "out", "", true);
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
    if (L.chain[CB_HESSC] > 0 && L.staged) {
        // the staged pair (exa_hesscl): pK_hessclL loads what is NOT x (multipliers, table columns, parameters), pK_hessclE
        // takes its x operands from the staged stretch
        std::map<std::string, std::string> off;
        for (const auto &kv : b.xoff) {
            const auto &cls = L.stage[pi].cl;
            for (size_t c = 0; c < cls.size(); c++)
                if (kv.second.second >= cls[c].cmin && kv.second.second <= cls[c].cmax)
                    off[kv.first] = std::to_string(cls[c].stretch * (64 + kStageHalo)) + " + xd[" + std::to_string(c) + "] + " + std::to_string(kv.second.second - cls[c].cmin);
        }
        const Split sp = split_body_staged(b.e, &off);
        g_handover[{CB_COUNT, pi}] = {sp.nin, sp.nik};
        os << "static __device__ __forceinline__ void " << fn_name(pi, "hesscl") << "L(const long* __restrict__ P, const double* __restrict__ x, const double* __restrict__ y, "
           << "const double* __restrict__ th, long tid, double* in, long* ik) {\n"
           << "    const long I0 = " << b.P(L.pat[pi].lo) << " + tid;\n    const long hi = " << b.P(L.pat[pi].hi) << ";\n"
           << "    const long I = I0 < hi ? I0 : (hi > 0 ? hi - 1 : 0);\n";
        for (const auto &l : sp.load) os << "    " << l << "\n";
        os << "    (void)I;\n}\n";
        os << "static __device__ __forceinline__ void " << fn_name(pi, "hesscl") << "E(const long* __restrict__ P, const double* in, const long* ik, "
           << "const double* xs, const int* xd, double* __restrict__ out, double* __restrict__ sink, double sigma, long tid, double* lds) {\n"
           << "    const long I0 = " << b.P(L.pat[pi].lo) << " + tid;\n    const long hi = " << b.P(L.pat[pi].hi) << ";\n"
           << "    const int lane = threadIdx.x & 63;\n    const long I = I0 < hi ? I0 : (hi > 0 ? hi - 1 : 0);\n";
        for (const auto &l : sp.eval) os << "    " << l << "\n";
        emit_coo_stores(os, b, L.pat[pi].o2, p.o2step, vals, tile, "out", "", true);
        os << "}\n";
    }
}

// ParamLayout::stage of pattern pi: the x indices of its second-order body of the form (ONE unit-step range column) + literal, their
// literals clustered (a new cluster where the gap to the cluster's first literal exceeds the halo: another variable block).  A pattern
// without such an index (only literal / data-indexed reads) qualifies with no cluster: all its loads stay in the load stage.  false:
// two different range columns, or more than 8 clusters.
bool pattern_stage(const Model &m, int pi, const ParamLayout &L, ParamLayout::Stage *out) {
    Body b(m, pi, L);
    const Pattern &p = b.p;
    b.forward(p.ad_root, 2, false);
    *out = ParamLayout::Stage();
    int col = -1;
    std::vector<int64_t> lits;
    for (const auto &kv : b.xoff) {
        if (col >= 0 && kv.second.first != col) return false;
        col = kv.second.first;
        if (std::find(lits.begin(), lits.end(), kv.second.second) == lits.end()) lits.push_back(kv.second.second);
    }
    if (col < 0) return true;
    std::sort(lits.begin(), lits.end());
    out->word = L.pat[pi].col[col];
    for (int64_t v : lits) {
        if (out->cl.empty() || v - out->cl.back().cmin > kStageHalo) { ParamLayout::Stage::Cluster c; c.cmin = c.cmax = v; out->cl.push_back(c); }
        else out->cl.back().cmax = v;
    }
    return out->cl.size() <= 8;
}

// jac_coord! / hess_coord! (exa_jac / exa_hess): one device function per FUSED GROUP — the patterns of exactly the same
// length, evaluated by thread I one after the other inside ONE emitter (shared loads, shared gathers, one sincos per
// argument for all of them), each pattern's slots staged and flushed to ITS OWN contiguous COO range as soon as they are
// complete (so only one pattern's values are live at a time).  A singleton group is the plain per-pattern function.
void gen_coo_group_fn(std::ostringstream &os, const Model &m, const ParamLayout &L, int cb, int gi, bool permuted) {
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

// jac_coord! group function in two stages for the pipelined tile loop (exa_jacl): gK_jacL = the loads of the whole fused group at point tid,
// gK_jacE = the arithmetic and the COO stores of every pattern of the group, each pattern's stores right behind the lines that complete its
// slots (as in gK_jac).  Lanes and whole wavefronts beyond the shard evaluate its last point and store nothing (the flush's range check).
void gen_jac_group_two_stage(std::ostringstream &os, const Model &m, const ParamLayout &L, int gi) {
    const int cb = CB_JAC;
    const auto &grp = L.groups[cb][gi];
    Emitter E;
    std::vector<size_t> line_marks, eval_marks;
    struct St { int pk, S, word; std::vector<std::string> vals; };
    std::vector<St> sts;
    for (int pk : grp) {
        Body b(m, pk, L, &E);
        const Pattern &p = b.p;
        b.forward(p.ad_root, 1, false);
        GenAlg a(b, p.comp1, p.o1step);
        grpass(p, p.ad_root, a, Emitter::litf(1.0));
        St st{pk, p.o1step, L.pat[pk].o1, {}};
        for (int s2 = 0; s2 < st.S; s2++) st.vals.push_back(E.sd(a.acc[s2]));
        sts.push_back(st);
        line_marks.push_back(E.lines.size());
    }
    const Split sp = split_body_staged(E, nullptr, &line_marks, &eval_marks);
    g_handover[{cb, gi}] = {sp.nin, sp.nik};
    Body b0(m, grp.front(), L);
    const std::string head = "    const long I0 = " + b0.P(L.pat[grp.front()].lo) + " + tid;\n    const long hi = " + b0.P(L.pat[grp.front()].hi) +
                             ";\n    const long I = I0 < hi ? I0 : (hi > 0 ? hi - 1 : 0);\n";
    os << "static __device__ __forceinline__ void g" << gi << "_jacL(const long* __restrict__ P, const double* __restrict__ x, const double* __restrict__ th, long tid, "
          "double* in, long* ik) {\n" << head;
    for (const auto &l : sp.load) os << "    " << l << "\n";
    os << "    (void)I;\n}\n";
    os << "static __device__ __forceinline__ void g" << gi << "_jacE(const long* __restrict__ P, const double* in, const long* ik, double* __restrict__ out, long tid, double* lds) {\n"
       << head << "    const int lane = threadIdx.x & 63;\n    (void)lane;\n";
    size_t at = 0;
    for (size_t j = 0; j < sts.size(); j++) {
        for (; at < eval_marks[j]; at++) os << "    " << sp.eval[at] << "\n";
        Body b(m, sts[j].pk, L);
        emit_coo_stores(os, b, sts[j].word, sts[j].S, sts[j].vals, use_tile(sts[j].S), "out", "_" + std::to_string(sts[j].pk));
    }
    for (; at < sp.eval.size(); at++) os << "    " << sp.eval[at] << "\n";
    os << "}\n";
}

// Tile loop of exa_consl / exa_jacl, SOFTWARE-PIPELINED (round 6): the workgroup walks `ppt` consecutive entries of the block map; the loads of
// the NEXT entry's tile are issued before the current tile is evaluated and stored (what exa_hessc does for hess_coord!), so a wavefront's loads
// have a whole evaluation to land in instead of stalling its first arithmetic.  Entries may belong to different dispatch units (patterns /
// fused groups): one hand-over array sized for the largest, a uniform branch per stage.  The last tile of a workgroup loads itself again (no
// extra HBM traffic, a fixed number of memory instructions per iteration).  Measured by hand on LV before it was built (tools/kernel_patch_ab.py
// cons_pipe): cons_nln! 1e7 0.0451 -> 0.0421 ms, 3e7 0.126 -> 0.121, 1e8 0.514 -> 0.448; bitwise equal.
void gen_dispatch_looped(std::ostringstream &os, const ParamLayout &L, int cb) {
    const bool jac = cb == CB_JAC;
    const size_t nunits = jac ? L.groups[cb].size() : L.active[cb].size();
    auto key = [&](size_t k) { return jac ? (int)k : L.active[cb][k]; };
    int nin = 0, nik = 0;       // (no placeholder registers: an unused long pair claimed by the asm statements below cost LV's exa_consl its 7th wavefront per SIMD)
    for (size_t k = 0; k < nunits; k++) { const auto ho = g_handover[{cb, key(k)}]; nin = std::max(nin, ho.first); nik = std::max(nik, ho.second); }
    const std::string bmap = "((const __attribute__((address_space(4))) long*)P[" + std::to_string(L.blk[cb]) + "])";
    const std::string fnL = jac ? "_jacL" : "_consL", pre = jac ? "g" : "p";
    auto loads = [&](const std::string &e, const std::string &sfx, const char *ind) {
        os << ind << "{ const int ps_ = (int)(" << e << " >> 40); const long t_ = (" << e << " & ((1L << 40) - 1)) * EXA_BLOCK + threadIdx.x;\n";
        // (ONE dispatch unit: no branch — the parameter-table reads of the stage functions are then unconditional and the compiler hoists them
        // out of the loop; under a uniform branch it reloads P[lo], P[hi], P[col] in every iteration, a dependent scalar round trip in front
        // of the loads: LV cons_nln! 0.0462 -> 0.0434 ms from that alone)
        for (size_t k = 0; k < nunits; k++)
            os << ind << "  " << (nunits == 1 ? "" : (k ? "else " : "") + ("if (ps_ == " + std::to_string(k) + ") ")) << pre << key(k) << fnL << "(P, x, th, t_, " << (nin ? "in" + sfx : std::string("nullptr")) << ", " << (nik ? "ik" + sfx : std::string("nullptr")) << ");\n";
        if (nunits == 1) os << ind << "  (void)ps_;\n";
        os << ind << "}\n";
    };
    os << "    const long b0_ = (long)blockIdx.x * ppt;\n    if (b0_ >= nent) return;\n    long en_ = " << bmap << "[b0_];\n"
       << "    double in_[" << std::max(1, nin) << "], inn_[" << std::max(1, nin) << "]; long ik_[" << std::max(1, nik) << "], ikn_[" << std::max(1, nik) << "];\n";
    // (several units: a tile of a unit with fewer hand-over values leaves the rest of the arrays as they were — defined once, here)
    if (nunits > 1) os << "#pragma unroll\n    for (int q = 0; q < " << std::max(1, nin) << "; q++) { in_[q] = 0.0; inn_[q] = 0.0; }\n"
                       << "#pragma unroll\n    for (int q = 0; q < " << std::max(1, nik) << "; q++) { ik_[q] = 0; ikn_[q] = 0; }\n";
    loads("en_", "_", "    ");
    os << "#pragma unroll\n    for (int q = 0; q < " << nin << "; q++) asm volatile(\"\" : \"+v\"(in_[q]));\n"
       << "#pragma unroll\n    for (int q = 0; q < " << nik << "; q++) asm volatile(\"\" : \"+v\"(ik_[q]));\n"
       << "#pragma unroll 1\n    for (int u_ = 0; u_ < ppt; u_++) {\n        const long b = b0_ + u_;\n        if (b >= nent) break;\n"
       << "        const long e_ = en_;\n        en_ = u_ + 1 < ppt && b + 1 < nent ? " << bmap << "[b + 1] : e_;\n";
    loads("en_", "n_", "        ");
    os << "        const int ps_ = (int)(e_ >> 40);\n        const long tid0 = (e_ & ((1L << 40) - 1)) * EXA_BLOCK + threadIdx.x;\n";
    if (nunits == 1) os << "        (void)ps_;\n";
    for (size_t k = 0; k < nunits; k++) {
        os << "        " << (nunits == 1 ? "" : (k ? "else " : "") + ("if (ps_ == " + std::to_string(k) + ") "));
        if (jac) os << "g" << key(k) << "_jacE(P, " << (nin ? "in_" : "nullptr") << ", " << (nik ? "ik_" : "nullptr") << ", out, tid0, lds);\n";
        else os << "{ const double v_ = p" << key(k) << "_consE(P, " << (nin ? "in_" : "nullptr") << ", " << (nik ? "ik_" : "nullptr") << ", tid0); p" << key(k) << "_conss(P, out, aug, tid0, v_); }\n";
    }
    // (the hand-over registers are claimed at the BOTTOM of the iteration: the wait for the next tile's loads then lands behind this tile's
    // arithmetic and stores — see gen_dispatch_chained)
    os << "#pragma unroll\n        for (int q = 0; q < " << nin << "; q++) { asm volatile(\"\" : \"+v\"(inn_[q])); in_[q] = inn_[q]; }\n"
       << "#pragma unroll\n        for (int q = 0; q < " << nik << "; q++) { asm volatile(\"\" : \"+v\"(ikn_[q])); ik_[q] = ikn_[q]; }\n    }\n";
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
          "long tid, double* lds, const long* __restrict__ augptr, const long* __restrict__ augsrc, const double* __restrict__ augcoef, "
          "double* __restrict__ gout) {\n";
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
        // (objective patterns: their first partials are grad! — added to gout when the caller wants all five callbacks from
        // this one sweep, exa_eval_all; gathered patterns are left to exa_grad_pull, which needs no atomics)
        const bool ingrad = isobj && p.o1step > 0 && std::find(L.pull.begin(), L.pull.end(), pk) == L.pull.end();
        if ((!isobj || ingrad) && p.o1step > 0) grpass(p, p.ad_root, a1, Emitter::litf(1.0));
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
                  "            for (; j < je; j++) v += __dmul_rn(augcoef[j], x[augsrc[j]]);\n        }\n        __builtin_nontemporal_store(v, &cout[r_]);\n    }\n";
        // (base rows: non-temporal — 8 B per row that nothing of this evaluation reads again must not push x / y out of the MALL)
        else if (p.kind == EXA_PAT_CONAUG) os << "    if (I0 < hi) augout[" << rowtxt << "] = " << E.sd(value) << ";\n";
        else os << "    if (I0 < hi) __builtin_nontemporal_store(" << E.sd(value) << ", &cout[" << rowtxt << "]);\n";
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
        if (ingrad) {
            // contributions of one point to the same variable first, then one add per distinct variable: a literal index
            // (the same variable for every point) by wavefront reduction, a data index with lanes of one target peeled,
            // anything else lane by lane; the whole wavefront is here (tail lanes are clamped, act = false)
            Scatter sc(b);
            for (int sl = 0; sl < p.o1step; sl++) sc.add(b, p.slotvar1[sl], a1.acc[sl]);
            sc.merge();
            for (; emitted < E.lines.size(); emitted++) os << "    " << E.lines[emitted] << "\n";
            os << "    if (gout) {\n        const bool act = I0 < hi;\n";
            // one-launch exa_eval_all (ParamLayout::gbits): the model build has PROVEN on the data that no variable is named twice by
            // these items — each is then the variable's whole gradient entry: a plain store, no zero-filled g, no atomics (the
            // untouched variables get their 0.0 from the zero tiles of the same launch)
            if (L.gbits >= 0) {
                os << "        if (P[" << L.gbits << "]) {\n";
                for (const Scatter::Item &it : sc.items)
                    os << "            if (act) gout[" << E.s(E.sub(it.vidx, Emitter::liti(1))) << "] = " << E.sd(it.val) << ";\n";
                os << "        } else {\n";
            }
            for (const Scatter::Item &it : sc.items) {
                const std::string idx = E.s(E.sub(it.vidx, Emitter::liti(1)));
                if (it.vidx.is_lit()) os << "        exa_wave_atomic_add(&gout[" << idx << "], act ? " << E.sd(it.val) << " : 0.0);\n";
                else if (!affine(*it.p, it.ir).ok) os << "        exa_scatter_add1(gout, " << idx << ", " << E.sd(it.val) << ", act);\n";
                else os << "        if (act) exa_atomic_add(&gout[" << idx << "], " << E.sd(it.val) << ");\n";
            }
            if (L.gbits >= 0) os << "        }\n";
            os << "    }\n";
        }
        if (isobj) ret = "(I0 < hi ? " + E.sd(value) + " : 0.0)";
    }
    os << "    return " << ret << ";\n}\n";
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

// ---- fused kernels: blockIdx -> (pattern, tile) ------------------------------------------------------
void gen_dispatch(std::ostringstream &os, const ParamLayout &L, int cb, const std::string &call_prefix, const std::string &call_args,
                  const std::string &tail_args, bool looped) {
    const auto &act = L.active[cb];
    const int ppt = L.ppt[cb];
    // Tile loop (exa_jacl, exa_consl; round 5): the workgroup walks `ppt` consecutive entries of the SAME block map, one tile each, in a loop
    // that is not unrolled — the registers of one tile, a workgroup that lives ppt times as long.  Why: a wavefront slot of the one-tile kernels
    // sits empty for ~1 500 - 2 000 cycles between two workgroups (average residency 6.1 of 8 wavefronts per SIMD in LV's exa_cons, profiles/
    // r5_instruction_mix.txt), and the compiler hoists the literal coefficients of the math routines out of the loop (once per workgroup).
    // (the entry of the NEXT tile is fetched — a scalar load through the constant address space: nothing on the device writes the map —
    // before the current tile is evaluated: no dependent round trip at the head of an iteration)
    const std::string bmap = "((const __attribute__((address_space(4))) long*)P[" + std::to_string(L.blk[cb]) + "])";
    if (looped) os << "    const long b0_ = (long)blockIdx.x * ppt;\n    if (b0_ >= nent) return;\n    long en_ = " << bmap << "[b0_];\n"
                      "#pragma unroll 1\n    for (int u_ = 0; u_ < ppt; u_++) {\n    const long b = b0_ + u_;\n    if (b >= nent) break;\n"
                      "    const long e_ = en_;\n    en_ = " << bmap << "[b + 1 < nent ? b + 1 : b];\n";
    else
    os << "    const long b = blockIdx.x;\n";
    // block map: which pattern and which tile this workgroup evaluates (interleaved by the runtime so that patterns
    // reading the same x ranges run on the same XCD at about the same time)
    if (!looped) os << "    const long e_ = ((const long*)P[" << L.blk[cb] << "])[b];\n";
    os << "    const int ps_ = (int)(e_ >> 40);\n"
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
    if (looped) os << "    }\n";
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
        const int first = grp.front();
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
        {
            // all loads of the iteration first — the other patterns of this tile AND the first pattern of the next tile —,
            // then all evaluations: every load has at least one evaluation's arithmetic to land in
            for (size_t j = 1; j < grp.size(); j++) { os << "            "; ld(grp[j], "tid", ""); os << "\n"; }
            os << "            "; ld(first, "(t + 1 < tend_ ? t + 1 : t) * EXA_BLOCK + threadIdx.x", "n"); os << "\n";
            for (size_t j = 0; j < grp.size(); j++) { os << "            "; ev(grp[j], "tid"); if (j + 1 < grp.size()) os << "\n"; }
        }
        // The hand-over registers are claimed HERE, at the bottom of the iteration, where the only memory instructions
        // issued after the loads are the fixed number of stores of the last pattern: the wait is vmcnt(#stores).  Left to
        // itself the compiler merges in*n into in* and waits at the loop header, where the first entry (no stores behind
        // its loads) forces vmcnt(0) — draining every store of the previous tile before the next evaluation starts.
        os << "\n#pragma unroll\n            for (int q = 0; q < " << std::max(1, h0.first) << "; q++) { asm volatile(\"\" : \"+v\"(in" << first << "n[q])); in" << first << "[q] = in" << first << "n[q]; }\n"
           << "#pragma unroll\n            for (int q = 0; q < " << std::max(1, h0.second) << "; q++) { asm volatile(\"\" : \"+v\"(ik" << first << "n[q])); ik" << first << "[q] = ik" << first << "n[q]; }\n        }\n    }\n";
    }
}

// exa_hesscl: the chained dispatch with the x operands staged through LDS (ParamLayout::stage).  Per wavefront, tile and STRETCH s of
// the group the run [B_s + 64 w + 256 t, + 64 + halo_s) of x — B_s = the smallest base of the stretch's member clusters — is loaded
// once (one 8-byte load per lane + a halo load by `halo_s` lanes; the next tile's, like the other loads, BEFORE the current tile is
// evaluated), written to the wavefront's LDS run of that stretch at the top of the iteration, and every pattern reads its operands
// from there at lane + (its cluster's base - B_s) + literal distance.
void gen_dispatch_chained_staged(std::ostringstream &os, const Model &m, const ParamLayout &L) {
    const int cb = CB_HESSC, T = L.chain[cb];
    const int RUN = 64 + kStageHalo;
    const auto &groups = L.groups[cb];
    auto ld = [&](int pk, const std::string &tid, const std::string &sfx) {
        os << "p" << pk << "_hessclL(P, x, y, th, " << tid << ", in" << pk << sfx << ", ik" << pk << sfx << ");";
    };
    auto ev = [&](int pk, const std::string &tid) {
        os << "p" << pk << "_hessclE(P, in" << pk << ", ik" << pk << ", xs, xd" << pk << "_, out, sink, sigma, " << tid << ", lds);";
    };
    os << "    const long e_ = ((const long*)P[" << L.blk[cb] << "])[blockIdx.x];\n    const int gs_ = (int)(e_ >> 40);\n"
       << "    const long t0_ = (e_ & ((1L << 40) - 1)) * " << T << ";\n    const int lane = threadIdx.x & 63;\n";
    for (size_t g = 0; g < groups.size(); g++) {
        const auto &grp = groups[g];
        const int first = grp.front();
        const int ns = L.gstretch[g];
        os << "    " << (g ? "else " : "") << "if (gs_ == " << g << ") {\n        const long tend_ = t0_ + " << T << " < P[" << L.gtiles[cb][g] << "] ? t0_ + " << T
           << " : P[" << L.gtiles[cb][g] << "];\n";
        // stretch geometry (scalars): base of every cluster, the stretches' bases, the clusters' distances from them, the halos, the
        // last variable any point of a stretch's members reads (loads beyond it are clamped onto it)
        for (int pk : grp)
            for (size_t c = 0; c < L.stage[pk].cl.size(); c++)
                os << "        const long B" << pk << "_" << c << "_ = P[" << L.stage[pk].word << "] + P[" << L.pat[pk].lo << "] + (" << L.stage[pk].cl[c].cmin << "L) - 1L;\n";
        for (int s = 0; s < ns; s++) {
            os << "        long S" << s << "_ = 0x7fffffffffffffffL; long xlast" << s << "_ = 0;\n";
            for (int pk : grp)
                for (size_t c = 0; c < L.stage[pk].cl.size(); c++) {
                    if (L.stage[pk].cl[c].stretch != s) continue;
                    os << "        S" << s << "_ = B" << pk << "_" << c << "_ < S" << s << "_ ? B" << pk << "_" << c << "_ : S" << s << "_;\n"
                       << "        { const long l_ = P[" << L.stage[pk].word << "] + P[" << L.pat[pk].hi << "] - 1L + (" << L.stage[pk].cl[c].cmax << "L) - 1L; xlast" << s
                       << "_ = l_ > xlast" << s << "_ ? l_ : xlast" << s << "_; }\n";
                }
            os << "        int halo" << s << "_ = 0;\n";
            for (int pk : grp)
                for (size_t c = 0; c < L.stage[pk].cl.size(); c++) {
                    if (L.stage[pk].cl[c].stretch != s) continue;
                    const int64_t span = L.stage[pk].cl[c].cmax - L.stage[pk].cl[c].cmin;
                    os << "        { const int h_ = (int)(B" << pk << "_" << c << "_ - S" << s << "_) + " << span << "; halo" << s << "_ = h_ > halo" << s << "_ ? h_ : halo" << s << "_; }\n";
                }
        }
        // per pattern: the lane's position inside each of its clusters' stretches
        for (int pk : grp) {
            const size_t nc = L.stage[pk].cl.size();
            os << "        int xd" << pk << "_[" << std::max<size_t>(nc, 1) << "] = {";
            for (size_t c = 0; c < nc; c++) os << (c ? ", " : "") << "lane + (int)(B" << pk << "_" << c << "_ - S" << L.stage[pk].cl[c].stretch << "_)";
            if (nc == 0) os << "0";
            os << "};\n";
        }
        for (int pk : grp) {
            const auto ho = g_handover[{CB_COUNT, pk}];
            os << "        double in" << pk << "[" << std::max(1, ho.first) << "]; long ik" << pk << "[" << std::max(1, ho.second) << "];\n";
        }
        const auto h0 = g_handover[{CB_COUNT, first}];
        os << "        double in" << first << "n[" << std::max(1, h0.first) << "]; long ik" << first << "n[" << std::max(1, h0.second) << "];\n";
        if (ns) os << "        double g0_[" << ns << "], g1_[" << ns << "];\n";
        auto G = [&](const std::string &t) {
            for (int s = 0; s < ns; s++)
                os << "        { const long a_ = S" << s << "_ + (" << t << ") * EXA_BLOCK + (threadIdx.x & ~63); long a0_ = a_ + lane; a0_ = a0_ < xlast" << s << "_ ? a0_ : xlast" << s
                   << "_; long a1_ = a_ + 64 + lane; a1_ = a1_ < xlast" << s << "_ ? a1_ : xlast" << s << "_; g0_[" << s << "] = x[a0_]; g1_[" << s << "] = x[lane < halo" << s
                   << "_ ? a1_ : a0_]; }\n";
        };
        auto claim = [&](const char *ind) {
            for (int s = 0; s < ns; s++) os << ind << "asm volatile(\"\" : \"+v\"(g0_[" << s << "])); asm volatile(\"\" : \"+v\"(g1_[" << s << "]));\n";
        };
        G("t0_");
        os << "        "; ld(first, "t0_ * EXA_BLOCK + threadIdx.x", "");
        os << "\n";
        claim("        ");
        os << "#pragma unroll\n        for (int q = 0; q < " << std::max(1, h0.first) << "; q++) asm volatile(\"\" : \"+v\"(in" << first << "[q]));\n"
           << "#pragma unroll\n        for (int q = 0; q < " << std::max(1, h0.second) << "; q++) asm volatile(\"\" : \"+v\"(ik" << first << "[q]));\n"
           << "#pragma unroll 1\n        for (long t = t0_; t < tend_; t++) {\n            const long tid = t * EXA_BLOCK + threadIdx.x;\n"
           << "            __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier();\n";
        for (int s = 0; s < ns; s++)
            os << "            xs[" << s * RUN << " + lane] = g0_[" << s << "];\n            if (lane < halo" << s << "_) xs[" << s * RUN + 64 << " + lane] = g1_[" << s << "];\n";
        os << "            __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");\n";
        for (size_t j = 1; j < grp.size(); j++) { os << "            "; ld(grp[j], "tid", ""); os << "\n"; }
        os << "    ";
        G("t + 1 < tend_ ? t + 1 : t");
        os << "            "; ld(first, "(t + 1 < tend_ ? t + 1 : t) * EXA_BLOCK + threadIdx.x", "n"); os << "\n";
        for (size_t j = 0; j < grp.size(); j++) { os << "            "; ev(grp[j], "tid"); os << "\n"; }
        claim("            ");
        os << "#pragma unroll\n            for (int q = 0; q < " << std::max(1, h0.first) << "; q++) { asm volatile(\"\" : \"+v\"(in" << first << "n[q])); in" << first << "[q] = in" << first << "n[q]; }\n"
           << "#pragma unroll\n            for (int q = 0; q < " << std::max(1, h0.second) << "; q++) { asm volatile(\"\" : \"+v\"(ik" << first << "n[q])); ik" << first << "[q] = ik" << first << "n[q]; }\n        }\n    }\n";
    }
    (void)m;
}

}  // namespace gen
}  // namespace exa
