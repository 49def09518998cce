// exa_gen_module.cpp — generate_module: the parameter-table layout, the fused groups, and the __global__ kernels of a
// model's first module (one fused launch per callback: exa_obj, exa_grad / exa_grad_pull / exa_gradv, exa_cons / exa_cons1,
// exa_jac, exa_hess / exa_hessc, exa_fused, exa_jprod / exa_jprod1, exa_jtprod, exa_hprod, exa_{j,h}struct{32,64}).
// Replaces the per-pattern kernel launches of ext/ExaModelsKernelAbstractions.jl:253-351, 369-547.
#include "exa_gen.hpp"

namespace exa {

using namespace gen;

Generated generate_module(const Model &m, bool loopfree_scatter, bool nostage) {
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
        pp.lo = w++; pp.hi = w++; pp.o0 = w++; pp.o1 = w++; pp.o2 = w++; pp.oa = w++; pp.ob = w++; pp.qlo = w++; pp.qhi = w++;
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
    for (int cb = 0; cb < CB_COUNT; cb++) { L.blk[cb] = w++; L.ppt[cb] = 1; L.chain[cb] = cb == CB_HESSC ? kChainTiles : 0; }
    // one-launch exa_eval_all (ParamLayout::gbits): only models whose objective gradient is scattered inside the sweep and nowhere else
    if (!L.active[CB_GRAD].empty() && L.pull.empty()) L.gbits = w++;
    g_handover.clear();
    // EXAHIP_GROUP=0: every pattern on its own everywhere (the ungrouped kernels the grouped ones must equal bit for bit)
    const bool grouping = env_int("EXAHIP_GROUP", 1) != 0;
    constexpr int kGroupMax = 8;
    // groups of co-indexed patterns (iterator lengths within 2 of each other), in dispatch order.  The lengths are what
    // decides, so instances of a model family share one module unless two unrelated blocks happen to be equally long.
    for (int cb : {CB_HESSC}) {
        if (L.chain[cb] == 0) continue;
        for (int k : L.active[cb]) {
            bool placed = false;
            if (grouping)
                for (auto &g : L.groups[cb])
                    if ((int)g.size() < kGroupMax && std::llabs(m.pats[g.front()].n - m.pats[k].n) <= 2) { g.push_back(k); placed = true; break; }
            if (!placed) L.groups[cb].push_back({k});
        }
        for (size_t g = 0; g < L.groups[cb].size(); g++) L.gtiles[cb].push_back(w++);
    }
    // fused groups of the scattering products and of the one-launch cons_nln!: patterns of EXACTLY the same length (one
    // thread evaluates point I of all)
    for (int cb : {CB_JTPROD, CB_HPROD, CB_CONS1, CB_JAC, CB_HESS, CB_FUSED}) {
        for (int k : L.active[cb]) {
            bool placed = false;
            // (fused sweep: an objective pattern stays alone — its workgroups also write the partial sums of obj)
            const bool alone = cb == CB_FUSED && m.pats[k].kind == EXA_PAT_OBJ;
            // (a group's body is the concatenation of its patterns' bodies: bounded by the slots it computes, so that a
            // model with many equally long wide patterns does not produce one register-starved monster)
            auto slots = [&](int q) { const Pattern &t = m.pats[q]; return cb == CB_JAC || cb == CB_JTPROD || cb == CB_CONS1 ? t.o1step : t.o1step + t.o2step; };
            const int cap = 128;
            // (the scattering products also by the RAW contributions their reverse sweeps walk — what the body's length
            // follows: fusing is for small bodies that share loads (ACOPF's branch rows: 48 first-order / 90 second-order
            // contributions per group; the rocket: 25 / 103); bodies of thousands of SSA values gain nothing from it and,
            // fused, were miscompiled by the hiprtc of ROCm 7.0 once wavefront operations sat in them — wrong entries of
            // J'v / Hv in 4 of 60 random depth-6 models, none with the patterns on their own)
            auto raw = [&](int q) { const Pattern &t = m.pats[q]; return (int)(cb == CB_JTPROD ? t.comp1.size() : cb == CB_HPROD ? t.comp2.size() : 0); };
            const int rawcap = cb == CB_JTPROD ? 64 : 160;
            if (!alone && grouping)
                for (auto &g : L.groups[cb]) {
                    int have = 0, have_raw = 0;
                    for (int q : g) { have += slots(q); have_raw += raw(q); }
                    // (Hessian callbacks: objective patterns only among themselves — the objective-only forms, y == NULL, launch
                    // the objective groups alone and leave exact zeros in the constraint slots, nlp.jl:1912-1914)
                    const bool obj_mix = (cb == CB_HESS || cb == CB_HPROD) && (m.pats[g.front()].kind == EXA_PAT_OBJ) != (m.pats[k].kind == EXA_PAT_OBJ);
                    if ((int)g.size() < kGroupMax && have + slots(k) <= cap && have_raw + raw(k) <= rawcap && m.pats[g.front()].n == m.pats[k].n && !obj_mix &&
                        !(cb == CB_FUSED && m.pats[g.front()].kind == EXA_PAT_OBJ)) {
                        g.push_back(k); placed = true; break;
                    }
                }
            if (!placed) L.groups[cb].push_back({k});
        }
    }

    // exa_hesscl (ParamLayout::stage): only when EVERY pattern of EVERY chained group qualifies
    L.stage.assign(np, ParamLayout::Stage());
    L.gstretch.assign(L.groups[CB_HESSC].size(), 0);
    L.max_stretch = 0;
    L.staged = !nostage && L.chain[CB_HESSC] > 0 && !L.groups[CB_HESSC].empty();
    if (L.staged)
        for (const auto &grp : L.groups[CB_HESSC])
            for (int k : grp) L.staged = L.staged && pattern_stage(m, k, L, &L.stage[k]);
    if (L.staged) {
        // the clusters of a group's patterns -> the group's stretches: in ascending order of their literals, merged while the union stays
        // within the halo (patterns whose ranges start a few points apart — LV's constraint and objective — share a stretch; whether the
        // actual bases do lie that close is checked per shard by the runtime)
        for (size_t g = 0; g < L.groups[CB_HESSC].size(); g++) {
            struct Ref { int k, c; };
            std::vector<Ref> refs;
            for (int k : L.groups[CB_HESSC][g]) for (size_t c = 0; c < L.stage[k].cl.size(); c++) refs.push_back({k, (int)c});
            std::sort(refs.begin(), refs.end(), [&](const Ref &a, const Ref &b) { return L.stage[a.k].cl[a.c].cmin < L.stage[b.k].cl[b.c].cmin; });
            int ns = 0;
            int64_t smin = 0;
            for (const Ref &r : refs) {
                ParamLayout::Stage::Cluster &cl = L.stage[r.k].cl[r.c];
                if (ns == 0 || cl.cmax - smin > kStageHalo) { ns++; smin = cl.cmin; }
                cl.stretch = ns - 1;
            }
            L.gstretch[g] = ns;
            L.max_stretch = std::max(L.max_stretch, ns);
        }
        // ONE stretch only: staging is a single-array-stencil feature.  Several stretches (one per variable array: cops_chain's u, x1, x2, x3, the
        // rocket's h, v, m, tau) were built in round 4 and lost every A/B — cops_chain 0.089 staged / 0.073 chained, rocket nh = 2e7 2.40 / 2.11
        // (profiles/r4_staging_ab.txt, r5_rocket_staging_ab.txt): the extra LDS round trips cost more than the overlapping loads they replace.
        L.staged = L.max_stretch == 1;
    }
    if (!L.staged) { L.stage.assign(np, ParamLayout::Stage()); L.gstretch.assign(L.groups[CB_HESSC].size(), 0); L.max_stretch = 0; }
    // streaming value kernels keep more loads in flight per wavefront with several points per thread (measured)
    // (measured, LV 1e7: obj 0.037 -> 0.020 ms with 8 points per thread; 2 - 16 points per thread moved cons / jac / hess by
    // +-2 %, profiles/NOTES.md)
    L.ppt[CB_OBJ] = 8;
    L.nwords = w;

    for (int &v : g_lds_need) v = 0;
    g_lit_idx.clear();
    g_scatter_lines.clear();
    std::ostringstream os;
    L.pull_ppt = 2;
    os << prelude_text(m, L);
    os << "// patterns=" << np << " (sizes, offsets and column pointers are run-time parameters in P[])\n";
    if (loopfree_scatter) os << "// scatter kernels without loops: the first build of this module spilled registers there\n";
    if (nostage) os << "// no LDS-staged chained kernel (exa_hesscl): it outgrew the architectural registers in the first build of this module\n";
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
        for (int cb : {CB_GRAD, CB_JTPROD, CB_HPROD}) g_loopfree[cb] = loopfree_scatter || (int)g_scatter_lines[cb] > kHugeBody;
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
            if (L.ppt[CB_CONS] == 1) gen_cons_two_stage(os, m, k, L);
            gen_jprod_fn(os, m, k, L);
            if (p.o1step > 0) gen_struct_fn(os, m, k, L, false);
        }
        if (p.o2step > 0) { gen_hess_fn(os, m, k, L); gen_struct_fn(os, m, k, L, true); }
    }
    for (int cb : {CB_JTPROD, CB_HPROD})
        for (size_t g = 0; g < L.groups[cb].size(); g++) gen_scatter_group_fn(os, m, L, cb, (int)g);
    for (int cb : {CB_JAC, CB_HESS})
        for (size_t g = 0; g < L.groups[cb].size(); g++) gen_coo_group_fn(os, m, L, cb, (int)g);
    if (L.ppt[CB_JAC] == 1)
        for (size_t g = 0; g < L.groups[CB_JAC].size(); g++) gen_jac_group_two_stage(os, m, L, (int)g);
    for (size_t g = 0; g < L.groups[CB_FUSED].size(); g++) gen_fused_group_fn(os, m, L, (int)g);
    // scatter kernels whose patterns have targets shared by ALL data points process 16 tiles per workgroup: the shared
    // target then receives one atomic per wavefront per 16 tiles (same-address atomics serialise chip-wide at ~10 ns:
    // the rocket's step variable took 47 000 of them per J'v, 0.47 ms)
    for (int cb : {CB_GRAD, CB_JTPROD, CB_HPROD}) {
        bool any = false;
        for (const auto &kv : g_lit_idx) any = any || (kv.first.first == cb && !kv.second.empty());
        if (any && !g_loopfree[cb]) L.ppt[cb] = 16;
    }
    // obj: per-workgroup partial sums
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_obj(const long* __restrict__ P, const double* __restrict__ x, "
          "const double* __restrict__ th, double* part, unsigned* done, double* __restrict__ out) {\n    const long b = blockIdx.x;\n    double v = 0.0;\n";
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
    // the workgroup that finishes LAST folds the partial sums itself, in index order (deterministic), and re-arms the counter(s):
    // one launch at any size (exa_obj_arrive).  Release / acquire at device scope around the counters; device-scope loads of the partials.
    os << "    const double s = exa_block_sum(v);\n    exa_obj_arrive(part, b, s, done, gridDim.x, out);\n}\n";
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
    if (L.gbits >= 0)
        // the zero tiles of the one-launch grad! (one more unit of the block map; the buffer holds nvar in front of the bitmap)
        os << "    else if (ps_ == " << L.active[CB_GRAD].size() << ") {\n        const long* gb_ = (const long*)P[" << L.gbits << "];\n"
              "        const unsigned long long* bits = (const unsigned long long*)(gb_ + 1);\n        const long tile_ = e_ & ((1L << 40) - 1);\n#pragma unroll\n"
              "        for (int u = 0; u < 8; u++) {\n            const long v = tile_ * (EXA_BLOCK * 8) + u * EXA_BLOCK + threadIdx.x;\n"
              "            if (v < gb_[0] && !((bits[v >> 6] >> (v & 63)) & 1ull)) __builtin_nontemporal_store(0.0, &out[v]);\n        }\n    }\n";
    os << "}\n";
    // grad!, gather part: one thread per variable of [v_begin, v_end); also provides the zero of untouched variables (no
    // memset).  The value is COMPLETE — every data point of every gathered pattern that touches the variable, whatever the
    // shard — for the variables [own_lo, own_hi) this rank owns, and zero elsewhere: owner computes.  A sharded model whose
    // objective patterns are all gathered launches it over its own variables only and needs no collective at all.
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_grad_pull(const long* __restrict__ P, const double* __restrict__ x, "
          "const double* __restrict__ th, double* __restrict__ out, long v_begin, long v_end, long own_lo, long own_hi) {\n"
          "    const long v0 = v_begin + (long)blockIdx.x * (EXA_BLOCK * EXA_PULL_PPT) + threadIdx.x;\n    double g[EXA_PULL_PPT];\n"
          "#pragma unroll\n    for (int u = 0; u < EXA_PULL_PPT; u++) {\n        const long v_ = v0 + u * EXA_BLOCK, v = v_ < v_end ? v_ : v_end - 1;\n        g[u] = 0.0;\n";
    for (int k : L.pull) os << "        g[u] += p" << k << "_pull(P, x, th, v + 1);\n";
    os << "    }\n#pragma unroll\n    for (int u = 0; u < EXA_PULL_PPT; u++) { const long v = v0 + u * EXA_BLOCK; if (v < v_end) __builtin_nontemporal_store(v >= own_lo && v < own_hi ? g[u] : 0.0, &out[v]); }\n}\n";
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
    // exa_consl: exa_cons as a tile loop over `ppt` consecutive block-map entries (see gen_dispatch, looped)
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_consl(const long* __restrict__ P, const double* __restrict__ x, "
          "const double* __restrict__ th, double* __restrict__ out, double* __restrict__ aug, long nent, int ppt) {\n";
    if (L.ppt[CB_CONS] == 1) gen_dispatch_looped(os, L, CB_CONS);
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
                if (!target) { os << "        __builtin_nontemporal_store(" << E.sd(vals[q]) << ", &out[P[" << pp.o0 << "] + I]);\n"; continue; }
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
                os << "        __builtin_nontemporal_store(v, &out[r_]);\n        }\n";
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
                if (!target) { os << "        __builtin_nontemporal_store(" << E.sd(sums[q]) << ", &out[P[" << L.pat[pk].o0 << "] + I]);\n"; continue; }
                os << "        {\n        double s_ = " << E.sd(sums[q]) << ";\n        const long r_ = P[" << L.pat[pk].o0 << "] + I;\n"
                      "        long j = augptr[r_];\n        const long je = augptr[r_ + 1];\n"
                      "        for (; j + 4 <= je; j += 4) {\n"
                      "            const long i0 = augsrc[j], i1 = augsrc[j + 1], i2 = augsrc[j + 2], i3 = augsrc[j + 3];\n"
                      "            const double c0 = augcoef[j], c1 = augcoef[j + 1], c2 = augcoef[j + 2], c3 = augcoef[j + 3];\n"
                      "            const double v0 = v[i0], v1 = v[i1], v2 = v[i2], v3 = v[i3];\n"
                      "            s_ += __dmul_rn(c0, v0); s_ += __dmul_rn(c1, v1); s_ += __dmul_rn(c2, v2); s_ += __dmul_rn(c3, v3);\n        }\n"
                      "        for (; j < je; j++) s_ += __dmul_rn(augcoef[j], v[augsrc[j]]);\n        __builtin_nontemporal_store(s_, &out[r_]);\n        }\n";
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
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_jac(const long* __restrict__ P, const double* __restrict__ x, "
          "const double* __restrict__ th, double* __restrict__ out) {\n";
    lds_decl(CB_JAC, false);
    gen_dispatch(os, L, CB_JAC, "jac", "P, x, th, out", ", lds");
    os << "}\n";
    // the same callback as a tile loop (gen_dispatch, looped): launched instead of exa_jac where the block map is long (exa_runtime.cpp do_jac)
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_jacl(const long* __restrict__ P, const double* __restrict__ x, "
          "const double* __restrict__ th, double* __restrict__ out, long nent, int ppt) {\n";
    lds_decl(CB_JAC, false);
    if (L.ppt[CB_JAC] == 1) gen_dispatch_looped(os, L, CB_JAC);
    os << "}\n";
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_hess(const long* __restrict__ P, const double* __restrict__ x, "
          "const double* __restrict__ y, const double* __restrict__ th, double* __restrict__ out, double sigma) {\n";
    lds_decl(CB_HESS, true);
    gen_dispatch(os, L, CB_HESS, "hess", "P, x, y, th, out, sigma", ", lds");
    os << "}\n";
    if (L.chain[CB_HESSC] > 0) {
        os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_hessc(const long* __restrict__ P, const double* __restrict__ x, "
              "const double* __restrict__ y, const double* __restrict__ th, double* __restrict__ out, double sigma, double* __restrict__ sink) {\n";
        lds_decl(CB_HESS, true);
        gen_dispatch_chained(os, L, CB_HESSC, "hessc", true);
        os << "}\n";
    }
    if (L.chain[CB_HESSC] > 0 && L.staged) {
        os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_hesscl(const long* __restrict__ P, const double* __restrict__ x, "
              "const double* __restrict__ y, const double* __restrict__ th, double* __restrict__ out, double sigma, double* __restrict__ sink) {\n";
        lds_decl(CB_HESS, true);
        os << "    __shared__ double xs_all[(EXA_BLOCK / 64) * " << L.max_stretch * (64 + kStageHalo) << "];\n    double* xs = xs_all + (threadIdx.x >> 6) * "
           << L.max_stretch * (64 + kStageHalo) << ";\n";
        gen_dispatch_chained_staged(os, m, L);
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
        // bmap: null = the block map of exa_eval_fused (P[blk]); exa_eval_all passes a map that holds ONE MORE unit after the
        // fused groups — the tiles of the gathered gradient (the body of exa_grad_pull: one thread per variable of
        // [v_begin, v_end), complete for [own_lo, own_hi), zero elsewhere) — so that grad! rides in the same launch; in the
        // interleaved block order a gradient tile runs right after the tiles that have just pulled its stretch of x into L2
        os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_fused(const long* __restrict__ P, const double* __restrict__ x, "
              "const double* __restrict__ y, const double* __restrict__ th, double* __restrict__ part, double* __restrict__ cout, "
              "double* __restrict__ augout, double* __restrict__ jout, double* __restrict__ hout, double sigma, "
              "const long* __restrict__ augptr, const long* __restrict__ augsrc, const double* __restrict__ augcoef, double* __restrict__ gout, "
              "const long* __restrict__ bmap, long v_begin, long v_end, long own_lo, long own_hi, unsigned* done, long nobj, double* __restrict__ obj_out) {\n";
        if (mx) os << "    __shared__ double lds_all[(EXA_BLOCK / 64) * " << mx << "];\n    double* lds = lds_all + (threadIdx.x >> 6) * " << mx << ";\n";
        else os << "    double* lds = nullptr;\n";
        // only the workgroups of OBJECTIVE patterns have something to add to obj: they write one partial sum each, at a
        // compact index (pattern's first slot + tile), so the reduction reads 1/3 of the workgroup count on LV
        os << "    const long b = blockIdx.x;\n"
           << "    const long e_ = (bmap ? bmap : (const long*)P[" << L.blk[CB_FUSED] << "])[b];\n    const int ps_ = (int)(e_ >> 40);\n"
              "    const long tile_ = e_ & ((1L << 40) - 1);\n    const long tid0 = tile_ * EXA_BLOCK + threadIdx.x;\n";
        const auto &grps = L.groups[CB_FUSED];
        for (size_t k = 0; k < grps.size(); k++) {
            os << "    " << (k ? "else " : "") << "if (ps_ == " << k << ") { const double v = g" << k
               << "_fused(P, x, y, th, cout, augout, jout, hout, sigma, tid0, lds, augptr, augsrc, augcoef, gout);";
            if (m.pats[grps[k].front()].kind == EXA_PAT_OBJ)
                os << " const double s = exa_block_sum(v); exa_obj_arrive(part, P[" << L.pat[grps[k].front()].ob << "] + tile_, s, done, nobj, obj_out);";
            else os << " (void)v;";
            os << " }\n";
        }
        if (L.gbits >= 0) {
            // the zero tiles of the one-launch exa_eval_all: 0.0 for every variable no objective point writes (bitmap built at model build)
            os << "    " << (grps.empty() ? "" : "else ") << "if (ps_ == " << grps.size() << ") {\n"
                  "        const unsigned long long* bits = (const unsigned long long*)((const long*)P[" << L.gbits << "] + 1);\n"
                  "#pragma unroll\n        for (int u = 0; u < 8; u++) {\n            const long v = tile_ * (EXA_BLOCK * 8) + u * EXA_BLOCK + threadIdx.x;\n"
                  "            if (v < v_end && !((bits[v >> 6] >> (v & 63)) & 1ull)) __builtin_nontemporal_store(0.0, &gout[v]);\n        }\n    }\n";
        }
        if (!L.pull.empty()) {
            os << "    " << (grps.empty() ? "" : "else ") << "if (ps_ == " << grps.size() << ") {\n"
                  "        const long v0 = v_begin + tile_ * (EXA_BLOCK * EXA_PULL_PPT) + threadIdx.x;\n        double g[EXA_PULL_PPT];\n"
                  "#pragma unroll\n        for (int u = 0; u < EXA_PULL_PPT; u++) {\n            const long v_ = v0 + u * EXA_BLOCK, v = v_ < v_end ? v_ : v_end - 1;\n            g[u] = 0.0;\n";
            for (int k : L.pull) os << "            g[u] += p" << k << "_pull(P, x, th, v + 1);\n";
            os << "        }\n#pragma unroll\n        for (int u = 0; u < EXA_PULL_PPT; u++) { const long v = v0 + u * EXA_BLOCK; if (v < v_end) __builtin_nontemporal_store(v >= own_lo && v < own_hi ? g[u] : 0.0, &gout[v]); }\n    }\n";
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

}  // namespace exa
