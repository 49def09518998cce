// exa_windows.cpp — owner-computes WINDOWS: planning (window_plan), the second / third modules of a model (compressed COO: exa_compress;
// products: plan_products / load_products), their launches, and the C ABI of the compressed COO (include/exahip.h).  Split off
// exa_runtime.cpp in round 4; the shared state is Handle (exa_rt.hpp).
#include "exa_rt.hpp"

using namespace exa;
using namespace exa::rt;

namespace exa {
namespace rt {


// ---- windowed compressed evaluation (kernels: exa_gen_window.cpp generate_window_module) -----------------
// Decides, per matrix, whether the sorted structure is regular enough for the fast path, and prepares its tables:
//   * every slot s of every active pattern sits at compressed entry a_s + b_s*I for all points but a few at the ends
//     (fit at the middle point, checked for every point on the device); at most kBlock such end points in total — they are
//     evaluated by the tail kernel exa_c*x;
//   * the slots of a pattern are split into PASSES: one per stride b_s and per cluster of targets within 48 points (the
//     x[i] and u[i] blocks of a discretised ODE lie millions of entries apart); slots with b_s = 0 (an entry every point
//     adds to) go to the shared-entry kernel exa_c*s instead; at most 24 passes, at most 6 evaluations per point;
//   * window size and kernel shape (one chunk per pass / chunk loops) from the strides, see below.
// Anything else (data-indexed targets, stepped ranges of different lengths meeting in the same columns) keeps the gather.
// Knobs: EXAHIP_CWINDOW=0 gather only (the reference's scheme, bit for bit); EXAHIP_VERBOSE=1 prints the pass table;
// EXAHIP_KEEP_SOURCE=1 keeps the generated source next to the cached code object.
// Products (wk = WK_JTPROD / WK_HPROD): the same plan over the dense output vector — entry = 0-based variable, the maps
// a_s + b_s*I come from the index expressions (product_items), nothing is fitted or checked on the device, and the data
// points are ALL points of every pattern whatever the shard (a rank of a sharded model owns a range of WINDOWS and
// evaluates whatever touches them: owner computes).  Host-only: also planned for exa_plan_only handles.
Handle::Window &window_of(Handle &h, int wk) { return wk == WK_CJAC ? h.wj : wk == WK_CHESS ? h.wh : h.wp[wk - WK_JTPROD]; }
bool window_plan(Handle &h, int wk, const int32_t *cmap, WindowMatrix &wm) {
    const Model &m = *h.m;
    const ParamLayout &L = h.gen.layout;
    const bool hess = wk == WK_CHESS, product = wk >= WK_JTPROD;
    std::vector<WindowPat> &pats = wm.pats;
    std::vector<WindowShared> &shared = wm.shared;
    bool &single = wm.single;
    int &nspaces = wm.nspaces, &zs = wm.zs;
    Handle::Window &w = window_of(h, wk);
    const int64_t ncomp = product ? m.nvar : (hess ? h.ch.cnnz : h.cj.cnnz);
    const auto &act = L.active[wk == WK_CJAC ? CB_JAC : wk == WK_CHESS ? CB_HESS : wk == WK_JTPROD ? CB_JTPROD : CB_HPROD];
    auto no = [&](const std::string &why) { w.why = why; return false; };
    if (act.empty() || ncomp == 0) return no("empty");
    if (product && ncomp > 0x7fffffffLL) return no("more than 2^31 variables");
    std::vector<int64_t> Q;
    int64_t bmax = 0, spread_max = 0, npts = 0, passes_pts = 0;
    int smax = 1;
    std::map<int, std::pair<std::vector<int64_t>, std::vector<int64_t>>> items;     // products: per pattern the static (a, b)
    for (int k : act) {
        if (product) {
            auto &ab = items[k];
            if (!product_items(m, L, wk, k, ab.first, ab.second)) return no("pattern " + std::to_string(k) + ": a target is reached through a data column");
            smax = std::max(smax, (int)ab.first.size());
        } else smax = std::max(smax, hess ? m.pats[k].o2step : m.pats[k].o1step);
    }
    struct Exc { int k; int64_t I; };
    std::vector<Exc> exc;
    struct Sh { int k; int64_t e_lo, e_hi; std::vector<int64_t> target; };
    std::vector<Sh> shs;
    for (size_t j = 0; j < act.size(); j++) {
        const int k = act[j];
        const Pattern &p = m.pats[k];
        const int S = product ? (int)items[k].first.size() : (hess ? p.o2step : p.o1step);
        if (S == 0) continue;
        // This process's data points of the pattern are [lo, hi) (all of them unless sharded) and slot s of point I sits at
        // o + S * I of the COO it writes — also for the packed local slice of a shard, whose offset word already holds
        // local_offset - S * lo (fill_params).  Everything below is in ABSOLUTE point indices, which is what the window
        // kernels evaluate.  (Products: every point of the pattern, see above.)
        const auto &pl = L.pat[k];
        const int64_t lo = product ? 0 : h.P[pl.lo], hi = product ? p.n : h.P[pl.hi], n = hi - lo, o = product ? 0 : h.P[hess ? pl.o2 : pl.o1];
        if (n <= 0) continue;
        const int64_t mid = lo + (n >= 2 ? std::min(n / 2, n - 2) : 0);
        std::vector<int64_t> a((size_t)S), bs((size_t)S), aloc((size_t)S);
        int64_t cnt = 0, e_lo = 0, e_hi = n;
        if (product) { a = items[k].first; bs = items[k].second; }
        else {
            std::vector<int32_t> two((size_t)2 * S);
            HIPCHK(hipMemcpy(two.data(), cmap + o + (int64_t)S * mid, 4 * (size_t)S * (n >= 2 ? 2 : 1), hipMemcpyDeviceToHost));
            for (int s = 0; s < S; s++) {
                bs[s] = n >= 2 ? (int64_t)two[S + s] - two[s] : 1;
                a[s] = (int64_t)two[s] - bs[s] * mid;
                aloc[s] = a[s] + bs[s] * lo;            // the same map in the local index I - lo (what the check kernel walks)
            }
            affine_exceptions(cmap, o + (int64_t)S * lo, S, n, aloc.data(), bs.data(), mid - lo, &cnt, &e_lo, &e_hi, h.stream);
        }
        if (n <= 8) { e_lo = n; e_hi = n; }       // a handful of points (boundary conditions): all of them go to the tail kernel
        if (e_lo + (n - e_hi) > kBlock) return no("pattern " + std::to_string(k) + ": " + std::to_string(cnt) + " points off the regular structure");
        e_lo += lo; e_hi += lo;
        for (int64_t I = lo; I < e_lo; I++) exc.push_back({k, I});
        for (int64_t I = e_hi; I < hi; I++) exc.push_back({k, I});
        if (e_hi <= e_lo) continue;     // every point of the pattern is irregular (tiny pattern): exa_c*x does it all
        npts += e_hi - e_lo;
        // stride classes
        std::vector<int64_t> strides;
        for (int s = 0; s < S; s++) if (std::find(strides.begin(), strides.end(), bs[s]) == strides.end()) strides.push_back(bs[s]);
        if (strides.size() > 8) return no("pattern " + std::to_string(k) + ": slots advance with " + std::to_string(strides.size()) + " different strides");
        for (int64_t b : strides) {
            if (b == 0) {
                // entries every point adds to: per-workgroup sums + fold
                WindowShared q;
                Sh sh{k, e_lo, e_hi, {}};
                q.k = k;
                for (int s = 0; s < S; s++) {
                    if (bs[s] != 0) continue;
                    size_t g = 0;
                    for (; g < sh.target.size(); g++) if (sh.target[g] == a[s]) break;
                    if (g == sh.target.size()) { sh.target.push_back(a[s]); q.groups.emplace_back(); }
                    q.groups[g].push_back(s);
                }
                shared.push_back(std::move(q));
                shs.push_back(std::move(sh));
                passes_pts += e_hi - e_lo;
                continue;
            }
            // distinct targets of this stride, ascending; targets more than 64 points apart (another block of
            // variables: x[i] and u[i] of a discretised ODE) form separate passes, each re-evaluating the points for
            // its own slots only (the compiler drops what those slots do not need)
            std::vector<int64_t> av;
            for (int s = 0; s < S; s++) if (bs[s] == b && std::find(av.begin(), av.end(), a[s]) == av.end()) av.push_back(a[s]);
            std::sort(av.begin(), av.end());
            const int64_t ab = b < 0 ? -b : b;
            for (size_t c0 = 0; c0 < av.size();) {
                size_t c1 = c0 + 1;
                while (c1 < av.size() && (av[c1] - av[c0]) / ab <= 48) c1++;
                WindowPat wp;
                wp.k = k;
                wp.qbase = (int)Q.size();
                wp.group.assign(S, -1);
                std::vector<int64_t> ga;      // groups in slot order (the order the values are added in)
                for (int s = 0; s < S; s++) {
                    if (bs[s] != b || a[s] < av[c0] || a[s] > av[c1 - 1]) continue;
                    int g = -1;
                    for (size_t q = 0; q < ga.size(); q++) if (ga[q] == a[s]) g = (int)q;
                    if (g < 0) { g = (int)ga.size(); ga.push_back(a[s]); }
                    wp.group[s] = g;
                }
                wp.phase.assign(ga.size(), 0);
                for (size_t g = 0; g < ga.size(); g++) {
                    int ph = 0;
                    for (bool again = true; again;) {
                        again = false;
                        for (size_t q = 0; q < g; q++)
                            if (wp.phase[q] == ph && (ga[g] - ga[q]) % ab == 0) { ph++; again = true; break; }
                    }
                    wp.phase[g] = ph;
                }
                const int64_t amin = av[c0], amax = av[c1 - 1];
                spread_max = std::max(spread_max, (amax - amin) / ab + 1);
                bmax = std::max(bmax, ab);
                passes_pts += e_hi - e_lo;
                Q.push_back(b); Q.push_back(e_lo); Q.push_back(e_hi); Q.push_back(amin); Q.push_back(amax);
                for (int64_t v : ga) Q.push_back(v);
                pats.push_back(std::move(wp));
                c0 = c1;
            }
        }
    }
    if ((int64_t)exc.size() > kBlock) return no(std::to_string(exc.size()) + " irregular end points");
    if (pats.empty()) return no("no regular pattern");
    if (pats.size() > 24 || (double)passes_pts > 6.0 * (double)npts)
        return no(std::to_string(pats.size()) + " passes over " + std::to_string((double)passes_pts / std::max<double>(1.0, (double)npts)) + "x the points");
    // ---- block-owned variant (WindowSpec): the passes fall into several far-apart output ranges (SPACES: the column
    // blocks of a model laid out as separate variable arrays).  Workgroup j owns window j of every space — n points'
    // worth of each — so a pattern is evaluated once per point, not once per pass (rocket chess: 1.84x the VALU
    // instructions of the uncompressed sweep with one window space).  Needs: positive strides, one stride per space,
    // every pattern's points of a block within one chunk.
    nspaces = 0; zs = 0;
    std::vector<int32_t> Rb;
    int64_t Wtot = 0, nblocks = 0;
    {
        bool ok = pats.size() >= 2;
        struct Sp { int64_t lo, hi, b, W = 0, off = 0, o = 0, end = 0; };
        std::vector<Sp> sp;
        std::vector<size_t> order(pats.size());
        auto out_lo = [&](size_t q) { const int64_t *t = &Q[pats[q].qbase]; return t[3] + t[0] * t[1]; };
        auto out_hi = [&](size_t q) { const int64_t *t = &Q[pats[q].qbase]; return t[4] + t[0] * (t[2] - 1) + 1; };
        for (size_t q = 0; q < pats.size() && ok; q++) { order[q] = q; if (Q[pats[q].qbase] <= 0) ok = false; }
        if (ok) {
            std::sort(order.begin(), order.end(), [&](size_t a, size_t c) { return out_lo(a) < out_lo(c); });
            for (size_t q : order) {
                const int64_t b = Q[pats[q].qbase];
                if (!sp.empty() && out_lo(q) < sp.back().hi) {
                    if (sp.back().b != b) { ok = false; break; }
                    sp.back().hi = std::max(sp.back().hi, out_hi(q));
                } else sp.push_back({out_lo(q), out_hi(q), b});
                pats[q].space = (int)sp.size() - 1;
            }
        }
        ok = ok && sp.size() >= 2 && sp.size() <= 16;
        int64_t n = 0;
        if (ok) {
            int64_t sumb = 0;
            for (const auto &q : sp) sumb += q.b;
            // 6144 doubles of LDS per workgroup (3072 / 4096 / 5120 / 6144 / 7680 measured on the rocket: profiles/NOTES.md)
            n = std::min<int64_t>(kBlock - 2 * spread_max - 2, 6144 / sumb) / 16 * 16;
            ok = n >= 64;
        }
        std::vector<int> pk;
        if (ok) {
            for (size_t q = 0; q < sp.size(); q++) {
                sp[q].W = sp[q].b * n; sp[q].off = Wtot; Wtot += sp[q].W;
                sp[q].o = q == 0 ? 0 : sp[q].lo;
            }
            for (size_t q = 0; q < sp.size(); q++) {
                sp[q].end = q + 1 < sp.size() ? sp[q + 1].o : ncomp;
                nblocks = std::max(nblocks, (sp[q].end - sp[q].o + sp[q].W - 1) / sp[q].W);
            }
            for (const auto &wp : pats) if (std::find(pk.begin(), pk.end(), wp.k) == pk.end()) pk.push_back(wp.k);
            ok = nblocks * (int64_t)pk.size() * 2 < (int64_t)1 << 28;
        }
        if (ok) {
            auto fdiv = [](int64_t a, int64_t b) { int64_t q = a / b; return (a % b != 0 && a < 0) ? q - 1 : q; };
            auto cdiv = [&](int64_t a, int64_t b) { return -fdiv(-a, b); };
            Rb.assign((size_t)nblocks * pk.size() * 2, 0);
            for (int64_t j = 0; j < nblocks && ok; j++)
                for (size_t u = 0; u < pk.size() && ok; u++) {
                    int64_t lo = INT64_MAX, hi = INT64_MIN;
                    for (const auto &wp : pats) {
                        if (wp.k != pk[u]) continue;
                        const Sp &q = sp[wp.space];
                        const int64_t c0 = q.o + j * q.W, c1 = std::min(c0 + q.W, q.end) - 1;
                        if (c1 < c0) continue;
                        const int64_t *t = &Q[wp.qbase];
                        int64_t l = std::max(cdiv(c0 - t[4], t[0]), t[1]), hh = std::min(fdiv(c1 - t[3], t[0]) + 1, t[2]);
                        if (hh <= l) continue;
                        lo = std::min(lo, l); hi = std::max(hi, hh);
                    }
                    if (hi <= lo) { lo = 0; hi = 0; }
                    if (hi - lo > kBlock) ok = false;
                    Rb[(j * pk.size() + u) * 2] = (int32_t)lo; Rb[(j * pk.size() + u) * 2 + 1] = (int32_t)hi;
                }
        }
        if (ok) {
            nspaces = (int)sp.size();
            zs = (int)Q.size();
            w.spaces.clear();
            for (const auto &q : sp) { Q.push_back(q.o); Q.push_back(q.end); Q.push_back(q.W); Q.push_back(q.off); w.spaces.push_back({q.o, q.end, q.W}); }
            if (verbose())
                for (size_t q = 0; q < sp.size(); q++)
                    fprintf(stderr, "[exahip]   space %zu: entries [%ld,%ld) stride %ld window %ld\n", q, (long)sp[q].o, (long)sp[q].end, (long)sp[q].b, (long)sp[q].W);
        } else {
            for (auto &wp : pats) wp.space = 0;
            Rb.clear();
        }
    }
    // Window size.  If every pass advances with the same stride, W = what kBlock points produce (less the straddling
    // points): every pass of every window is one chunk and the straight-line kernel applies (LV 1e7 chess: 0.097 ms
    // against 0.112 with chunk loops at any W).  With mixed strides the small-stride passes need several chunks per
    // window anyway, and large windows win (rocket 1e6 chess, W = 1008 / 2272 / 3024 / 4080: 0.334 / 0.175 / 0.145 /
    // 0.122 ms; cjac 0.090 / 0.056 / 0.054 / 0.057): W = 4080 (32 KB of LDS, 5 workgroups per CU) unless that leaves
    // fewer than ~8 windows per CU.
    int64_t bmin = bmax;
    for (const auto &wp : pats) bmin = std::min<int64_t>(bmin, std::llabs(Q[wp.qbase]));
    int64_t W = std::min<int64_t>((kBlock - spread_max - 1) * bmax, 4096) / 16 * 16;
    single = W >= 16 && W / bmin + spread_max + 1 <= kBlock;
    if (bmax == 1 && bmin == 1 && nspaces == 0) {
        // every pass advances one entry per point: the PLANES form (below) needs no swizzled window, so W is only rounded to
        // whole 64-byte lines of the output: W + spread - 1 points fill the 256 lanes
        bool unit = true;
        for (const auto &wp : pats) unit = unit && Q[wp.qbase] == 1;
        if (unit && kBlock - spread_max + 1 >= 16) { W = (kBlock - spread_max + 1) / 8 * 8; single = true; }
    }
    if (!single) {
        const int64_t fill = ncomp / 2048 / 16 * 16;
        W = std::max<int64_t>(std::min<int64_t>(4080, fill), std::min<int64_t>(W, 1024));
    }
    if (nspaces > 0) { W = Wtot; single = true; }
    if (W < 16) return no("window too small");
    const int64_t nwin = nspaces > 0 ? nblocks : (ncomp + W - 1) / W;
    // work amplification: points evaluated (whole chunks of kBlock) over points present
    double work = 0.0;
    for (const auto &wp : pats) {
        const int64_t ab = std::llabs(Q[wp.qbase]);
        const int64_t n = Q[wp.qbase + 2] - Q[wp.qbase + 1];
        const double per = (double)W / (double)ab + (double)spread_max;
        const double wins = std::min<double>((double)nwin, (double)n * (double)ab / (double)W + 1.0);
        work += wins * std::ceil(per / kBlock) * kBlock;
    }
    if (nspaces == 0 && work > 2.0 * (double)passes_pts + 4096.0 * pats.size())
        return no("windows would evaluate " + std::to_string(work / std::max<double>(1.0, (double)passes_pts)) + "x the points");
    // irregular points: targets straight from the slot map, grouped by distinct target
    w.nx = (int)exc.size();
    w.smax = smax;
    if (w.nx) {
        std::vector<int64_t> X;
        std::vector<int32_t> tgt((size_t)w.nx * smax, -1);
        for (int t = 0; t < w.nx; t++) {
            const Pattern &p = m.pats[exc[t].k];
            X.push_back(exc[t].k); X.push_back(exc[t].I);
            if (product) {
                const auto &ab = items[exc[t].k];
                for (size_t s = 0; s < ab.first.size(); s++) tgt[(size_t)t * smax + s] = (int32_t)(ab.first[s] + ab.second[s] * exc[t].I);
                continue;
            }
            const int S = hess ? p.o2step : p.o1step;
            const int64_t o = h.P[hess ? L.pat[exc[t].k].o2 : L.pat[exc[t].k].o1];
            HIPCHK(hipMemcpy(tgt.data() + (size_t)t * smax, cmap + o + (int64_t)S * exc[t].I, 4 * (size_t)S, hipMemcpyDeviceToHost));
        }
        std::map<int32_t, std::vector<int32_t>> by;
        for (size_t e = 0; e < tgt.size(); e++) if (tgt[e] >= 0) by[tgt[e]].push_back((int32_t)e);
        std::vector<int32_t> T{(int32_t)by.size()}, E;
        for (auto &kv : by) {
            T.push_back(kv.first); T.push_back((int32_t)E.size());
            E.insert(E.end(), kv.second.begin(), kv.second.end());
            T.push_back((int32_t)E.size());
        }
        w.hX = X; w.hT = T; w.hE = E; w.xbuf_doubles = (int64_t)tgt.size();
    }
    // shared entries: workgroup map, partial-sum layout, fold list
    w.ns_blocks = 0;
    w.hF.assign(1, 0);                 // F[0] = 0 groups unless filled below
    w.hS.clear(); w.nparts = 0;
    w.has_shared = !shs.empty();
    if (!shs.empty()) {
        std::vector<int64_t> St, F{0};
        int64_t blocks = 0, parts = 0;
        std::vector<WindowShared> own_kernel, in_kernel;
        for (size_t i = 0; i < shs.size(); i++) {
            const auto &sh = shs[i];
            // one-chunk kernels: a pattern that has a pass in the windows sums its all-points entries INSIDE the window
            // kernel (one partial per window: every regular point belongs to exactly one) — no second evaluation pass
            int attach = -1;
            if (single && !(product ? h.no_attach : h.no_attach_c)) for (size_t q = 0; q < pats.size() && attach < 0; q++) if (pats[q].k == sh.k) attach = (int)q;
            if (attach >= 0) {
                WindowShared r = shared[i];
                r.attach = attach; r.qs = (int)Q.size();
                Q.push_back(parts); Q.push_back(nwin);
                for (size_t g = 0; g < sh.target.size(); g++) { F.push_back(parts + (int64_t)g * nwin); F.push_back(nwin); F.push_back(sh.target[g]); F[0]++; }
                parts += nwin * (int64_t)sh.target.size();
                in_kernel.push_back(std::move(r));
                continue;
            }
            const int64_t per = (int64_t)kBlock * kSharedTiles, nt = (sh.e_hi - sh.e_lo + per - 1) / per;
            St.push_back(sh.e_lo); St.push_back(sh.e_hi); St.push_back(blocks); St.push_back(parts);
            for (size_t g = 0; g < sh.target.size(); g++) { F.push_back(parts + (int64_t)g * nt); F.push_back(nt); F.push_back(sh.target[g]); F[0]++; }
            blocks += nt;
            parts += nt * (int64_t)sh.target.size();
            own_kernel.push_back(shared[i]);
        }
        St.push_back(0); St.push_back(0); St.push_back(blocks); St.push_back(parts);     // sentinel
        shared.swap(own_kernel);
        wm.shared_in.swap(in_kernel);
        w.ns_blocks = blocks;
        w.hS = St; w.hF = F; w.nparts = parts;
    }
    w.hQ = Q;
    // R[window][pass] = [lo, hi): the regular points with a slot of that pass inside the window
    if (nspaces > 0) {
        w.hR = Rb;
    } else {
        auto fdiv = [](int64_t a, int64_t b) { int64_t q = a / b; return (a % b != 0 && a < 0) ? q - 1 : q; };   // b > 0
        auto cdiv = [&](int64_t a, int64_t b) { return -fdiv(-a, b); };
        const size_t np = pats.size();
        std::vector<int32_t> R((size_t)nwin * np * 2);
        for (int64_t j = 0; j < nwin; j++) {
            const int64_t c0 = j * W, c1 = c0 + W - 1;
            for (size_t q = 0; q < np; q++) {
                const int64_t *t = &Q[pats[q].qbase];
                const int64_t b = t[0], amin = t[3], amax = t[4];
                int64_t lo, hi;
                if (b > 0) { lo = cdiv(c0 - amax, b); hi = fdiv(c1 - amin, b) + 1; }
                else { lo = cdiv(amin - c1, -b); hi = fdiv(amax - c0, -b) + 1; }
                lo = std::max(lo, t[1]); hi = std::min(hi, t[2]);
                if (hi < lo) hi = lo;
                R[(j * np + q) * 2] = (int32_t)lo; R[(j * np + q) * 2 + 1] = (int32_t)hi;
            }
        }
        w.hR.swap(R);
        w.spaces.assign(1, {0, ncomp, W});
    }
    w.W = (int)W;
    w.nwin = nwin;
    {
        // PLANES form (WindowMatrix::planes): one-chunk kernels whose passes all advance by one entry per data point
        size_t groups = 0;
        bool unit = single;
        for (const auto &wp : pats) { groups += wp.phase.size(); unit = unit && Q[wp.qbase] == 1; }
        wm.planes = unit && groups > 0 && groups * kBlock * 8 <= 65536;
        w.lds_bytes = wm.planes ? (int)(groups * kBlock * 8) : (int)(8 * W);
    }
    w.why = (nspaces > 0 ? "block-owned windows, " + std::to_string(nspaces) + " spaces" : (single ? "one chunk per pass" : "chunk loops")) + (wm.planes ? ", planes" : "");
    if (verbose()) {
        fprintf(stderr, "[exahip] windowed %s (%s): W=%ld windows=%ld passes=%zu shared-entry workgroups=%ld irregular points=%d\n", wk == WK_CHESS ? "hess" : wk == WK_CJAC ? "jac" : wk == WK_JTPROD ? "jtprod" : "hprod", nspaces > 0 ? "block-owned, one evaluation per point" : (single ? "one chunk per pass" : "chunk loops"), (long)W,
                (long)nwin, pats.size(), (long)w.ns_blocks, w.nx);
        for (size_t q = 0; q < pats.size(); q++) {
            const int64_t *t = &Q[pats[q].qbase];
            fprintf(stderr, "[exahip]   pass %zu: pattern %d  b=%ld  points [%ld,%ld)  targets %ld..%ld  groups=%zu\n", q, pats[q].k, (long)t[0], (long)t[1], (long)t[2],
                    (long)t[3], (long)t[4], pats[q].phase.size());
        }
    }
    return true;
}
// device copies of a planned window's tables (the host copies are dropped: R alone is 8 B per window and pass)
void window_upload(Handle::Window &w) {
    auto up = [](DevBuf &b, const void *src, size_t bytes) { b.ensure(std::max<size_t>(bytes, 8)); if (bytes) HIPCHK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice)); };
    up(w.Q, w.hQ.data(), 8 * w.hQ.size()); up(w.R, w.hR.data(), 4 * w.hR.size());
    up(w.X, w.hX.data(), 8 * w.hX.size()); up(w.T, w.hT.data(), 4 * w.hT.size()); up(w.E, w.hE.data(), 4 * w.hE.size());
    up(w.S, w.hS.data(), 8 * w.hS.size()); up(w.F, w.hF.data(), 8 * w.hF.size());
    w.xbuf.ensure(8 * (size_t)std::max<int64_t>(w.xbuf_doubles, 1)); w.part.ensure(8 * (size_t)std::max<int64_t>(w.nparts, 1));
    for (auto *v : {&w.hQ, &w.hX, &w.hS, &w.hF}) std::vector<int64_t>().swap(*v);
    for (auto *v : {&w.hT, &w.hE}) std::vector<int32_t>().swap(*v);      // (hR stays: exa_shard_var_range reads the owned windows' point ranges)
}

// Owner-computes products (exa_jtprodw / exa_hprodw): planned on the host at model build — also for exa_plan_only handles, so
// that exa_compile / exahip.pack build the module ahead of time — whenever every scatter target of J'v / Hv is affine in
// a range column.  EXAHIP_PRODUCT_WINDOW=0 keeps the atomics / the sorted gather.
void plan_products(Handle &h) {
    h.pspec = WindowSpec();
    h.psource.clear();
    const char *env = getenv("EXAHIP_PRODUCT_WINDOW");
    bool any = false;
    for (int wk : {WK_JTPROD, WK_HPROD}) {
        Handle::Window &w = window_of(h, wk);
        w.ok = w.planned = false; w.has_shared = false; w.why.clear(); w.nx = 0; w.ns_blocks = 0; w.nwin = 0;
        if (env && atoi(env) == 0) { w.why = "disabled (EXAHIP_PRODUCT_WINDOW=0)"; continue; }
        w.planned = window_plan(h, wk, nullptr, h.pspec.mat[wk]);
        if (!w.planned) h.pspec.mat[wk] = WindowMatrix();
        any = any || w.planned;
    }
    if (any) h.psource = generate_window_module(*h.m, h.gen.layout, h.pspec);
    // no windows because a target is reached through a data column: the owner-pull kernels instead (exa_gen_pull.cpp)
    const char *pe = getenv("EXAHIP_PRODUCT_PULL");
    for (int k = 0; k < 2; k++) { h.pl[k].planned = h.pl[k].ready = false; h.pl[k].why.clear(); h.pl[k].nitems.clear(); h.pl[k].total = 0; }
    if (!any && !(pe && atoi(pe) == 0)) {
        bool want[2] = {false, false};
        for (int k = 0; k < 2; k++) {
            const int cb = k ? CB_HPROD : CB_JTPROD;
            if (h.gen.layout.active[cb].empty() || h.wp[k].why.find("data column") == std::string::npos) continue;
            h.pl[k].nitems = pull_item_counts(*h.m, h.gen.layout, cb);
            int tot = 0;
            for (int n : h.pl[k].nitems) tot += n;
            want[k] = tot > 0 && tot <= 256;          // (one specialised function per item: bounded module size)
            h.pl[k].planned = want[k];
        }
        if (want[0] || want[1]) h.psource = generate_pull_module(*h.m, h.gen.layout, want[0], want[1]);
    }
}
// A window module, compiled or fetched — and ASKED (see audited_code_object).  Window kernels that sum the all-points entries
// inside themselves (exa_block_sum) are first given the chance to fit by a re-plan: those sums in a kernel of their own
// (no_attach) — tests/sweeps/window_sweep.py 227 1 blocks was a 12-pass Hv kernel with 820 B of scratch per lane.  What still
// does not fit the 256 architectural VGPRs keeps its windows and is compiled with the conservative allocator flags.
bool window_kernels_spill(const CodeObject &co, const WindowSpec &spec, int wk) {
    static const char *nm[WK_COUNT] = {"exa_cjac", "exa_chess", "exa_jtprod", "exa_hprod"};
    const WindowMatrix &wm = spec.mat[wk];
    if (wm.pats.empty()) return false;
    std::vector<KernelInfo> ks;
    if (!code_object_kernels(co.image, ks)) return true;        // unreadable metadata: assume the worst
    bool bad = false;
    for (const char *sfx : {"w", "s"}) {
        if (sfx[0] == 's' && wm.shared.empty()) continue;
        const std::string name = std::string(nm[wk]) + sfx;
        for (const KernelInfo &k : ks) {
            if (k.name != name) continue;
            bad = bad || !k.fits();
            if (verbose()) fprintf(stderr, "[exahip] %s: %d VGPRs, %d AGPRs, %d bytes of scratch per lane, %d VGPRs / %d SGPRs spilled%s\n", name.c_str(), k.vgpr, k.agpr, k.scratch, k.vgpr_spill, k.sgpr_spill, k.fits() ? "" : "  <- beyond the architectural registers");
        }
    }
    return bad;
}
CodeObject product_module_for(Handle &h, bool memory_only_ok) {
    CodeObject co = get_code_object(h.psource, memory_only_ok, prefer_safe(h.psource));
    double spent = 0.0;
    bool attached = false;
    for (int wk : {WK_JTPROD, WK_HPROD}) attached = attached || !h.pspec.mat[wk].shared_in.empty();
    // (EXAHIP_WINDOW_REPLAN=0, test infrastructure: keep the first plan — the canary's over-sized kernel — and go straight to the flags)
    static const bool replan = [] { const char *e = getenv("EXAHIP_WINDOW_REPLAN"); return !(e && atoi(e) == 0); }();
    if (replan && !h.no_attach && attached && (window_kernels_spill(co, h.pspec, WK_JTPROD) || window_kernels_spill(co, h.pspec, WK_HPROD))) {
        h.no_attach = true;
        plan_products(h);
        spent = co.build_ms;
        if (h.psource.empty()) return CodeObject();
    }
    CodeObject fin = audited_code_object(h, "products", h.psource, memory_only_ok, &co);
    fin.build_ms += spent;
    return fin;
}
// loads the product module and uploads the tables; a module that cannot be built leaves the products on their other paths
void load_products(Handle &h) {
    if (h.psource.empty()) return;
    try {
        CodeObject co = product_module_for(h, true);
        if (h.psource.empty()) return;
        h.phsaco_path = co.path; h.build_ms += co.build_ms; h.pco_name = co.name;
        HIPCHK(hipModuleLoadData(&h.pmodule, co.image.data()));
        auto fn = [&](const std::string &name) { hipFunction_t f; HIPCHK(hipModuleGetFunction(&f, h.pmodule, name.c_str())); return f; };
        for (int wk : {WK_JTPROD, WK_HPROD}) {
            Handle::Window &w = window_of(h, wk);
            if (!w.planned) continue;
            const std::string nm = wk == WK_JTPROD ? "exa_jtprod" : "exa_hprod";
            w.fw = fn(nm + "w"); w.fx = fn(nm + "x");
            if (w.ns_blocks) w.fs = fn(nm + "s");
            window_upload(w);
            w.ok = true;
        }
        for (int k = 0; k < 2; k++) {
            if (!h.pl[k].planned) continue;
            h.pl[k].fkeys = fn(k ? "exa_hpkeys" : "exa_jtkeys");
            h.pl[k].fpull = fn(k ? "exa_hppull" : "exa_jtpull");
        }
    } catch (const std::exception &e) {
        std::string msg = e.what();
        if (msg.size() > 300) msg.resize(300);
        for (int wk : {WK_JTPROD, WK_HPROD}) { Handle::Window &w = window_of(h, wk); if (w.planned) { w.ok = false; w.why = "the window kernels could not be built (" + msg + ")"; } }
        for (auto &q : h.pl) if (q.planned) { q.planned = false; q.fkeys = q.fpull = nullptr; q.why = "the owner-pull kernels could not be built (" + msg + ")"; }
        if (h.pmodule) { (void)hipModuleUnload(h.pmodule); h.pmodule = nullptr; }
    }
}

void window_setup(Handle &h) {
    // exa_compress may be called again (e.g. with another EXAHIP_CWINDOW): start from scratch
    for (Handle::Window *w : {&h.wj, &h.wh}) { w->ok = false; w->has_shared = false; w->why.clear(); w->nx = 0; w->ns_blocks = 0; w->nwin = 0; }
    h.sj.ok = h.sh.ok = false; h.sj.f = h.sh.f = nullptr;
    h.merged = false; h.f_chessm = h.f_hstructm = nullptr; h.chm.release();
    if (h.wmodule) { (void)hipModuleUnload(h.wmodule); h.wmodule = nullptr; }
    const char *env = getenv("EXAHIP_CWINDOW");
    if (env && atoi(env) == 0) { h.wj.why = h.wh.why = "disabled (EXAHIP_CWINDOW=0)"; return; }
    h.sj.ok = h.sh.ok = false;
    const bool plan_windows = true;                 // (a shard plans the windows of its local slice: absolute point indices throughout)
    const Model &m = *h.m;
    if (std::max(h.lnnzj, h.lnnzh) > 0x7fffffffLL) { h.wj.why = h.wh.why = "nnz exceeds int32"; return; }
    WindowSpec spec;
    DevBuf cmap;
    cmap.ensure(4 * (size_t)std::max<int64_t>(std::max(h.lnnzj, h.lnnzh), 1));
    bool okj = false, okh = false;
    if (plan_windows) try {
        build_slot_map(h.cj, (int32_t *)cmap.p, h.stream);
        HIPCHK(hipStreamSynchronize(h.stream));
        okj = window_plan(h, WK_CJAC, (const int32_t *)cmap.p, spec.mat[WK_CJAC]);
        if (!okj) spec.mat[WK_CJAC] = WindowMatrix();
        else window_upload(h.wj);
        build_slot_map(h.ch, (int32_t *)cmap.p, h.stream);
        HIPCHK(hipStreamSynchronize(h.stream));
        okh = window_plan(h, WK_CHESS, (const int32_t *)cmap.p, spec.mat[WK_CHESS]);
        if (!okh) spec.mat[WK_CHESS] = WindowMatrix();
        else window_upload(h.wh);
    } catch (...) { cmap.release(); throw; }
    cmap.release();
    // what the windows do not cover goes through the permuted store when it can: 32-bit positions, no entry with more
    // than 512 duplicates (those are summed cooperatively through the gather lists)
    h.sj.ok = h.sh.ok = false;
    const char *se = getenv("EXAHIP_CSCATTER");
    const bool scatter_on = !(se && atoi(se) == 0);
    spec.jac_scatter = scatter_on && !okj && h.cj.nnz > 0 && h.cj.nlong == 0;
    spec.hess_scatter = scatter_on && !okh && h.ch.nnz > 0 && h.ch.nlong == 0;
    // Hessian: merged slots when the fused groups collapse enough of them (ACOPF: 5.7 M slots -> 1.9 M)
    std::vector<int64_t> M;
    if (spec.hess_scatter) {
        const ParamLayout &L = h.gen.layout;
        const std::vector<int> sm = merged_hess_slots(m, L);
        int64_t nm = 0;
        for (size_t g = 0; g < L.groups[CB_HESS].size(); g++) {
            const auto &pp = L.pat[L.groups[CB_HESS][g].front()];
            M.push_back(nm);
            nm += (int64_t)sm[g] * (h.P[pp.hi] - h.P[pp.lo]);
        }
        if (nm > 0 && nm < 0xffffffffLL && (double)nm <= 0.8 * (double)h.ch.nnz) { spec.hess_merged = true; h.nmerged = nm; }
    }
    if (!okj && !okh && !spec.jac_scatter && !spec.hess_scatter) return;
    const std::string src = generate_window_module(m, h.gen.layout, spec);
    std::vector<char> image;
    // the gather path needs no second module: a host without hipcc (a packed library's consumer) or a failed compilation
    // must not take exa_compress down with it
    try {
        CodeObject wco = get_code_object(src, true, prefer_safe(src));
        bool attached = !spec.mat[WK_CJAC].shared_in.empty() || !spec.mat[WK_CHESS].shared_in.empty();
        if (!h.no_attach_c && attached && (window_kernels_spill(wco, spec, WK_CJAC) || window_kernels_spill(wco, spec, WK_CHESS)))
            throw std::runtime_error("a window kernel that sums the all-points entries spills registers");     // exa_compress plans again (no_attach_c)
        wco = audited_code_object(h, "compressed", src, true, &wco);
        image = wco.image;
        HIPCHK(hipModuleLoadData(&h.wmodule, image.data()));
        auto fn = [&](const char *name) { hipFunction_t f; HIPCHK(hipModuleGetFunction(&f, h.wmodule, name)); return f; };
        if (okj) { h.wj.fw = fn("exa_cjacw"); h.wj.fx = fn("exa_cjacx"); if (h.wj.ns_blocks) h.wj.fs = fn("exa_cjacs"); }
        if (okh) { h.wh.fw = fn("exa_chessw"); h.wh.fx = fn("exa_chessx"); if (h.wh.ns_blocks) h.wh.fs = fn("exa_chesss"); }
        if (spec.jac_scatter) h.sj.f = fn("exa_cjacp");
        if (spec.hess_scatter) h.sh.f = fn("exa_chessp");
        if (spec.hess_merged) { h.f_chessm = fn("exa_chessm"); h.f_hstructm = fn("exa_hstructm"); }
    } catch (const std::exception &e) {
        std::string msg = e.what();
        if (msg.size() > 300) msg.resize(300);
        h.wj.why = h.wh.why = "the windowed kernels could not be built (" + msg + ")";
        if (h.wmodule) { (void)hipModuleUnload(h.wmodule); h.wmodule = nullptr; }
        return;
    }
    // a matrix on the windowed sweep never gathers: its sorted permutation (4 B per uncompressed slot: 3.6 GB for LV 1e8)
    // and pointer list can go
    if (okj) { h.cj.release_gather(); h.wj.ok = true; }
    if (okh) { h.ch.release_gather(); h.wh.ok = true; }
    if (spec.hess_merged && h.f_chessm) {
        // structure of the merged slot space -> its own sorted lists; it must describe the same matrix as the slots'
        DevBuf r, c;
        try {
            h.dM.ensure(8 * M.size());
            HIPCHK(hipMemcpy(h.dM.p, M.data(), 8 * M.size(), hipMemcpyHostToDevice));
            r.ensure(8 * (size_t)h.nmerged); c.ensure(8 * (size_t)h.nmerged);
            const void *P = h.dP.p, *Mp = h.dM.p;
            void *rp = r.p, *cp = c.p;
            void *a[] = {&P, &rp, &cp, &Mp};
            launch(h, h.f_hstructm, h.grid[CB_HESS], kBlock, a);
            build_compressed(h.chm, (const int64_t *)r.p, (const int64_t *)c.p, h.nmerged, std::max<int64_t>(m.nvar, 1), std::max<int64_t>(m.nvar, 1), h.stream);
            HIPCHK(hipStreamSynchronize(h.stream));
        } catch (...) { r.release(); c.release(); throw; }
        r.release(); c.release();
        if (h.chm.cnnz == h.ch.cnnz && h.chm.nlong == 0) {
            h.sh.pos.ensure(4 * (size_t)h.nmerged);
            build_positions(h.chm, (uint32_t *)h.sh.pos.p, h.stream);
            HIPCHK(hipStreamSynchronize(h.stream));
            h.merged = true;
            h.sh.ok = true;
            h.wh.why = "merged slots (" + std::to_string(h.nmerged) + " for " + std::to_string(h.ch.nnz) + "), permuted store + sequential sums";
            h.ch.release_gather();
        } else h.chm.release();
    }
    for (int hess = 0; hess < 2; hess++) {
        Handle::Scatter &sc = hess ? h.sh : h.sj;
        CompressedCOO &cc = hess ? h.ch : h.cj;
        if (hess && h.merged) continue;
        if (!(hess ? spec.hess_scatter : spec.jac_scatter) || !sc.f) continue;
        sc.pos.ensure(4 * (size_t)cc.nnz);
        build_positions(cc, (uint32_t *)sc.pos.p, h.stream);
        HIPCHK(hipStreamSynchronize(h.stream));
        sc.ok = true;
        (hess ? h.wh : h.wj).why = cc.cnnz == cc.nnz ? "permuted store (no duplicates: the sweep writes the compressed entries directly)"
                                                      : "permuted store + sequential sums of the sorted duplicates";
    }
}
void do_scatter(Handle &h, bool hess, const double *x, const double *y, double sigma, double *vals) {
    Handle::Scatter &sc = hess ? h.sh : h.sj;
    const CompressedCOO &cc = hess ? h.ch : h.cj;
    const void *P = h.dP.p, *th = h.dtheta.p, *pos = sc.pos.p;
    if (hess && h.merged) {
        const bool direct = h.chm.cnnz == h.chm.nnz;
        double *out = direct ? vals : (double *)h.cbuf.p;
        const void *Mp = h.dM.p;
        void *a[] = {&P, &x, &y, &th, &out, &sigma, &pos, &Mp};
        launch(h, h.f_chessm, h.grid[CB_HESS], kBlock, a);
        if (!direct) compress_sorted(h.chm, out, vals, h.stream);
        return;
    }
    const bool direct = cc.cnnz == cc.nnz;          // a permutation: the sorted order IS the compressed array
    double *out = direct ? vals : (double *)h.cbuf.p;
    if (hess) { void *a[] = {&P, &x, &y, &th, &out, &sigma, &pos}; launch(h, sc.f, h.grid[CB_HESS], kBlock, a); }
    else { void *a[] = {&P, &x, &th, &out, &pos}; launch(h, sc.f, h.grid[CB_JAC], kBlock, a); }
    if (!direct) compress_sorted(cc, out, vals, h.stream);
}

// wk: which window kernel set (WKind); v: the vector of a product (null for the compressed COO).  [w0, w1): the windows
// this launch evaluates — all of them, or the ones a rank of an owner-sharded product owns.
void do_window(Handle &h, int wk, const double *x, const double *y, const double *v, double sigma, double *vals, int64_t w0, int64_t w1) {
    Handle::Window &w = window_of(h, wk);
    const void *P = h.dP.p, *Q = w.Q.p, *R = w.R.p, *th = h.dtheta.p;
    int64_t ncomp = wk == WK_CHESS ? h.ch.cnnz : wk == WK_CJAC ? h.cj.cnnz : h.m->nvar;
    int W = w.W;
    void *part = w.part.p;
    const int64_t ns = w.ns_blocks;
    if (w1 < 0) w1 = w.nwin;
    if (ns) {
        const void *S = w.S.p;
        void *a1[] = {&P, &S, &x, &y, &th, &v, &part, &sigma};
        HIPCHK(hipModuleLaunchKernel(w.fs, (unsigned)ns, 1, 1, kBlock, 1, 1, 0, h.stream, a1, nullptr));
    }
    void *a[] = {&P, &Q, &R, &x, &y, &th, &v, &vals, &sigma, &ncomp, &W, &w0, &part};
    if (w1 > w0) HIPCHK(hipModuleLaunchKernel(w.fw, (unsigned)(w1 - w0), 1, 1, kBlock, 1, 1, (unsigned)w.lds_bytes, h.stream, a, nullptr));
    if (w.nx || w.has_shared) {
        // tail: the irregular end points, then the fold of the shared-entry partial sums (one workgroup)
        const void *X = w.X.p, *T = w.T.p, *E = w.E.p, *F = w.F.p;
        void *xbuf = w.xbuf.p;
        int nx = w.nx;
        void *a2[] = {&P, &X, &T, &E, &x, &y, &th, &v, &xbuf, &vals, &sigma, &nx, &part, &F};
        HIPCHK(hipModuleLaunchKernel(w.fx, 1, 1, 1, 512, 1, 1, 0, h.stream, a2, nullptr));      // one workgroup: the irregular points, then the fold
    }
}
int cstruct(int id, bool hess, bool wide, void *r, void *c) {
    return guard(id, true, [&](Handle &h) {
        if (!h.compressed) throw BadInput("exa_compress has not been called");
        const CompressedCOO &cc = hess ? h.ch : h.cj;
        if (!wide && cc.cnnz > 0x7fffffffLL) throw std::runtime_error("nnz exceeds int32");
        compressed_structure(cc, r, c, wide, h.stream);
    });
}
int ccsc(int id, bool hess, int64_t *colptr, int64_t *rowval) {
    if (!colptr || !rowval) return 1;
    return guard(id, true, [&](Handle &h) {
        if (!h.compressed) throw BadInput("exa_compress has not been called");
        compressed_csc(hess ? h.ch : h.cj, h.m->nvar, colptr, rowval, h.stream);
    });
}

}  // namespace rt
}  // namespace exa

extern "C" {


// ---- compressed COO (CompressedNLPModel, src/utils.jl:425-579) ---------------------------------------------
int exa_compress(int id) {
    return guard(id, true, [&](Handle &h) {
        // A sharded model compresses the COO it evaluates: its local slice.  Every rank then holds a duplicate-summed
        // matrix of its own data points (its own structure, exa_c*_structure); the model's matrix is the SUM of the ranks'
        // matrices — entries that data points of two ranks share (stencil neighbours at a shard boundary, bus rows) appear
        // on both, which is what a distributed assembly expects.
        if (h.world != 1 && !h.coo_local) throw BadInput("exa_compress of a sharded model needs the local-slice COO (exa_set_coo_local)");
        const Model &m = *h.m;
        const int64_t nnzj = h.lnnzj, nnzh = h.lnnzh;
        const int64_t mx = std::max<int64_t>(std::max(nnzj, nnzh), 1);
        DevBuf r, c;
        r.ensure(8 * (size_t)mx); c.ensure(8 * (size_t)mx);
        try {
            do_struct(h, false, true, r.p, c.p);
            build_compressed(h.cj, (const int64_t *)r.p, (const int64_t *)c.p, nnzj, std::max<int64_t>(m.ncon, 1), std::max<int64_t>(m.nvar, 1), h.stream);
            do_struct(h, true, true, r.p, c.p);
            build_compressed(h.ch, (const int64_t *)r.p, (const int64_t *)c.p, nnzh, std::max<int64_t>(m.nvar, 1), std::max<int64_t>(m.nvar, 1), h.stream);
        } catch (...) { r.release(); c.release(); throw; }
        r.release(); c.release();
        window_setup(h);
        if (!h.no_attach_c && (h.wj.why.find("spills registers") != std::string::npos || h.wh.why.find("spills registers") != std::string::npos)) {
            h.no_attach_c = true;          // once more with the all-points entries summed by the kernel of their own
            window_setup(h);
        }
        if (!(h.wj.ok || nnzj == 0) || !(h.wh.ok || nnzh == 0)) h.cbuf.ensure(8 * (size_t)mx);
        h.compressed = true;
    });
}
int64_t exa_cnnzj64(int id) { Handle *h = get(id); return h && h->compressed ? h->cj.cnnz : -1; }
int64_t exa_cnnzh64(int id) { Handle *h = get(id); return h && h->compressed ? h->ch.cnnz : -1; }
int exa_cjac_structure(int id, int32_t *r, int32_t *c) { return cstruct(id, false, false, r, c); }
int exa_chess_structure(int id, int32_t *r, int32_t *c) { return cstruct(id, true, false, r, c); }
int exa_cjac_structure64(int id, int64_t *r, int64_t *c) { return cstruct(id, false, true, r, c); }
int exa_chess_structure64(int id, int64_t *r, int64_t *c) { return cstruct(id, true, true, r, c); }
int exa_cjac_csc(int id, int64_t *colptr, int64_t *rowval) { return ccsc(id, false, colptr, rowval); }
int exa_chess_csc(int id, int64_t *colptr, int64_t *rowval) { return ccsc(id, true, colptr, rowval); }
int exa_compress_info(int id, int hess, char *buf, int cap, int *len_out) {
    Handle *h = get(id);
    if (!h || !h->compressed) return -1;
    const Handle::Window &w = hess ? h->wh : h->wj;
    const std::string &why = w.why;
    if (len_out) *len_out = (int)why.size();
    if (buf && cap > 0) {
        const int c = std::min<int>(cap - 1, (int)why.size());
        memcpy(buf, why.data(), (size_t)c);
        buf[c] = 0;
    }
    return w.ok ? 1 : ((hess ? h->sh.ok : h->sj.ok) ? 2 : 0);
}
int exa_cjac(int id, const double *x, double *vals) {
    if (!x) return 1;
    return guard(id, true, [&](Handle &h) {
        if (!h.compressed) throw BadInput("exa_compress has not been called");
        if (h.wj.ok) { do_window(h, WK_CJAC, x, nullptr, nullptr, 0.0, vals); return; }
        if (h.sj.ok) { do_scatter(h, false, x, nullptr, 0.0, vals); return; }
        do_jac(h, x, (double *)h.cbuf.p);
        compress_values(h.cj, (const double *)h.cbuf.p, vals, h.stream);
    });
}
int exa_chess(int id, const double *x, const double *y, double w, double *vals) {
    if (!x) return 1;
    return guard(id, true, [&](Handle &h) {
        if (!h.compressed) throw BadInput("exa_compress has not been called");
        // y == NULL = the objective-only form (as exa_hess).  The window / permuted-store kernels evaluate EVERY pattern and read y (and
        // have released the gather lists the other route needs): on a constrained model they refuse it instead of faulting on the GPU;
        // the uncompressed evaluation + gather handles it (objective groups alone, the rest zero-filled)
        if (!y && h.m->ncon > 0 && (h.wh.ok || h.sh.ok))
            throw BadInput("exa_chess: y == NULL (objective only) is not available where the compressed Hessian runs by windows / permuted store: pass y = 0");
        if (h.wh.ok) { do_window(h, WK_CHESS, x, y, nullptr, w, vals); return; }
        if (h.sh.ok) { do_scatter(h, true, x, y, w, vals); return; }
        do_hess(h, x, y, w, (double *)h.cbuf.p);
        compress_values(h.ch, (const double *)h.cbuf.p, vals, h.stream);
    });
}

// ---- test infrastructure: one launch of a product window kernel as a self-contained file ---------------------------------------
/* Writes everything ONE launch of exa_jtprodw (hess = 0) / exa_hprodw (hess = 1) needs into `path` — grid, block, LDS bytes,
 * every argument (scalars by value, buffers by content) and the output THIS build of the kernel produces (NaN where it writes
 * nothing) — and the module's source into `path`.hip.  tests/sweeps/canary/canary_host.cpp replays such a file against a code
 * object compiled from that source with any compiler and flags, without this library: the reproducer of the wrong sums an
 * over-sized window kernel returns under the default register allocator (profiles/NOTES.md).  Format: "EXADUMP1", then
 * int64 words and raw bytes as written below.  Status 1 when the model has no such windows. */
int exa_debug_dump_window_launch(int id, int hess, const double *x, const double *y, const double *v, double sigma, const char *path) {
    if (!x || !v || !path) return 1;
    return guard(id, true, [&](Handle &h) {
        Handle::Window &w = h.wp[hess ? 1 : 0];
        if (!w.ok) throw BadInput("no product windows on this model: " + w.why);
        const Model &m = *h.m;
        std::vector<double> expect((size_t)m.nvar, std::numeric_limits<double>::quiet_NaN());
        DevBuf out;
        out.ensure(8 * expect.size());
        struct Rel { DevBuf &b; ~Rel() { b.release(); } } rel{out};
        HIPCHK(hipMemcpy(out.p, expect.data(), 8 * expect.size(), hipMemcpyHostToDevice));
        const void *P = h.dP.p, *Q = w.Q.p, *R = w.R.p, *th = h.dtheta.p;
        int64_t ncomp = m.nvar, w0 = 0;
        int W = w.W;
        void *part = w.part.p, *vals = out.p;
        if (hess && m.ncon > 0 && !y) throw BadInput("the recorded launch evaluates every pattern: y is needed");
        const double *yy = hess ? y : nullptr;
        if (w.ns_blocks) {
            const void *S = w.S.p;
            void *a1[] = {&P, &S, &x, &yy, &th, &v, &part, &sigma};
            HIPCHK(hipModuleLaunchKernel(w.fs, (unsigned)w.ns_blocks, 1, 1, kBlock, 1, 1, 0, h.stream, a1, nullptr));
        }
        void *a[] = {&P, &Q, &R, &x, &yy, &th, &v, &vals, &sigma, &ncomp, &W, &w0, &part};
        HIPCHK(hipModuleLaunchKernel(w.fw, (unsigned)w.nwin, 1, 1, kBlock, 1, 1, (unsigned)w.lds_bytes, h.stream, a, nullptr));
        HIPCHK(hipStreamSynchronize(h.stream));
        HIPCHK(hipMemcpy(expect.data(), out.p, 8 * expect.size(), hipMemcpyDeviceToHost));
        std::ofstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error(std::string("cannot write ") + path);
        auto word = [&](int64_t q) { f.write((const char *)&q, 8); };
        auto scalar = [&](const void *q, int64_t n) { word(0); word(n); f.write((const char *)q, n); };
        auto buffer = [&](const void *dev, int64_t n, int64_t kind) {       // kind 1 input, 2 the output (contents = NaN fill)
            std::vector<char> tmp((size_t)std::max<int64_t>(n, 8), 0);
            if (dev && n) HIPCHK(hipMemcpy(tmp.data(), dev, (size_t)n, hipMemcpyDeviceToHost));
            word(kind); word((int64_t)tmp.size()); f.write(tmp.data(), (std::streamsize)tmp.size());
        };
        f.write("EXADUMP1", 8);
        const std::string kname = hess ? "exa_hprodw" : "exa_jtprodw";
        word((int64_t)kname.size()); f.write(kname.data(), (std::streamsize)kname.size());
        word(w.nwin); word(kBlock); word(w.lds_bytes); word(13);
        buffer(P, 8 * (int64_t)h.P.size(), 1); buffer(Q, (int64_t)w.Q.bytes, 1); buffer(R, (int64_t)w.R.bytes, 1); buffer(x, 8 * m.nvar, 1);
        buffer(yy, yy ? 8 * m.ncon : 0, 1); buffer(th, (int64_t)h.dtheta.bytes, 1); buffer(v, 8 * (hess ? m.nvar : std::max<int64_t>(m.ncon, 1)), 1);
        word(2); word(8 * m.nvar);                                           // the output: the host fills it with NaN
        scalar(&sigma, 8); scalar(&ncomp, 8); scalar(&W, 4); scalar(&w0, 8);
        buffer(part, (int64_t)w.part.bytes, 1);
        word(m.nvar); f.write((const char *)expect.data(), (std::streamsize)(8 * expect.size()));
        f.close();
        std::ofstream g(std::string(path) + ".hip", std::ios::binary);
        g << h.psource;
    });
}

}  // extern "C"
