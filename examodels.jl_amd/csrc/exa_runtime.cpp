// exa_runtime.cpp — libexahip.so runtime + C ABI (include/exahip.h).
//
// Owns: model registry, kernel-module build (hipcc --genco for gfx950, cached on disk by source hash), device
// copies of the SoA iterator columns / theta / parameter table, launches on the model's HIP stream.
// Replaces the host drivers of ext/ExaModelsKernelAbstractions.jl:253-351, 515-547 (one launch per pattern per
// callback + fill!) with ONE fused launch per callback and no fill! for the COO outputs.
#include "exa_rt.hpp"

using namespace exa;
using namespace exa::rt;

namespace exa {
namespace rt {

thread_local std::string g_err;
std::mutex g_mu;


std::vector<std::unique_ptr<Handle>> g_models;   // id = index + 1

Handle *get(int id) {
    g_err.clear();          // exa_last_error() describes the LAST failed call: every entry point comes through here first
    std::lock_guard<std::mutex> lk(g_mu);
    if (id < 1 || id > (int)g_models.size()) return nullptr;
    return g_models[id - 1].get();
}

int put(std::unique_ptr<Handle> h) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = 0; i < g_models.size(); i++)
        if (!g_models[i]) { g_models[i] = std::move(h); return (int)i + 1; }
    g_models.push_back(std::move(h));
    return (int)g_models.size();
}

// What a measured decision depends on: device, shard, sizes of every pattern (the module itself is the file's name).
std::string tune_signature(const Handle &h, const std::string &what) {
    std::string t = h.devname + "|" + std::to_string(h.rank) + "/" + std::to_string(h.world) + "|" + (h.coo_local ? "L" : "G");
    for (const Pattern &p : h.m->pats) t += "," + std::to_string(p.n);
    return what + ":" + sha256_hex(t).substr(0, 16);
}
// variables rank r of a sharded model owns (owner-computes grad!): [own_var_lo(r), own_var_lo(r + 1))
int64_t own_var_lo(const Handle &h, int r) { return part_lo(h.m->nvar, r, h.world); }

// ---- parameter table -------------------------------------------------------------------------------------
void fill_params(Handle &h) {
    const Model &m = *h.m;
    const ParamLayout &L = h.gen.layout;
    h.P.assign((size_t)L.nwords, 0);
    // Local-slice COO (exa_set_coo_local): this rank's slots of pattern k are the contiguous global range
    // [o + step*lo, o + step*hi); packed pattern after pattern they start at local offset l.  Every kernel addresses a
    // slot as P[o] + step*I, so installing P[o] = l - step*lo redirects all of them (values and structures) at once.
    const bool local = h.coo_local && h.world > 1;
    int64_t l1 = 0, l2 = 0;
    h.lo1.assign(m.pats.size(), 0); h.lo2.assign(m.pats.size(), 0);
    for (size_t k = 0; k < m.pats.size(); k++) {
        const Pattern &p = m.pats[k];
        const auto &pp = L.pat[k];
        const int64_t lo = part_lo(p.n, h.rank, h.world), hi = part_lo(p.n, h.rank + 1, h.world);
        h.P[pp.lo] = lo; h.P[pp.hi] = hi; h.P[pp.o0] = p.o0; h.P[pp.o1] = p.o1; h.P[pp.o2] = p.o2; h.P[pp.oa] = p.oa;
        // gathered objective patterns: the points of the whole pattern that touch the variables this rank owns
        h.P[pp.qlo] = 0; h.P[pp.qhi] = p.n;
        if (h.world > 1 && std::find(L.pull.begin(), L.pull.end(), (int)k) != L.pull.end()) {
            int64_t jlo = 0, jhi = 0;
            pull_point_range(p, own_var_lo(h, h.rank) + 1, own_var_lo(h, h.rank + 1), &jlo, &jhi);
            h.P[pp.qlo] = jlo; h.P[pp.qhi] = jhi;
        }
        h.lo1[k] = l1; h.lo2[k] = l2;
        if (local) {
            if (p.kind != EXA_PAT_OBJ) { h.P[pp.o1] = l1 - (int64_t)p.o1step * lo; l1 += (int64_t)p.o1step * (hi - lo); }
            h.P[pp.o2] = l2 - (int64_t)p.o2step * lo; l2 += (int64_t)p.o2step * (hi - lo);
        }
        for (size_t c = 0; c < p.cols.size(); c++) {
            if (p.cols[c].type == EXA_COL_RANGE) h.P[pp.col[c]] = p.cols[c].start;
            else h.P[pp.col[c]] = h.on_device ? (int64_t)(uintptr_t)h.dcols[h.colslot[k][c]].p : 0;
        }
    }
    if (h.on_device) h.dsink.ensure(8 * 64);
    h.lnnzj = local ? l1 : m.nnzj;
    h.lnnzh = local ? l2 : m.nnzh;
    // Block maps: workgroup b -> (pattern slot, tile).  Two orders are prepared per callback:
    //   [0] sequential  — patterns one after the other, each streaming its own contiguous COO range;
    //   [1] interleaved — patterns advance together in proportion to their tile counts, in runs of 128 workgroups: the
    //       stretch of x / y / columns one pattern just read is still in L2/MALL when the next one needs it.
    // Neither wins everywhere (MI355X, hess_coord!): LV N=1e7 0.143 -> 0.133 ms and rocket 0.087 -> 0.082 ms with [1],
    // but LV N=1e8 1.75 -> 1.86 ms and the cache-resident ACOPF 0.016 -> 0.019 ms; runs of <= 16 workgroups are always
    // slower (too many concurrent write streams).  So for callbacks that stream >= 128 MB from several patterns the
    // order is CHOSEN BY MEASUREMENT (exa_tune, persisted); everything else runs sequentially.
    for (int cb = 0; cb < CB_COUNT; cb++) {
        // dispatch units: one per active pattern, or (chained callbacks) one per group of co-indexed patterns; nb = how
        // many block-map entries (workgroups) a unit needs
        const bool chained = L.chain[cb] > 0, grouped = !L.groups[cb].empty();
        const size_t na = grouped ? L.groups[cb].size() : L.active[cb].size();
        std::vector<int64_t> nb(na);
        int64_t total = 0;
        double out_bytes = 0.0;
        auto per_point = [&](const Pattern &pt) {
            if (cb == CB_HESS || cb == CB_HSTRUCT) return pt.o2step;
            if (cb == CB_JAC || cb == CB_JSTRUCT) return pt.o1step;
            if (cb == CB_FUSED) return 1 + pt.o1step + pt.o2step;
            return 1;
        };
        for (size_t j = 0; j < na; j++) {
            if (grouped) {
                const int64_t tile = (int64_t)kBlock * L.ppt[cb];
                int64_t tiles = 0;
                for (int k : L.groups[cb][j]) {
                    const int64_t cnt = h.P[L.pat[k].hi] - h.P[L.pat[k].lo];
                    tiles = std::max(tiles, (cnt + tile - 1) / tile);
                    out_bytes += 8.0 * per_point(m.pats[k]) * (double)cnt;
                }
                if (chained) { h.P[L.gtiles[cb][j]] = tiles; nb[j] = (tiles + L.chain[cb] - 1) / L.chain[cb]; }
                else nb[j] = tiles;
            } else {
                const auto &pp = L.pat[L.active[cb][j]];
                const int64_t tile = (int64_t)kBlock * L.ppt[cb];
                const int64_t cnt = h.P[pp.hi] - h.P[pp.lo];
                nb[j] = (cnt + tile - 1) / tile;
                out_bytes += 8.0 * per_point(m.pats[L.active[cb][j]]) * (double)cnt;
            }
            total += nb[j];
        }
        if (cb == CB_GRAD && L.gbits >= 0 && h.on_device && h.world == 1 && h.dgbits.p) {
            // one-launch grad!: the zero tiles as one more unit (the objective tiles store, see gen_first_fn)
            nb.push_back((m.nvar + (int64_t)kBlock * 8 - 1) / ((int64_t)kBlock * 8));
            total += nb.back();
        }
        h.grid[cb] = total;
        if (cb == CB_FUSED) {
            // objective partial sums of the fused sweep: one per workgroup of an OBJECTIVE pattern, at a compact index
            int64_t nobj = 0;
            for (size_t j = 0; j < na; j++) {
                const int k = grouped ? L.groups[cb][j].front() : L.active[cb][j];       // objective patterns are groups of their own
                if (m.pats[k].kind == EXA_PAT_OBJ) { h.P[L.pat[k].ob] = nobj; nobj += nb[j]; }
            }
            h.fused_nobj = nobj;
        }
        auto build_units = [](const std::vector<int64_t> &nb, int64_t run_len) {
            const size_t na = nb.size();
            int64_t total = 0;
            for (int64_t v : nb) total += v;
            std::vector<int64_t> map, done(na, 0);
            map.reserve((size_t)total + 1);
            while ((int64_t)map.size() < total) {
                size_t best = na;   // sequential: first unfinished unit; interleaved: the one furthest behind
                for (size_t j = 0; j < na; j++) {
                    if (done[j] >= nb[j]) continue;
                    if (best == na || (run_len > 0 && (__int128)done[j] * nb[best] < (__int128)done[best] * nb[j])) best = j;
                }
                const int64_t run = run_len > 0 ? run_len : nb[best];
                for (int64_t r = 0; r < run && done[best] < nb[best]; r++) map.push_back(((int64_t)best << 40) | done[best]++);
            }
            return map;
        };
        auto build = [&](int64_t run_len) { return build_units(nb, run_len); };
        if (cb == CB_FUSED && L.gbits >= 0) {
            // one-launch exa_eval_all: unsharded models whose objective scatter was proven injective at the build (to_device)
            h.gridz = 0;
            h.P[L.gbits] = 0;
            if (h.on_device && h.world == 1 && h.dgbits.p) {
                h.P[L.gbits] = (int64_t)(uintptr_t)h.dgbits.p;
                std::vector<int64_t> nbz = nb;
                nbz.push_back((m.nvar + (int64_t)kBlock * 8 - 1) / ((int64_t)kBlock * 8));
                h.gridz = total + nbz.back();
                for (int k = 0; k < 2; k++) {
                    std::vector<int64_t> mp = build_units(nbz, k ? 128 : 0);
                    h.dmapz[k].ensure(sizeof(int64_t) * std::max<size_t>(mp.size(), 1));
                    if (!mp.empty()) HIPCHK(hipMemcpy(h.dmapz[k].p, mp.data(), sizeof(int64_t) * mp.size(), hipMemcpyHostToDevice));
                }
            }
        }
        if (cb == CB_FUSED) {
            // exa_eval_all: the same units + the tiles of the gathered gradient as one more unit (see exa_fused)
            h.gridg = 0;
            if (h.on_device && !L.pull.empty()) {
                const bool owner = L.active[CB_GRAD].empty();
                const int64_t vb = owner && h.world > 1 ? own_var_lo(h, h.rank) : 0, ve = owner && h.world > 1 ? own_var_lo(h, h.rank + 1) : m.nvar;
                const int64_t per = (int64_t)kBlock * L.pull_ppt;
                std::vector<int64_t> nbg = nb;
                nbg.push_back((ve - vb + per - 1) / per);
                h.gridg = total + nbg.back();
                {
                    int pv = -1;
                    h.orderg = tune_lookup(source_key(h.gen.source), tune_signature(h, "orderg"), &pv) && (pv == 0 || pv == 1) ? pv : -1;
                }
                for (int k = 0; k < 2; k++) {
                    std::vector<int64_t> mp = build_units(nbg, k ? 128 : 0);
                    h.dmapg[k].ensure(sizeof(int64_t) * std::max<size_t>(mp.size(), 1));
                    if (!mp.empty()) HIPCHK(hipMemcpy(h.dmapg[k].p, mp.data(), sizeof(int64_t) * mp.size(), hipMemcpyHostToDevice));
                }
            }
        }
        const bool tunable = cb == CB_HESS || cb == CB_HESSC || cb == CB_JAC || cb == CB_FUSED || cb == CB_CONS;
        const double stream_bytes = out_bytes + 8.0 * (double)(m.nvar + m.ncon) / h.world;
        if (cb == CB_HESS) h.hess_stream_bytes = stream_bytes;
        const bool two = h.on_device && total > 0 && na > 1 && tunable && out_bytes >= 128e6;
        // The order a model NOBODY TUNED runs (the reference's hess_coord! needs no tuning call, KA ext :526-537), decided here
        // from the plan: interleaved where (a) the two heaviest units read the same stretch of x as they advance — decidable when
        // every index expression is affine in a range column (pattern_var_range; LV: the objective and the constraint both walk
        // x[1..N]) — and (b) the call streams less than 1.5 GB, i.e. the stretch one unit just read is still in the Infinity Cache
        // when the other arrives (LV 1e7 hess_coord! 0.1453 -> 0.1359 ms, the fused sweep 0.197 -> 0.193; LV 1e8: sequential
        // 1.75 against 1.85 ms — the same 1.5 GB that switches hess_coord! to the chained kernel).  exa_tune still measures both
        // and its persisted decision wins over this default.
        auto default_order = [&]() -> int {
            if (!two || stream_bytes >= 1.5e9) return 0;
            struct U { double w; int64_t a, b; };
            std::vector<U> us;
            for (size_t j = 0; j < na; j++) {
                U u{0.0, INT64_MAX, INT64_MIN};
                for (int k : grouped ? L.groups[cb][j] : std::vector<int>{L.active[cb][j]}) {
                    const int64_t lo = h.P[L.pat[k].lo], hi = h.P[L.pat[k].hi];
                    if (hi <= lo) continue;
                    int64_t a = 0, b = 0;
                    if (!pattern_var_range(m.pats[k], lo, hi, &a, &b)) return 0;      // data-indexed: anywhere in x, nothing to align
                    if (a > b) continue;                                                // (a pattern without variables)
                    u.a = std::min(u.a, a); u.b = std::max(u.b, b);
                    u.w += 8.0 * per_point(m.pats[k]) * (double)(hi - lo);
                }
                if (u.w > 0 && u.a <= u.b) us.push_back(u);
            }
            if (us.size() < 2) return 0;
            std::partial_sort(us.begin(), us.begin() + 2, us.end(), [](const U &p, const U &q) { return p.w > q.w; });
            const double overlap = (double)(std::min(us[0].b, us[1].b) - std::max(us[0].a, us[1].a) + 1);
            const double shorter = (double)std::min(us[0].b - us[0].a, us[1].b - us[1].a) + 1.0;
            return overlap >= 0.5 * shorter ? 1 : 0;
        };
        if (cb == CB_FUSED) h.orderg_default = default_order() || !two ? 1 : 0;
        h.order[cb] = 0;
        h.norders[cb] = 1;
        h.P[L.blk[cb]] = 0;
        if (h.on_device && total > 0) {
            std::vector<int64_t> m0 = build(0);
            h.dmap[cb][0].ensure(sizeof(int64_t) * m0.size());
            HIPCHK(hipMemcpy(h.dmap[cb][0].p, m0.data(), sizeof(int64_t) * m0.size(), hipMemcpyHostToDevice));
            h.P[L.blk[cb]] = (int64_t)(uintptr_t)h.dmap[cb][0].p;
            if (two) {
                std::vector<int64_t> m1 = build(128);
                h.dmap[cb][1].ensure(sizeof(int64_t) * m1.size());
                HIPCHK(hipMemcpy(h.dmap[cb][1].p, m1.data(), sizeof(int64_t) * m1.size(), hipMemcpyHostToDevice));
                // both orders exist: exa_tune measures them; until then (and in later processes) the persisted
                // decision for this module / device / sizes applies, else the sequential order
                h.norders[cb] = 2;
                int pv = 0;
                if (!(tune_lookup(source_key(h.gen.source), tune_signature(h, "order" + std::to_string(cb)), &pv) && (pv == 0 || pv == 1))) pv = default_order();
                h.order[cb] = pv;
                h.P[L.blk[cb]] = (int64_t)(uintptr_t)h.dmap[cb][pv].p;
            }
        }
    }
    // hess_coord! kernel: the persisted measurement (exa_tune) if there is one, else by size — the chained kernel pays
    // for its pipelining with twice the registers: it wins where the call streams gigabytes through HBM (LV N = 3e7:
    // 0.472 against 0.529 ms, N = 1e8: 1.52 against 1.76) and loses where x, y and much of the output sit in the 256 MB
    // MALL or the arithmetic dominates (LV N = 1e7: 0.156 against 0.135 ms; rocket 0.108 / 0.086; ACOPF 0.034 / 0.018)
    // occupancy throttle of the hess_coord! launches: dynamic LDS nobody uses (static + dynamic <= 64 KB: no function attribute needed)
    // (chained kernels only; exa_tune decides between none / three / two workgroups per CU — the environment variable fixes it)
    const char *dl = getenv("EXAHIP_HESS_DYN_LDS");
    if (dl) h.hess_dyn_lds = (unsigned)std::min(49152, std::max(0, atoi(dl)));
    h.hess_variant = 0;
    if (L.chain[CB_HESSC] > 0) {
        const char *ce = getenv("EXAHIP_HESS_VARIANT");
        int pv = 0;
        if (ce) h.hess_variant = std::min(2, std::max(0, atoi(ce)));
        else if (tune_lookup(source_key(h.gen.source), tune_signature(h, "hessvariant"), &pv)) h.hess_variant = std::min(2, std::max(0, pv));
        else h.hess_variant = h.hess_stream_bytes >= 1.5e9;
        if (!dl) {
            int pd = 0;
            if (tune_lookup(source_key(h.gen.source), tune_signature(h, "hessdynlds"), &pd)) h.hess_dyn_lds = (unsigned)std::min(49152, std::max(0, pd));
            else h.hess_dyn_auto = !ce && h.hess_stream_bytes >= 1.5e9;        // (measured at LV 1e8: 1.70 -> 1.62 ms, profiles/r5_hess_occupancy.txt)
        }
    }
    // exa_hesscl stages, per wavefront, tile and stretch, ONE run of 64 + kStageHalo variables for the member clusters of the stretch: their
    // first variables (of THIS shard's first points) must lie within the halo of each other (ParamLayout::Stage)
    h.stage_ok = L.staged;
    if (L.staged)
        for (size_t g = 0; g < L.groups[CB_HESSC].size(); g++) {
            const auto &grp = L.groups[CB_HESSC][g];
            for (int k : grp) if (h.P[L.pat[k].hi] <= h.P[L.pat[k].lo]) h.stage_ok = false;
            for (int sidx = 0; sidx < L.gstretch[g]; sidx++) {
                int64_t bmin = INT64_MAX;
                for (int k : grp)
                    for (const auto &cl : L.stage[k].cl)
                        if (cl.stretch == sidx) bmin = std::min(bmin, h.P[L.stage[k].word] + h.P[L.pat[k].lo] + cl.cmin);
                for (int k : grp)
                    for (const auto &cl : L.stage[k].cl)
                        if (cl.stretch == sidx && h.P[L.stage[k].word] + h.P[L.pat[k].lo] + cl.cmax - bmin > kStageHalo) {
                            if (verbose() && h.stage_ok) fprintf(stderr, "[exahip] exa_hesscl not usable on this shard: group %zu stretch %d: pattern %d reaches %ld variables beyond the stretch's first\n", g, sidx, k, (long)(h.P[L.stage[k].word] + h.P[L.pat[k].lo] + cl.cmax - bmin));
                            h.stage_ok = false;
                        }
            }
        }
    // objective-only Hessian forms: block maps of the objective groups alone + the constraint patterns' slot ranges, merged
    h.gridobj[0] = h.gridobj[1] = 0;
    h.con_hess_ranges.clear();
    if (h.on_device && m.ncon > 0) {
        std::vector<int64_t> P2 = h.P;
        for (int which = 0; which < 2; which++) {
            const int cb = which ? CB_HPROD : CB_HESS;
            const bool grouped = !L.groups[cb].empty();
            const size_t na = grouped ? L.groups[cb].size() : L.active[cb].size();
            std::vector<int64_t> mp;
            const int64_t tile = (int64_t)kBlock * L.ppt[cb];
            for (size_t j = 0; j < na; j++) {
                const int k0 = grouped ? L.groups[cb][j].front() : L.active[cb][j];
                if (m.pats[k0].kind != EXA_PAT_OBJ) continue;
                int64_t tiles = 0;
                for (int k : grouped ? L.groups[cb][j] : std::vector<int>{k0}) tiles = std::max(tiles, (h.P[L.pat[k].hi] - h.P[L.pat[k].lo] + tile - 1) / tile);
                for (int64_t t = 0; t < tiles; t++) mp.push_back(((int64_t)j << 40) | t);
            }
            h.gridobj[which] = (int64_t)mp.size();
            h.dmapobj[which].ensure(8 * std::max<size_t>(mp.size(), 1));
            if (!mp.empty()) HIPCHK(hipMemcpy(h.dmapobj[which].p, mp.data(), 8 * mp.size(), hipMemcpyHostToDevice));
            P2[L.blk[cb]] = (int64_t)(uintptr_t)h.dmapobj[which].p;
        }
        for (size_t k = 0; k < m.pats.size(); k++) {
            const Pattern &p = m.pats[k];
            const int64_t cnt = (int64_t)p.o2step * (h.P[L.pat[k].hi] - h.P[L.pat[k].lo]);
            if (p.kind == EXA_PAT_OBJ || cnt <= 0) continue;
            const int64_t off = h.P[L.pat[k].o2] + (int64_t)p.o2step * h.P[L.pat[k].lo];
            if (!h.con_hess_ranges.empty() && h.con_hess_ranges.back().first + h.con_hess_ranges.back().second == off) h.con_hess_ranges.back().second += cnt;
            else h.con_hess_ranges.push_back({off, cnt});
        }
        h.dPobj.ensure(8 * P2.size());
        HIPCHK(hipMemcpy(h.dPobj.p, P2.data(), 8 * P2.size(), hipMemcpyHostToDevice));
    }
    if (h.on_device) {
        h.dP.ensure(sizeof(int64_t) * h.P.size());
        HIPCHK(hipMemcpy(h.dP.p, h.P.data(), sizeof(int64_t) * h.P.size(), hipMemcpyHostToDevice));
        // objective partials (+ the shard sums of the two-level fold, exa_obj_arrive: n rounded up to whole shards + up to 1024 of them) and the arrival
        // counters: [0] the single / top one, [1 + r] shard r's (up to 1024), 128 B apart, re-armed by the kernel
        h.dpart.ensure(sizeof(double) * (size_t)(std::max(h.grid[CB_OBJ], h.grid[CB_FUSED]) + 2 * 1024 + 1));
        if (!h.ddone.p) { h.ddone.ensure(1025 * 128); HIPCHK(hipMemset(h.ddone.p, 0, 1025 * 128)); }
    }
}

// ---- every compiled kernel is ASKED what it needs -----------------------------------------------------------------------
// Kernels of this library cooperate across lanes almost everywhere: the LDS-transposed COO epilogue, block sums, butterflies
// over shared scatter targets, the objective fold, LDS windows and planes.  A kernel that has outgrown the 256 architectural
// VGPRs — AGPRs (spill space in a kernel without MFMA) or scratch in use — has, under the default register allocator,
// returned wrong, run-to-run different sums: it read register lanes it never wrote (profiles/NOTES.md round 3, the standalone
// reproducer under tests/sweeps/canary/).  So after EVERY compilation — the model's module, its product windows, the
// windows / permuted stores of exa_compress — the code object's metadata is read for ALL of its kernels, and a module holding
// a kernel that does not fit is compiled again with the conservative allocator flags (exa_build.cpp safe_flags; another cache
// file, keyed by the flags).  Unreadable metadata counts as "does not fit".  The decision is remembered as the note "safe" of
// the module's key, so later processes (and plan-only handles, exa_compile, exahip.pack) compile the final object at once; a
// packed library hands the object over under the name <key>_safe and its consumer finds it without a compiler.
// exa_build_audit reports every kernel of every module with its numbers and the flags its module was built with.
bool prefer_safe(const std::string &source) {
    const std::string key = source_key(source);
    return !safe_flags().empty() && (note_lookup(key) == "safe" || (cache_has(key + "_safe") && !cache_has(key)));
}
bool all_fit(const CodeObject &co, std::vector<KernelInfo> &ks, bool *readable) {
    *readable = code_object_kernels(co.image, ks);
    if (!*readable) return false;
    for (const KernelInfo &k : ks) if (!k.fits()) return false;
    return true;
}
// `have`: an object of this very source fetched a moment ago (default flags or safe), reused instead of fetched again
CodeObject audited_code_object(Handle &h, const std::string &which, const std::string &source, bool memory_only_ok, const CodeObject *have) {
    CodeObject co = have && have->key == source_key(source) ? *have : get_code_object(source, memory_only_ok, prefer_safe(source));
    std::vector<KernelInfo> ks;
    bool readable = false;
    if (!all_fit(co, ks, &readable) && !co.safe && !safe_flags().empty()) {
        if (verbose()) {
            for (const KernelInfo &k : ks)
                if (!k.fits()) fprintf(stderr, "[exahip] %s: %d VGPRs, %d AGPRs, %d bytes of scratch per lane, %d VGPRs / %d SGPRs spilled: beyond the architectural registers\n", k.name.c_str(), k.vgpr, k.agpr, k.scratch, k.vgpr_spill, k.sgpr_spill);
            fprintf(stderr, "[exahip] module %s (%s) is compiled again with %s\n", co.key.c_str(), which.c_str(), safe_flags().c_str());
        }
        note_store(co.key, "safe", true);
        CodeObject c2 = get_code_object(source, memory_only_ok, true);
        c2.build_ms += co.build_ms;
        co = c2;
        (void)all_fit(co, ks, &readable);
    }
    Handle::Audit a;
    a.which = which; a.name = co.name; a.safe = co.safe; a.readable = readable; a.kernels = ks;
    bool replaced = false;
    for (auto &q : h.audits) if (q.which == which) { q = a; replaced = true; }
    if (!replaced) h.audits.push_back(a);
    return co;
}

// The model's first module, compiled or fetched.  Independently of the flags, scatter kernels (bodies of hundreds to
// thousands of SSA values) that do not fit the registers are generated again WITHOUT the loops around and inside their
// bodies: per-lane accumulators of shared targets carried across a 16-tile loop and the peeling loop of exa_scatter_add are
// what made them outgrow the register file (tests/test_random_expressions.py).  The generator avoids the loops for bodies it
// can see are huge (kHugeBody); here the COMPILED kernels are asked.  The decision is recorded as the note "loopfree" of the
// first module's key, so plan-only handles, exa_compile, exahip.pack, later processes and a packed library's consumer all
// arrive at the SAME final module directly.
bool scatter_kernels_spill(const CodeObject &co) {
    std::vector<KernelInfo> ks;
    if (!code_object_kernels(co.image, ks)) return true;        // unreadable metadata: assume the worst
    bool spills = false;
    for (const KernelInfo &k : ks) {
        if (k.name != "exa_grad" && k.name != "exa_jtprod" && k.name != "exa_hprod") continue;
        spills = spills || !k.fits();
        if (verbose()) fprintf(stderr, "[exahip] %s: %d VGPRs, %d AGPRs, %d bytes of scratch per lane, %d VGPRs / %d SGPRs spilled\n", k.name.c_str(), k.vgpr, k.agpr, k.scratch, k.vgpr_spill, k.sgpr_spill);
    }
    return spills;
}
CodeObject module_for(Handle &h, bool memory_only_ok) {
    CodeObject co = get_code_object(h.gen.source, memory_only_ok, prefer_safe(h.gen.source));
    double spent = 0.0;
    // ... and exa_hesscl (the LDS-staged chained kernel: a few registers more than exa_hessc per staged stretch) is dropped when it ALONE is
    // what outgrew the architectural registers (round 5's four-stretch kernel of the rocket: 278 against 254): the model then runs exa_hessc where it would have run
    // exa_hesscl, instead of having its whole module rebuilt with the conservative flags for a kernel it can do without
    bool regen = false;
    {
        std::vector<KernelInfo> ks;
        if (!h.nostage && h.gen.layout.staged && code_object_kernels(co.image, ks)) {
            bool cl_big = false, rest_big = false;
            for (const KernelInfo &k : ks) { if (k.name == "exa_hesscl") cl_big = !k.fits(); else if (k.name != "exa_grad" && k.name != "exa_jtprod" && k.name != "exa_hprod") rest_big = rest_big || !k.fits(); }
            if (cl_big && !rest_big) { h.nostage = true; regen = true; }
        }
    }
    if (!h.loopfree_scatter && scatter_kernels_spill(co)) { h.loopfree_scatter = true; regen = true; }
    if (regen) {
        h.first_key = co.key;
        h.first_note = std::string(h.loopfree_scatter ? "loopfree" : "") + (h.loopfree_scatter && h.nostage ? "+" : "") + (h.nostage ? "nostage" : "");
        note_store(co.key, h.first_note, true);
        h.gen = generate_module(*h.m, h.loopfree_scatter, h.nostage);
        spent = co.build_ms;
        // the kernels the regeneration does not touch are the same code in the new module: if one of THEM is over-sized the new module
        // will need the conservative flags too — note it now and save the compilation that would only find that out
        std::vector<KernelInfo> ks;
        bool others = !code_object_kernels(co.image, ks);
        for (const KernelInfo &k : ks) others = others || (!k.fits() && k.name != "exa_grad" && k.name != "exa_jtprod" && k.name != "exa_hprod" && !(h.nostage && k.name == "exa_hesscl"));
        if (others && !safe_flags().empty()) note_store(source_key(h.gen.source), "safe", true);
    }
    CodeObject fin = audited_code_object(h, "model", h.gen.source, memory_only_ok, &co);
    fin.build_ms += spent;
    return fin;
}

void to_device(Handle &h) {
    Model &m = *h.m;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        throw HipError("no HIP device available (libexahip has no CPU fallback): " + std::string(hipGetErrorString(e)));
    CodeObject co = module_for(h, true);
    std::vector<char> &image = co.image;
    h.hsaco_path = co.path; h.build_how = co.how; h.build_ms = co.build_ms; h.co_name = co.name;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess) h.device = dev;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess)
            h.devname = std::string(prop.gcnArchName) + "/" + std::to_string(prop.multiProcessorCount);
    }
    h.on_device = true;   // from here on the destructor releases whatever was acquired
    HIPCHK(hipModuleLoadData(&h.module, image.data()));
    auto fn = [&](const char *name) { hipFunction_t f; HIPCHK(hipModuleGetFunction(&f, h.module, name)); return f; };
    h.f_obj = fn("exa_obj"); h.f_zero = fn("exa_zero"); h.f_grad = fn("exa_grad"); h.f_cons = fn("exa_cons");
    h.f_auggather = fn("exa_aug_gather"); h.f_gradpull = fn("exa_grad_pull");
    h.f_auglong = fn("exa_aug_long"); h.f_augfold = fn("exa_aug_fold");
    h.f_fused = fn("exa_fused");
    h.f_jprod = fn("exa_jprod"); h.f_jtprod = fn("exa_jtprod"); h.f_hprod = fn("exa_hprod"); h.f_jac = fn("exa_jac"); h.f_hess = fn("exa_hess");
    if (h.gen.layout.chain[CB_HESSC] > 0) h.f_hessc = fn("exa_hessc");
    if (h.gen.layout.chain[CB_HESSC] > 0 && h.gen.layout.staged) h.f_hesscl = fn("exa_hesscl");
    h.f_cons1 = fn("exa_cons1");
    h.f_jacl = fn("exa_jacl"); h.f_consl = fn("exa_consl");
    if (const char *tl = getenv("EXAHIP_TILE_LOOP")) h.tile_loop = std::min(64, std::max(0, atoi(tl)));
    h.f_gradv = fn("exa_gradv"); h.f_gstruct = fn("exa_gstruct");
    if (m.aug_linear || m.nconaug == 0) h.f_jprod1 = fn("exa_jprod1");
    h.f_js32 = fn("exa_jstruct32"); h.f_js64 = fn("exa_jstruct64"); h.f_hs32 = fn("exa_hstruct32"); h.f_hs64 = fn("exa_hstruct64");
    h.colslot.resize(m.pats.size());
    // one-launch exa_eval_all (ParamLayout::gbits): is the objective's in-sweep scatter injective on THIS data?  (needs the host columns)
    if (h.gen.layout.gbits >= 0 && m.nvar > 0) {
        std::vector<uint64_t> bits;
        int64_t pts = 0;
        for (int k : h.gen.layout.active[CB_GRAD]) pts += m.pats[(size_t)k].n;
        // (a host pass over the objective's data points: not worth it beyond a few 1e7 of them — such a model keeps the atomics)
        if (pts <= 50000000 && scatter_bitmap(m, h.gen.layout.active[CB_GRAD], bits)) {
            bits.insert(bits.begin(), (uint64_t)m.nvar);        // [nvar, bitmap...]: the zero tiles of exa_grad read their bound from the buffer
            h.dgbits.ensure(8 * bits.size());
            HIPCHK(hipMemcpy(h.dgbits.p, bits.data(), 8 * bits.size(), hipMemcpyHostToDevice));
        }
    }
    for (size_t k = 0; k < m.pats.size(); k++) {
        Pattern &p = m.pats[k];
        h.colslot[k].assign(p.cols.size(), -1);
        for (size_t c = 0; c < p.cols.size(); c++) {
            Column &col = p.cols[c];
            if (col.type == EXA_COL_RANGE) continue;
            if (col.alias_pat >= 0) {      // a copy of a column that is already resident (exa_plan.cpp)
                h.colslot[k][c] = h.colslot[col.alias_pat][col.alias_col];
                std::vector<int64_t>().swap(col.idata);
                std::vector<double>().swap(col.fdata);
                continue;
            }
            DevBuf b;
            b.ensure(8 * (size_t)p.n);
            const void *src = col.type == EXA_COL_I64 ? (const void *)col.idata.data() : (const void *)col.fdata.data();
            if (p.n) HIPCHK(hipMemcpy(b.p, src, 8 * (size_t)p.n, hipMemcpyHostToDevice));
            h.colslot[k][c] = (int)h.dcols.size();
            h.dcols.push_back(b);
            // the host copy is no longer needed once resident in HBM
            std::vector<int64_t>().swap(col.idata);
            std::vector<double>().swap(col.fdata);
        }
    }
    h.dtheta.ensure(sizeof(double) * (size_t)(m.npar + 1));
    if (m.npar) HIPCHK(hipMemcpy(h.dtheta.p, m.theta.data(), sizeof(double) * (size_t)m.npar, hipMemcpyHostToDevice));
    h.dobj.ensure(sizeof(double));
    if (m.nconaug) {
        auto up = [&](DevBuf &b, const std::vector<int64_t> &v) {
            b.ensure(8 * v.size());
            HIPCHK(hipMemcpy(b.p, v.data(), 8 * v.size(), hipMemcpyHostToDevice));
        };
        up(h.daugrows, m.aug_rows); up(h.daugptr, m.aug_ptr); up(h.daugperm, m.aug_perm);
        h.daugbuf.ensure(8 * (size_t)m.nconaug);
        std::vector<int64_t> longs;
        int64_t maxlen = 0;
        for (size_t t = 0; t + 1 < m.aug_ptr.size(); t++) {
            const int64_t len = m.aug_ptr[t + 1] - m.aug_ptr[t];
            if (len > 512) { longs.push_back((int64_t)t); maxlen = std::max(maxlen, len); }     // EXA_AUG_LONG
        }
        h.aug_nlong = (int64_t)longs.size();
        h.aug_chunks = (maxlen + 8191) / 8192;                                                    // EXA_AUG_CHUNK
        if (h.aug_nlong) { up(h.dauglong, longs); h.daugpartial.ensure(8 * (size_t)(h.aug_nlong * h.aug_chunks)); }
        // one-launch cons_nln! (exa_cons1): per constraint row, its terms as (pattern, data point) in insertion order
        if (h.aug_nlong == 0) {
            std::vector<int64_t> rowptr((size_t)m.ncon + 1, 0), src((size_t)m.nconaug);
            for (size_t t = 0; t < m.aug_rows.size(); t++) rowptr[(size_t)m.aug_rows[t] + 1] = m.aug_ptr[t + 1] - m.aug_ptr[t];
            for (int64_t r = 0; r < m.ncon; r++) rowptr[(size_t)r + 1] += rowptr[(size_t)r];
            std::vector<std::pair<int64_t, int>> starts;          // (first buffer entry, pattern) of the augmentation patterns
            for (size_t k = 0; k < m.pats.size(); k++) if (m.pats[k].kind == EXA_PAT_CONAUG && m.pats[k].n > 0) starts.push_back({m.pats[k].oa, (int)k});
            std::sort(starts.begin(), starts.end());
            std::vector<double> coef;
            if (m.aug_linear) coef.resize((size_t)m.nconaug);
            for (int64_t j = 0; j < m.nconaug; j++) {
                const int64_t q = m.aug_perm[(size_t)j];
                if (m.aug_linear) { src[(size_t)j] = m.aug_var[(size_t)q]; coef[(size_t)j] = m.aug_coef[(size_t)q]; continue; }
                auto it = std::upper_bound(starts.begin(), starts.end(), std::make_pair(q, INT32_MAX));
                const auto &st = *(it - 1);
                src[(size_t)j] = ((int64_t)st.second << 40) | (q - st.first);
            }
            up(h.daugcsr, rowptr); up(h.daugsrc, src);
            h.daugcoef.ensure(8 * std::max<size_t>(coef.size(), 1));
            if (!coef.empty()) HIPCHK(hipMemcpy(h.daugcoef.p, coef.data(), 8 * coef.size(), hipMemcpyHostToDevice));
            h.cons1 = true;
        }
    }
    if (m.nconaug == 0) {
        // (default: where a fused group of exa_cons1 really fuses something; EXAHIP_CONS_FUSED=0 / 1 overrides)
        bool fuses = false;
        for (const auto &g : h.gen.layout.groups[CB_CONS1]) fuses = fuses || g.size() > 1;
        const char *cf = getenv("EXAHIP_CONS_FUSED");
        h.cons_fused = cf && *cf ? *cf == '1' : fuses;
    }
    HIPCHK(hipEventCreate(&h.ev0));
    HIPCHK(hipEventCreate(&h.ev1));
    fill_params(h);
}

void launch(Handle &h, hipFunction_t f, int64_t grid, unsigned block, void **args, unsigned dyn_lds) {
    if (grid <= 0) return;
    if (grid > 0x7fffffffLL) throw std::runtime_error("grid too large");
    HIPCHK(hipModuleLaunchKernel(f, (unsigned)grid, 1, 1, block, 1, 1, dyn_lds, h.stream, args, nullptr));
}

// Tiles per workgroup of the looped first-order kernels (exa_consl / exa_jacl), 0 = the one-tile kernel.  A workgroup of the one-tile kernels lives
// ~2 us and its wavefront slots then sit empty for most of another microsecond until the next workgroup arrives (LV 1e7 exa_cons: 6.1 of 8
// wavefronts resident on average, profiles/r5_instruction_mix.txt); a loop over 4 / 8 block-map entries amortises that, the compiler keeps the
// literal coefficients of exp / sincos in SGPRs across tiles, and — round 6 — the loop is software-pipelined (the next tile's loads issued before
// this tile is evaluated, gen_dispatch_looped).  One model per setting alternating in one process (tools/tile_loop_ab.py, profiles/r6_tile_loop_ab.txt):
// LV 1e8 cons_nln! 0.575 -> 0.468 ms, jac_coord! 0.790 -> 0.578 (round 5's loop without the pipelining: 0.513 / 0.709); 3e7 0.133 -> 0.131 and
// 0.180 -> 0.162; 1e7 0.0488 -> 0.0496 (cons_nln!: the one-tile kernel stays there) and 0.0615 -> 0.0569; bitwise equal.  Only where the block map is long enough to keep every CU busy
// with the longer workgroups.
int tile_loop_ppt(Handle &h, int cb) {
    hipFunction_t f = cb == CB_JAC ? h.f_jacl : h.f_consl;
    if (!f || h.gen.layout.ppt[cb] != 1 || h.tile_loop == 0 || h.tile_loop == 1) return 0;
    if (h.tile_loop > 1) return h.tile_loop;
    // (cons_nln! at LV 1e7 — 39 063 entries —: one tile 0.0474-0.0488 ms, the loop 0.0494-0.0502: it pays from ~1e5 entries on; jac_coord!,
    // whose tiles also flush COO slots, from 32 768)
    if (cb == CB_JAC) return h.grid[cb] >= 32768 ? 8 : 0;
    return h.grid[cb] >= 65536 ? 8 : 0;
}
// zero-fill of n doubles on the model's stream (exa_zero)
void zero_fill(Handle &h, void *p, int64_t n) {
    if (n <= 0) return;
    void *a[] = {&p, &n};
    launch(h, h.f_zero, (n + 4 * kBlock - 1) / (4 * kBlock), kBlock, a);
}

// second stage of cons_nln! / jprod_nln! / the fused sweep: add the buffered augmentation terms to their rows
void aug_gather(Handle &h, void *buf, double *c) {
    const void *rows = h.daugrows.p, *ptr = h.daugptr.p, *perm = h.daugperm.p;
    int64_t nrows = (int64_t)h.m->aug_rows.size();
    void *a3[] = {&rows, &ptr, &perm, &buf, &c, &nrows};
    launch(h, h.f_auggather, (nrows + kBlock - 1) / kBlock, kBlock, a3);
    if (h.aug_nlong == 0) return;
    const void *list = h.dauglong.p;
    void *partial = h.daugpartial.p;
    int chunks = (int)h.aug_chunks;
    void *a4[] = {&list, &ptr, &perm, &buf, &partial, &chunks};
    HIPCHK(hipModuleLaunchKernel(h.f_auglong, (unsigned)h.aug_nlong, (unsigned)chunks, 1, kBlock, 1, 1, 0, h.stream, a4, nullptr));
    int64_t nlong = h.aug_nlong;
    void *a5[] = {&list, &rows, &partial, &chunks, &c, &nlong};
    launch(h, h.f_augfold, (nlong + kBlock - 1) / kBlock, kBlock, a5);
}

// ---- callbacks (device pointers, asynchronous) ------------------------------------------------------------
void do_obj(Handle &h, const double *x, double *out_dev) {
    const void *P = h.dP.p, *th = h.dtheta.p;
    void *part = h.dpart.p;
    int64_t n = h.grid[CB_OBJ];
    if (n == 0) { HIPCHK(hipMemsetAsync(out_dev, 0, sizeof(double), h.stream)); allreduce(h, out_dev, 1); return; }
    // ONE launch at any size: the workgroup of exa_obj that finishes last folds the partial sums (beyond 512 workgroups through 32
    // sharded arrival counters and 32 shard sums, exa_obj_arrive)
    void *done = h.ddone.p;
    void *a1[] = {&P, &x, &th, &part, &done, &out_dev};
    launch(h, h.f_obj, n, kBlock, a1);
    allreduce(h, out_dev, 1);
}
// grad!.  Gathered (range-affine) objective patterns are evaluated per VARIABLE, so a sharded model shards them by variable
// range: rank r computes complete values for the variables [nvar*r/G, nvar*(r+1)/G) from whatever data points touch them
// (it needs the halo of x, exa_shard_var_range) — a disjoint slice, no zero-fill, no collective; all-gather-v only to
// make the vector whole on every rank.  Patterns that scatter through a data index add the partial sums of the shard's
// own data points on top (the gathered part then holds zeros outside the owned slice) and the vector is all-reduced.
void do_grad(Handle &h, const double *x, double *g) {
    const void *P = h.dP.p, *th = h.dtheta.p;
    const int64_t nvar = h.m->nvar;
    const bool scatter = !h.gen.layout.active[CB_GRAD].empty(), pull = !h.gen.layout.pull.empty();     // (the MODEL's patterns, not this shard's)
    const bool owner = pull && !scatter;
    const int gb = h.gen.layout.gbits;
    // (an objective whose scatter the build proved injective: its tiles store, the zero tiles ride in the same launch — no zero-fill launch)
    const bool direct = scatter && !pull && gb >= 0 && h.world == 1 && h.P[(size_t)gb] != 0;
    if (!direct) {
        // gathered patterns: plain coalesced store of every g[v] (zero where nothing contributes).  Without gathered patterns
        // the same kernel is the zero-fill under the atomics: a plain launch is cheaper than hipMemsetAsync (ACOPF grad!
        // 0.018 -> 0.009 ms, profiles/NOTES.md)
        int64_t own_lo = h.world > 1 ? own_var_lo(h, h.rank) : 0, own_hi = h.world > 1 ? own_var_lo(h, h.rank + 1) : nvar;
        int64_t vb = owner ? own_lo : 0, ve = owner ? own_hi : nvar;
        void *a0[] = {&P, &x, &th, &g, &vb, &ve, &own_lo, &own_hi};
        const int64_t per = (int64_t)kBlock * h.gen.layout.pull_ppt;
        launch(h, h.f_gradpull, (ve - vb + per - 1) / per, kBlock, a0);
    }
    void *a[] = {&P, &x, &th, &g};
    launch(h, h.f_grad, h.grid[CB_GRAD], kBlock, a);   // scattered patterns: FP64 hardware atomics on top
    if (owner) allgatherv(h, g, var_pieces(h));
    else if (scatter || pull) allreduce(h, g, nvar);
}
// grad! by sorted gather: exa_gradv writes the gradient COO, every variable's slots are added in slot order (long lists
// cooperatively).  Lists are built at the first use; a sharded model keeps the atomics (its ranks write disjoint parts of
// one COO whose other parts nobody fills).
bool grad_sorted_possible(const Handle &h) { return h.world == 1 && h.m->nnzg > 0 && h.m->nnzg < 0xffffffffLL && h.grid[CB_OBJ] > 0; }
void grad_setup(Handle &h) {
    if (h.grad_ready) return;
    const Model &m = *h.m;
    h.gbuf.ensure(8 * (size_t)m.nnzg);
    DevBuf cols;
    try {
        cols.ensure(8 * (size_t)m.nnzg);
        const void *P = h.dP.p;
        void *cp = cols.p;
        void *a[] = {&P, &cp};
        launch(h, h.f_gstruct, h.grid[CB_OBJ], kBlock, a);
        build_sorted_index(h.gbyvar, (const int64_t *)cols.p, m.nnzg, m.nvar, h.stream);
        attach_unit(h.gbyvar, h.stream);
        const double one = 1.0;
        h.gone.ensure(8);
        HIPCHK(hipMemcpy(h.gone.p, &one, 8, hipMemcpyHostToDevice));
        HIPCHK(hipStreamSynchronize(h.stream));
    } catch (...) { cols.release(); throw; }
    cols.release();
    h.grad_ready = true;
}
void do_grad_sorted(Handle &h, const double *x, double *g) {
    const void *P = h.dP.p, *th = h.dtheta.p;
    void *gb = h.gbuf.p;
    void *a[] = {&P, &x, &th, &gb};
    launch(h, h.f_gradv, h.grid[CB_OBJ], kBlock, a);
    spmv_gather(h.gbyvar, (const double *)h.gbuf.p, nullptr, nullptr, nullptr, false, (const double *)h.gone.p, g, false, h.stream);
}
// A callback never allocates, sorts or synchronises: the sorted lists a persisted / explicit decision needs are built
// eagerly (eager_setup at model build and after a reshard; exa_set_*_mode; exa_tune).  Should a call still find them
// missing while its stream is being CAPTURED (hipStreamBeginCapture), it runs the implementation that needs none.
bool capturing(const Handle &h) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(h.stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}
int resolve_grad_mode(Handle &h) {
    if (h.grad_mode < 0) {
        int v = 0;
        h.grad_mode = tune_lookup(source_key(h.gen.source), tune_signature(h, "grad"), &v) && v == 1 ? 1 : 0;
    }
    if (h.grad_mode == 1 && !grad_sorted_possible(h)) return 0;
    if (h.grad_mode == 1 && !h.grad_ready) { if (capturing(h)) return 0; grad_setup(h); }
    return h.grad_mode;
}
void run_grad(Handle &h, const double *x, double *g) {
    if (resolve_grad_mode(h) == 1) { do_grad_sorted(h, x, g); allreduce(h, g, h.m->nvar); }
    else do_grad(h, x, g);
}
// cons_nln!.  A base row belongs to one data point, so a sharded model's ranks own disjoint row slices; with exa_cons1 the
// thread of a row evaluates the row's augmentation terms itself, from whatever data points they come — owner computes: every
// rank's rows are COMPLETE, nothing is zero-filled, nothing is summed (all-gather-v only to make c whole on every rank).  Only
// the two-stage path (rows collecting > 512 terms) still forms partial sums over the shard's terms and all-reduces c.
bool rows_owner_complete(const Handle &h) { return h.m->nconaug == 0 || h.cons1; }
void do_cons(Handle &h, const double *x, double *c) {
    if (h.m->ncon == 0) return;
    void *buf = h.daugbuf.p;
    const bool owner = rows_owner_complete(h);
    if (h.world > 1 && !owner) {
        HIPCHK(hipMemsetAsync(c, 0, sizeof(double) * (size_t)h.m->ncon, h.stream));
        if (h.m->nconaug) HIPCHK(hipMemsetAsync(buf, 0, sizeof(double) * (size_t)h.m->nconaug, h.stream));
    }
    const void *P = h.dP.p, *th = h.dtheta.p;
    if (h.cons1 || (h.cons_fused && h.m->nconaug == 0)) {
        // ONE launch: every base row's thread evaluates the row's augmentation terms itself (exa_cons1)
        const void *ptr = h.daugcsr.p, *src = h.daugsrc.p, *coef = h.daugcoef.p;
        void *a1[] = {&P, &x, &th, &c, &ptr, &src, &coef};
        launch(h, h.f_cons1, h.grid[CB_CONS1], kBlock, a1);
        allgatherv(h, c, row_pieces(h));
        return;
    }
    // base rows (plain stores into c) and augmentation terms (into the value buffer, coalesced)
    if (const int ppt = tile_loop_ppt(h, CB_CONS)) {
        int64_t nent = h.grid[CB_CONS];
        int pp = ppt;
        void *al[] = {&P, &x, &th, &c, &buf, &nent, &pp};
        launch(h, h.f_consl, (nent + ppt - 1) / ppt, kBlock, al);
    } else {
        void *a[] = {&P, &x, &th, &c, &buf};
        launch(h, h.f_cons, h.grid[CB_CONS], kBlock, a);
    }
    if (h.m->nconaug) aug_gather(h, buf, c);       // then one deterministic gather per target row
    if (owner) allgatherv(h, c, row_pieces(h));
    else allreduce(h, c, h.m->ncon);
}
void do_jac(Handle &h, const double *x, double *v) {
    const void *P = h.dP.p, *th = h.dtheta.p;
    if (const int ppt = tile_loop_ppt(h, CB_JAC)) {
        int64_t nent = h.grid[CB_JAC];
        int pp = ppt;
        void *al[] = {&P, &x, &th, &v, &nent, &pp};
        launch(h, h.f_jacl, (nent + ppt - 1) / ppt, kBlock, al);
        return;
    }
    void *a[] = {&P, &x, &th, &v};
    launch(h, h.f_jac, h.grid[CB_JAC], kBlock, a);
}
// The objective-only forms hess_coord!(m, x, hess; obj_weight) / hprod!(m, x, v, Hv; obj_weight) (nlp.jl:1906-1915, :1942-1952):
// y == NULL.  Like the reference, the constraint patterns are NOT evaluated: the objective groups are launched alone (their own
// block maps, fill_params) and the constraint slots receive exact zeros — a constraint whose second derivative is Inf / NaN at x
// cannot leak 0 * Inf = NaN into the result.  (Only the sorted-gather product, which evaluates the COO through do_hess, and the
// fused sweeps, which always have y, come through here with constraints present.)
void do_hess(Handle &h, const double *x, const double *y, double sigma, double *v) {
    const void *P = h.dP.p, *th = h.dtheta.p;
    if (!y && h.m->ncon > 0) {
        for (const auto &r : h.con_hess_ranges) zero_fill(h, v + r.first, r.second);
        const void *Po = h.dPobj.p;
        void *a[] = {&Po, &x, &y, &th, &v, &sigma};
        launch(h, h.f_hess, h.gridobj[0], kBlock, a);
        return;
    }
    if (h.hess_variant >= 1 && h.f_hessc) {
        if (h.hess_dyn_auto) { h.hess_dyn_auto = false; h.hess_dyn_lds = hess_throttle_bytes(h, h.hess_variant, 3); }
        void *sink = h.dsink.p;
        void *a[] = {&P, &x, &y, &th, &v, &sigma, &sink};
        // (a throttle from the environment or from a persisted decision is checked once against THIS kernel's static LDS: static + dynamic
        // beyond 64 KB would fail the launch — then no throttle)
        if (h.hess_dyn_ok_for != h.hess_dyn_lds || h.hess_dyn_ok_variant != h.hess_variant) {
            h.hess_dyn_ok = hess_throttle_clamp(h, h.hess_variant, h.hess_dyn_lds);
            h.hess_dyn_ok_for = h.hess_dyn_lds; h.hess_dyn_ok_variant = h.hess_variant;
        }
        launch(h, h.hess_variant == 1 && h.f_hesscl && h.stage_ok ? h.f_hesscl : h.f_hessc, h.grid[CB_HESSC], kBlock, a, h.hess_dyn_ok);
        return;
    }
    void *a[] = {&P, &x, &y, &th, &v, &sigma};
    launch(h, h.f_hess, h.grid[CB_HESS], kBlock, a);
}
// fused obj + cons_nln! + jac_coord! + hess_coord! at one x
// gout: null, or the gradient vector the objective patterns that are NOT gathered per variable add their first partials
// to (exa_eval_all; the caller has zero-filled it or run exa_grad_pull into it)
void do_fused(Handle &h, const double *x, const double *y, double sigma, double *obj_dev, double *c, double *jv, double *hv, double *gout,
              bool with_pull, bool with_zero) {
    const void *P = h.dP.p, *th = h.dtheta.p;
    void *part = h.dpart.p, *buf = h.daugbuf.p;
    // linear augmentation terms are added inside the sweep through the row lists of exa_cons1 when those exist: the rows a
    // rank owns are then complete (as in do_cons); otherwise partial sums over the shard's terms + all-reduce
    const bool inline_aug = h.cons1 && h.m->aug_linear && h.m->nconaug > 0;
    const bool owner = h.m->nconaug == 0 || inline_aug;
    if (h.world > 1 && !owner) {
        if (h.m->ncon) HIPCHK(hipMemsetAsync(c, 0, sizeof(double) * (size_t)h.m->ncon, h.stream));
        if (h.m->nconaug) HIPCHK(hipMemsetAsync(buf, 0, sizeof(double) * (size_t)h.m->nconaug, h.stream));
    }
    int64_t n = h.grid[CB_FUSED];
    const void *ap = inline_aug ? h.daugcsr.p : nullptr, *as = inline_aug ? h.daugsrc.p : nullptr, *ac = inline_aug ? h.daugcoef.p : nullptr;
    // with_pull: the gathered gradient's tiles ride in this launch as one more unit of the block map (exa_eval_all)
    const void *bmap = nullptr;
    int64_t vb = 0, ve = 0, own_lo = 0, own_hi = 0;
    if (with_pull && h.gridg > 0) {
        const bool owner = h.gen.layout.active[CB_GRAD].empty();
        own_lo = h.world > 1 ? own_var_lo(h, h.rank) : 0; own_hi = h.world > 1 ? own_var_lo(h, h.rank + 1) : h.m->nvar;
        vb = owner ? own_lo : 0; ve = owner ? own_hi : h.m->nvar;
        bmap = h.dmapg[(h.orderg >= 0 ? h.orderg : h.orderg_default) ? 1 : 0].p;
        n = h.gridg;
    }
    if (with_zero && h.gridz > 0) {      // the zero tiles of the injective in-sweep gradient ride as one more unit (no order between them and anything)
        ve = h.m->nvar;
        bmap = h.dmapz[h.order[CB_FUSED] ? 1 : 0].p;
        n = h.gridz;
    }
    // objective partial sums: folded by the objective workgroup that arrives last (as in do_obj)
    int64_t nobj = h.fused_nobj;
    void *done = h.ddone.p;
    void *a[] = {&P, &x, &y, &th, &part, &c, &buf, &jv, &hv, &sigma, &ap, &as, &ac, &gout, &bmap, &vb, &ve, &own_lo, &own_hi, &done, &nobj, &obj_dev};
    launch(h, h.f_fused, n, kBlock, a);
    if (!(n > 0 && nobj > 0)) HIPCHK(hipMemsetAsync(obj_dev, 0, sizeof(double), h.stream));
    if (h.m->nconaug && !inline_aug) aug_gather(h, buf, c);
    allreduce(h, obj_dev, 1);
    if (h.m->ncon) { if (owner) allgatherv(h, c, row_pieces(h)); else allreduce(h, c, h.m->ncon); }
}
// All five callbacks of a solver iteration at one x (the call pattern of
// test/NLPModelsIpoptLite.jl/src/NLPModelsIpoptLite.jl:28-40): obj, grad!, cons_nln!, jac_coord!, hess_coord!.  The fused
// sweep's objective patterns hold their first partials already: those that scatter through a data index add them to g
// inside the sweep (no exa_grad launch, no second evaluation); range-affine ones are gathered per variable by exa_grad_pull
// BEFORE the sweep (it also provides the zeros; x is then warm in the MALL for the sweep, whose 1.2 GB of output would
// otherwise evict it before a grad! that ran afterwards).
void do_eval_all(Handle &h, const double *x, const double *y, double sigma, double *obj_dev, double *g, double *c, double *jv, double *hv) {
    const ParamLayout &L = h.gen.layout;
    const int64_t nvar = h.m->nvar;
    const bool scatter = !L.active[CB_GRAD].empty(), pull = !L.pull.empty();
    if (resolve_grad_mode(h) == 1) {       // grad! by sorted gather (explicit, or exa_tune's persisted decision — the same choice exa_grad makes): separately
        do_grad_sorted(h, x, g);
        allreduce(h, g, nvar);
        do_fused(h, x, y, sigma, obj_dev, c, jv, hv);
        return;
    }
    const bool owner = pull && !scatter;
    if (pull && scatter) {
        // both kinds of objective pattern: the gathered part first (it provides the zeros the atomics of the sweep add to)
        const void *P = h.dP.p, *th = h.dtheta.p;
        int64_t own_lo = h.world > 1 ? own_var_lo(h, h.rank) : 0, own_hi = h.world > 1 ? own_var_lo(h, h.rank + 1) : nvar;
        int64_t vb = 0, ve = nvar;
        void *a0[] = {&P, &x, &th, &g, &vb, &ve, &own_lo, &own_hi};
        const int64_t per = (int64_t)kBlock * L.pull_ppt;
        launch(h, h.f_gradpull, (ve - vb + per - 1) / per, kBlock, a0);
    }
    // scattered patterns only: the sweep adds their first partials to a zero-filled g by atomics — or, where the build proved the scatter
    // injective (ParamLayout::gbits), STORES them, and the zero tiles of the same launch cover the variables no objective point names
    const bool direct = !pull && scatter && L.gbits >= 0 && h.world == 1 && h.gridz > 0 && h.P[(size_t)L.gbits] != 0;
    if (!pull && !direct) zero_fill(h, g, nvar);
    // (only gathered patterns: their tiles ride inside the sweep's launch)
    do_fused(h, x, y, sigma, obj_dev, c, jv, hv, g, /*with_pull=*/pull && !scatter, /*with_zero=*/direct);
    if (owner) allgatherv(h, g, var_pieces(h));
    else if (scatter || pull) allreduce(h, g, nvar);
}
// matrix-free products (jprod_nln! / jtprod_nln! / hprod!, nlp.jl:1882-1978)
void do_jprod(Handle &h, const double *x, const double *v, double *Jv) {
    if (h.m->ncon == 0) return;
    void *buf = h.daugbuf.p;
    const bool one = h.f_jprod1 && (h.cons1 || h.m->nconaug == 0);      // rows complete on their owner, as in do_cons
    const bool owner = one || h.m->nconaug == 0;
    if (h.world > 1 && !owner) {
        zero_fill(h, Jv, h.m->ncon);
        zero_fill(h, buf, h.m->nconaug);
    }
    const void *P = h.dP.p, *th = h.dtheta.p;
    if (one) {
        // ONE launch (fused groups; augmentation terms c * x[k] contribute c * v[k] straight from the row lists)
        const void *ptr = h.daugcsr.p, *src = h.daugsrc.p, *coef = h.daugcoef.p;
        void *a1[] = {&P, &x, &th, &v, &Jv, &ptr, &src, &coef};
        launch(h, h.f_jprod1, h.grid[CB_CONS1], kBlock, a1);
        allgatherv(h, Jv, row_pieces(h));
        return;
    }
    void *a[] = {&P, &x, &th, &v, &Jv, &buf};
    launch(h, h.f_jprod, h.grid[CB_JPROD], kBlock, a);
    if (h.m->nconaug) aug_gather(h, buf, Jv);
    if (owner) allgatherv(h, Jv, row_pieces(h));
    else allreduce(h, Jv, h.m->ncon);
}
void do_jtprod(Handle &h, const double *x, const double *v, double *Jtv) {
    zero_fill(h, Jtv, h.m->nvar);
    const void *P = h.dP.p, *th = h.dtheta.p;
    void *a[] = {&P, &x, &th, &v, &Jtv};
    launch(h, h.f_jtprod, h.grid[CB_JTPROD], kBlock, a);
}
void do_hprod(Handle &h, const double *x, const double *y, const double *v, double sigma, double *Hv) {
    zero_fill(h, Hv, h.m->nvar);
    const bool obj_only = !y && h.m->ncon > 0;         // objective groups alone (see do_hess)
    const void *P = obj_only ? h.dPobj.p : h.dP.p, *th = h.dtheta.p;
    void *a[] = {&P, &x, &y, &th, &v, &Hv, &sigma};
    launch(h, h.f_hprod, obj_only ? h.gridobj[1] : h.grid[CB_HPROD], kBlock, a);
}
void do_struct(Handle &h, bool hess, bool wide, void *rows, void *cols) {
    const void *P = h.dP.p;
    void *a[] = {&P, &rows, &cols};
    hipFunction_t f = hess ? (wide ? h.f_hs64 : h.f_hs32) : (wide ? h.f_js64 : h.f_js32);
    launch(h, f, h.grid[hess ? CB_HSTRUCT : CB_JSTRUCT], kBlock, a);
}

// host-pointer variants of a sharded model: entries this rank does not own come back as zeros (owner pieces + zeros add
// up across ranks like partial sums do)
void zero_if_sharded(Handle &h, void *p, size_t bytes) { if (h.world > 1 && bytes) HIPCHK(hipMemsetAsync(p, 0, bytes, h.stream)); }
void h2d(Handle &h, DevBuf &b, const void *src, size_t bytes) {
    b.ensure(bytes);
    if (bytes) HIPCHK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, h.stream));
}
void d2h(Handle &h, void *dst, const void *src, size_t bytes) {
    if (bytes) HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h.stream));
    HIPCHK(hipStreamSynchronize(h.stream));
}

void (*g_eager_setup)(Handle &) = nullptr;      // eager_setup (defined with the mode logic further down)
int create(const exa_model_desc_t *desc, int *id_out, bool device) {
    if (!desc || !id_out) return 1;
    try {
        auto h = std::make_unique<Handle>();
        h->m = plan_model(desc);
        h->gen = generate_module(*h->m);
        const std::string note0 = note_lookup(source_key(h->gen.source));
        if (note0.find("loopfree") != std::string::npos || note0.find("nostage") != std::string::npos) {       // decided where this module was first compiled
            h->first_key = source_key(h->gen.source);
            h->first_note = note0;
            h->loopfree_scatter = note0.find("loopfree") != std::string::npos;
            const char *keep = getenv("EXAHIP_KEEP_STAGE");           // (test infrastructure: see module_for)
            h->nostage = note0.find("nostage") != std::string::npos && !(keep && *keep == '1');
            h->gen = generate_module(*h->m, h->loopfree_scatter, h->nostage);
        }
        plan_products(*h);
        if (device) { to_device(*h); load_products(*h); if (g_eager_setup) g_eager_setup(*h); }
        else fill_params(*h);
        *id_out = put(std::move(h));
        return 0;
    } catch (const BadInput &e) {
        g_err = e.what();      // malformed table: the caller's fault
        return 1;
    } catch (const std::exception &e) {
        g_err = e.what();      // HIP failure, hipcc failure, I/O failure: internal
        return 2;
    } catch (...) {
        g_err = "unknown error";
        return 2;
    }
}


int clamp32(int64_t v) { return v > 0x7fffffffLL ? -1 : (int)v; }
int value_block(int id, int k, double *get_to, const double *set_from, int len) {
    Handle *h = get(id);
    if (!h || k < 0 || k >= (int)h->blocks.size() || h->blocks[k].kind != 2 || (!get_to && !set_from)) return 1;
    const BlockInfo &b = h->blocks[k];
    if ((int64_t)len != b.length) return 3;
    return get_to ? exa_get_value(id, b.offset, get_to, len) : exa_set_value(id, b.offset, set_from, len);
}
int struct_host(int id, bool hess, bool wide, void *r, void *c) {
    return guard(id, true, [&](Handle &h) {
        const int64_t nz = hess ? h.lnnzh : h.lnnzj;
        if (!nz) return;
        if (!wide && nz > 0x7fffffffLL) throw std::runtime_error("nnz exceeds int32");
        const size_t n = (wide ? 8 : 4) * (size_t)nz;
        h.srows.ensure(n); h.scols.ensure(n);
        do_struct(h, hess, wide, h.srows.p, h.scols.p);
        HIPCHK(hipMemcpyAsync(r, h.srows.p, n, hipMemcpyDeviceToHost, h.stream));
        d2h(h, c, h.scols.p, n);
    });
}

}  // namespace rt
}  // namespace exa

namespace exa {
int create_model(const exa_model_desc_t *desc, int *id_out, bool device) { return rt::create(desc, id_out, device); }
int attach_blocks(int id, std::vector<BlockInfo> blocks) {
    rt::Handle *h = rt::get(id);
    if (!h) return 1;
    h->blocks = std::move(blocks);
    return 0;
}
void set_last_error(const std::string &text) { rt::g_err = text; }
}  // namespace exa

extern "C" {


int exa_abi_version(void) { return EXAHIP_ABI_VERSION; }
const char *exa_last_error(void) { return g_err.c_str(); }
int exa_register_univariate(const char *name, const char *f, const char *df, const char *ddf, const char *helpers) {
    g_err.clear();
    if (!name || !f || !df || !ddf) { g_err = "exa_register_univariate: NULL argument"; return -1; }
    UserFn u; u.name = name; u.f = f; u.d1 = df; u.d11 = ddf; u.helpers = helpers ? helpers : "";
    std::string err;
    const int id = register_user_fn(false, u, &err);
    if (id < 0) g_err = "exa_register_univariate: " + err;
    return id;
}
int exa_register_univariate_fused(const char *name, const char *stmt, const char *helpers) {
    g_err.clear();
    if (!name || !stmt) { g_err = "exa_register_univariate_fused: NULL argument"; return -1; }
    UserFn u; u.name = name; u.fused = stmt; u.helpers = helpers ? helpers : "";
    std::string err;
    const int id = register_user_fn(false, u, &err);
    if (id < 0) g_err = "exa_register_univariate_fused: " + err;
    return id;
}
int exa_register_bivariate(const char *name, const char *f, const char *d1, const char *d2, const char *d11, const char *d12, const char *d22, const char *helpers) {
    g_err.clear();
    if (!name || !f || !d1 || !d2 || !d11 || !d12 || !d22) { g_err = "exa_register_bivariate: NULL argument"; return -1; }
    UserFn u; u.name = name; u.f = f; u.d1 = d1; u.d2 = d2; u.d11 = d11; u.d12 = d12; u.d22 = d22; u.helpers = helpers ? helpers : "";
    std::string err;
    const int id = register_user_fn(true, u, &err);
    if (id < 0) g_err = "exa_register_bivariate: " + err;
    return id;
}

int exa_user_function(int bivariate, int fn, int which, char *buf, int cap) {
    const UserFn *u = user_fn(bivariate != 0, fn);
    if (!u || which < 0 || which > 8 || cap < 0 || (cap > 0 && !buf)) return -1;
    const std::string *t[9] = {&u->name, &u->f, &u->d1, &u->d2, &u->d11, &u->d12, &u->d22, &u->helpers, &u->fused};
    const std::string &s = *t[which];
    if (cap > 0) { const size_t n = std::min(s.size(), (size_t)cap - 1); std::memcpy(buf, s.data(), n); buf[n] = 0; }
    return (int)s.size();
}

int exa_new_from_table(const exa_model_desc_t *desc, int *id_out) { return create(desc, id_out, true); }
int exa_plan_only(const exa_model_desc_t *desc, int *id_out) { return create(desc, id_out, false); }

int exa_cache_add(const char *name, const void *code_object, size_t len) {
    if (!name || !code_object || len == 0) return 1;
    return cache_add(name, code_object, len) ? 0 : 1;     // refuses anything that is not an AMDGPU code object
}
int exa_cache_note(const char *name, const char *note) {
    if (!name || !note || !*name) return 1;
    note_store(name, note, false);
    return 0;
}
/* code objects of a compiled model (exa_compile / a device model): k = 0 the model's module, 1 the owner-computes product
 * windows (when the model has them).  name <- the module's name (what exa_cache_add takes), path <- its file. */
int exa_code_object_count(int id) { Handle *h = get(id); return h ? (h->psource.empty() ? 1 : 2) : -1; }
int exa_code_object(int id, int k, char *name, int ncap, char *path, int pcap) {
    Handle *h = get(id);
    if (!h || k < 0 || k > 1 || (k == 1 && h->psource.empty())) return 1;
    const std::string &known = k == 0 ? h->co_name : h->pco_name;       // (<key>_safe for a module built with the conservative flags)
    if (name && ncap > 0) snprintf(name, (size_t)ncap, "%s", (known.empty() ? source_key(k == 0 ? h->gen.source : h->psource) : known).c_str());
    if (path && pcap > 0) snprintf(path, (size_t)pcap, "%s", (k == 0 ? h->hsaco_path : h->phsaco_path).c_str());
    return 0;
}
/* "" or the name of the module WITH loops this model's module replaces: a packed library hands exa_cache_note(that name,
 * "loopfree") over with the code object, so that its consumer generates the final module at once */
const char *exa_module_alias(int id) {
    Handle *h = get(id);
    if (!h) return nullptr;
    static thread_local std::string name;
    name = h->first_key;
    return name.c_str();
}
const char *exa_module_alias_note(int id) {
    Handle *h = get(id);
    if (!h) return nullptr;
    static thread_local std::string note;
    note = h->first_note;
    return note.c_str();
}
const char *exa_module_name(int id) {
    Handle *h = get(id);
    if (!h) return nullptr;
    static thread_local std::string name;
    name = source_key(h->gen.source);
    return name.c_str();
}
int exa_compile(int id) {
    return guard(id, false, [&](Handle &h) {
        CodeObject co = module_for(h, false);
        h.hsaco_path = co.path; h.build_how = co.how; h.build_ms = co.build_ms; h.co_name = co.name;
        if (!h.psource.empty()) { CodeObject pc = product_module_for(h, false); h.phsaco_path = pc.path; h.build_ms += pc.build_ms; h.pco_name = pc.name; }
    });
}
const char *exa_code_object_path(int id) {
    Handle *h = get(id);
    return h ? h->hsaco_path.c_str() : nullptr;
}

int exa_free(int id) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (id < 1 || id > (int)g_models.size() || !g_models[id - 1]) return 1;
    g_models[id - 1].reset();
    return 0;
}
int exa_nvar(int id) { Handle *h = get(id); return h ? clamp32(h->m->nvar) : -1; }
int exa_ncon(int id) { Handle *h = get(id); return h ? clamp32(h->m->ncon) : -1; }
int exa_nnzj(int id) { Handle *h = get(id); return h ? clamp32(h->m->nnzj) : -1; }
int exa_nnzh(int id) { Handle *h = get(id); return h ? clamp32(h->m->nnzh) : -1; }
int64_t exa_nvar64(int id) { Handle *h = get(id); return h ? h->m->nvar : -1; }
int64_t exa_ncon64(int id) { Handle *h = get(id); return h ? h->m->ncon : -1; }
int64_t exa_nnzj64(int id) { Handle *h = get(id); return h ? h->m->nnzj : -1; }
int64_t exa_nnzh64(int id) { Handle *h = get(id); return h ? h->m->nnzh : -1; }
int64_t exa_nnzg64(int id) { Handle *h = get(id); return h ? h->m->nnzg : -1; }
int exa_npatterns(int id) { Handle *h = get(id); return h ? (int)h->m->pats.size() : -1; }

int exa_pattern_info(int id, int p, int64_t out[9]) {
    Handle *h = get(id);
    if (!h || !out || p < 0 || p >= (int)h->m->pats.size()) return 1;
    const Pattern &q = h->m->pats[p];
    out[0] = q.kind; out[1] = q.n; out[2] = q.o0; out[3] = q.o1; out[4] = q.o2; out[5] = q.o1step; out[6] = q.o2step;
    out[7] = (int64_t)q.comp1.size(); out[8] = (int64_t)q.comp2.size();
    return 0;
}
int exa_pattern_comp(int id, int p, int order, int32_t *out) {
    Handle *h = get(id);
    if (!h || !out || p < 0 || p >= (int)h->m->pats.size() || (order != 1 && order != 2)) return 1;
    const auto &c = order == 1 ? h->m->pats[p].comp1 : h->m->pats[p].comp2;
    for (size_t i = 0; i < c.size(); i++) out[i] = c[i];
    return 0;
}
int exa_meta(int id, double *x0, double *lvar, double *uvar, double *lcon, double *ucon) {
    Handle *h = get(id);
    if (!h) return 1;
    const Model &m = *h->m;
    // an empty vector is the unmaterialised default constant (plan_model)
    auto put = [](double *dst, const std::vector<double> &v, int64_t n, double dflt) {
        if (!dst) return;
        if (v.empty()) std::fill(dst, dst + n, dflt);
        else std::memcpy(dst, v.data(), 8 * (size_t)n);
    };
    put(x0, m.x0, m.nvar, 0.0);
    put(lvar, m.lvar, m.nvar, -INFINITY);
    put(uvar, m.uvar, m.nvar, INFINITY);
    put(lcon, m.lcon, m.ncon, 0.0);
    put(ucon, m.ucon, m.ncon, 0.0);
    return 0;
}
const char *exa_kernel_source(int id) { Handle *h = get(id); return h ? h->gen.source.c_str() : nullptr; }
const char *exa_module_source(int id, int k) { Handle *h = get(id); return !h || k < 0 || k > 1 ? nullptr : (k == 0 ? h->gen.source.c_str() : h->psource.c_str()); }
int exa_set_stream(int id, void *s) { return guard(id, true, [&](Handle &h) { h.stream = (hipStream_t)s; }); }
int exa_set_value(int id, int64_t offset, const double *vals, int64_t len) {
    Handle *hh = get(id);
    if (!hh || !vals || offset < 0 || len < 0 || offset > hh->m->npar || len > hh->m->npar - offset) return 1;
    return guard(id, false, [&](Handle &h) {
        std::memcpy(h.m->theta.data() + offset, vals, 8 * (size_t)len);
        if (h.on_device && len) {
            HIPCHK(hipMemcpyAsync((double *)h.dtheta.p + offset, h.m->theta.data() + offset, 8 * (size_t)len, hipMemcpyHostToDevice, h.stream));
            HIPCHK(hipStreamSynchronize(h.stream));
        }
    });
}

/* set_value! for a parameter vector that lives on the device (the reference's set_value! is a copyto! into the device-resident
 * θ, nlp.jl:1279-1287): theta[offset .. offset+len) <- dev_vals, a device-to-device copy ordered on the model's stream — no host
 * hop, no synchronisation, capturable.  dev_vals must stay valid until the stream has passed the copy. */
int exa_set_value_dev(int id, int64_t offset, const double *dev_vals, int64_t len) {
    Handle *hh = get(id);
    if (!hh || !dev_vals || offset < 0 || len < 0 || offset > hh->m->npar || len > hh->m->npar - offset) return 1;
    return guard(id, true, [&](Handle &h) {
        if (len) HIPCHK(hipMemcpyAsync((double *)h.dtheta.p + offset, dev_vals, 8 * (size_t)len, hipMemcpyDeviceToDevice, h.stream));
        h.theta_dev_newer = true;
    });
}
/* The device-resident parameter vector itself (npar doubles; the reference's get_value returns such a view, nlp.jl:1270-1277):
 * kernels of this model launched after a write to it — on the model's stream, or ordered against it — see the new values.
 * NULL for a bad id, a plan-only handle or a model without parameters. */
double *exa_theta_ptr(int id) {
    Handle *h = get(id);
    if (!h || !h->on_device || h->m->npar == 0) return nullptr;
    h->theta_dev_newer = true;          // the caller may write through it: the host copy is no longer authoritative
    return (double *)h->dtheta.p;
}

int exa_get_value(int id, int64_t offset, double *vals, int64_t len) {
    Handle *hh = get(id);
    if (!hh || !vals || offset < 0 || len < 0 || offset > hh->m->npar || len > hh->m->npar - offset) return 1;
    if (hh->on_device && hh->theta_dev_newer) {
        // the device copy was written (exa_set_value_dev / exa_theta_ptr): bring the host copy up to date first
        const int rc = guard(id, true, [&](Handle &h) {
            HIPCHK(hipMemcpyAsync(h.m->theta.data(), h.dtheta.p, 8 * (size_t)h.m->npar, hipMemcpyDeviceToHost, h.stream));
            HIPCHK(hipStreamSynchronize(h.stream));
            // (a pointer handed out by exa_theta_ptr stays writable: only a set_value_dev is known to be over)
        });
        if (rc) return rc;
    }
    std::memcpy(vals, hh->m->theta.data() + offset, 8 * (size_t)len);   // the host copy: authoritative unless the device copy was written
    return 0;
}

// ---- named blocks (cnlp P_nblocks / P_block_name / P_block / P_get_value / P_set_value, Compiler :1476-1535) ----
int exa_nblocks(int id) { Handle *h = get(id); return h ? (int)h->blocks.size() : -1; }
int exa_block_name(int id, int k, char *buf, int cap) {
    Handle *h = get(id);
    if (!h || k < 0 || k >= (int)h->blocks.size()) return -1;
    const std::string &s = h->blocks[k].name;
    const int n = (int)s.size(), c = std::min(cap, n);
    if (c > 0 && buf) std::memcpy(buf, s.data(), (size_t)c);
    return n;
}
int exa_block(int id, int k, int *out) {
    Handle *h = get(id);
    if (!h || !out || k < 0 || k >= (int)h->blocks.size()) return 1;
    const BlockInfo &b = h->blocks[k];
    out[0] = b.kind; out[1] = (int)b.offset; out[2] = (int)b.length; out[3] = (int)b.dims.size();
    for (size_t j = 0; j < b.dims.size(); j++) out[4 + j] = (int)b.dims[j];
    return 0;
}
int exa_get_value_block(int id, int k, double *vals, int len) { return value_block(id, k, vals, nullptr, len); }
int exa_set_value_block(int id, int k, const double *vals, int len) { return value_block(id, k, nullptr, vals, len); }

/* perm_out [n of pattern `pattern`] <- a locality-improving order of the pattern's data points (0-based, stable): by the smallest
 * variable any x[...] of the pattern reaches at the point (a branch table: by from-bus, the order of a case file).  The library never
 * re-orders the tables it is given — the order of a table's rows IS the order of the constraint rows and COO slots it produces
 * (nlp.jl:1991-1992) — so applying the permutation (to every pattern that iterates over the same table, and to the y / bounds
 * of their rows) is the caller's decision, made before the model is built.  Host columns are needed: a plan-only handle. */
int exa_locality_order(int id, int pattern, int64_t *perm_out) {
    Handle *h = get(id);
    if (!h || !perm_out || pattern < 0 || pattern >= (int)h->m->pats.size()) return 1;
    if (h->on_device) { set_last_error("exa_locality_order: the host columns were released when the model went to the device (use exa_plan_only)"); return 1; }
    try {
        const std::vector<int64_t> perm = locality_order(*h->m, pattern);
        std::memcpy(perm_out, perm.data(), 8 * perm.size());
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return 2; }
}

// pattern-table view of a planned model that still holds its host columns
int exa_describe(int id, exa_model_desc_t *out) {
    Handle *hh = get(id);
    if (!hh || !out) return 1;
    if (hh->on_device) { set_last_error("exa_describe: the host columns were released when the model went to the device"); return 1; }
    Handle &h = *hh;
    Model &m = *h.m;
    h.view_pats.assign(m.pats.size(), exa_pattern_t{});
    h.view_cols.assign(m.pats.size(), {});
    for (size_t k = 0; k < m.pats.size(); k++) {
        Pattern &p = m.pats[k];
        for (Column &c : p.cols) {
            exa_column_t v{};
            v.type = c.type;
            v.data = c.type == EXA_COL_I64 ? (const void *)c.idata.data() : c.type == EXA_COL_F64 ? (const void *)c.fdata.data() : nullptr;
            v.start = c.start; v.step = c.step;
            h.view_cols[k].push_back(v);
        }
        exa_pattern_t &v = h.view_pats[k];
        v.kind = p.kind; v.n_nodes = (int)p.nodes.size(); v.nodes = p.nodes.data();
        v.root = p.root; v.target = p.target; v.base = p.base;
        v.n_cols = (int)p.cols.size(); v.cols = h.view_cols[k].data(); v.n = p.n;
    }
    *out = exa_model_desc_t{};
    out->nvar = m.nvar; out->npar = m.npar;
    auto ptr = [](const std::vector<double> &v) { return v.empty() ? nullptr : v.data(); };     // NULL = default constant
    out->x0 = ptr(m.x0); out->lvar = ptr(m.lvar); out->uvar = ptr(m.uvar); out->theta0 = ptr(m.theta);
    out->n_patterns = (int)m.pats.size(); out->minimize = m.minimize; out->patterns = h.view_pats.data();
    out->y0 = ptr(m.y0); out->lcon = ptr(m.lcon); out->ucon = ptr(m.ucon);
    return 0;
}

int exa_obj_async(int id, const double *x, double *out_dev) {
    if (!x || !out_dev) return 1;
    return guard(id, true, [&](Handle &h) { do_obj(h, x, out_dev); });
}
int exa_obj(int id, const double *x, double *out_host) {
    if (!x || !out_host) return 1;
    return guard(id, true, [&](Handle &h) { do_obj(h, x, (double *)h.dobj.p); d2h(h, out_host, h.dobj.p, 8); });
}
int exa_grad(int id, const double *x, double *g) {
    if (!x || !g) return 1;
    return guard(id, true, [&](Handle &h) { run_grad(h, x, g); });
}
int exa_cons(int id, const double *x, double *c) {
    if (!x) return 1;
    return guard(id, true, [&](Handle &h) { if (h.m->ncon && !c) throw BadInput("null output"); do_cons(h, x, c); });
}
int exa_jac(int id, const double *x, double *v) {
    if (!x) return 1;
    return guard(id, true, [&](Handle &h) { if (h.m->nnzj && !v) throw BadInput("null output"); do_jac(h, x, v); });
}
int exa_hess(int id, const double *x, const double *y, double w, double *v) {
    if (!x) return 1;
    return guard(id, true, [&](Handle &h) {
        if (h.m->nnzh && !v) throw BadInput("null output");
        do_hess(h, x, y, w, v);
    });
}
int exa_eval_fused(int id, const double *x, const double *y, double w, double *obj_dev, double *c, double *jvals, double *hvals) {
    if (!x || !obj_dev) return 1;
    return guard(id, true, [&](Handle &h) {
        if ((h.m->ncon && (!c || !y)) || (h.m->nnzj && !jvals) || (h.m->nnzh && !hvals)) throw BadInput("null output");
        do_fused(h, x, y, w, obj_dev, c, jvals, hvals);
    });
}
/* How exa_eval_all produces grad! on this model, as it stands (shard, modes): 1 = the gathered gradient's tiles ride inside the sweep's
 * launch (range-affine objective: LV, the rocket); 2 = the sweep's objective tiles STORE their first partials and zero tiles of the same
 * launch cover the other variables (data-indexed objective whose scatter the model build proved injective on the data: ACOPF's
 * generator costs) — one launch in both cases; 0 = a zero-fill launch, then the sweep adds by atomics; 3 = the gathered part in a launch
 * of its own, then the sweep adds the scattered part by atomics; 4 = grad! by the sorted gather, separately.  -1: bad id. */
int exa_eval_all_mode(int id) {
    Handle *h = get(id);
    if (!h) return -1;
    const ParamLayout &L = h->gen.layout;
    const bool scatter = !L.active[CB_GRAD].empty(), pull = !L.pull.empty();
    if (h->on_device && resolve_grad_mode(*h) == 1) return 4;
    if (pull && scatter) return 3;
    if (pull) return 1;
    if (scatter && L.gbits >= 0 && h->world == 1 && h->gridz > 0 && h->P[(size_t)L.gbits] != 0) return 2;
    return 0;
}
int exa_eval_all(int id, const double *x, const double *y, double w, double *obj_dev, double *g, double *c, double *jvals, double *hvals) {
    if (!x || !obj_dev || !g) return 1;
    return guard(id, true, [&](Handle &h) {
        if ((h.m->ncon && (!c || !y)) || (h.m->nnzj && !jvals) || (h.m->nnzh && !hvals)) throw BadInput("null output");
        do_eval_all(h, x, y, w, obj_dev, g, c, jvals, hvals);
    });
}
int exa_jprod(int id, const double *x, const double *v, double *Jv) {
    if (!x || !v) return 1;
    return guard(id, true, [&](Handle &h) { if (h.m->ncon && !Jv) throw BadInput("null output"); do_jprod(h, x, v, Jv); });
}
int exa_jac_structure(int id, int32_t *r, int32_t *c) {
    return guard(id, true, [&](Handle &h) { if (h.lnnzj > 0x7fffffffLL) throw std::runtime_error("nnzj exceeds int32"); do_struct(h, false, false, r, c); });
}
int exa_hess_structure(int id, int32_t *r, int32_t *c) {
    return guard(id, true, [&](Handle &h) { if (h.lnnzh > 0x7fffffffLL) throw std::runtime_error("nnzh exceeds int32"); do_struct(h, true, false, r, c); });
}
int exa_jac_structure64(int id, int64_t *r, int64_t *c) { return guard(id, true, [&](Handle &h) { do_struct(h, false, true, r, c); }); }
int exa_hess_structure64(int id, int64_t *r, int64_t *c) { return guard(id, true, [&](Handle &h) { do_struct(h, true, true, r, c); }); }

// ---- host-pointer variants ----------------------------------------------------------------------------
int exa_obj_host(int id, const double *x, double *out) {
    if (!x || !out) return 1;
    return guard(id, true, [&](Handle &h) {
        h2d(h, h.sx, x, 8 * (size_t)h.m->nvar);
        do_obj(h, (const double *)h.sx.p, (double *)h.dobj.p);
        d2h(h, out, h.dobj.p, 8);
    });
}
int exa_grad_host(int id, const double *x, double *g) {
    if (!x || !g) return 1;
    return guard(id, true, [&](Handle &h) {
        const size_t n = 8 * (size_t)h.m->nvar;
        h2d(h, h.sx, x, n);
        h.sout.ensure(n);
        zero_if_sharded(h, h.sout.p, n);
        run_grad(h, (const double *)h.sx.p, (double *)h.sout.p);
        d2h(h, g, h.sout.p, n);
    });
}
int exa_cons_host(int id, const double *x, double *c) {
    if (!x) return 1;
    return guard(id, true, [&](Handle &h) {
        const size_t n = 8 * (size_t)h.m->ncon;
        if (!n) return;
        h2d(h, h.sx, x, 8 * (size_t)h.m->nvar);
        h.sout.ensure(n);
        zero_if_sharded(h, h.sout.p, n);
        do_cons(h, (const double *)h.sx.p, (double *)h.sout.p);
        d2h(h, c, h.sout.p, n);
    });
}
int exa_jac_host(int id, const double *x, double *v) {
    if (!x) return 1;
    return guard(id, true, [&](Handle &h) {
        const size_t n = 8 * (size_t)h.lnnzj;
        if (!n) return;
        h2d(h, h.sx, x, 8 * (size_t)h.m->nvar);
        h.sout.ensure(n);
        do_jac(h, (const double *)h.sx.p, (double *)h.sout.p);
        d2h(h, v, h.sout.p, n);
    });
}
int exa_hess_host(int id, const double *x, const double *y, double w, double *v) {
    if (!x) return 1;
    return guard(id, true, [&](Handle &h) {
        const size_t n = 8 * (size_t)h.lnnzh;
        if (!n) return;
        h2d(h, h.sx, x, 8 * (size_t)h.m->nvar);
        if (h.m->ncon && y) h2d(h, h.sy, y, 8 * (size_t)h.m->ncon);
        else h.sy.ensure(8);
        h.sout.ensure(n);
        do_hess(h, (const double *)h.sx.p, h.m->ncon && !y ? nullptr : (const double *)h.sy.p, w, (double *)h.sout.p);      // y == NULL: objective only
        d2h(h, v, h.sout.p, n);
    });
}
int exa_jprod_host(int id, const double *x, const double *v, double *Jv) {
    if (!x || !v) return 1;
    return guard(id, true, [&](Handle &h) {
        const size_t n = 8 * (size_t)h.m->ncon;
        if (!n) return;
        h2d(h, h.sx, x, 8 * (size_t)h.m->nvar);
        h2d(h, h.sv, v, 8 * (size_t)h.m->nvar);
        h.sout.ensure(n);
        zero_if_sharded(h, h.sout.p, n);
        do_jprod(h, (const double *)h.sx.p, (const double *)h.sv.p, (double *)h.sout.p);
        d2h(h, Jv, h.sout.p, n);
    });
}
int exa_jac_structure_host(int id, int32_t *r, int32_t *c) { return struct_host(id, false, false, r, c); }
int exa_hess_structure_host(int id, int32_t *r, int32_t *c) { return struct_host(id, true, false, r, c); }
int exa_jac_structure64_host(int id, int64_t *r, int64_t *c) { return struct_host(id, false, true, r, c); }
int exa_hess_structure64_host(int id, int64_t *r, int64_t *c) { return struct_host(id, true, true, r, c); }
int exa_sync(int id) { return guard(id, true, [&](Handle &h) { HIPCHK(hipStreamSynchronize(h.stream)); }); }

// ---- how the module was obtained ---------------------------------------------------------------------------------------
int exa_build_audit(int id, char *buf, int cap) {
    Handle *h = get(id);
    if (!h) return -1;
    std::string out;
    for (const Handle::Audit &a : h->audits) {
        const std::string head = a.which + " " + a.name + " " + (a.safe ? "safe" : "default") + " ";
        if (!a.readable) { out += head + "? unreadable\n"; continue; }
        for (const KernelInfo &k : a.kernels) {
            char line[256];
            snprintf(line, sizeof line, "%s %d %d %d %d %d %d %s\n", k.name.c_str(), k.vgpr, k.agpr, k.scratch, k.vgpr_spill, k.sgpr_spill, k.lds, k.fits() ? "fits" : "oversized");
            out += head + line;
        }
    }
    if (buf && cap > 0) snprintf(buf, (size_t)cap, "%s", out.c_str());
    return (int)out.size();
}
int exa_build_info(int id, char *how, int cap, double *build_ms) {
    Handle *h = get(id);
    if (!h) return 1;
    if (how && cap > 0) { snprintf(how, (size_t)cap, "%s", h->build_how.c_str()); }
    if (build_ms) *build_ms = h->build_ms;
    return 0;
}

}  // extern "C"
