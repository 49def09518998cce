// exa_runtime.cpp — libexahip.so runtime + C ABI (include/exahip.h).
//
// Owns: model registry, kernel-module build (hipcc --genco for gfx950, cached on disk by source hash), device
// copies of the SoA iterator columns / theta / parameter table, launches on the model's HIP stream.
// Replaces the host drivers of ext/ExaModelsKernelAbstractions.jl:253-351, 515-547 (one launch per pattern per
// callback + fill!) with ONE fused launch per callback and no fill! for the COO outputs.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <mutex>
#include <sstream>
#include <stdexcept>

#include "exa_build.hpp"
#include "exa_comm.hpp"
#include "exa_compress.hpp"
#include "exa_internal.hpp"
#include "../../include/exahip_recipe.h"

using namespace exa;

namespace {

thread_local std::string g_err;
std::mutex g_mu;
// EXAHIP_VERBOSE=1: one stderr line per decision (tuning, register spills of the scatter kernels, window plans)
bool verbose() { static const bool v = [] { const char *e = getenv("EXAHIP_VERBOSE"); return e && atoi(e) != 0; }(); return v; }

struct HipError : std::runtime_error { using std::runtime_error::runtime_error; };
#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) throw HipError(std::string(#expr) + ": " + hipGetErrorString(_e));              \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    void ensure(size_t n) {
        if (n <= bytes) return;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        HIPCHK(hipMalloc(&p, n ? n : 8));
        bytes = n;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

struct Handle {
    std::unique_ptr<Model> m;
    Generated gen;
    std::string hsaco_path, build_how, co_name, pco_name;      // co_name / pco_name: what exa_cache_add takes for the two modules
    double build_ms = 0.0;
    bool on_device = false;
    int rank = 0, world = 1;
    // multi-GPU: either an RCCL communicator or a host-supplied reducer completes obj / grad / cons / products
    void *nccl = nullptr;
    bool nccl_owned = false;
    exa_allreduce_fn hook = nullptr;
    void *hook_ctx = nullptr;
    bool reduce = true;
    bool theta_dev_newer = false;     // exa_set_value_dev wrote the device copy of theta: the host copy is refreshed on the next exa_get_value
    // COO outputs of a sharded model: false = global slot positions (ranks fill disjoint slices of one global vector),
    // true = this rank's slots packed into a slice-sized buffer, pattern after pattern (exa_set_coo_local)
    bool coo_local = false;
    int64_t lnnzj = 0, lnnzh = 0;            // length of the jac / hess COO buffers in the current mode
    std::vector<int64_t> lo1, lo2;            // local first slot per pattern (coo_local)
    std::string devname;
    hipStream_t stream = nullptr;
    hipModule_t module = nullptr;
    hipFunction_t f_auglong = nullptr, f_augfold = nullptr, f_auggather = nullptr, f_gradpull = nullptr, f_fused = nullptr, f_jprod = nullptr, f_jtprod = nullptr, f_hprod = nullptr, f_obj = nullptr, f_red = nullptr, f_zero = nullptr, f_grad = nullptr, f_cons = nullptr, f_jac = nullptr,
                  f_hess = nullptr, f_hessc = nullptr, f_hesscl = nullptr, f_cons1 = nullptr, f_jprod1 = nullptr, f_js32 = nullptr, f_js64 = nullptr, f_hs32 = nullptr, f_hs64 = nullptr;
    std::vector<int64_t> P;                 // host copy of the parameter table
    std::vector<int64_t> grid = std::vector<int64_t>(CB_COUNT, 0);
    DevBuf daugcoef;
    DevBuf daugcsr, daugsrc;                // exa_cons1: CSR over constraint rows of the augmentation terms (pattern << 40 | point)
    bool cons1 = false;
    DevBuf dsink;                           // 64 doubles nobody reads (ParamLayout::sink)
    DevBuf dP, dtheta, dpart, ddone, dobj, daugbuf, daugrows, daugptr, daugperm, dauglong, daugpartial;
    int64_t aug_nlong = 0, aug_chunks = 0;   // rows collecting > 512 augmentation terms: cooperative summation
    DevBuf dmap[CB_COUNT][2];               // per-callback block maps: [0] units one after the other, [1] interleaved in runs of 128
    DevBuf dmapg[2];                        // exa_eval_all: the fused sweep's units + the gathered-gradient tiles (same two orders)
    int64_t gridg = 0;
    // objective-only forms of hess_coord! / hprod! (y == NULL): a second parameter table whose CB_HESS / CB_HPROD block maps
    // hold the objective groups alone, and the COO ranges of the constraint patterns (they receive exact zeros)
    DevBuf dPobj, dmapobj[2];
    int64_t gridobj[2] = {0, 0};
    std::vector<std::pair<int64_t, int64_t>> con_hess_ranges;
    int order[CB_COUNT] = {0};              // which map is active
    int norders[CB_COUNT] = {1};            // how many maps exist: exa_tune measures all of them
    int hess_variant = 0;                   // hess_coord! kernel: 0 exa_hess (one tile per workgroup), 1 chained, grouped, pipelined: exa_hesscl
                                            // (x staged through LDS) where this shard's stretches fit, else exa_hessc; 2 exa_hessc always
    bool stage_ok = false;                  // exa_hesscl's stretch geometry holds for this shard (fill_params)
    double hess_stream_bytes = 0.0;         // HBM bytes one hess_coord! of this shard streams (outputs + x + y)
    int64_t fused_nobj = 0;                 // objective partial sums written by exa_fused
    std::vector<DevBuf> dcols;              // flattened over patterns
    std::vector<std::vector<int>> colslot;  // [pattern][col] -> index into dcols (or -1 for RANGE)
    DevBuf sx, sy, sv, sout, srows, scols;  // scratch of the *_host variants
    CompressedCOO cj, ch;                   // duplicate-summed COO maps (exa_compress)
    DevBuf cbuf;                            // uncompressed values of the last compressed evaluation
    bool compressed = false;
    // windowed fast path of exa_cjac / exa_chess (window_setup): second module, per-matrix tables
    struct Window {
        bool ok = false;
        int W = 0, nx = 0, smax = 1, lds_bytes = 0;
        int64_t nwin = 0;
        hipFunction_t fw = nullptr, fx = nullptr, fs = nullptr;
        int64_t ns_blocks = 0;             // workgroups of the shared-entry pass (exa_c*s)
        DevBuf Q, R, X, T, E, xbuf, S, F, part;
        // the tables as planned on the host (window_plan); window_upload puts them on the device.  Product windows are
        // planned without a device (plan-only handles generate and compile their module too) and uploaded by to_device.
        std::vector<int64_t> hQ, hX, hS, hF;
        std::vector<int32_t> hR, hT, hE;
        int64_t xbuf_doubles = 0, nparts = 0;
        // output ranges: window j covers entries [o + j*W, min(o + (j+1)*W, end)) of every space (one space unless block-owned)
        struct Space { int64_t o, end, W; };
        std::vector<Space> spaces;
        bool planned = false;              // products: the plan exists (host); ok = its kernels are loaded as well
        bool has_shared = false;           // some entry is added to by every data point (partial sums + fold: not owner-shardable)
        std::string why;                   // why the fast path was not taken (exa_compress_info / exa_product_info)
    } wj, wh, wp[2];                       // compressed Jacobian / Hessian; J'v / Hv (WK_JTPROD, WK_HPROD)
    hipModule_t wmodule = nullptr;
    // owner-computes products: third module (generated at model build when every scatter target is range-affine)
    WindowSpec pspec;
    std::string psource, phsaco_path;
    bool no_attach = false, no_attach_c = false;   // all-points entries by the kernel of their own, never inside a window kernel (products / compressed COO)
    hipModule_t pmodule = nullptr;
    // permuted-store path of exa_cjac / exa_chess for matrices the windows do not fit (exa_c*p, see WindowSpec)
    struct Scatter { bool ok = false; hipFunction_t f = nullptr; DevBuf pos; } sj, sh;
    // ... and its merged-slot form for the Hessian (exa_chessm): the merged slot space has its own sorted lists
    bool merged = false;
    int device = -1;            // the HIP device that was current in exa_create (DeviceScope)
    // what the compiled kernels of every module of this model need (audited_code_object): which -> report
    struct Audit { std::string which, name; bool safe = false, readable = false; std::vector<KernelInfo> kernels; };
    std::vector<Audit> audits;
    bool loopfree_scatter = false;     // the module is the one generated without loops in the scatter kernels (module_for)
    std::string first_key;             // ... and this is the key of the module with loops it replaces (its "loopfree" note)
    // grad! by sorted gather (the reference's scheme, deterministic): gradient COO + (variable, slot) lists, built on demand
    hipFunction_t f_gradv = nullptr, f_gstruct = nullptr;
    SortedIndex gbyvar;
    DevBuf gbuf, gone;
    bool grad_ready = false;
    int grad_mode = -1;         // 0 pull + atomics, 1 sorted gather, -1 undecided (persisted exa_tune decision, else 0)
    hipFunction_t f_chessm = nullptr, f_hstructm = nullptr;
    CompressedCOO chm;
    DevBuf dM;
    int64_t nmerged = 0;
    std::vector<BlockInfo> blocks;          // named blocks (recipes; empty for plain pattern tables)
    std::vector<exa_pattern_t> view_pats;   // exa_describe: pattern-table view of the host copy
    std::vector<std::vector<exa_column_t>> view_cols;
    // sorted-gather products (the reference's prod helper): COO coordinates + entries grouped by column / by row
    DevBuf pjrows, pjcols, phrows, phcols;
    SortedIndex jbycol, hbyrow, hbycol;
    bool prod_ready_j = false, prod_ready_h = false;
    int jt_mode = -1, hp_mode = -1;         // -1 undecided, 0 atomics in the sweep, 1 COO + sorted gather, 2 owner-computes windows, 3 owner pull
    // owner pull (exa_gen_pull.cpp; models whose scatter targets come from data columns): kernels of the product module, the
    // variable -> item-slot lists (built by pull_setup, never inside a callback)
    struct Pull {
        bool planned = false, ready = false;
        std::vector<int> nitems;           // items per fused group of CB_JTPROD / CB_HPROD
        hipFunction_t fkeys = nullptr, fpull = nullptr;
        SortedIndex idx;
        DevBuf first;                      // int64[groups]: first item slot of every group
        int64_t total = 0;
        std::string why;
    } pl[2];
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    ~Handle() {
        // (every other entry point runs with the model's device current, DeviceScope; so must the teardown: hipFree /
        // hipModuleUnload / ncclCommDestroy of a model created on GPU 1 from a thread whose current device is GPU 0)
        int prev = -1;
        const bool switched = on_device && device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != device && hipSetDevice(device) == hipSuccess;
        if (on_device) (void)hipStreamSynchronize(stream);
        if (on_device) {
            daugcoef.release(); daugcsr.release(); daugsrc.release(); dsink.release(); dP.release(); dtheta.release(); dpart.release(); ddone.release(); dobj.release();
            daugbuf.release(); daugrows.release(); daugptr.release(); daugperm.release(); dauglong.release(); daugpartial.release();
            for (auto &b : dmap) { b[0].release(); b[1].release(); }
            dmapg[0].release(); dmapg[1].release(); dPobj.release(); dmapobj[0].release(); dmapobj[1].release();
            cj.release(); ch.release(); cbuf.release();
            for (Window *w : {&wj, &wh, &wp[0], &wp[1]}) { w->Q.release(); w->R.release(); w->X.release(); w->T.release(); w->E.release(); w->xbuf.release(); w->S.release(); w->F.release(); w->part.release(); }
            sj.pos.release(); sh.pos.release(); chm.release(); dM.release();
            gbyvar.release(); gbuf.release(); gone.release();
            if (wmodule) (void)hipModuleUnload(wmodule);
            if (pmodule) (void)hipModuleUnload(pmodule);
            pjrows.release(); pjcols.release(); phrows.release(); phcols.release();
            jbycol.release(); hbyrow.release(); hbycol.release();
            for (auto &q : pl) { q.idx.release(); q.first.release(); }
            for (auto &b : dcols) b.release();
            sx.release(); sy.release(); sv.release(); sout.release(); srows.release(); scols.release();
            if (ev0) (void)hipEventDestroy(ev0);
            if (ev1) (void)hipEventDestroy(ev1);
            if (module) (void)hipModuleUnload(module);
        }
        if (nccl && nccl_owned) { try { rccl_comm_destroy(nccl); } catch (...) {} }
        if (switched) (void)hipSetDevice(prev);
    }
};

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

struct Handle;
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
                for (int k = 0; k < 2; k++) {
                    std::vector<int64_t> mp = build_units(nbg, k ? 128 : 0);
                    h.dmapg[k].ensure(sizeof(int64_t) * std::max<size_t>(mp.size(), 1));
                    if (!mp.empty()) HIPCHK(hipMemcpy(h.dmapg[k].p, mp.data(), sizeof(int64_t) * mp.size(), hipMemcpyHostToDevice));
                }
            }
        }
        const bool tunable = cb == CB_HESS || cb == CB_HESSC || cb == CB_JAC || cb == CB_FUSED || cb == CB_CONS;
        if (cb == CB_HESS) h.hess_stream_bytes = out_bytes + 8.0 * (double)(m.nvar + m.ncon) / h.world;
        const bool two = h.on_device && total > 0 && na > 1 && tunable && out_bytes >= 128e6;
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
                if (tune_lookup(source_key(h.gen.source), tune_signature(h, "order" + std::to_string(cb)), &pv) && (pv == 0 || pv == 1)) {
                    h.order[cb] = pv;
                    h.P[L.blk[cb]] = (int64_t)(uintptr_t)h.dmap[cb][pv].p;
                }
            }
        }
    }
    // hess_coord! kernel: the persisted measurement (exa_tune) if there is one, else by size — the chained kernel pays
    // for its pipelining with twice the registers: it wins where the call streams gigabytes through HBM (LV N = 3e7:
    // 0.472 against 0.529 ms, N = 1e8: 1.52 against 1.76) and loses where x, y and much of the output sit in the 256 MB
    // MALL or the arithmetic dominates (LV N = 1e7: 0.156 against 0.135 ms; rocket 0.108 / 0.086; ACOPF 0.034 / 0.018)
    h.hess_variant = 0;
    if (L.chain[CB_HESSC] > 0) {
        const char *ce = getenv("EXAHIP_HESS_VARIANT");
        int pv = 0;
        if (ce) h.hess_variant = std::min(2, std::max(0, atoi(ce)));
        else if (tune_lookup(source_key(h.gen.source), tune_signature(h, "hessvariant"), &pv)) h.hess_variant = std::min(2, std::max(0, pv));
        else h.hess_variant = h.hess_stream_bytes >= 1.5e9;
    }
    // exa_hesscl stages, per wavefront and tile, ONE stretch of 64 + kStageHalo variables for all patterns of a group: the
    // patterns' first variables (of THIS shard's first points) must lie within the halo of each other
    h.stage_ok = L.staged;
    if (L.staged)
        for (const auto &grp : L.groups[CB_HESSC]) {
            int64_t bmin = INT64_MAX;
            for (int k : grp) {
                if (h.P[L.pat[k].hi] <= h.P[L.pat[k].lo]) h.stage_ok = false;
                bmin = std::min(bmin, h.P[L.stage[k].word] + h.P[L.pat[k].lo] + L.stage[k].cmin);
            }
            for (int k : grp)
                if (h.P[L.stage[k].word] + h.P[L.pat[k].lo] + L.stage[k].cmax - bmin > kStageHalo) h.stage_ok = false;
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
        h.dpart.ensure(sizeof(double) * (size_t)(std::max(h.grid[CB_OBJ], h.grid[CB_FUSED]) + 1));
        if (!h.ddone.p) { h.ddone.ensure(64); HIPCHK(hipMemset(h.ddone.p, 0, 64)); }      // exa_obj's arrival counter (re-armed by the kernel)
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
CodeObject audited_code_object(Handle &h, const std::string &which, const std::string &source, bool memory_only_ok, const CodeObject *have = nullptr) {
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
    if (!h.loopfree_scatter && scatter_kernels_spill(co)) {
        h.first_key = co.key;
        note_store(co.key, "loopfree", true);
        h.loopfree_scatter = true;
        h.gen = generate_module(*h.m, true);
        spent = co.build_ms;
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
    h.f_obj = fn("exa_obj"); h.f_red = fn("exa_reduce_partials"); h.f_zero = fn("exa_zero"); h.f_grad = fn("exa_grad"); h.f_cons = fn("exa_cons");
    h.f_auggather = fn("exa_aug_gather"); h.f_gradpull = fn("exa_grad_pull");
    h.f_auglong = fn("exa_aug_long"); h.f_augfold = fn("exa_aug_fold");
    h.f_fused = fn("exa_fused");
    h.f_jprod = fn("exa_jprod"); h.f_jtprod = fn("exa_jtprod"); h.f_hprod = fn("exa_hprod"); h.f_jac = fn("exa_jac"); h.f_hess = fn("exa_hess");
    if (h.gen.layout.chain[CB_HESSC] > 0) h.f_hessc = fn("exa_hessc");
    if (h.gen.layout.chain[CB_HESSC] > 0 && h.gen.layout.staged) h.f_hesscl = fn("exa_hesscl");
    h.f_cons1 = fn("exa_cons1");
    h.f_gradv = fn("exa_gradv"); h.f_gstruct = fn("exa_gstruct");
    if (m.aug_linear || m.nconaug == 0) h.f_jprod1 = fn("exa_jprod1");
    h.f_js32 = fn("exa_jstruct32"); h.f_js64 = fn("exa_jstruct64"); h.f_hs32 = fn("exa_hstruct32"); h.f_hs64 = fn("exa_hstruct64");
    h.colslot.resize(m.pats.size());
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
    HIPCHK(hipEventCreate(&h.ev0));
    HIPCHK(hipEventCreate(&h.ev1));
    fill_params(h);
}

void launch(Handle &h, hipFunction_t f, int64_t grid, unsigned block, void **args) {
    if (grid <= 0) return;
    if (grid > 0x7fffffffLL) throw std::runtime_error("grid too large");
    HIPCHK(hipModuleLaunchKernel(f, (unsigned)grid, 1, 1, block, 1, 1, 0, h.stream, args, nullptr));
}

// zero-fill of n doubles on the model's stream (exa_zero)
void zero_fill(Handle &h, void *p, int64_t n) {
    if (n <= 0) return;
    void *a[] = {&p, &n};
    launch(h, h.f_zero, (n + 4 * kBlock - 1) / (4 * kBlock), kBlock, a);
}

// Chooses the block order of callback `cb` by measurement (exa_tune only — callbacks never measure): both orders are
// timed on the model's stream (the outputs are simply rewritten with the same values), the faster map is installed in
// P[] and the decision persisted next to the cached module.  Synchronises the stream.
template <class F>
float tune_order(Handle &h, int cb, F &&run) {
    const int n = std::max(1, h.norders[cb]);
    const ParamLayout &L = h.gen.layout;
    float t[2] = {1e30f, 1e30f};
    // bring the clocks up first: the governor idles at ~570 MHz and needs tens of ms of load, and at low clocks the
    // orders rank differently than in steady state (measured: cold tuning picked the slower order 2 times out of 3)
    {
        HIPCHK(hipEventRecord(h.ev0, h.stream));
        for (int it = 0; it < 200; it++) {
            for (int r = 0; r < 4; r++) run();
            HIPCHK(hipEventRecord(h.ev1, h.stream));
            HIPCHK(hipEventSynchronize(h.ev1));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, h.ev0, h.ev1));
            if (ms > 60.f) break;
        }
    }
    // A/B rounds, minimum per order: a single sample per order is within the run-to-run noise of the difference being
    // measured (5-7 %); round 0 only warms up
    auto install = [&](int k) {
        if (n < 2) return;
        h.P[L.blk[cb]] = (int64_t)(uintptr_t)h.dmap[cb][k].p;
        HIPCHK(hipMemcpyAsync((int64_t *)h.dP.p + L.blk[cb], &h.P[L.blk[cb]], 8, hipMemcpyHostToDevice, h.stream));
        h.order[cb] = k;
    };
    for (int round = 0; round < 4; round++) {
        for (int k = 0; k < n; k++) {
            install(k);
            run();
            HIPCHK(hipEventRecord(h.ev0, h.stream));
            for (int r = 0; r < 4; r++) run();
            HIPCHK(hipEventRecord(h.ev1, h.stream));
            HIPCHK(hipEventSynchronize(h.ev1));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, h.ev0, h.ev1));
            if (round > 0 && ms < t[k]) t[k] = ms;
        }
    }
    const int best = n > 1 && t[1] < t[0] ? 1 : 0;
    install(best);
    HIPCHK(hipStreamSynchronize(h.stream));
    if (n > 1) tune_store(source_key(h.gen.source), tune_signature(h, "order" + std::to_string(cb)), best);
    if (verbose()) fprintf(stderr, "[exahip] tune cb=%d: %.4f %.4f ms per 4 launches -> order %d\n", cb, t[0], n > 1 ? t[1] : 0.f, best);
    return t[best];
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

// Completes a partial result of a sharded model: sum over the ranks, in place, on the model's stream — RCCL
// (exa_comm_init / exa_comm_attach) or the host's reducer (exa_comm_hook).  A model without a communicator returns its
// partial sums (exa_set_shard alone: the host layer reduces).
void allreduce(Handle &h, double *buf, int64_t count) {
    if (!h.reduce || count <= 0) return;
    if (h.nccl) rccl_allreduce_sum_f64(h.nccl, buf, count, h.stream);
    else if (h.hook) {
        const int rc = h.hook(h.hook_ctx, buf, count, (void *)h.stream);
        if (rc != 0) throw std::runtime_error("the host's all-reduce hook returned status " + std::to_string(rc));
    }
}

// Windows [w0, w1) a rank of a sharded model owns, and the pieces of the output they cover (owner computes: complete
// values, nothing to sum).  pieces: (offset, count, owner rank) for EVERY rank — what an all-gather-v needs.
void owned_windows(const Handle &h, const Handle::Window &w, int rank, int64_t *w0, int64_t *w1) {
    *w0 = part_lo(w.nwin, rank, h.world);
    *w1 = part_lo(w.nwin, rank + 1, h.world);
}
std::vector<Piece> window_pieces(const Handle &h, const Handle::Window &w) {
    std::vector<Piece> out;
    for (int r = 0; r < h.world; r++) {
        int64_t w0, w1;
        owned_windows(h, w, r, &w0, &w1);
        for (size_t q = 0; q < w.spaces.size(); q++) {
            const auto &sp = w.spaces[q];
            const int64_t a = std::min(sp.o + w0 * sp.W, sp.end), b = std::min(sp.o + w1 * sp.W, sp.end);
            if (b > a) out.push_back({a, b - a, r, (int)q});
        }
    }
    return out;
}
// Makes a vector whole whose pieces are complete on their owners (in place): RCCL — per set of pieces ONE in-place ncclAllGather
// where they are regular (plan_allgather, exa_comm.cpp), grouped broadcasts otherwise; a host reducer (exa_comm_hook) only
// knows how to sum, so the other ranks' pieces are zeroed and the covering range summed.
void allgatherv(Handle &h, double *buf, const std::vector<Piece> &pieces, bool force = false) {
    if ((!h.reduce && !force) || h.world == 1 || pieces.empty()) return;
    if (h.nccl) {
        rccl_run_plan_f64(h.nccl, buf, plan_allgather(pieces, h.world), h.rank, h.stream);
    } else if (h.hook) {
        int64_t lo = INT64_MAX, hi = 0;
        for (const Piece &q : pieces) {
            if (q.root != h.rank) HIPCHK(hipMemsetAsync(buf + q.off, 0, 8 * (size_t)q.count, h.stream));
            lo = std::min(lo, q.off); hi = std::max(hi, q.off + q.count);
        }
        const int rc = h.hook(h.hook_ctx, buf + lo, hi - lo, (void *)h.stream);
        if (rc != 0) throw std::runtime_error("the host's all-reduce hook returned status " + std::to_string(rc));
    }
}
std::vector<Piece> var_pieces(const Handle &h) {
    std::vector<Piece> out;
    for (int r = 0; r < h.world; r++) out.push_back({own_var_lo(h, r), own_var_lo(h, r + 1) - own_var_lo(h, r), r, 0});
    return out;
}
// constraint rows the ranks own: the base rows of their data points, pattern by pattern
std::vector<Piece> row_pieces(const Handle &h) {
    std::vector<Piece> out;
    for (int r = 0; r < h.world; r++)
        for (size_t k = 0; k < h.m->pats.size(); k++) {
            const Pattern &p = h.m->pats[k];
            if (p.kind != EXA_PAT_CON || p.n <= 0) continue;
            const int64_t lo = part_lo(p.n, r, h.world), hi = part_lo(p.n, r + 1, h.world);
            if (hi > lo) out.push_back({p.o0 + lo, hi - lo, r, (int)k});
        }
    return out;
}
// slots of the Jacobian / Hessian COO the ranks own (global positions)
std::vector<Piece> coo_pieces(const Handle &h, bool hess) {
    std::vector<Piece> out;
    for (int r = 0; r < h.world; r++)
        for (size_t k = 0; k < h.m->pats.size(); k++) {
            const Pattern &p = h.m->pats[k];
            const int64_t step = hess ? p.o2step : (p.kind != EXA_PAT_OBJ ? p.o1step : 0);
            if (step <= 0 || p.n <= 0) continue;
            const int64_t lo = part_lo(p.n, r, h.world), hi = part_lo(p.n, r + 1, h.world);
            if (hi > lo) out.push_back({(hess ? p.o2 : p.o1) + step * lo, step * (hi - lo), r, (int)k});
        }
    return out;
}

// ---- callbacks (device pointers, asynchronous) ------------------------------------------------------------
void do_obj(Handle &h, const double *x, double *out_dev) {
    const void *P = h.dP.p, *th = h.dtheta.p;
    void *part = h.dpart.p;
    int64_t n = h.grid[CB_OBJ];
    if (n == 0) { HIPCHK(hipMemsetAsync(out_dev, 0, sizeof(double), h.stream)); allreduce(h, out_dev, 1); return; }
    // up to kObjFoldMax workgroups: the one of exa_obj that finishes last folds the partial sums (one launch); more: a second
    // launch of 1024 threads
    void *done = n <= kObjFoldMax ? h.ddone.p : nullptr;
    void *a1[] = {&P, &x, &th, &part, &done, &out_dev};
    launch(h, h.f_obj, n, kBlock, a1);
    if (!done) {
        void *a2[] = {&part, &n, &out_dev};
        launch(h, h.f_red, 1, 1024, a2);
    }
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
    {
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
static int resolve_grad_mode(Handle &h) {
    if (h.grad_mode < 0) {
        int v = 0;
        h.grad_mode = tune_lookup(source_key(h.gen.source), tune_signature(h, "grad"), &v) && v == 1 ? 1 : 0;
    }
    if (h.grad_mode == 1 && !grad_sorted_possible(h)) return 0;
    if (h.grad_mode == 1 && !h.grad_ready) { if (capturing(h)) return 0; grad_setup(h); }
    return h.grad_mode;
}
static void run_grad(Handle &h, const double *x, double *g) {
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
    if (h.cons1) {
        // ONE launch: every base row's thread evaluates the row's augmentation terms itself (exa_cons1)
        const void *ptr = h.daugcsr.p, *src = h.daugsrc.p, *coef = h.daugcoef.p;
        void *a1[] = {&P, &x, &th, &c, &ptr, &src, &coef};
        launch(h, h.f_cons1, h.grid[CB_CONS1], kBlock, a1);
        allgatherv(h, c, row_pieces(h));
        return;
    }
    // base rows (plain stores into c) and augmentation terms (into the value buffer, coalesced)
    void *a[] = {&P, &x, &th, &c, &buf};
    launch(h, h.f_cons, h.grid[CB_CONS], kBlock, a);
    if (h.m->nconaug) aug_gather(h, buf, c);       // then one deterministic gather per target row
    if (owner) allgatherv(h, c, row_pieces(h));
    else allreduce(h, c, h.m->ncon);
}
void do_jac(Handle &h, const double *x, double *v) {
    const void *P = h.dP.p, *th = h.dtheta.p;
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
        void *sink = h.dsink.p;
        void *a[] = {&P, &x, &y, &th, &v, &sigma, &sink};
        launch(h, h.hess_variant == 1 && h.f_hesscl && h.stage_ok ? h.f_hesscl : h.f_hessc, h.grid[CB_HESSC], kBlock, a);
        return;
    }
    void *a[] = {&P, &x, &y, &th, &v, &sigma};
    launch(h, h.f_hess, h.grid[CB_HESS], kBlock, a);
}
// fused obj + cons_nln! + jac_coord! + hess_coord! at one x
// gout: null, or the gradient vector the objective patterns that are NOT gathered per variable add their first partials
// to (exa_eval_all; the caller has zero-filled it or run exa_grad_pull into it)
void do_fused(Handle &h, const double *x, const double *y, double sigma, double *obj_dev, double *c, double *jv, double *hv, double *gout = nullptr,
              bool with_pull = false) {
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
        bmap = h.dmapg[h.order[CB_FUSED] ? 1 : 0].p;
        n = h.gridg;
    }
    // objective partial sums: up to kObjFoldMax of them are folded by the objective workgroup that arrives last (as in do_obj)
    int64_t nobj = h.fused_nobj;
    void *done = n > 0 && nobj > 0 && nobj <= kObjFoldMax ? h.ddone.p : nullptr;
    void *a[] = {&P, &x, &y, &th, &part, &c, &buf, &jv, &hv, &sigma, &ap, &as, &ac, &gout, &bmap, &vb, &ve, &own_lo, &own_hi, &done, &nobj, &obj_dev};
    launch(h, h.f_fused, n, kBlock, a);
    if (n > 0 && nobj > 0) { if (!done) { void *a2[] = {&part, &nobj, &obj_dev}; launch(h, h.f_red, 1, 1024, a2); } }
    else HIPCHK(hipMemsetAsync(obj_dev, 0, sizeof(double), h.stream));
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
    } else if (!pull) {
        zero_fill(h, g, nvar);
    }
    // (only gathered patterns: their tiles ride inside the sweep's launch)
    do_fused(h, x, y, sigma, obj_dev, c, jv, hv, g, /*with_pull=*/pull && !scatter);
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
void do_struct(Handle &h, bool hess, bool wide, void *rows, void *cols);
void do_jac(Handle &h, const double *x, double *v);
void do_hess(Handle &h, const double *x, const double *y, double sigma, double *v);

// Products, second implementation: evaluate the COO and gather it through build-time sorted lists — the reference's
// own scheme (kerspmv2 / kersyspmv, KA ext :482-511).  Deterministic and contention-free; costs one extra pass over
// the COO.  Which of the two implementations runs is decided per model by MEASURING both once (exa_jtprod/exa_hprod
// first call): atomics win on stencil models (LV), sorted gathers win when many data points hit few targets
// (rocket's shared step variable, ACOPF bus rows).
void prod_setup(Handle &h, bool hess) {
    const Model &m = *h.m;
    // sorted lists describe the COO this process evaluates: the whole model, or (sharded) the local slice
    if (h.world != 1 && !h.coo_local) throw BadInput("sorted products of a sharded model need the local-slice COO (exa_set_coo_local)");
    const int64_t nnzj = h.lnnzj, nnzh = h.lnnzh;
    h.cbuf.ensure(8 * (size_t)std::max<int64_t>(std::max(nnzj, nnzh), 1));
    if (!hess && !h.prod_ready_j) {
        h.pjrows.ensure(8 * (size_t)std::max<int64_t>(nnzj, 1)); h.pjcols.ensure(8 * (size_t)std::max<int64_t>(nnzj, 1));
        do_struct(h, false, true, h.pjrows.p, h.pjcols.p);
        build_sorted_index(h.jbycol, (const int64_t *)h.pjcols.p, nnzj, m.nvar, h.stream);
        attach_other(h.jbycol, (const int64_t *)h.pjrows.p, nullptr, nullptr, false, m.ncon, h.stream);
        h.prod_ready_j = true;
    }
    if (hess && !h.prod_ready_h) {
        h.phrows.ensure(8 * (size_t)std::max<int64_t>(nnzh, 1)); h.phcols.ensure(8 * (size_t)std::max<int64_t>(nnzh, 1));
        do_struct(h, true, true, h.phrows.p, h.phcols.p);
        build_sorted_index(h.hbyrow, (const int64_t *)h.phrows.p, nnzh, m.nvar, h.stream);
        build_sorted_index(h.hbycol, (const int64_t *)h.phcols.p, nnzh, m.nvar, h.stream);
        const int64_t *r = (const int64_t *)h.phrows.p, *c = (const int64_t *)h.phcols.p;
        attach_other(h.hbyrow, c, r, c, false, m.nvar, h.stream);     // lower triangle incl. diagonal: gathers v[col]
        attach_other(h.hbycol, r, r, c, true, m.nvar, h.stream);      // its transpose, off-diagonal only: gathers v[row]
        h.prod_ready_h = true;
    }
}
void drop_sorted(Handle &h, bool hess) {
    if (!hess) { h.jbycol.release(); h.pjrows.release(); h.pjcols.release(); h.prod_ready_j = false; }
    else { h.hbyrow.release(); h.hbycol.release(); h.phrows.release(); h.phcols.release(); h.prod_ready_h = false; }
}
void do_jtprod_sorted(Handle &h, const double *x, const double *v, double *Jtv) {
    do_jac(h, x, (double *)h.cbuf.p);
    spmv_gather(h.jbycol, (const double *)h.cbuf.p, (const int64_t *)h.pjrows.p, nullptr, nullptr, false, v, Jtv, false, h.stream);
}
void do_hprod_sorted(Handle &h, const double *x, const double *y, const double *v, double sigma, double *Hv) {
    do_hess(h, x, y, sigma, (double *)h.cbuf.p);
    const int64_t *r = (const int64_t *)h.phrows.p, *c = (const int64_t *)h.phcols.p;
    spmv_gather(h.hbyrow, (const double *)h.cbuf.p, c, r, c, false, v, Hv, false, h.stream);      // lower triangle incl. diagonal
    spmv_gather(h.hbycol, (const double *)h.cbuf.p, r, r, c, true, v, Hv, true, h.stream);        // its transpose, off-diagonal only
}
template <class A, class B>
int pick_faster(Handle &h, A &&atomics, B &&sorted) {
    float t[2] = {0.f, 0.f};
    for (int which = 0; which < 2; which++) {
        for (int rep = 0; rep < 4; rep++) {
            if (rep == 1) HIPCHK(hipEventRecord(h.ev0, h.stream));
            if (which == 0) atomics(); else sorted();
        }
        HIPCHK(hipEventRecord(h.ev1, h.stream));
        HIPCHK(hipEventSynchronize(h.ev1));
        HIPCHK(hipEventElapsedTime(&t[which], h.ev0, h.ev1));
    }
    return t[1] < t[0] ? 1 : 0;
}
void do_struct(Handle &h, bool hess, bool wide, void *rows, void *cols) {
    const void *P = h.dP.p;
    void *a[] = {&P, &rows, &cols};
    hipFunction_t f = hess ? (wide ? h.f_hs64 : h.f_hs32) : (wide ? h.f_js64 : h.f_js32);
    launch(h, f, h.grid[hess ? CB_HSTRUCT : CB_JSTRUCT], kBlock, a);
}

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
void do_window(Handle &h, int wk, const double *x, const double *y, const double *v, double sigma, double *vals, int64_t w0 = 0, int64_t w1 = -1) {
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


// The HIP "current device" is per host thread; a model lives on the device that was current in exa_create.  A call from
// a thread whose current device is another one (a Julia task that migrated, a worker thread that never called
// hipSetDevice) would allocate its scratch buffers on the wrong GPU: every device call runs with the model's device
// current and puts the caller's back.
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    explicit DeviceScope(int want) {
        if (want < 0) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != want) switched = hipSetDevice(want) == hipSuccess;
    }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
};

template <class F>
int guard(int id, bool need_device, F &&f) {
    Handle *h = get(id);
    if (!h) return 1;
    if (need_device && !h->on_device) { g_err = "model was planned without a device (exa_plan_only)"; return 1; }
    try {
        DeviceScope scope(h->on_device ? h->device : -1);
        f(*h);
        return 0;
    } catch (const BadInput &e) {
        g_err = e.what();
        return 1;
    } catch (const std::exception &e) {
        g_err = e.what();
        return 2;
    } catch (...) {
        g_err = "unknown error";
        return 2;
    }
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
        if (note_lookup(source_key(h->gen.source)) == "loopfree" && h->gen.source.find("// scatter kernels without loops") == std::string::npos) {       // decided where this module was first compiled
            h->first_key = source_key(h->gen.source);
            h->loopfree_scatter = true;
            h->gen = generate_module(*h->m, true);
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

}  // namespace

namespace exa {
int create_model(const exa_model_desc_t *desc, int *id_out, bool device) { return create(desc, id_out, device); }
int attach_blocks(int id, std::vector<BlockInfo> blocks) {
    Handle *h = get(id);
    if (!h) return 1;
    h->blocks = std::move(blocks);
    return 0;
}
void set_last_error(const std::string &text) { g_err = text; }
}  // namespace exa

extern "C" {

int exa_abi_version(void) { return EXAHIP_ABI_VERSION; }
const char *exa_last_error(void) { return g_err.c_str(); }

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

static int clamp32(int64_t v) { return v > 0x7fffffffLL ? -1 : (int)v; }
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

// new shard and/or COO addressing: the parameter table, and everything derived from the local COO, start over
static void reshard(Handle &h, int rank, int world, bool coo_local) {
    if (h.on_device) HIPCHK(hipStreamSynchronize(h.stream));
    if ((h.nccl || h.hook) && (rank != h.rank || world != h.world)) throw BadInput("the model's communicator fixes its shard (exa_comm_free first)");
    h.rank = rank; h.world = world; h.coo_local = coo_local;
    fill_params(h);
    if (h.on_device) {
        drop_sorted(h, false); drop_sorted(h, true);
        for (auto &q : h.pl) { q.idx.release(); q.ready = false; }
        if (h.compressed) {
            h.cj.release(); h.ch.release(); h.compressed = false;
            for (Handle::Window *w : {&h.wj, &h.wh}) { w->ok = false; w->why.clear(); }
            h.sj.ok = h.sh.ok = false;
        }
        if (g_eager_setup) g_eager_setup(h);
    }
}
int exa_set_stream(int id, void *s) { return guard(id, true, [&](Handle &h) { h.stream = (hipStream_t)s; }); }
int exa_set_shard(int id, int rank, int world) {
    if (world < 1 || rank < 0 || rank >= world) return 1;
    return guard(id, false, [&](Handle &h) {
        reshard(h, rank, world, h.coo_local);
    });
}
int exa_set_value(int id, int64_t offset, const double *vals, int64_t len) {
    Handle *hh = get(id);
    if (!hh || !vals || offset < 0 || len < 0 || offset + len > hh->m->npar) return 1;
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
    if (!hh || !dev_vals || offset < 0 || len < 0 || offset + len > hh->m->npar) return 1;
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
    if (!hh || !vals || offset < 0 || len < 0 || offset + len > hh->m->npar) return 1;
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
static int value_block(int id, int k, double *get_to, const double *set_from, int len) {
    Handle *h = get(id);
    if (!h || k < 0 || k >= (int)h->blocks.size() || h->blocks[k].kind != 2 || (!get_to && !set_from)) return 1;
    const BlockInfo &b = h->blocks[k];
    if ((int64_t)len != b.length) return 3;
    return get_to ? exa_get_value(id, b.offset, get_to, len) : exa_set_value(id, b.offset, set_from, len);
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
// Which implementation a product runs is a property of the model fixed BEFORE the call: explicit (exa_set_product_mode),
// measured once by exa_tune and persisted next to the cached module, or — undecided and never tuned — the atomics of the
// sweep.  Callbacks never measure and never synchronise.
// Owner pull: the variable -> item-slot lists.  Built once per shard geometry (here: unsharded models only — a rank of a sharded
// model would need the items of ALL data points that touch its variables, like the windows; the atomics + all-reduce stay there).
static bool pull_possible(const Handle &h, bool hess) {
    const Handle::Pull &q = h.pl[hess ? 1 : 0];
    return h.on_device && h.world == 1 && q.planned && q.fpull && q.why.empty();
}
static void pull_setup(Handle &h, bool hess) {
    Handle::Pull &q = h.pl[hess ? 1 : 0];
    if (q.ready) return;
    const Model &m = *h.m;
    const ParamLayout &L = h.gen.layout;
    const int cb = hess ? CB_HPROD : CB_JTPROD;
    std::vector<int64_t> first(L.groups[cb].size(), 0);
    int64_t total = 0;
    for (size_t g = 0; g < L.groups[cb].size(); g++) {
        const auto &pp = L.pat[L.groups[cb][g].front()];
        first[g] = total;
        total += (int64_t)q.nitems[g] * (h.P[pp.hi] - h.P[pp.lo]);
    }
    if (total <= 0 || total > 0xfffffff0LL) { q.why = "no items, or more than 2^32 of them"; return; }
    q.total = total;
    q.first.ensure(8 * std::max<size_t>(first.size(), 1));
    HIPCHK(hipMemcpy(q.first.p, first.data(), 8 * first.size(), hipMemcpyHostToDevice));
    DevBuf keys, zx, zy;
    struct Rel { DevBuf &a, &b, &c; ~Rel() { a.release(); b.release(); c.release(); } } rel{keys, zx, zy};
    keys.ensure(8 * (size_t)total);
    // (the key functions hold the whole body of their group; the compiler drops the value part — x, y, v are handed valid zero
    // vectors all the same)
    zx.ensure(8 * (size_t)std::max<int64_t>(m.nvar, 1)); zy.ensure(8 * (size_t)std::max<int64_t>(m.ncon, 1));
    HIPCHK(hipMemsetAsync(zx.p, 0, zx.bytes, h.stream)); HIPCHK(hipMemsetAsync(zy.p, 0, zy.bytes, h.stream));
    const void *P = h.dP.p, *th = h.dtheta.p, *xz = zx.p, *yz = zy.p, *vz = hess ? zx.p : zy.p, *fp = q.first.p;
    void *kp = keys.p;
    double sigma = 1.0;
    if (hess) { void *a[] = {&P, &xz, &yz, &th, &vz, &sigma, &kp, &fp}; launch(h, q.fkeys, h.grid[cb], kBlock, a); }
    else { void *a[] = {&P, &xz, &th, &vz, &kp, &fp}; launch(h, q.fkeys, h.grid[cb], kBlock, a); }
    build_sorted_index(q.idx, (const int64_t *)keys.p, total, m.nvar, h.stream);
    HIPCHK(hipStreamSynchronize(h.stream));
    if (q.idx.nlong > 0) {       // a variable collecting more than 512 contributions (a slack shared by every point): one thread would walk them all
        q.idx.release();
        q.why = "a variable collects more than 512 contributions (the atomics / the sorted gather handle it cooperatively)";
        return;
    }
    q.ready = true;
}
static void do_pull(Handle &h, bool hess, const double *x, const double *y, const double *v, double sigma, double *out) {
    Handle::Pull &q = h.pl[hess ? 1 : 0];
    const void *P = h.dP.p, *th = h.dtheta.p, *ptr = q.idx.ptr, *perm = q.idx.perm, *fp = q.first.p;
    int64_t vb = 0, ve = h.m->nvar;
    const int64_t grid = (ve - vb + kBlock - 1) / kBlock;
    if (hess) { void *a[] = {&P, &x, &y, &th, &v, &sigma, &out, &ptr, &perm, &fp, &vb, &ve}; launch(h, q.fpull, grid, kBlock, a); }
    else { void *a[] = {&P, &x, &th, &v, &out, &ptr, &perm, &fp, &vb, &ve}; launch(h, q.fpull, grid, kBlock, a); }
}
static bool sorted_possible(Handle &h, bool hess) {
    const int64_t nnz = hess ? h.lnnzh : h.lnnzj;
    return (h.world == 1 || h.coo_local) && nnz > 0;
}
// Owner-computes windows: possible when the model's targets are range-affine (plan_products) and the module is loaded; a
// SHARDED model takes them only when nothing is left to the tail kernel (no tiny patterns, no entry every point adds to):
// those belong to all ranks at once.
static bool window_possible(Handle &h, bool hess) {
    const Handle::Window &w = h.wp[hess ? 1 : 0];
    return w.ok && (h.world == 1 || (w.nx == 0 && !w.has_shared));
}
static int resolve_mode(Handle &h, bool hess) {
    int &mode = hess ? h.hp_mode : h.jt_mode;
    if (mode < 0) {
        int v = -1;
        const bool tuned = tune_lookup(source_key(h.gen.source), tune_signature(h, hess ? "hprod" : "jtprod"), &v) && v >= 0 && v <= 3;
        if (tuned && ((v == 1 && sorted_possible(h, hess)) || (v == 2 && window_possible(h, hess)) || (v == 3 && pull_possible(h, hess)) || v == 0)) mode = v;
        // undecided and never tuned: the windows where the model has them (rocket nh = 1e6: J'v 0.040 against 0.060 ms for the
        // atomics, Hv 0.053 against 0.067 — with the all-points entry summed inside the window kernel; as a separate
        // evaluation pass it was 0.079)
        else mode = window_possible(h, hess) ? 2 : 0;
    }
    if (mode == 2 && !window_possible(h, hess)) return 0;
    if (mode == 3) {
        if (!pull_possible(h, hess)) return 0;
        if (!h.pl[hess ? 1 : 0].ready) { if (capturing(h)) return 0; pull_setup(h, hess); if (!h.pl[hess ? 1 : 0].ready) return 0; }
        return 3;
    }
    if (mode == 1 && !sorted_possible(h, hess)) return 0;      // sharded at global positions: nothing to sort locally
    if (mode == 1 && !(hess ? h.prod_ready_h : h.prod_ready_j)) { if (capturing(h)) return 0; prod_setup(h, hess); }
    return mode;
}
// what the persisted decisions need, built at model build / reshard instead of inside the first callback
static void eager_setup(Handle &h) {
    if (!h.on_device) return;
    const int g = h.grad_mode, jt = h.jt_mode, hp = h.hp_mode;
    (void)resolve_grad_mode(h);
    (void)resolve_mode(h, false);
    (void)resolve_mode(h, true);
    h.grad_mode = g; h.jt_mode = jt; h.hp_mode = hp;          // (still "undecided" for exa_get_*_mode until a call resolves them)
}
static const bool g_eager_registered = (g_eager_setup = eager_setup, true);
static void run_product_window(Handle &h, bool hess, const double *x, const double *y, const double *v, double w, double *out) {
    Handle::Window &win = h.wp[hess ? 1 : 0];
    if (h.world == 1) { do_window(h, hess ? WK_HPROD : WK_JTPROD, x, y, v, w, out); return; }
    int64_t w0, w1;
    owned_windows(h, win, h.rank, &w0, &w1);
    do_window(h, hess ? WK_HPROD : WK_JTPROD, x, y, v, w, out, w0, w1);
    allgatherv(h, out, window_pieces(h, win));
}
static void run_jtprod(Handle &h, const double *x, const double *v, double *Jtv) {
    const int mode = resolve_mode(h, false);
    if (mode == 2) { run_product_window(h, false, x, nullptr, v, 0.0, Jtv); return; }
    if (mode == 3) { do_pull(h, false, x, nullptr, v, 0.0, Jtv); return; }        // (unsharded: nothing to complete)
    if (mode == 1) do_jtprod_sorted(h, x, v, Jtv); else do_jtprod(h, x, v, Jtv);
    allreduce(h, Jtv, h.m->nvar);
}
static void run_hprod(Handle &h, const double *x, const double *y, const double *v, double w, double *Hv) {
    int mode = resolve_mode(h, true);
    if ((mode == 2 || mode == 3) && !y && h.m->ncon > 0) mode = 0;      // objective only: the window / pull kernels evaluate every pattern; the atomics launch the objective groups alone
    if (mode == 2) { run_product_window(h, true, x, y, v, w, Hv); return; }
    if (mode == 3) { do_pull(h, true, x, y, v, w, Hv); return; }
    if (mode == 1) do_hprod_sorted(h, x, y, v, w, Hv); else do_hprod(h, x, y, v, w, Hv);
    allreduce(h, Hv, h.m->nvar);
}
int exa_jtprod(int id, const double *x, const double *v, double *Jtv) {
    if (!x || !Jtv) return 1;
    return guard(id, true, [&](Handle &h) { if (h.m->ncon && !v) throw BadInput("null input"); run_jtprod(h, x, v, Jtv); });
}
int exa_hprod(int id, const double *x, const double *y, const double *v, double w, double *Hv) {
    if (!x || !v || !Hv) return 1;
    return guard(id, true, [&](Handle &h) {
        run_hprod(h, x, y, v, w, Hv);
    });
}
/* 0 = atomics inside the sweep, 1 = COO + sorted gather, 2 = owner-computes windows, -1 = undecided (default): the decision
 * exa_tune persisted for this module / device / sizes if there is one, else the windows where the model has them, else 0 */
int exa_set_product_mode(int id, int jtprod_mode, int hprod_mode) {
    if (jtprod_mode < -1 || jtprod_mode > 3 || hprod_mode < -1 || hprod_mode > 3) return 1;
    return guard(id, true, [&](Handle &h) {
        for (int k = 0; k < 2; k++) {
            if ((k ? hprod_mode : jtprod_mode) != 3) continue;
            if (pull_possible(h, k != 0)) pull_setup(h, k != 0);
            if (!h.pl[k].ready) throw BadInput(std::string(k ? "Hv" : "J'v") + " has no owner-pull lists on this model: " +
                                               (h.pl[k].why.empty() ? (h.world > 1 ? "sharded model" : h.wp[k].why.empty() ? "not planned" : "the model has owner-computes windows or no data-indexed target") : h.pl[k].why));
        }
        if (jtprod_mode == 2 && !window_possible(h, false)) throw BadInput("J'v has no owner-computes windows on this model: " + h.wp[0].why);
        if (hprod_mode == 2 && !window_possible(h, true)) throw BadInput("Hv has no owner-computes windows on this model: " + h.wp[1].why);
        if (jtprod_mode == 1) prod_setup(h, false);      // refuses a sharded model at global positions (status 1)
        if (hprod_mode == 1) prod_setup(h, true);
        h.jt_mode = jtprod_mode; h.hp_mode = hprod_mode;
    });
}
/* What exa_jtprod (hess = 0) / exa_hprod (hess = 1) run: 0 atomics, 1 sorted gather, 2 owner-computes windows (resolved as a
 * call would resolve it, without building anything); buf <- the kernel shape of the windows or why the model has none. */
// the implementation a call WOULD run (explicit mode, else the persisted exa_tune decision, else the windows where the model
// has them), without building anything: shared by exa_product_info and exa_shard_layout so that the two cannot disagree
static int product_mode_query(Handle &h, bool hess) {
    const Handle::Window &w = h.wp[hess ? 1 : 0];
    const int mode = hess ? h.hp_mode : h.jt_mode;
    if (mode >= 0) return (mode == 2 && !window_possible(h, hess)) || (mode == 3 && !pull_possible(h, hess)) ? 0 : mode;
    int v = -1;
    if (h.on_device && tune_lookup(source_key(h.gen.source), tune_signature(h, hess ? "hprod" : "jtprod"), &v) && v >= 0 && v <= 3 &&
        (v != 2 || window_possible(h, hess)) && (v != 1 || sorted_possible(h, hess)) && (v != 3 || pull_possible(h, hess))) return v;
    return (h.on_device ? window_possible(h, hess) : w.planned) ? 2 : 0;
}
int exa_product_info(int id, int hess, char *buf, int cap) {
    Handle *h = get(id);
    if (!h) return -1;
    const Handle::Window &w = h->wp[hess ? 1 : 0];
    const Handle::Pull &q = h->pl[hess ? 1 : 0];
    std::string text = w.why;
    if (q.planned || !q.why.empty()) {
        int tot = 0;
        for (int n : q.nitems) tot += n;
        text += q.why.empty() ? "; owner pull available (" + std::to_string(tot) + " item functions)" : "; no owner pull: " + q.why;
    }
    if (buf && cap > 0) snprintf(buf, (size_t)cap, "%s", text.c_str());
    return product_mode_query(*h, hess != 0);
}
/* grad!: 0 = gathered (affine patterns) + FP64 atomics (data-indexed ones), 1 = gradient COO + sorted gather (the reference's
 * scheme: deterministic, and immune to many data points sharing a few variables), -1 = undecided: the persisted exa_tune
 * decision if there is one, else 0.  A sharded model always runs 0. */
int exa_set_grad_mode(int id, int mode) {
    if (mode < -1 || mode > 1) return 1;
    return guard(id, true, [&](Handle &h) {
        if (mode == 1 && grad_sorted_possible(h)) grad_setup(h);
        h.grad_mode = mode;
    });
}
/* All three at once: on = grad!, jtprod and hprod by sorted gather wherever the model allows it (bit-reproducible run to
 * run, like every other callback); off = back to undecided (-1: the persisted exa_tune decisions, else atomics). */
int exa_set_deterministic(int id, int on) {
    return guard(id, true, [&](Handle &h) {
        if (on) {
            if (grad_sorted_possible(h)) { grad_setup(h); h.grad_mode = 1; }
            // (the owner-computes windows and the owner pull are deterministic too: a fixed order of additions, no atomics)
            for (int k = 0; k < 2; k++) {
                int &mode = k ? h.hp_mode : h.jt_mode;
                if (window_possible(h, k != 0)) { mode = 2; continue; }
                if (pull_possible(h, k != 0)) { pull_setup(h, k != 0); if (h.pl[k].ready) { mode = 3; continue; } }
                if (sorted_possible(h, k != 0)) { prod_setup(h, k != 0); mode = 1; }
            }
        } else { h.grad_mode = -1; h.jt_mode = -1; h.hp_mode = -1; }
    });
}
int exa_get_grad_mode(int id, int *mode) {
    Handle *h = get(id);
    if (!h || !mode) return 1;
    *mode = h->grad_mode;
    return 0;
}
int exa_get_product_mode(int id, int *jtprod_mode, int *hprod_mode) {
    Handle *h = get(id);
    if (!h || !jtprod_mode || !hprod_mode) return 1;
    *jtprod_mode = h->jt_mode; *hprod_mode = h->hp_mode;
    return 0;
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
int exa_jtprod_host(int id, const double *x, const double *v, double *Jtv) {
    if (!x || !Jtv) return 1;
    return guard(id, true, [&](Handle &h) {
        const size_t n = 8 * (size_t)h.m->nvar;
        h2d(h, h.sx, x, n);
        if (h.m->ncon) { if (!v) throw std::runtime_error("null input"); h2d(h, h.sv, v, 8 * (size_t)h.m->ncon); }
        else h.sv.ensure(8);
        h.sout.ensure(n);
        zero_if_sharded(h, h.sout.p, n);
        run_jtprod(h, (const double *)h.sx.p, (const double *)h.sv.p, (double *)h.sout.p);
        d2h(h, Jtv, h.sout.p, n);
    });
}
int exa_hprod_host(int id, const double *x, const double *y, const double *v, double w, double *Hv) {
    if (!x || !v || !Hv) return 1;
    return guard(id, true, [&](Handle &h) {
        const size_t n = 8 * (size_t)h.m->nvar;
        h2d(h, h.sx, x, n);
        h2d(h, h.sv, v, n);
        if (h.m->ncon && y) h2d(h, h.sy, y, 8 * (size_t)h.m->ncon);
        else h.sy.ensure(8);
        h.sout.ensure(n);
        zero_if_sharded(h, h.sout.p, n);
        run_hprod(h, (const double *)h.sx.p, h.m->ncon && !y ? nullptr : (const double *)h.sy.p, (const double *)h.sv.p, w, (double *)h.sout.p);
        d2h(h, Hv, h.sout.p, n);
    });
}
static int struct_host(int id, bool hess, bool wide, void *r, void *c) {
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
int exa_jac_structure_host(int id, int32_t *r, int32_t *c) { return struct_host(id, false, false, r, c); }
int exa_hess_structure_host(int id, int32_t *r, int32_t *c) { return struct_host(id, true, false, r, c); }
int exa_jac_structure64_host(int id, int64_t *r, int64_t *c) { return struct_host(id, false, true, r, c); }
int exa_hess_structure64_host(int id, int64_t *r, int64_t *c) { return struct_host(id, true, true, r, c); }

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
static int cstruct(int id, bool hess, bool wide, void *r, void *c) {
    return guard(id, true, [&](Handle &h) {
        if (!h.compressed) throw BadInput("exa_compress has not been called");
        const CompressedCOO &cc = hess ? h.ch : h.cj;
        if (!wide && cc.cnnz > 0x7fffffffLL) throw std::runtime_error("nnz exceeds int32");
        compressed_structure(cc, r, c, wide, h.stream);
    });
}
int exa_cjac_structure(int id, int32_t *r, int32_t *c) { return cstruct(id, false, false, r, c); }
int exa_chess_structure(int id, int32_t *r, int32_t *c) { return cstruct(id, true, false, r, c); }
int exa_cjac_structure64(int id, int64_t *r, int64_t *c) { return cstruct(id, false, true, r, c); }
int exa_chess_structure64(int id, int64_t *r, int64_t *c) { return cstruct(id, true, true, r, c); }
static int ccsc(int id, bool hess, int64_t *colptr, int64_t *rowval) {
    if (!colptr || !rowval) return 1;
    return guard(id, true, [&](Handle &h) {
        if (!h.compressed) throw BadInput("exa_compress has not been called");
        compressed_csc(hess ? h.ch : h.cj, h.m->nvar, colptr, rowval, h.stream);
    });
}
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
        if (h.wh.ok) { do_window(h, WK_CHESS, x, y, nullptr, w, vals); return; }
        if (h.sh.ok) { do_scatter(h, true, x, y, w, vals); return; }
        do_hess(h, x, y, w, (double *)h.cbuf.p);
        compress_values(h.ch, (const double *)h.cbuf.p, vals, h.stream);
    });
}

// ---- measurement ------------------------------------------------------------------------------------------
int exa_time_callback(int id, int which, int reps, const double *x, const double *y, double w, double *out, float *ms_out) {
    if (reps < 1 || !ms_out || which < 0 || which > 5 || !x) return 1;   /* 0 obj 1 grad 2 cons 3 jac 4 hess 5 an (almost) empty launch: the floor */
    return guard(id, true, [&](Handle &h) {
        const Model &m = *h.m;
        if ((which == 1 && !out) || (which == 2 && m.ncon && !out) || (which == 3 && m.nnzj && !out) ||
            (which == 4 && ((m.nnzh && !out) || (m.ncon && !y))))
            throw BadInput("null pointer for a buffer the callback reads or writes");
        HIPCHK(hipEventRecord(h.ev0, h.stream));
        for (int r = 0; r < reps; r++) {
            switch (which) {
            case 0: do_obj(h, x, (double *)h.dobj.p); break;
            case 1: run_grad(h, x, out); break;
            case 2: do_cons(h, x, out); break;
            case 3: do_jac(h, x, out); break;
            case 4: do_hess(h, x, y, w, out); break;
            case 5: zero_fill(h, h.dobj.p, 1); break;      // one workgroup writing one double: what a launch costs on this stream
            }
        }
        HIPCHK(hipEventRecord(h.ev1, h.stream));
        HIPCHK(hipEventSynchronize(h.ev1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, h.ev0, h.ev1));
        *ms_out = ms / (float)reps;
    });
}
/* which: 3 jac, 4 hess (as exa_time_callback), 2 cons, 5 fused.  -1 = not measured yet, 0 sequential, 1 interleaved */
int exa_block_order(int id, int which) {
    Handle *h = get(id);
    if (!h) return -2;
    const int cb = which == 3 ? CB_JAC : which == 4 ? (h->hess_variant >= 1 ? CB_HESSC : CB_HESS) : which == 2 ? CB_CONS : which == 5 ? CB_FUSED : -1;
    return cb < 0 ? -2 : h->order[cb];
}
/* which hess_coord! kernel runs: 0 exa_hess (one tile per workgroup), 1 exa_hesscl (chained over groups of co-indexed
 * patterns, software-pipelined, x staged through LDS), 2 exa_hessc (the same without the staging: chosen, or what 1 falls
 * back to when the model / this shard does not fit the staging), -1 bad id */
int exa_hess_variant(int id) {
    Handle *h = get(id);
    if (!h) return -1;
    return h->hess_variant == 1 && !(h->f_hesscl && h->stage_ok) ? 2 : h->hess_variant;
}
int exa_sync(int id) { return guard(id, true, [&](Handle &h) { HIPCHK(hipStreamSynchronize(h.stream)); }); }

// ---- explicit tuning (the only place that measures; callbacks never do) ------------------------------------------------
int exa_tune(int id, int what, const double *x, const double *y) {
    if (what < 0 || what > 7) return 1;
    return guard(id, true, [&](Handle &h) {
        const Model &m = *h.m;
        struct Tmp { DevBuf b[8]; ~Tmp() { for (auto &q : b) q.release(); } } t;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(h.stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) throw BadInput("exa_tune cannot run inside a stream capture");
        if (!x) {
            std::vector<double> x0 = m.x0;
            if (x0.empty()) x0.assign((size_t)m.nvar, 0.0);
            t.b[0].ensure(8 * x0.size());
            HIPCHK(hipMemcpy(t.b[0].p, x0.data(), 8 * x0.size(), hipMemcpyHostToDevice));
            x = (const double *)t.b[0].p;
        }
        if (!y && m.ncon) {
            std::vector<double> ones((size_t)m.ncon, 1.0);
            t.b[1].ensure(8 * ones.size());
            HIPCHK(hipMemcpy(t.b[1].p, ones.data(), 8 * ones.size(), hipMemcpyHostToDevice));
            y = (const double *)t.b[1].p;
        }
        const bool reduce = h.reduce;
        h.reduce = false;                      // ranks measure on their own: no collective inside a measurement
        struct Restore { Handle &h; bool r; ~Restore() { h.reduce = r; } } restore{h, reduce};
        const double sigma = 0.5;
        double *c = nullptr, *jv = nullptr, *hv = nullptr, *obj = (double *)h.dobj.p, *g = nullptr;
        auto need = [&](int k, int64_t n) { t.b[k].ensure(8 * (size_t)std::max<int64_t>(n, 1)); return (double *)t.b[k].p; };
        if (what & 1) {
            if (h.norders[CB_CONS] > 1) { c = need(2, m.ncon); (void)tune_order(h, CB_CONS, [&] { do_cons(h, x, c); }); }
            if (h.norders[CB_JAC] > 1) { jv = need(3, h.lnnzj); (void)tune_order(h, CB_JAC, [&] { do_jac(h, x, jv); }); }
            if (h.f_hessc && h.lnnzh > 0) {
                // the two hess_coord! kernels, each at the better of its block orders
                hv = need(4, h.lnnzh);
                h.hess_variant = 0;
                const float t0 = tune_order(h, CB_HESS, [&] { do_hess(h, x, y, sigma, hv); });
                h.hess_variant = 2;
                const float t1 = tune_order(h, CB_HESSC, [&] { do_hess(h, x, y, sigma, hv); });
                h.hess_variant = t1 < t0 ? 2 : 0;
                if (h.f_hesscl && h.stage_ok) {
                    h.hess_variant = 1;
                    const float t2 = tune_order(h, CB_HESSC, [&] { do_hess(h, x, y, sigma, hv); });
                    if (!(t2 < std::min(t0, t1))) h.hess_variant = t1 < t0 ? 2 : 0;
                }
                tune_store(source_key(h.gen.source), tune_signature(h, "hessvariant"), h.hess_variant);
            } else if (h.norders[CB_HESS] > 1) { hv = need(4, h.lnnzh); tune_order(h, CB_HESS, [&] { do_hess(h, x, y, sigma, hv); }); }
            if (h.norders[CB_FUSED] > 1) {
                c = need(2, m.ncon); jv = need(3, h.lnnzj); hv = need(4, h.lnnzh);
                (void)tune_order(h, CB_FUSED, [&] { do_fused(h, x, y, sigma, obj, c, jv, hv); });
            }
        }
        if (what & 2) {
            g = need(5, m.nvar);
            // beyond 3e8 entries the sorted lists' memory (16 B per entry + the COO itself) is not worth a trial
            for (int hess = 0; hess < 2; hess++) {
                int &mode = hess ? h.hp_mode : h.jt_mode;
                const int64_t nnz = hess ? h.lnnzh : h.lnnzj;
                int best = 0;
                if (sorted_possible(h, hess != 0) && nnz <= 300000000LL) {
                    prod_setup(h, hess != 0);
                    if (hess) best = pick_faster(h, [&] { do_hprod(h, x, y, x, sigma, g); }, [&] { do_hprod_sorted(h, x, y, x, sigma, g); });
                    else best = pick_faster(h, [&] { do_jtprod(h, x, y, g); }, [&] { do_jtprod_sorted(h, x, y, g); });
                    if (best == 0) drop_sorted(h, hess != 0);
                }
                if (window_possible(h, hess != 0)) {
                    // the owner-computes windows against the winner so far
                    const int other = best;
                    auto base = [&] { if (hess) { if (other) do_hprod_sorted(h, x, y, x, sigma, g); else do_hprod(h, x, y, x, sigma, g); }
                                      else { if (other) do_jtprod_sorted(h, x, y, g); else do_jtprod(h, x, y, g); } };
                    auto wnd = [&] { if (hess) run_product_window(h, true, x, y, x, sigma, g); else run_product_window(h, false, x, nullptr, y, 0.0, g); };
                    if (pick_faster(h, base, wnd) == 1) { best = 2; if (other == 1) drop_sorted(h, hess != 0); }
                }
                if (pull_possible(h, hess != 0)) {
                    // the owner pull against the winner so far
                    pull_setup(h, hess != 0);
                    if (h.pl[hess].ready) {
                        const int other = best;
                        auto base = [&] { if (hess) { if (other == 1) do_hprod_sorted(h, x, y, x, sigma, g); else do_hprod(h, x, y, x, sigma, g); }
                                          else { if (other == 1) do_jtprod_sorted(h, x, y, g); else do_jtprod(h, x, y, g); } };
                        auto pull = [&] { if (hess) do_pull(h, true, x, y, x, sigma, g); else do_pull(h, false, x, nullptr, y, 0.0, g); };
                        if (other != 2 && pick_faster(h, base, pull) == 1) { best = 3; if (other == 1) drop_sorted(h, hess != 0); }
                        else { h.pl[hess].idx.release(); h.pl[hess].ready = false; }
                    }
                }
                mode = best;
                tune_store(source_key(h.gen.source), tune_signature(h, hess ? "hprod" : "jtprod"), best);
            }
        }
        if (what & 4) {
            // grad!: only worth a trial when some objective pattern scatters through a data index (the gathered patterns
            // of a stencil model are already a plain coalesced store)
            int best = 0;
            if (!h.gen.layout.active[CB_GRAD].empty() && grad_sorted_possible(h) && m.nnzg <= 300000000LL) {
                g = need(5, m.nvar);
                grad_setup(h);
                best = pick_faster(h, [&] { do_grad(h, x, g); }, [&] { do_grad_sorted(h, x, g); });
                if (best == 0) { h.gbyvar.release(); h.gbuf.release(); h.grad_ready = false; }
            }
            h.grad_mode = best;
            tune_store(source_key(h.gen.source), tune_signature(h, "grad"), best);
        }
        HIPCHK(hipStreamSynchronize(h.stream));
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

// ---- multi-GPU: collectives behind the ABI ----------------------------------------------------------------
int exa_comm_unique_id(void *out128) {
    if (!out128) return 1;
    try { rccl_unique_id(out128); return 0; } catch (const std::exception &e) { g_err = e.what(); return 2; }
}
int exa_comm_init(int id, int rank, int world, const void *unique_id128) {
    if (!unique_id128 || world < 1 || rank < 0 || rank >= world) return 1;
    return guard(id, true, [&](Handle &h) {
        if (h.nccl || h.hook) throw BadInput("the model already has a communicator (exa_comm_free first)");
        // the communicator first: if it cannot be created (librccl missing, init failure) the model stays as it was — not
        // sharded without a communicator, returning partial results
        void *comm = rccl_comm_init(rank, world, unique_id128);    // collective over all ranks; on the current HIP device
        try { reshard(h, rank, world, h.coo_local); } catch (...) { try { rccl_comm_destroy(comm); } catch (...) {} throw; }
        h.nccl = comm;
        h.nccl_owned = true;
    });
}
int exa_comm_attach(int id, void *nccl_comm) {
    if (!nccl_comm) return 1;
    return guard(id, true, [&](Handle &h) {
        if (h.nccl || h.hook) throw BadInput("the model already has a communicator (exa_comm_free first)");
        int rank = 0, world = 1;
        rccl_comm_shape(nccl_comm, &rank, &world);
        reshard(h, rank, world, h.coo_local);
        h.nccl = nccl_comm;
        h.nccl_owned = false;
    });
}
int exa_comm_hook(int id, int rank, int world, exa_allreduce_fn fn, void *ctx) {
    if (!fn || world < 1 || rank < 0 || rank >= world) return 1;
    return guard(id, false, [&](Handle &h) {
        if (h.nccl || h.hook) throw BadInput("the model already has a communicator (exa_comm_free first)");
        reshard(h, rank, world, h.coo_local);
        h.hook = fn; h.hook_ctx = ctx;
    });
}
int exa_comm_free(int id) {
    return guard(id, false, [&](Handle &h) {
        if (h.on_device) HIPCHK(hipStreamSynchronize(h.stream));
        if (h.nccl && h.nccl_owned) rccl_comm_destroy(h.nccl);
        h.nccl = nullptr; h.nccl_owned = false; h.hook = nullptr; h.hook_ctx = nullptr;
    });
}
int exa_comm_info(int id, int *rank, int *world, int *kind) {
    Handle *h = get(id);
    if (!h) return 1;
    if (rank) *rank = h->rank;
    if (world) *world = h->world;
    if (world && h->nccl) { try { *world = rccl_comm_count(h->nccl); } catch (...) {} }      // what RCCL itself says (ncclCommCount): the ranks it saw
    if (kind) *kind = h->nccl ? 1 : (h->hook ? 2 : 0);
    return 0;
}
int exa_set_reduce(int id, int on) { return guard(id, false, [&](Handle &h) { h.reduce = on != 0; }); }
int exa_allreduce(int id, double *dev_buf, int64_t count) {
    if (!dev_buf || count < 0) return 1;
    return guard(id, true, [&](Handle &h) {
        if (!h.nccl && !h.hook) throw BadInput("the model has no communicator");
        const bool r = h.reduce;
        h.reduce = true;
        try { allreduce(h, dev_buf, count); } catch (...) { h.reduce = r; throw; }
        h.reduce = r;
    });
}
int exa_set_coo_local(int id, int on) { return guard(id, false, [&](Handle &h) { reshard(h, h.rank, h.world, on != 0); }); }
int64_t exa_local_nnzj64(int id) { Handle *h = get(id); return h ? h->lnnzj : -1; }
int64_t exa_local_nnzh64(int id) { Handle *h = get(id); return h ? h->lnnzh : -1; }
int exa_coo_slices(int id, int hess, int64_t *out) {
    Handle *h = get(id);
    if (!h || !out) return 1;
    const Model &m = *h->m;
    for (size_t k = 0; k < m.pats.size(); k++) {
        const Pattern &p = m.pats[k];
        const int64_t lo = part_lo(p.n, h->rank, h->world), hi = part_lo(p.n, h->rank + 1, h->world);
        const bool has = hess ? p.o2step > 0 : (p.kind != EXA_PAT_OBJ && p.o1step > 0);
        const int64_t step = hess ? p.o2step : p.o1step, o = hess ? p.o2 : p.o1, cnt = has ? step * (hi - lo) : 0;
        out[3 * k] = o + step * lo;                                                     // first global slot (0-based)
        out[3 * k + 1] = h->coo_local && h->world > 1 ? (hess ? h->lo2[k] : h->lo1[k]) : o + step * lo;   // where it is in the caller's buffer
        out[3 * k + 2] = cnt;
    }
    return 0;
}
int exa_shard_var_range(int id, int64_t *lo_out, int64_t *hi_out) {
    Handle *h = get(id);
    if (!h || !lo_out || !hi_out) return 1;
    const Model &m = *h->m;
    const ParamLayout &L = h->gen.layout;
    int64_t vmin = INT64_MAX, vmax = INT64_MIN;
    bool anywhere = false;
    auto add = [&](const Pattern &p, int64_t lo, int64_t hi) {
        if (hi <= lo || anywhere) return;
        int64_t a = 0, b = 0;
        if (!pattern_var_range(p, lo, hi, &a, &b)) { anywhere = true; return; }      // data-indexed: anywhere
        if (a <= b) { vmin = std::min(vmin, a); vmax = std::max(vmax, b); }
    };
    for (size_t k = 0; k < m.pats.size(); k++) {
        const Pattern &p = m.pats[k];
        if (p.n <= 0) continue;
        const int64_t lo = part_lo(p.n, h->rank, h->world), hi = part_lo(p.n, h->rank + 1, h->world);
        // a shard holding nothing of a pattern still re-reads one point of it (the branch-free loads of the chained
        // kernels clamp there): the last point before the shard, or point 0
        const int64_t lo_ = hi > lo ? lo : (hi > 0 ? hi - 1 : 0), hi_ = hi > lo ? hi : lo_ + 1;
        add(p, lo_, hi_);
        if (h->world == 1) continue;
        // owner-computes callbacks reach beyond the shard's own data points:
        //   cons_nln! / jprod in one launch: a row's owner evaluates the row's augmentation terms wherever they come from;
        if (p.kind == EXA_PAT_CONAUG && (h->cons1 || !h->on_device)) add(p, 0, p.n);
        //   grad!: the points of a gathered objective pattern that touch the variables this rank owns;
        if (std::find(L.pull.begin(), L.pull.end(), (int)k) != L.pull.end()) add(p, h->P[L.pat[k].qlo], h->P[L.pat[k].qhi]);
    }
    //   J'v / Hv by windows: the points that touch the windows this rank owns
    for (int wk : {WK_JTPROD, WK_HPROD}) {
        const Handle::Window &w = h->wp[wk - WK_JTPROD];
        if (h->world == 1 || !w.planned || w.nx || w.has_shared || w.hR.empty()) continue;
        int64_t w0, w1;
        owned_windows(*h, w, h->rank, &w0, &w1);
        const WindowMatrix &wm = h->pspec.mat[wk];
        std::vector<int> pk;            // R is [window][pass] (one space) or [block][pattern] (block-owned)
        for (const auto &wp : wm.pats) if (wm.nspaces == 0 || std::find(pk.begin(), pk.end(), wp.k) == pk.end()) pk.push_back(wp.k);
        for (size_t q = 0; q < pk.size(); q++) {
            int64_t lo = INT64_MAX, hi = INT64_MIN;
            for (int64_t j = w0; j < w1; j++) {
                const int32_t a = w.hR[(j * pk.size() + q) * 2], b = w.hR[(j * pk.size() + q) * 2 + 1];
                if (b > a) { lo = std::min<int64_t>(lo, a); hi = std::max<int64_t>(hi, b); }
            }
            if (hi > lo) add(m.pats[pk[q]], lo, hi);
        }
    }
    if (anywhere) { *lo_out = 0; *hi_out = m.nvar; return 0; }
    if (vmin > vmax) { *lo_out = 0; *hi_out = 0; return 0; }
    *lo_out = vmin - 1; *hi_out = vmax;       // 0-based [lo, hi)
    return 0;
}
/* How a sharded model's rank leaves the output of callback `which` when nothing completes it (no communicator, or
 * exa_set_reduce(id, 0)): 1 = OWNER PIECES — complete values in disjoint pieces (rows of its data points, variables / windows
 * it owns), nothing else written, an all-gather makes the vector whole; 0 = PARTIAL SUMS over the whole vector, an
 * all-reduce(sum) completes it.  which: 0 obj, 1 grad, 2 cons, 5 jprod, 6 jtprod, 7 hprod (3 jac / 4 hess: always pieces),
 * 8 the cons vector of exa_eval_fused / exa_eval_all (differs from 2 for models with non-linear augmentation terms).
 * -1 bad id / argument. */
int exa_shard_layout(int id, int which) {
    Handle *hh = get(id);
    if (!hh) return -1;
    Handle &h = *hh;
    switch (which) {
    case 0: return 0;
    case 1: return h.gen.layout.active[CB_GRAD].empty() && !h.gen.layout.pull.empty() ? 1 : 0;
    case 2: return rows_owner_complete(h) || !h.on_device ? 1 : 0;
    case 3: case 4: return 1;
    case 5: return (h.m->nconaug == 0 || (h.m->aug_linear && (h.cons1 || !h.on_device))) ? 1 : 0;
    case 6: case 7: {
        // owner pieces only when the call really runs the windows: an explicit or a tuned mode 0 / 1 (atomics, sorted gather)
        // leaves partial sums over the whole vector
        const bool hess = which == 7;
        const Handle::Window &w = h.wp[hess ? 1 : 0];
        const bool can = (h.on_device ? w.ok : w.planned) && w.nx == 0 && !w.has_shared;
        return can && product_mode_query(h, hess) == 2 ? 1 : 0;
    }
    // cons as exa_eval_fused / exa_eval_all leave it: rows complete on their owner only when the augmentation terms are linear
    // (added inside the sweep through the row lists); non-linear ones are partial sums there although exa_cons (which = 2)
    // completes the rows itself
    case 8: return h.m->nconaug == 0 || (h.m->aug_linear && (h.cons1 || !h.on_device)) ? 1 : 0;
    }
    return -1;
}
/* How the library completes (or a host layer should complete) the output of callback `which` of a sharded model: out <- up to cap
 * operations of 4 words — kind (0 in-place all-gather: every rank `count` doubles, rank r's at offset + r * count; 1 broadcast of
 * [offset, offset + count) from `root`; 2 all-reduce(sum) of [offset, offset + count)), offset, count, root (-1 unless kind 1).
 * Returns the number of operations of the plan (call again with a larger buffer when > cap), 0 for world 1 / nothing to do, -1 bad
 * argument.  which as exa_shard_layout: 0 obj, 1 grad, 2 cons, 3 jac COO, 4 hess COO (exa_allgather_coo), 5 jprod, 6 jtprod,
 * 7 hprod, 8 the cons vector of the fused sweeps.  Host logic only: works for plan-only handles (tests/test_shard_layout.py). */
int exa_collective_plan(int id, int which, int64_t *out, int cap) {
    Handle *hh = get(id);
    if (!hh || which < 0 || which > 8 || (cap > 0 && !out)) return -1;
    Handle &h = *hh;
    if (h.world == 1) return 0;
    const int layout = exa_shard_layout(id, which);
    std::vector<CollOp> ops;
    const Model &m = *h.m;
    auto reduce_all = [&](int64_t n) { if (n > 0) ops.push_back({2, 0, n, -1}); };
    switch (which) {
    case 0: reduce_all(1); break;
    case 1: if (layout == 1) ops = plan_allgather(var_pieces(h), h.world); else reduce_all(m.nvar); break;
    case 2: case 5: case 8: if (layout == 1) ops = plan_allgather(row_pieces(h), h.world); else reduce_all(m.ncon); break;
    case 3: case 4: ops = plan_allgather(coo_pieces(h, which == 4), h.world); break;
    case 6: case 7: if (layout == 1) ops = plan_allgather(window_pieces(h, h.wp[which - 6]), h.world); else reduce_all(m.nvar); break;
    }
    for (size_t k = 0; k < ops.size() && (int)k < cap; k++) { out[4 * k] = ops[k].kind; out[4 * k + 1] = ops[k].off; out[4 * k + 2] = ops[k].count; out[4 * k + 3] = ops[k].root; }
    return (int)ops.size();
}
/* Makes a sharded Jacobian (hess = 0) / Hessian (hess = 1) COO vector whole on every rank: all-gather-v of the ranks' slot
 * ranges (a piece travels once; nothing is zero-filled or summed — an all-reduce of zero-padded vectors would move world x
 * the data).  `local`: what this rank's exa_jac / exa_hess wrote — the packed local slice (exa_set_coo_local) or
 * the global-length vector with this rank's slots in place; `global` [nnzj | nnzh]: receives everything (may equal `local`
 * when that is the global-length vector).  Needs a communicator; world 1: a device copy. */
int exa_allgather_coo(int id, int hess, const double *local, double *global) {
    if (!local || !global) return 1;
    return guard(id, true, [&](Handle &h) {
        const Model &m = *h.m;
        if (h.world > 1 && !h.nccl && !h.hook) throw BadInput("the model has no communicator");
        const bool packed = h.coo_local && h.world > 1;
        if (h.world == 1 || packed || local != global) {
            for (size_t k = 0; k < m.pats.size(); k++) {
                const Pattern &p = m.pats[k];
                const int64_t step = hess ? p.o2step : (p.kind != EXA_PAT_OBJ ? p.o1step : 0);
                if (step <= 0 || p.n <= 0) continue;
                const int64_t lo = part_lo(p.n, h.rank, h.world), hi = part_lo(p.n, h.rank + 1, h.world);
                const int64_t g0 = (hess ? p.o2 : p.o1) + step * lo, l0 = packed ? (hess ? h.lo2[k] : h.lo1[k]) : g0;
                if (hi > lo && local + l0 != global + g0)
                    HIPCHK(hipMemcpyAsync(global + g0, local + l0, 8 * (size_t)(step * (hi - lo)), hipMemcpyDeviceToDevice, h.stream));
            }
        }
        allgatherv(h, global, coo_pieces(h, hess != 0), true);
    });
}
}  // extern "C"
