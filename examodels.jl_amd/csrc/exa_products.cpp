// exa_products.cpp — which implementation J'v / Hv / grad! run (atomics, sorted gather, owner-computes windows, owner pull): mode resolution,
// the sorted and pull set-ups, exa_tune (the only place that measures), exa_time_callback, and their C ABI (include/exahip.h).  Split off
// exa_runtime.cpp in round 4; the shared state is Handle (exa_rt.hpp).
#include "exa_rt.hpp"

using namespace exa;
using namespace exa::rt;

namespace exa {
namespace rt {


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
// Which implementation a product runs is a property of the model fixed BEFORE the call: explicit (exa_set_product_mode),
// measured once by exa_tune and persisted next to the cached module, or — undecided and never tuned — the atomics of the
// sweep.  Callbacks never measure and never synchronise.
// Owner pull: the variable -> item-slot lists.  Built once per shard geometry (here: unsharded models only — a rank of a sharded
// model would need the items of ALL data points that touch its variables, like the windows; the atomics + all-reduce stay there).
bool pull_possible(const Handle &h, bool hess) {
    const Handle::Pull &q = h.pl[hess ? 1 : 0];
    return h.on_device && h.world == 1 && q.planned && q.fpull && q.why.empty();
}
void pull_setup(Handle &h, bool hess) {
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
    try {
        build_sorted_index(q.idx, (const int64_t *)keys.p, total, m.nvar, h.stream);
    } catch (const HipError &) {
        throw;
    } catch (const std::exception &e) {      // keys outside the variables: no pull lists (the other implementations stay)
        q.idx.release();
        q.why = e.what();
        return;
    }
    HIPCHK(hipStreamSynchronize(h.stream));
    if (q.idx.nlong > 0) {       // a variable collecting more than 512 contributions (a slack shared by every point): one thread would walk them all
        q.idx.release();
        q.why = "a variable collects more than 512 contributions (the atomics / the sorted gather handle it cooperatively)";
        return;
    }
    q.ready = true;
}
void do_pull(Handle &h, bool hess, const double *x, const double *y, const double *v, double sigma, double *out) {
    Handle::Pull &q = h.pl[hess ? 1 : 0];
    const void *P = h.dP.p, *th = h.dtheta.p, *ptr = q.idx.ptr, *perm = q.idx.perm, *fp = q.first.p;
    int64_t vb = 0, ve = h.m->nvar;
    const int64_t grid = (ve - vb + kBlock - 1) / kBlock;
    if (hess) { void *a[] = {&P, &x, &y, &th, &v, &sigma, &out, &ptr, &perm, &fp, &vb, &ve}; launch(h, q.fpull, grid, kBlock, a); }
    else { void *a[] = {&P, &x, &th, &v, &out, &ptr, &perm, &fp, &vb, &ve}; launch(h, q.fpull, grid, kBlock, a); }
}
bool sorted_possible(Handle &h, bool hess) {
    const int64_t nnz = hess ? h.lnnzh : h.lnnzj;
    return (h.world == 1 || h.coo_local) && nnz > 0;
}
// Owner-computes windows: possible when the model's targets are range-affine (plan_products) and the module is loaded; a
// SHARDED model takes them only when nothing is left to the tail kernel (no tiny patterns, no entry every point adds to):
// those belong to all ranks at once.
bool window_possible(Handle &h, bool hess) {
    const Handle::Window &w = h.wp[hess ? 1 : 0];
    return w.ok && (h.world == 1 || (w.nx == 0 && !w.has_shared));
}
int resolve_mode(Handle &h, bool hess) {
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
void eager_setup(Handle &h) {
    if (!h.on_device) return;
    const int g = h.grad_mode, jt = h.jt_mode, hp = h.hp_mode;
    (void)resolve_grad_mode(h);
    (void)resolve_mode(h, false);
    (void)resolve_mode(h, true);
    h.grad_mode = g; h.jt_mode = jt; h.hp_mode = hp;          // (still "undecided" for exa_get_*_mode until a call resolves them)
}
static const bool g_eager_registered = (g_eager_setup = eager_setup, true);
void run_product_window(Handle &h, bool hess, const double *x, const double *y, const double *v, double w, double *out) {
    Handle::Window &win = h.wp[hess ? 1 : 0];
    if (h.world == 1) { do_window(h, hess ? WK_HPROD : WK_JTPROD, x, y, v, w, out); return; }
    int64_t w0, w1;
    owned_windows(h, win, h.rank, &w0, &w1);
    do_window(h, hess ? WK_HPROD : WK_JTPROD, x, y, v, w, out, w0, w1);
    allgatherv(h, out, window_pieces(h, win));
}
void run_jtprod(Handle &h, const double *x, const double *v, double *Jtv) {
    const int mode = resolve_mode(h, false);
    if (mode == 2) { run_product_window(h, false, x, nullptr, v, 0.0, Jtv); return; }
    if (mode == 3) { do_pull(h, false, x, nullptr, v, 0.0, Jtv); return; }        // (unsharded: nothing to complete)
    if (mode == 1) do_jtprod_sorted(h, x, v, Jtv); else do_jtprod(h, x, v, Jtv);
    allreduce(h, Jtv, h.m->nvar);
}
void run_hprod(Handle &h, const double *x, const double *y, const double *v, double w, double *Hv) {
    int mode = resolve_mode(h, true);
    if ((mode == 2 || mode == 3) && !y && h.m->ncon > 0) mode = 0;      // objective only: the window / pull kernels evaluate every pattern; the atomics launch the objective groups alone
    if (mode == 2) { run_product_window(h, true, x, y, v, w, Hv); return; }
    if (mode == 3) { do_pull(h, true, x, y, v, w, Hv); return; }
    if (mode == 1) do_hprod_sorted(h, x, y, v, w, Hv); else do_hprod(h, x, y, v, w, Hv);
    allreduce(h, Hv, h.m->nvar);
}
/* What exa_jtprod (hess) / exa_hprod (hess) run: 0 atomics, 1 sorted gather, 2 owner-computes windows (resolved as a
 * call would resolve it, without building anything); buf <- the kernel shape of the windows or why the model has none. */
// the implementation a call WOULD run (explicit mode, else the persisted exa_tune decision, else the windows where the model
// has them), without building anything: shared by exa_product_info and exa_shard_layout so that the two cannot disagree
int product_mode_query(Handle &h, bool hess) {
    const Handle::Window &w = h.wp[hess ? 1 : 0];
    const int mode = hess ? h.hp_mode : h.jt_mode;
    if (mode >= 0) return (mode == 2 && !window_possible(h, hess)) || (mode == 3 && !pull_possible(h, hess)) ? 0 : mode;
    int v = -1;
    if (h.on_device && tune_lookup(source_key(h.gen.source), tune_signature(h, hess ? "hprod" : "jtprod"), &v) && v >= 0 && v <= 3 &&
        (v != 2 || window_possible(h, hess)) && (v != 1 || sorted_possible(h, hess)) && (v != 3 || pull_possible(h, hess))) return v;
    return (h.on_device ? window_possible(h, hess) : w.planned) ? 2 : 0;
}

// Dynamic LDS that leaves `wgs` workgroups of the chained hess_coord! kernel per CU (the CU's LDS asked from the device: 160 KB on MI355X;
// static + dynamic <= 64 KB, the launch limit without a function attribute); 0 when the kernel's own LDS already allows no more than
// that, or the kernel is not there.
static int hess_static_lds(Handle &h, int variant) {
    hipFunction_t f = variant == 1 && h.f_hesscl && h.stage_ok ? h.f_hesscl : h.f_hessc;
    int stat = -1;
    if (!f || hipFuncGetAttribute(&stat, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, f) != hipSuccess) return -1;
    return stat;
}
unsigned hess_throttle_bytes(Handle &h, int variant, int wgs) {
    const int stat = hess_static_lds(h, variant);
    if (stat < 0 || wgs < 1) return 0;
    int dev = 0, cu_lds = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu_lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || cu_lds <= 0)
        cu_lds = 160 * 1024;
    const int total = cu_lds / (wgs + 1) + 1024;          // just too much for wgs + 1 workgroups
    if (stat >= total || total > 64 * 1024) return 0;
    return (unsigned)(total - stat);
}
// A throttle that came from outside (EXAHIP_HESS_DYN_LDS, a persisted tuning decision of another library version): usable only when
// static + dynamic fit the 64 KB a launch may ask for without a function attribute — else 0 (no throttle) instead of a failing launch.
unsigned hess_throttle_clamp(Handle &h, int variant, unsigned want) {
    const int stat = hess_static_lds(h, variant);
    if (stat < 0 || want == 0) return 0;
    return (long)stat + (long)want <= 64 * 1024 ? want : 0;
}
}  // namespace rt
}  // namespace exa

extern "C" {

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
int exa_hess_throttle(int id) {
    Handle *h = get(id);
    if (!h) return -1;
    if (h->hess_variant >= 1 && h->hess_dyn_auto && h->f_hessc) { h->hess_dyn_auto = false; h->hess_dyn_lds = hess_throttle_bytes(*h, h->hess_variant, 3); }
    return h->hess_variant >= 1 ? (int)h->hess_dyn_lds : 0;
}

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
                // each kernel's block order first, then the kernels against each other, interleaved (see ab_min)
                h.hess_variant = 0;
                (void)tune_order(h, CB_HESS, [&] { do_hess(h, x, y, sigma, hv); });
                h.hess_variant = 2;
                (void)tune_order(h, CB_HESSC, [&] { do_hess(h, x, y, sigma, hv); });
                const int order_c = h.order[CB_HESSC];
                int order_cl = order_c;
                std::vector<int> cand = {0, 2};
                if (h.f_hesscl && h.stage_ok) {
                    h.hess_variant = 1;
                    (void)tune_order(h, CB_HESSC, [&] { do_hess(h, x, y, sigma, hv); });
                    order_cl = h.order[CB_HESSC];
                    cand.push_back(1);
                }
                // ... and the chained kernels at three occupancies: as many workgroups per CU as their registers allow, three, two — an LDS
                // throttle (dynamic LDS nobody uses, exa_rt.hpp hess_dyn_lds).  Fewer, longer streams per CU win where the output outgrows
                // the Infinity Cache (LV 1e8 exa_hesscl: 1.70 -> 1.62 ms at three workgroups) and on some boxes below it (profiles/r5_hess_occupancy.txt)
                struct Cand { int variant; unsigned dyn; };
                std::vector<Cand> cs;
                const bool fixed_dyn = getenv("EXAHIP_HESS_DYN_LDS") != nullptr;
                for (int v : cand) {
                    cs.push_back({v, v == 0 || !fixed_dyn ? 0u : h.hess_dyn_lds});
                    if (v == 0 || fixed_dyn) continue;
                    for (int wgs : {3, 2}) { const unsigned d = hess_throttle_bytes(h, v, wgs); if (d) cs.push_back({v, d}); }
                }
                const std::vector<float> tv = ab_min(h, (int)cs.size(), 7, 6, [&](int k) {
                    const int v = cs[(size_t)k].variant;
                    h.hess_variant = v;
                    h.hess_dyn_lds = cs[(size_t)k].dyn;
                    if (v != 0 && h.order[CB_HESSC] != (v == 1 ? order_cl : order_c)) install_order(h, CB_HESSC, v == 1 ? order_cl : order_c);
                    do_hess(h, x, y, sigma, hv);
                });
                // The PLAN-TIME DEFAULT stays unless another candidate wins by 3 % (round 6; round 5: "the plain kernel stays unless ..."): rounds of six
                // launches between other candidates rank candidates that are within a few per cent of each other by noise — one box tuned LV 1e7 to
                // exa_hesscl at 0.1404 ms in the timed region against exa_hess's 0.1307 (gpurun_out/r5am); another dropped LV 1e8's throttle for a 0.3 % lead
                // in the rounds and then ran 1.636 ms where the default ran 1.499 in the same process (profiles/r6_bench_default.json, first version).
                // Where a candidate really wins it is by 6 % and more.
                const int def_variant = h.hess_stream_bytes >= 1.5e9 ? (h.f_hesscl && h.stage_ok ? 1 : 2) : 0;
                const unsigned def_dyn = def_variant && !fixed_dyn ? hess_throttle_bytes(h, def_variant, 3) : (def_variant ? h.hess_dyn_lds : 0u);
                size_t def = 0;
                for (size_t k = 0; k < cs.size(); k++) if (cs[k].variant == def_variant && cs[k].dyn == def_dyn) def = k;
                size_t best = def;
                for (size_t k = 0; k < cs.size(); k++)
                    if (k != def && tv[k] < 0.97f * tv[def] && tv[k] < tv[best]) best = k;
                h.hess_variant = cs[best].variant;
                h.hess_dyn_lds = cs[best].dyn;
                h.hess_dyn_auto = false;
                install_order(h, CB_HESSC, h.hess_variant == 1 ? order_cl : order_c);
                HIPCHK(hipStreamSynchronize(h.stream));
                if (h.norders[CB_HESSC] > 1) tune_store(source_key(h.gen.source), tune_signature(h, "order" + std::to_string((int)CB_HESSC)), h.order[CB_HESSC]);
                if (verbose()) {
                    fprintf(stderr, "[exahip] tune hess_coord kernels (ms per 6 launches):");
                    for (size_t k = 0; k < cs.size(); k++) fprintf(stderr, " variant %d%s%s: %.4f", cs[k].variant, cs[k].dyn ? " +LDS " : "", cs[k].dyn ? std::to_string(cs[k].dyn).c_str() : "", tv[k]);
                    fprintf(stderr, " -> variant %d, %u bytes of throttle\n", h.hess_variant, h.hess_dyn_lds);
                }
                tune_store(source_key(h.gen.source), tune_signature(h, "hessvariant"), h.hess_variant);
                if (!fixed_dyn) tune_store(source_key(h.gen.source), tune_signature(h, "hessdynlds"), (int)h.hess_dyn_lds);
            } else if (h.norders[CB_HESS] > 1) { hv = need(4, h.lnnzh); tune_order(h, CB_HESS, [&] { do_hess(h, x, y, sigma, hv); }); }
            if (h.norders[CB_FUSED] > 1) {
                c = need(2, m.ncon); jv = need(3, h.lnnzj); hv = need(4, h.lnnzh);
                (void)tune_order(h, CB_FUSED, [&] { do_fused(h, x, y, sigma, obj, c, jv, hv); });
            }
            if (h.gridg > 0) {
                // exa_eval_all with the gathered gradient's tiles inside the sweep's launch: its own block order, measured with the real call
                c = need(2, m.ncon); jv = need(3, h.lnnzj); hv = need(4, h.lnnzh); g = need(5, m.nvar);
                const std::vector<float> tv = ab_min(h, 2, 7, 6, [&](int k) { h.orderg = k; do_eval_all(h, x, y, sigma, obj, g, c, jv, hv); });
                h.orderg = tv[0] < 0.99f * tv[1] ? 0 : 1;       // (the interleaved order is the default: the sequential one must win by more than the noise)
                HIPCHK(hipStreamSynchronize(h.stream));
                tune_store(source_key(h.gen.source), tune_signature(h, "orderg"), h.orderg);
                if (verbose()) fprintf(stderr, "[exahip] tune exa_eval_all block order (ms per 6 launches): sequential %.4f interleaved %.4f -> %d\n", tv[0], tv[1], h.orderg);
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
                // (the owner pull — mode 3 — is NOT a candidate: it lost to the atomics on every data-indexed model measured, profiles/r4_pull_ab.txt;
                // it stays the deterministic implementation, exa_set_deterministic / exa_set_product_mode(…, 3))
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

}  // extern "C"
