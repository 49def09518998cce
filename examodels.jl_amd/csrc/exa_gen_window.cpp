// exa_gen_window.cpp — the second module of a model: owner-computes WINDOW kernels (exa_chess / exa_cjac without the
// uncompressed round trip; CompressedNLPModel, src/utils.jl:425-579, KA ext :1290-1319), the permuted-store kernels of
// matrices the windows do not fit, and the merged-slot compressed Hessian.
#include <functional>

#include "exa_gen.hpp"

namespace exa {

using namespace gen;

// ---- windowed compressed COO ------------------------------------------------------------------
// exa_chess / exa_cjac without the uncompressed round trip.  For a pattern whose data point I puts slot s on compressed
// entry a_s + b*I (checked against the sorted structure at exa_compress time), a workgroup OWNS a window of W
// consecutive compressed entries: it evaluates, for every pattern, exactly the points that touch the window (the few
// points straddling two windows are evaluated by both, each keeping its own entries), adds the values into an LDS copy
// of the window — slot groups in a fixed order, a barrier between groups that could meet in one word, so the sum order
// is fixed and the result bit-reproducible — and streams the window out with plain coalesced stores: no zero-fill, no
// atomics, 8 B of HBM traffic per COMPRESSED entry instead of 16 B + 12 B per uncompressed one.
// The handful of points at a pattern's ends where the structure is irregular (first columns holding fewer rows) are
// left out of the windows and added afterwards by exa_c*x, sequentially.
// Kinds of window kernels (WKind): what the "slots" of a pattern are, which callback's patterns take part, how the kernels
// are called.  Every window kernel has the same argument list; `v` is the vector of a product (null for the compressed COO).
namespace {
struct KindNames { const char *nm, *fv, *fa; int cb; };
const KindNames kKind[WK_COUNT] = {{"cjac", "jacv", "jaca", CB_JAC}, {"chess", "hessv", "hessa", CB_HESS},
                                   {"jtprod", "jtpv", "jtpa", CB_JTPROD}, {"hprod", "hpv", "hpa", CB_HPROD}};
// contributions of one data point of pattern b.p to J'v / Hv, merged per distinct variable (the order is the order of first
// appearance: the same for the planner and for the value function, which both come through here)
std::vector<Scatter::Item> product_values(Body &b, int wk) {
    Scatter sc(b);
    if (wk == WK_HPROD) hprod_items(b, sc); else jtprod_items(b, sc);
    sc.merge();
    return sc.items;
}
}  // namespace

bool product_items(const Model &m, const ParamLayout &L, int wk, int k, std::vector<int64_t> &a, std::vector<int64_t> &bb) {
    a.clear(); bb.clear();
    Body b(m, k, L);
    for (const Scatter::Item &it : product_values(b, wk)) {
        const Affine f = affine(*it.p, it.ir);
        if (!f.ok) return false;
        if (f.col < 0) { a.push_back(f.c - 1); bb.push_back(0); continue; }
        const Column &c = it.p->cols[f.col];
        a.push_back(f.a * c.start + f.c - 1);
        bb.push_back(f.a * c.step);
    }
    return true;
}

// slots of pattern k in a window kernel of kind wk
static int window_slots(const Model &m, const ParamLayout &L, int wk, int k) {
    if (wk == WK_CJAC) return m.pats[k].o1step;
    if (wk == WK_CHESS) return m.pats[k].o2step;
    Body b(m, k, L);
    return (int)product_values(b, wk).size();
}

static void gen_window_value_fn(std::ostringstream &os, const Model &m, const ParamLayout &L, int k, int wk) {
    Body b(m, k, L);
    const Pattern &p = b.p;
    std::vector<std::string> vals;
    if (wk == WK_CHESS) {
        b.forward(p.ad_root, 2, false);
        Val adj;
        if (p.kind == EXA_PAT_OBJ) adj = b.e.raw("sigma", false);
        else adj = b.e.raw("y[" + b.row0() + "]", false);
        GenAlg a(b, p.comp2, p.o2step);
        hrpass0(p, p.ad_root, a, adj, zero_seed(b));
        for (int s = 0; s < p.o2step; s++) vals.push_back(b.e.sd(a.acc[s]));
    } else if (wk == WK_CJAC) {
        b.forward(p.ad_root, 1, false);
        GenAlg a(b, p.comp1, p.o1step);
        grpass(p, p.ad_root, a, Emitter::litf(1.0));
        for (int s = 0; s < p.o1step; s++) vals.push_back(b.e.sd(a.acc[s]));
    } else {
        for (const Scatter::Item &it : product_values(b, wk)) vals.push_back(b.e.sd(it.val));
    }
    // values of one data point, in slot order
    os << "static __device__ __forceinline__ void " << fn_name(k, kKind[wk].fv)
       << "(const long* __restrict__ P, const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ th, "
          "const double* __restrict__ v, double sigma, long I, double* o_) {\n";
    emit_lines(os, b.e);
    for (size_t s = 0; s < vals.size(); s++) os << "    o_[" << s << "] = " << vals[s] << ";\n";
    os << "}\n";
}

// adds one chunk's values of pass j into the window: groups of a phase never meet in one word, a barrier between phases
static void gen_window_fn(std::ostringstream &os, const WindowPat &wp, int j, int wk, int S) {
    const int ngroups = (int)wp.phase.size();
    int nphase = 0;
    for (int ph : wp.phase) nphase = std::max(nphase, ph + 1);
    os << "static __device__ __forceinline__ void w" << j << "_" << kKind[wk].fa
       << "(const long* __restrict__ Q, long I, bool act, long c0, int W, double* win, const double* o_) {\n"
       << "    const long cb_ = Q[" << wp.qbase << "] * I - c0;\n";
    for (int ph = 0; ph < nphase; ph++) {
        if (ph) os << "    __syncthreads();\n";
        for (int g = 0; g < ngroups; g++) {
            if (wp.phase[g] != ph) continue;
            std::string sum;
            for (int s = 0; s < S; s++)
                if (wp.group[s] == g) sum += (sum.empty() ? "" : " + ") + ("o_[" + std::to_string(s) + "]");
            os << "    { const long c = Q[" << wp.qbase + 5 + g << "] + cb_; if (act && (unsigned long)c < (unsigned long)W) win[EXA_WPOS((int)c)] += " << sum << "; }\n";
        }
    }
    os << "}\n";
}

// PLANES form (WindowMatrix::planes).  Plane of (pass j, group g): EXA_BLOCK doubles, written by the lane that evaluated the
// point; groups are numbered over all passes in (pass, phase, group) order — the order of the additions.
template <class Fv>
static void emit_planes(std::ostringstream &os, const std::vector<int> &S, const std::vector<WindowPat> &pats, std::function<std::string(int)>, std::function<std::string(int)>,
                        Fv vals) {
    int gg = 0;
    for (size_t j = 0; j < pats.size(); j++) {
        const WindowPat &wp = pats[j];
        int nphase = 0;
        for (int ph : wp.phase) nphase = std::max(nphase, ph + 1);
        for (int ph = 0; ph < nphase; ph++)
            for (size_t g = 0; g < wp.phase.size(); g++) {
                if (wp.phase[g] != ph) continue;
                std::string sum;
                for (int s = 0; s < S[wp.k]; s++)
                    if (wp.group[s] == (int)g) sum += (sum.empty() ? "" : " + ") + (vals((int)j) + "[" + std::to_string(s) + "]");
                os << "        win[" << gg * kBlock << " + threadIdx.x] = " << sum << ";\n";
                gg++;
            }
    }
    os << "        __syncthreads();\n";
}
// the owner of entry `e` adds what the planes hold for it; space < 0: every pass, else the passes of that space
static void emit_plane_gather(std::ostringstream &os, const std::vector<WindowPat> &pats, int space, const std::string &e,
                              std::function<std::string(int)> lo, std::function<std::string(int)> hi) {
    int gg = 0;
    for (size_t j = 0; j < pats.size(); j++) {
        const WindowPat &wp = pats[j];
        int nphase = 0;
        for (int ph : wp.phase) nphase = std::max(nphase, ph + 1);
        for (int ph = 0; ph < nphase; ph++)
            for (size_t g = 0; g < wp.phase.size(); g++) {
                if (wp.phase[g] != ph) continue;
                if (space < 0 || wp.space == space)
                    os << "        { const long q_ = " << e << " - Q[" << wp.qbase + 5 + g << "] - " << lo((int)j) << "; if ((unsigned long)q_ < (unsigned long)(" << hi((int)j)
                       << " - " << lo((int)j) << ")) acc_ += win[" << gg * kBlock << " + q_]; }\n";
                gg++;
            }
    }
}

// shared-entry sums formed inside a one-chunk window kernel (WindowShared::attach): `own` = the point's first target of the
// attached pass lies in this workgroup's window (c0 / Wexpr: first entry and length of that window), so every regular point is
// counted by exactly one workgroup; one block sum per group, written to this window's partial
static void emit_shared_in(std::ostringstream &os, const WindowMatrix &wm, const std::string &widx, std::function<std::string(int)> c0,
                           std::function<std::string(int)> Wexpr, std::function<std::string(int)> Ivar, std::function<std::string(int)> act,
                           std::function<std::string(int)> vals) {
    for (const WindowShared &sh : wm.shared_in) {
        const WindowPat &wp = wm.pats[sh.attach];
        size_t g0 = 0;
        for (size_t g = 0; g < wp.phase.size(); g++) if (wp.phase[g] < wp.phase[g0]) g0 = g;
        os << "    {\n        const long e_ = Q[" << wp.qbase + 5 + g0 << "] + Q[" << wp.qbase << "] * " << Ivar(sh.attach) << " - (" << c0(sh.attach) << ");\n"
           << "        const bool own_ = " << act(sh.attach) << " && (unsigned long)e_ < (unsigned long)(" << Wexpr(sh.attach) << ");\n";
        for (size_t g = 0; g < sh.groups.size(); g++) {
            std::string sum;
            for (int sl : sh.groups[g]) sum += (sum.empty() ? "" : " + ") + (vals(sh.attach) + "[" + std::to_string(sl) + "]");
            os << "        { const double s_ = exa_block_sum(own_ ? " << sum << " : 0.0); if (threadIdx.x == 0) part[Q[" << sh.qs << "] + " << g << " * Q[" << sh.qs + 1
               << "] + " << widx << "] = s_; __syncthreads(); }\n";
        }
        os << "    }\n";
    }
}

static const char *kWindowArgs = "(const long* __restrict__ P, const long* __restrict__ Q, const int* __restrict__ R, const double* __restrict__ x, "
                                 "const double* __restrict__ y, const double* __restrict__ th, const double* __restrict__ v, double* __restrict__ cout, "
                                 "double sigma, long ncomp, int W, long wb, double* __restrict__ part) {\n";

static void gen_window_kernels(std::ostringstream &os, const std::vector<int> &S, const WindowMatrix &wm, int wk) {
    const KindNames &kn = kKind[wk];
    const auto &pats = wm.pats;
    const int np = (int)pats.size();
    // R[window][pass] = first and one-past-last data point touching the window (host-computed: no 64-bit divisions at
    // the head of every workgroup's dependency chain).
    // Occupancy hint: the straight-line kernel is latency-bound between barriers (LV 1e7: 0.141 ms unhinted at 124
    // VGPRs, 0.10 ms at 8 waves per SIMD); only for small bodies, which fit 64 / 80 registers without spilling — the
    // chunk loops did spill under it (LV, two chunks per window: 0.118 -> 0.375 ms)
    int slots = 0;
    for (const auto &wp : pats) slots += S[wp.k];
    // (products: every pass's values are live until the planes are written — the 6-wave hint made a 12-pass kernel spill 820 B
    // per lane)
    const int waves = !wm.single ? 0 : (slots <= 16 ? 8 : (slots <= 40 && wk <= WK_CHESS ? 6 : 0));
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) ";
    if (waves > 0) os << "__attribute__((amdgpu_waves_per_eu(" << waves << "))) ";
    os << "exa_" << kn.nm << "w" << kWindowArgs << "    extern __shared__ double win[];\n";
    // (wb = first window of this launch: a rank that owns a range of the output evaluates its own windows only)
    os << "    const long wi_ = (long)blockIdx.x + wb;\n    const long c0 = wi_ * W;\n"
          "    const int* r_ = R + wi_ * " << 2 * np << ";\n";
    if (!wm.planes) os << "    for (int w = threadIdx.x; w < W; w += EXA_BLOCK) win[w] = 0.0;\n";
    for (int j = 0; j < np; j++)
        os << "    const long lo" << j << " = r_[" << 2 * j << "], hi" << j << " = r_[" << 2 * j + 1 << "];\n";
    if (wm.single && wm.planes) {
        // PLANES (see WindowMatrix): values of all passes, one plane per slot group, one barrier, the owner of an entry adds
        os << "    {\n";
        for (int j = 0; j < np; j++)
            os << "        const bool act" << j << " = lo" << j << " + threadIdx.x < hi" << j << ";\n        const long I" << j << " = act" << j << " ? lo" << j
               << " + threadIdx.x : 0;\n        double v" << j << "[" << std::max(1, S[pats[j].k]) << "];\n        " << fn_name(pats[j].k, kn.fv) << "(P, x, y, th, v, sigma, I" << j << ", v" << j << ");\n";
        emit_shared_in(os, wm, "wi_", [&](int) { return std::string("c0"); }, [&](int) { return std::string("W"); }, [&](int j) { return "I" + std::to_string(j); },
                       [&](int j) { return "act" + std::to_string(j); }, [&](int j) { return "v" + std::to_string(j); });
        emit_planes(os, S, pats, [&](int j) { return "lo" + std::to_string(j); }, [&](int j) { return "hi" + std::to_string(j); },
                    [&](int j) { return "v" + std::to_string(j); });
        os << "        const long e_ = c0 + threadIdx.x;\n        double acc_ = 0.0;\n";
        emit_plane_gather(os, pats, -1, "e_", [&](int j) { return "lo" + std::to_string(j); }, [&](int j) { return "hi" + std::to_string(j); });
        os << "        if ((int)threadIdx.x < W && e_ < ncomp) __builtin_nontemporal_store(acc_, &cout[e_]);\n    }\n}\n";
        return;
    }
    if (wm.single) {
        // every pass fits one chunk: all values first (the loads of all passes overlap), then the additions
        os << "    {\n";
        for (int j = 0; j < np; j++)
            os << "        const bool act" << j << " = lo" << j << " + threadIdx.x < hi" << j << ";\n        const long I" << j << " = act" << j << " ? lo" << j
               << " + threadIdx.x : 0;\n        double v" << j << "[" << std::max(1, S[pats[j].k]) << "];\n        " << fn_name(pats[j].k, kn.fv) << "(P, x, y, th, v, sigma, I" << j << ", v" << j << ");\n";
        emit_shared_in(os, wm, "wi_", [&](int) { return std::string("c0"); }, [&](int) { return std::string("W"); }, [&](int j) { return "I" + std::to_string(j); },
                       [&](int j) { return "act" + std::to_string(j); }, [&](int j) { return "v" + std::to_string(j); });
        for (int j = 0; j < np; j++)
            os << "        __syncthreads();\n        w" << j << "_" << kn.fa << "(Q, I" << j << ", act" << j << ", c0, W, win, v" << j << ");\n";
    } else {
        // chunk loops, software-pipelined: the next chunk's values are computed (its loads issued) before the current
        // chunk's additions wait at the barrier
        os << "    {\n";
        for (int j = 0; j < np; j++) {
            const int Sj = std::max(1, S[pats[j].k]);
            const std::string ev = fn_name(pats[j].k, kn.fv);
            os << "        if (lo" << j << " < hi" << j << ") {\n            long base = lo" << j << ";\n            bool act = base + threadIdx.x < hi" << j
               << ";\n            long I = act ? base + threadIdx.x : 0;\n            double vc[" << Sj << "], vn[" << Sj << "];\n            " << ev
               << "(P, x, y, th, v, sigma, I, vc);\n            while (base < hi" << j << ") {\n                const long nb = base + EXA_BLOCK;\n"
               << "                const bool actn = nb + threadIdx.x < hi" << j << ";\n                const long In = actn ? nb + threadIdx.x : 0;\n"
               << "                if (nb < hi" << j << ") " << ev << "(P, x, y, th, v, sigma, In, vn);\n                __syncthreads();\n                w" << j << "_" << kn.fa
               << "(Q, I, act, c0, W, win, vc);\n                for (int s = 0; s < " << Sj << "; s++) vc[s] = vn[s];\n                act = actn; I = In; base = nb;\n"
               << "            }\n        }\n";
        }
    }
    os << "    }\n    __syncthreads();\n"
          "    for (int w = threadIdx.x; w < W; w += EXA_BLOCK) if (c0 + w < ncomp) __builtin_nontemporal_store(win[EXA_WPOS(w)], &cout[c0 + w]);\n}\n";
}

// block-owned variant: see WindowMatrix.  R[block][pattern] = the points of the pattern with a slot in one of the block's
// windows (at most EXA_BLOCK of them: one chunk); every pattern is evaluated once, then each pass adds its slots into the
// window of its space; windows are clipped to their space when streamed out
static void gen_window_kernel_blocks(std::ostringstream &os, const std::vector<int> &S, const WindowMatrix &wm, int wk) {
    const KindNames &kn = kKind[wk];
    const auto &pats = wm.pats;
    std::vector<int> pk;
    for (const auto &wp : pats) if (std::find(pk.begin(), pk.end(), wp.k) == pk.end()) pk.push_back(wp.k);
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_" << kn.nm << "w" << kWindowArgs
       << "    extern __shared__ double win[];\n    const long j_ = (long)blockIdx.x + wb;\n"
          "    const int* r_ = R + j_ * " << 2 * pk.size() << ";\n";
    if (!wm.planes) os << "    for (int w = threadIdx.x; w < W; w += EXA_BLOCK) win[w] = 0.0;\n";
    for (size_t q = 0; q < pk.size(); q++)
        os << "    const bool act" << q << " = r_[" << 2 * q << "] + (long)threadIdx.x < r_[" << 2 * q + 1 << "];\n    const long I" << q << " = act" << q
           << " ? r_[" << 2 * q << "] + (long)threadIdx.x : 0;\n    double v" << q << "[" << std::max(1, S[pk[q]]) << "];\n    " << fn_name(pk[q], kn.fv) << "(P, x, y, th, v, sigma, I" << q
           << ", v" << q << ");\n";
    {
        auto qof_ = [&](int j) { return (size_t)(std::find(pk.begin(), pk.end(), pats[j].k) - pk.begin()); };
        auto zof_ = [&](int j) { return wm.zs + 4 * pats[j].space; };
        emit_shared_in(os, wm, "j_", [&](int j) { return "Q[" + std::to_string(zof_(j)) + "] + j_ * Q[" + std::to_string(zof_(j) + 2) + "]"; },
                       [&](int j) { return "Q[" + std::to_string(zof_(j) + 2) + "]"; }, [&](int j) { return "I" + std::to_string(qof_(j)); },
                       [&](int j) { return "act" + std::to_string(qof_(j)); }, [&](int j) { return "v" + std::to_string(qof_(j)); });
    }
    if (wm.planes) {
        auto qof = [&](int j) { return (size_t)(std::find(pk.begin(), pk.end(), pats[j].k) - pk.begin()); };
        auto lo = [&](int j) { return "(long)r_[" + std::to_string(2 * qof(j)) + "]"; };
        auto hi = [&](int j) { return "(long)r_[" + std::to_string(2 * qof(j) + 1) + "]"; };
        os << "    {\n";
        emit_planes(os, S, pats, lo, hi, [&](int j) { return "v" + std::to_string(qof(j)); });
        for (int sp = 0; sp < wm.nspaces; sp++) {
            const int z = wm.zs + 4 * sp;
            os << "        {\n        const long e_ = Q[" << z << "] + j_ * Q[" << z + 2 << "] + threadIdx.x;\n        double acc_ = 0.0;\n";
            emit_plane_gather(os, pats, sp, "e_", lo, hi);
            os << "        if ((long)threadIdx.x < Q[" << z + 2 << "] && e_ < Q[" << z + 1 << "]) __builtin_nontemporal_store(acc_, &cout[e_]);\n        }\n";
        }
        os << "    }\n    (void)ncomp;\n}\n";
        return;
    }
    for (size_t j = 0; j < pats.size(); j++) {
        const size_t q = std::find(pk.begin(), pk.end(), pats[j].k) - pk.begin();
        const int z = wm.zs + 4 * pats[j].space;
        os << "    __syncthreads();\n    w" << j << "_" << kn.fa << "(Q, I" << q << ", act" << q << ", Q[" << z << "] + j_ * Q[" << z + 2 << "], (int)Q[" << z + 2
           << "], win + Q[" << z + 3 << "], v" << q << ");\n";
    }
    os << "    __syncthreads();\n";
    for (int sp = 0; sp < wm.nspaces; sp++) {
        const int z = wm.zs + 4 * sp;
        os << "    {\n        const long c0 = Q[" << z << "] + j_ * Q[" << z + 2 << "], end = Q[" << z + 1 << "];\n        const int We = (int)Q[" << z + 2
           << "];\n        const double* wn = win + Q[" << z + 3 << "];\n"
           << "        for (int w = threadIdx.x; w < We; w += EXA_BLOCK) if (c0 + w < end) __builtin_nontemporal_store(wn[EXA_WPOS(w)], &cout[c0 + w]);\n    }\n";
    }
    os << "    (void)ncomp;\n}\n";
}

// irregular end points: X = [pattern, I] per point (up to EXA_BLOCK of them); values go through xbuf; then one thread
// per DISTINCT target entry adds that target's values in (point, slot) order: T = [ntargets, then per target:
// entry, first, one-past-last position in E], E = positions in xbuf
static void gen_window_x(std::ostringstream &os, const std::vector<int> &S, const std::vector<int> &pk, int wk) {
    const KindNames &kn = kKind[wk];
    int smax = 1;
    for (int k : pk) smax = std::max(smax, S[k]);
    // (512 threads: with 1024 the 128 registers a lane may have made the rocket's Hv tail kernel spill into scratch)
    os << "extern \"C\" __global__ void __launch_bounds__(512) exa_" << kn.nm << "x(const long* __restrict__ P, const long* __restrict__ X, "
          "const int* __restrict__ T, const int* __restrict__ E, const double* __restrict__ x, const double* __restrict__ y, "
          "const double* __restrict__ th, const double* __restrict__ v, double* __restrict__ xbuf, double* __restrict__ cout, double sigma, int nx, "
          "const double* __restrict__ part, const long* __restrict__ F) {\n"
          "    const int t = threadIdx.x;\n    if (t < nx) {\n        const long pk_ = X[2 * t], I = X[2 * t + 1];\n        double o_[" << smax << "];\n"
          "        for (int s = 0; s < " << smax << "; s++) o_[s] = 0.0;\n";
    for (size_t j = 0; j < pk.size(); j++)
        os << "        " << (j ? "else " : "") << "if (pk_ == " << pk[j] << ") " << fn_name(pk[j], kn.fv) << "(P, x, y, th, v, sigma, I, o_);\n";
    os << "        for (int s = 0; s < " << smax << "; s++) xbuf[t * " << smax << " + s] = o_[s];\n    }\n    __syncthreads();\n"
          "    for (int q = t; q < (nx > 0 ? T[0] : 0); q += 512) {\n        const int c = T[1 + 3 * q];\n        double s = cout[c];\n"
          "        for (int e = T[2 + 3 * q]; e < T[3 + 3 * q]; e++) s += xbuf[E[e]];\n        cout[c] = s;\n    }\n"
          // fold of the shared-entry partial sums: F = [ngroups, then per group: first partial, count, entry];
          // groups in order (several may share an entry), fixed summation order
          // (thread 0 carries the entry it is adding to in a register across consecutive groups of the same entry — the rocket's step length
          // collects a group per pattern — and writes it once: c, c + s0, (c + s0) + s1, ... as before, without a global read-modify-write per group)
          "    __shared__ double red[8];\n    long cur_ = -1;\n    double acc_ = 0.0;\n"
          "    for (long g = 0; g < F[0]; g++) {\n        __syncthreads();\n        const long off = F[1 + 3 * g], n = F[2 + 3 * g];\n        double a = 0.0;\n"
          // (eight loads in flight per thread, added in the same ascending order as a plain loop: with one load per iteration every
          // addition waited for its own load — the rocket's J'v tail took 13 us for a few thousand partials)
          "        for (long i0 = t; i0 < n; i0 += 512 * 8) {\n            double b_[8];\n#pragma unroll\n"
          "            for (int k = 0; k < 8; k++) { const long i = i0 + (long)k * 512; b_[k] = i < n ? part[off + i] : 0.0; }\n#pragma unroll\n"
          "            for (int k = 0; k < 8; k++) if (i0 + (long)k * 512 < n) a += b_[k];\n        }\n"
          "        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);\n"
          "        if ((t & 63) == 0) red[t >> 6] = a;\n        __syncthreads();\n"
          "        if (t == 0) {\n            double s = 0.0;\n            for (int w = 0; w < 8; w++) s += red[w];\n            const long e_ = F[3 + 3 * g];\n"
          "            if (e_ != cur_) { if (cur_ >= 0) cout[cur_] = acc_; cur_ = e_; acc_ = cout[e_]; }\n            acc_ += s;\n        }\n    }\n"
          "    if (t == 0 && cur_ >= 0) cout[cur_] = acc_;\n}\n";
}

// entries EVERY point adds to (b = 0: the rocket's step length): per-workgroup sums over the regular points (S_ = [per
// pattern j: e_lo, e_hi, first workgroup, first partial] + sentinel), folded by the tail kernel.  A launch of its own:
// as extra workgroups of the window kernel they each reserved a window's LDS and cost more than the launch (rocket chess
// 0.125 -> 0.137 ms)
static void gen_window_shared(std::ostringstream &os, const std::vector<int> &S, const std::vector<WindowShared> &sh, int wk) {
    const KindNames &kn = kKind[wk];
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_" << kn.nm << "s(const long* __restrict__ P, const long* __restrict__ S, "
          "const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ th, const double* __restrict__ v, "
          "double* __restrict__ part, double sigma) {\n    {\n        const long b = blockIdx.x;\n";
    for (size_t j = 0; j < sh.size(); j++) {
        const int St = std::max(1, S[sh[j].k]);
        const size_t ng = sh[j].groups.size();
        os << "        " << (j ? "else " : "") << "if (b < S[" << 4 * (j + 1) + 2 << "]) {\n            const long tile = b - S[" << 4 * j + 2 << "], nt = S["
           << 4 * (j + 1) + 2 << "] - S[" << 4 * j + 2 << "];\n            double acc[" << ng << "];\n            for (int g = 0; g < " << ng << "; g++) acc[g] = 0.0;\n"
           << "#pragma unroll 1\n            for (int u = 0; u < " << kSharedTiles << "; u++) {\n"
           << "                const long I0 = S[" << 4 * j << "] + (tile * " << kSharedTiles << " + u) * EXA_BLOCK + threadIdx.x;\n"
           << "                if (I0 - threadIdx.x >= S[" << 4 * j + 1 << "]) break;\n"
           << "                const bool act = I0 < S[" << 4 * j + 1 << "];\n                const long I = act ? I0 : 0;\n                double o_[" << St << "];\n                "
           << fn_name(sh[j].k, kn.fv) << "(P, x, y, th, v, sigma, I, o_);\n";
        for (size_t g = 0; g < ng; g++) {
            std::string sum;
            for (int s : sh[j].groups[g]) sum += (sum.empty() ? "" : " + ") + ("o_[" + std::to_string(s) + "]");
            os << "                acc[" << g << "] += act ? " << sum << " : 0.0;\n";
        }
        os << "            }\n";
        for (size_t g = 0; g < ng; g++)
            os << "            { const double s = exa_block_sum(acc[" << g << "]); if (threadIdx.x == 0) part[S[" << 4 * j + 3 << "] + " << g
               << " * nt + tile] = s; __syncthreads(); }\n";
        os << "        }\n";
    }
    os << "    }\n}\n";
}

std::vector<int> merged_hess_slots(const Model &m, const ParamLayout &L) {
    std::vector<int> out;
    for (const auto &grp : L.groups[CB_HESS]) out.push_back(merged_slot_count(m, L, grp));
    return out;
}

std::string generate_window_module(const Model &m, const ParamLayout &L, const WindowSpec &spec) {
    std::lock_guard<std::mutex> gen_lock(g_gen_mu);
    std::ostringstream os;
    os << prelude_text(m, L);
    // LDS position of window entry c: the low four bits (the 64-bit bank) are XOR-ed with the next four, so that lanes
    // striding through the window by 2, 3, 12 ... entries (the stride of a pass) spread over the banks instead of
    // hitting the same 2-4 of them (rocket, stride 12: 8-way conflicts on every read-modify-write); a bijection within
    // each aligned block of 16 entries, W is a multiple of 16
    os << "// window kernels\n#define EXA_WPOS(c) ((c) ^ (((c) >> 4) & 15))\n";
    if (spec.hess_merged) {
        // exa_chessm / exa_hstructm: the merged slot space (see merge_slot); M[g] = first merged slot of group g
        const auto &groups = L.groups[CB_HESS];
        for (size_t g = 0; g < groups.size(); g++) { gen_merged_hess_fn(os, m, L, (int)g); gen_merged_struct_fn(os, m, L, (int)g); }
        const std::string head = "    const long e_ = ((const long*)P[" + std::to_string(L.blk[CB_HESS]) + "])[blockIdx.x];\n    const int ps_ = (int)(e_ >> 40);\n"
                                 "    const long tid0 = (e_ & ((1L << 40) - 1)) * EXA_BLOCK + threadIdx.x;\n";
        os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_chessm(const long* __restrict__ P, const double* __restrict__ x, "
              "const double* __restrict__ y, const double* __restrict__ th, double* __restrict__ out, double sigma, const unsigned* __restrict__ pos, "
              "const long* __restrict__ M) {\n" << head;
        for (size_t g = 0; g < groups.size(); g++)
            os << "    " << (g ? "else " : "") << "if (ps_ == " << g << ") g" << g << "_hessm(P, x, y, th, out, sigma, tid0, pos, M[" << g << "]);\n";
        os << "}\nextern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_hstructm(const long* __restrict__ P, long* __restrict__ rows, "
              "long* __restrict__ cols, const long* __restrict__ M) {\n" << head;
        for (size_t g = 0; g < groups.size(); g++)
            os << "    " << (g ? "else " : "") << "if (ps_ == " << g << ") g" << g << "_hstm(P, rows, cols, tid0, M[" << g << "]);\n";
        os << "}\n";
    }
    for (int hess = 1; hess >= 0; hess--) {
        if (!(hess ? spec.hess_scatter : spec.jac_scatter)) continue;
        const int cb = hess ? CB_HESS : CB_JAC;
        for (size_t g = 0; g < L.groups[cb].size(); g++) gen_coo_group_fn(os, m, L, cb, (int)g, true);
        os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_" << (hess ? "chessp" : "cjacp")
           << "(const long* __restrict__ P, const double* __restrict__ x, " << (hess ? "const double* __restrict__ y, " : "")
           << "const double* __restrict__ th, double* __restrict__ out, " << (hess ? "double sigma, " : "") << "const unsigned* __restrict__ pos) {\n"
           << "    const long e_ = ((const long*)P[" << L.blk[cb] << "])[blockIdx.x];\n    const int ps_ = (int)(e_ >> 40);\n"
              "    const long tid0 = (e_ & ((1L << 40) - 1)) * EXA_BLOCK + threadIdx.x;\n";
        for (size_t g = 0; g < L.groups[cb].size(); g++)
            os << "    " << (g ? "else " : "") << "if (ps_ == " << g << ") g" << g << "_" << (hess ? "hessp" : "jacp") << "(P, x, " << (hess ? "y, " : "")
               << "th, out, " << (hess ? "sigma, " : "") << "tid0, pos);\n";
        os << "}\n";
    }
    for (int wk = WK_COUNT - 1; wk >= 0; wk--) {
        const WindowMatrix &wm = spec.mat[wk];
        if (wm.pats.empty()) continue;
        std::vector<int> S(m.pats.size(), 0);
        for (int k : L.active[kKind[wk].cb]) S[k] = window_slots(m, L, wk, k);
        // value function once per pattern, accumulate function per (pattern, stride class) pass
        std::vector<int> pk;
        for (const auto &wp : wm.pats) if (std::find(pk.begin(), pk.end(), wp.k) == pk.end()) pk.push_back(wp.k);
        for (const auto &q : wm.shared) if (std::find(pk.begin(), pk.end(), q.k) == pk.end()) pk.push_back(q.k);
        for (int k : pk) gen_window_value_fn(os, m, L, k, wk);
        for (size_t j = 0; j < wm.pats.size(); j++) gen_window_fn(os, wm.pats[j], (int)j, wk, S[wm.pats[j].k]);
        if (wm.nspaces > 0) gen_window_kernel_blocks(os, S, wm, wk);
        else gen_window_kernels(os, S, wm, wk);
        // every active pattern may own irregular end points
        std::vector<int> all;
        for (int k : L.active[kKind[wk].cb]) all.push_back(k);
        for (int k : all) if (std::find(pk.begin(), pk.end(), k) == pk.end()) { gen_window_value_fn(os, m, L, k, wk); pk.push_back(k); }
        gen_window_x(os, S, all, wk);
        if (!wm.shared.empty()) gen_window_shared(os, S, wm.shared, wk);
    }
    return os.str();
}

}  // namespace exa
