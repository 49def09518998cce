// exa_gen_window.cpp — the second module of a model: owner-computes WINDOW kernels (exa_chess / exa_cjac without the
// uncompressed round trip; CompressedNLPModel, src/utils.jl:425-579, KA ext :1290-1319), the permuted-store kernels of
// matrices the windows do not fit, and the merged-slot compressed Hessian.
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
static void gen_window_value_fn(std::ostringstream &os, const Model &m, const ParamLayout &L, int k, bool hess) {
    Body b(m, k, L);
    const Pattern &p = b.p;
    const int S = hess ? p.o2step : p.o1step;
    std::vector<Val> acc;
    if (hess) {
        b.forward(p.ad_root, 2, false);
        Val adj;
        if (p.kind == EXA_PAT_OBJ) adj = b.e.raw("sigma", false);
        else adj = b.e.raw("y[" + b.row0() + "]", false);
        GenAlg a(b, p.comp2, p.o2step);
        hrpass0(p, p.ad_root, a, adj, zero_seed(b));
        acc = a.acc;
    } else {
        b.forward(p.ad_root, 1, false);
        GenAlg a(b, p.comp1, p.o1step);
        grpass(p, p.ad_root, a, Emitter::litf(1.0));
        acc = a.acc;
    }
    std::vector<std::string> vals;
    for (int s = 0; s < S; s++) vals.push_back(b.e.sd(acc[s]));
    const char *tag = hess ? "hessv" : "jacv";
    // values of one data point, in slot order
    os << "static __device__ __forceinline__ void " << fn_name(k, tag)
       << "(const long* __restrict__ P, const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ th, "
          "double sigma, long I, double* v) {\n";
    emit_lines(os, b.e);
    for (int s = 0; s < S; s++) os << "    v[" << s << "] = " << vals[s] << ";\n";
    os << "}\n";
}

// adds one chunk's values of pass j into the window: groups of a phase never meet in one word, a barrier between phases
static void gen_window_fn(std::ostringstream &os, const WindowPat &wp, int j, bool hess, int S) {
    const int ngroups = (int)wp.phase.size();
    int nphase = 0;
    for (int ph : wp.phase) nphase = std::max(nphase, ph + 1);
    os << "static __device__ __forceinline__ void w" << j << (hess ? "_hessa" : "_jaca")
       << "(const long* __restrict__ Q, long I, bool act, long c0, int W, double* win, const double* v) {\n"
       << "    const long cb_ = Q[" << wp.qbase << "] * I - c0;\n";
    for (int ph = 0; ph < nphase; ph++) {
        if (ph) os << "    __syncthreads();\n";
        for (int g = 0; g < ngroups; g++) {
            if (wp.phase[g] != ph) continue;
            std::string sum;
            for (int s = 0; s < S; s++)
                if (wp.group[s] == g) sum += (sum.empty() ? "" : " + ") + ("v[" + std::to_string(s) + "]");
            os << "    { const long c = Q[" << wp.qbase + 5 + g << "] + cb_; if (act && (unsigned long)c < (unsigned long)W) win[EXA_WPOS((int)c)] += " << sum << "; }\n";
        }
    }
    os << "}\n";
}

static void emit_window_shared_body(std::ostringstream &os, const Model &m, const std::vector<WindowShared> &sh, bool hess);
static void gen_window_kernels(std::ostringstream &os, const Model &m, const std::vector<WindowPat> &pats, const std::vector<WindowShared> &sh,
                               bool hess, bool single) {
    const char *nm = hess ? "chess" : "cjac";
    const char *fa = hess ? "hessa" : "jaca";
    const char *fv = hess ? "hessv" : "jacv";
    const int np = (int)pats.size();
    // R[window][pass] = first and one-past-last data point touching the window (host-computed: no 64-bit divisions at
    // the head of every workgroup's dependency chain).
    // Occupancy hint: the straight-line kernel is latency-bound between barriers (LV 1e7: 0.141 ms unhinted at 124
    // VGPRs, 0.10 ms at 8 waves per SIMD); only for small bodies, which fit 64 / 80 registers without spilling — the
    // chunk loops did spill under it (LV, two chunks per window: 0.118 -> 0.375 ms)
    int slots = 0;
    for (const auto &wp : pats) slots += hess ? m.pats[wp.k].o2step : m.pats[wp.k].o1step;
    const int waves = !single ? 0 : (slots <= 16 ? 8 : (slots <= 40 ? 6 : 0));
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) ";
    if (waves > 0) os << "__attribute__((amdgpu_waves_per_eu(" << waves << "))) ";
    os << "exa_" << nm << "w(const long* __restrict__ P, const long* __restrict__ Q, "
          "const int* __restrict__ R, const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ th, "
          "double* __restrict__ cout, double sigma, long ncomp, int W) {\n"
          "    extern __shared__ double win[];\n";
    os << "    const long c0 = (long)blockIdx.x * W;\n"
          "    const int* r_ = R + (long)blockIdx.x * " << 2 * np << ";\n"
          "    for (int w = threadIdx.x; w < W; w += EXA_BLOCK) win[w] = 0.0;\n";
    for (int j = 0; j < np; j++)
        os << "    const long lo" << j << " = r_[" << 2 * j << "], hi" << j << " = r_[" << 2 * j + 1 << "];\n";
    if (single) {
        // every pass fits one chunk: all values first (the loads of all passes overlap), then the additions
        os << "    {\n";
        for (int j = 0; j < np; j++) {
            const int S = hess ? m.pats[pats[j].k].o2step : m.pats[pats[j].k].o1step;
            os << "        const bool act" << j << " = lo" << j << " + threadIdx.x < hi" << j << ";\n        const long I" << j << " = act" << j << " ? lo" << j
               << " + threadIdx.x : 0;\n        double v" << j << "[" << S << "];\n        " << fn_name(pats[j].k, fv) << "(P, x, y, th, sigma, I" << j << ", v" << j << ");\n";
        }
        for (int j = 0; j < np; j++)
            os << "        __syncthreads();\n        w" << j << "_" << fa << "(Q, I" << j << ", act" << j << ", c0, W, win, v" << j << ");\n";
    } else {
        // chunk loops, software-pipelined: the next chunk's values are computed (its loads issued) before the current
        // chunk's additions wait at the barrier
        os << "    {\n";
        for (int j = 0; j < np; j++) {
            const int S = hess ? m.pats[pats[j].k].o2step : m.pats[pats[j].k].o1step;
            const std::string ev = fn_name(pats[j].k, fv);
            os << "        if (lo" << j << " < hi" << j << ") {\n            long base = lo" << j << ";\n            bool act = base + threadIdx.x < hi" << j
               << ";\n            long I = act ? base + threadIdx.x : 0;\n            double v[" << S << "], vn[" << S << "];\n            " << ev
               << "(P, x, y, th, sigma, I, v);\n            while (base < hi" << j << ") {\n                const long nb = base + EXA_BLOCK;\n"
               << "                const bool actn = nb + threadIdx.x < hi" << j << ";\n                const long In = actn ? nb + threadIdx.x : 0;\n"
               << "                if (nb < hi" << j << ") " << ev << "(P, x, y, th, sigma, In, vn);\n                __syncthreads();\n                w" << j << "_" << fa
               << "(Q, I, act, c0, W, win, v);\n                for (int s = 0; s < " << S << "; s++) v[s] = vn[s];\n                act = actn; I = In; base = nb;\n"
               << "            }\n        }\n";
        }
    }
    os << "    }\n    __syncthreads();\n"
          "    for (int w = threadIdx.x; w < W; w += EXA_BLOCK) if (c0 + w < ncomp) __builtin_nontemporal_store(win[EXA_WPOS(w)], &cout[c0 + w]);\n}\n";
}

// block-owned variant: see WindowSpec.  R[block][pattern] = the points of the pattern with a slot in one of the block's
// windows (at most EXA_BLOCK of them: one chunk); every pattern is evaluated once, then each pass adds its slots into the
// window of its space; windows are clipped to their space when streamed out
static void gen_window_kernel_blocks(std::ostringstream &os, const Model &m, const std::vector<WindowPat> &pats, bool hess, int nspaces, int zs) {
    const char *nm = hess ? "chess" : "cjac";
    const char *fa = hess ? "hessa" : "jaca";
    const char *fv = hess ? "hessv" : "jacv";
    std::vector<int> pk;
    for (const auto &wp : pats) if (std::find(pk.begin(), pk.end(), wp.k) == pk.end()) pk.push_back(wp.k);
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_" << nm << "w(const long* __restrict__ P, const long* __restrict__ Q, "
          "const int* __restrict__ R, const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ th, "
          "double* __restrict__ cout, double sigma, long ncomp, int W) {\n"
          "    extern __shared__ double win[];\n    const long j_ = blockIdx.x;\n"
          "    const int* r_ = R + j_ * " << 2 * pk.size() << ";\n"
          "    for (int w = threadIdx.x; w < W; w += EXA_BLOCK) win[w] = 0.0;\n";
    for (size_t q = 0; q < pk.size(); q++) {
        const int S = hess ? m.pats[pk[q]].o2step : m.pats[pk[q]].o1step;
        os << "    const bool act" << q << " = r_[" << 2 * q << "] + (long)threadIdx.x < r_[" << 2 * q + 1 << "];\n    const long I" << q << " = act" << q
           << " ? r_[" << 2 * q << "] + (long)threadIdx.x : 0;\n    double v" << q << "[" << S << "];\n    " << fn_name(pk[q], fv) << "(P, x, y, th, sigma, I" << q
           << ", v" << q << ");\n";
    }
    for (size_t j = 0; j < pats.size(); j++) {
        const size_t q = std::find(pk.begin(), pk.end(), pats[j].k) - pk.begin();
        const int z = zs + 4 * pats[j].space;
        os << "    __syncthreads();\n    w" << j << "_" << fa << "(Q, I" << q << ", act" << q << ", Q[" << z << "] + j_ * Q[" << z + 2 << "], (int)Q[" << z + 2
           << "], win + Q[" << z + 3 << "], v" << q << ");\n";
    }
    os << "    __syncthreads();\n";
    for (int sp = 0; sp < nspaces; sp++) {
        const int z = zs + 4 * sp;
        os << "    {\n        const long c0 = Q[" << z << "] + j_ * Q[" << z + 2 << "], end = Q[" << z + 1 << "];\n        const int We = (int)Q[" << z + 2
           << "];\n        const double* wn = win + Q[" << z + 3 << "];\n"
           << "        for (int w = threadIdx.x; w < We; w += EXA_BLOCK) if (c0 + w < end) __builtin_nontemporal_store(wn[EXA_WPOS(w)], &cout[c0 + w]);\n    }\n";
    }
    os << "    (void)ncomp;\n}\n";
}

// irregular end points: X = [pattern, I] per point (up to EXA_BLOCK of them); values go through xbuf; then one thread
// per DISTINCT compressed target adds that target's values in (point, slot) order: T = [ntargets, then per target:
// compressed entry, first, one-past-last position in E], E = positions in xbuf
static void gen_window_x(std::ostringstream &os, const Model &m, const std::vector<int> &pk, bool hess) {
    const char *nm = hess ? "chess" : "cjac";
    const char *fv = hess ? "hessv" : "jacv";
    int smax = 1;
    for (int k : pk) smax = std::max(smax, hess ? m.pats[k].o2step : m.pats[k].o1step);
    os << "extern \"C\" __global__ void __launch_bounds__(1024) exa_" << nm << "x(const long* __restrict__ P, const long* __restrict__ X, "
          "const int* __restrict__ T, const int* __restrict__ E, const double* __restrict__ x, const double* __restrict__ y, "
          "const double* __restrict__ th, double* __restrict__ xbuf, double* __restrict__ cout, double sigma, int nx, "
          "const double* __restrict__ part, const long* __restrict__ F) {\n"
          "    const int t = threadIdx.x;\n    if (t < nx) {\n        const long pk_ = X[2 * t], I = X[2 * t + 1];\n        double v[" << smax << "];\n"
          "        for (int s = 0; s < " << smax << "; s++) v[s] = 0.0;\n";
    for (size_t j = 0; j < pk.size(); j++)
        os << "        " << (j ? "else " : "") << "if (pk_ == " << pk[j] << ") " << fn_name(pk[j], fv) << "(P, x, y, th, sigma, I, v);\n";
    os << "        for (int s = 0; s < " << smax << "; s++) xbuf[t * " << smax << " + s] = v[s];\n    }\n    __syncthreads();\n"
          "    for (int q = t; q < (nx > 0 ? T[0] : 0); q += 1024) {\n        const int c = T[1 + 3 * q];\n        double s = cout[c];\n"
          "        for (int e = T[2 + 3 * q]; e < T[3 + 3 * q]; e++) s += xbuf[E[e]];\n        cout[c] = s;\n    }\n"
          // fold of the shared-entry partial sums: F = [ngroups, then per group: first partial, count, compressed entry];
          // groups in order (several may share an entry), fixed summation order
          "    __shared__ double red[16];\n"
          "    for (long g = 0; g < F[0]; g++) {\n        __syncthreads();\n        const long off = F[1 + 3 * g], n = F[2 + 3 * g];\n        double a = 0.0;\n"
          "        for (long i = t; i < n; i += 1024) a += part[off + i];\n        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);\n"
          "        if ((t & 63) == 0) red[t >> 6] = a;\n        __syncthreads();\n"
          "        if (t == 0) { double s = 0.0; for (int w = 0; w < 16; w++) s += red[w]; cout[F[3 + 3 * g]] += s; }\n    }\n}\n";
}
static void emit_window_shared_body(std::ostringstream &os, const Model &m, const std::vector<WindowShared> &sh, bool hess) {
    const char *fv = hess ? "hessv" : "jacv";
    for (size_t j = 0; j < sh.size(); j++) {
        const int St = hess ? m.pats[sh[j].k].o2step : m.pats[sh[j].k].o1step;
        const size_t ng = sh[j].groups.size();
        os << "        " << (j ? "else " : "") << "if (b < S[" << 4 * (j + 1) + 2 << "]) {\n            const long tile = b - S[" << 4 * j + 2 << "], nt = S["
           << 4 * (j + 1) + 2 << "] - S[" << 4 * j + 2 << "];\n            double acc[" << ng << "];\n            for (int g = 0; g < " << ng << "; g++) acc[g] = 0.0;\n"
           << "#pragma unroll 1\n            for (int u = 0; u < " << kSharedTiles << "; u++) {\n"
           << "                const long I0 = S[" << 4 * j << "] + (tile * " << kSharedTiles << " + u) * EXA_BLOCK + threadIdx.x;\n"
           << "                if (I0 - threadIdx.x >= S[" << 4 * j + 1 << "]) break;\n"
           << "                const bool act = I0 < S[" << 4 * j + 1 << "];\n                const long I = act ? I0 : 0;\n                double v[" << St << "];\n                "
           << fn_name(sh[j].k, fv) << "(P, x, y, th, sigma, I, v);\n";
        for (size_t g = 0; g < ng; g++) {
            std::string sum;
            for (int s : sh[j].groups[g]) sum += (sum.empty() ? "" : " + ") + ("v[" + std::to_string(s) + "]");
            os << "                acc[" << g << "] += act ? " << sum << " : 0.0;\n";
        }
        os << "            }\n";
        for (size_t g = 0; g < ng; g++)
            os << "            { const double s = exa_block_sum(acc[" << g << "]); if (threadIdx.x == 0) part[S[" << 4 * j + 3 << "] + " << g
               << " * nt + tile] = s; __syncthreads(); }\n";
        os << "        }\n";
    }
}

// entries EVERY point adds to (b = 0: the rocket's step length): per-workgroup sums over the regular points (S = [per
// pattern j: e_lo, e_hi, first workgroup, first partial] + sentinel), folded by the tail kernel.  A launch of its own:
// as extra workgroups of the window kernel they each reserved a window's LDS and cost more than the launch (rocket chess
// 0.125 -> 0.137 ms)
static void gen_window_shared(std::ostringstream &os, const Model &m, const std::vector<WindowShared> &sh, bool hess) {
    os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_" << (hess ? "chess" : "cjac") << "s(const long* __restrict__ P, const long* __restrict__ S, "
          "const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ th, double* __restrict__ part, double sigma) {\n"
          "    {\n        const long b = blockIdx.x;\n";
    emit_window_shared_body(os, m, sh, hess);
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
    os << prelude_text(L);
    // LDS position of window entry c: the low four bits (the 64-bit bank) are XOR-ed with the next four, so that lanes
    // striding through the window by 2, 3, 12 ... entries (the stride of a pass) spread over the banks instead of
    // hitting the same 2-4 of them (rocket, stride 12: 8-way conflicts on every read-modify-write); a bijection within
    // each aligned block of 16 entries, W is a multiple of 16
    os << "// windowed compressed-COO kernels\n#define EXA_WPOS(c) ((c) ^ (((c) >> 4) & 15))\n";
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
    for (int hess = 1; hess >= 0; hess--) {
        const auto &pats = hess ? spec.hess : spec.jac;
        const auto &sh = hess ? spec.hess_shared : spec.jac_shared;
        if (pats.empty()) continue;
        // value function once per pattern, accumulate function per (pattern, stride class) pass
        std::vector<int> pk;
        for (const auto &wp : pats) if (std::find(pk.begin(), pk.end(), wp.k) == pk.end()) pk.push_back(wp.k);
        for (const auto &q : sh) if (std::find(pk.begin(), pk.end(), q.k) == pk.end()) pk.push_back(q.k);
        for (int k : pk) gen_window_value_fn(os, m, L, k, hess != 0);
        for (size_t j = 0; j < pats.size(); j++) gen_window_fn(os, pats[j], (int)j, hess != 0, hess ? m.pats[pats[j].k].o2step : m.pats[pats[j].k].o1step);
        if ((hess ? spec.hess_nspaces : spec.jac_nspaces) > 0)
            gen_window_kernel_blocks(os, m, pats, hess != 0, hess ? spec.hess_nspaces : spec.jac_nspaces, hess ? spec.hess_zs : spec.jac_zs);
        else
            gen_window_kernels(os, m, pats, sh, hess != 0, hess ? spec.hess_single : spec.jac_single);
        // every active pattern may own irregular end points
        std::vector<int> all;
        for (int k : L.active[hess ? CB_HESS : CB_JAC]) all.push_back(k);
        for (int k : all) if (std::find(pk.begin(), pk.end(), k) == pk.end()) { gen_window_value_fn(os, m, L, k, hess != 0); pk.push_back(k); }
        gen_window_x(os, m, all, hess != 0);
        if (!sh.empty()) gen_window_shared(os, m, sh, hess != 0);
    }
    return os.str();
}

}  // namespace exa
