// exa_gen_pull.cpp — J'v and Hv by OWNER PULL for models whose scatter targets come from data columns (ACOPF: every bus
// variable is reached through the branch table), the counterpart of kerspmv* / kersyspmv* (KA ext :482-511) without the COO.
//
// The atomics of exa_jtprod / exa_hprod add every branch's contribution to its two buses' entries: zero-fill + ~1.2 M FP64
// atomics, order varying from run to run (ACOPF 78k: 0.21 / 0.24 of the roofline, the vector pipe waiting on the atomics'
// returns).  Here the library builds, once, the list target variable -> the (group, data point, item) contributions that land
// on it — an "item" = what ONE thread of the scattering kernel would add to ONE distinct variable: the contributions of all
// patterns of the fused group to that variable, already merged (Scatter::merge) — and a thread per VARIABLE walks its list,
// RE-EVALUATES each item at its data point (a function specialised per item: the compiler keeps only what that one value
// needs) and stores the sum: no zero-fill, no atomics, a fixed order of additions (ascending item slot: bit-reproducible), and
// every output element written exactly once.  Degree of an ACOPF bus ~3.2 branches: each item is evaluated by exactly one
// thread, the shared subexpressions of a branch (one sincos) are recomputed per item — arithmetic is not what these
// launch-bound kernels wait for.  The list is a SortedIndex over the item slots q = first[g] + items_g * (I - lo_g) + t keyed by
// the target variable (exa_*keys writes the keys), built at exa_set_product_mode / exa_tune / model build (never in a callback).
#include "exa_gen.hpp"

namespace exa {
namespace gen {

namespace {
struct GroupItems {
    Emitter E;
    std::vector<std::unique_ptr<Body>> bodies;
    std::vector<Scatter::Item> items;
};
// the merged items of fused group gi of callback cb (CB_JTPROD / CB_HPROD), exactly as gen_scatter_group_fn forms them
void group_items(const Model &m, const ParamLayout &L, int cb, int gi, GroupItems &out) {
    Scatter sc(out.E, L);
    for (int pk : L.groups[cb][gi]) {
        out.bodies.emplace_back(new Body(m, pk, L, &out.E));
        if (cb == CB_HPROD) hprod_items(*out.bodies.back(), sc); else jtprod_items(*out.bodies.back(), sc);
    }
    sc.merge();
    out.items = sc.items;
}
}  // namespace

}  // namespace gen

std::vector<int> pull_item_counts(const Model &m, const ParamLayout &L, int cb) {
    std::lock_guard<std::mutex> gen_lock(gen::g_gen_mu);
    std::vector<int> out;
    for (size_t g = 0; g < L.groups[cb].size(); g++) {
        gen::GroupItems gi;
        gen::group_items(m, L, cb, (int)g, gi);
        out.push_back((int)gi.items.size());
    }
    return out;
}

std::string generate_pull_module(const Model &m, const ParamLayout &L, bool jt, bool hp) {
    using namespace gen;
    std::lock_guard<std::mutex> gen_lock(g_gen_mu);
    std::ostringstream os;
    os << prelude_text(m, L);
    os << "// owner-pull products: one thread per variable re-evaluates the items that land on it (exa_gen_pull.cpp)\n";
    for (int cb : {CB_JTPROD, CB_HPROD}) {
        if (!(cb == CB_JTPROD ? jt : hp)) continue;
        const bool h2 = cb == CB_HPROD;
        const char *nm = h2 ? "hp" : "jt";
        const std::string params = std::string("const long* __restrict__ P, const double* __restrict__ x, ") + (h2 ? "const double* __restrict__ y, " : "") +
                                   "const double* __restrict__ th, const double* __restrict__ v, " + (h2 ? "double sigma, " : "");
        const std::string args = std::string("P, x, ") + (h2 ? "y, " : "") + "th, v, " + (h2 ? "sigma, " : "");
        const size_t ng = L.groups[cb].size();
        std::vector<int> nt(ng, 0);
        for (size_t g = 0; g < ng; g++) {
            GroupItems gi;
            group_items(m, L, cb, (int)g, gi);
            nt[g] = (int)gi.items.size();
            // keys: the target variable (1-based) of every item of data point I, at keys[q .. q + items)
            os << "static __device__ __forceinline__ void g" << g << "_" << nm << "keys(" << params << "long* __restrict__ keys, long q, long I) {\n";
            emit_lines(os, gi.E);
            for (size_t t = 0; t < gi.items.size(); t++) os << "    keys[q + " << t << "] = " << gi.E.s(gi.items[t].vidx) << ";\n";
            os << "}\n";
            // one function per item: the whole body is there, the compiler keeps what this one value needs
            for (size_t t = 0; t < gi.items.size(); t++) {
                os << "static __device__ __forceinline__ double g" << g << "_" << nm << "item" << t << "(" << params << "long I) {\n";
                emit_lines(os, gi.E);
                os << "    return " << gi.E.sd(gi.items[t].val) << ";\n}\n";
            }
        }
        // the block map of the scattering kernel this list mirrors: an entry stands for EXA_BLOCK * ppt data points (ppt = 16 when a
        // target is shared by every point, gen_dispatch) — the keys kernel must visit all of them (it did not: the keys of 15 points
        // out of 16 stayed uninitialised on such models, found by the deep random sweep of round 4)
        const int ppt = std::max(1, L.ppt[cb]);
        const std::string head = "    const long e_ = ((const long*)P[" + std::to_string(L.blk[cb]) + "])[blockIdx.x];\n    const int ps_ = (int)(e_ >> 40);\n"
                                 "    const long tid0 = (e_ & ((1L << 40) - 1)) * (EXA_BLOCK * " + std::to_string(ppt) + ") + threadIdx.x;\n";
        os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_" << nm << "keys(" << params << "long* __restrict__ keys, const long* __restrict__ first) {\n" << head;
        for (size_t g = 0; g < ng; g++) {
            const auto &pp = L.pat[L.groups[cb][g].front()];
            os << "    " << (g ? "else " : "") << "if (ps_ == " << g << ") {\n#pragma unroll 1\n        for (int u = 0; u < " << ppt << "; u++) { const long t_ = tid0 + u * EXA_BLOCK, I = P["
               << pp.lo << "] + t_; if (I < P[" << pp.hi << "]) g" << g << "_" << nm << "keys(" << args << "keys, first[" << g << "] + " << nt[g] << "L * t_, I); }\n    }\n";
        }
        os << "}\n";
        os << "extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_" << nm << "pull(" << params
           << "double* __restrict__ out, const long* __restrict__ ptr, const unsigned* __restrict__ perm, const long* __restrict__ first, long v_begin, long v_end) {\n"
           << "    const long var = v_begin + (long)blockIdx.x * EXA_BLOCK + threadIdx.x;\n    if (var >= v_end) return;\n    double s = 0.0;\n"
           << "    const long e1 = ptr[var + 1];\n#pragma unroll 1\n    for (long e = ptr[var]; e < e1; e++) {\n        const long q = perm[e];\n        int g = 0;\n";
        for (size_t g = 1; g < ng; g++) os << "        g += q >= first[" << g << "];\n";
        os << "        const long r = q - first[g];\n        switch (g) {\n";
        for (size_t g = 0; g < ng; g++) {
            if (nt[g] == 0) continue;
            const auto &pp = L.pat[L.groups[cb][g].front()];
            os << "        case " << g << ": {\n            const long I = P[" << pp.lo << "] + r / " << nt[g] << ";\n            switch ((int)(r % " << nt[g] << ")) {\n";
            for (int t = 0; t < nt[g]; t++) os << "            case " << t << ": s += g" << g << "_" << nm << "item" << t << "(" << args << "I); break;\n";
            os << "            }\n        } break;\n";
        }
        os << "        }\n    }\n    __builtin_nontemporal_store(s, &out[var]);\n}\n";
    }
    return os.str();
}

}  // namespace exa
