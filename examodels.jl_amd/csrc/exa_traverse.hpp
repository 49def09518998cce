// exa_traverse.hpp — the reverse-sweep recursions, written once over an abstract adjoint algebra.
//
// The traversal ORDER is the contract: the k-th leaf visit of these recursions is the k-th entry of the
// reference's comp1/comp2 tuples (src/simdfunction.jl:78-100).  The planner instantiates them with a dummy
// algebra to list the visits; the code generator instantiates them with symbolic values to emit the
// straight-line adjoint code of a pattern.
//
//   grpass   <- src/gradient.jl:64-86 (== jrpass, src/jacobian.jl:16-40)
//   hrpass0  <- src/hessian.jl:382-517   (top-level linear peeling)
//   hrpass   <- src/hessian.jl:337-380, 580-592
//   hdrpass  <- src/hessian.jl:16-268, 317-320
#pragma once
#include "exa_internal.hpp"

namespace exa {

// Algebra concept:
//   using T;                       adjoint value
//   T y1(int n), y2(int n), h11(int n), h12(int n), h22(int n);   partials stored at AD node n
//   T mul(T,T), add(T,T), neg(T);
//   void leaf1(int n, T adj);                      first-order visit of VAR leaf n
//   void leaf2(int n1, int n2, T val, bool cross); second-order visit; cross => hdrpass VAR x VAR (value is `adj`,
//                                                  the algebra applies the i==j ? 2adj : adj rule, hessian.jl:251-268)

template <class Alg>
void grpass(const Pattern &p, int n, Alg &a, typename Alg::T adj) {
    const ADNode &t = p.ad[n];
    switch (t.kind) {
    case AD_CONST: case AD_NULL: return;
    case AD_UN: grpass(p, t.l, a, a.mul(adj, a.y1(n))); return;
    case AD_BIN:
        grpass(p, t.l, a, a.mul(adj, a.y1(n)));
        grpass(p, t.r, a, a.mul(adj, a.y2(n)));
        return;
    case AD_VAR: a.leaf1(n, adj); return;
    }
}

template <class Alg>
void hdrpass(const Pattern &p, int n1, int n2, Alg &a, typename Alg::T adj) {
    const ADNode &t1 = p.ad[n1], &t2 = p.ad[n2];
    const int k1 = t1.kind, k2 = t2.kind;
    if (k1 == AD_NULL || k2 == AD_NULL || k1 == AD_CONST || k2 == AD_CONST) return;
    if (k1 == AD_UN && k2 == AD_UN) { hdrpass(p, t1.l, t2.l, a, a.mul(a.mul(adj, a.y1(n1)), a.y1(n2))); return; }
    if (k1 == AD_VAR && k2 == AD_UN) { hdrpass(p, n1, t2.l, a, a.mul(adj, a.y1(n2))); return; }
    if (k1 == AD_UN && k2 == AD_VAR) { hdrpass(p, t1.l, n2, a, a.mul(adj, a.y1(n1))); return; }
    if (k1 == AD_BIN && k2 == AD_BIN) {
        hdrpass(p, t1.l, t2.l, a, a.mul(a.mul(adj, a.y1(n1)), a.y1(n2)));
        hdrpass(p, t1.l, t2.r, a, a.mul(a.mul(adj, a.y1(n1)), a.y2(n2)));
        hdrpass(p, t1.r, t2.l, a, a.mul(a.mul(adj, a.y2(n1)), a.y1(n2)));
        hdrpass(p, t1.r, t2.r, a, a.mul(a.mul(adj, a.y2(n1)), a.y2(n2)));
        return;
    }
    if (k1 == AD_UN && k2 == AD_BIN) {
        hdrpass(p, t1.l, t2.l, a, a.mul(a.mul(adj, a.y1(n1)), a.y1(n2)));
        hdrpass(p, t1.l, t2.r, a, a.mul(a.mul(adj, a.y1(n1)), a.y2(n2)));
        return;
    }
    if (k1 == AD_BIN && k2 == AD_UN) {
        hdrpass(p, t1.l, t2.l, a, a.mul(a.mul(adj, a.y1(n1)), a.y1(n2)));
        hdrpass(p, t1.r, t2.l, a, a.mul(a.mul(adj, a.y2(n1)), a.y1(n2)));
        return;
    }
    if (k1 == AD_VAR && k2 == AD_BIN) {
        hdrpass(p, n1, t2.l, a, a.mul(adj, a.y1(n2)));
        hdrpass(p, n1, t2.r, a, a.mul(adj, a.y2(n2)));
        return;
    }
    if (k1 == AD_BIN && k2 == AD_VAR) {
        hdrpass(p, t1.l, n2, a, a.mul(adj, a.y1(n1)));
        hdrpass(p, t1.r, n2, a, a.mul(adj, a.y2(n1)));
        return;
    }
    a.leaf2(n1, n2, adj, true);   // VAR x VAR
}

template <class Alg>
void hrpass(const Pattern &p, int n, Alg &a, typename Alg::T adj, typename Alg::T adj2) {
    const ADNode &t = p.ad[n];
    switch (t.kind) {
    case AD_CONST: case AD_NULL: return;
    case AD_UN: {
        auto y = a.y1(n);
        hrpass(p, t.l, a, a.mul(adj, y), a.add(a.mul(adj2, a.mul(y, y)), a.mul(adj, a.h11(n))));
        return;
    }
    case AD_BIN: {
        auto y1 = a.y1(n), y2 = a.y2(n);
        auto adj2y1y2 = a.mul(a.mul(adj2, y1), y2);
        auto adjh12 = a.mul(adj, a.h12(n));
        hrpass(p, t.l, a, a.mul(adj, y1), a.add(a.mul(adj2, a.mul(y1, y1)), a.mul(adj, a.h11(n))));
        hrpass(p, t.r, a, a.mul(adj, y2), a.add(a.mul(adj2, a.mul(y2, y2)), a.mul(adj, a.h22(n))));
        hdrpass(p, t.l, t.r, a, a.add(adj2y1y2, adjh12));
        return;
    }
    case AD_VAR: a.leaf2(n, n, adj2, false); return;
    }
}

template <class Alg>
void hrpass0(const Pattern &p, int n, Alg &a, typename Alg::T adj, typename Alg::T adj2) {
    const ADNode &t = p.ad[n];
    if (t.kind == AD_UN && t.fixed != FX_NONE) {
        if (t.fn == EXA_B_MUL) { auto y = a.y1(n); hrpass0(p, t.l, a, a.mul(adj, y), a.mul(adj2, a.mul(y, y))); return; }
        if (t.fn == EXA_B_ADD) { hrpass0(p, t.l, a, adj, adj2); return; }
        if (t.fn == EXA_B_SUB) { hrpass0(p, t.l, a, t.fixed == FX_FIRST ? a.neg(adj) : adj, adj2); return; }
    } else if (t.kind == AD_UN) {
        if (t.fn == EXA_U_PLUS) { hrpass0(p, t.l, a, adj, adj2); return; }
        if (t.fn == EXA_U_MINUS) { hrpass0(p, t.l, a, a.neg(adj), adj2); return; }
    } else if (t.kind == AD_BIN) {
        if (t.fn == EXA_B_ADD) { hrpass0(p, t.l, a, adj, adj2); hrpass0(p, t.r, a, adj, adj2); return; }
        if (t.fn == EXA_B_SUB) { hrpass0(p, t.l, a, adj, adj2); hrpass0(p, t.r, a, a.neg(adj), adj2); return; }
    } else if (t.kind == AD_VAR) {
        return;   // a linear term has no Hessian slot (hessian.jl:494-505)
    }
    hrpass(p, n, a, adj, adj2);
}

}  // namespace exa
