"""Test helper: a bare Newton-KKT iteration for equality-constrained models driven ONLY by the seven NLPModels
callbacks (the way the reference's test solver drives them, test/NLPModelsIpoptLite.jl/src/NLPModelsIpoptLite.jl:28-40).
Inertia-corrected, l1-merit backtracking.  Small dense/sparse problems only — it is a checker, not a product."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl


def newton_kkt(m, x0, ncon, tol=1e-10, maxit=60):
    n = len(x0)
    x, y = np.array(x0, float), np.zeros(ncon)
    jr, jc = m.jac_structure()
    hr, hc = m.hess_structure()

    def merit(z, mu):
        return m.obj(z) + mu * np.sum(np.abs(m.cons(z)))

    for it in range(maxit):
        g, c = np.asarray(m.grad(x)), np.asarray(m.cons(x))
        J = sp.coo_matrix((np.asarray(m.jac_coord(x)), (jr - 1, jc - 1)), shape=(ncon, n)).tocsr()
        L = sp.coo_matrix((np.asarray(m.hess_coord(x, y, 1.0)), (hr - 1, hc - 1)), shape=(n, n)).tocsr()
        H = L + sp.tril(L, -1).T
        if np.max(np.abs(np.concatenate([g + J.T @ y, c]))) < tol:
            return x, y, it
        delta = 0.0
        while True:
            K = sp.bmat([[H + delta * sp.eye(n), J.T], [J, None]]).tocsc()
            ev = np.linalg.eigvalsh(K.toarray())
            if np.sum(ev > 0) == n and np.sum(ev < 0) == ncon:
                break
            delta = max(1e-4, 10 * delta)
        d = spl.spsolve(K, -np.concatenate([g, c]))
        dx, ynew = d[:n], d[n:]
        mu = np.max(np.abs(ynew)) + 1.0
        a, m0, slope = 1.0, merit(x, mu), g @ dx - mu * np.sum(np.abs(c))
        while merit(x + a * dx, mu) > m0 + 1e-4 * a * slope and a > 1e-8:
            a *= 0.5
        x, y = x + a * dx, y + a * (ynew - y)
    raise AssertionError("Newton-KKT iteration did not converge")
