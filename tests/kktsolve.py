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


def newton_full_step(m, x0, ncon, iters):
    """What Ipopt does on an equality-constrained model without bounds when every step is accepted at alpha = 1 and the
    Hessian needs no regularisation (the three logs of docs/src/parameters.md:148-154, :207-213, :266-272 show exactly that:
    alpha_pr = alpha_du = 1, lg(rg) = '-'): least-squares initial multipliers  [I J'; J 0] [z; y0] = -[g; 0]  (Ipopt's
    default bound_mult_init / constr_mult_init_max = 1000 path), then the plain Newton step on the KKT conditions
    [H J'; J 0] [dx; dy] = -[g + J'y; c].  Returns (rows, x, y): one row (objective, inf_pr, inf_du, ||d||_inf) per iterate,
    0 .. iters, with inf_du the UNSCALED max |g + J'y| (Ipopt prints it multiplied by its objective scaling
    100 / max|grad f(x0)|, docs/src/parameters.md:159 shows the two columns)."""
    n = len(x0)
    x = np.array(x0, float)
    jr, jc = m.jac_structure()
    hr, hc = m.hess_structure()

    def first_order(z):
        g = np.asarray(m.grad(z), float)
        J = np.zeros((ncon, n))
        np.add.at(J, (np.asarray(jr) - 1, np.asarray(jc) - 1), np.asarray(m.jac_coord(z), float))
        return g, J

    g, J = first_order(x)
    Z = np.zeros((ncon, ncon))
    y = np.linalg.solve(np.block([[np.eye(n), J.T], [J, Z]]), -np.concatenate([g, np.zeros(ncon)]))[n:]
    rows = [(float(m.obj(x)), float(np.max(np.abs(m.cons(x)))), float(np.max(np.abs(g + J.T @ y))), 0.0)]
    for _ in range(iters):
        c = np.asarray(m.cons(x), float)
        L = np.zeros((n, n))
        np.add.at(L, (np.asarray(hr) - 1, np.asarray(hc) - 1), np.asarray(m.hess_coord(x, y, 1.0), float))
        H = L + np.tril(L, -1).T
        d = np.linalg.solve(np.block([[H, J.T], [J, Z]]), -np.concatenate([g + J.T @ y, c]))
        x, y = x + d[:n], y + d[n:]
        g, J = first_order(x)
        rows.append((float(m.obj(x)), float(np.max(np.abs(m.cons(x)))), float(np.max(np.abs(g + J.T @ y))), float(np.max(np.abs(d[:n])))))
    return rows, x, y
