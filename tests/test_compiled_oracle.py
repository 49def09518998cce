"""oracle/compiled.py — the gcc-compiled straight-line restatement of hess_coord! that bench.py times as the CPU baseline of
configs 3 and 4 — against the tree-walking interpreter (oracle/exa_oracle.c) it is generated beside: same recursions
(hessian.jl:16-517), same slot maps, same zero-fill + `+=` order, so the values agree to rounding (bit for bit wherever
the two spell the arithmetic identically)."""
import numpy as np
import pytest

from zoo import ZOO, point


@pytest.mark.parametrize("name", ["lv20", "lv20_objfirst", "lv_split_20x2", "lv1000", "rocket50", "acopf30", "conaug2d",
                                  "stepped", "cops_chain", "cops_elec", "trivialmax"])
def test_compiled_hessian_equals_interpreter(libs, name):
    import compiled
    import oracle
    ir = ZOO[name]().to_ir()
    o = oracle.OracleModel(ir)
    ch = compiled.CompiledHess(ir, o)
    for seed in (0, 5):
        x, y, s = point(ir.x0, o.ncon, seed=seed)
        a, b = ch(x, y, s), o.hess_coord(x, y, s)
        np.testing.assert_allclose(a, b, rtol=1e-13, atol=1e-13 * max(1.0, float(np.max(np.abs(b))) if b.size else 1.0))
    if o.nnzh:      # OpenMP over data points: slots are private to a data point, the threads never meet
        assert np.array_equal(ch(x, y, s, threads=4), ch(x, y, s, threads=1))


def test_functions_outside_the_covered_set_are_refused(libs):
    import compiled
    import oracle
    ir = ZOO["mixed"]().to_ir()          # uses tanh
    with pytest.raises(NotImplementedError):
        compiled.CompiledHess(ir, oracle.OracleModel(ir))
