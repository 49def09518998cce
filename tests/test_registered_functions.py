"""User-registered functions: exa_register_univariate / exa_register_bivariate (include/exahip.h) — the reference's
`@register_univariate` / `@register_bivariate` (src/register.jl:56-74, 123-276) with the rules as HIP device expressions.

How they are checked without teaching the test oracle new functions: a function registered with the rules of a TABLE entry (`mysin` =
sin, `myhyp` = hypot, ...) must give, inside any tree, the same slot maps as the built-in node (the layout depends on the node TYPE being
univariate / bivariate, never on the function) and the same values to rounding; the oracle evaluates the built-in twin.
CPU: registration semantics, planning, generated text.  `-m gpu`: values through the C ABI."""
import numpy as np
import pytest

from conftest import has_gpu

needs_gpu = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def _fns():
    from exahip import graph as G
    mysin = G.register_univariate("mysin", "sin($1)", "cos($1)", "-$2", py=np.sin)
    myexp = G.register_univariate("myexp", "exp($1)", "$2", "$3")
    mycube = G.register_univariate("mycube", "exa_user_cube($1)", "3.0 * $1 * $1", "6.0 * $1",
                                   helpers="static __device__ __forceinline__ double exa_user_cube(double t) { return t * t * t; }")
    myhyp = G.register_bivariate("myhyp", "hypot($1, $2)", "$1 / $3", "$2 / $3", "($3 * $3 - $1 * $1) / ($3 * $3 * $3)",
                                 "-($1 * $2) / ($3 * $3 * $3)", "($3 * $3 - $2 * $2) / ($3 * $3 * $3)")
    mymul = G.register_bivariate("mymul", "$1 * $2", "$2", "$1", "=0", "=1", "=0")
    return mysin, myexp, mycube, myhyp, mymul


def _fused_fns():
    """The one-statement form (exa_register_univariate_fused): value and both derivatives from one evaluation."""
    from exahip import graph as G
    fsin = G.register_univariate("fsin", fused="exa_sincos($1, &$2, &$3); $4 = -$2;", py=np.sin)
    fexp = G.register_univariate("fexp", fused="$2 = exa_exp($1); $3 = $2; $4 = $2")          # no trailing semicolon: the generator adds it
    return fsin, fexp


def _lv_pair(n=50):
    from exahip import ExaCore, graph as G, rng
    from exahip.models import lv_x0
    fsin, fexp = _fused_fns()

    def build(sin, exp):
        c = ExaCore()
        x = c.add_var(n, start=lv_x0(n))
        c.add_con(lambda i: 3 * x[i + 1] ** 3 + 2 * x[i + 2] - 5 + sin(x[i + 1] - x[i + 2]) * sin(x[i + 1] + x[i + 2]) + 4 * x[i + 1]
                  - x[i] * exp(x[i] - x[i + 1]) - 3, rng(1, n - 2))
        c.add_obj(lambda i: 100 * (x[i - 1] ** 2 - x[i]) ** 2 + (x[i - 1] - 1) ** 2 + exp(sin(x[i])) * sin(x[i]), rng(2, n))
        return c

    return build(fsin, fexp), build(G.sin, G.exp)


def _pair(n=60):
    """The same model twice: with registered functions and with their built-in twins."""
    from exahip import ExaCore, Table, graph as G, rng
    mysin, myexp, mycube, myhyp, mymul = _fns()

    def build(sin, exp, cube, hyp, mul):
        c = ExaCore()
        x = c.add_var(n, start=np.linspace(0.4, 1.4, n))
        th = c.add_par(2, value=[1.5, 0.25])
        tab = Table(i=np.arange(1, n - 1, 2), j=np.arange(3, n + 1, 2)[: len(np.arange(1, n - 1, 2))], w=np.linspace(0.5, 2.0, len(np.arange(1, n - 1, 2))))
        c.add_obj(lambda i: sin(x[i] - x[i + 1]) * exp(-x[i]) + cube(x[i + 1]) + hyp(x[i], x[i + 1] * th[1]), rng(1, n - 1))
        g = c.add_con(lambda i: mul(sin(x[i]), exp(x[i + 1] * 0.5)) + hyp(x[i + 2], 2.0) + hyp(th[2], cube(x[i])) + mul(3.0, x[i]) * x[i + 1], rng(1, n - 2))
        c.add_con(lambda d: d.w * sin(x[d.i] * x[d.j]) + cube(exp(x[d.i]) - x[d.j]), tab)
        c.add_con_aug(g, lambda k: (k, mul(x[k], x[k + 4]) + sin(x[k + 2])), rng(1, 5))
        return c

    user = build(mysin, myexp, mycube, myhyp, mymul)
    twin = build(G.sin, G.exp, lambda t: t ** 3, G.hypot, lambda a, b: a * b)
    return user, twin


def test_registration_semantics(libs):
    from exahip import capi, graph as G
    L = capi.lib()
    a = L.exa_register_univariate(b"reg_twice", b"tanh($1)", b"1.0 - $2 * $2", b"-2.0 * $2 * $3", None)
    assert a >= 1000 and a == L.exa_register_univariate(b"reg_twice", b"tanh($1)", b"1.0 - $2 * $2", b"-2.0 * $2 * $3", None)   # same rules: same id
    assert L.exa_register_univariate(b"reg_twice", b"tanh($1)", b"1.0", b"=0", None) == -1 and b"already registered" in L.exa_last_error()
    assert L.exa_register_univariate(b"bad name", b"$1", b"=1", b"=0", None) == -1 and b"identifier" in L.exa_last_error()
    assert L.exa_register_univariate(b"needs_rule", b"$1", b"", b"=0", None) == -1 and b"needs an expression" in L.exa_last_error()
    assert L.exa_register_univariate(b"bad_ph", b"$1 + $2", b"=1", b"=0", None) == -1 and b"placeholder" in L.exa_last_error()   # f has no $2
    assert L.exa_register_bivariate(b"bad_ph2", b"$1 * $2", b"$4", b"$1", b"=0", b"=1", b"=0", None) == -1
    assert L.exa_register_univariate(None, b"$1", b"=1", b"=0", None) == -1
    # a bare '$' at the end of a rule and a two-digit placeholder are refused at registration, not by the compiler at model build
    assert L.exa_register_univariate(b"bad_ph3", b"$1 + $", b"=1", b"=0", None) == -1 and b"placeholder" in L.exa_last_error()
    assert L.exa_register_univariate(b"bad_ph4", b"$1", b"$10", b"=0", None) == -1 and b"placeholder" in L.exa_last_error()
    b = L.exa_register_bivariate(b"reg_bin", b"$1 * $2", b"$2", b"$1", b"=0", b"=1", b"=0", None)
    assert b >= 1000
    with pytest.raises(ValueError):
        G.register_univariate("reg_twice", "tanh($1)", "1.0", "=0")


def test_unregistered_ids_are_refused_and_registered_ones_plan_like_their_twins(libs):
    from exahip import ExaModel
    import ctypes
    user, twin = _pair()
    mu, mt = ExaModel(user, device=False), ExaModel(twin, device=False)
    assert (mu.meta.nvar, mu.meta.ncon, mu.meta.nnzj, mu.meta.nnzh) == (mt.meta.nvar, mt.meta.ncon, mt.meta.nnzj, mt.meta.nnzh)
    for k in range(4):
        a, b = mu.pattern_info(k), mt.pattern_info(k)
        assert a == b, (k, a, b)
    src = mu.kernel_source()
    assert "exa_user_cube" in src and "user-registered function `mycube`" in src and "hypot(" in src
    assert "exa_user_cube" not in mt.kernel_source()
    mu.compile()                                # the rules compile for gfx950 (hipcc / hiprtc, no device needed)
    # an id nobody registered
    from exahip import ExaCore, graph as G, rng
    G.UN_ID["ghost"] = 4999
    try:
        c = ExaCore()
        x = c.add_var(4)
        c.add_obj(lambda i: G.Node1("ghost", x[i]), rng(1, 4))
        with pytest.raises(Exception, match="unknown univariate function"):
            ExaModel(c, device=False)
    finally:
        del G.UN_ID["ghost"]


def test_fused_registration(libs):
    from exahip import ExaModel, capi
    L = capi.lib()
    assert L.exa_register_univariate_fused(b"fz", b"$2 = $1;", None) == -1 and b"must use $1 and assign $2, $3 and $4" in L.exa_last_error()
    assert L.exa_register_univariate_fused(b"fz", b"$2 = $1; $3 = 1.0; $4 = $5;", None) == -1 and b"placeholder" in L.exa_last_error()
    user, twin = _lv_pair()
    mu, mt = ExaModel(user, device=False), ExaModel(twin, device=False)
    assert [mu.pattern_info(k) for k in range(2)] == [mt.pattern_info(k) for k in range(2)]
    src = mu.kernel_source()
    def body(name):
        i = src.index(name + "(")
        return src[i: src.index("\n}\n", i)]
    # one statement per DISTINCT argument: a - b and a + b in the constraint; sin(x[i]) appears twice in the objective -> once
    assert body("g0_hess").count("exa_sincos(") == 2 and body("g1_hess").count("exa_sincos(") == 1
    mu.compile()
    from exahip import Recipe
    again = Recipe(Recipe(user).bytes).instantiate(device=False)          # the wire format carries the fused statement
    assert again.kernel_source() == src


def test_a_rule_that_does_not_compile_fails_the_build_with_the_compiler_message(libs):
    from exahip import ExaCore, ExaModel, graph as G, rng
    broken = G.register_univariate("broken_rule", "no_such_function($1)", "=1", "=0")
    c = ExaCore()
    x = c.add_var(4)
    c.add_obj(lambda i: broken(x[i]), rng(1, 4))
    m = ExaModel(c, device=False)
    with pytest.raises(Exception, match="no_such_function"):
        m.compile()


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_fused_registered_functions_equal_the_table_on_hip(libs):
    """fsin's statement is the table's own (one exa_sincos): the LV-shaped model must agree with the built-in one to the last bit in
    cons / jac, and to rounding in the Hessian (fexp's second derivative is the value here, the table's is written the same way)."""
    from exahip import ExaModel
    user, twin = _lv_pair(2000)
    mu, mt = ExaModel(user), ExaModel(twin)
    x = np.asarray(mu.meta.x0) + 0.05 * np.random.default_rng(0).uniform(-1, 1, mu.meta.nvar)
    y = np.random.default_rng(1).standard_normal(mu.meta.ncon)
    assert mu.obj(x) == pytest.approx(mt.obj(x), rel=1e-14)
    assert np.array_equal(mu.cons(x), mt.cons(x)) and np.array_equal(mu.jac_coord(x), mt.jac_coord(x))
    np.testing.assert_allclose(mu.grad(x), mt.grad(x), rtol=1e-13, atol=1e-300)
    np.testing.assert_allclose(mu.hess_coord(x, y, 0.7), mt.hess_coord(x, y, 0.7), rtol=1e-12, atol=1e-300)
    import oracle
    o = oracle.OracleModel(ExaModel(twin, device=False).ir)
    np.testing.assert_allclose(mu.hess_coord(x, y, 0.7), o.hess_coord(x, y, 0.7), rtol=1e-10, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_registered_functions_equal_their_builtin_twins_on_hip(libs):
    from exahip import ExaModel
    import oracle
    user, twin = _pair()
    m = ExaModel(user)
    o = oracle.OracleModel(ExaModel(twin, device=False).ir)        # the oracle never sees a user function: it evaluates the twin
    x = np.asarray(m.meta.x0) + 0.05 * np.random.default_rng(0).uniform(-1, 1, m.meta.nvar)
    y = np.random.default_rng(1).standard_normal(m.meta.ncon)
    v = np.random.default_rng(2).standard_normal(m.meta.nvar)
    w = np.random.default_rng(3).standard_normal(m.meta.ncon)
    for a, b in zip(m.jac_structure() + m.hess_structure(), o.jac_structure() + o.hess_structure()):
        assert np.array_equal(a, b)

    def rel(a, b):
        a, b = np.asarray(a, float), np.asarray(b, float)
        return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3 * max(1.0, float(np.max(np.abs(b)))))))

    assert abs(m.obj(x) - o.obj(x)) <= 1e-12 * abs(o.obj(x))
    for name, a, b in (("cons", m.cons(x), o.cons(x)), ("grad", m.grad(x), o.grad(x)), ("jac", m.jac_coord(x), o.jac_coord(x)),
                       ("hess", m.hess_coord(x, y, 0.7), o.hess_coord(x, y, 0.7)), ("jprod", m.jprod(x, v), o.jprod(x, v)),
                       ("jtprod", m.jtprod(x, w), o.jtprod(x, w)), ("hprod", m.hprod(x, y, v, 0.7), o.hprod(x, y, v, 0.7))):
        assert rel(a, b) <= 1e-12, (name, rel(a, b))
    # the fused sweep and the compressed COO go through the same rules
    import torch
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    f, g, c, jac, hess = m.eval_all(xd, yd, 0.7)
    torch.cuda.synchronize()
    assert rel(hess.cpu().numpy(), o.hess_coord(x, y, 0.7)) <= 1e-12 and rel(g.cpu().numpy(), o.grad(x)) <= 1e-12
    from exahip import CompressedExaModel
    ch = CompressedExaModel(m).hess_coord(xd, yd, 0.7)          # duplicate-summed COO: same total
    torch.cuda.synchronize()
    assert abs(float(ch.sum()) - float(o.hess_coord(x, y, 0.7).sum())) <= 1e-10 * float(np.abs(o.hess_coord(x, y, 0.7)).sum())


_LOADER = r"""
import sys, numpy as np
sys.path[:0] = [{pkg!r}, {tests!r}]
from exahip import ExaCore, ExaModel, Recipe, capi, graph as G, rng
L = capi.lib()
# this process registers OTHER functions first, so every id of the file's writer means something else here ...
for k in range(7):
    assert L.exa_register_univariate(b"decoy%d" % k, b"cosh($1)", b"sinh($1)", b"$2", None) == 1000 + k
    assert L.exa_register_bivariate(b"decoyb%d" % k, b"$1 - $2", b"=1", b"=-1", b"=0", b"=0", b"=0", None) == 1000 + k
# ... and it has not said that model files may bring device code: the file is refused, nothing of it is registered
try:
    Recipe.load({path!r})
    raise SystemExit("a file carrying device code was accepted without opt-in")
except capi.ExaHipError as e:
    assert "carries device code" in str(e) and "mysin" in str(e), str(e)
assert L.exa_user_function(0, 1007, 0, None, 0) == -1 and L.exa_user_function(1, 1007, 0, None, 0) == -1
assert L.exa_recipe_trust_code(-1) == 0 and L.exa_recipe_trust_code(1) == 0 and L.exa_recipe_trust_code(-1) == 1
rec = Recipe.load({path!r})
m = rec.instantiate(device=False)
src = m.kernel_source()
assert "exa_user_cube" in src and "hypot(" in src and "cosh(" not in src, "the file's own rules, not the decoys"
names = set()
for biv in (0, 1):
    for fn in range(1000, 1020):
        n = L.exa_user_function(biv, fn, 0, None, 0)
        if n >= 0:
            import ctypes
            b = ctypes.create_string_buffer(n + 1); L.exa_user_function(biv, fn, 0, b, n + 1); names.add(b.value.decode())
assert {{"mysin", "myexp", "mycube", "myhyp", "mymul"}} <= names, names
print("KSRC", __import__("hashlib").sha256(src.encode()).hexdigest())
print("INFO", [m.pattern_info(k) for k in range(4)])
# ... and a name the file defines, already taken here by other rules, refuses the file
"""


def test_trusted_load_is_per_call_and_leaves_the_process_setting_alone(libs, tmp_path):
    """exa_recipe_load_trusted: what a packed library's loader uses for its own embedded recipe — device code in the bytes is
    accepted for that ONE call; the process-wide setting stays what it was (unset here: the environment still decides afterwards), and the
    same bytes through plain exa_recipe_load are still refused."""
    import os, subprocess, sys
    from exahip import Recipe
    user, _ = _pair()
    path = str(tmp_path / "user.exarcp")
    Recipe(user).save(path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys
sys.path[:0] = [{os.path.join(root, 'examodels.jl_amd')!r}]
from exahip import capi
L = capi.lib()
raw = open({path!r}, 'rb').read()
assert L.exa_recipe_trust_code(-1) == 0
assert L.exa_recipe_load(raw, len(raw)) == 0 and b'carries device code' in L.exa_last_error()
rid = L.exa_recipe_load_trusted(raw, len(raw))
assert rid > 0, L.exa_last_error()
assert L.exa_recipe_trust_code(-1) == 0                      # untouched: not latched to 0 or 1
import os
os.environ['EXAHIP_TRUST_MODEL_CODE'] = '1'                  # ... so the environment still decides afterwards
assert L.exa_recipe_trust_code(-1) == 1
print('OK')
"""
    env = {k: v for k, v in os.environ.items() if k != "EXAHIP_TRUST_MODEL_CODE"}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]


def test_a_model_file_carries_its_registered_functions(libs, tmp_path):
    """A recipe / model file that uses registered functions loads into a process that never registered them (or registered others
    first): the trailing section of the wire format (include/exahip_recipe.h) registers them at load and renumbers the nodes."""
    import os, subprocess, sys, hashlib
    from exahip import ExaModel, Recipe
    user, _ = _pair()
    path = str(tmp_path / "user.exarcp")
    Recipe(user).save(path)
    here = ExaModel(user, device=False)
    # the same process: load == direct build (ids already registered: same rules, same ids)
    again = Recipe.load(path).instantiate(device=False)
    assert again.kernel_source() == here.kernel_source()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _LOADER.format(pkg=os.path.join(root, "examodels.jl_amd"), tests=os.path.join(root, "tests"), path=path)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = dict(l.split(" ", 1) for l in out.stdout.splitlines() if l.startswith(("KSRC", "INFO")))
    # same generated text as here except for nothing: the rule texts are what the generator emits, ids never appear in the source
    assert lines["KSRC"] == hashlib.sha256(here.kernel_source().encode()).hexdigest()
    assert lines["INFO"] == str([here.pattern_info(k) for k in range(4)])


def test_a_model_file_whose_function_name_is_taken_by_other_rules_is_refused(libs, tmp_path):
    import os, subprocess, sys
    from exahip import Recipe
    user, _ = _pair()
    path = str(tmp_path / "user.exarcp")
    Recipe(user).save(path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys
sys.path[:0] = [{os.path.join(root, "examodels.jl_amd")!r}]
from exahip import Recipe, capi
L = capi.lib()
assert L.exa_register_bivariate(b"mymul", b"$1 + $2", b"=1", b"=1", b"=0", b"=0", b"=0", None) >= 1000      # NOT the file's mymul
try:
    Recipe.load({path!r})
except capi.ExaHipError as e:
    assert "already registered with other rules" in str(e), str(e)
    # ... and the refused file left nothing behind: its univariate entries come BEFORE mymul in the section, none is registered
    assert L.exa_user_function(0, 1000, 0, None, 0) == -1 and L.exa_user_function(1, 1001, 0, None, 0) == -1
    print("REFUSED")
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "REFUSED" in out.stdout, out.stderr[-2000:]


def test_a_model_file_with_code_needs_an_opt_in_unless_the_process_registered_the_same_rules(libs, tmp_path):
    """the trailing section is HIP device source.  Refused by default; accepted when every entry is, rule for rule, a
    registration the process made itself (this process: _pair registered them), under exa_recipe_trust_code(1), or under
    EXAHIP_TRUST_MODEL_CODE=1.  A file WITHOUT the section that names a registered id is refused (the id is the writer's)."""
    import os, struct, subprocess, sys
    from exahip import Recipe, capi
    user, _ = _pair()
    path = str(tmp_path / "user.exarcp")
    Recipe(user).save(path)
    L = capi.lib()
    assert L.exa_recipe_trust_code(-1) == 0
    Recipe.load(path)                     # same rules already registered here: nothing new would be compiled
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys
sys.path[:0] = [{os.path.join(root, "examodels.jl_amd")!r}]
from exahip import Recipe, capi
try:
    Recipe.load({path!r})
    print("ACCEPTED")
except capi.ExaHipError as e:
    print("REFUSED" if "carries device code" in str(e) else "OTHER " + str(e))
"""
    for env, want in (({}, "REFUSED"), ({"EXAHIP_TRUST_MODEL_CODE": "1"}, "ACCEPTED"), ({"EXAHIP_TRUST_MODEL_CODE": "0"}, "REFUSED")):
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        assert out.returncode == 0 and out.stdout.strip() == want, (env, out.stdout, out.stderr[-2000:])
    # the section cut off (and nothing else changed): nodes with fn >= 1000 and no definition
    good = Recipe(user).bytes
    k = good.rindex(struct.pack("<i", 5) + struct.pack("<ii", 0, 1000)) if struct.pack("<i", 5) + struct.pack("<ii", 0, 1000) in good else None
    if k is not None:
        with pytest.raises(capi.ExaHipError, match="does not define"):
            Recipe(good[:k])


def test_generated_text_does_not_depend_on_the_order_of_registration(libs):
    """ids are per process; the module's text (its cache key) must not carry them, nor their order."""
    import hashlib, os, subprocess, sys
    from exahip import ExaModel
    user, _ = _pair()
    here = hashlib.sha256(ExaModel(user, device=False).kernel_source().encode()).hexdigest()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys, hashlib
sys.path[:0] = [{os.path.join(root, "examodels.jl_amd")!r}, {os.path.join(root, "tests")!r}]
from exahip import ExaModel, graph as G
# the same five functions, registered in the REVERSE order (other ids)
G.register_bivariate("mymul", "$1 * $2", "$2", "$1", "=0", "=1", "=0")
G.register_bivariate("myhyp", "hypot($1, $2)", "$1 / $3", "$2 / $3", "($3 * $3 - $1 * $1) / ($3 * $3 * $3)", "-($1 * $2) / ($3 * $3 * $3)", "($3 * $3 - $2 * $2) / ($3 * $3 * $3)")
G.register_univariate("mycube", "exa_user_cube($1)", "3.0 * $1 * $1", "6.0 * $1", helpers="static __device__ __forceinline__ double exa_user_cube(double t) {{ return t * t * t; }}")
G.register_univariate("myexp", "exp($1)", "$2", "$3")
import test_registered_functions as T
user, _ = T._pair()
print("KSRC", hashlib.sha256(ExaModel(user, device=False).kernel_source().encode()).hexdigest())
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert [l for l in out.stdout.splitlines() if l.startswith("KSRC")][0].split()[1] == here


def test_user_function_section_is_validated(libs):
    """Malformed trailing sections are refused, never half-applied: truncated text, an id below 1000, a node whose function the section
    does not define, bytes after the section."""
    import struct
    from exahip import Recipe, capi
    user, _ = _pair()
    good = Recipe(user).bytes
    Recipe(good)
    for bad, why in ((good[:-3], "truncated"), (good + b"\0", "trailing")):
        with pytest.raises(capi.ExaHipError):
            Recipe(bad)
    # drop the last entry of the section (and fix the count): a node now uses an undefined function
    from exahip import recipe as R
    L = capi.lib()
    ids = sorted({(b, f) for b in (0, 1) for f in range(1000, 1100) if L.exa_user_function(b, f, 0, None, 0) >= 0})
    texts = lambda b, f: b"".join((lambda t: struct.pack("<i", len(t)) + t)(R._user_text(b, f, w).encode()) for w in range(9))
    used = [(b, f) for b, f in ids if struct.pack("<ii", b, f) + texts(b, f) in good]
    assert len(used) == 5
    tail = b"".join(struct.pack("<ii", b, f) + texts(b, f) for b, f in used)
    assert good.endswith(struct.pack("<i", 5) + tail)
    head = good[: -len(struct.pack("<i", 5) + tail)]
    short = head + struct.pack("<i", 4) + b"".join(struct.pack("<ii", b, f) + texts(b, f) for b, f in used[:-1])
    with pytest.raises(capi.ExaHipError, match="does not define"):
        Recipe(short)
    low = head + struct.pack("<i", 5) + tail.replace(struct.pack("<ii", *used[0]), struct.pack("<ii", used[0][0], 7), 1)
    with pytest.raises(capi.ExaHipError, match="bad user-function entry"):
        Recipe(low)


_CONSUMER = r"""
import ctypes, json, sys, numpy as np
lib = ctypes.CDLL({path!r})                  # nothing but the library: this process never registers a function
vp = ctypes.c_void_p
assert lib.cnlp_nmodels() == 1 and lib.userm_nargs() == 0
out = {{"loaded": True}}
if {gpu}:
    mid = lib.userm_new(0)
    assert mid > 0
    nvar, ncon, nnzh = lib.userm_nvar(mid), lib.userm_ncon(mid), lib.userm_nnzh(mid)
    x = np.linspace(0.5, 1.5, nvar); y = np.linspace(-1.0, 1.0, ncon)
    f = np.zeros(1); c = np.zeros(ncon); h = np.zeros(nnzh)
    assert lib.userm_obj(mid, vp(x.ctypes.data), vp(f.ctypes.data)) == 0
    assert lib.userm_cons(mid, vp(x.ctypes.data), vp(c.ctypes.data)) == 0
    assert lib.userm_hess(mid, vp(x.ctypes.data), vp(y.ctypes.data), ctypes.c_double(0.5), vp(h.ctypes.data)) == 0
    out.update(obj=float(f[0]), cons=c.tolist(), hess=h.tolist())
print("RESULT " + json.dumps(out))
"""


def _packed_consumer(tmp_path, gpu):
    import json, subprocess, sys
    from exahip.pack import pack_library
    user, twin = _pair()
    path = pack_library(str(tmp_path / "userlib"), ("userm", user))
    out = subprocess.run([sys.executable, "-c", _CONSUMER.format(path=path, gpu=gpu)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0][7:]), twin


def test_a_packed_library_carries_its_registered_functions(libs, tmp_path):
    """exahip.pack embeds the recipe bytes: the consumer process loads `libuserlib.so` alone and finds the model (its catalogue entry parses
    only if the loader accepted and registered the functions of the trailing section)."""
    res, _ = _packed_consumer(tmp_path, False)
    assert res == {"loaded": True}


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_a_packed_library_with_registered_functions_evaluates_in_a_fresh_process(libs, tmp_path):
    from exahip import ExaModel
    import oracle
    res, twin = _packed_consumer(tmp_path, True)
    o = oracle.OracleModel(ExaModel(twin, device=False).ir)
    x, y = np.linspace(0.5, 1.5, o.nvar), np.linspace(-1.0, 1.0, o.ncon)
    np.testing.assert_allclose(res["obj"], o.obj(x), rtol=1e-12)
    np.testing.assert_allclose(res["cons"], o.cons(x), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(res["hess"], o.hess_coord(x, y, 0.5), rtol=1e-10, atol=1e-12)


def _random_twins():
    import randexpr
    return randexpr.registered_twins()


def _random_pair(seed):
    import randexpr
    return randexpr.build_model(9000 + seed, npat=8, depth=4, user=_random_twins()), randexpr.build_model(9000 + seed, npat=8, depth=4)


@pytest.mark.parametrize("seed", range(6))
def test_random_trees_with_registered_twins_plan_like_the_table(libs, seed):
    from exahip import ExaModel
    user, twin = _random_pair(seed)
    mu, mt = ExaModel(user, device=False), ExaModel(twin, device=False)
    assert mu.meta.nnzj == mt.meta.nnzj and mu.meta.nnzh == mt.meta.nnzh
    assert [mu.pattern_info(k) for k in range(8)] == [mt.pattern_info(k) for k in range(8)]
    assert "user-registered function `rt_" in mu.kernel_source()
    if seed == 1:
        mu.compile()


def test_fused_statements_with_temporaries_do_not_collide(libs):
    """rt_cos declares a temporary (`double s_;`): two uses with different arguments in one kernel compile, each statement in its own block."""
    from exahip import ExaCore, ExaModel, rng
    cos = _random_twins()["cos"]
    c = ExaCore()
    x = c.add_var(10, start=0.5)
    c.add_obj(lambda i: cos(x[i]) * cos(x[i + 1]) + cos(x[i] * x[i + 1]), rng(1, 9))
    m = ExaModel(c, device=False)
    i = m.kernel_source().index("g0_hess(")
    assert m.kernel_source()[i: m.kernel_source().index("\n}\n", i)].count("double s_;") == 3
    m.compile()


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
@pytest.mark.parametrize("seed", range(6))
def test_random_trees_with_registered_twins_on_hip(libs, seed):
    """Registered functions at random places of random trees (under other functions, as fixed operands' partners, in shared augmentation
    rows): every callback against the oracle, which evaluates the table's tree."""
    from exahip import ExaModel
    import oracle
    user, twin = _random_pair(seed)
    m = ExaModel(user)
    o = oracle.OracleModel(ExaModel(twin, device=False).ir)
    x = np.asarray(m.meta.x0) + 0.02 * np.random.default_rng(seed).uniform(-1, 1, m.meta.nvar)
    y = np.random.default_rng(seed + 1).standard_normal(m.meta.ncon)
    v = np.random.default_rng(seed + 2).standard_normal(m.meta.nvar)
    w = np.random.default_rng(seed + 3).standard_normal(m.meta.ncon)
    for a, b in zip(m.jac_structure() + m.hess_structure(), o.jac_structure() + o.hess_structure()):
        assert np.array_equal(a, b)

    def rel(a, b):
        a, b = np.asarray(a, float), np.asarray(b, float)
        return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3 * max(1.0, float(np.max(np.abs(b)))))))

    assert abs(m.obj(x) - o.obj(x)) <= 1e-10 * max(1.0, abs(o.obj(x)))
    for name, a, b in (("cons", m.cons(x), o.cons(x)), ("grad", m.grad(x), o.grad(x)), ("jac", m.jac_coord(x), o.jac_coord(x)),
                       ("hess", m.hess_coord(x, y, 0.7), o.hess_coord(x, y, 0.7)), ("jprod", m.jprod(x, v), o.jprod(x, v)),
                       ("jtprod", m.jtprod(x, w), o.jtprod(x, w)), ("hprod", m.hprod(x, y, v, 0.7), o.hprod(x, y, v, 0.7))):
        assert rel(a, b) <= 1e-10, (seed, name, rel(a, b))
