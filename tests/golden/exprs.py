"""Expression list for the derivative golden vectors.

Each entry is written ONCE as `f(x, th, F)` over an abstract function namespace F and 1-based indexables x, th;
it is instantiated (a) with exahip nodes to build a pattern and (b) with sympy symbols by make_golden.py, which
differentiates it symbolically and evaluates with 40-digit mpmath — an oracle independent of both the C test
oracle and the HIP code generator.

Rows 1-63 restate the reference's own AD test list (test/ADTest/ADTest.jl:6-121: FUNCTIONS and
PARAMETER_FUNCTIONS) minus the SpecialFunctions rows, which are SPECIAL_EXPRS below (golden vectors:
special_golden.json, make_special_golden.py); the remaining rows cover every other table entry of src/functionlist.jl plus
the benchmark patterns.
"""

EXPRS = [
    # --- ADTest.jl:6-58 "basic-functions" ---
    ("basic-plus", lambda x, th, F: +(x[1])),
    ("basic-minus", lambda x, th, F: -(x[1])),
    ("basic-inv", lambda x, th, F: F.inv(x[1])),
    ("basic-abs", lambda x, th, F: F.abs(x[1])),
    ("basic-sqrt", lambda x, th, F: F.sqrt(x[1])),
    ("basic-cbrt", lambda x, th, F: F.cbrt(x[1])),
    ("basic-abs2", lambda x, th, F: F.abs2(x[1])),
    ("basic-exp", lambda x, th, F: F.exp(x[1])),
    ("basic-exp2", lambda x, th, F: F.exp2(x[1])),
    ("basic-exp10", lambda x, th, F: F.exp10(x[1])),
    ("basic-log", lambda x, th, F: F.log(x[1])),
    ("basic-log2", lambda x, th, F: F.log2(x[1])),
    ("basic-log1p", lambda x, th, F: F.log1p(x[1])),
    ("basic-log10", lambda x, th, F: F.log10(x[1])),
    ("basic-sin", lambda x, th, F: F.sin(x[1])),
    ("basic-cos", lambda x, th, F: F.cos(x[1])),
    ("basic-tan", lambda x, th, F: F.tan(x[1])),
    ("basic-asin", lambda x, th, F: F.asin(x[1])),
    ("basic-acos", lambda x, th, F: F.acos(x[1])),
    ("basic-csc", lambda x, th, F: F.csc(x[1])),
    ("basic-sec", lambda x, th, F: F.sec(x[1])),
    ("basic-cot", lambda x, th, F: F.cot(x[1])),
    ("basic-atan", lambda x, th, F: F.atan(x[1])),
    ("basic-acot", lambda x, th, F: F.acot(x[1])),
    ("basic-cscd", lambda x, th, F: F.cscd(x[1])),
    ("basic-secd", lambda x, th, F: F.secd(x[1])),
    ("basic-cotd", lambda x, th, F: F.cotd(x[1])),
    ("basic-sinh", lambda x, th, F: F.sinh(x[1])),
    ("basic-asinh", lambda x, th, F: F.asinh(x[1])),
    ("basic-cosh", lambda x, th, F: F.cosh(x[1])),
    ("basic-acosh", lambda x, th, F: F.acosh(x[1] + 1)),
    ("basic-tanh", lambda x, th, F: F.tanh(x[1])),
    ("basic-csch", lambda x, th, F: F.csch(x[1])),
    ("basic-sech", lambda x, th, F: F.sech(x[1])),
    ("basic-coth", lambda x, th, F: F.coth(x[1])),
    ("basic-atanh", lambda x, th, F: F.atanh(x[1])),
    ("basic-add2", lambda x, th, F: x[1] + x[2]),
    ("basic-sub2", lambda x, th, F: x[1] - x[2]),
    ("basic-mul2", lambda x, th, F: x[1] * x[2]),
    ("basic-pow2", lambda x, th, F: x[1] ** x[2]),
    ("basic-div2", lambda x, th, F: x[1] / x[2]),
    # --- ADTest.jl:80-106 "composite-functions" without SpecialFunctions ---
    ("composite-1-2", lambda x, th, F: 0 * x[1]),
    ("composite-1-4", lambda x, th, F: (0 * x[1] ** x[3] ** 1.0 + x[1]) / x[9] / x[10]),
    ("composite-1-5", lambda x, th, F: F.exp(x[1] + 1.0) ** x[2] * F.log(F.abs2(x[3]) + 3) / F.tanh(x[2])),
    ("composite-1-10", lambda x, th, F: F.sin(1 / x[1])),
    ("composite-1-11", lambda x, th, F: F.exp(x[2]) / F.cos(x[1]) ** 2 + F.sin(x[1] ** 2)),
    ("composite-1-12", lambda x, th, F: F.sin(x[9] * F.inv(x[1]) - x[8] * F.inv(x[2]))),
    ("composite-1-13", lambda x, th, F: x[1] / F.log(x[2] ** 2 + 9.0)),
    # --- ADTest.jl:108-121 "parameter" rows without SpecialFunctions ---
    ("parameter-basic-1", lambda x, th, F: x[1] + th[1]),
    ("parameter-basic-2", lambda x, th, F: x[1] * th[1]),
    ("parameter-basic-3", lambda x, th, F: x[1] ** 2 + th[1] * x[2]),
    ("parameter-basic-4", lambda x, th, F: F.sin(x[1]) + F.cos(th[1])),
    ("parameter-basic-5", lambda x, th, F: F.exp(x[1] + th[1])),
    ("parameter-basic-6", lambda x, th, F: F.log(x[1] ** 2 + th[1] ** 2)),
    ("parameter-basic-7", lambda x, th, F: x[1] / (1 + th[1])),
    ("parameter-basic-8", lambda x, th, F: th[1] * F.sin(x[1]) + th[2] * F.cos(x[2])),
    ("parameter-composite-1", lambda x, th, F: F.exp(x[1] * th[1]) + F.sin(x[2] + th[2])),
    ("parameter-composite-3", lambda x, th, F: F.sqrt(x[1] ** 2 + th[1] ** 2) * F.log(x[2] + th[2] + 1)),
    # --- remaining entries of src/functionlist.jl:6-81 ---
    ("table-sign", lambda x, th, F: F.sign(x[1] - 0.5) * x[2]),
    ("table-expm1", lambda x, th, F: F.expm1(x[1] * x[2])),
    ("table-sind", lambda x, th, F: F.sind(40 * x[1] + x[2])),
    ("table-cosd", lambda x, th, F: F.cosd(40 * x[1] * x[2])),
    ("table-tand", lambda x, th, F: F.tand(30 * x[1] + x[2])),
    ("table-atand", lambda x, th, F: F.atand(x[1] * x[2])),
    ("table-acotd", lambda x, th, F: F.acotd(x[1] + x[2])),
    ("table-sinpi", lambda x, th, F: F.sinpi(x[1] * x[2])),
    ("table-cospi", lambda x, th, F: F.cospi(x[1] + 2 * x[2])),
    ("table-sinc", lambda x, th, F: F.sinc(x[1] + x[2])),
    ("table-deg2rad", lambda x, th, F: F.deg2rad(x[1]) * x[2]),
    ("table-rad2deg", lambda x, th, F: F.rad2deg(x[1]) * x[1]),
    ("table-acoth", lambda x, th, F: F.acoth(x[1] + 1.5) * x[2]),
    ("table-atan2", lambda x, th, F: F.atan(x[1] * x[3], x[2] + x[3])),
    ("table-hypot", lambda x, th, F: F.hypot(x[1] * x[2], x[3])),
    ("table-max", lambda x, th, F: F.maximum(x[1] * x[2], x[3] ** 2)),
    ("table-min", lambda x, th, F: F.minimum(x[1] * x[2], x[3] ** 2)),
    ("table-floor-ceil", lambda x, th, F: (F.floor(3 * x[1]) + F.ceil(3 * x[2])) * x[3] ** 2),
    # --- integer / real powers, fixed operands on either side ---
    ("pow-int3", lambda x, th, F: 3 * x[1] ** 3 + x[2] ** 5 - x[3] ** 7),
    ("pow-negint", lambda x, th, F: x[1] ** -2 + x[2] ** -1 + (x[1] * x[2]) ** -3),
    ("pow-real", lambda x, th, F: x[1] ** 1.5 * x[2] ** 0.25 + 2.0 ** x[3] + th[1] ** x[2]),
    ("div-fixed", lambda x, th, F: 2.0 / x[1] + x[2] / 3.0 + th[1] / (x[1] * x[2])),
    ("sub-fixed", lambda x, th, F: (1 - x[1]) * (x[2] - 2) - (3.0 - x[1] * x[2]) ** 2),
    ("neg-chain", lambda x, th, F: -(-(x[1] * x[2])) - (-x[3]) ** 2),
    # --- benchmark patterns at one data point ---
    ("lv-obj", lambda x, th, F: 100 * (x[1] ** 2 - x[2]) ** 2 + (x[1] - 1) ** 2),
    ("lv-con", lambda x, th, F: 3 * x[2] ** 3 + 2 * x[3] - 5 + F.sin(x[2] - x[3]) * F.sin(x[2] + x[3]) + 4 * x[2]
     - x[1] * F.exp(x[1] - x[2]) - 3),
    ("acopf-flow", lambda x, th, F: x[5] - th[1] * x[1] ** 2 - th[2] * (x[1] * x[2] * F.cos(x[3] - x[4]))
     - 0.7 * (x[1] * x[2] * F.sin(x[3] - x[4]))),
    ("rocket-vel", lambda x, th, F: -x[1] + x[2] + 0.5 * x[9] * (
        (x[3] - 310.0 * x[1] ** 2 * F.exp(-5.0 * (x[5] + 0.5 - 1.0) / 1.0) - (x[7] + 0.3) * 1.0 * (1.0 / (x[5] + 0.5)) ** 2) / (x[7] + 0.3)
        + (x[4] - 310.0 * x[2] ** 2 * F.exp(-5.0 * (x[6] + 0.5 - 1.0) / 1.0) - (x[8] + 0.3) * 1.0 * (1.0 / (x[6] + 0.5)) ** 2) / (x[8] + 0.3))),
]

# The SpecialFunctions rows of the reference's AD test list (test/ADTest/ADTest.jl:59-64, 80-106, 118-120) and one row per entry
# of ext/functionlist.jl:6-124 (composed with the variables so that gradient and Hessian have several entries).
SPECIAL_EXPRS = [
    # --- ADTest.jl:59-64 ---
    ("special-erfi", lambda x, th, F: F.erfi(x[1])),
    ("special-erfcinv", lambda x, th, F: F.erfcinv(x[1])),
    ("special-airybiprime", lambda x, th, F: F.airybiprime(x[1])),
    ("special-besselj0", lambda x, th, F: F.besselj0(x[1])),
    ("special-beta", lambda x, th, F: F.beta(x[1], x[2])),
    ("special-logbeta", lambda x, th, F: F.logbeta(x[1], x[2])),
    # --- ADTest.jl:80-106 "composite-functions" with SpecialFunctions ---
    ("composite-1-1", lambda x, th, F: F.beta(F.erf(x[1] / x[2] / 3.0) + 3.0 * x[2], F.erf(x[9]) ** 2)),
    ("composite-1-3", lambda x, th, F: F.beta(F.cos(F.log(F.abs2(F.inv(F.inv(x[1]))) + 1.0)), F.erfc(F.tanh(0 * x[1])))),
    ("composite-1-6", lambda x, th, F: F.beta(2 * F.logbeta(x[1], x[5]), F.beta(x[2], x[3]))),
    ("composite-1-7", lambda x, th, F: F.besselj0(F.exp(F.erf(-x[1])))),
    ("composite-1-8", lambda x, th, F: F.erfc(F.abs2(x[1] ** 2 / x[2]) ** x[9] / x[10])),
    ("composite-1-9", lambda x, th, F: F.erfc(x[1]) ** F.erf(2.5 * x[2])),
    ("composite-1-14", lambda x, th, F: F.beta(F.beta(F.tan(F.beta(x[1], 1) + 2.0), F.cos(F.sin(x[2]))), x[3])),
    # --- ADTest.jl:118-120 ---
    ("parameter-composite-2", lambda x, th, F: F.beta(x[1] + th[1], x[2] + th[2])),
    ("parameter-composite-4", lambda x, th, F: F.gamma(x[1] + 1) * th[1] + F.erf(x[2] * th[2])),
    # --- every entry of ext/functionlist.jl ---
    ("sf-erf", lambda x, th, F: F.erf(x[1] * x[2] - x[3])),
    ("sf-erfc", lambda x, th, F: F.erfc(2 * x[1] + x[2] ** 2)),
    ("sf-erfi", lambda x, th, F: F.erfi(3 * x[1] - x[2] * x[3])),
    ("sf-erfi-large", lambda x, th, F: F.erfi(8 * x[1] + 4 * x[2])),
    ("sf-erfcx", lambda x, th, F: F.erfcx(x[1] * 5 - x[2])),
    ("sf-erfcx-neg", lambda x, th, F: F.erfcx(-3 * x[1] * x[2])),
    ("sf-digamma", lambda x, th, F: F.digamma(x[1] + 3 * x[2])),
    ("sf-digamma-neg", lambda x, th, F: F.digamma(-2.0 - x[1] * x[2])),
    ("sf-trigamma", lambda x, th, F: F.trigamma(x[1] * x[2] + x[3])),
    ("sf-trigamma-neg", lambda x, th, F: F.trigamma(x[1] - 3.0 - x[2] / 4)),
    ("sf-invdigamma", lambda x, th, F: F.invdigamma(x[1] - 4 * x[2])),
    ("sf-invdigamma-pos", lambda x, th, F: F.invdigamma(3 * x[1] * x[2] + 1)),
    ("sf-gamma", lambda x, th, F: F.gamma(4 * x[1] + x[2])),
    ("sf-gamma-neg", lambda x, th, F: F.gamma(x[1] * x[2] - 1.5)),
    ("sf-airyai", lambda x, th, F: F.airyai(3 * x[1] - 5 * x[2])),
    ("sf-airyai-pos", lambda x, th, F: F.airyai(6 * x[1] + 3 * x[2])),
    ("sf-airyai-asym", lambda x, th, F: F.airyai(9 + 4 * x[1] * x[2]) * 1e6 + F.airyai(-10 - 3 * x[1] - x[2])),
    ("sf-airybi", lambda x, th, F: F.airybi(2 * x[1] - 6 * x[2])),
    ("sf-airybi-asym", lambda x, th, F: F.airybi(9.5 + x[1] * x[2]) * 1e-8 + F.airybi(-12 - x[1] + x[2])),
    ("sf-airyaiprime", lambda x, th, F: F.airyaiprime(4 * x[1] - 7 * x[2]) + F.airyaiprime(10 + x[3]) * 1e8 + F.airyaiprime(-11 - x[3])),
    ("sf-airybiprime", lambda x, th, F: F.airybiprime(x[1] - 6 * x[2]) + F.airybiprime(9 + x[3]) * 1e-9 + F.airybiprime(-10 - x[3])),
    ("sf-besselj0", lambda x, th, F: F.besselj0(10 * x[1] + x[2])),
    ("sf-bessely0", lambda x, th, F: F.bessely0(3 * x[1] + 8 * x[2])),
    ("sf-besselj1", lambda x, th, F: F.besselj1(7 * x[1] * x[2] + x[3])),
    ("sf-bessely1", lambda x, th, F: F.bessely1(x[1] + 12 * x[2])),
    ("sf-dawson", lambda x, th, F: F.dawson(2 * x[1] - 3 * x[2])),
    ("sf-dawson-large", lambda x, th, F: F.dawson(9 * x[1] + 5 * x[2])),
    ("sf-erfinv", lambda x, th, F: F.erfinv(x[1] * x[2] - x[3] / 2)),
    ("sf-erfinv-edge", lambda x, th, F: F.erfinv(1 - x[1] * 1e-3)),
    ("sf-erfcinv", lambda x, th, F: F.erfcinv(x[1] + x[2])),
    ("sf-erfcinv-small", lambda x, th, F: F.erfcinv(x[1] * x[2] * 1e-6)),
    ("sf-beta", lambda x, th, F: F.beta(3 * x[1] + x[2], x[2] * x[3] + 0.5)),
    ("sf-beta-neg", lambda x, th, F: F.beta(x[1] - 1.5, x[2] + 2.0)),
    ("sf-beta-fixed", lambda x, th, F: F.beta(x[1] + x[2], 2.5) + F.beta(1.5, x[3]) + F.beta(th[1], x[1] * x[3])),
    ("sf-logbeta", lambda x, th, F: F.logbeta(5 * x[1], x[2] + 7 * x[3])),
    ("sf-logbeta-fixed", lambda x, th, F: F.logbeta(x[1] * x[2], 0.75) + F.logbeta(th[2], x[3])),
]

NVAR, NPAR = 10, 2
