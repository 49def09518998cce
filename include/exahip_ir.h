/* exahip_ir.h — wire format of a model handed to the evaluator ("pattern table").
 *
 * A model is a list of PATTERNS in insertion order.  A pattern is one algebraic
 * expression (a flat post-order node array) paired with a data iterator stored as
 * struct-of-arrays COLUMNS; it is what ExaModels calls a SIMDFunction + its `itr`
 * (reference: src/simdfunction.jl:21-30, src/nlp.jl:107-110,137-143,171-177).
 *
 * The node kinds restate the reference's symbolic tree (src/graph.jl:37-300):
 *   CONST_F / CONST_I  plain Real child of a Node2, Constant{v}, Val{v}
 *   DATA               DataSource / DataIndexed{...,field}: a field of the data point p = itr[I]
 *   PAR                ParameterNode(index-expr)       -> theta[idx]   (constant for AD)
 *   VAR                Var(index-expr)                 -> x[idx]       (differentiable leaf)
 *   UN / BIN           Node1{F} / Node2{F}
 *   NULLV              Null(v): constant row (empty constraint generator, nlp.jl:1570)
 * Index expressions (children of VAR/PAR, and the CONAUG target) are integer-typed
 * subtrees built only from CONST_I, DATA (of an I64/RANGE column) and BIN +,-,*.
 * All indices are 1-based, exactly as in the reference.
 *
 * This header is DATA FORMAT ONLY.  It is shared by the product (libexahip.so) and by
 * the test oracle (oracle/), which each parse it with their own code.
 */
#ifndef EXAHIP_IR_H
#define EXAHIP_IR_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum exa_opcode {
    EXA_OP_CONST_F = 0,
    EXA_OP_CONST_I = 1,
    EXA_OP_DATA    = 2,
    EXA_OP_PAR     = 3,
    EXA_OP_VAR     = 4,
    EXA_OP_UN      = 5,
    EXA_OP_BIN     = 6,
    EXA_OP_NULLV   = 7
};

/* Univariate functions; order = reference src/functionlist.jl:6-60 (_UNIVARIATES). */
enum exa_un_fn {
    EXA_U_PLUS = 0, EXA_U_MINUS, EXA_U_INV, EXA_U_SQRT, EXA_U_CBRT, EXA_U_ABS, EXA_U_ABS2,
    EXA_U_SIGN, EXA_U_EXP, EXA_U_EXP2, EXA_U_EXP10, EXA_U_EXPM1, EXA_U_LOG, EXA_U_LOG2,
    EXA_U_LOG1P, EXA_U_LOG10, EXA_U_SIN, EXA_U_COS, EXA_U_TAN, EXA_U_ASIN, EXA_U_ACOS,
    EXA_U_ATAN, EXA_U_ACOT, EXA_U_CSC, EXA_U_SEC, EXA_U_COT, EXA_U_SINH, EXA_U_COSH,
    EXA_U_TANH, EXA_U_ASINH, EXA_U_ACOSH, EXA_U_CSCH, EXA_U_SECH, EXA_U_COTH,
    EXA_U_SIND, EXA_U_COSD, EXA_U_TAND, EXA_U_CSCD, EXA_U_SECD, EXA_U_COTD,
    EXA_U_ATAND, EXA_U_ACOTD, EXA_U_SINPI, EXA_U_COSPI, EXA_U_SINC,
    EXA_U_DEG2RAD, EXA_U_RAD2DEG, EXA_U_SIGNBIT, EXA_U_FLOOR, EXA_U_CEIL,
    EXA_U_ATANH, EXA_U_ACOTH,
    /* the SpecialFunctions extension; order = reference ext/functionlist.jl:6-102 (erf ... erfcinv) */
    EXA_U_ERF, EXA_U_ERFC, EXA_U_ERFI, EXA_U_ERFCX, EXA_U_DIGAMMA, EXA_U_TRIGAMMA, EXA_U_INVDIGAMMA,
    EXA_U_GAMMA, EXA_U_AIRYAI, EXA_U_AIRYBI, EXA_U_AIRYAIPRIME, EXA_U_AIRYBIPRIME,
    EXA_U_BESSELJ0, EXA_U_BESSELY0, EXA_U_BESSELJ1, EXA_U_BESSELY1, EXA_U_DAWSON,
    EXA_U_ERFINV, EXA_U_ERFCINV,
    EXA_U_COUNT
};
#define EXA_U_FIRST_SPECIAL EXA_U_ERF

/* Bivariate functions; order = reference src/functionlist.jl:71-81 (_BIVARIATES). */
enum exa_bin_fn {
    EXA_B_ADD = 0, EXA_B_SUB, EXA_B_MUL, EXA_B_DIV, EXA_B_POW, EXA_B_ATAN2, EXA_B_HYPOT,
    EXA_B_MAX, EXA_B_MIN,
    /* the SpecialFunctions extension; reference ext/functionlist.jl:109-124 */
    EXA_B_BETA, EXA_B_LOGBETA,
    EXA_B_COUNT
};
#define EXA_B_FIRST_SPECIAL EXA_B_BETA

typedef struct exa_node {
    int32_t op;    /* enum exa_opcode */
    int32_t fn;    /* enum exa_un_fn / exa_bin_fn for UN / BIN, else 0 */
    int32_t a;     /* UN,BIN: first child; VAR,PAR: root of the index expression; DATA: column id */
    int32_t b;     /* BIN: second child; else -1 */
    double  fval;  /* CONST_F, NULLV */
    int64_t ival;  /* CONST_I */
} exa_node_t;      /* 32 bytes; children always precede parents (post-order) */

enum exa_coltype {
    EXA_COL_I64   = 0,  /* const int64_t data[n] */
    EXA_COL_F64   = 1,  /* const double  data[n] */
    EXA_COL_RANGE = 2   /* value(I) = start + step*I for I = 0..n-1; no storage (UnitRange/StepRange iterator) */
};

typedef struct exa_column {
    int32_t     type;   /* enum exa_coltype */
    int32_t     _pad;
    const void *data;   /* HOST pointer with n entries; NULL for RANGE */
    int64_t     start;  /* RANGE */
    int64_t     step;   /* RANGE */
} exa_column_t;

enum exa_patkind {
    EXA_PAT_OBJ    = 0,  /* Objective             (nlp.jl:1448-1482) */
    EXA_PAT_CON    = 1,  /* Constraint            (nlp.jl:1551-1611) */
    EXA_PAT_CONAUG = 2   /* ConstraintAugmentation(nlp.jl:1679-1738) */
};

typedef struct exa_pattern {
    int32_t             kind;     /* enum exa_patkind */
    int32_t             n_nodes;
    const exa_node_t   *nodes;
    int32_t             root;     /* value expression root */
    int32_t             target;   /* CONAUG: root of the integer expression giving the 1-based row
                                     inside the base constraint block (`idx` or column-major `idxx`,
                                     nlp.jl:1986-2015); otherwise -1 */
    int32_t             base;     /* CONAUG: position (in patterns[]) of the base CON pattern; else -1 */
    int32_t             n_cols;
    const exa_column_t *cols;
    int64_t             n;        /* number of data points, length(itr) */
} exa_pattern_t;

typedef struct exa_model_desc {
    int64_t              nvar;
    int64_t              npar;        /* length of theta */
    const double        *x0;          /* [nvar] or NULL (zeros) */
    const double        *lvar;        /* [nvar] or NULL (-inf) */
    const double        *uvar;        /* [nvar] or NULL (+inf) */
    const double        *theta0;      /* [npar] or NULL */
    int32_t              n_patterns;
    int32_t              minimize;    /* 1 = minimize (default), 0 = maximize: objective sign is NOT flipped by the
                                         evaluator (the reference leaves that to the solver; nlp.jl:486-501) */
    const exa_pattern_t *patterns;    /* insertion order == reference counter order (nlp.jl:1474-1482,1597-1611,1730-1738) */
    const double        *y0;          /* [ncon] or NULL */
    const double        *lcon;        /* [ncon] or NULL (0) */
    const double        *ucon;        /* [ncon] or NULL (0) */
} exa_model_desc_t;

#ifdef __cplusplus
}
#endif
#endif /* EXAHIP_IR_H */
