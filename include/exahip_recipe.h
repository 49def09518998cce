/* exahip_recipe.h — recipes: the on-the-wire model format and the schema/builder ABI of libexahip.so
 * (SURVEY §8f.4).
 *
 * What it replaces.  In the reference a core built against `ArgSource` placeholders is a *recipe*
 * (src/argument.jl:67-185, src/nlp.jl:507-523); `ExaModel(core, args...)` instantiates it (nlp.jl:809-863),
 * and ExaModelsCompiler publishes it from a compiled library through
 *
 *   P_nargs / P_argtype / P_new(n)                      ExaModelsCompiler.jl:1032-1047, :1075-1087
 *   P_schema / P_data_begin / P_set_scalar_{i64,f64} / P_set_array_{i64,f64} / P_set_col_{i64,f64} /
 *   P_data_ready / P_new_from_data                      ExaModelsCompiler.jl:1197-1330
 *   P_nblocks / P_block_name / P_block / P_get_value / P_set_value      :1476-1535
 *   cnlp_nmodels / cnlp_model_name                      :1376-1395
 *
 * The functions below are those entry points with the prefix made an argument (`recipe` id): a per-model
 * library with the reference's literal symbol names (`<prefix>_schema`, ...) is a few lines of C on top
 * (exahip.pack writes and compiles it).  Status codes as everywhere in the cnlp ABI:
 * 0 ok, 1 bad id / unknown field / bad argument, 2 internal error, 3 length disagrees with the block
 * (get/set_value); constructors return an id > 0 or 0 on failure; counts return -1 on a bad id.
 *
 * ---------------------------------------------------------------------------------------------------------
 * Wire format "EXARCP01" (little-endian; i32/i64/f64; str = i32 byte count + bytes; arr<T> = i64 count + data)
 *
 *   magic[8] = "EXARCP01"
 *   i32 minimize
 *   i32 nfields   { str name; i32 kind (0 scalar, 1 array, 2 table); i32 type (0 i64, 1 f64; scalar/array);
 *                   i32 ncols { str name; i32 type } }                       -- the schema, in binding order
 *   i32 nsyms     { i32 op; i64 a; i64 b }                                   -- size expressions, SSA, post-order:
 *                   0 CONST a | 1 SCALAR field a | 2 LEN field a | 3 ADD | 4 SUB | 5 MUL | 6 FLOORDIV (a,b = earlier
 *                   syms) | 7 MAX0 a | 8 NEG a
 *                   real-valued (a deferred coefficient such as 1/(N+1)): 9 FCONST (a = bits of the double) | 10 ITOF a |
 *                   11 FADD | 12 FSUB | 13 FMUL | 14 FDIV | 15 FNEG a   (operands real, ITOF's operand integer)
 *   ival          = i32 is_sym; i64 (sym id | literal)                       -- every size-like integer below
 *   ival nvar; ival npar
 *   7 vectors (x0, lvar, uvar, theta, y0, lcon, ucon), each:
 *       i32 nsegments { ival n; i32 src; payload }   src: 0 CONST f64 | 1 INLINE arr<f64> | 2 FIELD i32 field |
 *                                                         3 COL i32 field, i32 column
 *   i32 nblocks   { str name; i32 kind (0 var, 1 con, 2 par); ival offset; ival length; i32 ndims; ival dims[] }
 *   i32 npatterns {
 *       i32 kind; i32 root; i32 target; i32 base; ival n                     -- exa_pattern_t (exahip_ir.h)
 *       i32 nnodes { i32 op; i32 fn; i32 a; i32 b; f64 fval; i64 ival; i32 sym }   -- sym >= 0: CONST_I (CONST_F) leaf
 *                                                                                whose value is that integer (real) expression
 *       i32 ncols  { i32 kind; payload }
 *           0 RANGE        ival start; ival step
 *           1 INLINE_I64   arr<i64>             2 INLINE_F64  arr<f64>
 *           3 FIELD        i32 field            4 COL         i32 field; i32 column
 *           5 AXIS_RANGE   ival start; ival step; ival axis_len; ival inner      -- one axis of a product iterator:
 *           6 AXIS_FIELD   i32 field; i32 column (-1: array field); ival axis_len; ival inner
 *           7 AXIS_INLINE_I64 arr<i64>; ival axis_len; ival inner    8 AXIS_INLINE_F64 arr<f64>; ...
 *                           value(k) = axis[(k / inner) mod axis_len],  k = 0 .. n-1 (first axis fastest)
 *   }
 *   optional, only when a pattern uses a user-registered function (exahip.h: exa_register_univariate / _bivariate; node fn >= 1000):
 *   i32 nuserfns  { i32 bivariate; i32 fn (the id the nodes above use); str name; str f; str d1; str d2; str d11; str d12; str d22;
 *                   str helpers; str fused }      -- univariate: d1 = df, d11 = ddf, the others empty; fused: the one-statement form
 *                   (exa_register_univariate_fused; then f, d1, d11 are empty too).  exa_recipe_load registers each entry (the same
 *                   rules again give the id they already have; a name taken by OTHER rules refuses the file) and renumbers the nodes:
 *                   the file is self-contained.  Every fn >= 1000 a node uses must be defined by this section (a file without the
 *                   section must not use any).  All entries — and everything else about the file — are checked before any is
 *                   registered: a file that is refused leaves the process as it was.
 *                   TRUST BOUNDARY: the rule texts and `helpers` are HIP device SOURCE — hipcc / hiprtc compile it and the GPU runs
 *                   it in the host's context.  A file with this section is CODE, not data.  exa_recipe_load therefore refuses a file
 *                   whose entries are not, rule for rule, registrations the process has already made itself, unless the host opted
 *                   in: exa_recipe_trust_code(1), or EXAHIP_TRUST_MODEL_CODE=1 in the environment.  (A packed library, exahip.pack, is
 *                   native code already: its loader reads its own embedded recipe through exa_recipe_load_trusted.)
 * A fully concrete model is the special case nfields = 0, nsyms = 0 (everything literal / inline): the same
 * bytes are the library's model file format (exa_recipe_load + exa_recipe_new).
 */
#ifndef EXAHIP_RECIPE_H
#define EXAHIP_RECIPE_H
#include <stddef.h>
#include <stdint.h>
#include "exahip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- recipes ---------------------------------------------------------------------------------------- */
int exa_recipe_load(const void *bytes, size_t len);            /* -> recipe id > 0, 0 on a malformed (or refused) recipe  */
/* May model files bring device code of their own (registered functions this process has not registered itself)?  Default: no
 * (EXAHIP_TRUST_MODEL_CODE=1 in the environment: yes).  on = 1 / 0 sets it for the process, on < 0 only asks.  Returns the setting
 * in force before the call. */
int exa_recipe_trust_code(int on);
/* exa_recipe_load for bytes the CALLER vouches for (they are part of its own native code: the embedded recipe of a packed library,
 * exahip.pack): device code in the file is accepted for this one call on this one thread; the process-wide setting above — unset, 0 or
 * 1 — is neither consulted nor changed, and a concurrent exa_recipe_load of a foreign file on another thread stays refused. */
int exa_recipe_load_trusted(const void *bytes, size_t len);
int exa_recipe_free(int recipe);
/* P_nargs: how many values instantiation consumes — 0 for a fixed model, else the number of schema fields. */
int exa_recipe_nargs(int recipe);
/* P_argtype / P_schema: copy-out convention of the reference — returns the needed byte length, copies what fits. */
int exa_recipe_argtype(int recipe, char *buf, int cap);
int exa_recipe_schema(int recipe, char *buf, int cap);
/* P_new(n): fixed model (n ignored) or a recipe whose only field is one i64 scalar.  0 on failure (also for a
 * structured recipe: its entry point is the builder — the two surfaces are disjoint, as the consumers assume). */
int exa_recipe_new(int recipe, int n);
/* Same, planning only (no device): CPU-side checks. */
int exa_recipe_plan(int recipe, int n);

/* ---- builder ---------------------------------------------------------------------------------------- */
int exa_data_begin(int recipe);                                 /* -> builder id > 0, 0 on failure */
int exa_data_free(int builder);
int exa_set_scalar_i64(int builder, const char *field, int64_t v);
int exa_set_scalar_f64(int builder, const char *field, double v);
int exa_set_array_i64(int builder, const char *field, const int64_t *v, int len);   /* copies */
int exa_set_array_f64(int builder, const char *field, const double *v, int len);
int exa_set_col_i64(int builder, const char *table, const char *column, const int64_t *v, int len);
int exa_set_col_f64(int builder, const char *table, const char *column, const double *v, int len);
int exa_data_ready(int builder);            /* 1 iff every field is set and every table's columns agree in length */
int exa_new_from_data(int builder);         /* -> model id > 0 (an `exa_*` model, exahip.h), 0 on failure */
int exa_plan_from_data(int builder);        /* same, planning only (no device) */

/* ---- named blocks of an instance --------------------------------------------------------------------- */
int exa_nblocks(int id);
int exa_block_name(int id, int k, char *buf, int cap);          /* needed byte length, -1 on a bad id / k */
/* out = [kind, offset, length, ndims, dims...]; kind 0 variable, 1 constraint, 2 parameter; offset 0-based */
int exa_block(int id, int k, int *out);
int exa_get_value_block(int id, int k, double *vals, int len);  /* parameter block k; 3 = len != block length */
int exa_set_value_block(int id, int k, const double *vals, int len);

/* ---- introspection ----------------------------------------------------------------------------------- */
/* theta[offset .. offset+len) -> vals (HOST) */
int exa_get_value(int id, int64_t offset, double *vals, int64_t len);
/* View of a planned model as a pattern table (pointers into library-owned storage, valid until exa_free).
 * Only for models that still hold their host columns (exa_plan_only / exa_recipe_plan / exa_plan_from_data). */
int exa_describe(int id, exa_model_desc_t *out);

#ifdef __cplusplus
}
#endif
#endif /* EXAHIP_RECIPE_H */
