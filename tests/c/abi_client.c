/* abi_client.c — a plain-C consumer of libexahip.so: no Python, no torch types, only include/exahip.h.
 *
 * Builds the Luksan-Vlcek model (benchmark/runbenchmark.jl:163-169) as a pattern table by hand, exactly what the Julia
 * shim's lowering produces, then
 *   mode "plan": exa_plan_only -> sizes + slot maps (runs without a GPU);
 *   mode "eval": exa_new_from_table -> every callback through the *_host entry points, printed for the pytest wrapper.
 * Build: gcc -O1 -I include tests/c/abi_client.c -o abi_client -L examodels.jl_amd/exahip -lexahip -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "exahip.h"

static exa_node_t N[256];
static int nn;
static int ci(int64_t v) { N[nn] = (exa_node_t){EXA_OP_CONST_I, 0, -1, -1, 0.0, v}; return nn++; }
static int data(int col) { N[nn] = (exa_node_t){EXA_OP_DATA, 0, col, -1, 0.0, 0}; return nn++; }
static int bin(int fn, int a, int b) { N[nn] = (exa_node_t){EXA_OP_BIN, fn, a, b, 0.0, 0}; return nn++; }
static int un(int fn, int a) { N[nn] = (exa_node_t){EXA_OP_UN, fn, a, -1, 0.0, 0}; return nn++; }
static int var(int idx) { N[nn] = (exa_node_t){EXA_OP_VAR, 0, idx, -1, 0.0, 0}; return nn++; }
/* x[i + off]: Variable getindex builds Var(Node2(+, Node2(+|-, i, |off|), 0)) for a 1-D block at offset 0 (nlp.jl:900-910) */
static int xi(int off) {
    int i = data(0);
    int e = off == 0 ? i : bin(off > 0 ? EXA_B_ADD : EXA_B_SUB, i, ci(off > 0 ? off : -off));
    return var(bin(EXA_B_ADD, e, ci(0)));
}

int main(int argc, char **argv) {
    const char *mode = argc > 1 ? argv[1] : "plan";
    const int64_t Nv = 10;
    /* "plan-user" / "eval-user": the same model with sin and exp REGISTERED from C (the reference's @register_univariate, src/register.jl:56-74):
     * sin as one fused statement, exp as three rules; the ids take the place of EXA_U_SIN / EXA_U_EXP in the table */
    int fn_sin = EXA_U_SIN, fn_exp = EXA_U_EXP;
    if (strstr(mode, "-user")) {
        fn_sin = exa_register_univariate_fused("c_sin", "exa_sincos($1, &$2, &$3); $4 = -$2;", NULL);
        fn_exp = exa_register_univariate("c_exp", "exp($1)", "$2", "$3", NULL);
        if (fn_sin < 1000 || fn_exp < 1000 || fn_sin == fn_exp) { printf("FAIL register: %s\n", exa_last_error()); return 1; }
        if (exa_register_univariate("c_exp", "exp($1)", "$2", "$3", NULL) != fn_exp) { printf("FAIL same rules, other id\n"); return 1; }
        if (exa_register_univariate("c_exp", "exp2($1)", "$2", "$3", NULL) != -1) { printf("FAIL other rules accepted\n"); return 1; }
        char buf[64];
        if (exa_user_function(0, fn_sin, 8, buf, sizeof buf) != 35 || strncmp(buf, "exa_sincos(", 11)) { printf("FAIL read back\n"); return 1; }
        mode = !strncmp(mode, "plan", 4) ? "plan" : "eval";
    }
    /* constraint: 3x[i+1]^3 + 2x[i+2] - 5 + sin(x[i+1]-x[i+2])sin(x[i+1]+x[i+2]) + 4x[i+1] - x[i]exp(x[i]-x[i+1]) - 3, i = 1:N-2 */
    nn = 0;
    int t1 = bin(EXA_B_MUL, ci(3), bin(EXA_B_POW, xi(1), ci(3)));
    int t2 = bin(EXA_B_MUL, ci(2), xi(2));
    int s = bin(EXA_B_SUB, bin(EXA_B_ADD, t1, t2), ci(5));
    int sn = bin(EXA_B_MUL, un(fn_sin, bin(EXA_B_SUB, xi(1), xi(2))), un(fn_sin, bin(EXA_B_ADD, xi(1), xi(2))));
    s = bin(EXA_B_ADD, s, sn);
    s = bin(EXA_B_ADD, s, bin(EXA_B_MUL, ci(4), xi(1)));
    s = bin(EXA_B_SUB, s, bin(EXA_B_MUL, xi(0), un(fn_exp, bin(EXA_B_SUB, xi(0), xi(1)))));
    int con_root = bin(EXA_B_SUB, s, ci(3));
    int con_n = nn;
    exa_node_t *con_nodes = malloc(sizeof(exa_node_t) * con_n);
    memcpy(con_nodes, N, sizeof(exa_node_t) * con_n);
    /* objective: 100 (x[i-1]^2 - x[i])^2 + (x[i-1] - 1)^2, i = 2:N */
    nn = 0;
    int a = un(EXA_U_ABS2, bin(EXA_B_SUB, un(EXA_U_ABS2, xi(-1)), xi(0)));
    int o = bin(EXA_B_ADD, bin(EXA_B_MUL, ci(100), a), un(EXA_U_ABS2, bin(EXA_B_SUB, xi(-1), ci(1))));
    int obj_n = nn;
    exa_node_t *obj_nodes = malloc(sizeof(exa_node_t) * obj_n);
    memcpy(obj_nodes, N, sizeof(exa_node_t) * obj_n);

    exa_column_t ccol = {EXA_COL_RANGE, 0, NULL, 1, 1}, ocol = {EXA_COL_RANGE, 0, NULL, 2, 1};
    exa_pattern_t pats[2] = {
        {EXA_PAT_CON, con_n, con_nodes, con_root, -1, -1, 1, &ccol, Nv - 2},
        {EXA_PAT_OBJ, obj_n, obj_nodes, o, -1, -1, 1, &ocol, Nv - 1},
    };
    double x0[10];
    for (int i = 0; i < 10; i++) x0[i] = (i % 2 == 0) ? -1.2 : 1.0;
    exa_model_desc_t d;
    memset(&d, 0, sizeof d);
    d.nvar = Nv; d.x0 = x0; d.n_patterns = 2; d.minimize = 1; d.patterns = pats;
    int id = 0, st;
    if (!strcmp(mode, "plan")) {
        st = exa_plan_only(&d, &id);
        if (st) { printf("FAIL plan status %d: %s\n", st, exa_last_error()); return 1; }
    } else {
        st = exa_new_from_table(&d, &id);
        if (st) { printf("FAIL new status %d: %s\n", st, exa_last_error()); return 1; }
    }
    printf("abi %d nvar %d ncon %d nnzj %d nnzh %d\n", exa_abi_version(), exa_nvar(id), exa_ncon(id), exa_nnzj(id), exa_nnzh(id));
    int64_t info[9];
    int32_t comp[64];
    exa_pattern_info(id, 0, info);
    exa_pattern_comp(id, 0, 2, comp);
    printf("con o2step %lld comp2", (long long)info[6]);
    for (int k = 0; k < info[8]; k++) printf(" %d", comp[k]);
    printf("\n");
    if (exa_nvar(99) != -1 || exa_hess_host(99, x0, x0, 1.0, x0) != 1) { printf("FAIL bad-id convention\n"); return 1; }
    if (!strcmp(mode, "eval")) {
        double f, c[8], g[10], jv[24], hv[75], y[8];
        int32_t r[75], cc[75];
        for (int i = 0; i < 8; i++) y[i] = 1.0 + 0.1 * i;
        st = exa_obj_host(id, x0, &f) | exa_cons_host(id, x0, c) | exa_grad_host(id, x0, g) | exa_jac_host(id, x0, jv) |
             exa_hess_host(id, x0, y, 0.5, hv) | exa_hess_structure_host(id, r, cc);
        if (st) { printf("FAIL eval status %d: %s\n", st, exa_last_error()); return 1; }
        printf("obj %.17g\n", f);
        printf("cons"); for (int i = 0; i < 8; i++) printf(" %.17g", c[i]); printf("\n");
        printf("grad"); for (int i = 0; i < 10; i++) printf(" %.17g", g[i]); printf("\n");
        printf("jac"); for (int i = 0; i < 24; i++) printf(" %.17g", jv[i]); printf("\n");
        printf("hess"); for (int i = 0; i < 75; i++) printf(" %.17g", hv[i]); printf("\n");
        printf("hrows"); for (int i = 0; i < 75; i++) printf(" %d", r[i]); printf("\n");
        printf("hcols"); for (int i = 0; i < 75; i++) printf(" %d", cc[i]); printf("\n");
        /* products (jprod_nln! / jtprod_nln! / hprod!), the objective-only forms (y == NULL), the Jacobian structure */
        double v[10], w[8], Jv[8], Jtv[10], Hv[10], Hv0[10], hv0[75];
        int32_t jr[24], jc[24];
        for (int i = 0; i < 10; i++) v[i] = 0.3 * i - 1.0;
        for (int i = 0; i < 8; i++) w[i] = 0.5 - 0.2 * i;
        st = exa_jprod_host(id, x0, v, Jv) | exa_jtprod_host(id, x0, w, Jtv) | exa_hprod_host(id, x0, y, v, 0.5, Hv) |
             exa_hprod_host(id, x0, NULL, v, 0.5, Hv0) | exa_hess_host(id, x0, NULL, 0.5, hv0) | exa_jac_structure_host(id, jr, jc);
        if (st) { printf("FAIL products status %d: %s\n", st, exa_last_error()); return 1; }
        printf("jprod"); for (int i = 0; i < 8; i++) printf(" %.17g", Jv[i]); printf("\n");
        printf("jtprod"); for (int i = 0; i < 10; i++) printf(" %.17g", Jtv[i]); printf("\n");
        printf("hprod"); for (int i = 0; i < 10; i++) printf(" %.17g", Hv[i]); printf("\n");
        printf("hprodobj"); for (int i = 0; i < 10; i++) printf(" %.17g", Hv0[i]); printf("\n");
        printf("hessobj"); for (int i = 0; i < 75; i++) printf(" %.17g", hv0[i]); printf("\n");
        printf("jrows"); for (int i = 0; i < 24; i++) printf(" %d", jr[i]); printf("\n");
        printf("jcols"); for (int i = 0; i < 24; i++) printf(" %d", jc[i]); printf("\n");
    }
    if (exa_free(id) != 0 || exa_free(id) != 1) { printf("FAIL free convention\n"); return 1; }
    printf("OK\n");
    return 0;
}
