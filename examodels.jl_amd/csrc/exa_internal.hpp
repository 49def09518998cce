// exa_internal.hpp — shared declarations of libexahip.so (planner, HIP code generator, runtime).
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/exahip.h"

namespace exa {

// How n items (data points of a pattern, variables, windows) are split over G ranks: rank r holds [part_lo(n, r, G),
// part_lo(n, r + 1, G)) — floor(n / G) items each, the last rank the remainder (< G items) on top.  Equal pieces are what lets
// the owner-sharded vectors be completed by ONE in-place ncclAllGather (+ a broadcast of the remainder) instead of G broadcasts.
// Few items (n < 16 G: the remainder would be more than 6 % of a share) are split evenly instead, floor(n r / G): balance first.
inline int64_t part_lo(int64_t n, int r, int G) {
    if (r >= G) return n;
    return n >= 16LL * G ? (n / G) * (int64_t)r : (int64_t)((__int128)n * r / G);
}

// malformed pattern table / unsupported construct: the caller's fault -> C-ABI status 1 (everything else -> 2)
struct BadInput : std::runtime_error { using std::runtime_error::runtime_error; };

// ---------------------------------------------------------------------------------------------------
// Deep copy of the wire model (include/exahip_ir.h)
// ---------------------------------------------------------------------------------------------------
struct Column {
    int type = EXA_COL_RANGE;
    std::vector<int64_t> idata;
    std::vector<double> fdata;
    int64_t start = 0, step = 0;
    // the same column (type and contents) appeared earlier in the model — the branch table that ACOPF's four flow
    // constraints all iterate over arrives once per pattern in the wire format: ONE copy goes to HBM and both patterns
    // name the same parameter word, so their index expressions are textually equal (shared loads, merged scatter targets)
    int alias_pat = -1, alias_col = -1;
};

// One node of the per-pattern AD tree: what the reference builds with its adjoint node types
// (src/graph.jl:337-494) when the pattern's tree is called on an AdjointNodeSource.
enum ADKind { AD_CONST = 0, AD_VAR = 1, AD_UN = 2, AD_BIN = 3, AD_NULL = 4 };
enum Fixed { FX_NONE = 0, FX_FIRST = 1, FX_SECOND = 2 };   // FirstFixed / SecondFixed (register.jl:231-266)

struct ADNode {
    int kind = AD_CONST;
    int fn = 0;
    int fixed = FX_NONE;
    int ir = -1;     // AD_CONST: IR root of the Real subtree; AD_VAR: IR root of the index expression; else own IR id
    int cir = -1;    // fixed binary: IR root of the constant operand
    int key = -1;    // AD_VAR: id of the structural key of its index expression
    int l = -1, r = -1;
};

struct Pattern {
    int kind = EXA_PAT_OBJ;
    std::vector<exa_node_t> nodes;
    int root = -1, target = -1, base = -1;
    std::vector<Column> cols;
    int64_t n = 0;
    // ---- plan ----
    std::vector<char> isconst;      // per IR node
    std::vector<char> isint;        // per IR node: statically Int-typed
    std::vector<ADNode> ad;         // AD tree, ad_root is its root
    int ad_root = -1;
    std::vector<std::string> keys;  // structural keys of VAR index expressions
    std::vector<int> comp1, comp2;  // 1-based slot maps (Compressor, simdfunction.jl:11-14)
    std::vector<int> slotvar1;                      // per 1st-order slot: AD leaf (any visit) that defines its variable
    std::vector<std::pair<int, int>> slotvar2;      // per 2nd-order slot: ordered pair of AD leaves
    int o1step = 0, o2step = 0;
    int64_t o0 = 0, o1 = 0, o2 = 0;
    int64_t oa = 0;                 // CONAUG: first entry of this pattern in the augmentation value buffer (nlp.jl:1731)
};

struct Model {
    int64_t nvar = 0, npar = 0, ncon = 0, nnzj = 0, nnzh = 0, nnzg = 0, nobj = 0, nconaug = 0;
    int minimize = 1;
    std::vector<Pattern> pats;
    std::vector<double> x0, lvar, uvar, theta, y0, lcon, ucon;
    // Augmentation gather lists (the reference's conaugsparsity sorted by target row + conaugptr, KA ext :79-101):
    // target row aug_rows[t] receives buffer entries aug_perm[aug_ptr[t] .. aug_ptr[t+1])
    std::vector<int64_t> aug_rows, aug_ptr, aug_perm;
    // every augmentation term is coefficient * x[index] (ACOPF: p[a.i], -pg[g.i]): per buffer entry the 0-based variable
    // and the coefficient, evaluated once at build — the one-launch cons_nln! adds such a row with two loads per term
    bool aug_linear = false;
    std::vector<int64_t> aug_var;
    std::vector<double> aug_coef;
};

// Planner (exa_plan.cpp): copies the description, builds AD trees, slot maps and running offsets.
std::unique_ptr<Model> plan_model(const exa_model_desc_t *desc);   // throws std::runtime_error
std::vector<int64_t> locality_order(const Model &m, int pattern);   // exa_locality_order (exa_plan.cpp)
// per data point: the smallest variable any x[...] of the pattern names there (INT64_MAX: none).  Needs the host columns.
std::vector<int64_t> locality_keys(const Model &m, int pattern);
// bits[v / 64] >> (v % 64) & 1 <- some first-order slot of some point of the objective patterns `pats` is the 0-based variable v; false = a variable is
// named twice (by two points, or by two slots of one point whose index expressions differ): the scatter is not injective.  Needs the host columns.
bool scatter_bitmap(const Model &m, const std::vector<int> &pats, std::vector<uint64_t> &bits);

// ---------------------------------------------------------------------------------------------------
// Code generator (exa_gen_*.cpp; internals in exa_gen.hpp)
// ---------------------------------------------------------------------------------------------------
enum Callback { CB_OBJ = 0, CB_GRAD, CB_CONS, CB_JAC, CB_HESS, CB_JSTRUCT, CB_HSTRUCT,
                CB_JPROD, CB_JTPROD, CB_HPROD, CB_FUSED,
                CB_CONS1,      // cons_nln! in ONE launch (exa_cons1): the thread of a base row pulls the row's augmentation terms itself
                CB_HESSC,      // hess_coord!, second kernel (exa_hessc): chained + grouped + software-pipelined, see ParamLayout::chain
                CB_COUNT };

struct ParamLayout {
    // word indices into the int64 parameter table P that every kernel receives
    struct Pat {
        int lo = -1, hi = -1, o0 = -1, o1 = -1, o2 = -1, oa = -1;
        // gathered objective patterns (exa_grad_pull, owner computes): the data points [qlo, qhi) of the WHOLE pattern that
        // touch a variable this rank owns — all of them unless sharded (lo / hi are the shard's own data points)
        int qlo = -1, qhi = -1;
        int ob = -1;            // OBJ patterns: first slot of this pattern's workgroups among the fused sweep's objective partials
        std::vector<int> col;   // per column: device pointer (I64/F64) or range start (RANGE)
    };
    std::vector<Pat> pat;
    std::vector<int> active[CB_COUNT];   // patterns handled by each callback, in dispatch order
    int blk[CB_COUNT];                   // word holding the device address of the callback's block map: entry b =
                                         // (slot in active[cb] << 40) | tile index, built by the runtime (interleaved)
    int ppt[CB_COUNT];                   // data points per thread (a workgroup covers kBlock * ppt points)
    // Chained callbacks (CB_HESSC: the streaming variant of hess_coord!; which of exa_hess / exa_hessc runs is a measured,
    // persisted decision — the pipelined loop needs twice the registers, so cache-resident models prefer the plain
    // kernel and models streaming gigabytes the chained one).  The active patterns are partitioned into GROUPS of co-indexed patterns (iterators of
    // the same length: LV's constraint and objective, the four branch-flow constraints of ACOPF, the dynamics rows of a
    // discretised ODE).  A block-map entry is (group, first tile); its workgroup walks `chain` consecutive tiles and, for
    // every tile, all patterns of the group back to back — so co-indexed patterns read their stretch of x / their table
    // rows from HBM once (the second reader hits L1 / L2) at ANY model size.  The walk is software-pipelined: the inputs
    // of the next (pattern, tile) are loaded BEFORE the current one is evaluated and stored (on gfx9 loads and stores
    // share vmcnt in order, so loads issued after a tile's stores would wait for those stores to drain).
    // chain = 0: one (pattern, tile) per workgroup, the plain dispatch.  gtiles = P word with each group's tile count.
    int chain[CB_COUNT];
    // exa_hesscl — exa_hessc with the inputs STAGED THROUGH LDS.  Every x index of the form (unit-step range value) + literal is
    // staged; the literals of a pattern fall into CLUSTERS (gaps of more than kStageHalo: the variable blocks of a model laid out as
    // separate arrays — the rocket's h / v / m / tau — bake their offsets into the literals), and the clusters of all patterns of a
    // chained group are merged into the group's STRETCHES (by literal proximity; the runtime verifies the geometry, stage_ok).  Per
    // wavefront, tile and stretch ONE run of 64 + halo variables is loaded (one coalesced 8-byte load per lane + a halo load by a few
    // lanes) and written to LDS, from which every pattern takes its stencil operands (ds_read) — instead of two or three overlapping
    // wide loads per pattern and array.  x loads of any other form (a literal index: the rocket's step length; a data column) stay
    // ordinary loads of the load stage.  stage[k] = {word of the range column, clusters {smallest, largest literal, stretch of the
    // group}}; cluster (k, c) starts at B = P[word] + P[lo] + cmin - 1, stretch s at the smallest B of its members.
    struct Stage {
        int word = -1;
        struct Cluster { int64_t cmin = 0, cmax = 0; int stretch = 0; };
        std::vector<Cluster> cl;
    };
    std::vector<Stage> stage;
    std::vector<int> gstretch;          // stretches per chained group (CB_HESSC)
    int max_stretch = 0;
    bool staged = false;
    std::vector<std::vector<int>> groups[CB_COUNT];
    std::vector<int> gtiles[CB_COUNT];
    int nwords = 0;
    // exa_eval_all in ONE launch for models whose in-sweep objective gradient (data-indexed scatter) hits every variable AT MOST ONCE
    // (checked on the data at model build, unsharded models): word of the device address of the bitmap "variable is written by an
    // objective point" — non-zero = the sweep's objective tiles STORE their first partials (no atomics, no zero-filled g) and one more
    // unit of the block map writes 0.0 to every variable the bitmap does not name.  -1: the module has no such path.
    int gbits = -1;
    std::vector<int> pull;               // objective patterns whose gradient is GATHERED per variable (exa_grad_pull)
    int pull_ppt = 1;                    // variables per thread of exa_grad_pull
};

struct Generated {
    std::string source;
    ParamLayout layout;
};

// threads per workgroup: 4 wavefronts (128 / 512 / 1024 measured within +-2 %, profiles/NOTES.md)
constexpr int kBlock = 256;

// loopfree_scatter: no loop around or inside the bodies of exa_grad / exa_jtprod / exa_hprod (see kHugeBody in
// exa_gen.hpp); the generator turns it on by itself for huge bodies, the runtime asks for it when a scatter kernel of the
// compiled module turns out to spill registers.
// nostage: no exa_hesscl (the LDS-staged chained kernel outgrew the architectural registers where this module was first compiled).
Generated generate_module(const Model &m, bool loopfree_scatter = false, bool nostage = false);
bool pattern_var_range(const Pattern &p, int64_t lo, int64_t hi, int64_t *vmin, int64_t *vmax);
// data points [jlo, jhi) of a gathered objective pattern with a first-order slot on one of the 1-based variables [v_lo, v_hi]
void pull_point_range(const Pattern &p, int64_t v_lo, int64_t v_hi, int64_t *jlo, int64_t *jhi);       // exa_gen_scatter.cpp   // exa_gen_module.cpp

// Windowed compressed-COO kernels (exa_cjac / exa_chess fast path).  One pattern of such a kernel: every
// slot s of data point I lands on compressed entry a_s + b * I; slots with the same a_s are added in registers (group),
// and the groups are split into phases such that the lanes of a workgroup never touch one LDS word within a phase.
struct WindowPat {             // one (pattern, stride class) pass
    int k = 0;                  // pattern index
    std::vector<int> group;     // slot -> group (-1: the slot advances with another stride; another pass adds it)
    std::vector<int> phase;     // group -> phase
    int qbase = 0;              // first word of this pass's table in Q: b, e_lo, e_hi, amin, amax, a[group]...
    int space = 0;              // block-owned variant: which output space (column block) the pass writes to
};
constexpr int kStageHalo = 16;     // doubles of halo a wavefront stages beyond its 64 points (exa_hesscl)
constexpr int kSharedTiles = 8;   // chunks of kBlock points per workgroup of the shared-entry kernel (exa_c*s)
struct WindowShared {          // slots of a pattern that land on ONE entry for every data point (b = 0: a literal index):
    int k = 0;                  // summed per workgroup, folded in a fixed order by the tail kernel (exa_*x)
    std::vector<std::vector<int>> groups;   // slots of each such entry
    // attach >= 0: the sums are formed INSIDE the window kernel, by the pass `attach` of the same pattern (every regular
    // point belongs to exactly one window: the one that holds the first target of that pass) — one partial per window and
    // group at part[Q[qs] + g * Q[qs + 1] + window]; attach < 0: by the kernel of their own (exa_*s: one more evaluation pass)
    int attach = -1, qs = 0;
};
// What a window kernel produces.  CJAC / CHESS: the compressed (duplicate-summed) COO arrays of exa_cjac / exa_chess — entry
// = position in the (col, row)-sorted structure, fitted to a_s + b*I on the device at exa_compress.  JTPROD / HPROD: the
// dense vectors J'v / Hv — entry = 0-based variable, a_s + b*I known statically from the index expressions (range value
// times a literal plus a literal): the owner-computes form of the products, no zero-fill, no atomics, fixed summation
// order.  The "slots" of a product pattern are its contributions merged per distinct variable (product_items).
enum WKind { WK_CJAC = 0, WK_CHESS = 1, WK_JTPROD = 2, WK_HPROD = 3, WK_COUNT = 4 };
struct WindowMatrix {
    std::vector<WindowPat> pats;
    std::vector<WindowShared> shared;        // summed by exa_*s
    std::vector<WindowShared> shared_in;     // summed inside the window kernel (one-chunk kernels)
    // every pass of every window fits one chunk of kBlock points: straight-line kernel (all passes' loads first, then
    // the additions), compiled for 8 waves per SIMD; otherwise chunk loops, no occupancy hint (it made them spill)
    bool single = false;
    // Block-owned variant (models laid out as separate variable arrays: the outputs of one data point land in several
    // far-apart column blocks = SPACES).  Workgroup j owns window j of EVERY space — the outputs of the same block of
    // points — so each pattern is evaluated once per point instead of once per pass.  nspaces > 0 selects it; the space
    // table [origin, end, window, LDS offset] x nspaces starts at word zs of Q.
    int nspaces = 0, zs = 0;
    // PLANES form of the one-chunk kernels, when every pass advances by exactly one entry per data point (b = 1: the dense
    // product vectors of stencil models): each lane writes the sums of its slot groups to LDS planes [group][lane], ONE
    // barrier, then the lane that owns an entry adds the planes' values for it in (pass, group) order — instead of a
    // zero-filled window and one read-modify-write phase + barrier per group.  Same additions in the same order.
    bool planes = false;
    bool empty() const { return pats.empty(); }
};
struct WindowSpec {
    WindowMatrix mat[WK_COUNT];
    // Matrices the windows do not fit (data-indexed targets: ACOPF): exa_cjacp / exa_chessp — the uncompressed sweep with
    // every slot stored at its position in the (col, row)-SORTED order (pos[slot], built once), so that the duplicates of
    // an entry are contiguous: the reduction reads sequentially instead of gathering 8-byte values at random, and a matrix
    // without duplicates (ACOPF's Jacobian) needs no reduction at all.
    bool jac_scatter = false, hess_scatter = false;
    // Compressed Hessian through MERGED slots (exa_chessm, see merge_slot in exa_gen_coo.cpp): the patterns of a fused group
    // add the slots they put on one matrix entry in registers; the merged slots are stored at their sorted positions.
    bool hess_merged = false;
};
// Product "slots" of pattern k for wk = WK_JTPROD / WK_HPROD: per merged contribution the target 0-based variable
// a[s] + b[s] * I (b = 0: a literal index, the same variable for every data point).  false = some target is not affine in
// a range column (data-indexed) — such a model keeps the atomics / the sorted gather.
bool product_items(const Model &m, const ParamLayout &L, int wk, int k, std::vector<int64_t> &a, std::vector<int64_t> &b);
// Owner-pull products (exa_gen_pull.cpp): the module holding exa_jtkeys / exa_jtpull (jt) and exa_hpkeys / exa_hppull (hp), and
// the number of items per fused group of CB_JTPROD / CB_HPROD (the item slots of group g are first[g] + items[g] * point + t)
std::string generate_pull_module(const Model &m, const ParamLayout &L, bool jt, bool hp);
std::vector<int> pull_item_counts(const Model &m, const ParamLayout &L, int cb);
// merged Hessian slots per data point of every fused group of CB_HESS (what exa_chessm would write)
std::vector<int> merged_hess_slots(const Model &m, const ParamLayout &L);
// Source of the second module of a compressed model: exa_chessw / exa_chessx (and exa_cjacw / exa_cjacx).
std::string generate_window_module(const Model &m, const ParamLayout &L, const WindowSpec &spec);

// ---------------------------------------------------------------------------------------------------
// Runtime services used by the recipe layer (exa_recipe.cpp)
// ---------------------------------------------------------------------------------------------------
struct BlockInfo {          // one named variable / constraint / parameter block of an instance (cnlp P_block)
    std::string name;
    int kind = 0;           // 0 variable, 1 constraint, 2 parameter
    int64_t offset = 0, length = 0;
    std::vector<int64_t> dims;
};
int create_model(const exa_model_desc_t *desc, int *id_out, bool device);   // C-ABI status; sets the last-error text
int attach_blocks(int id, std::vector<BlockInfo> blocks);
void set_last_error(const std::string &text);


// ---- user-registered functions (exa_register_univariate / _bivariate; exa_gen_rules.cpp) ------------------------------------------
// The reference's @register_univariate / @register_bivariate (src/register.jl:56-74, 123-276) with the derivative rules given as HIP
// device expressions.  Function ids start at EXA_USER_FN_BASE in both tables; a registration lives as long as the process.
constexpr int EXA_USER_FN_BASE = 1000;
// `fused` (univariate only, exa_register_univariate_fused): ONE device statement that leaves value, first and second derivative in
// $2 $3 $4 from the argument $1 — for functions whose derivatives share work with the value (a range reduction, an exp): then f/d1/d11 are empty.
struct UserFn { std::string name, f, d1, d2, d11, d12, d22, helpers, fused; };
int register_user_fn(bool bivariate, const UserFn &fn, std::string *err, bool dry_run = false, bool *known = nullptr);   // known: these very rules were registered before   // the new id, or -1 with *err set
                                                                  // (dry_run: every check, nothing entered; the id it would get)
const UserFn *user_fn(bool bivariate, int fn);                                  // nullptr: not registered

}  // namespace exa
