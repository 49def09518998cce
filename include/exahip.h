/* exahip.h — C ABI of libexahip.so: the MI355X-native evaluator behind ExaModels' NLPModels surface.
 *
 * Drop-in boundary.  The entry points mirror, name for name, the cnlp ABI v0.1 that the reference
 * itself exports for this surface (ExaModelsCompiler/src/ExaModelsCompiler.jl:1560-1732, prefix P = "exa"),
 * and each one replaces one NLPModels callback of ext/ExaModelsKernelAbstractions.jl:
 *
 *   exa_obj             <- obj            (KA ext :253-271,  CPU src/nlp.jl:1827-1839)
 *   exa_cons            <- cons_nln!      (KA ext :273-308,  CPU src/nlp.jl:1841-1854)
 *   exa_grad            <- grad!          (KA ext :310-336,  CPU src/nlp.jl:1858-1868)
 *   exa_jac_structure   <- jac_structure! (KA ext :212-231,  CPU src/nlp.jl:1798-1807)
 *   exa_jac             <- jac_coord!     (KA ext :338-351,  CPU src/nlp.jl:1870-1880)
 *   exa_hess_structure  <- hess_structure!(KA ext :232-250,  CPU src/nlp.jl:1809-1825)
 *   exa_hess            <- hess_coord!    (KA ext :515-547,  CPU src/nlp.jl:1906-1940)
 *   exa_jprod           <- jprod_nln!     (KA ext :369-387,  CPU src/nlp.jl:1882-1892)   matrix-free: no COO buffer
 *   exa_jtprod          <- jtprod_nln!    (KA ext :389-407,  CPU src/nlp.jl:1894-1904)
 *   exa_hprod           <- hprod!         (KA ext :409-511,  CPU src/nlp.jl:1942-1978)
 *   exa_set_value       <- set_value!     (src/nlp.jl:1279-1287)
 *   exa_new_from_table  <- build_extension (KA ext :33-191) + the pattern records of src/simdfunction.jl:78-100
 *
 * Conventions (same as cnlp): every function returns an int status — 0 ok, 1 bad id / bad argument,
 * 2 internal error (HIP failure, kernel compile failure; exa_last_error() has the text) — and never
 * throws or aborts across the boundary.  Counts return >= 0, or -1 on a bad id.  Indices are 1-based,
 * the Hessian is the lower triangle (row >= col), outputs are FULLY OVERWRITTEN (callers may pass
 * uninitialised memory, cf. benchmark/runbenchmark.jl:88-92).  Evaluation never errors on the values
 * of x (NaN/Inf propagate).
 *
 * Pointers: the unsuffixed callbacks take DEVICE pointers (what Julia passes as pointer(::ROCArray)) and are asynchronous on
 * the model's stream (exa_set_stream; default the null stream): a callback never measures, tunes, sorts or synchronises —
 * exa_tune is the explicit, blocking set-up step for measuring; the sorted lists a persisted or explicit decision needs
 * (exa_set_product_mode / exa_set_grad_mode / exa_set_deterministic) are built at model build, at those calls and at
 * exa_set_shard, never inside a callback; the only allocation a callback may make is the scratch of its FIRST call
 * (run every callback once before capturing a graph) — except exa_obj whose `out` is a HOST double and therefore
 * synchronises.  The *_host variants take host pointers, stage through device scratch owned by the model, and synchronise
 * (the role of WrapperNLPModel, src/utils.jl:159-208); on a sharded model they return zeros where the rank owns nothing.
 * One call in flight per model id (reference callbacks share scratch too, KA ext :21-31); distinct ids are independent.
 *
 * There is NO CPU fallback: if no HIP device or the kernel module cannot be built, exa_new_from_table
 * fails with status 2.
 *
 * Kernel build.  The generated module is compiled IN THE PROCESS with hiprtc (the copy next to the HIP runtime the process
 * has loaded) — the consumer's machine needs no hipcc —, or by `hipcc --genco` as a subprocess (EXAHIP_COMPILER=hipcc, or
 * when hiprtc cannot be loaded), and cached on disk under SHA-256(source) + SHA-256(flags | compiler identity | arch).
 * Cache directory: $EXAHIP_CACHE_DIR, else <install>/kernel_cache, else $XDG_CACHE_HOME/exahip or ~/.cache/exahip (0700);
 * only directories owned by the user (or root) and not writable by others are read or written; never /tmp.
 */
#ifndef EXAHIP_H
#define EXAHIP_H
#include <stddef.h>
#include <stdint.h>
#include "exahip_ir.h"

#ifdef __cplusplus
extern "C" {
#endif

#define EXAHIP_ABI_VERSION 3

int         exa_abi_version(void);
const char *exa_last_error(void);                       /* thread-local text of the last status-2 failure */

/* ---- build / destroy -------------------------------------------------------------------------- */
/* Plans the COO layout (slot maps comp1/comp2, o1step/o2step, running offsets in insertion order),
 * generates + compiles one HIP module for the model (cached on disk by source hash), uploads the SoA
 * iterator columns.  Returns id > 0 in *id_out.  The table is validated (node links, column references, integer-typed
 * index expressions, augmentation targets inside their base block, every x[...] / theta[...] index inside 1..nvar /
 * 1..npar for every data point): a malformed table is status 1, never a crash or an out-of-bounds device access. */
int exa_new_from_table(const exa_model_desc_t *desc, int *id_out);
/* Same, but only plan + generate source (no device needed): used by the CPU-side build check and tests. */
int exa_plan_only(const exa_model_desc_t *desc, int *id_out);
/* Compile the model's generated module for gfx950 into the on-disk cache without loading it (works without a GPU:
 * the build check).  exa_code_object_path() then names the .hsaco. */
int exa_compile(int id);
const char *exa_code_object_path(int id);
/* Name of the model's generated module: "exa_" + the first 128 bits of SHA-256(source), independent of the compiler. */
const char *exa_module_name(int id);
/* Hand a compiled code object to the library in memory under that name: models whose module has this name load it
 * instead of looking on disk or compiling.  A packed library (exahip.pack) ships its module this way — ahead-of-time,
 * like a compile_library product of ExaModelsCompiler (ExaModelsCompiler.jl:108-222).  Status 1 for anything that is
 * not an AMDGPU ELF code object (or a clang offload bundle of one). */
int exa_cache_add(const char *name, const void *code_object, size_t len);
/* A fact that travels with a module.  One so far: note "loopfree" for the name of a module whose scatter kernels were found
 * to spill registers where it was compiled (the model's module is then the one generated without loops around their bodies,
 * under another name): a later build that finds the note generates the final module at once.  exa_module_alias(id) is the
 * name to attach it to ("" when the model's module replaces nothing). */
int exa_cache_note(const char *name, const char *note);
const char *exa_module_alias(int id);
/* ... and the note to attach to it: "loopfree", "nostage" (the LDS-staged chained kernel exa_hesscl alone outgrew the architectural
 * registers where the module was first compiled: the model's module is the one generated without it) or "loopfree+nostage"; "" when
 * the model's module replaces nothing. */
const char *exa_module_alias_note(int id);
/* The code objects of a compiled model (after exa_compile, or a device model): k = 0 the model's module, k = 1 the module of
 * the owner-computes product windows when the model has them (exa_product_info).  name <- what exa_cache_add takes,
 * path <- the file.  exahip.pack embeds all of them. */
int exa_code_object_count(int id);
int exa_code_object(int id, int k, char *name, int ncap, char *path, int pcap);
/* How the module of a device model was obtained: how <- "preloaded" | "disk" | "hiprtc" | "hipcc", *build_ms <- the
 * compiler's time (0 unless it ran).  The reference pays this at the first call of every callback (Julia's JIT). */
int exa_build_info(int id, char *how, int cap, double *build_ms);
/* What the compiled kernels need.  After every compilation (the model's module, the product windows, the second module of
 * exa_compress) the library reads the resources of ALL kernels of the code object from its metadata; a module holding a kernel
 * that has outgrown the 256 architectural VGPRs (AGPRs or scratch as spill space), or whose metadata cannot be read, is built
 * again with conservative register-allocation flags ($EXAHIP_SAFE_FLAGS, default "-mllvm -grow-region-complexity-budget=0";
 * "none" switches the fallback off) — such kernels have returned wrong sums under the default allocator
 * (tests/sweeps/canary/).  buf <- one line per kernel, "<module: model|products|compressed> <object name> <flags: default|safe>
 * <kernel> <vgpr> <agpr> <scratch bytes per lane> <spilled vgprs> <spilled sgprs> <lds bytes> <fits|oversized>", and a line
 * "<module> <object name> <flags> ? unreadable" for a module whose metadata could not be read.  Returns the length of the
 * whole report (call again with a larger buffer when it is >= cap), -1 for a bad id.  Works for plan-only handles after
 * exa_compile. */
int exa_build_audit(int id, char *buf, int cap);
/* Test infrastructure (tests/sweeps/canary/): everything one launch of exa_jtprodw (hess = 0) / exa_hprodw (hess = 1) needs —
 * launch shape, arguments, the output of THIS build — into `path`, the module's source into `path`.hip.  x, y, v: device. */
int exa_debug_dump_window_launch(int id, int hess, const double *x, const double *y, const double *v, double sigma, const char *path);
int exa_free(int id);

/* ---- sizes (cnlp: P_nvar/P_ncon/P_nnzj/P_nnzh, Compiler :1564-1582) ---------------------------- */
int     exa_nvar(int id);
int     exa_ncon(int id);
int     exa_nnzj(int id);
int     exa_nnzh(int id);
int64_t exa_nvar64(int id);
int64_t exa_ncon64(int id);
int64_t exa_nnzj64(int id);
int64_t exa_nnzh64(int id);
int64_t exa_nnzg64(int id);     /* number of sparse objective-gradient slots (ExaCore.nnzg) */
int     exa_npatterns(int id);
/* Per-pattern layout record, for tests and for hosts that consume the COO in place:
 * out[0]=kind out[1]=n out[2]=o0 out[3]=o1 out[4]=o2 out[5]=o1step out[6]=o2step out[7]=#leaf visits(1st) out[8]=#visits(2nd) */
int     exa_pattern_info(int id, int p, int64_t out[9]);
/* Slot maps: comp1 (length info[7]) / comp2 (length info[8]), 1-based like Compressor (simdfunction.jl:11-14). */
int     exa_pattern_comp(int id, int p, int order, int32_t *out);
/* x0,lvar,uvar [nvar]; lcon,ucon [ncon] — HOST pointers (cnlp P_meta, Compiler :1584-1608). */
int     exa_meta(int id, double *x0, double *lvar, double *uvar, double *lcon, double *ucon);
/* perm_out [n of the pattern] <- a locality-improving order of the pattern's data points (0-based, stable): by the smallest variable
 * any x[...] of the pattern reaches at the point — a branch table: by from-bus, the order of a MATPOWER / PGLIB case file.  The
 * gathers of data-indexed kernels are line-granular: the same ACOPF network runs 35-40 % faster with its branches in that order
 * (profiles/r3_acopf_topology.json).  The library never re-orders a table itself (row order = constraint-row and COO-slot order,
 * nlp.jl:1991-1992): the caller applies the permutation to the table (and to y / bounds of the rows it generates) BEFORE
 * building the model.  Needs the host columns: an exa_plan_only handle. */
int     exa_locality_order(int id, int pattern, int64_t *perm_out);
/* (Round 5 also let the library run its order-free kernels — grad!, J'v, Hv by atomics — on a locality-ordered COPY of every table
 * (exa_set_locality).  Measured on the 78 484-bus ACOPF with random branch ends the copies LOST (J'v 0.040 -> 0.054 ms, profiles/r5_locality_ab.txt:
 * sorting the branch rows by bus turns the coalesced v[row] / y[row] reads of the fused branch group into gathers); round 6 removed them —
 * they doubled the table memory of every model and put a row indirection on the hot path of kernels that never used it.) */
/* Generated HIP source of the model's module (NUL-terminated, owned by the library). */
const char *exa_kernel_source(int id);
/* ... of module k: 0 the model's module (= exa_kernel_source), 1 the product windows' ("" when the model has none). */
const char *exa_module_source(int id, int k);

/* ---- execution context -------------------------------------------------------------------------
 * Device: a model lives on the HIP device that was current in the thread that created it; every device call makes that
 * device current for its duration and restores the caller's (the HIP current device is per host thread — a Julia task
 * that migrated, a worker thread that never called hipSetDevice).
 * Graphs: once a callback has run once (its scratch buffers exist) it is a pure sequence of launches / memsets on the
 * model's stream — no allocation, no synchronisation — so the host may capture a whole solver iteration (exa_obj_async,
 * exa_grad, exa_cons, exa_jac, exa_hess, the products) with hipStreamBeginCapture on that stream and replay it on new
 * iterates held in the same buffers (tests/test_gpu_graph.py).  exa_obj (host result), the *_host variants, exa_tune,
 * exa_compress and the structure calls synchronise and cannot be captured. */
int exa_set_stream(int id, void *hip_stream);
/* Shard every pattern's iterator: this process evaluates data points [floor(n*rank/world), floor(n*(rank+1)/world))
 * of every pattern.  What a rank then writes (without a communicator, or with exa_set_reduce(id, 0)):
 *   jac / hess / structures   its data points' COO slots at their GLOBAL positions (or packed, exa_set_coo_local): disjoint;
 *   cons, jprod               the base rows of its data points, COMPLETE (the row's owner evaluates the row's augmentation
 *                             terms itself) — other rows are not touched.  Only models with rows collecting > 512 terms
 *                             fall back to partial sums over the whole vector;
 *   grad                      objective patterns gathered per variable (range-affine): the variables
 *                             [floor(nvar*rank/world), floor(nvar*(rank+1)/world)), COMPLETE, nothing else touched;
 *                             with a pattern that scatters through a data index: partial sums over the whole vector;
 *   jtprod, hprod             owner-computes windows (exa_set_product_mode 2): the windows the rank owns, COMPLETE;
 *                             otherwise partial sums over the whole vector;
 *   obj                       a partial sum.
 * rank=0, world=1 restores the unsharded model. */
int exa_set_shard(int id, int rank, int world);

/* ---- multi-GPU behind the ABI: one process per GPU, collectives on the model's stream -----------------
 * What the reference accumulates on one device — obj (KA ext :253-271), grad! (:310-336), cons_nln! with its
 * augmentation rows (:273-308) and the products — is a SUM over data points.  Here most of it is sharded by OWNER instead
 * (see exa_set_shard): ranks hold complete, disjoint pieces, and with a communicator attached the callbacks make every
 * vector whole on every rank by the cheapest collective that does it, enqueued right after the kernels:
 *   obj                              all-reduce(sum) of 1 double;
 *   cons, jprod (rows owned)         all-gather-v of the row slices (no zero-fill, nothing summed, each piece travels once);
 *   grad (variables owned)           all-gather-v of the variable slices;   grad with data-indexed patterns: all-reduce(nvar);
 *   jtprod / hprod                   windows: all-gather-v of the owned windows;  atomics / sorted gather: all-reduce(nvar);
 *   jac_coord! / hess_coord! / structures   NO collective: COO slots are private to a data point (exa_allgather_coo makes
 *                                    a sharded COO vector whole where a consumer wants that).
 * exa_set_reduce(id, 0) keeps the communicator but leaves the pieces / partial sums as they are.
 *
 *   exa_comm_unique_id   rank 0 obtains an ncclUniqueId (128 bytes) and the host distributes it (MPI.bcast, a file, ...);
 *   exa_comm_init        every rank: ncclCommInitRank on the process's current HIP device (collective) + exa_set_shard;
 *   exa_comm_attach      adopt a communicator the host created itself (rank / world are read from it; not destroyed by
 *                        exa_comm_free / exa_free);
 *   exa_comm_hook        no RCCL: the host supplies `fn(ctx, device_buffer, count, hip_stream)` that must leave the sum
 *                        over ranks in device_buffer, ordered after the work already enqueued on hip_stream (MPI.jl with
 *                        a GPU-aware MPI, or a staged host all-reduce: the CPU/gloo test path); non-zero return = failure;
 *   exa_comm_free        detach (the shard stays);   exa_set_reduce(id, 0) keeps the communicator but returns partial sums;
 *   exa_allreduce        the same sum for a buffer of the host's own (e.g. a zero-filled COO vector it wants whole).
 * librccl is loaded with dlopen at exa_comm_unique_id / exa_comm_init / exa_comm_attach: single-GPU use needs no RCCL. */
#define EXA_UNIQUE_ID_BYTES 128
typedef int (*exa_allreduce_fn)(void *ctx, double *device_buffer, int64_t count, void *hip_stream);
int exa_comm_unique_id(void *out128);
int exa_comm_init(int id, int rank, int world, const void *unique_id128);
int exa_comm_attach(int id, void *nccl_comm);
int exa_comm_hook(int id, int rank, int world, exa_allreduce_fn fn, void *ctx);
int exa_comm_free(int id);
int exa_comm_info(int id, int *rank, int *world, int *kind);      /* kind: 0 none, 1 RCCL, 2 host reducer */
int exa_set_reduce(int id, int on);
int exa_allreduce(int id, double *device_buffer, int64_t count);
/* How a rank of a sharded model leaves the output of callback `which` when nothing completes it (no communicator, or
 * exa_set_reduce(id, 0)): 1 = OWNER PIECES (complete values in disjoint pieces, nothing else written: an all-gather makes
 * the vector whole), 0 = PARTIAL SUMS over the whole vector (an all-reduce(sum) completes it), -1 bad id / argument.
 * which: 0 obj, 1 grad, 2 cons, 3 jac, 4 hess, 5 jprod, 6 jtprod, 7 hprod, 8 the cons vector of exa_eval_fused / exa_eval_all
 * (differs from 2 for models with NON-linear augmentation terms: the fused sweeps leave partial sums there, exa_cons complete
 * rows).  6 / 7 answer for the implementation a call would run — explicit mode, else the persisted exa_tune decision, else the
 * windows: atomics and the sorted gather leave partial sums, the windows owner pieces (the same resolution as exa_product_info). */
int exa_shard_layout(int id, int which);
/* The collective operations that complete the output of callback `which` (numbering of exa_shard_layout; 3 / 4 = the COO vectors as
 * exa_allgather_coo gathers them): out <- up to cap operations of 4 words: kind, offset, count, root.  kind 0 = in-place all-gather
 * (every rank contributes `count` doubles, rank r's at offset + r * count: ONE ncclAllGather — data points, variables and windows
 * are split into equal pieces, the remainder on the last rank), 1 = broadcast of [offset, offset + count) from `root` (the last
 * rank's surplus; irregular pieces), 2 = all-reduce(sum) of [offset, offset + count) (vectors left as partial sums).  This is what
 * the library issues through RCCL inside one ncclGroupStart / End, and what a host layer with its own transport (MPI.jl) should
 * issue.  Returns the number of operations (0 for world 1), -1 for a bad argument.  No device needed. */
int exa_collective_plan(int id, int which, int64_t *out, int cap);
/* Deferred completion: issues, on the model's stream through the attached RCCL communicator, exactly the operations of
 * exa_collective_plan(which) on buf (the callback's output vector, global length) — what the callback does itself at its end unless
 * exa_set_reduce(id, 0) switched that off: a host overlaps the collective of one callback with the kernels of the next.  With a WORLD-1
 * communicator the one-piece plan is issued all the same (a real in-place ncclAllGather / ncclAllReduce of the vector onto itself): how a
 * single-GPU machine exercises transport and plan.  0 ok, 1 bad argument / no RCCL communicator, 2 internal. */
int exa_comm_complete(int id, int which, double *buf);
/* A sharded Jacobian (hess = 0) / Hessian (hess = 1) COO vector made whole on every rank: all-gather-v of the ranks' slot
 * ranges (a piece travels once; an all-reduce of zero-padded vectors would move world x the data).  `local` = what this
 * rank's exa_jac / exa_hess wrote: the packed local slice (exa_set_coo_local) or the global-length vector with the rank's
 * slots in place; `global` [nnzj | nnzh] receives everything and may be `local` itself in the second case.  DEVICE pointers,
 * asynchronous on the model's stream.  Replaces a hess_coord! whose contract is the full vector on the caller's device
 * (src/nlp.jl:1906-1940) for consumers that are not sharded themselves. */
int exa_allgather_coo(int id, int hess, const double *local, double *global);
/* Local-slice COO.  By default a sharded rank writes its Jacobian / Hessian / structure slots at their GLOBAL positions
 * (the caller's buffers have nnzj / nnzh entries, the rank fills its disjoint part).  With exa_set_coo_local(id, 1) the
 * rank's slots are PACKED: pattern after pattern, each pattern's data points [lo, hi) in order — buffers of
 * exa_local_nnzj64 / exa_local_nnzh64 entries (1/world of the model: LV N=1e8 on 8 GPUs: 0.9 GB instead of 7.2 GB per
 * rank).  exa_coo_slices tells where each piece belongs: out[3k .. 3k+2] = first GLOBAL slot (0-based) of pattern k's
 * piece, its first position in the caller's buffer, its length.  Sorted products and exa_compress work on the local
 * slice of a sharded model (each rank compresses the matrix of its own data points; the model's matrix is the sum). */
int     exa_set_coo_local(int id, int on);
int64_t exa_local_nnzj64(int id);
int64_t exa_local_nnzh64(int id);
int     exa_coo_slices(int id, int hess, int64_t *out /* 3 * exa_npatterns */);
/* 0-based [lo, hi): the stretch of x this rank's data points read.  The whole vector when some index comes from a data
 * column; the shard's stencil footprint when all indices are range-affine — the host may then keep only that stretch
 * resident and pass `x_slice - lo` as x (and analogously y rows [o0+lo, o0+hi) per constraint pattern). */
int exa_shard_var_range(int id, int64_t *lo, int64_t *hi);
/* User-registered functions — the reference's @register_univariate / @register_bivariate (src/register.jl:56-74, 123-276: a function
 * with closed-form first and second derivatives becomes a node type).  There the rules are Julia lambdas; across a C ABI they are HIP
 * device EXPRESSIONS (C++ text, Float64) with placeholders, compiled into the modules of the models that use the function:
 *   univariate:  f in $1 (the argument);  df in $1, $2 (= f($1), already computed);  ddf in $1, $2, $3 (= df)
 *   bivariate:   f in $1, $2;  d1, d2, d11, d12, d22 in $1, $2, $3 (= f($1, $2))
 * A rule that is an exact constant may be written "=0", "=1", "=-2.5": the reverse sweep then folds it like the table's own constants.
 * `helpers` (may be NULL): device code placed once in front of the kernels of every module that uses the function (static __device__
 * functions, #defines) — anything the expressions call beyond the HIP math library.  Returns the function id (>= 1000) to put into
 * exa_node_t.fn with op = EXA_OP_UN / EXA_OP_BIN, or -1 (exa_last_error: bad name, missing rule, unknown placeholder, the name already
 * registered with other rules; the same rules again return the same id).  A registration lasts for the process; a model file / recipe that
 * uses such ids carries the registrations (trailing section of the wire format, include/exahip_recipe.h): loading registers them and
 * renumbers the nodes, whatever the loading process registered before.  A rule that does not compile fails the model build (status 4, the compiler's message in exa_last_error), not the registration. */
int exa_register_univariate(const char *name, const char *f, const char *df, const char *ddf, const char *helpers);
int exa_register_bivariate(const char *name, const char *f, const char *d1, const char *d2, const char *d11, const char *d12,
                           const char *d22, const char *helpers);
/* A univariate function whose derivatives share work with the value (a range reduction, an exponential, a series): ONE device STATEMENT
 * `stmt` computes all three — $1 the argument, $2 / $3 / $4 the variables that receive f, f', f'' — e.g.
 *   "exa_sincos($1, &$2, &$3); $4 = -$2;"        (what the table itself does for sin: src/functionlist.jl:22)
 * emitted once per distinct argument inside a block of its own (it may declare temporaries); kernels that need the value only leave
 * the rest to the compiler's dead-code elimination.  Measured on LV N = 1e7 (profiles/r4_userfn_ab.txt): a sine registered this way costs what the table's sine costs; as three separate rules it pays
 * the reduction twice (jac_coord! +26 %, hess_coord! +9 %).  Same id space, return values and lifetime as exa_register_univariate. */
int exa_register_univariate_fused(const char *name, const char *stmt, const char *helpers);
/* Read a registration back (what a writer of model files needs to make them self-contained, include/exahip_recipe.h): which = 0 name,
 * 1 f, 2 d1 (df), 3 d2, 4 d11 (ddf), 5 d12, 6 d22, 7 helpers, 8 the fused statement.  Copy-out convention of the cnlp ABI: returns the
 * byte length of the text and copies what fits into buf (NUL-terminated when cap > 0); -1 = no such function / bad argument. */
int exa_user_function(int bivariate, int fn, int which, char *buf, int cap);
/* theta update without rebuild (set_value!, nlp.jl:1279-1287; cnlp :1529-1535) */
int exa_set_value(int id, int64_t offset, const double *vals, int64_t len);   /* theta[offset .. offset+len) <- vals (HOST) */
/* ... and for parameters that live on the device (the reference's set_value! is a copyto! into the device-resident θ and its
 * get_value a device view, nlp.jl:1270-1287): theta[offset .. offset+len) <- dev_vals by a device-to-device copy ordered on the
 * model's stream — no PCIe hop, no synchronisation, capturable into a graph; exa_theta_ptr = the device vector itself (npar
 * doubles; NULL without a device or without parameters).  exa_get_value afterwards reads the device copy back first. */
int exa_set_value_dev(int id, int64_t offset, const double *dev_vals, int64_t len);
double *exa_theta_ptr(int id);

/* ---- callbacks, DEVICE pointers ------------------------------------------------------------------ */
int exa_obj (int id, const double *x, double *out_host);
int exa_obj_async(int id, const double *x, double *out_dev);                  /* result left on device, no sync */
int exa_grad(int id, const double *x, double *g);
int exa_cons(int id, const double *x, double *c);
int exa_jac (int id, const double *x, double *vals);
int exa_hess(int id, const double *x, const double *y, double obj_weight, double *vals);   /* y == NULL: objective only, see below */
int exa_jprod (int id, const double *x, const double *v, double *Jv);             /* Jv [ncon]  = J(x) v,   v [nvar] */
int exa_jtprod(int id, const double *x, const double *v, double *Jtv);            /* Jtv [nvar] = J(x)' v,  v [ncon] */
int exa_hprod (int id, const double *x, const double *y, const double *v, double obj_weight, double *Hv);  /* Hv [nvar] */
/* Objective-only forms — hess_coord!(m, x, hess; obj_weight), hprod!(m, x, v, Hv; obj_weight) (nlp.jl:1906-1915, :1942-1952):
 * exa_hess / exa_hprod (and their _host variants) with y == NULL.  As in the reference the constraint patterns are NOT evaluated:
 * the objective patterns are launched alone and the constraint slots receive exact zeros (a constraint whose second derivative
 * is Inf / NaN at x cannot leak 0 * Inf into the result); exa_hprod with y == NULL runs by atomics or sorted gather even where
 * the full product has windows.  Asynchronous like every callback, capturable from the first call. */
/* exa_jtprod / exa_hprod have four implementations.  mode 0: FP64 atomics inside the sweep (zero-fill + atomics; order of
 * additions varies).  mode 1: COO + gather through build-time sorted lists (the reference's prod helper, KA ext :56-178,
 * :482-511; deterministic).  mode 2: OWNER-COMPUTES WINDOWS — models whose every scatter target is (range value) * literal +
 * literal (stencil models: LV, discretised ODEs): a workgroup owns a window of consecutive variables, evaluates the data
 * points that touch it (the few straddling two windows twice), adds their contributions in LDS in a fixed order and streams
 * the window out with plain coalesced stores — no zero-fill, no atomics, bit-reproducible; a sharded model owns a range of
 * windows per rank (complete values, all-gather-v instead of all-reduce).  mode 3: OWNER PULL — models whose targets come from
 * data columns (ACOPF: bus variables reached through the branch table; no windows there): a list target variable -> the (fused
 * group, data point, item) contributions that land on it is built once (exa_set_product_mode(…, 3) / exa_set_deterministic or a persisted
 * decision at model build: never inside a callback; exa_tune no longer tries it — it lost to the atomics on every model measured), and a thread per variable re-evaluates its items — one specialised function
 * per item — and stores the sum: no zero-fill, no atomics, a fixed order of additions (bit-reproducible), every entry written
 * exactly once; unsharded models whose variables collect at most 512 contributions each (status 1 otherwise).  -1 (default)
 * undecided: the decision exa_tune measured and persisted for this module / device / sizes ("chosen by measured contention") if
 * there is one, else mode 2 where the model has windows, else 0.  The mode is fixed before a call; a callback never measures.
 * EXAHIP_PRODUCT_WINDOW=0 / EXAHIP_PRODUCT_PULL=0 keep the second module from being generated at all. */
int exa_set_product_mode(int id, int jtprod_mode, int hprod_mode);
int exa_get_product_mode(int id, int *jtprod_mode, int *hprod_mode);
/* What exa_jtprod (hess = 0) / exa_hprod (hess = 1) would run now: 0 | 1 | 2 | 3 as above, -1 bad id; buf <- the kernel shape of the
 * owner-computes windows ("one chunk per pass" | "chunk loops" | "block-owned windows, K spaces") or why the model has none.
 * Works on exa_plan_only handles too (the plan is host-only; EXAHIP_PRODUCT_WINDOW=0 disables it). */
int exa_product_info(int id, int hess, char *buf, int cap);
/* exa_grad likewise: 0 = affine-indexed objective patterns gathered per variable + data-indexed ones added with FP64
 * atomics, 1 = the reference's scheme (KA ext :310-336: gradient COO of ExaCore.nnzg slots, (variable, slot) lists sorted
 * once, every variable's slots added in slot order — deterministic; a variable shared by millions of data points is
 * summed cooperatively instead of by as many atomics on one cache line), -1 (default) undecided: the persisted exa_tune
 * decision (what & 4) if there is one, else 0.  A sharded model always runs 0. */
int exa_set_grad_mode(int id, int mode);
int exa_get_grad_mode(int id, int *mode);
/* All three switches at once.  on != 0: exa_grad, exa_jtprod and exa_hprod by sorted gather wherever the model allows it
 * (not a model sharded at global COO positions) — with that, EVERY callback is bit-reproducible run to run; on == 0: back
 * to undecided (-1). */
int exa_set_deterministic(int id, int on);
int exa_jac_structure  (int id, int32_t *rows, int32_t *cols);
int exa_hess_structure (int id, int32_t *rows, int32_t *cols);
int exa_jac_structure64 (int id, int64_t *rows, int64_t *cols);               /* Julia Vector{Int} */
int exa_hess_structure64(int id, int64_t *rows, int64_t *cols);

/* ---- callbacks, HOST pointers (synchronous) ------------------------------------------------------ */
int exa_obj_host (int id, const double *x, double *out);
int exa_grad_host(int id, const double *x, double *g);
int exa_cons_host(int id, const double *x, double *c);
int exa_jac_host (int id, const double *x, double *vals);
int exa_hess_host(int id, const double *x, const double *y, double obj_weight, double *vals);
int exa_jprod_host (int id, const double *x, const double *v, double *Jv);
int exa_jtprod_host(int id, const double *x, const double *v, double *Jtv);
int exa_hprod_host (int id, const double *x, const double *y, const double *v, double obj_weight, double *Hv);
int exa_jac_structure_host  (int id, int32_t *rows, int32_t *cols);
int exa_hess_structure_host (int id, int32_t *rows, int32_t *cols);
int exa_jac_structure64_host (int id, int64_t *rows, int64_t *cols);
int exa_hess_structure64_host(int id, int64_t *rows, int64_t *cols);

/* ---- fused sweep (new) ---------------------------------------------------------------------------- */
/* obj, cons_nln!, jac_coord! and hess_coord! at one x from ONE second-order forward sweep (value and first partials
 * are by-products of it, src/graph.jl:416-447): one launch instead of four, transcendental work done once.
 * All DEVICE pointers; *obj_dev receives the objective value on the device (no synchronisation). */
int exa_eval_fused(int id, const double *x, const double *y, double obj_weight, double *obj_dev, double *c, double *jvals, double *hvals);
/* ... and grad! with them: all FIVE callbacks of a solver iteration at one x (the call pattern of
 * test/NLPModelsIpoptLite.jl/src/NLPModelsIpoptLite.jl:28-40).  The objective patterns of the sweep hold their first partials
 * already: patterns that scatter through a data index add them to g inside the sweep (no second evaluation, no exa_grad
 * launch); range-affine ones are gathered per variable by exa_grad_pull, launched BEFORE the sweep so that x is still warm
 * in the 256 MB MALL when the sweep reads it.  g [nvar] is fully overwritten.  Results equal the separate callbacks
 * (bitwise, except the atomically added part of g).  With exa_set_grad_mode(1) the gradient comes from the sorted gather. */
int exa_eval_all(int id, const double *x, const double *y, double obj_weight, double *obj_dev, double *g, double *c, double *jvals, double *hvals);
/* How exa_eval_all produces grad! on this model as it stands (shard, modes) — in ONE launch where it can (round 5):
 *   1  the gathered gradient's tiles ride inside the sweep's launch as one more unit of its block map, interleaved with the tiles
 *      that pull their stretch of x into L2 (range-affine objective: LV, the rocket);
 *   2  the sweep's objective tiles STORE their first partials — no atomics, no zero-filled g — and zero tiles of the same launch write
 *      0.0 to every other variable: a data-indexed objective whose scatter the model build has PROVEN injective on the data (a bitmap of
 *      the written variables; ACOPF's generator costs); unsharded models;
 *   0  a zero-fill launch, then the sweep adds by FP64 atomics;   3  the gathered part in a launch of its own, then the sweep adds the
 *      scattered part by atomics;   4  grad! by the sorted gather, separately (exa_set_grad_mode(1) / exa_tune).      -1: bad id. */
int exa_eval_all_mode(int id);
/* (exa_grad on its own takes the form of mode 2 wherever exa_eval_all does: one launch — stores + zero tiles — instead of a zero-fill
 * launch followed by atomics.) */

/* ---- compressed COO: duplicate (row,col) entries summed (CompressedNLPModel, src/utils.jl:425-579; KA ext :1290-1319) --- */
/* One-off set-up on the device: sorts the (col,row) pairs of both structures (stable), builds ptr/perm.  Entries come
 * out sorted by (col, row); duplicates are added in ascending original slot order (utils.jl:476-478, 555-562). */
int     exa_compress(int id);
int64_t exa_cnnzj64(int id);                  /* number of distinct Jacobian entries, -1 before exa_compress */
int64_t exa_cnnzh64(int id);
int exa_cjac_structure   (int id, int32_t *rows, int32_t *cols);              /* DEVICE pointers */
int exa_chess_structure  (int id, int32_t *rows, int32_t *cols);
int exa_cjac_structure64 (int id, int64_t *rows, int64_t *cols);
int exa_chess_structure64(int id, int64_t *rows, int64_t *cols);
/* The compressed entries are sorted by column, then row: they ARE the nzval of a CSC matrix.  colptr [nvar+1] and
 * rowval [cnnz], 1-based like Julia's SparseMatrixCSC(ncon | nvar, nvar, colptr, rowval, vals) — the values written by
 * exa_cjac / exa_chess go straight into a sparse direct solver's KKT blocks, no reordering.  DEVICE pointers. */
int exa_cjac_csc (int id, int64_t *colptr, int64_t *rowval);
int exa_chess_csc(int id, int64_t *colptr, int64_t *rowval);
int exa_cjac (int id, const double *x, double *vals);                          /* vals [cnnzj], DEVICE pointers */
int exa_chess(int id, const double *x, const double *y, double obj_weight, double *vals);   /* y == NULL on a constrained model: objective only where the gather runs (exa_compress_info == 0), status 1 (bad input) where windows / the permuted store do */
/* Which implementation exa_cjac (hess = 0) / exa_chess (hess = 1) run: 2 = permuted store (matrices the windows do not fit
 * — data-indexed targets such as ACOPF's bus variables: the sweep stores every slot at its position in the (col, row)-sorted
 * order, so the duplicates of an entry are contiguous and are summed with sequential reads, in ascending slot order like the
 * gather; a matrix without duplicates is written in place with no reduction at all; EXAHIP_CSCATTER=0 disables), 1 = windowed (the sweep adds straight into
 * LDS-resident windows of the compressed array: no uncompressed round trip; taken when every slot of every pattern sits
 * at compressed entry a_s + b*I — stencil models), 0 = uncompressed evaluation + sorted gather (the reference's scheme;
 * data-indexed targets, variables shared by all points), -1 = bad id / not compressed.  buf receives the reason (0) or
 * the kernel shape (1: "one chunk per pass" | "chunk loops" | "block-owned windows, K spaces"); the needed byte length
 * is returned through *len_out if non-NULL.  EXAHIP_CWINDOW=0 forces the gather. */
int exa_compress_info(int id, int hess, char *buf, int cap, int *len_out);

/* ---- measurement hooks --------------------------------------------------------------------------- */
/* Runs the named callback `reps` times on the model's stream bracketed by hipEvents recorded on THAT
 * stream and returns the average milliseconds per call in *ms_out.  which: 0 obj,1 grad,2 cons,3 jac,4 hess, 5 an (almost) empty
 * launch of the model's module — the floor a launch-bound callback is measured against.
 * Device pointers as in the callbacks (unused ones may be NULL). */
int exa_time_callback(int id, int which, int reps, const double *x, const double *y, double obj_weight,
                      double *out, float *ms_out);
int exa_sync(int id);
/* Order of the (pattern, tile) workgroups of a multi-pattern callback: 0 patterns one after the other, 1 interleaved in runs of 128
 * workgroups, -2 bad argument.  which: 2 cons, 3 jac, 4 hess, 5 fused.  Decided AT PLAN TIME, no tuning call needed (the reference has none,
 * ext/ExaModelsKernelAbstractions.jl:526-537): 1 where the two heaviest dispatch units of the callback walk the same stretch of x (their index
 * expressions are affine in a range column and their variable ranges overlap by half) and the call streams < 1.5 GB — the stretch one unit
 * just read is then still in the Infinity Cache when the other arrives —, else 0; a decision exa_tune persisted wins over this default. */
int exa_block_order(int id, int which);
/* hess_coord! has three generated kernels: 0 = exa_hess, one (pattern, 256-point tile) per workgroup; 2 = exa_hessc, a
 * workgroup walks 4 consecutive tiles of a GROUP of co-indexed patterns with the next inputs loaded before the current
 * tile is stored — wins where a call streams gigabytes through HBM, loses on cache-resident models (twice the registers);
 * 1 = exa_hesscl, exa_hessc with each wavefront's stretch of x (64 points + halo) loaded once and staged through LDS —
 * generated when every pattern of every group reads x at (one unit-step range) + literal offsets no more than 16 apart,
 * and used when this shard's groups start within that halo of each other (else 1 runs exa_hessc and this returns 2).
 * Which one runs: by size at plan time (>= 1.5 GB streamed per call -> 1, else 0) — what exa_tune picks on the benchmark shapes —, unless exa_tune
 * measured and persisted another choice for this module / device / sizes. */
int exa_hess_variant(int id);
/* Dynamic LDS (bytes) the chained hess_coord! kernels (variants 1, 2) are launched with: an occupancy throttle — memory nobody uses that
 * leaves three or two workgroups per CU instead of as many as the registers allow.  Fewer, longer streams per CU win where the output
 * outgrows the Infinity Cache (LV 1e8: 1.70 -> 1.62 ms) and on some boxes below it; exa_tune measures none / three / two next to the
 * kernels themselves and persists the winner, EXAHIP_HESS_DYN_LDS=bytes fixes it.  Without a decision: three workgroups per CU for a model that
 * streams >= 1.5 GB per call, else 0 = no throttle.  A value from the environment or from a persisted decision is used only when the kernel's
 * static LDS + it fit the 64 KB a launch may ask for (else 0). */
int exa_hess_throttle(int id);
/* Explicit, BLOCKING tuning — the only entry point that measures, and OPTIONAL: every decision has a plan-time default (above) that equals what
 * this call picks on the benchmark shapes (bench.py prints both).  what: bit 0 = block order of cons / jac / hess / fused
 * (models streaming >= 128 MB from several patterns), bit 1 = exa_jtprod / exa_hprod implementation, bit 2 = exa_grad
 * implementation (exa_set_grad_mode; only models whose objective scatters through a data index).  Both candidates of
 * each decision are timed on the model's stream at (x, y) (DEVICE pointers; NULL = x0 / ones; outputs go to scratch),
 * the winner is installed and persisted next to the cached module under (module, device, shard, pattern sizes), so later
 * processes start with it.  Call it once after the build (and after exa_set_shard / exa_comm_init), outside any stream
 * capture. */
int exa_tune(int id, int what, const double *x, const double *y);

#ifdef __cplusplus
}
#endif
#endif /* EXAHIP_H */
