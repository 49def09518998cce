"""ctypes binding of libexahip.so — the C ABI declared in include/exahip.h (nothing else is called).

Fails loudly if the library is missing: there is no Python or CPU fallback for evaluation.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libexahip.so")
CSRC = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
_LIB = None

# every symbol include/exahip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "exa_abi_version", "exa_last_error", "exa_new_from_table", "exa_plan_only", "exa_compile", "exa_code_object_path",
    "exa_free", "exa_module_name", "exa_cache_add", "exa_cache_note", "exa_module_alias", "exa_module_alias_note", "exa_code_object_count", "exa_code_object", "exa_nvar", "exa_ncon", "exa_nnzj", "exa_nnzh", "exa_nvar64", "exa_ncon64", "exa_nnzj64", "exa_nnzh64",
    "exa_nnzg64", "exa_npatterns", "exa_pattern_info", "exa_pattern_comp", "exa_meta", "exa_locality_order", "exa_comm_complete", "exa_eval_all_mode", "exa_kernel_source", "exa_module_source",
    "exa_register_univariate", "exa_register_univariate_fused", "exa_register_bivariate", "exa_user_function", "exa_set_stream", "exa_set_shard", "exa_set_value", "exa_set_value_dev", "exa_theta_ptr", "exa_obj", "exa_obj_async", "exa_grad", "exa_cons", "exa_jac",
    "exa_hess", "exa_jprod", "exa_jtprod", "exa_hprod", "exa_jprod_host", "exa_jtprod_host", "exa_hprod_host", "exa_jac_structure", "exa_hess_structure", "exa_jac_structure64", "exa_hess_structure64",
    "exa_obj_host", "exa_grad_host", "exa_cons_host", "exa_jac_host", "exa_hess_host", "exa_jac_structure_host",
    "exa_hess_structure_host", "exa_jac_structure64_host", "exa_hess_structure64_host", "exa_time_callback", "exa_sync", "exa_block_order",
    "exa_eval_fused", "exa_eval_all", "exa_set_product_mode", "exa_get_product_mode", "exa_set_grad_mode", "exa_get_grad_mode", "exa_set_deterministic", "exa_compress", "exa_compress_info", "exa_cnnzj64", "exa_cnnzh64", "exa_cjac_structure", "exa_chess_structure", "exa_cjac_structure64",
    "exa_chess_structure64", "exa_cjac_csc", "exa_chess_csc", "exa_cjac", "exa_chess",
    "exa_build_info", "exa_build_audit", "exa_debug_dump_window_launch", "exa_tune", "exa_comm_unique_id", "exa_comm_init", "exa_comm_attach", "exa_comm_hook", "exa_comm_free",
    "exa_comm_info", "exa_set_reduce", "exa_allreduce", "exa_set_coo_local", "exa_local_nnzj64", "exa_local_nnzh64",
    "exa_coo_slices", "exa_shard_var_range", "exa_hess_variant", "exa_hess_throttle", "exa_product_info", "exa_allgather_coo", "exa_shard_layout", "exa_collective_plan",
]
# ... and include/exahip_recipe.h
RECIPE_SYMBOLS = [
    "exa_recipe_load", "exa_recipe_load_trusted", "exa_recipe_trust_code", "exa_recipe_free", "exa_recipe_nargs", "exa_recipe_argtype", "exa_recipe_schema", "exa_recipe_new",
    "exa_recipe_plan", "exa_data_begin", "exa_data_free", "exa_set_scalar_i64", "exa_set_scalar_f64", "exa_set_array_i64",
    "exa_set_array_f64", "exa_set_col_i64", "exa_set_col_f64", "exa_data_ready", "exa_new_from_data", "exa_plan_from_data",
    "exa_nblocks", "exa_block_name", "exa_block", "exa_get_value_block", "exa_set_value_block", "exa_get_value", "exa_describe",
]


def build(force=False):
    """Compile libexahip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cpp", ".hpp")) or f == "Makefile"]
    srcs += [os.path.join(CSRC, "..", "..", "include", f) for f in ("exahip.h", "exahip_ir.h", "exahip_recipe.h")]
    newest = max(os.path.getmtime(s) for s in srcs)
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < newest:
        subprocess.check_call(["make", "-C", CSRC, "-s"])
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C {CSRC}` (or __graft_entry__.build()). "
            "exahip has no CPU/Python fallback.")
    # One HIP runtime per process: when torch is installed its bundled libamdhip64.so.7 must be the copy that
    # libexahip.so binds to (same SONAME), otherwise device pointers/streams from torch tensors would belong to a
    # different runtime instance.  A host without torch (e.g. the Julia shim) just uses the system ROCm.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
    L.exa_last_error.restype = ctypes.c_char_p
    L.exa_kernel_source.restype = ctypes.c_char_p
    L.exa_kernel_source.argtypes = [i32]
    L.exa_module_source.restype = ctypes.c_char_p
    L.exa_module_source.argtypes = [i32, i32]
    L.exa_code_object_path.restype = ctypes.c_char_p
    L.exa_code_object_path.argtypes = [i32]
    L.exa_new_from_table.argtypes = [vp, vp]
    L.exa_plan_only.argtypes = [vp, vp]
    L.exa_compile.argtypes = [i32]
    L.exa_module_name.restype = ctypes.c_char_p
    L.exa_module_name.argtypes = [i32]
    L.exa_cache_add.argtypes = [ctypes.c_char_p, vp, ctypes.c_size_t]
    L.exa_cache_note.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    L.exa_module_alias.restype = ctypes.c_char_p
    L.exa_module_alias.argtypes = [i32]
    L.exa_module_alias_note.restype = ctypes.c_char_p
    L.exa_module_alias_note.argtypes = [i32]
    L.exa_code_object_count.argtypes = [i32]
    L.exa_code_object.argtypes = [i32, i32, ctypes.c_char_p, i32, ctypes.c_char_p, i32]
    L.exa_free.argtypes = [i32]
    for f in ("exa_nvar", "exa_ncon", "exa_nnzj", "exa_nnzh", "exa_npatterns"):
        getattr(L, f).argtypes = [i32]
    for f in ("exa_nvar64", "exa_ncon64", "exa_nnzj64", "exa_nnzh64", "exa_nnzg64"):
        getattr(L, f).argtypes = [i32]
        getattr(L, f).restype = i64
    L.exa_pattern_info.argtypes = [i32, i32, vp]
    L.exa_pattern_comp.argtypes = [i32, i32, i32, vp]
    L.exa_meta.argtypes = [i32, vp, vp, vp, vp, vp]
    L.exa_set_stream.argtypes = [i32, vp]
    L.exa_set_shard.argtypes = [i32, i32, i32]
    L.exa_set_value.argtypes = [i32, i64, vp, i64]
    for f in ("exa_obj", "exa_obj_async", "exa_grad", "exa_cons", "exa_jac", "exa_obj_host", "exa_grad_host",
              "exa_cons_host", "exa_jac_host"):
        getattr(L, f).argtypes = [i32, vp, vp]
    for f in ("exa_jprod", "exa_jtprod", "exa_jprod_host", "exa_jtprod_host"):
        getattr(L, f).argtypes = [i32, vp, vp, vp]
    L.exa_hprod.argtypes = [i32, vp, vp, vp, dbl, vp]
    L.exa_hprod_host.argtypes = [i32, vp, vp, vp, dbl, vp]
    L.exa_hess.argtypes = [i32, vp, vp, dbl, vp]
    L.exa_hess_host.argtypes = [i32, vp, vp, dbl, vp]
    for f in ("exa_jac_structure", "exa_hess_structure", "exa_jac_structure64", "exa_hess_structure64",
              "exa_jac_structure_host", "exa_hess_structure_host", "exa_jac_structure64_host",
              "exa_hess_structure64_host"):
        getattr(L, f).argtypes = [i32, vp, vp]
    L.exa_time_callback.argtypes = [i32, i32, i32, vp, vp, dbl, vp, vp]
    L.exa_sync.argtypes = [i32]
    L.exa_block_order.argtypes = [i32, i32]
    L.exa_eval_fused.argtypes = [i32, vp, vp, dbl, vp, vp, vp, vp]
    L.exa_eval_all.argtypes = [i32, vp, vp, dbl, vp, vp, vp, vp, vp]
    L.exa_set_product_mode.argtypes = [i32, i32, i32]
    L.exa_get_product_mode.argtypes = [i32, vp, vp]
    L.exa_set_grad_mode.argtypes = [i32, i32]
    L.exa_get_grad_mode.argtypes = [i32, vp]
    L.exa_set_deterministic.argtypes = [i32, i32]
    L.exa_compress.argtypes = [i32]
    for f in ("exa_cnnzj64", "exa_cnnzh64"):
        getattr(L, f).argtypes = [i32]
        getattr(L, f).restype = i64
    for f in ("exa_cjac_structure", "exa_chess_structure", "exa_cjac_structure64", "exa_chess_structure64", "exa_cjac", "exa_cjac_csc", "exa_chess_csc"):
        getattr(L, f).argtypes = [i32, vp, vp]
    L.exa_chess.argtypes = [i32, vp, vp, dbl, vp]
    L.exa_compress_info.argtypes = [i32, i32, ctypes.c_char_p, i32, vp]
    L.exa_build_info.argtypes = [i32, ctypes.c_char_p, i32, vp]
    L.exa_build_audit.argtypes = [i32, ctypes.c_char_p, i32]
    L.exa_locality_order.argtypes = [i32, i32, vp]
    L.exa_eval_all_mode.argtypes = [i32]
    L.exa_collective_plan.argtypes = [i32, i32, vp, i32]
    L.exa_comm_complete.argtypes = [i32, i32, vp]
    L.exa_set_value_dev.argtypes = [i32, ctypes.c_int64, vp, ctypes.c_int64]
    L.exa_register_univariate.argtypes = [ctypes.c_char_p] * 5
    L.exa_register_bivariate.argtypes = [ctypes.c_char_p] * 8
    L.exa_register_univariate_fused.argtypes = [ctypes.c_char_p] * 3
    L.exa_user_function.argtypes = [i32, i32, i32, vp, i32]
    L.exa_theta_ptr.restype = ctypes.c_void_p
    L.exa_theta_ptr.argtypes = [i32]
    L.exa_debug_dump_window_launch.argtypes = [i32, i32, vp, vp, vp, ctypes.c_double, ctypes.c_char_p]
    L.exa_tune.argtypes = [i32, i32, vp, vp]
    L.exa_comm_unique_id.argtypes = [vp]
    L.exa_comm_init.argtypes = [i32, i32, i32, vp]
    L.exa_comm_attach.argtypes = [i32, vp]
    L.exa_comm_hook.argtypes = [i32, i32, i32, vp, vp]
    L.exa_comm_free.argtypes = [i32]
    L.exa_comm_info.argtypes = [i32, vp, vp, vp]
    L.exa_set_reduce.argtypes = [i32, i32]
    L.exa_allreduce.argtypes = [i32, vp, i64]
    L.exa_set_coo_local.argtypes = [i32, i32]
    for f in ("exa_local_nnzj64", "exa_local_nnzh64"):
        getattr(L, f).argtypes = [i32]
        getattr(L, f).restype = i64
    L.exa_coo_slices.argtypes = [i32, i32, vp]
    L.exa_shard_var_range.argtypes = [i32, vp, vp]
    L.exa_hess_variant.argtypes = [i32]
    L.exa_hess_throttle.argtypes = [i32]
    L.exa_product_info.argtypes = [i32, i32, ctypes.c_char_p, i32]
    L.exa_allgather_coo.argtypes = [i32, i32, vp, vp]
    L.exa_shard_layout.argtypes = [i32, i32]
    # include/exahip_recipe.h
    cp, sz = ctypes.c_char_p, ctypes.c_size_t
    L.exa_recipe_load.argtypes = [vp, sz]
    L.exa_recipe_load_trusted.argtypes = [vp, sz]
    L.exa_recipe_trust_code.argtypes = [i32]
    for f in ("exa_recipe_free", "exa_recipe_nargs", "exa_data_begin", "exa_data_free", "exa_data_ready", "exa_new_from_data",
              "exa_plan_from_data", "exa_nblocks"):
        getattr(L, f).argtypes = [i32]
    for f in ("exa_recipe_argtype", "exa_recipe_schema"):
        getattr(L, f).argtypes = [i32, vp, i32]
    L.exa_recipe_new.argtypes = [i32, i32]
    L.exa_recipe_plan.argtypes = [i32, i32]
    L.exa_set_scalar_i64.argtypes = [i32, cp, i64]
    L.exa_set_scalar_f64.argtypes = [i32, cp, dbl]
    L.exa_set_array_i64.argtypes = [i32, cp, vp, i32]
    L.exa_set_array_f64.argtypes = [i32, cp, vp, i32]
    L.exa_set_col_i64.argtypes = [i32, cp, cp, vp, i32]
    L.exa_set_col_f64.argtypes = [i32, cp, cp, vp, i32]
    L.exa_block_name.argtypes = [i32, i32, vp, i32]
    L.exa_block.argtypes = [i32, i32, vp]
    L.exa_get_value_block.argtypes = [i32, i32, vp, i32]
    L.exa_set_value_block.argtypes = [i32, i32, vp, i32]
    L.exa_get_value.argtypes = [i32, i64, vp, i64]
    L.exa_describe.argtypes = [i32, vp]
    _LIB = L
    return L


class ExaHipError(RuntimeError):
    pass


def check(status, what):
    if status == 0:
        return
    msg = lib().exa_last_error().decode(errors="replace")
    if status != 2:
        msg = "bad id or argument" + (": " + msg if msg else "")
    raise ExaHipError(f"{what}: status {status}: {msg}")
