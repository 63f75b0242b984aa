"""ctypes loader for the in-tree C-ABI library (include/smcp_amd.h).

There is deliberately no fallback: if ``libsmcp_amd.so`` is missing the import fails loudly
(build it with ``python -m smcp_amd.build`` / ``__graft_entry__.build()``).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsmcp_amd.so")

c_i64 = ctypes.c_int64
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_dblp = ctypes.POINTER(ctypes.c_double)
c_vp = ctypes.c_void_p

# name -> (restype, argtypes); every symbol include/smcp_amd.h declares
SIGNATURES = {
    "csp_symbolic_create": (c_vp, [c_i64, c_vp, c_vp, c_vp, c_i64p]),
    "csp_symbolic_destroy": (None, [c_vp]),
    "csp_symbolic_replicate": (c_vp, [c_vp, c_i64, c_i64p]),
    "csp_trial_flags": (ctypes.c_int, [c_vp, c_i64, c_vp]),
    "csp_lazy_status": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "csp_status": (ctypes.c_int, [c_vp, c_vp]),
    "csp_symbolic_query": (c_i64, [c_vp, ctypes.c_int, c_vp]),
    "csp_maxcardsearch": (ctypes.c_int, [c_i64, c_vp, c_vp, c_vp]),
    "csp_mindegree": (ctypes.c_int, [c_i64, c_vp, c_vp, c_vp]),
    "csp_index_map": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp]),
    "csp_device_init": (ctypes.c_int, [c_vp, ctypes.c_int, c_i64]),
    "csp_device_bytes": (c_i64, [c_vp]),
    "csp_cholesky": (ctypes.c_int, [c_vp, c_vp, c_vp]),
    "csp_llt": (ctypes.c_int, [c_vp, c_vp, c_vp]),
    "csp_projected_inverse": (ctypes.c_int, [c_vp, c_vp, c_vp]),
    "csp_cholesky_projected_inverse": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_vp]),
    "csp_completion": (ctypes.c_int, [c_vp, c_vp, c_vp]),
    "csp_hessian": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, ctypes.c_int, ctypes.c_int, c_vp]),
    "csp_trsm": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, ctypes.c_int, c_vp]),
    "csp_dot": (ctypes.c_int, [c_vp, c_vp, c_vp, c_dblp, c_vp]),
    "csp_logdiagsum": (ctypes.c_int, [c_vp, c_vp, c_dblp, c_vp]),
    "csp_axpby": (ctypes.c_int, [c_i64, ctypes.c_double, c_vp, ctypes.c_double, c_vp, c_vp]),
    "kkt_set_constraints": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp]),
    "kkt_amap": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "kkt_aadj": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "kkt_schur_factor": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "kkt_schur_forget": (ctypes.c_int, [c_vp, c_vp]),
    "kkt_schur_columns": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp]),
    "kkt_schur_gram_part": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp]),
    "kkt_constraint_classes": (ctypes.c_int, [c_vp, c_vp]),
    "csp_cache_reset": (ctypes.c_int, [c_vp]),
    "csp_touch": (ctypes.c_int, [c_vp, c_vp]),
    "csp_tune": (ctypes.c_int, [c_vp, ctypes.c_int, c_i64]),
    "csp_tune_report": (ctypes.c_int, [c_vp, c_vp]),
    "csp_profile_enable": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "csp_profile_kinds": (c_i64, []),
    "csp_profile_filter": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "csp_profile_read": (c_i64, [c_vp, c_vp, c_vp]),
    "csp_profile_kernel_name": (ctypes.c_char_p, [ctypes.c_int]),
    "csp_set_partition": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int]),
    "kkt_gram_prepare": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "kkt_gram_sweep": (ctypes.c_int, [c_vp, ctypes.c_int, c_i64, c_i64, c_vp]),
    "kkt_gram_accumulate": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp]),
    "csp_exchange_sizes": (ctypes.c_int, [c_vp, c_i64, c_vp]),
    "csp_exchange_pack": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp]),
    "csp_exchange_unpack": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp]),
    "csp_exchange_pack_range": (ctypes.c_int, [c_vp, c_i64, c_i64, c_vp, c_vp]),
    "csp_exchange_unpack_all": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp]),
    "kkt_stack_rows": (ctypes.c_int, [c_vp, ctypes.c_int, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp]),
    "csp_exchange_combine": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, ctypes.c_int, c_vp]),
    "csp_cholesky_part": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, c_vp]),
    "csp_projected_inverse_part": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, c_vp]),
    "kkt_prepare_part": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, ctypes.c_int, c_vp]),
    "kkt_gram_prepare_part": (ctypes.c_int, [c_vp, c_vp]),
    "csp_hessian_sweep_part": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, ctypes.c_int, ctypes.c_int, c_vp]),
    "kkt_set_tnzcols": (ctypes.c_int, [c_vp, ctypes.c_double]),
    "dense_potrf": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_vp]),
    "dense_potrs": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp]),
    "kkt_solve": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, ctypes.c_double, c_vp, c_vp, c_vp]),
    "kkt_qr_factor": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "kkt_qr_solve": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_double, c_vp, c_vp, c_vp]),
    "kkt_qr_inspect": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "smcp_amd native library not built: %s is missing. Run `python -m smcp_amd.build` "
                "(there is no CPU fallback by design)." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
