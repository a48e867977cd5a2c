"""ctypes binding of ``libqprop_hip.so`` (C ABI in ``include/qprop.h``).

This is the only gateway from the Python host mirror to the engine.  There is no
CPU fallback: if the shared library is missing, or no MI355X is visible when a context
is created, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QPROP_HIP_LIB") or os.path.join(_HERE, "lib", "libqprop_hip.so")   # (override: A/B of two builds)

QP_OK = 0
QP_E_INTERNAL = 10
QP_E_RCCL = 12
STATUS = {
    0: "QP_OK", 1: "QP_E_BAD_ARG", 2: "QP_E_HIP", 3: "QP_E_DT_MISMATCH",
    4: "QP_E_TOO_FEW_COEFFS", 5: "QP_E_NORMALIZATION", 6: "QP_E_MAX_RESTARTS",
    7: "QP_E_DIVDIFF_UNDERFLOW", 8: "QP_E_NO_DEVICE", 9: "QP_E_ALLOC",
    10: "QP_E_INTERNAL", 11: "QP_E_M_MAX", 12: "QP_E_RCCL",
}
LAYOUT_CSR, LAYOUT_CSC = 0, 1
VAL_C128, VAL_F64 = 0, 1
FMT_AUTO, FMT_CSR, FMT_RBCSR, FMT_HRB, FMT_MATFREE, FMT_DENSE = 0, 1, 2, 3, 4, 5
WALK_REASONS = {0: "ok", 1: "not_hermitian", 2: "complex_coefficient", 3: "not_packed", 4: "too_few_blocks", 5: "no_uniform_run",
                6: "row_length", 7: "not_mirrored", 8: "no_far_distance", 9: "no_near_distance", 10: "near_too_far", 11: "too_many_near",
                12: "too_many_far", 13: "incommensurate_strides", 14: "no_kernel_instance", 15: "layout", 16: "disabled"}


class QPPerformanceWarning(UserWarning):
    """An operator left a fast path (e.g. the strip walk of the fused Chebyshev term); results are unaffected."""
FUNC_EXPMI, FUNC_EXP, FUNC_CALLBACK = 0, 1, 2


class QPError(RuntimeError):
    """A non-zero status from the C ABI.  ``status`` holds the code; subclasses mirror
    the exception types the reference raises for the same condition."""

    def __init__(self, status, message):
        super().__init__(f"{STATUS.get(status, status)}: {message}")
        self.status = status


class QPAssertionError(QPError, AssertionError):
    """Reference ``@assert`` failures (src/cheby.jl:157,165,196; src/newton.jl:209,375)."""


class QPArgumentError(QPError, ValueError):
    """Reference ``ArgumentError`` / ``error(...)`` on bad arguments."""


class qp_c128(C.Structure):
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


class qp_stats(C.Structure):
    _fields_ = [("n_matvec", C.c_uint64), ("n_cheby_steps", C.c_uint64),
                ("n_newton_steps", C.c_uint64), ("n_restarts", C.c_uint64),
                ("n_kernel_launches", C.c_uint64), ("spmv_bytes", C.c_double),
                ("n_graph_launches", C.c_uint64)]


class qp_newton_stats(C.Structure):
    _fields_ = [("restarts", C.c_int), ("n_a", C.c_int), ("n_leja", C.c_int),
                ("m_last", C.c_int), ("n_matvec", C.c_int), ("radius", C.c_double),
                ("last_relerr", C.c_double), ("norm_psi", C.c_double),
                ("ms_arnoldi", C.c_double), ("ms_eig", C.c_double), ("ms_leja", C.c_double),
                ("ms_coeffs", C.c_double), ("ms_poly", C.c_double), ("ms_update", C.c_double),
                ("ms_exposed", C.c_double), ("sweeps_onepass", C.c_int), ("sweeps_onepass_redone", C.c_int)]


FUNC_CB = C.CFUNCTYPE(None, C.POINTER(qp_c128), C.POINTER(qp_c128), C.c_void_p)


class qp_acc_defer(C.Structure):
    _fields_ = [("skip", C.c_int), ("n_defer", C.c_int), ("a_d1", C.c_double), ("a_d2", C.c_double)]


class qp_sharded_cheby_desc(C.Structure):
    _fields_ = [("op", C.c_void_p), ("split", C.c_void_p), ("comm", C.c_void_p), ("X0", C.c_void_p),
                ("X1", C.c_void_p), ("acc", C.c_void_p), ("slab", C.c_void_p), ("send_rows", C.POINTER(C.c_int64)),
                ("nsend", C.c_int64), ("M", C.c_int64), ("direct_send", C.c_int),
                ("send_to", C.POINTER(C.c_int)), ("n_send_to", C.c_int),
                ("recv_from", C.POINTER(C.c_int)), ("n_recv_from", C.c_int)]


class qp_pauli_string(C.Structure):
    _fields_ = [("xmask", C.c_uint64), ("zmask", C.c_uint64), ("coef", qp_c128), ("op", C.c_int)]


class qp_prop_spec(C.Structure):
    _fields_ = [("method", C.c_int), ("cheby", C.c_void_p), ("a", C.POINTER(C.c_double)), ("n_coeffs", C.c_int),
                ("Delta", C.c_double), ("E_min", C.c_double), ("wrk_dt", C.c_double), ("limit", C.c_double),
                ("check_normalization", C.c_int), ("newton", C.c_void_p), ("func_id", C.c_int), ("cb", FUNC_CB),
                ("user", C.c_void_p), ("norm_min", C.c_double), ("relerr", C.c_double), ("max_restarts", C.c_int)]

_P = C.c_void_p
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_dp = C.POINTER(C.c_double)
_cp = C.POINTER(qp_c128)

# every symbol include/qprop.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "qp_last_error": (C.c_char_p, []),
    "qp_status_name": (C.c_char_p, [C.c_int]),
    "qp_version": (C.c_int, []),
    "qp_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "qp_tuning_set": (C.c_int, [C.c_char_p, C.c_int]),
    "qp_ctx_tuning_set": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "qp_ctx_tuning_get": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int)]),
    "qp_ctx_create": (C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    "qp_ctx_destroy": (C.c_int, [_P]),
    "qp_sync": (C.c_int, [_P]),
    "qp_stats_get": (C.c_int, [_P, C.POINTER(qp_stats)]),
    "qp_stats_reset": (C.c_int, [_P]),
    "qp_timer_begin": (C.c_int, [_P]),
    "qp_timer_end": (C.c_int, [_P, _dp]),
    "qp_csc_to_csr_host": (C.c_int, [C.c_int64, C.c_int64, _i64p, _i64p, _cp, C.c_int, _i64p, _i32p, _cp]),
    "qp_partition_rows_host": (C.c_int, [_i64p, C.c_int64, C.c_int, C.c_int, _i64p]),
    "qp_matrix_create": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int64, _i64p, _i64p, _P, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "qp_matrix_destroy": (C.c_int, [_P]),
    "qp_matrix_info": (C.c_int, [_P, _i64p, _i64p, _i64p, C.POINTER(C.c_int), _i64p]),
    "qp_matrix_get_csr": (C.c_int, [_P, _i64p, _i32p, _cp]),
    "qp_operator_create": (C.c_int, [_P, C.POINTER(_P), C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "qp_operator_set_coeffs": (C.c_int, [_P, _cp, C.c_int]),
    "qp_operator_set_scale": (C.c_int, [_P, qp_c128]),
    "qp_operator_destroy": (C.c_int, [_P]),
    "qp_operator_info": (C.c_int, [_P, _i64p, _i64p, _i64p, C.POINTER(C.c_int)]),
    "qp_operator_get_csr": (C.c_int, [_P, _i64p, _i32p, _cp]),
    "qp_operator_layout_info": (C.c_int, [_P, _i64p]),
    "qp_operator_build_info": (C.c_int, [_P, _dp]),
    "qp_operator_walk_info": (C.c_int, [_P, _i64p]),
    "qp_operator_walk2_info": (C.c_int, [_P, _i64p]),
    "qp_operator_fill_info": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "qp_developer_build": (C.c_int, []),
    "qp_operator_walk_long": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "qp_operator_walk_reason": (C.c_int, [_P, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "qp_operator_walk_long_pairs": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "qp_operator_walk_shape": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "qp_operator_encoding_info": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "qp_operator_value_encoding_info": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "qp_operator_evaluate_info": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "qp_operator_colblock_info": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "qp_lattice_fill_host": (C.c_int, [C.c_int64, C.c_int64, _i64p, _i32p, C.c_int, _i64p, _i32p, C.c_int64, C.POINTER(C.c_int64)]),
    "qp_operator_spmm_walk": (C.c_int, [_P, C.c_int, _i64p]),
    "qp_operator_spmm_tiles": (C.c_int, [_P, C.c_int, _i64p]),
    "qp_state_create": (C.c_int, [_P, C.c_int64, C.POINTER(_P)]),
    "qp_state_wrap": (C.c_int, [_P, _P, C.c_int64, C.POINTER(_P)]),
    "qp_state_destroy": (C.c_int, [_P]),
    "qp_state_upload": (C.c_int, [_P, _cp]),
    "qp_state_download": (C.c_int, [_P, _cp]),
    "qp_host_register": (C.c_int, [_P, C.c_size_t]),
    "qp_host_unregister": (C.c_int, [_P]),
    "qp_state_ptr": (_P, [_P]),
    "qp_state_len": (C.c_int64, [_P]),
    "qp_copy": (C.c_int, [_P, _P]),
    "qp_scal": (C.c_int, [_P, qp_c128]),
    "qp_axpy": (C.c_int, [qp_c128, _P, _P]),
    "qp_fill": (C.c_int, [_P, qp_c128]),
    "qp_dot": (C.c_int, [_P, _P, _cp]),
    "qp_norm": (C.c_int, [_P, _dp]),
    "qp_mul": (C.c_int, [_P, _P, _P, qp_c128, qp_c128]),
    "qp_dot_op": (C.c_int, [_P, _P, _P, _P, _cp]),
    "qp_cheby_coeffs": (C.c_int, [C.c_double, C.c_double, C.c_double, _dp, C.c_int, C.POINTER(C.c_int)]),
    "qp_cheby_create": (C.c_int, [_P, C.c_int64, C.POINTER(_P)]),
    "qp_cheby_destroy": (C.c_int, [_P]),
    "qp_cheby_step": (C.c_int, [_P, _P, _P, _dp, C.c_int, C.c_double, C.c_double, C.c_double,
                                C.c_double, C.c_double, C.c_int]),
    "qp_cheby_step_batched": (C.c_int, [_P, _P, _P, C.c_int, _dp, C.c_int, C.c_double, C.c_double, C.c_double,
                                        C.c_double]),
    "qp_cheby_term": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P, _P, qp_c128, C.c_double, C.c_double,
                                C.c_double, qp_c128, C.POINTER(qp_acc_defer)]),
    "qp_acc_schedule_host": (C.c_int, [_dp, C.c_int, C.POINTER(qp_acc_defer)]),
    "qp_split_create": (C.c_int, [_P, _i64p, C.c_int64, C.POINTER(_P)]),
    "qp_split_destroy": (C.c_int, [_P]),
    "qp_split_info": (C.c_int, [_P, _i64p, _i64p]),
    "qp_split_walk_info": (C.c_int, [_P, _i64p]),
    "qp_split_check": (C.c_int, [_P]),
    "qp_cheby_term_split": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_int64, _P, _P, _P, _P, _P, qp_c128, C.c_double,
                                      C.c_double, C.c_double, qp_c128, C.POINTER(qp_acc_defer)]),
    "qp_liouvillian_create": (C.c_int, [_P, C.c_int64, C.POINTER(_cp), C.c_int, C.c_int, C.POINTER(_cp), C.c_int,
                                        C.c_int, C.POINTER(_P)]),
    "qp_pauli_operator_create": (C.c_int, [_P, C.c_int, C.POINTER(qp_pauli_string), C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "qp_comm_unique_id": (C.c_int, [C.c_char_p, C.c_char_p]),
    "qp_comm_create": (C.c_int, [_P, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(_P)]),
    "qp_comm_prepare": (C.c_int, [_P, C.c_char_p, C.c_int, C.c_int, C.POINTER(_P)]),
    "qp_comm_connect": (C.c_int, [_P, C.c_char_p]),
    "qp_comm_create_callback": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.POINTER(_P)]),
    "qp_comm_info": (C.c_int, [_P, _i32p, _i32p, _i32p, _i32p, _i32p]),
    "qp_comm_destroy": (C.c_int, [_P]),
    "qp_comm_allgather": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "qp_sharded_cheby_create": (C.c_int, [C.POINTER(qp_sharded_cheby_desc), C.POINTER(_P)]),
    "qp_sharded_cheby_destroy": (C.c_int, [_P]),
    "qp_sharded_cheby_step": (C.c_int, [_P, _dp, C.c_int, C.c_double, C.c_double, C.c_double]),
    "qp_krylov_create": (C.c_int, [_P, C.c_int64, C.c_int, C.POINTER(_P)]),
    "qp_krylov_destroy": (C.c_int, [_P]),
    "qp_krylov_download": (C.c_int, [_P, C.c_int, _cp]),
    "qp_krylov_vec": (C.c_int, [_P, C.c_int, C.POINTER(_P)]),
    "qp_krylov_multidot": (C.c_int, [_P, C.c_int, _P]),
    "qp_krylov_project": (C.c_int, [_P, C.c_int, C.c_double, _P, _P, _P]),
    "qp_krylov_normalize": (C.c_int, [_P, C.c_int, C.c_double, C.c_double, _P, _P]),
    "qp_combine": (C.c_int, [_P, C.c_int, qp_c128, _P, C.c_int, C.c_int, _cp, _P]),
    "qp_arnoldi": (C.c_int, [_P, _P, C.c_int, _P, C.c_double, C.c_int, C.c_double, _cp, C.c_int,
                             C.POINTER(C.c_int)]),
    "qp_arnoldi_extend": (C.c_int, [_P, _P, C.c_int, C.c_double, C.c_double, _cp, C.c_int,
                                    C.POINTER(C.c_int)]),
    "qp_hessenberg_eigvals": (C.c_int, [_cp, C.c_int, C.c_int, C.c_int, _cp]),
    "qp_extend_leja": (C.c_int, [_cp, C.c_int, _cp, C.c_int, C.c_int]),
    "qp_extend_newton_coeffs": (C.c_int, [_cp, C.c_int, _cp, C.c_int, FUNC_CB, _P, C.c_int, C.c_double]),
    "qp_newton_create": (C.c_int, [_P, C.c_int64, C.c_int, C.POINTER(_P)]),
    "qp_newton_destroy": (C.c_int, [_P]),
    "qp_newton_step": (C.c_int, [_P, _P, _P, C.c_double, C.c_int, FUNC_CB, _P, C.c_double, C.c_double,
                                 C.c_int, C.POINTER(qp_newton_stats)]),
    "qp_newton_get_coeffs": (C.c_int, [_P, _cp, _cp, C.c_int]),
    "qp_propagate": (C.c_int, [_P, _P, C.POINTER(qp_prop_spec), _dp, _cp, C.c_int, C.c_int, C.POINTER(_P), C.c_int,
                               _cp, _cp]),
    "qp_ritzvals": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_double, C.c_double, _cp, C.POINTER(C.c_int)]),
    "qp_specrange_arnoldi": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, _dp, _dp]),
}

_lib = None


def load():
    """Load the shared library once.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C quantumpropagators.jl_amd/csrc`).  There is no CPU fallback.")
    try:  # share torch's HIP runtime when torch is used in this process
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the library itself
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if os.environ.get("QPROP_HIP_LIB"):      # an older build loaded for an A/B run: newer entry points are absent
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status == QP_OK:
        return
    msg = load().qp_last_error().decode("utf-8", "replace")
    if status in (3, 4, 5, 6, 7):
        raise QPAssertionError(status, msg)
    if status in (1, 11):
        raise QPArgumentError(status, msg)
    raise QPError(status, msg)


def c128(z):
    z = complex(z)
    return qp_c128(z.real, z.imag)


def _as_c128(a):
    a = np.ascontiguousarray(a, dtype=np.complex128)
    return a, a.ctypes.data_as(_cp)


def _ptr(a, typ):
    return a.ctypes.data_as(typ)


_live_contexts = weakref.WeakSet()


def tuning_set(key, value, ctx=None):
    """Developer knob.  With ``ctx``: that context only (``qp_ctx_tuning_set``).  Without: the default
    for contexts created later (``qp_tuning_set``) AND every live ``Context`` of this Python process,
    one by one -- a convenience for single-threaded A/B scripts; the library itself keeps no
    process-global knob state."""
    if ctx is not None:
        ctx.tuning_set(key, value)
        return
    check(load().qp_tuning_set(key.encode(), int(value)))
    for c in list(_live_contexts):
        if c._h:
            c.tuning_set(key, value)


def device_count():
    n = C.c_int(0)
    check(load().qp_device_count(C.byref(n)))
    return n.value


# ------------------------------------------------------------------------------
# host-only helpers (run anywhere the library loads)
# ------------------------------------------------------------------------------

def cheby_coeffs(Delta, dt, limit=1e-12):
    lib = load()
    n = C.c_int(0)
    cap = 64
    while True:
        out = np.empty(cap, dtype=np.float64)
        st = lib.qp_cheby_coeffs(Delta, dt, limit, _ptr(out, _dp), cap, C.byref(n))
        if st == QP_OK:
            return out[: n.value].copy()
        if n.value > cap:
            cap = n.value
            continue
        check(st)


def hessenberg_eigvals(Hess, m, accumulate=False):
    Hess = np.asfortranarray(Hess, dtype=np.complex128)
    ldh = Hess.shape[0]
    n_out = m * (m + 1) // 2 if accumulate else m
    out = np.empty(n_out, dtype=np.complex128)
    check(load().qp_hessenberg_eigvals(_ptr(Hess, _cp), ldh, m, int(accumulate), _ptr(out, _cp)))
    return out


def lattice_fill_host(nrows, ncols, rowptr, col, min_blocks=3072):
    """The lattice completion of operator creation on a host CSR pattern (no device needed): returns (rowptr, col)."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    # the completion adds the missing entries of the rows (up to 12 %, engine_plans.hip: lattice_fill) plus the transposes that land
    # in the first and last `reach` rows -- bounded by the pattern, not by a constant: ask for the size first (a call with no room
    # reports the count it needs), then fill
    rp_out = np.empty(nrows + 1, dtype=np.int64)
    nnz = C.c_int64(0)
    cap = int(1.13 * int(rowptr[-1])) + 64 * 20 + 1024
    for _ in range(2):
        col_out = np.empty(cap, dtype=np.int32)
        rc = load().qp_lattice_fill_host(nrows, ncols, _ptr(rowptr, _i64p), _ptr(col, _i32p), int(min_blocks),
                                         _ptr(rp_out, _i64p), _ptr(col_out, _i32p), cap, C.byref(nnz))
        if rc == 0 or nnz.value <= cap:
            check(rc)
            break
        cap = int(nnz.value)      # (QP_E_BAD_ARG with *nnz_out = the entries the completed pattern has: retry with exactly that)
    return rp_out, col_out[:nnz.value].copy()


def extend_leja(leja, n, newpoints, n_use):
    """Zero-based; returns (leja, n + n_use).  ``newpoints`` is clobbered."""
    if len(leja) < n + n_use:
        new = np.zeros(2 * (n + n_use), dtype=np.complex128)
        new[:n] = leja[:n]
        leja = new
    assert leja.dtype == np.complex128 and newpoints.dtype == np.complex128
    check(load().qp_extend_leja(_ptr(leja, _cp), n, _ptr(newpoints, _cp), len(newpoints), n_use))
    return leja, n + n_use


def _func_args(func):
    """Map a Python callable / name to (func_id, callback, keepalive)."""
    if func is None or func == "expmi":
        return FUNC_EXPMI, FUNC_CB(0), None
    if func == "exp":
        return FUNC_EXP, FUNC_CB(0), None

    def tramp(zp, outp, _user):
        r = complex(func(complex(zp[0].re, zp[0].im)))
        outp[0].re = r.real
        outp[0].im = r.imag
    cb = FUNC_CB(tramp)
    return FUNC_CALLBACK, cb, cb


def extend_newton_coeffs(a, n_a, leja, func, n_leja, radius):
    if len(a) < n_leja:
        new = np.zeros(2 * n_leja, dtype=np.complex128)
        new[:n_a] = a[:n_a]
        a = new
    fid, cb, _keep = _func_args(func)
    leja = np.ascontiguousarray(leja, dtype=np.complex128)
    check(load().qp_extend_newton_coeffs(_ptr(a, _cp), n_a, _ptr(leja, _cp), fid, cb, None, n_leja, radius))
    return a, n_leja


def csc_to_csr(nrows, ncols, colptr, rowval, nzval, index_base=1):
    colptr = np.ascontiguousarray(colptr, dtype=np.int64)
    rowval = np.ascontiguousarray(rowval, dtype=np.int64)
    nzval = np.ascontiguousarray(nzval, dtype=np.complex128)
    nnz = int(colptr[-1]) - index_base
    rowptr = np.empty(nrows + 1, dtype=np.int64)
    col = np.empty(max(nnz, 1), dtype=np.int32)
    vals = np.empty(max(nnz, 1), dtype=np.complex128)
    check(load().qp_csc_to_csr_host(nrows, ncols, _ptr(colptr, _i64p), _ptr(rowval, _i64p), _ptr(nzval, _cp),
                                    index_base, _ptr(rowptr, _i64p), _ptr(col, _i32p), _ptr(vals, _cp)))
    return rowptr, col[:nnz], vals[:nnz]


def partition_rows(rowptr, nparts, balance="rows"):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    out = np.empty(nparts + 1, dtype=np.int64)
    check(load().qp_partition_rows_host(_ptr(rowptr, _i64p), len(rowptr) - 1, nparts,
                                        0 if balance == "rows" else 1, _ptr(out, _i64p)))
    return out


# ------------------------------------------------------------------------------
# handles
# ------------------------------------------------------------------------------

class Context:
    """Device + HIP stream.  ``stream=None``: the library creates its own stream;
    otherwise a raw hipStream_t (int), e.g. ``torch.cuda.current_stream().cuda_stream`` --
    which is 0 (HIP's null stream) unless the caller switched streams; 0 is passed on as
    QP_STREAM_NULL so that library kernels and the caller's copies share that stream."""

    def __init__(self, device=0, stream=None):
        self._h = _P()
        self.lib = load()
        if stream is None:
            sarg = None
        elif int(stream) == 0:
            sarg = _P(-1)            # QP_STREAM_NULL
        else:
            sarg = _P(int(stream))
        check(self.lib.qp_ctx_create(int(device), sarg, C.byref(self._h)))
        self.device = int(device)
        self._children = weakref.WeakSet()   # handles that must be destroyed before the context
        _live_contexts.add(self)

    def tuning_set(self, key, value):
        check(self.lib.qp_ctx_tuning_set(self._h, key.encode(), int(value)))

    def tuning_get(self, key):
        v = C.c_int(0)
        check(self.lib.qp_ctx_tuning_get(self._h, key.encode(), C.byref(v)))
        return v.value

    def _adopt(self, child):
        self._children.add(child)

    def sync(self):
        check(self.lib.qp_sync(self._h))

    def stats(self):
        s = qp_stats()
        check(self.lib.qp_stats_get(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in qp_stats._fields_}

    def reset_stats(self):
        check(self.lib.qp_stats_reset(self._h))

    def timer_begin(self):
        check(self.lib.qp_timer_begin(self._h))

    def timer_end(self):
        ms = C.c_double(0)
        check(self.lib.qp_timer_end(self._h, C.byref(ms)))
        return ms.value

    def close(self):
        if self._h:
            for child in list(self._children):   # native handles point at the native context
                child.close()
            self.lib.qp_ctx_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Matrix:
    """Canonical host CSR produced by the boundary's index work."""

    def __init__(self, ctx, nrows, ncols, ptr, idx, vals, layout=LAYOUT_CSR, index_base=0):
        self.ctx = ctx
        self.lib = ctx.lib
        ptr = np.ascontiguousarray(ptr, dtype=np.int64)
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        vals = np.asarray(vals)
        if np.iscomplexobj(vals):
            vals = np.ascontiguousarray(vals, dtype=np.complex128)
            dt = VAL_C128
        else:
            vals = np.ascontiguousarray(vals, dtype=np.float64)
            dt = VAL_F64
        nnz = int(ptr[-1]) - index_base
        self._h = _P()
        check(self.lib.qp_matrix_create(ctx._h, nrows, ncols, nnz, _ptr(ptr, _i64p), _ptr(idx, _i64p),
                                        vals.ctypes.data_as(_P), dt, layout, index_base, FMT_AUTO,
                                        C.byref(self._h)))
        self.nrows, self.ncols, self.nnz = nrows, ncols, nnz
        ctx._adopt(self)

    @classmethod
    def from_scipy(cls, ctx, A):
        import scipy.sparse as sp
        if sp.isspmatrix_csc(A):
            return cls(ctx, A.shape[0], A.shape[1], A.indptr, A.indices, A.data, layout=LAYOUT_CSC)
        A = sp.csr_matrix(A)
        return cls(ctx, A.shape[0], A.shape[1], A.indptr, A.indices, A.data, layout=LAYOUT_CSR)

    @classmethod
    def from_dense(cls, ctx, A):
        import scipy.sparse as sp
        return cls.from_scipy(ctx, sp.csr_matrix(np.asarray(A)))

    def get_csr(self):
        rowptr = np.empty(self.nrows + 1, dtype=np.int64)
        col = np.empty(max(self.nnz, 1), dtype=np.int32)
        vals = np.empty(max(self.nnz, 1), dtype=np.complex128)
        check(self.lib.qp_matrix_get_csr(self._h, _ptr(rowptr, _i64p), _ptr(col, _i32p), _ptr(vals, _cp)))
        return rowptr, col[: self.nnz], vals[: self.nnz]

    def close(self):
        if self._h:
            self.lib.qp_matrix_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Operator:
    """Device-resident lazy sum  sum_l c_l H_l  (Generators.Operator)."""

    def __init__(self, ctx, ops, ncoeffs=0, fmt=FMT_AUTO):
        self.ctx = ctx
        self.lib = ctx.lib
        self.ops = list(ops)
        arr = (_P * len(self.ops))(*[m._h for m in self.ops])
        self._h = _P()
        check(self.lib.qp_operator_create(ctx._h, arr, len(self.ops), int(ncoeffs), int(fmt), C.byref(self._h)))
        self.ncoeffs = int(ncoeffs)
        self._refresh_info()
        ctx._adopt(self)

    def _refresh_info(self):
        nr, nc, nnz, f = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
        check(self.lib.qp_operator_info(self._h, C.byref(nr), C.byref(nc), C.byref(nnz), C.byref(f)))
        self.nrows, self.ncols, self.nnz, self.format = nr.value, nc.value, nnz.value, f.value

    @property
    def shape(self):
        return (self.nrows, self.ncols)

    def layout_info(self):
        """Device layout: 64-row blocks, blocks with a stencil upper / lower column section,
        index bytes a mat-vec streams, stored values."""
        out = np.zeros(5, dtype=np.int64)
        check(self.lib.qp_operator_layout_info(self._h, _ptr(out, _i64p)))
        return dict(zip(("blocks", "stencil_upper_blocks", "stencil_lower_blocks", "index_bytes", "stored"),
                        (int(v) for v in out)))

    def build_info(self):
        """Host cost of the device layout: ms of the latest build, ms of all builds, re-layouts forced after
        creation (a complex coefficient on a Hermitian-packed operator), current device format."""
        out = np.zeros(4, dtype=np.float64)
        check(self.lib.qp_operator_build_info(self._h, _ptr(out, _dp)))
        return {"build_ms": float(out[0]), "build_ms_total": float(out[1]), "relayouts": int(out[2]), "format": int(out[3])}

    def fill_info(self):
        """Explicit zeros that operator creation added to complete a lattice operator's rows (0 for any other operator)."""
        n = C.c_int64(0)
        check(self.lib.qp_operator_fill_info(self._h, C.byref(n)))
        return n.value

    def walk2_info(self):
        """Does a whole-operator ``cheby!`` take the two-term strip walk (include/qprop.h: qp_operator_walk2_info)?"""
        out = np.zeros(8, dtype=np.int64)
        check(self.lib.qp_operator_walk2_info(self._h, _ptr(out, _i64p)))
        return dict(zip(("valid", "first_block", "end_block", "edge_blocks", "useful_rows_per_chunk", "chunks_per_strip_step",
                         "steps_per_wavefront", "segments"), (int(v) for v in out)))

    def walk_info(self):
        """Strip-walk plan of a Hermitian-packed lattice operator (see include/qprop.h)."""
        out = np.zeros(8, dtype=np.int64)
        check(self.lib.qp_operator_walk_info(self._h, _ptr(out, _i64p)))
        d = dict(zip(("valid", "near", "far", "diag", "rows_per_step", "first_block", "end_block", "edge_blocks"),
                     (int(v) for v in out)))
        gl = C.c_int64(0)
        check(self.lib.qp_operator_walk_long(self._h, C.byref(gl)))
        d["long_distance"] = gl.value        # rows; 0: no long pair (three-dimensional grids have one: the plane distance)
        lp = np.zeros(2, dtype=np.int64)
        check(self.lib.qp_operator_walk_long_pairs(self._h, _ptr(lp, _i64p)))
        d["long_distances"] = [int(v) for v in lp if v]      # [] / [L] / [L_0, L_1] (fourth-order stencils, four-dimensional grids)
        sh = np.zeros(8, dtype=np.int64)
        check(self.lib.qp_operator_walk_shape(self._h, _ptr(sh, _i64p)))
        d["far_diagonals"] = int(sh[4])      # 1: every far distance m g comes with m g - 1 and m g + 1 (nine-point stencils)
        d["upper_slots"] = int(sh[2] + sh[0] + sh[1] * (1 + 2 * sh[4]) + sh[3])   # value slots a walked row block streams
        return d

    def evaluate_info(self):
        """How evaluate! updates the values: first sparse control term (-1: none), positions rewritten, whether the latest
        update was a sparse one (include/qprop.h)."""
        out = np.zeros(3, dtype=np.int64)
        check(self.lib.qp_operator_evaluate_info(self._h, _ptr(out, _i64p)))
        return dict(zip(("first_sparse_term", "positions", "latest_update_sparse"), (int(v) for v in out)))

    def encoding_info(self):
        """Blocks per encoding of their column sections (include/qprop.h: qp_operator_encoding_info)."""
        out = np.zeros(8, dtype=np.int64)
        check(self.lib.qp_operator_encoding_info(self._h, _ptr(out, _i64p)))
        names = ("int32", "int16", "stencil", "block_map")
        return {"upper": dict(zip(names, (int(v) for v in out[:4]))), "lower": dict(zip(names, (int(v) for v in out[4:])))}

    def value_encoding_info(self):
        """Value-dictionary mirror: one byte + a shared table line per stored entry (include/qprop.h: qp_operator_value_encoding_info)."""
        out = np.zeros(6, dtype=np.int64)
        check(self.lib.qp_operator_value_encoding_info(self._h, _ptr(out, _i64p)))
        why = ("", "not a plain row-block operator", "a block with more than 256 distinct values", "no saving", "knob value_dict = 0",
               "column-blocked mirror in use")
        d = dict(zip(("valid", "table_entries", "tables", "coded_bytes", "plane_bytes"), (int(v) for v in out[:5])))
        d["reason"] = why[int(out[5])] if 0 <= int(out[5]) < len(why) else str(int(out[5]))
        return d

    def colblock_info(self):
        """Column-blocked mirror of an operator with irregular columns (include/qprop.h: qp_operator_colblock_info)."""
        out = np.zeros(6, dtype=np.int64)
        share = C.c_double(0.0)
        check(self.lib.qp_operator_colblock_info(self._h, _ptr(out, _i64p), C.byref(share)))
        d = dict(zip(("valid", "column_blocks", "log2_block_columns", "rows_per_tile", "longest_segment", "tiles"), (int(v) for v in out)))
        d["own_line_share"] = share.value
        return d

    def walk_reason(self):
        """(code, name, sentence): why the fused Chebyshev term of this operator does not take the strip walk -- QP_WALK_OK
        ("ok") when it does (include/qprop.h: qp_operator_walk_reason)."""
        code = C.c_int(0)
        buf = C.create_string_buffer(256)
        check(self.lib.qp_operator_walk_reason(self._h, C.byref(code), buf, len(buf)))
        return code.value, WALK_REASONS.get(code.value, str(code.value)), buf.value.decode()

    def spmm_walk(self, batch):
        """Row walk of the batched kernel for ``batch`` states: (inner dimension, strip width), (0, 0)
        for the natural row order."""
        out = np.zeros(2, dtype=np.int64)
        check(self.lib.qp_operator_spmm_walk(self._h, int(batch), _ptr(out, _i64p)))
        return int(out[0]), int(out[1])

    def spmm_tiles(self, batch):
        """LDS-staged tiles of the batched kernel for ``batch`` states (include/qprop.h: qp_operator_spmm_tiles)."""
        out = np.zeros(6, dtype=np.int64)
        check(self.lib.qp_operator_spmm_tiles(self._h, int(batch), _ptr(out, _i64p)))
        return dict(zip(("taken", "tiles", "rest_rows", "g", "K", "NN"), (int(v) for v in out)))

    def set_coeffs(self, coeffs):
        a, p = _as_c128(np.atleast_1d(coeffs))
        check(self.lib.qp_operator_set_coeffs(self._h, p, len(a)))
        self._refresh_info()      # (a complex coefficient takes a Hermitian-packed operator back to plain row blocks)

    def set_scale(self, s):
        check(self.lib.qp_operator_set_scale(self._h, c128(s)))
        self._refresh_info()

    def get_csr(self):
        if self.format == FMT_MATFREE:
            raise QPArgumentError(1, "a matrix-free operator has no stored entries")
        rowptr = np.empty(self.nrows + 1, dtype=np.int64)
        col = np.empty(max(self.nnz, 1), dtype=np.int32)
        vals = np.empty(max(self.nnz, 1), dtype=np.complex128)
        check(self.lib.qp_operator_get_csr(self._h, _ptr(rowptr, _i64p), _ptr(col, _i32p), _ptr(vals, _cp)))
        return rowptr, col[: self.nnz], vals[: self.nnz]

    def mul(self, x, y, alpha=1.0, beta=0.0):
        """``mul!(y, op, x, alpha, beta)``: y = beta y + alpha op x  (src/generators.jl:634-645)."""
        check(self.lib.qp_mul(self._h, x._h, y._h, c128(alpha), c128(beta)))
        return y

    def dot(self, x, y):
        """``dot(x, op, y)`` = <x| op |y>  (src/generators.jl:648-660)."""
        out = qp_c128()
        tmp = State(self.ctx, n=self.nrows)
        check(self.lib.qp_dot_op(x._h, self._h, y._h, tmp._h, C.byref(out)))
        return complex(out.re, out.im)

    def __mul__(self, state):
        """``op * state``: a new state."""
        if not isinstance(state, State):
            return NotImplemented
        return self.mul(state, State(self.ctx, n=self.nrows))

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def close(self):
        if self._h:
            self.lib.qp_operator_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


CONV_TDSE, CONV_LVN = 0, 1


def pauli_masks(string):
    """'XIZY' (qubit n-1 first, as one writes a tensor product) or a dict {qubit: 'X'|'Y'|'Z'} -> (xmask, zmask)."""
    if isinstance(string, dict):
        items = string.items()
    else:
        n = len(string)
        items = ((n - 1 - k, ch) for k, ch in enumerate(string))
    x = z = 0
    for q, ch in items:
        ch = ch.upper()
        if ch in "XY":
            x |= 1 << q
        if ch in "ZY":
            z |= 1 << q
        if ch not in "IXYZ":
            raise ValueError(f"not a Pauli label: {ch!r}")
    return x, z


class PauliOperator(Operator):
    """Qubit-register generator applied from its Pauli strings, no stored matrix (include/qprop.h, qp_pauli_operator_create):
    ``terms[l]`` = the strings of H_l of the lazy sum  sum_l c_l H_l  as (amplitude, string) pairs, string = 'XIZY'-style label
    (qubit n-1 first), {qubit: label} dict, or an (xmask, zmask) pair of bit masks; the last ``ncoeffs`` terms carry the
    coefficients of ``set_coeffs`` (the others are the drift), as for :class:`Operator`."""

    def __init__(self, ctx, nqubits, terms, ncoeffs=0):
        self.ctx, self.lib = ctx, ctx.lib
        flat = []
        for l, strings in enumerate(terms):
            for amp, st in strings:
                xm, zm = st if (isinstance(st, tuple) and len(st) == 2 and all(isinstance(v, (int, np.integer)) for v in st)) else pauli_masks(st)
                flat.append((int(xm), int(zm), complex(amp), l))
        arr = (qp_pauli_string * max(len(flat), 1))()
        for k, (xm, zm, amp, l) in enumerate(flat):
            arr[k].xmask, arr[k].zmask, arr[k].coef, arr[k].op = xm, zm, qp_c128(amp.real, amp.imag), l
        self._h = _P()
        check(self.lib.qp_pauli_operator_create(ctx._h, int(nqubits), arr, len(flat), len(terms), int(ncoeffs), C.byref(self._h)))
        self.ops = []
        self.ncoeffs = int(ncoeffs)
        self.nqubits = int(nqubits)
        self._refresh_info()
        ctx._adopt(self)


class Liouvillian(Operator):
    """Matrix-free ``liouvillian(H, c_ops; convention)`` (src/generators.jl:473-631) for dense
    ``H = sum_l c_l H_l`` and Lindblad operators: applies the superoperator to the column-major
    ``vec(rho)`` as n x n GEMMs instead of storing an n^2 x n^2 sparse matrix
    (include/qprop.h, qp_liouvillian_create).  Everything an :class:`Operator` offers except
    the entry points that need stored entries."""

    def __init__(self, ctx, H_terms, c_ops=(), ncoeffs=0, convention="TDSE"):
        self.ctx, self.lib = ctx, ctx.lib
        conv = {"TDSE": CONV_TDSE, "LVN": CONV_LVN}[str(convention).upper().lstrip(":")]
        mats = [np.asarray(H.toarray() if hasattr(H, "toarray") else H, dtype=np.complex128) for H in H_terms]
        cops = [np.asarray(A.toarray() if hasattr(A, "toarray") else A, dtype=np.complex128) for A in c_ops]
        if not mats and not cops:
            raise ValueError("Empty Liouvillian, must give at least one of `H` or `c_ops`")   # :627-630
        n = (mats + cops)[0].shape[0]
        for M in mats + cops:
            if M.shape != (n, n):
                raise ValueError("all operators must be square matrices of one size")
        # column-major (Julia) storage
        hbuf = [np.ascontiguousarray(M.T).reshape(-1) for M in mats]
        cbuf = [np.ascontiguousarray(M.T).reshape(-1) for M in cops]
        harr = (_cp * max(len(hbuf), 1))(*[_ptr(b, _cp) for b in hbuf])
        carr = (_cp * max(len(cbuf), 1))(*[_ptr(b, _cp) for b in cbuf])
        if "QP_ROCBLAS_PATH" not in os.environ:     # the rocBLAS this process already uses (PyTorch-ROCm's)
            try:
                import torch
                cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librocblas.so")
                if os.path.exists(cand):
                    os.environ["QP_ROCBLAS_PATH"] = cand
            except ImportError:
                pass
        self._h = _P()
        check(self.lib.qp_liouvillian_create(ctx._h, n, harr, len(hbuf), int(ncoeffs), carr, len(cbuf), conv,
                                             C.byref(self._h)))
        self.ops = []
        self.ncoeffs = int(ncoeffs)
        self.n_hilbert = n
        self._refresh_info()
        ctx._adopt(self)


def host_register(array):
    """Page-lock a NumPy array for fast uploads / downloads (qp_host_register); keep it alive until
    :func:`host_unregister`."""
    check(load().qp_host_register(array.ctypes.data_as(C.c_void_p), array.nbytes))
    return array


def host_unregister(array):
    check(load().qp_host_unregister(array.ctypes.data_as(C.c_void_p)))


class State:
    """ComplexF64[n] in HBM."""

    def __init__(self, ctx, n=None, data=None, device_ptr=None, keepalive=None):
        self.ctx = ctx
        self.lib = ctx.lib
        self._h = _P()
        self._keep = keepalive
        if device_ptr is not None:
            check(self.lib.qp_state_wrap(ctx._h, _P(device_ptr), int(n), C.byref(self._h)))
            self.n = int(n)
        else:
            if data is not None:
                data = np.ascontiguousarray(data, dtype=np.complex128).reshape(-1)
                n = len(data)
            check(self.lib.qp_state_create(ctx._h, int(n), C.byref(self._h)))
            self.n = int(n)
            if data is not None:
                self.upload(data)
        ctx._adopt(self)

    @classmethod
    def wrap_tensor(cls, ctx, t):
        """View a contiguous complex128 torch tensor (on the ctx device) as a State."""
        assert t.is_contiguous() and str(t.dtype) == "torch.complex128"
        return cls(ctx, n=t.numel(), device_ptr=t.data_ptr(), keepalive=t)

    def upload(self, data):
        a, p = _as_c128(np.asarray(data).reshape(-1))
        assert len(a) == self.n
        check(self.lib.qp_state_upload(self._h, p))
        return self

    def numpy(self):
        out = np.empty(self.n, dtype=np.complex128)
        check(self.lib.qp_state_download(self._h, _ptr(out, _cp)))
        return out

    def download(self, out):
        """Into an existing complex128 array (no allocation; pinned arrays -- :func:`host_register` -- at PCIe speed)."""
        assert out.dtype == np.complex128 and out.flags.c_contiguous and out.size == self.n
        check(self.lib.qp_state_download(self._h, _ptr(out, _cp)))
        return out

    def copy_from(self, other):
        check(self.lib.qp_copy(self._h, other._h))
        return self

    def scal(self, alpha):
        check(self.lib.qp_scal(self._h, c128(alpha)))
        return self

    def axpy(self, alpha, x):
        check(self.lib.qp_axpy(c128(alpha), x._h, self._h))
        return self

    def fill(self, alpha):
        check(self.lib.qp_fill(self._h, c128(alpha)))
        return self

    def dot(self, y):
        out = qp_c128()
        check(self.lib.qp_dot(self._h, y._h, C.byref(out)))
        return complex(out.re, out.im)

    def norm(self):
        out = C.c_double(0)
        check(self.lib.qp_norm(self._h, C.byref(out)))
        return out.value

    @property
    def ptr(self):
        return self.lib.qp_state_ptr(self._h)

    def __len__(self):
        return self.n

    # ---- the state interface of the reference (src/interfaces/state.jl:92-140), on the device:
    # similar / copy / copyto! / zero / fill! / lmul! / axpy! / norm / dot and the out-of-place
    # `state + state`, `state - state`, `a * state`.  Elements can be read (`state[i]`) but the
    # type is deliberately not a host array.
    dtype = np.complex128

    @property
    def shape(self):
        return (self.n,)

    def similar(self):
        return State(self.ctx, n=self.n)

    def copy(self):
        return State(self.ctx, n=self.n).copy_from(self)

    def zero(self):
        return State(self.ctx, n=self.n)          # qp_state_create zero-fills

    def lmul(self, a):
        return self.scal(a)

    def __add__(self, other):
        return self.copy().axpy(1.0, other)

    def __sub__(self, other):
        return self.copy().axpy(-1.0, other)

    def __mul__(self, a):
        return self.copy().scal(a)

    __rmul__ = __mul__

    def __getitem__(self, i):
        if not isinstance(i, (int, np.integer)):
            raise TypeError("a device state is indexed by one integer")
        if i < 0:
            i += self.n
        if not 0 <= i < self.n:
            raise IndexError(i)
        return complex(self.numpy()[i])

    def close(self):
        if self._h:
            self.lib.qp_state_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ChebyWrk:
    """Cheby.ChebyWrk (src/cheby.jl:87-124) with its vectors in HBM."""

    def __init__(self, ctx, n, Delta, E_min, dt, limit=1e-12):
        self.ctx = ctx
        self.lib = ctx.lib
        self.coeffs = cheby_coeffs(Delta, dt, limit)
        self.n_coeffs = len(self.coeffs)
        self.Delta, self.E_min, self.dt, self.limit = float(Delta), float(E_min), float(dt), float(limit)
        self._h = _P()
        check(self.lib.qp_cheby_create(ctx._h, int(n), C.byref(self._h)))
        ctx._adopt(self)

    def close(self):
        if self._h:
            self.lib.qp_cheby_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cheby(psi, H, dt, wrk, E_min=None, check_normalization=False):
    """``cheby!(psi, H, dt, wrk; E_min, check_normalization)`` on the device."""
    E_min = wrk.E_min if E_min is None else E_min
    if not getattr(H, "_walk_checked", False):
        _warn_if_off_the_walk(H)
    check(wrk.lib.qp_cheby_step(wrk._h, H._h, psi._h, _ptr(wrk.coeffs, _dp), wrk.n_coeffs, wrk.Delta,
                                float(E_min), float(dt), wrk.dt, wrk.limit, int(check_normalization)))
    return psi


def _warn_if_off_the_walk(H):
    """Once per operator: a Hermitian sparse operator large enough for the strip walk (knob walk_min_blocks) that does not
    take it pays up to 1.8 x per fused term on the per-block kernels -- say which property of its stencil broke the plan."""
    H._walk_checked = True
    try:
        if H.format not in (FMT_HRB, FMT_RBCSR, FMT_CSR) or H.shape[0] < 64 * H.ctx.tuning_get("walk_min_blocks"):
            return
        code, name, text = H.walk_reason()
    except Exception:       # an information query must never take a step down
        return
    if name in ("ok", "not_hermitian", "too_few_blocks", "disabled"):
        return
    try:
        # an operator that is no lattice at all but has a fast path of its own has nothing to be told: irregular columns take
        # the column-blocked mirror, a qubit-register Hamiltonian (row XOR mask) the block-map encoding of its row blocks
        if H.colblock_info()["valid"] or H.value_encoding_info()["valid"]:
            return
        enc = H.encoding_info()["upper"]
        if name in ("not_packed", "no_uniform_run") and enc["block_map"] > 0 and enc["int32"] + enc["int16"] == 0:
            return
    except Exception:
        pass
    import warnings
    warnings.warn(f"cheby!: this operator does not take the strip walk [{name}]: {text}", QPPerformanceWarning, stacklevel=3)


def cheby_batched(psi_panel, H, dt, wrk, batch, E_min=None):
    """``cheby!`` for ``batch`` states at once; ``psi_panel[i*batch + s]`` (state index contiguous)."""
    E_min = wrk.E_min if E_min is None else E_min
    check(wrk.lib.qp_cheby_step_batched(wrk._h, H._h, psi_panel._h, int(batch), _ptr(wrk.coeffs, _dp), wrk.n_coeffs,
                                        wrk.Delta, float(E_min), float(dt), wrk.dt))
    return psi_panel


def acc_schedule(coeffs):
    """Per term of a ``cheby!`` with these coefficients: a ``qp_acc_defer`` telling the fused
    term kernel whether it touches the Psi accumulator (every third term does, folding the
    two before it; include/qprop.h, qp_acc_defer).  Host only."""
    a = np.ascontiguousarray(coeffs, dtype=np.float64)
    out = (qp_acc_defer * (len(a) - 1))()
    check(load().qp_acc_schedule_host(_ptr(a, _dp), len(a), out))
    return out


def cheby_term(H, x, xoff, v0, vout, acc_in, acc_out, c, beta, a_prev, a, phase=1.0, defer=None):
    lib = H.lib
    check(lib.qp_cheby_term(H._h, x._h, int(xoff), v0._h if v0 is not None else None,
                            vout._h if vout is not None else None,
                            acc_in._h if acc_in is not None else None,
                            acc_out._h if acc_out is not None else None, c128(c), float(beta),
                            float(a_prev), float(a), c128(phase), C.byref(defer) if defer is not None else None))


class Split:
    """Boundary / interior partition of an operator's row blocks (multi-GPU overlap)."""

    def __init__(self, op, send_rows):
        self.op, self.lib = op, op.lib
        rows = np.ascontiguousarray(send_rows, dtype=np.int64)
        self._h = _P()
        check(self.lib.qp_split_create(op._h, _ptr(rows, _i64p), len(rows), C.byref(self._h)))
        nb, ni = C.c_int64(), C.c_int64()
        check(self.lib.qp_split_info(self._h, C.byref(nb), C.byref(ni)))
        self.n_boundary, self.n_interior = nb.value, ni.value
        op.ctx._adopt(self)

    def check(self):
        check(self.lib.qp_split_check(self._h))

    def walk_info(self):
        """The interior launch as a strip walk: (valid, first walked block, end, interior blocks on the per-block path)."""
        out = np.zeros(4, dtype=np.int64)
        check(self.lib.qp_split_walk_info(self._h, _ptr(out, _i64p)))
        return dict(zip(("valid", "first_block", "end_block", "edge_blocks"), (int(v) for v in out)))

    def close(self):
        if self._h:
            self.lib.qp_split_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cheby_term_split(H, split, boundary_stream, first, x, xoff, v0, vout, acc_in, acc_out, slab, c, beta, a_prev, a,
                     phase=1.0, defer=None):
    # hot in the multi-GPU loop (one call per term): keep the Python side minimal
    st = H.lib.qp_cheby_term_split(H._h, split._h, boundary_stream, 1 if first else 0, x._h, xoff,
                                   v0._h if v0 is not None else None, vout._h if vout is not None else None,
                                   acc_in._h if acc_in is not None else None,
                                   acc_out._h if acc_out is not None else None,
                                   slab._h if slab is not None else None, c128(c), beta, a_prev, a, c128(phase),
                                   C.byref(defer) if defer is not None else None)
    if st:
        check(st)


def rccl_library_path():
    """The librccl.so of this process: the one PyTorch-ROCm ships (torch.distributed's "nccl"
    backend is that RCCL), so that the library's communicator and torch's use the same code."""
    import torch
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if not os.path.exists(path):
        path = "/opt/rocm/lib/librccl.so"
    return path


class Comm:
    """RCCL communicator owned by the library (include/qprop.h, qp_comm).  ``exchange_id``:
    a callable that takes rank 0's 128-byte id (None on the other ranks) and returns it on every
    rank -- e.g. a ``torch.distributed.broadcast_object_list`` over any backend.  ``agree``: a callable
    that takes this rank's error (or None) and returns the first error of ANY rank (or None) on every
    rank.  Set-up runs in two phases: the part that can fail on one rank alone (dlopen of librccl,
    symbol resolution: ``qp_comm_prepare``) comes first, the ranks agree that all came through, and
    only then all enter the collective ``ncclCommInitRank`` (``qp_comm_connect``) -- a rank that failed
    locally would otherwise leave the others blocked in it.  Both callables are called exactly once, in
    this order, on every rank, whatever happens locally."""

    def __init__(self, ctx, rank, world, exchange_id, lib_path=None, agree=None):
        self.ctx, self.lib = ctx, ctx.lib
        self.rank, self.world = int(rank), int(world)
        path = (lib_path or rccl_library_path()).encode()
        self._h = _P()
        local_err = None
        if self.lib.qp_comm_prepare(ctx._h, path, self.rank, self.world, C.byref(self._h)) != QP_OK:
            local_err = self.lib.qp_last_error().decode(errors="replace")
            self._h = _P()
        uid = None
        if self.rank == 0 and local_err is None:
            buf = C.create_string_buffer(128)
            if self.lib.qp_comm_unique_id(path, buf) == QP_OK:   # on failure the other ranks learn it (None)
                uid = buf.raw
        try:
            uid = exchange_id(uid)                               # raises on every rank when rank 0 has no id
            err = agree(local_err) if agree is not None else local_err
        except Exception:
            self.close()
            raise
        if err is not None:
            self.close()
            raise QPError(QP_E_RCCL, f"RCCL communicator set-up failed on a rank: {err}")
        assert isinstance(uid, (bytes, bytearray)) and len(uid) == 128
        st = self.lib.qp_comm_connect(self._h, bytes(uid))
        if st != QP_OK:
            msg = self.lib.qp_last_error().decode(errors="replace")
            self.close()
            raise QPError(st, msg)
        ctx._adopt(self)

    def allgather(self, send, recv, count, stream=None):
        check(self.lib.qp_comm_allgather(self._h, send._h, recv._h, int(count), stream))

    def close(self):
        if self._h:
            self.lib.qp_comm_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


EXCHANGE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int), C.c_int,
                          C.POINTER(C.c_int), C.c_int, C.c_void_p)


def comm_info(comm):
    """qp_comm_info of a :class:`Comm` / :class:`CallbackComm`: what RCCL itself says the communicator is."""
    v = [C.c_int(0) for _ in range(5)]
    check(comm.lib.qp_comm_info(comm._h, *[C.byref(x) for x in v]))
    return dict(zip(("world", "rank", "rccl_ranks", "rccl_rank", "is_callback"), (int(x.value) for x in v)))


class CallbackComm:
    """A ``qp_comm`` whose exchange is done by Python (include/qprop.h, qp_comm_create_callback):
    ``exchange(send_ptr, count, recv_base_ptr, send_to, recv_from, stream)`` with device
    pointers as ints and the neighbour lists as Python lists (None = all-gather)."""

    def __init__(self, ctx, rank, world, exchange):
        self.ctx, self.lib = ctx, ctx.lib
        self.rank, self.world = int(rank), int(world)

        def _cb(user, send, count, recv, send_to, nst, recv_from, nrf, stream):
            try:
                st = None if nst < 0 else [send_to[i] for i in range(nst)]
                rf = None if nst < 0 else [recv_from[i] for i in range(nrf)]
                exchange(send, int(count), recv, st, rf, stream)
                return 0
            except Exception as exc:      # noqa: BLE001 -- must not propagate through the C frame
                import traceback
                traceback.print_exc()
                self.error = exc
                return 1
        self.error = None
        self._cb = EXCHANGE_CB(_cb)
        self._h = _P()
        check(self.lib.qp_comm_create_callback(ctx._h, self.rank, self.world, C.cast(self._cb, _P), None, C.byref(self._h)))
        ctx._adopt(self)

    def close(self):
        if self._h:
            self.lib.qp_comm_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedChebyStepper:
    """One host call per ``cheby!`` on a row-partitioned state (qp_sharded_cheby_step)."""

    def __init__(self, op, split, comm, X0, X1, acc, slab, send_rows, M, direct_send, send_to=None, recv_from=None):
        self.lib = op.lib
        rows = np.ascontiguousarray(send_rows, dtype=np.int64)
        d = qp_sharded_cheby_desc()
        if send_to is not None:           # neighbour exchange (ncclSend / ncclRecv) instead of the all-gather
            st = np.ascontiguousarray(send_to, dtype=np.int32)
            rf = np.ascontiguousarray(recv_from, dtype=np.int32)
            d.send_to, d.n_send_to = st.ctypes.data_as(C.POINTER(C.c_int)), len(st)
            d.recv_from, d.n_recv_from = rf.ctypes.data_as(C.POINTER(C.c_int)), len(rf)
        else:
            d.n_send_to = -1
        d.op, d.split, d.comm = op._h, (split._h if split is not None else None), (comm._h if comm is not None else None)
        d.X0, d.X1, d.acc, d.slab = X0._h, X1._h, acc._h, (slab._h if slab is not None else None)
        d.send_rows, d.nsend, d.M, d.direct_send = _ptr(rows, _i64p), len(rows), int(M), int(bool(direct_send))
        self._keep = (op, split, comm, X0, X1, acc, slab)
        self._h = _P()
        check(self.lib.qp_sharded_cheby_create(C.byref(d), C.byref(self._h)))
        op.ctx._adopt(self)

    def step(self, coeffs, Delta, E_min, dt):
        st = self.lib.qp_sharded_cheby_step(self._h, _ptr(coeffs, _dp), len(coeffs), Delta, E_min, dt)
        if st:
            check(st)

    def close(self):
        if self._h:
            self.lib.qp_sharded_cheby_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Krylov:
    def __init__(self, ctx, n, nvec):
        self.ctx, self.lib, self.n, self.nvec = ctx, ctx.lib, int(n), int(nvec)
        self._h = _P()
        check(self.lib.qp_krylov_create(ctx._h, self.n, self.nvec, C.byref(self._h)))
        ctx._adopt(self)

    def vec(self, i):
        out = np.empty(self.n, dtype=np.complex128)
        check(self.lib.qp_krylov_download(self._h, int(i), _ptr(out, _cp)))
        return out

    def view(self, i):
        """Non-owning State view of Arnoldi vector i."""
        h = _P()
        check(self.lib.qp_krylov_vec(self._h, int(i), C.byref(h)))
        s = State.__new__(State)
        s.ctx, s.lib, s._h, s._keep, s.n = self.ctx, self.lib, h, self, self.n
        self.ctx._adopt(s)
        return s

    # building blocks of a row-partitioned Arnoldi (the caller all-reduces between them)
    def multidot(self, j, reduced):
        check(self.lib.qp_krylov_multidot(self._h, int(j), reduced._h))

    def project(self, j, dt, reduced, hess_col, norm_partials):
        check(self.lib.qp_krylov_project(self._h, int(j), float(dt), reduced._h, hess_col._h, norm_partials._h))

    def normalize(self, j, dt, norm_min, norm_partials, hess_norm):
        check(self.lib.qp_krylov_normalize(self._h, int(j), float(dt), float(norm_min), norm_partials._h, hess_norm._h))

    def combine(self, out, use_out, s0, first, m, coefs, norm_partials=None):
        a, p = _as_c128(np.atleast_1d(coefs))
        check(self.lib.qp_combine(out._h, int(bool(use_out)), c128(s0), self._h, int(first), int(m), p,
                                  norm_partials._h if norm_partials is not None else None))

    def close(self):
        if self._h:
            self.lib.qp_krylov_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def arnoldi(Hess, q, m, psi, H, dt=1.0, extended=True, norm_min=1e-15):
    """``arnoldi!``; ``Hess`` is a Fortran-ordered square complex128 array."""
    assert Hess.flags.f_contiguous and Hess.dtype == np.complex128
    m_out = C.c_int(0)
    check(H.lib.qp_arnoldi(H._h, q._h, int(m), psi._h, float(dt), int(extended), float(norm_min),
                           _ptr(Hess, _cp), Hess.shape[0], C.byref(m_out)))
    return m_out.value


def extend_arnoldi(Hess, q, m, H, dt=1.0, norm_min=1e-15):
    assert Hess.flags.f_contiguous and Hess.dtype == np.complex128
    ext = C.c_int(0)
    check(H.lib.qp_arnoldi_extend(H._h, q._h, int(m), float(dt), float(norm_min), _ptr(Hess, _cp),
                                  Hess.shape[0], C.byref(ext)))
    return bool(ext.value)


class NewtonWrk:
    """Newton.NewtonWrk (src/newton.jl:23-60)."""

    def __init__(self, ctx, n, m_max=10):
        self.ctx, self.lib = ctx, ctx.lib
        self._h = _P()
        check(self.lib.qp_newton_create(ctx._h, int(n), int(m_max), C.byref(self._h)))
        ctx._adopt(self)
        self.m_max = min(int(m_max), int(n) - 1)
        self.restarts = 0
        self.n_a = 0
        self.n_leja = 0
        self.radius = 0.0
        self.stats = None

    def coeffs(self):
        a = np.empty(max(self.n_a, 1), dtype=np.complex128)
        leja = np.empty(max(self.n_a, 1), dtype=np.complex128)
        check(self.lib.qp_newton_get_coeffs(self._h, _ptr(a, _cp), _ptr(leja, _cp), len(a)))
        return a[: self.n_a], leja[: self.n_a]

    def close(self):
        if self._h:
            self.lib.qp_newton_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def newton(psi, H, dt, wrk, func=None, norm_min=1e-14, relerr=1e-12, max_restarts=50):
    """``newton!(psi, H, dt, wrk; func, norm_min, relerr, max_restarts)`` on the device."""
    fid, cb, _keep = _func_args(func)
    st = qp_newton_stats()
    check(wrk.lib.qp_newton_step(wrk._h, H._h, psi._h, float(dt), fid, cb, None, float(norm_min),
                                 float(relerr), int(max_restarts), C.byref(st)))
    wrk.restarts, wrk.n_a, wrk.n_leja, wrk.radius = st.restarts, st.n_a, st.n_leja, st.radius
    wrk.stats = {k: getattr(st, k) for k, _ in qp_newton_stats._fields_}
    return psi


def propagate_steps(H, psi, wrk, dts, coeff_table=None, observables=(), store_states=False, func=None,
                    norm_min=1e-14, relerr=1e-12, max_restarts=50, check_normalization=False):
    """The ``propagate`` step loop (src/propagate.jl:283-344) in one library call.
    ``wrk``: a ChebyWrk or NewtonWrk; ``dts``: signed time step per interval; ``coeff_table``:
    [nsteps, ncoeffs] operator coefficients per interval.  Returns (expvals [nsteps+1, nobs] or
    None, states [nsteps+1, n] or None); ``psi`` holds the final state."""
    dts = np.ascontiguousarray(dts, dtype=np.float64)
    nsteps = len(dts)
    spec = qp_prop_spec()
    keep = None
    if isinstance(wrk, ChebyWrk):
        spec.method, spec.cheby = 0, wrk._h
        spec.a, spec.n_coeffs = _ptr(wrk.coeffs, _dp), wrk.n_coeffs
        spec.Delta, spec.E_min, spec.wrk_dt, spec.limit = wrk.Delta, wrk.E_min, wrk.dt, wrk.limit
        spec.check_normalization = int(check_normalization)
    else:
        fid, cb, keep = _func_args(func)
        spec.method, spec.newton, spec.func_id, spec.cb = 1, wrk._h, fid, cb
        spec.norm_min, spec.relerr, spec.max_restarts = norm_min, relerr, int(max_restarts)
    ncoeffs = H.ncoeffs
    if ncoeffs:
        tab = np.ascontiguousarray(coeff_table, dtype=np.complex128).reshape(nsteps, ncoeffs)
        tabp = _ptr(tab, _cp)
    else:
        tabp = None
    nobs = len(observables)
    obs_arr = (_P * max(nobs, 1))(*[o._h for o in observables]) if nobs else None
    ev = np.empty((nsteps + 1, nobs), dtype=np.complex128) if nobs else None
    st = np.empty((nsteps + 1, psi.n), dtype=np.complex128) if store_states else None
    check(H.lib.qp_propagate(H._h, psi._h, C.byref(spec), _ptr(dts, _dp), tabp, ncoeffs, nsteps, obs_arr, nobs,
                             _ptr(ev, _cp) if nobs else None, _ptr(st, _cp) if store_states else None))
    del keep
    return ev, st


def ritzvals(G, state, m_min, m_max=None, prec=1e-5, norm_min=1e-15):
    m_max = 2 * m_min if m_max is None else m_max
    out = np.empty(max(m_max, 8), dtype=np.complex128)
    n = C.c_int(0)
    check(G.lib.qp_ritzvals(G._h, state._h, int(m_min), int(m_max), float(prec), float(norm_min),
                            _ptr(out, _cp), C.byref(n)))
    return out[: n.value].copy()


def specrange_arnoldi(H, state, m_min=25, m_max=60, prec=1e-3, norm_min=1e-15, enlarge=True):
    lo, hi = C.c_double(0), C.c_double(0)
    check(H.lib.qp_specrange_arnoldi(H._h, state._h, int(m_min), int(m_max), float(prec), float(norm_min),
                                     int(enlarge), C.byref(lo), C.byref(hi)))
    return lo.value, hi.value
