"""ctypes binding of libsla_hip.so (include/sla_hip.h).  There is NO CPU fallback: if the HIP
library is missing or no GPU is visible the calls raise."""
import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)                       # sparse-linear-algebra_amd/
# SLA_HIP_LIB: load another build of the same library (kernel A/B experiments, tools/build_variant.sh)
LIB_PATH = os.environ.get("SLA_HIP_LIB") or os.path.join(_ROOT, "lib", "libsla_hip.so")
CSRC = os.path.join(_ROOT, "csrc")

(OK, ERR_DIM_MISMATCH, ERR_UNSUPPORTED_METHOD, ERR_OOB, ERR_HIP, ERR_RCCL, ERR_ALLOC, ERR_INVALID,
 ERR_NO_DEVICE, ERR_NEEDS_PIVOTING) = range(10)

FLAG_CONVERGED, FLAG_MAX_ITERS, FLAG_DIAGONAL, FLAG_BREAKDOWN, FLAG_NONFINITE = 1, 2, 4, 8, 16
FLAG_SYNC_TIMEOUT, FLAG_RELAXED_ORDER = 32, 64
FOLD_EXACT, FOLD_REGROUPED, FOLD_RELAXED = 0, 1, 2   # sla_fold_kind
KERNEL_ALL = -1
KERNEL_SPMV, KERNEL_SPMV_DOT, KERNEL_SPMV_DOT2, KERNEL_SPMV_RES, KERNEL_SPMV_DUAL = 0, 1, 2, 3, 4
KERNEL_BICG_K2, KERNEL_BICG_K4, KERNEL_BICG_K5, KERNEL_CGS_C2, KERNEL_CGS_C4 = 5, 6, 7, 8, 9
KERNEL_BICG_K45 = 10
KERNEL_EXCHANGE, KERNEL_SUMS = 11, 12
KERNEL_ONCHIP = 13


class SlaError(RuntimeError):
    """Any libsla_hip failure that has no reference counterpart (HIP, RCCL, allocation, misuse)."""

    def __init__(self, code, msg):
        super().__init__(f"[sla status {code}] {msg}")
        self.code = code


class MatVecSizeMismatchException(SlaError):
    """Control/Exception/Common.hs:44-51 (also `error "matVec : mismatched dimensions"`, Common.hs:250)."""


class IterationException(SlaError):
    """IterE (Control/Exception/Common.hs:67-76), thrown by linSolve0 for GMRES_/BCG_ (Sparse.hs:1031)."""


class NeedsPivoting(SlaError):
    """NeedsPivoting (Control/Exception/Common.hs:58-61), thrown by triLowerSolve / triUpperSolve (Sparse.hs:757, :792)."""


class IndexOutOfBounds(SlaError):
    """`error "insertSpMatrix : index out of bounds"` (SpMatrix.hs:208)."""


class SolveOpts(C.Structure):
    """sla_solve_opts.  struct_size (ABI versioning, include/sla_hip.h) is filled in here: positional arguments start at max_iters."""
    _fields_ = [("struct_size", C.c_int32), ("max_iters", C.c_int32), ("tol_abs", C.c_double), ("tol_rel", C.c_double),
                ("check_every", C.c_int32), ("true_residual", C.c_int32), ("history", C.c_void_p), ("history_cap", C.c_int32)]

    def __init__(self, max_iters=200, tol_abs=1e-6, tol_rel=1e-4, check_every=16, true_residual=1, history=None, history_cap=0):
        super().__init__(C.sizeof(SolveOpts), max_iters, tol_abs, tol_rel, check_every, true_residual, history, history_cap)


class CsrProps(C.Structure):
    """sla_csr_props (include/sla_hip.h): typed properties of a lowered matrix."""
    _fields_ = [("struct_size", C.c_int32), ("fold", C.c_int32), ("x_exchange", C.c_int32), ("nranks", C.c_int32),
                ("rows_local", C.c_int64), ("nnz_local", C.c_int64), ("rowptr_bits", C.c_int32), ("reserved", C.c_int32)]

    def __init__(self):
        super().__init__(C.sizeof(CsrProps))


class SolveInfo(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("iters", C.c_int32), ("flags", C.c_int32), ("resnorm", C.c_double),
                ("r0norm", C.c_double), ("tol", C.c_double), ("history_len", C.c_int32)]

    def __init__(self):
        super().__init__(C.sizeof(SolveInfo))

    def as_dict(self):
        return {"iters": self.iters, "flags": self.flags, "resnorm": self.resnorm,
                "r0norm": self.r0norm, "tol": self.tol,
                "converged": bool(self.flags & FLAG_CONVERGED), "relaxed_order": bool(self.flags & FLAG_RELAXED_ORDER)}


# every symbol include/sla_hip.h declares: (name, restype, argtypes)
_vp, _i64, _int, _dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
_pp = C.POINTER(C.c_void_p)
_pi64, _pint, _pdbl = C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_double)
PROTOTYPES = [
    ("sla_ctx_create", _int, [_int, _pp]),
    ("sla_dist_unique_id", _int, [_vp]),
    ("sla_ctx_create_dist", _int, [_int, _int, _int, _vp, _pp]),
    ("sla_ctx_create_multi", _int, [_int, _vp, _pp]),
    ("sla_ctx_create_loopback", _int, [_int, _int, _int, _int, _pp]),
    ("sla_ctx_destroy", _int, [_vp]),
    ("sla_ctx_sync", _int, [_vp]),
    ("sla_ctx_rank", _int, [_vp, _pint, _pint]),
    ("sla_ctx_set_option", _int, [_vp, C.c_char_p, C.c_char_p]),
    ("sla_ctx_get_option", _int, [_vp, C.c_char_p, C.c_char_p, _int]),
    ("sla_debug_binding_violations", C.c_long, []),
    ("sla_stream_probe", _int, [_vp, _int, _int, _i64, _int, _pdbl, _pdbl]),
    ("sla_dist_p2p_selftest", _int, [_vp, _i64, _int, _pdbl]),
    ("sla_dist_preflight", _int, [_vp, _int, _i64, _pdbl, _pdbl]),
    ("sla_ctx_row_range", _int, [_vp, _i64, _pi64, _pi64]),
    ("sla_last_error", C.c_char_p, []),
    ("sla_version", C.c_char_p, []),
    ("sla_abi_version", _int, []),
    ("sla_csr_from_coo", _int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _int, _pp]),
    ("sla_csr_from_csr", _int, [_vp, _i64, _i64, _vp, _vp, _vp, _pp]),
    ("sla_csr_from_csr_rows", _int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _pp]),
    ("sla_csr_from_matrix_market", _int, [_vp, C.c_char_p, _int, _pp]),
    ("sla_csr_from_csc", _int, [_vp, _i64, _i64, _vp, _vp, _vp, _pp]),
    ("sla_csr_export_csc", _int, [_vp, _vp, _vp, _vp]),
    ("sla_csr_transpose", _int, [_vp, _pp]),
    ("sla_csr_from_csb", _int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _pp]),
    ("sla_vec_from_matrix_market", _int, [_vp, C.c_char_p, _pp]),
    ("sla_jacobi_pre", _int, [_vp, _pp]),
    ("sla_csr_diag_mul", _int, [_vp, _vp, _pp]),
    ("sla_tri_solve", _int, [_vp, _int, _vp, _vp, _pi64]),
    ("sla_tri_solve_info", _int, [_vp, _int, _pi64, _pi64]),
    ("sla_ssor_pre", _int, [_vp, C.c_double, _pp, _pp]),
    ("sla_csr_destroy", _int, [_vp]),
    ("sla_csr_dims", _int, [_vp, _pi64, _pi64, _pi64, _pi64]),
    ("sla_csr_matmat", _int, [_vp, _vp, _int, _pp]),
    ("sla_ilu0_pre", _int, [_vp, _int, _pp, _pp, _vp]),
    ("sla_csr_export", _int, [_vp, _vp, _vp, _vp]),
    ("sla_csr_is_diagonal", _int, [_vp, _pint]),
    ("sla_vec_create", _int, [_vp, _i64, _vp, _pp]),
    ("sla_vec_create_local", _int, [_vp, _i64, _vp, _pp]),
    ("sla_vec_destroy", _int, [_vp]),
    ("sla_vec_dim", _int, [_vp, _pi64, _pi64]),
    ("sla_vec_to_host", _int, [_vp, _vp]),
    ("sla_vec_to_host_local", _int, [_vp, _vp]),
    ("sla_vec_copy", _int, [_vp, _vp]),
    ("sla_spmv", _int, [_vp, _vp, _vp]),
    ("sla_spmv_t", _int, [_vp, _vp, _vp]),
    ("sla_dot", _int, [_vp, _vp, _pdbl]),
    ("sla_nrm2", _int, [_vp, _pdbl]),
    ("sla_axpby", _int, [_dbl, _vp, _dbl, _vp]),
    ("sla_scal", _int, [_dbl, _vp]),
    ("sla_solver_init", _int, [_int, _vp, _vp, _vp, _pp]),
    ("sla_solver_step", _int, [_vp, _int]),
    ("sla_solver_get", _int, [_vp, _int, _vp]),
    ("sla_solver_clone", _int, [_vp, _pp]),
    ("sla_solver_set_shadow", _int, [_vp, _vp]),
    ("sla_solver_destroy", _int, [_vp]),
    ("sla_bicgstab_init", _int, [_vp, _vp, _vp, _pp]),
    ("sla_bicgstab_step", _int, [_vp, _int]),
    ("sla_cgs_init", _int, [_vp, _vp, _vp, _pp]),
    ("sla_cgs_step", _int, [_vp, _int]),
    ("sla_linsolve0", _int, [_int, _vp, _vp, _vp, C.POINTER(SolveOpts), _vp, C.POINTER(SolveInfo)]),
    ("sla_arnoldi", _int, [_vp, _vp, _int, _vp, _vp, _pint]),
    ("sla_gmres", _int, [_vp, _vp, _vp, _int, C.POINTER(SolveOpts), _vp, C.POINTER(SolveInfo)]),
    ("sla_linsolve", _int, [_vp, _vp, _vp, C.POINTER(SolveInfo)]),
    ("sla_prof_start", _int, [_vp, _int, _int]),
    ("sla_prof_stop", _int, [_vp, _pint, _pdbl, _pdbl]),
    ("sla_prof_query", _int, [_vp, _int, _pint, _pdbl, _pdbl]),
    ("sla_device_count", _int, [_pint]),
    ("sla_ctx_comm_ranks", _int, [_vp, _pint]),
    ("sla_csr_kernel_info", _int, [_vp, C.c_char_p, _int]),
    ("sla_csr_lower_info", _int, [_vp, C.c_char_p, _int]),
    ("sla_csr_get_props", _int, [_vp, C.POINTER(CsrProps)]),
    ("sla_csr_exchange_plan", _int, [_vp, _vp, _vp, _int]),
    ("sla_plan_window_exchange", _int, [_int, _int, _i64, _vp, _vp, _vp, _vp, _vp, _pint]),
    ("sla_plan_allgather_passes", _int, [_int, _int, _i64, _int, _int, _int, _vp, _vp, _vp, _pint, _pint]),
    ("sla_plan_allgather_groups", _int, [_int, _i64, _int, _int, _int, _vp, _int, _pint]),
]

_LIB = None


def build(force=False, verbose=False):
    """Compile libsla_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"] + (["-B"] if force else [])
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
    if out.returncode != 0:
        raise RuntimeError("building libsla_hip.so failed")
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(libsla_hip has no CPU fallback)")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, res, args in PROTOTYPES:
            f = getattr(L, name)          # AttributeError here = header/library drift
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    if rc == OK:
        return
    msg = lib().sla_last_error().decode("utf-8", "replace")
    if rc == ERR_DIM_MISMATCH:
        raise MatVecSizeMismatchException(rc, msg)
    if rc == ERR_UNSUPPORTED_METHOD:
        raise IterationException(rc, msg)
    if rc == ERR_OOB:
        raise IndexOutOfBounds(rc, msg)
    if rc == ERR_NEEDS_PIVOTING:
        raise NeedsPivoting(rc, msg)
    raise SlaError(rc, msg)
