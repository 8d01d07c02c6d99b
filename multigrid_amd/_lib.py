"""ctypes binding of libmgx.so (the C ABI in include/mgx.h).

There is no CPU fallback: if the shared library is missing or a call fails the
caller gets an exception.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MGX_LIB selects another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("MGX_LIB") or os.path.join(_HERE, "libmgx.so")

_dp = ctypes.POINTER(ctypes.c_double)
_vp = ctypes.c_void_p


class MgxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmgx error %d: %s" % (code, msg))
        self.code = code


class mgx_options(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_int),
        ("device", ctypes.c_int),
        ("stream", ctypes.c_void_p),
        ("schedule", ctypes.c_int),
        ("verbose", ctypes.c_int),
        ("coarse_solve", ctypes.c_int),
        ("ngpus", ctypes.c_int),
        ("tail_m", ctypes.c_int),
        ("flags", ctypes.c_int),
        ("reserved", ctypes.c_int * 4),
    ]


class mgx_slab_info(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_int),
        ("device", ctypes.c_int), ("m", ctypes.c_int), ("col_first", ctypes.c_int), ("ncols", ctypes.c_int),
        ("own_first", ctypes.c_int), ("own_count", ctypes.c_int),
        ("f_dev", ctypes.c_void_p), ("v_dev", ctypes.c_void_p), ("stream", ctypes.c_void_p),
    ]


class mgx_mid_info(ctypes.Structure):
    _fields_ = [
        ("ok", ctypes.c_int), ("threads", ctypes.c_int), ("points_per_thread", ctypes.c_int),
        ("tile_i", ctypes.c_int), ("tile_j", ctypes.c_int), ("ntiles_i", ctypes.c_int), ("ntiles_j", ctypes.c_int),
        ("lds_bytes", ctypes.c_size_t),
    ]


class mgx_slab_op(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_int),
        ("nsweeps", ctypes.c_int), ("pro_stage", ctypes.c_int), ("norm_stage", ctypes.c_int),
        ("restrict_out", ctypes.c_int), ("zero_in", ctypes.c_int), ("store", ctypes.c_int),
        ("v_in", ctypes.c_void_p), ("v_out", ctypes.c_void_p), ("f", ctypes.c_void_p),
        ("coarse_v", ctypes.c_void_p), ("coarse_f", ctypes.c_void_p), ("norm_slot", ctypes.c_void_p),
        ("m", ctypes.c_int), ("ld", ctypes.c_int), ("ncols", ctypes.c_int),
        ("mc", ctypes.c_int), ("ldc", ctypes.c_int), ("nccols", ctypes.c_int),
        ("jlo", ctypes.c_int), ("jhi", ctypes.c_int),
        ("jfix_lo", ctypes.c_int), ("jfix_hi", ctypes.c_int),
        ("gf0", ctypes.c_int), ("gc0", ctypes.c_int),
        ("dx", ctypes.c_double), ("omega", ctypes.c_double),
        ("jlo2", ctypes.c_int), ("jhi2", ctypes.c_int),
    ]


SLAB_MAX_SWEEPS = 4

SCHEDULE_DEFAULT = 0
SCHEDULE_STEPWISE = 1
SCHEDULE_FUSED = 2
COARSE_REFERENCE = 0  # main.cpp:153-169: only a 3x3 coarsest grid is solved
COARSE_EXACT = 1      # extension: any coarsest grid up to 65x65 is solved exactly

# name -> (restype, argtypes); every symbol include/mgx.h declares.
SIGNATURES = {
    "mgx_version": (ctypes.c_int, []),
    "mgx_last_error": (ctypes.c_char_p, []),
    "mgx_depth": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "mgx_solve2d": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                   _vp, _vp, _vp]),
    "mgx_plan_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_double, ctypes.c_int, _vp]),
    "mgx_plan_destroy": (ctypes.c_int, [_vp]),
    "mgx_plan_depth": (ctypes.c_int, [_vp]),
    "mgx_plan_level_m": (ctypes.c_int, [_vp, ctypes.c_int]),
    "mgx_plan_bytes": (ctypes.c_size_t, [_vp]),
    "mgx_coarse_exact": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_double, _vp]),
    "mgx_plan_f_dev": (_vp, [_vp]),
    "mgx_plan_v_dev": (_vp, [_vp]),
    "mgx_plan_commit": (ctypes.c_int, [_vp]),
    "mgx_plan_profile": (ctypes.c_int, [_vp, ctypes.c_int]),
    "mgx_plan_profile_get": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_long),
                                            ctypes.POINTER(ctypes.c_double)]),
    "mgx_plan_profile_level": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_long),
                                              ctypes.POINTER(ctypes.c_double)]),
    "mgx_kernel_name": (ctypes.c_char_p, [ctypes.c_int]),
    "mgx_plan_set_dev": (ctypes.c_int, [_vp, _vp, _vp]),
    "mgx_plan_get_dev": (ctypes.c_int, [_vp, _vp]),
    "mgx_plan_upload": (ctypes.c_int, [_vp, _vp, _vp]),
    "mgx_plan_upload_boundary": (ctypes.c_int, [_vp, _vp]),
    "mgx_plan_download": (ctypes.c_int, [_vp, _vp]),
    "mgx_plan_cycles": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                       ctypes.c_int, _vp, _vp]),
    "mgx_plan_cycles_until": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                             ctypes.c_double, ctypes.c_int, _vp, _vp, ctypes.POINTER(ctypes.c_int),
                                             ctypes.POINTER(ctypes.c_int)]),
    "mgx_plan_fmg": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, _vp, _vp]),
    "mgx_relax": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double, _vp]),
    "mgx_residual": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_double, _vp]),
    "mgx_restrict": (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_int, _vp]),
    "mgx_refine": (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_int, _vp]),
    "mgx_prolong_add": (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_int, _vp]),
    "mgx_direct3x3": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_double, _vp]),
    "mgx_norminf": (ctypes.c_int, [_vp, ctypes.c_size_t, _vp, _vp]),
    "mgx_fused_down": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double,
                                      _vp]),
    "mgx_fused_up": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double,
                                    _vp, _vp]),
    "mgx_slab_launch": (ctypes.c_int, [_vp, _vp]),
    "mgx_norm_decode": (ctypes.c_double, [ctypes.c_ulonglong]),
    "mgx_plan_vcycle_async": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int]),
    "mgx_residual_norm": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_double, _vp, _vp]),
    "mgx_residual_norm_noalloc": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_double, _vp, _vp]),
    "mgx_mplan_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                        _vp, _vp]),
    "mgx_mplan_destroy": (ctypes.c_int, [_vp]),
    "mgx_mplan_partition": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_int]),
    "mgx_halo_plan": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                     ctypes.POINTER(ctypes.c_int)]),
    "mgx_mid_shape": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.POINTER(mgx_mid_info)]),
    "mgx_mid_spans": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "mgx_mplan_nslabs": (ctypes.c_int, [_vp]),
    "mgx_mplan_depth": (ctypes.c_int, [_vp]),
    "mgx_mplan_distributed_levels": (ctypes.c_int, [_vp]),
    "mgx_mplan_tail_m": (ctypes.c_int, [_vp]),
    "mgx_mplan_bytes": (ctypes.c_size_t, [_vp, ctypes.c_int]),
    "mgx_mplan_upload": (ctypes.c_int, [_vp, _vp, _vp]),
    "mgx_mplan_download": (ctypes.c_int, [_vp, _vp]),
    "mgx_mplan_slab": (ctypes.c_int, [_vp, ctypes.c_int, _vp]),
    "mgx_mplan_commit": (ctypes.c_int, [_vp]),
    "mgx_mplan_cycles": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                        _vp, _vp]),
    "mgx_mplan_fmg": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, _vp, _vp]),
    "mgx_mplan_tune": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_double]),
    "mgx_mplan_profile": (ctypes.c_int, [_vp, ctypes.c_int]),
    "mgx_mplan_profile_get": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_long),
                                             ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "mgx_mplan_stats": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long),
                                       ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "mgx_plan_tune": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_double]),
    "mgx_tune_mode": (ctypes.c_int, [ctypes.c_int]),
    "mgx_tune_reset": (ctypes.c_int, []),
    "mgx_tune_report": (ctypes.c_size_t, [ctypes.c_char_p, ctypes.c_size_t]),
    "mgx_plan_set_stream": (ctypes.c_int, [_vp, _vp]),
}

KERNEL_COUNT = 8

_LIB = None


def load():
    """Load libmgx.so (built in-tree by `make -C multigrid_amd/csrc`)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("multigrid_amd: %s is missing; build it with `make -C multigrid_amd/csrc` "
                              "(there is no CPU fallback)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    """Raise on a negative return code; pass non-negative codes through."""
    if rc < 0:
        msg = load().mgx_last_error()
        raise MgxError(rc, msg.decode("utf-8", "replace") if msg else "")
    return rc


def tune_report():
    """The launch shapes measured so far in this process (mgx_tune_report), a list of lines."""
    L = load()
    n = L.mgx_tune_report(None, 0)
    buf = ctypes.create_string_buffer(n)
    L.mgx_tune_report(buf, n)
    return [l for l in buf.value.decode().splitlines() if l]


FLAG_ZERO_GUESS = 1
STOP_MAX_CYCLES, STOP_TOLERANCE, STOP_FLOOR = 0, 1, 2


def make_options(device=-1, stream=None, schedule=SCHEDULE_DEFAULT, verbose=0, coarse_solve=0, ngpus=0, tail_m=0,
                 flags=0):
    o = mgx_options()
    o.struct_size = ctypes.sizeof(mgx_options)
    o.device = device
    o.stream = stream
    o.schedule = schedule
    o.verbose = verbose
    o.coarse_solve = coarse_solve
    o.ngpus = ngpus
    o.tail_m = tail_m
    o.flags = flags
    return o
