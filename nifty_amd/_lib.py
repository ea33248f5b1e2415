"""ctypes binding of libniftyk (the C ABI declared in include/niftyk.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc --offload-arch=gfx950).  Loading
fails loudly if it is missing: device code paths never fall back to anything else.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NK_LIB_PATH: developer override used for A/B runs of alternative builds (tools/); the product default is in-tree
LIB_PATH = os.environ.get("NK_LIB_PATH") or os.path.join(_HERE, "csrc", "libniftyk.so")

NK_OK, NK_ERR_INVALID, NK_ERR_UNSUPPORTED, NK_ERR_RUNTIME, NK_ERR_NOMEM = 0, -1, -2, -3, -4
NK_F32, NK_F64 = 0, 1
PRO_PLAIN, PRO_AMP, PRO_AMP_JVP, PRO_MUL = 0, 1, 2, 3
EPI_AFFINE, EPI_MUL, EPI_VJP, EPI_LIKELIHOOD, EPI_NONLIN = 0, 1, 2, 3, 4
LH_GAUSS, LH_POISSON = 0, 1
NL_ID, NL_EXP, NL_SIGMOID = 0, 1, 2
OP_ADD, OP_SUB, OP_MUL, OP_DIV = 0, 1, 2, 3

_vp, _i, _i64, _d, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_size_t


class Fuse(ctypes.Structure):
    """Mirror of ``struct nk_fuse`` (include/niftyk.h)."""

    _fields_ = [
        ("pro", _i), ("in_", _vp), ("in2", _vp), ("pidx", _vp), ("amp", _vp), ("damp", _vp),
        ("epi", _i), ("out", _vp), ("scale", _d), ("offset", _d), ("mul", _vp), ("mul_scalar", _d),
        ("xi", _vp), ("addend", _vp), ("addend_scale", _d), ("accumulate", _i), ("abar", _vp), ("lh_kind", _i), ("nonlin", _i), ("data", _vp),
        ("icov", _vp), ("icov_scalar", _d), ("out2", _vp), ("value", _vp),
        ("afield", _vp), ("dampT", _vp), ("abar_copies", _i), ("abar_stride", _i64), ("dafield", _vp), ("w8", _vp),
        ("field_octant", _i), ("value_slots", _i), ("pidx_octant", _vp), ("cg_r", _vp), ("cg_scal", _vp), ("w8max", _vp),
        ("pipe_chunks", _i), ("pipe_wait", _vp), ("pipe_record", _vp), ("wfull", _vp), ("io32", _i), ("carry1", _vp), ("carry2", _vp),
    ]


class Product(ctypes.Structure):
    """Mirror of ``struct nk_product`` (include/niftyk.h)."""

    _fields_ = [("nsub", _i), ("size", _i64 * 3), ("pidx", _vp * 3), ("tab", _vp * 3), ("dtab", _vp * 3), ("scale", _vp),
                ("dscale", _vp)]


class TiledCsr(ctypes.Structure):
    """Mirror of ``struct nk_tiled_csr`` (include/niftyk.h)."""

    _fields_ = [("n_rows", _i64), ("n_slots", _i64), ("n_items", ctypes.c_int32), ("ny", ctypes.c_int32), ("nx", ctypes.c_int32),
                ("th", ctypes.c_int32), ("tw", ctypes.c_int32), ("item_tile", _vp), ("item_blk", _vp), ("blk_slot", _vp),
                ("row_slot", _vp), ("loc", _vp), ("wgt", _vp)]


# name -> (restype, argtypes); the list is checked against include/niftyk.h by tests/test_abi.py
SIGNATURES = {
    "nk_last_error": (ctypes.c_char_p, []),
    "nk_version": (_i, []),
    "nk_plan_create": (_i, [ctypes.POINTER(_vp), _i, ctypes.POINTER(_i64), _i, _i64]),
    "nk_plan_destroy": (_i, [_vp]),
    "nk_plan_workspace_bytes": (_sz, [_vp]),
    "nk_hartley": (_i, [_vp, _vp, _vp, _d, _i, _vp, _vp]),
    "nk_hartley_fused": (_i, [_vp, ctypes.POINTER(Fuse), _i, _vp, _vp]),
    "nk_plan_sandwich": (_i, [_vp]),
    "nk_plan_pipe_ok": (_i, [_vp, _i]),
    "nk_hartley_sandwich": (_i, [_vp, ctypes.POINTER(Fuse), _d, _i, _vp, _vp]),
    "nk_hartley_sandwich_pair": (_i, [_vp, ctypes.POINTER(Fuse), ctypes.POINTER(Fuse), _d, _i, _vp, _vp, _vp]),
    "nk_fftn": (_i, [_vp, _vp, _vp, _i, _d, _vp, _vp]),
    "nk_profile_enable": (_i, [_i]),
    "nk_profile_collect": (_i, [_vp, _vp]),
    "nk_vdot": (_i, [_i64, _vp, _vp, _i, _vp, _i, _vp]),
    "nk_product_field": (_i, [ctypes.POINTER(Product), _i, _vp, _i, _vp]),
    "nk_mirror_combine": (_i, [_i, ctypes.POINTER(_i64), ctypes.POINTER(_i), _i, ctypes.POINTER(_d), _vp, _vp, _d, _d, _i, _vp]),
    "nk_product_marginal_scratch": (_sz, [ctypes.POINTER(Product), _i]),
    "nk_product_marginal": (_i, [ctypes.POINTER(Product), _i, _vp, _vp, _vp, _vp]),
    "nk_red_unit": (_i64, [_i64, _i]),
    "nk_red_layout": (_i, [_i64, _i, _i, _i, _i, _i, _vp]),
    "nk_red_finish": (_i, [_vp, _i, _i, _vp, _i, _vp]),
    "nk_sum": (_i, [_i64, _vp, _i, _vp, _i, _vp]),
    "nk_stats": (_i, [_i64, _vp, _i, _vp, _vp]),
    "nk_binary": (_i, [_i, _i64, _vp, _d, _vp, _d, _vp, _i, _vp]),
    "nk_cplx_muldiv": (_i, [_i64, _vp, _i, _d, _d, _vp, _i, _d, _d, _i, _i, _vp, _i, _vp]),
    "nk_cplx_pointwise": (_i, [_i, _i64, _vp, _vp, _i, _vp]),
    "nk_axpby": (_i, [_i64, _d, _vp, _d, _vp, _vp, _i, _vp]),
    "nk_axpby_sqnorm": (_i, [_i64, _d, _vp, _d, _vp, _vp, _i, _vp, _i, _vp]),
    "nk_pointwise": (_i, [_i, _d, _i64, _vp, _vp, _vp, _i, _vp]),
    "nk_clip": (_i, [_d, _d, _i64, _vp, _vp, _vp, _i, _vp]),
    "nk_gather": (_i, [_i64, _vp, _vp, _vp, _i, _vp]),
    "nk_scatter_add": (_i, [_i64, _vp, _vp, _i64, _vp, _i, _vp]),
    "nk_octant_expand": (_i, [_i, ctypes.POINTER(_i64), _vp, _vp, _vp, _i, _i, _vp]),
    "nk_octant_scatter": (_i, [_i, ctypes.POINTER(_i64), _vp, _vp, _vp, _i, _vp]),
    "nk_octant_expand_k2": (_i, [_i, ctypes.POINTER(_i64), _vp, _vp, _i64, _vp, _vp, _i, _vp, _vp]),
    "nk_segment_sum": (_i, [_i64, _vp, _vp, _vp, _vp, _i, _vp]),
    "nk_octant_scatter_k2": (_i, [_i, ctypes.POINTER(_i64), _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "nk_plan_octant_vjp": (_i, [_vp]),
    "nk_fold_copies": (_i, [_i64, _i, _i64, _vp, _vp, _vp]),
    "nk_cumsum": (_i, [_i64, _vp, _vp, _i, _i, _vp]),
    "nk_cplx_rows": (_i, [_i64, _i64, _i64, _vp, _vp, _vp, _i, _d, _i, _i, _vp]),
    "nk_csr_rowsum": (_i, [_i64, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "nk_tiled_rowsum": (_i, [ctypes.POINTER(TiledCsr), _i, _vp, _vp, _vp, _i, _vp]),
    "nk_bluestein_rows": (_i, [_i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _d, _i, _i, _i, _vp]),
    "nk_roll": (_i, [_i, ctypes.POINTER(_i64), ctypes.POINTER(_i64), _i, _vp, _vp, _vp]),
    "nk_spmv": (_i, [_i64, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "nk_spmv_t": (_i, [_i64, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "nk_pindex_from_k2": (_i, [_i, ctypes.POINTER(_i64), _vp, _vp, _vp, _vp]),
    "nk_cg_curv": (_i, [_i64, _vp, _vp, _i, _vp, _i, _vp]),
    "nk_cg_update": (_i, [_i64, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    "nk_cg_update_dr": (_i, [_i64, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    "nk_pcg64_normal_scratch_bytes": (_i64, [_i64, _i]),
    "nk_pcg64_normal": (_i, [_vp, _vp, _i64, _d, _d, _vp, _i, _vp, _i64, _i, _vp, _vp]),
    "nk_pcg64_fixed_scratch_bytes": (_i64, []),
    "nk_pcg64_uniform": (_i, [_vp, _vp, _i64, _d, _d, _vp, _i, _vp, _vp]),
    "nk_pcg64_pm1": (_i, [_vp, _vp, _i64, _vp, _i, _i, _vp, _vp]),
    "nk_pcg64_integers_scratch_bytes": (_i64, [_i64]),
    "nk_pcg64_integers": (_i, [_vp, _vp, _i64, _i64, ctypes.c_uint64, _i64, _vp, _vp, _vp, _vp]),
    "nk_cg_direction": (_i, [_i64, _vp, _vp, _i, _vp, _i, _vp]),
    "nk_amp_forward": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "nk_amp_jvp": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "nk_amp_vjp": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    # batched launches: per-member arguments are host arrays of `count` device pointers (ptr_array)
    "nk_plan_batch_ok": (_i, [_vp]),
    "nk_hartley_fused_batch": (_i, [_vp, ctypes.POINTER(Fuse), _i, _i, _vp, _vp]),
    "nk_amp_forward_batch": (_i, [_i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "nk_amp_jvp_batch": (_i, [_i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "nk_amp_vjp_batch": (_i, [_i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "nk_gather_batch": (_i, [_i64, _i, _vp, _vp, _vp, _i, _vp]),
    "nk_csr_rowsum_batch": (_i, [_i64, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _vp]),
    "nk_axpby_batch": (_i, [_i64, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "nk_axpby_sqnorm_batch": (_i, [_i64, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    "nk_binary_batch": (_i, [_i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "nk_vdot_batch": (_i, [_i64, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "nk_sum_tree": (_i, [_i64, _i, _vp, _vp, _i, _vp]),
    "nk_cg_curv_batch": (_i, [_i64, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "nk_cg_update_batch": (_i, [_i64, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    "nk_cg_update_dr_batch": (_i, [_i64, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    "nk_cg_direction_batch": (_i, [_i64, _i, _vp, _vp, _i, _vp, _i, _vp]),
}
MAX_BATCH = 8  # NK_MAX_BATCH of include/niftyk.h


def ptr_array(tensors):
    """Host array of device pointers for a `*_batch` entry point (None -> NULL)."""
    return (_vp * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def double_array(values):
    return (_d * len(values))(*[float(v) for v in values])

_lib = None


class NiftykError(RuntimeError):
    pass


def load():
    """Load libniftyk.so (once).  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NiftykError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). nifty_amd has no CPU fallback for device fields."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = ABI mismatch, fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    """Translate an nk_status into the reference's exception types (SURVEY 8b 'Errors')."""
    if rc == NK_OK:
        return
    msg = load().nk_last_error().decode(errors="replace")
    text = f"{what}: {msg}" if what else msg
    if rc == NK_ERR_INVALID:
        raise ValueError(text)
    if rc == NK_ERR_UNSUPPORTED:
        raise NotImplementedError(text)
    if rc == NK_ERR_NOMEM:
        raise MemoryError(text)
    raise NiftykError(text)
