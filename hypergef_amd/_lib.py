"""ctypes binding of libhgaggr.so (the C ABI in include/hg_aggr.h).

There is no fallback: if the shared library is missing or a call fails, the
caller gets an exception.  Nothing here computes on the CPU.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HG_AGGR_LIB: another build of the same library (A/B measurements on one box)
LIB_PATH = os.environ.get("HG_AGGR_LIB") or os.path.join(_HERE, "lib", "libhgaggr.so")

HG_OK = 0
HG_ERR_WORKSPACE = -4  # include/hg_aggr.h: caller workspace smaller than hg_plan_workspace_bytes
HG_VARIANT_AUTO = 0
HG_VARIANT_PULL = 1
HG_VARIANT_PUSH_ATOMIC = 2
HG_VARIANT_FUSED = 3
HG_PLAN_HOST_ONLY = 1
HG_PLAN_NO_XCD_REMAP = 2
HG_PLAN_DFS_ORDER = 4
HG_PLAN_NO_HUB_PASS = 8
HG_PLAN_NO_ROW_STREAM = 16

VARIANTS = {"auto": HG_VARIANT_AUTO, "pull": HG_VARIANT_PULL, "push_atomic": HG_VARIANT_PUSH_ATOMIC,
            "fused": HG_VARIANT_FUSED}

# every symbol include/hg_aggr.h declares (tests check the library exports all)
SYMBOLS = (
    "hg_version", "hg_last_error", "hg_status_string", "hg_balance_schedule", "hg_mtx_read", "hg_free",
    "hg_plan_create_host", "hg_plan_create_device", "hg_plan_destroy", "hg_plan_get_info",
    "hg_plan_get_vertex_csr", "hg_plan_get_vertex_csr_device", "hg_plan_get_schedule", "hg_plan_prepare", "hg_plan_auto_variant", "hg_plan_bind_scales", "hg_plan_tune_f32",
    "hg_plan_workspace_bytes",
    "hg_aggr_fused_f32", "hg_linear_pack_f32", "hg_linear_pack_ex_f32", "hg_linear_pack_floats", "hg_linear_rows_f32", "hg_linear_wgrad_workspace_bytes", "hg_linear_wgrad_f32", "hg_aggr_linear_workspace_bytes", "hg_aggr_linear_f32", "hg_aggr_linear_res_f32", "hg_aggr_linear_res_dev_f32",
    "hg_gather_rows_f32", "hg_aggr_push_groups_f32", "hg_gather_max_f32",
    "hg_scatter_record_f32",
)


class HgError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("libhgaggr: %s (status %d)" % (message, status))
        self.status = status


class PlanOpts(ctypes.Structure):
    _fields_ = [("short_max", ctypes.c_int32), ("split_len", ctypes.c_int32),
                ("panel_rows", ctypes.c_int32), ("panel_nnz", ctypes.c_int32),
                ("flags", ctypes.c_int32), ("t_big", ctypes.c_int32),
                ("fused_tile_bytes", ctypes.c_int32), ("fused_steps", ctypes.c_int32)]


class TuneInfo(ctypes.Structure):
    _fields_ = [("variant", ctypes.c_int32), ("pull_hop_kernels", ctypes.c_int32), ("us", ctypes.c_float * 10),
                ("reserved", ctypes.c_int32)]


class FusedInfo(ctypes.Structure):
    _fields_ = [("cap", ctypes.c_int32), ("t_big", ctypes.c_int32), ("vdeg_max", ctypes.c_int32),
                ("panels", ctypes.c_int32), ("n_mat", ctypes.c_int32), ("n_hub", ctypes.c_int32),
                ("slots", ctypes.c_int64), ("member_entries", ctypes.c_int64),
                ("n_split", ctypes.c_int32), ("fixups", ctypes.c_int32),
                ("hub_rounds", ctypes.c_int32), ("hub_workgroups", ctypes.c_int32),
                ("hub_entries", ctypes.c_int64), ("hub_pairs", ctypes.c_int64),
                ("partial_rows", ctypes.c_int64),
                ("record_words_max", ctypes.c_int32), ("stream_steps_max", ctypes.c_int32),
                ("lds_bytes", ctypes.c_int32), ("reserved", ctypes.c_int32)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class PlanInfo(ctypes.Structure):
    _fields_ = [("N", ctypes.c_int32), ("M", ctypes.c_int32), ("nnz", ctypes.c_int64),
                ("short_max", ctypes.c_int32), ("split_len", ctypes.c_int32),
                ("panel_rows", ctypes.c_int32), ("panel_nnz", ctypes.c_int32),
                ("flags", ctypes.c_int32),
                ("panels", ctypes.c_int32 * 2), ("tasks", ctypes.c_int32 * 2),
                ("partials", ctypes.c_int32 * 2), ("fixups", ctypes.c_int32 * 2),
                ("max_len", ctypes.c_int32 * 2), ("device_bytes", ctypes.c_int64)]

    def as_dict(self):
        out = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            out[name] = list(v) if hasattr(v, "__len__") else v
        return out


_lib = None


def lib():
    """Load libhgaggr.so once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C hypergef_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t
    L.hg_version.restype = ctypes.c_int
    L.hg_last_error.restype = ctypes.c_char_p
    L.hg_status_string.restype = ctypes.c_char_p
    L.hg_status_string.argtypes = [ctypes.c_int]
    L.hg_balance_schedule.restype = ctypes.c_int
    L.hg_balance_schedule.argtypes = [i32, i32, vp, ctypes.POINTER(i64), ctypes.POINTER(i64),
                                      vp, vp, vp, vp]
    L.hg_mtx_read.restype = ctypes.c_int
    L.hg_mtx_read.argtypes = [ctypes.c_char_p, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i64),
                              ctypes.POINTER(ctypes.POINTER(i32)), ctypes.POINTER(ctypes.POINTER(i32)),
                              ctypes.POINTER(ctypes.POINTER(i32)), ctypes.POINTER(ctypes.POINTER(i32))]
    L.hg_free.restype = None
    L.hg_free.argtypes = [vp]
    L.hg_plan_create_host.restype = ctypes.c_int
    L.hg_plan_create_host.argtypes = [ctypes.POINTER(vp), i32, i32, vp, vp, ctypes.POINTER(PlanOpts)]
    L.hg_plan_create_device.restype = ctypes.c_int
    L.hg_plan_create_device.argtypes = [ctypes.POINTER(vp), i32, i32, i64, vp, vp,
                                        ctypes.POINTER(PlanOpts), vp]
    L.hg_plan_destroy.restype = None
    L.hg_plan_destroy.argtypes = [vp]
    L.hg_plan_get_info.restype = ctypes.c_int
    L.hg_plan_get_info.argtypes = [vp, ctypes.POINTER(PlanInfo)]
    L.hg_plan_get_vertex_csr.restype = ctypes.c_int
    L.hg_plan_get_vertex_csr.argtypes = [vp, vp, vp]
    L.hg_plan_get_vertex_csr_device.restype = ctypes.c_int
    L.hg_plan_get_vertex_csr_device.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.hg_plan_get_schedule.restype = ctypes.c_int
    L.hg_plan_get_schedule.argtypes = [vp, i32, vp, vp, vp]
    L.hg_plan_prepare.restype = ctypes.c_int
    L.hg_plan_prepare.argtypes = [vp, i32, ctypes.POINTER(FusedInfo)]
    L.hg_plan_tune_f32.restype = ctypes.c_int
    L.hg_plan_tune_f32.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, sz, i32, vp, ctypes.POINTER(TuneInfo)]
    L.hg_linear_pack_f32.restype = ctypes.c_int
    L.hg_linear_pack_f32.argtypes = [i32, i32, vp, vp, vp]
    L.hg_linear_pack_ex_f32.restype = ctypes.c_int
    L.hg_linear_pack_ex_f32.argtypes = [i32, i32, vp, vp, i32, vp]
    L.hg_linear_pack_floats.restype = sz
    L.hg_linear_pack_floats.argtypes = [i32, i32, i32]
    L.hg_linear_wgrad_workspace_bytes.restype = sz
    L.hg_linear_wgrad_workspace_bytes.argtypes = [i64, i32, i32]
    L.hg_linear_wgrad_f32.restype = ctypes.c_int
    L.hg_linear_wgrad_f32.argtypes = [i64, i32, i32, vp, vp, vp, vp, sz, vp]
    L.hg_linear_rows_f32.restype = ctypes.c_int
    L.hg_linear_rows_f32.argtypes = [i64, i32, i32, vp, vp, vp, vp]
    L.hg_aggr_linear_workspace_bytes.restype = sz
    L.hg_aggr_linear_workspace_bytes.argtypes = [vp, i32]
    L.hg_aggr_linear_f32.restype = ctypes.c_int
    L.hg_aggr_linear_f32.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i32, vp]
    L.hg_aggr_linear_res_f32.restype = ctypes.c_int
    L.hg_aggr_linear_res_f32.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float,
                                         i32, vp, vp, vp, sz, i32, vp]
    L.hg_aggr_linear_res_dev_f32.restype = ctypes.c_int
    L.hg_aggr_linear_res_dev_f32.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_float, vp,
                                             i32, vp, vp, vp, sz, i32, vp]
    L.hg_plan_bind_scales.restype = ctypes.c_int
    L.hg_plan_bind_scales.argtypes = [vp, i32, vp, vp, vp, vp]
    L.hg_plan_auto_variant.restype = ctypes.c_int
    L.hg_plan_auto_variant.argtypes = [vp, i32]
    L.hg_plan_workspace_bytes.restype = sz
    L.hg_plan_workspace_bytes.argtypes = [vp, i32]
    L.hg_aggr_fused_f32.restype = ctypes.c_int
    L.hg_aggr_fused_f32.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, sz, i32, vp]
    L.hg_gather_rows_f32.restype = ctypes.c_int
    L.hg_gather_rows_f32.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.hg_gather_max_f32.restype = ctypes.c_int
    L.hg_gather_max_f32.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.hg_scatter_record_f32.restype = ctypes.c_int
    L.hg_scatter_record_f32.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp]
    L.hg_aggr_push_groups_f32.restype = ctypes.c_int
    L.hg_aggr_push_groups_f32.argtypes = [i32, i32, i32, i64, vp, vp, vp, vp, vp, vp,
                                          vp, vp, vp, vp, vp, vp]
    _lib = L
    return L


def check(status):
    if status != HG_OK:
        msg = lib().hg_last_error()
        raise HgError(status, (msg or b"").decode() or lib().hg_status_string(status).decode())
