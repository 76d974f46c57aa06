"""ctypes binding of the C ABI in include/nlsh_hip.h (lib/libnlsh_hip.so, built for gfx950).

The product path has NO fallback: if the library is missing or a call fails, an exception is
raised (`NlshHipError`).  Nothing here imports or calls the CPU oracle.
"""
import ctypes
import os
import subprocess

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("NLSH_HIP_LIB") or os.path.join(_PKG, "lib", "libnlsh_hip.so")   # override: diagnostic builds only
CSRC = os.path.join(_PKG, "csrc")

OK, E_INVALID, E_UNSUPPORTED, E_HIP, E_WORKSPACE = 0, -1, -2, -3, -4
ACT_SIGMOID, ACT_TANH = 0, 1
KEY_REF_INT16, KEY_FULL = 0, 1
METRIC_L2_EPS, METRIC_COSINE, METRIC_L2_EPS_FOLDED = 0, 1, 2
SCAN_QUERY_MAJOR, SCAN_BUCKET_MAJOR, SCAN_BUCKET_TILED = 0, 1, 2
MAX_LAYERS, MAX_HASH_BITS, MAX_PROBES, MAX_K, MAX_DIM, MAX_WIDTH = 8, 32, 64, 64, 1024, 632
PHASE_PLAN, PHASE_SCAN, PHASE_MERGE = 1, 2, 4
PHASE_ALL = 7
MAX_ENCODE_PROBES = 128  # nlsh_encode_hash generates up to this many keys per row; the scan takes them in slices of MAX_PROBES

# every symbol include/nlsh_hip.h declares (tests/test_host_cpu.py::test_capi_library_exports_every_declared_symbol checks the header against this)
SYMBOLS = (
    "nlsh_abi_version", "nlsh_last_error",
    "nlsh_encoder_packed_floats", "nlsh_encoder_pack", "nlsh_encode_hash", "nlsh_pack_codes",
    "nlsh_build_csr_workspace", "nlsh_build_csr", "nlsh_bucket_order_workspace", "nlsh_bucket_order", "nlsh_build_cells_workspace", "nlsh_build_cells", "nlsh_gather_rows",
    "nlsh_scan_workspace", "nlsh_scan_workspace_layout", "nlsh_scan_topk", "nlsh_scan_topk_phase", "nlsh_scan_topk_cells_phase",
    "nlsh_merge_topk",
    "nlsh_step_create", "nlsh_step_create_graph", "nlsh_step_destroy", "nlsh_step_set_weights", "nlsh_query_step_enqueue", "nlsh_step_release", "nlsh_step_busy", "nlsh_query_batch",
)


class NlshHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"nlsh_hip error {code}: {msg}")
        self.code = code


_lib = None
vp, i32, i64, u64, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint64, ctypes.c_size_t


class StepDesc(ctypes.Structure):
    """`nlsh_step_desc_t` of include/nlsh_hip.h, field for field (nlsh_step_create checks sizeof against the library's)."""
    _fields_ = [
        ("n_layers", ctypes.c_int32), ("act", ctypes.c_int32), ("key_mode", ctypes.c_int32), ("n_probes", ctypes.c_int32),
        ("dims", vp), ("packed", vp), ("n_multi_rows", i64),
        ("corpus_sorted", vp), ("row_stride", i64),
        ("gid", vp), ("uniq_keys", vp), ("offsets", vp), ("bucket_order", vp), ("cell_of", vp), ("cell_offsets", vp), ("inv_norm", vp),
        ("d", ctypes.c_int32), ("n_buckets", ctypes.c_int32), ("n_cells", ctypes.c_int32), ("k", ctypes.c_int32),
        ("metric", ctypes.c_int32), ("algo", ctypes.c_int32), ("seg_rows", ctypes.c_int32), ("hold_done", ctypes.c_int32),
        ("Q", i64), ("qkeys", vp), ("nkeys", vp), ("out_dist", vp), ("out_idx", vp), ("out_keys", vp), ("out_ncand", vp), ("status", vp),
        ("workspace", vp), ("workspace_bytes", sz), ("max_tasks", i64),
        ("front", vp), ("plan", vp), ("mid", vp), ("tail", vp),
    ]


def build_library(force=False):
    """Compile csrc/*.hip for gfx950 (hipcc cross-compiles without a GPU).  The optional list-construction helper is built for THIS
    interpreter (extension suffix and include directory from sysconfig), not for whichever python3-config is on PATH."""
    import sysconfig
    args = ["make", "-C", CSRC, "-j4"]
    suffix, inc = sysconfig.get_config_var("EXT_SUFFIX"), sysconfig.get_paths().get("include")
    if suffix and inc and os.path.exists(os.path.join(inc, "Python.h")):
        args += [f"PYSUFFIX={suffix}", f"PYINCLUDES=-I{inc}"]
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"])
    subprocess.check_call(args)
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NlshHipError(E_HIP, f"{LIB_PATH} not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
                                  f"or `make -C {CSRC}`); there is no CPU fallback")
    L = ctypes.CDLL(LIB_PATH)
    L.nlsh_abi_version.restype = i32
    L.nlsh_last_error.restype = ctypes.c_char_p
    L.nlsh_encoder_packed_floats.restype = i64
    L.nlsh_encoder_packed_floats.argtypes = [i32, vp]
    L.nlsh_encoder_pack.restype = i32
    L.nlsh_encoder_pack.argtypes = [i32, vp, vp, vp, vp, vp]
    L.nlsh_encode_hash.restype = i32
    L.nlsh_encode_hash.argtypes = [vp, i64, i64, i32, vp, vp, i32, i32, i32, i64, u64, i64, vp, vp, vp, vp, vp, vp]
    L.nlsh_pack_codes.restype = i32
    L.nlsh_pack_codes.argtypes = [vp, i64, i32, i32, i32, vp, vp]
    L.nlsh_build_csr_workspace.restype = sz
    L.nlsh_build_csr_workspace.argtypes = [i64]
    L.nlsh_build_csr.restype = i32
    L.nlsh_build_csr.argtypes = [vp, i64, vp, vp, vp, vp, vp, sz, vp]
    L.nlsh_bucket_order_workspace.restype = sz
    L.nlsh_bucket_order_workspace.argtypes = [i64]
    L.nlsh_bucket_order.restype = i32
    L.nlsh_bucket_order.argtypes = [vp, i64, vp, vp, sz, vp]
    L.nlsh_build_cells_workspace.restype = sz
    L.nlsh_build_cells_workspace.argtypes = [i64]
    L.nlsh_build_cells.restype = i32
    L.nlsh_build_cells.argtypes = [vp, i64, i32, vp, vp, vp, vp, vp, sz, vp]
    L.nlsh_gather_rows.restype = i32
    L.nlsh_gather_rows.argtypes = [vp, i64, i32, vp, i64, vp, i64, vp, vp, ctypes.c_int32, vp]
    L.nlsh_scan_workspace.restype = sz
    L.nlsh_scan_workspace.argtypes = [i64, i32, i32, i64, i64, i32]
    L.nlsh_scan_workspace_layout.restype = i32
    L.nlsh_scan_workspace_layout.argtypes = [i64, i32, i32, i64, i64, i32, i32, vp, vp, vp]
    L.nlsh_scan_topk.restype = i32
    L.nlsh_scan_topk.argtypes = [vp, i64, i32, vp, vp, vp, vp, ctypes.c_int32, vp, vp, i64, i64, vp, vp, i32, i32, i32, i32, i32,
                                 vp, vp, vp, vp, vp, vp, sz, i64, vp, vp, vp]
    L.nlsh_scan_topk_phase.restype = i32
    L.nlsh_scan_topk_phase.argtypes = list(L.nlsh_scan_topk.argtypes) + [i32]
    L.nlsh_scan_topk_cells_phase.restype = i32     # cell_of, cell_offsets, n_cells follow n_buckets
    L.nlsh_scan_topk_cells_phase.argtypes = L.nlsh_scan_topk_phase.argtypes[:8] + [vp, vp, ctypes.c_int32] + L.nlsh_scan_topk_phase.argtypes[8:]
    L.nlsh_merge_topk.restype = i32
    L.nlsh_merge_topk.argtypes = [vp, i64, i32, i64, i32, vp, vp, vp, vp, vp]
    L.nlsh_step_create.restype = i32
    L.nlsh_step_create.argtypes = [ctypes.POINTER(StepDesc), sz, ctypes.POINTER(vp)]
    L.nlsh_step_create_graph.restype = i32
    L.nlsh_step_create_graph.argtypes = [ctypes.POINTER(StepDesc), sz, vp, ctypes.POINTER(vp)]
    for name in ("nlsh_step_destroy", "nlsh_step_release", "nlsh_step_busy"):
        getattr(L, name).restype = i32
        getattr(L, name).argtypes = [vp]
    L.nlsh_step_set_weights.restype = i32
    L.nlsh_step_set_weights.argtypes = [vp, vp]
    L.nlsh_query_step_enqueue.restype = i32
    L.nlsh_query_step_enqueue.argtypes = [vp, vp, i64, u64, vp, vp, vp]
    L.nlsh_query_batch.restype = i32
    L.nlsh_query_batch.argtypes = [ctypes.POINTER(StepDesc), sz, vp, i64, u64, i64, i32, vp, vp, vp]
    _lib = L
    return L


def check(rc):
    if rc != OK:
        raise NlshHipError(rc, lib().nlsh_last_error().decode("utf-8", "replace"))


def ptr(t):
    """Device (or host) pointer of a torch tensor / None."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def int_array(values):
    return (ctypes.c_int * len(values))(*[int(v) for v in values])


def ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])
