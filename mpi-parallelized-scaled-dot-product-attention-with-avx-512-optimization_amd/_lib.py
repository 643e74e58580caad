"""ctypes binding of lib/libsdpa_hip.so -- the C ABI of include/sdpa_hip.h.

Nothing in here computes: it loads the shared library built from csrc/ and
declares the prototypes.  There is no fallback of any kind: if the library is
missing `load()` raises, and on a machine without a gfx950 device every compute
entry point returns SDPA_ENODEV, which `check()` turns into an exception.
"""
import ctypes
import os
import re

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
# $SDPA_HIP_LIB: a differently built copy of the same library (tools/build_variant.sh builds tuning
# variants into lib/variants/ for A/B timing); the product is lib/libsdpa_hip.so
LIB_PATH = os.environ.get("SDPA_HIP_LIB") or os.path.join(PKG_DIR, "lib", "libsdpa_hip.so")
HEADER_PATH = os.path.join(REPO_DIR, "include", "sdpa_hip.h")

SDPA_F_NO_PIPELINE, SDPA_F_BF16, SDPA_F_PLAN_QROWS, SDPA_F_MERGE_ALLREDUCE = 1, 2, 4, 8
SDPA_OK, SDPA_EINVAL, SDPA_ENODEV, SDPA_EHIP, SDPA_ERCCL, SDPA_ENOMEM, SDPA_EUNSUP = 0, -1, -2, -3, -4, -5, -6

_c_int, _c_long, _c_size_t, _c_void_p = ctypes.c_int, ctypes.c_long, ctypes.c_size_t, ctypes.c_void_p


class SdpaTiming(ctypes.Structure):
    _fields_ = [("total_us", ctypes.c_double), ("kv_stage_us", ctypes.c_double),
                ("pipeline_us", ctypes.c_double), ("kernel_us", ctypes.c_double),
                ("n_gpus", _c_int), ("q_batches", _c_int), ("kv_splits", _c_int),
                ("register_us", ctypes.c_double), ("head_us", ctypes.c_double), ("tail_us", ctypes.c_double),
                ("kv_chunks", _c_int), ("fused_launches", _c_int), ("plan", _c_int), ("merge", _c_int),
                ("virtual_ranks", _c_int),
                ("enqueue_total_us", ctypes.c_double), ("enqueue_first_kernel_us", ctypes.c_double * 16),
                ("egress", _c_int), ("enqueue_threads", _c_int), ("host_convert_threads", _c_int),
                # ABI 4
                ("compute_cus", _c_int), ("stream_k", _c_int), ("host_widen", _c_int), ("rccl_selftest", _c_int),
                ("merge_us", ctypes.c_double), ("reduce_us", ctypes.c_double), ("egress_us", ctypes.c_double),
                # ABI 5
                ("last_kernel", ctypes.c_char * 96), ("last_grid", _c_int), ("streamed", _c_int),
                ("host_convert_node", _c_int)]


SDPA_ABI_VERSION = 6          # the SDPA_ABI_VERSION of include/sdpa_hip.h this binding was written against


class SdpaError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__("%s failed: %s (code %d)" % (what, strerror(code), code))


_PROTOS = {
    "sdpa_init": (_c_int, [_c_int]),
    "sdpa_shutdown": (None, []),
    "sdpa_init_default": (_c_int, []),
    "sdpa_engine_ranks": (_c_int, []),
    "sdpa_device_count": (_c_int, []),
    "sdpa_strerror": (ctypes.c_char_p, [_c_int]),
    "sdpa_version": (ctypes.c_char_p, []),
    "sdpa_attention_f64": (_c_int, [_c_void_p] * 4 + [_c_int] * 5),
    "sdpa_last_timing": (_c_int, [ctypes.POINTER(SdpaTiming)]),
    "sdpa_last_timing_sized": (_c_int, [ctypes.POINTER(SdpaTiming), ctypes.c_size_t]),
    "sdpa_abi_version": (_c_int, []),
    "sdpa_reload_env": (None, []),
    "sdpa_prepare": (_c_int, [_c_int] * 5),
    "sdpa_plan_describe": (_c_int, [_c_int] * 6 + [ctypes.c_char_p, ctypes.c_size_t]),
    "sdpa_kv_prefetch": (_c_int, [_c_void_p, _c_void_p] + [_c_int] * 7),
    "sdpa_host_alloc": (_c_void_p, [ctypes.c_size_t]),
    "sdpa_host_free": (None, [_c_void_p]),
    "sdpa_host_declare_pinned": (_c_int, [_c_void_p, _c_size_t]),
    "sdpa_host_forget_pinned": (_c_int, [_c_void_p]),
    "sdpa_host_cvt_rows": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_int, ctypes.c_double, _c_int]),
    "sdpa_host_cvt_vt": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_long, _c_int, _c_int, _c_long, _c_int, _c_int]),
    "sdpa_host_widen": (_c_int, [_c_void_p, _c_void_p, ctypes.c_size_t, _c_int, _c_int]),
    "sdpa_owner_count": (_c_int, [_c_int] * 3),
    "sdpa_owner_disp": (_c_int, [_c_int] * 3),
    "sdpa_dev_stream_create": (_c_int, [_c_int, ctypes.POINTER(_c_void_p)]),
    "sdpa_dev_stream_destroy": (_c_int, [_c_void_p]),
    "sdpa_dev_last_launch": (_c_int, [ctypes.c_char_p, ctypes.c_size_t]),
    "sdpa_dev_dense_ld": (_c_int, [_c_int]),
    "sdpa_dev_cvt_d2f": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_void_p]),
    "sdpa_dev_cvt_f2d": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_long, _c_int, _c_void_p]),
    "sdpa_dev_kv_splits": (_c_int, [_c_int] * 4),
    "sdpa_dev_workspace_bytes": (_c_size_t, [_c_int] * 4),
    "sdpa_dev_shard_partial_f32": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int,
                                            _c_void_p, _c_int, _c_void_p, _c_void_p,
                                            _c_int, _c_int, _c_int, _c_int,
                                            _c_void_p, _c_size_t, _c_void_p]),
    "sdpa_dev_cvt_d2f_batch": (_c_int, [_c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "sdpa_dev_shard_attention_f64": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int,
                                              _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p,
                                              _c_int, _c_int, _c_int, _c_int,
                                              _c_void_p, _c_size_t, _c_void_p]),
    "sdpa_dev_merge_rescale": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p,
                                        _c_int, _c_int, _c_void_p]),
    "sdpa_dev_merge_normalise": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_void_p]),
    "sdpa_dev_merge_gathered": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p]),
    "sdpa_dev_finish_f64": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_int, _c_int, _c_void_p]),
    "sdpa_dev_bf16_ld": (_c_int, [_c_int]),
    "sdpa_dev_bf16_dvp": (_c_int, [_c_int]),
    "sdpa_dev_bf16_ldn": (_c_long, [_c_long]),
    "sdpa_dev_bf16_kvpos": (_c_long, [_c_long]),
    "sdpa_dev_cvt_d2bf": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_void_p]),
    "sdpa_dev_cvt_d2bf_q": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_void_p]),
    "sdpa_dev_cvt_d2bf_t": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_long, _c_void_p]),
    "sdpa_dev_bf16_tiled": (_c_int, [_c_int]),
    "sdpa_dev_cvt_d2bf_k": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_void_p]),
    "sdpa_dev_cvt_d2bf_v": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_int, _c_void_p]),
    "sdpa_dev_kv_splits_bf16": (_c_int, [_c_int] * 4),
    "sdpa_dev_workspace_bytes_bf16": (_c_size_t, [_c_int] * 4),
    "sdpa_dev_shard_partial_bf16": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_long,
                                             _c_void_p, _c_int, _c_void_p, _c_void_p,
                                             _c_int, _c_int, _c_int, _c_int,
                                             _c_void_p, _c_size_t, _c_void_p]),
}

_lib = None


def header_symbols():
    """Every entry point include/sdpa_hip.h declares (SDPA_API ... name(...))."""
    text = open(HEADER_PATH).read()
    return sorted(set(re.findall(r"SDPA_API\s+[\w\s\*]+?\b(sdpa_\w+)\s*\(", text)))


def load():
    """Load the library (once).  PyTorch is imported first when available so that its
    libamdhip64.so.7 is the one HIP runtime of the process (same SONAME as ROCm's)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `make -C %s` (hipcc, gfx950). "
                          "There is no CPU fallback." % (LIB_PATH, os.path.join(PKG_DIR, "csrc")))
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        if os.environ.get("SDPA_HIP_LIB") and not hasattr(lib, name):
            continue          # an older build loaded for an A/B (tools/): it simply lacks the newer entry points
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if not os.environ.get("SDPA_HIP_LIB"):
        got = lib.sdpa_abi_version()
        if got != SDPA_ABI_VERSION:      # a stale build: caller-allocated structs would not match
            raise ImportError("%s reports ABI %d, this binding needs %d: rebuild it (make -C %s)"
                              % (LIB_PATH, got, SDPA_ABI_VERSION, os.path.join(PKG_DIR, "csrc")))
    _lib = lib
    return lib


def reload_env():
    """sdpa_reload_env(): the launch paths read their $SDPA_* knobs from one snapshot; a device-level
    host that changes one between launches re-takes it (include/sdpa_hip.h)."""
    load().sdpa_reload_env()


def strerror(code):
    return load().sdpa_strerror(code).decode()


def check(code, what):
    if code != SDPA_OK:
        raise SdpaError(code, what)
