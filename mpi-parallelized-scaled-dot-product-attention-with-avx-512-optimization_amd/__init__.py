"""MI355X (gfx950) scaled-dot-product-attention engine: the reference's attention hot path
behind its own boundary.

    csrc/      hand-written HIP kernels + the C ABI of include/sdpa_hip.h -> lib/libsdpa_hip.so
    host/      attention-hip.c, the plain-C CLI drop-in (-> bin/attention-hip)
    engine.py  Python mirror of the reference's two attention() entry points
    _lib.py    ctypes prototypes

The directory name contains hyphens (it is the name the task prescribes), so import it with
    importlib.import_module("mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd")
"""
from . import _lib
from ._lib import SdpaError, header_symbols, load, reload_env
from .engine import (DEFAULT_Q_BATCH, HipBackend, ShardedAttention, attention, attention_mpi, attention_qrows, init,
                     last_launch, last_timing, owner_count, owner_disp, plan, prepare, round4, shutdown)

__all__ = ["SdpaError", "header_symbols", "load", "reload_env", "HipBackend", "ShardedAttention", "attention",
           "attention_mpi", "attention_qrows", "init", "last_launch", "last_timing", "owner_count", "owner_disp", "plan", "prepare", "round4", "shutdown",
           "DEFAULT_Q_BATCH", "_lib"]
