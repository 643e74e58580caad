"""Python host mirror of the reference's operator interface for the attention hot path.

Two entry points, same meaning as the reference's two `attention()` functions
(paths relative to the reference tree):

* `attention(Q, K, V)`            -- attention.c:20-21.  Host fp64 in, host fp64 out;
  the whole call runs inside libsdpa_hip.so (`sdpa_attention_f64`), one process
  driving the visible GPUs.
* `attention_mpi(Q, K, V, m, n, dk, dv, rank, world)` -- attention-mpi.c:191-192.  One
  process per GPU; only rank 0 holds the matrices (the others pass None, exactly like
  the reference's non-root ranks, attention-mpi.c:508-517).  K/V rows are sharded with
  owner_count/owner_disp (:19-27); per Q batch the shard-local triples are merged with
  all-reduce(MAX), all-reduce(SUM) and reduce(SUM) (:340-380) through
  `torch.distributed` (backend "nccl" = RCCL over xGMI).  Every compute stage is a HIP
  kernel reached through the C ABI (`HipBackend`).

PyTorch is used for device memory, streams and the process group only.  There is no
CPU compute path in this package: `HipBackend` raises if the library or the GPU is
missing.  (tests/ inject a checker-backed stand-in for the world_size-2 gloo run of
the collective choreography; that stand-in lives in tests/, not here.)
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import SdpaError, SdpaTiming, check

# Rows per Q batch.  The reference uses B = 512 (attention-mpi.c:200), an internal constant of its
# pipeline; here 32768 rows = 256 query blocks, which with 2 in-launch K/V splits is one full wave of
# 512 workgroups -- the fused kernel's best shape, and the C host's default (csrc/sdpa_host.hip,
# $SDPA_QBATCH)
DEFAULT_Q_BATCH = 32768


def round4(x):
    return (x + 3) // 4 * 4


def dense_ld(d):
    """leading dimension of an fp32 operand image: head dims in (32, 256] padded (zero columns) to
    64 / 128 / 256, the widths of the pipelined LDS-DMA kernel; beyond 256 a whole number of the dk-split kernels' lane runs
    (csrc/sdpa_internal.h: dense_ld)"""
    if d > 256:       # the dk-split kernels read 3 / 4 / 6 / 8 consecutive V columns per lane (dv <= 384 / 512 / 768 / beyond)
        q = 12 if d <= 384 else 4 if d <= 512 else 12 if d <= 768 else 8
        return (d + q - 1) // q * q
    return round4(d) if d <= 32 else 64 if d <= 64 else 128 if d <= 128 else 256


def _ld(be, d):
    return be.image_ld(d) if hasattr(be, "image_ld") else round4(d)


def owner_count(n, size, rank):
    """attention-mpi.c:19-22."""
    return _lib.load().sdpa_owner_count(n, size, rank)


def owner_disp(n, size, rank):
    """attention-mpi.c:24-27."""
    return _lib.load().sdpa_owner_disp(n, size, rank)


# --------------------------------------------------------------------------------------
# host level (single process, C ABI does everything)
# --------------------------------------------------------------------------------------
def init(n_gpus=0):
    check(_lib.load().sdpa_init(n_gpus), "sdpa_init")


def shutdown():
    _lib.load().sdpa_shutdown()


def attention(Q, K, V, flags=0, precision=None, plan=None, merge=None):
    """result = softmax(Q K^T / sqrt(dk)) V; numpy fp64 [m,dk],[n,dk],[n,dv] -> [m,dv].
    precision="bf16" selects the bf16-input MFMA path (flag SDPA_F_BF16); plan="qrows" shards the
    query rows over the engine's ranks instead of the K/V rows; merge="allreduce" is the
    reference's literal two-phase merge (default: one all-gather of the (lmax, lsum) pairs)."""
    lib = _lib.load()
    if precision == "bf16":
        flags |= _lib.SDPA_F_BF16
    if plan == "qrows":
        flags |= _lib.SDPA_F_PLAN_QROWS
    if merge == "allreduce":
        flags |= _lib.SDPA_F_MERGE_ALLREDUCE
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    K = np.ascontiguousarray(K, dtype=np.float64)
    V = np.ascontiguousarray(V, dtype=np.float64)
    if Q.ndim != 2 or K.ndim != 2 or V.ndim != 2 or Q.shape[1] != K.shape[1] or K.shape[0] != V.shape[0]:
        raise ValueError("expected Q[m,dk], K[n,dk], V[n,dv]")
    m, dk = Q.shape
    n, dv = V.shape
    out = np.empty((m, dv), dtype=np.float64)
    check(lib.sdpa_attention_f64(Q.ctypes.data, K.ctypes.data, V.ctypes.data, out.ctypes.data,
                                 m, n, dk, dv, flags), "sdpa_attention_f64")
    return out


def prepare(m, n, dk, dv, flags=0, precision=None):
    """sdpa_prepare(): buffers at their real sizes, the clock warm-up and one small call through the same code path -- which is
    also the start-up probe of the streamed launch (a runtime on which it cannot be fed is found out here and the engine keeps
    the launch-per-chunk schedule)."""
    if precision == "bf16":
        flags |= _lib.SDPA_F_BF16
    check(_lib.load().sdpa_prepare(m, n, dk, dv, flags), "sdpa_prepare")


def plan(m, n, dk, dv, flags=0, ranks=1):
    """The schedule sdpa_attention_f64 would run (sdpa_plan_describe): needs no GPU."""
    import ctypes
    import json
    buf = ctypes.create_string_buffer(1 << 18)
    check(_lib.load().sdpa_plan_describe(m, n, dk, dv, flags, ranks, buf, len(buf)), "sdpa_plan_describe")
    return json.loads(buf.value.decode())


def last_launch():
    """sdpa_dev_last_launch(): the calling thread's last fused launch through the device-level API -- kernel name as
    rocprofv3 prints it, grid, slabs, stream-K or not."""
    import json
    buf = ctypes.create_string_buffer(512)
    check(_lib.load().sdpa_dev_last_launch(buf, len(buf)), "sdpa_dev_last_launch")
    return json.loads(buf.value.decode())


def last_timing():
    t = SdpaTiming()
    check(_lib.load().sdpa_last_timing_sized(ctypes.byref(t), ctypes.sizeof(t)), "sdpa_last_timing_sized")
    out = {k: getattr(t, k) for k, _ in SdpaTiming._fields_}
    out["last_kernel"] = out["last_kernel"].decode("ascii", "replace")
    out["enqueue_first_kernel_us"] = list(out["enqueue_first_kernel_us"])[:max(1, out["n_gpus"])]
    return out


# --------------------------------------------------------------------------------------
# device level (one process per GPU)
# --------------------------------------------------------------------------------------
class HipBackend:
    """The stages of the hot path on torch CUDA tensors, each one a HIP kernel behind the
    C ABI.  All work is enqueued on torch's current stream of `device`."""

    name = "hip"

    def __init__(self, device=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise SdpaError(_lib.SDPA_ENODEV, "HipBackend")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._ws = None

    @property
    def comm_device(self):
        return self.device

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def to_device(self, a, dtype=None):
        """host array -> device, synchronously.  Not non_blocking: handed a pageable source with non_blocking=True the HIP
        runtime maps the user's pages into the GPU's address space for the duration of the copy, and two such transient
        mappings that share a page (neighbouring numpy arrays in the heap: K and V) pull it from under each other -- the one
        GPU memory fault this repo has seen was a read of a page-aligned host heap address in exactly this call
        (profiles/r04/gpu_memory_fault_on_a_host_heap_page.log).  And not `t.is_pinned()` to find out which sources could
        go asynchronously either: that query is hipPointerGetAttributes, which makes ROCm 7 log "Cannot get amd_mem_obj
        for ptr" at error level for every pageable array (round 5: 951 such lines in one -m gpu run came from here)."""
        t = torch.as_tensor(a)
        return t.to(self.device, dtype=dtype, non_blocking=t.device.type != "cpu")

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def image_ld(self, d):
        return dense_ld(d)

    def cvt_d2f(self, x64):
        """cvt_d2f_avx512 (attention-mpi.c:31-64): [rows, cols] f64 -> [rows, dense_ld(cols)] f32, pad columns zero."""
        rows, cols = x64.shape
        ld = dense_ld(cols)
        out = self.empty((rows, ld), torch.float32)
        if rows:
            assert x64.is_contiguous() and x64.dtype == torch.float64
            with torch.cuda.device(self.device):
                check(self.lib.sdpa_dev_cvt_d2f(x64.data_ptr(), out.data_ptr(), rows, cols, ld, self._stream()),
                      "sdpa_dev_cvt_d2f")
        return out

    def cvt_d2f_batch(self, xs):
        """up to three cvt_d2f's in ONE launch (sdpa_dev_cvt_d2f_batch): a list of [rows, cols] f64 -> the list of their images"""
        import ctypes
        k = len(xs)
        assert 1 <= k <= 3 and all(x.is_contiguous() and x.dtype == torch.float64 for x in xs)
        outs = [self.empty((x.shape[0], dense_ld(x.shape[1])), torch.float32) for x in xs]
        P, L, I = ctypes.c_void_p * k, ctypes.c_long * k, ctypes.c_int * k
        with torch.cuda.device(self.device):
            check(self.lib.sdpa_dev_cvt_d2f_batch(k, P(*[x.data_ptr() for x in xs]), P(*[o.data_ptr() for o in outs]),
                                                  L(*[x.shape[0] for x in xs]), I(*[x.shape[1] for x in xs]),
                                                  I(*[o.shape[1] for o in outs]), self._stream()), "sdpa_dev_cvt_d2f_batch")
        return outs

    def cvt_f2d(self, x32, cols):
        """cvt_f2d_avx512 (attention-mpi.c:68-101)."""
        rows, ld = x32.shape
        out = self.empty((rows, cols), torch.float64)
        if rows:
            with torch.cuda.device(self.device):
                check(self.lib.sdpa_dev_cvt_f2d(x32.data_ptr(), ld, out.data_ptr(), rows, cols, self._stream()),
                      "sdpa_dev_cvt_f2d")
        return out

    def shard_partial(self, Qf, Kf, Vf, dk, dv):
        """online_softmax_attention (attention-mpi.c:168-189) for every row of Qf against one shard."""
        m = Qf.shape[0]
        n_local = Kf.shape[0]
        ldo = max(round4(dv), Vf.shape[1])          # rows as wide as the V image's (the padded kernels write them whole)
        contrib = self.empty((m, ldo), torch.float32)
        lmax = self.empty((m,), torch.float32)
        lsum = self.empty((m,), torch.float32)
        need = self.lib.sdpa_dev_workspace_bytes(m, n_local, dk, dv)
        if need and (self._ws is None or self._ws.numel() < need):
            self._ws = self.empty((need,), torch.uint8)
        with torch.cuda.device(self.device):
            check(self.lib.sdpa_dev_shard_partial_f32(
                Qf.data_ptr(), Qf.shape[1], Kf.data_ptr() if n_local else None, Kf.shape[1],
                Vf.data_ptr() if n_local else None, Vf.shape[1], contrib.data_ptr(), ldo,
                lmax.data_ptr(), lsum.data_ptr(), m, n_local, dk, dv,
                self._ws.data_ptr() if need else None, need, self._stream()), "sdpa_dev_shard_partial_f32")
        return contrib, lmax, lsum

    def shard_attention_f64(self, Qf, Kf, Vf, dk, dv):
        """The single-shard call (sdpa_dev_shard_attention_f64): shard_partial + merge step 5 with gsum = lsum + the fp64
        writeback (attention-mpi.c:358-362, :373), the finish fused into the merge of the in-GPU splits.  -> result [m, dv] fp64."""
        m = Qf.shape[0]
        n_local = Kf.shape[0]
        ldo = max(round4(dv), Vf.shape[1])
        contrib = self.empty((m, ldo), torch.float32)
        stats = self.empty((2, m), torch.float32)
        out = self.empty((m, dv), torch.float64)
        need = self.lib.sdpa_dev_workspace_bytes(m, n_local, dk, dv)
        if need and (self._ws is None or self._ws.numel() < need):
            self._ws = self.empty((need,), torch.uint8)
        with torch.cuda.device(self.device):
            check(self.lib.sdpa_dev_shard_attention_f64(
                Qf.data_ptr(), Qf.shape[1], Kf.data_ptr() if n_local else None, Kf.shape[1],
                Vf.data_ptr() if n_local else None, Vf.shape[1], contrib.data_ptr(), ldo,
                stats[0].data_ptr(), stats[1].data_ptr(), out.data_ptr(), m, n_local, dk, dv,
                self._ws.data_ptr() if need else None, need, self._stream()), "sdpa_dev_shard_attention_f64")
        return out

    # ---- bf16-input MFMA variant (BASELINE config 5) -------------------------------------
    def cvt_d2bf(self, x64, ld=None):
        """fp64 [rows, cols] -> bf16 [rows, ld] (RNE, pad columns zero); ld defaults to the padded dk."""
        rows, cols = x64.shape
        ld = self.lib.sdpa_dev_bf16_ld(cols) if ld is None else ld
        check(min(ld, 0), "sdpa_dev_bf16_ld")
        out = self.empty((rows, ld), torch.bfloat16)
        if rows:
            assert x64.is_contiguous() and x64.dtype == torch.float64
            with torch.cuda.device(self.device):
                check(self.lib.sdpa_dev_cvt_d2bf(x64.data_ptr(), out.data_ptr(), rows, cols, ld, self._stream()),
                      "sdpa_dev_cvt_d2bf")
        return out

    def cvt_d2bf_q(self, q64):
        """fp64 Q [rows, dk] -> the bf16 Q image [rows, ld]: bf16(Q * log2(e)/sqrtf(dk)), one rounding
        (the softmax scale and the exp2 change of base live in the operand)."""
        rows, dk = q64.shape
        ld = self.lib.sdpa_dev_bf16_ld(dk)
        check(min(ld, 0), "sdpa_dev_bf16_ld")
        out = self.empty((rows, ld), torch.bfloat16)
        if rows:
            assert q64.is_contiguous() and q64.dtype == torch.float64
            with torch.cuda.device(self.device):
                check(self.lib.sdpa_dev_cvt_d2bf_q(q64.data_ptr(), out.data_ptr(), rows, dk, ld, self._stream()),
                      "sdpa_dev_cvt_d2bf_q")
        return out

    def cvt_d2bf_k(self, k64, dv):
        """fp64 K[n, dk] -> the K image of a (dk, dv) shape, [n_pad, ld] bf16: plain rows for dv <= 256, for dv > 256 the
        TILED image (chunk-swizzled rows, include/sdpa_hip.h) with its last tile's pad rows zeroed."""
        n, dk = k64.shape
        ld, ldn = self.lib.sdpa_dev_bf16_ld(dk), self.lib.sdpa_dev_bf16_ldn(n)
        check(min(ld, 0), "sdpa_dev_bf16_ld")
        out = self.empty((max(ldn, 32), ld), torch.bfloat16)
        if n:
            assert k64.is_contiguous() and k64.dtype == torch.float64
            with torch.cuda.device(self.device):
                check(self.lib.sdpa_dev_cvt_d2bf_k(k64.data_ptr(), out.data_ptr(), n, dk, dv, self._stream()),
                      "sdpa_dev_cvt_d2bf_k")
        return out

    def cvt_d2bf_t(self, v64):
        """fp64 V[n, dv] -> the transposed bf16 image the bf16 kernel of this dv reads: Vt[dv_pad, n_pad] (key positions
        permuted, dv <= 256) or the tiled image [n_pad/32, dv_pad/512, 512, 32] (dv > 256), returned as [dv_pad, n_pad] elements."""
        n, dv = v64.shape
        dvp, ldn = self.lib.sdpa_dev_bf16_dvp(dv), self.lib.sdpa_dev_bf16_ldn(n)
        check(min(dvp, 0), "sdpa_dev_bf16_dvp")
        out = self.empty((dvp, max(ldn, 32)), torch.bfloat16)
        assert v64.is_contiguous() and v64.dtype == torch.float64
        with torch.cuda.device(self.device):
            check(self.lib.sdpa_dev_cvt_d2bf_t(v64.data_ptr() if n else None, out.data_ptr(), n, dv, dvp,
                                               out.shape[1], self._stream()), "sdpa_dev_cvt_d2bf_t")
        return out

    def shard_partial_bf16(self, Qb, Kb, Vt, n_local, dk, dv):
        """online_softmax_attention (attention-mpi.c:168-189) on the bf16 MFMA kernel."""
        m = Qb.shape[0]
        if n_local == 0:    # empty shard: (0, -inf, 0) as attention-mpi.c:172-173
            zk = self.empty((0, 4), torch.float32)
            zv = self.empty((0, round4(dv)), torch.float32)
            return self.shard_partial(self.empty((m, 4), torch.float32).zero_(), zk, zv, 4, dv)
        ldo = round4(dv)
        contrib = self.empty((m, ldo), torch.float32)
        lmax = self.empty((m,), torch.float32)
        lsum = self.empty((m,), torch.float32)
        need = self.lib.sdpa_dev_workspace_bytes_bf16(m, n_local, dk, dv)
        if need and (self._ws is None or self._ws.numel() < need):
            self._ws = self.empty((need,), torch.uint8)
        with torch.cuda.device(self.device):
            check(self.lib.sdpa_dev_shard_partial_bf16(
                Qb.data_ptr(), Qb.shape[1], Kb.data_ptr(), Kb.shape[1], Vt.data_ptr(), Vt.shape[1],
                contrib.data_ptr(), ldo, lmax.data_ptr(), lsum.data_ptr(), m, n_local, dk, dv,
                self._ws.data_ptr() if need else None, need, self._stream()), "sdpa_dev_shard_partial_bf16")
        return contrib, lmax, lsum

    def merge_rescale(self, contrib, lsum, lmax, gmax, dv):
        """attention-mpi.c:346-351 (in place)."""
        with torch.cuda.device(self.device):
            check(self.lib.sdpa_dev_merge_rescale(contrib.data_ptr(), contrib.shape[1], lsum.data_ptr(),
                                                  lmax.data_ptr(), gmax.data_ptr(), contrib.shape[0], dv,
                                                  self._stream()), "sdpa_dev_merge_rescale")

    def merge_normalise(self, contrib, gsum, dv):
        """attention-mpi.c:358-362 (in place)."""
        with torch.cuda.device(self.device):
            check(self.lib.sdpa_dev_merge_normalise(contrib.data_ptr(), contrib.shape[1], gsum.data_ptr(),
                                                    contrib.shape[0], dv, self._stream()),
                  "sdpa_dev_merge_normalise")

    def merge_gathered(self, contrib, stats, self_index, dv):
        """attention-mpi.c:342-362 in one pass from all-gathered (lmax, lsum) pairs (in place)."""
        with torch.cuda.device(self.device):
            check(self.lib.sdpa_dev_merge_gathered(contrib.data_ptr(), contrib.shape[1], stats.data_ptr(),
                                                   stats.shape[0], self_index, contrib.shape[0], dv,
                                                   self._stream()), "sdpa_dev_merge_gathered")

    def finish_f64(self, contrib, lsum, dv):
        """single shard: step 5 with gsum = lsum fused with the fp64 writeback (:358-362,:373)."""
        out = self.empty((contrib.shape[0], dv), torch.float64)
        with torch.cuda.device(self.device):
            check(self.lib.sdpa_dev_finish_f64(contrib.data_ptr(), contrib.shape[1], lsum.data_ptr(),
                                               out.data_ptr(), contrib.shape[0], dv, self._stream()),
                  "sdpa_dev_finish_f64")
        return out


class ShardedAttention:
    """One rank's state for the K/V-sharded path: the resident fp32 shard plus the per-batch
    merge choreography of attention-mpi.c:307-399.  `dist` is torch.distributed (or None for a
    single rank)."""

    def __init__(self, backend, rank=0, world=1, dist=None, group=None, root=0, force_collectives=False,
                 precision="f32", merge="allreduce", egress="root"):
        assert precision in ("f32", "bf16") and merge in ("allreduce", "gather") and egress in ("root", "scatter")
        self.precision = precision
        self.merge = merge        # "allreduce": the reference's two-phase merge; "gather": one all-gather
        # how the merged rows of a batch leave (batch_merge_egress): "root" = reduce to the root
        # (attention-mpi.c:380); "scatter" = reduce-scatter, rank r keeps rows [r*share, (r+1)*share) of the batch
        # and widens them itself -- the C host's default schedule (csrc/sdpa_host.hip: tail_batch)
        self.egress = egress
        self.be = backend
        self.rank, self.world, self.root = rank, world, root
        self.dist = dist if (world > 1 or force_collectives) else None
        self.group = group
        self.Kf = self.Vf = None
        self.dk = self.dv = self.n = None

    # ---- K/V distribution: attention-mpi.c:210-266 ------------------------------------
    def load_kv_from_root(self, K64, V64, n, dk, dv):
        """Root holds K,V (fp64, any of numpy / CPU tensor / device tensor); every rank ends up
        with its own fp32 shard resident.  The root converts the whole of K and V to fp32 first
        (:224-225,:248-249) and the shards travel as fp32 (Scatterv, :258-264)."""
        be = self.be
        self.n, self.dk, self.dv = n, dk, dv
        cnt = owner_count(n, self.world, self.rank)
        if self.world == 1:
            self.load_kv_shard_f64(be.to_device(K64, torch.float64).contiguous(),
                                   be.to_device(V64, torch.float64).contiguous(), n, dk, dv)
            return
        assert self.precision == "f32", "the root-scatter path distributes fp32 shards"
        # Shard by shard (the Scatterv of :258-264 as point-to-point messages of the exact row counts):
        # the root moves ONE shard's fp64 rows to its device, converts them, keeps its own or sends the
        # rest -- never more than one shard of staging on the root, no padded copies.  The rows cross the
        # root's PCIe link once; that is inherent in the reference's contract that only rank 0 holds the
        # matrices (the C host, where every rank reads the caller's arrays itself, uses P links).
        ldk, ldv = _ld(be, dk), _ld(be, dv)
        if self.rank == self.root:
            K64 = K64 if torch.is_tensor(K64) else torch.as_tensor(K64)
            V64 = V64 if torch.is_tensor(V64) else torch.as_tensor(V64)
            for r in range(self.world):
                c, d = owner_count(n, self.world, r), owner_disp(n, self.world, r)
                if c == 0 and r != self.root:
                    continue
                kf = be.cvt_d2f(be.to_device(K64[d:d + c], torch.float64).contiguous())
                vf = be.cvt_d2f(be.to_device(V64[d:d + c], torch.float64).contiguous())
                if r == self.root:
                    self.Kf, self.Vf = kf, vf
                else:
                    self.dist.send(kf, dst=r, group=self.group)
                    self.dist.send(vf, dst=r, group=self.group)
        else:
            self.Kf = be.empty((cnt, ldk), torch.float32)
            self.Vf = be.empty((cnt, ldv), torch.float32)
            if cnt > 0:
                self.dist.recv(self.Kf, src=self.root, group=self.group)
                self.dist.recv(self.Vf, src=self.root, group=self.group)
        self.n_local = cnt

    def load_kv_shard_f64(self, K64_local, V64_local, n, dk, dv):
        """This rank's rows of K and V are on its device in fp64 (bench: resident inputs): convert
        them to the operand image of the selected precision (attention-mpi.c:224-225)."""
        self.n, self.dk, self.dv = n, dk, dv
        self.n_local = K64_local.shape[0]
        if self.precision == "bf16":
            self.Kf = self.be.cvt_d2bf_k(K64_local, dv)
            self.Vf = self.be.cvt_d2bf_t(V64_local)
        else:
            self.Kf = self.be.cvt_d2f(K64_local)
            self.Vf = self.be.cvt_d2f(V64_local)

    def load_kv_shard_and_q_f64(self, K64_local, V64_local, Q64, n, dk, dv):
        """load_kv_shard_f64 + convert_q of a call's ONLY Q batch in one launch (fp32 path: sdpa_dev_cvt_d2f_batch) -> the Q image"""
        if self.precision != "f32" or min(K64_local.shape[0], Q64.shape[0]) == 0:
            self.load_kv_shard_f64(K64_local, V64_local, n, dk, dv)
            return self.convert_q(Q64)
        self.n, self.dk, self.dv = n, dk, dv
        self.n_local = K64_local.shape[0]
        self.Kf, self.Vf, qf = self.be.cvt_d2f_batch([K64_local, V64_local, Q64])
        return qf

    def convert_q(self, Q64):
        """Q batch fp64 -> operand image (attention-mpi.c:303,:325)."""
        return self.be.cvt_d2bf_q(Q64) if self.precision == "bf16" else self.be.cvt_d2f(Q64)

    def load_kv_shard(self, Kf_local, Vf_local, n, dk, dv):
        """The shard is already on this rank's device as padded fp32 (bench: resident inputs)."""
        self.Kf, self.Vf, self.n, self.dk, self.dv = Kf_local, Vf_local, n, dk, dv

    # ---- one Q batch: attention-mpi.c:333-380 -----------------------------------------
    def batch_partial(self, Qf):
        if self.precision == "bf16":
            return self.be.shard_partial_bf16(Qf, self.Kf, self.Vf, self.n_local, self.dk, self.dv)
        return self.be.shard_partial(Qf, self.Kf, self.Vf, self.dk, self.dv)

    def batch_attention_f64(self, Qf):
        """one rank, no merge over ranks: the batch's finished fp64 rows in one call (fused kernel + ONE finishing pass)"""
        assert self.dist is None and self.precision == "f32"
        return self.be.shard_attention_f64(Qf, self.Kf, self.Vf, self.dk, self.dv)

    def batch_merge(self, contrib, lmax, lsum, async_reduce=False):
        """Steps 2-5 and 7 of the reference loop.  Returns (contrib, work): on the root `contrib`
        holds the normalised, shard-summed fp32 rows once `work` (if any) has completed."""
        be, dist = self.be, self.dist
        if dist is None:
            be.merge_normalise(contrib, lsum, self.dv)
            return contrib, None
        if self.merge == "gather":
            mine = torch.stack((lmax, lsum))                              # [2, m]
            stats = be.empty((self.world if self.world > 1 else 1, 2, lmax.shape[0]), torch.float32)
            # one ncclAllGather straight into the [world, 2, m] image (a list of output tensors would
            # cost a staging buffer and `world` copy kernels per batch)
            dist.all_gather_into_tensor(stats.view(-1, lmax.shape[0]), mine, group=self.group)
            be.merge_gathered(contrib, stats, self.rank if self.world > 1 else 0, self.dv)   # :342-362
            work = dist.reduce(contrib, dst=self.root, op=dist.ReduceOp.SUM, group=self.group,
                               async_op=async_reduce)                     # :380
            return contrib, work
        gmax = lmax.clone()
        dist.all_reduce(gmax, op=dist.ReduceOp.MAX, group=self.group)     # :342
        be.merge_rescale(contrib, lsum, lmax, gmax, self.dv)              # :346-351
        gsum = lsum.clone()
        dist.all_reduce(gsum, op=dist.ReduceOp.SUM, group=self.group)     # :354
        be.merge_normalise(contrib, gsum, self.dv)                        # :358-362
        work = dist.reduce(contrib, dst=self.root, op=dist.ReduceOp.SUM, group=self.group,
                           async_op=async_reduce)                         # :380
        return contrib, work

    def _merge_stats(self, contrib, lmax, lsum, marks=None):
        """Steps 2-5 of the reference loop (attention-mpi.c:340-362) in place on `contrib`: the statistics collective(s)
        and the merge kernel(s).  marks = (after the collective, after the merge kernel, ...) HIP events, optional."""
        be, dist = self.be, self.dist
        if self.merge == "gather":
            mine = torch.stack((lmax, lsum))                              # [2, m]
            stats = be.empty((self.world if self.world > 1 else 1, 2, lmax.shape[0]), torch.float32)
            dist.all_gather_into_tensor(stats.view(-1, lmax.shape[0]), mine, group=self.group)
            if marks:
                marks[0].record()
            be.merge_gathered(contrib, stats, self.rank if self.world > 1 else 0, self.dv)   # :342-362
        else:
            gmax = lmax.clone()
            dist.all_reduce(gmax, op=dist.ReduceOp.MAX, group=self.group)     # :342
            be.merge_rescale(contrib, lsum, lmax, gmax, self.dv)              # :346-351
            gsum = lsum.clone()
            dist.all_reduce(gsum, op=dist.ReduceOp.SUM, group=self.group)     # :354
            if marks:
                marks[0].record()
            be.merge_normalise(contrib, gsum, self.dv)                        # :358-362
        if marks:
            marks[1].record()

    def batch_merge_egress(self, contrib, lmax, lsum, async_reduce=False, marks=None):
        """batch_merge with the egress this object was created with.  Returns (rows, work, nrows): once `work`
        (if any) has completed, `rows[:nrows]` holds normalised, shard-summed fp32 rows --
          egress "root":    all rows of the batch on the root (nrows = rows of the batch; elsewhere the tensor is
                            this rank's send buffer),
          egress "scatter": rows [rank*share, rank*share + nrows) of the batch on EVERY rank, share = ceil(bs / world):
                            ncclReduceScatter instead of ncclReduce -- each rank then widens and owns its share
                            (the C host sends them home over P PCIe links, sdpa_host.hip:tail_batch)."""
        be, dist = self.be, self.dist
        bs = contrib.shape[0]
        if dist is None:
            be.merge_normalise(contrib, lsum, self.dv)
            if marks:
                for mk in marks:
                    mk.record()
            return contrib, None, bs
        self._merge_stats(contrib, lmax, lsum, marks)
        if self.egress == "root":
            work = dist.reduce(contrib, dst=self.root, op=dist.ReduceOp.SUM, group=self.group,
                               async_op=async_reduce)                         # :380
            if marks:
                marks[2].record()
            return contrib, work, bs
        world = self.world if self.world > 1 else 1
        share = (bs + world - 1) // world
        send = contrib
        if share * world != bs:                       # pad to P equal shares (the C host's contrib buffers carry +P rows)
            send = be.empty((share * world, contrib.shape[1]), torch.float32)
            send[:bs] = contrib
            send[bs:].zero_()
        out = be.empty((share, contrib.shape[1]), torch.float32)
        work = dist.reduce_scatter_tensor(out, send, op=dist.ReduceOp.SUM, group=self.group, async_op=async_reduce)
        if marks:
            marks[2].record()
        r0 = (self.rank if self.world > 1 else 0) * share
        # (`send` must outlive the collective: keep it referenced by the returned tensor's owner)
        out._sdpa_send_keepalive = send
        return out, work, max(0, min(share, bs - r0))

    def forward_batches(self, q_batches):
        """Run a sequence of fp32 Q batches (each already on every rank) through partial + merge,
        with batch i's reduce overlapped with batch i+1's kernel (the Ireduce of :364-380).
        Returns the list of per-batch normalised fp32 results (meaningful on the root)."""
        outs, pending = [], None
        for Qf in q_batches:
            contrib, lmax, lsum = self.batch_partial(Qf)
            if pending is not None:
                pending.wait()
            contrib, pending = self.batch_merge(contrib, lmax, lsum, async_reduce=True)
            outs.append(contrib)
        if pending is not None:
            pending.wait()
        return outs


def attention_qrows(Q, K, V, m, n, dk, dv, rank, world, dist=None, group=None, backend=None):
    """The other natural sharding of this path: query rows are independent
    (attention.c:28), so each rank takes rows [owner_disp(m), +owner_count(m)) against the WHOLE
    K/V and no merge collective exists at all -- only the distribution (K, V broadcast, Q rows
    scattered) and the gather of finished fp64 rows to rank 0.  An alternative to the K/V-sharded
    plan the reference uses, for problems whose K/V fit one GPU (every BASELINE shape does).
    Rank 0 passes the fp64 matrices, the others None; returns the result on rank 0."""
    be = backend if backend is not None else HipBackend()
    root = 0
    if world > 1:
        dims = torch.tensor([m, n, dk, dv] if rank == root else [0, 0, 0, 0], dtype=torch.int64,
                            device=be.comm_device)
        dist.broadcast(dims, src=root, group=group)
        m, n, dk, dv = (int(x) for x in dims.tolist())
    cnt, off = owner_count(m, world, rank), owner_disp(m, world, rank)
    cmax = owner_count(m, world, 0)
    if rank == root:
        K64 = be.to_device(K, torch.float64).contiguous()
        V64 = be.to_device(V, torch.float64).contiguous()
        Q64 = be.to_device(Q, torch.float64).contiguous()
    else:
        K64 = be.empty((n, dk), torch.float64)
        V64 = be.empty((n, dv), torch.float64)
        Q64 = None
    qloc = be.empty((cmax, dk), torch.float64)
    if world > 1:
        dist.broadcast(K64, src=root, group=group)
        dist.broadcast(V64, src=root, group=group)
        parts = None
        if rank == root:
            parts = []
            for r in range(world):
                c, d = owner_count(m, world, r), owner_disp(m, world, r)
                buf = be.empty((cmax, dk), torch.float64).zero_()
                buf[:c] = Q64[d:d + c]
                parts.append(buf)
        dist.scatter(qloc, parts, src=root, group=group)
    else:
        qloc = Q64
    sa = ShardedAttention(be)
    sa.load_kv_shard_f64(K64, V64, n, dk, dv)
    out = be.empty((cmax, dv), torch.float64).zero_()
    if cnt > 0:
        contrib, lmax, lsum = sa.batch_partial(sa.convert_q(qloc[:cnt].contiguous()))
        out[:cnt] = be.finish_f64(contrib, lsum, dv)
    if world == 1:
        return out[:cnt].cpu().numpy()
    gathered = [be.empty((cmax, dv), torch.float64) for _ in range(world)] if rank == root else None
    dist.gather(out, gathered, dst=root, group=group)
    if rank != root:
        return None
    result = np.empty((m, dv), dtype=np.float64)
    for r in range(world):
        c, d = owner_count(m, world, r), owner_disp(m, world, r)
        result[d:d + c] = gathered[r][:c].cpu().numpy()
    return result


def attention_mpi(Q, K, V, m, n, dk, dv, rank, world, dist=None, group=None, backend=None,
                  q_batch=DEFAULT_Q_BATCH, merge="allreduce"):
    """Mirror of the MPI `attention()` (attention-mpi.c:191-407): rank 0 passes the fp64 matrices,
    every other rank passes None and may pass garbage dims; returns the fp64 [m,dv] result on
    rank 0 (None elsewhere)."""
    be = backend if backend is not None else HipBackend()
    root = 0
    if world > 1:
        dims = torch.tensor([m, n, dk, dv] if rank == root else [0, 0, 0, 0], dtype=torch.int64,
                            device=be.comm_device)
        dist.broadcast(dims, src=root, group=group)                       # :196
        m, n, dk, dv = (int(x) for x in dims.tolist())
    sa = ShardedAttention(be, rank, world, dist, group, root, merge=merge)
    sa.load_kv_from_root(K, V, n, dk, dv)

    B = min(q_batch, m)
    nb = (m + B - 1) // B
    result = np.empty((m, dv), dtype=np.float64) if rank == root else None
    Q64 = None
    if rank == root:
        Q64 = torch.as_tensor(np.ascontiguousarray(Q, dtype=np.float64) if isinstance(Q, np.ndarray) else Q)

    def fetch(b):      # Q ping-pong prefetch: root converts, everyone receives (:303-305,:323-327)
        i0 = b * B
        bs = min(B, m - i0)
        if rank == root:
            qf = be.cvt_d2f(be.to_device(Q64[i0:i0 + bs], torch.float64).contiguous())
        else:
            qf = be.empty((bs, _ld(be, dk)), torch.float32)
        w = dist.broadcast(qf, src=root, group=group, async_op=True) if world > 1 else None
        return qf, w

    nxt = fetch(0)
    pending = None          # (work, contrib, i0, bs) of the previous batch's reduce
    for b in range(nb):
        qf, w = nxt
        if w is not None:
            w.wait()                                                      # :316
        if b + 1 < nb:
            nxt = fetch(b + 1)
        contrib, lmax, lsum = sa.batch_partial(qf)                        # :333-338
        if pending is not None:                                           # :365-376
            pw, pc, pi0, pbs = pending
            if pw is not None:
                pw.wait()
            if rank == root:
                result[pi0:pi0 + pbs] = be.cvt_f2d(pc, dv).cpu().numpy()
        contrib, work = sa.batch_merge(contrib, lmax, lsum, async_reduce=world > 1)
        pending = (work, contrib, b * B, min(B, m - b * B))
    pw, pc, pi0, pbs = pending                                            # :387-399
    if pw is not None:
        pw.wait()
    if rank == root:
        result[pi0:pi0 + pbs] = be.cvt_f2d(pc, dv).cpu().numpy()
    return result
