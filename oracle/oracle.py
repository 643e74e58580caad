"""ctypes front-end for the CPU checker in oracle/ (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package never does.

* `Oracle`  -- our own restatement (oracle/sdpa_oracle.c -> liboracle.so).
* `RefSerial` -- the reference's own `attention()` (attention.c:20-75) compiled
  unmodified into oracle/_ref/libref_serial.so by `make -C oracle ref`; exists
  only where that build ran (it travels to the GPU box as a prebuilt file).
* the on-disk test-file format of the reference (attention.c:92-121, :139-140):
  int32 m,n,dk,dv; Q[m*dk], K[n*dk], V[n*dv] fp64; answer[m*dv] fp64.
"""
import ctypes
import os
import struct
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = ctypes.POINTER(ctypes.c_double)
_fp = ctypes.POINTER(ctypes.c_float)


def _d(a):
    return a.ctypes.data_as(_dp)


def _f(a):
    return a.ctypes.data_as(_fp)


def build(ref=True):
    """Compile liboracle.so, and oracle/_ref when /root/reference is present."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    if ref and os.path.isdir(os.environ.get("SDPA_REFERENCE", "/root/reference")):
        subprocess.call(["make", "-s", "-C", HERE, "ref"],
                        stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


class Oracle:
    def __init__(self):
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = ctypes.CDLL(path)
        L.oracle_owner_count.restype = ctypes.c_int
        L.oracle_owner_disp.restype = ctypes.c_int
        L.oracle_threads.restype = ctypes.c_int
        L.oracle_attention_f64.argtypes = [_dp, _dp, _dp, _dp] + [ctypes.c_int] * 4
        L.oracle_attention_sharded_f32.argtypes = [_dp, _dp, _dp, _dp] + [ctypes.c_int] * 5
        L.oracle_shard_partial_f32.argtypes = [_fp] * 6 + [ctypes.c_int] * 4
        L.oracle_cvt_d2f.argtypes = [_fp, _dp, ctypes.c_size_t]
        L.oracle_cvt_f2d.argtypes = [_dp, _fp, ctypes.c_size_t]
        self.L = L

    def threads(self):
        return self.L.oracle_threads()

    def owner_count(self, n, size, rank):
        return self.L.oracle_owner_count(n, size, rank)

    def owner_disp(self, n, size, rank):
        return self.L.oracle_owner_disp(n, size, rank)

    def attention_f64(self, Q, K, V):
        Q, K, V = (np.ascontiguousarray(x, dtype=np.float64) for x in (Q, K, V))
        m, dk = Q.shape
        n, dv = V.shape
        out = np.empty((m, dv), dtype=np.float64)
        self.L.oracle_attention_f64(_d(Q), _d(K), _d(V), _d(out), m, n, dk, dv)
        return out

    def attention_sharded_f32(self, Q, K, V, parts):
        Q, K, V = (np.ascontiguousarray(x, dtype=np.float64) for x in (Q, K, V))
        m, dk = Q.shape
        n, dv = V.shape
        out = np.empty((m, dv), dtype=np.float64)
        self.L.oracle_attention_sharded_f32(_d(Q), _d(K), _d(V), _d(out), m, n, dk, dv, parts)
        return out

    def shard_partial_f32(self, Qf, Kf, Vf):
        Qf, Kf, Vf = (np.ascontiguousarray(x, dtype=np.float32) for x in (Qf, Kf, Vf))
        m, dk = Qf.shape
        n_local = Kf.shape[0]
        dv = Vf.shape[1]
        contrib = np.empty((m, dv), dtype=np.float32)
        lmax = np.empty(m, dtype=np.float32)
        lsum = np.empty(m, dtype=np.float32)
        self.L.oracle_shard_partial_f32(_f(Qf), _f(Kf), _f(Vf), _f(contrib), _f(lmax), _f(lsum),
                                        m, n_local, dk, dv)
        return contrib, lmax, lsum


class RefSerial:
    """The reference's own serial attention() from oracle/_ref/libref_serial.so."""

    PATH = os.path.join(HERE, "_ref", "libref_serial.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.PATH)

    def __init__(self):
        L = ctypes.CDLL(self.PATH)
        L.attention.argtypes = [_dp, _dp, _dp, _dp] + [ctypes.c_int] * 4
        L.attention.restype = None
        self.L = L

    def attention(self, Q, K, V):
        Q, K, V = (np.ascontiguousarray(x, dtype=np.float64) for x in (Q, K, V))
        m, dk = Q.shape
        n, dv = V.shape
        out = np.empty((m, dv), dtype=np.float64)
        self.L.attention(_d(Q), _d(K), _d(V), _d(out), m, n, dk, dv)
        return out


def numpy_attention_f64(Q, K, V, rows=None):
    """Vectorised fp64 attention (optionally a subset of query rows) for shapes
    where the serial C oracle would take too long.  Same maths as attention.c:20-75,
    different summation order (BLAS), agreement ~1e-15."""
    Q = np.asarray(Q, dtype=np.float64)
    if rows is not None:
        Q = Q[rows]
    s = (Q @ np.asarray(K, dtype=np.float64).T) * (1.0 / np.sqrt(float(K.shape[1])))
    s -= s.max(axis=1, keepdims=True)
    np.exp(s, out=s)
    s /= s.sum(axis=1, keepdims=True)
    return s @ np.asarray(V, dtype=np.float64)


# ----------------------------------------------------------------------------
# synthetic inputs: D1 U(-1,1), D2 N(0,1), D3 N(0,2^2),
# D4 adversarial (late spike key, constant rows)
# ----------------------------------------------------------------------------
def make_inputs(m, n, dk, dv, dist="D1", seed=1):
    rng = np.random.default_rng(seed)
    if dist == "D1":
        Q, K, V = (rng.uniform(-1.0, 1.0, s) for s in ((m, dk), (n, dk), (n, dv)))
    elif dist == "D2":
        Q, K, V = (rng.standard_normal(s) for s in ((m, dk), (n, dk), (n, dv)))
    elif dist == "D3":
        Q, K, V = (2.0 * rng.standard_normal(s) for s in ((m, dk), (n, dk), (n, dv)))
    elif dist == "D4":
        Q, K, V = (rng.standard_normal(s) for s in ((m, dk), (n, dk), (n, dv)))
        # a key aligned with query row 0, late in the sequence: the running max
        # of that row jumps by a large amount in the last shard / last tiles
        j = max(0, n - 3)
        K[j] = 8.0 * Q[0] / max(1e-9, np.linalg.norm(Q[0])) * np.sqrt(dk) ** 0.5
        if n > 4:
            K[1] = K[0]          # duplicate keys
            V[2] = 3.25          # a constant value row
        if m > 2:
            Q[1] = 0.0           # all scores equal -> uniform softmax
    else:
        raise ValueError(dist)
    return (np.ascontiguousarray(Q), np.ascontiguousarray(K), np.ascontiguousarray(V))


# ---- golden cases (tests/golden/INDEX.json) -------------------------------------------------------------------
# Two kinds.  "file": the whole case in the reference's on-disk format, inputs and reference-produced answer.
# "answer_file" (mid-size cases, round 4): ONLY the reference-produced answer block is committed (a quarter to one
# megabyte instead of 17-70); the inputs are the seeded make_inputs() draw named by the entry, checked against the
# sha256 of the bytes the reference was run on.
def golden_dir():
    return os.path.join(os.path.dirname(HERE), "tests", "golden")


def load_golden(case):
    """(Q, K, V, answer) of one INDEX.json entry"""
    if case.get("file"):
        return read_case(os.path.join(golden_dir(), case["file"]))
    import hashlib
    Q, K, V = make_inputs(case["m"], case["n"], case["dk"], case["dv"], case["dist"], case["seed"])
    for name, a in (("Q", Q), ("K", K), ("V", V)):
        got = hashlib.sha256(np.ascontiguousarray(a, dtype="<f8").tobytes()).hexdigest()
        if got != case["inputs_sha256"][name]:
            raise RuntimeError("golden case %s: the seeded %s differs from the array the reference was run on "
                               "(numpy's generator changed?)" % (case["name"], name))
    ans = np.fromfile(os.path.join(golden_dir(), case["answer_file"]), dtype="<f8").reshape(case["m"], case["dv"])
    return Q, K, V, ans


def golden_file(case, tmp_dir):
    """path of the case in the reference's file format (written into tmp_dir for an answer-only case)"""
    if case.get("file"):
        return os.path.join(golden_dir(), case["file"])
    path = os.path.join(str(tmp_dir), case["name"] + ".bin")
    if not os.path.exists(path):
        Q, K, V, ans = load_golden(case)
        write_case(path, Q, K, V, ans)
    return path


def write_case(path, Q, K, V, answer):
    m, dk = Q.shape
    n, dv = V.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", m, n, dk, dv))
        for a in (Q, K, V, answer):
            f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())


def read_case(path):
    with open(path, "rb") as f:
        m, n, dk, dv = struct.unpack("<4i", f.read(16))
        Q = np.frombuffer(f.read(8 * m * dk), dtype="<f8").reshape(m, dk)
        K = np.frombuffer(f.read(8 * n * dk), dtype="<f8").reshape(n, dk)
        V = np.frombuffer(f.read(8 * n * dv), dtype="<f8").reshape(n, dv)
        rest = f.read()
        ans = np.frombuffer(rest, dtype="<f8").reshape(m, dv) if len(rest) == 8 * m * dv else None
    return Q, K, V, ans
