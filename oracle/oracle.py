"""CPU oracle: restatement of the reference's query-time hot path (numpy + oracle/nlsh_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, `__graft_entry__.smoke()` and bench.py's
`cpu_baseline` leg; the product package never imports it (tests/test_host_cpu.py::test_product_never_references_the_oracle).

Parity pin: checked against golden vectors generated from the unmodified reference
(tests/golden/make_golden.py, tests/test_oracle_golden.py).  Reference lines restated:
nlsh/utils.pyx:6-32, eval.py:49-53, encoders.py:18-55, nlsh/hashings.py:13-27,66-92,
nlsh/indexer.py:6-24,40-96, nlsh/data.py:99-109,191-201, nlsh/metrics.py:4-25.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "nlsh_oracle.c")
_LIB = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """gcc the C restatement into oracle/_build/liboracle.so (x86-64-v3: the .so travels)."""
    if not force and os.path.exists(_LIB) and os.path.getmtime(_LIB) >= os.path.getmtime(_SRC):
        return _LIB
    os.makedirs(os.path.dirname(_LIB), exist_ok=True)
    cmd = ["gcc", "-O2", "-std=c99", "-mavx2", "-mfma", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC",
           "-o", _LIB, _SRC, "-lm"]
    subprocess.check_call(cmd)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_pack_full.restype = ctypes.c_int64
        _lib.oracle_build_csr.restype = ctypes.c_int64
        _lib.oracle_row_keys.restype = ctypes.c_int
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(ty)


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(n):
    lib().oracle_set_num_threads(ctypes.c_int(int(n)))


# ----------------------------------------------------------------------------- bit packing
def pack_keys(codes, mode="ref_int16"):
    """codes int32 [B, n, H] -> int64 keys [B, n]  (nlsh/utils.pyx:6-15 | eval.py:49-53)."""
    codes = np.ascontiguousarray(codes, dtype=np.int32)
    B, n, H = codes.shape
    keys = np.empty((B, n), dtype=np.int64)
    lib().oracle_hash_codes(_p(codes, c_i32p), ctypes.c_int64(B), n, H, 0 if mode == "ref_int16" else 1,
                            _p(keys, c_i64p))
    return keys


def hash_codes(codes, mode="ref_int16"):
    """nlsh/utils.pyx:18-32: list of B Python sets of packed keys."""
    keys = pack_keys(codes, mode)
    return [set(int(k) for k in row) for row in keys]


# ----------------------------------------------------------------------------- hasher forward
def mlp_forward(x, Ws, bs):
    """Pre-activation z [n, H] of encoder + output layer (k-ordered fp32 fma chains)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = x.shape[0]
    L = len(Ws)
    dims = np.array([x.shape[1]] + [W.shape[0] for W in Ws], dtype=np.int32)
    Wc = [np.ascontiguousarray(W, dtype=np.float32) for W in Ws]
    bc = [None if b is None else np.ascontiguousarray(b, dtype=np.float32) for b in bs]
    for l, W in enumerate(Wc):
        assert W.shape == (dims[l + 1], dims[l])
    Wp = (c_f32p * L)(*[_p(W, c_f32p) for W in Wc])
    bp = (c_f32p * L)(*[(_p(b, c_f32p) if b is not None else ctypes.cast(None, c_f32p)) for b in bc])
    z = np.empty((n, int(dims[-1])), dtype=np.float32)
    lib().oracle_mlp_forward(_p(x, c_f32p), ctypes.c_int64(n), L, Wp, bp, _p(dims, c_i32p), _p(z, c_f32p))
    return z


def mlp_forward_blas(x, Ws, bs):
    """Same forward with numpy/BLAS summation order (what torch-CPU would compute, up to order)."""
    h = np.asarray(x, dtype=np.float32)
    for l, (W, b) in enumerate(zip(Ws, bs)):
        h = h @ W.T.astype(np.float32)
        if b is not None:
            h = h + b
        if l + 1 < len(Ws):
            h = np.maximum(h, 0)
    return h.astype(np.float32)


def head_probs(z, act="sigmoid"):
    """(module output, Bernoulli probability): hashings.py:22-26 and :67-69."""
    z = np.ascontiguousarray(z, dtype=np.float32)
    raw = np.empty_like(z)
    p01 = np.empty_like(z)
    lib().oracle_head_probs(_p(z, c_f32p), ctypes.c_int64(z.size), 0 if act == "sigmoid" else 1,
                            _p(raw, c_f32p), _p(p01, c_f32p))
    return raw, p01


def hard_bits(p01):
    """hashings.py:72: strict `prob > 0.5` on the fp32 probability."""
    return (np.asarray(p01, dtype=np.float32) > np.float32(0.5)).astype(np.int32)


def row_keys(p01, n_probes, key_mode="ref_int16", seed=0, n_multi_rows=None, row0=0):
    """Per-row distinct keys (first = hard key): hashings.py:66-92 + utils.pyx:26-31 semantics.

    Rows >= n_multi_rows are single-probe (Indexer.hash trailing-batch rule, indexer.py:51-53).
    Returns (keys int64 [n, n_probes] first-occurrence order, nkeys int32 [n]).
    """
    p01 = np.ascontiguousarray(p01, dtype=np.float32)
    n, H = p01.shape
    if n_multi_rows is None:
        n_multi_rows = n
    keys = np.zeros((n, n_probes), dtype=np.int64)
    nk = np.zeros(n, dtype=np.int32)
    lib().oracle_rows_keys(_p(p01, c_f32p), ctypes.c_int64(n), H, n_probes, 0 if key_mode == "ref_int16" else 1,
                           ctypes.c_uint64(seed), ctypes.c_int64(row0), ctypes.c_int64(n_multi_rows),
                           _p(keys, c_i64p), _p(nk, c_i32p))
    return keys, nk


def philox(seed, c0, c1, c2, c3):
    out = (ctypes.c_uint32 * 4)()
    lib().oracle_philox4x32(ctypes.c_uint64(seed), ctypes.c_uint32(c0), ctypes.c_uint32(c1), ctypes.c_uint32(c2),
                            ctypes.c_uint32(c3), out)
    return [int(v) for v in out]


# ----------------------------------------------------------------------------- index build
def build_index(indexes):
    """nlsh/indexer.py:6-24 verbatim semantics: {key: ascending int64 rows}; multi-key sets allowed."""
    out = {}
    for row, key_set in enumerate(indexes):
        for key in key_set:
            out.setdefault(int(key), []).append(row)
    return {k: np.asarray(v, dtype=np.int64) for k, v in out.items()}


def build_csr(keys):
    """One key per row -> (perm int32 [N], uniq_keys int64 [nb] ascending, offsets int64 [nb+1])."""
    keys = np.ascontiguousarray(keys, dtype=np.int64)
    n = len(keys)
    perm = np.empty(n, dtype=np.int32)
    uniq = np.empty(max(n, 1), dtype=np.int64)
    offs = np.empty(max(n, 1) + 1, dtype=np.int64)
    nb = lib().oracle_build_csr(_p(keys, c_i64p), ctypes.c_int64(n), _p(perm, c_i32p), _p(uniq, c_i64p),
                                _p(offs, c_i64p))
    return perm, uniq[:nb].copy(), offs[:nb + 1].copy()


# ----------------------------------------------------------------------------- distances / scan
def distances(q, corpus, rows, metric="l2", f64=False):
    corpus = np.ascontiguousarray(corpus, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    o32 = np.empty(len(rows), dtype=np.float32)
    o64 = np.empty(len(rows), dtype=np.float64) if f64 else None
    lib().oracle_distances(_p(q, c_f32p), _p(corpus, c_f32p), corpus.shape[1], _p(rows, c_i32p),
                           ctypes.c_int64(len(rows)), 0 if metric == "l2" else 1, _p(o32, c_f32p),
                           _p(o64, c_f64p) if f64 else ctypes.cast(None, c_f64p))
    return (o32, o64) if f64 else o32


def query_batch(corpus, perm, uniq_keys, offsets, queries, qkeys, nkeys, k, metric="l2", simd=False):
    """Gather + distance + top-k for every query (indexer.py:62-95) on a CSR index.
    simd=True: the AVX2 form (8 candidates per lane set, same bits; what bench.py's cpu_baseline times).

    Returns (dist fp32 [Q,k] +inf padded, idx int32 [Q,k] -1 padded, ncand int64 [Q]);
    order = (distance, row id) ascending.
    """
    corpus = np.ascontiguousarray(corpus, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    perm = np.ascontiguousarray(perm, dtype=np.int32)
    uniq_keys = np.ascontiguousarray(uniq_keys, dtype=np.int64)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    qkeys = np.ascontiguousarray(qkeys, dtype=np.int64)
    nkeys = np.ascontiguousarray(nkeys, dtype=np.int32)
    Q = queries.shape[0]
    P = qkeys.shape[1] if qkeys.ndim == 2 and qkeys.shape[1] > 0 else 1
    if qkeys.size == 0:
        qkeys = np.zeros((Q, 1), dtype=np.int64)
    od = np.empty((Q, k), dtype=np.float32)
    oi = np.empty((Q, k), dtype=np.int32)
    nc = np.empty(Q, dtype=np.int64)
    fn = lib().oracle_query_batch_simd if simd else lib().oracle_query_batch
    fn(_p(corpus, c_f32p), corpus.shape[1], _p(perm, c_i32p), _p(uniq_keys, c_i64p),
                             _p(offsets, c_i64p), ctypes.c_int64(len(uniq_keys)), _p(queries, c_f32p),
                             ctypes.c_int64(Q), _p(qkeys, c_i64p), _p(nkeys, c_i32p), P, k,
                             0 if metric == "l2" else 1, _p(od, c_f32p), _p(oi, c_i32p), _p(nc, c_i64p))
    return od, oi, nc


def keys_from_lists(key_lists):
    """List of per-query key lists (iteration order) -> (qkeys int64 [Q,P], nkeys int32 [Q])."""
    Q = len(key_lists)
    P = max([len(k) for k in key_lists] + [1])
    qk = np.zeros((Q, P), dtype=np.int64)
    nk = np.zeros(Q, dtype=np.int32)
    for i, ks in enumerate(key_lists):
        nk[i] = len(ks)
        qk[i, :len(ks)] = list(ks)
    return qk, nk


# ----------------------------------------------------------------------------- metrics
def calculate_recall(y_true, y_pred, reduce_func=None):
    """nlsh/metrics.py:4-25: |set(true) & set(pred)| / len(true) per query."""
    assert len(y_true) == len(y_pred)
    rec = [len(set(int(t) for t in yt) & set(int(p) for p in yp)) / len(yt) for yt, yp in zip(y_true, y_pred)]
    return reduce_func(rec) if reduce_func is not None else rec


# ----------------------------------------------------------------------------- Indexer restatement
class OracleIndexer:
    """nlsh/indexer.py:27-96 restated on numpy arrays (hasher = explicit weights).

    compat=True keeps the reference quirks: int16 key wrap (F2), trailing partial batch
    single-probe (F6), `<k` fallback = last key's bucket rows (F7).
    """

    def __init__(self, Ws, bs, corpus, metric="l2", act="sigmoid", key_mode="ref_int16", seed=0):
        self.Ws, self.bs, self.act, self.key_mode, self.seed = Ws, bs, act, key_mode, seed
        self.metric = metric
        self.corpus = np.ascontiguousarray(corpus, dtype=np.float32)
        keys, _ = self.hash_arrays(self.corpus, hash_times=1)               # indexer.py:36-38
        self.corpus_keys = keys[:, 0].copy()
        self.perm, self.uniq_keys, self.offsets = build_csr(self.corpus_keys)

    @classmethod
    def from_keys(cls, corpus, corpus_keys, metric="l2", key_mode="ref_int16"):
        """An oracle index over rows whose bucket keys are GIVEN (one per row, as recorded from another run of the same hash):
        the scan stage of nlsh/indexer.py:56-96 on an identical index, independent of hashing.  Full-size parity
        tests use it with the device's keys, so that the oracle does not have to push 10^6..10^8 rows through its scalar MLP."""
        self = cls.__new__(cls)
        self.Ws = self.bs = None
        self.act, self.key_mode, self.seed, self.metric = "sigmoid", key_mode, 0, metric
        self.corpus = np.ascontiguousarray(corpus, dtype=np.float32)
        self.corpus_keys = np.ascontiguousarray(corpus_keys, dtype=np.int64)
        self.perm, self.uniq_keys, self.offsets = build_csr(self.corpus_keys)
        return self

    @property
    def index2row(self):
        return {int(k): self.perm[self.offsets[i]:self.offsets[i + 1]].astype(np.int64)
                for i, k in enumerate(self.uniq_keys)}

    def probs(self, x):
        return head_probs(mlp_forward(x, self.Ws, self.bs), self.act)

    def hash_arrays(self, x, batch_size=4096, hash_times=1):
        n = x.shape[0]
        n_multi = (n // batch_size) * batch_size                            # indexer.py:43-53 (F6)
        _, p01 = self.probs(x)
        return row_keys(p01, hash_times, self.key_mode, self.seed, n_multi_rows=n_multi)

    def hash(self, x, batch_size=4096, hash_times=1):
        keys, nk = self.hash_arrays(x, batch_size, hash_times)
        return [set(int(v) for v in keys[i, :nk[i]]) for i in range(len(nk))]

    def query_with_keys(self, queries, key_lists, k=10, simd=False):
        """Scan stage on injected key lists (each in the set-iteration order the caller saw).
        simd=True: the AVX2/OpenMP form of the scan (bit-identical to the scalar one, tests/test_oracle_golden.py)."""
        qk, nk = keys_from_lists(key_lists)
        od, oi, nc = query_batch(self.corpus, self.perm, self.uniq_keys, self.offsets, queries, qk, nk, k,
                                 self.metric, simd=simd)
        i2r = None
        results = []
        for q in range(len(key_lists)):
            if nc[q] >= k:
                results.append([int(v) for v in oi[q]])
            else:                                                           # indexer.py:92-93 (F7)
                if i2r is None:
                    i2r = self.index2row
                last = list(key_lists[q])[-1] if len(key_lists[q]) else None
                rows = i2r.get(int(last), np.zeros(0, np.int64)) if last is not None else np.zeros(0, np.int64)
                results.append([int(v) for v in rows])
        return results, [int(c) for c in nc], od, oi

    def query(self, queries, k=10, hash_times=10):
        keys, nk = self.hash_arrays(queries, hash_times=hash_times)
        key_lists = [list(set(int(v) for v in keys[i, :nk[i]])) for i in range(len(nk))]
        res, nc, _, _ = self.query_with_keys(queries, key_lists, k)
        return res, nc
