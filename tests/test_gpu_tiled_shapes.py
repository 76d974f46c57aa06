"""Every hand-scheduled task body of the LDS-tiled scan (csrc/scan_bucket.hip: l2_task<NW, NQ, NTL, METRIC>, NQ = queries
a wave holds 0..4, NTL = 64-row tiles of the task 1..4, L2 and cosine = 40 bodies, plus the odd-chunk tail of l2_kblock and
the fat-stage geometry of short segments) on a deterministic index: buckets of chosen sizes, each probed by a chosen
number of queries, so that the task table provably holds every (queries, tiles) shape -- asserted from the table the PLAN
phase left in the workspace -- and the results are held to the oracle (L2: bit-identical; cosine: <= 2e-5)."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import assert_lists_differ_only_at_ties, dev, make_hashing
from nlsh_amd import _capi, synth
from oracle import oracle

pytestmark = pytest.mark.gpu

SIZES = (1, 20, 40, 64, 65, 100, 128, 150, 192, 230, 256, 300, 513)  # rows per bucket: single-stage tasks (20 rows; 40 up to 100-d), 1..4 tiles, 2- and 3-segment buckets
GROUPS = tuple(range(1, 17)) + (17, 21, 33)                          # queries probing a bucket: every nq 1..16, and > 16 (several groups)
Q = 64


def _build(d, metric, seed):
    """Corpus + per-row bucket keys + per-query key lists with every (size, group) combination present once."""
    rng = np.random.default_rng(seed)
    buckets = [(s, m) for s in SIZES for m in GROUPS]
    N = sum(s for s, _ in buckets)
    gen = synth.sift_like if metric == "l2" else synth.glove_like
    corpus, queries = gen(N, d, seed=seed), gen(Q, d, seed=seed + 1)
    corpus[N // 3:N // 3 + 30] = corpus[:30]                        # exact distance ties across buckets
    keys = np.repeat(np.arange(len(buckets), dtype=np.int32) * 3 - 400, [s for s, _ in buckets])   # signed, gaps between keys
    order = rng.permutation(N)                                      # rows of a bucket are scattered over the corpus
    corpus_keys = np.empty(N, np.int32)
    corpus_keys[order] = keys
    key_lists = [[] for _ in range(Q)]
    for b, (s, m) in enumerate(buckets):
        for q in rng.choice(Q, size=m, replace=False):
            key_lists[q].append(int(b * 3 - 400))
    for q in range(Q):
        rng.shuffle(key_lists[q])
    key_lists[5].insert(2, 999999)                                  # an unknown key in the middle of a list
    assert max(len(ks) for ks in key_lists) <= _capi.MAX_PROBES
    return corpus, queries, corpus_keys, key_lists, buckets


def _task_table(indexer, Qn, P, k, d, full=False):
    """(nq, nrows) of every task the last tiled scan laid out, read from the workspace through the diagnostic layout call;
    full=True: the whole table [tasks, 4] plus the tasks' query ids and packed row ranges, each [tasks, 16]."""
    L = _capi.lib()
    max_tasks = indexer._last_max_tasks                              # the table the LAST launch ran with (it may be trimmed afterwards)
    off_task, off_q, off_r = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
    _capi.check(L.nlsh_scan_workspace_layout(Qn, P, k, max_tasks, indexer.n_buckets, d, _capi.SCAN_BUCKET_TILED,
                                             ctypes.byref(off_task), ctypes.byref(off_q), ctypes.byref(off_r)))
    ws = next(w for (stream, bm), w in indexer._ws.items() if bm)
    n_tasks = int(indexer.last_status.cpu()[0])
    tab = ws[off_task.value:off_task.value + 16 * n_tasks].view(torch.int32).view(-1, 4).cpu().numpy()
    if not full:
        return tab[:, 1], tab[:, 3]
    assert off_r.value == off_q.value + 4                            # one interleaved table of {query id, row range} records
    qr = ws[off_q.value:off_q.value + 128 * n_tasks].view(torch.int32).view(-1, 16, 2).cpu().numpy()
    return tab, qr[:, :, 0], qr[:, :, 1]


@pytest.mark.parametrize("d", [128, 100, 96, 72])
@pytest.mark.parametrize("metric", ["l2", "cosine"])
def test_every_tiled_task_body_matches_the_oracle(metric, d):
    from nlsh_amd.data import Glove, SIFT
    from nlsh_amd.indexer import Indexer
    k = 10
    corpus, queries, corpus_keys, key_lists, buckets = _build(d, metric, seed=1000 + d)
    Ws, bs = synth.make_weights([d, 32, 16], seed=d)
    hashing = make_hashing(d, (32,), 16, Ws, bs, compat=False)     # the hash is not used: keys are injected on both sides
    indexer = Indexer(hashing, dev(corpus), SIFT.distance if metric == "l2" else Glove.distance, compat=False, algo="tiled",
                      corpus_keys=dev(corpus_keys), window_rows=0)   # one task list per bucket: the shapes below are per bucket
    res, nc, dist, idx = indexer.query_with_keys(dev(queries), key_lists, k=k)
    assert indexer.last_algo == _capi.SCAN_BUCKET_TILED

    # ---- the task table holds every (queries per wave 0..4) x (tiles 1..4) shape
    P = max(len(ks) for ks in key_lists)
    nq, nrows = _task_table(indexer, Q, P, k, d)
    assert nq.min() >= 1 and nq.max() == 16 and nrows.min() >= 1 and nrows.max() == 256
    shapes = set()
    for a, r in zip(nq.tolist(), nrows.tolist()):
        nt = (r + 63) // 64
        for wave in range(4):                                       # queries are dealt round-robin over the 4 waves (NLSH_SLOT)
            shapes.add((max(0, min(4, (a - wave + 3) // 4)), nt))
    assert shapes == {(q_, t_) for q_ in range(5) for t_ in range(1, 5)}, sorted(shapes)
    # one-tile tasks come in two forms (r05): the whole task in ONE stage when rows x 16-byte chunks fit 1024 slots (l2_task_single), the
    # fat two-stage form otherwise -- both present for every number of queries a wave can hold
    d4 = (d + 3) // 4
    for one_stage in (True, False):
        got = set()
        for a, r in zip(nq.tolist(), nrows.tolist()):
            if r <= 64 and (r * d4 <= 1024) == one_stage:
                got |= {max(0, min(4, (a - wave + 3) // 4)) for wave in range(4)}
        assert got == set(range(5)), (one_stage, sorted(got))
    assert {(a, (r + 63) // 64) for a, r in zip(nq.tolist(), nrows.tolist())} >= {(a, t) for a in range(1, 17) for t in range(1, 5)}
    # expected task count: per bucket ceil(m / 16) query groups x ceil(size / 256) segments
    assert len(nq) == sum(((m + 15) // 16) * ((s + 255) // 256) for s, m in buckets)

    # ---- results vs the oracle on the same candidate sets
    perm, uniq, offs = oracle.build_csr(corpus_keys.astype(np.int64))
    assert np.array_equal(indexer.perm.cpu().numpy(), perm)
    qk, nk = oracle.keys_from_lists(key_lists)
    od, oi, onc = oracle.query_batch(corpus, perm, uniq, offs, queries, qk, nk, k, metric)
    assert nc == onc.tolist()
    dist, idx = dist.cpu().numpy(), idx.cpu().numpy()
    if metric == "l2":                                              # same k-ascending fmaf chain as the oracle: bit for bit
        assert np.array_equal(idx, oi)
        assert np.array_equal(dist.view(np.uint32), od.view(np.uint32))
    else:
        fin = np.isfinite(od)
        assert np.array_equal(np.isfinite(dist), fin)
        assert np.all(np.abs(dist[fin] - od[fin]) <= 2e-5 * np.maximum(1.0, np.abs(od[fin])))
        for q in range(Q):
            n = min(k, int(onc[q]))
            assert_lists_differ_only_at_ties(idx[q][:n], oi[q][:n], queries[q], corpus, metric)
            assert np.all(idx[q][n:] == -1)


def test_repeated_probe_keys_probe_a_bucket_once():
    """A query's keys are a set (nlsh/utils.pyx:27-31).  The facade de-duplicates caller-supplied lists; a raw key table
    with repeats handed to the C ABI is de-duplicated by the plan kernels of every schedule: same result as the set."""
    from nlsh_amd.data import SIFT
    from nlsh_amd.indexer import Indexer
    d, k = 128, 10
    corpus, queries, corpus_keys, key_lists, _ = _build(d, "l2", seed=77)
    Ws, bs = synth.make_weights([d, 32, 16], seed=1)
    hashing = make_hashing(d, (32,), 16, Ws, bs, compat=False)
    qd = dev(queries)
    key_lists = [ks[:30] for ks in key_lists]                       # twice the keys must still fit one scan call (64 probes)
    for algo in ("query", "bucket", "tiled"):
        indexer = Indexer(hashing, dev(corpus), SIFT.distance, compat=False, algo=algo, corpus_keys=dev(corpus_keys))
        base = indexer.query_with_keys(qd, key_lists, k=k)
        doubled = [ks + ks[::-1] for ks in key_lists[:20]] + [ks[:1] * 3 + ks for ks in key_lists[20:]]
        again = indexer.query_with_keys(qd, doubled, k=k)           # facade path: de-duplicated on the host, first occurrence kept
        assert again[0] == base[0] and again[1] == base[1]
        # raw table with repeats straight to nlsh_scan_topk (the facade is bypassed)
        P = max(len(ks) for ks in key_lists)
        tab = np.zeros((Q, 2 * P), np.int32)
        cnt = np.zeros((Q,), np.int32)
        for q, ks in enumerate(key_lists):
            row = [kk for pair in zip(ks, ks) for kk in pair]       # every key twice, adjacent
            tab[q, :len(row)] = np.asarray(row, np.int64).astype(np.int32)
            cnt[q] = len(row)
        dist, idx, ncand, _ = indexer.scan_tensors(qd, dev(tab), dev(cnt), k=k)
        assert torch.equal(idx, base[3]) and torch.equal(dist, base[2]) and ncand.cpu().tolist() == base[1]


@pytest.mark.parametrize("algo", ["bucket", "tiled"])
def test_non_zero_workspace_head_is_detected_not_trusted(algo):
    """Workspace contract of the bucket-major schedules (include/nlsh_hip.h): the per-bucket pair counters at the head of
    the workspace must be zero on entry.  A violated contract is detected on the device (status[1] = 2, the batch gets no
    task, nothing is addressed through a stale count) and surfaces as NLSH_E_WORKSPACE -- never a fault, never wrong ids."""
    from nlsh_amd.data import SIFT
    from nlsh_amd.indexer import Indexer
    d, k = 128, 10
    corpus, queries, corpus_keys, key_lists, _ = _build(d, "l2", seed=91)
    Ws, bs = synth.make_weights([d, 32, 16], seed=1)
    hashing = make_hashing(d, (32,), 16, Ws, bs, compat=False)
    indexer = Indexer(hashing, dev(corpus), SIFT.distance, compat=False, algo=algo, corpus_keys=dev(corpus_keys))
    qd = dev(queries)
    good = indexer.query_with_keys(qd, key_lists, k=k)
    for poison in ("ff", "plus3", "minus_plus"):
        (wkey, ws), = [(kk, w) for kk, w in indexer._ws.items() if kk[1]]
        head = ws[:4 * indexer.n_buckets].view(torch.int32)
        assert int(head.abs().sum()) == 0                            # every completed call hands the counters back as zeros
        if poison == "ff":
            ws.fill_(0xFF)                                           # an uninitialised buffer
        elif poison == "plus3":
            head[7] = 3                                              # a stale count on one bucket
        else:
            head[3], head[11] = -2, 2                                # stale counts that cancel in the sum
        with pytest.raises(_capi.NlshHipError) as err:
            indexer.query_with_keys(qd, key_lists, k=k)
        assert err.value.code == _capi.E_WORKSPACE
        torch.cuda.synchronize()
        assert not indexer._ws                                       # the poisoned buffers were dropped ...
        again = indexer.query_with_keys(qd, key_lists, k=k)          # ... and the next call runs on a fresh zeroed one
        assert again[0] == good[0] and again[1] == good[1] and torch.equal(again[3], good[3])


@pytest.mark.parametrize("d", [128, 100])
def test_folded_l2_form_is_an_opt_in_within_the_stated_tolerance(d):
    """NLSH_METRIC_L2_EPS_FOLDED (`Indexer(l2_form="folded")`): sqrt(sum(((q + 1e-6) - c)^2)) instead of the reference's
    sqrt(sum(((q - c) + 1e-6)^2)) (nlsh/data.py:201) -- one rounding per element differs, so it is NOT bit-identical to the oracle
    and never the default.  Its bar is BASELINE.json's: every returned distance within 1e-4 * max(1, d) of the fp64 value,
    candidate counts exact, id lists equal to the oracle's except where distances tie to that tolerance; on every task body."""
    from nlsh_amd.data import SIFT
    from nlsh_amd.indexer import Indexer
    k = 10
    corpus, queries, corpus_keys, key_lists, _ = _build(d, "l2", seed=2000 + d)
    # standardised like the bench workload (values of order 1, where eps = 1e-6 is a few ulps: on raw SIFT-like integers the two
    # forms round to the same fp32 distances almost everywhere)
    corpus, mean, std = synth.standardise(corpus)
    queries, _, _ = synth.standardise(queries, mean, std)
    Ws, bs = synth.make_weights([d, 32, 16], seed=d)
    hashing = make_hashing(d, (32,), 16, Ws, bs, compat=False)
    exact = Indexer(hashing, dev(corpus), SIFT.distance, compat=False, algo="tiled", corpus_keys=dev(corpus_keys))
    assert exact.l2_form == "exact"                                  # the default is the oracle's operation order
    folded = Indexer(hashing, dev(corpus), SIFT.distance, compat=False, algo="tiled", corpus_keys=dev(corpus_keys), l2_form="folded")
    r0 = exact.query_with_keys(dev(queries), key_lists, k=k)
    r1 = folded.query_with_keys(dev(queries), key_lists, k=k)
    assert r1[1] == r0[1]                                           # candidate counts: exact
    d0, d1 = r0[2].cpu().numpy(), r1[2].cpu().numpy()
    i0, i1 = r0[3].cpu().numpy(), r1[3].cpu().numpy()
    fin = np.isfinite(d0)
    assert np.array_equal(np.isfinite(d1), fin)
    assert np.all(np.abs(d1[fin] - d0[fin]) <= 1e-4 * np.maximum(1.0, np.abs(d0[fin])))
    assert not np.array_equal(d1.view(np.uint32), d0.view(np.uint32))   # it really is another rounding (else it would be the default)
    perm, uniq, offs = oracle.build_csr(corpus_keys.astype(np.int64))
    i2r = {int(u): perm[offs[j]:offs[j + 1]] for j, u in enumerate(uniq)}
    for q in range(Q):
        n = min(k, r0[1][q])
        rows = np.concatenate([i2r.get(kk, np.zeros(0, np.int32)) for kk in dict.fromkeys(key_lists[q])]) if key_lists[q] else np.zeros(0, np.int32)
        _, d64 = oracle.distances(queries[q], corpus, rows, "l2", f64=True)
        from helpers import check_topk_against_candidates
        check_topk_against_candidates(i1[q], d1[q], rows, d64, k, rtol=1e-4)
        # where the two forms name different ids, the candidates tie to the tolerance (on integer SIFT-like rows the folded form
        # sees EXACT ties where the reference's +eps-per-element separates mirror-image differences by ~1e-6 relative)
        assert_lists_differ_only_at_ties(i1[q][:n], i0[q][:n], queries[q], corpus, "l2", rtol=1e-4)
    # schedules without a folded form answer the same request with the exact one
    other = Indexer(hashing, dev(corpus), SIFT.distance, compat=False, algo="query", corpus_keys=dev(corpus_keys), l2_form="folded")
    r2 = other.query_with_keys(dev(queries), key_lists, k=k)
    assert r2[1] == r0[1] and np.all(np.abs(r2[2].cpu().numpy()[fin] - d0[fin]) <= 2e-5 * np.maximum(1.0, np.abs(d0[fin])))


def _greedy_cells(offsets, W, span=1024):
    """Host restatement of nlsh_build_cells' packing rule (include/nlsh_hip.h): CSR order, restarted every `span` buckets; a
    bucket of more than W rows is its own cell, smaller ones join the open cell while its rows stay <= W."""
    nb = len(offsets) - 1
    starts = np.zeros(nb, np.int32)
    cur = -1
    for b in range(nb):
        if b % span == 0:
            cur = -1
        s_, e_ = int(offsets[b]), int(offsets[b + 1])
        if e_ - s_ > W:
            starts[b], cur = 1, -1
        elif cur < 0 or e_ - cur > W:
            starts[b], cur = 1, s_
    cell_of = np.cumsum(starts) - 1
    cell_offsets = np.append(offsets[:-1][starts == 1], offsets[-1])
    return cell_of.astype(np.int32), cell_offsets.astype(np.int32)


@pytest.mark.parametrize("window", [64, 128, 256])
def test_build_cells_is_the_greedy_packing(window):
    from nlsh_amd.data import SIFT
    from nlsh_amd.indexer import Indexer
    rng = np.random.default_rng(window)
    # 5,000 buckets (several 1024-bucket spans): mostly tiny, some around the window size, a few far larger
    sizes = np.concatenate([rng.integers(1, 20, 3000), rng.integers(1, 2 * window, 1500), rng.integers(257, 3000, 500)])
    rng.shuffle(sizes)
    keys = np.repeat(np.arange(len(sizes), dtype=np.int32) * 2 - 5000, sizes)
    d = 8
    corpus = rng.standard_normal((len(keys), d)).astype(np.float32)
    Ws, bs = synth.make_weights([d, 8, 16], seed=1)
    ix = Indexer(make_hashing(d, (8,), 16, Ws, bs, compat=False), dev(corpus), SIFT.distance, compat=False, corpus_keys=dev(keys))
    cell_of, cell_offsets, cell_order, nc = ix.cells(window)
    offs = ix.offsets.cpu().numpy()
    want_of, want_offs = _greedy_cells(offs, window)
    assert nc == len(want_offs) - 1 and nc < ix.n_buckets
    assert np.array_equal(cell_of.cpu().numpy(), want_of)
    assert np.array_equal(cell_offsets.cpu().numpy(), want_offs)
    rows = np.diff(want_offs)
    order = cell_order.cpu().numpy()
    assert sorted(order.tolist()) == list(range(nc))                 # a permutation of the cells ...
    assert np.all(np.diff(rows[order]) <= 0)                         # ... by descending rows
    multi = np.bincount(want_of) > 1
    assert multi.any() and rows[multi].max() <= window               # shared windows exist and none exceeds the window


@pytest.mark.parametrize("scale", [1e-25, 3e-16, 1e-12, 1.0, 3e18])
def test_tiled_l2_at_extreme_magnitudes_is_still_the_oracle_bit_for_bit(scale):
    """The r05 epilogue takes square roots without hipcc's range scaling when every accumulator of a list is >= 2^-96 and falls back to
    sqrtf otherwise (one wave-uniform guard per list).  Both sides of the guard, the zero distance of a query that IS a corpus row
    (sqrt(d) * 1e-6 through the eps term), sums that underflow to denormals and sums that overflow to +inf must all give the oracle's
    bits and its (distance, id) order."""
    from nlsh_amd.data import SIFT
    from nlsh_amd.indexer import Indexer
    rng = np.random.default_rng(11)
    d, k, N, Qn = 64, 10, 3000, 48
    corpus = (rng.standard_normal((N, d)) * scale).astype(np.float32)
    queries = (rng.standard_normal((Qn, d)) * scale).astype(np.float32)
    twins = rng.choice(N, 16, replace=False)
    queries[:16] = corpus[twins]                                       # exact matches: every term is (0 + eps)^2
    keys = rng.integers(0, 12, N).astype(np.int32) * 7 - 20           # 12 buckets of ~250 rows: several tiles per task
    Ws, bs = synth.make_weights([d, 8, 16], seed=2)
    ix = Indexer(make_hashing(d, (8,), 16, Ws, bs, compat=False), dev(corpus), SIFT.distance, compat=False, algo="tiled", corpus_keys=dev(keys))
    key_lists = [[int(v) * 7 - 20 for v in rng.choice(12, 3, replace=False)] for _ in range(Qn)]
    for i, row in enumerate(twins):                                    # a twin's bucket is among the buckets its query probes
        key_lists[i] = list(dict.fromkeys([int(keys[row])] + key_lists[i]))
    res, nc, dist, idx = ix.query_with_keys(dev(queries), key_lists, k=k)
    perm, uniq, offs = oracle.build_csr(keys.astype(np.int64))
    qk, nk = oracle.keys_from_lists(key_lists)
    od, oi, onc = oracle.query_batch(corpus, perm, uniq, offs, queries, qk, nk, k, "l2")
    assert nc == onc.tolist()
    assert np.array_equal(dist.cpu().numpy().view(np.uint32), od.view(np.uint32))
    assert np.array_equal(idx.cpu().numpy(), oi)
    if scale == 1.0:
        assert np.all(od[:16, 0] < 1e-4) and np.array_equal(oi[:16, 0], twins)   # the twins came first, at sqrt(d) * 1e-6


@pytest.mark.parametrize("metric,d", [("cosine", 100), ("l2", 72)])
def test_padded_rows_change_no_result_bit(metric, d):
    """`Indexer(row_align=32)` starts every row of the bucket-sorted copy on a 128-byte line (100-d: 400 -> 512 bytes per row; an
    r05 traffic experiment).  The kernels walk d, not the stride: every schedule must return the bits of the
    packed layout."""
    from nlsh_amd.data import Glove, SIFT
    from nlsh_amd.indexer import Indexer
    corpus, queries, corpus_keys, key_lists, _ = _build(d, metric, seed=4000 + d)
    Ws, bs = synth.make_weights([d, 32, 16], seed=d)
    hashing = make_hashing(d, (32,), 16, Ws, bs, compat=False)
    dist_fn = SIFT.distance if metric == "l2" else Glove.distance
    for algo in ("tiled", "query"):
        base = Indexer(hashing, dev(corpus), dist_fn, compat=False, algo=algo, corpus_keys=dev(corpus_keys))
        wide = Indexer(hashing, dev(corpus), dist_fn, compat=False, algo=algo, corpus_keys=dev(corpus_keys), row_align=32)
        assert wide.row_stride == (d + 31) // 32 * 32 > base.row_stride == (d + 3) // 4 * 4
        r0, r1 = base.query_with_keys(dev(queries), key_lists, k=10), wide.query_with_keys(dev(queries), key_lists, k=10)
        assert r0[0] == r1[0] and r0[1] == r1[1]
        assert torch.equal(r0[3], r1[3]) and torch.equal(r0[2].view(torch.int32), r1[2].view(torch.int32))


def test_foreign_cells_with_a_window_wider_than_a_segment_are_refused():
    """the tiled scan assumes that a window shared by several buckets fits ONE 256-row segment (what nlsh_build_cells
    guarantees).  Cells from elsewhere that pack small buckets into a wider window are refused on the device (status flag 3 ->
    NlshHipError E_INVALID), never scanned with rows missing; the workspace stays usable afterwards."""
    from nlsh_amd.data import SIFT
    from nlsh_amd.indexer import Indexer
    rng = np.random.default_rng(5)
    d, k = 32, 5
    sizes = np.full(40, 20)                                         # 40 buckets of 20 rows: 800 rows
    keys = np.repeat(np.arange(40, dtype=np.int32) * 3, sizes)
    corpus = rng.standard_normal((len(keys), d)).astype(np.float32)
    queries = rng.standard_normal((8, d)).astype(np.float32)
    Ws, bs = synth.make_weights([d, 8, 16], seed=1)
    ix = Indexer(make_hashing(d, (8,), 16, Ws, bs, compat=False), dev(corpus), SIFT.distance, compat=False, algo="tiled", corpus_keys=dev(keys),
                 window_rows=64)
    key_lists = [[int(3 * ((q * 5 + j) % 40)) for j in range(4)] for q in range(8)]
    good = ix.query_with_keys(dev(queries), key_lists, k=k)
    # foreign cells: buckets 0..19 (400 rows) in ONE cell, the rest one cell each
    cell_of = np.concatenate([np.zeros(20, np.int32), np.arange(1, 21, dtype=np.int32)])
    offs = ix.offsets.cpu().numpy()
    cell_offsets = np.concatenate([[0], offs[20:]]).astype(np.int32)
    nc = 21
    order = np.argsort(-np.diff(cell_offsets), kind="stable").astype(np.int32)
    ix._cells[64] = (dev(cell_of), dev(cell_offsets), dev(order), nc)
    ix._max_tasks.clear()
    with pytest.raises(_capi.NlshHipError) as err:
        ix.query_with_keys(dev(queries), key_lists, k=k)
    assert err.value.code == _capi.E_INVALID and "256-row segment" in str(err.value)
    del ix._cells[64]                                               # back to the library's own cells: same answer as before
    again = ix.query_with_keys(dev(queries), key_lists, k=k)
    assert again[0] == good[0] and again[1] == good[1]


@pytest.mark.parametrize("metric", ["l2", "cosine"])
@pytest.mark.parametrize("d", [128, 100])
def test_shared_windows_change_no_result_bit(metric, d):
    """Small-bucket packing (nlsh_build_cells): consecutive small buckets share a row window and its tasks, each (task, query) with
    the row range of its own bucket.  Every distance is the same fmaf chain over the same row, so ids, distances and counts must be
    bit-identical to the one-task-list-per-bucket layout for every window size -- L2 and cosine alike -- while the task count drops;
    the ranges the PLAN phase wrote are checked against the CSR."""
    from nlsh_amd.data import Glove, SIFT
    from nlsh_amd.indexer import Indexer
    k = 10
    rng = np.random.default_rng(7 * d)
    # tiny buckets in runs (shared windows), mid-size ones, and big ones that break the runs; every bucket probed by 0..20 queries
    sizes = np.concatenate([rng.integers(1, 12, 400), rng.integers(12, 130, 120), rng.integers(130, 257, 30), rng.integers(257, 700, 12)])
    rng.shuffle(sizes)
    N = int(sizes.sum())
    gen = synth.sift_like if metric == "l2" else synth.glove_like
    corpus, queries = gen(N, d, seed=d), gen(Q, d, seed=d + 1)
    corpus[N // 2:N // 2 + 40] = corpus[:40]                        # exact distance ties across buckets and windows
    bkeys = np.arange(len(sizes), dtype=np.int32) * 5 - 1300
    corpus_keys = np.repeat(bkeys, sizes)                           # rows in bucket order: windows are runs of the corpus itself
    key_lists = [[] for _ in range(Q)]
    for b in range(len(sizes)):
        for q in rng.choice(Q, size=int(rng.integers(0, 21)) if b % 3 else int(rng.integers(0, 3)), replace=False):
            if len(key_lists[q]) < _capi.MAX_PROBES:
                key_lists[q].append(int(bkeys[b]))
    for q in range(Q):
        rng.shuffle(key_lists[q])
    Ws, bs = synth.make_weights([d, 32, 16], seed=d)
    hashing = make_hashing(d, (32,), 16, Ws, bs, compat=False)
    dist_fn = SIFT.distance if metric == "l2" else Glove.distance
    P = max(len(ks) for ks in key_lists)
    base = Indexer(hashing, dev(corpus), dist_fn, compat=False, algo="tiled", corpus_keys=dev(corpus_keys), window_rows=0)
    r0 = base.query_with_keys(dev(queries), key_lists, k=k)
    n0 = int(base.last_status.cpu()[0])
    offs = base.offsets.cpu().numpy()
    uniq = base.uniq_keys.cpu().numpy()
    for window in (64, 128, 256):
        ix = Indexer(hashing, dev(corpus), dist_fn, compat=False, algo="tiled", corpus_keys=dev(corpus_keys), window_rows=window)
        r1 = ix.query_with_keys(dev(queries), key_lists, k=k)
        assert ix.last_window == window
        assert r1[0] == r0[0] and r1[1] == r0[1]
        assert torch.equal(r1[3], r0[3]) and torch.equal(r1[2].view(torch.int32), r0[2].view(torch.int32))
        tab, tq, tr = _task_table(ix, Q, P, k, d, full=True)
        assert len(tab) < n0                                         # fewer, fuller tasks
        cell_of, cell_offsets, _, nc = ix.cells(window)
        coffs = cell_offsets.cpu().numpy()
        shared = 0
        for (pair0, nq, row0, nrows), qs_, rs_ in zip(tab.tolist(), tq.tolist(), tr.tolist()):
            assert 1 <= nq <= 16 and 1 <= nrows <= 256
            c = int(np.searchsorted(coffs, row0, side="right")) - 1
            assert coffs[c] <= row0 < coffs[c + 1]
            spans = set()
            for slot in range(nq):
                lo, hi = rs_[slot] & 0xFFFF, rs_[slot] >> 16
                assert 0 <= lo < hi <= nrows
                b = int(np.searchsorted(offs, row0 + lo, side="right")) - 1
                # the range is exactly the part of ONE bucket that lies in this task's rows, and the query probes that bucket
                assert max(offs[b], row0) == row0 + lo and min(offs[b + 1], row0 + nrows) == row0 + hi
                assert int(uniq[b]) in key_lists[qs_[slot]]
                spans.add((lo, hi))
            shared += len(spans) > 1
        assert shared > 0                                            # windows shared by queries of DIFFERENT buckets were exercised
