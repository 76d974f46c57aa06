"""BASELINE.json configs[2], [3] and [4] as WORKLOADS (the pieces are covered in test_gpu_parity.py; these run the
whole path at the configured shapes):

* configs[2] GloVe-1.2M-shaped: 1,183,514 x 100-d, cosine (nlsh/data.py:99-109), learned 24-bit hash with FULL-width
  keys (eval.py:49-53 `_binarr_to_int`, no int16 wrap), 10k queries;
* configs[3] SIFT1M split into 8 bucket shards (one GPU emulates the 8 ranks: same Indexer per shard, same merge kernel
  the all-gather feeds) -- merged result == the single index, bitwise;
* configs[4] Deep100M-shaped at the largest N that keeps the test in seconds (2M x 96-d), 32-bit full-width keys:
  keys >= 2^31 travel as negative int32 bit patterns (index2row / _rows_of_key name them as non-negative ints).

Parity = size-independent properties on all queries + the oracle: on ALL 10^4 queries of configs[2] (its AVX2/OpenMP scan,
pinned bit-identical to the scalar form: counts exact, cosine distances <= 1e-4, ids equal except where the k-th distances tie)
and on a 64-query slice through the scalar form."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import check_scan_properties, dev, make_hashing
from nlsh_amd import synth
from oracle import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CKPT_DIR = os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "checkpoints")


def _load_ckpt(name):
    arrs = np.load(os.path.join(CKPT_DIR, name))
    return [arrs[f"W{i}"] for i in range(3)], [arrs[f"b{i}"] for i in range(3)]


# ------------------------------------------------------------------------------------------- configs[2]
@pytest.fixture(scope="module")
def glove():
    from nlsh_amd.data import Glove
    from nlsh_amd.indexer import Indexer
    N, Q, d, H = 1_183_514, 10_000, 100, 24
    corpus = synth.glove_manifold(N, d, seed=synth.SEED_DATA)
    queries = synth.glove_manifold(Q, d, seed=synth.SEED_QUERY)
    Ws, bs = _load_ckpt("glove_manifold_h24.npz")
    hashing = make_hashing(d, (256, 256), H, Ws, bs, compat=False)
    cg, qg = dev(corpus), dev(queries)
    ix = Indexer(hashing, cg, Glove.distance, compat=False)
    return dict(corpus=corpus, queries=queries, cg=cg, qg=qg, Ws=Ws, bs=bs, hashing=hashing, ix=ix)


def test_glove_1m2_cosine_24bit_workload(glove):
    from nlsh_amd import _capi
    from nlsh_amd.indexer import Indexer
    from nlsh_amd.data import Glove
    ix, qg, cg = glove["ix"], glove["qg"], glove["cg"]
    k, P, seed = 10, 10, 2024
    assert ix.metric == "cosine" and ix._hashing.key_mode == _capi.KEY_FULL
    assert ix.row_stride == 100 and ix.inv_norm is not None
    keys, nkeys = ix.hash_device(qg, hash_times=P, seed=seed)
    # full-width 24-bit keys: non-negative, beyond the int16 range the compat mode would wrap them into
    assert int(keys.min()) >= 0 and int(keys.max()) < (1 << 24) and int(ix.corpus_keys.max()) > 32767
    # compat=False hashes every row with hash_times (no F6 single-probe tail)
    assert int(nkeys[8192:].max()) > 1
    dist, idx, nc, _ = ix.scan_tensors(qg, keys, nkeys, k=k)
    check_scan_properties(ix, qg, cg, keys, nkeys, dist, idx, nc, k, "cosine")
    # the three schedules answer identically up to fp32 summation order (the chooser picks the tiled schedule here since r03:
    # reuse 6.9 pairs per row, size-biased bucket 82 rows -- Indexer.choose_algo)
    assert ix.last_algo == _capi.SCAN_BUCKET_TILED
    for algo in ("bucket", "query"):
        other = Indexer(ix._hashing, cg, Glove.distance, compat=False, algo=algo)
        d2, i2, n2, _ = other.scan_tensors(qg, keys, nkeys, k=k)
        assert torch.equal(n2, nc)
        both = (idx >= 0) & (i2 >= 0)
        assert bool(((d2 - dist).abs() <= 2e-5)[both].all())
        differ = (i2 != idx).any(1)
        assert float(differ.float().mean()) < 0.01
        # where id lists differ, the distance profiles still agree: only near-ties re-ordered
        assert bool(((d2 - dist).abs()[differ] <= 2e-5).all())
    # idempotence
    again = ix.scan_tensors(qg, keys, nkeys, k=k)
    assert torch.equal(again[0], dist) and torch.equal(again[1], idx)


def test_glove_oracle_slice_and_index(glove):
    ix = glove["ix"]
    qg = glove["qg"][:64]
    keys, nkeys = ix._hashing.hash_device(qg, n=10, seed=5)
    dist, idx, nc, _ = ix.scan_tensors(qg, keys, nkeys, k=10)
    ck = ix.corpus_keys.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    perm, uniq, offs = oracle.build_csr(ck)
    assert np.array_equal(perm, ix.perm.cpu().numpy())
    assert np.array_equal(offs, ix.offsets.cpu().numpy())
    od, oi, onc = oracle.query_batch(glove["corpus"], perm, uniq, offs, glove["queries"][:64],
                                     keys.cpu().numpy().astype(np.int64) & 0xFFFFFFFF, nkeys.cpu().numpy(), 10, "cosine")
    assert np.array_equal(nc.cpu().numpy(), onc)
    dh, ih = dist.cpu().numpy(), idx.cpu().numpy()
    fin = np.isfinite(od)
    assert np.array_equal(np.isfinite(dh), fin)
    assert np.all(np.abs(dh[fin] - od[fin]) <= 2e-5)
    for a, b, da, db in zip(ih, oi, dh, od):      # ids may differ only where the two distance profiles tie to tolerance
        for j in np.nonzero(a != b)[0]:
            assert abs(da[j] - db[j]) <= 2e-5
    # the corpus keys themselves on a slice: oracle forward + full-width pack
    z = oracle.mlp_forward(glove["corpus"][:4096], glove["Ws"], glove["bs"])
    _, p01 = oracle.head_probs(z)
    ko, _ = oracle.row_keys(p01, 1, "full")
    assert np.array_equal(ko[:, 0].astype(np.int64) & 0xFFFFFFFF, ck[:4096])


def test_glove_oracle_on_all_queries(glove):
    """VERDICT r04 item 3, configs[2]: all 10^4 queries against the oracle.  Candidate counts exact; cosine distances within 1e-4
    (BASELINE.json's tolerance; measured ~1e-6: the device pre-divides the query by its norm and multiplies by the row's inverse norm,
    nlsh/data.py:99-109 divides by the product); id lists identical except where the two distance profiles tie to that tolerance,
    and then the fp64 distance profiles of the two lists agree (helpers.assert_lists_differ_only_at_ties)."""
    from helpers import assert_lists_differ_only_at_ties
    ix, qg = glove["ix"], glove["qg"]
    k, P, seed = 10, 10, 2024
    keys, nkeys = ix.hash_device(qg, hash_times=P, seed=seed)
    dist, idx, nc, _ = ix.scan_tensors(qg, keys, nkeys, k=k)
    ck = ix.corpus_keys.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    ox = oracle.OracleIndexer.from_keys(glove["corpus"], ck, metric="cosine", key_mode="full")
    kh, nh = keys.cpu().numpy().astype(np.int64) & 0xFFFFFFFF, nkeys.cpu().numpy()
    od, oi, onc = oracle.query_batch(ox.corpus, ox.perm, ox.uniq_keys, ox.offsets, glove["queries"], kh, nh, k, "cosine", simd=True)
    assert np.array_equal(nc.cpu().numpy(), onc)
    dh, ih = dist.cpu().numpy(), idx.cpu().numpy()
    fin = np.isfinite(od)
    assert np.array_equal(np.isfinite(dh), fin) and np.array_equal(ih >= 0, oi >= 0)
    assert np.all(np.abs(dh[fin] - od[fin]) <= 1e-4)
    differ = np.nonzero((ih != oi).any(1))[0]
    assert len(differ) < 0.01 * len(ih), len(differ)
    for q in differ:                      # different members of a tie, nothing else
        n = int((oi[q] >= 0).sum())
        assert_lists_differ_only_at_ties(ih[q, :n], oi[q, :n], glove["queries"][q], glove["corpus"], "cosine", rtol=1e-4)
    # the reference-typed lists of the facade are these rows (compat=False: short lists hold the candidates there are)
    ids, ncl = ix.query(qg, k=k, hash_times=P, seed=seed)
    assert ncl == onc.tolist()
    assert ids == [[int(v) for v in row if v >= 0] for row in ih]
    print(f"[oracle, GloVe-1.2M] all {len(ih)} queries: counts exact, max |d - d_oracle| {np.abs(dh[fin] - od[fin]).max():.2e}, "
          f"{len(differ)} id lists differ at ties")


def test_glove_recall_and_reference_return_types(glove):
    from nlsh_amd.data import brute_force_topk
    from nlsh_amd.metrics import calculate_recall
    ix = glove["ix"]
    gt = brute_force_topk(glove["qg"], glove["cg"], 10, "cosine").cpu().numpy()
    ids, nc = ix.query(glove["qg"], k=10, hash_times=10)
    assert isinstance(ids, list) and isinstance(ids[0], list) and isinstance(nc, list) and isinstance(nc[0], int)
    rec = calculate_recall(list(gt), ids, np.mean)
    assert 0.30 < rec < 0.55, rec                      # profiles/r01_bench_glove.json: 0.40 at ~317 candidates/query
    assert 150 < np.mean(nc) < 700
    stats = ix.bucket_stats()
    assert stats["n_indexes"] > 50_000                 # ~104k buckets of ~11 rows
    assert len(ix.index2row) == stats["n_indexes"]
    assert all(0 <= key < (1 << 24) for key in list(ix.index2row)[:1000])


# ------------------------------------------------------------------------------------------- configs[3]
def test_sift1m_eight_bucket_shards_merge_equals_single_index():
    from nlsh_amd.data import SIFT
    from nlsh_amd.distributed import merge_topk_device, plan_bucket_shards
    from nlsh_amd.indexer import Indexer
    N, Q, d, G, k, P = 1_000_000, 10_000, 128, 8, 10, 10
    corpus, mean, std = synth.standardise(synth.sift_manifold(N, d, seed=synth.SEED_DATA))
    queries, _, _ = synth.standardise(synth.sift_manifold(Q, d, seed=synth.SEED_QUERY), mean, std)
    Ws, bs = _load_ckpt("sift1m_manifold_h16.npz")
    hashing = make_hashing(d, (256, 256), 16, Ws, bs)
    cg, qg = dev(corpus), dev(queries)
    single = Indexer(hashing, cg, SIFT.distance)
    d1, i1, n1, _ = single.query_tensors(qg, k=k, hash_times=P, seed=31)
    owner, stats = plan_bucket_shards(single.corpus_keys, G)
    keys_all, nc_all, rows = [], [], []
    for r in range(G):
        sel = torch.nonzero(owner == r).view(-1)
        sh = Indexer(hashing, cg[sel], SIFT.distance, row_ids=sel.int(), schedule_stats=stats)
        assert sh.choose_algo(Q, P) == single.choose_algo(Q, P)
        _, _, nc, k64 = sh.query_tensors(qg, k=k, hash_times=P, seed=31, want_keys=True)
        keys_all.append(k64); nc_all.append(nc); rows.append(int(sel.numel()))
        del sh
    assert sum(rows) == N and max(rows) - min(rows) < 0.02 * N / G          # snake deal balances the rows
    packed = torch.cat([torch.stack(keys_all), torch.stack(nc_all).long()[:, :, None]], dim=2)
    dm, im, nm = merge_topk_device(packed, k)
    assert torch.equal(nm, n1) and torch.equal(im, i1) and torch.equal(dm, d1)


# ------------------------------------------------------------------------------------------- configs[4]
@pytest.fixture(scope="module")
def deep():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scale_deep100m as gen
    from nlsh_amd import training
    from nlsh_amd.data import SIFT
    from nlsh_amd.encoders import MultiLayerRelu
    from nlsh_amd.hashings import MultivariateBernoulli
    from nlsh_amd.indexer import Indexer
    N, Q, d, H = 2_000_000, 10_000, 96, 32
    device = torch.device("cuda", 0)
    params = gen._manifold_params(device, d)
    cg = gen.deep_manifold_device(0, N, d, 1234, device, params)
    qg = gen.deep_manifold_device(0, Q, d, 4321, device, params)
    torch.manual_seed(0)
    hashing = MultivariateBernoulli(MultiLayerRelu(d, [256, 256]), H, None, compat=False)
    sample = cg[::10][:200_000].contiguous()
    knn = training.self_knn(sample, 10)
    training.fit_triplet(hashing, sample, knn, n_steps=600, margin=1.0, log=lambda s: None)
    hashing.train_mode(False)
    ix = Indexer(hashing, cg, SIFT.distance, compat=False)
    Ws = [w.cpu().numpy() for w, _ in hashing.linear_stack()]
    bs = [b.cpu().numpy() for _, b in hashing.linear_stack()]
    return dict(cg=cg, qg=qg, ix=ix, hashing=hashing, Ws=Ws, bs=bs)


def test_deep_96d_32bit_full_width_keys_workload(deep):
    from nlsh_amd import _capi
    ix, qg, cg = deep["ix"], deep["qg"], deep["cg"]
    k, P = 10, 10
    assert ix._hashing.key_mode == _capi.KEY_FULL and ix.dim == 96
    ckeys = ix.corpus_keys
    assert int((ckeys < 0).sum()) > 0 and int((ckeys >= 0).sum()) > 0       # codes on both sides of 2^31
    assert int(torch.unique(ckeys >> 16).numel()) > 64                       # the high 16 bits carry information (no int16 wrap)
    assert bool((ix.uniq_keys[1:] > ix.uniq_keys[:-1]).all())                # CSR in ascending SIGNED int32 order
    keys, nkeys = ix.hash_device(qg, hash_times=P, seed=9)
    assert int((keys < 0).sum()) > 0
    results = {}
    for algo in ("query", "tiled"):
        ix.algo = algo
        results[algo] = ix.scan_tensors(qg, keys, nkeys, k=k)[:3]
    ix.algo = None
    dist, idx, nc = results["tiled"]
    assert int((nc > 0).sum()) > 0.5 * qg.shape[0]
    check_scan_properties(ix, qg, cg, keys, nkeys, dist, idx, nc, k, "l2")
    dq, iq, nq = results["query"]
    assert torch.equal(nq, nc)
    assert bool(((dq - dist).abs() <= 2e-5)[(idx >= 0)].all())
    # reference-typed views name buckets by the NON-NEGATIVE code (eval.py:49-53)
    names = list(ix.index2row)
    assert min(names) >= 0 and max(names) >= (1 << 31)
    big = next(n for n in names if n >= (1 << 31))
    assert ix._rows_of_key(big) == ix.index2row[big].cpu().tolist()
    sets = ix.hash(qg[:256], hash_times=P)
    assert all(0 <= v < (1 << 32) for s_ in sets for v in s_)
    assert any(v >= (1 << 31) for s_ in sets for v in s_)
    # query() (lists) agrees with the tensors it is built from; short lists follow the non-compat rule
    ids, ncl = ix.query(qg[:2048], k=k, hash_times=1)
    kk, nk = ix.hash_device(qg[:2048], hash_times=1)
    d2, i2, n2, _ = ix.scan_tensors(qg[:2048], kk, nk, k=k)
    assert ncl == n2.cpu().tolist()
    assert ids == [[int(v) for v in row if v >= 0] for row in i2.cpu().numpy()]


def test_deep_oracle_slice(deep):
    ix = deep["ix"]
    corpus = deep["cg"].cpu().numpy()
    queries = deep["qg"][:64].cpu().numpy()
    keys, nkeys = ix._hashing.hash_device(deep["qg"][:64], n=10, seed=3)
    dist, idx, nc, _ = ix.scan_tensors(deep["qg"][:64], keys, nkeys, k=10)
    ck = ix.corpus_keys.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    perm, uniq, offs = oracle.build_csr(ck)
    # the oracle orders buckets by the unsigned code, the device CSR by the signed bit pattern: same buckets, same rows
    assert len(uniq) == ix.n_buckets
    od, oi, onc = oracle.query_batch(corpus, perm, uniq, offs, queries, keys.cpu().numpy().astype(np.int64) & 0xFFFFFFFF,
                                     nkeys.cpu().numpy(), 10, "l2")
    assert np.array_equal(nc.cpu().numpy(), onc)
    if ix.last_algo == 2:       # tiled: k-ascending fmaf chain == oracle, bit for bit
        assert np.array_equal(idx.cpu().numpy(), oi)
        assert np.array_equal(dist.cpu().numpy().view(np.uint32), od.view(np.uint32))
    else:
        fin = np.isfinite(od)
        assert np.all(np.abs(dist.cpu().numpy()[fin] - od[fin]) <= 2e-5)
    # keys of a corpus slice: oracle forward + full-width pack, as unsigned codes
    z = oracle.mlp_forward(corpus[:2048], deep["Ws"], deep["bs"])
    _, p01 = oracle.head_probs(z)
    ko, _ = oracle.row_keys(p01, 1, "full")
    assert np.array_equal(ko[:, 0].astype(np.int64) & 0xFFFFFFFF, ck[:2048])
