"""GPU parity tests: HIP path (through the C ABI / ctypes) vs the CPU oracle and vs golden vectors
produced by the unmodified reference.  Integer/index work is bit-exact; fp32 tolerances are stated
at each assert.  Run on the MI355X box: python -m pytest tests -m gpu."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import assert_lists_differ_only_at_ties, G, cases, check_topk_against_candidates, dev, make_hashing
from nlsh_amd import synth
from oracle import oracle

pytestmark = pytest.mark.gpu


# ----------------------------------------------------------------------------- a4/a5 bit packing
def test_pack_codes_golden_and_oracle():
    from nlsh_amd.utils import hash_codes, pack_codes
    g = json.load(open(os.path.join(G, "g1_hash_codes.json")))
    for case in g["hand"] + g["random"]:
        codes = np.asarray(case["codes"], dtype=np.int32)
        assert [sorted(s) for s in hash_codes(codes)] == case["ref_int16_sets"]
        full = pack_codes(codes, "full").cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        assert full.tolist() == case["full_keys"]
    assert hash_codes(np.zeros((0, 1, 8), np.int32)) == []
    rng = np.random.default_rng(3)
    for H in (1, 7, 16, 17, 31, 32):
        codes = rng.integers(0, 2, size=(257, 9, H)).astype(np.int32)
        for mode in ("ref_int16", "full"):
            got = pack_codes(codes, mode).cpu().numpy().astype(np.int64)
            if mode == "full":
                got &= 0xFFFFFFFF
            assert np.array_equal(got, oracle.pack_keys(codes, mode))


# ----------------------------------------------------------------------------- a1-a3 hasher forward
@pytest.mark.parametrize("case", cases.G2_CASES, ids=[c[0] for c in cases.G2_CASES])
def test_encode_hash_vs_oracle_and_reference(case):
    name, d, hidden, H, tanh, two_layer, kind = case
    g = np.load(os.path.join(G, "g2_hasher.npz"))
    Ws, bs = cases.g2_weights(case)
    x = cases.g2_inputs(kind, d)
    hashing = make_hashing(d, hidden, H, Ws, bs, tanh=tanh, two_layer=two_layer)
    z, probs, code = hashing.forward_device(dev(x))
    z, probs, code = z.cpu().numpy(), probs.cpu().numpy(), code.cpu().numpy().view(np.uint32)
    # fp32 MFMA chain == k-ordered fmaf chain of the oracle: bit exact
    zo = oracle.mlp_forward(x, Ws, bs)
    assert np.array_equal(z.view(np.uint32), zo.view(np.uint32)), f"z not bit-exact, max diff {np.abs(z - zo).max()}"
    raw_o, p01_o = oracle.head_probs(zo, "tanh" if tanh else "sigmoid")
    assert np.abs(probs - raw_o).max() <= 2e-7          # device expf/tanhf vs glibc: <= 1-2 ulp of a value in (0,1)
    bits_o = oracle.hard_bits(p01_o)
    code_o = oracle.pack_keys(bits_o[:, None, :], "full")[:, 0]
    flip_rows = code.astype(np.int64) != code_o
    assert np.all(np.abs(zo[flip_rows]).min(axis=1) < 1e-6) if flip_rows.any() else True
    # against the reference's own bits (BLAS summation order): flips only where |z| is at rounding level
    ref_bits = g[name + "/bits"].astype(np.int64)
    my_bits = (code[:, None].astype(np.int64) >> np.arange(H - 1, -1, -1)[None, :]) & 1
    flips = my_bits != ref_bits
    scale = max(np.abs(g[name + "/z"]).max(), 1.0)
    assert np.all(np.abs(g[name + "/z"][flips]) < 1e-4 * scale)
    assert np.abs(z - g[name + "/z"]).max() <= 2e-5 * scale
    assert np.abs(probs - g[name + "/probs"]).max() <= 5e-6
    # keys (compat int16) for rows without flips equal the reference's
    keys, nkeys = hashing.hash_device(dev(x), n=1)
    same = ~flips.any(axis=1)
    assert np.array_equal(keys.cpu().numpy()[same, 0], g[name + "/key_ref_int16"][same])
    assert np.all(nkeys.cpu().numpy() == 1)
    sets = hashing.hash(dev(x), 1)
    assert all(len(s) == 1 for s in sets)
    with pytest.raises(ValueError):
        hashing.hash(dev(x), 0)                          # hashings.py:83


# the three forms of the encoder: <= 4096 rows 16-row workgroups of 16x16x4 tiles, <= 16384 rows 32-row workgroups on one LDS image,
# beyond that the 128-row index-build form -- every one the same k-ascending fmaf chain as the oracle, at and around the switch-overs and
# around one / two workgroups per CU (8192 rows)
@pytest.mark.parametrize("n_rows", [1, 15, 16, 17, 63, 64, 65, 1000, 4096, 4097, 5000, 8192, 8193, 8209, 10000, 12287, 12288, 12289, 16384, 16385, 20001])
def test_encode_hash_ragged_sizes_bit_exact(n_rows):
    d, hidden, H = 128, (256, 256), 16
    Ws, bs = synth.make_weights([d] + list(hidden) + [H], seed=5)
    x, _, _ = synth.standardise(synth.sift_like(max(n_rows, 8), d, seed=21))
    x = x[:n_rows]
    hashing = make_hashing(d, hidden, H, Ws, bs)
    z, _, _ = hashing.forward_device(dev(x))
    assert np.array_equal(z.cpu().numpy().view(np.uint32), oracle.mlp_forward(x, Ws, bs).view(np.uint32))


@pytest.mark.parametrize("dims", [[25, 96, 8], [50, 64, 64, 12], [200, 256, 256, 24], [96, 320, 40, 32],
                                  [128, 600, 16], [100, 256, 256, 256, 256, 20], [128, 33, 1], [960, 256, 256, 32], [1024, 64, 7]])
def test_encode_hash_odd_architectures(dims):
    Ws, bs = synth.make_weights(dims, seed=7)
    x = synth.glove_like(130, dims[0], seed=9)
    hashing = make_hashing(dims[0], dims[1:-1], dims[-1], Ws, bs)
    z, _, code = hashing.forward_device(dev(x))
    zo = oracle.mlp_forward(x, Ws, bs)
    assert np.array_equal(z.cpu().numpy().view(np.uint32), zo.view(np.uint32))


def test_multiprobe_keys_of_the_headline_batch_shape_match_oracle_sampler():
    """10^4 rows x 10 probes with the F6 rule (rows >= 8192 single-probe): the batch shape of the headline, 313 32-row workgroups
    -- keys, key counts and first-occurrence order against the oracle's Philox sampler, row for row."""
    d, hidden, H, n, rows = 128, (256, 256), 16, 10, 10_000
    Ws, bs = synth.make_weights([d] + list(hidden) + [H], seed=11)
    x, _, _ = synth.standardise(synth.sift_like(rows, d, seed=33))
    hashing = make_hashing(d, hidden, H, Ws, bs)
    _, probs, _ = hashing.forward_device(dev(x))
    keys, nkeys = hashing.hash_device(dev(x), n=n, n_multi_rows=8192, seed=99)
    ko, no = oracle.row_keys(probs.cpu().numpy(), n, "ref_int16", seed=99, n_multi_rows=8192)
    kd, nd = keys.cpu().numpy().astype(np.int64), nkeys.cpu().numpy()
    assert np.array_equal(nd, no) and np.all(no[8192:] == 1) and no[:8192].max() > 1
    live = np.arange(n)[None, :] < no[:, None]
    assert np.array_equal(kd[live], ko[live])


def test_multiprobe_keys_match_oracle_sampler():
    d, hidden, H, n = 128, (64, 64), 12, 10
    Ws, bs = synth.make_weights([d] + list(hidden) + [H], seed=300)
    x, _, _ = synth.standardise(synth.sift_like(300, d, seed=32))
    for compat, mode in ((True, "ref_int16"), (False, "full")):
        hashing = make_hashing(d, hidden, H, Ws, bs, compat=compat)
        _, probs, _ = hashing.forward_device(dev(x))
        p01 = probs.cpu().numpy()
        for n_multi in (300, 256, 0):
            keys, nkeys = hashing.hash_device(dev(x), n=n, n_multi_rows=n_multi, seed=1234, row0=7)
            ko, no = oracle.row_keys(p01, n, mode, seed=1234, n_multi_rows=n_multi, row0=7)
            kd = keys.cpu().numpy().astype(np.int64)
            if mode == "full":
                kd &= 0xFFFFFFFF
            assert np.array_equal(nkeys.cpu().numpy(), no)
            for r in range(300):
                assert np.array_equal(kd[r, :no[r]], ko[r, :no[r]])
            assert np.all(no[n_multi:] == 1)                 # F6: trailing rows single-probe


def test_indexer_hash_batching_rule_g3():
    g = json.load(open(os.path.join(G, "g3_batching.json")))
    Ws, bs = synth.make_weights([128, 64, 64, 12], seed=300)
    corpus, _, _ = synth.standardise(synth.sift_like(64, 128, seed=31))
    from nlsh_amd.data import SIFT
    from nlsh_amd.indexer import Indexer
    hashing = make_hashing(128, (64, 64), 12, Ws, bs)
    indexer = Indexer(hashing, dev(corpus), SIFT.distance)
    for case in g:
        x, _, _ = synth.standardise(synth.sift_like(case["Q"], 128, seed=32))
        sets = indexer.hash(dev(x), batch_size=case["batch_size"], hash_times=case["hash_times"])
        hard = indexer.hash(dev(x), batch_size=case["batch_size"], hash_times=1)
        assert [int(list(h)[0]) for h in hard] == case["hard_keys"]          # reference's hard keys
        n_multi = (case["Q"] // case["batch_size"]) * case["batch_size"]
        for i, s in enumerate(sets):
            assert case["hard_keys"][i] in s
            assert (len(s) == 1) if i >= n_multi else (1 <= len(s) <= case["hash_times"])


# ----------------------------------------------------------------------------- a7 index build
def test_build_index_reference_vector_and_random():
    from nlsh_amd.indexer import build_csr_device, build_index
    gj = json.load(open(os.path.join(G, "g4_build_index.json")))["ref_test"]
    got = build_index([set(s) for s in gj["input"]], cuda=True)
    assert {str(k): v.cpu().tolist() for k, v in got.items()} == gj["expected"]
    g = np.load(os.path.join(G, "g4_build_index.npz"))
    perm, uniq, offs = build_csr_device(dev(g["keys"]))
    assert np.array_equal(uniq.cpu().numpy(), g["uniq_keys"])
    assert np.array_equal(np.diff(offs.cpu().numpy()), g["sizes"])
    assert np.array_equal(perm.cpu().numpy(), g["rows_concat"])
    rng = np.random.default_rng(0)
    for n, lo, hi in ((1, -5, 5), (1000, -32768, 32768), (200000, -2 ** 31, 2 ** 31), (70000, 0, 3)):
        keys = rng.integers(lo, hi, size=n).astype(np.int32)
        perm, uniq, offs = build_csr_device(dev(keys))
        po, uo, oo = oracle.build_csr(keys)
        assert np.array_equal(perm.cpu().numpy(), po) and np.array_equal(uniq.cpu().numpy(), uo)
        assert np.array_equal(offs.cpu().numpy(), oo)


# ----------------------------------------------------------------------------- a8-a10 scan on injected keys
@pytest.mark.parametrize("algo", ["query", "bucket", "tiled"])
@pytest.mark.parametrize("name", ["l2_small", "cos_small", "l2_k3"])
def test_scan_golden_injected_keys(name, algo):
    from nlsh_amd.data import Glove, SIFT
    from nlsh_amd.indexer import Indexer
    meta = json.load(open(os.path.join(G, "g5_query.json")))[name]
    g = np.load(os.path.join(G, "g5_query.npz"))
    corpus, queries, Ws, bs = cases.g5_inputs(meta)
    cos = meta["metric"] == "cosine"
    hashing = make_hashing(meta["d"], (64, 64), meta["H"], Ws, bs, tanh=cos)
    # our own hard keys agree with the reference's except where |z| sits at rounding level (F5 flip policy, tested above) ...
    ours, _ = hashing.hash_device(dev(corpus), n=1)
    assert (ours.view(-1).cpu().numpy() == g[name + "/corpus_keys"]).mean() > 0.999
    # ... and the index under test is built on the REFERENCE's recorded keys, so the scan always runs on the reference's
    # buckets (identical candidate sets) and this test can never skip
    indexer = Indexer(hashing, dev(corpus), Glove.distance if cos else SIFT.distance, algo=algo,
                      corpus_keys=dev(g[name + "/corpus_keys"].astype(np.int32)))
    assert np.array_equal(indexer.corpus_keys.cpu().numpy(), g[name + "/corpus_keys"])
    res, nc, dist, idx = indexer.query_with_keys(dev(queries), meta["injected_iter"], k=meta["k"])
    assert nc == g[name + "/ncand"].tolist()                      # n_candidates: exact
    off = g[name + "/cand_off"]
    dist, idx = dist.cpu().numpy(), idx.cpu().numpy()
    for q in range(meta["Q"]):
        rows = g[name + "/cand_rows"][off[q]:off[q + 1]]
        dref = g[name + "/cand_dist"][off[q]:off[q + 1]]
        ref_ids = meta["result_ids"][q]
        if nc[q] < meta["k"]:
            assert res[q] == ref_ids                               # F7 fallback list: exact
        else:
            cases.assert_topk_equivalent(res[q], ref_ids, rows, dref, meta["k"], 1e-4)
        _, d64 = oracle.distances(queries[q], corpus, rows, meta["metric"], f64=True)
        check_topk_against_candidates(idx[q], dist[q], rows, d64, meta["k"])
    rec = oracle.calculate_recall(list(g[name + "/ground_truth"]), res)
    # every list was shown above to equal the reference's up to k-th-distance ties: recall can move by at most the
    # number of ids those ties substituted
    swapped = sum(len(set(res[q]) - set(meta["result_ids"][q])) for q in range(meta["Q"]))
    assert abs(np.mean(rec) - meta["mean_recall"]) <= swapped / (meta["k"] * meta["Q"]) + 1e-12


SCAN_CASES = [
    # metric, d, N, Q, H, k, seg_rows, P
    ("l2", 128, 20000, 64, 4, 10, 64, 3),      # 16 fat buckets, many segments -> merge kernel
    ("l2", 128, 20000, 64, 10, 10, 0, 10),
    ("cosine", 100, 15000, 50, 6, 10, 128, 6),
    ("l2", 96, 9000, 40, 8, 64, 64, 5),        # k = NLSH_MAX_K
    ("cosine", 25, 5000, 33, 5, 1, 0, 4),      # LPR=16, d % 4 != 0
    ("l2", 50, 5000, 33, 5, 7, 0, 4),
    ("l2", 200, 6000, 20, 6, 10, 64, 4),       # LPR=64
    ("cosine", 300, 3000, 20, 4, 5, 0, 3),     # VPL=2
    ("l2", 600, 1500, 10, 3, 10, 64, 2),       # VPL=4
]


@pytest.mark.parametrize("algo", ["query", "bucket", "tiled"])
@pytest.mark.parametrize("metric,d,N,Q,H,k,seg,P", SCAN_CASES)
def test_scan_vs_oracle_shapes(metric, d, N, Q, H, k, seg, P, algo):
    from nlsh_amd.data import Glove, SIFT
    from nlsh_amd.indexer import Indexer
    rng = np.random.default_rng(d * 7 + N)
    gen = synth.sift_like if metric == "l2" else synth.glove_like
    corpus, queries = gen(N, d, seed=d), gen(Q, d, seed=d + 1)
    corpus[N // 2:N // 2 + 25] = corpus[:25]                      # exact distance ties
    Ws, bs = synth.make_weights([d, 64, H], seed=d)
    hashing = make_hashing(d, (64,), H, Ws, bs, compat=False)
    indexer = Indexer(hashing, dev(corpus), SIFT.distance if metric == "l2" else Glove.distance, compat=False, seg_rows=seg,
                      algo=algo)
    ck = indexer.corpus_keys.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    perm, uniq, offs = oracle.build_csr(ck)
    assert np.array_equal(indexer.perm.cpu().numpy(), perm)
    present = uniq.tolist()
    key_lists = []
    for q in range(Q):
        ks = [int(present[i]) for i in rng.choice(len(present), size=min(P, len(present)), replace=False)]
        if q % 7 == 0:
            ks = ks[:1] + [4000000000 + q] + ks[1:]              # unknown key in the middle (full-width)
        key_lists.append(ks)
    res, nc, dist, idx = indexer.query_with_keys(dev(queries), key_lists, k=k)
    qk, nk = oracle.keys_from_lists(key_lists)
    od, oi, onc = oracle.query_batch(corpus, perm, uniq, offs, queries, qk, nk, k, metric)
    assert nc == onc.tolist()
    dist, idx = dist.cpu().numpy(), idx.cpu().numpy()
    i2r = {int(u): perm[offs[i]:offs[i + 1]] for i, u in enumerate(uniq)}
    exact = 0
    for q in range(Q):
        rows = np.concatenate([i2r.get(kk, np.zeros(0, np.int32)) for kk in key_lists[q]]) if key_lists[q] else np.zeros(0, np.int32)
        _, d64 = oracle.distances(queries[q], corpus, rows, metric, f64=True)
        check_topk_against_candidates(idx[q], dist[q], rows, d64, k)
        n = min(k, int(onc[q]))
        exact += int(assert_lists_differ_only_at_ties(idx[q][:n], oi[q][:n], queries[q], corpus, metric))
        assert np.all(idx[q][n:] == -1)
    if algo == "tiled" and metric == "l2":                        # same k-ascending fmaf chain as the oracle: no near-ties to move
        assert exact == Q


def test_hash_times_100_keys_and_sliced_scan():
    """eval.py:148 sweeps n_samples up to 100: encode_hash generates up to 128 keys per row, the scan takes them in
    slices of 64 and merges -- same keys as the oracle's sampler, same results as its single pass."""
    from nlsh_amd.data import SIFT
    from nlsh_amd.indexer import Indexer
    d, H, N, Q, k, P = 128, 14, 30000, 96, 10, 100
    Ws, bs = synth.make_weights([d, 64, H], seed=77)
    corpus, _, _ = synth.standardise(synth.sift_like(N, d, seed=70))
    queries, _, _ = synth.standardise(synth.sift_like(Q, d, seed=71))
    hashing = make_hashing(d, (64,), H, Ws, bs, compat=False)
    qd = dev(queries)
    keys, nkeys = hashing.hash_device(qd, n=P, seed=5)
    _, probs, _ = hashing.forward_device(qd)
    ko, no = oracle.row_keys(probs.cpu().numpy(), P, "full", seed=5, n_multi_rows=Q)
    kd = keys.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    assert np.array_equal(nkeys.cpu().numpy(), no) and no.max() > 64          # the case needs more than one slice
    for r in range(Q):
        assert np.array_equal(kd[r, :no[r]], ko[r, :no[r]])
    indexer = Indexer(hashing, dev(corpus), SIFT.distance, compat=False)
    dist, idx, nc, k64 = indexer.scan_tensors(qd, keys, nkeys, k=k, want_keys=True)
    ck = indexer.corpus_keys.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    perm, uniq, offs = oracle.build_csr(ck)
    od, oi, onc = oracle.query_batch(corpus, perm, uniq, offs, queries, ko, no, k, "l2")
    assert np.array_equal(nc.cpu().numpy(), onc)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.allclose(dist.cpu().numpy(), od, rtol=2e-5, atol=0) or np.array_equal(np.isinf(dist.cpu().numpy()), np.isinf(od))
    # the merged lists' sort keys are what a shard would all-gather: monotone(dist) << 32 | id
    kk = k64.cpu().numpy().view(np.uint64)
    assert np.array_equal((kk & np.uint64(0xFFFFFFFF)).astype(np.int64)[idx.cpu().numpy() >= 0], idx.cpu().numpy()[idx.cpu().numpy() >= 0])
    assert np.all(kk[:, 1:] >= kk[:, :-1])
    res, ncl = indexer.query(qd, k=k, hash_times=P)                            # the reference-typed call takes the same path
    assert len(res) == Q and all(len(r) <= max(k, 1) or ncl[i] < k for i, r in enumerate(res))


def test_scan_edge_cases():
    from nlsh_amd.data import SIFT
    from nlsh_amd.indexer import Indexer
    d, H = 128, 6
    corpus = synth.sift_like(500, d, seed=1)
    Ws, bs = synth.make_weights([d, 64, H], seed=1)
    hashing = make_hashing(d, (64,), H, Ws, bs)
    indexer = Indexer(hashing, dev(corpus), SIFT.distance)
    # Q = 0
    res, nc = indexer.query(dev(np.zeros((0, d), np.float32)), k=10, hash_times=3)
    assert res == [] and nc == []
    # every key unknown / no keys at all -> C_q = 0 -> [] (reference: candidate_rows.tolist() of the empty default)
    q = synth.sift_like(4, d, seed=2)
    res, nc, dist, idx = indexer.query_with_keys(dev(q), [[123456], [], [7, 8, 9], [-5]], k=10)
    assert res == [[], [], [], []] and nc == [0, 0, 0, 0]
    assert np.all(idx.cpu().numpy() == -1) and np.all(np.isinf(dist.cpu().numpy()))
    # argument validation comes back as exceptions, never as a silent fallback
    from nlsh_amd import _capi
    with pytest.raises(_capi.NlshHipError):
        indexer.scan_tensors(dev(q), torch.zeros((4, 1), dtype=torch.int32, device="cuda"),
                             torch.ones((4,), dtype=torch.int32, device="cuda"), k=65)
    with pytest.raises(_capi.NlshHipError):
        indexer.query(torch.from_numpy(q), k=10)                   # host tensor: no CPU path


def test_task_table_overflow_is_detected_and_retried():
    from nlsh_amd.data import SIFT
    from nlsh_amd.indexer import Indexer
    d, H = 128, 3
    corpus = synth.sift_like(6000, d, seed=3)
    queries = synth.sift_like(80, d, seed=4)
    Ws, bs = synth.make_weights([d, 64, H], seed=3)
    hashing = make_hashing(d, (64,), H, Ws, bs)
    indexer = Indexer(hashing, dev(corpus), SIFT.distance, seg_rows=64)
    for algo in (0, 1, 2):
        indexer._max_tasks[indexer._tkey(algo, 80, 1)] = 5                       # far too small: must grow, not truncate
    ids, nc = indexer.query(dev(queries), k=10, hash_times=1)
    assert indexer._max_tasks[indexer._tkey(indexer.last_algo, 80, 1)] > 5
    # the device-resident form detects it too (check=True) and a batch of another shape gets its own table
    indexer._max_tasks[indexer._tkey(indexer.last_algo, 80, 1)] = 5
    _, i2, n2, _ = indexer.query_tensors(dev(queries), k=10, hash_times=1)
    assert n2.cpu().tolist() == nc and indexer._max_tasks[indexer._tkey(indexer.last_algo, 80, 1)] > 5
    ids40, nc40 = indexer.query(dev(queries[:40]), k=10, hash_times=1)
    assert ids40 == ids[:40] and nc40 == nc[:40] and indexer._tkey(indexer.last_algo, 40, 1) in indexer._max_tasks
    ox = oracle.OracleIndexer(Ws, bs, corpus)
    oids, onc = ox.query(queries, k=10, hash_times=1)
    assert nc == onc
    for q, (a, b) in enumerate(zip(ids, oids)):
        assert_lists_differ_only_at_ties(a, b, queries[q], corpus, "l2")


# ----------------------------------------------------------------------------- end to end (config 0)
def test_sift_small_end_to_end_g7():
    from nlsh_amd.data import SIFT
    from nlsh_amd.indexer import Indexer
    from nlsh_amd.metrics import calculate_recall
    meta = json.load(open(os.path.join(G, "g7_sift_small.json")))
    g = np.load(os.path.join(G, "g7_sift_small.npz"))
    corpus, queries, Ws, bs = cases.g7_inputs()
    hashing = make_hashing(128, (256, 256), 8, Ws, bs)
    indexer = Indexer(hashing, dev(corpus), SIFT.distance)
    sizes = {int(k): int(len(v)) for k, v in indexer.index2row.items()}
    ref_sizes = {int(k): v for k, v in meta["bucket_sizes"].items()}
    moved = sum(abs(sizes.get(k, 0) - ref_sizes.get(k, 0)) for k in set(sizes) | set(ref_sizes))
    assert moved <= 0.002 * meta["N"]
    ids, nc = indexer.query(dev(queries), k=10, hash_times=1)
    assert sum(a == b for a, b in zip(nc, meta["ncand"])) >= 97
    rec = calculate_recall(list(g["ground_truth"]), ids, np.mean)
    assert abs(rec - meta["mean_recall"]) < 0.01
    # oracle on the same inputs: identical keys, identical candidate counts, identical ids
    ox = oracle.OracleIndexer(Ws, bs, corpus)
    assert np.array_equal(ox.corpus_keys, indexer.corpus_keys.cpu().numpy())
    oids, onc = ox.query(queries, k=10, hash_times=1)
    assert nc == onc and sum(a == b for a, b in zip(ids, oids)) >= 98
    # multi-probe query (hash_times=10): Q=100 < 4096 -> every query single-probe in compat mode (F6)
    ids10, nc10 = indexer.query(dev(queries), k=10, hash_times=10)
    assert nc10 == nc


# ----------------------------------------------------------------------------- (e) shards: merge == single GPU
@pytest.mark.parametrize("partition", ["buckets", "rows"])
@pytest.mark.parametrize("G", [2, 3, 8])
def test_sharded_scan_plus_merge_equals_single_index(G, partition):
    from nlsh_amd.data import SIFT
    from nlsh_amd.distributed import corpus_statistics, merge_topk_device, plan_bucket_shards, shard_range
    from nlsh_amd.indexer import Indexer
    N, Q, d, H, k, P = 30000, 200, 128, 7, 10, 6
    corpus, _, _ = synth.standardise(synth.sift_like(N, d, seed=8))
    queries, _, _ = synth.standardise(synth.sift_like(Q, d, seed=9))
    corpus[100:140] = corpus[20000:20040]                          # exact ties across shards
    Ws, bs = synth.make_weights([d, 64, H], seed=8)
    hashing = make_hashing(d, (64,), H, Ws, bs)
    qd, cd = dev(queries), dev(corpus)
    single = Indexer(hashing, cd, SIFT.distance)
    d1, i1, n1, _ = single.query_tensors(qd, k=k, hash_times=P, seed=77)
    owner, stats = plan_bucket_shards(single.corpus_keys, G)        # what ShardedIndexer derives from the gathered keys
    assert stats == corpus_statistics(torch.as_tensor(single.bucket_sizes))
    keys_all, nc_all, seen = [], [], []
    for r in range(G):
        if partition == "rows":
            lo, hi = shard_range(N, r, G)
            sh = Indexer(hashing, cd[lo:hi], SIFT.distance, id_base=lo, schedule_stats=stats)
        else:
            sel = torch.nonzero(owner == r).view(-1)
            sh = Indexer(hashing, cd[sel], SIFT.distance, row_ids=sel.int(), schedule_stats=stats)
            seen.append(set(sh.uniq_keys.cpu().tolist()))
        assert sh.choose_algo(Q, P) == single.choose_algo(Q, P)
        _, _, nc, k64 = sh.query_tensors(qd, k=k, hash_times=P, seed=77, want_keys=True)
        keys_all.append(k64); nc_all.append(nc)
    if partition == "buckets":                                      # every bucket lives whole on exactly one rank
        assert sum(len(s_) for s_ in seen) == single.n_buckets and len(set().union(*seen)) == single.n_buckets
    packed = torch.cat([torch.stack(keys_all), torch.stack(nc_all).long()[:, :, None]], dim=2)
    dm, im, nm = merge_topk_device(packed, k)
    assert torch.equal(nm, n1)                                      # candidate counts add up exactly
    assert torch.equal(im, i1)                                      # same comparator -> identical ids
    assert torch.equal(dm, d1)                                      # and bit-identical distances


def test_bucket_shard_keeps_reference_views():
    """index2row / the F7 fallback of a bucket shard speak GLOBAL row ids."""
    from nlsh_amd.data import SIFT
    from nlsh_amd.distributed import plan_bucket_shards
    from nlsh_amd.indexer import Indexer
    N, d, H = 5000, 128, 6
    corpus, _, _ = synth.standardise(synth.sift_like(N, d, seed=18))
    Ws, bs = synth.make_weights([d, 64, H], seed=18)
    hashing = make_hashing(d, (64,), H, Ws, bs)
    cd = dev(corpus)
    single = Indexer(hashing, cd, SIFT.distance)
    owner, stats = plan_bucket_shards(single.corpus_keys, 2)
    merged = {}
    for r in range(2):
        sel = torch.nonzero(owner == r).view(-1)
        sh = Indexer(hashing, cd[sel], SIFT.distance, row_ids=sel.int(), schedule_stats=stats)
        for key, rows in sh.index2row.items():
            assert key not in merged
            merged[key] = rows.cpu().tolist()
            assert sh._rows_of_key(key) == merged[key]
    assert merged == {k_: v.cpu().tolist() for k_, v in single.index2row.items()}


def test_both_schedules_are_bit_identical():
    """Query-major and bucket-major share the arithmetic: same distances (bitwise), ids, counts."""
    from nlsh_amd.data import Glove, SIFT
    from nlsh_amd.indexer import Indexer
    for metric, d, fn in (("l2", 128, SIFT.distance), ("cosine", 100, Glove.distance), ("l2", 200, SIFT.distance)):
        gen = synth.sift_like if metric == "l2" else synth.glove_like
        corpus, queries = gen(40000, d, seed=5), gen(700, d, seed=6)
        Ws, bs = synth.make_weights([d, 64, 7], seed=5)
        hashing = make_hashing(d, (64,), 7, Ws, bs)
        out = []
        for algo in ("query", "bucket"):
            ix = Indexer(hashing, dev(corpus), fn, algo=algo, seg_rows=128)
            out.append(ix.query_tensors(dev(queries), k=10, hash_times=8, seed=3, want_keys=True))
        for a, b in zip(out[0], out[1]):
            assert torch.equal(a, b)


def test_tiled_schedule_l2_distances_bit_exact_vs_oracle():
    """The LDS-tiled schedule sums each distance as a k-ascending fmaf chain: bit-identical to the
    oracle's scalar loop (oracle_l2), so ids AND distances must equal the oracle's exactly."""
    from nlsh_amd.data import SIFT
    from nlsh_amd.indexer import Indexer
    for d, seg in ((128, 128), (100, 0), (200, 64), (50, 0)):
        corpus, queries = synth.sift_like(30000, d, seed=15), synth.sift_like(400, d, seed=16)
        corpus[200:260] = corpus[20000:20060]                       # exact ties -> id tie-break
        Ws, bs = synth.make_weights([d, 64, 6], seed=15)
        hashing = make_hashing(d, (64,), 6, Ws, bs)
        ix = Indexer(hashing, dev(corpus), SIFT.distance, algo="tiled", seg_rows=seg)
        keys, nkeys = ix.hash_device(dev(queries), hash_times=6, seed=5)
        dist, idx, nc, _ = ix.scan_tensors(dev(queries), keys, nkeys, k=10)
        perm, uniq, offs = oracle.build_csr(ix.corpus_keys.cpu().numpy().astype(np.int64))
        od, oi, onc = oracle.query_batch(corpus, perm, uniq, offs, queries, keys.cpu().numpy().astype(np.int64),
                                         nkeys.cpu().numpy(), 10, "l2")
        assert np.array_equal(nc.cpu().numpy(), onc)
        assert np.array_equal(idx.cpu().numpy(), oi)
        assert np.array_equal(dist.cpu().numpy().view(np.uint32), od.view(np.uint32))


def test_save_then_load_checkpoint_roundtrip(tmp_path):
    """hashing.save() (TorchScript _cpu/_gpu + state dict, reference hashings.py:53-57) -> io loaders ->
    a fresh hashing produces bit-identical keys."""
    from nlsh_amd import io as nio
    from nlsh_amd.encoders import TwoLayer256Relu
    from nlsh_amd.hashings import MultivariateBernoulli
    Ws, bs = synth.make_weights([128, 256, 256, 16], seed=21)
    hashing = make_hashing(128, None, 16, Ws, bs, two_layer=True)
    base = str(tmp_path / "run_1_0.5000")
    hashing.save(base)
    x, _, _ = synth.standardise(synth.sift_like(500, 128, seed=22))
    k0, n0 = hashing.hash_device(dev(x), n=4, seed=9)
    for suffix in ("_cpu.pt", "_gpu.pt", "_state.pt"):
        W2, b2 = nio.load_hasher_weights(base + suffix)
        h2 = nio.hashing_from_weights(W2, b2, compat=True)
        k1, n1 = h2.hash_device(dev(x), n=4, seed=9)
        assert torch.equal(k0, k1) and torch.equal(n0, n1)
    h3 = MultivariateBernoulli(TwoLayer256Relu(128), 16, None)
    h3.load_state(base + "_state.pt")
    h3.train_mode(False)
    k2, _ = h3.hash_device(dev(x), n=4, seed=9)
    assert torch.equal(k0, k2)


def test_pipelined_batches_are_bit_identical_to_sequential_calls():
    """nlsh_amd/pipeline.py: encode + PLAN, SCAN and MERGE of consecutive batches on three streams -- same kernels,
    same arguments."""
    from nlsh_amd.data import Glove, SIFT
    from nlsh_amd.indexer import Indexer
    from nlsh_amd.pipeline import QueryPipeline
    for metric, d, fn, algo in (("l2", 128, SIFT.distance, "tiled"), ("cosine", 100, Glove.distance, "query"), ("l2", 64, SIFT.distance, "bucket")):
        gen = synth.sift_like if metric == "l2" else synth.glove_like
        N, Q, H, k, P = 40000, 512, 9, 10, 8
        corpus = gen(N, d, seed=41)
        Ws, bs = synth.make_weights([d, 64, H], seed=41)
        hashing = make_hashing(d, (64,), H, Ws, bs, compat=False)
        indexer = Indexer(hashing, dev(corpus), fn, compat=False, algo=algo)
        batches = [dev(gen(Q, d, seed=50 + i)) for i in range(5)]
        want = [indexer.query_tensors(b, k=k, hash_times=P, seed=900 + i, want_keys=True) for i, b in enumerate(batches)]
        want = [tuple(t.clone() for t in w) for w in want]
        for depth, graph in ((2, False), (3, False), (2, True), (3, True), (3, "no-graph")):    # staged streams (r03-r05), graph slots (r06), and a graph slot whose capture failed
            if graph == "no-graph":
                os.environ["NLSH_STEP_NO_GRAPH"] = "1"      # read by nlsh_step_create_graph: the slot launches eagerly on its own stream
                graph = True
            else:
                os.environ.pop("NLSH_STEP_NO_GRAPH", None)
            pipe = QueryPipeline(indexer, batches[0], k=k, hash_times=P, depth=depth, want_keys=True, graph=graph)
            assert pipe.graph == (graph and algo != "query")      # graph slots exist for the bucket-major schedules
            got = []
            for i, b in enumerate(batches):
                got.append(pipe.submit(b, seed=900 + i))
                if (i + 1) % depth == 0 or i + 1 == len(batches):   # a slot is overwritten `depth` submits later: read now
                    pipe.synchronize()
                    for j in range(i + 1 - ((i % depth) + 1), i + 1):
                        for a, w in zip(got[j], want[j]):
                            assert torch.equal(a, w), (metric, algo, depth, graph, j)
            assert not pipe.overflowed()
            pipe.close()
        os.environ.pop("NLSH_STEP_NO_GRAPH", None)


@pytest.mark.parametrize("metric,d,algo", [("l2", 128, "tiled"), ("cosine", 100, "tiled"), ("l2", 72, "tiled"), ("l2", 128, "bucket"), ("cosine", 100, "bucket")])
def test_one_call_batch_equals_hash_then_scan(metric, d, algo):
    """ABI v4, `nlsh_query_batch` (what `Indexer.query_tensors` issues): encode_hash with the bucket lookup in its EPILOGUE + the rest
    of the PLAN phase + scan + merge -- five launches -- against the separate calls (`hash_device`, then `scan_tensors`, whose PLAN phase
    looks the caller-supplied key table up with the stand-alone kernel): key tables, key counts, ids, distance BITS, candidate counts and
    64-bit sort keys identical, at batch sizes on both sides of every encoder form (16-row, 32-row, the 32 + 16-row launch), with and
    without the reference's single-probe tail (F6), and for a row RANGE of a batch (row0, the range's share of the multi-probe rows).
    The key table itself is the oracle's (Philox sampler, first-occurrence order)."""
    from nlsh_amd.data import Glove, SIFT
    from nlsh_amd.indexer import Indexer
    gen = synth.sift_like if metric == "l2" else synth.glove_like
    N, H, k = 60000, 11, 10
    corpus = gen(N, d, seed=61)
    Ws, bs = synth.make_weights([d, 64, 64, H], seed=61)
    for compat in (True, False):
        hashing = make_hashing(d, (64, 64), H, Ws, bs, compat=compat)
        indexer = Indexer(hashing, dev(corpus), SIFT.distance if metric == "l2" else Glove.distance, compat=compat, algo=algo)
        for Q, P in ((1, 1), (100, 10), (4097, 10), (8193, 10), (9000, 3), (12288, 10), (333, 64)):
            q = dev(gen(Q, d, seed=70 + Q))
            keys, nkeys = indexer.hash_device(q, hash_times=P, seed=4000 + Q)
            want = indexer.scan_tensors(q, keys, nkeys, k=k, want_keys=True)
            got = indexer._batch_tensors(q, k, P, 4000 + Q, want_keys=True)
            assert indexer._fuses(q, P, indexer.last_algo)
            assert torch.equal(got[4], keys) and torch.equal(got[5], nkeys), (compat, Q, P)
            for a, w in zip(got[:4], want):
                assert torch.equal(a, w), (compat, Q, P)
            if Q in (100, 8193):     # the key table against the oracle's sampler
                _, probs, _ = hashing.forward_device(q)
                ko, no = oracle.row_keys(probs.cpu().numpy(), P, "ref_int16" if compat else "full", seed=4000 + Q, n_multi_rows=indexer._n_multi_rows(Q))
                kd = keys.cpu().numpy().astype(np.int64) & (0xFFFFFFFF if not compat else -1)
                live = np.arange(P)[None, :] < no[:, None]
                assert np.array_equal(nkeys.cpu().numpy(), no) and np.array_equal(kd[live], (ko if compat else ko & 0xFFFFFFFF)[live])
        # a row range of a larger batch: hashed and scanned by one call into its slice of the batch's key table
        Q, P, lo, hi = 9000, 10, 4000, 8500
        q = dev(gen(Q, d, seed=99))
        keys, nkeys = indexer.hash_device(q, hash_times=P, seed=77)
        want = indexer.scan_tensors(q[lo:hi], keys[lo:hi].contiguous(), nkeys[lo:hi].contiguous(), k=k, want_keys=True)
        tab, cnt = torch.zeros_like(keys), torch.zeros_like(nkeys)
        n_multi = indexer._n_multi_rows(Q)
        got = indexer._batch_tensors(q[lo:hi], k, P, 77, want_keys=True, row0=lo, n_multi=min(max(n_multi - lo, 0), hi - lo), out=(tab[lo:hi], cnt[lo:hi]))
        assert torch.equal(tab[lo:hi], keys[lo:hi]) and torch.equal(cnt[lo:hi], nkeys[lo:hi])
        for a, w in zip(got[:4], want):
            assert torch.equal(a, w), (compat, "range")
