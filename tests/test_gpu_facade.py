"""Callers and facade paths either side of the hot path, on the GPU:
N2 minimal trainer (nlsh/trainers/base.py:36-115, triplet.py:16-26,101-131, precompute.py:57-67), the
unknown-callable fallback of `Indexer.query` (nlsh/indexer.py:84-87 accepts ANY distance_func), BatchNorm encoders
(encoders.py:49-50) through `encode_hash`, and the eval.py:103-197 flow (tools/eval_curve.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import assert_lists_differ_only_at_ties, dev, make_hashing
from nlsh_amd import synth
from oracle import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ----------------------------------------------------------------------------- N2 trainer
def test_self_knn_is_exact():
    from nlsh_amd.data import brute_force_topk
    from nlsh_amd.training import self_knn
    x = dev(synth.glove_manifold(5000, 100, seed=21))                 # continuous values: no exact ties
    for metric in ("l2", "cosine"):
        knn = self_knn(x, 10, chunk=1024, metric=metric)
        gt = brute_force_topk(x, x, 11, metric)                       # first hit is the row itself
        assert bool((gt[:, 0] == torch.arange(5000, device=x.device)).all())
        assert torch.equal(knn, gt[:, 1:])
        assert not bool((knn == torch.arange(5000, device=x.device)[:, None]).any())


def test_fit_triplet_learns_and_validates_through_the_hip_indexer():
    from nlsh_amd import training
    from nlsh_amd.data import SIFT, brute_force_topk
    from nlsh_amd.encoders import MultiLayerRelu
    from nlsh_amd.hashings import MultivariateBernoulli
    N, Q, d, H, k = 20000, 500, 128, 12, 10
    corpus, mean, std = synth.standardise(synth.sift_manifold(N, d, seed=31))
    queries, _, _ = synth.standardise(synth.sift_manifold(Q, d, seed=32), mean, std)
    cg, qg = dev(corpus), dev(queries)
    gt = brute_force_topk(qg, cg, k, "l2").cpu().numpy()
    torch.manual_seed(0)
    hashing = MultivariateBernoulli(MultiLayerRelu(d, [256, 256]), H, None)
    knn = training.self_knn(cg, 10)
    validate = training.make_validator(hashing, cg, qg, gt, SIFT.distance, k=k, hash_times=10)
    hashing.train_mode(False)
    before = validate(0)
    # a fixed batch of triplets to read the loss on, before and after
    gen = torch.Generator(device=cg.device)
    gen.manual_seed(99)
    a, p, ng = next(training.triplet_batches(N, knn, 10, 4096, gen))

    def fixed_loss():
        hashing.train_mode(False)
        with torch.no_grad():
            m = hashing._hasher
            return float(training.triplet_loss(m(cg[a]), m(cg[p]), m(cg[ng]), margin=1.0))
    loss0 = fixed_loss()
    logs = []
    history = training.fit_triplet(hashing, cg, knn, n_steps=300, batch_size=1024, margin=1.0, validate=validate,
                                   test_every_updates=100, log=logs.append)
    loss1 = fixed_loss()
    assert [h["step"] for h in history] == [100, 200, 300] and len(logs) == 3
    for h in history:          # names of nlsh/trainers/base.py:87-90,105-108
        assert {"test/n_indexes", "test/std_index_rows", "test/recall", "test/query_size", "test/qps", "loss"} <= set(h)
        assert h["test/qps"] > 0 and 0 <= h["test/recall"] <= 1 and h["test/n_indexes"] >= 1
    assert loss1 < 0.8 * loss0, (loss0, loss1)
    # recall per candidate examined: the learned hash must beat the random-init one (a random hash gets recall by
    # putting a large share of the corpus in few buckets)
    eff0 = before["test/recall"] / max(before["test/query_size"], 1.0)
    eff1 = history[-1]["test/recall"] / max(history[-1]["test/query_size"], 1.0)
    assert eff1 > eff0, (before, history[-1])
    assert history[-1]["test/n_indexes"] > before["test/n_indexes"]
    # the validator went through the HIP Indexer with the CURRENT weights: repacked after every optimiser step
    z, probs, code = hashing.forward_device(qg[:64])
    with torch.no_grad():
        ref = hashing._hasher(qg[:64])
    assert float((probs - ref).abs().max()) < 1e-5
    # export / reload round trip reproduces the keys (the headline checkpoint was made this way: tools/train_hash.py)
    from nlsh_amd import io
    arrays = training.export_weights(hashing)
    again = io.hashing_from_weights([arrays[f"W{i}"] for i in range(3)], [arrays[f"b{i}"] for i in range(3)], compat=True)
    assert torch.equal(again.hash_device(qg, n=1)[0], hashing.hash_device(qg, n=1)[0])


# ----------------------------------------------------------------------------- unknown distance callable
@pytest.mark.parametrize("metric", ["l2", "cosine"])
def test_untagged_distance_callable_takes_the_generic_path(metric):
    """nlsh/indexer.py:84-87 calls whatever `distance_func` it was given; an untagged callable must give the same
    answers as the tagged metric (which runs the fused kernel), including the F7 fallback lists."""
    import torch.nn.functional as F
    from nlsh_amd.data import Glove, SIFT
    from nlsh_amd.indexer import Indexer
    N, Q, d, H, k = 6000, 120, 64, 9, 10
    gen = synth.sift_like if metric == "l2" else synth.glove_like
    corpus, queries = gen(N, d, seed=41), gen(Q, d, seed=42)
    if metric == "l2":
        corpus, mean, std = synth.standardise(corpus)
        queries, _, _ = synth.standardise(queries, mean, std)
    Ws, bs = synth.make_weights([d, 64, H], seed=41)
    hashing = make_hashing(d, (64,), H, Ws, bs)
    cg, qg = dev(corpus), dev(queries)
    plain = (lambda v1, v2: F.pairwise_distance(v1, v2)) if metric == "l2" else (lambda v1, v2: 1 - F.cosine_similarity(v1, v2, dim=-1))
    generic = Indexer(hashing, cg, plain)
    fused = Indexer(hashing, cg, SIFT.distance if metric == "l2" else Glove.distance)
    assert generic.metric is None and fused.metric == metric
    # hash_times=1: both calls see the same (deterministic) hard key
    ids_g, nc_g = generic.query(qg, k=k, hash_times=1)
    ids_f, nc_f = fused.query(qg, k=k, hash_times=1)
    assert nc_g == nc_f
    short = [q for q in range(Q) if nc_f[q] < k]
    for q in range(Q):
        if q in short:
            assert ids_g[q] == ids_f[q]                                # F7: last key's rows (or [] if it has no bucket)
        else:
            assert_lists_differ_only_at_ties(ids_g[q], ids_f[q], queries[q], corpus, metric)
    # multi-probe: same keys injected on both sides; an unknown LAST key yields [] under compat (indexer.py:68,92)
    key_sets = fused.hash(qg, hash_times=5)
    generic.hash = lambda *a, **kw: key_sets
    ids_g5, nc_g5 = generic.query(qg, k=k, hash_times=5)
    res, nc5, _, _ = fused.query_with_keys(qg, [list(s) for s in key_sets], k=k)
    assert nc_g5 == nc5
    for q in range(Q):
        if nc5[q] < k:
            assert ids_g5[q] == res[q]
        else:
            assert_lists_differ_only_at_ties(ids_g5[q], res[q], queries[q], corpus, metric)
    absent = 30000 if 30000 not in fused.index2row else 30001
    sizes = {kk: len(v) for kk, v in fused.index2row.items()}
    small_key = min(sizes, key=sizes.get)
    if sizes[small_key] < k:
        generic.hash = lambda *a, **kw: [[small_key, absent]] * 4
        ids_e, nc_e = generic.query(qg[:4], k=k, hash_times=2)
        assert ids_e == [[]] * 4 and nc_e == [sizes[small_key]] * 4
        res_e, nc_e2, _, _ = fused.query_with_keys(qg[:4], [[small_key, absent]] * 4, k=k)
        assert res_e == [[]] * 4 and nc_e2 == nc_e


# ----------------------------------------------------------------------------- BatchNorm encoder
def test_batchnorm_encoder_through_encode_hash():
    """encoders.py:49-50: Linear -> BatchNorm1d -> ReLU blocks; eval-mode BN is folded into the Linear the kernel packs."""
    from nlsh_amd.encoders import MultiLayerRelu
    from nlsh_amd.hashings import MultivariateBernoulli
    d, H, B = 96, 12, 700
    torch.manual_seed(7)
    enc = MultiLayerRelu(d, [128, 64], with_batchnorm=True)
    hashing = MultivariateBernoulli(enc, H, None)
    with torch.no_grad():
        for m in hashing._hasher.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.5)
                m.running_var.uniform_(0.3, 2.0)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.3)
    hashing.train_mode(False)
    x = synth.glove_like(B, d, seed=7)
    xg = dev(x)
    z, probs, code = hashing.forward_device(xg)
    stack = hashing.linear_stack()
    Ws = [w.cpu().numpy() for w, _ in stack]
    bs = [b.cpu().numpy() for _, b in stack]
    assert len(Ws) == 3 and Ws[0].shape == (128, d) and Ws[1].shape == (64, 128)
    zo = oracle.mlp_forward(x, Ws, bs)
    assert np.array_equal(z.cpu().numpy().view(np.uint32), zo.view(np.uint32))          # same folded weights: bit-exact
    _, p01 = oracle.head_probs(zo)
    ko, _ = oracle.row_keys(p01, 1, "ref_int16")
    assert np.array_equal(hashing.hash_device(xg, n=1)[0].cpu().numpy()[:, 0], ko[:, 0])
    # and the folding itself is right: the torch module (real BatchNorm layers, eval mode) gives the same probabilities
    with torch.no_grad():
        ref = hashing._hasher(xg)
    assert float((probs - ref).abs().max()) < 5e-6
    # train mode goes through autograd (batch statistics), not the kernel; back in eval mode the kernel sees updated buffers
    hashing.train_mode(True)
    out = hashing.predict(xg)
    assert out.requires_grad
    hashing.train_mode(False)
    z2, _, _ = hashing.forward_device(xg)
    with torch.no_grad():
        ref2 = hashing._hasher(xg)
    assert float((torch.sigmoid(z2) - ref2).abs().max()) < 5e-6                          # running stats moved; repacked


# ----------------------------------------------------------------------------- eval.py flow
def test_eval_curve_tool_on_a_small_corpus():
    """tools/eval_curve.py (eval.py:103-197): recall-vs-candidates over n_samples.  Probe sets are nested in n_samples
    (same Philox stream), so candidates and recall are monotone."""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "eval_curve.py"), "--model", "checkpoints/sift1m_manifold_h16.npz",
           "--data", "synth:sift1m", "--n", "20000", "--q", "400", "--max-samples", "8"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    curve = rec["curve"]
    assert [r["n_samples"] for r in curve] == list(range(1, 9)) and rec["metric"] == "l2" and rec["k"] == 10
    cand = [r["avg_n_candidates"] for r in curve]
    recall = [r["recall"] for r in curve]
    assert all(b >= a for a, b in zip(cand, cand[1:])) and cand[-1] > cand[0]
    assert all(b >= a - 1e-12 for a, b in zip(recall, recall[1:])) and recall[-1] > recall[0]
    assert all(r["qps"] > 0 for r in curve)


# ----------------------------------------------------------------------------- pipeline buffer lifetimes
def test_pipeline_survives_dropped_batches_and_weight_updates():
    """A staging loop frees / reallocates its batch tensor right after `submit`; a training step between batches
    repacks the hasher's weights.  Neither may change the answers (nlsh_amd/pipeline.py buffer-lifetime contract)."""
    from nlsh_amd.data import SIFT
    from nlsh_amd.indexer import Indexer
    from nlsh_amd.pipeline import QueryPipeline
    N, Q, d, H, k, P = 60000, 2000, 128, 10, 10, 6
    corpus, mean, std = synth.standardise(synth.sift_like(N, d, seed=51))
    batches = [synth.standardise(synth.sift_like(Q, d, seed=52 + i), mean, std)[0] for i in range(6)]
    Ws, bs = synth.make_weights([d, 64, H], seed=51)
    hashing = make_hashing(d, (64,), H, Ws, bs)
    ix = Indexer(hashing, dev(corpus), SIFT.distance)
    want = [tuple(t.clone() for t in ix.query_tensors(dev(b), k=k, hash_times=P, seed=90 + i)[:3]) for i, b in enumerate(batches)]
    pipe = QueryPipeline(ix, dev(batches[0]), k=k, hash_times=P, depth=3)
    got = []
    for i, b in enumerate(batches):
        q = dev(b)
        out = pipe.submit(q, seed=90 + i)
        del q                                                        # the caller drops its batch at once ...
        junk = [torch.full((Q, d), float("nan"), device="cuda") for _ in range(4)]   # ... and the allocator is asked for more
        pipe.synchronize()
        got.append(tuple(t.clone() for t in out[:3]))
        del junk
    for (d0, i0, n0), (d1, i1, n1) in zip(want, got):
        assert torch.equal(i0, i1) and torch.equal(d0, d1) and torch.equal(n0, n1)
    # weights change between submits (what a training step / load_state does): the next batch is hashed with the new ones
    lin = [m for m in hashing._hasher.modules() if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        lin[-1].weight.mul_(-1.0)
        lin[-1].bias.mul_(-1.0)
    qd = dev(batches[1])
    ref = ix.scan_tensors(qd, *hashing.hash_device(qd, n=P, n_multi_rows=ix._n_multi_rows(Q), seed=7), k=k)
    out = pipe.submit(qd, seed=7)
    pipe.synchronize()
    assert torch.equal(out[1], ref[1]) and torch.equal(out[0], ref[0])
    assert not torch.equal(out[1], want[1][1])                        # flipped bits: other buckets, other answers


def test_pipeline_waits_for_a_batch_still_being_produced_on_the_default_stream():
    """torch produces a tensor on the DEFAULT stream (handle NULL) unless told otherwise, and the pipeline's stage streams are
    non-blocking ones that do not order themselves behind it: `nlsh_query_step_enqueue` must make the front stream wait for the
    producer stream whatever its handle (include/nlsh_hip.h).  The batch buffer is written LAST on the default stream, behind a few
    milliseconds of unrelated work; submitted at once, the answers must be those of the finished batch."""
    from nlsh_amd.data import SIFT
    from nlsh_amd.indexer import Indexer
    from nlsh_amd.pipeline import QueryPipeline
    N, Q, d, H, k, P = 60000, 2000, 128, 10, 10, 6
    corpus, mean, std = synth.standardise(synth.sift_like(N, d, seed=61))
    batch = dev(synth.standardise(synth.sift_like(Q, d, seed=62), mean, std)[0])
    Ws, bs = synth.make_weights([d, 64, H], seed=61)
    ix = Indexer(make_hashing(d, (64,), H, Ws, bs), dev(corpus), SIFT.distance)
    want = tuple(t.clone() for t in ix.query_tensors(batch, k=k, hash_times=P, seed=3)[:3])
    pipe = QueryPipeline(ix, batch, k=k, hash_times=P, depth=3)
    assert torch.cuda.current_stream().cuda_stream == 0              # the case under test: the producer is the NULL handle
    big = torch.randn(4096, 4096, device="cuda")
    for trial in range(3):
        staged = torch.zeros_like(batch)                             # zeros hash to one bucket: a front stage that ran early shows
        torch.cuda.synchronize()
        acc = big
        for _ in range(12):
            acc = acc @ big                                          # milliseconds of work queued in front of the copy
        staged.copy_(batch)
        out = pipe.submit(staged, seed=3)
        pipe.synchronize()
        assert torch.equal(out[1], want[1]) and torch.equal(out[0], want[0]) and torch.equal(out[2], want[2]), trial
        del acc


# ----------------------------------------------------------------------------- query() in row ranges (host/device overlap)
@pytest.mark.parametrize("compat", [True, False])
def test_query_in_row_ranges_equals_the_single_range_call(compat):
    """`Indexer.query` scans a large batch in `query_chunks` row ranges and converts one range while the device scans the
    next; the lists and counts must not depend on the split (including the F7 rule of the short queries and a task-table
    overflow inside a range)."""
    from nlsh_amd.data import SIFT
    from nlsh_amd.indexer import Indexer
    N, Q, d, H, k, P = 30000, 4500, 128, 12, 10, 10
    corpus, mean, std = synth.standardise(synth.sift_like(N, d, seed=91))
    queries = synth.standardise(synth.sift_like(Q, d, seed=92), mean, std)[0]
    Ws, bs = synth.make_weights([d, 64, H], seed=93)
    cg, qg = dev(corpus), dev(queries)
    answers = []
    for chunks in (1, 2, 3):
        ix = Indexer(make_hashing(d, [64], H, Ws, bs, compat=compat, seed=5), cg, SIFT.distance, compat=compat)
        ix.query_chunks, ix._CHUNK_MIN_ROWS = chunks, 1024
        if chunks == 3:
            ix._estimate_tasks = lambda *a, **kw: 64          # every range overflows its first task table and is repeated
        got = [ix.query(qg, k=k, hash_times=P) for _ in range(2)]       # two calls: the call counter seeds the probes
        answers.append(got)
        if chunks == 3:
            assert min(ix._max_tasks.values()) > 64
    short = sum(1 for n in answers[0][0][1] if n < k)
    assert short > 0, "the case should hold queries with fewer than k candidates (F7 rule)"
    for got in answers[1:]:
        for call in range(2):
            assert got[call][1] == answers[0][call][1]
            assert got[call][0] == answers[0][call][0]
    assert answers[0][0] != answers[0][1]                      # different probe draws per call: the comparison is not vacuous


def test_hash_in_train_mode_uses_batch_statistics_like_the_reference():
    """nlsh/hashings.py:66-67: `hash` runs `self._hasher(x)` in the module's CURRENT mode (nlsh/trainers/proposed.py:101-104
    calls it in train mode).  With BatchNorm that means batch statistics + a running-statistics update; without BatchNorm
    both modes are the same function and the fused kernel serves both."""
    from nlsh_amd.encoders import MultiLayerRelu
    from nlsh_amd.hashings import MultivariateBernoulli
    torch.manual_seed(3)
    d, H, B = 64, 12, 300
    x = torch.randn(B, d, device="cuda") * 2 + 0.5
    hashing = MultivariateBernoulli(MultiLayerRelu(d, [48, 32], with_batchnorm=True), H, None)
    with torch.no_grad():
        for m in hashing._hasher.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(); m.running_var.uniform_(0.5, 2.0); m.weight.uniform_(0.5, 1.5); m.bias.normal_()
    hashing.train_mode(False)
    eval_sets = hashing.hash(x, 1)
    hashing.train_mode(True)
    bn = [m for m in hashing._hasher.modules() if isinstance(m, torch.nn.BatchNorm1d)][0]
    before = bn.running_mean.clone()
    train_sets = hashing.hash(x, 1)
    assert not torch.equal(bn.running_mean, before)                  # the train-mode forward updated the running statistics
    with torch.no_grad():
        probs = hashing._hasher(x)                                   # the same train-mode forward (batch statistics)
    bits = (probs > 0.5).int().cpu().numpy()
    want = oracle.pack_keys(bits[:, None, :], "ref_int16")[:, 0]
    got = np.array([next(iter(s)) for s in train_sets])
    assert all(len(s) == 1 for s in train_sets) and np.array_equal(got, want)
    assert sum(a != b for a, b in zip(train_sets, eval_sets)) > B // 10     # and that is NOT the eval-mode (folded) function
    # multi-probe in train mode: slot 0 is the hard key, rows past n_multi_rows stay single-probe, keys distinct per row
    keys, nkeys = hashing.hash_device(x, n=6, n_multi_rows=256)
    kh, nh = keys.cpu().numpy(), nkeys.cpu().numpy()
    assert np.array_equal(kh[:, 0], want) and np.all(nh[256:] == 1) and nh[:256].max() > 1
    assert all(len(set(kh[r, :nh[r]].tolist())) == nh[r] for r in range(B))
    # an encoder without BatchNorm takes the fused kernel in either mode: identical keys
    plain = MultivariateBernoulli(MultiLayerRelu(d, [48, 32]), H, None)
    plain.train_mode(True)
    a = plain.hash_device(x, n=1)[0].clone()
    plain.train_mode(False)
    assert torch.equal(a, plain.hash_device(x, n=1)[0])


def test_indexer_hashes_a_batchnorm_encoder_in_train_mode_batch_by_batch_like_the_reference():
    """nlsh/indexer.py:40-53: `Indexer.hash` feeds the module `batch_size`-row batches -- in TRAIN mode with BatchNorm that means
    per-batch statistics, one running-statistics update per batch, and the trailing partial batch on its own statistics with n = 1
    (one forward over all rows used other statistics).  The index build goes the same way (indexer.py:36-38)."""
    import copy
    from nlsh_amd.data import SIFT
    from nlsh_amd.encoders import MultiLayerRelu
    from nlsh_amd.hashings import MultivariateBernoulli
    from nlsh_amd.indexer import Indexer
    torch.manual_seed(5)
    d, H, N, bsz = 32, 10, 700, 256
    x = torch.randn(N, d, device="cuda") * 1.5 + 0.3
    hashing = MultivariateBernoulli(MultiLayerRelu(d, [24], with_batchnorm=True), H, None)
    hashing.train_mode(False)
    ix = Indexer(hashing, x, SIFT.distance)                          # built in eval mode: the fused kernel
    hashing.train_mode(True)
    twin = copy.deepcopy(hashing._hasher)                            # same weights, same running statistics, same mode
    keys, nkeys = ix.hash_device(x, batch_size=bsz, hash_times=1)
    want = []
    with torch.no_grad():
        for lo in range(0, N, bsz):                                  # the reference's loop: 256 + 256 + 188 rows
            want.append((twin(x[lo:lo + bsz]) > 0.5).int())
    want = oracle.pack_keys(torch.cat(want).cpu().numpy()[:, None, :], "ref_int16")[:, 0]
    assert np.array_equal(keys[:, 0].cpu().numpy(), want) and bool((nkeys == 1).all())
    bn, bn_twin = ([m for m in mod.modules() if isinstance(m, torch.nn.BatchNorm1d)][0] for mod in (hashing._hasher, twin))
    assert torch.allclose(bn.running_mean, bn_twin.running_mean) and int(bn.num_batches_tracked) == int(bn_twin.num_batches_tracked) == 3
    # multi-probe: full batches get hash_times keys, the trailing partial batch stays single-probe (F6)
    keys5, nkeys5 = ix.hash_device(x, batch_size=bsz, hash_times=5)
    assert keys5.shape == (N, 5) and int(nkeys5[:512].max()) > 1 and bool((nkeys5[512:] == 1).all())
    # the pipeline runs the fused eval-mode kernel only: it refuses a train-mode BatchNorm hasher instead of hashing with other statistics
    from nlsh_amd import _capi
    from nlsh_amd.pipeline import QueryPipeline
    hashing.train_mode(False)
    pipe = QueryPipeline(ix, x[:128].contiguous(), k=5, hash_times=2)
    hashing.train_mode(True)
    with pytest.raises(_capi.NlshHipError):
        pipe.submit(x[:128].contiguous())
    hashing.train_mode(False)


def test_bench_runs_on_real_files_given_as_a_texmex_directory(tmp_path):
    """the day real SIFT1M / GloVe files are supplied, `bench.py --dataset PATH --hash-checkpoint CKPT` reads them through
    the facade's dataset classes (nlsh/data.py:14-46,112-140) and scores recall against the file's own `neighbors`.  Here: a TEXMEX
    directory written from seeded data with exact ground truth, a checkpoint in the reference's parameter naming."""
    import json
    import subprocess
    import sys
    from nlsh_amd import io
    from nlsh_amd.data import brute_force_topk
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(11)
    N, Q, d, H = 20000, 700, 32, 10
    base = rng.standard_normal((N, d)).astype(np.float32)
    query = (base[rng.integers(0, N, Q)] + 0.05 * rng.standard_normal((Q, d))).astype(np.float32)
    gt = brute_force_topk(dev(query), dev(base), 100, "l2").cpu().numpy().astype(np.int32)
    root = tmp_path / "toy"
    root.mkdir()
    io.write_vecs(str(root / "toy_base.fvecs"), base)
    io.write_vecs(str(root / "toy_query.fvecs"), query)
    io.write_vecs(str(root / "toy_groundtruth.ivecs"), gt)
    Ws, bs = synth.make_weights([d, 48, H], seed=4)
    np.savez(tmp_path / "hash.npz", **{f"W{i}": w for i, w in enumerate(Ws)}, **{f"b{i}": b for i, b in enumerate(bs)})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dataset", str(root), "--hash-checkpoint", str(tmp_path / "hash.npz"),
                          "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["data"] == "real" and "toy" in rec["config"]["workload"] and "hash.npz" in rec["config"]["hash"]
    assert rec["config"]["shape"]["N"] == N and rec["config"]["shape"]["d"] == d and rec["config"]["shape"]["H"] == H and rec["config"]["shape"]["Q"] == Q
    # the same recall from the facade directly, against the file's neighbours
    from nlsh_amd.data import SIFT
    from nlsh_amd.indexer import Indexer
    from nlsh_amd.metrics import calculate_recall
    ix = Indexer(make_hashing(d, (48,), H, Ws, bs), dev(base), SIFT.distance)
    ids, _ = ix.query(dev(query), k=10, hash_times=10, seed=5000)
    assert abs(rec["recall_at_10"] - float(np.mean(calculate_recall(list(gt[:, :10]), ids)))) < 1e-12


def test_train_hash_tool_on_a_texmex_directory(tmp_path):
    """tools/train_hash.py --dataset: the minimal triplet trainer (N2) on files read through the dataset classes (N4); the checkpoint it
    writes loads back through io.load_hasher_weights and its validation history went through the HIP Indexer."""
    import json
    import subprocess
    import sys
    from nlsh_amd import io
    from nlsh_amd.data import brute_force_topk
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    N, Q, d = 6000, 200, 24
    base = synth.sift_like(N, d, seed=31)
    query = synth.sift_like(Q, d, seed=32)
    gt = brute_force_topk(dev(query), dev(base), 10, "l2").cpu().numpy().astype(np.int32)
    root = tmp_path / "toy"
    root.mkdir()
    io.write_vecs(str(root / "toy_base.fvecs"), base)
    io.write_vecs(str(root / "toy_query.fvecs"), query)
    io.write_vecs(str(root / "toy_groundtruth.ivecs"), gt)
    out_path = tmp_path / "h.npz"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_hash.py"), "--dataset", str(root), "--metric", "l2", "--unit-norm",
                          "--hash-size", "8", "--steps", "120", "--every", "60", "--q", str(Q), "--out", str(out_path)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    last = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert 0.0 <= last["test/recall"] <= 1.0 and last["test/query_size"] > 0          # the reference's metric names (base.py:105-108)
    Ws, bs = io.load_hasher_weights(str(out_path))
    assert [w.shape for w in Ws] == [(256, d), (256, 256), (8, 256)] and all(b is not None for b in bs)


def test_query_keeps_its_range_plans_and_follows_weight_updates_and_table_growth():
    """r06: `Indexer.query` builds the buffers and the `nlsh_query_batch` descriptor of each row range once per batch shape and reuses
    them call after call.  The kept descriptor must follow what can change underneath it: the hasher's weights (a training step between
    two validation calls, nlsh/trainers/base.py:80-96), the task table (grown after an overflow, trimmed after the first call) and the
    batch shape -- every call equals the lists derived from a `query_tensors` call on the same seed, which builds everything afresh."""
    from nlsh_amd.data import SIFT
    from nlsh_amd.indexer import Indexer
    N, d, H, k, P = 30000, 128, 12, 10, 10
    corpus, mean, std = synth.standardise(synth.sift_like(N, d, seed=191))
    Ws, bs = synth.make_weights([d, 64, H], seed=193)
    hashing = make_hashing(d, [64], H, Ws, bs, compat=False, seed=5)
    ix = Indexer(hashing, dev(corpus), SIFT.distance, compat=False)
    ix.query_chunks, ix._CHUNK_MIN_ROWS = 2, 1024

    def fresh(q, seed):
        _, idx, nc, _ = ix.query_tensors(q, k=k, hash_times=P, seed=seed)
        return [[int(v) for v in row if v >= 0] for row in idx.cpu().numpy()], nc.cpu().tolist()

    qa = dev(synth.standardise(synth.sift_like(4500, d, seed=192), mean, std)[0])
    qb = dev(synth.standardise(synth.sift_like(3000, d, seed=194), mean, std)[0])
    for step, q in enumerate((qa, qa, qb, qa)):
        want = fresh(q, 700 + step)
        assert ix.query(q, k=k, hash_times=P, seed=700 + step) == tuple(want) or list(ix.query(q, k=k, hash_times=P, seed=700 + step)) == list(want)
    plans_before = len(ix._range_plans)
    assert plans_before >= 3                                  # two ranges of the 4500-row shape, two of the 3000-row one; none rebuilt per call
    with torch.no_grad():                                     # "a training step": new weights, same tensors
        for prm in hashing.parameters():
            prm.mul_(-0.5)
    want = fresh(qa, 900)
    got = ix.query(qa, k=k, hash_times=P, seed=900)
    assert list(got) == list(want) and len(ix._range_plans) == plans_before
    ix._max_tasks = {key: 64 for key in ix._max_tasks}        # every kept descriptor now names a table that is too small
    got = ix.query(qa, k=k, hash_times=P, seed=901)
    assert list(got) == list(fresh(qa, 901)) and min(ix._max_tasks[key] for key in ix._max_tasks if key[1] in (2250,)) > 64
