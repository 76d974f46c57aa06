"""CPU-only tests: C-ABI library loads and exports every declared symbol (no compute calls),
host logic of the facade, and hygiene rules (the product never touches oracle/)."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "neural-locality-sensitive-hashing_amd")


def test_capi_library_exports_every_declared_symbol():
    from nlsh_amd import _capi
    header = open(os.path.join(ROOT, "include", "nlsh_hip.h")).read()
    declared = set(re.findall(r"\b(nlsh_[a-z0-9_]+)\s*\(", header))
    declared -= {"nlsh_stream_t"}
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    if not os.path.exists(_capi.LIB_PATH):
        _capi.build_library()
    lib = _capi.lib()
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} not exported"
    assert lib.nlsh_abi_version() == 4
    # pure host-side argument validation (no device needed): errors come back as codes + message
    dims = _capi.int_array([128, 256, 256, 16])
    # the 32x32x2 fragments of every layer + biases, then (r04) the hidden layers once more packed for 16x16x4 tiles (the 16-row form)
    assert lib.nlsh_encoder_packed_floats(3, dims) == (256 * 128 + 256 + 256 * 256 + 256 + 32 * 256 + 32) + (256 * 128 + 256 * 256)
    assert lib.nlsh_encoder_packed_floats(3, _capi.int_array([128, 256, 256, 33])) == -1
    assert b"hash_size" in lib.nlsh_last_error()
    assert lib.nlsh_encoder_packed_floats(2, _capi.int_array([128, 700, 16])) == -1       # 128 ok, 700 too wide
    assert lib.nlsh_scan_workspace(10, 4, 10, 100, 50, 128) > 0


def test_step_descriptor_layout_and_host_side_validation():
    """ABI v3: `_capi.StepDesc` mirrors `nlsh_step_desc_t` field for field (names and order checked against the header's text, the
    size against the library, which refuses a descriptor of another size), and the host-side argument checks of the step calls
    answer with codes + messages -- no device call is made for any of these."""
    import ctypes
    from nlsh_amd import _capi
    header = open(os.path.join(ROOT, "include", "nlsh_hip.h")).read()
    body = header[header.index("typedef struct nlsh_step_desc {"):header.index("} nlsh_step_desc_t;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if decl:
            names += [re.sub(r"[\s*]", "", part).split(" ")[-1] for part in re.sub(r"^(const\s+)?[\w]+\s", "", decl, count=1).split(",")]
    assert names == [f[0] for f in _capi.StepDesc._fields_], names
    lib = _capi.lib()
    desc, handle = _capi.StepDesc(), ctypes.c_void_p()
    assert lib.nlsh_step_create(ctypes.byref(desc), ctypes.sizeof(desc) - 8, ctypes.byref(handle)) == _capi.E_INVALID
    assert str(ctypes.sizeof(desc)).encode() in lib.nlsh_last_error()        # "... this library's is <sizeof>": the two layouts agree
    assert lib.nlsh_step_create(ctypes.byref(desc), ctypes.sizeof(desc), ctypes.byref(handle)) == _capi.E_INVALID
    assert b"streams" in lib.nlsh_last_error() and not handle.value
    assert lib.nlsh_query_step_enqueue(None, None, 0, 0, None, None, None) == _capi.E_INVALID
    assert lib.nlsh_step_busy(None) == _capi.E_INVALID and lib.nlsh_step_release(None) == _capi.E_INVALID
    assert lib.nlsh_step_destroy(None) == _capi.OK
    # a workspace that does not start on a 16-byte boundary is refused before anything is launched (fake non-null pointers: none is read)
    fake = 0x10000
    need = lib.nlsh_bucket_order_workspace(1000)
    assert lib.nlsh_bucket_order(fake, 1000, fake, fake + 4, need, None) == _capi.E_INVALID and b"16-byte" in lib.nlsh_last_error()


def test_weights_signature_follows_the_module_as_it_is_now(monkeypatch):
    """the cached walk behind `_weights_signature` (called per hashing call: `QueryPipeline.submit` re-binds the packed
    weights when it changes) must see an in-place update, a parameter or buffer registered later, a swapped layer and an added submodule."""
    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)      # no GPU here; the signature is host logic
    from nlsh_amd.encoders import MultiLayerRelu
    from nlsh_amd.hashings import MultivariateBernoulli
    h = MultivariateBernoulli(MultiLayerRelu(8, [16, 16]), 4, None)
    s0 = h._weights_signature()
    assert s0 == h._weights_signature()
    with torch.no_grad():
        h._hasher.output_layer.weight.add_(1)
    s1 = h._weights_signature()
    assert s1 != s0
    h._hasher.register_buffer("extra", torch.zeros(3))
    s2 = h._weights_signature()
    assert s2 != s1 and len(s2) == len(s1) + 1
    enc = h._hasher._encoder
    setattr(enc, list(enc._modules)[0], torch.nn.Linear(8, 16))
    s3 = h._weights_signature()
    assert s3 != s2 and len(s3) == len(s2)
    enc.add_module("late", torch.nn.Linear(2, 2))
    assert len(h._weights_signature()) == len(s3) + 2


def test_product_never_references_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(PKG):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", text, re.M) or "nlsh_oracle" in text or "liboracle" in text:
                    bad.append(os.path.join(dirpath, fn))
    assert not bad, bad


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from nlsh_amd import _capi
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_capi.NlshHipError):
        _capi.lib()


def test_build_index_host_variant_reference_vector():
    # the reference's own unit test (nlsh/tests/test_indexer.py:6-26), cuda=False
    from nlsh_amd.indexer import build_index
    got = build_index([set([1, 2]), set([2, 3, 4]), set([1, 5])], cuda=False)
    expected = {1: [0, 2], 2: [0, 1], 3: [1], 4: [1], 5: [2]}
    assert got.keys() == expected.keys()
    for k, v in expected.items():
        assert torch.equal(got[k], torch.LongTensor(v))
    assert build_index([], cuda=False) == {}


def test_calculate_recall_matches_oracle():
    from nlsh_amd.metrics import calculate_recall
    from oracle import oracle
    rng = np.random.default_rng(0)
    yt = rng.integers(0, 50, size=(20, 10))
    yp = [rng.integers(0, 50, size=rng.integers(0, 12)).tolist() for _ in range(20)]
    assert calculate_recall(list(yt), yp) == oracle.calculate_recall(list(yt), yp)
    assert calculate_recall(list(yt), yp, np.mean) == pytest.approx(np.mean(oracle.calculate_recall(list(yt), yp)))
    with pytest.raises(AssertionError):
        calculate_recall([[1]], [])


def test_distance_callables_match_oracle_and_carry_metric_tag():
    from nlsh_amd.data import Glove, SIFT, metric_of
    from oracle import oracle
    rng = np.random.default_rng(1)
    c = rng.standard_normal((50, 100)).astype(np.float32)
    q = rng.standard_normal(100).astype(np.float32)
    rows = np.arange(50, dtype=np.int32)
    assert metric_of(SIFT.distance) == "l2" and metric_of(Glove.distance) == "cosine" and metric_of(len) is None
    l2 = SIFT.distance(torch.from_numpy(q), torch.from_numpy(c)).numpy()
    cs = Glove.distance(torch.from_numpy(q), torch.from_numpy(c)).numpy()
    assert np.allclose(l2, oracle.distances(q, c, rows, "l2"), rtol=1e-5, atol=1e-5)
    assert np.allclose(cs, oracle.distances(q, c, rows, "cosine"), rtol=1e-5, atol=1e-6)


def test_encoder_parameter_names_and_batchnorm_folding():
    from nlsh_amd.encoders import MultiLayerRelu, TwoLayer256Relu
    two = TwoLayer256Relu(100)
    assert set(dict(two.named_parameters())) == {"fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"}
    m = MultiLayerRelu(64, [32, 16], with_batchnorm=True)
    names = set(dict(m.named_parameters()))
    assert {"0_linear.weight", "0_linear.bias", "0_batch_norm.weight", "1_linear.weight", "1_batch_norm.bias"} <= names
    m.eval()
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.uniform_(-1, 1); mod.running_var.uniform_(0.5, 2); mod.weight.uniform_(0.5, 2); mod.bias.uniform_(-1, 1)
        x = torch.randn(7, 64)
        ref = m(x)
        h = x
        for w, b in m.linear_stack():
            h = torch.relu(h @ w.T + b)
    assert torch.allclose(h, ref, atol=1e-5)
    assert m.output_dim == 16 and two.output_dim == 256


def test_synthetic_generators_are_deterministic():
    from nlsh_amd import synth
    a, b = synth.sift_like(100, 128, seed=5), synth.sift_like(100, 128, seed=5)
    assert np.array_equal(a, b) and a.min() >= 0 and a.max() <= 218 and np.array_equal(a, np.rint(a))
    assert np.allclose(np.linalg.norm(synth.deep_like(10, 96), axis=1), 1.0, atol=1e-5)
    Ws, bs = synth.make_weights([128, 256, 16], seed=0)
    assert Ws[0].shape == (256, 128) and bs[1].shape == (16,)


def test_reference_import_paths_resolve_to_the_hip_package():
    """compat/ mirrors the reference's module paths, so `from nlsh.indexer import Indexer` needs no edit."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from nlsh.indexer import Indexer, build_index\n"
        "from nlsh.hashings import MultivariateBernoulli\n"
        "from nlsh.utils import hash_codes\n"
        "from nlsh.metrics import calculate_recall\n"
        "from nlsh.data import SIFT, Glove\n"
        "from encoders import MultiLayerRelu, TwoLayer256Relu\n"
        "import nlsh_amd.indexer, nlsh_amd.hashings, nlsh_amd.encoders\n"
        "assert Indexer is nlsh_amd.indexer.Indexer and MultivariateBernoulli is nlsh_amd.hashings.MultivariateBernoulli\n"
        "assert MultiLayerRelu is nlsh_amd.encoders.MultiLayerRelu\n"
        "assert build_index([{3}, {5}, {3}], cuda=False)[3].tolist() == [0, 2]\n"
        "print('ok')\n") % os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "compat")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-2000:]


def test_bench_gpus_n_launches_n_ranks_or_fails(tmp_path):
    """`python bench.py --gpus 2` (no launcher, no WORLD_SIZE) must start 2 ranks itself; a world size that disagrees
    with --gpus is an error, never a 1-GPU run under an N-GPU label.  --rehearse-launch stops after the rendezvous
    (gloo), so this runs without a GPU; the full 2-rank run is tests/test_gpu_distributed.py."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    bench = os.path.join(ROOT, "bench.py")
    out = subprocess.run([sys.executable, bench, "--gpus", "2", "--rehearse-launch"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2
    bad = subprocess.run([sys.executable, bench, "--gpus", "2", "--rehearse-launch"], env=dict(env, WORLD_SIZE="3", RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "disagrees" in bad.stderr
    one = subprocess.run([sys.executable, bench, "--gpus", "1", "--rehearse-launch"], env=env, capture_output=True, text=True, timeout=120)
    assert one.returncode == 0 and json.loads(one.stdout.splitlines()[-1])["n_gpus"] == 1


def test_result_lists_skip_the_young_generation_and_full_collections_still_happen():
    """`Indexer._plain_lists` with the OPT-IN promotion on: the fresh lists are promoted past the collector's young
    generations (no collection is triggered by them), the application's own frozen objects are respected, and cyclic
    garbage is still reclaimed by the periodic full collection.  By default (flag off) the collector's generations are
    left alone: only paused for the conversion itself."""
    import gc
    import weakref
    import numpy as np
    from nlsh_amd.indexer import Indexer
    idx, nc = np.arange(20000, dtype=np.int32).reshape(2000, 10), np.full((2000,), 12, dtype=np.int32)
    seen = []
    cb = lambda phase, info: seen.append(info["generation"]) if phase == "start" else None   # noqa: E731
    assert Indexer.promote_results is False          # a library call does not rewrite the application's GC state unasked
    gc.collect()
    frozen0 = gc.get_freeze_count()
    gc.callbacks.append(cb)
    Indexer.promote_results = True
    try:
        lists, counts = Indexer._plain_lists(idx, nc)
        probe = [[i] for i in range(50)]          # container allocations right after: would trip a young collection
        after, young = list(seen), gc.get_count()[0]
        assert after == [0] and young < 700 and probe    # only the explicit young pass BEFORE the conversion; none after
        assert lists == idx.tolist() and counts == nc.tolist()
        Indexer.promote_results = False
        try:
            del seen[:]
            lists2, counts2 = Indexer._plain_lists(idx, nc)     # (the first result stays alive: a freed one hands its count back)
            assert gc.isenabled() and gc.get_freeze_count() == frozen0   # default: collector back on, nothing frozen or promoted
            probe = [[i] for i in range(50)]
            after = list(seen)
            assert after and after[0] == 0 and lists2 == lists  # the young collection the promotion avoids happens here
        finally:
            Indexer.promote_results = True
        # an application that froze its own objects keeps them frozen
        gc.freeze()
        frozen = gc.get_freeze_count()
        Indexer._plain_lists(idx, nc)
        assert gc.get_freeze_count() >= frozen > 0
        gc.unfreeze()
        # cyclic garbage made between calls is reclaimed by the periodic full collection
        class Node:
            pass
        a, b = Node(), Node()
        a.other, b.other = b, a
        ref = weakref.ref(a)
        del a, b
        Indexer._plain_lists(idx, nc)                 # young cyclic garbage goes at the next call (generation-0 pass before it)
        assert ref() is None
        # ... and garbage that was promoted before it died is reclaimed by the periodic full collection
        a, b = Node(), Node()
        a.other, b.other = b, a
        ref = weakref.ref(a)
        Indexer._plain_lists(idx, nc)                 # a and b are alive here: promoted to the oldest generation
        del a, b
        Indexer._plain_lists(idx, nc)
        assert ref() is not None
        old_every, Indexer._FULL_COLLECT_EVERY = Indexer._FULL_COLLECT_EVERY, 8
        try:
            for _ in range(8):
                Indexer._plain_lists(idx, nc)
        finally:
            Indexer._FULL_COLLECT_EVERY = old_every
        assert ref() is None
    finally:
        Indexer.promote_results = False
        gc.callbacks.remove(cb)


def test_deferred_result_release_is_opt_in_and_drops_the_older_result_under_the_next_call():
    """`Indexer.defer_result_release`: off by default (a library call does not change object lifetimes unasked); when on, the
    indexer holds its last two results and `_release_held()` -- called once the next call's device work is queued -- frees the
    older one, i.e. the lists the caller stopped referencing when it rebound its names after the previous call."""
    import weakref
    from nlsh_amd.indexer import Indexer

    class Lists(list):      # plain lists cannot be weak-referenced
        pass

    ix = Indexer.__new__(Indexer)
    assert Indexer.defer_result_release is False
    a = Lists([[1, 2]])
    ra = weakref.ref(a)
    assert ix._keep((a, [2]))[0] is a and not ix.__dict__.get("_held")
    del a
    assert ra() is None                                    # default: nothing is retained
    Indexer.defer_result_release = True
    try:
        a, b = Lists([[1]]), Lists([[2]])
        ra, rb = weakref.ref(a), weakref.ref(b)
        ix._keep((a, [1]))                                 # call i returns; the caller holds `a`
        ix._release_held()                                 # call i+1 queues its work: only the newest is kept, `a` is the newest
        ix._keep((b, [1]))                                 # call i+1 returns; the caller rebinds: drops `a`, holds `b`
        del a
        assert ra() is not None                            # not freed in the caller's rebind ...
        ix._release_held()                                 # ... but under call i+2's device work
        assert ra() is None and rb() is not None
        del b
        assert rb() is not None
        Indexer.defer_result_release = False
        ix._keep(([], []))                                 # switching it off releases what was held
        assert rb() is None
    finally:
        Indexer.defer_result_release = False


def test_query_pauses_the_collector_for_the_call_and_leaves_it_as_it_found_it(monkeypatch):
    """r05: `Indexer.query` runs with the cyclic collector paused from entry to exit (not only inside each row range's conversion), so
    that the fresh result lists are walked once, after the call, whatever the row-range split.  The application's collector state is
    restored on every path: enabled stays enabled (also when the call raises), disabled stays disabled, and the switch turns it off."""
    import gc
    from nlsh_amd.indexer import Indexer
    ix = Indexer.__new__(Indexer)                      # host logic only: no index, no device
    seen = []

    def fake_query(self, q, k, hash_times, seed):
        seen.append(gc.isenabled())
        if q == "boom":
            raise RuntimeError("boom")
        return [[1]], [1]
    monkeypatch.setattr(Indexer, "_query", fake_query)
    assert gc.isenabled()
    assert ix.query("x") == ([[1]], [1]) and seen == [False] and gc.isenabled()
    with pytest.raises(RuntimeError):
        ix.query("boom")
    assert gc.isenabled()
    gc.disable()
    try:
        ix.query("x")
        assert not gc.isenabled() and seen[-1] is False
    finally:
        gc.enable()
    monkeypatch.setattr(Indexer, "pause_collector_for_call", False)
    ix.query("x")
    assert seen[-1] is True and gc.isenabled()


def test_collector_pause_is_reentrant_and_shared_between_threads(monkeypatch):
    """r06: two threads inside `query()` at once, and the pause `_plain_lists` nests inside it, leave the collector
    as the FIRST region found it -- the thread that finishes first must not switch the collector back on under the other one, a caller
    that had it disabled gets it back disabled, and one that enables it during a call keeps it enabled."""
    import gc
    import threading
    from nlsh_amd import indexer
    from nlsh_amd.indexer import Indexer
    pause = indexer._collector_pause
    ix = Indexer.__new__(Indexer)
    a_inside, a_may_leave, b_done = threading.Event(), threading.Event(), threading.Event()
    seen = {}

    def fake_query(self, q, k, hash_times, seed):
        if q == "a":                                    # thread A: enters first, leaves last
            a_inside.set()
            assert a_may_leave.wait(10)
            seen["a_after_b_left"] = gc.isenabled()
        else:                                           # thread B: enters while A is inside, leaves before it
            seen["b_inside"] = gc.isenabled()
            seen["b_app_state"] = pause.application_state()
            with pause:                                 # what `_plain_lists` does inside a paused call
                pass
            seen["b_after_nested"] = gc.isenabled()
        return [[1]], [1]
    monkeypatch.setattr(Indexer, "_query", fake_query)
    assert gc.isenabled() and pause._depth == 0
    ta = threading.Thread(target=lambda: ix.query("a"))
    ta.start()
    assert a_inside.wait(10)

    def run_b():
        ix.query("b")
        seen["after_b_returned"] = gc.isenabled()       # A is still inside: the collector must still be paused
        b_done.set()
    tb = threading.Thread(target=run_b)
    tb.start()
    assert b_done.wait(10)
    a_may_leave.set()
    ta.join(10)
    tb.join(10)
    assert seen == {"b_inside": False, "b_app_state": True, "b_after_nested": False, "after_b_returned": False, "a_after_b_left": False}
    assert gc.isenabled() and pause._depth == 0         # the last region to close restored what the first one found
    # an application that runs with the collector off gets it back off, whatever happened in between
    gc.disable()
    try:
        with pause:
            with pause:
                assert pause.application_state() is False
        assert not gc.isenabled()
    finally:
        gc.enable()
    # a caller that switches the collector ON during a call keeps it on; the exit never disables
    with pause:
        gc.enable()
    assert gc.isenabled()


def test_promote_results_still_promotes_through_a_paused_query(monkeypatch):
    """with `pause_collector_for_call` on (the default) `gc.isenabled()` is False inside `query()`, and
    `_plain_lists` took that for "the application runs without a collector" -- `Indexer.promote_results = True` was a silent no-op on
    the `query()` path.  It now asks the pause for the state the APPLICATION left: the promotion (freeze + unfreeze: the fresh lists
    are in the oldest generation, the young counters at zero) happens through `query()` too, and not when the caller runs gc-off."""
    import gc
    import numpy as np
    from nlsh_amd.indexer import Indexer
    ix = Indexer.__new__(Indexer)
    idx, nc = np.arange(10240, dtype=np.int32).reshape(1024, 10), np.full((1024,), 12, dtype=np.int32)
    monkeypatch.setattr(Indexer, "_query", lambda self, q, k, hash_times, seed: Indexer._plain_lists(idx, nc))
    monkeypatch.setattr(Indexer, "promote_results", True)
    gc.collect()
    before = Indexer._promotions
    lists, counts = ix.query("x")
    assert Indexer._promotions == before + 1 and lists == idx.tolist() and gc.isenabled()
    assert gc.get_count()[0] < 64                       # the young generation does not hold the 1024 fresh lists any more
    gc.disable()
    try:
        ix.query("x")
        assert Indexer._promotions == before + 1 and not gc.isenabled()   # the application's own choice: nothing is promoted
    finally:
        gc.enable()
    monkeypatch.setattr(Indexer, "pause_collector_for_call", False)
    ix.query("x")
    assert Indexer._promotions == before + 2


def test_step_create_refuses_equal_streams_and_bad_scan_arguments_before_anything_is_enqueued():
    """ADVICE r05 / VERDICT r05 item 5: `nlsh_step_create` used to check that front, mid and tail were non-NULL, not that they differ, and
    the scan call's own argument checks (workspace alignment, strides) only ran inside the first enqueue, behind its encode.  Both are
    refused by the constructor now -- on the host, before any HIP call, so this runs without a GPU."""
    import ctypes
    from nlsh_amd import _capi
    L = _capi.lib()
    dims = _capi.int_array([128, 64, 16])
    buf = (ctypes.c_char * 4096)()
    a = ctypes.addressof(buf)
    def desc(**kw):
        base = dict(n_layers=2, act=0, key_mode=0, n_probes=10, dims=ctypes.cast(dims, ctypes.c_void_p), packed=a, n_multi_rows=0,
                    corpus_sorted=a, row_stride=128, gid=a, uniq_keys=a, offsets=a, bucket_order=a, cell_of=None, cell_offsets=None, inv_norm=None,
                    d=128, n_buckets=8, n_cells=0, k=10, metric=0, algo=2, seg_rows=0, hold_done=0, Q=64, qkeys=a, nkeys=a, out_dist=a, out_idx=a,
                    out_keys=None, out_ncand=a, status=a, workspace=a, workspace_bytes=4096, max_tasks=16, front=0x1000, plan=None, mid=0x2000, tail=0x3000)
        base.update(kw)
        return _capi.StepDesc(**base)
    handle = ctypes.c_void_p()
    for bad in (dict(mid=0x1000), dict(tail=0x2000), dict(tail=0x1000), dict(plan=0x3000), dict(front=None)):
        d_ = desc(**bad)
        rc = L.nlsh_step_create(ctypes.byref(d_), ctypes.sizeof(d_), ctypes.byref(handle))
        assert rc == _capi.E_INVALID and not handle.value, bad
        assert b"stream" in L.nlsh_last_error()
    # distinct streams, but a scan argument the scan call itself refuses: a workspace that is not 16-byte aligned / too small
    for bad, word in ((dict(workspace=a + 4), b"aligned"), (dict(workspace_bytes=64), b"workspace"), (dict(row_stride=126), b"row_stride")):
        d_ = desc(**bad)
        rc = L.nlsh_step_create(ctypes.byref(d_), ctypes.sizeof(d_), ctypes.byref(handle))
        assert rc in (_capi.E_INVALID, _capi.E_WORKSPACE) and not handle.value, bad
        assert word in L.nlsh_last_error(), (bad, L.nlsh_last_error())
    # graph slots: the slot's stream must be a real one
    d_ = desc()
    assert L.nlsh_step_create_graph(ctypes.byref(d_), ctypes.sizeof(d_), None, ctypes.byref(handle)) == _capi.E_INVALID


def test_query_batch_argument_errors_are_reported_before_any_launch():
    """`nlsh_query_batch` (ABI v4): a descriptor of another size (a binding built against another header), a missing batch pointer, an
    encoder whose input width is not the corpus dimension and a probe count beyond one scan call are refused on the host; an empty batch
    is a no-op.  None of these paths touches the device, so they are checked here."""
    import ctypes
    from nlsh_amd import _capi
    L = _capi.lib()
    dims = _capi.int_array([128, 64, 16])
    buf = (ctypes.c_char * 4096)()
    a = ctypes.addressof(buf)
    def desc(**kw):
        base = dict(n_layers=2, act=0, key_mode=0, n_probes=10, dims=ctypes.cast(dims, ctypes.c_void_p), packed=a, n_multi_rows=0,
                    corpus_sorted=a, row_stride=128, gid=a, uniq_keys=a, offsets=a, bucket_order=a, cell_of=None, cell_offsets=None, inv_norm=None,
                    d=128, n_buckets=8, n_cells=0, k=10, metric=0, algo=2, seg_rows=0, hold_done=0, Q=64, qkeys=a, nkeys=a, out_dist=a, out_idx=a,
                    out_keys=None, out_ncand=a, status=a, workspace=a, workspace_bytes=4096, max_tasks=16, front=None, plan=None, mid=None, tail=None)
        base.update(kw)
        return _capi.StepDesc(**base)
    d_ = desc()
    assert L.nlsh_query_batch(ctypes.byref(d_), ctypes.sizeof(d_) - 8, a, 128, 1, 0, 0, None, None, None) == _capi.E_INVALID
    assert b"descriptor" in L.nlsh_last_error()
    assert L.nlsh_query_batch(ctypes.byref(d_), ctypes.sizeof(d_), None, 128, 1, 0, 0, None, None, None) == _capi.E_INVALID
    d0 = desc(Q=0)
    assert L.nlsh_query_batch(ctypes.byref(d0), ctypes.sizeof(d0), None, 128, 1, 0, 0, None, None, None) == _capi.OK      # empty batch: nothing to do
    dd = desc(d=96)
    assert L.nlsh_query_batch(ctypes.byref(dd), ctypes.sizeof(dd), a, 128, 1, 0, 0, None, None, None) == _capi.E_INVALID
    dp = desc(n_probes=65)
    assert L.nlsh_query_batch(ctypes.byref(dp), ctypes.sizeof(dp), a, 128, 1, 0, 0, None, None, None) == _capi.E_UNSUPPORTED


def test_fastlists_builds_the_same_lists_as_ndarray_tolist():
    """csrc/fastlists.c: the host-side list builder of `Indexer._plain_lists` is `ndarray.tolist()` element for element (types too),
    refuses a short buffer, and is what the facade uses when it is built."""
    import numpy as np
    from nlsh_amd import indexer
    assert indexer._rows_to_lists is not None, "csrc/fastlists.c was not built (make -C csrc)"
    rng = np.random.default_rng(5)
    for shape in ((0, 10), (1, 1), (257, 10), (1000, 64)):
        a = rng.integers(-1, 2 ** 31 - 1, size=shape).astype(np.int32)
        got = indexer._rows_to_lists(a, shape[0], shape[1])
        assert got == a.tolist() and all(type(v) is int for row in got for v in row)
        assert all(type(row) is list for row in got)
        # by default every list is an ordinary, collector-tracked one (what `.tolist()` returns: nlsh/indexer.py:91) ...
        import gc
        assert gc.is_tracked(got) and all(gc.is_tracked(row) for row in got)
        # ... and only the opt-in hands the inner rows out untracked (ints only: no cycle possible unless the caller builds one later)
        opt = indexer._rows_to_lists(a, shape[0], shape[1], True)
        assert opt == got and gc.is_tracked(opt) and not any(gc.is_tracked(row) for row in opt)
    with pytest.raises(ValueError):
        indexer._rows_to_lists(np.zeros((4, 4), np.int32), 5, 4)
    idx, nc = np.arange(5120, dtype=np.int32).reshape(512, 10), np.full((512,), 12, dtype=np.int32)
    lists, counts = indexer.Indexer._plain_lists(idx, nc)
    assert lists == idx.tolist() and counts == nc.tolist()
    assert indexer.Indexer.untracked_results is False and all(gc.is_tracked(row) for row in lists)   # the facade's default: tracked
    indexer.Indexer.untracked_results = True
    try:
        assert not any(gc.is_tracked(row) for row in indexer.Indexer._plain_lists(idx, nc)[0])
    finally:
        indexer.Indexer.untracked_results = False


def test_deferred_release_never_holds_more_than_two_results():
    """`defer_result_release` keeps the newest results alive so that the older one is freed under the next call's device work; the
    paths that do not pass through `_release_held` (generic metric, hash_times > 64) used to grow the list by one result set per call."""
    from nlsh_amd.indexer import Indexer
    ix = Indexer.__new__(Indexer)
    Indexer.defer_result_release = True
    try:
        for i in range(50):
            ix._keep(([[i]], [1]))
            assert len(ix._held) <= 2
        assert ix._held[-1] == ([[49]], [1])
    finally:
        Indexer.defer_result_release = False
    ix._keep(([], []))
    assert ix._held == []


def test_design_md_stays_within_120_columns():
    """DESIGN.md is wrapped (tools/wrap_md.py re-flows it; tables and code are reported, not altered)."""
    over = [(i + 1, len(ln)) for i, ln in enumerate(open(os.path.join(ROOT, "DESIGN.md")).read().split("\n")) if len(ln) > 120]
    assert not over, over[:5]


def test_graft_entry_build_passes_on_cpu():
    """`__graft_entry__.build()` is the driver's "does it build" check: hipcc cross-compiles gfx950 without a GPU, the library's ABI version
    must be the header's (r04 bumped it to 2 and the entry point still asserted 1 until this test existed), every declared symbol resolves."""
    import __graft_entry__ as g
    g.build()
