"""On-disk formats (N4): TEXMEX vecs round trips and hasher checkpoint loaders (CPU only)."""
import numpy as np
import pytest
import torch

from nlsh_amd import io as nio
from nlsh_amd import synth
from nlsh_amd.encoders import MultiLayerRelu, TwoLayer256Relu
from nlsh_amd.hashings import _Hasher


def test_fvecs_ivecs_bvecs_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    f = rng.standard_normal((37, 96)).astype(np.float32)
    i = rng.integers(0, 1 << 30, size=(11, 100)).astype(np.int32)
    b = rng.integers(0, 256, size=(5, 128)).astype(np.uint8)
    nio.write_vecs(tmp_path / "a.fvecs", f)
    nio.write_vecs(tmp_path / "a.ivecs", i)
    nio.write_vecs(tmp_path / "a.bvecs", b)
    assert np.array_equal(nio.read_fvecs(tmp_path / "a.fvecs"), f)
    assert np.array_equal(nio.read_fvecs(tmp_path / "a.fvecs", max_rows=10), f[:10])
    assert np.array_equal(nio.read_ivecs(tmp_path / "a.ivecs"), i)
    assert np.array_equal(nio.read_bvecs(tmp_path / "a.bvecs"), b.astype(np.float32))
    (tmp_path / "bad.fvecs").write_bytes(b"\x03\x00\x00\x00" + b"\x00" * 10)
    with pytest.raises(ValueError):
        nio.read_fvecs(tmp_path / "bad.fvecs")


def test_checkpoint_loaders_npz_statedict_torchscript(tmp_path):
    Ws, bs = synth.make_weights([100, 256, 256, 12], seed=3)
    hasher = _Hasher(TwoLayer256Relu(100), 12)                      # same child names as the reference module
    lin = [m for m in hasher.modules() if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        for m, W, b in zip(lin, Ws, bs):
            m.weight.copy_(torch.from_numpy(W)); m.bias.copy_(torch.from_numpy(b))
    # the reference's save format: TorchScript (works for TwoLayer256Relu under torch 2.x)
    torch.jit.save(torch.jit.script(hasher), str(tmp_path / "m_cpu.pt"))
    torch.save(hasher.state_dict(), str(tmp_path / "m_state.pt"))
    np.savez(tmp_path / "m.npz", **{f"W{i}": w for i, w in enumerate(Ws)}, **{f"b{i}": b for i, b in enumerate(bs)})
    for name in ("m_cpu.pt", "m_state.pt", "m.npz"):
        W2, b2 = nio.load_hasher_weights(tmp_path / name)
        assert len(W2) == 3 and all(np.array_equal(a, b) for a, b in zip(W2, Ws)) and all(np.array_equal(a, b) for a, b in zip(b2, bs))
    # MultiLayerRelu naming + BatchNorm folding
    enc = MultiLayerRelu(20, [16, 8], with_batchnorm=True)
    h = _Hasher(enc, 4).eval()
    with torch.no_grad():
        for mod in h.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.uniform_(-1, 1); mod.running_var.uniform_(0.5, 2); mod.weight.uniform_(0.5, 2); mod.bias.uniform_(-1, 1)
    W3, b3 = nio.weights_from_state_dict(h.state_dict())
    x = torch.randn(9, 20)
    ref = h.output_layer(h._encoder(x))
    y = x
    for i, (W, b) in enumerate(zip(W3, b3)):
        y = y @ torch.from_numpy(W).T + torch.from_numpy(b)
        if i + 1 < len(W3):
            y = torch.relu(y)
    assert torch.allclose(y, ref, atol=1e-5)
    with pytest.raises(ValueError):
        nio.weights_from_state_dict({"foo.weight": np.zeros((2, 2))})


def test_dataset_classes_read_the_path_they_are_given(tmp_path):
    """nlsh/data.py:14-46,112-140: `SIFT(path, unit_norm)` / `Glove(path, unit_norm, unit_ball)` load THAT file (TEXMEX
    directory here; ann-benchmarks HDF5 needs h5py) and never substitute generated data."""
    import pytest
    from nlsh_amd import data, io
    rng = np.random.default_rng(0)
    base = rng.integers(0, 200, size=(300, 16)).astype(np.float32)
    query = rng.integers(0, 200, size=(20, 16)).astype(np.float32)
    gt = rng.integers(0, 300, size=(20, 10)).astype(np.int32)
    d = tmp_path / "sift"
    d.mkdir()
    io.write_vecs(str(d / "sift_base.fvecs"), base)
    io.write_vecs(str(d / "sift_query.fvecs"), query)
    io.write_vecs(str(d / "sift_groundtruth.ivecs"), gt)
    ds = data.SIFT(str(d), unit_norm=True)
    with pytest.raises(ValueError):
        ds.training                                   # not prepared (nlsh/data.py:142-144)
    ds.load()
    mean, std = base.mean(0), base.std(0)
    assert ds.prepared and ds.dim == 16
    assert np.allclose(ds.training, (base - mean) / std, atol=1e-6) and np.allclose(ds.testing, (query - mean) / std, atol=1e-6)
    assert np.array_equal(ds.ground_truth, gt)
    with pytest.raises(AttributeError):
        ds.training_self_knn                          # no train_knn in the dataset
    io.write_vecs(str(d / "sift_train_knn.ivecs"), gt[:, :5])
    g = data.Glove(str(d), unit_ball=True)
    g.load()
    assert np.allclose(np.linalg.norm(g.training, axis=1), 1.0, atol=1e-5) and g.training_self_knn.shape == (20, 5)
    assert data.metric_of(data.SIFT.distance) == "l2" and data.metric_of(g.distance) == "cosine"
    with pytest.raises(ValueError):
        data.SIFT(None)
    with pytest.raises(FileNotFoundError):
        data.SIFT(str(tmp_path / "missing.hdf5")).load()
    (tmp_path / "fake.hdf5").write_bytes(b"not hdf5")
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError):
            data.SIFT(str(tmp_path / "fake.hdf5")).load()
    syn = data.SyntheticSIFT(n_train=500, n_test=10, dim=32, k=5, unit_norm=True, with_train_knn=True)
    syn.load()
    assert syn.training.shape == (500, 32) and syn.ground_truth.shape == (10, 5) and syn.training_self_knn.shape == (500, 5)
