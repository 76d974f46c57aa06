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
