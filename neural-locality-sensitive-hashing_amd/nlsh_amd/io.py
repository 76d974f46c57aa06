"""On-disk formats either side of the hot path.

* TEXMEX `.fvecs` / `.bvecs` / `.ivecs` (SIFT1M, Deep1B, BIGANN are distributed this way);
* ann-benchmarks HDF5 (`train` / `test` / `neighbors` / `distances`, plus the reference's
  `train_knn` from precompute.py:91-97) -- what nlsh/data.py:17-46,114-138 reads; needs `h5py`, which
  this image does not ship, so it is imported lazily and its absence is an explicit error;
* hasher checkpoints: our `.npz` (`training.export_weights`), a plain state dict, or the reference's
  TorchScript `<base>_cpu.pt` (nlsh/hashings.py:53-57; eval.py:113 loads it with torch.jit.load).
  Parameter names follow the reference modules: `_encoder.{i}_linear.*` (MultiLayerRelu),
  `_encoder.fc1|fc2.*` (TwoLayer256Relu), `output_layer.*`.
"""
import re
from typing import Dict, List, Optional, Tuple

import numpy as np


# ----------------------------------------------------------------------------- TEXMEX vectors
def _read_vecs(path, dtype, max_rows=None):
    """Every record is `int32 dim` followed by `dim` components of `dtype`."""
    raw = np.memmap(path, dtype=np.uint8, mode="r")
    if raw.size == 0:
        return np.zeros((0, 0), dtype=dtype)
    dim = int(raw[:4].view(np.int32)[0])
    item = np.dtype(dtype).itemsize
    rec = 4 + dim * item
    if raw.size % rec:
        raise ValueError(f"{path}: size {raw.size} is not a multiple of the record size {rec} (dim={dim})")
    n = raw.size // rec
    if max_rows is not None:
        n = min(n, int(max_rows))
    body = raw[: n * rec].reshape(n, rec)
    if not np.all(body[:, :4].view(np.int32)[:, 0] == dim):
        raise ValueError(f"{path}: records with differing dimension")
    return np.ascontiguousarray(body[:, 4:]).view(dtype).reshape(n, dim)


def read_fvecs(path, max_rows=None) -> np.ndarray:
    return _read_vecs(path, np.float32, max_rows)


def read_ivecs(path, max_rows=None) -> np.ndarray:
    return _read_vecs(path, np.int32, max_rows)


def read_bvecs(path, max_rows=None) -> np.ndarray:
    """uint8 components (BIGANN / SIFT1B); returned as float32, the dtype the hot path computes in."""
    return _read_vecs(path, np.uint8, max_rows).astype(np.float32)


def write_vecs(path, array) -> None:
    a = np.ascontiguousarray(array)
    n, dim = a.shape
    rec = np.empty((n, 4 + dim * a.dtype.itemsize), dtype=np.uint8)
    rec[:, :4] = np.full((n, 1), dim, dtype=np.int32).view(np.uint8)
    rec[:, 4:] = a.view(np.uint8).reshape(n, -1)
    rec.tofile(path)


# ----------------------------------------------------------------------------- ann-benchmarks HDF5
def load_hdf5(path, with_train_knn=False) -> Dict[str, np.ndarray]:
    try:
        import h5py
    except ImportError as e:  # pragma: no cover - h5py is absent from this image
        raise ImportError("reading ann-benchmarks HDF5 files needs h5py (not installed in this image)") from e
    with h5py.File(path, "r") as f:
        out = {"training": np.asarray(f["train"], dtype=np.float32), "testing": np.asarray(f["test"], dtype=np.float32),
               "ground_truth": np.asarray(f["neighbors"])}
        if "distances" in f:
            out["ground_truth_distances"] = np.asarray(f["distances"])
        if with_train_knn and "train_knn" in f:
            out["training_self_knn"] = np.asarray(f["train_knn"])
    return out


# ----------------------------------------------------------------------------- hasher checkpoints
_LAYER_PATTERNS = (re.compile(r"^_encoder\.(\d+)_linear\.(weight|bias)$"), re.compile(r"^_encoder\.fc(\d+)\.(weight|bias)$"))


def weights_from_state_dict(state) -> Tuple[List[np.ndarray], List[Optional[np.ndarray]]]:
    """Reference-named parameters -> ([W_l], [b_l | None]) in forward order, output layer last.
    BatchNorm entries (`_encoder.{i}_batch_norm.*`) are folded into the preceding Linear (eval mode)."""
    layers: Dict[int, Dict[str, np.ndarray]] = {}
    bn: Dict[int, Dict[str, np.ndarray]] = {}
    out_layer: Dict[str, np.ndarray] = {}
    for name, t in state.items():
        a = np.asarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t)
        m = next((p.match(name) for p in _LAYER_PATTERNS if p.match(name)), None)
        if m:
            layers.setdefault(int(m.group(1)), {})[m.group(2)] = a
        elif name.startswith("output_layer."):
            out_layer[name.split(".", 1)[1]] = a
        else:
            mb = re.match(r"^_encoder\.(\d+)_batch_norm\.(weight|bias|running_mean|running_var)$", name)
            if mb:
                bn.setdefault(int(mb.group(1)), {})[mb.group(2)] = a
    if not layers or "weight" not in out_layer:
        raise ValueError("not a reference hasher state dict (expected _encoder.*_linear|fc* and output_layer.*)")
    Ws, bs = [], []
    for i in sorted(layers):
        w, b = layers[i]["weight"].astype(np.float32), layers[i].get("bias")
        if i in bn:
            s = bn[i]["weight"] / np.sqrt(bn[i]["running_var"] + 1e-5)
            w = w * s[:, None]
            b = ((b if b is not None else 0.0) - bn[i]["running_mean"]) * s + bn[i]["bias"]
        Ws.append(np.ascontiguousarray(w, dtype=np.float32))
        bs.append(None if b is None else np.ascontiguousarray(b, dtype=np.float32))
    Ws.append(np.ascontiguousarray(out_layer["weight"], dtype=np.float32))
    bs.append(None if "bias" not in out_layer else np.ascontiguousarray(out_layer["bias"], dtype=np.float32))
    return Ws, bs


def load_hasher_weights(path) -> Tuple[List[np.ndarray], List[Optional[np.ndarray]]]:
    """`.npz` (ours), TorchScript module (reference `_cpu.pt`) or pickled state dict."""
    if str(path).endswith(".npz"):
        arrs = np.load(path)
        n = len([k for k in arrs.files if re.match(r"^W\d+$", k)])
        return [arrs[f"W{i}"] for i in range(n)], [arrs[f"b{i}"] if f"b{i}" in arrs.files else None for i in range(n)]
    import torch
    try:
        module = torch.jit.load(str(path), map_location="cpu")
        state = {k: v for k, v in module.named_parameters()}
        state.update({k: v for k, v in module.named_buffers()})
    except RuntimeError:
        state = torch.load(str(path), map_location="cpu")
    return weights_from_state_dict(state)


def hashing_from_weights(Ws, bs, tanh_output=False, compat=None, seed=0):
    """Device-resident `MultivariateBernoulli` (MultiLayerRelu encoder) carrying the given weights."""
    import torch
    from .encoders import MultiLayerRelu
    from .hashings import MultivariateBernoulli
    dims = [Ws[0].shape[1]] + [w.shape[0] for w in Ws]
    H = dims[-1]
    hashing = MultivariateBernoulli(MultiLayerRelu(dims[0], dims[1:-1], with_bias=bs[0] is not None), H, None,
                                    tanh_output=tanh_output, compat=(H <= 16) if compat is None else compat, seed=seed)
    linears = [m for m in hashing._hasher.modules() if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        for lin, W, b in zip(linears, Ws, bs):
            lin.weight.copy_(torch.as_tensor(W))
            if lin.bias is not None and b is not None:
                lin.bias.copy_(torch.as_tensor(b))
    hashing.train_mode(False)
    return hashing
