"""Learnable hash function with the reference's `MultivariateBernoulli` surface
(nlsh/hashings.py:11-92), backed by the fused gfx950 `encode_hash` kernel.

`hash()` / `predict()` (eval mode) run ONE HIP launch: MLP forward on fp32 MFMA, sigmoid/tanh,
`> 0.5` bits, MSB-first packing, Philox multi-probe sampling and per-row de-duplication, with no
device->host copy of codes.  `hash_device()` is the device-resident form `Indexer` uses.
There is no CPU fallback: a missing library or a non-device tensor raises.
"""
import itertools
from typing import List, Set

import torch
import torch.nn as nn

from . import _capi


class _Hasher(nn.Module):
    """encoder -> Linear(hash_size) -> sigmoid | tanh; same child names as the reference module
    (hashings.py:13-27: `_encoder`, `output_layer`) so state dicts interchange."""

    def __init__(self, encoder, hash_size, tanh_output=False):
        super().__init__()
        self._encoder = encoder
        self._tanh_output = tanh_output
        self.output_layer = nn.Linear(encoder.output_dim, hash_size)

    def forward(self, x):  # stock autograd path: training only (out of the hot path)
        z = self.output_layer(self._encoder(x))
        return torch.tanh(z) if self._tanh_output else torch.sigmoid(z)


class MultivariateBernoulli:
    _Hasher = _Hasher

    def __init__(self, encoder, hash_size, distance_func, tanh_output=False, compat=True, seed=0):
        if not 1 <= hash_size <= _capi.MAX_HASH_BITS:
            raise ValueError(f"hash_size must be in [1, {_capi.MAX_HASH_BITS}], got {hash_size}")
        self._encoder = encoder
        self._hash_size = hash_size
        self._distance_func = distance_func  # code-space distance: training only (hashings.py:33-34)
        self._tanh_output = tanh_output
        # compat=True: bucket keys wrap to int16 exactly like nlsh/utils.pyx:7-15
        self.key_mode = _capi.KEY_REF_INT16 if compat else _capi.KEY_FULL
        self._hasher = _Hasher(encoder, hash_size, tanh_output).cuda()
        self._seed = int(seed)
        self._calls = itertools.count()
        self._packed = None
        self._packed_sig = None

    # ------------------------------------------------------------------ reference surface
    @property
    def distance(self):
        return self._distance_func

    @property
    def output_dim(self):
        return self._hash_size

    def parameters(self):
        return self._hasher.parameters()

    def train_mode(self, on):
        self._hasher.train(bool(on))

    def save(self, base_name):
        """TorchScript `_cpu.pt` / `_gpu.pt` like hashings.py:53-57, plus a plain state dict
        (`_state.pt`), which `load_state` reads back (the reference has no loader: :58)."""
        torch.save(self._hasher.state_dict(), base_name + "_state.pt")
        for suffix, module in (("_cpu.pt", self._hasher.cpu()), ("_gpu.pt", self._hasher.cuda())):
            torch.jit.save(torch.jit.script(module), base_name + suffix)

    def load_state(self, path):
        self._hasher.load_state_dict(torch.load(path, map_location="cpu"))
        self._hasher.cuda()
        self._packed_sig = None

    def predict(self, x):
        """Module output [B, H].  Eval mode: HIP kernel.  Train mode: autograd forward (the
        training losses need gradients; training is outside the query-time hot path)."""
        if self._hasher.training:
            return self._hasher(x)
        return self._run(x, 1, want_probs=True)[2]

    def hash(self, query_vectors, n=1) -> List[Set[int]]:
        """hashings.py:66-92: list of B sets of bucket keys (1 hard + n-1 sampled probes).
        Like the reference, the forward runs in whatever mode the module is in (`train_mode`): see `_run_train_mode`."""
        if n < 1:
            raise ValueError(f"`n` should be positive integer, but got {n}")
        keys, nkeys, _ = self._run(query_vectors, n)
        return keys_to_sets(keys, nkeys, self.key_mode)

    # ------------------------------------------------------------------ device-resident form
    def hash_device(self, x, n=1, n_multi_rows=None, seed=None, row0=0, out=None):
        """-> (keys int32 [B, n] distinct, first-occurrence order; nkeys int32 [B]) on the device.

        Rows >= n_multi_rows are single-probe (Indexer.hash's trailing-batch rule).  `seed`
        defaults to a per-call stream (base seed + call counter): identical on every rank.
        """
        if n < 1:
            raise ValueError(f"`n` should be positive integer, but got {n}")
        keys, nkeys, _ = self._run(x, n, n_multi_rows=n_multi_rows, seed=seed, row0=row0, out=out)
        return keys, nkeys

    def forward_device(self, x):
        """-> (z [B,H], probs [B,H], code uint32-as-int32 [B]) in one launch (tests / eval flow)."""
        B = x.shape[0]
        z = torch.empty((B, self._hash_size), dtype=torch.float32, device=x.device)
        code = torch.empty((B,), dtype=torch.int32, device=x.device)
        _, _, probs = self._run(x, 1, want_probs=True, z_out=z, code_out=code)
        return z, probs, code

    # ------------------------------------------------------------------ internals
    def linear_stack(self):
        stack = list(self._encoder.linear_stack())
        ol = self._hasher.output_layer
        stack.append((ol.weight.detach(), None if ol.bias is None else ol.bias.detach()))
        return stack

    def dims(self):
        stack = self.linear_stack()
        return [stack[0][0].shape[1]] + [w.shape[0] for w, _ in stack]

    def _weights_signature(self):
        """(name, storage address, version counter) of every parameter and buffer of the hasher: changes whenever an optimiser step, a
        `load_state_dict`, a `.cuda()`, a reassigned or a newly registered parameter / buffer changes what the packed blob was built
        from.  Called on every hashing call, so it walks a cached list of the modules' own `_parameters` / `_buffers` dicts (whatever
        they hold NOW: `register_buffer`, `parametrize` or an assignment on an existing module is seen through the dict) instead of
        `module.parameters()` -- the recursive module walk cost 30 us per call, a third of the host time of a pipelined batch (r04,
        cProfile of `QueryPipeline.submit`).  The list itself is rebuilt when the module tree changed shape (a submodule added,
        removed or swapped: the count and identities of the `_modules` entries are part of the cache key)."""
        cached = self.__dict__.get("_sig_slots")
        if cached is not None and cached[0] is self._hasher:
            for md, n, ids in cached[2]:                          # every module's child table: same length, same child objects
                if len(md) != n or tuple(map(id, md.values())) != ids:
                    cached = None
                    break
        if cached is None or cached[0] is not self._hasher:           # a replaced `_hasher` module gets its own walk
            mods = list(self._hasher.modules())
            cached = self._sig_slots = (self._hasher, [d for m in mods for d in (m._parameters, m._buffers)],
                                        [(m._modules, len(m._modules), tuple(map(id, m._modules.values()))) for m in mods])
        sig = []
        for d in cached[1]:
            for name, t in d.items():
                sig.append((name, None) if t is None else (name, t.data_ptr(), t._version))
        return tuple(sig)

    def packed_weights(self):
        """MFMA-fragment-ordered weight blob on the device; repacked when a parameter changes."""
        sig = self._weights_signature()
        if self._packed is not None and sig == self._packed_sig:
            return self._packed
        L = _capi.lib()
        stack = [(w.float().contiguous(), None if b is None else b.float().contiguous()) for w, b in self.linear_stack()]
        dev = stack[0][0].device
        if dev.type != "cuda":
            raise _capi.NlshHipError(_capi.E_INVALID, "hasher weights are not on the GPU; there is no CPU path")
        dims = self.dims()
        n_floats = L.nlsh_encoder_packed_floats(len(stack), _capi.int_array(dims))
        if n_floats < 0:
            _capi.check(_capi.E_UNSUPPORTED)
        packed = torch.empty((n_floats,), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _capi.check(L.nlsh_encoder_pack(len(stack), _capi.int_array(dims), _capi.ptr_array([w for w, _ in stack]),
                                        _capi.ptr_array([b for _, b in stack]), _capi.ptr(packed), stream))
        self._packed, self._packed_sig, self._keep = packed, sig, stack
        return packed

    def encode_args(self, n, keys, nkeys):
        """Fixed part of an `nlsh_encode_hash` call that fills a caller-owned key table (weights as they are NOW):
        (n_layers, dims array, packed weights ptr, act, key_mode, n_probes) and the output pointers, as plain values
        for callers that launch many batches (nlsh_amd/pipeline.py)."""
        dims = self.dims()
        self._dims_arr = _capi.int_array(dims)
        packed = self.packed_weights()
        return ((len(dims) - 1, self._dims_arr, packed.data_ptr(), _capi.ACT_TANH if self._tanh_output else _capi.ACT_SIGMOID,
                 self.key_mode, n), (None, None, None, keys.data_ptr(), nkeys.data_ptr()))

    def next_seed(self):
        return (self._seed + 0x9E3779B97F4A7C15 * (next(self._calls) + 1)) & 0xFFFFFFFFFFFFFFFF

    def _needs_train_forward(self):
        """Train mode changes the forward only for encoders with BatchNorm (batch statistics instead of the running ones the
        fused kernel folds into its weights); the plain Linear+ReLU stacks compute the same function in both modes."""
        return self._hasher.training and any(isinstance(m, nn.modules.batchnorm._BatchNorm) for m in self._hasher.modules())

    def _run_train_mode(self, x, n, n_multi_rows, want_probs, out):
        """`hash()` / `hash_device()` while the module is in TRAIN mode and has BatchNorm layers: the reference's `hash` runs
        `self._hasher(query_vectors)` in the module's current mode (nlsh/hashings.py:66-67; nlsh/trainers/proposed.py:101-104
        calls it between optimiser steps, in train mode), i.e. with BATCH statistics and a running-statistics update.  That
        forward cannot be folded into the fused kernel's weights, and training is outside the query-time hot path, so this
        branch mirrors the reference op for op on the device: the module's own forward, `> 0.5`, `torch.bernoulli` draws (the
        global torch RNG, as `Bernoulli.sample` uses), then `nlsh_pack_codes` and a first-occurrence de-duplication per row."""
        from .utils import pack_codes
        with torch.no_grad():
            probs = self._hasher(x)
        p01 = probs / 2. + 0.5 if self._tanh_output else probs
        base = (p01 > 0.5).int().unsqueeze(1)
        if n > 1:
            sampled = torch.bernoulli(p01.unsqueeze(0).expand(n - 1, -1, -1)).int().permute(1, 0, 2)
            codes = torch.cat((base, sampled), dim=1)
        else:
            codes = base
        B = x.shape[0]
        every = pack_codes(codes.contiguous(), "ref_int16" if self.key_mode == _capi.KEY_REF_INT16 else "full", device=x.device)
        earlier = torch.tril(torch.ones((n, n), dtype=torch.bool, device=x.device), diagonal=-1)
        dup = ((every[:, :, None] == every[:, None, :]) & earlier[None]).any(2)
        if n_multi_rows is not None and n > 1:      # Indexer.hash's trailing-batch rule: rows >= n_multi_rows are single-probe
            dup[int(n_multi_rows):, 1:] = True
        order = torch.argsort(dup.to(torch.int8), dim=1, stable=True)
        keys_t, nkeys_t = torch.gather(every, 1, order).contiguous(), (~dup).sum(1).to(torch.int32)
        if out is not None:
            out[0].copy_(keys_t)
            out[1].copy_(nkeys_t)
            keys_t, nkeys_t = out
        return keys_t, nkeys_t, (probs if want_probs else None)

    def _run(self, x, n, n_multi_rows=None, seed=None, row0=0, want_probs=False, z_out=None, code_out=None, out=None):
        if x.device.type != "cuda":
            raise _capi.NlshHipError(_capi.E_INVALID, "encode_hash needs a device tensor; there is no CPU path")
        if self._needs_train_forward():
            if z_out is not None or code_out is not None:
                raise _capi.NlshHipError(_capi.E_UNSUPPORTED, "forward_device() needs eval mode for BatchNorm encoders (train_mode(False))")
            if x.dtype != torch.float32:
                x = x.float()
            return self._run_train_mode(x.detach(), n, n_multi_rows, want_probs, out)
        if n > _capi.MAX_ENCODE_PROBES:
            raise _capi.NlshHipError(_capi.E_UNSUPPORTED, f"hash_times={n} > {_capi.MAX_ENCODE_PROBES}")
        L = _capi.lib()
        x = x.detach()
        if x.dtype != torch.float32:
            x = x.float()
        if x.dim() != 2 or x.stride(1) != 1:
            x = x.contiguous()
        dims = self.dims()
        if x.shape[1] != dims[0]:
            raise ValueError(f"input dim {x.shape[1]} != encoder input dim {dims[0]}")
        B = x.shape[0]
        packed = self.packed_weights()
        if out is not None:   # caller-owned key table (pipelined batches keep theirs across streams)
            keys, nkeys = out
            if keys.shape != (B, n) or keys.dtype != torch.int32 or nkeys.shape != (B,) or nkeys.dtype != torch.int32:
                raise ValueError("out must be (int32 [B, n], int32 [B])")
        else:
            keys = torch.empty((B, n), dtype=torch.int32, device=x.device)
            nkeys = torch.empty((B,), dtype=torch.int32, device=x.device)
        probs = torch.empty((B, self._hash_size), dtype=torch.float32, device=x.device) if want_probs else None
        if seed is None:
            seed = (self._seed + 0x9E3779B97F4A7C15 * (next(self._calls) + 1)) & 0xFFFFFFFFFFFFFFFF
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _capi.check(L.nlsh_encode_hash(
            _capi.ptr(x), B, x.stride(0) if B else dims[0], len(dims) - 1, _capi.int_array(dims), _capi.ptr(packed),
            _capi.ACT_TANH if self._tanh_output else _capi.ACT_SIGMOID, self.key_mode, n,
            B if n_multi_rows is None else int(n_multi_rows), seed, row0,
            _capi.ptr(z_out), _capi.ptr(probs), _capi.ptr(code_out), _capi.ptr(keys), _capi.ptr(nkeys), stream))
        return keys, nkeys, probs


def host_key_set(row, count, key_mode=_capi.KEY_REF_INT16) -> Set[int]:
    """One row of a HOST key table -> the reference's key set (same construction as `keys_to_sets`, so the same
    Python iteration order: the F7 fallback names the LAST key of that order)."""
    if key_mode == _capi.KEY_FULL:
        row = row.astype("int64") & 0xFFFFFFFF
    return set(row[:count].tolist())


def keys_to_sets(keys, nkeys, key_mode=_capi.KEY_REF_INT16) -> List[Set[int]]:
    """Device key table -> the reference's `List[Set[int]]` (one D2H copy, then host objects)."""
    kh = keys.cpu().numpy()
    nh = nkeys.cpu().numpy()
    if key_mode == _capi.KEY_FULL:
        kh = kh.astype("int64") & 0xFFFFFFFF  # eval.py:49-53 yields non-negative ints
    return [set(row[:c].tolist()) for row, c in zip(kh, nh)]
