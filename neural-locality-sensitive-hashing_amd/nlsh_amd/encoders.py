"""Encoder modules with the reference's constructor signatures and parameter names
(reference encoders.py:8-55: `TwoLayer256Relu` -> fc1/fc2, `MultiLayerRelu` ->
`{i}_linear` / `{i}_batch_norm` / `{i}_relu`), so state dicts are interchangeable.

They are ordinary `torch.nn` modules: training (out of scope, stock autograd) runs their
torch forward; the query-time hot path never does -- `MultivariateBernoulli` reads their
weights through `linear_stack()` and runs the fused gfx950 kernel instead.
The third-party SIREN encoder (encoders.py:58-79) is out of scope.
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

LinearSpec = Tuple[torch.Tensor, Optional[torch.Tensor]]


def _fold_batchnorm(weight, bias, bn: nn.BatchNorm1d) -> LinearSpec:
    """Eval-mode BatchNorm1d after a Linear is an affine map: fold it into (W, b)."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps) if bn.affine else 1.0 / torch.sqrt(bn.running_var + bn.eps)
    shift = bn.bias if bn.affine else torch.zeros_like(bn.running_mean)
    w = weight * scale[:, None]
    b0 = bias if bias is not None else torch.zeros_like(bn.running_mean)
    return w, (b0 - bn.running_mean) * scale + shift


class _ReluMlp:
    """Mixin of the Linear(+BatchNorm1d)+ReLU encoders: exposes the folded Linear stack to the HIP
    kernel.  `forward` stays TorchScript-compatible in the subclasses (`hashing.save` scripts them)."""

    def _blocks(self):
        raise NotImplementedError

    def linear_stack(self) -> List[LinearSpec]:
        """[(W [out,in], b [out] | None)] with eval-mode BatchNorm folded in: what the HIP kernel packs."""
        out = []
        for linear, bn in self._blocks():
            w, b = linear.weight, linear.bias
            if bn is not None:
                w, b = _fold_batchnorm(w, b, bn)
            out.append((w.detach(), None if b is None else b.detach()))
        return out


class TwoLayer256Relu(_ReluMlp, nn.Module):

    def __init__(self, input_dim: int, with_bias=True):
        nn.Module.__init__(self)
        self._input_dim = input_dim
        self.output_dim = 256
        self.fc1 = nn.Linear(input_dim, 256, bias=with_bias)
        self.fc2 = nn.Linear(256, 256, bias=with_bias)

    def forward(self, x):
        return torch.relu(self.fc2(torch.relu(self.fc1(x))))

    def _blocks(self):
        return [(self.fc1, None), (self.fc2, None)]


class MultiLayerRelu(_ReluMlp, nn.Sequential):
    """nn.Sequential of `{i}_linear` [, `{i}_batch_norm`], `{i}_relu` (Sequential's own forward scripts)."""

    def __init__(self, input_dim, hidden_dims: List[int], with_batchnorm=False, with_bias=True):
        nn.Sequential.__init__(self)
        self._input_dim = input_dim
        self._hidden_dims = list(hidden_dims)
        self._with_batchnorm = with_batchnorm
        self.output_dim = self._hidden_dims[-1]
        widths = [input_dim] + self._hidden_dims
        for i, (fan_in, fan_out) in enumerate(zip(widths[:-1], widths[1:])):
            self.add_module(f"{i}_linear", nn.Linear(fan_in, fan_out, bias=with_bias))
            if with_batchnorm:
                self.add_module(f"{i}_batch_norm", nn.BatchNorm1d(fan_out))
            self.add_module(f"{i}_relu", nn.ReLU())

    def _blocks(self):
        mods = dict(self.named_children())
        return [(mods[f"{i}_linear"], mods.get(f"{i}_batch_norm")) for i in range(len(self._hidden_dims))]


class Siren(nn.Module):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "SIREN encoder is out of scope: its arithmetic lives in the un-vendored, unpinned PyPI package "
            "`siren-torch` (reference encoders.py:5,58-79; SURVEY.md F15)")
