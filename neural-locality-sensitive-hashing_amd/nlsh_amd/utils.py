"""`hash_codes` with the reference's signature (nlsh/utils.pyx:18-32), packed on the device.

The hot path never calls this (packing is fused into `encode_hash`); it exists so callers of the
reference's Cython entry point keep working.  Codes are uploaded, packed by `nlsh_pack_codes`
(MSB-first, int16 wrap unless mode="full"), and the per-row set() is formed on the host.
"""
from typing import List, Set

import numpy as np
import torch

from . import _capi


def pack_codes(codes, mode="ref_int16", device="cuda"):
    """int32 [B, n, H] (numpy or tensor) -> int32 keys [B, n] on the device."""
    t = torch.as_tensor(np.ascontiguousarray(codes) if isinstance(codes, np.ndarray) else codes)
    if t.dim() != 3:
        raise ValueError("Buffer has wrong number of dimensions (expected 3, got %d)" % t.dim())
    t = t.to(device=device, dtype=torch.int32).contiguous()
    B, n, H = t.shape
    keys = torch.empty((B, n), dtype=torch.int32, device=t.device)
    stream = torch.cuda.current_stream(t.device).cuda_stream
    _capi.check(_capi.lib().nlsh_pack_codes(_capi.ptr(t), B, n, max(H, 1) if B * n == 0 else H,
                                            _capi.KEY_REF_INT16 if mode == "ref_int16" else _capi.KEY_FULL,
                                            _capi.ptr(keys), stream))
    return keys


def hash_codes(codes, mode="ref_int16") -> List[Set[int]]:
    keys = pack_codes(codes, mode).cpu().numpy()
    if mode != "ref_int16":
        keys = keys.astype(np.int64) & 0xFFFFFFFF
    return [set(row.tolist()) for row in keys]
