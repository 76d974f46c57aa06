#!/usr/bin/env python3
"""Does a CU-masked HIP stream (hipExtStreamCreateWithCUMask) confine a kernel on this stack?  Times the headline scan on streams
whose mask enables a share of the 256 CUs and prints ms per launch.  r04 result (profiles/r04_cu_mask_probe.txt): honoured -- all
256 CUs 0.279 ms, 128 CUs 0.436-0.443, 32 CUs 1.68, 232 CUs 0.297 -- but a pipeline whose front and tail stages ran on 16-64 CUs
of their own (masked streams) with the scan on the rest took 0.45-1.0 ms per step against 0.31 unmasked: the masked queues did not
overlap each other.  Experiment only; nothing in the package uses it.

    python tools/cu_mask_probe.py
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nlsh_amd import io, synth  # noqa: E402
from nlsh_amd.data import SIFT  # noqa: E402
from nlsh_amd.indexer import Indexer  # noqa: E402


def masked_stream(device, cu_lo, cu_hi, _keep=[]):
    """A HIP stream confined to compute units [cu_lo, cu_hi) of the device, wrapped as a torch stream."""
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
    n_cu = torch.cuda.get_device_properties(device).multi_processor_count
    words = (n_cu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for cu in range(max(0, cu_lo), min(n_cu, cu_hi)):
        mask[cu // 32] |= 1 << (cu % 32)
    handle = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), words, mask)
    assert rc == 0 and handle.value, rc
    _keep.append(handle)
    return torch.cuda.ExternalStream(handle.value, device=device)


N, d, Q = 1_000_000, 128, 10_000
corpus_h, mean, std = synth.standardise(synth.sift_manifold(N, d, seed=synth.SEED_DATA))
queries_h, _, _ = synth.standardise(synth.sift_manifold(Q, d, seed=synth.SEED_QUERY), mean, std)
Ws, bs = io.load_hasher_weights(os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "checkpoints", "sift1m_manifold_h16.npz"))
ix = Indexer(io.hashing_from_weights(Ws, bs, compat=True), torch.from_numpy(corpus_h).cuda(), SIFT.distance, algo="tiled")
q = torch.from_numpy(queries_h).cuda()
keys, nkeys = ix.hash_device(q, hash_times=10, seed=7)
ref = ix.scan_tensors(q, keys, nkeys, k=10)
torch.cuda.synchronize()
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
out = {"multi_processor_count": n_cu}
for lo, hi in ((0, n_cu), (0, n_cu // 2), (n_cu // 2, n_cu), (0, n_cu // 8), (24, n_cu)):
    st = masked_stream(q.device, lo, hi)
    with torch.cuda.stream(st):
        ix.scan_tensors(q, keys, nkeys, k=10)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in ev:
            a.record(); b.record()
        for i in range(20):
            got = ix.scan_tensors(q, keys, nkeys, k=10, check=False, events=ev[i])
        st.synchronize()
    assert torch.equal(got[1], ref[1]) and torch.equal(got[0], ref[0])
    out[f"cus[{lo}:{hi})"] = round(float(np.mean([a.elapsed_time(b) for a, b in ev])), 4)
print(json.dumps(out))
