#!/usr/bin/env python3
"""Phase timeline of the query batch's encode (+ fused bucket lookup) launch INSIDE the device step (diagnostic, r06).

Needs the library built with `make -C neural-locality-sensitive-hashing_amd/csrc VARIANT=enctrace EXTRA=-DNLSH_ENC_TRACE` and
NLSH_HIP_LIB pointing at it.  Runs the headline batch through `Indexer.query_tensors` (one `nlsh_query_batch` call: the encode
launch does the lookup in its epilogue) and prints, per workgroup kind (rows per workgroup), the mean 100 MHz-tick deltas per phase,
when the workgroups start and end on the launch's clock, and the launch's length = last end - first start."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nlsh_amd import _capi, io, synth  # noqa: E402
from nlsh_amd.data import SIFT  # noqa: E402
from nlsh_amd.indexer import Indexer  # noqa: E402

N, d, Q = 1_000_000, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
corpus_h, mean, std = synth.standardise(synth.sift_manifold(N, d, seed=synth.SEED_DATA))
Ws, bs = io.load_hasher_weights(os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "checkpoints", "sift1m_manifold_h16.npz"))
ix = Indexer(io.hashing_from_weights(Ws, bs, compat=True), torch.from_numpy(corpus_h).cuda(), SIFT.distance)
q = torch.from_numpy(synth.standardise(synth.sift_manifold(Q, d, seed=synth.SEED_QUERY), mean, std)[0]).cuda()
ix.query_tensors(q, k=10, hash_times=10, seed=1)
torch.cuda.synchronize()
L = _capi.lib()
names = ["stage", "layer1", "layer2", "-", "-", "-", "layer_out", "sigmoid", "keys", "dedup", "store+lookup"]
for rep in range(3):
    ix.query_tensors(q, k=10, hash_times=10, seed=2 + rep, check=False)
    torch.cuda.synchronize()
    buf = np.zeros((4096 * 16,), dtype=np.float32)
    L.nlsh_debug_enc_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    t = buf.reshape(4096, 16)
    t = t[(t[:, 13] > 0) & (t[:, 13] < 128)]          # the query batch's workgroups (slots beyond them still hold the index build's)
    t0 = t[:, 12].min()
    print(f"rep {rep}: {len(t)} workgroups, launch {(t[:, 12] + t[:, 11] - t0).max() / 100:.1f} us (first start -> last end)")
    for rows in sorted(set(t[:, 13].tolist())):
        w = t[t[:, 13] == rows]
        st = w[:, :12].copy()
        st[:, 3:7] = np.where(st[:, 3:7] == 0, st[:, [2]], st[:, 3:7])     # stamps a 3-layer encoder never sets
        dl = np.diff(st, axis=1)
        dl[:, 6] = st[:, 7] - st[:, 3]
        keep = [0, 1, 2, 6, 7, 8, 9, 10]
        print(f"  {int(rows):3d}-row workgroups x {len(w)}: total {w[:, 11].mean() / 100:.1f} us (max {w[:, 11].max() / 100:.1f}), start {((w[:, 12] - t0).mean()) / 100:.1f} us (max {((w[:, 12] - t0).max()) / 100:.1f}), "
              f"end max {((w[:, 12] + w[:, 11] - t0).max()) / 100:.1f};  " + "  ".join(f"{names[i]} {dl[:, i].mean() / 100:.2f}" for i in keep))
