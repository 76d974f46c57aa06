#!/usr/bin/env python3
"""Index build (N1: encode corpus -> key sort -> CSR + grouped corpus, nlsh/indexer.py:6-24,36-38) wall time per build
and per stage, headline corpus (1M x 128, 16-bit learned hash).  The first build pays the allocator (hipMalloc of the
512 MB grouped copy); the reference rebuilds every 300 training steps (main.py:402), so the repeat figure is the one that recurs."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nlsh_amd import io, synth  # noqa: E402
from nlsh_amd import indexer as ixm  # noqa: E402
from nlsh_amd.data import SIFT  # noqa: E402

N, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 128
corpus_h, mean, std = synth.standardise(synth.sift_manifold(N, d, seed=synth.SEED_DATA))
Ws, bs = io.load_hasher_weights(os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "checkpoints", "sift1m_manifold_h16.npz"))
hashing = io.hashing_from_weights(Ws, bs, compat=True)
cg = torch.from_numpy(corpus_h).cuda()
walls = []
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix = ixm.Indexer(hashing, cg, SIFT.distance)
    torch.cuda.synchronize(); walls.append(time.perf_counter() - t0)
    del ix
# stages of a repeat build
now = time.perf_counter
sync = torch.cuda.synchronize
st = {}
sync(); t = now(); keys, _ = hashing.hash_device(cg, n=1); sync(); st["encode"] = now() - t
t = now(); perm, uniq, offs = ixm.build_csr_device(keys.view(-1)); sync(); st["sort+csr (incl. one .item())"] = now() - t
t = now(); out = torch.empty((N, d), dtype=torch.float32, device="cuda"); gid = torch.empty((N,), dtype=torch.int32, device="cuda"); sync(); st["alloc grouped copy"] = now() - t
from nlsh_amd import _capi  # noqa: E402
L = _capi.lib()
t = now(); _capi.check(L.nlsh_gather_rows(_capi.ptr(cg), cg.stride(0), d, _capi.ptr(perm), N, _capi.ptr(out), d, None, _capi.ptr(gid), 0, ixm._stream(cg.device))); sync(); st["gather rows"] = now() - t
t = now(); uh = uniq.cpu().numpy(); oh = offs.cpu().numpy(); st["directory to host"] = now() - t
print(json.dumps({"N": N, "build_wall_ms": [round(1e3 * w, 3) for w in walls], "stages_ms": {k: round(1e3 * v, 3) for k, v in st.items()}}))
