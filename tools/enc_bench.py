#!/usr/bin/env python3
"""encode_hash timing: the 1M-row index-build launch (MFMA utilisation) and the 10k-query multi-probe launch (latency)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nlsh_amd import io, synth  # noqa: E402

d, H = 128, 16
Ws, bs = io.load_hasher_weights(os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "checkpoints", "sift1m_manifold_h16.npz"))
hashing = io.hashing_from_weights(Ws, bs, compat=True)
x = torch.randn((1_000_000, d), device="cuda")
q = x[:10_000].contiguous()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


big = timed(lambda: hashing.hash_device(x, n=1), 10)
small = timed(lambda: hashing.hash_device(q, n=10, n_multi_rows=8192, seed=3), 200)
by_rows = {}
for rows in (4096, 8192, 8193, 9000, 10_000, 11_000, 12_288, 12_289, 16_384):   # the query-batch forms at and around their switch-overs
    qr = x[:rows].contiguous()
    by_rows[rows] = 1e3 * timed(lambda: hashing.hash_device(qr, n=10, n_multi_rows=(rows // 4096) * 4096, seed=3), 200)
flops = 2.0 * (d * 256 + 256 * 256 + 256 * H)
print(json.dumps({"tag": sys.argv[1] if len(sys.argv) > 1 else "", "rows_1M_ms": big, "mfma_util_1M": flops * 1e6 / (big * 1e-3) / 157.3e12,
                  "queries_10k_us": small * 1e3, "us_by_rows": by_rows}))
