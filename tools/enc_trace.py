#!/usr/bin/env python3
"""Phase timeline of one encode_hash workgroup (diagnostic).  Needs the library built with
`make -C neural-locality-sensitive-hashing_amd/csrc EXTRA=-DNLSH_ENC_TRACE` (never the shipped build: it
overwrites the first floats of z_out).  Prints the mean wall_clock64 (100 MHz) deltas per phase over all workgroups."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nlsh_amd import io, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
Ws, bs = io.load_hasher_weights(os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "checkpoints", "sift1m_manifold_h16.npz"))
hashing = io.hashing_from_weights(Ws, bs, compat=True)
x = torch.from_numpy(synth.sift_manifold(n, 128, seed=1)).cuda()
H = 16
RPW = int(sys.argv[3]) if len(sys.argv) > 3 else 64   # rows per workgroup of the form the launcher picks for n (128: index builds)
names = ["stage", "layer1", "layer2", "(l3)", "(l4)", "(l5)", "layer_out", "sigmoid", "probes", "dedup+store"]
for rep in range(3):
    z = torch.zeros((n, H), dtype=torch.float32, device="cuda")
    hashing._run(x, int(sys.argv[2]) if len(sys.argv) > 2 else 10, z_out=z, seed=rep)
    torch.cuda.synchronize()
    st = z.cpu().numpy().reshape(-1)[: (n // RPW) * RPW * H].reshape(n // RPW, RPW * H)[:, :12]
    core = st[:, 11]; st = st[:, :11]
    d = np.diff(st, axis=1)
    keep = [0, 1, 2, 6, 7, 8, 9]
    # stamps 4..6 are unset for a 3-layer encoder: layer_out = stamp7 - stamp3
    d[:, 6] = st[:, 7] - st[:, 3]
    if os.environ.get("ENC_FINE"): print("fine (layer 2): kloop", (st[:, 4] - st[:, 2]).mean(), "barrier", (st[:, 5] - st[:, 4]).mean(), "writeback", (st[:, 6] - st[:, 5]).mean(), "barrier2", (st[:, 3] - st[:, 6]).mean())
    print(f"rep {rep}: total {st[:, 10].mean():.0f} ticks (100 MHz => {st[:, 10].mean() / 100:.1f} us)  " +
          "  ".join(f"{names[i]} {d[:, i].mean():.0f}" for i in keep) + f"  clock {np.median(core / st[:, 10]) * 0.1:.3f} GHz")
